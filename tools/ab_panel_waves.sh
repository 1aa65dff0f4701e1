root=$GRAFT_REPO_ROOT
for W in w1; do echo "== parity with $W"; MPMC_ENERGY_LIB=$root/mpmcxx_amd/libmpmc_energy_$W.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config5.py tests/test_gpu_triclinic.py -x -q 2>&1 | tail -1; done
bash tools/ab_libs.sh w4=$root/mpmcxx_amd/libmpmc_energy.so w2=$root/mpmcxx_amd/libmpmc_energy_w2.so w1=$root/mpmcxx_amd/libmpmc_energy_w1.so
for rep in 1 2; do for W in "" _w2 _w1; do echo -n "rep$rep lib$W alone: "; MPMC_ENERGY_LIB=$root/mpmcxx_amd/libmpmc_energy$W.so timeout -k 10 200 python tools/alone_ab.py "x:" 2>&1 | grep "polar r[12]" | tr '\n' ' '; echo; done; done
