root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_pair_sweep.py tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_triclinic.py tests/test_gpu_parity_margin.py tests/test_gpu_config5.py -x -q > $out/r05_t3.log 2>&1; echo "pytest rc=$?"; tail -3 $out/r05_t3.log
for r in 1 2; do
echo "new build:"; PAB_ROUNDS=2 timeout -k 10 200 python tools/pair_ab.py "new:" 
echo "base build:"; MPMC_ENERGY_LIB=$root/mpmcxx_amd/libmpmc_energy_base.so PAB_ROUNDS=2 timeout -k 10 200 python tools/pair_ab.py "base:"
done
