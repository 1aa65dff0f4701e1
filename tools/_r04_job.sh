bash tools/ab_libs.sh w4=mpmcxx_amd/.abl/lib_w4.so new=mpmcxx_amd/libmpmc_energy.so 2>&1 | tee gpurun_out/r04_abl_w4.txt
for lib in mpmcxx_amd/.abl/lib_w4.so mpmcxx_amd/libmpmc_energy.so; do echo "== $lib"; MPMC_ENERGY_LIB=$lib python tools/host_step_profile.py 1 200; MPMC_ENERGY_LIB=$lib python tools/host_step_profile.py 4 50; done 2>&1 | tee -a gpurun_out/r04_abl_w4.txt
