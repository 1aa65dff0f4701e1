bash tools/host_asan.sh run > gpurun_out/r04_host_asan.txt 2>&1; cat gpurun_out/r04_host_asan.txt | tail -5
python tools/soak.py 2>&1 | tee gpurun_out/r04_soak.txt
bash tools/fuzz_long.sh > gpurun_out/r04_fuzz_long.txt 2>&1; cat gpurun_out/r04_fuzz_long.txt | tail -30
