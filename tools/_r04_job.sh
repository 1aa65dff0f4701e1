python -m pytest tests -m gpu -x -q > gpurun_out/r04_t8.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r04_t8.log
python tools/solver_time.py 2>&1 | tee gpurun_out/r04_solver_time.txt
