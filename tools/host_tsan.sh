#!/bin/bash
# Host-side ThreadSanitizer over the host translation units (csrc/context.cpp, evaluate.cpp, trial.cpp, pi.cpp, comm.cpp, gibbs.cpp, erfc_table.cpp): the kernels
# are compiled as usual, the host files by g++ with -fsanitize=thread, linked into mpmcxx_amd/libmpmc_energy_tsan.so.  What it watches: the library's own
# shared state under concurrent callers (the reference calls energy() from P OpenMP threads, one per System: PathIntegral.cpp:772-779) -- the process-wide
# tuning defaults, the RCCL loader, the per-device worker threads of mpmc_pi_allreduce and their hand-off, the atexit teardown.  The HIP runtime, RCCL and
# Python are not instrumented: races reported INSIDE them are suppressed (tools/tsan.supp), races between our frames are findings.
#   here:      bash tools/host_tsan.sh build
#   GPU box:   gpurun -- 'bash tools/host_tsan.sh run'      -> gpurun_out/tsan_*.log, gpurun_out/tsan_report.* (none = clean)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
	tmp=$(mktemp -d)
	for f in kernels kernels_sym kernels_pair kernels_panel kernels_delta kernels_gs kernels_dense; do
		/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -c $root/mpmcxx_amd/csrc/$f.hip -o $tmp/$f.o &
	done
	for f in context evaluate trial pi comm gibbs erfc_table; do
		g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$root/include \
			-c $root/mpmcxx_amd/csrc/$f.cpp -o $tmp/host_$f.o &
	done
	wait
	g++ -shared -fPIC -fsanitize=thread $tmp/*.o -L/opt/rocm/lib -lamdhip64 -ldl -lpthread -Wl,-rpath,/opt/rocm/lib -o $root/mpmcxx_amd/libmpmc_energy_tsan.so
	echo built $root/mpmcxx_amd/libmpmc_energy_tsan.so
else
	cd $root && mkdir -p gpurun_out && rm -f gpurun_out/tsan_report.*
	export MPMC_ENERGY_LIB=$root/mpmcxx_amd/libmpmc_energy_tsan.so
	export TSAN_OPTIONS="suppressions=$root/tools/tsan.supp:halt_on_error=0:second_deadlock_stack=1:history_size=4:log_path=$root/gpurun_out/tsan_report:ignore_noninstrumented_modules=1"
	pre=$(gcc -print-file-name=libtsan.so)
	# (ThreadSanitizer's shadow layout does not survive this kernel's 32-bit mmap randomisation -- "unexpected memory mapping": the test programs
	# are started with address-space randomisation off; setarch execs python BEFORE anything has touched the GPU)
	noaslr="setarch $(uname -m) -R"
	echo "== 1. the threaded fuzzer: one live context per host thread through random operation sequences, 4 threads"
	timeout -k 10 600 $noaslr env LD_PRELOAD=$pre python tools/fuzz_state.py 0 24 4 > gpurun_out/tsan_fuzz_state.log 2>&1 || echo "fuzz_state rc=$?"
	tail -n 2 gpurun_out/tsan_fuzz_state.log
	echo "== 2. several host threads creating contexts and running their FIRST evaluation at once (the OpenMP bead loop)"
	timeout -k 10 300 $noaslr env LD_PRELOAD=$pre python tools/first_eval_stress.py 8 8 > gpurun_out/tsan_first_eval.log 2>&1 || echo "first_eval rc=$?"
	tail -n 2 gpurun_out/tsan_first_eval.log
	echo "== 3. mpmc_pi_allreduce with one host thread per (virtual) device: worker hand-off, ordered combine, atexit teardown"
	timeout -k 10 300 $noaslr env LD_PRELOAD=$pre python -m pytest tests/test_gpu_comm.py -q -m gpu > gpurun_out/tsan_comm.log 2>&1 || echo "comm tests rc=$?"
	tail -n 2 gpurun_out/tsan_comm.log
	echo "== 4. bench.py --launch inprocess --gpus 2 --force-device 0 (the driver-shaped step through mpmc_pi_allreduce)"
	timeout -k 10 300 $noaslr env LD_PRELOAD=$pre python bench.py --gpus 2 --launch inprocess --force-device 0 --beads 4 --natoms 1000 --steps 3 --warmup 1 --cpu-baseline none --no-extra-passes > gpurun_out/tsan_inprocess.log 2>&1 || echo "inprocess rc=$?"
	tail -c 300 gpurun_out/tsan_inprocess.log; echo
	ls gpurun_out | grep "tsan_report" || echo "no ThreadSanitizer reports"
	for f in gpurun_out/tsan_report.*; do [ -f "$f" ] && { echo "--- $f"; grep -c "WARNING: ThreadSanitizer" $f; grep -A12 "WARNING: ThreadSanitizer" $f | head -60; }; done
fi
