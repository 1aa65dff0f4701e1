#!/bin/bash
# Host-side AddressSanitizer + UBSan over the host translation units (csrc/context.cpp, evaluate.cpp, trial.cpp, pi.cpp, comm.cpp, gibbs.cpp, erfc_table.cpp: the
# pointer work above the kernels): the kernels are compiled as usual, the host files by g++ with -fsanitize=address,undefined, linked into mpmcxx_amd/libmpmc_energy_asan.so.  Device code is NOT
# instrumented (GPU ASan / xnack+ are not available on the pool).
#   here:      bash tools/host_asan.sh build
#   GPU box:   gpurun -- 'bash tools/host_asan.sh run'      -> gpurun_out/asan_tests.log, gpurun_out/*san_report* (none = clean)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
	tmp=$(mktemp -d)
	for f in kernels kernels_sym kernels_pair kernels_panel kernels_delta kernels_gs kernels_dense; do
		/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -c $root/mpmcxx_amd/csrc/$f.hip -o $tmp/$f.o &
	done
	for f in context evaluate trial pi comm gibbs erfc_table; do
		g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$root/include \
			-c $root/mpmcxx_amd/csrc/$f.cpp -o $tmp/host_$f.o &
	done
	wait
	g++ -shared -fPIC -fsanitize=address,undefined $tmp/*.o -L/opt/rocm/lib -lamdhip64 -ldl -lpthread -Wl,-rpath,/opt/rocm/lib -o $root/mpmcxx_amd/libmpmc_energy_asan.so
	echo built $root/mpmcxx_amd/libmpmc_energy_asan.so
else
	cd $root && mkdir -p gpurun_out
	export MPMC_ENERGY_LIB=$root/mpmcxx_amd/libmpmc_energy_asan.so
	export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:log_path=$root/gpurun_out/asan_report
	export UBSAN_OPTIONS=print_stacktrace=1:log_path=$root/gpurun_out/ubsan_report
	LD_PRELOAD=$(gcc -print-file-name=libasan.so) timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_trial_moves.py \
		tests/test_gpu_box_moves.py tests/test_gpu_edge_cases.py tests/test_gpu_round2_fixes.py tests/test_gpu_round3_fixes.py tests/test_gpu_pair_sweep.py tests/test_gpu_triclinic.py tests/test_gibbs.py tests/test_gpu_parity_margin.py tests/test_gpu_config5.py -q -m gpu > gpurun_out/asan_tests.log 2>&1 || true
	tail -n 3 gpurun_out/asan_tests.log
	ls gpurun_out | grep san_report || echo "no sanitizer reports"
fi
