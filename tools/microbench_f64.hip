// tools/microbench_f64.hip -- issue cost of the fp64 VALU instructions the pair / Jacobi kernels are made of, on the box itself.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/microbench_f64.hip && /tmp/mb
// Each kernel runs NCHAIN independent dependency chains of one instruction per lane (so the result reflects issue rate, not
// latency) for ITER iterations on `waves` waves per SIMD of every CU, and reports shader cycles per wave-instruction per SIMD
// (s_memtime deltas of wave 0) together with the whole-chip rate from the wall clock.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
	do {                                                                                  \
		hipError_t e_ = (x);                                                              \
		if (e_ != hipSuccess) {                                                           \
			std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));            \
			std::exit(1);                                                                 \
		}                                                                                 \
	} while (0)

constexpr int NCHAIN = 8;
constexpr int ITER = 16384;

enum Op { FMA, MUL, ADD, RNDNE, RSQ, DPP_ROL, FMA_SGPR, MIX_STEP };

template <int OP>
__global__ __launch_bounds__(256) void k_bench(double *out, long long *cyc, double seed, double sconst) {
	double v[NCHAIN];
	for (int c = 0; c < NCHAIN; ++c) v[c] = seed + 1e-3 * (threadIdx.x + c);
	const double a = 1.0000001, b = 1e-9;
	const long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < ITER; ++it) {
#pragma unroll
		for (int c = 0; c < NCHAIN; ++c) {
			if (OP == FMA) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));
			if (OP == MUL) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[c]) : "v"(a));
			if (OP == ADD) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[c]) : "v"(b));
			if (OP == RNDNE) asm volatile("v_rndne_f64 %0, %0" : "+v"(v[c]));
			if (OP == RSQ) asm volatile("v_rsq_f64 %0, %0" : "+v"(v[c]));
			if (OP == FMA_SGPR) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[c]) : "s"(sconst), "v"(b));
			if (OP == DPP_ROL) {
				int lo = __double2loint(v[c]), hi = __double2hiint(v[c]);
				asm volatile("v_mov_b32_dpp %0, %0 wave_rol:1 row_mask:0xf bank_mask:0xf" : "+v"(lo));
				asm volatile("v_mov_b32_dpp %0, %0 wave_rol:1 row_mask:0xf bank_mask:0xf" : "+v"(hi));
				v[c] = __hiloint2double(hi, lo);
			}
		}
	}
	const long long t1 = __builtin_readcyclecounter();
	double s = 0;
	for (int c = 0; c < NCHAIN; ++c) s += v[c];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char *name, int waves_per_simd, int instr_per_chain_step = 1) {
	hipDeviceProp_t p;
	CHECK(hipGetDeviceProperties(&p, 0));
	const int cus = p.multiProcessorCount;
	const int blocks = cus * waves_per_simd; // 256 threads = 4 waves = one per SIMD
	double *out;
	long long *cyc;
	CHECK(hipMalloc(&out, sizeof(double) * blocks * 256));
	CHECK(hipMalloc(&cyc, sizeof(long long) * blocks));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k_bench<OP>), dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5, 1.0000001);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	hipLaunchKernelGGL((k_bench<OP>), dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5, 1.0000001);
	CHECK(hipEventRecord(e1));
	CHECK(hipDeviceSynchronize());
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<long long> h(blocks);
	CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
	double mean = 0;
	for (long long c : h) mean += (double)c;
	mean /= blocks;
	const double n_instr = (double)ITER * NCHAIN * instr_per_chain_step; // per wave
	// s_memtime counts at a fixed 100 MHz on gfx9: convert through the wall clock instead
	const double wave_instr_total = n_instr * blocks * 4.0;
	const double per_simd_rate = wave_instr_total / (cus * 4.0) / (ms * 1e-3); // wave-instr / s / SIMD
	std::printf("%-10s waves/SIMD %d: %.3f ms  %.3e wave-instr/s/SIMD  => %.2f cycles/instr at 2.4 GHz (%.2f at 2.0 GHz); counter ticks/instr %.3f\n", name,
	            waves_per_simd, ms, per_simd_rate, 2.4e9 / per_simd_rate, 2.0e9 / per_simd_rate, mean / n_instr);
	CHECK(hipFree(out));
	CHECK(hipFree(cyc));
}

int main() {
	for (int w : {1, 2, 4}) {
		run<FMA>("fma_f64", w);
		run<MUL>("mul_f64", w);
		run<ADD>("add_f64", w);
		run<RNDNE>("rndne_f64", w);
		run<RSQ>("rsq_f64", w);
		run<FMA_SGPR>("fma_sgpr", w);
		run<DPP_ROL>("dpp_rol", w, 2);
	}
	return 0;
}
