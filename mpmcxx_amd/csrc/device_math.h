// device_math.h -- device-only helpers shared by the HIP translation units (wave reductions, SGPR-constant Horner
// steps, exp, erfc).  Included after kernels.h; compiled with -ffp-contract=off like everything else.
#pragma once

#include <hip/hip_runtime.h>

#include "erfcx_coeffs.h"
#include "pair_math.h"

namespace mpmc {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	return v;
}

// lane l receives the value held by lane (l+1) & 63: two v_mov_b32_dpp wave_rol:1 (verified once per device at context creation,
// launch_rot_selftest)
__device__ __forceinline__ double rot_from_next(double v) {
	int lo = __double2loint(v), hi = __double2hiint(v);
	lo = __builtin_amdgcn_update_dpp(lo, lo, 0x134 /* wave_rol:1 */, 0xf, 0xf, false);
	hi = __builtin_amdgcn_update_dpp(hi, hi, 0x134, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
}

// the tensor store is read exactly once per launch: stream it past the caches (global_load ... nt).
// Measured on MI355X (10k atoms): 0.159 ms vs 0.170 ms per launch with default-policy loads.
template <bool NT>
__device__ __forceinline__ double2 ld_stream(const double2 *p) {
	if (NT) {
		double2 v;
		v.x = __builtin_nontemporal_load(&p->x);
		v.y = __builtin_nontemporal_load(&p->y);
		return v;
	}
	return *p;
}

// One Horner step p*t + c with the constant in an SGPR pair.  hipcc's own choice for fma(p, t, literal) on gfx950 is
// "2 x v_mov_b32 (literal -> VGPR pair) + v_fmac_f64", i.e. three VALU issues per step; v_fma_f64 may read one SGPR operand,
// and s_mov_b32 runs on the scalar unit, so this form costs ONE VALU issue per step.
__device__ __forceinline__ double hstep(double p, double t, double c) {
	double o;
	asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(t), "s"(c));
	return o;
}

// exp(x) = 2^k exp(r), k = rint(x log2 e), r = x - k ln2 (two-part ln2), exp(r) by the degree-11 polynomial of
// tools/fit_erfcx.py on [-ln2/2, ln2/2] (rel. err 1.6e-15).  Arguments here are <= 0 (Gaussian, Thole damping).
__device__ __forceinline__ double exp_fast(double x) {
	const double k = rint(x * 1.4426950408889634);
	double r = fma(-k, 6.93147180369123816490e-01, x);
	r = fma(-k, 1.90821492927058770002e-10, r);
	constexpr double e[MPMC_EXP_DEG + 1] = {MPMC_EXP_COEFFS};
	double p = e[MPMC_EXP_DEG];
#pragma unroll
	for (int i = MPMC_EXP_DEG - 1; i >= 0; --i) p = hstep(p, r, e[i]);
	return ldexp(p, (int)k);
}

// erfc(x) = exp(-x^2) * erfcx(x) for every x >= 0; (1+2x) erfcx(x) is the degree-20 polynomial of tools/fit_erfcx.py in
// t = (x-K)/(x+K) (rel. err < 1e-14), so there is no range branch and no libm erfc/erf in the kernel.
// Also returns e = exp(-x^2), which the Ewald field term needs anyway.
__device__ __forceinline__ double erfc_and_gauss(double x, double &e) {
	e = exp_fast(-x * x);
	constexpr double c[MPMC_ERFCX_DEG + 1] = {MPMC_ERFCX_COEFFS};
	const double d1 = x + MPMC_ERFCX_K, d2 = fma(2.0, x, 1.0);
	const double den = d1 * d2;
	double inv = __builtin_amdgcn_rcp(den); // ~2^-23 seed, one Newton step: ~1e-14
	inv = fma(fma(-den, inv, 1.0), inv, inv);
	const double t = (x - MPMC_ERFCX_K) * (d2 * inv); // (x-K)/(x+K)
	double p = c[MPMC_ERFCX_DEG];
#pragma unroll
	for (int k = MPMC_ERFCX_DEG - 1; k >= 0; --k) p = hstep(p, t, c[k]);
	return e * (p * (d1 * inv)); // p / (1+2x)
}

} // namespace mpmc
