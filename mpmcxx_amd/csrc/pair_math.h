// pair_math.h -- per-pair arithmetic of the energy hot path, shared by every HIP kernel.
//
// These are the ONLY places where the physics formulas live.  They are __host__ __device__ (plain C++ without hipcc) so the same
// expressions can be compiled for the host by a sanitizer build (tools/host_asan.sh); the product library only instantiates them
// in device code.
//
// Build contract: this translation unit is compiled with -ffp-contract=off.  The minimum-image distance
// decides pair inclusion and must round exactly like the reference's x86-64 build (no FMA): SURVEY §7
// "bit-exact pair inclusion".  Where a fused multiply-add is wanted for speed it is spelled fma().
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MPMC_HD __host__ __device__ __forceinline__
#else
#define MPMC_HD inline
#endif

namespace mpmc {

constexpr double kPi = 3.141592653589793238462643383279502884;      // reference constants.h:13
constexpr double kOneOverSqrtPi = 0.5641895835477562869480794515607725858440506293289988; // constants.h:48
constexpr double kSmallDR = 1.0e-12;                                  // constants.h:54
constexpr double kMaxValue = 1.0e40;                                  // constants.h:53
constexpr double kDebye2SKA = 85.10597636;                            // constants.h:41
constexpr int kMaxIterationCount = 128;                               // constants.h:52

// per-atom flag bits (host packs them in mpmc_set_atoms)
enum : int {
	AF_FROZEN = 1,     // Atom::frozen
	AF_NULL_RD = 2,    // epsilon == 0 || sigma == 0   (System.cpp:1050)
	AF_HAS_DISP = 4,   // any of c6,c8,c10 != 0        (System.cpp:1052)
	AF_NEG_SIGMA = 8,  // sigma < 0  -> attractive_only (System.cpp:1167)
	AF_ZERO_SIGMA = 16,// sigma == 0                   (System.cpp:1170)
	AF_ZERO_Q = 32,    // charge == 0                  (System.cpp:1059)
	AF_ZERO_ALPHA = 64,// polarizability == 0
	AF_PAD = 128       // padding slot beyond n (never a real atom)
};
// Atoms whose flags change the MIXING of lj_mix (sigma < 0: attractive only; dispersion coefficients), not just the masks of
// pair_exclusions: a tile pair that contains one is left to k_pair_fused.  ONE definition for the host list (context.cpp: upload_atoms)
// and the sweep's skip test (kernels_pair.hip) -- a tile pair must be served by exactly one of the two kernels.
constexpr int kAtomFlagsMixing = AF_HAS_DISP | AF_NEG_SIGMA;

// box constants, passed by value to kernels
struct Box {
	double b[9];   // pbc.basis[q][p]           row-major
	double r[9];   // pbc.reciprocal_basis[q][p] row-major
	double volume, cutoff;
	// squared-distance forms of the reference's cutoff predicates, found on the host by bisection over doubles
	// with a correctly rounded sqrt, so the device never needs an exactly rounded sqrt to decide inclusion:
	//   ri2 <= t_lj   <=>   sqrt(ri2) - 1e-12 < cutoff      (lj :934, thole_field_nopbc :3319)
	//   ri2 <= t_es   <=>   !(sqrt(ri2) > cutoff)           (coulombic_real :1490, real_term :2917)
	double t_lj, t_es;
	double t_wolf; //   ri2 <= t_wolf <=>   sqrt(ri2) < cutoff              (coulombic_wolf :1443)
	int ortho;     // 1 when basis (and therefore reciprocal) is diagonal
};

// ---- minimum_image, reference System.cpp:1202-1279 -----------------------------------------------------
// d = r_i - r_j (caller), returns rimg and dimg; association order as in the reference:
//   img[p] = rint(((0 + R[0][p] d0) + R[1][p] d1) + R[2][p] d2) ; di[p] = d[p] - (((0 + B[0][p] i0) + B[1][p] i1) + B[2][p] i2)
// For a diagonal box the off-diagonal products are exact zeros, so the short form below returns the same
// bits for every finite input.
template <bool ORTHO>
MPMC_HD double min_image(const Box &bx, double dx, double dy, double dz, double &ox, double &oy, double &oz) {
	double ix, iy, iz, tx, ty, tz;
	if (ORTHO) {
		ix = rint(bx.r[0] * dx);
		iy = rint(bx.r[4] * dy);
		iz = rint(bx.r[8] * dz);
		tx = bx.b[0] * ix;
		ty = bx.b[4] * iy;
		tz = bx.b[8] * iz;
	} else {
		ix = rint(((bx.r[0] * dx) + bx.r[3] * dy) + bx.r[6] * dz);
		iy = rint(((bx.r[1] * dx) + bx.r[4] * dy) + bx.r[7] * dz);
		iz = rint(((bx.r[2] * dx) + bx.r[5] * dy) + bx.r[8] * dz);
		tx = ((bx.b[0] * ix) + bx.b[3] * iy) + bx.b[6] * iz;
		ty = ((bx.b[1] * ix) + bx.b[4] * iy) + bx.b[7] * iz;
		tz = ((bx.b[2] * ix) + bx.b[5] * iy) + bx.b[8] * iz;
	}
	double ex = dx - tx, ey = dy - ty, ez = dz - tz;
	double ri2 = ((ex * ex) + ey * ey) + ez * ez;
	double ri = sqrt(ri2);
	if (ri != ri) { // isnan guard, System.cpp:1265
		ox = dx; oy = dy; oz = dz;
		return sqrt(((dx * dx) + dy * dy) + dz * dz);
	}
	ox = ex; oy = ey; oz = ez;
	return ri;
}

// same displacement, squared length only (no sqrt): the inclusion predicates are applied on ri2 (Box::t_lj/t_es)
template <bool ORTHO>
MPMC_HD double min_image_sq(const Box &bx, double dx, double dy, double dz, double &ox, double &oy, double &oz) {
	double ix, iy, iz, tx, ty, tz;
	if (ORTHO) {
		ix = rint(bx.r[0] * dx);
		iy = rint(bx.r[4] * dy);
		iz = rint(bx.r[8] * dz);
		tx = bx.b[0] * ix;
		ty = bx.b[4] * iy;
		tz = bx.b[8] * iz;
	} else {
		ix = rint(((bx.r[0] * dx) + bx.r[3] * dy) + bx.r[6] * dz);
		iy = rint(((bx.r[1] * dx) + bx.r[4] * dy) + bx.r[7] * dz);
		iz = rint(((bx.r[2] * dx) + bx.r[5] * dy) + bx.r[8] * dz);
		tx = ((bx.b[0] * ix) + bx.b[3] * iy) + bx.b[6] * iz;
		ty = ((bx.b[1] * ix) + bx.b[4] * iy) + bx.b[7] * iz;
		tz = ((bx.b[2] * ix) + bx.b[5] * iy) + bx.b[8] * iz;
	}
	ox = dx - tx;
	oy = dy - ty;
	oz = dz - tz;
	return ((ox * ox) + oy * oy) + oz * oz;
}

// 1/sqrt(x) to ~1 ulp: hardware seed (v_rsq_f64, ~2^-23) + two Newton steps.  Values only, never predicates.
MPMC_HD double fast_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
	double y = __builtin_amdgcn_rsq(x);
	double e = fma(-(x * y), y, 1.0);
	y = fma(0.5 * y, e, y);
	e = fma(-(x * y), y, 1.0);
	y = fma(0.5 * y, e, y);
	return y;
#else
	return 1.0 / sqrt(x);
#endif
}

// one Newton step only (~2e-14 relative): enough for the far-field dipole tensor, whose entries feed sums that are
// compared at 1e-9
MPMC_HD double fast_rsqrt_1(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
	double y = __builtin_amdgcn_rsq(x);
	const double e = fma(-(x * y), y, 1.0);
	return fma(0.5 * y, e, y);
#else
	return 1.0 / sqrt(x);
#endif
}

// ---- pair_exclusions, reference System.cpp:1035-1197 (Lorentz-Berthelot branch) --------------------------
struct PairFlags {
	bool intra, frozen, rd_excluded, es_excluded, attractive_only;
};
MPMC_HD PairFlags pair_flags(int mol_i, int fl_i, int mol_j, int fl_j) {
	PairFlags f;
	int any = fl_i | fl_j;
	f.intra = (mol_i == mol_j);
	f.frozen = (fl_i & fl_j & AF_FROZEN) != 0;
	f.rd_excluded = f.intra || ((any & AF_NULL_RD) && !(any & AF_HAS_DISP));
	f.es_excluded = f.intra || (any & AF_ZERO_Q);
	f.attractive_only = (any & AF_NEG_SIGMA) != 0;
	return f;
}
// mixed LJ parameters.  sabs = |sigma|, sqe = sqrt(epsilon) per atom (host precomputes).
// sqrt(ei)*sqrt(ej) replaces the reference's sqrt(ei*ej): <= 1 ulp apart, energies only (never a predicate).
MPMC_HD void lj_mix(int fl_i, int fl_j, double sabs_i, double sqe_i, double sabs_j, double sqe_j, double &sigma, double &epsilon) {
	int any = fl_i | fl_j;
	sigma = (any & AF_ZERO_SIGMA) && !(any & AF_NEG_SIGMA) ? 0.0 : 0.5 * (sabs_i + sabs_j);
	epsilon = (any & AF_NEG_SIGMA) ? 0.0 : sqe_i * sqe_j; // epsilon is never assigned on the sigma<0 branch (System.cpp:1167-1169)
}

// lj_lrc_corr / lj_lrc_self, reference System.Energy.cpp:1036-1096
MPMC_HD double lrc_term(double sigma_abs, double epsilon, double cutoff, double volume) {
	double sig_cut = sigma_abs / cutoff;
	double sig3 = sigma_abs * sigma_abs * sigma_abs;
	double sig_cut3 = sig_cut * sig_cut * sig_cut;
	double sig_cut9 = sig_cut3 * sig_cut3 * sig_cut3;
	return ((16.0 / 3.0) * kPi * epsilon * sig3) * ((1.0 / 3.0) * sig_cut9 - sig_cut3) / volume;
}

// lj pair term, reference System.Energy.cpp:965-993 (caller has applied the inclusion predicate :934-937)
MPMC_HD double lj_term(double sigma_abs, double epsilon, double rimg, bool attractive_only) {
	double s = sigma_abs / rimg;
	double s6 = s * s * s;
	s6 *= s6;
	double s12 = s6 * s6;
	double t12 = attractive_only ? 0.0 : s12;
	return 4.0 * epsilon * (t12 - s6);
}

// Feynman-Hibbs corrections (reference System.Energy.cpp: lj_fh_corr :1100-1148, coulombic_real_FH :1521-1557), per pair, shared by the
// pair sweep and the per-move delta kernels.  c2 = M2A2 hbar^2 / (24 kB T amu2kg), c4 = M2A4 hbar^4 / (1152 kB^2 T^2 amu2kg^2);
// imu = 1/M_i + 1/M_j (molecule masses, amu): the reduced mass enters as its inverse.
MPMC_HD double fh_lj_corr(int order, double c2, double c4, double imu, double eps, double t12, double s6, double ir) {
	const double ir2 = ir * ir;
	const double dE = -24.0 * eps * (2.0 * t12 - s6) * ir;
	const double d2E = 24.0 * eps * (26.0 * t12 - 7.0 * s6) * ir2;
	double corr = c2 * imu * (d2E + 2.0 * dE * ir);
	if (order >= 4) {
		const double ir3 = ir2 * ir;
		const double d3E = -1344.0 * eps * (6.0 * t12 - s6) * ir3;
		const double d4E = 12096.0 * eps * (10.0 * t12 - s6) * (ir2 * ir2);
		corr += c4 * (imu * imu) * (15.0 * dE * ir3 + 4.0 * d3E * ir + d4E);
	}
	return corr;
}
// (added WITHOUT the charge product, as the reference does, :1499-1500); erfc_a = erfc(alpha r), gauss_a = exp(-alpha^2 r^2)
MPMC_HD double fh_es_corr(int order, double c2, double c4, double imu, double al, double erfc_a, double gauss_a, double ri2, double r, double ir) {
	const double a2 = al * al, a3 = a2 * al;
	const double ir2 = ir * ir, ir3 = ir2 * ir, ir4 = ir2 * ir2;
	const double isp = kOneOverSqrtPi;
	const double du = -2.0 * al * gauss_a * ir * isp - erfc_a * ir2;
	const double d2u = 4.0 * isp * gauss_a * (a3 + ir2) + 2.0 * erfc_a * ir3;
	double corr = c2 * imu * (d2u + 2.0 * du * ir);
	if (order >= 4) {
		const double d3u = gauss_a * isp * (-8.0 * (a3 * a2) * r - 8.0 * a3 * ir - 12.0 * al * ir3) - 6.0 * erfc_a * ir4;
		const double d4u = gauss_a * isp * (8.0 * a3 * a2 + 16.0 * a3 * (a3 * al) * ri2 + 32.0 * a3 * ir2 + 48.0 * ir4) + 24.0 * erfc_a * (ir4 * ir);
		corr += c4 * (imu * imu) * (15.0 * du * ir3 + 4.0 * d3u * ir + d4u);
	}
	return corr;
}

// Thole exponential damping, reference System.Energy.cpp:2731-2757:  T = a*I - b*(d (x) d),
//   a = damp1/r^3, b = 3*damp2/r^5.  r == 0 gives the reference's MAXVALUE guard (:2704-2705).
MPMC_HD void thole_ab(double r, double lambda, double &a, double &b) {
	double ir3, ir5;
	if (r == 0.0) {
		ir3 = ir5 = kMaxValue;
	} else {
		double ir = 1.0 / r;
		ir3 = ir * ir * ir;
		ir5 = ir3 * ir * ir;
	}
	double r2 = r * r;
	double l2 = lambda * lambda, l3 = l2 * lambda;
	double explr = exp(-lambda * r);
	double damp1 = 1.0 - explr * (0.5 * l2 * r2 + lambda * r + 1.0);
	double damp2 = damp1 - explr * (l3 * r2 * r / 6.0);
	a = damp1 * ir3;
	b = 3.0 * damp2 * ir5;
}

// real_term factor, reference System.Energy.cpp:2919-2934 (caller applied  !(r > rc || r == 0) and !frozen)
MPMC_HD double field_real_factor(double r, double alpha, bool es_excluded) {
	double r2 = r * r;
	double g = 2.0 * alpha * kOneOverSqrtPi * exp(-alpha * alpha * r2) * r;
	return es_excluded ? (g - erf(alpha * r)) / (r * r2) : (g + erfc(alpha * r)) / (r2 * r);
}

} // namespace mpmc
