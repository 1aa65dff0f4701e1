// kernels_sym.hip -- the symmetric ("each unordered pair once") gfx950 kernels of the production path.
//
//   k_pair_fused          ONE pass over all N(N-1)/2 pairs does everything position dependent that is pairwise:
//                         lj() + pair LRC, coulombic_real() (erfc and intramolecular erf term), the real-space / no-PBC
//                         static field (real_term / thole_field_nopbc, both atoms of the pair), the in-cutoff pair
//                         counts, and the Thole tensor store (a = d1/r^3, b = 3 d2/r^5 per pair, 16 B) that the
//                         dipole iterations stream afterwards.
//   k_dipole_iter_compact one Jacobi contraction  F = -(A - diag) mu  streaming that store: 16 B per unordered pair,
//                         each pair's tensor applied to both of its atoms.  HBM-bound by construction.
//
// Wave-level schedule (both kernels): one wavefront per (I <= J) tile pair of 64 x 64 atoms.  Lane l owns i-atom
// I*64+l in registers.  The j-atoms sit in LDS; at step s lane l pairs with j = (l + s) & 63, so the 64 lanes touch 64
// DIFFERENT j-atoms in every step: the per-atom vector accumulators of the j-side travel in registers and are
// rotated by one lane per step (v_mov_b32_dpp wave_rol:1), never reduced across lanes and never sent through
// atomics.  Diagonal tiles use s = 1..32 (s = 32: lanes 0..31 only) which enumerates each pair once.
// Per-atom partial results go to slot [source tile][atom] of a partial buffer that is completely overwritten by
// every launch and summed in tile order by the follow-up kernel => bit-reproducible.
#include <algorithm>
#include <cstdlib>

#include "kernels.h"
#include "device_math.h"

namespace mpmc {

// first step of an off-diagonal tile walk: 16 different chunk offsets spread over consecutive blocks
__device__ __forceinline__ int stagger_start(int block) { return ((block * 5) & 15) * 4; }

// self-test of the rotation primitive: out[l] = lane id received by lane l (expected (l+1)&63)
__global__ void k_rot_selftest(int *out) {
	const int lane = threadIdx.x;
	out[lane] = (int)rot_from_next((double)lane);
}
void launch_rot_selftest(hipStream_t st, int *out) { hipLaunchKernelGGL(k_rot_selftest, dim3(1), dim3(64), 0, st, out); }

// ------------------------------------------------------------------------------------------------------
// fused symmetric pair kernel
//   ES    : electrostatics on (coulombic_real)
//   FIELD : 0 none, 1 Ewald real_term (:2900-2940), 2 thole_field_nopbc (:3300-3333)
//   THOLE : write the (a,b) tensor store
// ------------------------------------------------------------------------------------------------------
//   EXT   : adjacent physics compiled in (Wolf electrostatics, Feynman-Hibbs corrections); the default instantiations
//           carry none of that code
//   ALPHA2: polar_ewald_alpha differs from ewald_alpha (both user-set): the field term needs an erfc of its own.  An instantiation,
//           not a call: an out-of-line second erfc put a function call into the loop, and the SGPRs saved around it (13 spilled
//           through v_writelane / v_readlane, 16 B of scratch) were paid by every step of the common case.
//   TAIL  : small systems, LJ only: the block that finishes last adds up the block partials (fixed order) and writes the scalar
//           results straight into the caller's pinned host buffer -- the whole evaluation is ONE launch, no classes, no copy back
struct PairTail {
	int *counter;     // zero between launches (the last block resets it)
	double *out_host; // [S_COUNT + C_COUNT + 1] device-visible pinned host memory (mpmc_ctx::h_scal); the last slot receives `seq`
	double seq;       // launch number: written AFTER the results (system-scope fence in between), so a host that polls it sees them
};
//   WAVES : waves per tile pair (1 or 4).  A lone wave's 64 dependent steps take 16 us with LJ alone and 65 us with Ewald + field +
//           Thole store; a table of a few hundred tile pairs leaves most SIMDs empty, so small systems split the steps over four
//           waves whose sums meet in LDS in wave order (reproducible).  Large tables keep one wave per tile pair (one prologue).
template <bool ORTHO, bool ES, int FIELD, bool THOLE, bool EXT = false, bool ALPHA2 = false, bool TAIL = false, int WAVES = (TAIL ? 4 : 1)>
__global__ __launch_bounds__(64 * WAVES) void k_pair_fused(AtomsDev at, Box bx, FusedParams fp, const int2 *__restrict__ tile_pairs,
                                                   const int *__restrict__ cls, double *__restrict__ block_part, int *__restrict__ block_cnt,
                                                   double *__restrict__ fpart /*[nt][n_pad][3]*/, double2 *__restrict__ ab, PairTail tail = PairTail{},
                                                   const int *__restrict__ tp_list = nullptr /*the tile pairs of this launch (null: all, block = tile pair)*/) {
	__shared__ double s_x[kTile], s_y[kTile], s_z[kTile], s_q[kTile], s_sig[kTile], s_sqe[kTile];
	__shared__ int s_mol[kTile], s_fl[kTile];
	__shared__ double s_g[3 * kTile];

	static_assert(WAVES == 1 || WAVES == 4, "one wave per tile pair, or four");
	static_assert(!TAIL || WAVES == 4, "the single-launch form is the four-wave form");
	constexpr int W = WAVES;
	const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int tp = (!TAIL && tp_list) ? tp_list[blockIdx.x] : (int)blockIdx.x;
	const int2 IJ = tile_pairs[tp];
	if (THOLE && fp.store_only && fp.touch_n >= 0) { // (block-uniform) a trial move: only the tile pairs of the moved atoms' tiles are rebuilt
		bool hit = false;
		for (int k = 0; k < fp.touch_n; ++k) hit = hit || (IJ.x == fp.touch[k]) || (IJ.y == fp.touch[k]);
		if (!hit) return;
	}
	const bool diag = (IJ.x == IJ.y);
	const int i = IJ.x * kTile + lane;
	const int j0 = IJ.y * kTile;
	// tile-pair class (wave-uniform): whole tile pair beyond the cutoff / beyond the Thole damping range
	const int cl = TAIL ? 0 : cls[tp]; // (the single-launch form of small systems carries no classes: every tile pair is "near")
	const bool beyond = (cl & CLS_BEYOND_CUTOFF) != 0;
	const bool store_thole = THOLE && !(cl & CLS_THOLE_FAR);
	if (beyond && !store_thole) { // nothing position dependent to do: publish zeros so the fixed-shape reductions stay valid
		if (FIELD != 0) {
			const int nt_pad3 = at.n_pad * 3;
			double *oi = fpart + (size_t)IJ.y * nt_pad3 + 3 * (size_t)i;
			double *oj = fpart + (size_t)IJ.x * nt_pad3 + 3 * (size_t)(j0 + lane);
			oi[0] = oi[1] = oi[2] = 0.0;
			oj[0] = oj[1] = oj[2] = 0.0;
		}
		if (lane == 0) {
			block_part[2 * (size_t)tp] = 0.0;
			block_part[2 * (size_t)tp + 1] = 0.0;
			block_cnt[2 * (size_t)tp] = 0;
			block_cnt[2 * (size_t)tp + 1] = 0;
		}
		return;
	}

	const double4 pi = at.xyzq[i];
	const double2 li = at.lj[i];
	const int2 mi = at.mf[i];
	__shared__ double s_imm[EXT ? kTile : 1];
	double imm_i = 0.0;
	if (EXT && fp.fh_order) {
		imm_i = at.inv_molmass[i];
		s_imm[lane] = at.inv_molmass[j0 + lane];
	}
	if (w == 0) {
		const double4 pj = at.xyzq[j0 + lane];
		const double2 lj = at.lj[j0 + lane];
		const int2 mj = at.mf[j0 + lane];
		s_x[lane] = pj.x;
		s_y[lane] = pj.y;
		s_z[lane] = pj.z;
		s_q[lane] = pj.w;
		s_sig[lane] = lj.x;
		s_sqe[lane] = lj.y;
		s_mol[lane] = mj.x;
		s_fl[lane] = mj.y;
	}
	__syncthreads();

	const bool i_real = !(mi.y & AF_PAD);
	constexpr bool same_alpha = !ALPHA2;
	const double lam = fp.polar_damp;
	double e_lj = 0, e_re = 0;
	int n_lj = 0, n_es = 0;
	double eix = 0, eiy = 0, eiz = 0; // field on my i-atom
	double gx = 0, gy = 0, gz = 0;    // field on the j-atom currently paired with this lane (rotates)

	// step order: diagonal tiles s = 1..32; off-diagonal tiles walk all 64 steps starting at a per-block offset
	// (multiple of 4) so that concurrently running waves do not hit the same HBM channels in lock step
	const int s_first = diag ? 1 : stagger_start(tp), n_steps = diag ? 32 : 64;
	double2 *ab_tile = store_thole ? ab + (size_t)tp * (kTile * kTile) : nullptr;

	// "plain" tile pair (wave-uniform): no atom of either tile is frozen, padded, chargeless, has a zero / negative
	// sigma, zero epsilon or dispersion coefficients -- then the exclusion logic of pair_exclusions collapses to
	// "same molecule" and the Lorentz-Berthelot mixing to one add and one multiply
	constexpr int kSpecial = AF_FROZEN | AF_NULL_RD | AF_HAS_DISP | AF_NEG_SIGMA | AF_ZERO_SIGMA | AF_ZERO_Q | AF_PAD;
	const bool plain = !__any(((mi.y | s_fl[lane]) & kSpecial) != 0);

	const int k_begin = w * (n_steps / W), k_end = k_begin + n_steps / W;
	for (int k = k_begin; k < k_end; ++k) {
		const int s = diag ? (s_first + k) : ((s_first + k) & 63);
		const bool last = (k == k_end - 1); // this wave's last step: its j-side accumulators stay where they are
		const int jl = (lane + s) & 63;
		const int molj = s_mol[jl];
		bool act;
		int flj;
		PairFlags f;
		if (plain) {
			flj = 0;
			act = (!diag || s < 32 || lane < 32);
			f.intra = (mi.x == molj);
			f.frozen = false;
			f.rd_excluded = f.es_excluded = f.intra;
			f.attractive_only = false;
		} else {
			flj = s_fl[jl];
			act = i_real && !(flj & AF_PAD) && (!diag || s < 32 || lane < 32);
			f = pair_flags(mi.x, mi.y, molj, flj);
		}
		double ta = 0.0, tb = 0.0;
		if (act) {
			const double qj = s_q[jl];
			const double dx = pi.x - s_x[jl], dy = pi.y - s_y[jl], dz = pi.z - s_z[jl];
			double ox, oy, oz;
			const double ri2 = min_image_sq<ORTHO>(bx, dx, dy, dz, ox, oy, oz);
			const double ir = fast_rsqrt_1(ri2); // 1/rimg to ~2e-14 (inf/NaN when ri2 == 0: coincident atoms => non-finite energy, as in the reference)
			const double r = ri2 * ir;
			const bool in_lj = (ri2 <= bx.t_lj); // rimg - 1e-12 < rc
			const bool in_es = (ri2 <= bx.t_es); // !(rimg > rc)

			if (store_thole) { // thole_amatrix couples every pair: no cutoff, no exclusions, frozen included (:2694-2767)
				double ir3, ir5;
				if (ri2 == 0.0) {
					ir3 = ir5 = kMaxValue;
				} else {
					ir3 = ir * ir * ir;
					ir5 = ir3 * ir * ir;
				}
				const double rr = (ri2 == 0.0) ? 0.0 : r;
				const double lr = lam * rr;
				double damp1 = 1.0, damp2 = 1.0;
				if (__any(lr < fp.thole_far_x)) { // wave-uniform: beyond lambda r = kTholeFarX the damping is dropped (as in CLS_THOLE_FAR)
					const double explr = exp_fast(-lr);
					damp1 = fma(-explr, fma(lr, fma(0.5, lr, 1.0), 1.0), 1.0);   // 1 - e^{-lr} (lr^2/2 + lr + 1)
					damp2 = fma(-explr, (lr * lr) * (lr * (1.0 / 6.0)), damp1);    // damp1 - e^{-lr} lr^3/6
				}
				ta = damp1 * ir3;
				tb = 3.0 * damp2 * ir5;
			}

			if (!f.frozen && !beyond && !fp.store_only) {
				double sig, eps;
				if (plain) {
					sig = 0.5 * (li.x + s_sig[jl]);
					eps = li.y * s_sqe[jl];
				} else {
					lj_mix(mi.y, flj, li.x, li.y, s_sig[jl], s_sqe[jl], sig, eps);
				}
				if (in_lj && !f.rd_excluded) {
					const double sr = sig * ir;
					double s6 = sr * sr * sr;
					s6 *= s6;
					const double t12 = f.attractive_only ? 0.0 : s6 * s6;
					e_lj = fma(4.0 * eps, t12 - s6, e_lj);
					n_lj++;
					if (EXT && fp.fh_order) // lj_fh_corr :1100-1148
						e_lj += fh_lj_corr(fp.fh_order, fp.fh_c2, fp.fh_c4, imm_i + s_imm[jl], eps, t12, s6, ir);
				}
				if (ES) {
					const double qq = pi.w * qj;
					const bool wolf_on = EXT && fp.wolf;
					const bool es_pair = in_es && !f.es_excluded && !wolf_on; // coulombic_real :1490
					if (wolf_on && !f.es_excluded && (ri2 <= bx.t_wolf)) { // coulombic_wolf :1443-1445 (r < R)
						e_re = fma(qq, ir - fp.wolf_erfa_over_r - fp.wolf_inv_r2 * (bx.cutoff - r), e_re);
						n_es++;
					}
					const bool fld_pair = (FIELD == 1) && in_es && (ri2 != 0.0) && !(pi.w == 0.0 && qj == 0.0); // real_term :2916-2917
					double erfc_a = 0.0, gauss_a = 0.0;
					if (es_pair || (fld_pair && same_alpha)) erfc_a = erfc_and_gauss(fp.ewald_alpha * r, gauss_a); // the ONE erfc of this pair
					if (es_pair) {
						e_re = fma(qq * erfc_a, ir, e_re);
						n_es++;
						if (EXT && fp.fh_order) // coulombic_real_FH :1521-1557
							e_re += fh_es_corr(fp.fh_order, fp.fh_c2, fp.fh_c4, imm_i + s_imm[jl], fp.ewald_alpha, erfc_a, gauss_a, ri2, r, ir);
					} // (the intramolecular charge-to-screen term, :1503-1504, is summed by k_intra_terms)
					if (FIELD == 1 && fld_pair) { // real_term :2919-2934: erfc form, or erf form (= 1 - erfc) for es_excluded pairs
						const double ap = fp.polar_ewald_alpha;
						double ec = erfc_a, ga = gauss_a;
						if (!same_alpha) ec = erfc_and_gauss(ap * r, ga);
						const double g = (2.0 * kOneOverSqrtPi * ap) * (ga * r);
						const double fac = (f.es_excluded ? (g - (1.0 - ec)) : (g + ec)) * (ir * ir * ir);
						const double fj = fac * qj, fi = fac * pi.w;
						eix = fma(fj, ox, eix);
						eiy = fma(fj, oy, eiy);
						eiz = fma(fj, oz, eiz);
						gx = fma(-fi, ox, gx);
						gy = fma(-fi, oy, gy);
						gz = fma(-fi, oz, gz);
					}
					if (FIELD == 2 && !f.intra && in_lj && ri2 != 0.0) { // thole_field_nopbc :3311-3326
						const double ir3 = ir * ir * ir;
						const double fj = qj * ir3, fi = pi.w * ir3;
						eix = fma(fj, ox, eix);
						eiy = fma(fj, oy, eiy);
						eiz = fma(fj, oz, eiz);
						gx = fma(-fi, ox, gx);
						gy = fma(-fi, oy, gy);
						gz = fma(-fi, oz, gz);
					}
				}
			}
		}
		if (store_thole) ab_tile[s * kTile + lane] = make_double2(ta, tb);
		if (FIELD != 0 && !last) {
			gx = rot_from_next(gx);
			gy = rot_from_next(gy);
			gz = rot_from_next(gz);
		}
	}

	if (FIELD != 0) {
		int jown = (lane + s_first + k_end - 1) & 63; // the j-atom whose accumulator this lane ended up holding
		const int nt_pad3 = at.n_pad * 3;
		bool writer = true;
		if (W > 1) { // the waves' shares meet in LDS, parked at the atom they belong to, and are added in wave order by wave 0
			__shared__ double s_fw[W][6][kTile];
			s_fw[w][0][lane] = eix;
			s_fw[w][1][lane] = eiy;
			s_fw[w][2][lane] = eiz;
			s_fw[w][3][jown] = gx;
			s_fw[w][4][jown] = gy;
			s_fw[w][5][jown] = gz;
			__syncthreads();
			writer = (w == 0);
			if (writer) {
				double v[6];
#pragma unroll
				for (int d = 0; d < 6; ++d) {
					v[d] = s_fw[0][d][lane];
#pragma unroll
					for (int k = 1; k < W; ++k) v[d] += s_fw[k][d][lane];
				}
				eix = v[0], eiy = v[1], eiz = v[2];
				gx = v[3], gy = v[4], gz = v[5];
				jown = lane;
			}
		}
		if (writer) {
			if (diag) { // both sides belong to the same 64 atoms: one slot [I][I-atoms]
				if (W == 1) { // fold through LDS
					s_g[3 * jown + 0] = gx;
					s_g[3 * jown + 1] = gy;
					s_g[3 * jown + 2] = gz;
					__syncthreads();
					gx = s_g[3 * lane + 0];
					gy = s_g[3 * lane + 1];
					gz = s_g[3 * lane + 2];
				}
				double *o = fpart + (size_t)IJ.x * nt_pad3 + 3 * (size_t)i;
				o[0] = eix + gx;
				o[1] = eiy + gy;
				o[2] = eiz + gz;
			} else {
				double *oi = fpart + (size_t)IJ.y * nt_pad3 + 3 * (size_t)i; // i-atoms, contribution of tile J
				oi[0] = eix;
				oi[1] = eiy;
				oi[2] = eiz;
				double *oj = fpart + (size_t)IJ.x * nt_pad3 + 3 * (size_t)(j0 + jown); // j-atoms, contribution of tile I
				oj[0] = gx;
				oj[1] = gy;
				oj[2] = gz;
			}
		}
	}

	e_lj = wave_sum(e_lj);
	e_re = wave_sum(e_re);
	n_lj = wave_sum_i(n_lj);
	n_es = wave_sum_i(n_es);
	if (W > 1) { // the waves' shares of the tile pair, added in wave order
		__shared__ double s_pe[W][2];
		__shared__ int s_pc[W][2];
		if (lane == 0) {
			s_pe[w][0] = e_lj;
			s_pe[w][1] = e_re;
			s_pc[w][0] = n_lj;
			s_pc[w][1] = n_es;
		}
		__syncthreads();
		if (w != 0) return;
		for (int k = 1; k < W; ++k) {
			e_lj += s_pe[k][0];
			e_re += s_pe[k][1];
			n_lj += s_pc[k][0];
			n_es += s_pc[k][1];
		}
	}
	if (lane == 0) {
		block_part[2 * (size_t)tp] = e_lj;
		block_part[2 * (size_t)tp + 1] = e_re;
		block_cnt[2 * (size_t)tp] = n_lj;
		block_cnt[2 * (size_t)tp + 1] = n_es;
	}
	if (TAIL) {
		// last-arriving block: everybody publishes its partials (release), takes a ticket; the holder of the last ticket sees them all
		// (acquire) and folds them in block order -- the same order whichever block comes last, so the result is reproducible
		// (only wave 0 of a block is still here)
		int is_last = 0;
		if (lane == 0) {
			__threadfence();
			const int ticket = atomicAdd(tail.counter, 1);
			is_last = (ticket == (int)gridDim.x - 1);
		}
		if (!__builtin_amdgcn_readfirstlane(is_last)) return; // lane 0's verdict for the whole wave
		__threadfence();
		double s0 = 0, s1 = 0;
		long long c0 = 0, c1 = 0;
		for (int b = lane; b < (int)gridDim.x; b += kTile) {
			s0 += __builtin_nontemporal_load(block_part + 2 * (size_t)b);
			s1 += __builtin_nontemporal_load(block_part + 2 * (size_t)b + 1);
			c0 += __builtin_nontemporal_load(block_cnt + 2 * (size_t)b);
			c1 += __builtin_nontemporal_load(block_cnt + 2 * (size_t)b + 1);
		}
		s0 = wave_sum(s0);
		s1 = wave_sum(s1);
		for (int off = 32; off > 0; off >>= 1) {
			c0 += __shfl_down(c0, off, 64);
			c1 += __shfl_down(c1, off, 64);
		}
		if (lane < S_COUNT + C_COUNT) {
			double v = 0.0;
			if (lane == S_LJ) v = s0;
			if (lane == S_ES_REAL) v = s1;
			tail.out_host[lane] = v;
		}
		if (lane == 0) {
			long long *cnt = reinterpret_cast<long long *>(tail.out_host + S_COUNT);
			cnt[C_LJ_IN] = c0;
			cnt[C_ES_IN] = c1;
			*tail.counter = 0;
		}
		__threadfence_system();
		if (lane == 0) __hip_atomic_store(tail.out_host + S_COUNT + C_COUNT, tail.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

template <bool ORTHO, bool ES, int FIELD, bool THOLE>
static void launch_fused_t(hipStream_t st, const AtomsDev &at, const Box &bx, const FusedParams &fp, const int2 *tp, const int *cls, int ntp,
                           double *bpart, int *bcnt, double *fpart, double2 *ab, const int *tp_list) {
#define MPMC_PF(EXT, A2, W) \
	hipLaunchKernelGGL((k_pair_fused<ORTHO, ES, FIELD, THOLE, EXT, A2, false, W>), dim3(ntp), dim3(W * kTile), 0, st, at, bx, fp, tp, cls, bpart, bcnt, fpart, ab, PairTail{}, tp_list)
	if (FIELD == 1 && fp.polar_ewald_alpha != fp.ewald_alpha) MPMC_PF(true, true, 1); // two different Ewald alphas: every extension compiled in
	else if ((ES && fp.wolf) || fp.fh_order) MPMC_PF(true, false, 1);                 // Wolf / Feynman-Hibbs
	else if (fp.pair_waves == 4) MPMC_PF(false, false, 4);                            // small table: four waves per tile pair
	else MPMC_PF(false, false, 1);
#undef MPMC_PF
}

template <bool ORTHO>
static void launch_fused_o(hipStream_t st, const AtomsDev &at, const Box &bx, const FusedParams &fp, const int2 *tp, const int *cls,
                           int ntp, double *bpart, int *bcnt, double *fpart, double2 *ab, const int *tp_list) {
	const bool thole = fp.do_thole && ab;
	if (!fp.do_es && thole) // (store-only sweeps of polarizable trial moves)
		launch_fused_t<ORTHO, false, 0, true>(st, at, bx, fp, tp, cls, ntp, bpart, bcnt, fpart, ab, tp_list);
	else if (!fp.do_es)
		launch_fused_t<ORTHO, false, 0, false>(st, at, bx, fp, tp, cls, ntp, bpart, bcnt, fpart, ab, tp_list);
	else if (fp.do_field == 0)
		launch_fused_t<ORTHO, true, 0, false>(st, at, bx, fp, tp, cls, ntp, bpart, bcnt, fpart, ab, tp_list);
	else if (fp.do_field == 1) {
		if (thole) launch_fused_t<ORTHO, true, 1, true>(st, at, bx, fp, tp, cls, ntp, bpart, bcnt, fpart, ab, tp_list);
		else launch_fused_t<ORTHO, true, 1, false>(st, at, bx, fp, tp, cls, ntp, bpart, bcnt, fpart, ab, tp_list);
	} else {
		if (thole) launch_fused_t<ORTHO, true, 2, true>(st, at, bx, fp, tp, cls, ntp, bpart, bcnt, fpart, ab, tp_list);
		else launch_fused_t<ORTHO, true, 2, false>(st, at, bx, fp, tp, cls, ntp, bpart, bcnt, fpart, ab, tp_list);
	}
}

// small systems, LJ (+ LRC) only: one launch, results land in `out_host` (pinned, device-visible) when the stream has drained
void launch_pair_lj_single(hipStream_t st, const AtomsDev &at, const Box &bx, const FusedParams &fp, const int2 *tile_pairs, int n_tile_pairs,
                           double *block_part, int *block_cnt, int *counter, double *out_host, double seq) {
	const PairTail tail{counter, out_host, seq};
	dim3 grid(n_tile_pairs), block(4 * kTile);
	if (bx.ortho)
		hipLaunchKernelGGL((k_pair_fused<true, false, 0, false, false, false, true>), grid, block, 0, st, at, bx, fp, tile_pairs, nullptr, block_part, block_cnt, nullptr, nullptr, tail);
	else
		hipLaunchKernelGGL((k_pair_fused<false, false, 0, false, false, false, true>), grid, block, 0, st, at, bx, fp, tile_pairs, nullptr, block_part, block_cnt, nullptr, nullptr, tail);
}

void launch_pair_fused(hipStream_t st, const AtomsDev &at, const Box &bx, const FusedParams &fp, const int2 *tile_pairs,
                       const int *cls, int n_tile_pairs, double *block_part, int *block_cnt, double *fpart, double2 *ab, const int *tp_list) {
	if (n_tile_pairs <= 0) return;
	if (bx.ortho) launch_fused_o<true>(st, at, bx, fp, tile_pairs, cls, n_tile_pairs, block_part, block_cnt, fpart, ab, tp_list);
	else launch_fused_o<false>(st, at, bx, fp, tile_pairs, cls, n_tile_pairs, block_part, block_cnt, fpart, ab, tp_list);
}

// ------------------------------------------------------------------------------------------------------
// intramolecular charge-to-screen term of coulombic_real (:1503-1504): sum over non-frozen same-molecule pairs of
// q_i q_j erf(alpha r)/r with the PLAIN distance, whatever the cutoff.  Molecules are contiguous runs of the original
// atom order (validated in mpmc_set_atoms), so the sum is O(N * atoms per molecule).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_intra_terms(AtomsDev at, const int *__restrict__ slot_of, double alpha, double *__restrict__ scal) {
	__shared__ double sh[4];
	double acc = 0;
	for (int i = threadIdx.x; i < at.n; i += 256) {
		const int si = slot_of[i];
		const int2 mi = at.mf[si];
		const double4 pi = at.xyzq[si];
		for (int j = i + 1; j < at.n; ++j) {
			const int sj = slot_of[j];
			const int2 mj = at.mf[sj];
			if (mj.x != mi.x) break;
			if (mi.y & mj.y & AF_FROZEN) continue;
			const double4 pj = at.xyzq[sj];
			const double qq = pi.w * pj.w;
			if (qq == 0.0) continue;
			const double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
			const double r2 = ((dx * dx) + dy * dy) + dz * dz;
			const double ir = fast_rsqrt(r2);
			double g;
			acc += qq * (1.0 - erfc_and_gauss(alpha * (r2 * ir), g)) * ir; // erf = 1 - erfc
		}
	}
	acc = wave_sum(acc);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) scal[S_ES_INTRA] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}
void launch_intra_terms(hipStream_t st, const AtomsDev &at, const int *slot_of, double ewald_alpha, double *scal) {
	hipLaunchKernelGGL(k_intra_terms, dim3(1), dim3(256), 0, st, at, slot_of, ewald_alpha, scal);
}

// ------------------------------------------------------------------------------------------------------
// tile bounding boxes in wrapped fractional coordinates and tile-pair classes (any cell)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_tile_bounds(AtomsDev at, Box bx, double3 origin_f, double *__restrict__ tb /*[nt][12]*/) {
	const int lane = threadIdx.x;
	const int k = blockIdx.x * kTile + lane;
	const double4 p = at.xyzq[k];
	const bool real = !(at.mf[k].y & AF_PAD);
	double lo[6], hi[6]; // 0..2 wrapped fractional coordinates; 3..5 raw coordinates: Cartesian (orthorhombic cell) / fractional (any other)
	const double pos[3] = {p.x, p.y, p.z};
	const double of[3] = {origin_f.x, origin_f.y, origin_f.z};
	for (int d = 0; d < 3; ++d) {
		// fractional coordinate d = column d of the reciprocal basis . position (the off-diagonal entries of a diagonal cell are exact zeros),
		// wrapped into one period counted from the origin of the spatial sort
		const double fraw = ((bx.r[d] * pos[0]) + bx.r[3 + d] * pos[1]) + bx.r[6 + d] * pos[2];
		double f = fraw - of[d];
		f -= floor(f);
		lo[d] = real ? f : 2.0;
		hi[d] = real ? f : -1.0;
		const double raw = bx.ortho ? pos[d] : fraw;
		lo[3 + d] = real ? raw : 1e300;
		hi[3 + d] = real ? raw : -1e300;
	}
	for (int off = 32; off > 0; off >>= 1)
		for (int d = 0; d < 6; ++d) {
			lo[d] = fmin(lo[d], __shfl_down(lo[d], off, 64));
			hi[d] = fmax(hi[d], __shfl_down(hi[d], off, 64));
		}
	if (lane == 0)
		for (int d = 0; d < 3; ++d) {
			tb[12 * (size_t)blockIdx.x + d] = lo[d];
			tb[12 * (size_t)blockIdx.x + 3 + d] = hi[d];
			tb[12 * (size_t)blockIdx.x + 6 + d] = lo[3 + d];
			tb[12 * (size_t)blockIdx.x + 9 + d] = hi[3 + d];
		}
}

// Lower bound of the minimum-image distance between two tiles from the gaps g_d >= 0 of their fractional bounding boxes on the unit
// circle.  The image displacement d = B^T f has |f_d| >= g_d in every dimension, so
//   orthorhombic cell:  |d|^2 >= sum_d (L_d g_d)^2
//   any cell:           |d| >= g_d / |R_d|  for every d  (R_d: column d of the reciprocal basis, the normal of the planes f_d = const, whose
//                       spacing is 1 / |R_d|),  and  |d|^2 >= lambda_min(B B^T) |g|^2
// (cell.plane[d] = 1 / |R_d|, cell.lam_min = lambda_min, both from the host).
struct CellBounds {
	double plane[3];
	double lam_min;
};
__global__ __launch_bounds__(256) void k_classify(const double *__restrict__ tb, const int2 *__restrict__ tile_pairs, int ntp, Box bx, CellBounds cell,
                                                  double thr_cut2, double thr_far2, int *__restrict__ cls, double4 *__restrict__ tp_shift) {
	const int t = blockIdx.x * 256 + threadIdx.x;
	if (t >= ntp) return;
	const int2 IJ = tile_pairs[t];
	int c = 0;
	if (IJ.x != IJ.y) {
		double d2 = 0, g2 = 0, dmax = 0;
		for (int d = 0; d < 3; ++d) {
			const double a0 = tb[12 * (size_t)IJ.x + d], a1 = tb[12 * (size_t)IJ.x + 3 + d];
			const double b0 = tb[12 * (size_t)IJ.y + d], b1 = tb[12 * (size_t)IJ.y + 3 + d];
			double gap = 0.0; // distance between the two intervals on the unit circle
			if (a1 < b0) gap = fmin(b0 - a1, a0 + 1.0 - b1);
			else if (b1 < a0) gap = fmin(a0 - b1, b0 + 1.0 - a1);
			gap = fmax(0.0, gap - 1e-12); // rounding guard: the bound must stay a LOWER bound
			if (bx.ortho) {
				const double L = fabs(bx.b[4 * d]);
				d2 += (L * gap) * (L * gap);
			} else {
				g2 += gap * gap;
				dmax = fmax(dmax, gap * cell.plane[d]);
			}
		}
		if (!bx.ortho) d2 = fmax(dmax * dmax, cell.lam_min * g2) * (1.0 - 1e-12);
		if (d2 > thr_cut2) c |= CLS_BEYOND_CUTOFF;
		if (thr_far2 > 0.0 && d2 > thr_far2) c |= CLS_THOLE_FAR;
	}
	if (tp_shift) {
		// Is the periodic image index rint(R d) the same for every atom pair of the tile pair?  The subtraction x_i - x_j, the products, the
		// sums and rint are all monotone in each argument (also after rounding), so it suffices that the very expression the pair would
		// evaluate (minimum_image, src/System.cpp:1228-1246, same association order) rounds to the same integer at the two extreme corners of
		// the tiles' RAW coordinate ranges -- even a pair sitting exactly on the half-box tie gets the reference's image.
		//   orthorhombic cell: per dimension (CLS_UNIFORM_X/Y/Z), shift component B_dd img_d;
		//   any other cell: bit p = the index along lattice vector p is common; the translation B^T img mixes the Cartesian components, so
		//   the shift is a whole vector (the sum over the common directions), and
		//   the test runs on the tiles' RAW FRACTIONAL ranges (a tile is compact in fractional coordinates; its Cartesian box is not):
		//   the pair's own value of (R d)_p differs from the difference of the two atoms' fractional coordinates by rounding only, so a range
		//   that stays 1e-9 clear of the half-integers on both sides has one image index (a pair ON a half-box tie is left to the general
		//   path, which rounds it like the reference).
		double lo[3], hi[3];
		bool ok = true;
		for (int q = 0; q < 3; ++q) {
			const double ilo = tb[12 * (size_t)IJ.x + 6 + q], ihi = tb[12 * (size_t)IJ.x + 9 + q];
			const double jlo = tb[12 * (size_t)IJ.y + 6 + q], jhi = tb[12 * (size_t)IJ.y + 9 + q];
			lo[q] = ilo - jhi;
			hi[q] = ihi - jlo;
			ok = ok && (ihi >= ilo) && (jhi >= jlo);
		}
		double sh[3] = {0, 0, 0};
		double band = 1e30; // see below: relative half-width of the band in which the pair sweep's fused geometry defers to the reference's form
		if (bx.ortho) {
			double cmax = 0.0;
			for (int d = 0; d < 3; ++d) {
				const double m0 = rint(bx.r[4 * d] * lo[d]), m1 = rint(bx.r[4 * d] * hi[d]);
				if ((m0 == m1) && ok) c |= (CLS_UNIFORM_X << d);
				sh[d] = bx.b[4 * d] * m0;
				const double ilo = tb[12 * (size_t)IJ.x + 6 + d], ihi = tb[12 * (size_t)IJ.x + 9 + d];
				const double jlo = tb[12 * (size_t)IJ.y + 6 + d], jhi = tb[12 * (size_t)IJ.y + 9 + d];
				cmax = fmax(cmax, fmax(fabs(ilo), fabs(ihi)) + fmax(fabs(jlo), fabs(jhi)) + fabs(sh[d]) + fabs(bx.b[4 * d]));
			}
			// The pair sweep's fused geometry (kernels_pair.hip: i-atom pre-shifted by the common image, fma) forms the same displacement
			// from the same operands in another order: each component differs from the reference's by at most ~4 ulp of the largest
			// magnitude it subtracts (<= cmax), the squared distance by 2 r sqrt(3) 4 eps cmax + 4 eps r^2, i.e. relative to the cutoff
			// threshold t ~ r^2 by  err = 14 eps cmax / sqrt(t) + 4 eps  (eps = 2^-52).  A band of 1e-9 is a hundred times that for
			// coordinates up to ~3e4 cutoffs from the origin; beyond, the band is "everything" and the sweep takes the reference's form.
			const double tmin = fmin(bx.t_lj, bx.t_es);
			const double err = (tmin > 0.0) ? (14.0 * cmax / sqrt(tmin) + 4.0) * 2.220446049250313e-16 : 1.0;
			if (ok && err * 100.0 < 1e-9) band = 1e-9;
		} else {
			double img[3]; // (lo, hi: extreme differences of the raw fractional coordinate p)
			for (int p = 0; p < 3; ++p) {
				const double m0 = rint(lo[p]);
				const bool uni = ok && (lo[p] - m0 > -0.5 + 1e-9) && (hi[p] - m0 < 0.5 - 1e-9);
				if (uni) c |= (CLS_UNIFORM_X << p);
				img[p] = uni ? m0 : 0.0;
			}
			// B^T img over the lattice directions with a common index, summed as the reference sums it (x + 0.0 = x: with all three common
			// this IS the reference's translation, which is what the pair sweep needs -- it uses a skewed tile pair's shift only then; the
			// Jacobi walk, values only, takes the partial sums too and rounds the remaining indices per pair)
			for (int p = 0; p < 3; ++p) sh[p] = ((bx.b[p] * img[0]) + bx.b[3 + p] * img[1]) + bx.b[6 + p] * img[2];
		}
		tp_shift[t] = make_double4(sh[0], sh[1], sh[2], band);
	}
	cls[t] = c;
}

void launch_tile_classes(hipStream_t st, const AtomsDev &at, const Box &bx, const int2 *tile_pairs, int n_tile_pairs, double polar_damp,
                         double *tile_bounds, int *cls, double4 *tp_shift, const double origin_f[3], double thole_far_x) {
	const double tmax = (bx.t_lj > bx.t_es) ? bx.t_lj : bx.t_es;
	const double thr_cut2 = tmax * (1.0 + 1e-9);
	double thr_far2 = 0.0;
	if (polar_damp > 0.0) {
		const double rf = thole_far_x / polar_damp;
		thr_far2 = rf * rf * (1.0 + 1e-9);
	}
	CellBounds cell{{0, 0, 0}, 0.0};
	if (!bx.ortho) { // plane spacings 1 / |R_d| and the smallest eigenvalue of the metric B B^T (Jacobi rotations on the 3 x 3 matrix)
		for (int d = 0; d < 3; ++d) cell.plane[d] = 1.0 / std::sqrt(bx.r[d] * bx.r[d] + bx.r[3 + d] * bx.r[3 + d] + bx.r[6 + d] * bx.r[6 + d]);
		double G[3][3];
		for (int i = 0; i < 3; ++i)
			for (int j = 0; j < 3; ++j) G[i][j] = bx.b[3 * i] * bx.b[3 * j] + bx.b[3 * i + 1] * bx.b[3 * j + 1] + bx.b[3 * i + 2] * bx.b[3 * j + 2];
		for (int sweep = 0; sweep < 32; ++sweep)
			for (int p = 0; p < 2; ++p)
				for (int q = p + 1; q < 3; ++q) {
					if (std::fabs(G[p][q]) < 1e-300) continue;
					const double th = 0.5 * std::atan2(2.0 * G[p][q], G[q][q] - G[p][p]), cs = std::cos(th), sn = std::sin(th);
					double Rm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
					Rm[p][p] = cs, Rm[q][q] = cs, Rm[p][q] = sn, Rm[q][p] = -sn;
					double T[3][3], N[3][3];
					for (int i = 0; i < 3; ++i)
						for (int j = 0; j < 3; ++j) T[i][j] = G[i][0] * Rm[0][j] + G[i][1] * Rm[1][j] + G[i][2] * Rm[2][j];
					for (int i = 0; i < 3; ++i)
						for (int j = 0; j < 3; ++j) N[i][j] = Rm[0][i] * T[0][j] + Rm[1][i] * T[1][j] + Rm[2][i] * T[2][j];
					for (int i = 0; i < 3; ++i)
						for (int j = 0; j < 3; ++j) G[i][j] = N[i][j];
				}
		cell.lam_min = std::max(0.0, std::min(G[0][0], std::min(G[1][1], G[2][2])) * (1.0 - 1e-9));
	}
	hipLaunchKernelGGL(k_tile_bounds, dim3(at.n_pad / kTile), dim3(kTile), 0, st, at, bx, make_double3(origin_f[0], origin_f[1], origin_f[2]), tile_bounds);
	hipLaunchKernelGGL(k_classify, dim3((n_tile_pairs + 255) / 256), dim3(256), 0, st, tile_bounds, tile_pairs, n_tile_pairs, bx, cell, thr_cut2,
	                   thr_far2, cls, tp_shift);
}

// ------------------------------------------------------------------------------------------------------
// One Jacobi contraction (reference contract_dipoles :3564-3598 over the A matrix of thole_amatrix :2661-2770).
// For the pair (i,j):   F_i -= a mu_j - b d (d.mu_j),   F_j -= a mu_i - b d (d.mu_i),   T = a I - b d(x)d.
// ------------------------------------------------------------------------------------------------------
template <bool ORTHO>
__device__ __forceinline__ void image_vec(const Box &bx, double dx, double dy, double dz, double &ox, double &oy, double &oz) {
	// displacement as VALUES only (no predicate here): the image index is the reference's rint(R d), the back-projection may be fused
	if (ORTHO) {
		ox = fma(-bx.b[0], rint(bx.r[0] * dx), dx);
		oy = fma(-bx.b[4], rint(bx.r[4] * dy), dy);
		oz = fma(-bx.b[8], rint(bx.r[8] * dz), dz);
	} else {
		(void)min_image_sq<false>(bx, dx, dy, dz, ox, oy, oz);
	}
}

// ------------------------------------------------------------------------------------------------------
// Single-launch form of the same contraction (the default): one kernel walks ALL tile pairs, streaming the stored ones
// and recomputing the far ones (wave-uniform branch), so HBM-bound and fp64-bound waves share the CUs.
//   The j-side accumulators meet their atoms in registers, rotated by one lane per step with v_mov_b32_dpp wave_rol:1 (6 VALU issues
//   per step; ds_bpermute_b32 and ds_add_f64 into an LDS image of the j-atoms were measured and dropped: 0.113 against 0.105 ms).
// ------------------------------------------------------------------------------------------------------
// FAR and the uniform-image mask UM (3 bits, one per dimension) are wave-uniform properties of the tile pair: the walk is instantiated
// for each combination so that the inner loop carries no branch.  A set UM bit: in that dimension the periodic image index is the same
// for all 4096 atom pairs (k_classify), the caller has already moved the i-atom by that lattice vector component and the displacement
// is one subtraction.  At the benchmark box 1.9 of the 3 dimensions are uniform on average (all three for 25 % of the tile pairs).
struct HybLds {
	const double *j; // ONE array [7][2 * kTile]: x, y, z, mu_x, mu_y, mu_z, valid -- one base register, compile-time offsets
	double *gx, *gy, *gz;
};
constexpr int kJ2 = 2 * kTile;
struct HybAcc {
	double fx, fy, fz, gx, gy, gz;
};

// one pair step: lane l against j = (l + s) & 63.  The LDS images of the j-tile hold every value twice (slots k and k + 64),
// so the slot is jl = l + s with no wrap and, inside an unrolled round, a compile-time offset from one base address.
// ROT: rotate the j-side accumulators afterwards (not after the last step).
// RECOMP (with !FAR): nothing is stored at all (solver MATRIX_FREE, the store does not fit its budget): the damped tensor of a tile pair
// inside the damping range is rebuilt from the positions with the arithmetic of the pair sweep (thole_amatrix :2731-2757); t.x then
// carries the pair's 0/1 mask (padded slots, the half-counted step 32 of diagonal tiles) and t.y the damping constant lambda.
template <bool ORTHO, bool FAR, int UM, bool ROT, bool RECOMP = false>
__device__ __forceinline__ void hyb_step(const Box &bx, const HybLds &L, const int jl, const int src4, const double pix, const double piy,
                                         const double piz, const double mix, const double miy, const double miz, double2 t, const double padi,
                                         HybAcc &A) {
	double ox, oy, oz;
	const double xj = L.j[jl + 0 * kJ2], yj = L.j[jl + 1 * kJ2], zj = L.j[jl + 2 * kJ2];
	if (ORTHO) { // per dimension: one subtraction when the image index is known for the whole tile pair (UM bit), else sub-mul-rint-fma
		ox = pix - xj;
		oy = piy - yj;
		oz = piz - zj;
		if (!(UM & 1)) ox = fma(-bx.b[0], rint(bx.r[0] * ox), ox);
		if (!(UM & 2)) oy = fma(-bx.b[4], rint(bx.r[4] * oy), oy);
		if (!(UM & 4)) oz = fma(-bx.b[8], rint(bx.r[8] * oz), oz);
	} else {
		image_vec<ORTHO>(bx, pix - xj, piy - yj, piz - zj, ox, oy, oz);
	}
	if (FAR) { // undamped dipole tensor: a = 1/r^3, b = 3/r^5 (damping dropped beyond lambda r = kTholeFarX, kernels.h)
		const double r2 = fma(oz, oz, fma(oy, oy, ox * ox));
		const double ir = fast_rsqrt_1(r2);
		const double ir2 = ir * ir;
		t.x = ir2 * ir;
		if (padi >= 0.0) t.x *= padi * L.j[jl + 6 * kJ2]; // wave-uniform: only tile pairs that touch the padded last tile
		t.y = t.x * (3.0 * ir2);
	} else if (RECOMP) {
		const double mask = t.x, lam = t.y;
		const double r2 = fma(oz, oz, fma(oy, oy, ox * ox));
		const double ir = fast_rsqrt(r2);
		const double r = r2 * ir;
		const double ir3 = ir * ir * ir, ir5 = ir3 * ir * ir;
		const double lr = lam * r;
		const double explr = exp_fast(-lr);
		const double damp1 = fma(-explr, fma(lr, fma(0.5, lr, 1.0), 1.0), 1.0);
		const double damp2 = fma(-explr, (lr * lr) * (lr * (1.0 / 6.0)), damp1);
		const double live = (r2 > 0.0) ? mask * L.j[jl + 6 * kJ2] : 0.0;
		t.x = live * (damp1 * ir3);
		t.y = live * (3.0 * damp2 * ir5);
	}
	const double mjx = L.j[jl + 3 * kJ2], mjy = L.j[jl + 4 * kJ2], mjz = L.j[jl + 5 * kJ2];
	const double dj = t.y * fma(oz, mjz, fma(oy, mjy, ox * mjx));
	const double di = t.y * fma(oz, miz, fma(oy, miy, ox * mix));
	A.fx = fma(-t.x, mjx, fma(dj, ox, A.fx));
	A.fy = fma(-t.x, mjy, fma(dj, oy, A.fy));
	A.fz = fma(-t.x, mjz, fma(dj, oz, A.fz));
	A.gx = fma(-t.x, mix, fma(di, ox, A.gx));
	A.gy = fma(-t.x, miy, fma(di, oy, A.gy));
	A.gz = fma(-t.x, miz, fma(di, oz, A.gz));
	if (ROT) {
		A.gx = rot_from_next(A.gx);
		A.gy = rot_from_next(A.gy);
		A.gz = rot_from_next(A.gz);
	}
}

template <bool ORTHO, int PIPE, bool FAR, int UM>
__device__ __forceinline__ void hyb_walk(const Box &bx, const HybLds &L, const double pix, const double piy, const double piz, const double mix,
                                         const double miy, const double miz, const double2 *__restrict__ abt, const int s_first,
                                         const int n_steps, const int lane, const int src4, const double padi, HybAcc &A, const double lambda = 0.0,
                                         const double vi = 1.0, const bool diag = false) {
	int jb = lane + s_first; // LDS slot of the round's first step
	if (!FAR && abt == nullptr) { // matrix-free: rebuild the damped tensors (RECOMP); step 32 of a diagonal tile counts lanes 0..31 only
		for (int k = 0; k < n_steps; ++k) {
			const int s = s_first + k;
			const double mask = (diag && s == 32 && lane >= 32) ? 0.0 : vi;
			if (k != n_steps - 1) hyb_step<ORTHO, false, UM, true, true>(bx, L, jb + k, src4, pix, piy, piz, mix, miy, miz, make_double2(mask, lambda), padi, A);
			else hyb_step<ORTHO, false, UM, false, true>(bx, L, jb + k, src4, pix, piy, piz, mix, miy, miz, make_double2(mask, lambda), padi, A);
		}
		return;
	}
	if (FAR) {
		for (int kc = 0; kc < n_steps - 4; kc += 4, jb += 4) {
#pragma unroll
			for (int u = 0; u < 4; ++u) hyb_step<ORTHO, true, UM, true>(bx, L, jb + u, src4, pix, piy, piz, mix, miy, miz, make_double2(0, 0), padi, A);
		}
#pragma unroll
		for (int u = 0; u < 3; ++u) hyb_step<ORTHO, true, UM, true>(bx, L, jb + u, src4, pix, piy, piz, mix, miy, miz, make_double2(0, 0), padi, A);
		hyb_step<ORTHO, true, UM, false>(bx, L, jb + 3, src4, pix, piy, piz, mix, miy, miz, make_double2(0, 0), padi, A);
		return;
	}
	// rolling prefetch ring: the (a,b) of step k + PIPE is requested as soon as the registers of step k are consumed, so PIPE
	// loads of 1 KiB per wave stay in flight all the time (latency x bandwidth of HBM needs > 16 MB in flight chip-wide).
	// All rounds but the last prefetch unconditionally (no branch => the ring registers are reused in place); the addresses
	// are one running pointer per round plus compile-time offsets.
	double2 buf[PIPE];
	const double2 *__restrict__ pn = abt + s_first * kTile;
#pragma unroll
	for (int u = 0; u < PIPE; ++u) buf[u] = ld_stream<true>(pn + u * kTile);
	for (int kc = 0; kc < n_steps - PIPE; kc += PIPE, jb += PIPE) {
		pn += PIPE * kTile;
#pragma unroll
		for (int u = 0; u < PIPE; ++u) {
			hyb_step<ORTHO, false, UM, true>(bx, L, jb + u, src4, pix, piy, piz, mix, miy, miz, buf[u], padi, A);
			buf[u] = ld_stream<true>(pn + u * kTile); // refill the slot just consumed: in place, no register copies
			__builtin_amdgcn_sched_barrier(0); // keep the steps in program order (no hoisting of all LDS reads to the top)
		}
	}
#pragma unroll
	for (int u = 0; u < PIPE; ++u) {
		if (u != PIPE - 1) hyb_step<ORTHO, false, UM, true>(bx, L, jb + u, src4, pix, piy, piz, mix, miy, miz, buf[u], padi, A);
		else hyb_step<ORTHO, false, UM, false>(bx, L, jb + u, src4, pix, piy, piz, mix, miy, miz, buf[u], padi, A);
		__builtin_amdgcn_sched_barrier(0);
	}
}

// W waves share one tile pair: wave w walks the steps [w n/W, (w+1) n/W) of the same 64 i-atoms, so a tile pair is W short
// waves on W SIMDs instead of one long one (12 403 long waves on 1024 SIMDs left a quarter of the CU-time idle in the
// tail of the launch).  The W partial sums of every atom meet in LDS and are added in wave order (fixed => reproducible).
template <bool ORTHO, int PIPE, int W>
__device__ __forceinline__ void hyb_block(const AtomsDev &at, const Box &bx, const double *__restrict__ mu, const int2 *__restrict__ tile_pairs,
                                          const int *__restrict__ cls, const double4 *__restrict__ tp_shift, const double2 *__restrict__ ab,
                                          double *__restrict__ part /*[nt][n_pad][3]*/, const int tp, const double lambda = 0.0) {
	static_assert((32 / W) % PIPE == 0, "a wave's share of a diagonal tile (32 / W steps) must be whole rounds of PIPE");
	__shared__ double s_j[7 * kJ2];
	__shared__ double s_g[W][3][kTile];
	__shared__ double s_f[W][3][kTile];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int2 IJ = tile_pairs[tp];
	const bool diag = (IJ.x == IJ.y);
	const int i = IJ.x * kTile + lane;
	const int j0 = IJ.y * kTile;
	const int src4 = ((lane + 1) & 63) * 4;

	const double4 pi = at.xyzq[i];
	const int c = cls[tp];
	const int um = (ORTHO && tp_shift) ? ((c / CLS_UNIFORM_X) & 7) : 0; // dimensions with one image index for the whole tile pair
	const double mix = mu[3 * (size_t)i], miy = mu[3 * (size_t)i + 1], miz = mu[3 * (size_t)i + 2];
	if (w == 0) {
		const double4 pj = at.xyzq[j0 + lane];
		const double vals[7] = {pj.x, pj.y, pj.z, mu[3 * (size_t)(j0 + lane)], mu[3 * (size_t)(j0 + lane) + 1], mu[3 * (size_t)(j0 + lane) + 2],
		                        (at.mf[j0 + lane].y & AF_PAD) ? 0.0 : 1.0};
#pragma unroll
		for (int c = 0; c < 7; ++c) s_j[c * kJ2 + lane] = s_j[c * kJ2 + lane + kTile] = vals[c]; // twice: slot l + s never wraps (hyb_step)
	}
	__syncthreads();
	const bool far = (c & CLS_THOLE_FAR) != 0; // wave-uniform: beyond the damping range, nothing was stored
	const double vi = (at.mf[i].y & AF_PAD) ? 0.0 : 1.0;
	const bool has_pad = (at.n != at.n_pad) && (IJ.y == at.n_pad / kTile - 1); // only the last tile holds padding slots (I <= J)

	const double2 *__restrict__ abt = ab ? ab + (size_t)tp * (kTile * kTile) + lane : nullptr; // null: matrix-free (nothing stored)
	HybAcc A = {0, 0, 0, 0, 0, 0};
	const HybLds L = {s_j, &s_g[0][0][0], &s_g[0][1][0], &s_g[0][2][0]};
	const double padi = has_pad ? vi : -1.0;
	// 64 steps (off-diagonal, s = 0..63) or 32 steps (diagonal, s = 1..32), W equal shares of whole PIPE rounds
	const int n_steps = (diag ? 32 : 64) / W;
	const int s_first = (diag ? 1 : 0) + w * n_steps;
#define MPMC_WALK(F, U) hyb_walk<ORTHO, PIPE, F, U>(bx, L, qx, qy, qz, mix, miy, miz, abt, s_first, n_steps, lane, src4, padi, A, lambda, vi, diag)
#define MPMC_WALK_UM(F)                                                       \
	switch (um) {                                                             \
	case 1: MPMC_WALK(F, 1); break;                                           \
	case 2: MPMC_WALK(F, 2); break;                                           \
	case 3: MPMC_WALK(F, 3); break;                                           \
	case 4: MPMC_WALK(F, 4); break;                                           \
	case 5: MPMC_WALK(F, 5); break;                                           \
	case 6: MPMC_WALK(F, 6); break;                                           \
	case 7: MPMC_WALK(F, 7); break;                                           \
	default: MPMC_WALK(F, 0); break;                                          \
	}
	double qx = pi.x, qy = pi.y, qz = pi.z;
	if (um) { // the i-atom moves by the common lattice vector of the uniform dimensions, once
		const double4 sh = tp_shift[tp];
		if (um & 1) qx -= sh.x;
		if (um & 2) qy -= sh.y;
		if (um & 4) qz -= sh.z;
	}
	if (ORTHO) {
		if (far) { MPMC_WALK_UM(true) } else { MPMC_WALK_UM(false) }
	} else {
		if (far) MPMC_WALK(true, 0);
		else MPMC_WALK(false, 0);
	}
#undef MPMC_WALK_UM
#undef MPMC_WALK
	{ // park the rotated accumulators at their atoms' LDS slots
		const int jl_last = (lane + s_first + n_steps - 1) & 63;
		s_g[w][0][jl_last] = A.gx;
		s_g[w][1][jl_last] = A.gy;
		s_g[w][2][jl_last] = A.gz;
	}
	if (W > 1) {
		s_f[w][0][lane] = A.fx;
		s_f[w][1][lane] = A.fy;
		s_f[w][2][lane] = A.fz;
	}
	__syncthreads();
	if (w != 0) return;
	double f[3] = {A.fx, A.fy, A.fz}, g[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		g[d] = s_g[0][d][lane];
#pragma unroll
		for (int k = 1; k < W; ++k) {
			f[d] += s_f[k][d][lane];
			g[d] += s_g[k][d][lane];
		}
	}
	const int nt_pad3 = at.n_pad * 3;
	if (diag) {
		double *o = part + (size_t)IJ.x * nt_pad3 + 3 * (size_t)i;
		o[0] = f[0] + g[0];
		o[1] = f[1] + g[1];
		o[2] = f[2] + g[2];
	} else {
		double *oi = part + (size_t)IJ.y * nt_pad3 + 3 * (size_t)i;
		oi[0] = f[0];
		oi[1] = f[1];
		oi[2] = f[2];
		double *oj = part + (size_t)IJ.x * nt_pad3 + 3 * (size_t)(j0 + lane);
		oj[0] = g[0];
		oj[1] = g[1];
		oj[2] = g[2];
	}
}

template <bool ORTHO, int PIPE = 8, int W = 1>
__global__ __launch_bounds__(64 * W) void k_dipole_iter_hybrid(AtomsDev at, Box bx, const double *__restrict__ mu,
                                                                const int2 *__restrict__ tile_pairs, const int *__restrict__ cls,
                                                                const double4 *__restrict__ tp_shift, const double2 *__restrict__ ab,
                                                                double *__restrict__ part, double lambda, const int *__restrict__ converged) {
	if (converged && *converged != 0) return; // an iteration enqueued ahead of the verdict of a precision-terminated solve
	hyb_block<ORTHO, PIPE, W>(at, bx, mu, tile_pairs, cls, tp_shift, ab, part, blockIdx.x, lambda);
}

void launch_dipole_iter_hybrid(hipStream_t st, const AtomsDev &at, const Box &bx, const double *mu, const int2 *tile_pairs,
                               const int *cls, const double4 *tp_shift, int n_tile_pairs, const double2 *ab, double *part, double polar_damp,
                               const int *converged) {
	dim3 grid(n_tile_pairs);
	if (bx.ortho) hipLaunchKernelGGL((k_dipole_iter_hybrid<true>), grid, dim3(kTile), 0, st, at, bx, mu, tile_pairs, cls, tp_shift, ab, part, polar_damp, converged);
	else hipLaunchKernelGGL((k_dipole_iter_hybrid<false>), grid, dim3(kTile), 0, st, at, bx, mu, tile_pairs, cls, tp_shift, ab, part, polar_damp, converged);
}

} // namespace mpmc
