// kernels_pair.hip -- the pair sweep of the production path (any cell, Ewald electrostatics), rebuilt for VALU issue.
//
// Same work as k_pair_fused (kernels_sym.hip): lj() (src/System.Energy.cpp:897-993), the erfc part of coulombic_real() (:1484-1510), the
// real-space static field real_term() (:2900-2940, both atoms of a pair), the in-cutoff pair counts, and the Thole tensor store
// (thole_amatrix :2694-2767 as 16 B per unordered pair) -- every unordered pair once, one wave per tile pair of 64 x 64 atoms, lane l
// owning i-atom l, the j-side field accumulators rotating by one lane per step.  What changed is the instruction stream: the generic
// kernel issues 160 VALU instructions per pair of which 97 are fp64 arithmetic (profiles/r02_pmc_stalls.txt), this one ~80:
//
//  * erfc and the Gaussian come from a TABLE in LDS instead of a degree-20 polynomial plus a range-reduced exp: 512 pieces of width
//    1/128 in x = alpha r, per piece a degree-5 interpolant p of erfc(x); the Gaussian of the field factor is its derivative
//    (exp(-x^2) = -(sqrt(pi) / 2) erfc'(x)), so erfc + 2 x / sqrt(pi) exp(-x^2) = p - (x / H) p' costs one Horner pass for both and one fma.
//    Three ds_read_b128 gathers and 11 (energy) / 16 (energy and field) VALU instructions replace 57 (rounds 2-3: erfcx to degree 4 times
//    G_k exp(t), 22); relative error of erfc 3e-14 over [0, 4), of the field factor 1e-13 below x = 2 and 3e-11 at 4
//    (tools/fit_erfc_table.py, tests/test_erfc_table.py).  The table is why a workgroup is four waves: they share it;
//  * the four waves of a workgroup take four tile pairs (I0 .. I0+3, J) behind ONE j-tile image in LDS, stored as three double2
//    arrays (x,y | z,q | sigma/2, 2 sqrt(eps)) and read with three ds_read_b128 per step (twice over, so that l + s never wraps);
//  * per dimension with a tile-pair-wide periodic image (CLS_UNIFORM_*, k_classify) the displacement is  (x_i - x_j) - B img : two
//    subtractions that round exactly like the reference's  d - B rint(R d)  (same operands, same order), instead of five operations.
//    The squared distance is still  ((dx^2) + dy^2) + dz^2  unfused: pair inclusion stays bit-exact (pair_math.h);
//  * three grades of exclusion logic, chosen per tile pair (wave-uniform): MODE 0 none at all (plain atoms, every molecule one atom);
//    MODE 1 "same molecule" (plain atoms); MODE 2 the flag words of pair_exclusions for tile pairs with frozen / chargeless / sigma- or
//    epsilon-less atoms (masks by and / or of the two flag words, ~11 VALU instructions more).  Only what lj_mix treats specially --
//    sigma < 0 ("attractive only") and dispersion coefficients -- is left to k_pair_fused, which the host launches on the list of those
//    tile pairs (launch_pair_fused ... tp_list);
//  * in-cutoff counts are popcounts of the execution mask on the scalar unit, not per-lane counters; LJ / Coulomb / field code is
//    spelled with fma where the reference's rounding does not decide a predicate.
#include "kernels.h"
#include "device_math.h"
#include "erfc_table.inc"

namespace mpmc {

constexpr int kSweepWaves = 4;
constexpr int kSpecialAtom = AF_FROZEN | AF_NULL_RD | AF_HAS_DISP | AF_NEG_SIGMA | AF_ZERO_SIGMA | AF_ZERO_Q; // (what pair_flags / lj_mix look at)
constexpr int kUnmaskable = kAtomFlagsMixing; // (pair_math.h) these change the MIXING (lj_mix), not just the masks: the generic kernel keeps them

__device__ __forceinline__ int sweep_tp_index(int I, int J, int nt) { return I * nt - (I * (I - 1)) / 2 + (J - I); }

typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr double kSweepEsScale = 0.5, kSweepFieldScale = 0.125; // the walk works with Y = 2 / r: its Coulomb sum carries 2 / r, its field sums 8 / r^3

struct SweepI { // the i-atom a lane owns
	double x, y, z, q, hs, e2; // position, charge, sigma / 4 (half of the mixing rule's term: the walk multiplies by 2 / r), 2 sqrt(epsilon)
	double xs, ys, zs;         // position minus the tile pair's common image translation (used per uniform dimension)
	int mol, fl;
};
struct SweepAcc {
	double e_lj, e_re;
	double ex, ey, ez; // field on the i-atom
	double gx, gy, gz; // field on the j-atom this lane is paired with (rotates)
};

// what follows the geometry of a step: exclusions, the Thole tensor store, the inclusion predicates and their counts, LJ / Coulomb / field
template <bool FIELD, int MODE, bool PAD>
__device__ __forceinline__ void sweep_finish(const double2 *__restrict__ s_se, const double2 *__restrict__ s_tab, const int jl, const int lane,
                                             const SweepI &I, const double2 zq, const int2 mfj, const bool ok, const double ox, const double oy,
                                             const double oz, const double ri2, const unsigned long long m_cut_lj, const unsigned long long m_cut_es, const Box &bx,
                                             const PairSweepParams &pp, const bool half, const bool store, const rsrc_t ab_rsrc, const int ab_soffset,
                                             SweepAcc &A, int &n_lj, int &n_es) {
	// Y = 2 / r: hardware seed and one Newton step in its product form y (3 - x y^2) -- one instruction less than the step towards 1 / r,
	// and every use below absorbs the power of two exactly: the atoms' sigma / 2 arrive as sigma / 4, alpha / H and lambda as halves, the
	// Coulomb and field sums are scaled once per wave behind the walk (kSweepEsScale, kSweepFieldScale), the Thole factors carry 1 / 8 and
	// 3 / 32 in their constants.  ~2e-14 relative like fast_rsqrt_1.
	const double y0 = __builtin_amdgcn_rsq(ri2);
	const double Y = y0 * fma(-(ri2 * y0), y0, 3.0);
	const double R2 = ri2 * Y; // 2 r
	const double Y2 = Y * Y;   // 4 / r^2
	// pair_exclusions (src/System.cpp:1035-1197) for what this grade of tile pair can hold:
	//   frozen pair (both frozen): no LJ, no Coulomb, no field -- only the Thole tensor;   excl_rd / excl_es: same molecule, or an atom without
	//   sigma / epsilon (rd) resp. without charge (es);   no_field: both charges zero (real_term :2916).  MODE 0 / 1 know none / only "same molecule".
	const bool intra = (MODE >= 1) && (I.mol == mfj.x);
	bool frozen = false, excl_rd = intra, excl_es = intra, no_field = false;
	if (MODE == 2) {
		const int any = I.fl | mfj.y, both = I.fl & mfj.y;
		frozen = (both & AF_FROZEN) != 0;
		excl_rd = intra || (any & AF_NULL_RD) != 0;
		excl_es = intra || (any & AF_ZERO_Q) != 0;
		no_field = (both & AF_ZERO_Q) != 0 || (ri2 == 0.0);
	}

	if (store) { // thole_amatrix couples every pair: no cutoff, no exclusions, frozen included (:2694-2767)
		const double Y3 = Y2 * Y, Y5 = (Y2 * Y2) * Y; // 8 / r^3, 32 / r^5
		const double lr = pp.polar_damp_half * R2;
		double damp1s = 0.125, damp2s = 0.09375; // damp1 / 8 and 3 damp2 / 32 (the powers of two of Y^3 and Y^5 taken out)
		if (__any(lr < pp.thole_far_x)) { // wave-uniform: beyond lambda r = kTholeFarX the damping is dropped (as in CLS_THOLE_FAR)
			const double explr = exp_fast(-lr);
			damp1s = fma(-explr, fma(lr, fma(0.0625, lr, 0.125), 0.125), 0.125); // (1 - e^{-lr} (lr^2/2 + lr + 1)) / 8
			damp2s = fma(-explr, (lr * lr) * (lr * 0.015625), 0.75 * damp1s);     // 3 (damp1 - e^{-lr} lr^3/6) / 32
		}
		double ta = damp1s * Y3, tb = damp2s * Y5;
		if (PAD || half || MODE == 2) {
			// (coincident sites -- a dummy site on top of an atom -- exist among special atoms: the reference's MAXVALUE guard (:2704-2705) times
			// its vanishing damping factors is 0)
			const bool live = ok && !(MODE == 2 && ri2 == 0.0);
			ta = live ? ta : 0.0;
			tb = live ? tb : 0.0;
		}
		// one 16-byte store per lane through a buffer descriptor of the tile pair's block: the row advances in a SCALAR offset (a per-lane 64-bit
		// pointer cost a v_lshl_add_u64 in every step of every tile pair, stored or not)
		__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, make_double2(ta, tb)), ab_rsrc, lane * 16, ab_soffset, 0);
	}

	// the inclusion predicates, and the counts of the pairs they admit: wave-level masks and popcounts (scalar unit), taken OUTSIDE the
	// divergent region.  t_es <= t_lj (pair_math.h Box), so the Coulomb pairs are a subset of the LJ shell.
	// (m_cut_*: the cutoff predicates of the real pairs of this step as wave masks, from sweep_step; masks combine on the scalar unit, and in
	// MODE 0 -- no exclusions -- they ARE the masks of the admitted pairs)
	const unsigned long long m_in_lj = m_cut_lj & ~__builtin_amdgcn_ballot_w64(frozen); // rimg - 1e-12 < rc  (lj :934)
	const unsigned long long m_in_es = m_cut_es & ~__builtin_amdgcn_ballot_w64(frozen); // (implies in_lj)  !(rimg > rc)  (coulombic_real :1490, real_term :2917)
	const unsigned long long m_lj_on = m_in_lj & ~__builtin_amdgcn_ballot_w64(excl_rd), m_es_on = m_in_es & ~__builtin_amdgcn_ballot_w64(excl_es);
	n_lj += __popcll(m_lj_on);
	n_es += __popcll(m_es_on);
	const bool in_lj = __builtin_amdgcn_inverse_ballot_w64(m_in_lj), in_es = __builtin_amdgcn_inverse_ballot_w64(m_in_es);
	const bool lj_on = __builtin_amdgcn_inverse_ballot_w64(m_lj_on), es_on = __builtin_amdgcn_inverse_ballot_w64(m_es_on);
	asm volatile("" : "+s"(n_lj), "+s"(n_es)); // (the sums are wanted HERE, in scalar registers: sunk behind the divergent region the masks make a round trip through VGPRs)
	if (in_lj) {
		const double2 se = s_se[jl];
		if (lj_on) {
			const double sig = I.hs + se.x; // Lorentz-Berthelot: (s_i + s_j)/2, here as its half ((s_i + s_j)/4 times Y = 2/r); of 4 sqrt(e_i e_j) the j-atom's factor here, the lane's own at the end of the walk
			const double sr = sig * Y;
			const double s3 = (sr * sr) * sr;
			const double s6 = s3 * s3;
			A.e_lj = fma(se.y, fma(s6, s6, -s6), A.e_lj); // 4 eps (s^12 - s^6)  (:965-993)
		}
		const bool fld_on = FIELD && in_es && !(MODE == 2 && no_field);
		if (es_on || fld_on) {
			const double xs = R2 * pp.alpha_scaled_half; // alpha r / H: the integer part is the piece, the fraction the argument of its polynomial
			const int it = (int)xs;
			const double dd = __builtin_amdgcn_fract(xs);
			const double2 c01 = s_tab[it], c23 = s_tab[MPMC_ERFTAB_PIECES + it], c45 = s_tab[2 * MPMC_ERFTAB_PIECES + it];
			// p(dd) = erfc(x); with the field also p'(dd) = -(2 / sqrt(pi)) exp(-x^2) / 128 from the same coefficients (Horner's pass for a
			// polynomial and its derivative)
			double b = fma(c45.y, dd, c45.x), d1 = 0.0;
			if (FIELD) d1 = fma(c45.y, dd, b);
			b = fma(b, dd, c23.y);
			if (FIELD) d1 = fma(d1, dd, b);
			b = fma(b, dd, c23.x);
			if (FIELD) d1 = fma(d1, dd, b);
			b = fma(b, dd, c01.y);
			if (FIELD) d1 = fma(d1, dd, b);
			const double erfc_x = fma(b, dd, c01.x);
			if (es_on) A.e_re = fma(zq.y * erfc_x, Y, A.e_re); // 2 q_j erfc(alpha r) / r; the lane's own q_i and the half at the end of the walk
			if (fld_on) { // real_term :2919-2934: (2 alpha r / sqrt(pi) exp(-alpha^2 r^2) + erfc) / r^3, erf form (= that - 1) for es_excluded pairs
				double B = fma(-xs, d1, erfc_x); // erfc(x) + 2 x / sqrt(pi) exp(-x^2) = p - (x / H) p'
				if (MODE >= 1) B -= excl_es ? 1.0 : 0.0;
				const double fac = B * (Y2 * Y); // 8 B / r^3 (kSweepFieldScale behind the walk)
				const double fj = fac * zq.y, fi = fac * I.q;
				A.ex = fma(fj, ox, A.ex);
				A.ey = fma(fj, oy, A.ey);
				A.ez = fma(fj, oz, A.ez);
				A.gx = fma(-fi, ox, A.gx);
				A.gy = fma(-fi, oy, A.gy);
				A.gz = fma(-fi, oz, A.gz);
			}
		}
	}
}

// one step: lane l against j = slot jl of the (doubled) j-tile image
template <int UM, bool FIELD, int MODE, bool PAD, bool TRI>
__device__ __forceinline__ void sweep_step(const double2 *__restrict__ s_xy, const double2 *__restrict__ s_zq, const double2 *__restrict__ s_se,
                                           const int2 *__restrict__ s_mf, const double2 *__restrict__ s_tab, const int jl, const int lane,
                                           const SweepI &I, const double shx, const double shy, const double shz, const double t_lo,
                                           const double t_hi, const Box &bx, const PairSweepParams &pp, const bool half, const bool i_real, const bool store,
                                           const rsrc_t ab_rsrc, const int ab_soffset /*this step's row of the tile pair's block of the tensor store: byte offset, scalar*/,
                                           SweepAcc &A, int &n_lj, int &n_es) {
	const double2 xy = s_xy[jl], zq = s_zq[jl];
	int2 mfj = make_int2(0, 0);
	if (MODE >= 1 || PAD) mfj = s_mf[jl]; // molecule id, flags (padding slots: negative ids, AF_PAD)
	// lanes that form a real pair at this step: all of them, except in tile pairs with padding slots (PAD: the last tile) and in the
	// closing half step of a diagonal tile pair (half, wave-uniform: sweep_walk spells that step out)
	bool ok = true;
	if (PAD) ok = i_real && (mfj.x >= 0);
	if (half) ok = ok && (lane < 32);
	// minimum image (src/System.cpp:1228-1246), diagonal cell: d - B rint(R d); with a tile-pair-wide image index B rint(R d) is shx
	// general cell (TRI): the translation B^T img mixes the dimensions, so a tile pair has one image for all three (UM = 7, k_classify) or
	// every pair takes the reference's full form rint(R d), B^T img
	double ox, oy, oz, ri2;
	// the reference's two cutoff predicates -- rimg - 1e-12 < rc (lj :934), !(rimg > rc) (coulombic_real :1490, real_term :2917) -- of the
	// real pairs of this step, as WAVE MASKS in scalar registers (two predicates that were merged behind a branch as per-lane booleans would travel
	// through vector registers; a mask is a scalar value and merges for free)
	unsigned long long m_lj, m_es;
	auto exact_geometry = [&]() { // the reference's operands in the reference's order, unfused: ri2 decides pair inclusion bit for bit
		const double dx = I.x - xy.x, dy = I.y - xy.y, dz = I.z - zq.x;
		if (TRI && UM == 0) {
			ri2 = min_image_sq<false>(bx, dx, dy, dz, ox, oy, oz);
		} else {
			if (UM & 1) ox = dx - shx;
			else ox = dx - bx.b[0] * rint(bx.r[0] * dx);
			if (UM & 2) oy = dy - shy;
			else oy = dy - bx.b[4] * rint(bx.r[4] * dy);
			if (UM & 4) oz = dz - shz;
			else oz = dz - bx.b[8] * rint(bx.r[8] * dz);
			ri2 = ((ox * ox) + oy * oy) + oz * oz;
		}
		m_lj = __builtin_amdgcn_ballot_w64(ok && ri2 <= bx.t_lj);
		m_es = __builtin_amdgcn_ballot_w64(ok && ri2 <= bx.t_es);
	};
	if (TRI) { // (skewed cells keep the reference's form throughout)
		exact_geometry();
	} else {
		// FAST form (round 4): the i-atom carries the tile pair's common image already (I.xs = x_i - shx per uniform dimension, once per
		// wave), the remaining dimensions and the squared distance are fused: 6 instead of 11 instructions with three uniform dimensions.
		// Its ri2 differs from the reference's by a few ulp of the COORDINATES (relative 1e-14 here), which can only change a cutoff
		// predicate inside the band t_lo .. t_hi = t_es (1 - 1e-9) .. t_lj (1 + 1e-9) around the thresholds (k_classify checks per tile
		// pair that the coordinates are small enough for that and widens the band to "everything" if not: tp_shift.w).  The two compares
		// the predicates need anyway are taken against the band's edges: inside t_lo both predicates hold, beyond t_hi neither does, and a
		// step with a lane in between (about one pair in 1e9) redoes its geometry the reference's way -- pair inclusion stays bit-exact.
		if (UM & 1) ox = I.xs - xy.x;
		else {
			const double dx = I.x - xy.x;
			ox = fma(-bx.b[0], rint(bx.r[0] * dx), dx);
		}
		if (UM & 2) oy = I.ys - xy.y;
		else {
			const double dy = I.y - xy.y;
			oy = fma(-bx.b[4], rint(bx.r[4] * dy), dy);
		}
		if (UM & 4) oz = I.zs - zq.x;
		else {
			const double dz = I.z - zq.x;
			oz = fma(-bx.b[8], rint(bx.r[8] * dz), dz);
		}
		ri2 = fma(oz, oz, fma(oy, oy, ox * ox));
		m_lj = m_es = __builtin_amdgcn_ballot_w64(ok && ri2 <= t_lo);
		// (t_lo < t_hi: some lane lies in between exactly when the two masks differ)
		if (__builtin_expect(__builtin_amdgcn_ballot_w64(ok && ri2 <= t_hi) != m_lj, 0)) {
			asm volatile("" ::: "memory"); // (a real branch: the compiler must not run the reference's form speculatively and select)
			exact_geometry();
		}
	}
	sweep_finish<FIELD, MODE, PAD>(s_se, s_tab, jl, lane, I, zq, mfj, ok, ox, oy, oz, ri2, m_lj, m_es, bx, pp, half, store, ab_rsrc, ab_soffset, A, n_lj, n_es);
}

template <int UM, bool FIELD, int MODE, bool PAD, bool TRI = false>
__device__ __forceinline__ void sweep_walk(const double2 *__restrict__ s_xy, const double2 *__restrict__ s_zq, const double2 *__restrict__ s_se,
                                        const int2 *__restrict__ s_mf, const double2 *__restrict__ s_tab, const int lane, const SweepI &I,
                                        const double shx, const double shy, const double shz, const double t_lo, const double t_hi,
                                        const Box &bx, const PairSweepParams &pp, const bool tail_half, const int s_begin, const int s_end, const bool i_real, const bool store,
                                        const rsrc_t ab_rsrc, SweepAcc &A, int &n_lj, int &n_es) {
	// diagonal tile pair: s = 1..32, the last one with lanes 0..31 only (each pair once); off-diagonal: s = 0..63.  A wave walks the
	// steps [s_begin, s_end) of that sequence: all of them, or one half of them when two waves share a tile pair (pp.split);
	// tail_half: this wave's last step is the diagonal tile pair's closing half step
	// (the closing step is spelled out: inside the loop "half" is a compile-time false and costs nothing)
	for (int s = s_begin; s < s_end - 1; ++s) {
		sweep_step<UM, FIELD, MODE, PAD, TRI>(s_xy, s_zq, s_se, s_mf, s_tab, lane + s, lane, I, shx, shy, shz, t_lo, t_hi, bx, pp, false, i_real, store,
		                                  ab_rsrc, s * (kTile * 16), A, n_lj, n_es);
		if (FIELD) {
			A.gx = rot_from_next(A.gx);
			A.gy = rot_from_next(A.gy);
			A.gz = rot_from_next(A.gz);
		}
	}
	const int s = s_end - 1;
	sweep_step<UM, FIELD, MODE, PAD, TRI>(s_xy, s_zq, s_se, s_mf, s_tab, lane + s, lane, I, shx, shy, shz, t_lo, t_hi, bx, pp, tail_half, i_real, store,
	                                  ab_rsrc, s * (kTile * 16), A, n_lj, n_es);
}

// blocks: { J, I0 } -- the workgroup's waves take the tile pairs (I0 + w, J), w = 0..3, as far as I0 + w <= J
// (ORTHO: the orthorhombic instantiation reads the diagonals of the cell and its inverse only, which keeps the rest of the Box out of its
// scalar registers -- spilled SGPRs are v_writelane / v_readlane on the VALU)
template <bool FIELD, bool INTRA, bool ORTHO, int SPLIT>
__global__ __launch_bounds__(64 * kSweepWaves) void k_pair_sweep(AtomsDev at, Box bx, PairSweepParams pp, const int2 *__restrict__ blocks,
                                                                 const int *__restrict__ cls, const double4 *__restrict__ tp_shift,
                                                                 const double2 *__restrict__ erf_tab, double *__restrict__ block_part,
                                                                 int *__restrict__ block_cnt, double *__restrict__ fpart /*[nt][n_pad][3]*/,
                                                                 double2 *__restrict__ ab) {
	__shared__ double2 s_tab[3 * MPMC_ERFTAB_PIECES];
	__shared__ double2 s_xy[2 * kTile], s_zq[2 * kTile], s_se[2 * kTile];
	__shared__ int2 s_mf[2 * kTile];
	__shared__ int s_jflags[3];
	const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// pp.split: the workgroup takes TWO tile pairs (I0, J), (I0 + 1, J) and two waves share each -- wave w walks half (w >> 1) of the
	// steps of tile pair (w & 1); the halves meet in LDS behind the walk.  Half-length workgroups: a lone launch drains on a tail half as
	// long (CUs busy 77 % -> ~90 % of the launch at 10 000 atoms).  The table stays { J, I0 in steps of 4 }: two workgroups per entry.
	// SPLIT 0: never; 1: every entry; 2 (default, round 5): the LAST entries of the table only -- workgroups [0, n_main) take whole entries, the
	// ones behind them half entries: the launch ends on units half as long (a lone launch drains on its last units: 17 % of it at 10 000
	// atoms), while most of the work keeps the cheaper whole form.  Which entries are halved is a function of the table alone, so an
	// evaluation gives the same bits alone and inside an ensemble.
	const int n_main = SPLIT == 1 ? 0 : pp.n_main;
	const bool split = SPLIT == 1 ? true : (SPLIT == 0 ? false : ((int)blockIdx.x >= n_main));
	const int b_half = (int)blockIdx.x - n_main; // (split workgroups: two per entry)
	const int2 blk = blocks[split ? n_main + (b_half >> 1) : (int)blockIdx.x];
	const int pw = split ? (w & 1) : w, half = split ? (w >> 1) : 0;
	const int J = __builtin_amdgcn_readfirstlane(blk.x), I = __builtin_amdgcn_readfirstlane(blk.y) + (split ? 2 * (b_half & 1) : 0) + pw;
	const int j0 = J * kTile;
#pragma unroll
	for (int k = 0; k < (3 * MPMC_ERFTAB_PIECES) / (64 * kSweepWaves); ++k) s_tab[threadIdx.x + k * 64 * kSweepWaves] = erf_tab[threadIdx.x + k * 64 * kSweepWaves];
	if (w == 0) { // the j-tile, every value twice (slot l + s never wraps)
		const double4 pj = at.xyzq[j0 + lane];
		const double2 lj = at.lj[j0 + lane];
		const int2 mj = at.mf[j0 + lane];
		s_xy[lane] = s_xy[lane + kTile] = make_double2(pj.x, pj.y);
		s_zq[lane] = s_zq[lane + kTile] = make_double2(pj.z, pj.w);
		s_se[lane] = s_se[lane + kTile] = make_double2(0.25 * lj.x, 2.0 * lj.y); // sigma / 4, 2 sqrt(eps)
		s_mf[lane] = s_mf[lane + kTile] = mj; // (padding slots carry negative ids and AF_PAD)
		const bool padj = (mj.y & AF_PAD) != 0;
		const int any_unmask = __any(!padj && (mj.y & kUnmaskable) != 0), any_pad = __any(padj), any_spec = __any(!padj && (mj.y & kSpecialAtom) != 0);
		if (lane == 0) {
			s_jflags[0] = any_unmask;
			s_jflags[1] = any_pad;
			s_jflags[2] = any_spec;
		}
	}
	__syncthreads();
	if (!split && I > J) return; // (a j-tile's last workgroup may have fewer tile pairs than waves; with pp.split every wave stays for the barriers)
	const bool have_tp = (I <= J);
	const int nt = pp.nt;
	const int tp = have_tp ? sweep_tp_index(I, J, nt) : 0;
	const int i = (have_tp ? I : J) * kTile + lane;
	const double4 pi = at.xyzq[i];
	const double2 li = at.lj[i];
	const int2 mi = at.mf[i];
	const bool i_real = !(mi.y & AF_PAD);
	// a tile pair with an atom whose flags change the MIXING (sigma < 0, dispersion coefficients) belongs to the generic kernel (the host
	// launches it on exactly these: same predicate); any other special atom (frozen, chargeless, sigma- or epsilon-less) selects the masked grade
	const bool generic_tp = __builtin_amdgcn_readfirstlane(s_jflags[0]) || __any(i_real && (mi.y & kUnmaskable) != 0);
	if (!split && generic_tp) return;
	const bool special = __builtin_amdgcn_readfirstlane(s_jflags[2]) || __any(i_real && (mi.y & kSpecialAtom) != 0);
	const bool pad = __builtin_amdgcn_readfirstlane(s_jflags[1]) || __any(!i_real);
	const bool diag = (I == J);
	const int cl = cls[tp];
	const bool beyond = (cl & CLS_BEYOND_CUTOFF) != 0;
	const bool store = pp.store && !(cl & CLS_THOLE_FAR);
	const int nt_pad3 = at.n_pad * 3;
	// (a tile pair beyond the cutoff whose tensors are stored all the same -- a cutoff shorter than the damping range -- takes the
	// ordinary walk: no pair of it passes a cutoff predicate, by the class's construction)
	const bool nothing = beyond && !store; // nothing to do: publish zeros so that the fixed-shape reductions stay valid
	const bool live = have_tp && !generic_tp; // this wave's tile pair is the sweep's
	if (live && nothing && half == 0) {
		if (FIELD) {
			double *oi = fpart + (size_t)J * nt_pad3 + 3 * (size_t)i;
			double *oj = fpart + (size_t)I * nt_pad3 + 3 * (size_t)(j0 + lane);
			oi[0] = oi[1] = oi[2] = 0.0;
			oj[0] = oj[1] = oj[2] = 0.0;
		}
		if (lane == 0) {
			block_part[2 * (size_t)tp] = 0.0;
			block_part[2 * (size_t)tp + 1] = 0.0;
			block_cnt[2 * (size_t)tp] = 0;
			block_cnt[2 * (size_t)tp + 1] = 0;
		}
	}
	if (!split && nothing) return;
	const bool walk = live && !nothing;
	SweepAcc A = {};
	int n_lj = 0, n_es = 0;
	// steps of this wave: s = 0..63 (diagonal tile pair: 1..32), or one half of them
	const int n_steps = diag ? 32 : 64, s_first = diag ? 1 : 0;
	const int s_begin = s_first + (split ? half * (n_steps / 2) : 0), s_end = s_first + (split ? (half + 1) * (n_steps / 2) : n_steps);
	const bool tail_half = diag && (s_end == 33);
	if (walk) {
		SweepI Ai;
		Ai.x = pi.x, Ai.y = pi.y, Ai.z = pi.z, Ai.q = pi.w;
		Ai.hs = 0.25 * li.x, Ai.e2 = 2.0 * li.y;
		Ai.mol = mi.x;
		Ai.fl = mi.y;
		int um = (pp.have_shift && !pad) ? ((cl / CLS_UNIFORM_X) & 7) : 0; // (the padded tile's pairs take the general image path: one variant)
		if (!ORTHO && um != 7) um = 0; // a skewed cell's translation mixes the components: one common image for all three indices, or the full form
		double shx = 0.0, shy = 0.0, shz = 0.0; // B img of the tile pair's common image, per uniform dimension (wave-uniform: scalar loads)
		double band = 1e30;                     // relative half-width of the fast geometry's band around the cutoff thresholds (1e30: everything)
		if (pp.have_shift) {
			const double4 sh = tp_shift[tp];
			shx = sh.x, shy = sh.y, shz = sh.z;
			if (pp.fast) band = sh.w;
		}
		const double t_lo = bx.t_es - bx.t_es * band, t_hi = bx.t_lj + bx.t_lj * band;
		Ai.xs = Ai.x - shx, Ai.ys = Ai.y - shy, Ai.zs = Ai.z - shz;
		// the tile pair's 64 x 64 block of the tensor store behind a buffer descriptor (64 KiB; raw, 32-bit elements): rows are scalar offsets
		const rsrc_t ab_rsrc = __builtin_amdgcn_make_buffer_rsrc(store ? (void *)(ab + (size_t)tp * (kTile * kTile)) : (void *)nullptr, 0, store ? kTile * kTile * 16 : 0, 0x00027000);
#define MPMC_SWEEP_ARGS s_xy, s_zq, s_se, s_mf, s_tab, lane, Ai, shx, shy, shz, t_lo, t_hi, bx, pp, tail_half, s_begin, s_end, i_real, store, ab_rsrc, A, n_lj, n_es
#define MPMC_SWEEP_UM(MODE)                                                        \
	switch (um) {                                                                  \
	case 0:                                                                        \
		if (!ORTHO) {                                                              \
			if (pad) sweep_walk<0, FIELD, MODE, true, true>(MPMC_SWEEP_ARGS);      \
			else sweep_walk<0, FIELD, MODE, false, true>(MPMC_SWEEP_ARGS);         \
		} else if (pad) sweep_walk<0, FIELD, MODE, true>(MPMC_SWEEP_ARGS);         \
		else sweep_walk<0, FIELD, MODE, false>(MPMC_SWEEP_ARGS);                   \
		break;                                                                     \
	case 1: sweep_walk<1, FIELD, MODE, false>(MPMC_SWEEP_ARGS); break;             \
	case 2: sweep_walk<2, FIELD, MODE, false>(MPMC_SWEEP_ARGS); break;             \
	case 3: sweep_walk<3, FIELD, MODE, false>(MPMC_SWEEP_ARGS); break;             \
	case 4: sweep_walk<4, FIELD, MODE, false>(MPMC_SWEEP_ARGS); break;             \
	case 5: sweep_walk<5, FIELD, MODE, false>(MPMC_SWEEP_ARGS); break;             \
	case 6: sweep_walk<6, FIELD, MODE, false>(MPMC_SWEEP_ARGS); break;             \
	default: sweep_walk<7, FIELD, MODE, false>(MPMC_SWEEP_ARGS); break;            \
	}
		if (special) {
			MPMC_SWEEP_UM(2)
		} else {
			MPMC_SWEEP_UM((INTRA ? 1 : 0))
		}
#undef MPMC_SWEEP_UM
#undef MPMC_SWEEP_ARGS
		A.e_lj *= Ai.e2; // (the i-atom's factors of 4 sqrt(e_i e_j) and q_i q_j, once per wave instead of once per pair)
		A.e_re *= kSweepEsScale * Ai.q;
		if (FIELD) {
			A.ex *= kSweepFieldScale, A.ey *= kSweepFieldScale, A.ez *= kSweepFieldScale;
			A.gx *= kSweepFieldScale, A.gy *= kSweepFieldScale, A.gz *= kSweepFieldScale;
		}
	}
	// the j-atom whose accumulator this lane ended up holding: after step s (no rotation behind the last one) lane l pairs with (l + s) & 63
	const int jown = (lane + s_end - 1) & 63;
	double e_lj = A.e_lj, e_re = A.e_re;
	if (split) {
		// the two halves of a tile pair meet in LDS (the erfc table's space: nobody reads it behind the first barrier): the wave of the second
		// half leaves its i-side sums per lane, its j-side sums per j-atom, its energies per lane and its counts; the wave of the first half
		// adds them to its own -- first half + second half, a fixed order -- and writes the tile pair's outputs as the unsplit kernel does
		double *xch = reinterpret_cast<double *>(s_tab) + (size_t)pw * (9 * kTile);
		__syncthreads();
		if (half == 1 && walk) {
			xch[0 * kTile + lane] = A.ex;
			xch[1 * kTile + lane] = A.ey;
			xch[2 * kTile + lane] = A.ez;
			xch[3 * kTile + jown] = A.gx;
			xch[4 * kTile + jown] = A.gy;
			xch[5 * kTile + jown] = A.gz;
			xch[6 * kTile + lane] = A.e_lj;
			xch[7 * kTile + lane] = A.e_re;
			if (lane == 0) {
				reinterpret_cast<int *>(xch + 8 * kTile)[0] = n_lj;
				reinterpret_cast<int *>(xch + 8 * kTile)[1] = n_es;
			}
		}
		__syncthreads();
		if (half == 1 || !walk) return;
		A.ex += xch[0 * kTile + lane];
		A.ey += xch[1 * kTile + lane];
		A.ez += xch[2 * kTile + lane];
		A.gx += xch[3 * kTile + jown];
		A.gy += xch[4 * kTile + jown];
		A.gz += xch[5 * kTile + jown];
		e_lj += xch[6 * kTile + lane];
		e_re += xch[7 * kTile + lane];
		n_lj += reinterpret_cast<const int *>(xch + 8 * kTile)[0];
		n_es += reinterpret_cast<const int *>(xch + 8 * kTile)[1];
	}

	if (FIELD) {
		if (diag) { // both sides are the same 64 atoms: one slot [I][I atoms]; atom a's j-side sum sits in the lane with jown = a
			double *o = fpart + (size_t)I * nt_pad3 + 3 * (size_t)i;
			const int src = (2 * lane - jown) & 63; // jown(src) = lane  (jown = lane + c  =>  src = lane - c)
			o[0] = A.ex + __shfl(A.gx, src, 64);
			o[1] = A.ey + __shfl(A.gy, src, 64);
			o[2] = A.ez + __shfl(A.gz, src, 64);
		} else {
			double *oi = fpart + (size_t)J * nt_pad3 + 3 * (size_t)i; // i-atoms, contribution of tile J
			oi[0] = A.ex;
			oi[1] = A.ey;
			oi[2] = A.ez;
			double *oj = fpart + (size_t)I * nt_pad3 + 3 * (size_t)(j0 + jown); // j-atoms, contribution of tile I
			oj[0] = A.gx;
			oj[1] = A.gy;
			oj[2] = A.gz;
		}
	}
	e_lj = wave_sum(e_lj), e_re = wave_sum(e_re);
	if (lane == 0) {
		block_part[2 * (size_t)tp] = e_lj;
		block_part[2 * (size_t)tp + 1] = e_re;
		block_cnt[2 * (size_t)tp] = n_lj;
		block_cnt[2 * (size_t)tp + 1] = n_es;
	}
}

int pair_sweep_blocks(int n_tiles, int2 *out) {
	int n = 0;
	for (int J = 0; J < n_tiles; ++J)
		for (int I0 = 0; I0 <= J; I0 += kSweepWaves) {
			if (out) out[n] = make_int2(J, I0);
			n++;
		}
	return n;
}

bool pair_sweep_covers(const Box &bx, const FusedParams &fp, double ewald_alpha) {
	if (!fp.do_es || fp.do_field == 2 || fp.wolf || fp.fh_order || fp.store_only) return false;
	if (fp.do_field == 1 && fp.polar_ewald_alpha != fp.ewald_alpha) return false;
	const double tmax = (bx.t_lj > bx.t_es) ? bx.t_lj : bx.t_es;
	return ewald_alpha * std::sqrt(tmax) * (1.0 + 1e-9) < MPMC_ERFTAB_XMAX; // every in-cutoff pair inside the table
}

void launch_pair_sweep(hipStream_t st, const AtomsDev &at, const Box &bx, const FusedParams &fp, bool intra, const int2 *blocks, int n_blocks,
                       const int *cls, const double4 *tp_shift, const double2 *erf_tab, double *block_part, int *block_cnt, double *fpart, double2 *ab,
                       int split_mode, int n_split_tail, bool fast_geometry, int lds_pad_bytes, int replicas) {
	PairSweepParams pp;
	// workgroups [0, n_main) take whole entries, the rest half entries (two per entry): mode 0 -> all whole, 1 -> all halved, 2 -> the last n_split_tail halved
	const int n_tail = split_mode == 1 ? n_blocks : (split_mode == 2 ? std::min(std::max(n_split_tail, 0), n_blocks) : 0);
	pp.split = split_mode;
	pp.n_main = n_blocks - n_tail;
	pp.alpha_scaled_half = 0.5 * (fp.ewald_alpha * MPMC_ERFTAB_INV_H); // (powers of two: (alpha r) / H and (alpha / 2H) (2r) are the same double)
	pp.polar_damp_half = 0.5 * fp.polar_damp;
	pp.thole_far_x = fp.thole_far_x;
	pp.store = (fp.do_thole && ab) ? 1 : 0;
	pp.nt = at.n_pad / kTile;
	pp.have_shift = tp_shift ? 1 : 0;
	pp.fast = fast_geometry ? 1 : 0; // (the per-tile-pair band comes from k_classify: tp_shift.w)
	dim3 grid(pp.n_main + 2 * n_tail, replicas > 1 ? replicas : 1), block(64 * kSweepWaves); // (replicas: measurement only -- the same work blockIdx.y times)
	const unsigned lds = lds_pad_bytes > 0 ? (unsigned)lds_pad_bytes : 0u; // unused dynamic LDS: caps the workgroups per CU (see evaluate.cpp)
#define MPMC_PS2(F, N, O, S) hipLaunchKernelGGL((k_pair_sweep<F, N, O, S>), grid, block, lds, st, at, bx, pp, blocks, cls, tp_shift, erf_tab, block_part, block_cnt, fpart, ab)
#define MPMC_PS(F, N)                                    \
	do {                                                 \
		if (bx.ortho) {                                  \
			if (n_tail == 0) MPMC_PS2(F, N, true, 0);    \
			else if (n_tail == n_blocks) MPMC_PS2(F, N, true, 1); \
			else MPMC_PS2(F, N, true, 2);                \
		} else {                                         \
			if (n_tail == 0) MPMC_PS2(F, N, false, 0);   \
			else if (n_tail == n_blocks) MPMC_PS2(F, N, false, 1); \
			else MPMC_PS2(F, N, false, 2);               \
		}                                                \
	} while (0)
	if (fp.do_field == 1) {
		if (intra) MPMC_PS(true, true);
		else MPMC_PS(true, false);
	} else {
		if (intra) MPMC_PS(false, true);
		else MPMC_PS(false, false);
	}
#undef MPMC_PS
#undef MPMC_PS2
}

} // namespace mpmc
