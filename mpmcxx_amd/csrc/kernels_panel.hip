// kernels_panel.hip -- the Jacobi contraction of the production path, "panel" form.
//
// One Jacobi iteration is  F = -(A - diag) mu  (reference contract_dipoles, src/System.Energy.cpp:3564-3598, over the matrix of
// thole_amatrix :2661-2770), evaluated pair by pair:  F_i -= a mu_j - b d (d.mu_j),  F_j -= a mu_i - b d (d.mu_i),  T = a I - b d(x)d.
//
// The single-tile-pair walk of kernels_sym.hip (one wave = 64 i-atoms in registers x 64 j-atoms in LDS, j-side accumulators rotated by one
// lane per step) pays, per PAIR, 6 LDS reads of the j-atom and 6 v_mov_b32_dpp for the rotation next to ~27 (stored tensor) or ~40
// (recomputed far-field tensor) fp64 operations -- and the kernel is VALU-issue bound with the LDS pipe 70 % busy.  Here one wave takes a
// PANEL: NI tile pairs (I_0, J) ... (I_{NI-1}, J) that share the j-tile.  Lane l holds NI i-atoms (one per member); at step s all of
// them meet the same j = (l + s) & 63, whose data is read from LDS ONCE and whose accumulator, fed by all NI members, is rotated ONCE:
// LDS reads and lane rotations per pair drop by NI.
//
// Members of a panel share their class: all stored or all far-field, and one set of "non-uniform" dimensions (k_classify: in a uniform
// dimension the periodic image index is the same for all 4096 atom pairs of a tile pair, the i-atom is pre-shifted by that lattice
// vector and the displacement costs one subtraction).  A member that is uniform in more dimensions than the panel simply takes the
// general path there (same image index by construction).  The panel's coordinates are PERMUTED so that its NU non-uniform dimensions
// come first: the walk is instantiated for NU = 0..3 instead of the 8 masks, the LDS image of the j-tile is written in permuted order,
// and F / G are un-permuted when they are stored.
//
// k_build_panels pairs up, for every j-tile J, the tile pairs (I < J, J) of equal class (then leftovers of equal far/stored kind, with
// the intersection of their uniform masks); what stays single -- the diagonal tile pair and at most one odd leftover per kind -- runs
// through the same walk with one member, in the same launch.  Skewed cells: "dimension" = lattice direction; the i-atom is pre-shifted by the
// lattice vectors of the directions with a common index (a whole vector: B^T img mixes the Cartesian components), the others cost the row of R,
// rint and the lattice vector per pair (TRI, instantiated per mask for the far-field walk).
// Partial sums: F_k -> part[J][I_k atoms] (i-side), the entry's combined G -> gpart[entry] (j-side), each written exactly once per
// iteration.  Inside this path a slot holds its 64 x 3 doubles COMPONENT-major ([3][64]): every store instruction of the closing wave
// writes four whole 128-byte lines (what a write-through store wants) and every load of the update is unit-stride.
//
// The update of the dipoles rides the same launch (round 5): the workgroup that delivers the LAST contribution to tile X -- the nt - X
// i-side slots part[S][X], S >= X, and the slots of the entries of X's segment -- runs new_mu = alpha (E0 + F) for X's 64 atoms, sums
// taken from the slots in the fixed (group, slot) order of k_dipole_update_panel, so the result does not depend on who arrives last.
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): the closing wave stores its slots write-through (sc1), waits for its own
// stores (s_waitcnt vmcnt(0)), then adds to the tiles' arrival counters (agent-scope atomics; the add whose return value completes the
// count is the last one); the arriving workgroup takes ONE agent-scope acquire, waits for it, and passes a workgroup barrier before any
// of its waves loads a slot (sc1 loads on top: L1 is not trusted).  The entries are launched in DESCENDING j-tile order: tile X is
// complete when segment X is (its i-side pairs (X, S) live in the segments S > X), so the updates spread over the whole launch and the
// longest segments start first; iteration k + 1 still starts at the kernel boundary.
#include "kernels.h"
#include "device_math.h"

namespace mpmc {

// Work table: one entry per workgroup of kPanelWaves waves, which split the entry's walk by steps.  Entries of j-tile J occupy [seg[J], seg[J + 1]) (host-made): its diagonal tile pair, at most floor(J / 2)
// panels of two tile pairs and up to two single off-diagonal tile pairs (one per kind); unused entries carry tp = -1.
//   entry = { tile pair A, tile pair B (-1: single), uniform mask | far << 3 | diagonal << 4, J }
int panel_segment_entries(int J) { return J / 2 + 3; }
constexpr int kPanFar = 8, kPanDiag = 16;
#ifndef MPMC_PANEL_WAVES
#define MPMC_PANEL_WAVES 2
#endif
// waves per workgroup: they split the steps of ONE entry's walk.  TWO since round 5 (32 steps each): a wave's prologue and epilogue -- atom
// loads, the partial sums' trip through LDS, the closing wave's fold -- are ~90 VALU instructions, 7 % of the launch with four waves of 16
// steps; same-box builds (tools/ab_panel_waves.sh, two boxes): 2 waves +1.8 % evaluations/s with 32 beads in flight and level alone,
// 1 wave +1 % in flight and 8 % slower alone (units of 48 us), 8 waves -13 %.
constexpr int kPanelWaves = MPMC_PANEL_WAVES;
#ifndef MPMC_PANEL_PIPE
#define MPMC_PANEL_PIPE 4 // depth of the stored walk's prefetch ring (steps ahead): 2 / 8 measured in round 5, tools/README.md
#endif
constexpr double kFarSumScale = 0.125; // the far-field walk works with 2 / r (pan_step): its sums carry 8 / r^3

__device__ __forceinline__ int tp_index(int I, int J, int nt) { return I * nt - (I * (I - 1)) / 2 + (J - I); }

// One WAVE per j-tile J.  Its off-diagonal tile pairs (I < J, J) are taken 64 at a time: lane = one tile pair, key = its class
// (far << 3 | uniform mask).  Within a class the members pair up by rank (ballot + popcount): ranks (0,1), (2,3), ...; an odd one out
// is carried to the next chunk of 64 as that class's pending member.  At the end the pending members of equal kind (stored / far)
// pair up across classes -- the panel then keeps only the dimensions uniform for BOTH -- and what is still single becomes a
// one-member entry.  Every step is wave-uniform or a fixed function of the lane: the table is the same whatever the timing.
__global__ __launch_bounds__(64) void k_build_panels(const int *__restrict__ cls, int nt, const int *__restrict__ seg, int4 *__restrict__ panels,
                                                     int *__restrict__ arrive /*[nt] arrival counters of the fused update: left at zero*/) {
	__shared__ int s_odd[16][33]; // the odd-rank member of every pair of one chunk, by class and pair index
	const int J = blockIdx.x, lane = threadIdx.x;
	if (J >= nt) return;
	if (arrive && lane == 0) arrive[J] = 0; // (an evaluation that failed half way may have left a count behind)
	int4 *out = panels + seg[J];
	const int cap = seg[J + 1] - seg[J];
	int n = 0; // entries written so far (wave-uniform)
	if (lane == 0) out[0] = make_int4(tp_index(J, J, nt), -1, ((cls[tp_index(J, J, nt)] / CLS_UNIFORM_X) & 7) | kPanDiag, J);
	n = 1;
	int pend[16]; // pending member of every class (wave-uniform values, kept in every lane)
#pragma unroll
	for (int k = 0; k < 16; ++k) pend[k] = -1;
	const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane)); // lanes below this one
	for (int I0 = 0; I0 < J; I0 += 64) {
		const int I = I0 + lane;
		const bool live = I < J;
		const int tp = live ? tp_index(I, J, nt) : -1;
		const int c = live ? cls[tp] : 0;
		const int key = ((c & CLS_THOLE_FAR) ? 8 : 0) | ((c / CLS_UNIFORM_X) & 7);
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			const unsigned long long mask = __ballot(live && key == k);
			if (mask == 0ull) continue; // wave-uniform
			const int cnt = __popcll(mask);
			const bool mine = live && key == k;
			// with a pending member from an earlier chunk the ranks shift by one: pending = rank 0
			const int shift = (pend[k] >= 0) ? 1 : 0;
			const int rank = __popcll(mask & lt) + shift;
			const int total = cnt + shift;
			if (mine && (rank & 1)) s_odd[k][rank >> 1] = tp;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			if (mine && !(rank & 1) && rank + 1 < total) out[n + (rank >> 1)] = make_int4(tp, s_odd[k][rank >> 1], k, J);
			if (shift && lane == 0) out[n] = make_int4(pend[k], s_odd[k][0], k, J); // the pending member is rank 0: its partner is rank 1
			// the new pending member: the last one when the total is odd
			int np = -1;
			if (total & 1) {
				// (total odd) the member of rank total - 1: the pending one itself when nobody joined (cannot happen: cnt >= 1), else the top lane of the class
				const int top = 63 - __clzll(mask);
				np = __shfl(tp, top, 64);
			}
			n += total >> 1;
			pend[k] = np;
			__builtin_amdgcn_wave_barrier();
		}
	}
	if (lane == 0) {
		for (int far = 0; far < 2; ++far) { // leftovers of one kind pair up in class order
			int prev = -1, prev_um = 0;
#pragma unroll
			for (int um = 7; um >= 0; --um) {
				const int p = pend[far * 8 + um];
				if (p < 0) continue;
				if (prev >= 0) {
					out[n++] = make_int4(prev, p, (prev_um & um) | (far ? kPanFar : 0), J);
					prev = -1;
				} else {
					prev = p;
					prev_um = um;
				}
			}
			if (prev >= 0) out[n++] = make_int4(prev, -1, prev_um | (far ? kPanFar : 0), J); // odd one out: a single
		}
		for (; n < cap; ++n) out[n] = make_int4(-1, -1, 0, J);
	}
}

// write-through store / L1-bypassing load (global_store / global_load ... sc1): the hand-off of the partial slots between workgroups
template <bool SC1>
__device__ __forceinline__ void st_slot(double *p, double v) {
	if (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else *p = v;
}
template <bool SC1>
__device__ __forceinline__ double ld_slot(const double *p) {
	if (SC1) return __hip_atomic_load(const_cast<double *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return *p;
}

template <int NI>
struct PanAcc {
	double f[NI][3];
	double g[3];
};

// one step of the walk: every member against j = (lane + s) & 63 (LDS slot jl = lane + s, no wrap: the image holds every value twice)
template <bool FAR, int NU, int NI, bool ROT, bool PAD, bool TRI>
__device__ __forceinline__ void pan_step(const Box &bx, const double2 *__restrict__ s_xy, const double2 *__restrict__ s_zm, const double2 *__restrict__ s_mm,
                                         const double *__restrict__ s_valid, const int jl, const int src4, const double (&L)[3],
                                         const double (&iL)[3], const double (&q)[NI][3], const double (&m)[NI][3], const double2 (&t)[NI],
                                         PanAcc<NI> &A) {
	const double2 xy = s_xy[jl], zm = s_zm[jl], mm = s_mm[jl];
	const double xj = xy.x, yj = xy.y, zj = zm.x, mjx = zm.y, mjy = mm.x, mjz = mm.y;
	double vj = 1.0;
	if (FAR && PAD) vj = s_valid[jl];
#pragma unroll
	for (int k = 0; k < NI; ++k) {
		double ox = q[k][0] - xj, oy = q[k][1] - yj, oz = q[k][2] - zj;
		if (TRI) { // general cell: NU is the MASK of the lattice directions without a tile-pair-wide image index; each one costs the row of R,
			// rint and the lattice vector (values only: nothing here is a predicate)
			const double dx = ox, dy = oy, dz = oz;
			if (NU & 1) {
				const double n0 = rint(fma(bx.r[6], dz, fma(bx.r[3], dy, bx.r[0] * dx)));
				ox = fma(-n0, bx.b[0], ox), oy = fma(-n0, bx.b[1], oy), oz = fma(-n0, bx.b[2], oz);
			}
			if (NU & 2) {
				const double n1 = rint(fma(bx.r[7], dz, fma(bx.r[4], dy, bx.r[1] * dx)));
				ox = fma(-n1, bx.b[3], ox), oy = fma(-n1, bx.b[4], oy), oz = fma(-n1, bx.b[5], oz);
			}
			if (NU & 4) {
				const double n2 = rint(fma(bx.r[8], dz, fma(bx.r[5], dy, bx.r[2] * dx)));
				ox = fma(-n2, bx.b[6], ox), oy = fma(-n2, bx.b[7], oy), oz = fma(-n2, bx.b[8], oz);
			}
		} else {
			if (NU > 0) ox = fma(-L[0], rint(iL[0] * ox), ox);
			if (NU > 1) oy = fma(-L[1], rint(iL[1] * oy), oy);
			if (NU > 2) oz = fma(-L[2], rint(iL[2] * oz), oz);
		}
		double ta, tb;
		if (FAR) { // bare dipole tensor T = a (w d (x) d - 1), a = 1/r^3, w = 3/r^2: beyond lambda r = kTholeFarX the Thole damping is dropped (kernels.h)
			const double r2 = fma(oz, oz, fma(oy, oy, ox * ox));
			// 2 / r: hardware seed and one Newton step in its product form y (3 - x y^2) -- the halving that 1 / r would need is a power of two and is
			// applied once per wave to the sums instead (kFarSumScale: a = 8 / r^3 here); ~2e-14 relative like fast_rsqrt_1, one instruction less per pair
			const double y0 = __builtin_amdgcn_rsq(r2);
			const double y2 = y0 * fma(-(r2 * y0), y0, 3.0);
			const double i4 = y2 * y2; // 4 / r^2
			ta = i4 * y2;              // 8 / r^3
			if (PAD) ta *= vj; // padded slots of the last tile
			tb = 0.75 * i4; // 3 / r^2  (w, not b = a w: the factor a is applied once per component below, which saves the product a w)
			const double sj = tb * fma(oz, mjz, fma(oy, mjy, ox * mjx));
			const double si = tb * fma(oz, m[k][2], fma(oy, m[k][1], ox * m[k][0]));
			A.f[k][0] = fma(ta, fma(sj, ox, -mjx), A.f[k][0]);
			A.f[k][1] = fma(ta, fma(sj, oy, -mjy), A.f[k][1]);
			A.f[k][2] = fma(ta, fma(sj, oz, -mjz), A.f[k][2]);
			A.g[0] = fma(ta, fma(si, ox, -m[k][0]), A.g[0]);
			A.g[1] = fma(ta, fma(si, oy, -m[k][1]), A.g[1]);
			A.g[2] = fma(ta, fma(si, oz, -m[k][2]), A.g[2]);
		} else {
			ta = t[k].x;
			tb = t[k].y;
			const double dj = tb * fma(oz, mjz, fma(oy, mjy, ox * mjx));
			const double di = tb * fma(oz, m[k][2], fma(oy, m[k][1], ox * m[k][0]));
			A.f[k][0] = fma(-ta, mjx, fma(dj, ox, A.f[k][0]));
			A.f[k][1] = fma(-ta, mjy, fma(dj, oy, A.f[k][1]));
			A.f[k][2] = fma(-ta, mjz, fma(dj, oz, A.f[k][2]));
			A.g[0] = fma(-ta, m[k][0], fma(di, ox, A.g[0]));
			A.g[1] = fma(-ta, m[k][1], fma(di, oy, A.g[1]));
			A.g[2] = fma(-ta, m[k][2], fma(di, oz, A.g[2]));
		}
	}
	if (ROT) {
		A.g[0] = rot_from_next(A.g[0]);
		A.g[1] = rot_from_next(A.g[1]);
		A.g[2] = rot_from_next(A.g[2]);
	}
}

// the walk: n_steps steps from s_first on (64 from 0 for an off-diagonal tile pair, 32 from 1 for a diagonal one); n_steps is a multiple of PIPE
template <bool FAR, int NU, int NI, int PIPE, bool TRI>
__device__ __forceinline__ void pan_walk(const Box &bx, const double2 *__restrict__ s_xy, const double2 *__restrict__ s_zm, const double2 *__restrict__ s_mm,
                                         const double *__restrict__ s_valid, const bool pad, const int lane, const int src4,
                                         const double (&L)[3], const double (&iL)[3], const double (&q)[NI][3], const double (&m)[NI][3],
                                         const double2 *__restrict__ ab, const size_t (&ab_tile)[NI], const int s_first, const int n_steps,
                                         PanAcc<NI> &A) {
	int jb = lane + s_first;
	if (FAR) {
		const double2 none[NI] = {};
#define MPMC_FAR_LOOP(P)                                                                                                               \
	for (int kc = 0; kc < n_steps - 4; kc += 4, jb += 4) {                                                                             \
		_Pragma("unroll") for (int u = 0; u < 4; ++u)                                                                                  \
		    pan_step<true, NU, NI, true, P, TRI>(bx, s_xy, s_zm, s_mm, s_valid, jb + u, src4, L, iL, q, m, none, A);                      \
	}                                                                                                                                  \
	_Pragma("unroll") for (int u = 0; u < 3; ++u) pan_step<true, NU, NI, true, P, TRI>(bx, s_xy, s_zm, s_mm, s_valid, jb + u, src4, L, iL, q, m, none, A); \
	pan_step<true, NU, NI, false, P, TRI>(bx, s_xy, s_zm, s_mm, s_valid, jb + 3, src4, L, iL, q, m, none, A);
		if (pad) { // wave-uniform: only the panels whose j-tile is the padded last tile
			MPMC_FAR_LOOP(true)
		} else {
			MPMC_FAR_LOOP(false)
		}
#undef MPMC_FAR_LOOP
		return;
	}
	// stored tensors: one rolling prefetch ring of PIPE loads per member (the (a, b) of step k + PIPE is requested as soon as the
	// registers of step k are consumed); masked pairs were stored as (0, 0), so the walk carries no predicate.  Addresses are a
	// wave-uniform base per member (scalar registers, derived from the kernel argument so the loads stay global_load: a flat_load would
	// also count against the LDS counter) plus ONE per-lane element offset shared by the members.
	double2 buf[NI][PIPE];
	int voff = s_first * kTile + lane;
#pragma unroll
	for (int k = 0; k < NI; ++k) {
#pragma unroll
		for (int u = 0; u < PIPE; ++u) buf[k][u] = ld_stream<true>(ab + ab_tile[k] + voff + u * kTile);
	}
	for (int kc = 0; kc < n_steps - PIPE; kc += PIPE, jb += PIPE) {
		voff += PIPE * kTile;
#pragma unroll
		for (int u = 0; u < PIPE; ++u) {
			double2 t[NI];
#pragma unroll
			for (int k = 0; k < NI; ++k) t[k] = buf[k][u];
			pan_step<false, NU, NI, true, false, TRI>(bx, s_xy, s_zm, s_mm, s_valid, jb + u, src4, L, iL, q, m, t, A);
#pragma unroll
			for (int k = 0; k < NI; ++k) buf[k][u] = ld_stream<true>(ab + ab_tile[k] + voff + u * kTile); // refill in place
			__builtin_amdgcn_sched_barrier(0);
		}
	}
#pragma unroll
	for (int u = 0; u < PIPE; ++u) {
		double2 t[NI];
#pragma unroll
		for (int k = 0; k < NI; ++k) t[k] = buf[k][u];
		if (u != PIPE - 1) pan_step<false, NU, NI, true, false, TRI>(bx, s_xy, s_zm, s_mm, s_valid, jb + u, src4, L, iL, q, m, t, A);
		else pan_step<false, NU, NI, false, false, TRI>(bx, s_xy, s_zm, s_mm, s_valid, jb + u, src4, L, iL, q, m, t, A);
		__builtin_amdgcn_sched_barrier(0);
	}
}

template <int PIPE, int NI, bool ORTHO, bool FUSED>
__device__ __forceinline__ void panel_block(const AtomsDev &at, const Box &bx, const double *__restrict__ mu, const int2 *__restrict__ tile_pairs,
                                            const double4 *__restrict__ tp_shift, const double2 *__restrict__ ab, double *__restrict__ part,
                                            double *__restrict__ gslot, const int tpA, const int tpB, const int flags, const int J,
                                            double2 *__restrict__ s_xy, double2 *__restrict__ s_zm, double2 *__restrict__ s_mm,
                                            double *__restrict__ s_valid, double (*__restrict__ s_F)[2][3][kTile], double (*__restrict__ s_G)[3][kTile]) {
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int src4 = ((lane + 1) & 63) * 4;
	const int tps[2] = {tpA, tpB};
	const int um = flags & 7; // dimensions uniform for every member
	const bool far = (flags & kPanFar) != 0, diag = (flags & kPanDiag) != 0;
	const int j0 = J * kTile;
	// permutation (p0, p1, p2): the non-uniform dimensions first, ascending; then the uniform ones, ascending
	//   mask of non-uniform dims -> packed permutation p0 | p1 << 2 | p2 << 4
	const int nonuni = (~um) & 7;
	const int nu = __popc(nonuni);
	unsigned packed = 0x24; // x y z
	switch (ORTHO ? nonuni : 7) { // (a skewed cell keeps its order: the walk is instantiated per mask there)
	case 2: packed = 0x21; break; // y | x z
	case 4: packed = 0x12; break; // z | x y
	case 5: packed = 0x18; break; // x z | y
	case 6: packed = 0x09; break; // y z | x
	default: break;
	}
	const int p0 = packed & 3, p1 = (packed >> 2) & 3, p2 = (packed >> 4) & 3;
	auto pick = [](const double x, const double y, const double z, const int p) { return p == 0 ? x : (p == 1 ? y : z); };
	const double L[3] = {pick(bx.b[0], bx.b[4], bx.b[8], p0), pick(bx.b[0], bx.b[4], bx.b[8], p1), pick(bx.b[0], bx.b[4], bx.b[8], p2)};
	const double iL[3] = {pick(bx.r[0], bx.r[4], bx.r[8], p0), pick(bx.r[0], bx.r[4], bx.r[8], p1), pick(bx.r[0], bx.r[4], bx.r[8], p2)};

	// The permutation is applied by the ADDRESS of the loads (component p_d of an atom's record: a wave-uniform base, the lane's offset), not by
	// selects on loaded values: the prologue of a wave is paid once per 16 steps.
	const double *__restrict__ xyz = reinterpret_cast<const double *>(at.xyzq); // { x, y, z, q } per atom
	const double *__restrict__ xp[3] = {xyz + p0, xyz + p1, xyz + p2}, *__restrict__ mp[3] = {mu + p0, mu + p1, mu + p2};
	if (w == 0) { // j-tile into LDS, permuted, every value twice (slot l + s never wraps)
		const unsigned a4 = 4u * (unsigned)(j0 + lane), a3 = 3u * (unsigned)(j0 + lane);
		const double2 xy = make_double2(xp[0][a4], xp[1][a4]);
		const double2 zm = make_double2(xp[2][a4], mp[0][a3]);
		const double2 mm = make_double2(mp[1][a3], mp[2][a3]);
		s_xy[lane] = s_xy[lane + kTile] = xy;
		s_zm[lane] = s_zm[lane + kTile] = zm;
		s_mm[lane] = s_mm[lane + kTile] = mm;
		s_valid[lane] = s_valid[lane + kTile] = (at.mf[j0 + lane].y & AF_PAD) ? 0.0 : 1.0;
	}
	double q[NI][3], m[NI][3];
	size_t ab_tile[NI]; // wave-uniform element offset of each member's 64 x 64 block of the store
	int Is[NI];
#pragma unroll
	for (int k = 0; k < NI; ++k) {
		const int I = tile_pairs[tps[k]].x;
		Is[k] = I;
		const unsigned a4 = 4u * (unsigned)(I * kTile + lane), a3 = 3u * (unsigned)(I * kTile + lane);
		// the i-atom moves by the tile pair's common lattice vector in the uniform dimensions (wave-uniform: scalar registers; x - 0.0 = x)
		const double4 sh = tp_shift[tps[k]];
		// (skewed cell: the shift is the whole vector of THIS member's common directions -- zero where it has none; a direction common for the
		// member but not for the panel is rounded per pair all the same and comes out as index 0)
		const bool whole = !ORTHO;
		const double sx = (whole || (um & 1)) ? sh.x : 0.0, sy = (whole || (um & 2)) ? sh.y : 0.0, sz = (whole || (um & 4)) ? sh.z : 0.0;
		q[k][0] = xp[0][a4] - pick(sx, sy, sz, p0);
		q[k][1] = xp[1][a4] - pick(sx, sy, sz, p1);
		q[k][2] = xp[2][a4] - pick(sx, sy, sz, p2);
		m[k][0] = mp[0][a3];
		m[k][1] = mp[1][a3];
		m[k][2] = mp[2][a3];
		ab_tile[k] = (size_t)tps[k] * (kTile * kTile);
	}
	__syncthreads();
	const bool pad = (at.n != at.n_pad) && (J == at.n_pad / kTile - 1);
	// the waves split the walk: 64 / kPanelWaves steps each of the 64 of an off-diagonal tile pair (s = 0..63), 32 / kPanelWaves of the 32 of a diagonal one (s = 1..32)
	const int n_steps = (diag ? 32 : 64) / kPanelWaves, s_first = (diag ? 1 : 0) + w * n_steps;
	PanAcc<NI> A = {};
#define MPMC_PWALK(F, N) pan_walk<F, N, NI, PIPE, false>(bx, s_xy, s_zm, s_mm, s_valid, pad, lane, src4, L, iL, q, m, ab, ab_tile, s_first, n_steps, A)
#define MPMC_PWALK_TRI(F, M) pan_walk<F, M, NI, PIPE, true>(bx, s_xy, s_zm, s_mm, s_valid, pad, lane, src4, L, iL, q, m, ab, ab_tile, s_first, n_steps, A)
	if (!ORTHO && nu > 0) { // skewed cell: the far-field walk per mask of directions without a common index; the (few) stored ones take all three
		if (far) {
			switch (nonuni) {
			case 1: MPMC_PWALK_TRI(true, 1); break;
			case 2: MPMC_PWALK_TRI(true, 2); break;
			case 3: MPMC_PWALK_TRI(true, 3); break;
			case 4: MPMC_PWALK_TRI(true, 4); break;
			case 5: MPMC_PWALK_TRI(true, 5); break;
			case 6: MPMC_PWALK_TRI(true, 6); break;
			default: MPMC_PWALK_TRI(true, 7); break;
			}
		} else MPMC_PWALK_TRI(false, 7);
	} else if (!ORTHO) { // (every direction has a common index: plain differences of the pre-shifted positions)
		if (far) MPMC_PWALK(true, 0);
		else MPMC_PWALK(false, 0);
	} else if (far) {
		switch (nu) {
		case 0: MPMC_PWALK(true, 0); break;
		case 1: MPMC_PWALK(true, 1); break;
		case 2: MPMC_PWALK(true, 2); break;
		default: MPMC_PWALK(true, 3); break;
		}
	} else {
		switch (nu) {
		case 0: MPMC_PWALK(false, 0); break;
		case 1: MPMC_PWALK(false, 1); break;
		case 2: MPMC_PWALK(false, 2); break;
		default: MPMC_PWALK(false, 3); break;
		}
	}
#undef MPMC_PWALK
#undef MPMC_PWALK_TRI
	if (far) { // the far-field walk accumulates with a = 8 / r^3 (pan_step)
#pragma unroll
		for (int d = 0; d < 3; ++d) {
#pragma unroll
			for (int k = 0; k < NI; ++k) A.f[k][d] *= kFarSumScale;
			A.g[d] *= kFarSumScale;
		}
	}
	// the waves' partial sums meet in LDS: F of lane l's own atoms, G parked at the atom it belongs to -- after n_steps - 1 rotations
	// lane l holds the accumulator of j = (l + s_first + n_steps - 1) & 63
	{
		const int jl_last = (lane + s_first + n_steps - 1) & 63;
#pragma unroll
		for (int d = 0; d < 3; ++d) {
#pragma unroll
			for (int k = 0; k < NI; ++k) s_F[w][k][d][lane] = A.f[k][d];
			s_G[w][d][jl_last] = A.g[d];
		}
	}
	__syncthreads();
	if (w != 0) return;
	// wave 0: sums in wave order (fixed => reproducible) and stores; the permutation is undone by the ADDRESS (component d of the
	// walk is component p_d of the cell).  i-side: slot [J][I_k atoms]; j-side: this entry's slot.
	const int nt_pad3 = at.n_pad * 3;
	double g[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		g[d] = s_G[0][d][lane];
#pragma unroll
		for (int v = 1; v < kPanelWaves; ++v) g[d] += s_G[v][d][lane];
	}
#pragma unroll
	for (int k = 0; k < NI; ++k) {
		double f[3];
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			f[d] = s_F[0][k][d][lane];
#pragma unroll
			for (int v = 1; v < kPanelWaves; ++v) f[d] += s_F[v][k][d][lane];
		}
		if (NI == 1 && diag) { // both sides are the same 64 atoms: one slot [J][J atoms] = F + G, nothing on the j-side
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				f[d] += g[d];
				g[d] = 0.0;
			}
		}
		double *oi = part + (size_t)J * nt_pad3 + 3 * (size_t)(Is[k] * kTile) + lane; // slot [J][I_k atoms], component-major
		st_slot<FUSED>(oi + p0 * kTile, f[0]);
		st_slot<FUSED>(oi + p1 * kTile, f[1]);
		st_slot<FUSED>(oi + p2 * kTile, f[2]);
	}
	st_slot<FUSED>(gslot + p0 * kTile + lane, g[0]);
	st_slot<FUSED>(gslot + p1 * kTile + lane, g[1]);
	st_slot<FUSED>(gslot + p2 * kTile + lane, g[2]);
}

// ---- new_mu = alpha (E0 + F) for the 64 atoms of tile X ------------------------------------------------------------------------------
// F = sum of the panel kernel's slots of this tile: part[S][X atoms] for S = X .. nt-1 (i-side, the diagonal included) and gpart[e] for
// the entries of X's segment of the work table (j-side).  Tail as k_dipole_update (contract_dipoles :3586-3593, calc_dipole_rrms
// :3147-3177, are_we_done_yet :3227-3236).  The sums are taken in kUpdGroups strided groups -- group g: slots g, g + 16, ... of the i-side,
// then of the j-side -- and the groups added in order: fixed by (g, t) alone, whoever runs it and with however many waves (NW waves take
// kUpdGroups / NW groups each, with independent accumulators: the loads of a wave are in flight together).
constexpr int kUpdGroups = 16;
struct PanelUpdate {
	const double *e_static;
	const int *seg;
	double *mu_new, *e_induced /*may be null: not wanted (timing launches)*/, *rrms_atom;
	double allowed_sqerr;
	int *ctl, *host_flag;
	int *arrive; // [nt] arrival counters; null: the update is a launch of its own (k_dipole_update_panel)
	int nt, it, want_rrms, reverse;
	int probe; // measurement only: arrive, but skip the update (the producer side of the hand-off alone; results are NOT valid)
};

template <int NW, bool SC1, int ROUNDS = ((8 * NW) / 16 > 0 ? (8 * NW) / 16 : 1)>
__device__ __forceinline__ void panel_update_tile(const AtomsDev &at, const PanelUpdate &u, const double *__restrict__ part, const double *__restrict__ gpart,
                                                  const double *__restrict__ mu_old, const int X, double (*__restrict__ sh)[kTile][3]) {
	constexpr int GW = kUpdGroups / NW; // groups per wave
	const int a = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int i = X * kTile + a;
	const int nF = u.nt - X, wg0 = u.seg[X], nG = u.seg[X + 1] - wg0;
	// what the closing wave needs behind the barrier is requested now (it would be a second latency chain there)
	double al = 0.0, es[3] = {0, 0, 0}, mo[3] = {0, 0, 0};
	if (w == 0) {
		al = at.alpha[i];
		for (int p = 0; p < 3; ++p) {
			es[p] = u.e_static[3 * (size_t)i + p];
			mo[p] = mu_old[3 * (size_t)i + p];
		}
	}
	// One list of T = nF + nG slots (i-side first); group g takes t = g, g + 16, ...  Every load is issued unconditionally (a slot index
	// past the end is clamped and its value replaced by zero: no branch between the loads), ROUNDS rounds of the wave's groups per trip: 3 GW
	// ROUNDS independent loads in flight per lane (24; 48 measured no faster) -- the update is
	// a latency chain (few workgroups, data fresh from other CUs), not a bandwidth one.  The order of the additions does not depend on ROUNDS.
	double f[GW][3] = {};
	const size_t n_pad = (size_t)at.n_pad;
	const int T = nF + nG;
	const double *__restrict__ pX = part + ((size_t)X * n_pad + (size_t)X * kTile) * 3 + a; // slot X, this tile's block; slot X + t is t * n_pad * 3 further
	const double *__restrict__ gX = gpart + (size_t)wg0 * (kTile * 3) + a;
	for (int t0 = 0; t0 < T; t0 += ROUNDS * kUpdGroups) {
		double v[ROUNDS][GW][3];
#pragma unroll
		for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
			for (int c = 0; c < GW; ++c) {
				const int t = t0 + r * kUpdGroups + w * GW + c;
				const int tc = t < T ? t : T - 1;
				const double *__restrict__ q = tc < nF ? pX + (size_t)tc * n_pad * 3 : gX + (size_t)(tc - nF) * (kTile * 3);
				v[r][c][0] = ld_slot<SC1>(q);
				v[r][c][1] = ld_slot<SC1>(q + kTile);
				v[r][c][2] = ld_slot<SC1>(q + 2 * kTile);
			}
		}
#pragma unroll
		for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
			for (int c = 0; c < GW; ++c) {
				const bool ok = t0 + r * kUpdGroups + w * GW + c < T;
#pragma unroll
				for (int p = 0; p < 3; ++p) f[c][p] += ok ? v[r][c][p] : 0.0;
			}
		}
	}
#pragma unroll
	for (int c = 0; c < GW; ++c) {
		sh[w * GW + c][a][0] = f[c][0];
		sh[w * GW + c][a][1] = f[c][1];
		sh[w * GW + c][a][2] = f[c][2];
	}
	__syncthreads();
	if (w != 0) return;
	double fsum[3];
	for (int p = 0; p < 3; ++p) {
		double v = sh[0][a][p];
		for (int k = 1; k < kUpdGroups; ++k) v += sh[k][a][p];
		fsum[p] = v;
	}
	const bool live = (i < at.n) && (al != 0.0);
	double fo[3] = {0, 0, 0}, nm[3] = {0, 0, 0};
	if (live) {
		for (int p = 0; p < 3; ++p) {
			fo[p] = fsum[p];
			nm[p] = al * (es[p] + fo[p]);
		}
	}
	bool broke = false;
	double acc = 0, nn = 0;
	for (int p = 0; p < 3; ++p) {
		const double d = nm[p] - mo[p];
		acc += d * d;
		nn += nm[p] * nm[p];
		if (d * d > u.allowed_sqerr) broke = true;
		u.mu_new[3 * (size_t)i + p] = nm[p];
		if (u.e_induced) u.e_induced[3 * (size_t)i + p] = fo[p];
	}
	if (u.want_rrms) {
		double r = sqrt(acc / nn);
		if (!isfinite(r)) r = 0.0;
		u.rrms_atom[i] = (i < at.n) ? r : 0.0;
	}
	if (u.ctl) { // are_we_done_yet on the device (iteration_verdict in kernels.hip; repeated here: separate translation unit)
		int *ctl = u.ctl;
		const bool wave_broke = __any(u.allowed_sqerr > 0.0 && broke && i < at.n);
		if (a != 0) return;
		if (wave_broke) atomicOr(&ctl[0], 1);
		__threadfence();
		const int ticket = atomicAdd(&ctl[2], 1);
		if (ticket != u.nt - 1) return;
		__threadfence();
		const int any_broke = atomicOr(&ctl[0], 0);
		if (!any_broke) ctl[1] = u.it;
		ctl[0] = 0;
		ctl[2] = 0;
		if (u.host_flag) { // pinned { last closed iteration, converged-at }: the host spins on it instead of synchronising the stream
			__hip_atomic_store(u.host_flag + 1, any_broke ? 0 : u.it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(u.host_flag, u.it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

// (per FUSED two instantiations: the orthorhombic one reads three diagonal elements of the cell and its inverse, which keeps most of the
// Box out of its scalar registers; the other holds the walks of skewed cells)
constexpr int kPanelWalkDouble2 = 6 * kTile + kTile + (kPanelWaves * 2 * 3 * kTile + kPanelWaves * 3 * kTile) / 2; // j-tile image (every value twice), valid flags, s_F, s_G
constexpr int kPanelLdsDouble2 = kPanelWalkDouble2; // (the fused update's group sums reuse the walk's LDS: with four waves and more it is large enough)
template <int PIPE, bool ORTHO, bool FUSED>
__global__ __launch_bounds__(64 * kPanelWaves) void k_dipole_iter_panel(AtomsDev at, Box bx, const double *__restrict__ mu,
                                                                        const int2 *__restrict__ tile_pairs, const double4 *__restrict__ tp_shift,
                                                                        const int4 *__restrict__ panels, const double2 *__restrict__ ab,
                                                                        double *__restrict__ part, double *__restrict__ gpart /*[entries][3][64]*/,
                                                                        const int *__restrict__ converged /*null, or &ctl[1] of the precision-terminated solve*/,
                                                                        long long *__restrict__ trace /*null; measurement only: [entries][4] = start, end (100 MHz ticks), HW_ID, XCC_ID*/,
                                                                        const PanelUpdate u) {
	long long t_start = 0;
	if (trace) t_start = wall_clock64();
	__shared__ double2 s_all[(FUSED && kPanelLdsDouble2 * 2 < kUpdGroups * kTile * 3) ? (kUpdGroups * kTile * 3) / 2 : kPanelLdsDouble2];
	__shared__ int s_todo[4];
	if (converged && *converged != 0) return; // an iteration enqueued ahead of the verdict: nothing to do
	double2 *s_xy = s_all, *s_zm = s_all + 2 * kTile, *s_mm = s_all + 4 * kTile;
	double *s_valid = reinterpret_cast<double *>(s_all + 6 * kTile);
	double(*s_F)[2][3][kTile] = reinterpret_cast<double(*)[2][3][kTile]>(s_all + 7 * kTile);
	double(*s_G)[3][kTile] = reinterpret_cast<double(*)[3][kTile]>(s_all + 7 * kTile + kPanelWaves * 2 * 3 * kTile / 2);
	// entries in descending j-tile order (the table is ascending): the longest segments first, tile X complete when segment X is
	const int ent = u.reverse ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
	const int4 e = panels[ent];
	// everything that describes the entry is wave-uniform: keep it in scalar registers
	const int tpA = __builtin_amdgcn_readfirstlane(e.x), tpB = __builtin_amdgcn_readfirstlane(e.y);
	const int flags = __builtin_amdgcn_readfirstlane(e.z), J = __builtin_amdgcn_readfirstlane(e.w);
	double *gslot = gpart + (size_t)ent * kTile * 3;
	if (tpA < 0) { // unused entry of this j-tile's segment: its slot is read by the update all the same
		if (threadIdx.x < kTile) {
			st_slot<FUSED>(gslot + threadIdx.x, 0.0);
			st_slot<FUSED>(gslot + kTile + threadIdx.x, 0.0);
			st_slot<FUSED>(gslot + 2 * kTile + threadIdx.x, 0.0);
		}
		if (trace && threadIdx.x == 0) trace[4 * (size_t)ent + 1] = 0; // no work: the reader drops entries whose end stamp is 0
		if (!FUSED) return;
	} else if (tpB >= 0) panel_block<PIPE, 2, ORTHO, FUSED>(at, bx, mu, tile_pairs, tp_shift, ab, part, gslot, tpA, tpB, flags, J, s_xy, s_zm, s_mm, s_valid, s_F, s_G);
	else panel_block<PIPE, 1, ORTHO, FUSED>(at, bx, mu, tile_pairs, tp_shift, ab, part, gslot, tpA, tpB, flags, J, s_xy, s_zm, s_mm, s_valid, s_F, s_G);
	if (FUSED) {
		// arrival: wave 0 stored every slot of this entry; its lanes 0..2 add to the counters of the tiles those slots belong to
		//   diagonal or unused entry: { J };  off-diagonal: { I_A, I_B (panels of two), J }
		const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
		if (w == 0) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's write-through stores have been acknowledged
			int target = -1;
			const bool offdiag = tpA >= 0 && !(flags & kPanDiag);
			if (lane == 0) target = J;
			else if (lane == 1 && offdiag) target = tile_pairs[tpA].x;
			else if (lane == 2 && offdiag && tpB >= 0) target = tile_pairs[tpB].x;
			bool last = false;
			if (target >= 0) {
				const int expected = (u.nt - target) + (u.seg[target + 1] - u.seg[target] - 1);
				last = __hip_atomic_fetch_add(&u.arrive[target], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expected - 1;
			}
			if (lane < 3) s_todo[lane] = last ? target : -1;
			if (__any(last)) { // ONE agent-scope acquire per arriving workgroup, complete before the barrier lets anybody load
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			}
		}
		__syncthreads();
		double(*sh)[kTile][3] = reinterpret_cast<double(*)[kTile][3]>(s_all);
#pragma unroll 1
		for (int k = 0; k < 3; ++k) {
			const int X = __builtin_amdgcn_readfirstlane(s_todo[k]);
			if (X < 0) continue; // (block-uniform)
			if (!u.probe) panel_update_tile<kPanelWaves, true>(at, u, part, gpart, mu, X, sh);
			if (threadIdx.x == 0) u.arrive[X] = 0; // (nobody else touches it any more in this launch; the next one starts from zero)
			__syncthreads();
		}
	}
	if (trace && threadIdx.x == 0 && tpA >= 0) { // (wave 0 is the last one to leave a workgroup: it folds the partial sums)
		long long *o = trace + 4 * (size_t)ent;
		o[0] = t_start;
		o[1] = wall_clock64();
		o[2] = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
		o[3] = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (3 << 11));
	}
}

// the update as a launch of its own (the default; fused_update = 1 rides the contraction instead): one workgroup per tile, the same function,
// the same sums whatever NW.  Four waves by default (update_waves): a sixteen-wave workgroup must find sixteen free wave slots on one CU,
// which an ensemble's other kernels rarely leave (the launch then waits ~170 us for them), and is no faster alone.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_dipole_update_panel(AtomsDev at, const double *__restrict__ part, const double *__restrict__ gpart,
                                                                 const double *__restrict__ mu_old, const PanelUpdate u) {
	__shared__ double sh[kUpdGroups][kTile][3];
	if (u.ctl && u.ctl[1] != 0) return; // converged in an earlier iteration (block-uniform)
	panel_update_tile<NW, false>(at, u, part, gpart, mu_old, blockIdx.x, sh); // (48 loads per lane and trip instead of 24: no faster, round 5)
}

void launch_build_panels(hipStream_t st, const int *cls, int n_tiles, const int *seg, int4 *panels, int *arrive) {
	hipLaunchKernelGGL(k_build_panels, dim3(n_tiles), dim3(64), 0, st, cls, n_tiles, seg, panels, arrive);
}

void launch_dipole_iter_panel(hipStream_t st, const AtomsDev &at, const Box &bx, const double *mu, const int2 *tile_pairs,
                              const double4 *tp_shift, const int4 *panels, int n_entries, const double2 *ab, double *part, double *gpart,
                              const int *converged, long long *trace, int replicas, const PanelFuse *fuse) {
	if (n_entries <= 0) return;
	dim3 grid(n_entries, replicas > 1 ? replicas : 1), block(kTile * kPanelWaves); // (replicas: measurement only -- the same work blockIdx.y times)
	PanelUpdate u{};
	u.reverse = 1;
	const bool fused = fuse && fuse->arrive && replicas <= 1;
	if (fuse) u.reverse = fuse->reverse;
	if (fused) {
		u.e_static = fuse->e_static, u.seg = fuse->seg, u.mu_new = fuse->mu_new, u.e_induced = fuse->e_induced, u.rrms_atom = fuse->rrms_atom;
		u.allowed_sqerr = fuse->allowed_sqerr, u.ctl = fuse->ctl, u.host_flag = fuse->host_flag, u.arrive = fuse->arrive;
		u.nt = at.n_pad / kTile, u.it = fuse->it, u.want_rrms = fuse->want_rrms, u.probe = fuse->probe;
	}
#define MPMC_PANEL(O, F) hipLaunchKernelGGL((k_dipole_iter_panel<MPMC_PANEL_PIPE, O, F>), grid, block, 0, st, at, bx, mu, tile_pairs, tp_shift, panels, ab, part, gpart, converged, trace, u)
	if (bx.ortho) {
		if (fused) MPMC_PANEL(true, true);
		else MPMC_PANEL(true, false);
	} else {
		if (fused) MPMC_PANEL(false, true);
		else MPMC_PANEL(false, false);
	}
#undef MPMC_PANEL
}

void launch_dipole_update_panel(hipStream_t st, const AtomsDev &at, const double *e_static, const double *part, const double *gpart, const int *seg,
                                const double *mu_old, double *mu_new, double *e_induced, int want_rrms, double *rrms_atom, double allowed_sqerr,
                                int *ctl, int *host_flag, int it, int waves) {
	PanelUpdate u{};
	u.e_static = e_static, u.seg = seg, u.mu_new = mu_new, u.e_induced = e_induced, u.rrms_atom = rrms_atom, u.allowed_sqerr = allowed_sqerr;
	u.ctl = ctl, u.host_flag = host_flag, u.arrive = nullptr, u.nt = at.n_pad / kTile, u.it = it, u.want_rrms = want_rrms;
	if (waves == 4) hipLaunchKernelGGL(k_dipole_update_panel<4>, dim3(at.n_pad / kTile), dim3(kTile * 4), 0, st, at, part, gpart, mu_old, u);
	else if (waves == 2) hipLaunchKernelGGL(k_dipole_update_panel<2>, dim3(at.n_pad / kTile), dim3(kTile * 2), 0, st, at, part, gpart, mu_old, u);
	else if (waves == 1) hipLaunchKernelGGL(k_dipole_update_panel<1>, dim3(at.n_pad / kTile), dim3(kTile), 0, st, at, part, gpart, mu_old, u);
	else hipLaunchKernelGGL(k_dipole_update_panel<kUpdGroups>, dim3(at.n_pad / kTile), dim3(kTile * kUpdGroups), 0, st, at, part, gpart, mu_old, u);
}

} // namespace mpmc
