// kernels_panel.hip -- the Jacobi contraction of the production path, "panel" form (orthorhombic cells).
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
// through the same walk with one member, in the same launch.  (Triclinic cells carry no classes: kernels_sym.hip serves them.)
// Every partial slot part[source tile][atom] is still written exactly once per iteration: F_k -> part[J][I_k atoms], the combined
// G -> part[I_0][J atoms], zeros -> part[I_k][J atoms], k > 0.
#include "kernels.h"
#include "device_math.h"

namespace mpmc {

// Work table: one entry per wave, four entries (= one workgroup of four waves) share their j-tile.  Entries of j-tile J occupy
// [seg[J], seg[J + 1]) (multiples of 4, host-made): its diagonal tile pair, at most floor(J / 2) panels of two tile pairs and up to two
// single off-diagonal tile pairs (one per kind); unused entries carry tp = -1.
//   entry = { tile pair A, tile pair B (-1: single), uniform mask | far << 3 | diagonal << 4, J }
int panel_segment_entries(int J) { return ((J / 2 + 3) + 3) / 4 * 4; }
constexpr int kPanFar = 8, kPanDiag = 16;

__device__ __forceinline__ int tp_index(int I, int J, int nt) { return I * nt - (I * (I - 1)) / 2 + (J - I); }

// one thread per j-tile; pending members per class live in LDS (16 classes: far << 3 | uniform mask).  The classes are read in
// batches of 8 independent loads (a dependent load per tile pair made this kernel latency-bound: 96 us at 157 tiles).
__global__ __launch_bounds__(64) void k_build_panels(const int *__restrict__ cls, int nt, const int *__restrict__ seg, int4 *__restrict__ panels) {
	__shared__ int s_pend[16][64];
	const int J = blockIdx.x * 64 + threadIdx.x;
	if (J >= nt) return;
	const int t = threadIdx.x;
	for (int k = 0; k < 16; ++k) s_pend[k][t] = -1;
	int4 *out = panels + seg[J];
	const int cap = seg[J + 1] - seg[J];
	int n = 0;
	out[n++] = make_int4(tp_index(J, J, nt), -1, ((cls[tp_index(J, J, nt)] / CLS_UNIFORM_X) & 7) | kPanDiag, J);
	for (int I0 = 0; I0 < J; I0 += 8) {
		int c[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) c[u] = (I0 + u < J) ? cls[tp_index(I0 + u, J, nt)] : -1;
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			if (I0 + u >= J) break;
			const int tp = tp_index(I0 + u, J, nt);
			const int key = ((c[u] & CLS_THOLE_FAR) ? 8 : 0) | ((c[u] / CLS_UNIFORM_X) & 7);
			const int p = s_pend[key][t];
			if (p >= 0) {
				out[n++] = make_int4(p, tp, key, J);
				s_pend[key][t] = -1;
			} else {
				s_pend[key][t] = tp;
			}
		}
	}
	for (int far = 0; far < 2; ++far) { // leftovers of one kind: pair them in class order, the panel keeps the dimensions uniform for BOTH members
		int prev = -1, prev_um = 0;
		for (int um = 7; um >= 0; --um) {
			const int p = s_pend[far * 8 + um][t];
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

template <int NI>
struct PanAcc {
	double f[NI][3];
	double g[3];
};

// one step of the walk: every member against j = (lane + s) & 63 (LDS slot jl = lane + s, no wrap: the image holds every value twice)
template <int JACC, bool FAR, int NU, int NI, bool ROT, bool PAD>
__device__ __forceinline__ void pan_step(const double2 *__restrict__ s_xy, const double2 *__restrict__ s_zm, const double2 *__restrict__ s_mm,
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
		if (NU > 0) ox = fma(-L[0], rint(iL[0] * ox), ox);
		if (NU > 1) oy = fma(-L[1], rint(iL[1] * oy), oy);
		if (NU > 2) oz = fma(-L[2], rint(iL[2] * oz), oz);
		double ta, tb;
		if (FAR) { // bare dipole tensor a = 1/r^3, b = 3/r^5: beyond lambda r = kTholeFarX the Thole damping is dropped (kernels.h)
			const double r2 = fma(oz, oz, fma(oy, oy, ox * ox));
			const double ir = fast_rsqrt_1(r2);
			const double ir2 = ir * ir;
			ta = ir2 * ir;
			if (PAD) ta *= vj; // padded slots of the last tile
			tb = ta * (3.0 * ir2);
		} else {
			ta = t[k].x;
			tb = t[k].y;
		}
		const double dj = tb * fma(oz, mjz, fma(oy, mjy, ox * mjx));
		const double di = tb * fma(oz, m[k][2], fma(oy, m[k][1], ox * m[k][0]));
		A.f[k][0] = fma(-ta, mjx, fma(dj, ox, A.f[k][0]));
		A.f[k][1] = fma(-ta, mjy, fma(dj, oy, A.f[k][1]));
		A.f[k][2] = fma(-ta, mjz, fma(dj, oz, A.f[k][2]));
		A.g[0] = fma(-ta, m[k][0], fma(di, ox, A.g[0]));
		A.g[1] = fma(-ta, m[k][1], fma(di, oy, A.g[1]));
		A.g[2] = fma(-ta, m[k][2], fma(di, oz, A.g[2]));
	}
	if (ROT) {
		A.g[0] = rot_from_next<JACC == 0>(A.g[0], src4);
		A.g[1] = rot_from_next<JACC == 0>(A.g[1], src4);
		A.g[2] = rot_from_next<JACC == 0>(A.g[2], src4);
	}
}

// the walk: n_steps steps from s_first on (64 from 0 for an off-diagonal tile pair, 32 from 1 for a diagonal one); n_steps is a multiple of PIPE
template <int JACC, bool FAR, int NU, int NI, int PIPE>
__device__ __forceinline__ void pan_walk(const double2 *__restrict__ s_xy, const double2 *__restrict__ s_zm, const double2 *__restrict__ s_mm,
                                         const double *__restrict__ s_valid, const bool pad, const int lane, const int src4,
                                         const double (&L)[3], const double (&iL)[3], const double (&q)[NI][3], const double (&m)[NI][3],
                                         const double2 *__restrict__ ab, const size_t (&ab_tile)[NI], const int s_first, const int n_steps,
                                         PanAcc<NI> &A) {
	int jb = lane + s_first;
	if (FAR) { // (never a diagonal tile pair: always 64 steps)
		const double2 none[NI] = {};
#define MPMC_FAR_LOOP(P)                                                                                                               \
	for (int kc = 0; kc < kTile - 4; kc += 4, jb += 4) {                                                                               \
		_Pragma("unroll") for (int u = 0; u < 4; ++u)                                                                                  \
		    pan_step<JACC, true, NU, NI, true, P>(s_xy, s_zm, s_mm, s_valid, jb + u, src4, L, iL, q, m, none, A);                      \
	}                                                                                                                                  \
	_Pragma("unroll") for (int u = 0; u < 3; ++u) pan_step<JACC, true, NU, NI, true, P>(s_xy, s_zm, s_mm, s_valid, jb + u, src4, L, iL, q, m, none, A); \
	pan_step<JACC, true, NU, NI, false, P>(s_xy, s_zm, s_mm, s_valid, jb + 3, src4, L, iL, q, m, none, A);
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
			pan_step<JACC, false, NU, NI, true, false>(s_xy, s_zm, s_mm, s_valid, jb + u, src4, L, iL, q, m, t, A);
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
		if (u != PIPE - 1) pan_step<JACC, false, NU, NI, true, false>(s_xy, s_zm, s_mm, s_valid, jb + u, src4, L, iL, q, m, t, A);
		else pan_step<JACC, false, NU, NI, false, false>(s_xy, s_zm, s_mm, s_valid, jb + u, src4, L, iL, q, m, t, A);
		__builtin_amdgcn_sched_barrier(0);
	}
}

template <int JACC, int PIPE, int NI>
__device__ __forceinline__ void panel_block(const AtomsDev &at, const Box &bx, const double *__restrict__ mu, const int2 *__restrict__ tile_pairs,
                                            const double4 *__restrict__ tp_shift, const double2 *__restrict__ ab, double *__restrict__ part,
                                            const int tpA, const int tpB, const int flags, const int J, double2 *__restrict__ s_xy,
                                            double2 *__restrict__ s_zm, double2 *__restrict__ s_mm, double *__restrict__ s_valid,
                                            double (*__restrict__ s_G)[kTile]) {
	const int lane = threadIdx.x & 63;
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
	switch (nonuni) {
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

	{ // j-tile into LDS, permuted, every value twice (slot l + s never wraps)
		const double4 pj = at.xyzq[j0 + lane];
		const double mx = mu[3 * (size_t)(j0 + lane)], my = mu[3 * (size_t)(j0 + lane) + 1], mz = mu[3 * (size_t)(j0 + lane) + 2];
		const double2 xy = make_double2(pick(pj.x, pj.y, pj.z, p0), pick(pj.x, pj.y, pj.z, p1));
		const double2 zm = make_double2(pick(pj.x, pj.y, pj.z, p2), pick(mx, my, mz, p0));
		const double2 mm = make_double2(pick(mx, my, mz, p1), pick(mx, my, mz, p2));
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
		const int i = I * kTile + lane;
		const double4 pi = at.xyzq[i];
		const double4 sh = tp_shift[tps[k]];
		const double px = (um & 1) ? pi.x - sh.x : pi.x, py = (um & 2) ? pi.y - sh.y : pi.y, pz = (um & 4) ? pi.z - sh.z : pi.z;
		q[k][0] = pick(px, py, pz, p0);
		q[k][1] = pick(px, py, pz, p1);
		q[k][2] = pick(px, py, pz, p2);
		const double mx = mu[3 * (size_t)i], my = mu[3 * (size_t)i + 1], mz = mu[3 * (size_t)i + 2];
		m[k][0] = pick(mx, my, mz, p0);
		m[k][1] = pick(mx, my, mz, p1);
		m[k][2] = pick(mx, my, mz, p2);
		ab_tile[k] = (size_t)tps[k] * (kTile * kTile);
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // the LDS image is this wave's own: no workgroup barrier, program order suffices
	__builtin_amdgcn_wave_barrier();
	const bool pad = (at.n != at.n_pad) && (J == at.n_pad / kTile - 1);
	const int s_first = diag ? 1 : 0, n_steps = diag ? 32 : 64;
	PanAcc<NI> A = {};
#define MPMC_PWALK(F, N) pan_walk<JACC, F, N, NI, PIPE>(s_xy, s_zm, s_mm, s_valid, pad, lane, src4, L, iL, q, m, ab, ab_tile, s_first, n_steps, A)
	if (far) {
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
	// un-permute and store.  After n_steps - 1 rotations lane l holds the accumulator of j = (l + s_first + n_steps - 1) & 63.
	const int nt_pad3 = at.n_pad * 3;
	auto unperm = [&](const double (&v)[3], double *o) { // o[p_d] = v[d] without dynamically indexed registers
		o[0] = (p0 == 0) ? v[0] : ((p1 == 0) ? v[1] : v[2]);
		o[1] = (p0 == 1) ? v[0] : ((p1 == 1) ? v[1] : v[2]);
		o[2] = (p0 == 2) ? v[0] : ((p1 == 2) ? v[1] : v[2]);
	};
	// i-side: slot [J][I_k atoms] (the diagonal tile pair: [J][J atoms]).  j-side: parked in LDS at its atom; the workgroup's four
	// waves share the j-tile, their G meet there and leave as ONE slot (k_dipole_iter_panel).
#pragma unroll
	for (int k = 0; k < NI; ++k) {
		double o[3];
		unperm(A.f[k], o);
		double *oi = part + (size_t)J * nt_pad3 + 3 * (size_t)(Is[k] * kTile + lane);
		oi[0] = o[0];
		oi[1] = o[1];
		oi[2] = o[2];
	}
	{
		const int jl_last = (lane + s_first + n_steps - 1) & 63;
		double o[3];
		unperm(A.g, o);
		s_G[0][jl_last] = o[0];
		s_G[1][jl_last] = o[1];
		s_G[2][jl_last] = o[2];
	}
}

constexpr int kPanelWaves = 4;
template <int JACC, int PIPE>
__global__ __launch_bounds__(64 * kPanelWaves) void k_dipole_iter_panel(AtomsDev at, Box bx, const double *__restrict__ mu,
                                                                        const int2 *__restrict__ tile_pairs, const double4 *__restrict__ tp_shift,
                                                                        const int4 *__restrict__ panels, const double2 *__restrict__ ab,
                                                                        double *__restrict__ part, double *__restrict__ gpart /*[n_wg][64][3]*/) {
	__shared__ double2 s_xy[kPanelWaves][2 * kTile], s_zm[kPanelWaves][2 * kTile], s_mm[kPanelWaves][2 * kTile];
	__shared__ double s_valid[kPanelWaves][2 * kTile];
	__shared__ double s_G[kPanelWaves][3][kTile];
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int4 e = panels[blockIdx.x * kPanelWaves + w];
	// everything that describes the entry is wave-uniform: keep it in scalar registers
	const int tpA = __builtin_amdgcn_readfirstlane(e.x), tpB = __builtin_amdgcn_readfirstlane(e.y);
	const int flags = __builtin_amdgcn_readfirstlane(e.z), J = __builtin_amdgcn_readfirstlane(e.w);
	if (tpA < 0) { // unused entry of this j-tile's segment
		s_G[w][0][lane] = s_G[w][1][lane] = s_G[w][2][lane] = 0.0;
	} else if (tpB >= 0) {
		panel_block<JACC, PIPE, 2>(at, bx, mu, tile_pairs, tp_shift, ab, part, tpA, tpB, flags, J, s_xy[w], s_zm[w], s_mm[w], s_valid[w], s_G[w]);
	} else {
		panel_block<JACC, PIPE, 1>(at, bx, mu, tile_pairs, tp_shift, ab, part, tpA, tpB, flags, J, s_xy[w], s_zm[w], s_mm[w], s_valid[w], s_G[w]);
	}
	__syncthreads();
	if (w != 0) return;
	double *o = gpart + ((size_t)blockIdx.x * kTile + lane) * 3; // wave order: fixed => reproducible
	o[0] = ((s_G[0][0][lane] + s_G[1][0][lane]) + s_G[2][0][lane]) + s_G[3][0][lane];
	o[1] = ((s_G[0][1][lane] + s_G[1][1][lane]) + s_G[2][1][lane]) + s_G[3][1][lane];
	o[2] = ((s_G[0][2][lane] + s_G[1][2][lane]) + s_G[2][2][lane]) + s_G[3][2][lane];
}

// new_mu = alpha (E0 + F), F = sum of the panel kernel's slots of this tile X: part[S][X atoms] for S = X .. nt-1 (i-side, the diagonal
// included) and gpart[wg][.] for the workgroups of X's segment (j-side).  Same tail as k_dipole_update (contract_dipoles :3586-3593,
// calc_dipole_rrms :3147-3177, are_we_done_yet :3227-3236).
constexpr int kUpdGroups = 8;
__global__ __launch_bounds__(64 * kUpdGroups) void k_dipole_update_panel(AtomsDev at, const double *__restrict__ e_static, const double *__restrict__ part,
                                                                         const double *__restrict__ gpart, const int *__restrict__ seg, int nt,
                                                                         const double *__restrict__ mu_old, double *__restrict__ mu_new,
                                                                         double *__restrict__ e_induced, int want_rrms, double *__restrict__ rrms_atom,
                                                                         double allowed_sqerr, int *__restrict__ not_done_flag) {
	__shared__ double sh[kUpdGroups][kTile][3];
	const int a = threadIdx.x & 63, g = threadIdx.x >> 6;
	const int X = blockIdx.x, i = X * kTile + a;
	const int nF = nt - X, wg0 = seg[X] / kPanelWaves, nG = seg[X + 1] / kPanelWaves - wg0;
	double f[3] = {0, 0, 0};
	for (int t = g; t < nF + nG; t += kUpdGroups) {
		const double *q = (t < nF) ? part + ((size_t)(X + t) * at.n_pad + i) * 3 : gpart + ((size_t)(wg0 + t - nF) * kTile + a) * 3;
		f[0] += q[0];
		f[1] += q[1];
		f[2] += q[2];
	}
	sh[g][a][0] = f[0];
	sh[g][a][1] = f[1];
	sh[g][a][2] = f[2];
	__syncthreads();
	if (g != 0) return;
	double fsum[3];
	for (int p = 0; p < 3; ++p) {
		double v = sh[0][a][p];
		for (int k = 1; k < kUpdGroups; ++k) v += sh[k][a][p];
		fsum[p] = v;
	}
	const double al = at.alpha[i];
	const bool live = (i < at.n) && (al != 0.0);
	double fo[3] = {0, 0, 0}, nm[3] = {0, 0, 0};
	if (live) {
		for (int p = 0; p < 3; ++p) {
			fo[p] = fsum[p];
			nm[p] = al * (e_static[3 * (size_t)i + p] + fo[p]);
		}
	}
	bool broke = false;
	double acc = 0, nn = 0;
	for (int p = 0; p < 3; ++p) {
		const double d = nm[p] - mu_old[3 * (size_t)i + p];
		acc += d * d;
		nn += nm[p] * nm[p];
		if (d * d > allowed_sqerr) broke = true;
		mu_new[3 * (size_t)i + p] = nm[p];
		e_induced[3 * (size_t)i + p] = fo[p];
	}
	if (want_rrms) {
		double r = sqrt(acc / nn);
		if (!isfinite(r)) r = 0.0;
		rrms_atom[i] = (i < at.n) ? r : 0.0;
	}
	if (allowed_sqerr > 0.0 && broke && i < at.n) atomicOr(not_done_flag, 1);
}

void launch_build_panels(hipStream_t st, const int *cls, int n_tiles, const int *seg, int4 *panels) {
	hipLaunchKernelGGL(k_build_panels, dim3((n_tiles + 63) / 64), dim3(64), 0, st, cls, n_tiles, seg, panels);
}

void launch_dipole_iter_panel(hipStream_t st, int jacc, const AtomsDev &at, const Box &bx, const double *mu, const int2 *tile_pairs,
                              const double4 *tp_shift, const int4 *panels, int n_entries, const double2 *ab, double *part, double *gpart) {
	if (n_entries <= 0) return;
	dim3 grid(n_entries / kPanelWaves), block(kTile * kPanelWaves);
	if (jacc == 1)
		hipLaunchKernelGGL((k_dipole_iter_panel<1, 4>), grid, block, 0, st, at, bx, mu, tile_pairs, tp_shift, panels, ab, part, gpart);
	else
		hipLaunchKernelGGL((k_dipole_iter_panel<0, 4>), grid, block, 0, st, at, bx, mu, tile_pairs, tp_shift, panels, ab, part, gpart);
}

void launch_dipole_update_panel(hipStream_t st, const AtomsDev &at, const double *e_static, const double *part, const double *gpart, const int *seg,
                                const double *mu_old, double *mu_new, double *e_induced, int want_rrms, double *rrms_atom, double allowed_sqerr,
                                int *not_done_flag) {
	hipLaunchKernelGGL(k_dipole_update_panel, dim3(at.n_pad / kTile), dim3(kTile * kUpdGroups), 0, st, at, e_static, part, gpart, seg, at.n_pad / kTile,
	                   mu_old, mu_new, e_induced, want_rrms, rrms_atom, allowed_sqerr, not_done_flag);
}

} // namespace mpmc
