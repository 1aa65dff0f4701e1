// kernels_gs.hip -- Gauss-Seidel dipole sweeps (`polar_gs on`, reference contract_dipoles src/System.Energy.cpp:3564-3598
// with the in-place update of :3590-3592).
//
// The reference walks the atoms in atom_array order; atom i sees the NEW dipoles of every j < i and the OLD dipoles of every
// j > i.  That order is part of the result, so this solver runs with the identity atom order (no spatial sort, context.cpp).
// With tiles of 64 consecutive atoms, the induced field of an atom of tile K is
//     U (tiles > K, old dipoles)  +  L (tiles < K, this sweep's dipoles)  +  the in-tile part (old for j > i, new for j < i).
// Round 3 (the round-2 review's item 10): the sweep is a pipeline instead of 2 x 157 launches of 113 us per tile --
//   * once per evaluation, k_gs_blocks: the in-tile 3 x 3 blocks behind the diagonal, times -alpha of their row (packed lower triangle,
//     96 KB per tile: they depend on positions and polarizabilities only);
//   * once per sweep, everything that comes from OLD dipoles: the matrix-free symmetric Jacobi kernel (k_dipole_iter_hybrid, null store)
//     gives every tile pair's row sums; k_gs_upper_sum adds, per tile, the slots of the tiles above it (tile order) and the in-tile atoms
//     behind each row's own -> U;
//   * then ONE launch per tile K (k_gs_stage, nt - K workgroups of 16 waves): workgroup b adds T_{K+b, K-1} mu_{K-1}^{new} -- the dipoles
//     the launch before has just finished -- into L of tile K + b; workgroup 0 goes on to walk tile K: its blocks were requested from global
//     memory before the push and are in LDS by now; each lane carries its candidate dipole y = alpha (E0 + row); the y of the atom whose
//     turn it is is final, is read with v_readlane, and every later lane adds (-alpha_j T_jk) y_k to its own -- nine fma per step, NO wave
//     reduction, no LDS round trip on the dependency chain.
//   Nothing spins on a flag: the order is the stream's.  Sums are taken in tile order, then lane order: reproducible.
//   k_gs_finish: rrms / "broke tolerance" flag from (mu before the sweep, mu after), as calc_dipole_rrms :3147-3177 and
//   are_we_done_yet :3227-3236 do with old_mu / new_mu.
#include "kernels.h"
#include "device_math.h"

namespace mpmc {

constexpr int kGsWaves = 16;

// (a, b) of thole_amatrix (:2731-2757) from the squared image distance; r = 0 gives (0, 0): the reference's MAXVALUE guard times its
// vanishing damping factors (:2704-2705)
__device__ __forceinline__ double2 gs_thole_ab(double r2, double lambda) {
	const double ir = fast_rsqrt(r2);
	const double r = r2 * ir;
	const double ir3 = ir * ir * ir, ir5 = ir3 * ir * ir;
	const double lr = lambda * r;
	const double explr = exp_fast(-lr);
	const double damp1 = fma(-explr, fma(lr, fma(0.5, lr, 1.0), 1.0), 1.0);
	const double damp2 = fma(-explr, (lr * lr) * (lr * (1.0 / 6.0)), damp1);
	return (r2 > 0.0) ? make_double2(damp1 * ir3, 3.0 * damp2 * ir5) : make_double2(0.0, 0.0);
}

// value of lane k (wave-uniform k) of a double: two v_readlane_b32, no LDS round trip
__device__ __forceinline__ double gs_lane_value(double v, int k) {
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
	return __hiloint2double(hi, lo);
}

// slot of (column k, row j > k) in the packed strictly-lower triangle of a 64 x 64 block, columns one after the other.  The walk reads
// slot kTile + gs_tri(k) + lane - k - 1 in EVERY lane, one step ahead of its use -- hence kTile slots in front: the lanes at or in front
// of k read someone else's numbers and drop them.
__host__ __device__ constexpr int gs_tri(int k) { return (kTile - 1) * k - (k * (k - 1)) / 2; }
constexpr int kGsTri = gs_tri(kTile - 1);    // 2016 pairs
constexpr int kGsSlots = kTile + kGsTri;     // slots of one component array of one tile
constexpr int kGsTileBlock = 3 * kGsSlots;   // double2 elements per tile: (xx, xy) | (xz, yy) | (yz, zz)
size_t gs_block_store_elements(int n_tiles) { return (size_t)n_tiles * kGsTileBlock; }

// Once per evaluation: -alpha_j T_jk of every in-tile pair behind the diagonal (k < j) -- what the walk of k_gs_stage applies to the NEW dipole
// of k -- as whole 3 x 3 blocks, so that the walk carries the row's candidate dipole y = alpha (E0 + row) itself and spends nine fma per
// step on it.  One workgroup per tile, lane = row j, wave w the columns [kShare w, kShare (w + 1)).
template <bool ORTHO>
__global__ __launch_bounds__(64 * kGsWaves) void k_gs_blocks(AtomsDev at, Box bx, double lambda, double2 *__restrict__ blocks) {
	__shared__ double4 s_pos[kTile];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int i = blockIdx.x * kTile + lane;
	const double4 pi = at.xyzq[i];
	if (w == 0) s_pos[lane] = pi;
	const double al_raw = at.alpha[i];
	const bool live = (i < at.n) && (al_raw != 0.0) && !(at.mf[i].y & AF_PAD);
	const double al = live ? al_raw : 0.0; // (a row that is not live gets zeros)
	__syncthreads();
	double2 *__restrict__ t0 = blocks + (size_t)blockIdx.x * kGsTileBlock, *__restrict__ t1 = t0 + kGsSlots, *__restrict__ t2 = t1 + kGsSlots;
	constexpr int kShare = kTile / kGsWaves;
	for (int c = w * kShare; c < (w + 1) * kShare; ++c) {
		if (c >= lane) continue;
		const double4 pc = s_pos[c];
		double ox, oy, oz;
		const double r2 = min_image_sq<ORTHO>(bx, pi.x - pc.x, pi.y - pc.y, pi.z - pc.z, ox, oy, oz);
		const double2 t = gs_thole_ab(r2, lambda);
		const double a = -al * t.x, bx_ = (al * t.y) * ox, by_ = (al * t.y) * oy, bz_ = (al * t.y) * oz;
		const int slot = kTile + gs_tri(c) + (lane - c - 1);
		t0[slot] = make_double2(fma(bx_, ox, a), bx_ * oy);
		t1[slot] = make_double2(bx_ * oz, fma(by_, oy, a));
		t2[slot] = make_double2(by_ * oz, fma(bz_, oz, a));
	}
}

// Once per sweep, one workgroup per tile I: what the rows of the tile get from OLD dipoles --
//   the slots the symmetric kernel wrote for the tiles above, part[J][i], J > I, in tile order (wave 0), and
//   the in-tile atoms behind each row's own (column c > lane), wave w the columns [kShare w, kShare (w + 1)) --
// summed into U; L (the part from this sweep's dipoles) starts from zero.
template <bool ORTHO>
__global__ __launch_bounds__(64 * kGsWaves) void k_gs_upper_sum(AtomsDev at, Box bx, double lambda, const double *__restrict__ mu,
                                                               const double *__restrict__ part, int nt, double *__restrict__ U, double *__restrict__ L) {
	__shared__ double4 s_pos[kTile];
	__shared__ double s_mu[3][kTile];
	__shared__ double s_p[kGsWaves][3][kTile];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int I = blockIdx.x, i = I * kTile + lane;
	const double4 pi = at.xyzq[i];
	if (w == 0) {
		const bool live = (i < at.n) && (at.alpha[i] != 0.0) && !(at.mf[i].y & AF_PAD);
		s_pos[lane] = pi;
		for (int d = 0; d < 3; ++d) s_mu[d][lane] = live ? mu[3 * (size_t)i + d] : 0.0;
	}
	__syncthreads();
	double f[3] = {0, 0, 0};
	constexpr int kShare = kTile / kGsWaves;
	for (int c = w * kShare; c < (w + 1) * kShare; ++c) {
		if (c <= lane) continue;
		const double4 pc = s_pos[c];
		double ox, oy, oz;
		const double r2 = min_image_sq<ORTHO>(bx, pi.x - pc.x, pi.y - pc.y, pi.z - pc.z, ox, oy, oz);
		const double2 t = gs_thole_ab(r2, lambda);
		const double mx = s_mu[0][c], my = s_mu[1][c], mz = s_mu[2][c];
		const double t3 = t.y * fma(oz, mz, fma(oy, my, ox * mx));
		f[0] = fma(-t.x, mx, fma(t3, ox, f[0]));
		f[1] = fma(-t.x, my, fma(t3, oy, f[1]));
		f[2] = fma(-t.x, mz, fma(t3, oz, f[2]));
	}
	for (int d = 0; d < 3; ++d) s_p[w][d][lane] = f[d];
	double up[3] = {0, 0, 0};
	if (w == 0) {
#pragma unroll 4
		for (int J = I + 1; J < nt; ++J) {
			const double *q = part + ((size_t)J * at.n_pad + i) * 3;
			up[0] += q[0];
			up[1] += q[1];
			up[2] += q[2];
		}
	}
	__syncthreads();
	if (w != 0) return;
	for (int d = 0; d < 3; ++d) {
		double sum = s_p[0][d][lane];
#pragma unroll
		for (int v = 1; v < kGsWaves; ++v) sum += s_p[v][d][lane];
		U[3 * (size_t)i + d] = up[d] + sum;
		L[3 * (size_t)i + d] = 0.0; // the lower part starts every sweep from zero
	}
}

// One launch per tile K, nt - K workgroups: workgroup b adds T_{K+b, K-1} mu_{K-1}^{new} -- the dipoles the launch before has just
// finished -- into L of tile K + b; workgroup 0 goes on to walk tile K.
template <bool ORTHO>
__global__ __launch_bounds__(64 * kGsWaves) void k_gs_stage(AtomsDev at, Box bx, double lambda, const double *__restrict__ e_static,
                                                           const double *__restrict__ U, double *__restrict__ L, const double2 *__restrict__ blocks,
                                                           int K, double *__restrict__ mu, double *__restrict__ e_induced) {
	__shared__ double4 s_pos[kTile];          // source tile (K - 1) of the push
	__shared__ double s_mu[3][kTile];
	__shared__ double s_p[kGsWaves][3][kTile]; // partial sums of the waves
	__shared__ double2 s_T[kGsTileBlock];      // workgroup 0: this tile's blocks (k_gs_blocks)
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const int J = K + (int)blockIdx.x;
	const int i = J * kTile + lane;
	const double4 pi = at.xyzq[i];
	constexpr int kShare = kTile / kGsWaves; // source atoms per wave
	// what the walk needs from global memory is requested before the push, so that it has arrived by then (workgroup 0 only)
	constexpr int kStage = (kGsTileBlock + 64 * kGsWaves - 1) / (64 * kGsWaves);
	double2 stage[kStage];
	double e0[3] = {0, 0, 0}, up[3] = {0, 0, 0}, al = 0.0;
	bool live = false;
	if (blockIdx.x == 0) {
		const double2 *__restrict__ src = blocks + (size_t)K * kGsTileBlock;
#pragma unroll
		for (int q = 0; q < kStage; ++q) {
			const int e = q * (64 * kGsWaves) + (int)threadIdx.x;
			stage[q] = (e < kGsTileBlock) ? src[e] : make_double2(0.0, 0.0);
		}
		if (w == 0) {
			const double al_raw = at.alpha[i];
			live = (i < at.n) && (al_raw != 0.0) && !(at.mf[i].y & AF_PAD);
			al = live ? al_raw : 0.0;
			if (live)
				for (int d = 0; d < 3; ++d) {
					e0[d] = e_static[3 * (size_t)i + d];
					up[d] = U[3 * (size_t)i + d];
				}
		}
	}
	double low[3] = {0, 0, 0};
	if (K > 0) { // L_J += T_{J, K-1} mu_{K-1}: lane = row atom, the waves share the 64 source atoms
		if (w == 0) {
			const int sg = (K - 1) * kTile + lane;
			s_pos[lane] = at.xyzq[sg];
			const bool src_live = !(at.mf[sg].y & (AF_PAD | AF_ZERO_ALPHA));
			for (int d = 0; d < 3; ++d) s_mu[d][lane] = src_live ? mu[3 * (size_t)sg + d] : 0.0;
		}
		__syncthreads();
		double f[3] = {0, 0, 0};
#pragma unroll 2
		for (int jj = w * kShare; jj < (w + 1) * kShare; ++jj) {
			const double4 pj = s_pos[jj];
			double ox, oy, oz;
			const double r2 = min_image_sq<ORTHO>(bx, pi.x - pj.x, pi.y - pj.y, pi.z - pj.z, ox, oy, oz);
			const double2 t = gs_thole_ab(r2, lambda);
			const double mx = s_mu[0][jj], my = s_mu[1][jj], mz = s_mu[2][jj];
			const double t3 = t.y * fma(oz, mz, fma(oy, my, ox * mx));
			f[0] = fma(-t.x, mx, fma(t3, ox, f[0]));
			f[1] = fma(-t.x, my, fma(t3, oy, f[1]));
			f[2] = fma(-t.x, mz, fma(t3, oz, f[2]));
		}
		for (int d = 0; d < 3; ++d) s_p[w][d][lane] = f[d];
	}
	if (blockIdx.x == 0) {
#pragma unroll
		for (int q = 0; q < kStage; ++q) {
			const int e = q * (64 * kGsWaves) + (int)threadIdx.x;
			if (e < kGsTileBlock) s_T[e] = stage[q];
		}
	}
	__syncthreads();
	if (w != 0) return;
	if (K > 0)
		for (int d = 0; d < 3; ++d) {
			double sum = s_p[0][d][lane];
#pragma unroll
			for (int v = 1; v < kGsWaves; ++v) sum += s_p[v][d][lane];
			low[d] = L[3 * (size_t)i + d] + sum;
			L[3 * (size_t)i + d] = low[d];
		}
	if (blockIdx.x != 0) return; // (every workgroup but the first has only pushed)
	// ---- walk tile K (:3586-3592) -----------------------------------------------------------------------------------------------
	// y = alpha (E0 + row), row = the field pieces that are complete before the walk: old dipoles (U), tiles below (L).  Atom k's row is
	// complete when its turn comes (everything in front of it has been added), so its y IS its new dipole; it is read with v_readlane and
	// every later atom of the tile takes y_j += (-alpha_j T_jk) y_k at once.  One wave, no reduction; the blocks of column k + 1 are read
	// from LDS while column k is applied.
	const double2 *__restrict__ s_T0 = s_T, *__restrict__ s_T1 = s_T + kGsSlots, *__restrict__ s_T2 = s_T + 2 * kGsSlots;
	double y[3];
	for (int d = 0; d < 3; ++d) y[d] = al * (e0[d] + (up[d] + low[d]));
	double2 t0 = s_T0[kTile + lane - 1], t1 = s_T1[kTile + lane - 1], t2 = s_T2[kTile + lane - 1]; // column 0
#pragma unroll
	for (int k = 0; k < kTile - 1; ++k) {
		const int next = (k + 1 < kTile - 1) ? kTile + gs_tri(k + 1) + (lane - k - 2) : 0;
		const double2 n0 = s_T0[next], n1 = s_T1[next], n2 = s_T2[next];
		const double mkx = gs_lane_value(y[0], k), mky = gs_lane_value(y[1], k), mkz = gs_lane_value(y[2], k);
		if (lane > k) {
			y[0] = fma(t1.x, mkz, fma(t0.y, mky, fma(t0.x, mkx, y[0])));
			y[1] = fma(t2.x, mkz, fma(t1.y, mky, fma(t0.y, mkx, y[1])));
			y[2] = fma(t2.y, mkz, fma(t2.x, mky, fma(t1.x, mkx, y[2])));
		}
		t0 = n0;
		t1 = n1;
		t2 = n2;
	}
	for (int d = 0; d < 3; ++d) {
		mu[3 * (size_t)i + d] = y[d]; // (0 for the slots that are not live)
		e_induced[3 * (size_t)i + d] = live ? y[d] / al - e0[d] : 0.0; // the row sum the dipole was made from
	}
}

__global__ __launch_bounds__(256) void k_gs_finish(AtomsDev at, const double *__restrict__ mu_old, const double *__restrict__ mu_new, int want_rrms,
                                                   double *__restrict__ rrms_atom, double allowed_sqerr, int *__restrict__ not_done_flag) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= at.n_pad) return;
	bool broke = false;
	double acc = 0, nn = 0;
	for (int p = 0; p < 3; ++p) {
		const double nm = mu_new[3 * (size_t)i + p];
		const double d = nm - mu_old[3 * (size_t)i + p];
		acc += d * d;
		nn += nm * nm;
		if (d * d > allowed_sqerr) broke = true;
	}
	if (want_rrms) {
		double r = sqrt(acc / nn);
		if (!isfinite(r)) r = 0.0;
		rrms_atom[i] = (i < at.n) ? r : 0.0;
	}
	if (allowed_sqerr > 0.0 && broke && i < at.n) atomicOr(not_done_flag, 1);
}

void launch_gs_blocks(hipStream_t st, const AtomsDev &at, const Box &bx, double polar_damp, double2 *blocks) {
	const int nt = at.n_pad / kTile;
	if (bx.ortho) hipLaunchKernelGGL(k_gs_blocks<true>, dim3(nt), dim3(kTile * kGsWaves), 0, st, at, bx, polar_damp, blocks);
	else hipLaunchKernelGGL(k_gs_blocks<false>, dim3(nt), dim3(kTile * kGsWaves), 0, st, at, bx, polar_damp, blocks);
}

void launch_gs_sweep(hipStream_t st, const AtomsDev &at, const Box &bx, double polar_damp, const double *e_static, double *mu, double *e_induced,
                     double *part, const int2 *tile_pairs, const int *cls, const double4 *tp_shift, int n_tile_pairs, double *U, double *L,
                     const double2 *blocks) {
	const int nt = at.n_pad / kTile;
	// old dipoles: every tile pair's row sums in one launch, then per tile the slots of the tiles above it + the in-tile atoms behind each row
	launch_dipole_iter_hybrid(st, at, bx, mu, tile_pairs, cls, tp_shift, n_tile_pairs, nullptr, part, polar_damp, nullptr);
	if (bx.ortho) hipLaunchKernelGGL(k_gs_upper_sum<true>, dim3(nt), dim3(kTile * kGsWaves), 0, st, at, bx, polar_damp, mu, part, nt, U, L);
	else hipLaunchKernelGGL(k_gs_upper_sum<false>, dim3(nt), dim3(kTile * kGsWaves), 0, st, at, bx, polar_damp, mu, part, nt, U, L);
	for (int K = 0; K < nt; ++K) {
		if (bx.ortho) hipLaunchKernelGGL(k_gs_stage<true>, dim3(nt - K), dim3(kTile * kGsWaves), 0, st, at, bx, polar_damp, e_static, U, L, blocks, K, mu, e_induced);
		else hipLaunchKernelGGL(k_gs_stage<false>, dim3(nt - K), dim3(kTile * kGsWaves), 0, st, at, bx, polar_damp, e_static, U, L, blocks, K, mu, e_induced);
	}
}

void launch_gs_finish(hipStream_t st, const AtomsDev &at, const double *mu_old, const double *mu_new, int want_rrms, double *rrms_atom,
                      double allowed_sqerr, int *not_done_flag) {
	hipLaunchKernelGGL(k_gs_finish, dim3((at.n_pad + 255) / 256), dim3(256), 0, st, at, mu_old, mu_new, want_rrms, rrms_atom, allowed_sqerr,
	                   not_done_flag);
}

} // namespace mpmc
