// kernels_dense.hip -- the reference's own data layout for the dipole solve: the dense 3N x 3N matrix of thole_amatrix
// (src/System.Energy.cpp:2661-2770) in device memory and contract_dipoles (:3564-3598) as a dense matrix-vector product on the
// fp64 matrix cores (`solver = MPMC_SOLVER_DENSE`; BASELINE configs[3] "dense 3N MFMA").
//
// It exists to be MEASURED next to the production path, not to be fast: one contraction reads (3N)^2 x 8 B = 7.2 GB at 10 000 atoms
// against 0.34 GB of the compact store (16 B per unordered pair instead of 72 B per ordered pair, and nothing at all beyond the
// damping range), so it is HBM-bound at ~9x the bytes.  A matrix-VECTOR product has no operand reuse: v_mfma_f64_16x16x4_f64 is fed
// with the vector replicated over the 16 rows of its A operand, i.e. 1/16 of its flops are useful, and even so the matrix cores idle
// behind HBM (16 B/clk/SIMD of matrix operand = 33 TB/s of appetite against 8 TB/s of supply).
//
//   k_dense_build    A_off[3i+p][3j+q] = delta_pq d1/r^3 - 3 d_p d_q d2/r^5 for i != j (slot order, padded slots zero), diagonal
//                    3x3 blocks zero: the contraction skips j == i (:3578), the 1/alpha diagonal never enters it.
//   k_dense_matvec   part[chunk][n] = - sum_{k in chunk} x[k] A_off[k][n]   ( = -(A x)[n] by symmetry of A ): one wave per
//                    (16 columns, row chunk); B operand = 4 rows x 16 consecutive columns of A (four 128-byte segments per MFMA),
//                    A operand = x[k] replicated; lanes 0..15 hold the 16 column sums.  (rounds 1-3; `dense_symmetric = 0`)
//   k_dense_symv     (round 4, the default) thole_amatrix fills A symmetrically (:2748-2757), so only the upper BLOCK triangle is read:
//                    one workgroup per tile pair (I <= J) of 192 x 192 doubles, and every 16 x 16 block of it feeds TWO products --
//                    y_J += x_I^T M on the matrix cores (B operand = 4 rows x 16 columns, as above) and z_I += M x_J on the vector unit
//                    (one fma per loaded register, a 16-lane sum per row) -- i.e. half the bytes of the matrix per contraction.
// The partial slots have the layout of the other solvers' ([source tile or chunk][n_pad][3]), so k_dipole_update finishes the iteration.
#include "kernels.h"
#include "device_math.h"

namespace mpmc {

template <bool ORTHO>
__global__ __launch_bounds__(256) void k_dense_build(AtomsDev at, Box bx, double lambda, double *__restrict__ a, int upper_only) {
	const int j = blockIdx.x * 256 + threadIdx.x; // column atom (slot)
	const int i = blockIdx.y;                     // row atom (slot)
	if (j >= at.n_pad) return;
	if (upper_only && (j / kTile) < (i / kTile)) return; // (the symmetric contraction reads tile pairs I <= J only; the rest stays as allocated: zero)
	const size_t ld = 3 * (size_t)at.n_pad;
	double *blk = a + (3 * (size_t)i) * ld + 3 * (size_t)j;
	double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	const bool live = (i != j) && !(at.mf[i].y & AF_PAD) && !(at.mf[j].y & AF_PAD);
	if (live) {
		const int lo = min(i, j), hi = max(i, j);
		const double4 pl = at.xyzq[lo], ph = at.xyzq[hi];
		double d[3];
		const double r = min_image<ORTHO>(bx, pl.x - ph.x, pl.y - ph.y, pl.z - ph.z, d[0], d[1], d[2]);
		double ta, tb;
		thole_ab(r, lambda, ta, tb); // ta = d1/r^3, tb = 3 d2/r^5
		for (int p = 0; p < 3; ++p)
			for (int q = 0; q < 3; ++q) v[3 * p + q] = ((p == q) ? ta : 0.0) - tb * d[p] * d[q];
	}
	for (int p = 0; p < 3; ++p)
		for (int q = 0; q < 3; ++q) blk[p * ld + q] = v[3 * p + q];
}

typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k_dense_matvec(const double *__restrict__ a, int ld, const double *__restrict__ x, int rows_per_chunk,
                                                     double *__restrict__ part /*[chunks][ld]*/) {
	const int lane = threadIdx.x;
	const int n0 = blockIdx.x * 16;
	const int k_begin = blockIdx.y * rows_per_chunk, k_end = min(ld, k_begin + rows_per_chunk);
	const int kk = lane >> 4, nn = lane & 15;
	v4f64 acc = {0.0, 0.0, 0.0, 0.0};
	const double *col = a + n0 + nn;
	for (int k0 = k_begin; k0 < k_end; k0 += 4) {
		const int k = k0 + kk;
		const double xa = x[k];                                       // A operand: A[m][kk] = x[k0 + kk] for every m
		const double bm = __builtin_nontemporal_load(col + (size_t)k * ld); // B operand: B[kk][nn] = A_off[k0 + kk][n0 + nn]
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, bm, acc, 0, 0, 0);
	}
	// D[m][n] is the same for every m (identical A rows); lanes 0..15 hold n = lane of rows 0..3
	if (lane < 16) part[(size_t)blockIdx.y * ld + n0 + lane] = -acc[0];
}

// Upper block triangle only.  Workgroup = tile pair (I <= J): rows 192 I .. 192 I + 191, columns 192 J ..; four waves, wave w owns the
// column strips { w, w + 4, w + 8 } (16 columns each) for all twelve 16-row blocks.  Every 16 x 16 block is loaded ONCE, as four
// registers of 4 rows x 16 columns (four 128-byte segments per load), and feeds two products:
//   y (columns, contraction over rows) on the matrix cores: B operand = the register, A operand = x_I replicated; three accumulators
//       per wave, one per strip, alive over the twelve row blocks; nobody else touches those columns, so lanes 0..15 write the strip's
//       sums straight into the slot part[I][J atoms];
//   z (rows, contraction over columns) on the vector unit: one fma per register (x_J of the lane's column), the 16 lanes of a row are
//       added by four xor-shuffles once per row block and register, the wave's partial rows wait in LDS and the four waves' parts are
//       added in wave order (fixed) into the slot part[J][I atoms].  (Both products on the matrix cores -- the round's first form, the
//       same 2 KB loaded a second time in the A operand's layout -- ran at 0.99 ms per contraction: 1152 MFMAs of 64 cycles per tile
//       pair with 15 of 16 columns wasted are then as long as the HBM stream.)
// Diagonal tile pairs (both triangles of the block are stored) take the y product alone.  Padded rows / columns are zeros in A.
// The loads of row block k + 1 are issued before the products of row block k (two register sets).
__global__ __launch_bounds__(256) void k_dense_symv(const double *__restrict__ a, int ld, int n_pad, const double *__restrict__ x,
                                                    const int2 *__restrict__ tile_pairs, double *__restrict__ part) {
	__shared__ double s_xi[3 * kTile], s_z[4][3 * kTile];
	const int2 IJ = tile_pairs[blockIdx.x];
	const int I = IJ.x, J = IJ.y;
	const bool diag = (I == J);
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	if (threadIdx.x < 3 * kTile) s_xi[threadIdx.x] = x[(size_t)I * (3 * kTile) + threadIdx.x];
	const int hi = lane >> 4, lo = lane & 15; // the register's element: row hi of 4, column lo of 16
	double xj[3];
#pragma unroll
	for (int t = 0; t < 3; ++t) xj[t] = x[(size_t)J * (3 * kTile) + (w + 4 * t) * 16 + lo];
	__syncthreads();
	const double *base = a + (size_t)I * (3 * kTile) * ld + (size_t)J * (3 * kTile) + (size_t)hi * ld + w * 16 + lo;
	v4f64 acc_y[3] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
	double cur[3][4], nxt[3][4];
	auto load_block = [&](double (&dst)[3][4], const int kb) {
		const double *rows = base + (size_t)(kb * 16) * ld;
#pragma unroll
		for (int t = 0; t < 3; ++t)
#pragma unroll
			for (int r = 0; r < 4; ++r) dst[t][r] = __builtin_nontemporal_load(rows + (size_t)(4 * r) * ld + 64 * t);
	};
	load_block(cur, 0);
	for (int kb = 0; kb < 12; ++kb) {
		if (kb < 11) load_block(nxt, kb + 1);
		double zp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const double xi = s_xi[kb * 16 + 4 * r + hi];
#pragma unroll
			for (int t = 0; t < 3; ++t) {
				acc_y[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, cur[t][r], acc_y[t], 0, 0, 0);
				zp[r] = fma(cur[t][r], xj[t], zp[r]);
			}
		}
		if (!diag) {
#pragma unroll
			for (int r = 0; r < 4; ++r) { // row 4 r + hi of the block: its 16 column lanes
				double v = zp[r];
				v += __shfl_xor(v, 1, 64);
				v += __shfl_xor(v, 2, 64);
				v += __shfl_xor(v, 4, 64);
				v += __shfl_xor(v, 8, 64);
				if (lo == 0) s_z[w][kb * 16 + 4 * r + hi] = v;
			}
		}
#pragma unroll
		for (int t = 0; t < 3; ++t)
#pragma unroll
			for (int r = 0; r < 4; ++r) cur[t][r] = nxt[t][r];
	}
	// y: D[m][n] is the same for every m (identical A rows); lanes 0..15 hold column n = lane (row 0 in register 0)
	if (lane < 16) {
#pragma unroll
		for (int t = 0; t < 3; ++t) part[((size_t)I * n_pad + (size_t)J * kTile) * 3 + (w + 4 * t) * 16 + lane] = -acc_y[t][0];
	}
	if (diag) return;
	__syncthreads();
	if (threadIdx.x < 3 * kTile)
		part[((size_t)J * n_pad + (size_t)I * kTile) * 3 + threadIdx.x] =
		    -(((s_z[0][threadIdx.x] + s_z[1][threadIdx.x]) + s_z[2][threadIdx.x]) + s_z[3][threadIdx.x]);
}

void launch_dense_symv(hipStream_t st, const double *a, int n_pad, const double *x, const int2 *tile_pairs, int n_tile_pairs, double *part) {
	hipLaunchKernelGGL(k_dense_symv, dim3(n_tile_pairs), dim3(256), 0, st, a, 3 * n_pad, n_pad, x, tile_pairs, part);
}

void launch_dense_build(hipStream_t st, const AtomsDev &at, const Box &bx, double polar_damp, double *a, bool upper_only) {
	dim3 grid((at.n_pad + 255) / 256, at.n_pad), block(256);
	if (bx.ortho) hipLaunchKernelGGL(k_dense_build<true>, grid, block, 0, st, at, bx, polar_damp, a, upper_only ? 1 : 0);
	else hipLaunchKernelGGL(k_dense_build<false>, grid, block, 0, st, at, bx, polar_damp, a, upper_only ? 1 : 0);
}

void launch_dense_matvec(hipStream_t st, const double *a, int n_pad, const double *x, int n_chunks, double *part) {
	const int ld = 3 * n_pad;
	const int rows = (((ld + n_chunks - 1) / n_chunks) + 3) / 4 * 4; // whole MFMA k-steps
	hipLaunchKernelGGL(k_dense_matvec, dim3(ld / 16, n_chunks), dim3(64), 0, st, a, ld, x, rows, part);
}

} // namespace mpmc
