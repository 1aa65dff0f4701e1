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
//                    A operand = x[k] replicated; lanes 0..15 hold the 16 column sums.
// The partial slots have the layout of the other solvers' ([chunk][n_pad][3]), so k_dipole_update finishes the iteration.
#include "kernels.h"
#include "device_math.h"

namespace mpmc {

template <bool ORTHO>
__global__ __launch_bounds__(256) void k_dense_build(AtomsDev at, Box bx, double lambda, double *__restrict__ a) {
	const int j = blockIdx.x * 256 + threadIdx.x; // column atom (slot)
	const int i = blockIdx.y;                     // row atom (slot)
	if (j >= at.n_pad) return;
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

void launch_dense_build(hipStream_t st, const AtomsDev &at, const Box &bx, double polar_damp, double *a) {
	dim3 grid((at.n_pad + 255) / 256, at.n_pad), block(256);
	if (bx.ortho) hipLaunchKernelGGL(k_dense_build<true>, grid, block, 0, st, at, bx, polar_damp, a);
	else hipLaunchKernelGGL(k_dense_build<false>, grid, block, 0, st, at, bx, polar_damp, a);
}

void launch_dense_matvec(hipStream_t st, const double *a, int n_pad, const double *x, int n_chunks, double *part) {
	const int ld = 3 * n_pad;
	const int rows = (((ld + n_chunks - 1) / n_chunks) + 3) / 4 * 4; // whole MFMA k-steps
	hipLaunchKernelGGL(k_dense_matvec, dim3(ld / 16, n_chunks), dim3(64), 0, st, a, ld, x, rows, part);
}

} // namespace mpmc
