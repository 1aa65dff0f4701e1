// kernels_gs.hip -- Gauss-Seidel dipole sweeps (`polar_gs on`, reference contract_dipoles src/System.Energy.cpp:3564-3598
// with the in-place update of :3590-3592).
//
// The reference walks the atoms in atom_array order; atom i sees the NEW dipoles of every j < i and the OLD dipoles of every
// j > i.  That order is part of the result, so this solver runs with the identity atom order (no spatial sort, context.cpp)
// and sweeps the tiles of 64 consecutive atoms one after the other:
//
//   k_gs_rows   for the 64 rows of tile I: partial induced fields from every OTHER tile J, one workgroup per J, read straight
//               from the in-place dipole array (tiles J < I already hold this sweep's values, tiles J > I the previous ones).
//               Matrix-free: T_ij is rebuilt from the positions with the reference's damping (thole_amatrix :2731-2757).
//   k_gs_tile   one wave: adds the partials in tile order, then walks the 64 atoms of tile I sequentially -- at step k every
//               lane j contributes T_kj mu_j with its CURRENT dipole, a wave sum gives row k, lane k stores its new dipole.
//   k_gs_finish rrms / "broke tolerance" flag from (mu before the sweep, mu after), as calc_dipole_rrms :3147-3177 and
//               are_we_done_yet :3227-3236 do with old_mu / new_mu.
// The sweep is inherently serial over tiles (2 launches per tile); it is here for coverage of the reference's option, the
// production path is the Jacobi iteration of kernels_sym.hip.
#include "kernels.h"
#include "device_math.h"

namespace mpmc {

template <bool ORTHO>
__global__ __launch_bounds__(64) void k_gs_rows(AtomsDev at, Box bx, double lambda, const double *__restrict__ mu, int I,
                                                double *__restrict__ part /*[nt][64][3]*/) {
	__shared__ double4 s_xyzq[kTile];
	__shared__ double s_mu[kTile * 3];
	__shared__ int s_fl[kTile];
	const int lane = threadIdx.x, J = blockIdx.x;
	double fx = 0, fy = 0, fz = 0;
	if (J != I) {
		const int i = I * kTile + lane;
		const double4 pi = at.xyzq[i];
		const int jg = J * kTile + lane;
		s_xyzq[lane] = at.xyzq[jg];
		s_fl[lane] = at.mf[jg].y;
		s_mu[3 * lane + 0] = mu[3 * (size_t)jg + 0];
		s_mu[3 * lane + 1] = mu[3 * (size_t)jg + 1];
		s_mu[3 * lane + 2] = mu[3 * (size_t)jg + 2];
		__syncthreads();
		for (int jj = 0; jj < kTile; ++jj) {
			if (s_fl[jj] & (AF_PAD | AF_ZERO_ALPHA)) continue; // mu_j == 0 for non-polarizable sites (:3571-3576)
			const double4 pj = s_xyzq[jj];
			double ox, oy, oz;
			const double r = min_image<ORTHO>(bx, pi.x - pj.x, pi.y - pj.y, pi.z - pj.z, ox, oy, oz);
			double a, b;
			thole_ab(r, lambda, a, b);
			const double mx = s_mu[3 * jj], my = s_mu[3 * jj + 1], mz = s_mu[3 * jj + 2];
			const double t3 = b * (((ox * mx) + oy * my) + oz * mz);
			fx -= a * mx - t3 * ox;
			fy -= a * my - t3 * oy;
			fz -= a * mz - t3 * oz;
		}
	}
	double *o = part + ((size_t)J * kTile + lane) * 3;
	o[0] = fx;
	o[1] = fy;
	o[2] = fz;
}

__device__ __forceinline__ double wave_sum_all(double v) { // every lane receives the total (fixed butterfly order)
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}

template <bool ORTHO>
__global__ __launch_bounds__(64) void k_gs_tile(AtomsDev at, Box bx, double lambda, const double *__restrict__ e_static,
                                                const double *__restrict__ part, int n_tiles, int I, double *__restrict__ mu,
                                                double *__restrict__ e_induced) {
	const int lane = threadIdx.x;
	const int i = I * kTile + lane;
	const double4 p = at.xyzq[i];
	const double al = at.alpha[i];
	const bool live = (i < at.n) && (al != 0.0) && !(at.mf[i].y & AF_PAD);
	double acc[3] = {0, 0, 0};
	for (int J = 0; J < n_tiles; ++J) { // tile order = the reference's j order up to association
		if (J == I) continue;
		const double *q = part + ((size_t)J * kTile + lane) * 3;
		acc[0] += q[0];
		acc[1] += q[1];
		acc[2] += q[2];
	}
	double m[3] = {mu[3 * (size_t)i], mu[3 * (size_t)i + 1], mu[3 * (size_t)i + 2]};
	if (!live) m[0] = m[1] = m[2] = 0.0;
	const double e0[3] = {e_static[3 * (size_t)i], e_static[3 * (size_t)i + 1], e_static[3 * (size_t)i + 2]};
	double eind[3] = {0, 0, 0};
	for (int k = 0; k < kTile; ++k) {
		// row k of the tile: lane j supplies - T_kj mu_j (displacement = pos_k - pos_j, the pair order of minimum_image :1202)
		const double kx = __shfl(p.x, k, 64), ky = __shfl(p.y, k, 64), kz = __shfl(p.z, k, 64);
		double cx = 0, cy = 0, cz = 0;
		if (lane != k && live) {
			double ox, oy, oz;
			const double r = min_image<ORTHO>(bx, kx - p.x, ky - p.y, kz - p.z, ox, oy, oz);
			double a, b;
			thole_ab(r, lambda, a, b);
			const double t3 = b * (((ox * m[0]) + oy * m[1]) + oz * m[2]);
			cx = -(a * m[0] - t3 * ox);
			cy = -(a * m[1] - t3 * oy);
			cz = -(a * m[2] - t3 * oz);
		}
		const double sx = wave_sum_all(cx), sy = wave_sum_all(cy), sz = wave_sum_all(cz);
		if (lane == k) {
			if (live) {
				eind[0] = acc[0] + sx;
				eind[1] = acc[1] + sy;
				eind[2] = acc[2] + sz;
				m[0] = al * (e0[0] + eind[0]); // :3586-3592: new_mu, and mu = new_mu at once
				m[1] = al * (e0[1] + eind[1]);
				m[2] = al * (e0[2] + eind[2]);
			}
		}
	}
	for (int d = 0; d < 3; ++d) {
		mu[3 * (size_t)i + d] = m[d];
		e_induced[3 * (size_t)i + d] = eind[d];
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

void launch_gs_sweep(hipStream_t st, const AtomsDev &at, const Box &bx, double polar_damp, const double *e_static, double *mu, double *e_induced,
                     double *part) {
	const int nt = at.n_pad / kTile;
	for (int I = 0; I < nt; ++I) {
		if (bx.ortho) {
			hipLaunchKernelGGL(k_gs_rows<true>, dim3(nt), dim3(kTile), 0, st, at, bx, polar_damp, mu, I, part);
			hipLaunchKernelGGL(k_gs_tile<true>, dim3(1), dim3(kTile), 0, st, at, bx, polar_damp, e_static, part, nt, I, mu, e_induced);
		} else {
			hipLaunchKernelGGL(k_gs_rows<false>, dim3(nt), dim3(kTile), 0, st, at, bx, polar_damp, mu, I, part);
			hipLaunchKernelGGL(k_gs_tile<false>, dim3(1), dim3(kTile), 0, st, at, bx, polar_damp, e_static, part, nt, I, mu, e_induced);
		}
	}
}

void launch_gs_finish(hipStream_t st, const AtomsDev &at, const double *mu_old, const double *mu_new, int want_rrms, double *rrms_atom,
                      double allowed_sqerr, int *not_done_flag) {
	hipLaunchKernelGGL(k_gs_finish, dim3((at.n_pad + 255) / 256), dim3(256), 0, st, at, mu_old, mu_new, want_rrms, rrms_atom, allowed_sqerr,
	                   not_done_flag);
}

} // namespace mpmc
