// kernels_delta.hip -- trial-move ("delta") energies: pair energies, structure factors and (polarizable boxes) the real-space static field.
//
// The reference avoids recomputing unchanged pairs through its per-pair cache (`recalculate_energy`, set by
// minimum_image when a pair's raw displacement changed, System.cpp:1211-1224; lj() :925 and coulombic_real() :1484 then
// re-evaluate only flagged pairs and re-sum the cached values).  The device-side counterpart keeps the accepted
// configuration resident and, for a trial move of m atoms, evaluates only the pairs that involve a moved atom -- each one
// in its old and in its new geometry, with exactly the per-pair arithmetic of k_pair_fused -- plus the change of the
// structure factors:   E_trial = E_accepted + sum_{pairs with a moved atom} (u_new - u_old) + (E_recip[S + dS] - E_recip[S]).
// Cost O(m N) + O(K m) instead of O(N^2) + O(K N).
#include "kernels.h"
#include "device_math.h"

namespace mpmc {

// u(i,j) of lj() and coulombic_real() (erfc part) for one geometry; adds to e_lj/e_re and the in-cutoff counts with sign sg.
// EXT: the adjacent physics of the pair sweep's extended variant -- coulombic_wolf (:1420-1462) instead of the erfc term, Feynman-Hibbs
// corrections of both terms (:1100-1148, :1521-1557) -- with the same per-pair arithmetic (pair_math.h); imu = 1/M_i + 1/M_j.
template <bool ORTHO, bool EXT>
__device__ __forceinline__ void pair_terms(const Box &bx, const FusedParams &fp, int do_es, const double4 &pi, const double4 &pj, double sig,
                                           double eps, const PairFlags &f, double imu, double sg, double &e_lj, double &e_re, int &n_lj, int &n_es) {
	double ox, oy, oz;
	const double ri2 = min_image_sq<ORTHO>(bx, pi.x - pj.x, pi.y - pj.y, pi.z - pj.z, ox, oy, oz);
	const double ir = fast_rsqrt(ri2);
	if ((ri2 <= bx.t_lj) && !f.rd_excluded) {
		const double sr = sig * ir;
		double s6 = sr * sr * sr;
		s6 *= s6;
		const double t12 = f.attractive_only ? 0.0 : s6 * s6;
		e_lj = fma(sg * 4.0 * eps, t12 - s6, e_lj);
		n_lj += (sg > 0) ? 1 : -1;
		if (EXT && fp.fh_order) e_lj = fma(sg, fh_lj_corr(fp.fh_order, fp.fh_c2, fp.fh_c4, imu, eps, t12, s6, ir), e_lj);
	}
	if (!do_es || f.es_excluded) return;
	if (EXT && fp.wolf) {
		if (ri2 <= bx.t_wolf) { // r < R
			e_re = fma(sg * (pi.w * pj.w), ir - fp.wolf_erfa_over_r - fp.wolf_inv_r2 * (bx.cutoff - ri2 * ir), e_re);
			n_es += (sg > 0) ? 1 : -1;
		}
		return;
	}
	if (ri2 <= bx.t_es) {
		double g;
		const double r = ri2 * ir;
		const double ec = erfc_and_gauss(fp.ewald_alpha * r, g);
		e_re = fma(sg * (pi.w * pj.w) * ec, ir, e_re);
		n_es += (sg > 0) ? 1 : -1;
		if (EXT && fp.fh_order) e_re = fma(sg, fh_es_corr(fp.fh_order, fp.fh_c2, fp.fh_c4, imu, fp.ewald_alpha, ec, g, ri2, r, ir), e_re);
	}
}

// Where the kernels read the move from: device arrays (any m; staged by one host-to-device copy) or the kernel arguments themselves
// (MvInline, m <= kMvInline: no copy command at all -- a trial move is launch-bound on the host, and every call counts).
struct MvDev {
	const int *slot_, *orig_;
	const double4 *nw_;
	__device__ __forceinline__ int slot(int k) const { return slot_[k]; }
	__device__ __forceinline__ int orig(int k) const { return orig_[k]; }
	__device__ __forceinline__ double4 nw(int k) const { return nw_[k]; }
};
struct MvArg { // every element is read with a STATIC index (scalar loads from the argument segment) and selected: k may differ per lane
	MvInline d;
	__device__ __forceinline__ int slot(int k) const {
		int r = d.slot[0];
#pragma unroll
		for (int q = 1; q < kMvInline; ++q) r = (k == q) ? d.slot[q] : r;
		return r;
	}
	__device__ __forceinline__ int orig(int k) const {
		int r = d.orig[0];
#pragma unroll
		for (int q = 1; q < kMvInline; ++q) r = (k == q) ? d.orig[q] : r;
		return r;
	}
	__device__ __forceinline__ double4 nw(int k) const {
		double4 r = make_double4(d.nw[0][0], d.nw[0][1], d.nw[0][2], d.nw[0][3]);
#pragma unroll
		for (int q = 1; q < kMvInline; ++q)
			if (k == q) r = make_double4(d.nw[q][0], d.nw[q][1], d.nw[q][2], d.nw[q][3]);
		return r;
	}
};

// thread = atom slot j of tile `tile`; loops over the m moved atoms.
// mv.slot(k): slot of moved atom k; mv.nw(k): its trial position (+ charge); moved_idx[slot]: k or -1.
template <bool ORTHO, bool EXT, class MV>
__device__ __forceinline__ void delta_pairs_tile(const AtomsDev &at, const Box &bx, const FusedParams &fp, int do_es, const MV &mv, int m,
                                                 const int *__restrict__ moved_idx, double *__restrict__ block_part, int *__restrict__ block_cnt,
                                                 int tile) {
	const int j = tile * kTile + threadIdx.x;
	const double4 pj_old = at.xyzq[j];
	const double2 lj = at.lj[j];
	const int2 mj = at.mf[j];
	int kj = -1; // >= 0 when j itself is a moved atom
	if (moved_idx) kj = moved_idx[j];
	else
		for (int k = 0; k < m; ++k)
			if (mv.slot(k) == j) kj = k;
	const double4 pj_new = (kj >= 0) ? mv.nw(kj) : pj_old;
	double e_lj = 0, e_re = 0;
	int n_lj = 0, n_es = 0;
	double imm_j = 0.0;
	if (EXT && fp.fh_order) imm_j = at.inv_molmass[j];
	if (!(mj.y & AF_PAD)) {
		for (int k = 0; k < m; ++k) {
			if (kj >= 0 && kj <= k) continue; // moved-moved pairs once (k < kj), never the atom with itself
			const int si = mv.slot(k);
			const int2 mi = at.mf[si];
			const PairFlags f = pair_flags(mi.x, mi.y, mj.x, mj.y);
			if (f.frozen) continue;
			const double2 li = at.lj[si];
			double sig, eps;
			lj_mix(mi.y, mj.y, li.x, li.y, lj.x, lj.y, sig, eps);
			double imu = 0.0;
			if (EXT && fp.fh_order) imu = at.inv_molmass[si] + imm_j;
			pair_terms<ORTHO, EXT>(bx, fp, do_es, mv.nw(k), pj_new, sig, eps, f, imu, 1.0, e_lj, e_re, n_lj, n_es);
			pair_terms<ORTHO, EXT>(bx, fp, do_es, at.xyzq[si], pj_old, sig, eps, f, imu, -1.0, e_lj, e_re, n_lj, n_es);
		}
	}
	e_lj = wave_sum(e_lj);
	e_re = wave_sum(e_re);
	n_lj = wave_sum_i(n_lj);
	n_es = wave_sum_i(n_es);
	if (threadIdx.x == 0) {
		block_part[2 * (size_t)tile] = e_lj;
		block_part[2 * (size_t)tile + 1] = e_re;
		block_cnt[2 * (size_t)tile] = n_lj;
		block_cnt[2 * (size_t)tile + 1] = n_es;
	}
}

// change of the intramolecular charge-to-screen sum (coulombic_real :1503-1504) for the moved atoms: one block.
// mv.orig(k): original index of moved atom k; molecules are contiguous runs of the original order.
template <class MV>
__device__ __forceinline__ void delta_intra_block(const AtomsDev &at, const int *__restrict__ slot_of, double alpha, const MV &mv, int m,
                                                  const int *__restrict__ moved_idx, double *__restrict__ out) {
	double acc = 0;
	for (int k = threadIdx.x; k < m; k += 64) {
		const int i = mv.orig(k);
		const int si = slot_of[i];
		const int2 mi = at.mf[si];
		const double4 pi_old = at.xyzq[si], pi_new = mv.nw(k);
		// walk the molecule of atom i in both directions of the original order
		for (int dir = -1; dir <= 1; dir += 2)
			for (int jo = i + dir; jo >= 0 && jo < at.n; jo += dir) {
				const int sj = slot_of[jo];
				const int2 mj = at.mf[sj];
				if (mj.x != mi.x) break;
				int kj = -1;
				if (moved_idx) kj = moved_idx[sj];
				else
					for (int q = 0; q < m; ++q)
						if (mv.orig(q) == jo) kj = q;
				if (kj >= 0 && kj < k) continue; // a moved-moved pair is counted once, from its lower list index
				if (mi.y & mj.y & AF_FROZEN) continue;
				const double4 pj_old = at.xyzq[sj];
				const double4 pj_new = (kj >= 0) ? mv.nw(kj) : pj_old;
				const double qq = pi_old.w * pj_old.w;
				if (qq == 0.0) continue;
				for (int pass = 0; pass < 2; ++pass) {
					const double4 a = pass ? pi_old : pi_new, b = pass ? pj_old : pj_new;
					const double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
					const double r2 = ((dx * dx) + dy * dy) + dz * dz;
					const double ir = fast_rsqrt(r2);
					double g;
					const double t = qq * (1.0 - erfc_and_gauss(alpha * (r2 * ir), g)) * ir;
					acc += pass ? -t : t;
				}
			}
	}
	acc = wave_sum(acc);
	if (threadIdx.x == 0) out[0] = acc;
}

// trial structure factors: sf_trial[k] = sf[k] + sum_moved q (e^{i k.r_new} - e^{i k.r_old}); one wave per k-vector
template <class MV>
__device__ __forceinline__ void delta_recip_k(const AtomsDev &at, const RecipDev &rc, const MV &mv, int m, double4 *__restrict__ sf_trial, int kidx) {
	const double4 kv = rc.kvec[kidx];
	double re = 0, im = 0, C = 0, S = 0;
	for (int k = threadIdx.x; k < m; k += 64) {
		const int si = mv.slot(k);
		const int fl = at.mf[si].y;
		const double4 po = at.xyzq[si], pn = mv.nw(k);
		double s0, c0, s1, c1;
		sincos(((kv.x * po.x) + kv.y * po.y) + kv.z * po.z, &s0, &c0);
		sincos(((kv.x * pn.x) + kv.y * pn.y) + kv.z * pn.z, &s1, &c1);
		const double dc = po.w * (c1 - c0), ds = po.w * (s1 - s0);
		C += dc;
		S += ds;
		if (!(fl & (AF_FROZEN | AF_ZERO_Q))) {
			re += dc;
			im += ds;
		}
	}
	re = wave_sum(re);
	im = wave_sum(im);
	C = wave_sum(C);
	S = wave_sum(S);
	if (threadIdx.x == 0) {
		const double4 o = rc.sf[kidx];
		sf_trial[kidx] = make_double4(o.x + re, o.y + im, o.z + C, o.w + S);
	}
}

// ONE launch for the three parts of a trial that do not depend on one another (they were three): blocks [0, nt) the pair terms of one
// tile each, block nt the intramolecular term, blocks (nt, nt + 1 + K) one k-vector each.  Without Ewald electrostatics (none at all, or
// Wolf: no intramolecular and no reciprocal term, coulombic() :1404-1413) the grid is nt.
template <bool ORTHO, bool EXT, class MV>
__global__ __launch_bounds__(64) void k_delta_all(AtomsDev at, Box bx, RecipDev rc, const int *__restrict__ slot_of, FusedParams fp, int do_es, MV mv, int m,
                                                  const int *__restrict__ moved_idx, double4 *__restrict__ sf_trial, double *__restrict__ block_part,
                                                  int *__restrict__ block_cnt, double *__restrict__ out_intra) {
	const int nt = at.n_pad / kTile;
	const int b = blockIdx.x; // (block-uniform roles)
	if (b < nt) delta_pairs_tile<ORTHO, EXT>(at, bx, fp, do_es, mv, m, moved_idx, block_part, block_cnt, b);
	else if (b == nt) delta_intra_block(at, slot_of, fp.ewald_alpha, mv, m, moved_idx, out_intra);
	else delta_recip_k(at, rc, mv, m, sf_trial, b - nt - 1);
}

// sums the per-tile partials, the reciprocal energy of the trial structure factors, into out[0..3] and counts
__global__ __launch_bounds__(256) void k_delta_finish(const double *__restrict__ block_part, const int *__restrict__ block_cnt, int nb,
                                                      RecipDev rc, const double4 *__restrict__ sf_trial, Box bx, int do_es,
                                                      double *__restrict__ out /* dlj, des_real, (dintra at [2]), e_recip_trial at [3] */,
                                                      long long *__restrict__ dcnt,
                                                      double *__restrict__ host_out /* pinned [9]: the same 5 doubles + 2 counts, spare, launch number */,
                                                      double seq) {
	__shared__ double sh[4];
	__shared__ long long shc[256];
	double s0 = 0, s1 = 0, e = 0;
	long long c0 = 0, c1 = 0;
	for (int b = threadIdx.x; b < nb; b += 256) {
		s0 += block_part[2 * (size_t)b];
		s1 += block_part[2 * (size_t)b + 1];
		c0 += block_cnt[2 * (size_t)b];
		c1 += block_cnt[2 * (size_t)b + 1];
	}
	if (do_es)
		for (int k = threadIdx.x; k < rc.K; k += 256) {
			const double4 sf = sf_trial[k];
			e += rc.w_en[k] * (sf.x * sf.x + sf.y * sf.y);
		}
	auto bsum = [&](double v) {
		v = wave_sum(v);
		__syncthreads();
		if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
		__syncthreads();
		return ((sh[0] + sh[1]) + sh[2]) + sh[3];
	};
	s0 = bsum(s0);
	s1 = bsum(s1);
	e = bsum(e);
	if (threadIdx.x == 0) {
		out[0] = s0;
		out[1] = s1;
		if (!do_es) out[2] = 0.0; // (k_delta_intra wrote it otherwise)
		out[3] = e * (4.0 * kPi / bx.volume);
		out[4] = (double)0;
	}
	long long tot[2] = {0, 0};
	for (int k = 0; k < 2; ++k) {
		__syncthreads();
		shc[threadIdx.x] = k ? c1 : c0;
		__syncthreads();
		for (int off = 128; off > 0; off >>= 1) {
			if (threadIdx.x < off) shc[threadIdx.x] += shc[threadIdx.x + off];
			__syncthreads();
		}
		if (threadIdx.x == 0) dcnt[k] = tot[k] = shc[0];
	}
	// the result goes to the caller's pinned block from here (no copy-back command, no stream synchronisation: the host polls the launch
	// number, which is stored last, behind a system-scope fence)
	if (host_out && threadIdx.x == 0) {
		host_out[0] = s0;
		host_out[1] = s1;
		host_out[2] = do_es ? out[2] : 0.0;
		host_out[3] = e * (4.0 * kPi / bx.volume);
		host_out[4] = 0.0;
		long long *hc = reinterpret_cast<long long *>(host_out + 5);
		hc[0] = tot[0];
		hc[1] = tot[1];
		__threadfence_system();
		__hip_atomic_store(host_out + 8, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// set / clear the moved-atom index map
__global__ void k_mark_moved(int *__restrict__ moved_idx, const int *__restrict__ mv_slot, int m, int set) {
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k < m) moved_idx[mv_slot[k]] = set ? k : -1;
}

// ---- polarizable boxes: change of the REAL-SPACE static field ---------------------------------------------------------------------
// thole_field's real part (real_term :2900-2940 with polar_ewald, thole_field_nopbc :3300-3333 without) is a pair sum: a trial move of
// m atoms changes, for every atom j, only the terms with a moved partner.  Thread = atom j, loop over the moved atoms k, each pair in
// its new and in its old geometry with the arithmetic of k_pair_fused:
//   E_j += q_k [F(r_j - r_k')  - F(r_j - r_k)],      E_k += q_j [F(r_k' - r_j) - F(r_k - r_j)]   (F odd: the pair is evaluated once)
// FIELD 1: F(d) = fac(|d|) d with fac = (2 a/sqrt(pi) e^{-a^2 r^2} r +- erfc/erf(a r)) / r^3 inside the cutoff (es_excluded pairs take
// the erf form); FIELD 2: fac = 1/r^3 for inter-molecular pairs inside the cutoff.  The moved atoms' own changes are reduced per wave and
// land in dk_part[tile][k][3]; k_delta_field_finish adds them up over the tiles.
template <bool ORTHO, int FIELD>
__device__ __forceinline__ void field_pair(const Box &bx, double ap, const double4 &pi, const double4 &pj, const PairFlags &f, double sg,
                                           double (&ei)[3], double (&ej)[3]) {
	double ox, oy, oz;
	const double ri2 = min_image_sq<ORTHO>(bx, pi.x - pj.x, pi.y - pj.y, pi.z - pj.z, ox, oy, oz);
	if (ri2 == 0.0) return;
	double fac;
	if (FIELD == 1) {
		if (!(ri2 <= bx.t_es) || (pi.w == 0.0 && pj.w == 0.0)) return; // real_term :2916-2917
		const double ir = fast_rsqrt_1(ri2), r = ri2 * ir;
		double ga;
		const double ec = erfc_and_gauss(ap * r, ga);
		const double g = (2.0 * kOneOverSqrtPi * ap) * (ga * r);
		fac = (f.es_excluded ? (g - (1.0 - ec)) : (g + ec)) * (ir * ir * ir);
	} else {
		if (f.intra || !(ri2 <= bx.t_lj)) return; // thole_field_nopbc :3311-3326
		const double ir = fast_rsqrt_1(ri2);
		fac = ir * ir * ir;
	}
	const double fj = sg * fac * pj.w, fi = sg * fac * pi.w;
	ei[0] = fma(fj, ox, ei[0]);
	ei[1] = fma(fj, oy, ei[1]);
	ei[2] = fma(fj, oz, ei[2]);
	ej[0] = fma(-fi, ox, ej[0]);
	ej[1] = fma(-fi, oy, ej[1]);
	ej[2] = fma(-fi, oz, ej[2]);
}

template <bool ORTHO, int FIELD>
__global__ __launch_bounds__(64) void k_delta_field(AtomsDev at, Box bx, double ap, const int *__restrict__ mv_slot, const double4 *__restrict__ mv_new,
                                                    int m, const int *__restrict__ moved_idx, const double *__restrict__ e_real,
                                                    double *__restrict__ e_real_trial, double *__restrict__ dk_part /*[n_tiles][m][3]*/) {
	const int j = blockIdx.x * kTile + threadIdx.x;
	const double4 pj_old = at.xyzq[j];
	const int2 mj = at.mf[j];
	int kj = -1;
	if (moved_idx) kj = moved_idx[j];
	else
		for (int k = 0; k < m; ++k)
			if (mv_slot[k] == j) kj = k;
	const double4 pj_new = (kj >= 0) ? mv_new[kj] : pj_old;
	const bool j_real = !(mj.y & AF_PAD);
	double ej[3] = {0, 0, 0};
	for (int k = 0; k < m; ++k) {
		double ek[3] = {0, 0, 0};
		if (j_real && !(kj >= 0 && kj <= k)) { // moved-moved pairs once (from the higher list index), never an atom with itself
			const int si = mv_slot[k];
			const int2 mi = at.mf[si];
			const PairFlags f = pair_flags(mi.x, mi.y, mj.x, mj.y);
			if (!f.frozen) {
				field_pair<ORTHO, FIELD>(bx, ap, mv_new[k], pj_new, f, 1.0, ek, ej);
				field_pair<ORTHO, FIELD>(bx, ap, at.xyzq[si], pj_old, f, -1.0, ek, ej);
			}
		}
		for (int d = 0; d < 3; ++d) ek[d] = wave_sum(ek[d]);
		if (threadIdx.x == 0) {
			double *o = dk_part + ((size_t)blockIdx.x * m + k) * 3;
			o[0] = ek[0];
			o[1] = ek[1];
			o[2] = ek[2];
		}
	}
	for (int d = 0; d < 3; ++d) e_real_trial[3 * (size_t)j + d] = e_real[3 * (size_t)j + d] + ej[d];
}
// the moved atoms' own share: e_real_trial[slot_k] += sum over tiles of dk_part[tile][k]  (one thread per moved atom, tiles in order)
__global__ __launch_bounds__(64) void k_delta_field_finish(const int *__restrict__ mv_slot, int m, int n_tiles, const double *__restrict__ dk_part,
                                                           double *__restrict__ e_real_trial) {
	const int k = blockIdx.x * 64 + threadIdx.x;
	if (k >= m) return;
	double s[3] = {0, 0, 0};
	for (int t = 0; t < n_tiles; ++t) {
		const double *q = dk_part + ((size_t)t * m + k) * 3;
		s[0] += q[0];
		s[1] += q[1];
		s[2] += q[2];
	}
	double *o = e_real_trial + 3 * (size_t)mv_slot[k];
	o[0] += s[0];
	o[1] += s[1];
	o[2] += s[2];
}

void launch_delta_field(hipStream_t st, const AtomsDev &at, const Box &bx, double polar_ewald_alpha, int polar_ewald, const int *mv_slot,
                        const double4 *mv_new, int m, int *moved_idx, const double *e_real, double *e_real_trial, double *dk_part) {
	const int nt = at.n_pad / kTile;
	const bool use_map = (m > 8);
	if (use_map) hipLaunchKernelGGL(k_mark_moved, dim3((m + 63) / 64), dim3(64), 0, st, moved_idx, mv_slot, m, 1);
	else moved_idx = nullptr;
#define MPMC_DF(O, F) hipLaunchKernelGGL((k_delta_field<O, F>), dim3(nt), dim3(kTile), 0, st, at, bx, polar_ewald_alpha, mv_slot, mv_new, m, moved_idx, e_real, e_real_trial, dk_part)
	if (bx.ortho) {
		if (polar_ewald) MPMC_DF(true, 1);
		else MPMC_DF(true, 2);
	} else {
		if (polar_ewald) MPMC_DF(false, 1);
		else MPMC_DF(false, 2);
	}
#undef MPMC_DF
	hipLaunchKernelGGL(k_delta_field_finish, dim3((m + 63) / 64), dim3(64), 0, st, mv_slot, m, nt, dk_part, e_real_trial);
	if (use_map) hipLaunchKernelGGL(k_mark_moved, dim3((m + 63) / 64), dim3(64), 0, st, moved_idx, mv_slot, m, 0);
}

// swap: the resident positions of the moved atoms become the trial ones, the old ones are kept in mv_new's place (a second call undoes it)
__global__ void k_swap_positions(double4 *__restrict__ xyzq, const int *__restrict__ mv_slot, double4 *__restrict__ mv_new, int m) {
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k < m) {
		const double4 old = xyzq[mv_slot[k]];
		xyzq[mv_slot[k]] = mv_new[k];
		mv_new[k] = old;
	}
}
void launch_swap_positions(hipStream_t st, double4 *xyzq, const int *mv_slot, double4 *mv_new, int m) {
	hipLaunchKernelGGL(k_swap_positions, dim3((m + 63) / 64), dim3(64), 0, st, xyzq, mv_slot, mv_new, m);
}

// accept: write the trial positions into the resident arrays
__global__ void k_commit_positions(double4 *__restrict__ xyzq, const int *__restrict__ mv_slot, const double4 *__restrict__ mv_new, int m) {
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k < m) xyzq[mv_slot[k]] = mv_new[k];
}
void launch_delta(hipStream_t st, const AtomsDev &at, const int *slot_of, const Box &bx, const RecipDev &rc, const FusedParams &fp, int do_es,
                  const int *mv_slot, const int *orig_of_mv, const double4 *mv_new, int m, int *moved_idx, double4 *sf_trial,
                  double *block_part, int *block_cnt, double *out4, long long *dcnt2, double *host_out, double seq, const MvInline *inl) {
	const int nt = at.n_pad / kTile;
	const bool ext = (do_es && fp.wolf) || fp.fh_order;
	const int do_ewald = (do_es && !fp.wolf) ? 1 : 0; // intramolecular + reciprocal parts exist
	const int grid = do_ewald ? nt + 1 + rc.K : nt;
#define MPMC_DELTA(O, E, MVT, MVV, MAP)                                                                                                              \
	hipLaunchKernelGGL((k_delta_all<O, E, MVT>), dim3(grid), dim3(kTile), 0, st, at, bx, rc, slot_of, fp, do_es, MVV, m, MAP, sf_trial, block_part, block_cnt, \
	                   out4 + 2)
#define MPMC_DELTA_OE(MVT, MVV, MAP)              \
	if (bx.ortho) {                               \
		if (ext) MPMC_DELTA(true, true, MVT, MVV, MAP);   \
		else MPMC_DELTA(true, false, MVT, MVV, MAP);      \
	} else {                                      \
		if (ext) MPMC_DELTA(false, true, MVT, MVV, MAP);  \
		else MPMC_DELTA(false, false, MVT, MVV, MAP);     \
	}
	if (inl) { // the move travels in the kernel arguments (m <= kMvInline: the lists are scanned in the kernels, no map)
		MvArg mv;
		mv.d = *inl;
		MPMC_DELTA_OE(MvArg, mv, nullptr)
		hipLaunchKernelGGL(k_delta_finish, dim3(1), dim3(256), 0, st, block_part, block_cnt, nt, rc, sf_trial, bx, do_ewald, out4, dcnt2, host_out, seq);
		return;
	}
	const bool use_map = (m > 8); // short lists are scanned in the kernels; long ones go through the slot -> list-index map
	if (use_map) hipLaunchKernelGGL(k_mark_moved, dim3((m + 63) / 64), dim3(64), 0, st, moved_idx, mv_slot, m, 1);
	else moved_idx = nullptr;
	const MvDev mv{mv_slot, orig_of_mv, mv_new};
	MPMC_DELTA_OE(MvDev, mv, moved_idx)
#undef MPMC_DELTA_OE
#undef MPMC_DELTA
	hipLaunchKernelGGL(k_delta_finish, dim3(1), dim3(256), 0, st, block_part, block_cnt, nt, rc, sf_trial, bx, do_ewald, out4, dcnt2, host_out, seq);
	if (use_map) hipLaunchKernelGGL(k_mark_moved, dim3((m + 63) / 64), dim3(64), 0, st, moved_idx, mv_slot, m, 0);
}
// accept, the move in the kernel arguments
__global__ void k_commit_positions_arg(double4 *__restrict__ xyzq, MvArg mv, int m) {
	const int k = threadIdx.x;
	if (k < m) xyzq[mv.slot(k)] = mv.nw(k);
}
void launch_commit_positions_inline(hipStream_t st, double4 *xyzq, const MvInline &inl, int m) {
	MvArg mv;
	mv.d = inl;
	hipLaunchKernelGGL(k_commit_positions_arg, dim3(1), dim3(64), 0, st, xyzq, mv, m);
}
void launch_commit_positions(hipStream_t st, double4 *xyzq, const int *mv_slot, const double4 *mv_new, int m) {
	hipLaunchKernelGGL(k_commit_positions, dim3((m + 63) / 64), dim3(64), 0, st, xyzq, mv_slot, mv_new, m);
}

} // namespace mpmc
