// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the energy hot path.
//
// Common shape: one wavefront (64 lanes) owns 64 i-atoms in registers; j-atoms are staged 64 at a time in LDS
// and broadcast-read by all lanes (same address per wave instruction => no bank conflicts).  All arithmetic
// is fp64.  Every sum is reduced in a fixed order (per-lane serial -> wave shuffle tree -> per-block partial ->
// single-block final pass), so results are bit-reproducible run to run.  No atomics on floating point.
//
// Compiled with -ffp-contract=off (see pair_math.h).
#include "kernels.h"

namespace mpmc {

// ------------------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	return v; // valid in lane 0
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	return v;
}
// sum over a 256-thread block; result valid in thread 0.  `sh` must hold 4 doubles.
__device__ __forceinline__ double block_sum_256(double v, double *sh) {
	v = wave_sum(v);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
	__syncthreads();
	return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// ------------------------------------------------------------------------------------------------------
// fixed-order final reduction of the per-tile-pair partials of k_pair_fused: {lj, es_real} and {n_lj, n_es}
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void reduce_pairs_block(const double *__restrict__ block_part, const int *__restrict__ block_cnt, int nb,
                                                   double *__restrict__ scal, long long *__restrict__ cnt) {
	__shared__ double sh[4];
	__shared__ long long shc[256];
	double s0 = 0, s1 = 0;
	long long c0 = 0, c1 = 0;
#pragma unroll 4
	for (int b = threadIdx.x; b < nb; b += 256) { // (one block over all tile pairs' partials: unrolled, the loads overlap; same order of sums)
		s0 += block_part[2 * (size_t)b];
		s1 += block_part[2 * (size_t)b + 1];
		c0 += block_cnt[2 * (size_t)b];
		c1 += block_cnt[2 * (size_t)b + 1];
	}
	s0 = block_sum_256(s0, sh);
	s1 = block_sum_256(s1, sh);
	if (threadIdx.x == 0) {
		scal[S_LJ] = s0;
		scal[S_ES_REAL] = s1;
	}
	for (int k = 0; k < 2; ++k) {
		__syncthreads();
		shc[threadIdx.x] = k ? c1 : c0;
		__syncthreads();
		for (int off = 128; off > 0; off >>= 1) {
			if (threadIdx.x < off) shc[threadIdx.x] += shc[threadIdx.x + off];
			__syncthreads();
		}
		if (threadIdx.x == 0) cnt[k ? C_ES_IN : C_LJ_IN] = shc[0];
	}
}
__global__ __launch_bounds__(256) void k_reduce_pairs(const double *__restrict__ block_part, const int *__restrict__ block_cnt, int nb,
                                                      double *__restrict__ scal, long long *__restrict__ cnt) {
	reduce_pairs_block(block_part, block_cnt, nb, scal, cnt);
}

// position-independent pair flags (reference pair_exclusions :1035-1067): one wave per tile pair, broadcast j loop
__global__ __launch_bounds__(64) void k_static_counts(AtomsDev at, const int2 *__restrict__ tile_pairs, int *__restrict__ block_cnt) {
	__shared__ int2 s_mf[kTile];
	const int lane = threadIdx.x;
	const int2 IJ = tile_pairs[blockIdx.x];
	const int2 mi = at.mf[IJ.x * kTile + lane];
	s_mf[lane] = at.mf[IJ.y * kTile + lane];
	__syncthreads();
	int n_intra = 0, n_rdx = 0, n_esx = 0, n_fr = 0;
	const bool diag = (IJ.x == IJ.y);
	for (int jj = 0; jj < kTile; ++jj) {
		const int2 mj = s_mf[jj];
		if ((mi.y & AF_PAD) || (mj.y & AF_PAD) || (diag && jj <= lane)) continue;
		const PairFlags f = pair_flags(mi.x, mi.y, mj.x, mj.y);
		n_intra += f.intra;
		n_rdx += f.rd_excluded;
		n_esx += f.es_excluded;
		n_fr += f.frozen;
	}
	n_intra = wave_sum_i(n_intra);
	n_rdx = wave_sum_i(n_rdx);
	n_esx = wave_sum_i(n_esx);
	n_fr = wave_sum_i(n_fr);
	if (lane == 0) {
		int *bc = block_cnt + 4 * (size_t)blockIdx.x;
		bc[0] = n_intra;
		bc[1] = n_rdx;
		bc[2] = n_esx;
		bc[3] = n_fr;
	}
}
__global__ __launch_bounds__(256) void k_reduce_counts4(const int *__restrict__ block_cnt, int nb, long long *__restrict__ cnt4) {
	__shared__ long long shc[256];
	long long c[4] = {0, 0, 0, 0};
#pragma unroll 4
	for (int b = threadIdx.x; b < nb; b += 256) // (one block over all tile pairs: unrolled, the loads overlap)
		for (int k = 0; k < 4; ++k) c[k] += block_cnt[4 * (size_t)b + k];
	for (int k = 0; k < 4; ++k) {
		__syncthreads();
		shc[threadIdx.x] = c[k];
		__syncthreads();
		for (int off = 128; off > 0; off >>= 1) {
			if (threadIdx.x < off) shc[threadIdx.x] += shc[threadIdx.x + off];
			__syncthreads();
		}
		if (threadIdx.x == 0) cnt4[k] = shc[0];
	}
}
void launch_static_counts(hipStream_t st, const AtomsDev &at, const int2 *tile_pairs, int n_tile_pairs, int *block_cnt, long long *cnt4) {
	hipLaunchKernelGGL(k_static_counts, dim3(n_tile_pairs), dim3(kTile), 0, st, at, tile_pairs, block_cnt);
	hipLaunchKernelGGL(k_reduce_counts4, dim3(1), dim3(256), 0, st, block_cnt, n_tile_pairs, cnt4);
}

void launch_reduce_pairs(hipStream_t st, const double *block_part, const int *block_cnt, int nb, double *scal, long long *cnt) {
	hipLaunchKernelGGL(k_reduce_pairs, dim3(1), dim3(256), 0, st, block_part, block_cnt, nb, scal, cnt);
}

// ------------------------------------------------------------------------------------------------------
// reciprocal space (reference coulombic_reciprocal :1561-1622, recip_term :2834-2896, coulombic_self :1626-1643,
// lj_lrc_self :1072-1096)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_recip_sf(AtomsDev at, RecipDev rc) {
	__shared__ double sh[4];
	const double4 kv = rc.kvec[blockIdx.x];
	double re = 0, im = 0, C = 0, S = 0;
	for (int a = threadIdx.x; a < at.n; a += 256) {
		const double4 p = at.xyzq[a];
		const int fl = at.mf[a].y;
		const double ph = ((kv.x * p.x) + kv.y * p.y) + kv.z * p.z;
		double s, c;
		sincos(ph, &s, &c);
		const double qc = p.w * c, qs = p.w * s;
		C += qc; // recip_term sums over ALL atoms (:2868-2872)
		S += qs;
		if (!(fl & (AF_FROZEN | AF_ZERO_Q))) { // coulombic_reciprocal skips frozen and q == 0 (:1599-1602)
			re += qc;
			im += qs;
		}
	}
	re = block_sum_256(re, sh);
	im = block_sum_256(im, sh);
	C = block_sum_256(C, sh);
	S = block_sum_256(S, sh);
	if (threadIdx.x == 0) rc.sf[blockIdx.x] = make_double4(re, im, C, S);
}

// binomial coefficients C(6,k), C(12,k)
__device__ const double kBinom6[7] = {1, 6, 15, 20, 15, 6, 1};
__device__ const double kBinom12[13] = {1, 12, 66, 220, 495, 792, 924, 792, 495, 220, 66, 12, 1};

// Reciprocal energy, coulombic_self, lj_lrc_self and the pair LRC.
// Pair LRC (reference lj_lrc_corr :1036-1069, summed over ALL non-frozen pairs with eps_ij, sigma_ij != 0):
//   g_ij = (16 pi / 3V) eps_ij (sig_ij^12 / (3 rc^9) - sig_ij^6 / rc^3),  eps_ij = sqrt(eps_i) sqrt(eps_j),  sig_ij = (s_i + s_j)/2
//   sum_{i<j} sqe_i sqe_j (s_i+s_j)^m = 1/2 [ sum_k C(m,k) M_k M_{m-k} - sum_i sqe_i^2 (2 s_i)^m ],   M_k = sum_i sqe_i s_i^k
// over the atoms with eps != 0, sigma > 0 (sigma == 0 or < 0 gives sig_ij or eps_ij = 0), minus the same sum over the
// frozen subset (frozen-frozen pairs are excluded, :1049).  O(N) instead of O(N^2), position independent.
// Two launches: the per-atom sums over as many 256-thread blocks as there are atoms for (one block looping over 10 000 atoms with thirteen
// dependent multiplies each took 81 us -- most of an evaluation it rides along with after every insertion, removal or volume move),
// 32 partial sums per block; then one wave adds the blocks' partials in block order (reproducible) and forms the three results.
constexpr int kAtomTermSums = 32, kAtomTermBlocks = 64; // self, lrc, M_0..12, Mfrozen_0..12, d6, d12, df6, df12
static_assert((size_t)kAtomTermSums * kAtomTermBlocks <= kAtomTermScratch, "scratch of launch_atom_terms");
__global__ __launch_bounds__(256) void k_atom_terms_part(AtomsDev at, Box bx, double ewald_alpha, int rd_lrc, int do_es,
                                                         double *__restrict__ part /*[gridDim.x][kAtomTermSums]*/) {
	double self = 0, lrc = 0;
	double m[13], mf[13], d6 = 0, d12 = 0, df6 = 0, df12 = 0;
	for (int k = 0; k < 13; ++k) m[k] = mf[k] = 0;
	for (int a = blockIdx.x * 256 + threadIdx.x; a < at.n; a += gridDim.x * 256) {
		const int fl = at.mf[a].y;
		const double2 l = at.lj[a];
		if (rd_lrc && !(fl & (AF_NULL_RD | AF_NEG_SIGMA))) {
			double pw = l.y;
			const bool fr = (fl & AF_FROZEN) != 0;
			for (int k = 0; k < 13; ++k) {
				m[k] += pw;
				if (fr) mf[k] += pw;
				pw *= l.x;
			}
			const double t2 = 2.0 * l.x, t6 = (t2 * t2 * t2) * (t2 * t2 * t2), e2 = l.y * l.y;
			d6 += e2 * t6;
			d12 += e2 * t6 * t6;
			if (fr) {
				df6 += e2 * t6;
				df12 += e2 * t6 * t6;
			}
		}
		if (fl & AF_FROZEN) continue;
		const double q = at.xyzq[a].w;
		if (do_es) self -= ewald_alpha * q * q / sqrt(kPi);
		if (rd_lrc && !(fl & AF_NULL_RD)) lrc += lrc_term(l.x, at.eps[a], bx.cutoff, bx.volume);
	}
	// the 32 sums of the block: every wave reduces all of them, ONE barrier, then 32 threads fold the four waves in wave order (the order
	// block_sum_256 uses -- 32 calls of it were 64 barriers and most of the kernel's 15 us)
	__shared__ double s_w[4][kAtomTermSums];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	auto put = [&](int slot, double v) {
		v = wave_sum(v);
		if (lane == 0) s_w[w][slot] = v;
	};
	put(0, self);
	put(1, lrc);
	for (int k = 0; k < 13; ++k) {
		put(2 + k, m[k]);
		put(15 + k, mf[k]);
	}
	put(28, d6);
	put(29, d12);
	put(30, df6);
	put(31, df12);
	__syncthreads();
	if (threadIdx.x < kAtomTermSums)
		part[(size_t)blockIdx.x * kAtomTermSums + threadIdx.x] = ((s_w[0][threadIdx.x] + s_w[1][threadIdx.x]) + s_w[2][threadIdx.x]) + s_w[3][threadIdx.x];
}
__global__ __launch_bounds__(64) void k_atom_terms_finish(const double *__restrict__ part, int nb, Box bx, int rd_lrc, double *__restrict__ scal) {
	__shared__ double s[kAtomTermSums];
	if (threadIdx.x < kAtomTermSums) {
		double v = 0;
#pragma unroll 8
		for (int b = 0; b < nb; ++b) v += part[(size_t)b * kAtomTermSums + threadIdx.x]; // block order: the same whatever ran first (unrolled: loads in flight together)
		s[threadIdx.x] = v;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		const double *m = s + 2, *mf = s + 15;
		const double d6 = s[28], d12 = s[29], df6 = s[30], df12 = s[31];
		double s6 = 0, s12 = 0, f6 = 0, f12 = 0;
		for (int k = 0; k <= 6; ++k) {
			s6 += kBinom6[k] * m[k] * m[6 - k];
			f6 += kBinom6[k] * mf[k] * mf[6 - k];
		}
		for (int k = 0; k <= 12; ++k) {
			s12 += kBinom12[k] * m[k] * m[12 - k];
			f12 += kBinom12[k] * mf[k] * mf[12 - k];
		}
		const double p6 = (0.5 * (s6 - d6) - 0.5 * (f6 - df6)) / 64.0;       // sum_{pairs} eps_ij sig_ij^6
		const double p12 = (0.5 * (s12 - d12) - 0.5 * (f12 - df12)) / 4096.0; // sum_{pairs} eps_ij sig_ij^12
		const double rc3 = bx.cutoff * bx.cutoff * bx.cutoff, rc9 = rc3 * rc3 * rc3;
		scal[S_LRC_PAIR] = rd_lrc ? (16.0 / 3.0) * kPi * (p12 / (3.0 * rc9) - p6 / rc3) / bx.volume : 0.0;
		scal[S_ES_SELF] = s[0];
		scal[S_LRC_SELF] = s[1];
	}
}

// ---- the same sums without one sincos per (k, atom) ------------------------------------------------------------------------
// k = 2 pi R l (l integer, :1586-1590), so exp(i k.r) = prod_q exp(2 pi i g_q)^{l_q} with g_q = sum_p R[p][q] r_p: three sincos per
// atom, the powers m = 0..kmax by complex multiplication, and every (k, atom) term is two complex products of table entries
// (the classic Ewald factorisation; relative deviation from sincos(k.r) ~ kmax x 1e-16).
__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 phase_base(const Box &bx, const double4 &p, int q) {
	const double g = (bx.r[q] * p.x + bx.r[3 + q] * p.y) + bx.r[6 + q] * p.z;
	double s, c;
	sincos(2.0 * kPi * g, &s, &c);
	return make_double2(c, s);
}

// structure factors: block (atom tile, chunk of 256 k-vectors), thread = k-vector, table [dir][atom][m] in LDS (a wave's reads of one
// atom fall into one 16 (kmax+1)-byte row: distinct banks, equal m broadcast).  part[tile][k] = this tile's share of (re, im, C, S).
__global__ __launch_bounds__(256) void k_recip_sf_tab(AtomsDev at, Box bx, RecipDev rc, int kmax, double4 *__restrict__ part) {
	extern __shared__ double2 tab[];
	__shared__ double s_q[kTile], s_qe[kTile];
	const int KM1 = kmax + 1;
	const int tile = blockIdx.x, tid = threadIdx.x;
	for (int idx = tid; idx < 3 * kTile; idx += 256) {
		const int a = idx & 63, q = idx >> 6;
		const double2 base = phase_base(bx, at.xyzq[tile * kTile + a], q);
		double2 *row = tab + (size_t)(q * kTile + a) * KM1;
		double2 v = make_double2(1.0, 0.0);
		row[0] = v;
		for (int m = 1; m <= kmax; ++m) {
			v = cmul(v, base);
			row[m] = v;
		}
	}
	if (tid < kTile) {
		const int i = tile * kTile + tid;
		const int fl = at.mf[i].y;
		const double q = (fl & AF_PAD) ? 0.0 : at.xyzq[i].w;
		s_q[tid] = q;                                                   // recip_term sums over ALL atoms (:2868-2872)
		s_qe[tid] = (fl & (AF_FROZEN | AF_ZERO_Q | AF_PAD)) ? 0.0 : q;  // coulombic_reciprocal skips frozen and q == 0 (:1599-1602)
	}
	__syncthreads();
	const int k = blockIdx.y * 256 + tid;
	if (k >= rc.K) return;
	const int4 l = rc.lvec[k];
	const int ay = abs(l.y), az = abs(l.z);
	const double sy = (l.y < 0) ? -1.0 : 1.0, sz = (l.z < 0) ? -1.0 : 1.0;
	double re = 0, im = 0, C = 0, S = 0;
	for (int a = 0; a < kTile; ++a) {
		const double2 ex = tab[(size_t)a * KM1 + l.x];
		double2 ey = tab[(size_t)(kTile + a) * KM1 + ay];
		double2 ez = tab[(size_t)(2 * kTile + a) * KM1 + az];
		ey.y *= sy;
		ez.y *= sz;
		const double2 e = cmul(cmul(ex, ey), ez);
		const double q = s_q[a], qe = s_qe[a];
		C += q * e.x;
		S += q * e.y;
		re += qe * e.x;
		im += qe * e.y;
	}
	part[(size_t)tile * rc.K + k] = make_double4(re, im, C, S);
}
// sf[k] = sum over tiles of part[tile][k]: 16 k-vectors x 16 tile groups per block (the loads of one thread are few and independent),
// folded in a fixed order => reproducible
__global__ __launch_bounds__(256) void k_recip_sf_reduce(RecipDev rc, const double4 *__restrict__ part, int n_tiles) {
	__shared__ double4 sh[16][16];
	const int kk = threadIdx.x & 15, g = threadIdx.x >> 4;
	const int k = blockIdx.x * 16 + kk;
	double4 acc = make_double4(0, 0, 0, 0);
	if (k < rc.K)
		for (int t = g; t < n_tiles; t += 16) {
			const double4 v = part[(size_t)t * rc.K + k];
			acc.x += v.x;
			acc.y += v.y;
			acc.z += v.z;
			acc.w += v.w;
		}
	sh[g][kk] = acc;
	__syncthreads();
	if (g == 0 && k < rc.K) {
		double4 r = sh[0][kk];
		for (int j = 1; j < 16; ++j) {
			r.x += sh[j][kk].x;
			r.y += sh[j][kk].y;
			r.z += sh[j][kk].z;
			r.w += sh[j][kk].w;
		}
		rc.sf[k] = r;
	}
}

void launch_recip_sf(hipStream_t st, const AtomsDev &at, const Box &bx, const RecipDev &rc, int kmax, double4 *sf_part) {
	if (rc.K <= 0) return;
	if (sf_part && rc.lvec && kmax <= kRecipTabMaxK) {
		const int nt = at.n_pad / kTile;
		const size_t lds = (size_t)3 * kTile * (kmax + 1) * sizeof(double2);
		hipLaunchKernelGGL(k_recip_sf_tab, dim3(nt, (rc.K + 255) / 256), dim3(256), lds, st, at, bx, rc, kmax, sf_part);
		hipLaunchKernelGGL(k_recip_sf_reduce, dim3((rc.K + 15) / 16), dim3(256), 0, st, rc, sf_part, nt);
		return;
	}
	hipLaunchKernelGGL(k_recip_sf, dim3(rc.K), dim3(256), 0, st, at, rc);
}
void launch_atom_terms(hipStream_t st, const AtomsDev &at, const Box &bx, double ewald_alpha, int rd_lrc, int do_es, double *part_scratch,
                       double *scal) {
	const int nb = std::max(1, std::min(kAtomTermBlocks, (at.n + 255) / 256));
	hipLaunchKernelGGL(k_atom_terms_part, dim3(nb), dim3(256), 0, st, at, bx, ewald_alpha, rd_lrc, do_es, part_scratch);
	hipLaunchKernelGGL(k_atom_terms_finish, dim3(1), dim3(64), 0, st, part_scratch, nb, bx, rd_lrc, scal);
}

// the position-dependent part of coulombic_reciprocal alone (:1609-1618): (4 pi / V) sum_k w_k |S_k|^2 -- K terms, one block
__global__ __launch_bounds__(256) void k_recip_energy(RecipDev rc, Box bx, double *__restrict__ scal) {
	__shared__ double sh[4];
	double e = 0;
	for (int k = threadIdx.x; k < rc.K; k += 256) {
		const double4 sf = rc.sf[k];
		e += rc.w_en[k] * (sf.x * sf.x + sf.y * sf.y);
	}
	e = block_sum_256(e, sh);
	if (threadIdx.x == 0) scal[S_ES_RECIP] = e * (4.0 * kPi / bx.volume);
}
void launch_recip_energy(hipStream_t st, const RecipDev &rc, const Box &bx, double *scal) {
	hipLaunchKernelGGL(k_recip_energy, dim3(1), dim3(256), 0, st, rc, bx, scal);
}

// ------------------------------------------------------------------------------------------------------
// static field
// ------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(64) void k_field_recip(AtomsDev at, RecipDev rc, double *__restrict__ e_part /*[gridDim.y][n_pad][3]*/) {
	const int i = blockIdx.x * kTile + threadIdx.x;
	const int per = (rc.K + (int)gridDim.y - 1) / (int)gridDim.y; // gridDim.y = recip_ksplit(n_pad) slices
	const int k0 = blockIdx.y * per, k1 = min(rc.K, k0 + per);
	const double4 p = at.xyzq[i];
	double ex = 0, ey = 0, ez = 0;
	for (int k = k0; k < k1; ++k) {
		const double4 kv = rc.kvec[k];
		const double4 sf = rc.sf[k];
		const double4 kw = rc.kw[k];
		const double ph = ((kv.x * p.x) + kv.y * p.y) + kv.z * p.z;
		double s, c;
		sincos(ph, &s, &c);
		const double g = s * sf.z - c * sf.w; // sin(k.r) C_k - cos(k.r) S_k  (:2877-2878)
		ex += kw.x * g;
		ey += kw.y * g;
		ez += kw.z * g;
	}
	double *o = e_part + ((size_t)blockIdx.y * at.n_pad + i) * 3;
	o[0] = ex;
	o[1] = ey;
	o[2] = ez;
}

// sum of the per-source-tile partial slots of 64 atoms: 8 groups of 64 threads stride over the slots, then a fixed-order
// fold through LDS (bit-reproducible).  Result valid in the threads with g == 0.
constexpr int kSlotGroups = 8;
__device__ __forceinline__ void slot_sum_64(const double *__restrict__ part, int n_slots, int n_pad, int i, int a, int g,
                                            double (*sh)[kTile][3], double out[3]) {
	double f[3] = {0, 0, 0};
	// (slot addresses as a wave-uniform base -- the group index and the tile's first atom are scalars -- plus the lane's 32-bit offset: a
	// per-lane 64-bit pointer costs more vector instructions than the sums themselves; same trick as k_dipole_update_panel, round 4)
	const int gs = __builtin_amdgcn_readfirstlane(g);
	const size_t tile0 = (size_t)__builtin_amdgcn_readfirstlane(i - a);
	const unsigned lane3 = 3u * (unsigned)a;
#pragma unroll 4
	for (int t = gs; t < n_slots; t += kSlotGroups) { // (unrolled: the loads of a group of four are in flight together; same order of sums)
		const double *__restrict__ q = part + ((size_t)t * n_pad + tile0) * 3;
		f[0] += q[lane3];
		f[1] += q[lane3 + 1];
		f[2] += q[lane3 + 2];
	}
	sh[g][a][0] = f[0];
	sh[g][a][1] = f[1];
	sh[g][a][2] = f[2];
	__syncthreads();
	if (g == 0) {
		for (int p = 0; p < 3; ++p) {
			double v = sh[0][a][p];
			for (int k = 1; k < kSlotGroups; ++k) v += sh[k][a][p];
			out[p] = v;
		}
	}
}

__global__ __launch_bounds__(512) void k_field_finalize(AtomsDev at, Box bx, int polar_ewald, const double *__restrict__ e_recip_part,
                                                        const double *__restrict__ part, int n_split, double gamma,
                                                        double *__restrict__ e_static, double *__restrict__ mu, double *__restrict__ e_real_out, int n_kslices) {
	__shared__ double sh[kSlotGroups][kTile][3];
	__shared__ double shk[kSlotGroups][kTile][3];
	const int a = threadIdx.x & 63, g = threadIdx.x >> 6;
	const int i = blockIdx.x * kTile + a;
	double real[3];
	double e[3] = {0, 0, 0};
	// the k-slices of the reciprocal field are added the way the real-space slots are: the eight groups stride over them, fixed-order
	// fold (a small system has up to kKSplitMax slices; a serial walk by one thread was a 19 us chain)
	if (polar_ewald) slot_sum_64(e_recip_part, n_kslices, at.n_pad, i, a, g, shk, e); // (block-uniform branch)
	slot_sum_64(part, n_split, at.n_pad, i, a, g, sh, real);
	if (g != 0) return;
	if (e_real_out)
		for (int p = 0; p < 3; ++p) e_real_out[3 * (size_t)i + p] = real[p];
	if (polar_ewald) {
		const double sc = 8.0 * kPi / bx.volume; // :2890
		for (int p = 0; p < 3; ++p) e[p] *= sc;
	}
	for (int p = 0; p < 3; ++p) e[p] += real[p]; // real_term is added after the reciprocal part was scaled
	const double al = at.alpha[i];
	for (int p = 0; p < 3; ++p) {
		e_static[3 * (size_t)i + p] = e[p];
		mu[3 * (size_t)i + p] = (al * e[p]) * gamma; // init_dipoles :3553-3556
	}
}

// the field sum with the factorised phases: lane = atom, its table column [dir][m][lane] in LDS (every lane reads the same m: no
// conflicts), k-vectors walked by all lanes together (l from uniform loads).  A workgroup is FOUR waves behind one table (round 4: one
// wave per (tile, slice) walked its 89 k-vectors behind three sincos and 21 dependent complex products of its own -- a 30 us latency
// chain at one wave per SIMD): waves 0..2 build one direction each, then every wave takes a quarter of the slice's k-vectors and the
// quarters are added in wave order.
constexpr int kFieldRecipWaves = 4;
__global__ __launch_bounds__(64 * kFieldRecipWaves) void k_field_recip_tab(AtomsDev at, Box bx, RecipDev rc, int kmax, double *__restrict__ e_part) {
	extern __shared__ double2 tab[];
	__shared__ double s_e[kFieldRecipWaves][3][kTile];
	const int KM1 = kmax + 1;
	const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = blockIdx.x * kTile + lane;
	if (w < 3) {
		const double2 base = phase_base(bx, at.xyzq[i], w);
		double2 v = make_double2(1.0, 0.0);
		tab[(size_t)(w * KM1) * kTile + lane] = v;
		for (int m = 1; m <= kmax; ++m) {
			v = cmul(v, base);
			tab[(size_t)(w * KM1 + m) * kTile + lane] = v;
		}
	}
	__syncthreads();
	const int per = (rc.K + (int)gridDim.y - 1) / (int)gridDim.y; // gridDim.y = recip_ksplit(n_pad) slices
	const int s0 = blockIdx.y * per, s1 = min(rc.K, s0 + per);
	const int quarter = (s1 - s0 + kFieldRecipWaves - 1) / kFieldRecipWaves;
	const int k0 = min(s1, s0 + w * quarter), k1 = min(s1, k0 + quarter);
	double ex = 0, ey = 0, ez = 0;
#pragma unroll 8
	for (int k = k0; k < k1; ++k) { // (unrolled: the wave-uniform loads of eight k-vectors are requested together instead of one latency per k-vector)
		const int4 l = rc.lvec[k];
		const double4 sf = rc.sf[k];
		const double4 kw = rc.kw[k];
		const double2 a = tab[(size_t)l.x * kTile + lane];
		double2 b = tab[(size_t)(KM1 + abs(l.y)) * kTile + lane];
		double2 c = tab[(size_t)(2 * KM1 + abs(l.z)) * kTile + lane];
		if (l.y < 0) b.y = -b.y;
		if (l.z < 0) c.y = -c.y;
		const double2 e = cmul(cmul(a, b), c);
		const double g = e.y * sf.z - e.x * sf.w; // sin(k.r) C_k - cos(k.r) S_k  (:2877-2878)
		ex += kw.x * g;
		ey += kw.y * g;
		ez += kw.z * g;
	}
	s_e[w][0][lane] = ex;
	s_e[w][1][lane] = ey;
	s_e[w][2][lane] = ez;
	__syncthreads();
	if (w == 0) {
		double *o = e_part + ((size_t)blockIdx.y * at.n_pad + i) * 3;
#pragma unroll
		for (int d = 0; d < 3; ++d) o[d] = ((s_e[0][d][lane] + s_e[1][d][lane]) + s_e[2][d][lane]) + s_e[3][d][lane];
	}
}

__global__ __launch_bounds__(64) void k_post_results(double *__restrict__ scal, double *__restrict__ out_host, double seq) {
	static_assert(S_COUNT + C_COUNT <= 64, "one wave posts the scalar block");
	const int t = threadIdx.x;
	if (t < S_COUNT + C_COUNT) { // (the counts travel as bit patterns: loads and stores only)
		const double v = scal[t];
		out_host[t] = v;
		scal[t] = 0.0;
	}
	__threadfence_system();
	if (t == 0) __hip_atomic_store(out_host + S_COUNT + C_COUNT, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_post_results(hipStream_t st, double *scal, double *out_host, double seq) {
	hipLaunchKernelGGL(k_post_results, dim3(1), dim3(64), 0, st, scal, out_host, seq);
}

void launch_field_recip(hipStream_t st, const AtomsDev &at, const Box &bx, const RecipDev &rc, int kmax, double *e_recip_part) {
	if (rc.lvec && kmax <= kRecipTabMaxK) {
		const size_t lds = (size_t)3 * kTile * (kmax + 1) * sizeof(double2);
		hipLaunchKernelGGL(k_field_recip_tab, dim3(at.n_pad / kTile, recip_ksplit(at.n_pad)), dim3(kTile * kFieldRecipWaves), lds, st, at, bx, rc, kmax, e_recip_part);
		return;
	}
	hipLaunchKernelGGL(k_field_recip, dim3(at.n_pad / kTile, recip_ksplit(at.n_pad)), dim3(kTile), 0, st, at, rc, e_recip_part);
}

void launch_field_finalize(hipStream_t st, const AtomsDev &at, const Box &bx, int polar_ewald, const double *e_recip_part, const double *part,
                           int n_split, double gamma, double *e_static, double *mu, double *e_real_out) {
	hipLaunchKernelGGL(k_field_finalize, dim3(at.n_pad / kTile), dim3(kTile * kSlotGroups), 0, st, at, bx, polar_ewald, e_recip_part, part, n_split,
	                   gamma, e_static, mu, e_real_out, recip_ksplit(at.n_pad));
}

// are_we_done_yet (:3215-3239) on the device.  ctl = { "some atom broke the tolerance in this iteration", iteration at which the solve
// converged (0: still iterating), ticket counter }.  Every update block ORs its verdict into ctl[0] and takes a ticket; the block
// that comes last closes the iteration: nobody broke => ctl[1] = it.  From then on the contraction and update kernels of the
// iterations the host had already enqueued return at once (the dipoles stay as they were), so the host looks at ctl[1] only once
// every few iterations instead of synchronising after each one.  Call from the threads of wave 0 of the block.
// host_flag (pinned, device-visible, may be null) = { last closed iteration, iteration at which the solve converged }: the host spins on
// it instead of synchronising the stream.
__device__ __forceinline__ void iteration_verdict(int *__restrict__ ctl, int *__restrict__ host_flag, int it, bool lane_broke) {
	const bool wave_broke = __any(lane_broke);
	if ((threadIdx.x & 63) != 0) return;
	if (wave_broke) atomicOr(&ctl[0], 1);
	__threadfence();
	const int ticket = atomicAdd(&ctl[2], 1);
	if (ticket != (int)gridDim.x - 1) return;
	__threadfence();
	const int broke = atomicOr(&ctl[0], 0);
	if (!broke) ctl[1] = it;
	ctl[0] = 0;
	ctl[2] = 0;
	if (host_flag) {
		__hip_atomic_store(host_flag + 1, broke ? 0 : it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(host_flag, it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// contract_dipoles tail :3586-3593, calc_dipole_rrms :3147-3177, are_we_done_yet :3227-3236
__device__ __forceinline__ void dipole_update_block(const AtomsDev &at, const double *__restrict__ e_static, const double *__restrict__ part,
                                                    int n_split, const double *__restrict__ mu_old, double *__restrict__ mu_new,
                                                    double *__restrict__ e_induced, int want_rrms, double *__restrict__ rrms_atom,
                                                    double allowed_sqerr, int *__restrict__ ctl, int *__restrict__ host_flag, int it) {
	__shared__ double sh[kSlotGroups][kTile][3];
	if (ctl && ctl[1] != 0) return; // converged in an earlier iteration (block-uniform)
	const int a = threadIdx.x & 63, g = threadIdx.x >> 6;
	const int i = blockIdx.x * kTile + a;
	double fsum[3];
	slot_sum_64(part, n_split, at.n_pad, i, a, g, sh, fsum);
	if (g != 0) return;
	const double al = at.alpha[i];
	const bool live = (i < at.n) && (al != 0.0);
	double f[3] = {0, 0, 0}, nm[3] = {0, 0, 0};
	if (live) {
		for (int p = 0; p < 3; ++p) {
			f[p] = fsum[p];
			nm[p] = al * (e_static[3 * (size_t)i + p] + f[p]);
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
		e_induced[3 * (size_t)i + p] = f[p];
	}
	if (want_rrms) {
		double r = sqrt(acc / nn);
		if (!isfinite(r)) r = 0.0;
		rrms_atom[i] = (i < at.n) ? r : 0.0;
	}
	if (ctl) iteration_verdict(ctl, host_flag, it, allowed_sqerr > 0.0 && broke && i < at.n);
}
__global__ __launch_bounds__(512) void k_dipole_update(AtomsDev at, const double *__restrict__ e_static, const double *__restrict__ part,
                                                       int n_split, const double *__restrict__ mu_old, double *__restrict__ mu_new,
                                                       double *__restrict__ e_induced, int want_rrms, double *__restrict__ rrms_atom,
                                                       double allowed_sqerr, int *__restrict__ ctl, int *__restrict__ host_flag, int it) {
	dipole_update_block(at, e_static, part, n_split, mu_old, mu_new, e_induced, want_rrms, rrms_atom, allowed_sqerr, ctl, host_flag, it);
}

__global__ __launch_bounds__(256) void k_dipole_reset(AtomsDev at, const double *__restrict__ e_static, double *__restrict__ mu) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= at.n_pad) return;
	const double a = at.alpha[i];
	for (int p = 0; p < 3; ++p) mu[3 * (size_t)i + p] = a * e_static[3 * (size_t)i + p]; // :3486
}

__device__ __forceinline__ void polar_energy_block(const AtomsDev &at, const double *__restrict__ mu, const double *__restrict__ e_static,
                                                   const double *__restrict__ rrms_atom, double *__restrict__ scal) {
	__shared__ double sh[4];
	double u = 0, rr = 0;
#pragma unroll 4
	for (int i = threadIdx.x; i < at.n; i += 256) { // (one block: unrolled so that the loads of four atoms are in flight together; same order of sums)
		const size_t b = 3 * (size_t)i;
		u += ((mu[b] * e_static[b]) + mu[b + 1] * e_static[b + 1]) + mu[b + 2] * e_static[b + 2];
		if (rrms_atom) {
			const double r = rrms_atom[i];
			if (isfinite(r)) rr += r;
		}
	}
	u = block_sum_256(u, sh);
	rr = block_sum_256(rr, sh);
	if (threadIdx.x == 0) {
		scal[S_POLAR] = -0.5 * u;       // :2618
		scal[S_RRMS] = rr / (double)at.n; // get_dipole_rrms :2656
	}
}
__global__ __launch_bounds__(256) void k_polar_energy(AtomsDev at, const double *__restrict__ mu, const double *__restrict__ e_static,
                                                      const double *__restrict__ rrms_atom, double *__restrict__ scal) {
	polar_energy_block(at, mu, e_static, rrms_atom, scal);
}

// the tail of a polarizable evaluation in ONE launch: block 0 the polarization energy, block 1 the fold of the pair sweep's partials (both
// single-block, fixed-order sums; the fold used to run on the side stream, whose fork and join each cost the main stream a barrier packet)
__global__ __launch_bounds__(256) void k_polar_energy_and_pairs(AtomsDev at, const double *__restrict__ mu, const double *__restrict__ e_static,
                                                                const double *__restrict__ rrms_atom, const double *__restrict__ block_part,
                                                                const int *__restrict__ block_cnt, int nb, double *__restrict__ scal,
                                                                long long *__restrict__ cnt) {
	if (blockIdx.x == 0) polar_energy_block(at, mu, e_static, rrms_atom, scal);
	else reduce_pairs_block(block_part, block_cnt, nb, scal, cnt);
}

void launch_dipole_update(hipStream_t st, const AtomsDev &at, const double *e_static, const double *part, int n_split, const double *mu_old,
                          double *mu_new, double *e_induced, int want_rrms, double *rrms_atom, double allowed_sqerr, int *ctl, int *host_flag, int it) {
	hipLaunchKernelGGL(k_dipole_update, dim3(at.n_pad / kTile), dim3(kTile * kSlotGroups), 0, st, at, e_static, part, n_split, mu_old, mu_new,
	                   e_induced, want_rrms, rrms_atom, allowed_sqerr, ctl, host_flag, it);
}
void launch_dipole_reset(hipStream_t st, const AtomsDev &at, const double *e_static, double *mu) {
	hipLaunchKernelGGL(k_dipole_reset, dim3((at.n_pad + 255) / 256), dim3(256), 0, st, at, e_static, mu);
}
void launch_polar_energy(hipStream_t st, const AtomsDev &at, const double *mu, const double *e_static, const double *rrms_atom, double *scal) {
	hipLaunchKernelGGL(k_polar_energy, dim3(1), dim3(256), 0, st, at, mu, e_static, rrms_atom, scal);
}
void launch_polar_energy_and_pairs(hipStream_t st, const AtomsDev &at, const double *mu, const double *e_static, const double *rrms_atom,
                                   const double *block_part, const int *block_cnt, int nb, double *scal, long long *cnt) {
	hipLaunchKernelGGL(k_polar_energy_and_pairs, dim3(2), dim3(256), 0, st, at, mu, e_static, rrms_atom, block_part, block_cnt, nb, scal, cnt);
}

// ------------------------------------------------------------------------------------------------------
// dense thole_amatrix rows (reference :2661-2770).  One thread per (row atom, column atom) 3x3 block.
// ------------------------------------------------------------------------------------------------------
template <bool ORTHO>
__global__ __launch_bounds__(256) void k_amatrix_rows(AtomsDev at, const int *__restrict__ slot_of, Box bx, double lambda, int atom0,
                                                      int natoms_rows, double *__restrict__ a) {
	const int j = blockIdx.x * 256 + threadIdx.x;
	const int ir = blockIdx.y; // row atom index relative to atom0
	if (j >= at.n || ir >= natoms_rows) return;
	const int i = atom0 + ir;
	const size_t ld = 3 * (size_t)at.n;
	double *blk = a + (3 * (size_t)ir) * ld + 3 * (size_t)j;
	if (i == j) {
		const double al = at.alpha[slot_of[i]];
		for (int p = 0; p < 3; ++p)
			for (int q = 0; q < 3; ++q) blk[p * ld + q] = (p == q) ? ((al != 0.0) ? 1.0 / al : kMaxValue) : 0.0;
		return;
	}
	const int lo = min(i, j), hi = max(i, j); // the reference fills the (lo,hi) block and COPIES it to (hi,lo) (:2762-2764)
	const double4 pl = at.xyzq[slot_of[lo]], ph = at.xyzq[slot_of[hi]];
	double d[3];
	const double r = min_image<ORTHO>(bx, pl.x - ph.x, pl.y - ph.y, pl.z - ph.z, d[0], d[1], d[2]);
	double ir3, ir5;
	if (r == 0.0)
		ir3 = ir5 = kMaxValue;
	else {
		const double inv = 1.0 / r;
		ir3 = inv * inv * inv;
		ir5 = ir3 * inv * inv;
	}
	const double r2 = r * r, l2 = lambda * lambda, l3 = l2 * lambda;
	const double explr = exp(-lambda * r);
	const double damp1 = 1.0 - explr * (0.5 * l2 * r2 + lambda * r + 1.0);
	const double damp2 = damp1 - explr * (l3 * r2 * r / 6.0);
	for (int p = 0; p < 3; ++p)
		for (int q = 0; q < 3; ++q) {
			double v = -3.0 * d[p] * d[q] * damp2 * ir5;
			if (p == q) v += damp1 * ir3;
			blk[p * ld + q] = v;
		}
}

void launch_amatrix_rows(hipStream_t st, const AtomsDev &at, const int *slot_of, const Box &bx, double polar_damp, int row0, int nrows, double *a) {
	const int atom0 = row0 / 3, nat = nrows / 3;
	dim3 grid((at.n + 255) / 256, nat), block(256);
	if (bx.ortho)
		hipLaunchKernelGGL(k_amatrix_rows<true>, grid, block, 0, st, at, slot_of, bx, polar_damp, atom0, nat, a);
	else
		hipLaunchKernelGGL(k_amatrix_rows<false>, grid, block, 0, st, at, slot_of, bx, polar_damp, atom0, nat, a);
}

// ------------------------------------------------------------------------------------------------------
// positions that already live in device memory ([n][3] fp64) -> xyzq.xyz (charge kept)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_set_positions(const double *__restrict__ pos, const int *__restrict__ perm, double4 *__restrict__ xyzq, int n) {
	const int k = blockIdx.x * 256 + threadIdx.x;
	if (k >= n) return;
	const int i = perm[k];
	double4 v = xyzq[k];
	v.x = pos[3 * (size_t)i + 0];
	v.y = pos[3 * (size_t)i + 1];
	v.z = pos[3 * (size_t)i + 2];
	xyzq[k] = v;
}
void launch_set_positions(hipStream_t st, const double *pos_dev, const int *perm, double4 *xyzq, int n) {
	if (n > 0) hipLaunchKernelGGL(k_set_positions, dim3((n + 255) / 256), dim3(256), 0, st, pos_dev, perm, xyzq, n);
}

} // namespace mpmc
