// kernels.h -- launch wrappers of the gfx950 kernels (implemented in kernels.hip), used by context.cpp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "pair_math.h"

namespace mpmc {

constexpr int kTile = 64; // one wavefront owns 64 i-atoms; j-atoms are staged in LDS 64 at a time
constexpr int kKSplit = 8; // k-vector range split of the reciprocal field kernel: at least this many slices, ...
constexpr int kKSplitMax = 64, kKSplitWaves = 1024;
// ... and more when there are few tiles: one wave walks its slice of the k-vectors serially (26 us for the whole range over 8 slices,
// whatever the number of atoms), so a small system cuts the range until about kKSplitWaves waves share it.  A function of n_pad alone:
// the launcher of the field kernel and the kernel that adds the slices up both call it.
inline int recip_ksplit(int n_pad) {
	const int nt = n_pad / kTile;
	int ks = kKSplit;
	while (ks < kKSplitMax && nt * ks < kKSplitWaves) ks *= 2;
	return ks;
}
// doubles in the slice buffer [recip_ksplit(n_pad)][n_pad][3] for any n_pad <= max_pad
inline size_t recip_slices_capacity(size_t max_pad) { return std::max((size_t)kKSplit * 3 * max_pad, (size_t)2 * kKSplitWaves * kTile * 3); }

// device view of one System's atoms (struct-of-arrays, padded to a multiple of kTile)
struct AtomsDev {
	const double4 *xyzq; // x, y, z, charge
	const double2 *lj;   // |sigma|, sqrt(epsilon)
	const int2 *mf;      // molecule id, AF_* flags
	const double *alpha; // polarizability
	const double *eps;   // epsilon (atom self-LRC term)
	const double *inv_molmass; // 1 / (mass of the owning molecule, amu): Feynman-Hibbs reduced masses (may be null)
	int n, n_pad;
};

// slots of the scalar result vector on the device
enum : int {
	S_LJ = 0, S_LRC_PAIR, S_ES_REAL, S_ES_INTRA, // pair kernel
	S_ES_RECIP, S_ES_SELF, S_LRC_SELF,           // recip / atom kernel
	S_POLAR, S_RRMS,                             // polarization
	S_COUNT = 16
};
enum : int { C_LJ_IN = 0, C_ES_IN, C_INTRA, C_RDX, C_ESX, C_FROZEN, C_COUNT = 8 };
// tile-pair classes written by k_classify (any cell: DESIGN section 3 has the bound used for skewed cells): lower bound of the
// minimum-image distance between the two tiles' bounding boxes
enum : int {
	CLS_BEYOND_CUTOFF = 1, // > cutoff: no pair of the tile pair passes any cutoff predicate
	CLS_THOLE_FAR = 2,     // lambda*r > kTholeFarX for every pair: T is the bare dipole tensor (see below for what that drops)
	CLS_UNIFORM_X = 4,     // per dimension (x, y, z = 4, 8, 16): one periodic image index serves all 4096 pairs of the tile pair in
	CLS_UNIFORM_Y = 8,     // that dimension (wrapped coordinates); tp_shift holds the lattice vector component B img
	CLS_UNIFORM_Z = 16
};
// Beyond lambda r = kTholeFarX a tile pair's tensors are the bare dipole tensors (nothing stored, recomputed from the positions).  The
// Thole factors differ from 1 by exp(-x)(x^3/6 + x^2/2 + x + 1) = 4.7e-10 at x = 30 -- a RELATIVE error of tensors that are themselves
// < 1/(14 A)^3 = 3.6e-4 of the self term 1/alpha, and only for the few pairs that sit right at the boundary (the class is decided by the
// tiles' bounding boxes, so almost every pair of a far tile pair is much further out).  Measured on the 10 000-atom box: the polarization
// energy moves by 6e-15 relative between x = 40 and x = 30 (1.4e-12 at x = 26), dipoles by < 1e-13 of the largest one.  The store shrinks from 5299 to
// 2576 of 12 403 tile pairs (347 -> 169 MB read per Jacobi iteration), which is worth 6-8 % of the whole-job rate: with 32 beads in
// flight the aggregate HBM traffic (3.7 TB/s at x = 40) is a co-bottleneck.  A compile-time constant: it is the knob of an approximation
// (rounds 1-2 read it from the environment).
constexpr double kTholeFarX = 30.0;

// reciprocal space: structure factors for every k, then energy + O(N) atom terms
struct RecipDev {
	const double4 *kvec; // kx, ky, kz, k^2            [K]
	const double *w_en;  // exp(-k^2/4a^2)/k^2         [K]
	const double4 *kw;   // k_p/k^2 exp(-k^2/4ap^2), _ [K]
	const int4 *lvec;    // integer l of k = 2 pi R l       [K] (null: one sincos per (k, atom))
	double4 *sf;         // SF_re, SF_im (energy: non-frozen, q != 0), C_k, S_k (field: all atoms)  [K]
	int K;
};
constexpr int kRecipTabMaxK = 15; // phase tables of 3 x 64 x (kmax+1) complex numbers must fit 64 KiB of LDS
// sf_part: [n_tiles][K] scratch of the factorised form (may be null: direct sincos form)
void launch_recip_sf(hipStream_t st, const AtomsDev &at, const Box &bx, const RecipDev &rc, int kmax, double4 *sf_part);
// reciprocal energy + O(N) atom terms: coulombic_self, lj_lrc_self, and the PAIR long-range correction summed in O(N)
// through moments of (sqrt(eps), |sigma|) (Lorentz-Berthelot makes the pair term a polynomial in sigma_i + sigma_j)
// position-independent terms (pair LRC, self LRC, Ewald self) into their three slots of the scalar block; part_scratch: kAtomTermScratch doubles
constexpr size_t kAtomTermScratch = 64 * 32;
void launch_atom_terms(hipStream_t st, const AtomsDev &at, const Box &bx, double ewald_alpha, int rd_lrc, int do_es, double *part_scratch,
                       double *scal);

// static field
void launch_field_recip(hipStream_t st, const AtomsDev &at, const Box &bx, const RecipDev &rc, int kmax, double *e_recip /*[recip_ksplit(n_pad)][n_pad][3]*/);
// E0 = recip*(8 pi/V) + sum_s part ; mu0 = gamma * alpha * E0
void launch_field_finalize(hipStream_t st, const AtomsDev &at, const Box &bx, int polar_ewald, const double *e_recip,
                           const double *part, int n_split, double gamma, double *e_static, double *mu,
                           double *e_real_out = nullptr /*the real-space part alone (sum of the slots): what a trial move updates incrementally*/);

// new_mu = alpha (E0 + F) ; optionally rrms per atom.  Precision-terminated solves pass ctl = { broke, converged-at, ticket } (device
// ints, zeroed at the start of the solve) and the iteration number: are_we_done_yet runs on the device, the kernels of iterations
// enqueued after convergence return at once (see iteration_verdict in kernels.hip)
void launch_dipole_update(hipStream_t st, const AtomsDev &at, const double *e_static, const double *part, int n_split,
                          const double *mu_old, double *mu_new, double *e_induced, int want_rrms, double *rrms_atom,
                          double allowed_sqerr, int *ctl, int *host_flag /*pinned {closed iteration, converged-at}, may be null*/, int it);
// iterator failure: mu = alpha * E0
void launch_dipole_reset(hipStream_t st, const AtomsDev &at, const double *e_static, double *mu);
// end of an evaluation: the scalar block [S_COUNT doubles][C_COUNT int64] goes to pinned host memory, the launch number behind it (a host
// that polls that slot finds the results complete), and the device block is left zeroed for the next evaluation
void launch_post_results(hipStream_t st, double *scal, double *out_host, double seq);
void launch_polar_energy(hipStream_t st, const AtomsDev &at, const double *mu, const double *e_static, const double *rrms_atom,
                         double *scal);

// dense thole_amatrix rows (parity / DENSE solver)
// rows and columns are ORIGINAL atom indices; slot_of maps them to the device order
void launch_amatrix_rows(hipStream_t st, const AtomsDev &at, const int *slot_of, const Box &bx, double polar_damp, int row0, int nrows,
                         double *a /*[nrows][3n]*/);

// ---- symmetric production kernels (kernels_sym.hip) -------------------------------------------------------
struct FusedParams {
	double ewald_alpha, polar_ewald_alpha, polar_damp;
	int rd_lrc;
	int do_es;    // electrostatics on
	int do_field; // 0 none, 1 Ewald real_term, 2 thole_field_nopbc
	int do_thole; // write the (a,b) tensor store
	// adjacent physics (extended kernel variant only)
	int wolf;             // coulombic_wolf instead of the erfc term
	int fh_order;         // 0 off, 2 or 4: Feynman-Hibbs corrections
	double fh_c2, fh_c4;  // M2A2 hbar^2 / (24 kB T amu2kg),  M2A4 hbar^4 / (1152 kB^2 T^2 amu2kg^2)
	double wolf_erfa_over_r, wolf_inv_r2; // erf(alpha R)/R, 1/R^2
	double thole_far_x; // lambda r beyond which the exponential damping is dropped (the value the tile classes were made with)
	int store_only;     // nothing but the Thole tensor store (trial moves of polarizable boxes: energies and field come from the delta kernels)
	int touch_n;        // store-only passes: >= 0 restricts the pass to the tile pairs that contain one of touch[0 .. touch_n) (the tiles of
	int touch[8];       // the moved atoms: every other tile pair keeps the tensors it has); < 0: all tile pairs
	int pair_waves;     // waves per tile pair of the sweep: 4 for small tables (n_tile_pairs <= kPairSplitMax), else 1
};
// Tables up to this many tile pairs run everything on ONE stream: forking the side stream (reciprocal space, panel table) and joining it
// costs two cross-stream waits, more than the 20-odd us of work they overlap.  Measured (profiles/r02_one_stream.txt), one evaluation at
// a time: -11 % at 216 atoms, -7 % at 1000, -5 % at 2000, -2 % at 3000 (1128 tile pairs), level at 4000 (2016), +3 % at 5000, +2 % at
// 10 000; with 32 beads in flight: +7 % evaluations/s at 3000 atoms on one stream, level at 5000 and 7000, -2.6 % at 10 000.
constexpr int kOneStreamMaxPairs = 1536;
// Tables up to this many tile pairs run the pair sweep with four waves per tile pair.  Measured (profiles/r02_pair_waves.txt): one
// evaluation at a time four waves win at every size (-12 % at 216 atoms, -11 % at 3000, -7 % at 5000, -3 % at 7000, -1 % at 10 000);
// with 32 beads in flight they are level at 3000 atoms, +3..7 % at 5000, +1.4 % at 7000 (6105 tile pairs) and level (-0.3 %) at
// 10 000 (12 403) -- so the switch sits between the last size with a gain in both regimes and the first without one.
constexpr int kPairSplitMax = 8192;
// every unordered pair once: energies + counts (block partials), static-field partials fpart[nt][n_pad][3],
// Thole store ab[n_tile_pairs][64*64] (double2 = 16 B per pair)
// tp_list (may be null): the launch covers the tile pairs tp_list[0 .. n_tile_pairs) only -- what the fast sweep below leaves to it
void launch_pair_fused(hipStream_t st, const AtomsDev &at, const Box &bx, const FusedParams &fp, const int2 *tile_pairs,
                       const int *cls, int n_tile_pairs, double *block_part /*[ntp][2]*/, int *block_cnt /*[ntp][2]*/, double *fpart, double2 *ab,
                       const int *tp_list = nullptr);
// ---- the fast pair sweep (kernels_pair.hip): any cell, Ewald electrostatics with alpha r_c inside the erfc table; every tile pair
// except those with an atom that changes lj_mix (kAtomFlagsMixing: sigma < 0, dispersion coefficients) ----
struct PairSweepParams {
	double alpha_scaled_half, polar_damp_half, thole_far_x; // half of: the Ewald alpha over the erfc table's piece width, the Thole damping parameter (the sweep works with 2 r)
	int store;      // write the Thole tensor store
	int nt;         // tiles
	int have_shift; // tp_shift / CLS_UNIFORM_* are valid
	int split;      // 0 | 1 | 2: two waves per tile pair (half the steps each), two tile pairs per workgroup -- never / every entry / the entries behind n_main
	int n_main;     // workgroups [0, n_main) take whole entries of the table, the ones behind them half entries
	int fast;       // fused geometry with the tile pair's band around the cutoff thresholds (tp_shift.w); 0: the reference's form everywhere
};
// Tables with more tile pairs than this take the sweep by default.  Measured (profiles/r03_sweep_sizes.txt), k_pair_fused (four waves per
// tile pair up to kPairSplitMax) against the sweep: one evaluation at a time +2 % at 3000 atoms (1128 tile pairs), -3 % at 5000 (3160),
// -5 % at 7000, -9 % at 10 000; 32 beads in flight the sweep wins from 3000 atoms on (+4 % evaluations/s, +4.5 % at 5000, +9 % at 7000).
constexpr int kSweepMinPairs = 2048;
// two waves per tile pair (half the steps each; kernels_pair.hip): for the last kSweepSplitTailPermille / 1000 of the work table (round 5; round 4
// measured "all" against "none": faster alone, 0.7 % slower with 32 evaluations in flight)
constexpr int kSweepSplitTailPermille = 250;
// host: the device layout of the erfc table, 3 * 512 double2 = (c0,c1)[512], (c2,c3)[512], (c4,c5)[512]  (erfc_table.cpp)
constexpr int kErfTableDouble2 = 3 * 512;
void erfc_table_device_layout(double2 *out /*[kErfTableDouble2]*/);
// work table of the sweep: one workgroup per entry { J, I0 }, its four waves take the tile pairs (I0 .. I0+3, J); returns the
// number of entries (out may be null: count only)
int pair_sweep_blocks(int n_tiles, int2 *out);
// whether the sweep can serve this evaluation (switches, alpha r_c inside the erfc table).  Frozen, chargeless, sigma- or epsilon-less
// atoms are masked by the sweep itself (its MODE 2); only tile pairs with a kAtomFlagsMixing atom are skipped by it and must go
// through launch_pair_fused with their list (context.cpp builds it with the same constant)
bool pair_sweep_covers(const Box &bx, const FusedParams &fp, double ewald_alpha);
void launch_pair_sweep(hipStream_t st, const AtomsDev &at, const Box &bx, const FusedParams &fp, bool intra /*some molecule has more than one atom*/,
                       const int2 *blocks, int n_blocks, const int *cls, const double4 *tp_shift /*null: no uniform images*/,
                       const double2 *erf_tab, double *block_part, int *block_cnt, double *fpart, double2 *ab, int split_mode = 0 /*0 | 1 | 2: see PairSweepParams*/,
                       int n_split_tail = 0 /*mode 2: how many entries at the end of the table are halved*/, bool fast_geometry = false, int lds_pad_bytes = 0 /*unused dynamic LDS per workgroup: fewer workgroups per CU*/,
                       int replicas = 1 /*measurement only: the grid repeated in y (mpmc_debug_time_pair with panel_replicas)*/);
void launch_reduce_pairs(hipStream_t st, const double *block_part, const int *block_cnt, int nb, double *scal, long long *cnt);
// polarizable evaluations: launch_polar_energy and launch_reduce_pairs as the two blocks of one launch (the tail of the evaluation)
void launch_polar_energy_and_pairs(hipStream_t st, const AtomsDev &at, const double *mu, const double *e_static, const double *rrms_atom,
                                   const double *block_part, const int *block_cnt, int nb, double *scal, long long *cnt);
// LJ (+ counts) of a small system in ONE launch: no tile classes, the last-arriving block folds the partials and writes the scalar
// vector [S_COUNT doubles][C_COUNT int64][seq] into pinned host memory (seq last: a host polling that slot finds the results
// complete); `counter` is a zeroed int the kernel leaves zeroed
void launch_pair_lj_single(hipStream_t st, const AtomsDev &at, const Box &bx, const FusedParams &fp, const int2 *tile_pairs, int n_tile_pairs,
                           double *block_part, int *block_cnt, int *counter, double *out_host, double seq);
// S_ES_RECIP = (4 pi / V) sum_k w_k |S_k|^2 from the structure factors of this configuration
void launch_recip_energy(hipStream_t st, const RecipDev &rc, const Box &bx, double *scal);
// position-independent pair-flag counts (n_intra, n_rd_excluded, n_es_excluded, n_frozen): once per atom upload
void launch_static_counts(hipStream_t st, const AtomsDev &at, const int2 *tile_pairs, int n_tile_pairs, int *block_cnt /*[ntp][4]*/,
                          long long *cnt4);
// sum over intramolecular non-frozen pairs of q_i q_j erf(alpha r)/r with the plain (non-image) distance (:1503-1504)
void launch_intra_terms(hipStream_t st, const AtomsDev &at, const int *slot_of, double ewald_alpha, double *scal);
// per-tile bounding boxes (wrapped fractional coordinates) and tile-pair classes (CLS_*)
void launch_tile_classes(hipStream_t st, const AtomsDev &at, const Box &bx, const int2 *tile_pairs, int n_tile_pairs, double polar_damp,
                         double *tile_bounds /*[nt][12]*/, int *cls, double4 *tp_shift /*[ntp], may be null*/,
                         const double origin_f[3] /*fractional origin of the spatial sort: where the periodic wrap is cut*/,
                         double thole_far_x = kTholeFarX /*lambda r beyond which a tile pair is CLS_THOLE_FAR*/);
// one Jacobi contraction over all tile pairs (class read per block): part[nt][n_pad][3].  The matrix-free solver (and panels = 0);
// evaluations with a tensor store take the panel form below, in any cell
void launch_dipole_iter_hybrid(hipStream_t st, const AtomsDev &at, const Box &bx, const double *mu, const int2 *tile_pairs,
                               const int *cls, const double4 *tp_shift /*null: no uniform-image fast path*/, int n_tile_pairs,
                               const double2 *ab /*null: matrix-free, tensors inside the damping range rebuilt from the positions*/,
                               double *part, double polar_damp, const int *converged = nullptr);
// ---- panel form of the contraction (kernels_panel.hip; any cell): two tile pairs that share their j-tile per workgroup ----
int panel_segment_entries(int J); // entries the work table reserves for j-tile J; seg[J] = their running sum
// the work table of the panel kernel: per j-tile its diagonal tile pair, panels of two tile pairs of equal class, odd singles
void launch_build_panels(hipStream_t st, const int *cls, int n_tiles, const int *seg /*[n_tiles + 1]*/, int4 *panels,
                         int *arrive /*[n_tiles] arrival counters of the fused update, zeroed here; may be null*/);
// the dipole update riding the contraction's launch: the workgroup that delivers the last contribution to a tile updates that tile
struct PanelFuse {
	int *arrive; // [n_tiles] arrival counters, zero between launches; null: no fused update (launch_dipole_update_panel follows)
	int reverse; // entries in descending j-tile order (default); 0: table order (measurement)
	const double *e_static;
	const int *seg;
	double *mu_new, *e_induced /*may be null*/, *rrms_atom;
	double allowed_sqerr;
	int *ctl, *host_flag;
	int it, want_rrms;
	int probe; // measurement only (fused_update = 2): arrive but skip the update
};
// i-side partial sums -> part[J][I atoms] (the usual slots, upper triangle + diagonal only; [3][64] inside a slot); j-side -> gpart[entry][3][64]
void launch_dipole_iter_panel(hipStream_t st, const AtomsDev &at, const Box &bx, const double *mu, const int2 *tile_pairs,
                              const double4 *tp_shift, const int4 *panels, int n_entries, const double2 *ab, double *part, double *gpart,
                              const int *converged = nullptr, long long *trace = nullptr /*measurement only (trace_panel)*/,
                              int replicas = 1 /*measurement only: the grid repeated in y (mpmc_debug_time_panel)*/,
                              const PanelFuse *fuse = nullptr);
void launch_dipole_update_panel(hipStream_t st, const AtomsDev &at, const double *e_static, const double *part, const double *gpart, const int *seg,
                                const double *mu_old, double *mu_new, double *e_induced, int want_rrms, double *rrms_atom, double allowed_sqerr,
                                int *ctl, int *host_flag, int it, int waves = 16 /*waves per workgroup: 16 | 4*/);
// lane-rotation primitive self-test: out[l] = lane whose value lane l received (must be (l+1)&63)
void launch_rot_selftest(hipStream_t st, int *out);

// ---- dense solver (kernels_dense.hip): the reference's 3N x 3N layout, contraction on the fp64 matrix cores ------------------------
// a: (3 n_pad)^2 doubles, slot order, diagonal 3x3 blocks and padded slots zero
void launch_dense_build(hipStream_t st, const AtomsDev &at, const Box &bx, double polar_damp, double *a, bool upper_only = false);
// the symmetric form (default): tile pairs I <= J only, both products per 16 x 16 block; part[source tile][n_pad][3] like the other solvers
void launch_dense_symv(hipStream_t st, const double *a, int n_pad, const double *x, const int2 *tile_pairs, int n_tile_pairs, double *part);
// part[chunk][n_pad][3] = - (rows of the chunk of A) . x  (A symmetric); k_dipole_update sums the chunks
void launch_dense_matvec(hipStream_t st, const double *a, int n_pad, const double *x, int n_chunks, double *part);

// ---- Gauss-Seidel sweeps (kernels_gs.hip): `polar_gs on`, identity atom order, matrix-free ---------------------------
// one sweep over all tiles in atom order: mu is updated in place, e_induced receives each atom's induced field.  part: the slots of the
// symmetric kernel [nt][n_pad][3]; U, L: [n_pad][3] (tiles above / below); tile_pairs, cls, tp_shift: this evaluation's tile-pair tables
size_t gs_block_store_elements(int n_tiles); // double2 elements of the in-tile block store of the Gauss-Seidel sweeps
void launch_gs_blocks(hipStream_t st, const AtomsDev &at, const Box &bx, double polar_damp, double2 *blocks);
void launch_gs_sweep(hipStream_t st, const AtomsDev &at, const Box &bx, double polar_damp, const double *e_static, double *mu, double *e_induced,
                     double *part, const int2 *tile_pairs, const int *cls, const double4 *tp_shift, int n_tile_pairs, double *U, double *L,
                     const double2 *blocks);
void launch_gs_finish(hipStream_t st, const AtomsDev &at, const double *mu_old, const double *mu_new, int want_rrms, double *rrms_atom,
                      double allowed_sqerr, int *not_done_flag);

// ---- trial moves (kernels_delta.hip) ----------------------------------------------------------------------
// out4 = { d lj_pairs, d es_real(erfc part), d intramolecular term, E_recip of the trial structure factors }, dcnt2 = { d n_lj, d n_es }
// a trial move short enough to travel in the kernel arguments (no staging copy): trial positions + charge, slot and original index
constexpr int kMvInline = 8;
struct MvInline {
	double nw[kMvInline][4];
	int slot[kMvInline], orig[kMvInline];
};
void launch_commit_positions_inline(hipStream_t st, double4 *xyzq, const MvInline &inl, int m);
void launch_delta(hipStream_t st, const AtomsDev &at, const int *slot_of, const Box &bx, const RecipDev &rc,
                  const FusedParams &fp /*ewald_alpha and the Wolf / Feynman-Hibbs fields*/, int do_es,
                  const int *mv_slot, const int *orig_of_mv, const double4 *mv_new, int m, int *moved_idx, double4 *sf_trial,
                  double *block_part, int *block_cnt, double *out4, long long *dcnt2,
                  double *host_out /*pinned [9]: the result and, last, the launch number `seq`*/, double seq,
                  const MvInline *inl = nullptr /*non-null: the move is in here (m <= kMvInline), the device lists are not read*/);
void launch_commit_positions(hipStream_t st, double4 *xyzq, const int *mv_slot, const double4 *mv_new, int m);
// polarizable boxes: e_real_trial = e_real + (real-space static field of the pairs with a moved atom, new minus old geometry);
// dk_part: scratch [n_tiles][m][3]
void launch_delta_field(hipStream_t st, const AtomsDev &at, const Box &bx, double polar_ewald_alpha, int polar_ewald, const int *mv_slot,
                        const double4 *mv_new, int m, int *moved_idx, const double *e_real, double *e_real_trial, double *dk_part);
// resident positions of the moved atoms <-> mv_new (call again to undo)
void launch_swap_positions(hipStream_t st, double4 *xyzq, const int *mv_slot, double4 *mv_new, int m);

// device-resident positions [n][3] in original atom order -> xyzq[slot].xyz (perm[slot] = original index)
void launch_set_positions(hipStream_t st, const double *pos_dev, const int *perm, double4 *xyzq, int n);

} // namespace mpmc
