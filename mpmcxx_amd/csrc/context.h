// context.h -- INTERNAL to libmpmc_energy.so: the device-resident state behind an mpmc_ctx handle and the helpers its translation units
// share (context.cpp: lifetime, box, options, atoms; evaluate.cpp: one energy evaluation and its pieces; trial.cpp: per-move delta
// energies; pi.cpp: the path-integral bead loop).  Not installed, not part of the C ABI (include/mpmc_energy.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mpmc_energy.h"
#include "kernels.h"

using namespace mpmc;

namespace mpmc {
extern thread_local std::string g_create_error; // last create-time error of this thread (context.cpp)
}

struct EvPair {
	hipEvent_t a, b;
	int cls;
};

// Measurement / A-B switches (mpmc_debug_configure; the library reads no environment variable for any of them).  The defaults are the
// production path; none of them changes a result beyond the last bits (tests/test_gpu_round3_fixes.py holds every one to the reference).
// Pinned host memory comes from the HIP runtime's own pool: an address freed by one context is handed to the next.  The runtime is not
// instrumented, so the sanitizer build (tools/host_tsan.sh) is told here what the pool's lock orders: every release happens before every
// later allocation -- otherwise the old owner's writes and the new owner's count as a race between two threads that never shared anything.
#if defined(__SANITIZE_THREAD__)
extern "C" void AnnotateHappensBefore(const char *file, int line, const volatile void *tag);
extern "C" void AnnotateHappensAfter(const char *file, int line, const volatile void *tag);
inline char g_pinned_pool_tag;
#endif
template <class T>
inline hipError_t pinned_alloc(T **p, size_t bytes) {
	hipError_t e = hipHostMalloc((void **)p, bytes);
#if defined(__SANITIZE_THREAD__)
	AnnotateHappensAfter(__FILE__, __LINE__, &g_pinned_pool_tag);
#endif
	return e;
}
inline hipError_t pinned_free(void *p) {
#if defined(__SANITIZE_THREAD__)
	AnnotateHappensBefore(__FILE__, __LINE__, &g_pinned_pool_tag);
#endif
	return hipHostFree(p);
}

struct mpmc_tuning {
	int stream_mode = -1;   // "side_stream": -1 by table size (kOneStreamMaxPairs), 0 never fork the side stream, 1 always
	int pair_kernel = 0;    // "pair_kernel": 0 the fast sweep (kernels_pair.hip) where it applies and the table is large, 1 never, 2 wherever it applies
	int pair_waves = 0;     // "pair_waves": waves per tile pair of k_pair_fused, 0 by table size (kPairSplitMax), 1 | 4
	int sort_grid = -1;           // "sort_grid": -1 the aligned-grid spatial order where the table is large enough (round 4), 0 the nested count-based bisection of rounds 1-3
	int sort_nx = 0, sort_ny = 0; // "sort_nx" / "sort_ny": > 0: the aligned grid with exactly this many x slabs / y strips (measurement)
	bool side_after_sweep = true; // "side_after_sweep": two streams: the side stream's kernels are enqueued behind the pair sweep's launch (0: in front, rounds 1-3)
	bool lazy_side_stream = true; // "lazy_side_stream": the side stream is created when an evaluation first forks, not with the context -- the runtime deals
	                              // hardware queues to streams in turn, and an ensemble that never forks then has its main streams on all four (+0.6 %)
	bool poll_retire = true;      // "poll_retire": a polled-for evaluation queries its streams afterwards so that the runtime retires the finished commands
	bool poll_long = true;        // "poll_long": evaluations of large tables are polled for before the wait synchronises the stream (0: rounds 1-3)
	bool tail_fused = true;       // "tail_fused": polarization energy and the fold of the pair partials in one launch (0: the fold forks the side stream)
	bool dense_symmetric = true; // "dense_symmetric": the dense solver reads the upper block triangle of A only (0: rounds 1-3, the whole matrix)
	bool fast_geometry = true; // "fast_geometry": fused minimum image in the pair sweep, the reference's form only inside a 1e-9 band around the cutoff (0: everywhere)
	int pair_split = -1;    // "pair_split": two waves per tile pair in the fast sweep (half-length workgroups): -1 the last part of the table (round 5), 0 never | 1 everywhere
	int pair_split_tail = -1; // "pair_split_tail": per mille of the sweep's work table that is halved under pair_split = -1 (default kSweepSplitTailPermille)
	int sweep_order = 1; // "sweep_order": the pair sweep's work table by descending j-tile (1, round 5: the short rows with their partial entries end the launch; -2 to -4 % per lone
	                     // launch without field and store, level with them) | 0 ascending (rounds 3-4)
	int update_waves = 0; // "update_waves": waves per workgroup of the dipole update launch: 0 = 4 (round 5) | 1 | 2 | 4 | 16 (rounds 2-4).  A 16-wave workgroup needs
	                      // sixteen free wave slots on ONE CU at once: with 32 beads in flight the launch waited ~170 us for them (0.02 ms with four waves, +2.3 %
	                      // evaluations/s); alone four waves are faster too (1008 against 1019 us per evaluation): profiles/r05_update_waves.txt
	int sweep_lds_pad = 2048; // "sweep_lds_pad": bytes of unused dynamic LDS on the pair sweep's launch when the side stream runs beside it: four
	                          // workgroups per CU instead of five (no loss: 133 -> 131 us) leave 30 KB of LDS for the reciprocal-space kernels, which otherwise
	                          // wait for sweep workgroups to retire (k_recip_sf_tab 68 -> 53 us beside the sweep; profiles/r05_one_evaluation_timeline.txt)
	int fused_update = 0; // "fused_update" = 1 (2: measurement only, arrivals without the update -- results invalid): the dipole update rides the panel launch (last-arriving workgroup per tile); 0 (default): its own launch per iteration -- measured in round 5, profiles/r05_fused_update.txt
	bool panel_reverse = true; // "panel_reverse": panel entries launched in descending j-tile order; 0: table order (rounds 2-4)
	bool use_panels = true; // "panels": panel form of the Jacobi contraction (orthorhombic cells, stored tensors); 0: one tile pair per workgroup
	bool no_uniform = false;   // "uniform_images" = 0: no tile-pair-wide periodic images
	bool no_classes = false;   // "tile_classes" = 0: every tile pair is "near" (nothing skipped, every tensor stored)
	bool single_launch = true; // "single_launch": small LJ-only boxes in one launch
	bool no_recip_tab = false; // "recip_table" = 0: one sincos per (k, atom) instead of the factorised phases
	bool no_sort = false;      // "spatial_sort" = 0: atoms stay in the caller's order
	bool no_order_carry = false; // "order_carry" = 0: every upload of the atom list sorts
	bool no_polar_delta = false; // "polar_delta" = 0: trial moves of polarizable boxes run a full evaluation
	bool no_inline_move = false; // "inline_move" = 0: trial moves always travel through the staging block
	int virtual_device = -1;     // "virtual_device" = v >= 0 (test hook): mpmc_pi_allreduce treats this context as living on a device of its own (csrc/comm.cpp group_beads)
	int fail_next_wait = 0;      // "fail_next_wait" = 1: the next wait of this context fails as if the runtime had refused it (test of the recovery path)
	bool trace_panel = false;    // "trace_panel" = 1: per-workgroup time stamps of the panel kernel (tools/panel_trace.py)
	long long tensor_budget_mb = 4096; // "tensor_budget_mb": AUTO solver: largest tensor store it will allocate
};

struct mpmc_ctx {
	mpmc_tuning tune;
	int device = 0;
	hipStream_t stream = nullptr;
	// second stream for work that is independent of the main chain inside ONE evaluation (reciprocal space next to the
	// pair sweep; the far-field Jacobi kernel next to the streaming one); always joined back before results are used
	hipStream_t stream2 = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	bool two_streams = true; // the side stream is forked in THIS evaluation (set per evaluation from stream_mode and the table size)
	int max_atoms = 0, max_pad = 0;
	int n = 0, n_pad = 0, n_tiles = 0, n_tile_pairs = 0, n_split = 1;
	int n_molecules = 0;
	double N_movable = 0; // countN
	std::string err;

	// host mirrors of the flattened System
	std::vector<double> h_pos, h_q, h_alpha, h_eps, h_sigma, h_mass;
	std::vector<int32_t> h_mol, h_frozen, h_disp;

	// spatial order: device slot k holds original atom perm[k]; slot_of[i] is the slot of original atom i.
	// Atoms are sorted (nested x / y / z bisection of the wrapped fractional coordinates) so that each tile of 64
	// consecutive slots is spatially compact; every result that leaves the library is returned in ORIGINAL order.
	std::vector<int32_t> perm, slot_of;
	int32_t *d_slot_of = nullptr, *d_perm = nullptr;
	bool atoms_dirty = true; // host mirror newer than the device arrays (full upload pending)

	// device atom arrays
	// the per-atom arrays below are pieces of ONE device block laid out like the pinned staging block of upload_atoms
	// ([xyzq][lj][mf][alpha][eps][inv_molmass][perm][slot_of], each max_pad long): an upload of the atoms is one copy
	char *d_atoms_blob = nullptr;
	double4 *d_xyzq = nullptr;
	double2 *d_lj = nullptr;
	int2 *d_mf = nullptr;
	double *d_alpha = nullptr, *d_eps = nullptr, *d_inv_molmass = nullptr;

	// pair kernel
	int2 *d_tile_pairs = nullptr;
	double *d_block_part = nullptr; // [ntp][2]
	int *d_block_cnt = nullptr;     // [ntp][4] (2 used by the pair kernel, 4 by the static-count kernel)
	int *d_cls = nullptr;           // tile-pair classes (CLS_*), recomputed every evaluation
	double *d_tile_bounds = nullptr; // [n_tiles][12]: wrapped fractional lo/hi, raw Cartesian lo/hi
	double4 *d_tp_shift = nullptr;   // [n_tile_pairs] lattice vector components of the common image index (CLS_UNIFORM_X/Y/Z)
	int4 *d_panels = nullptr;        // work table of the panel form of the Jacobi contraction (k_build_panels), rebuilt every evaluation
	int *d_seg = nullptr;            // [n_tiles + 1] first entry of every j-tile's segment of that table
	double *d_gpart = nullptr;       // [entries][3][64] j-side partial sums, one slot per entry of the table
	int *d_arrive = nullptr;         // [n_tiles] arrival counters of the fused dipole update (zero between launches)
	size_t cap_panels = 0, cap_seg = 0;
	long long *d_trace = nullptr;    // measurement only (tune.trace_panel): [entries][4] start / end ticks, HW_ID, XCC_ID of every workgroup of the LAST panel launch
	int n_panel_entries = 0, seg_tiles = -1; // entries of the table / the tile count its layout was made for
	bool panels_built = false;       // this evaluation's classes carry CLS_GROUPED bits and d_panels is valid
	// the fast pair sweep (kernels_pair.hip): its erfc table, its work table { J, I0 } (depends on the tile count only) and the list of
	// tile pairs it leaves to k_pair_fused (a tile with a kAtomFlagsMixing atom -- sigma < 0 or dispersion coefficients: rebuilt with every upload)
	double2 *d_erf_tab = nullptr;
	int2 *d_sweep_blocks = nullptr;
	size_t cap_sweep_blocks = 0;
	int n_sweep_blocks = 0, sweep_tiles = -1;
	int *d_generic_list = nullptr;
	size_t cap_generic = 0;
	int n_generic = 0;
	std::vector<int> h_generic;
	int inflight_hint = 1; // evaluations the caller keeps in flight together with this one (mpmc_hint_in_flight; the PI loops set their bead count)
	bool last_pair_was_sweep = false; // (diagnostics: which kernel the last evaluation's pair pass ran)
	FusedParams last_fp{};            // the pair pass's parameters in the last evaluation (mpmc_debug_time_pair replays it)
	bool last_fp_valid = false;
	int debug_panel_replicas = 1;     // mpmc_debug_configure "panel_replicas": grid repetitions of mpmc_debug_time_panel's launches
	double4 *h_xyzq = nullptr;       // PINNED host mirror of d_xyzq (slot order, max_pad entries): position updates copy from it asynchronously;
	hipEvent_t ev_xyzq = nullptr;    // marks the last copy out of it done -- whoever is about to write the mirror waits for that copy only
	bool xyzq_in_flight = false;     // (mirror_guard), not for the evaluations queued behind it
	std::vector<double> h_pos_sorted; // positions at the time of the last spatial sort
	// The spatial order is a locality heuristic: ANY permutation gives the same physics (sums in another order).  A contiguous insertion
	// or removal (uVT / Gibbs) therefore carries the order it finds -- the survivors keep their sequence, inserted atoms are appended --
	// instead of paying an O(N log N) host sort per move; a real sort follows once kTile atoms have come or gone since the last one.
	bool order_sorted = false;  // perm is a spatial sort of the current atom list (not the identity of small / Gauss-Seidel systems)
	bool order_carried = false; // set_atoms has already brought perm / slot_of up to date: upload_atoms does not sort
	bool atoms_dirty_order = false; // a NEW sort was asked for (cell, options, drift) and is pending: nothing is carried across it
	long long n_uploads_carried = 0, n_uploads_sorted = 0; // (diagnostics: mpmc_debug_upload_counts)
	int edits_since_sort = 0;   // atoms inserted + removed since the last real sort
	double sort_origin_f[3] = {0, 0, 0}; // fractional coordinate at which the spatial sort cuts the periodic wrap
	hipStream_t sync_stream = nullptr;  // stream that carries this context's final copies (null: its own)
	size_t cap_tile_pairs = 0;
	std::vector<double> molmass_tmp; // (scratch of upload_atoms)
	long long *static_cnt = nullptr; // pinned [4]: n_intra, n_rd_excluded, n_es_excluded, n_frozen (position independent; copied back behind every upload of the atoms)
	// upload_atoms stages every per-atom array in ONE persistent pinned block (eight truly asynchronous copies, no synchronisation);
	// ev_stage marks the copies done, the next upload waits for it before it refills the block
	char *h_kstage = nullptr; // the same for the k-vector tables of build_k_tables ([kvec][kw][lvec][w_en], each cap_kstage long: 16-byte types first)
	size_t cap_kstage = 0;
	hipEvent_t ev_kstage = nullptr;
	bool kstage_in_flight = false;
	int lvec_kmax = -1;       // the integer l-vectors on the device belong to this kmax (they depend on nothing else)
	char *h_stage = nullptr;
	hipEvent_t ev_stage = nullptr;
	bool stage_in_flight = false;
	// the position-independent terms (LRC, Ewald self) ride along with the next general evaluation when they are stale: static_gen
	// counts the events that make them stale, static_ride_gen is the count the pending evaluation's ride-along belongs to (0: none)
	unsigned static_gen = 1, static_ride_gen = 0;

	// position-independent scalars (pair LRC, self LRC, Ewald self term): functions of the atom parameters, the cell and the options
	// only -- computed once (k_atom_terms) whenever one of those changed, kept on the host, added when a result is assembled
	bool static_dirty = true;
	double h_static[3] = {0, 0, 0}; // lrc_pair, lrc_self, es_self
	int *d_counter = nullptr;       // ticket counter of the single-launch small-system kernels (zero between launches)
	bool scal_clean = false;        // d_scal is all zeros (the post kernel of the last evaluation left it so): no clear needed in front of this one
	int poll_budget_us = 1000;      // ... for at most this long before the wait falls back to hipStreamSynchronize
	bool spin_on_post = false;      // the pending evaluation ends in k_post_results and is short: wait_and_fill polls the launch number first
	bool last_was_single = false;   // the pending evaluation wrote h_scal from the device: nothing to copy back
	double single_seq = 0;          // launch number the single-launch kernel posts behind its results (host polls h_scal[S_COUNT + C_COUNT])
	// scalars
	double *d_atom_part = nullptr;   // scratch of launch_atom_terms (per-block partial sums)
	double *d_scal = nullptr;
	long long *d_cnt = nullptr;
	double *h_scal = nullptr; // pinned
	long long *h_cnt = nullptr;
	int *d_flag = nullptr;
	int *h_flag = nullptr; // pinned

	// reciprocal tables
	int K = 0, cap_K = 0; // cap_K: capacity of the k tables (d_kvec, d_kw, d_lvec, d_w_en)
	int cap_sf = 0;       // capacity of d_sf, which trades places with d_sf_trial when a trial move is accepted
	double4 *d_kvec = nullptr, *d_kw = nullptr, *d_sf = nullptr;
	int4 *d_lvec = nullptr;       // integer l-vectors of the k table
	double4 *d_sf_part = nullptr; // [n_tiles][K] per-tile structure-factor partials (factorised phases)
	size_t cap_sf_part = 0;
	double *d_w_en = nullptr;

	// polarization work
	double *d_e_recip_part = nullptr, *d_part = nullptr, *d_e_static = nullptr, *d_mu[2] = {nullptr, nullptr}, *d_e_induced = nullptr,
	       *d_rrms = nullptr;
	size_t cap_part = 0;
	double2 *d_gs_blocks = nullptr; // Gauss-Seidel sweeps: the in-tile 3 x 3 blocks (k_gs_blocks), cap_gs_blocks double2 elements
	size_t cap_gs_blocks = 0;
	double *d_gs_ul = nullptr; // Gauss-Seidel sweeps (kernels_gs.hip): [2][max_pad][3] induced-field parts from the tiles above / below
	int mu_cur = 0;
	// dense A rows scratch
	double *d_arows = nullptr;
	double *d_adense = nullptr; // solver DENSE: the (3 n_pad)^2 matrix of thole_amatrix without its diagonal blocks
	size_t cap_adense = 0;
	size_t cap_arows = 0;
	// compact Thole tensor store: (a,b) per unordered pair, tile-pair major, 64*64 double2 per tile pair
	double2 *d_ab = nullptr;
	size_t cap_ab = 0; // in double2 elements
	int solver_used = MPMC_SOLVER_MATRIX_FREE;

	Box box{};
	double box_in[20] = {0}; // what mpmc_set_box was last called with (basis, reciprocal, volume, cutoff): an identical call is a no-op
	bool box_in_has_recip = false;
	bool box_set = false, atoms_set = false, opts_set = false, k_dirty = true;
	mpmc_options opts{};
	double ewald_alpha = 0, polar_ewald_alpha = 0;

	// results of the last evaluation
	bool pending = false;
	bool have_polar = false;
	int iters = 0, failed = 0;
	unsigned run_mask = 0;

	// trial moves (delta energies)
	bool cache_valid = false;   // last_full = totals of the accepted configuration, d_sf = its structure factors
	mpmc_result last_full{};
	mpmc_result trial_res{};
	int trial_last_kind = -1; // (diagnostics) the last enqueued trial: 1 full evaluation, 0 delta energies
	bool trial_open = false, trial_evaluated = false, trial_was_full = false, trial_enqueued = false, trial_noop = false;
	mpmc_result trial_keep{}; // accepted totals while a full-evaluation trial is in flight
	int trial_first = 0, trial_count = 0;
	std::vector<double> trial_new, trial_old;
	int *d_mv_slot = nullptr, *d_mv_orig = nullptr, *d_moved_idx = nullptr; // d_mv_slot/d_mv_orig/d_mv_new live in ONE allocation (d_mv_blob)
	double4 *d_mv_new = nullptr, *d_sf_trial = nullptr;
	// polarizable boxes: the real-space static field of the accepted configuration (k_field_finalize) and of the trial one
	// (e_real + delta of the pairs with a moved atom); they trade places on accept.  dk_part: scratch of k_delta_field.
	double *d_e_real = nullptr, *d_e_real_trial = nullptr, *d_dk_part = nullptr;
	size_t cap_dk_part = 0;
	bool e_real_valid = false;     // d_e_real describes the accepted configuration
	bool trial_polar_delta = false; // the open trial took the incremental polarizable path (positions swapped on the device)
	// the tensor store between trial moves: a trial rebuilds only the tile pairs of the tiles its moved atoms live in; after a REJECTED
	// trial those tiles hold the rejected geometry's tensors and are rebuilt by the next trial (or by any full evaluation)
	std::vector<int> store_dirty_tiles, trial_tiles;
	int touch_n = -1, touch[8] = {0}; // what enqueue(RUN_STORE) passes to the store-only sweep (-1: all tile pairs)
	unsigned char *d_mv_blob = nullptr, *h_mv_blob = nullptr; // device / pinned host staging of a trial's moved-atom list
	int cap_sf_trial = 0;
	double *d_delta_out = nullptr, *h_delta_out = nullptr; // h: pinned [9] = 5 doubles, 2 int64 counts, spare, launch number (k_delta_finish posts it)
	MvInline mv_inline{};          // the pending trial's move when it travelled in the kernel arguments (trial_inline)
	bool trial_inline = false;
	double trial_seq = 0;          // launch number of the pending trial's k_delta_finish
	long long *d_delta_cnt = nullptr, *h_delta_cnt = nullptr;

	// how the host waits ended (mpmc_debug_wait_counters): polls that saw the device's post, polls that ran out of their budget (the
	// wait then fell back to a stream synchronisation), plain stream synchronisations, and yields taken inside long polls
	long long n_poll_hits = 0, n_poll_timeouts = 0, n_stream_syncs = 0, n_poll_yields = 0;

	// profiling
	bool prof = false;
	std::vector<EvPair> ev_free, ev_used;
	mpmc_timings tim{};

	int64_t bytes_total = 0;
};

// ---------------------------------------------------------------------------------------------------------
#define HIP_TRY(ctx, call)                                                                                        \
	do {                                                                                                          \
		hipError_t _e = (call);                                                                                   \
		if (_e != hipSuccess) {                                                                                   \
			(ctx)->err = std::string(#call) + ": " + hipGetErrorString(_e);                                       \
			return MPMC_ERR_HIP;                                                                                  \
		}                                                                                                         \
	} while (0)

namespace mpmc { // internal helpers: mangled names, nothing here can collide with a symbol of the host program

constexpr size_t kAtomRecordBytes = sizeof(double4) + sizeof(double2) + sizeof(int2) + 3 * sizeof(double) + 2 * sizeof(int32_t); // per atom, all arrays of the block
template <typename T>
inline int dev_alloc(mpmc_ctx *c, T **p, size_t count);
// carve the per-atom arrays out of a block of P records (device block and pinned staging block share the layout)
template <typename F>
inline void atom_block_layout(char *base, size_t P, F &&set) {
	double4 *xyzq = reinterpret_cast<double4 *>(base);
	double2 *lj = reinterpret_cast<double2 *>(xyzq + P);
	int2 *mf = reinterpret_cast<int2 *>(lj + P);
	double *al = reinterpret_cast<double *>(mf + P), *ep = al + P, *imm = ep + P;
	int32_t *perm = reinterpret_cast<int32_t *>(imm + P), *slot = perm + P;
	set(xyzq, lj, mf, al, ep, imm, perm, slot);
}
template <typename T>
inline int dev_alloc(mpmc_ctx *c, T **p, size_t count) {
	const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
	HIP_TRY(c, hipMalloc((void **)p, bytes));
	// Every buffer starts from zeros: what a kernel finds in a slot it has not written yet must not depend on what an earlier process
	// left in that memory.  The fill is WAITED for -- buffers are also allocated in the middle of an evaluation, after the side stream
	// was forked, and the first writer may be a side-stream kernel that is not ordered behind a fill on the main stream (seen: structure
	// factors zeroed under the reciprocal-space kernels).  Allocations happen once per context, the wait costs nothing in steady state.
	HIP_TRY(c, hipMemsetAsync(*p, 0, bytes, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	c->bytes_total += (int64_t)(count * sizeof(T));
	return MPMC_OK;
}
// before the slot-ordered mirror is written: the last asynchronous copy out of it must have read it
inline int mirror_guard(mpmc_ctx *c) {
	if (c->xyzq_in_flight) {
		HIP_TRY(c, hipEventSynchronize(c->ev_xyzq));
		c->xyzq_in_flight = false;
	}
	return MPMC_OK;
}
template <typename T>
inline void dev_free(mpmc_ctx *c, T **p, size_t count) {
	if (*p) {
		(void)hipFree(*p);
		c->bytes_total -= (int64_t)(count * sizeof(T));
		*p = nullptr;
	}
}

inline int fail(mpmc_ctx *c, int code, const std::string &msg) {
	if (c) c->err = msg;
	else g_create_error = msg;
	return code;
}

// Poll a pinned word the device posts behind its results (system-scope release on the device side) until `seen()` or until `budget` has
// passed; true = seen.  A short evaluation ends a few microseconds earlier this way than through the driver's completion path.  Past
// 50 us the poller yields between checks: on a host with fewer free cores than polling / OpenMP threads (one run of fourteen in round 2
// took 3.5 ms per Monte Carlo step instead of 0.13 -- the signature of spinning threads time-slicing on too few cores) a spinning thread
// must not keep the thread that would feed the device off the CPU.  The counters tell afterwards which way the waits went.
template <class Pred>
inline bool poll_posted(mpmc_ctx *c, Pred seen, std::chrono::microseconds budget) {
	const auto t0 = std::chrono::steady_clock::now();
	for (int spins = 0;; ++spins) {
		if (seen()) {
			std::atomic_thread_fence(std::memory_order_acquire);
			c->n_poll_hits++;
			return true;
		}
		if ((spins & 255) == 255) {
			const auto dt = std::chrono::steady_clock::now() - t0;
			if (dt > budget) {
				c->n_poll_timeouts++;
				return false;
			}
			if (dt > std::chrono::microseconds(50)) {
				c->n_poll_yields++;
				std::this_thread::yield();
			}
		}
	}
}

// ---- profiling ------------------------------------------------------------------------------------------
inline void prof_begin(mpmc_ctx *c, int cls, int &cur, hipStream_t st) {
	cur = -1;
	if (!c->prof) return;
	EvPair e;
	if (!c->ev_free.empty()) {
		e = c->ev_free.back();
		c->ev_free.pop_back();
	} else {
		if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
	}
	e.cls = cls;
	(void)hipEventRecord(e.a, st);
	c->ev_used.push_back(e);
	cur = (int)c->ev_used.size() - 1;
}
inline void prof_end(mpmc_ctx *c, int cur, hipStream_t st) {
	if (cur >= 0 && cur < (int)c->ev_used.size()) (void)hipEventRecord(c->ev_used[cur].b, st);
}
inline void prof_harvest(mpmc_ctx *c) { // stream must be idle
	for (auto &e : c->ev_used) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
			c->tim.ms[e.cls] += ms;
			c->tim.launches[e.cls] += 1;
		}
		c->ev_free.push_back(e);
	}
	c->ev_used.clear();
}
struct ProfScope { // HIP-event bracket on the stream the kernels are launched on
	mpmc_ctx *c;
	int cur;
	hipStream_t st;
	ProfScope(mpmc_ctx *c_, int cls, hipStream_t st_ = nullptr) : c(c_), st(st_ ? st_ : c_->stream) { prof_begin(c, cls, cur, st); }
	~ProfScope() { prof_end(c, cur, st); }
};
// side stream: starts after everything enqueued so far on the main stream / main stream waits for the side stream
inline hipStream_t fork_side(mpmc_ctx *c) {
	if (!c->two_streams) return c->stream;
	if (!c->stream2 && hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) { // (lazy_side_stream)
		c->stream2 = nullptr;
		c->two_streams = false;
		return c->stream;
	}
	(void)hipEventRecord(c->ev_fork, c->stream);
	(void)hipStreamWaitEvent(c->stream2, c->ev_fork, 0);
	return c->stream2;
}
inline void join_side(mpmc_ctx *c) {
	if (!c->two_streams || !c->stream2) return;
	(void)hipEventRecord(c->ev_join, c->stream2);
	(void)hipStreamWaitEvent(c->stream, c->ev_join, 0);
}


// ---- shared between the translation units ---------------------------------------------------------------------
enum : unsigned {
	RUN_PAIR = 1, RUN_PAIR_ES = 2, RUN_RECIP = 4, RUN_ATOMTERMS = 8, RUN_FIELD = 16, RUN_SOLVE = 32, RUN_WOLF = 64,
	RUN_STORE = 128 // tile classes + the Thole tensor store alone (no energies, no field): trial moves of polarizable boxes
};
int prepare(mpmc_ctx *c, bool defer_static = false); // uploads what is dirty, (re)builds the k tables; the position-independent terms unless deferred (evaluate.cpp)
int enqueue(mpmc_ctx *c, unsigned mask);         // one evaluation (the pieces in `mask`) on the context's streams (evaluate.cpp)
int wait_and_fill(mpmc_ctx *c, mpmc_result *out); // waits for it and assembles the result (evaluate.cpp)
unsigned full_mask(const mpmc_ctx *c);           // what double System::energy() runs under the current options
void ext_params(const mpmc_ctx *c, FusedParams &fp, bool wolf_on); // Wolf / Feynman-Hibbs fields of the pair parameters (evaluate.cpp)
AtomsDev atoms_view(const mpmc_ctx *c);
RecipDev recip_view(const mpmc_ctx *c);
int upload_atoms(mpmc_ctx *c);                   // spatial order + device atom arrays (context.cpp)

} // namespace mpmc
