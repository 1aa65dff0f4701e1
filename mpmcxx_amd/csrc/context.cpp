// context.cpp -- host side of libmpmc_energy.so: the C ABI of include/mpmc_energy.h.
//
// One mpmc_ctx = the device-resident state of one reference `System` (one box / one PI bead):
// its own HIP stream, struct-of-arrays atom buffers, k-vector tables, work buffers and result scalars.
// Replaces, for the energy path only, the per-System pair lists (reference src/System.Pairs.cpp:21) and the
// A matrix (src/System.cpp:1430-1473).  There is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mpmc_energy.h"
#include "kernels.h"

using namespace mpmc;

static thread_local std::string g_create_error;

struct EvPair {
	hipEvent_t a, b;
	int cls;
};

struct mpmc_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	// second stream for work that is independent of the main chain inside ONE evaluation (reciprocal space next to the
	// pair sweep; the far-field Jacobi kernel next to the streaming one); always joined back before results are used
	hipStream_t stream2 = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	bool two_streams = true; // MPMC_ONE_STREAM=1 disables the fork/join
	int jacc = 0; // hybrid Jacobi kernel variant (MPMC_JACC): 0 DPP lane rotation, 1 ds_bpermute (when the DPP self-test fails)
	bool jacobi_hybrid = true; // one launch per Jacobi iteration over all tile pairs; MPMC_JACOBI=split: two kernels (stream / far)
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
	double4 *d_xyzq = nullptr;
	double2 *d_lj = nullptr;
	int2 *d_mf = nullptr;
	double *d_alpha = nullptr, *d_eps = nullptr, *d_inv_molmass = nullptr;

	// pair kernel
	int2 *d_tile_pairs = nullptr;
	double *d_block_part = nullptr; // [ntp][2]
	int *d_block_cnt = nullptr;     // [ntp][4] (2 used by the pair kernel, 4 by the static-count kernel)
	int *d_cls = nullptr;           // tile-pair classes (CLS_*), recomputed every evaluation
	int *d_lists = nullptr;         // [2 ntp] work lists of the two Jacobi kernels + [2] their lengths (at the end)
	double *d_tile_bounds = nullptr; // [n_tiles][12]: wrapped fractional lo/hi, raw Cartesian lo/hi
	double4 *d_tp_shift = nullptr;   // [n_tile_pairs] lattice vector components of the common image index (CLS_UNIFORM_X/Y/Z)
	std::vector<double4> h_xyzq;     // host mirror of d_xyzq (slot order), for bulk position updates
	std::vector<double> h_pos_sorted; // positions at the time of the last spatial sort
	double sort_origin_f[3] = {0, 0, 0}; // fractional coordinate at which the spatial sort cuts the periodic wrap
	bool no_uniform = false;         // MPMC_NO_UNI=1
	// lockstep solve of several systems (mpmc_pi_potential_local): enqueue() stops before the dipole iterations when asked to and
	// possible; the batch driver then runs the iterations of all deferred systems in shared launches on one stream
	bool defer_solve = false, solve_deferred = false, reduce_pending_join = false;
	hipEvent_t ev_phase = nullptr;      // "everything before the solve is enqueued" marker on this context's stream
	hipStream_t sync_stream = nullptr;  // stream that carries this context's final copies (null: its own)
	SolveBead *d_solve_args = nullptr;  // device array of per-system pointers (owned by the first system of a batch)
	std::vector<SolveBead> h_solve_args; // its host image (must outlive the asynchronous copy)
	int cap_solve_args = 0;
	int last_batch = 1;                 // systems per launch in the last evaluation's solve
	size_t cap_tile_pairs = 0;
	long long static_cnt[4] = {0, 0, 0, 0}; // n_intra, n_rd_excluded, n_es_excluded, n_frozen (position independent)

	// scalars
	double *d_scal = nullptr;
	long long *d_cnt = nullptr;
	double *h_scal = nullptr; // pinned
	long long *h_cnt = nullptr;
	int *d_flag = nullptr;
	int *h_flag = nullptr; // pinned

	// reciprocal tables
	int K = 0, cap_K = 0;
	double4 *d_kvec = nullptr, *d_kw = nullptr, *d_sf = nullptr;
	int4 *d_lvec = nullptr;       // integer l-vectors of the k table
	double4 *d_sf_part = nullptr; // [n_tiles][K] per-tile structure-factor partials (factorised phases)
	size_t cap_sf_part = 0;
	bool no_recip_tab = false;    // MPMC_NO_RECIP_TAB=1: one sincos per (k, atom)
	double *d_w_en = nullptr;

	// polarization work
	double *d_e_recip_part = nullptr, *d_part = nullptr, *d_e_static = nullptr, *d_mu[2] = {nullptr, nullptr}, *d_e_induced = nullptr,
	       *d_rrms = nullptr;
	size_t cap_part = 0;
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
	bool use_dpp = true;   // lane rotation by v_mov_b32_dpp wave_rol:1 (verified at create), else ds_bpermute
	bool no_classes = false; // MPMC_NO_CLASSES=1: treat every tile pair as near (A/B comparisons only)

	Box box{};
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
	bool trial_open = false, trial_evaluated = false, trial_was_full = false, trial_enqueued = false, trial_noop = false;
	mpmc_result trial_keep{}; // accepted totals while a full-evaluation trial is in flight
	int trial_first = 0, trial_count = 0;
	std::vector<double> trial_new, trial_old;
	int *d_mv_slot = nullptr, *d_mv_orig = nullptr, *d_moved_idx = nullptr; // d_mv_slot/d_mv_orig/d_mv_new live in ONE allocation (d_mv_blob)
	double4 *d_mv_new = nullptr, *d_sf_trial = nullptr;
	unsigned char *d_mv_blob = nullptr, *h_mv_blob = nullptr; // device / pinned host staging of a trial's moved-atom list
	int cap_sf_trial = 0;
	double *d_delta_out = nullptr, *h_delta_out = nullptr;
	long long *d_delta_cnt = nullptr, *h_delta_cnt = nullptr;

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

template <typename T>
static int dev_alloc(mpmc_ctx *c, T **p, size_t count) {
	HIP_TRY(c, hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
	c->bytes_total += (int64_t)(count * sizeof(T));
	return MPMC_OK;
}
template <typename T>
static void dev_free(mpmc_ctx *c, T **p, size_t count) {
	if (*p) {
		(void)hipFree(*p);
		c->bytes_total -= (int64_t)(count * sizeof(T));
		*p = nullptr;
	}
}

static int fail(mpmc_ctx *c, int code, const std::string &msg) {
	if (c) c->err = msg;
	else g_create_error = msg;
	return code;
}

// ---- profiling ------------------------------------------------------------------------------------------
static void prof_begin(mpmc_ctx *c, int cls, int &cur, hipStream_t st) {
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
static void prof_end(mpmc_ctx *c, int cur, hipStream_t st) {
	if (cur >= 0 && cur < (int)c->ev_used.size()) (void)hipEventRecord(c->ev_used[cur].b, st);
}
static void prof_harvest(mpmc_ctx *c) { // stream must be idle
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
static hipStream_t fork_side(mpmc_ctx *c) {
	if (!c->two_streams) return c->stream;
	(void)hipEventRecord(c->ev_fork, c->stream);
	(void)hipStreamWaitEvent(c->stream2, c->ev_fork, 0);
	return c->stream2;
}
static void join_side(mpmc_ctx *c) {
	if (!c->two_streams) return;
	(void)hipEventRecord(c->ev_join, c->stream2);
	(void)hipStreamWaitEvent(c->stream, c->ev_join, 0);
}

// ---- library --------------------------------------------------------------------------------------------
extern "C" int mpmc_abi_version(void) { return MPMC_ABI_VERSION; }

extern "C" int mpmc_device_count(int *count) {
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (count) *count = (e == hipSuccess) ? n : 0;
	return (e == hipSuccess) ? MPMC_OK : MPMC_ERR_NO_DEVICE;
}

extern "C" const char *mpmc_last_error(const mpmc_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// PeriodicBoundary::update, reference src/PeriodicBoundary.cpp:31-101 (same association order)
extern "C" int mpmc_pbc_compute(const double b[9], double R[9], double *volume, double *cutoff) {
	if (!b || !R || !volume || !cutoff) return MPMC_ERR_ARG;
#define B(i, j) b[3 * (i) + (j)]
	double vol;
	vol = B(0, 0) * (B(1, 1) * B(2, 2) - B(1, 2) * B(2, 1));
	vol += B(0, 1) * (B(1, 2) * B(2, 0) - B(1, 0) * B(2, 2));
	vol += B(0, 2) * (B(1, 0) * B(2, 1) - B(1, 1) * B(2, 0));
	*volume = vol;
	if (vol <= 0) {
		*cutoff = kMaxValue;
	} else {
		double shortest = kMaxValue;
		for (int i = -15; i <= 15; i++)
			for (int j = -15; j <= 15; j++)
				for (int k = -15; k <= 15; k++) {
					if (!i && !j && !k) continue;
					double v[3];
					for (int p = 0; p < 3; p++) v[p] = i * B(0, p) + j * B(1, p) + k * B(2, p);
					double m = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
					if (m < shortest) shortest = m;
				}
		*cutoff = 0.5 * shortest;
	}
	const double iv = 1.0 / vol;
	R[0] = iv * (B(1, 1) * B(2, 2) - B(1, 2) * B(2, 1));
	R[1] = iv * (B(0, 2) * B(2, 1) - B(0, 1) * B(2, 2));
	R[2] = iv * (B(0, 1) * B(1, 2) - B(0, 2) * B(1, 1));
	R[3] = iv * (B(1, 2) * B(2, 0) - B(1, 0) * B(2, 2));
	R[4] = iv * (B(0, 0) * B(2, 2) - B(0, 2) * B(2, 0));
	R[5] = iv * (B(0, 2) * B(1, 0) - B(0, 0) * B(1, 2));
	R[6] = iv * (B(1, 0) * B(2, 1) - B(1, 1) * B(2, 0));
	R[7] = iv * (B(0, 1) * B(2, 0) - B(0, 0) * B(2, 1));
	R[8] = iv * (B(0, 0) * B(1, 1) - B(0, 1) * B(1, 0));
#undef B
	return (vol > 0) ? MPMC_OK : MPMC_ERR_BOX;
}

extern "C" void mpmc_default_options(mpmc_options *o) {
	if (!o) return;
	std::memset(o, 0, sizeof(*o));
	o->rd_lrc = 1;          // reference src/System.h: rd_lrc default on
	o->polar_max_iter = 10; // polar_max_iter default
	o->ewald_kmax = 7;      // ewald_kmax default
	o->polar_gamma = 1.0;
	o->damp_type = MPMC_DAMPING_EXPONENTIAL;
	o->solver = MPMC_SOLVER_AUTO;
}

// largest double t >= 0 with pred(t) true, for a predicate that is true below and false above some point near rc^2
template <typename Pred>
static double bisect_threshold(double rc, Pred pred) {
	double lo = rc * rc * (1.0 - 1e-6), hi = rc * rc * (1.0 + 1e-6) + 1e-300;
	if (!pred(lo) || pred(hi)) return pred(hi) ? hi : -1.0; // degenerate box; callers validated rc > 0
	uint64_t a, b;
	std::memcpy(&a, &lo, 8);
	std::memcpy(&b, &hi, 8);
	while (b - a > 1) { // positive doubles order like their bit patterns
		uint64_t m = a + (b - a) / 2;
		double x;
		std::memcpy(&x, &m, 8);
		if (pred(x)) a = m;
		else b = m;
	}
	double out;
	std::memcpy(&out, &a, 8);
	return out;
}

// per-device result of the lane-rotation self-test (0 unknown, 1 dpp ok, 2 dpp wrong -> ds_bpermute)
static int g_rot_mode[64] = {0};
static int rot_selftest(mpmc_ctx *c) {
	if (c->device < 64 && g_rot_mode[c->device]) {
		c->use_dpp = (g_rot_mode[c->device] == 1);
		return MPMC_OK;
	}
	int *d = nullptr, h[128];
	HIP_TRY(c, hipMalloc((void **)&d, 128 * sizeof(int)));
	launch_rot_selftest(c->stream, d, d + 64);
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	(void)hipFree(d);
	bool dpp_ok = true, perm_ok = true;
	for (int l = 0; l < 64; l++) {
		if (h[l] != ((l + 1) & 63)) dpp_ok = false;
		if (h[64 + l] != ((l + 1) & 63)) perm_ok = false;
	}
	if (!perm_ok) {
		c->err = "lane-rotation self-test failed (ds_bpermute)";
		return MPMC_ERR_INTERNAL;
	}
	c->use_dpp = dpp_ok;
	if (c->device < 64) g_rot_mode[c->device] = dpp_ok ? 1 : 2;
	return MPMC_OK;
}

// ---- lifetime --------------------------------------------------------------------------------------------
extern "C" int mpmc_ctx_create(int device, int max_atoms, mpmc_ctx **out) {
	if (!out || max_atoms <= 0) return fail(nullptr, MPMC_ERR_ARG, "mpmc_ctx_create: bad argument");
	*out = nullptr;
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
		return fail(nullptr, MPMC_ERR_NO_DEVICE,
		            std::string("mpmc_ctx_create: no HIP device (") + hipGetErrorString(e) + "); this library has no CPU path");
	if (device < 0 || device >= ndev) return fail(nullptr, MPMC_ERR_ARG, "mpmc_ctx_create: device index out of range");
	if (hipSetDevice(device) != hipSuccess) return fail(nullptr, MPMC_ERR_NO_DEVICE, "mpmc_ctx_create: hipSetDevice failed");

	mpmc_ctx *c = new mpmc_ctx();
	c->device = device;
	c->max_atoms = max_atoms;
	c->max_pad = ((max_atoms + kTile - 1) / kTile) * kTile;
	mpmc_default_options(&c->opts);
	int rc = MPMC_OK;
	auto A = [&](int r) { if (rc == MPMC_OK) rc = r; };
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
	    hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
	    hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
		delete c;
		return fail(nullptr, MPMC_ERR_HIP, "mpmc_ctx_create: hipStreamCreate failed");
	}
	if (const char *e = std::getenv("MPMC_ONE_STREAM")) c->two_streams = !(e[0] == '1');
	if (const char *e = std::getenv("MPMC_JACOBI")) c->jacobi_hybrid = (e[0] != 's');
	c->jacc = c->use_dpp ? 0 : 1;
	if (const char *e = std::getenv("MPMC_JACC")) c->jacc = std::atoi(e);
	if (const char *e = std::getenv("MPMC_NO_UNI")) c->no_uniform = (e[0] == '1');
	if (const char *e = std::getenv("MPMC_NO_RECIP_TAB")) c->no_recip_tab = (e[0] == '1');
	const size_t P = (size_t)c->max_pad;
	A(dev_alloc(c, &c->d_xyzq, P));
	A(dev_alloc(c, &c->d_lj, P));
	A(dev_alloc(c, &c->d_mf, P));
	A(dev_alloc(c, &c->d_alpha, P));
	A(dev_alloc(c, &c->d_eps, P));
	A(dev_alloc(c, &c->d_inv_molmass, P));
	A(dev_alloc(c, &c->d_tile_bounds, 12 * (P / kTile)));
	A(dev_alloc(c, &c->d_slot_of, P));
	A(dev_alloc(c, &c->d_perm, P));
	A(dev_alloc(c, &c->d_scal, (size_t)S_COUNT + (size_t)C_COUNT)); // scalars and counts share one buffer: one clear, one read-back
	if (rc == MPMC_OK) c->d_cnt = reinterpret_cast<long long *>(c->d_scal + S_COUNT);
	A(dev_alloc(c, &c->d_flag, (size_t)1));
	static_assert(sizeof(long long) == sizeof(double), "scalars and counts share one buffer");
	if (rc == MPMC_OK && hipHostMalloc((void **)&c->h_scal, (S_COUNT + C_COUNT) * sizeof(double)) != hipSuccess) rc = MPMC_ERR_HIP;
	if (rc == MPMC_OK) c->h_cnt = reinterpret_cast<long long *>(c->h_scal + S_COUNT);
	if (rc == MPMC_OK && hipHostMalloc((void **)&c->h_flag, sizeof(int)) != hipSuccess) rc = MPMC_ERR_HIP;
	if (rc == MPMC_OK) rc = rot_selftest(c);
	if (rc != MPMC_OK) {
		g_create_error = "mpmc_ctx_create: device allocation failed: " + c->err;
		mpmc_ctx_destroy(c);
		return rc;
	}
	if (const char *e = std::getenv("MPMC_NO_CLASSES")) c->no_classes = (e[0] == '1');
	if (const char *e = std::getenv("MPMC_NO_DPP")) if (e[0] == '1') c->use_dpp = false;
	*out = c;
	return MPMC_OK;
}

extern "C" int mpmc_ctx_destroy(mpmc_ctx *c) {
	if (!c) return MPMC_ERR_ARG;
	(void)hipSetDevice(c->device);
	if (c->stream2) (void)hipStreamSynchronize(c->stream2);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->ev_phase) (void)hipEventDestroy(c->ev_phase);
	if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
	if (c->ev_join) (void)hipEventDestroy(c->ev_join);
	if (c->stream2) (void)hipStreamDestroy(c->stream2);
	for (auto &e : c->ev_used) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
	for (auto &e : c->ev_free) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
	void *ptrs[] = {c->d_xyzq, c->d_lj, c->d_mf, c->d_alpha, c->d_eps, c->d_inv_molmass, c->d_tile_pairs, c->d_block_part, c->d_block_cnt, c->d_scal,
	                c->d_flag, c->d_kvec, c->d_kw, c->d_sf, c->d_w_en, c->d_e_recip_part, c->d_part, c->d_e_static, c->d_mu[0], c->d_mu[1],
	                c->d_e_induced, c->d_rrms, c->d_arows, c->d_adense, c->d_ab, c->d_slot_of, c->d_perm, c->d_cls, c->d_tp_shift, c->d_lvec, c->d_sf_part, c->d_solve_args, c->d_tile_bounds, c->d_lists, c->d_mv_blob, c->d_moved_idx,
	                c->d_sf_trial, c->d_delta_out};
	for (void *p : ptrs)
		if (p) (void)hipFree(p);
	if (c->h_scal) (void)hipHostFree(c->h_scal);
	if (c->h_flag) (void)hipHostFree(c->h_flag);
	if (c->h_delta_out) (void)hipHostFree(c->h_delta_out);
	if (c->h_mv_blob) (void)hipHostFree(c->h_mv_blob);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
	return MPMC_OK;
}

// ---- box / options ---------------------------------------------------------------------------------------
extern "C" int mpmc_set_box(mpmc_ctx *c, const double basis[9], const double *reciprocal, double volume, double cutoff) {
	if (!c || !basis) return MPMC_ERR_ARG;
	double R[9], vol = 0, cut = 0;
	int rc = mpmc_pbc_compute(basis, R, &vol, &cut);
	if (rc != MPMC_OK) return fail(c, MPMC_ERR_BOX, "mpmc_set_box: non-positive cell volume");
	if (reciprocal) std::memcpy(R, reciprocal, sizeof(R));
	if (volume > 0) vol = volume;
	if (cutoff > 0) cut = cutoff;
	if (!(vol > 0) || !(cut > 0)) return fail(c, MPMC_ERR_BOX, "mpmc_set_box: invalid volume / cutoff");
	std::memcpy(c->box.b, basis, sizeof(R));
	std::memcpy(c->box.r, R, sizeof(R));
	c->box.volume = vol;
	c->box.cutoff = cut;
	// squared-distance forms of the reference's cutoff predicates (see pair_math.h Box)
	c->box.t_lj = bisect_threshold(cut, [cut](double t) { return std::sqrt(t) - kSmallDR < cut; });
	c->box.t_es = bisect_threshold(cut, [cut](double t) { return !(std::sqrt(t) > cut); });
	c->box.t_wolf = bisect_threshold(cut, [cut](double t) { return std::sqrt(t) < cut; });
	c->box.ortho = 1;
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++)
			if (i != j && (basis[3 * i + j] != 0.0 || R[3 * i + j] != 0.0)) c->box.ortho = 0;
	c->box_set = true;
	c->k_dirty = true;
	c->atoms_dirty = true; // the spatial order depends on the cell
	c->cache_valid = false;
	return MPMC_OK;
}

extern "C" int mpmc_set_options(mpmc_ctx *c, const mpmc_options *o) {
	if (!c || !o) return MPMC_ERR_ARG;
	if (o->unsupported_flags & ~(uint64_t)(MPMC_FLAG_WOLF | MPMC_FLAG_FEYNMAN_HIBBS)) { // (Wolf / Feynman-Hibbs travel in their own option fields)
		char buf[160];
		std::snprintf(buf, sizeof buf, "mpmc_set_options: reference option(s) outside the energy hot path are ON (flag mask 0x%llx)",
		              (unsigned long long)o->unsupported_flags);
		return fail(c, MPMC_ERR_UNSUPPORTED, buf);
	}
	if (o->polarization && !o->rd_only) {
		if (!o->polar_iterative)
			return fail(c, MPMC_ERR_UNSUPPORTED, "mpmc_set_options: polarization by matrix inversion (polar_iterative off) is not supported");
		if (o->damp_type != MPMC_DAMPING_EXPONENTIAL)
			return fail(c, MPMC_ERR_UNSUPPORTED, "mpmc_set_options: only polar_damp_type exponential is supported");
		if (o->polar_precision == 0.0 && o->polar_max_iter < 1)
			return fail(c, MPMC_ERR_INVALID_SETTING, "mpmc_set_options: polar_max_iter must be >= 1 when polar_precision is 0 (the reference never terminates)");
		if (o->polar_precision < 0.0) return fail(c, MPMC_ERR_INVALID_SETTING, "mpmc_set_options: polar_precision < 0");
		if (o->solver < MPMC_SOLVER_AUTO || o->solver > MPMC_SOLVER_DENSE) return fail(c, MPMC_ERR_INVALID_SETTING, "mpmc_set_options: bad solver");
	}
	if (o->ewald_kmax < 0 || o->ewald_kmax > 64) return fail(c, MPMC_ERR_INVALID_SETTING, "mpmc_set_options: ewald_kmax out of range");
	if (o->feynman_hibbs) {
		if (!(o->temperature > 0)) return fail(c, MPMC_ERR_INVALID_SETTING, "mpmc_set_options: feynman_hibbs requires positive temperature"); // SimulationControl.cpp:2509
		if (o->wolf && !o->rd_only) return fail(c, MPMC_ERR_INCOMPATIBLE, "mpmc_set_options: FH + es_wolf is not implemented"); // System.Energy.cpp:1448-1450
	}
	if (c->opts_set && std::memcmp(&c->opts, o, sizeof(mpmc_options)) == 0) return MPMC_OK; // unchanged: keep the accepted configuration's totals
	if (c->opts_set && (c->opts.polar_gs != 0) != (o->polar_gs != 0)) c->atoms_dirty = true; // Gauss-Seidel sweeps need the reference's atom order
	c->opts = *o;
	c->opts_set = true;
	c->k_dirty = true;
	c->cache_valid = false;
	return MPMC_OK;
}

// ---- atoms -----------------------------------------------------------------------------------------------
// nested bisection sort of the wrapped fractional coordinates: nx slabs in x, ny strips in y per slab, z order inside a
// strip; consecutive groups of 64 slots (tiles) are then roughly cubic cells.  Pure host code, O(N log N).
constexpr double kResortDrift = 2.0; // Angstrom; a tile is ~16 A wide at liquid density
static void compute_spatial_order(mpmc_ctx *c) {
	const int n = c->n;
	c->perm.resize(n);
	c->slot_of.resize(n);
	for (int i = 0; i < n; i++) c->perm[i] = i;
	bool enable = c->box_set && n > 2 * kTile;
	if (c->opts_set && c->opts.polar_gs && c->opts.polarization && !c->opts.rd_only) enable = false; // the sweep order IS the atom order (:3569)
	if (const char *e = std::getenv("MPMC_NO_SORT")) if (e[0] == '1') enable = false;
	if (enable) {
		// fractional coordinates counted from the smallest one in each dimension: with all atoms inside one period (the usual case) the
		// periodic wrap is cut at the edge of the occupied range, so tiles are compact in the RAW coordinates too -- which is what lets
		// whole tile pairs share one periodic image index (k_classify)
		std::vector<double> f(3 * (size_t)n);
		double org[3] = {1e300, 1e300, 1e300};
		for (int i = 0; i < n; i++)
			for (int p = 0; p < 3; p++) {
				double v = 0;
				for (int q = 0; q < 3; q++) v += c->box.r[3 * q + p] * c->h_pos[3 * i + q];
				f[3 * (size_t)i + p] = v;
				if (v < org[p]) org[p] = v;
			}
		for (int p = 0; p < 3; p++) c->sort_origin_f[p] = org[p];
		for (int i = 0; i < n; i++)
			for (int p = 0; p < 3; p++) {
				double v = f[3 * (size_t)i + p] - org[p];
				v -= std::floor(v);
				f[3 * (size_t)i + p] = v;
			}
		const int T = (n + kTile - 1) / kTile;
		const int nx = std::max(1, (int)std::lround(std::cbrt((double)T)));
		const int tiles_per_slab = (T + nx - 1) / nx;
		const int ny = std::max(1, (int)std::lround(std::sqrt((double)tiles_per_slab)));
		const int tiles_per_strip = (tiles_per_slab + ny - 1) / ny;
		auto by = [&](int dim) { return [&f, dim](int a, int b) { return f[3 * (size_t)a + dim] < f[3 * (size_t)b + dim] || (f[3 * (size_t)a + dim] == f[3 * (size_t)b + dim] && a < b); }; };
		std::sort(c->perm.begin(), c->perm.end(), by(0));
		const int slab = tiles_per_slab * kTile, strip = tiles_per_strip * kTile;
		for (int s0 = 0; s0 < n; s0 += slab) {
			const int s1 = std::min(n, s0 + slab);
			std::sort(c->perm.begin() + s0, c->perm.begin() + s1, by(1));
			for (int t0 = s0; t0 < s1; t0 += strip) {
				const int t1 = std::min(s1, t0 + strip);
				std::sort(c->perm.begin() + t0, c->perm.begin() + t1, by(2));
			}
		}
	}
	for (int k = 0; k < n; k++) c->slot_of[c->perm[k]] = k;
}

static int upload_atoms(mpmc_ctx *c) {
	compute_spatial_order(c);
	const int n = c->n, np = c->n_pad;
	std::vector<double4> xyzq(np);
	std::vector<double2> lj(np);
	std::vector<int2> mf(np);
	std::vector<double> al(np, 0.0), ep(np, 0.0), imm(np, 0.0);
	std::vector<int32_t> perm(np, -1), slot(np, -1);
	std::vector<double> molmass(n, 0.0); // Molecule::mass = sum of its atoms' masses (System.cpp:687), per atom
	if (!c->h_mass.empty())
		for (int i0 = 0; i0 < n;) {
			int i1 = i0;
			double m = 0;
			while (i1 < n && c->h_mol[i1] == c->h_mol[i0]) m += c->h_mass[i1++];
			for (int i = i0; i < i1; i++) molmass[i] = m;
			i0 = i1;
		}
	for (int k = 0; k < np; k++) {
		if (k < n) {
			const int i = c->perm[k];
			perm[k] = i;
			slot[i] = k;
			xyzq[k] = make_double4(c->h_pos[3 * i], c->h_pos[3 * i + 1], c->h_pos[3 * i + 2], c->h_q[i]);
			lj[k] = make_double2(std::fabs(c->h_sigma[i]), std::sqrt(c->h_eps[i]));
			int fl = 0;
			if (c->h_frozen[i]) fl |= AF_FROZEN;
			if (c->h_eps[i] == 0.0 || c->h_sigma[i] == 0.0) fl |= AF_NULL_RD;
			if (c->h_disp[i]) fl |= AF_HAS_DISP;
			if (c->h_sigma[i] < 0.0) fl |= AF_NEG_SIGMA;
			if (c->h_sigma[i] == 0.0) fl |= AF_ZERO_SIGMA;
			if (c->h_q[i] == 0.0) fl |= AF_ZERO_Q;
			if (c->h_alpha[i] == 0.0) fl |= AF_ZERO_ALPHA;
			mf[k] = make_int2(c->h_mol[i], fl);
			al[k] = c->h_alpha[i];
			ep[k] = c->h_eps[i];
			imm[k] = (molmass[i] > 0.0) ? 1.0 / molmass[i] : 0.0;
		} else {
			xyzq[k] = make_double4(0, 0, 0, 0);
			lj[k] = make_double2(0, 0);
			mf[k] = make_int2(-1 - k, AF_PAD | AF_FROZEN | AF_NULL_RD | AF_ZERO_SIGMA | AF_ZERO_Q | AF_ZERO_ALPHA);
		}
	}
	HIP_TRY(c, hipMemcpyAsync(c->d_xyzq, xyzq.data(), np * sizeof(double4), hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->d_lj, lj.data(), np * sizeof(double2), hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->d_mf, mf.data(), np * sizeof(int2), hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->d_alpha, al.data(), np * sizeof(double), hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->d_eps, ep.data(), np * sizeof(double), hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->d_inv_molmass, imm.data(), np * sizeof(double), hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->d_perm, perm.data(), np * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->d_slot_of, slot.data(), np * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
	// position-independent pair-flag counts (diagnostics of pair_exclusions), once per upload
	{
		AtomsDev at;
		at.xyzq = c->d_xyzq;
		at.lj = c->d_lj;
		at.mf = c->d_mf;
		at.alpha = c->d_alpha;
		at.eps = c->d_eps;
		at.inv_molmass = c->d_inv_molmass;
		at.n = c->n;
		at.n_pad = c->n_pad;
		launch_static_counts(c->stream, at, c->d_tile_pairs, c->n_tile_pairs, c->d_block_cnt, c->d_cnt);
		HIP_TRY(c, hipGetLastError());
		HIP_TRY(c, hipMemcpyAsync(c->static_cnt, c->d_cnt, 4 * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
	}
	HIP_TRY(c, hipStreamSynchronize(c->stream)); // staging vectors die here
	c->h_xyzq.swap(xyzq);          // slot-ordered mirror for bulk position updates
	c->h_pos_sorted = c->h_pos;    // where every atom stood when this order was made
	c->atoms_dirty = false;
	return MPMC_OK;
}

// A context that has to hold more atoms than it was created for is rebuilt in place: a fresh context of the larger capacity takes
// over the caller's handle (same device, box, options, profiling state), the old device buffers are released.  Everything sized by
// the capacity is allocated on first use, so nothing else has to know.
static int grow_capacity(mpmc_ctx *c, int n) {
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream2);
	(void)hipStreamSynchronize(c->stream);
	mpmc_ctx *f = nullptr;
	const int cap = n + n / 4 + kTile;
	int rc = mpmc_ctx_create(c->device, cap, &f);
	if (rc != MPMC_OK) return fail(c, rc, "mpmc_set_atoms: cannot grow the context to " + std::to_string(cap) + " atoms: " + g_create_error);
	if (c->box_set) rc = mpmc_set_box(f, c->box.b, c->box.r, c->box.volume, c->box.cutoff);
	if (rc == MPMC_OK && c->opts_set) rc = mpmc_set_options(f, &c->opts);
	if (rc != MPMC_OK) {
		c->err = "mpmc_set_atoms: growing the context failed: " + f->err;
		mpmc_ctx_destroy(f);
		return rc;
	}
	f->prof = c->prof;
	f->tim = c->tim;
	std::swap(*c, *f);
	mpmc_ctx_destroy(f); // now owns the old, smaller buffers
	return MPMC_OK;
}

extern "C" int mpmc_set_atoms(mpmc_ctx *c, int n, const double *pos, const double *charge, const double *polarizability, const double *epsilon,
                              const double *sigma, const int32_t *mol_id, const int32_t *frozen, const int32_t *has_disp, const double *mass) {
	if (!c || n <= 0 || !pos || !charge || !polarizability || !epsilon || !sigma || !mol_id || !frozen) return MPMC_ERR_ARG;
	if (n > c->max_atoms) { // insertions (uVT / Gibbs callers) outgrew the capacity hint given at creation
		const int rc_grow = grow_capacity(c, n);
		if (rc_grow != MPMC_OK) return rc_grow;
	}
	for (int i = 0; i < n; i++) {
		if (!std::isfinite(pos[3 * i]) || !std::isfinite(pos[3 * i + 1]) || !std::isfinite(pos[3 * i + 2]))
			return fail(c, MPMC_ERR_INVALID_DATUM, "mpmc_set_atoms: non-finite position");
		if (epsilon[i] < 0.0 || !std::isfinite(epsilon[i]) || !std::isfinite(sigma[i]) || !std::isfinite(charge[i]) || !std::isfinite(polarizability[i]))
			return fail(c, MPMC_ERR_INVALID_DATUM, "mpmc_set_atoms: epsilon < 0 or non-finite atom parameter");
	}
	{ // molecules are contiguous runs of the atom list (reference System.cpp:672): an id may not reappear later
		std::vector<int32_t> firsts;
		for (int i = 0; i < n; i++)
			if (i == 0 || mol_id[i] != mol_id[i - 1]) firsts.push_back(mol_id[i]);
		std::vector<int32_t> sorted = firsts;
		std::sort(sorted.begin(), sorted.end());
		if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
			return fail(c, MPMC_ERR_INVALID_DATUM, "mpmc_set_atoms: atoms of one molecule (equal mol_id) must be contiguous");
	}
	HIP_TRY(c, hipSetDevice(c->device));
	c->n = n;
	c->n_pad = ((n + kTile - 1) / kTile) * kTile;
	c->n_tiles = c->n_pad / kTile;
	c->h_pos.assign(pos, pos + 3 * (size_t)n);
	c->h_q.assign(charge, charge + n);
	c->h_alpha.assign(polarizability, polarizability + n);
	c->h_eps.assign(epsilon, epsilon + n);
	c->h_sigma.assign(sigma, sigma + n);
	c->h_mol.assign(mol_id, mol_id + n);
	c->h_frozen.assign(frozen, frozen + n);
	if (has_disp) c->h_disp.assign(has_disp, has_disp + n);
	else c->h_disp.assign(n, 0);
	if (mass) c->h_mass.assign(mass, mass + n);
	else c->h_mass.clear();

	// countN (reference src/System.cpp:909-931): molecules that are not frozen.  A molecule's flag is the
	// flag of its last atom row (the PQR reader overwrites molecule->frozen per atom, src/System.cpp:687).
	c->n_molecules = 0;
	c->N_movable = 0;
	for (int i = 0; i < n; i++) {
		const bool last_of_mol = (i == n - 1) || (mol_id[i + 1] != mol_id[i]);
		if (last_of_mol) {
			c->n_molecules++;
			if (!frozen[i]) c->N_movable += 1.0;
		}
	}

	int rc = MPMC_OK;
	c->atoms_dirty = true; // uploaded (in spatial order) by the next evaluation, when the box is known too

	// upper-triangular tile-pair schedule of the pair kernel
	const int nt = c->n_tiles;
	const size_t ntp = (size_t)nt * (nt + 1) / 2;
	if (ntp > c->cap_tile_pairs) {
		dev_free(c, &c->d_tile_pairs, c->cap_tile_pairs);
		dev_free(c, &c->d_block_part, 2 * c->cap_tile_pairs);
		dev_free(c, &c->d_block_cnt, 4 * c->cap_tile_pairs);
		dev_free(c, &c->d_cls, c->cap_tile_pairs);
		dev_free(c, &c->d_tp_shift, c->cap_tile_pairs);
		dev_free(c, &c->d_lists, 2 * c->cap_tile_pairs + 2);
		c->cap_tile_pairs = 0;
		if ((rc = dev_alloc(c, &c->d_tile_pairs, ntp)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_block_part, 2 * ntp)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_block_cnt, 4 * ntp)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_cls, ntp)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_tp_shift, ntp)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_lists, 2 * ntp + 2)) != MPMC_OK) return rc;
		HIP_TRY(c, hipMemset(c->d_cls, 0, ntp * sizeof(int)));
		c->cap_tile_pairs = ntp;
	}
	std::vector<int2> tp;
	tp.reserve(ntp);
	for (int I = 0; I < nt; I++)
		for (int J = I; J < nt; J++) tp.push_back(make_int2(I, J));
	HIP_TRY(c, hipMemcpy(c->d_tile_pairs, tp.data(), ntp * sizeof(int2), hipMemcpyHostToDevice));
	c->n_tile_pairs = (int)ntp;

	// j-range split of the per-atom (row) kernels: aim for >= ~4096 one-wave blocks
	c->n_split = std::max(1, std::min(nt, (4096 + nt - 1) / nt));
	c->atoms_set = true;
	c->pending = false;
	c->cache_valid = false;
	c->trial_open = false;
	return MPMC_OK;
}

extern "C" int mpmc_update_positions(mpmc_ctx *c, int first, int count, const double *pos) {
	if (!c || !pos || first < 0 || count < 0) return MPMC_ERR_ARG;
	if (!c->atoms_set || first + count > c->n) return fail(c, MPMC_ERR_ARG, "mpmc_update_positions: range outside the atom list");
	if (count == 0) return MPMC_OK;
	HIP_TRY(c, hipSetDevice(c->device));
	for (int t = 0; t < count; t++) {
		if (!std::isfinite(pos[3 * t]) || !std::isfinite(pos[3 * t + 1]) || !std::isfinite(pos[3 * t + 2]))
			return fail(c, MPMC_ERR_INVALID_DATUM, "mpmc_update_positions: non-finite position");
	}
	for (int t = 0; t < count; t++) {
		const int i = first + t;
		c->h_pos[3 * i] = pos[3 * t];
		c->h_pos[3 * i + 1] = pos[3 * t + 1];
		c->h_pos[3 * i + 2] = pos[3 * t + 2];
	}
	c->cache_valid = false; // the accepted totals no longer describe the resident configuration
	if (c->atoms_dirty) return MPMC_OK; // a full (re-sorted) upload is pending anyway
	if (count > 256) { // bulk update (typically: all positions handed over in host memory for every evaluation)
		// The atoms keep their slots -- the spatial order only matters for speed, the tile classes are recomputed from the actual
		// bounding boxes every evaluation -- and the whole position array goes up in ONE copy.  The order is refreshed (full upload)
		// once some atom has drifted further than kResortDrift from where it stood at the last sort.
		double worst = 0.0;
		for (int t = 0; t < count; t++) {
			const int i = first + t;
			double d2 = 0;
			for (int p = 0; p < 3; p++) {
				const double d = pos[3 * t + p] - c->h_pos_sorted[3 * (size_t)i + p];
				d2 += d * d;
			}
			if (d2 > worst) worst = d2;
		}
		if (c->h_pos_sorted.empty() || !(worst <= kResortDrift * kResortDrift)) {
			c->atoms_dirty = true;
			return MPMC_OK;
		}
		for (int t = 0; t < count; t++) {
			const int i = first + t;
			c->h_xyzq[c->slot_of[i]] = make_double4(pos[3 * t], pos[3 * t + 1], pos[3 * t + 2], c->h_q[i]);
		}
		HIP_TRY(c, hipMemcpyAsync(c->d_xyzq, c->h_xyzq.data(), (size_t)c->n_pad * sizeof(double4), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		return MPMC_OK;
	}
	for (int t = 0; t < count; t++) { // the moved atoms keep their slots (the order only matters for speed)
		const int i = first + t;
		const double4 v = make_double4(pos[3 * t], pos[3 * t + 1], pos[3 * t + 2], c->h_q[i]);
		HIP_TRY(c, hipMemcpyAsync(c->d_xyzq + c->slot_of[i], &v, sizeof(double4), hipMemcpyHostToDevice, c->stream));
	}
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	return MPMC_OK;
}

extern "C" int mpmc_set_positions_device(mpmc_ctx *c, const double *pos_device) {
	if (!c || !pos_device) return MPMC_ERR_ARG;
	if (!c->atoms_set) return fail(c, MPMC_ERR_ARG, "mpmc_set_positions_device: no atoms set");
	HIP_TRY(c, hipSetDevice(c->device));
	if (c->atoms_dirty) { // need the slot order first
		int rc = upload_atoms(c);
		if (rc != MPMC_OK) return rc;
	}
	launch_set_positions(c->stream, pos_device, c->d_perm, c->d_xyzq, c->n);
	HIP_TRY(c, hipGetLastError());
	// keep the host mirror coherent (update_com / later partial updates read it)
	HIP_TRY(c, hipMemcpyAsync(c->h_pos.data(), pos_device, 3 * (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	return MPMC_OK;
}

// ---- k-vector tables (hemisphere enumeration of coulombic_reciprocal :1577-1590 / recip_term :2849-2865) --------
static int build_k_tables(mpmc_ctx *c) {
	const int kmax = c->opts.ewald_kmax;
	const double alpha = c->ewald_alpha, ea = c->polar_ewald_alpha;
	std::vector<double4> kvec, kw;
	std::vector<double> wen;
	std::vector<int4> lvec;
	int l[3];
	for (l[0] = 0; l[0] <= kmax; l[0]++)
		for (l[1] = (!l[0] ? 0 : -kmax); l[1] <= kmax; l[1]++)
			for (l[2] = ((!l[0] && !l[1]) ? 1 : -kmax); l[2] <= kmax; l[2]++) {
				if (l[0] * l[0] + l[1] * l[1] + l[2] * l[2] > kmax * kmax) continue;
				double k[3];
				for (int p = 0; p < 3; p++) {
					k[p] = 0;
					for (int q = 0; q < 3; q++) k[p] += 2.0 * kPi * c->box.r[3 * p + q] * l[q];
				}
				const double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
				kvec.push_back(make_double4(k[0], k[1], k[2], k2));
				lvec.push_back(make_int4(l[0], l[1], l[2], 0));
				wen.push_back(std::exp(-k2 / (4.0 * alpha * alpha)) / k2);
				const double g = std::exp(-k2 / (4.0 * ea * ea));
				kw.push_back(make_double4(k[0] / k2 * g, k[1] / k2 * g, k[2] / k2 * g, 0.0));
			}
	const int K = (int)kvec.size();
	if (K > c->cap_K) {
		dev_free(c, &c->d_kvec, (size_t)c->cap_K);
		dev_free(c, &c->d_kw, (size_t)c->cap_K);
		dev_free(c, &c->d_lvec, (size_t)c->cap_K);
		dev_free(c, &c->d_sf, (size_t)c->cap_K);
		dev_free(c, &c->d_w_en, (size_t)c->cap_K);
		c->cap_K = 0;
		int rc;
		if ((rc = dev_alloc(c, &c->d_kvec, (size_t)K)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_kw, (size_t)K)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_lvec, (size_t)K)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_sf, (size_t)K)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_w_en, (size_t)K)) != MPMC_OK) return rc;
		c->cap_K = K;
	}
	if (K > 0) {
		HIP_TRY(c, hipMemcpy(c->d_kvec, kvec.data(), K * sizeof(double4), hipMemcpyHostToDevice));
		HIP_TRY(c, hipMemcpy(c->d_kw, kw.data(), K * sizeof(double4), hipMemcpyHostToDevice));
		HIP_TRY(c, hipMemcpy(c->d_lvec, lvec.data(), K * sizeof(int4), hipMemcpyHostToDevice));
		HIP_TRY(c, hipMemcpy(c->d_w_en, wen.data(), K * sizeof(double), hipMemcpyHostToDevice));
	}
	c->K = K;
	return MPMC_OK;
}

static int ensure_polar_buffers(mpmc_ctx *c) {
	const size_t np = (size_t)c->max_pad;
	int rc;
	if (!c->d_e_static) {
		if ((rc = dev_alloc(c, &c->d_e_static, 3 * np)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_mu[0], 3 * np)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_mu[1], 3 * np)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_e_induced, 3 * np)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_rrms, np)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_e_recip_part, (size_t)kKSplit * 3 * np)) != MPMC_OK) return rc;
		HIP_TRY(c, hipMemset(c->d_e_static, 0, 3 * np * sizeof(double)));
		HIP_TRY(c, hipMemset(c->d_mu[0], 0, 3 * np * sizeof(double)));
		HIP_TRY(c, hipMemset(c->d_mu[1], 0, 3 * np * sizeof(double)));
		HIP_TRY(c, hipMemset(c->d_e_induced, 0, 3 * np * sizeof(double)));
		HIP_TRY(c, hipMemset(c->d_rrms, 0, np * sizeof(double)));
	}
	// per-atom partial slots: one per source tile (symmetric kernels) -- also covers the n_split <= n_tiles slots
	// of the matrix-free row kernel
	const size_t need = (size_t)c->n_tiles * c->n_pad * 3;
	if (need > c->cap_part) {
		dev_free(c, &c->d_part, c->cap_part);
		c->cap_part = 0;
		if ((rc = dev_alloc(c, &c->d_part, need)) != MPMC_OK) return rc;
		c->cap_part = need;
	}
	return MPMC_OK;
}

// decide how the dipole iteration runs and (COMPACT) make room for the tensor store
static int resolve_solver(mpmc_ctx *c) {
	const size_t need = (size_t)c->n_tile_pairs * (kTile * kTile); // double2 elements, 16 B each
	int want = c->opts.solver;
	if (c->opts.polar_gs) want = MPMC_SOLVER_MATRIX_FREE; // Gauss-Seidel sweeps rebuild the tensors row block by row block (kernels_gs.hip)
	if (want == MPMC_SOLVER_AUTO) {
		size_t budget_mb = 4096;
		if (const char *e = std::getenv("MPMC_TENSOR_BUDGET_MB")) budget_mb = (size_t)std::strtoull(e, nullptr, 10);
		want = (need * sizeof(double2) <= budget_mb * (size_t)1048576) ? MPMC_SOLVER_COMPACT : MPMC_SOLVER_MATRIX_FREE;
		// building the store costs about as much as three iterations save (0.10 ms against 0.03 ms per iteration at 10 000 atoms)
		if (c->opts.polar_precision == 0.0 && c->opts.polar_max_iter <= 3) want = MPMC_SOLVER_MATRIX_FREE;
	}
	if (want == MPMC_SOLVER_DENSE) { // the reference's layout, on request only: (3 n_pad)^2 doubles
		const size_t nd = (size_t)3 * c->n_pad * (size_t)3 * c->n_pad;
		if (nd > c->cap_adense) {
			dev_free(c, &c->d_adense, c->cap_adense);
			c->cap_adense = 0;
			int rc = dev_alloc(c, &c->d_adense, nd);
			if (rc != MPMC_OK) return rc;
			c->cap_adense = nd;
		}
	}
	if (want == MPMC_SOLVER_COMPACT && need > c->cap_ab) {
		dev_free(c, &c->d_ab, c->cap_ab);
		c->cap_ab = 0;
		int rc = dev_alloc(c, &c->d_ab, need);
		if (rc != MPMC_OK) {
			if (c->opts.solver == MPMC_SOLVER_COMPACT) return rc; // explicitly requested: report
			(void)hipGetLastError();
			want = MPMC_SOLVER_MATRIX_FREE; // AUTO: fall back to recomputing the tensors (still the HIP path)
		} else {
			c->cap_ab = need;
		}
	}
	c->solver_used = want;
	return MPMC_OK;
}

// resolve alpha defaults, rebuild k tables when box/options changed
static int prepare(mpmc_ctx *c) {
	if (!c->box_set) return fail(c, MPMC_ERR_BOX, "energy: no box set (mpmc_set_box)");
	if (!c->atoms_set) return fail(c, MPMC_ERR_INVALID_DATUM, "energy: no atoms set (mpmc_set_atoms)");
	HIP_TRY(c, hipSetDevice(c->device));
	if (c->opts.feynman_hibbs && c->h_mass.empty())
		return fail(c, MPMC_ERR_INVALID_DATUM, "energy: feynman_hibbs needs atom masses (mpmc_set_atoms was called without them)");
	if (c->atoms_dirty) {
		int rc = upload_atoms(c);
		if (rc != MPMC_OK) return rc;
	}
	if (c->k_dirty) {
		// System::update_pbc, reference src/System.cpp:871-874
		c->ewald_alpha = (c->opts.ewald_alpha > 0) ? c->opts.ewald_alpha : 3.5 / c->box.cutoff;
		c->polar_ewald_alpha = (c->opts.polar_ewald_alpha > 0) ? c->opts.polar_ewald_alpha : 3.5 / c->box.cutoff;
		int rc = build_k_tables(c);
		if (rc != MPMC_OK) return rc;
		c->k_dirty = false;
	}
	return MPMC_OK;
}

static AtomsDev atoms_view(const mpmc_ctx *c) {
	AtomsDev a;
	a.xyzq = c->d_xyzq;
	a.lj = c->d_lj;
	a.mf = c->d_mf;
	a.alpha = c->d_alpha;
	a.eps = c->d_eps;
	a.inv_molmass = c->d_inv_molmass;
	a.n = c->n;
	a.n_pad = c->n_pad;
	return a;
}
static RecipDev recip_view(const mpmc_ctx *c) {
	RecipDev r;
	r.kvec = c->d_kvec;
	r.w_en = c->d_w_en;
	r.kw = c->d_kw;
	r.lvec = c->no_recip_tab ? nullptr : c->d_lvec;
	r.sf = c->d_sf;
	r.K = c->K;
	return r;
}

// which pieces of energy() to run
enum : unsigned { RUN_PAIR = 1, RUN_PAIR_ES = 2, RUN_RECIP = 4, RUN_ATOMTERMS = 8, RUN_FIELD = 16, RUN_SOLVE = 32, RUN_WOLF = 64 };

static int enqueue(mpmc_ctx *c, unsigned mask) {
	int rc = prepare(c);
	if (rc != MPMC_OK) return rc;
	const AtomsDev at = atoms_view(c);
	const RecipDev rcp = recip_view(c);
	const mpmc_options &o = c->opts;
	hipStream_t st = c->stream;
	c->run_mask = mask;
	c->have_polar = false;
	c->iters = 0;
	c->failed = 0;

	HIP_TRY(c, hipMemsetAsync(c->d_scal, 0, (S_COUNT + C_COUNT) * sizeof(double), st));

	if (mask & (RUN_FIELD | RUN_SOLVE)) {
		if ((rc = ensure_polar_buffers(c)) != MPMC_OK) return rc;
		if ((rc = resolve_solver(c)) != MPMC_OK) return rc;
	}
	const bool compact = (mask & RUN_SOLVE) && c->solver_used == MPMC_SOLVER_COMPACT;

	// ---- reciprocal space + O(N) atom terms on the side stream, next to the pair sweep ------------------------------
	const bool need_sf = (mask & RUN_RECIP) || ((mask & RUN_FIELD) && o.polar_ewald);
	// intramolecular charge-to-screen term of coulombic_real: position dependent but independent of the pair sweep; identically zero
	// when every molecule is a single atom
	const bool need_intra = (mask & RUN_PAIR) && (mask & RUN_PAIR_ES) && !(o.wolf && (mask & RUN_WOLF)) && (c->n_molecules != c->n);
	const bool side_work = need_sf || (mask & RUN_ATOMTERMS) || need_intra;
	// a fork/join costs ~20 us of dispatch latency: worth it next to reciprocal-space work, not for the O(N) atom terms alone
	const bool side_fork = c->two_streams && (need_sf || need_intra);
	if (side_work) {
		hipStream_t s2 = side_fork ? fork_side(c) : st;
		if (need_intra) {
			ProfScope p(c, MPMC_K_PAIR, s2);
			launch_intra_terms(s2, at, c->d_slot_of, c->ewald_alpha, c->d_scal);
		}
		{
			ProfScope p(c, MPMC_K_RECIP, s2);
			if (need_sf) {
				const size_t need_part = (size_t)c->n_tiles * (size_t)c->K;
				if (rcp.lvec && o.ewald_kmax <= kRecipTabMaxK && need_part > c->cap_sf_part) {
					dev_free(c, &c->d_sf_part, c->cap_sf_part);
					c->cap_sf_part = 0;
					if ((rc = dev_alloc(c, &c->d_sf_part, need_part)) != MPMC_OK) return rc;
					c->cap_sf_part = need_part;
				}
				launch_recip_sf(s2, at, c->box, rcp, o.ewald_kmax, c->d_sf_part);
			}
			if (mask & (RUN_RECIP | RUN_ATOMTERMS))
				launch_atom_terms(s2, at, rcp, c->box, c->ewald_alpha, (mask & RUN_ATOMTERMS) ? o.rd_lrc : 0, (mask & RUN_RECIP) ? 1 : 0, c->d_scal);
		}
		if ((mask & RUN_FIELD) && o.polar_ewald) {
			ProfScope p(c, MPMC_K_FIELD, s2);
			launch_field_recip(s2, at, c->box, rcp, o.ewald_kmax, c->d_e_recip_part);
		}
	}

	// ---- pairwise pass: one symmetric sweep (energies + counts, static-field partials, Thole tensor store) ----------
	if (mask & (RUN_PAIR | RUN_FIELD)) {
		ProfScope p(c, MPMC_K_PAIR);
		// tile-pair classes from this configuration's tile bounding boxes (orthorhombic cells; all "near" otherwise)
		if (c->no_classes) HIP_TRY(c, hipMemsetAsync(c->d_cls, 0, (size_t)c->n_tile_pairs * sizeof(int), st));
		else launch_tile_classes(st, at, c->box, c->d_tile_pairs, c->n_tile_pairs, (o.polarization && !o.rd_only) ? o.polar_damp : 0.0,
		                         c->d_tile_bounds, c->d_cls, c->no_uniform ? nullptr : c->d_tp_shift, c->sort_origin_f);
		FusedParams fp;
		fp.ewald_alpha = c->ewald_alpha;
		fp.polar_ewald_alpha = c->polar_ewald_alpha;
		fp.polar_damp = o.polar_damp;
		fp.rd_lrc = o.rd_lrc;
		fp.do_es = ((mask & RUN_PAIR_ES) || (mask & RUN_FIELD)) ? 1 : 0;
		fp.do_field = (mask & RUN_FIELD) ? (o.polar_ewald ? 1 : 2) : 0;
		fp.do_thole = compact ? 1 : 0;
		fp.wolf = (o.wolf && (mask & RUN_WOLF)) ? 1 : 0;
		fp.fh_order = o.feynman_hibbs ? ((o.feynman_hibbs_order == 4) ? 4 : 2) : 0;
		fp.fh_c2 = fp.fh_c4 = 0.0;
		if (fp.fh_order) { // reference constants.h:15-33: M2A2 hBar2 / (24 kB T) and M2A4 hBar4 / (1152 kB2 T^2), reduced mass in kg
			const double hBar2 = 1.11211999e-68, hBar4 = 1.23681087e-136, kB = 1.3806503e-23, kB2 = 1.90619525e-46, amu = 1.66053873e-27;
			fp.fh_c2 = 1.0e20 * (hBar2 / (24.0 * kB * o.temperature)) / amu;
			fp.fh_c4 = 1.0e40 * (hBar4 / (1152.0 * kB2 * o.temperature * o.temperature)) / (amu * amu);
		}
		fp.wolf_erfa_over_r = std::erf(c->ewald_alpha * c->box.cutoff) / c->box.cutoff;
		fp.wolf_inv_r2 = 1.0 / (c->box.cutoff * c->box.cutoff);
		if (compact && !c->jacobi_hybrid) // work lists of the two-kernel Jacobi form only
			launch_build_lists(st, c->d_cls, c->n_tile_pairs, c->d_lists, c->d_lists + 2 * (size_t)c->n_tile_pairs);
		launch_pair_fused(st, c->use_dpp, at, c->box, fp, c->d_tile_pairs, c->d_cls, c->n_tile_pairs, c->d_block_part, c->d_block_cnt, c->d_part,
		                  compact ? c->d_ab : nullptr);
	}
	if (side_work && side_fork) join_side(c);
	bool reduce_forked = false;
	if (mask & RUN_PAIR) { // the scalar totals of the sweep are only read back at the very end: fold them beside the field / dipole work
		reduce_forked = c->two_streams && (mask & RUN_FIELD) != 0;
		hipStream_t s3 = reduce_forked ? fork_side(c) : st;
		ProfScope p(c, MPMC_K_REDUCE, s3);
		launch_reduce_pairs(s3, c->d_block_part, c->d_block_cnt, c->n_tile_pairs, c->d_scal, c->d_cnt);
	}

	// ---- static field ---------------------------------------------------------------------------------------
	if (mask & RUN_FIELD) {
		ProfScope p(c, MPMC_K_FIELD);
		c->mu_cur = 0;
		launch_field_finalize(st, at, c->box, o.polar_ewald, c->d_e_recip_part, c->d_part, c->n_tiles, o.polar_gamma, c->d_e_static,
		                      c->d_mu[0]);
	}

	// ---- thole_iterative, reference src/System.Energy.cpp:3450-3543 ------------------------------------------------
	c->solve_deferred = false;
	c->last_batch = 1;
	if ((mask & RUN_SOLVE) && c->defer_solve && compact && c->jacobi_hybrid && o.polar_precision == 0.0 && !o.polar_gs) {
		// fixed iteration count, stored-tensor single-launch Jacobi: the caller runs the iterations of several systems together
		if (reduce_forked) join_side(c);
		if (!c->ev_phase) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_phase, hipEventDisableTiming));
		HIP_TRY(c, hipEventRecord(c->ev_phase, st));
		HIP_TRY(c, hipGetLastError());
		c->solve_deferred = true;
		return MPMC_OK;
	}
	if (mask & RUN_SOLVE) {
		const bool by_precision = (o.polar_precision != 0.0);
		const int want_rrms = (o.polar_rrms || o.polar_precision > 0) ? 1 : 0;
		const double allowed = by_precision ? o.polar_precision * o.polar_precision * kDebye2SKA * kDebye2SKA : 0.0;
		const bool dense = (c->solver_used == MPMC_SOLVER_DENSE) && !o.polar_gs;
		constexpr int kDenseChunks = 16;
		const int iter_slots = dense ? kDenseChunks : c->n_tiles;
		if (dense) { // thole_amatrix into device memory, once per evaluation (the positions changed)
			ProfScope p(c, MPMC_K_TENSOR);
			launch_dense_build(st, at, c->box, o.polar_damp, c->d_adense);
		}
		int it = 0;
		bool keep = true;
		while (keep) {
			it++;
			if (it >= kMaxIterationCount && by_precision) { // divergence: mu = alpha E0, iterator_failed (:3483-3494)
				launch_dipole_reset(st, at, c->d_e_static, c->d_mu[c->mu_cur]);
				c->failed = 1;
				break;
			}
			if (by_precision) HIP_TRY(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), st));
			if (o.polar_gs) { // in-place sweep in atom order; old_mu is kept only when rrms / precision need it (:3503-3507)
				double *mu = c->d_mu[c->mu_cur], *mu_old = c->d_mu[1 - c->mu_cur];
				if (want_rrms) HIP_TRY(c, hipMemcpyAsync(mu_old, mu, 3 * (size_t)at.n_pad * sizeof(double), hipMemcpyDeviceToDevice, st));
				{
					ProfScope p(c, MPMC_K_DIPOLE_ITER);
					launch_gs_sweep(st, at, c->box, o.polar_damp, c->d_e_static, mu, c->d_e_induced, c->d_part);
				}
				if (want_rrms) {
					ProfScope p(c, MPMC_K_REDUCE);
					launch_gs_finish(st, at, mu_old, mu, want_rrms, c->d_rrms, allowed, c->d_flag);
				}
				if (by_precision) {
					HIP_TRY(c, hipMemcpyAsync(c->h_flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
					HIP_TRY(c, hipStreamSynchronize(st));
					keep = (*c->h_flag != 0);
				} else {
					keep = (it != o.polar_max_iter);
				}
				continue;
			}
			if (dense) {
				ProfScope p(c, MPMC_K_DIPOLE_ITER);
				launch_dense_matvec(st, c->d_adense, c->n_pad, c->d_mu[c->mu_cur], kDenseChunks, c->d_part);
			} else if (compact && c->jacobi_hybrid) {
				ProfScope p(c, MPMC_K_DIPOLE_ITER);
				launch_dipole_iter_hybrid(st, c->jacc, at, c->box, c->d_mu[c->mu_cur], c->d_tile_pairs, c->d_cls,
				                          (c->no_uniform || c->no_classes) ? nullptr : c->d_tp_shift, c->n_tile_pairs, c->d_ab, c->d_part, o.polar_damp);
			} else if (compact) {
				const int *counts = c->d_lists + 2 * (size_t)c->n_tile_pairs;
				hipStream_t s2 = fork_side(c); // the fp64-bound far-field kernel runs beside the HBM-bound streaming kernel
				{
					ProfScope p(c, MPMC_K_DIPOLE_FAR, s2);
					launch_dipole_iter_far(s2, c->use_dpp, at, c->box, c->d_mu[c->mu_cur], c->d_tile_pairs, c->d_lists, counts, c->n_tile_pairs, c->d_part);
				}
				{
					ProfScope p(c, MPMC_K_DIPOLE_ITER);
					launch_dipole_iter_stream(st, c->use_dpp, at, c->box, c->d_mu[c->mu_cur], c->d_tile_pairs, c->d_lists, counts, c->n_tile_pairs,
					                          c->d_ab, c->d_part);
				}
				join_side(c);
			} else { // matrix-free: the same symmetric tile-pair walk with nothing stored (null store => damped tensors rebuilt)
				ProfScope p(c, MPMC_K_DIPOLE_ITER);
				launch_dipole_iter_hybrid(st, c->jacc, at, c->box, c->d_mu[c->mu_cur], c->d_tile_pairs, c->d_cls,
				                          (c->no_uniform || c->no_classes) ? nullptr : c->d_tp_shift, c->n_tile_pairs, nullptr, c->d_part, o.polar_damp);
			}
			{
				ProfScope p(c, MPMC_K_REDUCE);
				launch_dipole_update(st, at, c->d_e_static, c->d_part, iter_slots, c->d_mu[c->mu_cur], c->d_mu[1 - c->mu_cur], c->d_e_induced,
				                     want_rrms, c->d_rrms, allowed, c->d_flag);
			}
			c->mu_cur = 1 - c->mu_cur;
			if (by_precision) { // are_we_done_yet needs the verdict on the host
				HIP_TRY(c, hipMemcpyAsync(c->h_flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
				HIP_TRY(c, hipStreamSynchronize(st));
				keep = (*c->h_flag != 0);
			} else {
				keep = (it != o.polar_max_iter);
			}
		}
		c->iters = it;
		{
			ProfScope p(c, MPMC_K_REDUCE);
			launch_polar_energy(st, at, c->d_mu[c->mu_cur], c->d_e_static, want_rrms ? c->d_rrms : nullptr, c->d_scal);
		}
		c->have_polar = true;
	}
	if (reduce_forked) join_side(c);
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipMemcpyAsync(c->h_scal, c->d_scal, (S_COUNT + C_COUNT) * sizeof(double), hipMemcpyDeviceToHost, st));
	c->pending = true;
	return MPMC_OK;
}

static unsigned full_mask(const mpmc_ctx *c);

static int wait_and_fill(mpmc_ctx *c, mpmc_result *out) {
	if (!c->pending) return fail(c, MPMC_ERR_ARG, "mpmc_energy_wait: nothing enqueued");
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->sync_stream ? c->sync_stream : c->stream));
	c->sync_stream = nullptr;
	c->pending = false;
	prof_harvest(c);
	if (!out) return MPMC_OK;
	std::memset(out, 0, sizeof(*out));
	const double *s = c->h_scal;
	out->lj_pairs = s[S_LJ];
	out->lrc_pair = s[S_LRC_PAIR];
	out->lrc_self = s[S_LRC_SELF];
	out->rd_energy = (s[S_LJ] + s[S_LRC_PAIR]) + s[S_LRC_SELF];
	out->es_real = s[S_ES_REAL] - s[S_ES_INTRA];
	out->es_recip = s[S_ES_RECIP];
	out->es_self = s[S_ES_SELF];
	out->coulombic_energy = (out->es_real + out->es_recip) + out->es_self; // coulombic() :1412
	out->polarization_energy = s[S_POLAR];
	out->dipole_rrms = s[S_RRMS];
	out->energy = out->rd_energy + out->coulombic_energy + out->polarization_energy + out->vdw_energy + out->three_body_energy; // :136
	out->N = c->N_movable;
	out->NU = out->N * out->energy; // :162
	out->n_pairs = (int64_t)c->n * (c->n - 1) / 2;
	out->n_lj_in_cutoff = c->h_cnt[C_LJ_IN];
	out->n_es_in_cutoff = c->h_cnt[C_ES_IN];
	out->n_intra = c->static_cnt[0];
	out->n_rd_excluded = c->static_cnt[1];
	out->n_es_excluded = c->static_cnt[2];
	out->n_frozen = c->static_cnt[3];
	out->polar_iterations = c->iters;
	out->iterator_failed = c->failed;
	if (c->run_mask == full_mask(c)) { // a complete energy(): it re-bases the trial-move totals
		c->last_full = *out;
		c->cache_valid = true;
	}
	return MPMC_OK;
}

static unsigned full_mask(const mpmc_ctx *c) {
	unsigned m = RUN_PAIR | RUN_ATOMTERMS;
	if (!c->opts.rd_only) {
		m |= RUN_PAIR_ES;
		m |= c->opts.wolf ? RUN_WOLF : RUN_RECIP; // coulombic() :1404-1413: Wolf replaces real + reciprocal + self
		if (c->opts.polarization) m |= RUN_FIELD | RUN_SOLVE;
	}
	return m;
}

extern "C" int mpmc_energy_async(mpmc_ctx *c) {
	if (!c) return MPMC_ERR_ARG;
	return enqueue(c, full_mask(c));
}
extern "C" int mpmc_energy_wait(mpmc_ctx *c, mpmc_result *out) {
	if (!c) return MPMC_ERR_ARG;
	return wait_and_fill(c, out);
}
extern "C" int mpmc_energy(mpmc_ctx *c, mpmc_result *out) {
	if (!c || !out) return MPMC_ERR_ARG;
	int rc = enqueue(c, full_mask(c));
	if (rc != MPMC_OK) return rc;
	return wait_and_fill(c, out);
}

// ---- trial moves -----------------------------------------------------------------------------------------------
constexpr size_t kMvBlobBytes = MPMC_TRIAL_MAX_ATOMS * (2 * sizeof(int) + sizeof(double4));
static int ensure_trial_buffers(mpmc_ctx *c) {
	int rc;
	if (!c->d_mv_blob) {
		if ((rc = dev_alloc(c, &c->d_mv_blob, kMvBlobBytes)) != MPMC_OK) return rc;
		c->d_mv_new = reinterpret_cast<double4 *>(c->d_mv_blob); // 32-byte records first (alignment), then the two int lists
		c->d_mv_slot = reinterpret_cast<int *>(c->d_mv_blob + MPMC_TRIAL_MAX_ATOMS * sizeof(double4));
		c->d_mv_orig = c->d_mv_slot + MPMC_TRIAL_MAX_ATOMS;
		if ((rc = dev_alloc(c, &c->d_moved_idx, (size_t)c->max_pad)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_delta_out, (size_t)8)) != MPMC_OK) return rc; // 5 doubles + 2 int64 counts
		c->d_delta_cnt = reinterpret_cast<long long *>(c->d_delta_out + 5);
		HIP_TRY(c, hipMemset(c->d_moved_idx, 0xff, (size_t)c->max_pad * sizeof(int))); // all -1
		HIP_TRY(c, hipHostMalloc((void **)&c->h_delta_out, 8 * sizeof(double)));
		c->h_delta_cnt = reinterpret_cast<long long *>(c->h_delta_out + 5);
		HIP_TRY(c, hipHostMalloc((void **)&c->h_mv_blob, kMvBlobBytes));
	}
	if (c->K > c->cap_sf_trial) {
		dev_free(c, &c->d_sf_trial, (size_t)c->cap_sf_trial);
		c->cap_sf_trial = 0;
		if ((rc = dev_alloc(c, &c->d_sf_trial, (size_t)std::max(c->K, 1))) != MPMC_OK) return rc;
		c->cap_sf_trial = std::max(c->K, 1);
	}
	return MPMC_OK;
}

extern "C" int mpmc_trial_begin(mpmc_ctx *c, int first, int count, const double *new_pos) {
	if (!c || !new_pos || first < 0 || count <= 0) return MPMC_ERR_ARG;
	if (!c->atoms_set || first + count > c->n) return fail(c, MPMC_ERR_ARG, "mpmc_trial_begin: range outside the atom list");
	if (c->trial_open) return fail(c, MPMC_ERR_ARG, "mpmc_trial_begin: a trial move is already open (accept or reject it first)");
	if (!c->cache_valid) return fail(c, MPMC_ERR_ARG, "mpmc_trial_begin: no accepted configuration (call mpmc_energy first)");
	for (int t = 0; t < 3 * count; t++)
		if (!std::isfinite(new_pos[t])) return fail(c, MPMC_ERR_INVALID_DATUM, "mpmc_trial_begin: non-finite position");
	c->trial_first = first;
	c->trial_count = count;
	c->trial_new.assign(new_pos, new_pos + 3 * (size_t)count);
	c->trial_old.assign(c->h_pos.begin() + 3 * (size_t)first, c->h_pos.begin() + 3 * (size_t)(first + count));
	c->trial_open = true;
	c->trial_evaluated = false;
	c->trial_enqueued = false;
	// a "move" that leaves every coordinate as it is (e.g. the box a two-box move does not touch): the trial totals ARE the accepted
	// totals, nothing is evaluated.  (The reference's bead moves re-centre the whole chain, so they do touch every image.)
	c->trial_noop = (std::memcmp(new_pos, c->h_pos.data() + 3 * (size_t)first, 3 * (size_t)count * sizeof(double)) == 0);
	return MPMC_OK;
}

// the two halves of mpmc_trial_energy: everything up to the last enqueue, then the wait + host arithmetic (P images of a
// path-integral move overlap on the device when a driver enqueues all of them before the first wait)
extern "C" int mpmc_trial_energy_async(mpmc_ctx *c) {
	if (!c) return MPMC_ERR_ARG;
	if (!c->trial_open) return fail(c, MPMC_ERR_ARG, "mpmc_trial_energy: no trial move is open");
	if (c->trial_enqueued) return fail(c, MPMC_ERR_ARG, "mpmc_trial_energy_async: already enqueued");
	if (c->trial_noop) {
		c->trial_was_full = false;
		c->trial_enqueued = true;
		return MPMC_OK;
	}
	const mpmc_options &o = c->opts;
	const bool polar = o.polarization && !o.rd_only;
	const int m = c->trial_count;
	if (polar || m > MPMC_TRIAL_MAX_ATOMS || o.wolf || o.feynman_hibbs) { // (the delta kernels carry the base LJ + Ewald terms only)
		// the dipole solve couples every atom: evaluate the trial configuration in full (still on the device)
		c->trial_keep = c->last_full;
		int rc = mpmc_update_positions(c, c->trial_first, m, c->trial_new.data());
		if (rc != MPMC_OK) return rc;
		if ((rc = mpmc_energy_async(c)) != MPMC_OK) return rc;
		c->trial_was_full = true;
		c->trial_enqueued = true;
		return MPMC_OK;
	}
	int rc = prepare(c);
	if (rc != MPMC_OK) return rc;
	if ((rc = ensure_trial_buffers(c)) != MPMC_OK) return rc;
	hipStream_t st = c->stream;
	{ // one pinned staging record, one host-to-device copy
		double4 *nw = reinterpret_cast<double4 *>(c->h_mv_blob);
		int *slots = reinterpret_cast<int *>(c->h_mv_blob + MPMC_TRIAL_MAX_ATOMS * sizeof(double4));
		int *origs = slots + MPMC_TRIAL_MAX_ATOMS;
		for (int t = 0; t < m; t++) {
			const int i = c->trial_first + t;
			origs[t] = i;
			slots[t] = c->slot_of[i];
			nw[t] = make_double4(c->trial_new[3 * t], c->trial_new[3 * t + 1], c->trial_new[3 * t + 2], c->h_q[i]);
		}
		HIP_TRY(c, hipMemcpyAsync(c->d_mv_blob, c->h_mv_blob, kMvBlobBytes, hipMemcpyHostToDevice, st));
	}
	const int do_es = o.rd_only ? 0 : 1;
	{
		ProfScope p(c, MPMC_K_PAIR);
		launch_delta(st, atoms_view(c), c->d_slot_of, c->box, recip_view(c), c->ewald_alpha, do_es, c->d_mv_slot, c->d_mv_orig, c->d_mv_new, m,
		             c->d_moved_idx, c->d_sf_trial, c->d_block_part, c->d_block_cnt, c->d_delta_out, c->d_delta_cnt);
	}
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipMemcpyAsync(c->h_delta_out, c->d_delta_out, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
	c->trial_was_full = false;
	c->trial_enqueued = true;
	return MPMC_OK;
}

extern "C" int mpmc_trial_energy_wait(mpmc_ctx *c, mpmc_result *out) {
	if (!c || !out) return MPMC_ERR_ARG;
	if (!c->trial_open || !c->trial_enqueued) return fail(c, MPMC_ERR_ARG, "mpmc_trial_energy_wait: nothing enqueued");
	c->trial_enqueued = false;
	if (c->trial_noop) {
		c->trial_res = c->last_full;
		c->trial_evaluated = true;
		*out = c->trial_res;
		return MPMC_OK;
	}
	if (c->trial_was_full) {
		int rc = mpmc_energy_wait(c, out);
		if (rc != MPMC_OK) return rc;
		c->trial_res = *out;
		c->last_full = c->trial_keep; // still the ACCEPTED configuration's totals until mpmc_trial_accept
		c->trial_evaluated = true;
		return MPMC_OK;
	}
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	prof_harvest(c);
	const int do_es = c->opts.rd_only ? 0 : 1;
	const mpmc_result &a = c->last_full;
	mpmc_result r = a;
	r.lj_pairs = a.lj_pairs + c->h_delta_out[0];
	r.rd_energy = (r.lj_pairs + r.lrc_pair) + r.lrc_self;
	r.n_lj_in_cutoff = a.n_lj_in_cutoff + c->h_delta_cnt[0];
	if (do_es) {
		r.es_real = a.es_real + (c->h_delta_out[1] - c->h_delta_out[2]);
		r.es_recip = c->h_delta_out[3];
		r.coulombic_energy = (r.es_real + r.es_recip) + r.es_self;
		r.n_es_in_cutoff = a.n_es_in_cutoff + c->h_delta_cnt[1];
	}
	r.energy = r.rd_energy + r.coulombic_energy + r.polarization_energy + r.vdw_energy + r.three_body_energy;
	r.NU = r.N * r.energy;
	c->trial_res = r;
	c->trial_evaluated = true;
	*out = r;
	return MPMC_OK;
}

extern "C" int mpmc_trial_energy(mpmc_ctx *c, mpmc_result *out) {
	if (!c || !out) return MPMC_ERR_ARG;
	int rc = mpmc_trial_energy_async(c);
	if (rc != MPMC_OK) return rc;
	return mpmc_trial_energy_wait(c, out);
}

extern "C" int mpmc_trial_accept(mpmc_ctx *c) {
	if (!c) return MPMC_ERR_ARG;
	if (!c->trial_open || !c->trial_evaluated) return fail(c, MPMC_ERR_ARG, "mpmc_trial_accept: no evaluated trial move");
	HIP_TRY(c, hipSetDevice(c->device));
	const int m = c->trial_count;
	if (c->trial_noop) {
		c->trial_open = false;
		return MPMC_OK;
	}
	if (!c->trial_was_full) {
		launch_commit_positions(c->stream, c->d_xyzq, c->d_mv_slot, c->d_mv_new, m);
		HIP_TRY(c, hipGetLastError());
		std::swap(c->d_sf, c->d_sf_trial); // the trial structure factors become the accepted ones
		std::swap(c->cap_K, c->cap_sf_trial);
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		for (int t = 0; t < 3 * m; t++) c->h_pos[3 * (size_t)c->trial_first + t] = c->trial_new[t];
	}
	c->last_full = c->trial_res;
	c->cache_valid = true;
	c->trial_open = false;
	return MPMC_OK;
}

extern "C" int mpmc_trial_reject(mpmc_ctx *c) {
	if (!c) return MPMC_ERR_ARG;
	if (!c->trial_open) return fail(c, MPMC_ERR_ARG, "mpmc_trial_reject: no trial move is open");
	if (c->trial_enqueued) { // enqueued but never waited for: drain it first
		mpmc_result drop;
		int rc = mpmc_trial_energy_wait(c, &drop);
		if (rc != MPMC_OK) return rc;
	}
	c->trial_open = false;
	if (c->trial_evaluated && c->trial_was_full) { // the resident configuration is the trial one: put the old positions back
		const mpmc_result keep = c->last_full;
		int rc = mpmc_update_positions(c, c->trial_first, c->trial_count, c->trial_old.data());
		if (rc != MPMC_OK) return rc;
		c->last_full = keep;
		c->cache_valid = true;
		const bool polar = c->opts.polarization && !c->opts.rd_only;
		if (!polar && !c->opts.wolf) { // the resident structure factors are the trial ones: re-base on the restored configuration
			mpmc_result tmp;
			if ((rc = mpmc_energy(c, &tmp)) != MPMC_OK) return rc;
		}
	}
	return MPMC_OK;
}

// ---- component entry points --------------------------------------------------------------------------------
static int run_piece(mpmc_ctx *c, unsigned mask, mpmc_result *r) {
	if (!c) return MPMC_ERR_ARG;
	int rc = enqueue(c, mask);
	if (rc != MPMC_OK) return rc;
	return wait_and_fill(c, r);
}
extern "C" int mpmc_lj(mpmc_ctx *c, double *out) {
	mpmc_result r;
	int rc = run_piece(c, RUN_PAIR | RUN_ATOMTERMS, &r);
	if (rc == MPMC_OK && out) *out = r.rd_energy;
	return rc;
}
extern "C" int mpmc_coulombic_real(mpmc_ctx *c, double *out) {
	mpmc_result r;
	int rc = run_piece(c, RUN_PAIR | RUN_PAIR_ES, &r);
	if (rc == MPMC_OK && out) *out = r.es_real;
	return rc;
}
extern "C" int mpmc_coulombic_reciprocal(mpmc_ctx *c, double *out) {
	mpmc_result r;
	int rc = run_piece(c, RUN_RECIP, &r);
	if (rc == MPMC_OK && out) *out = r.es_recip;
	return rc;
}
extern "C" int mpmc_coulombic_self(mpmc_ctx *c, double *out) {
	mpmc_result r;
	int rc = run_piece(c, RUN_RECIP, &r);
	if (rc == MPMC_OK && out) *out = r.es_self;
	return rc;
}
extern "C" int mpmc_coulombic(mpmc_ctx *c, double *out) {
	if (!c) return MPMC_ERR_ARG;
	mpmc_result r;
	int rc = run_piece(c, RUN_PAIR | RUN_PAIR_ES | (c->opts.wolf ? RUN_WOLF : RUN_RECIP), &r);
	if (rc == MPMC_OK && out) *out = r.coulombic_energy;
	return rc;
}
extern "C" int mpmc_polar(mpmc_ctx *c, double *out) {
	if (!c) return MPMC_ERR_ARG;
	if (!c->opts.polarization) return fail(c, MPMC_ERR_INVALID_SETTING, "mpmc_polar: polarization is off");
	mpmc_result r;
	int rc = run_piece(c, RUN_FIELD | RUN_SOLVE, &r);
	if (rc == MPMC_OK && out) *out = r.polarization_energy;
	return rc;
}
// device per-atom vectors are in slot order; everything handed to the caller is in original atom order
static int fetch_atoms3(mpmc_ctx *c, const double *d_src, double *out) {
	std::vector<double> tmp(3 * (size_t)c->n);
	HIP_TRY(c, hipMemcpy(tmp.data(), d_src, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
	for (int k = 0; k < c->n; k++) {
		const int i = c->perm[k];
		out[3 * (size_t)i] = tmp[3 * (size_t)k];
		out[3 * (size_t)i + 1] = tmp[3 * (size_t)k + 1];
		out[3 * (size_t)i + 2] = tmp[3 * (size_t)k + 2];
	}
	return MPMC_OK;
}

extern "C" int mpmc_thole_field(mpmc_ctx *c, double *ef_static) {
	if (!c) return MPMC_ERR_ARG;
	mpmc_result r;
	int rc = run_piece(c, RUN_FIELD, &r);
	if (rc != MPMC_OK) return rc;
	if (ef_static) return fetch_atoms3(c, c->d_e_static, ef_static);
	return MPMC_OK;
}

extern "C" int mpmc_thole_amatrix(mpmc_ctx *c, int row0, int nrows, double *a) {
	if (!c || !a || row0 < 0 || nrows <= 0) return MPMC_ERR_ARG;
	int rc = prepare(c);
	if (rc != MPMC_OK) return rc;
	if (row0 % 3 || nrows % 3 || row0 + nrows > 3 * c->n) return fail(c, MPMC_ERR_ARG, "mpmc_thole_amatrix: rows must cover whole atoms (multiples of 3) inside 3N");
	const size_t need = (size_t)nrows * 3 * c->n;
	if (need > c->cap_arows) {
		dev_free(c, &c->d_arows, c->cap_arows);
		c->cap_arows = 0;
		if ((rc = dev_alloc(c, &c->d_arows, need)) != MPMC_OK) return rc;
		c->cap_arows = need;
	}
	{
		ProfScope p(c, MPMC_K_TENSOR);
		launch_amatrix_rows(c->stream, atoms_view(c), c->d_slot_of, c->box, c->opts.polar_damp, row0, nrows, c->d_arows);
	}
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipMemcpyAsync(a, c->d_arows, need * sizeof(double), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	prof_harvest(c);
	return MPMC_OK;
}

extern "C" int mpmc_get_dipoles(mpmc_ctx *c, double *mu, double *ef_static, double *ef_induced) {
	if (!c) return MPMC_ERR_ARG;
	if (!c->d_e_static) return fail(c, MPMC_ERR_ARG, "mpmc_get_dipoles: no polarization evaluation has run");
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	int rc = MPMC_OK;
	if (mu && rc == MPMC_OK) rc = fetch_atoms3(c, c->d_mu[c->mu_cur], mu);
	if (ef_static && rc == MPMC_OK) rc = fetch_atoms3(c, c->d_e_static, ef_static);
	if (ef_induced && rc == MPMC_OK) rc = fetch_atoms3(c, c->d_e_induced, ef_induced);
	return rc;
}

// update_com + wrap_all, reference src/System.cpp:1347-1425 (host side: O(N), consumed by I/O only)
extern "C" int mpmc_update_com(mpmc_ctx *c, double *com, double *wrapped_com, double *wrapped_pos, int *n_molecules) {
	if (!c) return MPMC_ERR_ARG;
	if (!c->atoms_set || !c->box_set) return fail(c, MPMC_ERR_ARG, "mpmc_update_com: atoms and box must be set");
	if (c->h_mass.empty()) return fail(c, MPMC_ERR_ARG, "mpmc_update_com: mpmc_set_atoms was called without masses");
	if (n_molecules) *n_molecules = c->n_molecules;
	int m = 0;
	for (int i0 = 0; i0 < c->n;) {
		int i1 = i0;
		while (i1 + 1 < c->n && c->h_mol[i1 + 1] == c->h_mol[i0]) i1++;
		double cm[3] = {0, 0, 0}, mass = 0;
		for (int i = i0; i <= i1; i++) {
			mass += c->h_mass[i];
			for (int p = 0; p < 3; p++) cm[p] += c->h_mass[i] * c->h_pos[3 * i + p];
		}
		for (int p = 0; p < 3; p++) cm[p] /= mass;
		const bool mol_frozen = c->h_frozen[i1] != 0;
		double w[3] = {0, 0, 0};
		if (!mol_frozen) {
			double d[3];
			for (int p = 0; p < 3; p++) {
				d[p] = 0;
				for (int q = 0; q < 3; q++) d[p] += c->box.r[3 * q + p] * cm[q];
				d[p] = std::rint(d[p]);
			}
			for (int p = 0; p < 3; p++) {
				w[p] = 0;
				for (int q = 0; q < 3; q++) w[p] += c->box.b[3 * q + p] * d[q];
			}
		}
		if (com)
			for (int p = 0; p < 3; p++) com[3 * m + p] = cm[p];
		if (wrapped_com)
			for (int p = 0; p < 3; p++) wrapped_com[3 * m + p] = w[p]; // the reference stores the lattice shift here (:1404)
		if (wrapped_pos)
			for (int i = i0; i <= i1; i++)
				for (int p = 0; p < 3; p++) wrapped_pos[3 * i + p] = mol_frozen ? c->h_pos[3 * i + p] : c->h_pos[3 * i + p] - w[p];
		m++;
		i0 = i1 + 1;
	}
	return MPMC_OK;
}

// ---- path integral ---------------------------------------------------------------------------------------
// Enqueue one full evaluation of every system.  Systems whose solve can be deferred (same device, same box and options, fixed
// iteration count, stored-tensor single-launch Jacobi) run everything up to the static field on their own streams -- the pair
// sweeps of different systems overlap -- and then their dipole iterations together: one launch per iteration for the whole group
// (SolveBead array, blockIdx.y = system) on the first system's stream, which also carries the final copies of every member.
static bool same_solve_shape(const mpmc_ctx *a, const mpmc_ctx *b) {
	return a->device == b->device && a->n == b->n && a->n_pad == b->n_pad && a->n_tile_pairs == b->n_tile_pairs &&
	       std::memcmp(&a->opts, &b->opts, sizeof(mpmc_options)) == 0 && std::memcmp(a->box.b, b->box.b, sizeof(a->box.b)) == 0 &&
	       a->jacc == b->jacc && a->no_uniform == b->no_uniform && a->no_classes == b->no_classes;
}
static int pi_enqueue_all(mpmc_ctx **beads, int n_local) {
	// Opt-in (MPMC_PI_LOCKSTEP=1).  Measured on MI355X, 32 beads of the 10 000-atom box: the lockstep launches run each bead's
	// contraction exactly as fast as a launch of its own (0.107 ms per bead: the kernel is issue-bound, not tail-bound), while
	// independent streams let one bead's pair sweep fill the stalls of another bead's iterations -- 650 evaluations/s in lockstep
	// against 737 on independent streams.  The lockstep form stays for its clean per-launch timings.
	const char *e = std::getenv("MPMC_PI_LOCKSTEP");
	const bool lockstep = e && e[0] == '1';
	for (int b = 0; b < n_local; b++) {
		mpmc_ctx *c = beads[b];
		if (!c) return MPMC_ERR_ARG;
		c->defer_solve = lockstep && n_local > 1;
		int rc = enqueue(c, full_mask(c));
		c->defer_solve = false;
		if (rc != MPMC_OK) return rc;
	}
	std::vector<char> done(n_local, 0);
	for (int lead = 0; lead < n_local; lead++) {
		if (done[lead] || !beads[lead]->solve_deferred) continue;
		std::vector<mpmc_ctx *> grp;
		for (int b = lead; b < n_local; b++)
			if (!done[b] && beads[b]->solve_deferred && same_solve_shape(beads[lead], beads[b])) {
				grp.push_back(beads[b]);
				done[b] = 1;
			}
		mpmc_ctx *L = grp[0];
		const int nb = (int)grp.size();
		const mpmc_options &o = L->opts;
		HIP_TRY(L, hipSetDevice(L->device));
		hipStream_t st = L->stream;
		if (nb > L->cap_solve_args) {
			dev_free(L, &L->d_solve_args, (size_t)L->cap_solve_args);
			L->cap_solve_args = 0;
			int rc = dev_alloc(L, &L->d_solve_args, (size_t)nb);
			if (rc != MPMC_OK) return rc;
			L->cap_solve_args = nb;
		}
		std::vector<SolveBead> &args = L->h_solve_args;
		args.resize(nb);
		for (int k = 0; k < nb; k++) {
			mpmc_ctx *c = grp[k];
			SolveBead &a = args[k];
			a.at = atoms_view(c);
			a.tile_pairs = c->d_tile_pairs;
			a.cls = c->d_cls;
			a.tp_shift = (c->no_uniform || c->no_classes) ? nullptr : c->d_tp_shift;
			a.ab = c->d_ab;
			a.part = c->d_part;
			a.mu[0] = c->d_mu[0];
			a.mu[1] = c->d_mu[1];
			a.e_static = c->d_e_static;
			a.e_induced = c->d_e_induced;
			a.rrms = c->d_rrms;
			a.scal = c->d_scal;
			if (k > 0) HIP_TRY(L, hipStreamWaitEvent(st, c->ev_phase, 0)); // the member's pre-solve work (its own stream) is done
		}
		HIP_TRY(L, hipMemcpyAsync(L->d_solve_args, args.data(), (size_t)nb * sizeof(SolveBead), hipMemcpyHostToDevice, st));
		const int want_rrms = o.polar_rrms ? 1 : 0;
		int cur = 0; // field_finalize wrote mu[0]
		for (int it = 1; it <= o.polar_max_iter; it++) {
			{
				ProfScope p(L, MPMC_K_DIPOLE_ITER);
				launch_dipole_iter_hybrid_batched(st, L->jacc, L->d_solve_args, nb, L->box, cur, L->n_tile_pairs);
			}
			{
				ProfScope p(L, MPMC_K_REDUCE);
				launch_dipole_update_batched(st, L->d_solve_args, nb, L->n_pad, L->n_tiles, cur, want_rrms);
			}
			cur = 1 - cur;
		}
		{
			ProfScope p(L, MPMC_K_REDUCE);
			launch_polar_energy_batched(st, L->d_solve_args, nb, cur, want_rrms);
		}
		HIP_TRY(L, hipGetLastError());
		for (int k = 0; k < nb; k++) {
			mpmc_ctx *c = grp[k];
			c->mu_cur = cur;
			c->iters = o.polar_max_iter;
			c->have_polar = true;
			c->last_batch = nb;
			HIP_TRY(L, hipMemcpyAsync(c->h_scal, c->d_scal, (S_COUNT + C_COUNT) * sizeof(double), hipMemcpyDeviceToHost, st));
			c->sync_stream = st;
			c->pending = true;
			c->solve_deferred = false;
		}
	}
	return MPMC_OK;
}

// systems per launch in the dipole iterations of this context's last evaluation (1: it ran on its own)
extern "C" int mpmc_last_batch_size(mpmc_ctx *c) { return c ? c->last_batch : 0; }

extern "C" int mpmc_pi_potential_local(mpmc_ctx **beads, int n_local, double sums4[4], mpmc_result *per_bead, int *any_failed) {
	if (!beads || n_local < 0 || !sums4) return MPMC_ERR_ARG;
	{ // every bead enqueued before the first wait; the dipole iterations of compatible beads run in lockstep, in shared launches
		int rc = pi_enqueue_all(beads, n_local);
		if (rc != MPMC_OK) return rc;
	}
	sums4[0] = sums4[1] = sums4[2] = sums4[3] = 0;
	int failed = 0;
	for (int b = 0; b < n_local; b++) {
		mpmc_result r;
		int rc = mpmc_energy_wait(beads[b], &r);
		if (rc != MPMC_OK) return rc;
		sums4[0] += r.rd_energy; // ordered accumulation, PathIntegral.cpp:791-796
		sums4[1] += r.coulombic_energy;
		sums4[2] += r.polarization_energy;
		sums4[3] += r.vdw_energy;
		failed |= r.iterator_failed;
		if (per_bead) per_bead[b] = r;
	}
	if (any_failed) *any_failed = failed;
	return MPMC_OK;
}

extern "C" double mpmc_pi_finish(const double s[4], int P, double obs4[4]) {
	double o[4];
	for (int k = 0; k < 4; k++) o[k] = s[k] / P; // :798-801
	if (obs4)
		for (int k = 0; k < 4; k++) obs4[k] = o[k];
	return o[0] + o[1] + o[3] + o[2]; // rd + coulombic + vdw + polarization, :803-804
}

// PI_chain_mass_length2_ENTIRE_SYSTEM / PI_chain_mass_length2(vector<Molecule*>&), PathIntegral.cpp:851-965
extern "C" double mpmc_pi_chain_mass_length2(int P, int nmol, const double *com, const double *mol_mass, const int32_t *movable) {
	const double AMU2KG = 1.66053873e-27, ANGSTROM2METER = 1.0e-10; // src/constants.h:31,40
	double sum = 0;
	for (int m = 0; m < nmol; m++) {
		if (movable && !movable[m]) continue; // :881
		double len2 = 0;
		for (int i = 0; i < P; i++) { // closed loop over adjacent images, :956-960
			const int j = (i + 1) % P;
			const double *a = com + 3 * ((size_t)i * nmol + m), *b = com + 3 * ((size_t)j * nmol + m);
			const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
			len2 += dx * dx + dy * dy + dz * dz;
		}
		len2 *= (mol_mass[m] * AMU2KG) * (ANGSTROM2METER * ANGSTROM2METER); // :961
		sum += len2;
	}
	return sum;
}
// PI_calculate_kinetic, PathIntegral.cpp:806-824
extern "C" double mpmc_pi_kinetic(double chain_mass_len2, double orient_mu_len2, double N, int nP, double T) {
	const double kB = 1.3806503e-23, hBar2 = 1.11211999e-68; // src/constants.h:17,20
	(void)orient_mu_len2; // computed but not used by the reference's estimator (:815)
	const double d = 3.0, P = (double)nP;
	const double beta = 1.0 / (kB * T);
	const double omega2 = P / (beta * beta * hBar2);
	const double t1 = 0.5 * d * N * kB * T * P;
	const double t2 = 0.5 * omega2 * chain_mass_len2;
	return (1.0 / kB) * (t1 - t2);
}

// ---- measurement -----------------------------------------------------------------------------------------
extern "C" int mpmc_set_profiling(mpmc_ctx *c, int enabled) {
	if (!c) return MPMC_ERR_ARG;
	c->prof = enabled != 0;
	return MPMC_OK;
}
extern "C" int mpmc_get_timings(mpmc_ctx *c, mpmc_timings *out, int reset) {
	if (!c || !out) return MPMC_ERR_ARG;
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	prof_harvest(c);
	*out = c->tim;
	if (reset) std::memset(&c->tim, 0, sizeof(c->tim));
	return MPMC_OK;
}
extern "C" int mpmc_synchronize(mpmc_ctx *c) {
	if (!c) return MPMC_ERR_ARG;
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	return MPMC_OK;
}
extern "C" int mpmc_get_tile_stats(mpmc_ctx *c, int64_t out4[4]) {
	if (!c || !out4) return MPMC_ERR_ARG;
	if (!c->atoms_set || !c->d_cls) return fail(c, MPMC_ERR_ARG, "mpmc_get_tile_stats: no evaluation has run");
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	std::vector<int> cls((size_t)c->n_tile_pairs);
	HIP_TRY(c, hipMemcpy(cls.data(), c->d_cls, cls.size() * sizeof(int), hipMemcpyDeviceToHost));
	out4[0] = c->n_tile_pairs;
	out4[1] = out4[2] = out4[3] = 0;
	for (int v : cls) {
		if (v & CLS_THOLE_FAR) out4[2]++;
		else out4[1]++;
		if (v & CLS_BEYOND_CUTOFF) out4[3]++;
	}
	return MPMC_OK;
}

extern "C" int mpmc_memory_usage(mpmc_ctx *c, int64_t *total, int64_t *tensor) {
	if (!c) return MPMC_ERR_ARG;
	if (total) *total = c->bytes_total;
	if (tensor) *tensor = (c->solver_used == MPMC_SOLVER_COMPACT) ? (int64_t)(c->cap_ab * sizeof(double2)) : 0;
	return MPMC_OK;
}
