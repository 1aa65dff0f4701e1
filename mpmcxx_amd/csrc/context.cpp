// context.cpp -- host side of libmpmc_energy.so: the C ABI of include/mpmc_energy.h.
//
// One mpmc_ctx = the device-resident state of one reference `System` (one box / one PI bead):
// its own HIP stream, struct-of-arrays atom buffers, k-vector tables, work buffers and result scalars.
// Replaces, for the energy path only, the per-System pair lists (reference src/System.Pairs.cpp:21) and the
// A matrix (src/System.cpp:1430-1473).  There is no CPU fallback anywhere in this file.
#include "context.h"

using namespace mpmc;

thread_local std::string mpmc::g_create_error;

// ---- library --------------------------------------------------------------------------------------------
extern "C" int mpmc_abi_version(void) { return MPMC_ABI_VERSION; }

extern "C" int mpmc_device_count(int *count) {
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (count) *count = (e == hipSuccess) ? n : 0;
	return (e == hipSuccess) ? MPMC_OK : MPMC_ERR_NO_DEVICE;
}

// (ABI 6) what a host program without its own HIP binding needs around the path: a fence over everything this process enqueued on a
// device (the timing bracket of a benchmark, the reference's MPI_Barrier companion) and the device's marketing name for its log
extern "C" int mpmc_device_synchronize(int device) {
	if (hipSetDevice(device) != hipSuccess) return MPMC_ERR_NO_DEVICE;
	return hipDeviceSynchronize() == hipSuccess ? MPMC_OK : MPMC_ERR_HIP;
}
extern "C" int mpmc_device_name(int device, char *name, int capacity) {
	if (!name || capacity < 1) return MPMC_ERR_ARG;
	name[0] = 0;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MPMC_ERR_NO_DEVICE;
	std::snprintf(name, (size_t)capacity, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
	return MPMC_OK;
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

// per-device result of the lane-rotation self-test (0 unknown, 1 ok): every symmetric kernel rotates its j-side accumulators with
// v_mov_b32_dpp wave_rol:1; a device on which that does not deliver lane (l + 1) & 63 cannot run this library
static std::atomic<int> g_rot_ok[64];
static int rot_selftest(mpmc_ctx *c) {
	if (c->device < 64 && g_rot_ok[c->device].load(std::memory_order_acquire)) return MPMC_OK;
	int *d = nullptr, h[64];
	HIP_TRY(c, hipMalloc((void **)&d, 64 * sizeof(int)));
	launch_rot_selftest(c->stream, d);
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	(void)hipFree(d);
	for (int l = 0; l < 64; l++)
		if (h[l] != ((l + 1) & 63)) {
			c->err = "lane-rotation self-test failed (v_mov_b32_dpp wave_rol:1): not a gfx950 device?";
			return MPMC_ERR_INTERNAL;
		}
	if (c->device < 64) g_rot_ok[c->device].store(1, std::memory_order_release);
	return MPMC_OK;
}

static mpmc_tuning g_tuning_default; // what contexts created from now on start from (mpmc_debug_configure with a null context)
static std::mutex g_tuning_mu;       // ... written by mpmc_debug_configure(NULL, ...) and copied by mpmc_ctx_create on any thread

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
	{
		std::lock_guard<std::mutex> lk(g_tuning_mu);
		c->tune = g_tuning_default;
	}
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
	    (!c->tune.lazy_side_stream && hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) ||
	    hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
		delete c;
		return fail(nullptr, MPMC_ERR_HIP, "mpmc_ctx_create: hipStreamCreate failed");
	}
	c->two_streams = (c->tune.stream_mode != 0);
	const size_t P = (size_t)c->max_pad;
	A(dev_alloc(c, &c->d_atoms_blob, P * kAtomRecordBytes)); // every per-atom array, one block (layout: atom_block_layout)
	if (rc == MPMC_OK)
		atom_block_layout(c->d_atoms_blob, P, [c](double4 *xyzq, double2 *lj, int2 *mf, double *al, double *ep, double *imm, int32_t *perm, int32_t *slot) {
			c->d_xyzq = xyzq, c->d_lj = lj, c->d_mf = mf, c->d_alpha = al, c->d_eps = ep, c->d_inv_molmass = imm, c->d_perm = perm, c->d_slot_of = slot;
		});
	A(dev_alloc(c, &c->d_tile_bounds, 12 * (P / kTile)));
	A(dev_alloc(c, &c->d_scal, (size_t)S_COUNT + (size_t)C_COUNT)); // scalars and counts share one buffer: one clear, one read-back
	if (rc == MPMC_OK) c->d_cnt = reinterpret_cast<long long *>(c->d_scal + S_COUNT);
	A(dev_alloc(c, &c->d_flag, (size_t)4)); // [0]: Gauss-Seidel's per-sweep flag; [1..3]: iteration control of the precision-terminated Jacobi solve
	A(dev_alloc(c, &c->d_counter, (size_t)1));
	A(dev_alloc(c, &c->d_atom_part, kAtomTermScratch));
	A(dev_alloc(c, &c->d_erf_tab, (size_t)kErfTableDouble2));
	if (rc == MPMC_OK) { // the erfc table of the pair sweep: 24 KB, once per context
		std::vector<double2> tab(kErfTableDouble2);
		erfc_table_device_layout(tab.data());
		if (hipMemcpyAsync(c->d_erf_tab, tab.data(), tab.size() * sizeof(double2), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
		    hipStreamSynchronize(c->stream) != hipSuccess)
			rc = MPMC_ERR_HIP;
	}
	static_assert(sizeof(long long) == sizeof(double), "scalars and counts share one buffer");
	if (rc == MPMC_OK && pinned_alloc(&c->h_scal, (S_COUNT + C_COUNT + 1) * sizeof(double)) != hipSuccess) rc = MPMC_ERR_HIP;
	if (rc == MPMC_OK) {
		c->h_cnt = reinterpret_cast<long long *>(c->h_scal + S_COUNT);
		std::memset(c->h_scal, 0, (S_COUNT + C_COUNT + 1) * sizeof(double)); // (the launch-number slot the waits poll starts at 0: a recycled pinned block may hold an old context's 1.0)
	}
	if (rc == MPMC_OK && pinned_alloc(&c->h_flag, 4 * sizeof(int)) != hipSuccess) rc = MPMC_ERR_HIP;
	if (rc == MPMC_OK) rc = rot_selftest(c);
	if (rc != MPMC_OK) {
		g_create_error = "mpmc_ctx_create: device allocation failed: " + c->err;
		mpmc_ctx_destroy(c);
		return rc;
	}
	*out = c;
	return MPMC_OK;
}

extern "C" int mpmc_ctx_destroy(mpmc_ctx *c) {
	if (!c) return MPMC_ERR_ARG;
	(void)hipSetDevice(c->device);
	if (c->stream2) (void)hipStreamSynchronize(c->stream2);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
	if (c->ev_join) (void)hipEventDestroy(c->ev_join);
	if (c->stream2) (void)hipStreamDestroy(c->stream2);
	for (auto &e : c->ev_used) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
	for (auto &e : c->ev_free) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
	void *ptrs[] = {c->d_atoms_blob, c->d_atom_part, c->d_tile_pairs, c->d_block_part, c->d_block_cnt, c->d_scal,
	                c->d_flag, c->d_counter, c->d_kvec, c->d_kw, c->d_sf, c->d_w_en, c->d_e_recip_part, c->d_part, c->d_e_static, c->d_mu[0], c->d_mu[1],
	                c->d_e_induced, c->d_rrms, c->d_arows, c->d_adense, c->d_ab, c->d_cls, c->d_tp_shift, c->d_lvec, c->d_sf_part, c->d_tile_bounds, c->d_panels, c->d_seg, c->d_arrive, c->d_gpart, c->d_trace, c->d_mv_blob, c->d_moved_idx,
	                c->d_sf_trial, c->d_delta_out, c->d_e_real, c->d_e_real_trial, c->d_dk_part, c->d_gs_ul, c->d_gs_blocks, c->d_erf_tab, c->d_sweep_blocks, c->d_generic_list};
	for (void *p : ptrs)
		if (p) (void)hipFree(p);
	if (c->h_stage) (void)pinned_free(c->h_stage);
	if (c->h_xyzq) (void)pinned_free(c->h_xyzq);
	if (c->ev_xyzq) (void)hipEventDestroy(c->ev_xyzq);
	if (c->h_kstage) (void)pinned_free(c->h_kstage);
	if (c->ev_kstage) (void)hipEventDestroy(c->ev_kstage);
	if (c->static_cnt) (void)pinned_free(c->static_cnt);
	if (c->ev_stage) (void)hipEventDestroy(c->ev_stage);
	if (c->h_scal) (void)pinned_free(c->h_scal);
	if (c->h_flag) (void)pinned_free(c->h_flag);
	if (c->h_delta_out) (void)pinned_free(c->h_delta_out);
	if (c->h_mv_blob) (void)pinned_free(c->h_mv_blob);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
	return MPMC_OK;
}

// ---- box / options ---------------------------------------------------------------------------------------
extern "C" int mpmc_set_box(mpmc_ctx *c, const double basis[9], const double *reciprocal, double volume, double cutoff) {
	if (!c || !basis) return MPMC_ERR_ARG;
	// the same cell again (a caller that hands the box over before every evaluation): nothing to invalidate
	double in[20];
	std::memcpy(in, basis, 9 * sizeof(double));
	if (reciprocal) std::memcpy(in + 9, reciprocal, 9 * sizeof(double));
	else std::memset(in + 9, 0, 9 * sizeof(double));
	in[18] = volume;
	in[19] = cutoff;
	if (c->box_set && c->box_in_has_recip == (reciprocal != nullptr) && std::memcmp(in, c->box_in, sizeof(in)) == 0) return MPMC_OK;
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
	std::memcpy(c->box_in, in, sizeof(in)); // (only a call that was accepted is remembered)
	c->box_in_has_recip = (reciprocal != nullptr);
	c->k_dirty = true;
	c->static_dirty = true;
	c->static_gen++;
	// (No re-sort, no upload of the atoms: the spatial order is a locality heuristic and a new cell leaves it as good as the positions
	// leave it -- a volume move scales both together; atoms that really wander are caught by the drift check of mpmc_update_positions.
	// The fractional origin of the last sort stays valid too: it only says where the tile bounds cut the periodic wrap.)
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
	{ // Gauss-Seidel sweeps run in the reference's atom order (System.Energy.cpp:3569): whenever "this evaluation sweeps in atom order"
	  // changes -- through polar_gs, polarization or rd_only, or on the first options after an upload under the defaults -- re-upload
		auto atom_order = [](const mpmc_options &q) { return q.polar_gs && q.polarization && !q.rd_only; };
		const bool was = c->opts_set && atom_order(c->opts);
		if (was != (bool)atom_order(*o)) c->atoms_dirty = c->atoms_dirty_order = true;
	}
	c->opts = *o;
	c->opts_set = true;
	c->k_dirty = true;
	c->static_dirty = true;
	c->static_gen++;
	c->cache_valid = false;
	return MPMC_OK;
}

// ---- atoms -----------------------------------------------------------------------------------------------
// nested bisection sort of the wrapped fractional coordinates: nx slabs in x, ny strips in y per slab, z order inside a
// strip; consecutive groups of 64 slots (tiles) are then roughly cubic cells.  Pure host code, O(N log N).
constexpr double kResortDrift = 2.0; // Angstrom; a tile is ~16 A wide at liquid density
constexpr int kSortGridMinTiles = 100; // the aligned-grid order from this many tiles on (about 6400 atoms)
static void compute_spatial_order(mpmc_ctx *c) {
	const int n = c->n;
	c->perm.resize(n);
	c->slot_of.resize(n);
	for (int i = 0; i < n; i++) c->perm[i] = i;
	bool enable = c->box_set && n > 2 * kTile;
	if (c->opts_set && c->opts.polar_gs && c->opts.polarization && !c->opts.rd_only) enable = false; // the sweep order IS the atom order (:3569)
	if (c->tune.no_sort) enable = false;
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
		// The aligned grid (round 4).  A dimension costs the Jacobi walk and the pair sweep three instructions per pair when the tile pair has
		// no common periodic image in it, i.e. when tile A's interval meets the half-period image of tile B's.  With count-based boundaries
		// (the nested bisection below) that happens for a fraction (len_A + len_B) / L of the tile pairs -- 1.38 dimensions per far pair at the
		// benchmark box.  Slabs and strips on a FIXED grid of the fractional coordinates with an EVEN number of cells per dimension map onto
		// themselves under a shift by half a period, so an interval meets the image of exactly one other: 0.99 dimensions per far pair, -2.0 %
		// instructions in the contraction, -3.7 % in the sweep, +1.4 % evaluations/s with 32 beads in flight (profiles/r04_sort_grid.txt).
		// The columns are walked in serpentine order, z up one and down the next, so that a tile that runs over the end of a column
		// continues in the neighbouring one at the same z edge and stays compact.  Cells per dimension: 2 ceil(|b_d| / 2e) for the edge e of a
		// cube of one tile's volume; used when every column holds at least two tiles (below that the bisection's cubes are better).
		int grid_x = c->tune.sort_nx, grid_y = c->tune.sort_ny;
		if ((grid_x <= 0 || grid_y <= 0) && c->tune.sort_grid != 0) {
			const int T = (n + kTile - 1) / kTile;
			const double e = std::cbrt(std::fabs(c->box.volume) * (double)kTile / (double)n);
			double len[2];
			for (int p = 0; p < 2; p++) len[p] = std::sqrt(c->box.b[3 * p] * c->box.b[3 * p] + c->box.b[3 * p + 1] * c->box.b[3 * p + 1] + c->box.b[3 * p + 2] * c->box.b[3 * p + 2]);
			const int ax = 2 * std::max(1, (int)std::ceil(len[0] / (2.0 * e))), ay = 2 * std::max(1, (int)std::ceil(len[1] / (2.0 * e)));
			// (measured over sizes, profiles/r04_sort_grid.txt: +1.4 to +1.7 % evaluations/s in flight at 7000 and 10 000 atoms, +1 % at 20 000, but -4 % at
			// 4000 atoms, where four cells per dimension leave the bisection's tiles aligned already: from kSortGridMinTiles tiles on)
			if (e > 0.0 && T >= kSortGridMinTiles && (long long)ax * ay * 2 <= T) grid_x = ax, grid_y = ay;
		}
		if (grid_x > 0 && grid_y > 0) {
			const int gx = grid_x, gy = grid_y;
			std::vector<long long> key((size_t)n);
			std::vector<double> zk((size_t)n);
			for (int i = 0; i < n; i++) {
				int sx = std::min(gx - 1, (int)(f[3 * (size_t)i] * gx)), sy = std::min(gy - 1, (int)(f[3 * (size_t)i + 1] * gy));
				if (sx & 1) sy = gy - 1 - sy;
				const long long col = (long long)sx * gy + sy;
				key[i] = col;
				zk[i] = (col & 1) ? -f[3 * (size_t)i + 2] : f[3 * (size_t)i + 2];
			}
			std::sort(c->perm.begin(), c->perm.end(), [&](int a, int b) {
				if (key[a] != key[b]) return key[a] < key[b];
				if (zk[a] != zk[b]) return zk[a] < zk[b];
				return a < b;
			});
			for (int k = 0; k < n; k++) c->slot_of[c->perm[k]] = k;
			c->order_sorted = true;
			c->edits_since_sort = 0;
			return;
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
	c->order_sorted = enable;
	c->edits_since_sort = 0;
}

// set_atoms with a list that differs from the one before by ONE contiguous run of atoms (inserted or removed): positions before the run
// and behind it are bitwise those of the old list.  Brings perm / slot_of up to date without sorting; false = not such an edit.
static bool carry_spatial_order(mpmc_ctx *c, const double *new_pos, int n_new) {
	const int n_old = (int)c->perm.size();
	if (!c->order_sorted || c->atoms_dirty_order || n_old == 0 || (int)c->h_pos.size() != 3 * n_old) return false;
	if (n_new <= 2 * kTile || n_new == n_old) return false; // (small systems keep the identity order; same size = not an insertion / removal)
	const int diff = n_new - n_old, m = diff > 0 ? diff : -diff, lo = std::min(n_old, n_new);
	if (m > kTile || c->edits_since_sort + m > kTile) return false; // time for a real sort
	const double *old_pos = c->h_pos.data();
	int p = 0;
	while (p < lo && old_pos[3 * p] == new_pos[3 * p] && old_pos[3 * p + 1] == new_pos[3 * p + 1] && old_pos[3 * p + 2] == new_pos[3 * p + 2]) p++;
	int s = 0;
	while (s < lo - p && old_pos[3 * (n_old - 1 - s)] == new_pos[3 * (n_new - 1 - s)] && old_pos[3 * (n_old - 1 - s) + 1] == new_pos[3 * (n_new - 1 - s) + 1] &&
	       old_pos[3 * (n_old - 1 - s) + 2] == new_pos[3 * (n_new - 1 - s) + 2])
		s++;
	if (p + s != lo) return false; // more than one run changed (or atoms moved as well): sort
	std::vector<int32_t> np_;
	np_.reserve(n_new);
	for (int k = 0; k < n_old; k++) {
		const int i = c->perm[k];
		if (i < p) np_.push_back(i);
		else if (i >= n_old - s) np_.push_back(i + diff);
		// else: removed
	}
	for (int i = p; i < n_new - s; i++) np_.push_back(i); // inserted atoms: appended (the last tile loses some locality until the next sort)
	if ((int)np_.size() != n_new) return false;
	c->perm.swap(np_);
	c->slot_of.assign(n_new, -1);
	for (int k = 0; k < n_new; k++) c->slot_of[c->perm[k]] = k;
	c->edits_since_sort += m;
	return true;
}

int mpmc::upload_atoms(mpmc_ctx *c) {
	// (atoms_dirty_order: somebody asked for a NEW order since the list was carried -- a set_options that switches Gauss-Seidel sweeps on
	// needs the identity order of System.Energy.cpp:3569, not the carried spatial one)
	if (!c->order_carried || c->atoms_dirty_order || c->tune.no_order_carry || (int)c->perm.size() != c->n) {
		compute_spatial_order(c);
		c->n_uploads_sorted++;
	} else {
		c->n_uploads_carried++;
	}
	c->order_carried = false;
	c->atoms_dirty_order = false;
	const int n = c->n, np = c->n_pad;
	// One persistent pinned staging block for all per-atom arrays: the copies below are asynchronous for real (from pageable vectors
	// every one of them was a staged, blocking copy, and a stream synchronisation kept the vectors alive) -- an insertion or removal
	// (uVT, Gibbs) pays for a sort and eight enqueues here, nothing else.
	const size_t P = (size_t)c->max_pad;
	if (!c->h_stage) {
		HIP_TRY(c, pinned_alloc(&c->h_stage, P * kAtomRecordBytes));
		HIP_TRY(c, pinned_alloc(&c->h_xyzq, P * sizeof(double4)));
		HIP_TRY(c, hipEventCreateWithFlags(&c->ev_xyzq, hipEventDisableTiming));
		HIP_TRY(c, hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming));
		HIP_TRY(c, pinned_alloc(&c->static_cnt, 4 * sizeof(long long)));
		for (int k = 0; k < 4; k++) c->static_cnt[k] = 0;
	}
	if (c->stage_in_flight) { // (an upload per evaluation at most, and evaluations are waited for: normally long done)
		HIP_TRY(c, hipEventSynchronize(c->ev_stage));
		c->stage_in_flight = false;
	}
	double4 *xyzq = nullptr;
	double2 *lj = nullptr;
	int2 *mf = nullptr;
	double *al = nullptr, *ep = nullptr, *imm = nullptr;
	int32_t *perm = nullptr, *slot = nullptr;
	atom_block_layout(c->h_stage, P, [&](double4 *a0, double2 *a1, int2 *a2, double *a3, double *a4, double *a5, int32_t *a6, int32_t *a7) {
		xyzq = a0, lj = a1, mf = a2, al = a3, ep = a4, imm = a5, perm = a6, slot = a7;
	});
	c->molmass_tmp.assign(n, 0.0); // Molecule::mass = sum of its atoms' masses (System.cpp:687), per atom
	std::vector<double> &molmass = c->molmass_tmp;
	if (!c->h_mass.empty())
		for (int i0 = 0; i0 < n;) {
			int i1 = i0;
			double m = 0;
			while (i1 < n && c->h_mol[i1] == c->h_mol[i0]) m += c->h_mass[i1++];
			for (int i = i0; i < i1; i++) molmass[i] = m;
			i0 = i1;
		}
	for (int k = 0; k < np; k++) slot[k] = -1;
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
			perm[k] = -1;
			al[k] = ep[k] = imm[k] = 0.0;
			xyzq[k] = make_double4(0, 0, 0, 0);
			lj[k] = make_double2(0, 0);
			mf[k] = make_int2(-1 - k, AF_PAD | AF_FROZEN | AF_NULL_RD | AF_ZERO_SIGMA | AF_ZERO_Q | AF_ZERO_ALPHA);
		}
	}
	{ // tile pairs with an atom whose flags change lj_mix (sigma < 0, dispersion coefficients) are the generic pair kernel's (the sweep masks
	  // everything else itself): their list, in tile-pair order
		constexpr int kSpecial = kAtomFlagsMixing; // (pair_math.h: the same constant the sweep skips tile pairs by)
		const int nt = c->n_tiles;
		std::vector<char> special((size_t)nt, 0);
		bool any = false;
		for (int k = 0; k < n; k++)
			if (mf[k].y & kSpecial) special[k / kTile] = 1, any = true;
		c->h_generic.clear();
		if (any)
			for (int I = 0, t = 0; I < nt; I++)
				for (int J = I; J < nt; J++, t++)
					if (special[I] || special[J]) c->h_generic.push_back(t);
		c->n_generic = (int)c->h_generic.size();
		if (c->n_generic > 0) {
			if ((size_t)c->n_generic > c->cap_generic) {
				dev_free(c, &c->d_generic_list, c->cap_generic);
				c->cap_generic = 0;
				const int rc_a = dev_alloc(c, &c->d_generic_list, (size_t)c->n_tile_pairs);
				if (rc_a != MPMC_OK) return rc_a;
				c->cap_generic = (size_t)c->n_tile_pairs;
			}
			HIP_TRY(c, hipMemcpyAsync(c->d_generic_list, c->h_generic.data(), (size_t)c->n_generic * sizeof(int), hipMemcpyHostToDevice, c->stream));
			HIP_TRY(c, hipStreamSynchronize(c->stream)); // pageable source that the next upload clears: wait, as the tile-pair and block tables do
		}
	}
	// ONE copy: the device block has the layout of the staging block (the tails beyond n_pad travel along; nobody reads them)
	if (np < (int)P && (size_t)np * 2 >= P)
		for (size_t k = (size_t)np; k < P; k++) { // (the tails travel along with the single copy: defined values)
			xyzq[k] = make_double4(0, 0, 0, 0);
			lj[k] = make_double2(0, 0);
			mf[k] = make_int2(-1 - (int)k, AF_PAD | AF_FROZEN | AF_NULL_RD | AF_ZERO_SIGMA | AF_ZERO_Q | AF_ZERO_ALPHA);
			al[k] = ep[k] = imm[k] = 0.0;
			perm[k] = slot[k] = -1;
		}
	if ((size_t)np * 2 >= P) {
		HIP_TRY(c, hipMemcpyAsync(c->d_atoms_blob, c->h_stage, P * kAtomRecordBytes, hipMemcpyHostToDevice, c->stream));
	} else { // a context created with much more room than atoms: the used prefix of every array instead of the whole block
		HIP_TRY(c, hipMemcpyAsync(c->d_xyzq, xyzq, np * sizeof(double4), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->d_lj, lj, np * sizeof(double2), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->d_mf, mf, np * sizeof(int2), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->d_alpha, al, np * sizeof(double), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->d_eps, ep, np * sizeof(double), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->d_inv_molmass, imm, np * sizeof(double), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->d_perm, perm, np * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->d_slot_of, slot, np * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
	}
	HIP_TRY(c, hipEventRecord(c->ev_stage, c->stream));
	c->stage_in_flight = true;
	// position-independent pair-flag counts (diagnostics of pair_exclusions), once per upload; they arrive in pinned memory in front of
	// the evaluation that follows on this stream, and are read when that evaluation has been waited for
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
		c->scal_clean = false; // (d_cnt is part of the scalar block: the next evaluation clears it instead of trusting which kernel overwrites what)
		HIP_TRY(c, hipGetLastError());
		HIP_TRY(c, hipMemcpyAsync(c->static_cnt, c->d_cnt, 4 * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
	}
	{ // slot-ordered mirror for later position updates
		const int rc_g = mirror_guard(c);
		if (rc_g != MPMC_OK) return rc_g;
		std::memcpy(c->h_xyzq, xyzq, (size_t)np * sizeof(double4));
	}
	c->h_pos_sorted = c->h_pos;        // where every atom stood when this order was made
	c->atoms_dirty = false;
	return MPMC_OK;
}

// A context that has to hold more atoms than it was created for is rebuilt in place: a fresh context of the larger capacity takes
// over the caller's handle (same device, box, options, profiling state), the old device buffers are released.  Everything sized by
// the capacity is allocated on first use, so nothing else has to know.
static int grow_capacity(mpmc_ctx *c, int n) {
	(void)hipSetDevice(c->device);
	if (c->stream2) (void)hipStreamSynchronize(c->stream2);
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
	f->tune = c->tune;
	f->n_poll_hits = c->n_poll_hits, f->n_poll_timeouts = c->n_poll_timeouts, f->n_stream_syncs = c->n_stream_syncs, f->n_poll_yields = c->n_poll_yields;
	f->n_uploads_carried = c->n_uploads_carried; // (diagnostics survive the growth; the order itself does not: the new context sorts)
	f->n_uploads_sorted = c->n_uploads_sorted;
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
	// (before the old positions go: is this the list of before with one run of atoms inserted or removed?  Then the spatial order is carried)
	c->order_carried = carry_spatial_order(c, pos, n);
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
	c->static_dirty = true;
	c->static_gen++;

	// upper-triangular tile-pair schedule of the pair kernel
	const int nt = c->n_tiles;
	const size_t ntp = (size_t)nt * (nt + 1) / 2;
	if (ntp > c->cap_tile_pairs) {
		dev_free(c, &c->d_tile_pairs, c->cap_tile_pairs);
		dev_free(c, &c->d_block_part, 2 * c->cap_tile_pairs);
		dev_free(c, &c->d_block_cnt, 4 * c->cap_tile_pairs);
		dev_free(c, &c->d_cls, c->cap_tile_pairs);
		dev_free(c, &c->d_tp_shift, c->cap_tile_pairs);
		c->cap_tile_pairs = 0;
		if ((rc = dev_alloc(c, &c->d_tile_pairs, ntp)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_block_part, 2 * ntp)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_block_cnt, 4 * ntp)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_cls, ntp)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_tp_shift, ntp)) != MPMC_OK) return rc;
		c->cap_tile_pairs = ntp;
	}
	std::vector<int2> tp;
	tp.reserve(ntp);
	for (int I = 0; I < nt; I++)
		for (int J = I; J < nt; J++) tp.push_back(make_int2(I, J));
	HIP_TRY(c, hipMemcpyAsync(c->d_tile_pairs, tp.data(), ntp * sizeof(int2), hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream)); // `tp` dies with this function
	c->n_tile_pairs = (int)ntp;

	if (c->sweep_tiles != nt) { // work table of the fast pair sweep: a function of the tile count
		const int nb = pair_sweep_blocks(nt, nullptr);
		std::vector<int2> blocks((size_t)nb);
		pair_sweep_blocks(nt, blocks.data());
		if (c->tune.sweep_order == 1) std::reverse(blocks.begin(), blocks.end()); // j-tiles descending (measurement)
		if ((size_t)nb > c->cap_sweep_blocks) {
			dev_free(c, &c->d_sweep_blocks, c->cap_sweep_blocks);
			c->cap_sweep_blocks = 0;
			if ((rc = dev_alloc(c, &c->d_sweep_blocks, (size_t)nb)) != MPMC_OK) return rc;
			c->cap_sweep_blocks = (size_t)nb;
		}
		HIP_TRY(c, hipMemcpyAsync(c->d_sweep_blocks, blocks.data(), (size_t)nb * sizeof(int2), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream)); // `blocks` dies here
		c->n_sweep_blocks = nb;
		c->sweep_tiles = nt;
	}

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
			c->atoms_dirty = c->atoms_dirty_order = true;
			return MPMC_OK;
		}
	}
	// the atoms keep their slots; their new positions go through the pinned slot-ordered mirror: one asynchronous copy of the whole array
	// (a handful of atoms: one small copy each), an event behind it.  No stream synchronisation: the next writer of the mirror waits
	// for this copy (mirror_guard), nobody waits for the evaluations that may be queued in front of it.
	{
		const int rc_g = mirror_guard(c);
		if (rc_g != MPMC_OK) return rc_g;
	}
	for (int t = 0; t < count; t++) {
		const int i = first + t;
		c->h_xyzq[c->slot_of[i]] = make_double4(pos[3 * t], pos[3 * t + 1], pos[3 * t + 2], c->h_q[i]);
	}
	if (count > 4) {
		HIP_TRY(c, hipMemcpyAsync(c->d_xyzq, c->h_xyzq, (size_t)c->n_pad * sizeof(double4), hipMemcpyHostToDevice, c->stream));
	} else {
		for (int t = 0; t < count; t++) {
			const int k = c->slot_of[first + t];
			HIP_TRY(c, hipMemcpyAsync(c->d_xyzq + k, c->h_xyzq + k, sizeof(double4), hipMemcpyHostToDevice, c->stream));
		}
	}
	HIP_TRY(c, hipEventRecord(c->ev_xyzq, c->stream));
	c->xyzq_in_flight = true;
	return MPMC_OK;
}

extern "C" int mpmc_set_positions_device(mpmc_ctx *c, const double *pos_device) {
	if (!c || !pos_device) return MPMC_ERR_ARG;
	if (!c->atoms_set) return fail(c, MPMC_ERR_ARG, "mpmc_set_positions_device: no atoms set");
	if (c->trial_open) return fail(c, MPMC_ERR_ARG, "mpmc_set_positions_device: a trial move is open (accept or reject it first)");
	if (c->pending) return fail(c, MPMC_ERR_ARG, "mpmc_set_positions_device: an evaluation is in flight (mpmc_energy_wait first)");
	c->cache_valid = false; // the accepted totals / structure factors no longer describe the resident configuration
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
	{
		const int rc_g = mirror_guard(c);
		if (rc_g != MPMC_OK) return rc_g;
	}
	for (int i = 0; i < c->n; i++) { // ... and the slot-ordered mirror a later mpmc_update_positions uploads as a whole
		double4 &v = c->h_xyzq[c->slot_of[i]];
		v.x = c->h_pos[3 * (size_t)i], v.y = c->h_pos[3 * (size_t)i + 1], v.z = c->h_pos[3 * (size_t)i + 2];
	}
	return MPMC_OK;
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
	HIP_TRY(c, hipMemcpyAsync(cls.data(), c->d_cls, cls.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	out4[0] = c->n_tile_pairs;
	out4[1] = out4[2] = out4[3] = 0;
	for (int v : cls) {
		if (v & CLS_THOLE_FAR) out4[2]++;
		else out4[1]++;
		if (v & CLS_BEYOND_CUTOFF) out4[3]++;
	}
	return MPMC_OK;
}

// measurement (bench.py's roofline): exact ATOM-pair counts behind the tile-pair classes of the last evaluation.  out[12] =
//   0 all pairs N(N-1)/2      1 pairs of the tile pairs whose tensors are stored      2 ... of the far-field tile pairs      3 ... of the
//   tile pairs beyond the cutoff      4 / 5  sum over the stored / far tile pairs of pairs x (dimensions WITHOUT a tile-pair-wide image)
//   6 pairs the pair sweep walks (not beyond the cutoff, or stored)      7 the same sum of pairs x non-uniform dimensions over those
//   8 tile pairs      9 stored      10 far      11 beyond the cutoff
extern "C" int mpmc_debug_pair_stats(mpmc_ctx *c, int64_t out[12]) {
	if (!c || !out) return MPMC_ERR_ARG;
	if (!c->atoms_set || !c->d_cls) return fail(c, MPMC_ERR_ARG, "mpmc_debug_pair_stats: no evaluation has run");
	HIP_TRY(c, hipSetDevice(c->device));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	std::vector<int> cls((size_t)c->n_tile_pairs);
	HIP_TRY(c, hipMemcpyAsync(cls.data(), c->d_cls, cls.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	for (int k = 0; k < 12; k++) out[k] = 0;
	const int nt = c->n_tiles;
	const bool uni = !(c->tune.no_uniform || c->tune.no_classes);
	auto real = [&](int T) { return (int64_t)std::max(0, std::min(kTile, c->n - T * kTile)); };
	size_t t = 0;
	for (int I = 0; I < nt; I++)
		for (int J = I; J < nt; J++, t++) {
			const int v = cls[t];
			const int64_t pairs = (I == J) ? real(I) * (real(I) - 1) / 2 : real(I) * real(J);
			const int um = uni ? ((v / CLS_UNIFORM_X) & 7) : 0;
			const int64_t nu = 3 - ((um & 1) + ((um >> 1) & 1) + ((um >> 2) & 1));
			const bool far = (v & CLS_THOLE_FAR) != 0, beyond = (v & CLS_BEYOND_CUTOFF) != 0;
			out[0] += pairs;
			out[far ? 2 : 1] += pairs;
			out[far ? 5 : 4] += pairs * nu;
			if (beyond) out[3] += pairs;
			if (!beyond || !far) {
				out[6] += pairs;
				out[7] += pairs * nu;
			}
			out[8]++;
			out[far ? 10 : 9]++;
			if (beyond) out[11]++;
		}
	return MPMC_OK;
}

// Measurement / A-B switches (struct mpmc_tuning, context.h).  With a context: that context, from its next evaluation on; with a null
// context: the default that contexts created afterwards in this process start from.  The library reads no environment variable for
// any of this.  Keys (value 1 = on, 0 = off unless said otherwise):
//   side_stream -1 | 0 | 1     pair_kernel 0 | 1 | 2     pair_waves 0 | 1 | 4     panels     uniform_images     tile_classes
//   single_launch     recip_table     spatial_sort     order_carry     polar_delta     inline_move     trace_panel     tensor_budget_mb N
extern "C" int mpmc_debug_configure(mpmc_ctx *c, const char *key, double value) {
	if (!key) return MPMC_ERR_ARG;
	std::unique_lock<std::mutex> tuning_lk(g_tuning_mu, std::defer_lock);
	if (!c) tuning_lk.lock();
	mpmc_tuning &t = c ? c->tune : g_tuning_default;
	const std::string k(key);
	const int v = (int)value;
	const bool on = (value != 0.0);
	if (k == "side_stream") {
		if (v < -1 || v > 1) return MPMC_ERR_ARG;
		t.stream_mode = v;
	} else if (k == "pair_kernel") {
		if (v < 0 || v > 2) return MPMC_ERR_ARG;
		t.pair_kernel = v;
	} else if (k == "pair_waves") {
		if (v != 0 && v != 1 && v != 4) return MPMC_ERR_ARG;
		t.pair_waves = v;
	} else if (k == "fast_geometry") t.fast_geometry = on;
	else if (k == "dense_symmetric") t.dense_symmetric = on;
	else if (k == "sort_grid") {
		if (v < -1 || v > 0) return MPMC_ERR_ARG;
		t.sort_grid = v;
		if (c) c->atoms_dirty = c->atoms_dirty_order = true;
	} else if (k == "sort_nx") {
		t.sort_nx = v;
		if (c) c->atoms_dirty = c->atoms_dirty_order = true;
	} else if (k == "sort_ny") {
		t.sort_ny = v;
		if (c) c->atoms_dirty = c->atoms_dirty_order = true;
	}
	else if (k == "side_after_sweep") t.side_after_sweep = on;
	else if (k == "poll_long") t.poll_long = on;
	else if (k == "poll_retire") t.poll_retire = on;
	else if (k == "lazy_side_stream") t.lazy_side_stream = on;
	else if (k == "tail_fused") t.tail_fused = on;
	else if (k == "pair_split_tail") {
		if (v < -1 || v > 1000) return MPMC_ERR_ARG;
		t.pair_split_tail = v;
	} else if (k == "pair_split") {
		if (v < -1 || v > 1) return MPMC_ERR_ARG;
		t.pair_split = v;
	} else if (k == "panels") t.use_panels = on;
	else if (k == "fused_update") {
		if (v < 0 || v > 2) return MPMC_ERR_ARG;
		t.fused_update = v;
	}
	else if (k == "panel_reverse") t.panel_reverse = on;
	else if (k == "sweep_order") {
		if (v < 0 || v > 1) return MPMC_ERR_ARG;
		t.sweep_order = v;
		if (c) c->sweep_tiles = -1, c->atoms_dirty = true; // (the table is rebuilt with the next upload)
	} else if (k == "update_waves") {
		if (v != 0 && v != 1 && v != 2 && v != 4 && v != 16) return MPMC_ERR_ARG;
		t.update_waves = v;
	} else if (k == "sweep_lds_pad") {
		if (v < 0 || v > 65536) return MPMC_ERR_ARG;
		t.sweep_lds_pad = v;
	}
	else if (k == "uniform_images") t.no_uniform = !on;
	else if (k == "tile_classes") t.no_classes = !on;
	else if (k == "single_launch") t.single_launch = on;
	else if (k == "recip_table") t.no_recip_tab = !on;
	else if (k == "spatial_sort") {
		t.no_sort = !on;
		if (c) c->atoms_dirty = c->atoms_dirty_order = true; // (the order changes with the next upload)
	} else if (k == "order_carry") t.no_order_carry = !on;
	else if (k == "polar_delta") t.no_polar_delta = !on;
	else if (k == "inline_move") t.no_inline_move = !on;
	else if (k == "trace_panel") t.trace_panel = on;
	else if (k == "virtual_device") {
		if (!c || v < -1 || v > 63) return MPMC_ERR_ARG;
		t.virtual_device = v;
	} else if (k == "fail_next_wait") {
		if (!c) return MPMC_ERR_ARG;
		t.fail_next_wait = on ? 1 : 0;
	}
	else if (k == "panel_replicas") {
		if (!c || v < 1 || v > 64) return MPMC_ERR_ARG;
		c->debug_panel_replicas = v;
	} else if (k == "tensor_budget_mb") {
		if (value < 0) return MPMC_ERR_ARG;
		t.tensor_budget_mb = (long long)value;
	} else return MPMC_ERR_ARG;
	return MPMC_OK;
}
// the last evaluated trial move: 1 = a full evaluation of the trial configuration, 0 = per-move delta energies, -1 = none
extern "C" int mpmc_debug_last_trial_was_full(mpmc_ctx *c) { return (c && c->trial_last_kind >= 0) ? c->trial_last_kind : -1; }
// which kernel ran the pair pass of the last evaluation: 1 the fast sweep, 0 k_pair_fused
extern "C" int mpmc_debug_last_pair_kernel(mpmc_ctx *c) { return c ? (c->last_pair_was_sweep ? 1 : 0) : -1; }

// how this context's host waits ended since it was created: out[4] = polls that saw the device's post, polls that ran out of their
// budget (the wait then synchronised the stream), stream synchronisations, yields taken inside long polls (poll_posted, context.h)
extern "C" int mpmc_debug_wait_counters(mpmc_ctx *c, long long *out4) {
	if (!c || !out4) return -1;
	out4[0] = c->n_poll_hits;
	out4[1] = c->n_poll_timeouts;
	out4[2] = c->n_stream_syncs;
	out4[3] = c->n_poll_yields;
	return 0;
}

// diagnostics only (tests assert that the order was really carried): uploads of the atom list that kept the order / that sorted
extern "C" int mpmc_debug_upload_counts(mpmc_ctx *c, long long *out2) {
	if (!c || !out2) return -1;
	out2[0] = c->n_uploads_carried;
	out2[1] = c->n_uploads_sorted;
	return 0;
}

// measurement only: per-workgroup time stamps of the last panel launch (tools/panel_trace.py); 0 entries unless the context was configured with trace_panel = 1
extern "C" int mpmc_debug_panel_trace(mpmc_ctx *c, long long *out4, int max_entries) {
	if (!c || !out4) return -1;
	if (!c->d_trace) return 0;
	const int n = std::min(max_entries, c->n_panel_entries);
	if (hipSetDevice(c->device) != hipSuccess) return -1;
	if (hipMemcpyAsync(out4, c->d_trace, (size_t)n * 4 * sizeof(long long), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
	if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
	return n;
}

// measurement only: the work table of the panel kernel, { tile pair A, tile pair B or -1, uniform mask | far << 3 | diagonal << 4, J } per workgroup
extern "C" int mpmc_debug_panel_table(mpmc_ctx *c, int *out4, int max_entries) {
	if (!c || !out4) return -1;
	if (!c->d_panels || !c->panels_built) return 0;
	const int n = std::min(max_entries, c->n_panel_entries);
	if (hipSetDevice(c->device) != hipSuccess) return -1;
	if (hipMemcpyAsync(out4, c->d_panels, (size_t)n * 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
	if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
	return n;
}

extern "C" int mpmc_memory_usage(mpmc_ctx *c, int64_t *total, int64_t *tensor) {
	if (!c) return MPMC_ERR_ARG;
	if (total) *total = c->bytes_total;
	if (tensor) *tensor = (c->solver_used == MPMC_SOLVER_COMPACT) ? (int64_t)(c->cap_ab * sizeof(double2)) : 0;
	return MPMC_OK;
}
