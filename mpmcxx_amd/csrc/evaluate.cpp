// evaluate.cpp -- one energy evaluation: k tables, work buffers, the enqueue of every kernel of double System::energy(), result assembly, component entry points
// (part of libmpmc_energy.so; shared state and helpers: context.h.  There is no CPU fallback anywhere in this library.)
#include "context.h"


using namespace mpmc;

// ---- k-vector tables (hemisphere enumeration of coulombic_reciprocal :1577-1590 / recip_term :2849-2865) --------
static int build_k_tables(mpmc_ctx *c) {
	const int kmax = c->opts.ewald_kmax;
	const double alpha = c->ewald_alpha, ea = c->polar_ewald_alpha;
	// the set of l-vectors depends on kmax alone: count it, make room (device tables and ONE persistent pinned staging block), fill the
	// staging block in place and copy asynchronously -- a volume move rebuilds these tables every time, and four blocking copies from
	// pageable vectors plus a stream synchronisation cost more than the reciprocal-space kernels they feed
	int K = 0;
	int l[3];
	for (l[0] = 0; l[0] <= kmax; l[0]++)
		for (l[1] = (!l[0] ? 0 : -kmax); l[1] <= kmax; l[1]++)
			for (l[2] = ((!l[0] && !l[1]) ? 1 : -kmax); l[2] <= kmax; l[2]++)
				if (l[0] * l[0] + l[1] * l[1] + l[2] * l[2] <= kmax * kmax) K++;
	if (K > c->cap_K) {
		dev_free(c, &c->d_kvec, (size_t)c->cap_K);
		dev_free(c, &c->d_kw, (size_t)c->cap_K);
		dev_free(c, &c->d_lvec, (size_t)c->cap_K);
		dev_free(c, &c->d_w_en, (size_t)c->cap_K);
		c->cap_K = 0;
		c->lvec_kmax = -1;
		int rc;
		if ((rc = dev_alloc(c, &c->d_kvec, (size_t)K)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_kw, (size_t)K)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_lvec, (size_t)K)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_w_en, (size_t)K)) != MPMC_OK) return rc;
		c->cap_K = K;
	}
	if (K > c->cap_sf) { // the structure factors swap buffers with the trial ones on accept: sized on their own
		dev_free(c, &c->d_sf, (size_t)c->cap_sf);
		c->cap_sf = 0;
		int rc = dev_alloc(c, &c->d_sf, (size_t)K);
		if (rc != MPMC_OK) return rc;
		c->cap_sf = K;
	}
	if (K > 0) {
		if (c->kstage_in_flight) { // (one rebuild per evaluation at most, and evaluations are waited for: normally long done)
			HIP_TRY(c, hipEventSynchronize(c->ev_kstage));
			c->kstage_in_flight = false;
		}
		if ((size_t)K > c->cap_kstage) {
			if (c->h_kstage) HIP_TRY(c, pinned_free(c->h_kstage));
			c->h_kstage = nullptr;
			c->cap_kstage = 0;
			HIP_TRY(c, pinned_alloc(&c->h_kstage, (size_t)K * (2 * sizeof(double4) + sizeof(double) + sizeof(int4))));
			c->cap_kstage = (size_t)K;
			if (!c->ev_kstage) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_kstage, hipEventDisableTiming));
		}
		// (the 16-byte types first: behind an odd number of doubles an int4 array would be misaligned -- found by tools/host_asan.sh)
		double4 *kvec = reinterpret_cast<double4 *>(c->h_kstage), *kw = kvec + c->cap_kstage;
		int4 *lvec = reinterpret_cast<int4 *>(kw + c->cap_kstage);
		double *wen = reinterpret_cast<double *>(lvec + c->cap_kstage);
		int n = 0;
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
					kvec[n] = make_double4(k[0], k[1], k[2], k2);
					lvec[n] = make_int4(l[0], l[1], l[2], 0);
					wen[n] = std::exp(-k2 / (4.0 * alpha * alpha)) / k2;
					const double g = std::exp(-k2 / (4.0 * ea * ea));
					kw[n] = make_double4(k[0] / k2 * g, k[1] / k2 * g, k[2] / k2 * g, 0.0);
					n++;
				}
		HIP_TRY(c, hipMemcpyAsync(c->d_kvec, kvec, K * sizeof(double4), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->d_kw, kw, K * sizeof(double4), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->d_w_en, wen, K * sizeof(double), hipMemcpyHostToDevice, c->stream));
		if (c->lvec_kmax != kmax) {
			HIP_TRY(c, hipMemcpyAsync(c->d_lvec, lvec, K * sizeof(int4), hipMemcpyHostToDevice, c->stream));
			c->lvec_kmax = kmax;
		}
		HIP_TRY(c, hipEventRecord(c->ev_kstage, c->stream));
		c->kstage_in_flight = true;
	}
	c->K = K;
	return MPMC_OK;
}

constexpr int kDenseChunks = 16; // row chunks (= partial slots) of the dense matrix-vector product
constexpr int kOneStreamMinInflight = 4; // evaluations in flight from which on an evaluation keeps to one stream
constexpr int kCheckEvery = 4;         // precision-terminated Jacobi solve: host reads the device-side verdict once per this many iterations
constexpr int kSingleLaunchTiles = 32; // <= 2048 atoms (528 tile pairs): LJ-only evaluations run as ONE launch

static int ensure_polar_buffers(mpmc_ctx *c) {
	const size_t np = (size_t)c->max_pad;
	int rc;
	if (!c->d_e_static) {
		if ((rc = dev_alloc(c, &c->d_e_static, 3 * np)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_mu[0], 3 * np)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_mu[1], 3 * np)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_e_induced, 3 * np)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_rrms, np)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_e_recip_part, recip_slices_capacity(np))) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_e_real, 3 * np)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_e_real_trial, 3 * np)) != MPMC_OK) return rc;
		if ((rc = dev_alloc(c, &c->d_gs_ul, 6 * np)) != MPMC_OK) return rc; // Gauss-Seidel sweeps: fields of the tiles above / below
		// (dev_alloc zero-fills on the context's stream; nothing in this library touches the null stream, which is unordered against
		// our non-blocking streams)
	}
	// per-atom partial slots: one per source tile (symmetric kernels) -- also covers the n_split <= n_tiles slots of the matrix-free
	// row kernel -- and never fewer than the kDenseChunks row chunks the dense matrix-vector product writes (small systems have fewer
	// tiles than that: the dense solver used to write past the end of this buffer, into the matrix that was allocated right behind it)
	const size_t need = (size_t)std::max(c->n_tiles, kDenseChunks) * c->n_pad * 3;
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
		const size_t budget_mb = (size_t)c->tune.tensor_budget_mb;
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
int mpmc::prepare(mpmc_ctx *c, bool defer_static) {
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
	// pair LRC (O(N) moment form), self LRC, Ewald self term: position independent (lj_lrc_corr / lj_lrc_self :1036-1096, coulombic_self
	// :1626-1643), cached in h_static.  A general evaluation takes them along when they are stale (defer_static: enqueue() launches the
	// kernel behind its clear of the scalar block and wait_and_fill adopts the three values); everybody else gets them here and now.
	if (c->static_dirty && !defer_static) {
		c->scal_clean = false; // (the kernel writes its slots of the scalar block: the next evaluation clears the block first)
		launch_atom_terms(c->stream, atoms_view(c), c->box, c->ewald_alpha, c->opts.rd_lrc, /*self term*/ 1, c->d_atom_part, c->d_scal);
		HIP_TRY(c, hipGetLastError());
		double tmp[S_COUNT];
		HIP_TRY(c, hipMemcpyAsync(tmp, c->d_scal, sizeof(tmp), hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		c->h_static[0] = tmp[S_LRC_PAIR];
		c->h_static[1] = tmp[S_LRC_SELF];
		c->h_static[2] = tmp[S_ES_SELF];
		c->static_dirty = false;
	}
	return MPMC_OK;
}

AtomsDev mpmc::atoms_view(const mpmc_ctx *c) {
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
RecipDev mpmc::recip_view(const mpmc_ctx *c) {
	RecipDev r;
	r.kvec = c->d_kvec;
	r.w_en = c->d_w_en;
	r.kw = c->d_kw;
	r.lvec = c->tune.no_recip_tab ? nullptr : c->d_lvec;
	r.sf = c->d_sf;
	r.K = c->K;
	return r;
}

// the Wolf / Feynman-Hibbs fields of the pair parameters (pair sweep and per-move delta kernels)
void mpmc::ext_params(const mpmc_ctx *c, FusedParams &fp, bool wolf_on) {
	const mpmc_options &o = c->opts;
	fp.wolf = wolf_on ? 1 : 0;
	fp.fh_order = o.feynman_hibbs ? ((o.feynman_hibbs_order == 4) ? 4 : 2) : 0;
	fp.fh_c2 = fp.fh_c4 = 0.0;
	if (fp.fh_order) { // reference constants.h:15-33: M2A2 hBar2 / (24 kB T) and M2A4 hBar4 / (1152 kB2 T^2), reduced mass in kg
		const double hBar2 = 1.11211999e-68, hBar4 = 1.23681087e-136, kB = 1.3806503e-23, kB2 = 1.90619525e-46, amu = 1.66053873e-27;
		fp.fh_c2 = 1.0e20 * (hBar2 / (24.0 * kB * o.temperature)) / amu;
		fp.fh_c4 = 1.0e40 * (hBar4 / (1152.0 * kB2 * o.temperature * o.temperature)) / (amu * amu);
	}
	fp.wolf_erfa_over_r = std::erf(c->ewald_alpha * c->box.cutoff) / c->box.cutoff;
	fp.wolf_inv_r2 = 1.0 / (c->box.cutoff * c->box.cutoff);
}

// two waves per tile pair in the fast pair sweep (half-length workgroups): by default for the LAST quarter of the work table only -- a lone
// launch drains on units half as long, an ensemble (whose other kernels fill the drain anyway) pays the halved form's overhead on a quarter of
// the work.  The rule is a function of the table alone (never of the call): an evaluation gives the same bits alone and inside an ensemble.
static inline int sweep_split_mode(const mpmc_ctx *c) { return c->tune.pair_split < 0 ? 2 : (c->tune.pair_split != 0 ? 1 : 0); }
static inline int sweep_split_tail(const mpmc_ctx *c) {
	const int permille = c->tune.pair_split_tail >= 0 ? c->tune.pair_split_tail : kSweepSplitTailPermille;
	return (int)((long long)c->n_sweep_blocks * permille / 1000);
}

// which pieces of energy() to run

int mpmc::enqueue(mpmc_ctx *c, unsigned mask) {
	// one evaluation per context at a time: a second enqueue would overwrite the scalar block and the result slots under the first
	if (c->pending) return fail(c, MPMC_ERR_ARG, "an evaluation of this context is still in flight (mpmc_energy_wait first)");
	// stale position-independent terms ride along with this evaluation (an insertion / removal makes them stale every time)
	int rc = prepare(c, true);
	if (rc != MPMC_OK) return rc;
	const bool static_ride = c->static_dirty;
	c->static_ride_gen = 0;
	const AtomsDev at = atoms_view(c);
	const RecipDev rcp = recip_view(c);
	const mpmc_options &o = c->opts;
	hipStream_t st = c->stream;
	c->run_mask = mask;
	c->have_polar = false;
	c->iters = 0;
	c->failed = 0;

	c->last_was_single = false;
	c->spin_on_post = false;
	if (c->tune.single_launch && mask == (RUN_PAIR | RUN_ATOMTERMS) && !o.feynman_hibbs && c->n_tiles <= kSingleLaunchTiles && !c->prof && !static_ride) {
		// small LJ box (BASELINE configs[1]): the whole evaluation is one launch -- pair sweep without classes, the block that finishes last
		// folds the partials into the pinned result vector; the LRC terms are the cached position-independent ones
		FusedParams fp{};
		fp.rd_lrc = o.rd_lrc;
		c->single_seq += 1.0;
		launch_pair_lj_single(st, at, c->box, fp, c->d_tile_pairs, c->n_tile_pairs, c->d_block_part, c->d_block_cnt, c->d_counter, c->h_scal,
		                      c->single_seq);
		HIP_TRY(c, hipGetLastError());
		c->last_was_single = true;
		c->pending = true;
		return MPMC_OK;
	}
	// the scalar block: zeroed by the post kernel of the evaluation before this one; cleared here only when something else used it since
	// (the static-terms pass, an upload of the atoms, an evaluation that failed half way)
	if (!c->scal_clean) HIP_TRY(c, hipMemsetAsync(c->d_scal, 0, (S_COUNT + C_COUNT) * sizeof(double), st));
	c->scal_clean = false;
	if (static_ride) { // first thing on the main stream: its slots are nobody else's (S_LRC_PAIR, S_LRC_SELF, S_ES_SELF)
		launch_atom_terms(st, at, c->box, c->ewald_alpha, o.rd_lrc, /*self term*/ 1, c->d_atom_part, c->d_scal);
		HIP_TRY(c, hipGetLastError());
		c->static_ride_gen = c->static_gen;
	}

	if (mask & (RUN_FIELD | RUN_SOLVE | RUN_STORE)) {
		if ((rc = ensure_polar_buffers(c)) != MPMC_OK) return rc;
		if ((rc = resolve_solver(c)) != MPMC_OK) return rc;
	}
	const bool compact = (mask & RUN_SOLVE) && c->solver_used == MPMC_SOLVER_COMPACT;

	// ---- reciprocal space + O(N) atom terms on the side stream, next to the pair sweep ------------------------------
	const bool need_sf = (mask & RUN_RECIP) || ((mask & RUN_FIELD) && o.polar_ewald);
	// intramolecular charge-to-screen term of coulombic_real: position dependent but independent of the pair sweep; identically zero
	// when every molecule is a single atom
	const bool need_intra = (mask & RUN_PAIR) && (mask & RUN_PAIR_ES) && !(o.wolf && (mask & RUN_WOLF)) && (c->n_molecules != c->n);
	const bool side_work = need_sf || need_intra;
	// a fork/join costs ~20 us of dispatch latency: worth it next to reciprocal-space work, not for the O(N) atom terms alone -- and not
	// for small tables at all (kOneStreamMaxPairs).  Decided here, once per evaluation: nothing is forked at this point.
	// (round 4: ... and not when the caller keeps kOneStreamMinInflight or more evaluations in flight: other evaluations fill the device then,
	// and the fork and join are pure cost -- 1018-1025 against 1006-1012 evaluations/s with 32 beads, 1021 against 1002 with 8; one
	// evaluation at a time the fork is worth 1.5 %.  The choice of streams does not touch the arithmetic.)
	// Only for evaluations with a dipole solve: there the side work is 5 % of the evaluation; a 10 000-atom LJ + Ewald evaluation (sweep 95 us,
	// reciprocal space 25 us) loses 4 % in flight without the overlap (9365 against 9600-9960 evaluations/s).
	c->two_streams = (c->tune.stream_mode == 1) ||
	                 (c->tune.stream_mode < 0 && c->n_tile_pairs > kOneStreamMaxPairs &&
	                  !((mask & RUN_SOLVE) && c->inflight_hint >= kOneStreamMinInflight));
	const bool side_fork = c->two_streams && (need_sf || need_intra);
	bool panel_side = false; // the panel table of the Jacobi contraction is being built on the side stream
	// (two streams: enqueued BEHIND the pair sweep -- the main stream's critical path (classes, sweep) reaches the device first; one
	// evaluation at a time the host used to be ~15 us late with the sweep because nine API calls of side work stood in front of it)
	int side_rc = MPMC_OK;
	auto enqueue_side_work = [&](hipStream_t s2) {
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
					if ((side_rc = dev_alloc(c, &c->d_sf_part, need_part)) != MPMC_OK) return;
					c->cap_sf_part = need_part;
				}
				launch_recip_sf(s2, at, c->box, rcp, o.ewald_kmax, c->d_sf_part);
			}
			if (mask & RUN_RECIP) launch_recip_energy(s2, rcp, c->box, c->d_scal); // (the LRC and self terms are cached: prepare())
		}
		if ((mask & RUN_FIELD) && o.polar_ewald) {
			ProfScope p(c, MPMC_K_FIELD, s2);
			launch_field_recip(s2, at, c->box, rcp, o.ewald_kmax, c->d_e_recip_part);
		}
	};
	const bool side_deferred = side_work && side_fork && (mask & (RUN_PAIR | RUN_FIELD | RUN_STORE)) != 0 && c->tune.side_after_sweep;
	if (side_work && !side_deferred) {
		enqueue_side_work(side_fork ? fork_side(c) : st);
		if (side_rc != MPMC_OK) return side_rc;
	}

	// ---- pairwise pass: one symmetric sweep (energies + counts, static-field partials, Thole tensor store) ----------
	if (mask & (RUN_PAIR | RUN_FIELD | RUN_STORE)) {
		// tile-pair classes from this configuration's tile bounding boxes
		{
			ProfScope pc(c, MPMC_K_CLASSES);
			if (c->tune.no_classes) HIP_TRY(c, hipMemsetAsync(c->d_cls, 0, (size_t)c->n_tile_pairs * sizeof(int), st));
			else launch_tile_classes(st, at, c->box, c->d_tile_pairs, c->n_tile_pairs, (o.polarization && !o.rd_only) ? o.polar_damp : 0.0,
			                         c->d_tile_bounds, c->d_cls, c->tune.no_uniform ? nullptr : c->d_tp_shift, c->sort_origin_f, kTholeFarX);
		}
		FusedParams fp;
		fp.ewald_alpha = c->ewald_alpha;
		fp.polar_ewald_alpha = c->polar_ewald_alpha;
		fp.polar_damp = o.polar_damp;
		fp.rd_lrc = o.rd_lrc;
		fp.do_es = ((mask & RUN_PAIR_ES) || (mask & RUN_FIELD)) ? 1 : 0;
		fp.do_field = (mask & RUN_FIELD) ? (o.polar_ewald ? 1 : 2) : 0;
		fp.do_thole = compact ? 1 : 0;
		ext_params(c, fp, (o.wolf && (mask & RUN_WOLF)) != 0);
		fp.thole_far_x = kTholeFarX;
		fp.pair_waves = c->tune.pair_waves ? c->tune.pair_waves : (c->n_tile_pairs <= kPairSplitMax ? 4 : 1);
		fp.store_only = ((mask & RUN_STORE) && !(mask & (RUN_PAIR | RUN_FIELD))) ? 1 : 0;
		fp.touch_n = fp.store_only ? c->touch_n : -1;
		for (int k = 0; k < 8; k++) fp.touch[k] = c->touch[k];
		if (!fp.store_only && compact) c->store_dirty_tiles.clear(); // a full sweep rebuilds every stored tile pair
		// panels of the Jacobi contraction: two tile pairs of equal class behind one j-tile per workgroup (compact solver)
		c->panels_built = false;
		if (compact && c->tune.use_panels && !c->tune.no_classes && !c->tune.no_uniform && c->n_tiles >= 3) {
			if (c->seg_tiles != c->n_tiles) { // the table's layout depends on the tile count only
				std::vector<int> seg((size_t)c->n_tiles + 1, 0);
				for (int J = 0; J < c->n_tiles; J++) seg[J + 1] = seg[J] + panel_segment_entries(J);
				if ((size_t)c->n_tiles + 1 > c->cap_seg) {
					dev_free(c, &c->d_seg, c->cap_seg);
					dev_free(c, &c->d_arrive, c->cap_seg);
					c->cap_seg = 0;
					if ((rc = dev_alloc(c, &c->d_seg, (size_t)c->n_tiles + 1)) != MPMC_OK) return rc;
					if ((rc = dev_alloc(c, &c->d_arrive, (size_t)c->n_tiles + 1)) != MPMC_OK) return rc;
					c->cap_seg = (size_t)c->n_tiles + 1;
				}
				HIP_TRY(c, hipMemcpyAsync(c->d_seg, seg.data(), seg.size() * sizeof(int), hipMemcpyHostToDevice, st));
				HIP_TRY(c, hipStreamSynchronize(st)); // `seg` dies here
				c->n_panel_entries = seg[c->n_tiles];
				c->seg_tiles = c->n_tiles;
			}
			const size_t need = (size_t)c->n_panel_entries;
			if (need > c->cap_panels) {
				dev_free(c, &c->d_panels, c->cap_panels);
				dev_free(c, &c->d_gpart, c->cap_panels * kTile * 3);
				c->cap_panels = 0;
				if ((rc = dev_alloc(c, &c->d_panels, need)) != MPMC_OK) return rc;
				if ((rc = dev_alloc(c, &c->d_gpart, need * kTile * 3)) != MPMC_OK) return rc;
				c->cap_panels = need;
				if (c->tune.trace_panel) {
					if (c->d_trace) (void)hipFree(c->d_trace);
					c->d_trace = nullptr;
					if ((rc = dev_alloc(c, &c->d_trace, need * 4)) != MPMC_OK) return rc;
				}
			}
			// the table is needed by the first Jacobi launch only: it is made beside the pair sweep (side stream, joined after the sweep)
			panel_side = c->two_streams;
			if (!panel_side) {
				ProfScope pc(c, MPMC_K_CLASSES, st);
				launch_build_panels(st, c->d_cls, c->n_tiles, c->d_seg, c->d_panels, c->d_arrive);
			}
			c->panels_built = true;
		}
		// the side stream starts behind the classes (what it reads of the main stream's work: positions, classes); its kernels are
		// enqueued after the sweep's launch call
		hipStream_t s_side = (side_deferred || panel_side) ? fork_side(c) : st;
		c->last_fp = fp;
		c->last_fp_valid = !fp.store_only;
		ProfScope p(c, MPMC_K_PAIR);
		// the fast sweep (kernels_pair.hip) where it applies -- Ewald electrostatics, alpha r_c inside its erfc table --
		// and, by default, where the table has more than kSweepMinPairs tile pairs (below that the 64 dependent steps of its one wave per
		// tile pair are a latency chain: four waves per tile pair in k_pair_fused); the tile pairs with a special atom, which it skips,
		// go through k_pair_fused on their list
		const bool sweep = c->tune.pair_kernel != 1 && c->d_sweep_blocks && (c->tune.pair_kernel == 2 || c->n_tile_pairs > kSweepMinPairs) &&
		                   pair_sweep_covers(c->box, fp, c->ewald_alpha);
		c->last_pair_was_sweep = sweep;
		if (sweep) {
			launch_pair_sweep(st, at, c->box, fp, c->n_molecules != c->n, c->d_sweep_blocks, c->n_sweep_blocks, c->d_cls,
			                  (c->tune.no_uniform || c->tune.no_classes) ? nullptr : c->d_tp_shift, c->d_erf_tab, c->d_block_part, c->d_block_cnt, c->d_part,
			                  compact ? c->d_ab : nullptr, sweep_split_mode(c), sweep_split_tail(c), c->tune.fast_geometry,
			                  (side_deferred || panel_side) ? c->tune.sweep_lds_pad : 0);
			if (c->n_generic > 0)
				launch_pair_fused(st, at, c->box, fp, c->d_tile_pairs, c->d_cls, c->n_generic, c->d_block_part, c->d_block_cnt, c->d_part,
				                  compact ? c->d_ab : nullptr, c->d_generic_list);
		} else if (!(fp.store_only && !compact)) // (a store-only pass without a store to fill has nothing to do beyond the classes)
			launch_pair_fused(st, at, c->box, fp, c->d_tile_pairs, c->d_cls, c->n_tile_pairs, c->d_block_part, c->d_block_cnt, c->d_part,
			                  compact ? c->d_ab : nullptr);
		if (side_deferred) {
			enqueue_side_work(s_side);
			if (side_rc != MPMC_OK) return side_rc;
		}
		if (panel_side) {
			ProfScope pc(c, MPMC_K_CLASSES, s_side);
			launch_build_panels(s_side, c->d_cls, c->n_tiles, c->d_seg, c->d_panels, c->d_arrive);
		}
	}
	if ((side_work && side_fork) || panel_side) join_side(c);
	// the scalar totals of the sweep are only read back at the very end.  Polarizable evaluations on two streams fold them in the launch
	// that closes the evaluation (second block of the polarization-energy kernel: launch_polar_energy_and_pairs) -- rounds 2-3 forked the
	// side stream for it, which put an event record in front of the static field and a join in front of the posted results (~10 us of
	// barrier packets on the main stream, one evaluation at a time)
	const bool reduce_in_tail = (mask & RUN_PAIR) && (mask & RUN_SOLVE) && c->tune.tail_fused; // (on one stream too: one launch fewer)
	bool reduce_forked = false;
	if ((mask & RUN_PAIR) && !reduce_in_tail) {
		reduce_forked = c->two_streams && (mask & RUN_FIELD) != 0; // (tail_fused = 0: the rounds 2-3 arrangement)
		hipStream_t s3 = reduce_forked ? fork_side(c) : st;
		ProfScope p(c, MPMC_K_REDUCE, s3);
		launch_reduce_pairs(s3, c->d_block_part, c->d_block_cnt, c->n_tile_pairs, c->d_scal, c->d_cnt);
	}

	// ---- static field ---------------------------------------------------------------------------------------
	if (mask & RUN_FIELD) {
		ProfScope p(c, MPMC_K_FIELD);
		c->mu_cur = 0;
		launch_field_finalize(st, at, c->box, o.polar_ewald, c->d_e_recip_part, c->d_part, c->n_tiles, o.polar_gamma, c->d_e_static,
		                      c->d_mu[0], c->d_e_real);
		c->e_real_valid = (mask == full_mask(c)); // (with the accepted positions resident: what trial moves update incrementally)
	}

	// ---- thole_iterative, reference src/System.Energy.cpp:3450-3543 ------------------------------------------------
	if (mask & RUN_SOLVE) {
		const bool by_precision = (o.polar_precision != 0.0);
		const int want_rrms = (o.polar_rrms || o.polar_precision > 0) ? 1 : 0;
		const double allowed = by_precision ? o.polar_precision * o.polar_precision * kDebye2SKA * kDebye2SKA : 0.0;
		const bool dense = (c->solver_used == MPMC_SOLVER_DENSE) && !o.polar_gs;
		const bool dense_sym = dense && c->tune.dense_symmetric;
		const int iter_slots = (dense && !dense_sym) ? kDenseChunks : c->n_tiles;
		if (dense) { // thole_amatrix into device memory, once per evaluation (the positions changed)
			ProfScope p(c, MPMC_K_TENSOR);
			launch_dense_build(st, at, c->box, o.polar_damp, c->d_adense, dense_sym);
		}
		// Precision-terminated Jacobi solves: are_we_done_yet (:3215-3239) runs on the device (ctl = {broke, converged-at, ticket}); the host
		// enqueues kCheckEvery iterations at a time and reads the verdict once per batch -- the iterations enqueued behind the one that
		// converged return at once and leave the dipoles alone.  (Gauss-Seidel sweeps and the dense solver still ask after every iteration.)
		int *ctl = (by_precision && !o.polar_gs) ? c->d_flag + 1 : nullptr;
		const int *converged = ctl ? ctl + 1 : nullptr;
		const int check_every = dense ? 1 : kCheckEvery;
		const int mu_start = c->mu_cur;
		int done_at = 0;
		int *host_flag = ctl ? c->h_flag + 2 : nullptr; // pinned {last closed iteration, converged-at} the closing update block posts
		if (ctl) {
			HIP_TRY(c, hipMemsetAsync(ctl, 0, 3 * sizeof(int), st));
			c->h_flag[2] = c->h_flag[3] = 0; // (nothing of an earlier solve can still be in flight: every evaluation is waited for)
		}
		if (o.polar_gs) { // the in-tile blocks of the sweeps: positions and polarizabilities only, once per evaluation
			const size_t need = gs_block_store_elements(c->n_tiles);
			if (need > c->cap_gs_blocks) {
				dev_free(c, &c->d_gs_blocks, c->cap_gs_blocks);
				c->cap_gs_blocks = 0;
				if ((rc = dev_alloc(c, &c->d_gs_blocks, need)) != MPMC_OK) return rc;
				c->cap_gs_blocks = need;
			}
			ProfScope p(c, MPMC_K_TENSOR);
			launch_gs_blocks(st, at, c->box, o.polar_damp, c->d_gs_blocks);
		}
		// compact solver on the panel path: the update of the dipoles is the tail of the contraction's own launch (kernels_panel.hip)
		const bool fused_update = compact && c->panels_built && !dense && !o.polar_gs && c->tune.fused_update;
		int it = 0;
		bool keep = true;
		while (keep) {
			it++;
			if (it >= kMaxIterationCount && by_precision) { // divergence: mu = alpha E0, iterator_failed (:3483-3494)
				launch_dipole_reset(st, at, c->d_e_static, c->d_mu[c->mu_cur]);
				c->failed = 1;
				break;
			}
			if (by_precision && o.polar_gs) HIP_TRY(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), st));
			if (o.polar_gs) { // in-place sweep in atom order; old_mu is kept only when rrms / precision need it (:3503-3507)
				double *mu = c->d_mu[c->mu_cur], *mu_old = c->d_mu[1 - c->mu_cur];
				if (want_rrms) HIP_TRY(c, hipMemcpyAsync(mu_old, mu, 3 * (size_t)at.n_pad * sizeof(double), hipMemcpyDeviceToDevice, st));
				{
					ProfScope p(c, MPMC_K_DIPOLE_ITER);
					launch_gs_sweep(st, at, c->box, o.polar_damp, c->d_e_static, mu, c->d_e_induced, c->d_part, c->d_tile_pairs, c->d_cls,
					                (c->tune.no_uniform || c->tune.no_classes) ? nullptr : c->d_tp_shift, c->n_tile_pairs, c->d_gs_ul, c->d_gs_ul + 3 * (size_t)c->max_pad, c->d_gs_blocks);
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
				if (dense_sym) launch_dense_symv(st, c->d_adense, c->n_pad, c->d_mu[c->mu_cur], c->d_tile_pairs, c->n_tile_pairs, c->d_part);
				else launch_dense_matvec(st, c->d_adense, c->n_pad, c->d_mu[c->mu_cur], kDenseChunks, c->d_part);
			} else if (compact) {
				ProfScope p(c, MPMC_K_DIPOLE_ITER);
				if (c->panels_built) { // every tile pair through the panel table: two per wave where classes allow; the update rides along
					PanelFuse fu{};
					fu.arrive = fused_update ? c->d_arrive : nullptr, fu.reverse = c->tune.panel_reverse ? 1 : 0;
					fu.e_static = c->d_e_static, fu.seg = c->d_seg, fu.mu_new = c->d_mu[1 - c->mu_cur], fu.e_induced = c->d_e_induced, fu.rrms_atom = c->d_rrms;
					fu.allowed_sqerr = allowed, fu.ctl = ctl, fu.host_flag = host_flag, fu.it = it, fu.want_rrms = want_rrms, fu.probe = c->tune.fused_update == 2;
					launch_dipole_iter_panel(st, at, c->box, c->d_mu[c->mu_cur], c->d_tile_pairs, c->d_tp_shift, c->d_panels,
					                         c->n_panel_entries, c->d_ab, c->d_part, c->d_gpart, converged, c->d_trace, 1, &fu);
				} else
					launch_dipole_iter_hybrid(st, at, c->box, c->d_mu[c->mu_cur], c->d_tile_pairs, c->d_cls,
					                          (c->tune.no_uniform || c->tune.no_classes) ? nullptr : c->d_tp_shift, c->n_tile_pairs, c->d_ab, c->d_part, o.polar_damp,
					                          converged);
			} else { // matrix-free: the same symmetric tile-pair walk with nothing stored (null store => damped tensors rebuilt)
				ProfScope p(c, MPMC_K_DIPOLE_ITER);
				launch_dipole_iter_hybrid(st, at, c->box, c->d_mu[c->mu_cur], c->d_tile_pairs, c->d_cls,
				                          (c->tune.no_uniform || c->tune.no_classes) ? nullptr : c->d_tp_shift, c->n_tile_pairs, nullptr, c->d_part, o.polar_damp,
				                          converged);
			}
			if (!fused_update) {
				ProfScope p(c, MPMC_K_REDUCE);
				if (compact && c->panels_built && !dense)
					launch_dipole_update_panel(st, at, c->d_e_static, c->d_part, c->d_gpart, c->d_seg, c->d_mu[c->mu_cur], c->d_mu[1 - c->mu_cur],
					                           c->d_e_induced, want_rrms, c->d_rrms, allowed, ctl, host_flag, it,
					                           c->tune.update_waves ? c->tune.update_waves : 4);
				else
					launch_dipole_update(st, at, c->d_e_static, c->d_part, iter_slots, c->d_mu[c->mu_cur], c->d_mu[1 - c->mu_cur], c->d_e_induced,
					                     want_rrms, c->d_rrms, allowed, ctl, host_flag, it);
			}
			c->mu_cur = 1 - c->mu_cur;
			if (by_precision) {
				if (it % check_every == 0 || it + 1 >= kMaxIterationCount) { // the verdict of this batch
					HIP_TRY(c, hipGetLastError());
					// spin on the pinned flag until iteration `it` is closed (or an earlier one converged); a stream synchronisation costs
					// ~15 us, this ~2.  Past a generous budget fall back to the blocking read (correct either way).
					volatile const int *hf = host_flag;
					const bool seen = poll_posted(c, [&] { return hf[0] >= it || hf[1] != 0; }, std::chrono::microseconds(50000));
					if (seen) {
						done_at = hf[1];
					} else {
						c->n_stream_syncs++;
						HIP_TRY(c, hipMemcpyAsync(c->h_flag, ctl + 1, sizeof(int), hipMemcpyDeviceToHost, st));
						HIP_TRY(c, hipStreamSynchronize(st));
						done_at = *c->h_flag;
					}
					keep = (done_at == 0);
				}
			} else {
				keep = (it != o.polar_max_iter);
			}
		}
		if (done_at > 0) { // the iterations enqueued behind the converged one did nothing: the result is where iteration done_at left it
			it = done_at;
			c->mu_cur = (mu_start + done_at) & 1;
		}
		c->iters = it;
		{
			ProfScope p(c, MPMC_K_REDUCE);
			if (reduce_in_tail)
				launch_polar_energy_and_pairs(st, at, c->d_mu[c->mu_cur], c->d_e_static, want_rrms ? c->d_rrms : nullptr, c->d_block_part, c->d_block_cnt,
				                              c->n_tile_pairs, c->d_scal, c->d_cnt);
			else launch_polar_energy(st, at, c->d_mu[c->mu_cur], c->d_e_static, want_rrms ? c->d_rrms : nullptr, c->d_scal);
		}
		c->have_polar = true;
	}
	if (reduce_forked) join_side(c);
	HIP_TRY(c, hipGetLastError());
	// results to the pinned block by a kernel of ours (a blit and a stream synchronisation cost more than the whole reciprocal space of a
	// small box): copy, zero the device block for the next evaluation, launch number last
	c->single_seq += 1.0;
	launch_post_results(st, c->d_scal, c->h_scal, c->single_seq);
	HIP_TRY(c, hipGetLastError());
	c->scal_clean = true;
	// short evaluations are polled for (a few us against ~10-15 for the synchronisation); long ones, and profiled ones (the event
	// harvest needs an idle stream), are waited for the ordinary way
	// (round 4: long evaluations are polled for as well, with a budget of a few of their own durations -- one evaluation at a time the
	// posted launch number is seen ~10 us before hipStreamSynchronize returns; an ensemble's first wait outlasts the budget and synchronises)
	c->spin_on_post = c->ev_used.empty() && (c->tune.poll_long || c->n_tile_pairs <= kOneStreamMaxPairs);
	c->poll_budget_us = (c->n_tile_pairs <= kOneStreamMaxPairs) ? 1000 : 4000;
	c->pending = true;
	return MPMC_OK;
}


// measurement only (bench.py's roofline, cross-check): the panel Jacobi kernel `reps` times back to back between ONE pair of HIP events
// on the context's stream, per launch.  It agrees with the profiling mode's per-launch event brackets (92.7 against 92.9 us): what
// separates both from rocprofv3's kernel trace (86 us) is the dispatch / completion time between consecutive kernels of a stream, not
// the event records.  Needs the state a polarizable evaluation of a box on the panel path leaves behind; the partial slots it
// overwrites are dead by then.
extern "C" int mpmc_debug_time_panel(mpmc_ctx *c, int reps, double *ms_per_launch) {
	if (!c || !ms_per_launch || reps <= 0) return MPMC_ERR_ARG;
	if (c->pending) return fail(c, MPMC_ERR_ARG, "mpmc_debug_time_panel: an evaluation is in flight");
	if (!c->have_polar || !c->panels_built || c->solver_used != MPMC_SOLVER_COMPACT)
		return fail(c, MPMC_ERR_ARG, "mpmc_debug_time_panel: the last evaluation did not run the panel kernel");
	HIP_TRY(c, hipSetDevice(c->device));
	const AtomsDev at = atoms_view(c);
	hipEvent_t e0, e1;
	HIP_TRY(c, hipEventCreate(&e0));
	HIP_TRY(c, hipEventCreate(&e1));
	// the launch as the solve makes it: with the fused update (default) the new dipoles go to the buffer that is dead after the solve and the
	// induced field is not stored, so the evaluation's results stay as they are
	PanelFuse fu{};
	fu.arrive = c->tune.fused_update ? c->d_arrive : nullptr, fu.reverse = c->tune.panel_reverse ? 1 : 0;
	fu.e_static = c->d_e_static, fu.seg = c->d_seg, fu.mu_new = c->d_mu[1 - c->mu_cur], fu.e_induced = nullptr, fu.rrms_atom = c->d_rrms, fu.probe = c->tune.fused_update == 2;
	for (int r = 0; r < 3; r++) // (warm)
		launch_dipole_iter_panel(c->stream, at, c->box, c->d_mu[c->mu_cur], c->d_tile_pairs, c->d_tp_shift, c->d_panels, c->n_panel_entries,
		                         c->d_ab, c->d_part, c->d_gpart, nullptr, nullptr, 1, &fu);
	HIP_TRY(c, hipEventRecord(e0, c->stream));
	const int replicas = c->debug_panel_replicas; // (> 1: every launch carries the grid that many times, without the update: what a batched launch would cost per system)
	for (int r = 0; r < reps; r++)
		launch_dipole_iter_panel(c->stream, at, c->box, c->d_mu[c->mu_cur], c->d_tile_pairs, c->d_tp_shift, c->d_panels, c->n_panel_entries,
		                         c->d_ab, c->d_part, c->d_gpart, nullptr, nullptr, replicas, &fu);
	HIP_TRY(c, hipEventRecord(e1, c->stream));
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipEventSynchronize(e1));
	float ms = 0;
	HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	*ms_per_launch = (double)ms / reps;
	return MPMC_OK;
}

// the same for the pair pass of the last evaluation (fast sweep or k_pair_fused, whichever ran): `reps` launches back to back between one
// pair of events.  Needs the classes of a complete evaluation; what it overwrites (block partials, field slots, tensor store) is rewritten
// identically, the configuration being the same.
extern "C" int mpmc_debug_time_pair(mpmc_ctx *c, int reps, double *ms_per_launch) {
	if (!c || !ms_per_launch || reps <= 0) return MPMC_ERR_ARG;
	if (c->pending) return fail(c, MPMC_ERR_ARG, "mpmc_debug_time_pair: an evaluation is in flight");
	if (!c->cache_valid || !c->last_fp_valid) return fail(c, MPMC_ERR_ARG, "mpmc_debug_time_pair: no complete evaluation has run");
	HIP_TRY(c, hipSetDevice(c->device));
	const AtomsDev at = atoms_view(c);
	const FusedParams &fp = c->last_fp;
	const bool compact = c->solver_used == MPMC_SOLVER_COMPACT && fp.do_thole;
	bool timed = false; // (the warm-up launches run the plain grid; the timed ones carry the replicas of "panel_replicas", if any)
	auto launch = [&] {
		if (c->last_pair_was_sweep) {
			launch_pair_sweep(c->stream, at, c->box, fp, c->n_molecules != c->n, c->d_sweep_blocks, c->n_sweep_blocks, c->d_cls,
			                  (c->tune.no_uniform || c->tune.no_classes) ? nullptr : c->d_tp_shift, c->d_erf_tab, c->d_block_part, c->d_block_cnt, c->d_part,
			                  compact ? c->d_ab : nullptr, sweep_split_mode(c), sweep_split_tail(c), c->tune.fast_geometry,
			                  c->two_streams ? c->tune.sweep_lds_pad : 0, timed ? c->debug_panel_replicas : 1);
			if (c->n_generic > 0)
				launch_pair_fused(c->stream, at, c->box, fp, c->d_tile_pairs, c->d_cls, c->n_generic, c->d_block_part, c->d_block_cnt, c->d_part,
				                  compact ? c->d_ab : nullptr, c->d_generic_list);
		} else {
			launch_pair_fused(c->stream, at, c->box, fp, c->d_tile_pairs, c->d_cls, c->n_tile_pairs, c->d_block_part, c->d_block_cnt, c->d_part,
			                  compact ? c->d_ab : nullptr);
		}
	};
	hipEvent_t e0, e1;
	HIP_TRY(c, hipEventCreate(&e0));
	HIP_TRY(c, hipEventCreate(&e1));
	for (int r = 0; r < 2; r++) launch();
	timed = true;
	HIP_TRY(c, hipEventRecord(e0, c->stream));
	for (int r = 0; r < reps; r++) launch();
	HIP_TRY(c, hipEventRecord(e1, c->stream));
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipEventSynchronize(e1));
	float ms = 0;
	HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	*ms_per_launch = (double)ms / reps;
	return MPMC_OK;
}

int mpmc::wait_and_fill(mpmc_ctx *c, mpmc_result *out) {
	if (!c->pending) return fail(c, MPMC_ERR_ARG, "mpmc_energy_wait: nothing enqueued");
	// a wait that fails must not leave the context refusing every later enqueue ("still in flight"): whatever happens below, the
	// evaluation is over for the host -- best-effort drain, state back to idle, scalar block marked dirty so that the next one clears it
	auto abandon = [c](hipError_t e, const char *what) {
		(void)hipStreamSynchronize(c->stream);
		if (c->two_streams && c->stream2) (void)hipStreamSynchronize(c->stream2);
		c->sync_stream = nullptr;
		c->pending = false;
		c->scal_clean = false;
		c->static_ride_gen = 0;
		return fail(c, MPMC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
	};
	hipError_t werr = c->tune.fail_next_wait ? hipErrorUnknown : hipSetDevice(c->device);
	c->tune.fail_next_wait = 0;
	if (werr != hipSuccess) return abandon(werr, "mpmc_energy_wait: hipSetDevice");
	bool seen = false;
	if (c->last_was_single || c->spin_on_post) {
		// the kernel posts its launch number behind the results (system-scope release): a short spin on the pinned slot returns a few
		// microseconds before the driver's own completion path would; past the budget, or if anything is off, fall back to the sync
		volatile const double *flag = c->h_scal + S_COUNT + C_COUNT;
		const double want = c->single_seq;
		seen = poll_posted(c, [&] { return *flag == want; }, std::chrono::microseconds(c->last_was_single ? 200 : c->poll_budget_us));
	}
	if (!seen) {
		c->n_stream_syncs++;
		werr = hipStreamSynchronize(c->sync_stream ? c->sync_stream : c->stream);
		if (werr != hipSuccess) return abandon(werr, "mpmc_energy_wait: hipStreamSynchronize");
	} else if (c->tune.poll_retire && c->n_tile_pairs > kOneStreamMaxPairs) {
		// the results are in, but the runtime has not been told: a stream that is never synchronised keeps its finished commands, and
		// the next asynchronous copy on it pays for the backlog (measured with positions handed over in host memory: 900 against 966
		// evaluations/s).  A query is enough to let it retire them.  Long evaluations only: the query costs a few microseconds, which is
		// a third of a 1000-atom LJ evaluation (23 -> 30 us when it ran behind every poll).
		(void)hipStreamQuery(c->stream);
		if (c->two_streams && c->stream2) (void)hipStreamQuery(c->stream2);
	}
	c->sync_stream = nullptr;
	c->pending = false;
	prof_harvest(c);
	if (!out) return MPMC_OK;
	std::memset(out, 0, sizeof(*out));
	const double *s = c->h_scal;
	if (c->static_ride_gen) { // the position-independent terms came along: adopt them, unless something made them stale again meanwhile
		if (c->static_ride_gen == c->static_gen) {
			c->h_static[0] = s[S_LRC_PAIR];
			c->h_static[1] = s[S_LRC_SELF];
			c->h_static[2] = s[S_ES_SELF];
			c->static_dirty = false;
		}
		c->static_ride_gen = 0;
	}
	const bool lrc = (c->run_mask & RUN_ATOMTERMS) && c->opts.rd_lrc;
	out->lj_pairs = s[S_LJ];
	out->lrc_pair = lrc ? c->h_static[0] : 0.0;
	out->lrc_self = lrc ? c->h_static[1] : 0.0;
	out->rd_energy = (out->lj_pairs + out->lrc_pair) + out->lrc_self;
	out->es_real = s[S_ES_REAL] - s[S_ES_INTRA];
	out->es_recip = s[S_ES_RECIP];
	out->es_self = (c->run_mask & RUN_RECIP) ? c->h_static[2] : 0.0;
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

unsigned mpmc::full_mask(const mpmc_ctx *c) {
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
extern "C" int mpmc_hint_in_flight(mpmc_ctx *c, int n) {
	if (!c || n < 1) return MPMC_ERR_ARG;
	c->inflight_hint = n;
	return MPMC_OK;
}
extern "C" int mpmc_energy_wait(mpmc_ctx *c, mpmc_result *out) {
	if (!c) return MPMC_ERR_ARG;
	return wait_and_fill(c, out);
}
extern "C" int mpmc_energy(mpmc_ctx *c, mpmc_result *out) {
	if (!c || !out) return MPMC_ERR_ARG;
	c->inflight_hint = 1; // (a synchronous call: nothing else of this caller is in flight)
	int rc = enqueue(c, full_mask(c));
	if (rc != MPMC_OK) return rc;
	return wait_and_fill(c, out);
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
	HIP_TRY(c, hipMemcpyAsync(tmp.data(), d_src, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
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
