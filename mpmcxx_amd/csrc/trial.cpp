// trial.cpp -- per-move delta energies (mpmc_trial_*): the device-side counterpart of the reference's per-pair cache
// (part of libmpmc_energy.so; shared state and helpers: context.h.  There is no CPU fallback anywhere in this library.)
#include <atomic>
#include <chrono>

#include "context.h"


using namespace mpmc;

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
		HIP_TRY(c, hipMemsetAsync(c->d_moved_idx, 0xff, (size_t)c->max_pad * sizeof(int), c->stream)); // all -1; on our stream (ordered before the first delta kernel)
		HIP_TRY(c, pinned_alloc(&c->h_delta_out, 9 * sizeof(double)));
		c->h_delta_out[8] = 0.0;
		c->h_delta_cnt = reinterpret_cast<long long *>(c->h_delta_out + 5);
		HIP_TRY(c, pinned_alloc(&c->h_mv_blob, kMvBlobBytes));
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
	c->trial_polar_delta = false;
	const bool no_polar_delta = c->tune.no_polar_delta;
	// (Wolf electrostatics and the Feynman-Hibbs corrections are per-pair terms like the others: the delta kernels carry them; a
	// polarizable box under Wolf keeps the full evaluation -- its static field is the Ewald one, outside the reference's own combinations)
	const bool polar_delta = polar && c->e_real_valid && !no_polar_delta && m <= MPMC_TRIAL_MAX_ATOMS && !o.wolf;
	if ((polar && !polar_delta) || m > MPMC_TRIAL_MAX_ATOMS) {
		// the dipole solve couples every atom: evaluate the trial configuration in full (still on the device)
		c->trial_keep = c->last_full;
		int rc = mpmc_update_positions(c, c->trial_first, m, c->trial_new.data());
		if (rc != MPMC_OK) return rc;
		if ((rc = mpmc_energy_async(c)) != MPMC_OK) return rc;
		c->trial_was_full = true;
		c->trial_last_kind = 1;
		c->trial_enqueued = true;
		return MPMC_OK;
	}
	int rc = prepare(c);
	if (rc != MPMC_OK) return rc;
	if ((rc = ensure_trial_buffers(c)) != MPMC_OK) return rc;
	hipStream_t st = c->stream;
	// short moves of non-polarizable boxes travel in the kernel arguments: no staging copy (a trial is launch-bound on the host: every
	// call saved is ~5 us of a ~25 us move).  The polarizable path keeps the device lists (its field / store kernels read them).
	const bool no_inline = c->tune.no_inline_move;
	c->trial_inline = !polar_delta && m <= kMvInline && !no_inline;
	if (c->trial_inline) {
		for (int t = 0; t < m; t++) {
			const int i = c->trial_first + t;
			c->mv_inline.orig[t] = i;
			c->mv_inline.slot[t] = c->slot_of[i];
			for (int d = 0; d < 3; d++) c->mv_inline.nw[t][d] = c->trial_new[3 * t + d];
			c->mv_inline.nw[t][3] = c->h_q[i];
		}
		for (int t = m; t < kMvInline; t++) { // (unused entries: defined values, never selected)
			c->mv_inline.orig[t] = c->mv_inline.slot[t] = 0;
			for (int d = 0; d < 4; d++) c->mv_inline.nw[t][d] = 0.0;
		}
	} else { // one pinned staging record, one host-to-device copy
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
		FusedParams fp{};
		fp.ewald_alpha = c->ewald_alpha;
		ext_params(c, fp, o.wolf && do_es);
		ProfScope p(c, MPMC_K_PAIR);
		launch_delta(st, atoms_view(c), c->d_slot_of, c->box, recip_view(c), fp, do_es, c->d_mv_slot, c->d_mv_orig, c->d_mv_new, m,
		             c->d_moved_idx, c->d_sf_trial, c->d_block_part, c->d_block_cnt, c->d_delta_out, c->d_delta_cnt, c->h_delta_out,
		             (c->trial_seq += 1.0), c->trial_inline ? &c->mv_inline : nullptr);
	}
	HIP_TRY(c, hipGetLastError()); // (k_delta_finish posts the result into h_delta_out itself)
	c->trial_was_full = false;
	c->trial_last_kind = 0;
	if (polar_delta) {
		// Polarizable box: the pair energies and structure factors above are O(m N); the static field follows the same way -- real part:
		// delta of the pairs with a moved atom; reciprocal part: recomputed from the trial structure factors (O(K N), 30 us at 10 000
		// atoms) -- then the tensor store is rebuilt for the trial geometry (store-only sweep of the near tile pairs) and the dipoles are
		// solved from alpha E0 exactly as in a full evaluation (thole_iterative restarts every call, :3547-3560).  What is saved: the
		// energy / erfc / field arithmetic of the N^2/2 pair sweep and the O(K N) structure factors.
		const size_t need = (size_t)c->n_tiles * (size_t)m * 3;
		if (need > c->cap_dk_part) {
			dev_free(c, &c->d_dk_part, c->cap_dk_part);
			c->cap_dk_part = 0;
			if ((rc = dev_alloc(c, &c->d_dk_part, (size_t)c->n_tiles * MPMC_TRIAL_MAX_ATOMS * 3)) != MPMC_OK) return rc;
			c->cap_dk_part = (size_t)c->n_tiles * MPMC_TRIAL_MAX_ATOMS * 3;
		}
		const AtomsDev at = atoms_view(c);
		{
			ProfScope p(c, MPMC_K_FIELD);
			launch_delta_field(st, at, c->box, c->polar_ewald_alpha, o.polar_ewald, c->d_mv_slot, c->d_mv_new, m, c->d_moved_idx, c->d_e_real,
			                   c->d_e_real_trial, c->d_dk_part);
			// from here on the resident positions are the TRIAL ones (the old ones wait in d_mv_new: reject swaps them back)
			launch_swap_positions(st, c->d_xyzq, c->d_mv_slot, c->d_mv_new, m);
			if (o.polar_ewald) {
				RecipDev rt = recip_view(c);
				rt.sf = c->d_sf_trial;
				launch_field_recip(st, at, c->box, rt, o.ewald_kmax, c->d_e_recip_part);
			}
			c->mu_cur = 0;
			launch_field_finalize(st, at, c->box, o.polar_ewald, c->d_e_recip_part, c->d_e_real_trial, 1, o.polar_gamma, c->d_e_static, c->d_mu[0]);
		}
		HIP_TRY(c, hipGetLastError());
		c->trial_polar_delta = true;
		// the store-only sweep rebuilds the tile pairs of the tiles the moved atoms live in, and of the tiles a rejected trial left behind:
		// no other tile pair's geometry (or class) changed
		c->trial_tiles.clear();
		for (int t = 0; t < m; t++) {
			const int tile = c->slot_of[c->trial_first + t] / kTile;
			if (std::find(c->trial_tiles.begin(), c->trial_tiles.end(), tile) == c->trial_tiles.end()) c->trial_tiles.push_back(tile);
		}
		std::vector<int> touch = c->trial_tiles;
		for (int tile : c->store_dirty_tiles)
			if (std::find(touch.begin(), touch.end(), tile) == touch.end()) touch.push_back(tile);
		c->touch_n = (touch.size() <= 8) ? (int)touch.size() : -1;
		for (int k = 0; k < c->touch_n; k++) c->touch[k] = touch[k];
		rc = enqueue(c, RUN_STORE | RUN_SOLVE); // classes + store of the trial geometry, the iterations, -1/2 mu.E0
		c->touch_n = -1;
		if (rc != MPMC_OK) return rc;
		c->store_dirty_tiles = c->trial_tiles; // until accepted: these tiles hold the TRIAL geometry's tensors
	}
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
	mpmc_result solved{};
	if (c->trial_polar_delta) { // the solve half went through enqueue(): its scalars (polarization energy, rrms, iterations) arrive the usual way
		const mpmc_result keep = c->last_full;
		int rc = wait_and_fill(c, &solved);
		c->last_full = keep;
		if (rc != MPMC_OK) return rc;
	} else {
		// the finish kernel posts its launch number behind the result: poll for it (a trial is a few tens of microseconds; the stream
		// synchronisation alone costs 10-15); with profiling events outstanding, or past the budget, wait the ordinary way
		bool seen = false;
		if (c->ev_used.empty()) {
			volatile const double *flag = c->h_delta_out + 8;
			const double want = c->trial_seq;
			seen = poll_posted(c, [&] { return *flag == want; }, std::chrono::microseconds(1000));
		}
		if (!seen) {
			c->n_stream_syncs++;
			HIP_TRY(c, hipStreamSynchronize(c->stream));
		}
		prof_harvest(c);
	}
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
	if (c->trial_polar_delta) {
		r.polarization_energy = solved.polarization_energy;
		r.dipole_rrms = solved.dipole_rrms;
		r.polar_iterations = solved.polar_iterations;
		r.iterator_failed = solved.iterator_failed;
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
		if (c->trial_polar_delta) { // the trial positions are already resident; the trial real-space field becomes the accepted one
			std::swap(c->d_e_real, c->d_e_real_trial);
			c->store_dirty_tiles.clear(); // ... and so is the store
		} else {
			if (c->trial_inline) launch_commit_positions_inline(c->stream, c->d_xyzq, c->mv_inline, m);
			else launch_commit_positions(c->stream, c->d_xyzq, c->d_mv_slot, c->d_mv_new, m);
		}
		HIP_TRY(c, hipGetLastError());
		std::swap(c->d_sf, c->d_sf_trial); // the trial structure factors become the accepted ones
		std::swap(c->cap_sf, c->cap_sf_trial); // (d_sf has its own capacity: cap_K sizes the k tables, which do not move)
		// (no wait: whatever comes next on this context is enqueued behind the commit on the same stream, and the host mirrors below are
		// the host's own -- the stream synchronisation that stood here cost 13 us of a 41 us accepted move)
		for (int t = 0; t < 3 * m; t++) c->h_pos[3 * (size_t)c->trial_first + t] = c->trial_new[t];
		{
			const int rc_g = mirror_guard(c);
			if (rc_g != MPMC_OK) return rc_g;
		}
		for (int t = 0; t < m; t++) { // the slot-ordered mirror follows (a later position update uploads it as a whole)
			double4 &v = c->h_xyzq[c->slot_of[c->trial_first + t]];
			v.x = c->trial_new[3 * t], v.y = c->trial_new[3 * t + 1], v.z = c->trial_new[3 * t + 2];
		}
	}
	c->last_full = c->trial_res;
	c->cache_valid = true;
	c->trial_open = false;
	c->trial_polar_delta = false;
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
	if (c->trial_polar_delta) { // (enqueued or evaluated) the device holds the trial positions: swap the accepted ones back
		HIP_TRY(c, hipSetDevice(c->device));
		launch_swap_positions(c->stream, c->d_xyzq, c->d_mv_slot, c->d_mv_new, c->trial_count);
		HIP_TRY(c, hipGetLastError()); // (stream-ordered in front of whatever comes next: nothing to wait for)
		c->trial_polar_delta = false;
		return MPMC_OK; // (the store, classes and dipoles describe the rejected geometry; the next trial or energy() rebuilds them)
	}
	if (c->trial_evaluated && c->trial_was_full) { // the resident configuration is the trial one: put the old positions back
		const mpmc_result keep = c->last_full;
		int rc = mpmc_update_positions(c, c->trial_first, c->trial_count, c->trial_old.data());
		if (rc != MPMC_OK) return rc;
		c->last_full = keep;
		c->cache_valid = true;
		c->e_real_valid = false; // (the real-space field on the device is the rejected configuration's: the next polarizable trial evaluates in full)
		const bool polar = c->opts.polarization && !c->opts.rd_only;
		if (!polar && !c->opts.wolf) { // the resident structure factors are the trial ones: re-base on the restored configuration
			mpmc_result tmp;
			if ((rc = mpmc_energy(c, &tmp)) != MPMC_OK) return rc;
		}
	}
	return MPMC_OK;
}
