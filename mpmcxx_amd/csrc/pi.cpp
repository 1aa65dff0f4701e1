// pi.cpp -- the path-integral bead loop (PI_calculate_potential) and the PI estimators
// (part of libmpmc_energy.so; shared state and helpers: context.h.  There is no CPU fallback anywhere in this library.)
#include "context.h"


using namespace mpmc;

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
