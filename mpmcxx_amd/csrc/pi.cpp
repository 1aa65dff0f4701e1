// pi.cpp -- the path-integral bead loop (PI_calculate_potential) and the PI estimators
// (part of libmpmc_energy.so; shared state and helpers: context.h.  There is no CPU fallback anywhere in this library.)
#include "context.h"


using namespace mpmc;

// ---- path integral ---------------------------------------------------------------------------------------
// PI_calculate_potential (src/SimulationControl.PathIntegral.cpp:752-805) over the beads of this process: every bead is a complete,
// independent evaluation on its own context and streams; all of them are enqueued before the first wait, so one bead's pair sweep
// fills the launch ramps and tails of another bead's dipole iterations (a lockstep form that shared launches between the beads was
// measured in round 1 -- 650 against 737 evaluations/s -- and removed in round 3).
// An error in the middle of the loop must not leave evaluations in flight that nobody waits for: the caller's next step would write a
// bead's buffers under a running evaluation.  Beads [first, last) are waited for, their results dropped; the first error stays the answer.
static void pi_drain(mpmc_ctx **beads, int first, int last) {
	for (int b = first; b < last; b++)
		if (beads[b] && beads[b]->pending) {
			mpmc_result dropped;
			(void)mpmc_energy_wait(beads[b], &dropped);
		}
}

static int pi_wait_all(mpmc_ctx **beads, int n_local, double sums4[4], mpmc_result *per_bead, int *any_failed) {
	sums4[0] = sums4[1] = sums4[2] = sums4[3] = 0;
	int failed = 0;
	for (int b = 0; b < n_local; b++) {
		mpmc_result r;
		int rc = mpmc_energy_wait(beads[b], &r);
		if (rc != MPMC_OK) {
			pi_drain(beads, b + 1, n_local);
			return rc;
		}
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

// systems per launch in the dipole iterations of this context's last evaluation: always 1 (kept for ABI stability)
extern "C" int mpmc_last_batch_size(mpmc_ctx *c) { return c ? 1 : 0; }

extern "C" int mpmc_pi_potential_local(mpmc_ctx **beads, int n_local, double sums4[4], mpmc_result *per_bead, int *any_failed) {
	if (!beads || n_local < 0 || !sums4) return MPMC_ERR_ARG;
	for (int b = 0; b < n_local; b++) {
		if (beads[b]) beads[b]->inflight_hint = n_local;
		int rc = beads[b] ? enqueue(beads[b], full_mask(beads[b])) : MPMC_ERR_ARG;
		if (rc != MPMC_OK) {
			pi_drain(beads, 0, b);
			return rc;
		}
	}
	return pi_wait_all(beads, n_local, sums4, per_bead, any_failed);
}

// The same step with every bead's positions handed over in HOST memory (pos[b]: n x 3 doubles, original atom order) -- the boundary as
// a host program with its own copy of the coordinates uses it.  Bead b's upload (host pass over its 3 n doubles, one asynchronous
// 32 n-byte copy from the context's pinned mirror on the bead's own stream) is followed at once by bead b's enqueue, so the device
// works on bead b while the host prepares bead b + 1: the uploads hide behind the evaluations instead of standing in front of them.
extern "C" int mpmc_pi_potential_local_host(mpmc_ctx **beads, int n_local, const double *const *pos, double sums4[4], mpmc_result *per_bead,
                                            int *any_failed) {
	if (!beads || n_local < 0 || !sums4 || !pos) return MPMC_ERR_ARG;
	for (int b = 0; b < n_local; b++) {
		mpmc_ctx *c = beads[b];
		int rc = (c && pos[b]) ? mpmc_update_positions(c, 0, c->n, pos[b]) : MPMC_ERR_ARG;
		if (rc == MPMC_OK) {
			c->inflight_hint = n_local;
			rc = enqueue(c, full_mask(c));
		}
		if (rc != MPMC_OK) {
			pi_drain(beads, 0, b);
			return rc;
		}
	}
	return pi_wait_all(beads, n_local, sums4, per_bead, any_failed);
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
