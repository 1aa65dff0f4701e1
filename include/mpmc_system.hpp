// include/mpmc_system.hpp -- C++ host facade over the C ABI (include/mpmc_energy.h).
//
// The reference is compiled C++ whose energy path is a set of `System` member functions; this header gives host
// C++ code the same surface -- same member names, argument meaning, side effects and error behaviour -- so that a
// Monte Carlo accept/reject loop written against the reference's `System` reads the same against this one:
//
//   reference (src/System.h)                         here (namespace mpmc)
//   -----------------------------------------------  ---------------------------------------------------------
//   double System::energy()               :315        double System::energy()
//   lj / coulombic / coulombic_real / _reciprocal /   same names, same return values
//     _self / polar / thole_field         :346-402
//   observables_t *observables            :94-113     observables_t *observables (same field names)
//   nodestats->polarization_iterations    :151-185    nodestats->polarization_iterations
//   int iterator_failed                   :680        int iterator_failed
//   PeriodicBoundary pbc (+ update_pbc)               PeriodicBoundary pbc, update_pbc()
//   option fields rd_only, rd_lrc, polarization, polar_* , ewald_* (:510-831)   same names
//   errors: `throw <int>` (src/constants.h:108-147)   `throw <int>` with the same codes
//   SimulationControl::PI_calculate_potential()       PathIntegralEnsemble::PI_calculate_potential()
//     (src/SimulationControl.PathIntegral.cpp:752)
//
// Data model: flat vectors instead of Molecule -> Atom linked lists (the flattening is what the adapter of
// INTEGRATION.md does for the real reference objects).  Header-only; link with -lmpmc_energy.
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "mpmc_energy.h"

namespace mpmc {

enum { DAMPING_OFF = 0, DAMPING_LINEAR = 1, DAMPING_EXPONENTIAL = 2 }; // reference constants.h:66-70

// reference src/PeriodicBoundary.h
class PeriodicBoundary {
public:
	double cutoff = 0, volume = 0;
	double basis[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
	double reciprocal_basis[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
	void update() { // PeriodicBoundary::update, src/PeriodicBoundary.cpp:31-35
		int rc = mpmc_pbc_compute(&basis[0][0], &reciprocal_basis[0][0], &volume, &cutoff);
		if (rc != MPMC_OK) throw (int)MPMC_ERR_BOX;
	}
};

// reference src/System.h:94-113
struct observables_t {
	double energy = 0, coulombic_energy = 0, rd_energy = 0, polarization_energy = 0, vdw_energy = 0, three_body_energy = 0,
	       dipole_rrms = 0, kinetic_energy = 0, temperature = 0, volume = 0, N = 0, NU = 0, spin_ratio = 0, frozen_mass = 0, total_mass = 0;
	double potential() const { return coulombic_energy + rd_energy + polarization_energy + vdw_energy + three_body_energy; }
};
struct nodestats_t {
	double polarization_iterations = 0;
};

// one row of the flattened atom list (reference src/Atom.h:21-56, the fields the path reads / writes)
struct Atom {
	double pos[3] = {0, 0, 0};
	double mass = 0, charge = 0, polarizability = 0, epsilon = 0, sigma = 0;
	double c6 = 0, c8 = 0, c10 = 0;
	int frozen = 0;
	int molecule = 0; // index of the owning molecule (consecutive atoms with equal index form one Molecule)
	int moltype = -1; // index into System::moltype_names (the molecule-type column of the PQR row; -1: not recorded)
	// written by energy():
	double mu[3] = {0, 0, 0}, ef_static[3] = {0, 0, 0}, ef_induced[3] = {0, 0, 0};
};

class System {
public:
	// ---- options, reference names and defaults (src/System.h:510-831) ----
	int rd_only = 0, rd_lrc = 1;
	int polarization = 0, polar_iterative = 0, polar_ewald = 0, polar_max_iter = 10, polar_gs = 0, polar_rrms = 0;
	int damp_type = DAMPING_EXPONENTIAL;
	int ewald_kmax = 7;
	int wolf = 0, feynman_hibbs = 0, feynman_hibbs_order = 0;
	double temperature = 0;
	double polar_precision = 0, polar_gamma = 1.0, polar_damp = 0;
	double ewald_alpha = 0.5, polar_ewald_alpha = 0.5;
	int ewald_alpha_set = 0, polar_ewald_alpha_set = 0;
	uint64_t unsupported_flags = 0; // MPMC_FLAG_*: reference switches outside the hot path that are ON
	int solver = MPMC_SOLVER_AUTO;
	int device = 0;
	// the reference's energy() leaves mu / ef_static / ef_induced in every Atom, but reads them only when it writes the dipole and
	// field files (src/System.Output.cpp:1132-1229): a Monte Carlo driver may switch the per-call copy-back off and call
	// update_dipoles() when it samples
	bool eager_dipoles = true;

	// ---- state ----
	PeriodicBoundary pbc;
	std::vector<Atom> atoms;
	std::vector<std::string> moltype_names; // molecule-type labels met by read_pqr, in order of first appearance (Atom::moltype indexes it)
	int natoms = 0;
	int iterator_failed = 0;
	double last_volume = 0;
	observables_t *observables = &obs_;
	nodestats_t *nodestats = &stats_;
	mpmc_result last_result{};

	System() = default;
	System(const System &) = delete;
	System &operator=(const System &) = delete;
	~System() {
		if (ctx_) mpmc_ctx_destroy(ctx_);
	}

	// System::update_pbc, src/System.cpp:859-876
	void update_pbc() {
		pbc.update();
		if (ewald_alpha_set != 1) ewald_alpha = 3.5 / pbc.cutoff;
		if (polar_ewald_alpha_set != 1) polar_ewald_alpha = 3.5 / pbc.cutoff;
		box_dirty_ = true;
	}
	int countNatoms() const { return (int)atoms.size(); }
	// System::countN, src/System.cpp:909-931: molecules that are not frozen.  The flag of a molecule is that of its LAST atom row:
	// the PQR reader overwrites molecule->frozen on every row (src/System.cpp:684); adiabatic / target molecules are refused by the readers
	unsigned int countN() {
		unsigned int count = 0;
		for (size_t i = 0; i < atoms.size(); i++)
			if ((i + 1 == atoms.size() || atoms[i + 1].molecule != atoms[i].molecule) && !atoms[i].frozen) count++;
		observables->N = count;
		return count;
	}
	// Molecule::update_COM for every molecule (src/Molecule.cpp:259-281): com [n_molecules][3], mass, movable flag
	void molecule_coms(std::vector<double> &com, std::vector<double> &mol_mass, std::vector<int32_t> &movable) const {
		com.clear();
		mol_mass.clear();
		movable.clear();
		for (size_t a0 = 0; a0 < atoms.size();) {
			size_t a1 = a0;
			double m = 0, c[3] = {0, 0, 0};
			for (; a1 < atoms.size() && atoms[a1].molecule == atoms[a0].molecule; a1++) {
				m += atoms[a1].mass;
				for (int d = 0; d < 3; d++) c[d] += atoms[a1].mass * atoms[a1].pos[d];
			}
			for (int d = 0; d < 3; d++) com.push_back(c[d] / m);
			mol_mass.push_back(m);
			movable.push_back(atoms[a1 - 1].frozen ? 0 : 1); // last row decides (src/System.cpp:684)
			a0 = a1;
		}
	}

	// call after changing atom parameters or the number of atoms (positions alone: move_atoms)
	void atoms_changed() { atoms_dirty_ = true; }
	// after a Monte Carlo move that displaced atoms [first, first+count)
	void move_atoms(int first, int count) {
		if (atoms_dirty_ || !ctx_) return; // a full upload is pending anyway
		std::vector<double> p(3 * (size_t)count);
		for (int k = 0; k < count; k++)
			for (int d = 0; d < 3; d++) p[3 * k + d] = atoms[first + k].pos[d];
		check(mpmc_update_positions(ctx_, first, count, p.data()), "mpmc_update_positions");
	}

	// ---- double System::energy(), src/System.Energy.cpp:19-171 ----
	double energy() {
		sync_state();
		mpmc_result r;
		check(mpmc_energy(ctx_, &r), "mpmc_energy");
		absorb(r);
		return r.energy;
	}
	void energy_async() {
		sync_state();
		check(mpmc_hint_in_flight(ctx_, in_flight_hint_), "mpmc_hint_in_flight");
		check(mpmc_energy_async(ctx_), "mpmc_energy_async");
	}
	// how many evaluations the caller keeps in flight together with this system's (a scheduling hint for energy_async: mpmc_hint_in_flight;
	// kept here because the context is created lazily)
	void hint_in_flight(int n) { in_flight_hint_ = n < 1 ? 1 : n; }
	double energy_wait() {
		mpmc_result r;
		check(mpmc_energy_wait(ctx_, &r), "mpmc_energy_wait");
		absorb(r);
		return r.energy;
	}

	// ---- trial moves: the role of the reference's per-pair recalculate_energy cache (src/System.cpp:1211-1224) ----
	// energy of the configuration in which atoms [first, first+count) sit at new_pos ([count][3]); then accept_trial()
	// (atoms[].pos are updated to the trial positions) or reject_trial().  Needs a prior energy() of the accepted state.
	double energy_trial(int first, int count, const double *new_pos) {
		sync_state();
		check(mpmc_trial_begin(ctx_, first, count, new_pos), "mpmc_trial_begin");
		mpmc_result r;
		int rc = mpmc_trial_energy(ctx_, &r);
		if (rc != MPMC_OK) {
			mpmc_trial_reject(ctx_);
			check(rc, "mpmc_trial_energy");
		}
		trial_first_ = first;
		trial_pos_.assign(new_pos, new_pos + 3 * (size_t)count);
		trial_result_ = r;
		return r.energy;
	}
	// the same in two halves, so that a driver can put the trials of many Systems in flight before the first wait
	void energy_trial_async(int first, int count, const double *new_pos) {
		sync_state();
		check(mpmc_trial_begin(ctx_, first, count, new_pos), "mpmc_trial_begin");
		int rc = mpmc_trial_energy_async(ctx_);
		if (rc != MPMC_OK) {
			mpmc_trial_reject(ctx_);
			check(rc, "mpmc_trial_energy_async");
		}
		trial_first_ = first;
		trial_pos_.assign(new_pos, new_pos + 3 * (size_t)count);
	}
	double energy_trial_wait() {
		mpmc_result r;
		int rc = mpmc_trial_energy_wait(ctx_, &r);
		if (rc != MPMC_OK) {
			mpmc_trial_reject(ctx_);
			check(rc, "mpmc_trial_energy_wait");
		}
		trial_result_ = r;
		return r.energy;
	}
	const mpmc_result &trial_result() const { return trial_result_; } // the trial configuration's totals (observables keep the accepted ones)
	void accept_trial() {
		check(mpmc_trial_accept(ctx_), "mpmc_trial_accept");
		for (size_t k = 0; k < trial_pos_.size() / 3; k++)
			for (int d = 0; d < 3; d++) atoms[trial_first_ + k].pos[d] = trial_pos_[3 * k + d];
		absorb(trial_result_);
	}
	void reject_trial() { check(mpmc_trial_reject(ctx_), "mpmc_trial_reject"); }

	double lj() { return piece(mpmc_lj); }
	double coulombic() { return piece(mpmc_coulombic); }
	double coulombic_real() { return piece(mpmc_coulombic_real); }
	double coulombic_reciprocal() { return piece(mpmc_coulombic_reciprocal); }
	double coulombic_self() { return piece(mpmc_coulombic_self); }
	double polar() {
		double v = piece(mpmc_polar);
		fetch_dipoles();
		return v;
	}
	void thole_field() {
		sync_state();
		std::vector<double> e(3 * atoms.size());
		check(mpmc_thole_field(ctx_, e.data()), "mpmc_thole_field");
		for (size_t i = 0; i < atoms.size(); i++)
			for (int p = 0; p < 3; p++) atoms[i].ef_static[p] = e[3 * i + p];
	}
	// rows [row0, row0+nrows) of the dense 3N x 3N matrix of System::thole_amatrix (row-major nrows x 3N)
	std::vector<double> thole_amatrix(int row0, int nrows) {
		sync_state();
		std::vector<double> a((size_t)nrows * 3 * atoms.size());
		check(mpmc_thole_amatrix(ctx_, row0, nrows, a.data()), "mpmc_thole_amatrix");
		return a;
	}

	void update_dipoles() { // atoms[].mu / ef_static / ef_induced of the last evaluation
		if (polarization && !rd_only && ctx_) fetch_dipoles();
	}

	mpmc_ctx *context() {
		sync_state();
		return ctx_;
	}

private:
	observables_t obs_;
	nodestats_t stats_;
	mpmc_ctx *ctx_ = nullptr;
	int in_flight_hint_ = 1;
	int capacity_ = 0;
	bool atoms_dirty_ = true, box_dirty_ = true;
	int trial_first_ = 0;
	std::vector<double> trial_pos_;
	mpmc_result trial_result_{};

	void check(int rc, const char *what) {
		if (rc == MPMC_OK) return;
		last_error_ = std::string(what) + ": " + (ctx_ ? mpmc_last_error(ctx_) : mpmc_last_error(nullptr));
		throw (rc > 0 ? rc : (int)MPMC_ERR_INTERNAL); // reference convention: throw <int>
	}

	void sync_state() {
		natoms = countNatoms();
		const int n = natoms;
		if (!ctx_ || capacity_ < n) {
			if (ctx_) mpmc_ctx_destroy(ctx_);
			ctx_ = nullptr;
			capacity_ = n + n / 4 + 64;
			check(mpmc_ctx_create(device, capacity_, &ctx_), "mpmc_ctx_create");
			atoms_dirty_ = box_dirty_ = true;
		}
		if (box_dirty_) {
			check(mpmc_set_box(ctx_, &pbc.basis[0][0], &pbc.reciprocal_basis[0][0], pbc.volume, pbc.cutoff), "mpmc_set_box");
			box_dirty_ = false;
		}
		mpmc_options o;
		mpmc_default_options(&o);
		o.rd_only = rd_only;
		o.rd_lrc = rd_lrc;
		o.polarization = polarization;
		o.polar_iterative = polar_iterative;
		o.polar_ewald = polar_ewald;
		o.polar_max_iter = polar_max_iter;
		o.polar_gs = polar_gs;
		o.polar_rrms = polar_rrms;
		o.damp_type = damp_type;
		o.ewald_kmax = ewald_kmax;
		o.solver = solver;
		o.wolf = wolf;
		o.feynman_hibbs = feynman_hibbs;
		o.feynman_hibbs_order = feynman_hibbs_order;
		o.temperature = temperature;
		o.polar_precision = polar_precision;
		o.polar_gamma = polar_gamma;
		o.polar_damp = polar_damp;
		o.ewald_alpha = ewald_alpha;
		o.polar_ewald_alpha = polar_ewald_alpha;
		o.unsupported_flags = unsupported_flags | ((polarization && !polar_iterative) ? MPMC_FLAG_POLAR_MATRIX_INVERSION : 0);
		check(mpmc_set_options(ctx_, &o), "mpmc_set_options");
		if (atoms_dirty_) {
			std::vector<double> pos(3 * (size_t)n), q(n), al(n), ep(n), sg(n), ms(n);
			std::vector<int32_t> mol(n), fr(n), dp(n);
			for (int i = 0; i < n; i++) {
				const Atom &a = atoms[i];
				for (int p = 0; p < 3; p++) pos[3 * i + p] = a.pos[p];
				q[i] = a.charge;
				al[i] = a.polarizability;
				ep[i] = a.epsilon;
				sg[i] = a.sigma;
				ms[i] = a.mass;
				mol[i] = a.molecule;
				fr[i] = a.frozen;
				dp[i] = (a.c6 != 0.0 || a.c8 != 0.0 || a.c10 != 0.0) ? 1 : 0;
			}
			check(mpmc_set_atoms(ctx_, n, pos.data(), q.data(), al.data(), ep.data(), sg.data(), mol.data(), fr.data(), dp.data(), ms.data()),
			      "mpmc_set_atoms");
			atoms_dirty_ = false;
		}
	}

	void fetch_dipoles() {
		const size_t n = atoms.size();
		std::vector<double> mu(3 * n), e0(3 * n), ei(3 * n);
		check(mpmc_get_dipoles(ctx_, mu.data(), e0.data(), ei.data()), "mpmc_get_dipoles");
		for (size_t i = 0; i < n; i++)
			for (int p = 0; p < 3; p++) {
				atoms[i].mu[p] = mu[3 * i + p];
				atoms[i].ef_static[p] = e0[3 * i + p];
				atoms[i].ef_induced[p] = ei[3 * i + p];
			}
	}

	void absorb(const mpmc_result &r) {
		last_result = r;
		obs_.energy = r.energy;
		obs_.coulombic_energy = r.coulombic_energy;
		obs_.rd_energy = r.rd_energy;
		obs_.polarization_energy = r.polarization_energy;
		obs_.vdw_energy = r.vdw_energy;
		obs_.three_body_energy = r.three_body_energy;
		obs_.dipole_rrms = r.dipole_rrms;
		obs_.N = r.N;
		obs_.NU = r.NU;
		stats_.polarization_iterations = (double)r.polar_iterations;
		iterator_failed = r.iterator_failed;
		last_volume = pbc.volume;
		if (polarization && !rd_only && eager_dipoles) fetch_dipoles();
	}

	double piece(int (*fn)(mpmc_ctx *, double *)) {
		sync_state();
		double v = 0;
		check(fn(ctx_, &v), "component");
		return v;
	}

public:
	std::string last_error_;
};

// SimulationControl's multi-System part for ensemble pi_nvt (src/SimulationControl.h:60, PathIntegral.cpp:752-805).
// `systems` holds the beads owned by THIS process; `nSys` is the Trotter number P over all processes.
// The cross-process exchange of 4 doubles per bead (MPI_Allgather x4 in the reference) is delegated to
// `allgather`: it receives this process's per-bead values (n_local x stride, local order; stride = 4 for the
// potential, 3 * n_molecules for the centres of mass of the kinetic estimator) and must return all P x stride
// values in bead order.  With one process it may be left empty.
// (The reference keeps all P images on every MPI rank, so its kinetic estimator needs no exchange; with the beads
// sharded over ranks the ring of adjacent images crosses ranks and the centres of mass are gathered once per call.)
// one image's share of a loop that may run under OpenMP: the facade reports errors by throwing an int, and an exception must not
// leave a parallel region -- the first one is kept and thrown again behind the loop
template <class F>
inline void each_image_guarded(int &first_error, F &&body) {
	try {
		body();
	} catch (int e) {
#ifdef _OPENMP
#pragma omp critical(mpmc_image_error)
#endif
		if (!first_error) first_error = e;
	}
}

template <class SystemT>
class PathIntegralEnsembleT {
public:
	std::vector<SystemT *> systems;
	int nSys = 0;
	double temperature = 0;        // sys.temperature
	observables_t sys_observables; // the aggregate "sys.observables" of the reference
	std::function<std::vector<double>(const std::vector<double> &)> allgather;

	// one process per GPU: the exchange runs on RCCL inside libmpmc_energy.so (mpmc_pi_gather_beads: ONE ncclAllGather per call, the
	// reference's 4 x MPI_Allgather, PathIntegral.cpp:763-766).  `comm` comes from mpmc_comm_init_rank; nSys = P over all ranks.
	void use_comm(mpmc_comm *comm) {
		allgather = [this, comm](const std::vector<double> &mine) {
			const int n_local = (int)systems.size();
			int n_ranks = 1;
			(void)mpmc_comm_info(comm, &n_ranks, nullptr, nullptr);
			const int stride = n_local > 0 ? (int)(mine.size() / (size_t)n_local) : 0;
			std::vector<double> all((size_t)n_ranks * mine.size());
			const int rc = mpmc_pi_gather_beads(comm, mine.data(), n_local, stride, all.data());
			if (rc != MPMC_OK) throw rc;
			return all;
		};
	}

	// PI_calculate_potential for a TRIAL configuration in which atoms [first, first+count) of every image sit at new_pos[image]:
	// per-move delta energies behind the same aggregate (every image enqueued before the first wait); the Systems keep the accepted
	// configuration until accept_trial() / reject_trial()
	double PI_trial_potential(int first, int count, const std::vector<std::vector<double>> &new_pos) {
		const int n_local = (int)systems.size();
		int err_img = 0;
		// The images' enqueues are independent (one context, one stream each) and a trial is bound by the host's HIP calls, not by the GPU:
		// built with -fopenmp the images are enqueued by a few threads, as the reference's own bead loop is (PathIntegral.cpp:759-775);
		// without it this is the plain loop.  The waits and the sums below stay in image order: the result does not depend on threads.
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_local < 4 ? n_local : 4) if (n_local > 1)
#endif
		for (int b = 0; b < n_local; b++) each_image_guarded(err_img, [&] { systems[b]->energy_trial_async(first, count, new_pos[b].data()); });
		if (err_img) throw err_img;
		std::vector<double> mine(4 * (size_t)n_local);
		for (int b = 0; b < n_local; b++) {
			systems[b]->energy_trial_wait();
			const mpmc_result &r = systems[b]->trial_result();
			mine[4 * b + 0] = r.rd_energy;
			mine[4 * b + 1] = r.coulombic_energy;
			mine[4 * b + 2] = r.polarization_energy;
			mine[4 * b + 3] = r.vdw_energy;
		}
		const std::vector<double> all = allgather ? allgather(mine) : mine;
		const int P = nSys ? nSys : n_local;
		double acc[4] = {0, 0, 0, 0};
		for (int s = 0; s < P; s++)
			for (int k = 0; k < 4; k++) acc[k] += all[4 * (size_t)s + k];
		for (int k = 0; k < 4; k++) acc[k] /= P;
		sys_observables.rd_energy = acc[0];
		sys_observables.coulombic_energy = acc[1];
		sys_observables.polarization_energy = acc[2];
		sys_observables.vdw_energy = acc[3];
		return acc[0] + acc[1] + acc[3] + acc[2];
	}

	// SimulationControl::PI_calculate_energy, PathIntegral.cpp:734-749
	double PI_calculate_energy() {
		const double kinetic = PI_calculate_kinetic();
		const double potential = PI_calculate_potential();
		sys_observables.energy = kinetic + potential;
		return sys_observables.energy;
	}

	// SimulationControl::PI_calculate_kinetic, PathIntegral.cpp:806-824
	double PI_calculate_kinetic() {
		const double N = (double)systems.at(0)->countN();
		const int P = nSys ? nSys : (int)systems.size();
		const double chain_mass_len2 = PI_chain_mass_length2_ENTIRE_SYSTEM();
		const double orient_mu_len2 = PI_orientational_mu_length2_ENTIRE_SYSTEM();
		sys_observables.N = N;
		sys_observables.kinetic_energy = mpmc_pi_kinetic(chain_mass_len2, orient_mu_len2, N, P, temperature);
		return sys_observables.kinetic_energy;
	}

	// SimulationControl::PI_chain_mass_length2_ENTIRE_SYSTEM, PathIntegral.cpp:851-896
	double PI_chain_mass_length2_ENTIRE_SYSTEM() {
		std::vector<double> com, mass;
		std::vector<int32_t> movable;
		const int nmol = gather_coms(com, mass, movable);
		const int P = nSys ? nSys : (int)systems.size();
		return mpmc_pi_chain_mass_length2(P, nmol, com.data(), mass.data(), movable.data());
	}
	// SimulationControl::PI_chain_mass_length2() for one chain, PathIntegral.cpp:897-965 (`molecule` plays the part of
	// checkpoint->molecule_altered: the index of the molecule whose bead chain was perturbed)
	double PI_chain_mass_length2(int molecule) {
		std::vector<double> com, mass;
		std::vector<int32_t> movable;
		const int nmol = gather_coms(com, mass, movable);
		if (molecule < 0 || molecule >= nmol) throw (int)MPMC_ERR_ARG;
		const int P = nSys ? nSys : (int)systems.size();
		std::vector<double> one(3 * (size_t)P);
		for (int s = 0; s < P; s++)
			for (int d = 0; d < 3; d++) one[3 * s + d] = com[3 * ((size_t)s * nmol + molecule) + d];
		return mpmc_pi_chain_mass_length2(P, 1, one.data(), &mass[molecule], nullptr);
	}
	double PI_orientational_mu_length2_ENTIRE_SYSTEM() { return 0.0; } // PathIntegral.cpp:970-972

private:
	int gather_coms(std::vector<double> &com, std::vector<double> &mass, std::vector<int32_t> &movable) {
		std::vector<double> mine, c, m0;
		std::vector<int32_t> mv;
		for (size_t b = 0; b < systems.size(); b++) {
			systems[b]->molecule_coms(c, b == 0 ? mass : m0, b == 0 ? movable : mv);
			mine.insert(mine.end(), c.begin(), c.end());
		}
		com = allgather ? allgather(mine) : mine;
		return (int)mass.size();
	}

public:

	double PI_calculate_potential() {
		const int n_local = (int)systems.size();
		int err_img = 0;
		// every bead enqueued on its own stream before the first wait (by a few threads when built with -fopenmp, see PI_trial_potential)
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_local < 4 ? n_local : 4) if (n_local > 1)
#endif
		for (int b = 0; b < n_local; b++)
			each_image_guarded(err_img, [&] {
				systems[b]->hint_in_flight(n_local);
				systems[b]->energy_async();
			});
		if (err_img) throw err_img;
		std::vector<double> mine(4 * (size_t)n_local);
		for (int b = 0; b < n_local; b++) {
			systems[b]->energy_wait();
			const observables_t *o = systems[b]->observables;
			mine[4 * b + 0] = o->rd_energy;
			mine[4 * b + 1] = o->coulombic_energy;
			mine[4 * b + 2] = o->polarization_energy;
			mine[4 * b + 3] = o->vdw_energy;
		}
		const std::vector<double> all = allgather ? allgather(mine) : mine;
		const int P = nSys ? nSys : n_local;
		double acc[4] = {0, 0, 0, 0};
		for (int s = 0; s < P; s++) // ordered sum, :791-796
			for (int k = 0; k < 4; k++) acc[k] += all[4 * (size_t)s + k];
		for (int k = 0; k < 4; k++) acc[k] /= P; // :798-801
		sys_observables.rd_energy = acc[0];
		sys_observables.coulombic_energy = acc[1];
		sys_observables.polarization_energy = acc[2];
		sys_observables.vdw_energy = acc[3];
		return acc[0] + acc[1] + acc[3] + acc[2]; // :803-804
	}
};
using PathIntegralEnsemble = PathIntegralEnsembleT<System>;

} // namespace mpmc
