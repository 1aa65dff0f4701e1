// include/mpmc_pimc.hpp -- path-integral NVT Monte Carlo on top of the energy path.
//
// The steps either side of the hot path, written against the C++ facade (include/mpmc_system.hpp) so that a GPU box runs an
// end-to-end PIMC without the reference's sources.  Behaviour follows the reference's driver step for step -- the same
// random-number stream (std::mt19937 + the standard uniform / normal distributions, reference src/Rando.h), the same order of
// draws and the same floating-point association -- so a run reproduces the stock binary's `energy.dat` rows
// (tests/golden/pi001, pi_ion27).  Reference functions mirrored (src/SimulationControl.PathIntegral.cpp):
//   PI_nvt_mc                    :31-198     main loop, Metropolis test, accept / restore
//   PI_NVT_boltzmann_factor      :490-547
//   PI_pick_NVT_move             :1047-1116
//   PI_displace                  :1320-1387  common random translation (+ one random rotation about the chain's centre)
//   PI_perturb_bead_COMs(n)      :1453-1554  Coker et al. J. Chem. Phys. 86, 5689 (1987) staging of n beads
//   PI_perturb_bead_COMs_ENTIRE_SYSTEM :1402-1449
//   PI_calculate_energy / kinetic / potential :734-824 (mpmc::PathIntegralEnsembleT)
//   Molecule::update_COM / translate / translate_rand_pbc / move_to_   src/Molecule.cpp:259-345
//   PI_perturb_beads_orientations / generate_orientation_configs / apply_orientation_configs :1559-1698, Molecule::orient
//                                src/Molecule.cpp:211-254, PI_orientational_mu_length2 :978-1039 (molecule types that carry
//                                sorbate_orientation_site / sorbate_bondlength / sorbate_reducedMass entries)
//   rotations                    include/mpmc_rotation.hpp (arithmetic of src/Quaternion.cpp:16-127, src/Vector3D.*)
// Scope: rigid molecules, no spin flips, no insert / remove, no simulated annealing.  All P images live in this process (one context per image; images may sit on different devices).
#pragma once

#include <array>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "mpmc_io.hpp"
#include "mpmc_rotation.hpp"
#include "mpmc_system.hpp"

namespace mpmc {

// reference src/Rando.h: ONE engine, one uniform and one normal distribution object shared by every draw of the run
class Rando {
public:
	void seed(unsigned int s) { mt.seed(s); }
	double rand() { return uniform_distribution(mt); }
	double rand_normal() { return normal_distribution(mt); }

private:
	std::mt19937 mt;
	std::normal_distribution<double> normal_distribution{0.0, 1.0};
	std::uniform_real_distribution<double> uniform_distribution{0.0, 1.0};
};

enum { MOVETYPE_DISPLACE = 2, MOVETYPE_PERTURB_BEADS = 6 };

// per-molecule-type data of the orientational bead moves (input keys sorbate_orientation_site / sorbate_bondlength / sorbate_reducedMass;
// reference table: src/SimulationControl.cpp:2975-3072).  Records keep the order in which their molecule type was first named.
struct SorbateRecord {
	std::string moltype;
	int orientation_site = -1;
	double bond_length = 0, reduced_mass = 0; // Angstrom, kg
};

struct PimcSettings {
	std::vector<SorbateRecord> sorbates;
	SorbateRecord &sorbate(const std::string &moltype) { // the record of a type, appended when new
		for (SorbateRecord &r : sorbates)
			if (r.moltype == moltype) return r;
		sorbates.push_back(SorbateRecord());
		sorbates.back().moltype = moltype;
		return sorbates.back();
	}
	int sorbate_position(const std::string &moltype) const {
		for (size_t k = 0; k < sorbates.size(); k++)
			if (sorbates[k].moltype == moltype) return (int)k;
		return -1;
	}
	std::string job_name = "job";
	unsigned int numsteps = 0, corrtime = 0, seed = 0;
	bool seed_set = false, parallel_restarts = false;
	double move_factor = 1.0, rot_factor = 1.0, bead_perturb_probability = 0.0, temperature = 0.0; // src/System.h:526-538
	int PI_trial_chain_length = 0;
};

// the Monte Carlo keywords of a reference input file (src/SimulationControl.cpp:204-267, :801-863)
inline PimcSettings read_pimc_settings(const std::string &path) {
	using namespace io_detail;
	std::ifstream f(path);
	if (!f) throw 1000; // fopen_fail_read
	PimcSettings c;
	std::string line;
	while (std::getline(f, line)) {
		const std::vector<std::string> t = tokens(line);
		if (t.size() < 2 || t[0][0] == '!' || t[0][0] == '#') continue;
		const std::string k = lower(t[0]);
		double v = 0;
		if (k == "job_name") c.job_name = t[1];
		else if (k == "ensemble") {
			if (lower(t[1]) != "pi_nvt") throw 4004; // unsupported_setting: this driver is the pi_nvt ensemble
		} else if (k == "parallel_restarts") c.parallel_restarts = onoff(t[1]) != 0;
		else if (k == "spinflip_probability" || k == "simulated_annealing") {
			if (!to_double(t[1], v) || v != 0.0) throw 4004;
		} else if (k == "feynman_hibbs") {
			if (onoff(t[1])) throw 3000; // invalid_input: the reference refuses Feynman-Hibbs corrections in a path-integral run (SimulationControl.cpp:1938-1944)
		} else if (k == "sorbate_orientation_site" || k == "sorbate_bondlength" || k == "sorbate_reducedmass") {
			if (t.size() < 3 || !to_double(t[2], v)) throw 3000; // invalid_input (src/SimulationControl.cpp:306-339)
			SorbateRecord &r = c.sorbate(t[1]);
			if (k == "sorbate_orientation_site") r.orientation_site = (int)v;
			else if (k == "sorbate_bondlength") r.bond_length = v;
			else r.reduced_mass = v;
		} else if (k == "numsteps" || k == "corrtime" || k == "seed" || k == "move_factor" || k == "rot_factor" || k == "bead_perturb_probability" ||
		           k == "temperature" || k == "pi_trial_chain_length") {
			if (!to_double(t[1], v)) throw 3000; // invalid_input
			if (k == "numsteps") c.numsteps = (unsigned int)v;
			else if (k == "corrtime") c.corrtime = (unsigned int)v;
			else if (k == "seed") { c.seed = (unsigned int)v; c.seed_set = true; }
			else if (k == "move_factor") c.move_factor = v;
			else if (k == "rot_factor") c.rot_factor = v;
			else if (k == "bead_perturb_probability") c.bead_perturb_probability = v;
			else if (k == "temperature") c.temperature = v;
			else c.PI_trial_chain_length = (int)v;
		}
	}
	if (c.bead_perturb_probability > 1.0) throw 3000; // "probabilities for all MC moves sum to a value greater than 1.0" (SimulationControl.cpp:1946-1949)
	return c;
}

template <class SystemT>
class PathIntegralNVT {
public:
	PimcSettings cfg;
	std::vector<SystemT *> systems; // the P images, all local
	std::vector<std::string> moltype_names; // System::moltype_names of the PQR the images were read from (only the orientational bead moves look at it)
	PathIntegralEnsembleT<SystemT> pi;
	Rando rng;
	unsigned int step = 0;
	// acceptance bookkeeping (System::register_accept / register_reject, src/System.MonteCarlo.cpp:1475-1760)
	long accept = 0, reject = 0, accept_displace = 0, reject_displace = 0, accept_bead = 0, reject_bead = 0;
	long energy_calls = 0; // evaluations of System::energy() (or per-move delta evaluations) issued so far
	double boltzmann_factor = 0;
	// per-move delta energies instead of full evaluations: the images keep the accepted configuration resident and
	// evaluate only the pairs of the moved molecule (LJ / Ewald boxes; polarizable boxes run a full evaluation behind the same calls)
	bool use_trial_moves = false;

	void init() {
		nSys = (int)systems.size();
		int bits = 0;
		for (unsigned v = (unsigned)nSys; v; v >>= 1) bits += (int)(v & 1u);
		if (nSys < 4 || bits != 1) throw 9003;                                                   // invalid_MPI_size_for_PI (:568)
		if (!cfg.PI_trial_chain_length || cfg.PI_trial_chain_length >= nSys) throw 4001;           // invalid_setting (:582-599)
		if (!cfg.numsteps || !cfg.corrtime || !(cfg.temperature > 0)) throw 4001;
		mol_first.clear();
		const std::vector<Atom> &a = systems[0]->atoms;
		for (size_t i = 0; i < a.size(); i++)
			if (i == 0 || a[i].molecule != a[i - 1].molecule) mol_first.push_back((int)i);
		mol_first.push_back((int)a.size());
		for (SystemT *s : systems) {
			if (s->atoms.size() != a.size()) throw 9000; // internal_error: images are not consistent (:1427)
			s->temperature = cfg.temperature;
		}
		pi.systems = systems;
		pi.nSys = nSys;
		pi.temperature = cfg.temperature;
		rng.seed(cfg.seed);
		chain_start = 0;
		com.assign(nSys, {{0, 0, 0}});
		mol_mass.assign(nSys, 0.0);
		backup_pos.assign(nSys, {});
		checkpoint_obs.assign(nSys, observables_t());
	}

	// SimulationControl::PI_nvt_mc, :31-198.  fp_energy receives the rows of JOB.energy.dat (System::write_observables,
	// src/System.Output.cpp:251-271); on_sample (optional) is called after each row.
	bool run(FILE *fp_energy, void (*on_sample)(PathIntegralNVT &, void *) = nullptr, void *user = nullptr) {
		observables_t &obs = pi.sys_observables;
		for (SystemT *s : systems) {
			s->observables->temperature = cfg.temperature;
			s->observables->volume = s->pbc.volume;
		}
		if (!cfg.parallel_restarts) PI_perturb_bead_COMs_ENTIRE_SYSTEM();
		PI_calculate_energy(true);
		refresh_aggregate();
		if (fp_energy) {
			std::fprintf(fp_energy, "#step #energy #coulombic #rd #polar #vdw #kinetic #kin_temp #N #spin_ratio #volume #core_temp\n");
			write_observables(fp_energy);
		}
		if (on_sample) on_sample(*this, user);

		int move = PI_pick_NVT_move();
		backup_observables_ALL_SYSTEMS();
		double potential_current = obs.potential();
		if (!std::isfinite(potential_current)) obs.energy = potential_current = kMaxValue;

		for (step = 1; step <= cfg.numsteps; step++) {
			const double potential_init = potential_current;
			const double chain_init = (move == MOVETYPE_PERTURB_BEADS) ? PI_chain_mass_length2() : 0;
			const double orient_init = (move == MOVETYPE_PERTURB_BEADS) ? PI_orientational_mu_length2() : 0;
			PI_make_move(move, !use_trial_moves);
			const double chain_trial = (move == MOVETYPE_PERTURB_BEADS) ? PI_chain_mass_length2() : 0;
			const double orient_trial = (move == MOVETYPE_PERTURB_BEADS) ? PI_orientational_mu_length2() : 0;
			double potential_trial = use_trial_moves ? PI_trial_potential() : PI_calculate_potential();
			const int failed = use_trial_moves ? systems[0]->trial_result().iterator_failed : systems[0]->iterator_failed;
			if (!std::isfinite(potential_trial)) { // a bad contact is a reject (:128-131)
				potential_trial = obs.energy = kMaxValue;
				boltzmann_factor = 0;
			} else {
				boltzmann_factor = PI_NVT_boltzmann_factor(move, potential_trial - potential_init, chain_trial - chain_init, orient_trial - orient_init);
			}
			if ((rng.rand() < boltzmann_factor) && (failed == 0)) {
				register_move(move, true);
				potential_current = potential_trial;
				if (use_trial_moves)
					{
						const int n_img = (int)systems.size();
						int err_img = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_img < 4 ? n_img : 4) if (n_img > 1)
#endif
						for (int b = 0; b < n_img; b++) each_image_guarded(err_img, [&] { systems[b]->accept_trial(); });
						if (err_img) throw err_img;
					}
				// same geometry: the potential of the trial evaluation stands, the kinetic part is new.  It is a pure function of
				// the geometry (O(P N) host work), so the delta-energy mode computes it when a row is written instead of per accept.
				if (!use_trial_moves) PI_calculate_energy(false);
				backup_observables_ALL_SYSTEMS();
			} else {
				if (use_trial_moves) {
					for (SystemT *s : systems) {
						s->reject_trial();
						s->iterator_failed = 0;
					}
				} else {
					restore_PI_systems();
				}
				obs = checkpoint_sys_obs;
				register_move(move, false);
			}
			move = PI_pick_NVT_move();
			if (!(step % cfg.corrtime) || (step == cfg.numsteps)) {
				if (use_trial_moves) PI_calculate_energy(false);
				refresh_aggregate();
				if (fp_energy) write_observables(fp_energy);
				if (on_sample) on_sample(*this, user);
			}
		}
		step = cfg.numsteps;
		return true;
	}

	double acceptance_rate() const { return (accept + reject) ? (double)accept / (double)(accept + reject) : 0.0; }

	// ---- pieces (public so that tests can drive them one by one) --------------------------------------------------------
	// SimulationControl::PI_pick_NVT_move, :1047-1116
	int PI_pick_NVT_move() {
		const double u_move = rng.rand();
		const double u_target = rng.rand();
		std::vector<int> perturbable;
		const std::vector<Atom> &a = systems[0]->atoms;
		for (size_t m = 0; m + 1 < mol_first.size(); m++)
			if (!a[mol_first[m + 1] - 1].frozen) perturbable.push_back((int)m); // Molecule::frozen = flag of the last atom row (src/System.cpp:684)
		if (perturbable.empty()) throw 3001; // no_molecules_in_system
		target = perturbable[(int)std::floor(perturbable.size() * u_target)];
		movetype = (u_move < cfg.bead_perturb_probability) ? MOVETYPE_PERTURB_BEADS : MOVETYPE_DISPLACE;
		const int first = mol_first[target], count = mol_first[target + 1] - first;
		for (int s = 0; s < nSys; s++) { // checkpoint->molecule_backup = copy of the molecule about to be altered (:1107)
			backup_pos[s].resize(3 * (size_t)count);
			for (int k = 0; k < count; k++)
				for (int d = 0; d < 3; d++) backup_pos[s][3 * k + d] = systems[s]->atoms[first + k].pos[d];
		}
		return movetype;
	}

	void PI_make_move(int mv, bool upload = true) { // :1121-1160
		if (mv == MOVETYPE_DISPLACE) PI_displace();
		else if (mv == MOVETYPE_PERTURB_BEADS) { // PI_perturb_beads :1392-1397
			PI_perturb_beads_orientations();
			PI_perturb_bead_COMs(cfg.PI_trial_chain_length);
		}
		else throw 12000; // invalid_monte_carlo_move
		if (!upload) return;
		const int first = mol_first[target], count = mol_first[target + 1] - first;
		for (SystemT *s : systems) s->move_atoms(first, count);
	}

	// the potential of the configuration PI_make_move just produced, through the per-move delta path: the new positions of the
	// altered molecule go to the images as a TRIAL (their atoms[] return to the accepted configuration until accept_trial())
	double PI_trial_potential() {
		const int first = mol_first[target], count = mol_first[target + 1] - first;
		std::vector<std::vector<double>> trial(nSys, std::vector<double>(3 * (size_t)count));
		for (int s = 0; s < nSys; s++)
			for (int k = 0; k < count; k++)
				for (int d = 0; d < 3; d++) {
					trial[s][3 * k + d] = systems[s]->atoms[first + k].pos[d];
					systems[s]->atoms[first + k].pos[d] = backup_pos[s][3 * k + d];
				}
		energy_calls += nSys;
		const double v = pi.PI_trial_potential(first, count, trial);
		// the rest of the step (chain measures were taken before; an accepted move re-reads atoms[]) sees the trial geometry only
		// after accept_trial(), which is what the estimator needs
		return v;
	}

	// SimulationControl::PI_displace, :1320-1387
	void PI_displace() {
		double draws[6];
		for (int i = 0; i < 6; i++) draws[i] = rng.rand();
		double pi_com[3] = {0, 0, 0};
		for (int s = 0; s < nSys; s++) {
			update_COM(s);
			translate_rand_pbc(s, cfg.move_factor, systems[s]->pbc.cutoff, draws);
			for (int d = 0; d < 3; d++) pi_com[d] = pi_com[d] + com[s][d];
		}
		for (int d = 0; d < 3; d++) pi_com[d] /= nSys;
		const double axis_x = rng.rand_normal();
		const double axis_y = rng.rand_normal();
		const double axis_z = rng.rand_normal();
		const double turn_deg = rng.rand() * cfg.rot_factor;
		const Rotor spin = Rotor::about_axis_degrees(axis_x, axis_y, axis_z, turn_deg);
		const int first = mol_first[target], last = mol_first[target + 1];
		for (int s = 0; s < nSys; s++) {
			translate(s, -pi_com[0], -pi_com[1], -pi_com[2]);
			for (int i = first; i < last; i++) {
				double *p = systems[s]->atoms[i].pos;
				const Vec3 r = spin.turn_left_first(Vec3{{p[0], p[1], p[2]}});
				p[0] = r[0];
				p[1] = r[1];
				p[2] = r[2];
			}
			translate(s, pi_com[0], pi_com[1], pi_com[2]);
			update_COM(s);
		}
	}

	// SimulationControl::PI_perturb_bead_COMs(int n), :1453-1554
	void PI_perturb_bead_COMs(int n) {
		const double kB = 1.3806503e-23, hBar2 = 1.11211999e-68, AMU2KG = 1.66053873e-27, METER2ANGSTROM = 1.0e10; // src/constants.h
		const double beta = 1.0 / (kB * cfg.temperature);
		const double P = (double)nSys;
		update_COM(0);
		const double Mass = AMU2KG * mol_mass[0];
		int anchor = chain_start;
		int placed = (anchor + 1) % nSys;
		const int far_end = (anchor + n + 1) % nSys;
		chain_start = (chain_start + 1) % nSys;

		std::vector<std::array<double, 3>> beads(nSys);
		double chain_COM[3] = {0, 0, 0};
		for (int s = 0; s < nSys; s++) {
			update_COM(s);
			beads[s] = com[s];
			for (int d = 0; d < 3; d++) chain_COM[d] = chain_COM[d] + com[s][d];
		}
		for (int d = 0; d < 3; d++) chain_COM[d] /= P;

		double tB = (double)n;
		double tA = 1.0 + n;
		for (int j = 1; j <= n; j++) {
			const double init_factor = tB-- / tA--;
			const double term_factor = 1.0 - init_factor;
			const double sigma_factor = std::sqrt((hBar2 * beta * init_factor) / (P * Mass)) * METER2ANGSTROM;
			// `Vector3D perturbation(rand_normal(), rand_normal(), rand_normal())` at :1524: the order in which the three arguments
			// are evaluated is unspecified in C++; the reference as built by g++ (the stock binary of the goldens) evaluates them
			// right to left, i.e. the FIRST draw lands in z.
			double perturbation[3];
			perturbation[2] = rng.rand_normal();
			perturbation[1] = rng.rand_normal();
			perturbation[0] = rng.rand_normal();
			for (int d = 0; d < 3; d++)
				beads[placed][d] = ((init_factor * beads[anchor][d]) + (term_factor * beads[far_end][d])) + (sigma_factor * perturbation[d]);
			anchor = (anchor + 1) % nSys;
			placed = (anchor + 1) % nSys;
		}
		double delta_COM[3] = {0, 0, 0};
		for (int s = 0; s < nSys; s++)
			for (int d = 0; d < 3; d++) delta_COM[d] = delta_COM[d] + beads[s][d];
		for (int d = 0; d < 3; d++) delta_COM[d] = (delta_COM[d] / P) - chain_COM[d];
		for (int s = 0; s < nSys; s++)
			for (int d = 0; d < 3; d++) beads[s][d] = beads[s][d] - delta_COM[d];
		for (int s = 0; s < nSys; s++) // Molecule::move_to_ :326-328
			translate(s, beads[s][0] - com[s][0], beads[s][1] - com[s][1], beads[s][2] - com[s][2]);
	}

	// SimulationControl::PI_perturb_bead_COMs_ENTIRE_SYSTEM, :1402-1449
	void PI_perturb_bead_COMs_ENTIRE_SYSTEM() {
		const int saved = target;
		const std::vector<Atom> &a = systems[0]->atoms;
		for (size_t m = 0; m + 1 < mol_first.size(); m++) {
			if (a[mol_first[m + 1] - 1].frozen) continue;
			target = (int)m;
			PI_perturb_bead_COMs(nSys);
		}
		target = saved;
		for (SystemT *s : systems) s->atoms_changed();
	}

	// ---- orientational bead moves (:1559-1698): a molecule type with an orientation site and a bond length gets, before the COM
	// perturbation, one orientation per image from the bisection sampler of Subramanian et al., J. Chem. Phys. 146, 094105 (2017)
	int sorbate_of_target() const {
		const Atom &last = systems[0]->atoms[mol_first[target + 1] - 1]; // Molecule::moleculetype = the label of its last row (src/System.cpp:681)
		if (last.moltype < 0 || (size_t)last.moltype >= moltype_names.size()) return -1;
		return cfg.sorbate_position(moltype_names[last.moltype]);
	}
	// The reference's get_orientation_site returns the POSITION of the type's record in its table, not the site number stored in
	// it (src/SimulationControl.cpp:2996-3004: `return (int) it->second`): the first type named in the input orients by its atom 0,
	// the second by its atom 1, ...  The stock binary's trajectories are made that way, so this driver does the same.
	int orientation_handle() const {
		const int k = sorbate_of_target();
		if (k >= 0 && k >= mol_first[target + 1] - mol_first[target]) throw 9000; // (the reference walks off the molecule's atom list here)
		return k;
	}
	double sorbate_bond_length() const {
		const int k = sorbate_of_target();
		return k < 0 ? 0.0 : cfg.sorbates[k].bond_length;
	}

	// SimulationControl::PI_orientational_mu_length2, :978-1039
	double PI_orientational_mu_length2() {
		const int handle = orientation_handle();
		const double bond_length = sorbate_bond_length();
		if (handle < 0 || bond_length <= 0) return 0.0;
		const double ANGSTROM2METER = 1.0e-10;
		std::vector<Vec3> bonds;
		for (int s = 0; s < nSys; s++) {
			update_COM(s);
			const double *hp = systems[s]->atoms[mol_first[target] + handle].pos;
			const Vec3 from_com = sub3(Vec3{{hp[0], hp[1], hp[2]}}, com[s]);
			bonds.push_back(scaled3(bond_length, unit3(from_com)));
		}
		double sum = 0.0;
		for (int i = 0; i < nSys; i++) {
			const Vec3 d = sub3(bonds[i], bonds[(i + 1) % nSys]);
			sum += dot3(d, d);
		}
		sum *= (ANGSTROM2METER * ANGSTROM2METER);
		return sum;
	}

	void PI_perturb_beads_orientations() { // :1559-1570
		const int handle = orientation_handle();
		if (handle < 0 || sorbate_bond_length() <= 0) return;
		generate_orientation_configs();
		for (int s = 0; s < nSys; s++) orient(s, orientations[s], handle); // apply_orientation_configs :1684-1698
	}

	void generate_orientation_configs() { // :1575-1598
		const double kB = 1.3806503e-23, METER2ANGSTROM = 1.0e10;
		const int k = sorbate_of_target();
		const double reduced_mass = k < 0 ? -1.0 : cfg.sorbates[k].reduced_mass;
		if (reduced_mass < 0) throw 6000; // missing_required_datum
		double bond = sorbate_bond_length();
		if (bond < 0) throw 6000;
		bond /= METER2ANGSTROM;
		const double b2 = bond * bond;
		const double mu_kT = reduced_mass * kB * cfg.temperature;
		orientations.assign(nSys, Vec3{{0, 0, 0}});
		Vec3 first;
		first[0] = rng.rand_normal(); // Vector3D::randomize, src/Vector3D.cpp:119-124
		first[1] = rng.rand_normal();
		first[2] = rng.rand_normal();
		orientations[0] = unit3(first);
		place_orientations(0, (unsigned)nSys, 2, (unsigned)nSys, b2, mu_kT);
	}
	// :1599-1679 -- the orientation halfway (in chain index) between images `lo` and `hi`, then the two halves
	void place_orientations(unsigned lo, unsigned hi, unsigned level, unsigned n_images, double b2, double mu_kT) {
		const double pi_ = 3.141592653589793238462643383279502884L, h = 6.626068e-34;
		const double two_pi = 2.0 * pi_;
		if (level > n_images) return;
		const unsigned mid = (lo + hi) / 2;
		const Vec3 a = orientations[lo], c = orientations[(hi == n_images) ? 0 : hi];
		const Vec3 mean_dir = unit3(divided3(add3(a, c), 2.0));
		Vec3 side; // a vector orthogonal to mean_dir
		double opening = 0; // angle between a and c
		if (level > 2) {
			side = sub3(c, a);
			opening = angle3(a, c);
		} else { // a == c: any direction that differs from mean_dir serves to build an orthogonal one
			const Vec3 other = unit3(add3(Vec3{{1, 2, -3}}, mean_dir));
			side = cross3(other, mean_dir);
		}
		const double u = rng.rand();
		const double lambda2 = h * h / (two_pi * mu_kT);
		const double kh = pi_ * b2 / lambda2;                       // Eq. (13b) of the paper
		const double K = 4.0 * kh * level * std::cos(opening * 0.5); // Eq. (17b)
		const double tilt = std::acos(1.0 + (1.0 / K) * std::log(1.0 - u * (1.0 - std::exp(-2.0 * K)))); // Eq. (18)
		const double azimuth = rng.rand() * two_pi;
		const Vec3 tilt_axis = Rotor::about_axis(mean_dir[0], mean_dir[1], mean_dir[2], azimuth).turn_left_first(side);
		orientations[mid] = Rotor::about_axis(tilt_axis[0], tilt_axis[1], tilt_axis[2], tilt).turn_left_first(mean_dir);
		if (level < n_images) {
			place_orientations(lo, mid, level * 2, n_images, b2, mu_kT);
			place_orientations(mid, hi, level * 2, n_images, b2, mu_kT);
		}
	}
	// Molecule::orient, src/Molecule.cpp:211-254: turn image s of the altered molecule about its COM so that atom `handle` points along `dir`
	void orient(int s, const Vec3 &dir, int handle) {
		update_COM(s);
		const Vec3 centre = com[s];
		translate(s, -centre[0], -centre[1], -centre[2]);
		const int first = mol_first[target], last = mol_first[target + 1];
		const double *hp = systems[s]->atoms[first + handle].pos;
		const Vec3 now = unit3(Vec3{{hp[0], hp[1], hp[2]}});
		const double turn = std::acos(dot3(now, dir) / length3(dir));
		const Vec3 axis = cross3(now, dir);
		const Rotor spin = Rotor::about_axis(axis[0], axis[1], axis[2], turn);
		for (int i = first; i < last; i++) {
			double *p = systems[s]->atoms[i].pos;
			const Vec3 r = spin.turn_left_first(Vec3{{p[0], p[1], p[2]}});
			p[0] = r[0];
			p[1] = r[1];
			p[2] = r[2];
		}
		translate(s, centre[0], centre[1], centre[2]);
	}

	// SimulationControl::PI_chain_mass_length2() for the altered molecule, :905-965
	double PI_chain_mass_length2() {
		const double AMU2KG = 1.66053873e-27, ANGSTROM2METER = 1.0e-10;
		for (int s = 0; s < nSys; s++) update_COM(s);
		double len2 = 0;
		for (int i = 0; i < nSys; i++) {
			const int j = (i + 1) % nSys;
			const double dx = com[i][0] - com[j][0], dy = com[i][1] - com[j][1], dz = com[i][2] - com[j][2];
			len2 += dx * dx + dy * dy + dz * dz;
		}
		len2 *= (mol_mass[0] * AMU2KG) * (ANGSTROM2METER * ANGSTROM2METER);
		return len2;
	}

	// SimulationControl::PI_NVT_boltzmann_factor, :490-547
	double PI_NVT_boltzmann_factor(int mv, double delta_energy, double delta_chain, double delta_orient = 0) const {
		const double pi_ = 3.141592653589793238462643383279502884L, h = 6.626068e-34, kB = 1.3806503e-23; // src/constants.h:13-20
		const double T = cfg.temperature;
		if (mv == MOVETYPE_PERTURB_BEADS) {
			const size_t P = (size_t)nSys;
			const double PIchain_2_K = (P * pi_ * pi_ * kB * T) / (2.0 * h * h);
			const double potential_contrib = delta_energy / T;
			const double PI_COM_contrib = delta_chain * PIchain_2_K;
			// The orientational chain measure enters WITHOUT the reduced mass (:515-520 reads it and does not use it), i.e. ~1e27 per
			// A^2: from images that start with one common orientation every orientational trial has delta_orient > 0 and is rejected,
			// and a trial that shortened the chain would be accepted whatever its energy.  Part of the reference's trajectories.
			const double PI_orientation_contrib = (sorbate_of_target() >= 0) ? delta_orient * PIchain_2_K : 0;
			return std::exp(-potential_contrib - PI_COM_contrib - PI_orientation_contrib);
		}
		return std::exp(-delta_energy / T);
	}

	double PI_calculate_potential() {
		energy_calls += nSys;
		return pi.PI_calculate_potential();
	}
	// PI_calculate_energy :734-749; with_potential = false reuses the per-image energies of the evaluation just made for the
	// same geometry (the reference evaluates them a second time, to the same values)
	double PI_calculate_energy(bool with_potential) {
		observables_t &o = pi.sys_observables;
		const double kinetic = pi.PI_calculate_kinetic();
		const double potential = with_potential ? PI_calculate_potential() : ((o.rd_energy + o.coulombic_energy) + o.vdw_energy) + o.polarization_energy;
		o.energy = kinetic + potential;
		return o.energy;
	}

	void write_observables(FILE *fp) const { // System::write_observables, src/System.Output.cpp:251-271
		const observables_t &o = pi.sys_observables;
		std::fprintf(fp, "%d %f %f %f %f %f %f %f %f %f %f %f\n", (int)step, o.energy, o.coulombic_energy, o.rd_energy, o.polarization_energy,
		             o.vdw_energy, o.kinetic_energy, o.temperature, o.N, o.spin_ratio, o.volume, cfg.temperature);
		std::fflush(fp);
	}

	int current_target() const { return target; }
	int current_movetype() const { return movetype; }

private:
	static constexpr double kMaxValue = 1.0e40; // src/constants.h:56
	int nSys = 0, target = 0, movetype = MOVETYPE_DISPLACE, chain_start = 0;
	std::vector<int> mol_first;
	std::vector<std::array<double, 3>> com; // Molecule::com of the altered molecule in every image
	std::vector<double> mol_mass;           // Molecule::mass
	std::vector<Vec3> orientations;         // one unit vector per image (SimulationControl::orientations)
	std::vector<std::vector<double>> backup_pos;
	std::vector<observables_t> checkpoint_obs;
	observables_t checkpoint_sys_obs;

	void update_COM(int s) { // Molecule::update_COM, src/Molecule.cpp:259-281
		const int first = mol_first[target], last = mol_first[target + 1];
		double m = 0, c0 = 0, c1 = 0, c2 = 0;
		for (int i = first; i < last; i++) {
			const Atom &a = systems[s]->atoms[i];
			m += a.mass;
			c0 += a.mass * a.pos[0];
			c1 += a.mass * a.pos[1];
			c2 += a.mass * a.pos[2];
		}
		mol_mass[s] = m;
		com[s][0] = c0 / m;
		com[s][1] = c1 / m;
		com[s][2] = c2 / m;
	}
	void translate(int s, double x, double y, double z) { // Molecule::translate, :333-345
		com[s][0] += x;
		com[s][1] += y;
		com[s][2] += z;
		const int first = mol_first[target], last = mol_first[target + 1];
		for (int i = first; i < last; i++) {
			double *p = systems[s]->atoms[i].pos;
			p[0] += x;
			p[1] += y;
			p[2] += z;
		}
	}
	void translate_rand_pbc(int s, double scale, double cutoff, const double u[6]) { // :296-321
		double trans_x = scale * u[0] * cutoff;
		double trans_y = scale * u[1] * cutoff;
		double trans_z = scale * u[2] * cutoff;
		if (u[3] < 0.5) trans_x *= -1.0;
		if (u[4] < 0.5) trans_y *= -1.0;
		if (u[5] < 0.5) trans_z *= -1.0;
		const int first = mol_first[target], last = mol_first[target + 1];
		for (int i = first; i < last; i++) {
			double *p = systems[s]->atoms[i].pos;
			p[0] += trans_x;
			p[1] += trans_y;
			p[2] += trans_z;
		}
		update_COM(s);
	}

	void backup_observables_ALL_SYSTEMS() { // src/SimulationControl.cpp:2838-2848
		checkpoint_sys_obs = pi.sys_observables;
		for (int s = 0; s < nSys; s++) checkpoint_obs[s] = *systems[s]->observables;
	}
	void restore_PI_systems() { // :201-206 + System::restore (src/System.MonteCarlo.cpp:1510-1575)
		const int first = mol_first[target], count = mol_first[target + 1] - first;
		for (int s = 0; s < nSys; s++) {
			systems[s]->iterator_failed = 0;
			*systems[s]->observables = checkpoint_obs[s];
			for (int k = 0; k < count; k++)
				for (int d = 0; d < 3; d++) systems[s]->atoms[first + k].pos[d] = backup_pos[s][3 * k + d];
			systems[s]->move_atoms(first, count);
		}
	}
	void register_move(int mv, bool ok) {
		(ok ? accept : reject)++;
		if (mv == MOVETYPE_DISPLACE) (ok ? accept_displace : reject_displace)++;
		else (ok ? accept_bead : reject_bead)++;
	}
	void refresh_aggregate() { // average_current_observables_into_PI_avgObservables :211-232 (the instantaneous part)
		observables_t &o = pi.sys_observables;
		o.N = systems[0]->observables->N;
		o.volume = systems[0]->observables->volume;
		o.temperature = systems[0]->observables->temperature;
		o.spin_ratio = systems[0]->observables->spin_ratio;
		o.NU = systems[0]->observables->NU;
	}
};

} // namespace mpmc
