// include/mpmc_gibbs.hpp -- the two-box part of the host facade: what SimulationControl::Gibbs_mc asks of the energy path.
//
// Reference (src/SimulationControl.Gibbs.cpp):  :152  initial_energy[i] = systems[i]->mc_initial_energy();
//                                              :179  final_energy[0] = systems[0]->energy();
//                                              :180  final_energy[1] = systems[1]->energy();
//                                              :190  boltzmann_factor_NVT_Gibbs(*systems[0], ..., *systems[1], ...);   (:358-522)
// Here the two boxes are two device contexts; `place_on_devices()` puts box 0 on device 0 and box 1 on device 1 when the node
// shows two (north_star: "Gibbs dual-box energies shard naturally across the GPUs"), `energy()` enqueues both evaluations before
// it waits for either.  The acceptance factor is the library's restatement of boltzmann_factor_NVT_Gibbs, pinned by the reference's
// own function (tests/golden/gibbs_bf.json).  GibbsNVT below restates the loop of Gibbs_mc with its move generators (make_move_Gibbs,
// volume_change_Gibbs) and reproduces trajectories made by the reference's own functions (oracle/ref_gibbs_traj.cpp ->
// tests/golden/gibbs_*; tests/test_gibbs_driver.py); what the moves DO to a live context -- set_box, set_atoms with a new N,
// update_positions, restore -- is also exercised directly by tests/test_gpu_box_moves.py.
#pragma once
#include <array>
#include <cmath>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "mpmc_io.hpp"
#include "mpmc_rotation.hpp"
#include "mpmc_pimc.hpp" // Rando: the reference's ONE global engine (src/Rando.h)
#include "mpmc_system.hpp"

namespace mpmc {

template <class SystemT>
class GibbsBoxesT {
public:
	SystemT *systems[2] = {nullptr, nullptr};
	double initial_energy[2] = {0, 0}, final_energy[2] = {0, 0};
	double boltzmann_factor[2] = {0, 0}; // sys[i]->nodestats->boltzmann_factor

	GibbsBoxesT(SystemT &a, SystemT &b) {
		systems[0] = &a;
		systems[1] = &b;
	}
	// box i -> device i mod G (call before the first evaluation: a System creates its context lazily on `device`)
	void place_on_devices() {
		int n = 0;
		if (mpmc_device_count(&n) != MPMC_OK || n < 1) throw (int)MPMC_ERR_NO_DEVICE;
		systems[0]->device = 0;
		systems[1]->device = 1 % n;
	}
	// Gibbs.cpp:179-180 -- both boxes enqueued, then both waited for
	void energy() {
		systems[0]->energy_async();
		systems[1]->energy_async();
		final_energy[0] = systems[0]->energy_wait();
		final_energy[1] = systems[1]->energy_wait();
	}
	// Gibbs.cpp:152
	void mc_initial_energy() {
		energy();
		initial_energy[0] = final_energy[0];
		initial_energy[1] = final_energy[1];
	}
	// boltzmann_factor_NVT_Gibbs (:358-522).  movetype: MPMC_MOVETYPE_* of the two boxes; N / volume: observables AFTER the move;
	// checkpoint_volume_0: the volume box 0 started the move from.  A bad contact on a coordinated move sets both factors to 0 and
	// observables->energy to MAXVALUE, as the reference does.  Throws the reference's ints (20000, 102).
	void boltzmann_factor_NVT_Gibbs(int movetype_a, int movetype_b, double temperature, double checkpoint_volume_0) {
		mpmc_gibbs_move m{};
		m.movetype[0] = movetype_a;
		m.movetype[1] = movetype_b;
		m.temperature = temperature;
		for (int i = 0; i < 2; i++) {
			m.init_energy[i] = initial_energy[i];
			m.final_energy[i] = final_energy[i];
			m.N[i] = (double)systems[i]->countN();
			m.volume[i] = systems[i]->pbc.volume;
		}
		m.checkpoint_volume_0 = checkpoint_volume_0;
		double en[2] = {systems[0]->observables->energy, systems[1]->observables->energy};
		const int rc = mpmc_gibbs_boltzmann_factor(&m, boltzmann_factor, en);
		if (rc != MPMC_OK) throw rc;
		systems[0]->observables->energy = en[0];
		systems[1]->observables->energy = en[1];
	}
	// accepted: the trial energies become the accepted ones (Gibbs.cpp:172-173 restores them at the top of the next step)
	void accept() {
		initial_energy[0] = final_energy[0];
		initial_energy[1] = final_energy[1];
	}
};
using GibbsBoxes = GibbsBoxesT<System>;

// ---------------------------------------------------------------------------------------------------------------------------------
// The Gibbs-ensemble Monte Carlo driver: the steps either side of the two energy() calls, written against the
// facade so that the GPU box runs `ensemble nvt_gibbs` inputs end to end.  Behaviour follows the reference function by function --
// the same random-number streams in the same order (every System owns a std::mt19937 that the Gibbs set-up never seeds, so both boxes
// draw the SAME default-seeded sequence; acceptance and rotations draw from the global Rando), the same floating-point association --
// so a run reproduces, step for step, the trajectory the reference's own functions make (tests/golden/gibbs_*: oracle/ref_gibbs_traj.cpp
// drives System::pick_Gibbs_move / make_move_Gibbs / energy / boltzmann_factor_NVT_Gibbs / restore; the stock Gibbs_mc loop itself
// dies in bookkeeping that has nothing to do with the trajectory).  Reference functions mirrored:
//   SimulationControl::Gibbs_mc            src/SimulationControl.Gibbs.cpp:133-330   loop skeleton, Metropolis tests, accept / restore
//   System::pick_Gibbs_move                src/System.MonteCarlo.cpp:509-720
//   System::make_move_Gibbs                :905-1120   displacement of one molecule per box, coupled volume change, particle transfer
//   System::volume_change_Gibbs            :1287-1340  /  revert_volume_change :1690-1727
//   System::displace                       :1226-1230  = Molecule::translate_rand_pbc (src/Molecule.cpp:286-321) + rotate_rand (:128-137)
//   Molecule::rotate                       src/Molecule.cpp:138-203 (quaternion q (p q*), about the stored centre of mass)
//   System::restore                        :1510-1580
// Scope: no quantum rotation (spin flips), no adiabatic / cavity-biased / insertion-list moves, no simulated annealing or tempering.
struct GibbsSettings {
	unsigned int numsteps = 0, seed = 0;
	bool seed_set = false;
	double temperature = 0, move_factor = 1.0, rot_factor = 1.0; // src/System.h:526-538
	double volume_probability = 0, transfer_probability = 0, volume_change_factor = 0.25;
	std::string pqr_input, pqr_input_B;
};

// the Monte Carlo keywords of a reference nvt_gibbs input file (src/SimulationControl.cpp:204-267, :801-863, :1351)
inline GibbsSettings read_gibbs_settings(const std::string &path) {
	using namespace io_detail;
	std::ifstream f(path);
	if (!f) throw 1000; // fopen_fail_read
	GibbsSettings c;
	std::string line;
	while (std::getline(f, line)) {
		const std::vector<std::string> t = tokens(line);
		if (t.size() < 2 || t[0][0] == '!' || t[0][0] == '#') continue;
		const std::string k = lower(t[0]);
		double v = 0;
		if (k == "ensemble") {
			if (lower(t[1]) != "nvt_gibbs") throw 4004; // unsupported_setting: this driver is the nvt_gibbs ensemble
		} else if (k == "pqr_input") c.pqr_input = t[1];
		else if (k == "pqr_input_b") c.pqr_input_B = t[1];
		else if (k == "spinflip_probability" || k == "simulated_annealing" || k == "parallel_tempering" || k == "cavity_bias") {
			if (to_double(t[1], v) ? (v != 0.0) : (onoff(t[1]) != 0)) throw 4004;
		} else if (k == "numsteps" || k == "seed" || k == "preset_seed" || k == "move_factor" || k == "rot_factor" || k == "temperature" ||
		           k == "volume_probability" || k == "transfer_probability" || k == "volume_change_factor") {
			if (!to_double(t[1], v)) throw 3000; // invalid_input
			if (k == "numsteps") c.numsteps = (unsigned int)v;
			else if (k == "seed" || k == "preset_seed") { c.seed = (unsigned int)v; c.seed_set = true; }
			else if (k == "move_factor") c.move_factor = v;
			else if (k == "rot_factor") c.rot_factor = v;
			else if (k == "temperature") c.temperature = v;
			else if (k == "volume_probability") c.volume_probability = v;
			else if (k == "transfer_probability") c.transfer_probability = v;
			else c.volume_change_factor = v;
		}
	}
	if (c.transfer_probability == 0.0) throw 4003; // missing_setting (SimulationControl.Gibbs.cpp:112-115)
	return c;
}

template <class SystemT>
class GibbsNVT {
public:
	GibbsSettings cfg;
	SystemT *systems[2] = {nullptr, nullptr};
	Rando rng;                  // Rando:: (global engine of the reference): acceptance tests and rotations
	std::mt19937 mt_rand[2];    // System::mt_rand of each box (never seeded by the Gibbs set-up: default seed, identical streams)
	unsigned int step = 0;
	long accept[2] = {0, 0}, reject[2] = {0, 0};
	long energy_calls = 0;

	struct Record { // what one step of the loop decided (tests compare it with the reference-made trajectory)
		int movetype[2];
		double final_energy[2], boltzmann_factor[2];
		int accepted[2];
		double energy[2], N[2], volume[2];
		int natoms[2];
	};
	std::vector<Record> trace;
	double initial_energy[2] = {0, 0};

	GibbsNVT(SystemT &a, SystemT &b) {
		systems[0] = &a;
		systems[1] = &b;
	}

	// SimulationControl::initialize_Gibbs_systems (volume probability default :93-98) + the head of Gibbs_mc (:141-158)
	void init() {
		rng.seed(cfg.seed);
		for (int i = 0; i < 2; i++) {
			systems[i]->temperature = cfg.temperature;
			systems[i]->observables->temperature = cfg.temperature;
			rebuild_molecules(i);
			systems[i]->countN();
		}
		if (cfg.volume_probability == 0.0) cfg.volume_probability = 1.0 / (double)(systems[0]->observables->N + systems[1]->observables->N);
		evaluate_both(initial_energy);
		for (int i = 0; i < 2; i++) {
			systems[i]->observables->volume = systems[i]->pbc.volume;
			if (!std::isfinite(initial_energy[i])) initial_energy[i] = systems[i]->observables->energy = kGibbsMaxValue; // mc_initial_energy :161-175
			ckpt_obs[i] = *systems[i]->observables;
		}
		move = pick_Gibbs_move();
	}

	// one pass of the main loop of Gibbs_mc (:166-272)
	void mc_step() {
		step++;
		double init[2], fin[2];
		init[0] = systems[0]->observables->energy;
		init[1] = systems[1]->observables->energy;
		Record r{};
		r.movetype[0] = movetype[0];
		r.movetype[1] = movetype[1];
		make_move_Gibbs();
		evaluate_both(fin);
		double bf[2] = {stored_bf[0], stored_bf[1]}, en[2] = {systems[0]->observables->energy, systems[1]->observables->energy};
		mpmc_gibbs_move m{};
		for (int i = 0; i < 2; i++) {
			m.movetype[i] = movetype[i];
			m.init_energy[i] = init[i];
			m.final_energy[i] = fin[i];
			m.N[i] = systems[i]->observables->N;
			m.volume[i] = systems[i]->observables->volume;
		}
		m.temperature = systems[0]->temperature;
		m.checkpoint_volume_0 = ckpt_obs[0].volume;
		const int rc = mpmc_gibbs_boltzmann_factor(&m, bf, en);
		if (rc != MPMC_OK) throw rc;
		for (int i = 0; i < 2; i++) {
			stored_bf[i] = bf[i];
			systems[i]->observables->energy = en[i];
			r.final_energy[i] = fin[i];
			r.boltzmann_factor[i] = bf[i];
		}
		if (move == MPMC_MOVETYPE_DISPLACE || move == MPMC_MOVETYPE_SPINFLIP) { // independent Metropolis tests (:196-226)
			for (int i = 0; i < 2; i++) {
				if ((rng.rand() < stored_bf[i]) && !systems[i]->iterator_failed) {
					r.accepted[i] = 1;
					accept[i]++;
				} else {
					systems[i]->iterator_failed = 0;
					restore(i);
					reject[i]++;
				}
			}
		} else { // transfers and volume exchanges stand or fall together (:228-270)
			if ((rng.rand() < stored_bf[0]) && !systems[0]->iterator_failed && !systems[1]->iterator_failed) {
				for (int i = 0; i < 2; i++) {
					r.accepted[i] = 1;
					ckpt_obs[i] = *systems[i]->observables;
					accept[i]++;
				}
			} else {
				for (int i = 0; i < 2; i++) {
					systems[i]->iterator_failed = 0;
					restore(i);
					reject[i]++;
				}
			}
		}
		for (int i = 0; i < 2; i++) {
			r.energy[i] = systems[i]->observables->energy;
			r.N[i] = systems[i]->observables->N;
			r.volume[i] = systems[i]->pbc.volume;
			r.natoms[i] = (int)systems[i]->atoms.size();
		}
		trace.push_back(r);
		for (int i = 0; i < 2; i++) ckpt_obs[i] = *systems[i]->observables; // backup_observables_SYS_VECTOR (:274)
		move = pick_Gibbs_move();
	}
	void run() {
		for (unsigned int s = 0; s < cfg.numsteps; s++) mc_step();
	}

private:
	static constexpr double kGibbsMaxValue = 1.0e40; // MAXVALUE, src/constants.h:53
	int move = 0, movetype[2] = {0, 0};
	double stored_bf[2] = {0, 0};                  // nodestats->boltzmann_factor (entries the acceptance rule leaves alone keep their value)
	observables_t ckpt_obs[2];                     // checkpoint->observables
	std::vector<int> mol_first[2];                 // first atom of every molecule (+ end), list order
	std::vector<std::array<double, 3>> com[2];     // Molecule::com as the reference holds it (updated where the reference updates it)
	int altered[2] = {0, 0};                       // checkpoint->molecule_altered (index into the molecule list)
	std::vector<Atom> backup[2];                   // checkpoint->molecule_backup
	std::array<double, 3> backup_com[2];
	std::uniform_real_distribution<double> dist{0, 1}; // System::dist

	double get_rand(int i) { return dist(mt_rand[i]); } // System::get_rand, src/System.cpp:1478-1482

	void rebuild_molecules(int i) { // enumerate_particles: molecule ids in list order; first atoms
		std::vector<Atom> &a = systems[i]->atoms;
		mol_first[i].clear();
		int id = -1, prev = 0;
		for (size_t k = 0; k < a.size(); k++) {
			if (k == 0 || a[k].molecule != prev) {
				id++;
				mol_first[i].push_back((int)k);
			}
			prev = a[k].molecule;
			a[k].molecule = -1 - id; // (tagged negative first: an old id may equal a new one of another molecule)
		}
		for (Atom &x : a) x.molecule = -1 - x.molecule;
		mol_first[i].push_back((int)a.size());
		systems[i]->natoms = (int)a.size();
	}
	int n_molecules(int i) const { return (int)mol_first[i].size() - 1; }
	bool mol_frozen(int i, int m) const { return systems[i]->atoms[mol_first[i][m + 1] - 1].frozen != 0; } // last row decides (src/System.cpp:684)

	void update_COM(int i, int m, std::array<double, 3> &c) const { // Molecule::update_COM, src/Molecule.cpp:259-281
		double mass = 0;
		c = {{0, 0, 0}};
		for (int k = mol_first[i][m]; k < mol_first[i][m + 1]; k++) {
			const Atom &a = systems[i]->atoms[k];
			mass += a.mass;
			c[0] += a.mass * a.pos[0];
			c[1] += a.mass * a.pos[1];
			c[2] += a.mass * a.pos[2];
		}
		c[0] /= mass;
		c[1] /= mass;
		c[2] /= mass;
	}
	// systems[0]->energy(), systems[1]->energy() (Gibbs.cpp:179-180): both boxes -- usually on two devices -- are enqueued before either is
	// waited for, results taken in box order.  The reference's pairs() refreshes every Molecule::com on the way (update_com, src/System.cpp:1347-1378).
	void evaluate_both(double e[2]) {
		energy_calls += 2;
		// a throw between the first enqueue and the last wait must not leave a box with an evaluation in flight: the next move would
		// change that box's cell or atoms under it.  Whatever was enqueued and not yet waited for is drained before the throw travels on.
		bool in_flight[2] = {false, false};
		try {
			for (int i = 0; i < 2; i++) {
				systems[i]->energy_async();
				in_flight[i] = true;
			}
			for (int i = 0; i < 2; i++) {
				in_flight[i] = false; // (energy_wait closes the evaluation whether it returns or throws)
				e[i] = systems[i]->energy_wait();
				com[i].resize(n_molecules(i));
				for (int m = 0; m < n_molecules(i); m++) update_COM(i, m, com[i][m]);
			}
		} catch (...) {
			for (int i = 0; i < 2; i++)
				if (in_flight[i]) {
					try {
						(void)systems[i]->energy_wait();
					} catch (...) {
					}
				}
			throw;
		}
	}

	// Molecule::rotate_rand + Molecule::rotate on `atoms` about `c` (src/Molecule.cpp:128-203)
	void rotate_rand(std::vector<Atom> &atoms, int first, int last, const std::array<double, 3> &c, double scale) {
		const double x = rng.rand_normal();
		const double y = rng.rand_normal();
		const double z = rng.rand_normal();
		const double angle = rng.rand() * 360 * scale;
		const Rotor spin = Rotor::about_axis_degrees(x, y, z, angle);
		for (int k = first; k < last; k++) {
			Atom &a = atoms[k];
			a.pos[0] -= c[0];
			a.pos[1] -= c[1];
			a.pos[2] -= c[2];
		}
		for (int k = first; k < last; k++) {
			Atom &a = atoms[k];
			const Vec3 r = spin.turn_right_first(Vec3{{a.pos[0], a.pos[1], a.pos[2]}});
			a.pos[0] = r[0];
			a.pos[1] = r[1];
			a.pos[2] = r[2];
			a.pos[0] += c[0];
			a.pos[1] += c[1];
			a.pos[2] += c[2];
		}
	}

	// System::pick_Gibbs_move, src/System.MonteCarlo.cpp:509-720
	int pick_Gibbs_move() {
		int n_exchangeable[2] = {0, 0};
		std::vector<int> exchange[2];
		for (int i = 0; i < 2; i++)
			for (int m = 0; m < n_molecules(i); m++)
				if (!mol_frozen(i, m)) {
					exchange[i].push_back(m);
					++n_exchangeable[i];
				}
		{
			const double p_volume = cfg.volume_probability + 0.0; // (+ spinflip probability: quantum rotation is off)
			const double p_transfer = cfg.transfer_probability + p_volume;
			const double pick = get_rand(0);
			if (pick < p_volume) {
				movetype[0] = movetype[1] = MPMC_MOVETYPE_VOLUME;
			} else if (pick < p_transfer) {
				if (get_rand(0) < 0.5) {
					movetype[0] = MPMC_MOVETYPE_REMOVE;
					movetype[1] = MPMC_MOVETYPE_INSERT;
				} else {
					movetype[0] = MPMC_MOVETYPE_INSERT;
					movetype[1] = MPMC_MOVETYPE_REMOVE;
				}
			} else {
				movetype[0] = movetype[1] = MPMC_MOVETYPE_DISPLACE;
			}
		}
		for (int i = 0; i < 2; i++) {
			--n_exchangeable[i];
			const int pick = (int)std::floor(get_rand(i) * systems[i]->observables->N);
			if (pick < 0 || pick >= (int)exchange[i].size()) throw 3001; // no_molecules_in_system
			altered[i] = exchange[i][pick];
			// the box must keep one molecule: a removal of the last one becomes a displacement (checked inside the loop, as the reference does)
			if ((!n_exchangeable[0] && movetype[0] == MPMC_MOVETYPE_REMOVE) || (!n_exchangeable[1] && movetype[1] == MPMC_MOVETYPE_REMOVE))
				movetype[0] = movetype[1] = MPMC_MOVETYPE_DISPLACE;
		}
		for (int i = 0; i < 2; i++) { // checkpoint->molecule_backup = new Molecule(*molecule_altered)
			backup[i].assign(systems[i]->atoms.begin() + mol_first[i][altered[i]], systems[i]->atoms.begin() + mol_first[i][altered[i] + 1]);
			backup_com[i] = com[i][altered[i]];
		}
		return movetype[0];
	}

	// System::make_move_Gibbs, :905-1120
	void make_move_Gibbs() {
		switch (movetype[0]) {
		case MPMC_MOVETYPE_DISPLACE:
			for (int i = 0; i < 2; i++) { // System::displace = translate_rand_pbc (6 draws of the box's own engine) + rotate_rand (global engine)
				const int m = altered[i], first = mol_first[i][m], last = mol_first[i][m + 1];
				double dice[6];
				for (int k = 0; k < 6; k++) dice[k] = dist(mt_rand[i]);
				double tx = cfg.move_factor * dice[0] * systems[i]->pbc.cutoff;
				double ty = cfg.move_factor * dice[1] * systems[i]->pbc.cutoff;
				double tz = cfg.move_factor * dice[2] * systems[i]->pbc.cutoff;
				if (dice[3] < 0.5) tx *= -1.0;
				if (dice[4] < 0.5) ty *= -1.0;
				if (dice[5] < 0.5) tz *= -1.0;
				for (int k = first; k < last; k++) {
					Atom &a = systems[i]->atoms[k];
					a.pos[0] += tx;
					a.pos[1] += ty;
					a.pos[2] += tz;
				}
				update_COM(i, m, com[i][m]);
				rotate_rand(systems[i]->atoms, first, last, com[i][m], cfg.rot_factor);
				systems[i]->move_atoms(first, last - first);
			}
			break;
		case MPMC_MOVETYPE_VOLUME:
			volume_change_Gibbs();
			break;
		case MPMC_MOVETYPE_INSERT:
		case MPMC_MOVETYPE_REMOVE:
			for (int s = 0; s < 2; s++) {
				if (movetype[s] != MPMC_MOVETYPE_INSERT) continue;
				// box s: a copy of ITS picked molecule goes to a random place of the cell in a random orientation (:1009-1035)
				double rnd[3], c[3];
				for (int p = 0; p < 3; p++) rnd[p] = 0.5 - get_rand(s);
				for (int p = 0; p < 3; p++) {
					c[p] = 0;
					for (int q = 0; q < 3; q++) c[p] += systems[s]->pbc.basis[q][p] * rnd[q];
				}
				std::vector<Atom> mol = backup[s];
				for (Atom &a : mol)
					for (int p = 0; p < 3; p++) a.pos[p] += c[p] - backup_com[s][p];
				const std::array<double, 3> cc = {{c[0], c[1], c[2]}};
				rotate_rand(mol, 0, (int)mol.size(), cc, 1.0);
				// into the list in front of the picked molecule, which it then replaces as "altered" (:1052-1066)
				const int at = mol_first[s][altered[s]];
				for (Atom &a : mol) a.molecule = -1;
				systems[s]->atoms.insert(systems[s]->atoms.begin() + at, mol.begin(), mol.end());
				com[s].insert(com[s].begin() + altered[s], cc);
				backup[s].clear(); // checkpoint->molecule_backup = nullptr
				rebuild_molecules(s);
				systems[s]->atoms_changed();
				// the OTHER box loses its picked molecule (:1098-1112)
				const int o = 1 - s;
				systems[o]->atoms.erase(systems[o]->atoms.begin() + mol_first[o][altered[o]], systems[o]->atoms.begin() + mol_first[o][altered[o] + 1]);
				com[o].erase(com[o].begin() + altered[o]);
				rebuild_molecules(o);
				systems[o]->atoms_changed();
			}
			break;
		default:
			throw (int)MPMC_ERR_INVALID_MC_MOVE_KIND;
		}
	}

	// System::volume_change_Gibbs, :1287-1340
	void volume_change_Gibbs() {
		double new_volume[2];
		do {
			const double log_new_volume = std::log(systems[0]->pbc.volume) + (get_rand(0) - 0.5) * cfg.volume_change_factor;
			new_volume[0] = std::exp(log_new_volume);
			new_volume[1] = systems[1]->pbc.volume + systems[0]->pbc.volume - new_volume[0];
		} while (new_volume[1] <= 0.0);
		for (int s = 0; s < 2; s++) scale_box(s, std::pow(new_volume[s] / systems[s]->pbc.volume, 1.0 / 3.0), false);
	}
	// the common part of volume_change_Gibbs and revert_volume_change (:1690-1727; the revert also moves the stored centres of mass)
	void scale_box(int s, double basis_scale_factor, bool reverting) {
		SystemT &S = *systems[s];
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) S.pbc.basis[i][j] *= basis_scale_factor;
		S.update_pbc();
		S.observables->volume = S.pbc.volume;
		for (int m = 0; m < n_molecules(s); m++) {
			double delta_pos[3];
			for (int i = 0; i < 3; i++) {
				const double old_com = com[s][m][i];
				const double new_com = com[s][m][i] * basis_scale_factor;
				if (reverting) com[s][m][i] = new_com;
				delta_pos[i] = new_com - old_com;
			}
			for (int k = mol_first[s][m]; k < mol_first[s][m + 1]; k++)
				for (int i = 0; i < 3; i++) S.atoms[k].pos[i] += delta_pos[i];
		}
		S.move_atoms(0, (int)S.atoms.size());
	}

	// System::restore, :1510-1580
	void restore(int i) {
		*systems[i]->observables = ckpt_obs[i];
		switch (movetype[i]) {
		case MPMC_MOVETYPE_INSERT: // take the inserted molecule out again
			systems[i]->atoms.erase(systems[i]->atoms.begin() + mol_first[i][altered[i]], systems[i]->atoms.begin() + mol_first[i][altered[i] + 1]);
			com[i].erase(com[i].begin() + altered[i]);
			rebuild_molecules(i);
			systems[i]->atoms_changed();
			break;
		case MPMC_MOVETYPE_REMOVE: { // put the backup back where the molecule was
			const int at = (altered[i] < n_molecules(i)) ? mol_first[i][altered[i]] : (int)systems[i]->atoms.size();
			for (Atom &a : backup[i]) a.molecule = -1;
			systems[i]->atoms.insert(systems[i]->atoms.begin() + at, backup[i].begin(), backup[i].end());
			com[i].insert(com[i].begin() + altered[i], backup_com[i]);
			rebuild_molecules(i);
			systems[i]->atoms_changed();
			break;
		}
		case MPMC_MOVETYPE_VOLUME:
			scale_box(i, std::pow(ckpt_obs[i].volume / systems[i]->pbc.volume, 1.0 / 3.0), true);
			break;
		default: { // displacement: the backup replaces the rejected configuration
			const int first = mol_first[i][altered[i]];
			for (size_t k = 0; k < backup[i].size(); k++) {
				const int id = systems[i]->atoms[first + k].molecule;
				systems[i]->atoms[first + k] = backup[i][k];
				systems[i]->atoms[first + k].molecule = id;
			}
			com[i][altered[i]] = backup_com[i];
			systems[i]->move_atoms(first, (int)backup[i].size());
		}
		}
	}
};

} // namespace mpmc
