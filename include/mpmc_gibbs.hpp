// include/mpmc_gibbs.hpp -- the two-box part of the host facade: what SimulationControl::Gibbs_mc asks of the energy path.
//
// Reference (src/SimulationControl.Gibbs.cpp):  :152  initial_energy[i] = systems[i]->mc_initial_energy();
//                                              :179  final_energy[0] = systems[0]->energy();
//                                              :180  final_energy[1] = systems[1]->energy();
//                                              :190  boltzmann_factor_NVT_Gibbs(*systems[0], ..., *systems[1], ...);   (:358-522)
// Here the two boxes are two device contexts; `place_on_devices()` puts box 0 on device 0 and box 1 on device 1 when the node
// shows two (north_star: "Gibbs dual-box energies shard naturally across the GPUs"), `energy()` enqueues both evaluations before
// it waits for either.  The acceptance factor is the library's restatement of boltzmann_factor_NVT_Gibbs, pinned by the reference's
// own function (tests/golden/gibbs_bf.json).  The move generators of Gibbs_mc (make_move_Gibbs, volume_change_Gibbs) are outside the
// path (MC drivers) and the reference's own loop cannot run in this image, so no trajectory exists to reproduce; what the
// moves DO to a live context -- set_box, set_atoms with a new N, update_positions, restore -- is exercised by tests/test_gpu_box_moves.py.
#pragma once
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

} // namespace mpmc
