// gibbs.cpp -- the two-box side of the energy path: SimulationControl::Gibbs_mc evaluates both boxes after every move
// (reference src/SimulationControl.Gibbs.cpp:179-180) and turns the two energy differences into ONE acceptance factor
// (boltzmann_factor_NVT_Gibbs, :358-522).  (part of libmpmc_energy.so; there is no CPU fallback for the evaluations.)
#include "context.h"

using namespace mpmc;

// final_energy[0] = systems[0]->energy(); final_energy[1] = systems[1]->energy();   -- both enqueued before either is waited for.
// The boxes are independent evaluations: put them on two devices (box 0 -> device 0, box 1 -> device 1) and they run side by side;
// on one device their kernels share it through the contexts' own streams.
extern "C" int mpmc_gibbs_energy(mpmc_ctx *box_a, mpmc_ctx *box_b, mpmc_result *out_a, mpmc_result *out_b) {
	if (!box_a || !box_b || !out_a || !out_b || box_a == box_b) return MPMC_ERR_ARG;
	int rc = mpmc_energy_async(box_a);
	if (rc != MPMC_OK) return rc;
	rc = mpmc_energy_async(box_b);
	if (rc != MPMC_OK) {
		mpmc_result drop;
		(void)mpmc_energy_wait(box_a, &drop); // leave nothing in flight behind an error
		return rc;
	}
	const int ra = mpmc_energy_wait(box_a, out_a);
	const int rb = mpmc_energy_wait(box_b, out_b);
	return ra != MPMC_OK ? ra : rb;
}

// boltzmann_factor_NVT_Gibbs, reference src/SimulationControl.Gibbs.cpp:358-522 (same branch order, same expressions)
extern "C" int mpmc_gibbs_boltzmann_factor(const mpmc_gibbs_move *m, double boltzmann_factor[2], double energy[2]) {
	if (!m || !boltzmann_factor) return MPMC_ERR_ARG;
	const double dE[2] = {m->final_energy[0] - m->init_energy[0], m->final_energy[1] - m->init_energy[1]};
	const int mv0 = m->movetype[0], mv1 = m->movetype[1];
	const bool fin0 = std::isfinite(m->final_energy[0]), fin1 = std::isfinite(m->final_energy[1]);
	// a bad contact on a coordinated move rejects it for both systems (:372-380); on the other moves nothing is touched
	if (!fin0 || !fin1) {
		if (mv0 == MPMC_MOVETYPE_INSERT || mv0 == MPMC_MOVETYPE_REMOVE || mv0 == MPMC_MOVETYPE_VOLUME) {
			boltzmann_factor[0] = boltzmann_factor[1] = 0.0;
			if (energy) energy[0] = energy[1] = kMaxValue;
		}
		return MPMC_OK;
	}
	if (mv0 == MPMC_MOVETYPE_DISPLACE || mv1 == MPMC_MOVETYPE_DISPLACE) { // independent displacements (:388-414)
		if (mv0 != mv1) return MPMC_ERR_INVALID_MC_MOVE;
		boltzmann_factor[0] = std::exp(-dE[0] / m->temperature);
		boltzmann_factor[1] = std::exp(-dE[1] / m->temperature);
		return MPMC_OK;
	}
	if ((mv0 == MPMC_MOVETYPE_INSERT && mv1 == MPMC_MOVETYPE_REMOVE) || (mv0 == MPMC_MOVETYPE_REMOVE && mv1 == MPMC_MOVETYPE_INSERT)) {
		// transfer from box A (the one that loses the molecule) to box B (:421-441)
		const int A = (mv0 == MPMC_MOVETYPE_REMOVE) ? 0 : 1, B = 1 - A;
		const double V_A = m->volume[A], N_A = m->N[A], V_B = m->volume[B], N_B = m->N[B];
		const double Beta = 1.0 / m->temperature;
		boltzmann_factor[0] = boltzmann_factor[1] = (N_A / V_A) * (V_B / (N_B + 1)) * std::exp(-Beta * dE[A] - Beta * dE[B]);
		return MPMC_OK;
	}
	if (mv0 == MPMC_MOVETYPE_VOLUME) { // coupled volume shift (:447-470)
		if (mv0 != mv1) return MPMC_ERR_INVALID_MC_MOVE;
		const double beta = 1.0 / m->temperature;
		const double dV = m->checkpoint_volume_0 - m->volume[0];
		boltzmann_factor[0] = boltzmann_factor[1] = std::pow((m->volume[0] + dV) / m->volume[0], m->N[0]) * std::pow((m->volume[1] - dV) / m->volume[1], m->N[1]) *
		                                            std::exp(-beta * dE[0] - beta * dE[1]);
		return MPMC_OK;
	}
	if (mv0 == MPMC_MOVETYPE_SPINFLIP) return MPMC_ERR_UNSUPPORTED; // rotational partition functions (:475-519): quantum rotation is outside the energy path
	return MPMC_ERR_INVALID_MC_MOVE_KIND; // :521
}
