// include/mpmc_gibbs_run.hpp -- shared body of examples/gibbs_nvt.cpp (HIP path) and tests/cpp/gibbs_check.cpp (oracle as evaluator):
// load the two boxes of an `ensemble nvt_gibbs` input, run GibbsNVT, print the trajectory as one JSON object in the layout of
// oracle/ref_gibbs_traj.cpp (so the same test code reads both).
#pragma once
#include <cstdio>
#include <string>

#include "mpmc_gibbs.hpp"

namespace mpmc {

template <class SystemT>
inline void print_gibbs_box(const SystemT &s, const char *key) {
	std::printf("\"%s\": {\"basis\": [", key);
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) std::printf("%.17g%s", s.pbc.basis[i][j], (i == 2 && j == 2) ? "" : ", ");
	std::printf("], \"mol_id\": [");
	for (size_t k = 0; k < s.atoms.size(); k++) std::printf("%s%d", k ? ", " : "", s.atoms[k].molecule);
	std::printf("], \"charge\": [");
	for (size_t k = 0; k < s.atoms.size(); k++) std::printf("%s%.17g", k ? ", " : "", s.atoms[k].charge);
	std::printf("], \"pos\": [");
	for (size_t k = 0; k < s.atoms.size(); k++) std::printf("%s%.17g, %.17g, %.17g", k ? ", " : "", s.atoms[k].pos[0], s.atoms[k].pos[1], s.atoms[k].pos[2]);
	std::printf("]}");
}

// a and b: already loaded boxes (options, cell, atoms); steps < 0: cfg.numsteps
template <class SystemT>
inline void run_gibbs_and_print(SystemT &a, SystemT &b, const GibbsSettings &cfg, int steps) {
	GibbsNVT<SystemT> mc(a, b);
	mc.cfg = cfg;
	if (steps >= 0) mc.cfg.numsteps = (unsigned int)steps;
	mc.init();
	std::printf("{\"initial_energy\": [%.17g, %.17g], \"N\": [%.17g, %.17g], \"volume\": [%.17g, %.17g], \"volume_probability\": %.17g,\n \"steps\": [",
	            mc.initial_energy[0], mc.initial_energy[1], a.observables->N, b.observables->N, a.pbc.volume, b.pbc.volume, mc.cfg.volume_probability);
	mc.run();
	for (size_t s = 0; s < mc.trace.size(); s++) {
		const typename GibbsNVT<SystemT>::Record &r = mc.trace[s];
		std::printf("%s\n  {\"step\": %d, \"movetype\": [%d, %d], \"final_energy\": [%.17g, %.17g], \"boltzmann_factor\": [%.17g, %.17g], \"accepted\": [%d, %d], "
		            "\"energy\": [%.17g, %.17g], \"N\": [%.17g, %.17g], \"volume\": [%.17g, %.17g], \"natoms\": [%d, %d]}",
		            s ? "," : "", (int)s + 1, r.movetype[0], r.movetype[1], r.final_energy[0], r.final_energy[1], r.boltzmann_factor[0], r.boltzmann_factor[1],
		            r.accepted[0], r.accepted[1], r.energy[0], r.energy[1], r.N[0], r.N[1], r.volume[0], r.volume[1], r.natoms[0], r.natoms[1]);
	}
	std::printf("],\n \"accept\": [%ld, %ld], \"reject\": [%ld, %ld], \"energy_calls\": %ld,\n ", mc.accept[0], mc.accept[1], mc.reject[0], mc.reject[1], mc.energy_calls);
	print_gibbs_box(a, "final_box_0");
	std::printf(",\n ");
	print_gibbs_box(b, "final_box_1");
	std::printf("}\n");
}

} // namespace mpmc
