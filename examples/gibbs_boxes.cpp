// examples/gibbs_boxes.cpp -- the two boxes of a Gibbs ensemble through the C++ facade (include/mpmc_gibbs.hpp).
//   gibbs_boxes A.in B.in [--displace dx dy dz]
// Box A goes to device 0, box B to device 1 (device 0 when the node shows one GPU).  Prints one JSON line: the devices, both initial
// energies (Gibbs.cpp:152) and -- with --displace, which shifts the first molecule of each box -- the trial energies (:179-180) and the
// two Boltzmann factors of an independent displacement move (boltzmann_factor_NVT_Gibbs :388-414).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mpmc_gibbs.hpp"
#include "mpmc_io.hpp"

int main(int argc, char **argv) {
	if (argc < 3) {
		std::fprintf(stderr, "usage: %s A.in B.in [--displace dx dy dz]\n", argv[0]);
		return 2;
	}
	try {
		mpmc::System a, b;
		mpmc::load_system(argv[1], a);
		mpmc::load_system(argv[2], b);
		mpmc::GibbsBoxes g(a, b);
		g.place_on_devices();
		g.mc_initial_energy();
		std::printf("{\"devices\": [%d, %d], \"initial_energy\": [%.17g, %.17g]", a.device, b.device, g.initial_energy[0], g.initial_energy[1]);
		if (argc >= 7 && !std::strcmp(argv[3], "--displace")) {
			const double d[3] = {std::atof(argv[4]), std::atof(argv[5]), std::atof(argv[6])};
			for (mpmc::System *s : g.systems) {
				int count = 0;
				while (count < (int)s->atoms.size() && s->atoms[count].molecule == s->atoms[0].molecule) count++;
				for (int i = 0; i < count; i++)
					for (int p = 0; p < 3; p++) s->atoms[i].pos[p] += d[p];
				s->move_atoms(0, count);
			}
			g.energy();
			g.boltzmann_factor_NVT_Gibbs(MPMC_MOVETYPE_DISPLACE, MPMC_MOVETYPE_DISPLACE, a.temperature, a.pbc.volume);
			std::printf(", \"final_energy\": [%.17g, %.17g], \"boltzmann_factor\": [%.17g, %.17g], \"temperature\": %.17g", g.final_energy[0],
			            g.final_energy[1], g.boltzmann_factor[0], g.boltzmann_factor[1], a.temperature);
		}
		std::printf("}\n");
	} catch (int code) {
		std::printf("{\"error\": %d}\n", code);
		return 1;
	}
	return 0;
}
