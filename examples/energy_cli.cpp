// examples/energy_cli.cpp -- single-point energy of a reference input file through the C++ facade.
//   energy_cli INPUT.in            one JSON line: energy components (%.17g), counts, first dipole
//   energy_cli INPUT.in --time N   N back-to-back energy() calls through the C++ facade: microseconds per call (no Python in the loop)
//   energy_cli INPUT.in --parse    parse only (no GPU): n, basis, options and per-atom arrays, for checking the readers
//   energy_cli INPUT.in --write OUT.pqr   re-write the geometry in the reference's PQR row layout
//   energy_cli INPUT.in --pi B0.pqr B1.pqr ...          path-integral energy estimator over the P bead geometries given
//   energy_cli INPUT.in --pi-rccl B0.pqr B1.pqr ...     the same with the cross-rank exchange on RCCL (a one-rank communicator of
//                                                      mpmc_comm_init_rank bound to the ensemble: PathIntegralEnsemble::use_comm)
//   energy_cli INPUT.in --pi-kinetic B0.pqr B1.pqr ...  kinetic part only (host arithmetic, no GPU)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "mpmc_io.hpp"

int main(int argc, char **argv) {
	if (argc < 2) {
		std::fprintf(stderr, "usage: %s INPUT.in [--parse | --write OUT.pqr]\n", argv[0]);
		return 2;
	}
	try {
		if (argc > 3 && !std::strncmp(argv[2], "--pi", 4)) {
			const bool kinetic_only = !std::strcmp(argv[2], "--pi-kinetic");
			std::vector<std::unique_ptr<mpmc::System>> beads;
			mpmc::PathIntegralEnsemble pi;
			for (int b = 3; b < argc; b++) {
				beads.emplace_back(new mpmc::System());
				mpmc::read_input(argv[1], *beads.back());
				mpmc::read_pqr(argv[b], *beads.back());
				beads.back()->update_pbc();
				pi.systems.push_back(beads.back().get());
			}
			pi.nSys = (int)beads.size();
			pi.temperature = beads[0]->temperature;
			mpmc_comm *comm = nullptr;
			if (!std::strcmp(argv[2], "--pi-rccl")) {
				char id[MPMC_COMM_ID_BYTES];
				if (mpmc_comm_unique_id(id) != MPMC_OK || mpmc_comm_init_rank(&comm, 1, 0, id, 0) != MPMC_OK) {
					std::printf("{\"error\": \"%s\"}\n", mpmc_comm_last_error(nullptr));
					return 1;
				}
				pi.use_comm(comm);
			}
			const mpmc::observables_t &o = pi.sys_observables;
			if (kinetic_only) {
				const double k = pi.PI_calculate_kinetic();
				std::printf("{\"P\": %d, \"N\": %.17g, \"kinetic\": %.17g, \"chain_mass_len2\": %.17g, \"chain0\": %.17g}\n", pi.nSys, o.N, k,
				            pi.PI_chain_mass_length2_ENTIRE_SYSTEM(), pi.PI_chain_mass_length2(0));
				return 0;
			}
			const double e = pi.PI_calculate_energy();
			std::printf("{\"P\": %d, \"N\": %.17g, \"energy\": %.17g, \"kinetic\": %.17g, \"rd\": %.17g, \"es\": %.17g, \"polar\": %.17g}\n", pi.nSys,
			            o.N, e, o.kinetic_energy, o.rd_energy, o.coulombic_energy, o.polarization_energy);
			if (comm) mpmc_comm_destroy(comm);
			return 0;
		}
		mpmc::System s;
		mpmc::load_system(argv[1], s);
		if (argc > 3 && !std::strcmp(argv[2], "--write")) {
			mpmc::write_pqr(argv[3], s);
			return 0;
		}
		if (argc > 2 && !std::strcmp(argv[2], "--parse")) {
			std::printf("{\"n\": %d, \"volume\": %.17g, \"cutoff\": %.17g, \"ewald_alpha\": %.17g, \"polar_ewald_alpha\": %.17g, "
			            "\"options\": [%d, %d, %d, %d, %d, %d, %d, %d, %.17g, %.17g, %.17g], \"unsupported\": %llu, \"atoms\": [",
			            (int)s.atoms.size(), s.pbc.volume, s.pbc.cutoff, s.ewald_alpha, s.polar_ewald_alpha, s.rd_only, s.rd_lrc, s.polarization,
			            s.polar_iterative, s.polar_ewald, s.polar_max_iter, s.polar_rrms, s.ewald_kmax, s.polar_precision, s.polar_gamma, s.polar_damp,
			            (unsigned long long)s.unsupported_flags);
			for (size_t i = 0; i < s.atoms.size(); i++) {
				const mpmc::Atom &a = s.atoms[i];
				std::printf("%s[%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %d, %d]", i ? ", " : "", a.pos[0], a.pos[1], a.pos[2], a.mass,
				            a.charge, a.polarizability, a.epsilon, a.sigma, a.molecule, a.frozen);
			}
			std::printf("]}\n");
			return 0;
		}
		if (argc > 3 && !std::strcmp(argv[2], "--time")) {
			const int reps = std::atoi(argv[3]);
			s.eager_dipoles = false; // per-atom vectors are fetched on demand, not after every call
			double e = s.energy();
			e = s.energy();
			const auto t0 = std::chrono::steady_clock::now();
			for (int k = 0; k < reps; k++) e = s.energy();
			const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps > 0 ? reps : 1);
			std::printf("{\"natoms\": %d, \"total\": %.17g, \"calls\": %d, \"us_per_call\": %.3f}\n", s.natoms, e, reps, us);
			return 0;
		}
		const double e = s.energy();
		const mpmc::observables_t *o = s.observables;
		std::printf("{\"natoms\": %d, \"total\": %.17g, \"rd\": %.17g, \"es\": %.17g, \"polar\": %.17g, \"es_real\": %.17g, \"es_recip\": %.17g, "
		            "\"es_self\": %.17g, \"n_lj_in_cutoff\": %lld, \"n_es_in_cutoff\": %lld, \"polar_iterations\": %d, \"mu0\": [%.17g, %.17g, %.17g]}\n",
		            s.natoms, e, o->rd_energy, o->coulombic_energy, o->polarization_energy, s.last_result.es_real, s.last_result.es_recip,
		            s.last_result.es_self, (long long)s.last_result.n_lj_in_cutoff, (long long)s.last_result.n_es_in_cutoff,
		            s.last_result.polar_iterations, s.atoms[0].mu[0], s.atoms[0].mu[1], s.atoms[0].mu[2]);
	} catch (int code) {
		std::printf("{\"error\": %d}\n", code);
		return 1;
	}
	return 0;
}
