// tests/cpp/facade_check.cpp -- host C++ written against include/mpmc_system.hpp exactly the way a Monte Carlo
// driver is written against the reference's System: fill atoms + options, update_pbc(), energy(), read observables.
// Input: a whitespace text dump produced by tests/test_gpu_cpp_facade.py from a golden fixture.
// Output: one JSON line (%.17g) that the test compares with the reference's golden values.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mpmc_system.hpp"

static bool load(const char *path, mpmc::System &s) {
	FILE *f = std::fopen(path, "r");
	if (!f) return false;
	int n = 0;
	if (std::fscanf(f, "%d", &n) != 1) return false;
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++)
			if (std::fscanf(f, "%lf", &s.pbc.basis[i][j]) != 1) return false;
	double ea, pea;
	if (std::fscanf(f, "%d %d %d %d %d %d %d %d %lf %lf %lf %lf %lf", &s.rd_only, &s.rd_lrc, &s.polarization, &s.polar_iterative, &s.polar_ewald,
	                &s.polar_max_iter, &s.polar_rrms, &s.ewald_kmax, &s.polar_precision, &s.polar_gamma, &s.polar_damp, &ea, &pea) != 13)
		return false;
	if (ea > 0) {
		s.ewald_alpha = ea;
		s.ewald_alpha_set = 1;
	}
	if (pea > 0) {
		s.polar_ewald_alpha = pea;
		s.polar_ewald_alpha_set = 1;
	}
	s.atoms.resize(n);
	for (int i = 0; i < n; i++) {
		mpmc::Atom &a = s.atoms[i];
		if (std::fscanf(f, "%lf %lf %lf %lf %lf %lf %lf %lf %d %d", &a.pos[0], &a.pos[1], &a.pos[2], &a.mass, &a.charge, &a.polarizability, &a.epsilon,
		                &a.sigma, &a.molecule, &a.frozen) != 10)
			return false;
	}
	std::fclose(f);
	s.update_pbc();
	s.atoms_changed();
	return true;
}

int main(int argc, char **argv) {
	if (argc < 2) return 2;
	try {
		mpmc::System s;
		if (!load(argv[1], s)) {
			std::fprintf(stderr, "cannot read %s\n", argv[1]);
			return 2;
		}
		const double e = s.energy();
		std::printf("{\"energy\": %.17g, \"rd\": %.17g, \"es\": %.17g, \"polar\": %.17g, \"iters\": %.17g, \"failed\": %d, \"N\": %.17g, \"NU\": %.17g, "
		            "\"mu0\": [%.17g, %.17g, %.17g], \"lj\": %.17g, \"coulombic\": %.17g",
		            e, s.observables->rd_energy, s.observables->coulombic_energy, s.observables->polarization_energy,
		            s.nodestats->polarization_iterations, s.iterator_failed, s.observables->N, s.observables->NU, s.atoms[0].mu[0], s.atoms[0].mu[1],
		            s.atoms[0].mu[2], s.lj(), s.rd_only ? 0.0 : s.coulombic());

		// MC-style use: displace one atom, re-evaluate, move back (reject), energy must return bit for bit
		const double old = s.atoms[1].pos[0];
		s.atoms[1].pos[0] += 0.25;
		s.move_atoms(1, 1);
		const double e_trial = s.energy();
		s.atoms[1].pos[0] = old;
		s.move_atoms(1, 1);
		const double e_back = s.energy();
		std::printf(", \"e_trial\": %.17g, \"e_back\": %.17g", e_trial, e_back);

		// path-integral aggregate over 4 beads that share the box: ordered mean of 4 energies
		mpmc::PathIntegralEnsemble pi;
		std::vector<mpmc::System *> beads;
		for (int b = 0; b < 4; b++) {
			mpmc::System *t = new mpmc::System();
			load(argv[1], *t);
			for (size_t i = 0; i < t->atoms.size(); i++) t->atoms[i].pos[b % 3] += 0.01 * (b + 1) * ((i % 2) ? 1 : -1);
			beads.push_back(t);
		}
		pi.systems = beads;
		pi.nSys = 4;
		const double v = pi.PI_calculate_potential();
		double acc = 0;
		for (auto *t : beads) acc += t->observables->rd_energy;
		std::printf(", \"pi_V\": %.17g, \"pi_rd\": %.17g, \"pi_rd_check\": %.17g", v, pi.sys_observables.rd_energy, acc / 4);
		for (auto *t : beads) delete t;

		// error convention: an out-of-scope switch must throw the reference's int code (unsupported_setting = 4004)
		int thrown = 0;
		try {
			mpmc::System u;
			load(argv[1], u);
			u.rd_only = 0;
			u.polarization = 1;
			u.polar_iterative = 1;
			u.damp_type = mpmc::DAMPING_LINEAR;
			u.energy();
		} catch (int code) {
			thrown = code;
		}
		std::printf(", \"thrown\": %d}\n", thrown);
	} catch (int code) {
		std::printf("{\"error\": %d}\n", code);
		return 1;
	}
	return 0;
}
