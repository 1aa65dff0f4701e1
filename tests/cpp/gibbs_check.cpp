// tests/cpp/gibbs_check.cpp -- the Gibbs-ensemble driver of include/mpmc_gibbs.hpp run with the CPU ORACLE as the energy evaluator
// (test infrastructure: links oracle/libmpmc_oracle.so; the product never does).  It pins the driver's host logic -- random-number
// streams, move selection, displacement / volume exchange / particle transfer, Boltzmann factors, accept / restore -- against the
// trajectory the reference's own functions made for the same input (tests/golden/gibbs_*, oracle/ref_gibbs_traj.cpp), no GPU needed.
//   gibbs_check INPUT.in [STEPS]      prints the trajectory JSON of include/mpmc_gibbs_run.hpp
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "mpmc_gibbs_run.hpp"
#include "../../oracle/mpmc_oracle.h"

namespace {
class OracleBox {
public:
	int rd_only = 0, rd_lrc = 1, polarization = 0, polar_iterative = 0, polar_ewald = 0, polar_max_iter = 10, polar_gs = 0, polar_rrms = 0;
	int damp_type = mpmc::DAMPING_EXPONENTIAL, ewald_kmax = 7, wolf = 0, feynman_hibbs = 0, feynman_hibbs_order = 0;
	double temperature = 0, polar_precision = 0, polar_gamma = 1.0, polar_damp = 0, ewald_alpha = 0.5, polar_ewald_alpha = 0.5;
	mpmc::PeriodicBoundary pbc;
	std::vector<mpmc::Atom> atoms;
	int iterator_failed = 0;
	mpmc::observables_t obs_, *observables = &obs_;

	int natoms = 0, ewald_alpha_set = 0, polar_ewald_alpha_set = 0;
	void update_pbc() { // System::update_pbc, src/System.cpp:859-876
		pbc.update();
		if (ewald_alpha_set != 1) ewald_alpha = 3.5 / pbc.cutoff;
		if (polar_ewald_alpha_set != 1) polar_ewald_alpha = 3.5 / pbc.cutoff;
	}
	void atoms_changed() {}
	void move_atoms(int, int) {}
	void energy_async() {}
	void hint_in_flight(int) {}
	double energy_wait() { return energy(); }
	unsigned int countN() {
		unsigned int c = 0;
		for (size_t i = 0; i < atoms.size(); i++)
			if ((i + 1 == atoms.size() || atoms[i + 1].molecule != atoms[i].molecule) && !atoms[i].frozen) c++; // last row decides (src/System.cpp:684)
		observables->N = c;
		return c;
	}
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
			movable.push_back(atoms[a0].frozen ? 0 : 1);
			a0 = a1;
		}
	}
	double energy() {
		const int n = (int)atoms.size();
		std::vector<double> pos(3 * (size_t)n), q(n), al(n), ep(n), sg(n), ms(n);
		std::vector<int> mol(n), fr(n), dp(n);
		for (int i = 0; i < n; i++) {
			const mpmc::Atom &a = atoms[i];
			for (int p = 0; p < 3; p++) pos[3 * i + p] = a.pos[p];
			q[i] = a.charge, al[i] = a.polarizability, ep[i] = a.epsilon, sg[i] = a.sigma, ms[i] = a.mass;
			mol[i] = a.molecule, fr[i] = a.frozen, dp[i] = (a.c6 != 0 || a.c8 != 0 || a.c10 != 0);
		}
		orc_system s{};
		s.n = n;
		s.pos = pos.data(), s.charge = q.data(), s.polarizability = al.data(), s.epsilon = ep.data(), s.sigma = sg.data();
		s.mol_id = mol.data(), s.frozen = fr.data(), s.has_disp = dp.data(), s.mass = ms.data();
		for (int i = 0; i < 9; i++) s.basis[i] = (&pbc.basis[0][0])[i], s.recip[i] = (&pbc.reciprocal_basis[0][0])[i];
		s.volume = pbc.volume, s.cutoff = pbc.cutoff;
		s.rd_only = rd_only, s.rd_lrc = rd_lrc, s.polarization = polarization, s.polar_iterative = polar_iterative, s.polar_ewald = polar_ewald;
		s.polar_max_iter = polar_max_iter, s.polar_gs = polar_gs, s.polar_rrms = polar_rrms, s.ewald_kmax = ewald_kmax;
		s.polar_precision = polar_precision, s.polar_gamma = polar_gamma, s.polar_damp = polar_damp;
		s.ewald_alpha = ewald_alpha, s.polar_ewald_alpha = polar_ewald_alpha;
		s.wolf = wolf, s.feynman_hibbs = feynman_hibbs, s.feynman_hibbs_order = feynman_hibbs_order, s.temperature = temperature;
		orc_result r{};
		orc_energy(&s, &r, nullptr, nullptr, nullptr);
		observables->energy = r.energy;
		observables->rd_energy = r.rd_energy;
		observables->coulombic_energy = r.coulombic_energy;
		observables->polarization_energy = r.polarization_energy;
		observables->vdw_energy = r.vdw_energy;
		countN();
		observables->NU = observables->N * r.energy;
		iterator_failed = r.iterator_failed;
		return r.energy;
	}
};
} // namespace

static void copy_box(const mpmc::System &p, OracleBox &o) {
	o.rd_only = p.rd_only, o.rd_lrc = p.rd_lrc, o.polarization = p.polarization, o.polar_iterative = p.polar_iterative;
	o.polar_ewald = p.polar_ewald, o.polar_max_iter = p.polar_max_iter, o.polar_gs = p.polar_gs, o.polar_rrms = p.polar_rrms;
	o.ewald_kmax = p.ewald_kmax, o.wolf = p.wolf, o.feynman_hibbs = p.feynman_hibbs, o.feynman_hibbs_order = p.feynman_hibbs_order;
	o.polar_precision = p.polar_precision, o.polar_gamma = p.polar_gamma, o.polar_damp = p.polar_damp;
	o.ewald_alpha = p.ewald_alpha, o.polar_ewald_alpha = p.polar_ewald_alpha;
	o.ewald_alpha_set = p.ewald_alpha_set, o.polar_ewald_alpha_set = p.polar_ewald_alpha_set;
	o.temperature = p.temperature;
	o.pbc = p.pbc;
	o.atoms = p.atoms;
	o.natoms = (int)p.atoms.size();
}

int main(int argc, char **argv) {
	if (argc < 2) return 2;
	try {
		const mpmc::GibbsSettings cfg = mpmc::read_gibbs_settings(argv[1]);
		mpmc::System pa, pb; // the facade's readers fill options + geometry; the evaluation goes to the oracle
		const std::string pqr_a = mpmc::read_input(argv[1], pa);
		(void)mpmc::read_input(argv[1], pb);
		std::string pqr_b = cfg.pqr_input_B.empty() ? pqr_a : cfg.pqr_input_B;
		if (pqr_b[0] != '/') pqr_b = mpmc::io_detail::dirname_of(argv[1]) + "/" + pqr_b;
		mpmc::read_pqr(pqr_a, pa);
		mpmc::read_pqr(pqr_b, pb);
		pa.update_pbc();
		pb.update_pbc();
		OracleBox a, b;
		copy_box(pa, a);
		copy_box(pb, b);
		mpmc::run_gibbs_and_print(a, b, cfg, argc > 2 ? std::atoi(argv[2]) : -1);
	} catch (int code) {
		std::printf("{\"error\": %d}\n", code);
		return 1;
	}
	return 0;
}
