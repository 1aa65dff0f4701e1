// tests/cpp/pimc_check.cpp -- the PI-NVT driver of include/mpmc_pimc.hpp run with the CPU ORACLE as the per-image energy
// evaluator (test infrastructure: links oracle/libmpmc_oracle.so; the product never does).  It pins the driver's host logic --
// random-number stream, move generation, Boltzmann factors, accept / restore, estimator bookkeeping -- against the rows the stock
// reference binary wrote for the same input (tests/golden/pi001, pi_ion27) without needing a GPU.
//   pimc_check INPUT.in P OUT_DIR [--trial]     writes OUT_DIR/energy.dat and OUT_DIR/final-%04d.pqr, prints a JSON summary
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "mpmc_pimc.hpp"
#include "../../oracle/mpmc_oracle.h"

namespace {
// the members PathIntegralNVT / PathIntegralEnsembleT use, backed by orc_energy()
class OracleBead {
public:
	int rd_only = 0, rd_lrc = 1, polarization = 0, polar_iterative = 0, polar_ewald = 0, polar_max_iter = 10, polar_gs = 0, polar_rrms = 0;
	int damp_type = mpmc::DAMPING_EXPONENTIAL, ewald_kmax = 7, wolf = 0, feynman_hibbs = 0, feynman_hibbs_order = 0;
	double temperature = 0, polar_precision = 0, polar_gamma = 1.0, polar_damp = 0, ewald_alpha = 0.5, polar_ewald_alpha = 0.5;
	mpmc::PeriodicBoundary pbc;
	std::vector<mpmc::Atom> atoms;
	int iterator_failed = 0;
	mpmc::observables_t obs_, *observables = &obs_;

	void atoms_changed() {}
	void move_atoms(int, int) {}
	void energy_async() {}
	void hint_in_flight(int) {}
	double energy_wait() { return energy(); }
	unsigned int countN() {
		unsigned int c = 0;
		for (size_t i = 0; i < atoms.size(); i++)
			if ((i == 0 || atoms[i].molecule != atoms[i - 1].molecule) && !atoms[i].frozen) c++;
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
	// trial moves behind the facade's names: the oracle has no delta path, the trial configuration is evaluated in full
	mpmc_result trial_{};
	int trial_first_ = 0;
	std::vector<double> trial_pos_;
	mpmc::observables_t trial_obs_;
	void energy_trial_async(int first, int count, const double *new_pos) {
		trial_first_ = first;
		trial_pos_.assign(new_pos, new_pos + 3 * (size_t)count);
	}
	double energy_trial_wait() {
		const mpmc::observables_t keep = *observables;
		const int keep_failed = iterator_failed;
		std::vector<double> old(trial_pos_.size());
		for (size_t k = 0; k < trial_pos_.size() / 3; k++)
			for (int d = 0; d < 3; d++) {
				old[3 * k + d] = atoms[trial_first_ + k].pos[d];
				atoms[trial_first_ + k].pos[d] = trial_pos_[3 * k + d];
			}
		const double e = energy();
		trial_obs_ = *observables;
		trial_ = mpmc_result{};
		trial_.energy = e;
		trial_.rd_energy = observables->rd_energy;
		trial_.coulombic_energy = observables->coulombic_energy;
		trial_.polarization_energy = observables->polarization_energy;
		trial_.vdw_energy = observables->vdw_energy;
		trial_.iterator_failed = iterator_failed;
		for (size_t k = 0; k < trial_pos_.size() / 3; k++)
			for (int d = 0; d < 3; d++) atoms[trial_first_ + k].pos[d] = old[3 * k + d];
		*observables = keep;
		iterator_failed = keep_failed;
		return e;
	}
	const mpmc_result &trial_result() const { return trial_; }
	void accept_trial() {
		for (size_t k = 0; k < trial_pos_.size() / 3; k++)
			for (int d = 0; d < 3; d++) atoms[trial_first_ + k].pos[d] = trial_pos_[3 * k + d];
		*observables = trial_obs_;
		iterator_failed = trial_.iterator_failed;
	}
	void reject_trial() {}

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

int main(int argc, char **argv) {
	if (argc < 4) return 2;
	const bool trial = argc > 4 && std::string(argv[4]) == "--trial";
	try {
		const int P = std::atoi(argv[2]);
		const std::string outdir = argv[3];
		mpmc::System proto; // the facade's reader fills options + geometry; the evaluation goes to the oracle
		mpmc::load_system(argv[1], proto);
		std::vector<std::unique_ptr<OracleBead>> beads;
		mpmc::PathIntegralNVT<OracleBead> mc;
		mc.cfg = mpmc::read_pimc_settings(argv[1]);
		mc.moltype_names = proto.moltype_names;
		for (int b = 0; b < P; b++) {
			beads.emplace_back(new OracleBead());
			OracleBead &o = *beads.back();
			o.rd_only = proto.rd_only, o.rd_lrc = proto.rd_lrc, o.polarization = proto.polarization, o.polar_iterative = proto.polar_iterative;
			o.polar_ewald = proto.polar_ewald, o.polar_max_iter = proto.polar_max_iter, o.polar_gs = proto.polar_gs, o.polar_rrms = proto.polar_rrms;
			o.ewald_kmax = proto.ewald_kmax, o.wolf = proto.wolf, o.feynman_hibbs = proto.feynman_hibbs, o.feynman_hibbs_order = proto.feynman_hibbs_order;
			o.polar_precision = proto.polar_precision, o.polar_gamma = proto.polar_gamma, o.polar_damp = proto.polar_damp;
			o.ewald_alpha = proto.ewald_alpha, o.polar_ewald_alpha = proto.polar_ewald_alpha;
			o.pbc = proto.pbc;
			o.atoms = proto.atoms;
			if (mc.cfg.parallel_restarts) { // one geometry per image: JOB.restart-%04d.pqr next to the input file
				char name[64];
				std::snprintf(name, sizeof name, ".restart-%04d.pqr", b);
				mpmc::System image;
				mpmc::read_pqr(mpmc::io_detail::dirname_of(argv[1]) + "/" + mc.cfg.job_name + name, image);
				o.atoms = image.atoms;
			}
			mc.systems.push_back(&o);
		}
		mc.init();
		mc.use_trial_moves = trial;
		FILE *fp = std::fopen((outdir + "/energy.dat").c_str(), "w");
		if (!fp) return 3;
		mc.run(fp);
		std::fclose(fp);
		for (int b = 0; b < P; b++) {
			char name[64];
			std::snprintf(name, sizeof name, "/final-%04d.pqr", b);
			mpmc::System out;
			out.atoms = beads[b]->atoms;
			mpmc::write_pqr(outdir + name, out);
		}
		std::printf("{\"P\": %d, \"steps\": %u, \"AR\": %.5f, \"accept\": %ld, \"reject\": %ld, \"accept_bead\": %ld, \"reject_bead\": %ld, \"accept_displace\": %ld, "
		            "\"reject_displace\": %ld, \"energy_calls\": %ld}\n",
		            P, mc.step, mc.acceptance_rate(), mc.accept, mc.reject, mc.accept_bead, mc.reject_bead, mc.accept_displace, mc.reject_displace,
		            mc.energy_calls);
	} catch (int code) {
		std::printf("{\"error\": %d}\n", code);
		return 1;
	}
	return 0;
}
