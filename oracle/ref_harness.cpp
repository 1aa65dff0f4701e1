// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, never part of the product path.
//
// Driver that links the *reference's own* object files (compiled in place from
// /root/reference/src by oracle/Makefile into oracle/_ref/) and calls
// System::energy() (reference src/System.Energy.cpp:19) and its public component
// functions at full precision.  It is used for two things only:
//   1. generating the golden vectors under tests/golden/ (oracle/make_golden.py)
//   2. the "reference" CPU baseline leg of bench.py (kind = "reference")
// No reference source text lives in this file; it only uses public members
// declared in the reference headers (System.h, SimulationControl.h).
//
// usage: ref_harness INPUT.in [--time K] [--dump-atoms] [--sample-atoms STRIDE] [--dump-com] [--amatrix i,j ...]
//   --sample-atoms S : per-atom mu / ef_static / ef_induced of atoms 0, S, 2S, ... only (large boxes)
//   --dump-com       : what pairs() leaves behind through update_com() + wrap_all() (src/System.cpp:1347-1425):
//                      Molecule::com, Molecule::wrapped_com, Atom::wrapped_pos
// Prints one JSON object on the LAST line of stdout (the reference prints its own
// banner lines before it).

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "Atom.h"
#include "Molecule.h"
#include "Pair.h"
#include "SimulationControl.h"
#include "System.h"
#include "constants.h"

// globals the reference's main.cpp normally owns (src/main.cpp:18-20)
int rank = 0;
int size = 0;
bool mpi = false;

static double now_s() {
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void force_full_recompute(System &s) {
	// energy() re-flags every pair when observables->energy == 0.0 (System.Energy.cpp:42)
	s.flag_all_pairs();
	s.observables->energy = 0.0;
}

int main(int argc, char **argv) {
	if (argc < 2) {
		fprintf(stderr, "usage: %s INPUT.in [--time K] [--dump-atoms] [--amatrix i,j ...]\n", argv[0]);
		return 2;
	}
	int time_reps = 0;
	bool dump_atoms = false, dump_com = false;
	int sample_stride = 0;
	std::vector<std::pair<int, int>> aspots;
	for (int a = 2; a < argc; a++) {
		if (!strcmp(argv[a], "--time") && a + 1 < argc)
			time_reps = atoi(argv[++a]);
		else if (!strcmp(argv[a], "--dump-atoms"))
			dump_atoms = true;
		else if (!strcmp(argv[a], "--dump-com"))
			dump_com = true;
		else if (!strcmp(argv[a], "--sample-atoms") && a + 1 < argc)
			sample_stride = atoi(argv[++a]);
		else if (!strcmp(argv[a], "--amatrix")) {
			while (a + 1 < argc && argv[a + 1][0] != '-') {
				int i = 0, j = 0;
				if (sscanf(argv[++a], "%d,%d", &i, &j) == 2) aspots.push_back({i, j});
			}
		}
	}

	try {
		SimulationControl sc(argv[1], 0, false, nullptr);
		sc.initializeSimulationObjects();
		System &s = sc.sys;

		double t0 = now_s();
		double total = s.energy();
		double t_first = now_s() - t0;

		double rd = s.observables->rd_energy;
		double es = s.observables->coulombic_energy;
		double pol = s.observables->polarization_energy;
		int n = s.natoms;

		// component functions (public: src/System.h:346-368), after re-flagging so that
		// every pair is recomputed; they are pure functions of the pair list state.
		double es_real = 0, es_recip = 0, es_self = 0;
		if (!(s.use_sg || s.rd_only)) {
			s.flag_all_pairs();
			es_real = s.coulombic_real();
			es_recip = s.coulombic_reciprocal();
			es_self = s.coulombic_self();
		}

		// predicate counts and LRC parts, walking the pair lists the way lj()/coulombic_real() do
		long long n_pairs = 0, n_intra = 0, n_rd_excl = 0, n_es_excl = 0, n_lj_in = 0, n_es_in = 0, n_frozen = 0;
		double lrc_pair = 0, lrc_self = 0, lj_pairs_only = 0;
		double cutoff = s.pbc.cutoff;
		for (int i = 0; i < n; i++) {
			for (Pair *p = s.atom_array[i]->pairs; p; p = p->next) {
				n_pairs++;
				if (s.molecule_array[i] == p->molecule) n_intra++;
				if (p->rd_excluded) n_rd_excl++;
				if (p->es_excluded) n_es_excl++;
				if (p->frozen) n_frozen++;
				if ((p->rimg - SMALL_dR < cutoff) && !p->rd_excluded && !p->frozen) n_lj_in++;
				if (!p->frozen && !((p->rimg > cutoff) || p->es_excluded)) n_es_in++;
				lrc_pair += p->lrc;
				lj_pairs_only += p->rd_energy;
			}
			if (s.rd_lrc) lrc_self += s.lj_lrc_self(s.atom_array[i], cutoff);
		}

		printf("\n{\"natoms\": %d, \"cutoff\": %.17g, \"volume\": %.17g, \"ewald_alpha\": %.17g, \"polar_ewald_alpha\": %.17g,\n",
		       n, cutoff, s.pbc.volume, s.ewald_alpha, s.polar_ewald_alpha);
		printf(" \"basis\": [");
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) printf("%.17g%s", s.pbc.basis[i][j], (i == 2 && j == 2) ? "" : ", ");
		printf("],\n \"reciprocal_basis\": [");
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) printf("%.17g%s", s.pbc.reciprocal_basis[i][j], (i == 2 && j == 2) ? "" : ", ");
		printf("],\n");
		printf(" \"total\": %.17g, \"rd\": %.17g, \"es\": %.17g, \"polar\": %.17g,\n", total, rd, es, pol);
		printf(" \"es_real\": %.17g, \"es_recip\": %.17g, \"es_self\": %.17g,\n", es_real, es_recip, es_self);
		printf(" \"lj_pairs\": %.17g, \"lrc_pair\": %.17g, \"lrc_self\": %.17g,\n", lj_pairs_only, lrc_pair, lrc_self);
		printf(" \"n_pairs\": %lld, \"n_intra\": %lld, \"n_rd_excluded\": %lld, \"n_es_excluded\": %lld, \"n_frozen\": %lld,\n",
		       n_pairs, n_intra, n_rd_excl, n_es_excl, n_frozen);
		printf(" \"n_lj_in_cutoff\": %lld, \"n_es_in_cutoff\": %lld,\n", n_lj_in, n_es_in);
		printf(" \"polar_iterations\": %.17g, \"dipole_rrms\": %.17g, \"iterator_failed\": %d,\n",
		       s.polarization ? s.nodestats->polarization_iterations : 0.0, s.observables->dipole_rrms, s.iterator_failed);
		printf(" \"N\": %.17g, \"NU\": %.17g", s.observables->N, s.observables->NU);

		if (!aspots.empty() && s.polarization && s.A_matrix) {
			printf(",\n \"amatrix\": [");
			for (size_t k = 0; k < aspots.size(); k++) {
				int i = aspots[k].first, j = aspots[k].second;
				printf("%s{\"i\": %d, \"j\": %d, \"block\": [", k ? ", " : "", i, j);
				for (int p = 0; p < 3; p++)
					for (int q = 0; q < 3; q++)
						printf("%.17g%s", s.A_matrix[3 * i + p][3 * j + q], (p == 2 && q == 2) ? "" : ", ");
				printf("]}");
			}
			printf("]");
		}

		if (dump_atoms) {
			printf(",\n \"ef_static\": [");
			for (int i = 0; i < n; i++)
				printf("%s%.17g, %.17g, %.17g", i ? ", " : "", s.atom_array[i]->ef_static[0], s.atom_array[i]->ef_static[1],
				       s.atom_array[i]->ef_static[2]);
			printf("],\n \"mu\": [");
			for (int i = 0; i < n; i++)
				printf("%s%.17g, %.17g, %.17g", i ? ", " : "", s.atom_array[i]->mu[0], s.atom_array[i]->mu[1],
				       s.atom_array[i]->mu[2]);
			printf("],\n \"ef_induced\": [");
			for (int i = 0; i < n; i++)
				printf("%s%.17g, %.17g, %.17g", i ? ", " : "", s.atom_array[i]->ef_induced[0], s.atom_array[i]->ef_induced[1],
				       s.atom_array[i]->ef_induced[2]);
			printf("]");
		}

		if (sample_stride > 0) {
			printf(",\n \"sample_stride\": %d", sample_stride);
			const char *names[3] = {"ef_static_sample", "mu_sample", "ef_induced_sample"};
			for (int w = 0; w < 3; w++) {
				printf(",\n \"%s\": [", names[w]);
				for (int i = 0; i < n; i += sample_stride) {
					const double *v = (w == 0) ? s.atom_array[i]->ef_static : (w == 1) ? s.atom_array[i]->mu : s.atom_array[i]->ef_induced;
					printf("%s%.17g, %.17g, %.17g", i ? ", " : "", v[0], v[1], v[2]);
				}
				printf("]");
			}
		}

		if (dump_com) {
			int nm = 0;
			printf(",\n \"com\": [");
			for (Molecule *m = s.molecules; m; m = m->next, nm++) printf("%s%.17g, %.17g, %.17g", nm ? ", " : "", m->com[0], m->com[1], m->com[2]);
			printf("],\n \"wrapped_com\": [");
			nm = 0;
			for (Molecule *m = s.molecules; m; m = m->next, nm++)
				printf("%s%.17g, %.17g, %.17g", nm ? ", " : "", m->wrapped_com[0], m->wrapped_com[1], m->wrapped_com[2]);
			printf("],\n \"wrapped_pos\": [");
			for (int i = 0; i < n; i++)
				printf("%s%.17g, %.17g, %.17g", i ? ", " : "", s.atom_array[i]->wrapped_pos[0], s.atom_array[i]->wrapped_pos[1],
				       s.atom_array[i]->wrapped_pos[2]);
			printf("],\n \"n_molecules\": %d", nm);
		}

		if (time_reps > 0) {
			// steady-state full recompute: every pair re-flagged, like the first step of a run
			double best = 1e300, sum = 0;
			for (int k = 0; k < time_reps; k++) {
				force_full_recompute(s);
				double a = now_s();
				volatile double e = s.energy();
				(void)e;
				double dt = now_s() - a;
				sum += dt;
				if (dt < best) best = dt;
			}
			printf(",\n \"time_first_s\": %.6g, \"time_reps\": %d, \"time_mean_s\": %.6g, \"time_best_s\": %.6g", t_first, time_reps,
			       sum / time_reps, best);
		}
		printf("}\n");
		fflush(stdout);
	} catch (int e) {
		printf("\n{\"error\": %d}\n", e);
		return 1;
	}
	return 0;
}
