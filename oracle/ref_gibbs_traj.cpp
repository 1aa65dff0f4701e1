// oracle/ref_gibbs_traj.cpp -- TEST INFRASTRUCTURE, never part of the product path.
//
// A Gibbs-ensemble (nvt_gibbs) Monte Carlo trajectory made by the REFERENCE's own functions.  The stock binary cannot run this
// ensemble in a build without MPI (its loop sizes buffers by the MPI world size and writes through null statistics pointers,
// DESIGN.md 8.4); what fails there is bookkeeping, not the moves.  This driver repeats the skeleton of SimulationControl::Gibbs_mc
// (src/SimulationControl.Gibbs.cpp:133-330: which function is called when) and calls the reference's object code for everything
// that decides the trajectory: System::pick_Gibbs_move, System::make_move_Gibbs (displace / volume_change_Gibbs / transfer),
// System::energy, SimulationControl::boltzmann_factor_NVT_Gibbs, Rando::rand, System::restore, register_accept / register_reject,
// backup_observables.  Skipped: setup_mpi, averages, output files.  Compiled with -fno-access-control (private members).
// No reference source text lives in this file.
//
// usage: ref_gibbs_traj INPUT.in STEPS       one JSON object on the last line of stdout
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "Atom.h"
#include "Molecule.h"
#include "Rando.h"
#include "SimulationControl.h"
#include "System.h"
#include "constants.h"

int rank = 0;
int size = 0;
bool mpi = false;

static void dump_box(System *s, const char *key) {
	printf("\"%s\": {\"basis\": [", key);
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) printf("%.17g%s", s->pbc.basis[i][j], (i == 2 && j == 2) ? "" : ", ");
	printf("], \"mol_id\": [");
	bool first = true;
	int m = 0;
	for (Molecule *mol = s->molecules; mol; mol = mol->next, m++)
		for (Atom *a = mol->atoms; a; a = a->next) {
			printf("%s%d", first ? "" : ", ", m);
			first = false;
		}
	printf("], \"charge\": [");
	first = true;
	for (Molecule *mol = s->molecules; mol; mol = mol->next)
		for (Atom *a = mol->atoms; a; a = a->next) {
			printf("%s%.17g", first ? "" : ", ", a->charge);
			first = false;
		}
	printf("], \"pos\": [");
	first = true;
	for (Molecule *mol = s->molecules; mol; mol = mol->next)
		for (Atom *a = mol->atoms; a; a = a->next) {
			printf("%s%.17g, %.17g, %.17g", first ? "" : ", ", a->pos[0], a->pos[1], a->pos[2]);
			first = false;
		}
	printf("]}");
}

int main(int argc, char **argv) {
	if (argc < 3) {
		fprintf(stderr, "usage: %s INPUT.in STEPS\n", argv[0]);
		return 2;
	}
	const int steps = atoi(argv[2]);
	try {
		SimulationControl sc(argv[1], 0, false, nullptr);
		sc.initializeSimulationObjects();
		std::vector<System *> &systems = sc.systems;
		if (systems.size() != 2) {
			printf("\n{\"error\": \"not a two-system input\"}\n");
			return 1;
		}
		double initial_energy[2], final_energy[2];
		for (int i = 0; i < 2; i++) {
			systems[i]->observables->volume = systems[i]->pbc.volume;
			initial_energy[i] = systems[i]->mc_initial_energy();
		}
		// (Gibbs_mc calls backup_observables_ALL_SYSTEMS() here, whose first statement writes through the never-initialised checkpoint of
		// the TEMPLATE system `sc.sys` -- one of the reasons the stock loop dies; the per-box half is what the trajectory depends on)
		sc.backup_observables_SYS_VECTOR();
		int move = System::pick_Gibbs_move(systems);
		printf("\n{\"initial_energy\": [%.17g, %.17g], \"N\": [%.17g, %.17g], \"volume\": [%.17g, %.17g], \"volume_probability\": %.17g,\n \"steps\": [",
		       initial_energy[0], initial_energy[1], systems[0]->observables->N, systems[1]->observables->N, systems[0]->pbc.volume, systems[1]->pbc.volume,
		       systems[0]->volume_probability);
		for (int s = 1; s <= steps; s++) {
			systems[0]->step = systems[1]->step = s;
			initial_energy[0] = systems[0]->observables->energy;
			initial_energy[1] = systems[1]->observables->energy;
			const int mv[2] = {systems[0]->checkpoint->movetype, systems[1]->checkpoint->movetype};
			System::make_move_Gibbs(systems);
			// polar() lets the Thole matrices follow N only in the uVT / replay ensembles (src/System.Energy.cpp:2544-2545): a polarizable
			// Gibbs box overruns them at its first particle transfer.  The resize is allocation only -- the same call, made here.
			for (int i = 0; i < 2; i++)
				if (systems[i]->polarization && !systems[i]->polar_zodid) systems[i]->thole_resize_matrices();
			final_energy[0] = systems[0]->energy();
			final_energy[1] = systems[1]->energy();
			SimulationControl::boltzmann_factor_NVT_Gibbs(*systems[0], initial_energy[0], final_energy[0], *systems[1], initial_energy[1], final_energy[1]);
			const double bf[2] = {systems[0]->nodestats->boltzmann_factor, systems[1]->nodestats->boltzmann_factor};
			int accepted[2] = {0, 0};
			if (move == MOVETYPE_DISPLACE || move == MOVETYPE_SPINFLIP) {
				for (int i = 0; i < 2; i++) {
					if ((Rando::rand() < systems[i]->nodestats->boltzmann_factor) && !systems[i]->iterator_failed) {
						accepted[i] = 1;
						systems[i]->register_accept();
					} else {
						systems[i]->iterator_failed = 0;
						systems[i]->restore();
						systems[i]->register_reject();
					}
				}
			} else {
				const double b = systems[0]->nodestats->boltzmann_factor;
				if ((Rando::rand() < b) && !systems[0]->iterator_failed && !systems[1]->iterator_failed) {
					for (int i = 0; i < 2; i++) {
						accepted[i] = 1;
						*systems[i]->checkpoint->observables = *systems[i]->observables;
						systems[i]->register_accept();
					}
				} else {
					for (int i = 0; i < 2; i++) {
						systems[i]->iterator_failed = 0;
						systems[i]->restore();
						systems[i]->register_reject();
					}
				}
			}
			printf("%s\n  {\"step\": %d, \"movetype\": [%d, %d], \"final_energy\": [%.17g, %.17g], \"boltzmann_factor\": [%.17g, %.17g], \"accepted\": [%d, %d], "
			       "\"energy\": [%.17g, %.17g], \"N\": [%.17g, %.17g], \"volume\": [%.17g, %.17g], \"natoms\": [%d, %d]}",
			       s > 1 ? "," : "", s, mv[0], mv[1], final_energy[0], final_energy[1], bf[0], bf[1], accepted[0], accepted[1], systems[0]->observables->energy,
			       systems[1]->observables->energy, systems[0]->observables->N, systems[1]->observables->N, systems[0]->pbc.volume, systems[1]->pbc.volume,
			       systems[0]->countNatoms(), systems[1]->countNatoms());
			sc.backup_observables_SYS_VECTOR();
			move = System::pick_Gibbs_move(systems);
		}
		printf("],\n ");
		dump_box(systems[0], "final_box_0");
		printf(",\n ");
		dump_box(systems[1], "final_box_1");
		printf("}\n");
		fflush(stdout);
	} catch (int e) {
		printf("\n{\"error\": %d}\n", e);
		return 1;
	}
	return 0;
}
