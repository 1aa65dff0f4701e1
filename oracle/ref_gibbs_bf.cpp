// oracle/ref_gibbs_bf.cpp -- TEST INFRASTRUCTURE, never part of the product path.
//
// Calls the reference's own SimulationControl::boltzmann_factor_NVT_Gibbs (src/SimulationControl.Gibbs.cpp:358) -- object code
// compiled in place from /root/reference/src by oracle/Makefile -- on two bare System objects whose observables / checkpoint /
// nodestats this driver fills from its input.  The reference's Gibbs_mc loop itself cannot run in this image, this
// one function can: it pins mpmc_gibbs_boltzmann_factor (tests/golden/gibbs_bf.json, generator oracle/make_gibbs_golden.py).
// The function is a private static member: this driver is compiled with -fno-access-control.  No reference source text lives here.
//
// stdin, one case per line:  moveA moveB T initA finalA initB finalB NA VA NB VB ckptVolA
//   ("nan" / "inf" allowed for energies); stdout, one line per case:  bfA bfB energyA energyB status   (%.17g; status 0 ok, else the thrown int)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "SimulationControl.h"
#include "System.h"
#include "constants.h"

int rank = 0;
int size = 0;
bool mpi = false;

template <typename T>
static T *zeroed() {
	return static_cast<T *>(calloc(1, sizeof(T)));
}

int main() {
	System a, b;
	System *s[2] = {&a, &b};
	for (int i = 0; i < 2; i++) {
		s[i]->observables = zeroed<System::observables_t>();
		s[i]->nodestats = zeroed<System::nodestats_t>();
		s[i]->checkpoint = zeroed<System::checkpoint_t>();
		s[i]->checkpoint->observables = zeroed<System::observables_t>();
	}
	char line[1024];
	while (fgets(line, sizeof line, stdin)) {
		int mv[2];
		double T, e[4], N[2], V[2], ck;
		char es[4][64];
		if (sscanf(line, "%d %d %lf %63s %63s %63s %63s %lf %lf %lf %lf %lf", &mv[0], &mv[1], &T, es[0], es[1], es[2], es[3], &N[0], &V[0], &N[1], &V[1], &ck) != 12)
			continue;
		for (int k = 0; k < 4; k++) e[k] = strtod(es[k], nullptr); // strtod reads nan / inf
		for (int i = 0; i < 2; i++) {
			s[i]->temperature = T;
			s[i]->checkpoint->movetype = mv[i];
			s[i]->observables->N = N[i];
			s[i]->observables->volume = V[i];
			s[i]->observables->energy = e[2 * i + 1];
			s[i]->nodestats->boltzmann_factor = -1.0; // sentinel: "left untouched"
		}
		a.checkpoint->observables->volume = ck;
		int status = 0;
		try {
			SimulationControl::boltzmann_factor_NVT_Gibbs(a, e[0], e[1], b, e[2], e[3]);
		} catch (int code) {
			status = code;
		}
		printf("%.17g %.17g %.17g %.17g %d\n", a.nodestats->boltzmann_factor, b.nodestats->boltzmann_factor, a.observables->energy, b.observables->energy, status);
	}
	return 0;
}
