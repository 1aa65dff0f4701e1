// oracle/ref_adapter.cpp -- DROP-IN PROOF (test infrastructure; built only into oracle/_ref/, never shipped).
//
// This is the reference-side binding a maintainer of b-tudor/mpmcxx would add: it
// replaces `double System::energy()` (reference src/System.Energy.cpp:19) by a call into the C ABI of
// libmpmc_energy.so, WITHOUT touching any reference source: the reference's own objects are linked with
//      -Wl,--wrap=_ZN6System6energyEv
// so that every caller (System::mc, mc_initial_energy, SimulationControl::PI_calculate_potential, Gibbs_mc)
// reaches __wrap__ZN6System6energyEv below.  The stock Monte Carlo loop then runs unchanged on top of the
// MI355X energy path.
//
// What the adapter does per call (everything else the reference's energy() does is host-side bookkeeping that is
// delegated to the reference's own public methods):
//   1. flatten the Molecule -> Atom linked lists into the struct-of-arrays the C ABI takes  (O(N))
//   2. mpmc_set_box / mpmc_set_options / mpmc_set_atoms on the per-System context, mpmc_energy()
//   3. copy mpmc_result into System::observables / nodestats / iterator_failed, and mu / ef_static / ef_induced
//      back into the Atom objects
//   4. update_com() + wrap_all() (the tail of pairs()), countN(), NU, last_volume -- as energy() does.
//
// MPMC_WRAP_MODE=gpu (default) | both (also run the original and abort if any component differs by > 1e-9 rel)
//               | passthrough (original only).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "Atom.h"
#include "Molecule.h"
#include "Pair.h"
#include "System.h"
#include "constants.h"

#include "../include/mpmc_energy.h"

extern "C" double __real__ZN6System6energyEv(System *self);

namespace {
struct Binding {
	mpmc_ctx *ctx = nullptr;
	int capacity = 0;
};
// what was last handed to mpmc_set_atoms for a System: while the atom list itself is unchanged (the normal case between two Monte
// Carlo moves) only the positions travel, through mpmc_update_positions
struct AtomsSeen {
	std::vector<double> q, alpha, eps, sig, mass;
	std::vector<int32_t> mol, frozen, disp;
};
std::unordered_map<System *, AtomsSeen> g_seen;
std::mutex g_mu;
std::unordered_map<System *, Binding> g_ctx;
long g_calls = 0;

int device_for(System *s) {
	// bead / box -> device round robin; single-GPU boxes map everything to device 0
	int ndev = 1;
	mpmc_device_count(&ndev);
	if (ndev < 1) ndev = 1;
	std::lock_guard<std::mutex> lk(g_mu);
	return (int)(g_ctx.size() % (size_t)ndev);
}

uint64_t unsupported_mask(System *s) {
	uint64_t m = 0;
	if (s->rd_crystal) m |= MPMC_FLAG_RD_CRYSTAL;
	if (s->spectre) m |= MPMC_FLAG_SPECTRE;
	if (s->gwp) m |= MPMC_FLAG_GWP;
	if (s->use_sg) m |= MPMC_FLAG_USE_SG;
	if (s->polarvdw) m |= MPMC_FLAG_POLARVDW;
	if (s->polar_ewald_full) m |= MPMC_FLAG_POLAR_EWALD_FULL;
	if (s->polar_wolf || s->polar_wolf_full) m |= MPMC_FLAG_POLAR_WOLF;
	if (s->polar_palmo) m |= MPMC_FLAG_POLAR_PALMO;
	if (s->polar_gs_ranked) m |= MPMC_FLAG_POLAR_GS_RANKED;
	if (s->polar_sor || s->polar_esor) m |= MPMC_FLAG_POLAR_SOR;
	if (s->polar_zodid) m |= MPMC_FLAG_POLAR_ZODID;
	if (s->waldmanhagler || s->halgren_mixing || s->c6_mixing || s->cdvdw_9th_repulsion || s->cdvdw_sig_repulsion || s->cdvdw_exp_repulsion)
		m |= MPMC_FLAG_NON_LB_MIXING;
	if (s->rd_anharmonic || s->use_dreiding || s->using_lj_buffered_14_7 || s->using_disp_expansion) m |= MPMC_FLAG_OTHER_RD;
	if (s->using_axilrod_teller) m |= MPMC_FLAG_AXILROD_TELLER;
	if (s->cavity_autoreject || s->cavity_autoreject_absolute) m |= MPMC_FLAG_CAVITY_AUTOREJECT;
	if (s->polarization && !s->polar_iterative) m |= MPMC_FLAG_POLAR_MATRIX_INVERSION;
	return m;
}

[[noreturn]] void die(System *s, mpmc_ctx *c, int rc, const char *what) {
	std::fprintf(stderr, "ref_adapter: %s failed (%d): %s\n", what, rc, c ? mpmc_last_error(c) : mpmc_last_error(nullptr));
	throw (rc > 0 ? rc : (int)internal_error); // reference convention: throw <int>, caught in main()
}
} // namespace

extern "C" double __wrap__ZN6System6energyEv(System *s) {
	static const char *mode_env = std::getenv("MPMC_WRAP_MODE");
	const bool passthrough = mode_env && !std::strcmp(mode_env, "passthrough");
	const bool both = mode_env && !std::strcmp(mode_env, "both");
	if (passthrough) return __real__ZN6System6energyEv(s);
	// both: the original runs FIRST, on the state the driver left -- its "first call / volume changed" test (System.Energy.cpp:42:
	// last_volume, observables->energy == 0) and its per-pair caches must see what they would see in the stock binary, not what
	// step 3/4 below write.  Its results are kept, the HIP path is evaluated on the same positions, compared, and the System is
	// left exactly as the original left it.
	double e_ref = 0;
	System::observables_t ref_obs;
	int ref_failed = 0;
	if (both) {
		e_ref = __real__ZN6System6energyEv(s);
		ref_obs = *s->observables;
		ref_failed = s->iterator_failed;
	}

	// ---- 1. flatten (atom_array order = list order, System.cpp:881-904) ---------------------------------
	s->natoms = s->countNatoms();
	const int n = s->natoms;
	std::vector<double> pos(3 * (size_t)n), q(n), alpha(n), eps(n), sig(n), mass(n);
	std::vector<int32_t> mol(n), frozen(n), disp(n);
	std::vector<Atom *> atoms(n);
	{
		int k = 0, m = 0;
		for (Molecule *mp = s->molecules; mp; mp = mp->next, m++)
			for (Atom *a = mp->atoms; a; a = a->next, k++) {
				atoms[k] = a;
				for (int p = 0; p < 3; p++) pos[3 * k + p] = a->pos[p];
				q[k] = a->charge;
				alpha[k] = a->polarizability;
				eps[k] = a->epsilon;
				sig[k] = a->sigma;
				mass[k] = a->mass;
				mol[k] = m;
				frozen[k] = a->frozen;
				disp[k] = (a->c6 != 0.0 || a->c8 != 0.0 || a->c10 != 0.0) ? 1 : 0;
			}
	}

	// ---- 2. context + evaluation --------------------------------------------------------------------------------
	Binding b;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		auto it = g_ctx.find(s);
		if (it != g_ctx.end()) b = it->second;
	}
	bool fresh_ctx = false;
	if (!b.ctx || b.capacity < n) {
		fresh_ctx = true;
		if (b.ctx) mpmc_ctx_destroy(b.ctx);
		b.capacity = n + n / 4 + 64; // uVT head-room
		int rc = mpmc_ctx_create(device_for(s), b.capacity, &b.ctx);
		if (rc != MPMC_OK) die(s, nullptr, rc, "mpmc_ctx_create");
		std::lock_guard<std::mutex> lk(g_mu);
		g_ctx[s] = b;
	}
	mpmc_ctx *c = b.ctx;
	int rc;
	if ((rc = mpmc_set_box(c, &s->pbc.basis[0][0], &s->pbc.reciprocal_basis[0][0], s->pbc.volume, s->pbc.cutoff)) != MPMC_OK)
		die(s, c, rc, "mpmc_set_box");
	mpmc_options o;
	mpmc_default_options(&o);
	o.rd_only = s->rd_only || s->use_sg; // energy() :46
	o.rd_lrc = s->rd_lrc;
	o.polarization = s->polarization;
	o.polar_iterative = s->polar_iterative;
	o.polar_ewald = s->polar_ewald;
	o.polar_max_iter = s->polar_max_iter;
	o.polar_gs = s->polar_gs;
	o.polar_rrms = s->polar_rrms;
	o.damp_type = s->damp_type;
	o.ewald_kmax = s->ewald_kmax;
	o.wolf = s->wolf;
	o.feynman_hibbs = s->feynman_hibbs;
	o.feynman_hibbs_order = s->feynman_hibbs_order;
	o.temperature = s->temperature;
	o.polar_precision = s->polar_precision;
	o.polar_gamma = s->polar_gamma;
	o.polar_damp = s->polar_damp;
	o.ewald_alpha = s->ewald_alpha;             // already resolved by System::update_pbc (System.cpp:871-874)
	o.polar_ewald_alpha = s->polar_ewald_alpha;
	o.unsupported_flags = unsupported_mask(s);
	if (!s->polarization) o.damp_type = MPMC_DAMPING_EXPONENTIAL; // damp_type has no initializer in the reference (System.h:705)
	if ((rc = mpmc_set_options(c, &o)) != MPMC_OK) die(s, c, rc, "mpmc_set_options");
	{
		std::unique_lock<std::mutex> lk(g_mu);
		AtomsSeen &seen = g_seen[s];
		const bool same = !fresh_ctx && seen.q == q && seen.alpha == alpha && seen.eps == eps && seen.sig == sig && seen.mass == mass && seen.mol == mol &&
		                  seen.frozen == frozen && seen.disp == disp;
		if (!same) {
			seen.q = q, seen.alpha = alpha, seen.eps = eps, seen.sig = sig, seen.mass = mass;
			seen.mol = mol, seen.frozen = frozen, seen.disp = disp;
		}
		lk.unlock();
		if (same) {
			if ((rc = mpmc_update_positions(c, 0, n, pos.data())) != MPMC_OK) die(s, c, rc, "mpmc_update_positions");
		} else if ((rc = mpmc_set_atoms(c, n, pos.data(), q.data(), alpha.data(), eps.data(), sig.data(), mol.data(), frozen.data(), disp.data(),
		                                mass.data())) != MPMC_OK) {
			die(s, c, rc, "mpmc_set_atoms");
		}
	}
	mpmc_result r;
	if ((rc = mpmc_energy(c, &r)) != MPMC_OK) die(s, c, rc, "mpmc_energy");

	// ---- 3. write back what energy() leaves behind ------------------------------------------------------------------
	s->observables->coulombic_energy = r.coulombic_energy;
	s->observables->rd_energy = r.rd_energy;
	if (s->polarization && !o.rd_only) {
		s->observables->polarization_energy = r.polarization_energy;
		s->observables->dipole_rrms = r.dipole_rrms;
		s->nodestats->polarization_iterations = (double)r.polar_iterations;
		s->iterator_failed = r.iterator_failed;
		std::vector<double> mu(3 * (size_t)n), e0(3 * (size_t)n), ei(3 * (size_t)n);
		if ((rc = mpmc_get_dipoles(c, mu.data(), e0.data(), ei.data())) != MPMC_OK) die(s, c, rc, "mpmc_get_dipoles");
		for (int k = 0; k < n; k++)
			for (int p = 0; p < 3; p++) {
				atoms[k]->mu[p] = mu[3 * k + p];
				atoms[k]->ef_static[p] = e0[3 * k + p];
				atoms[k]->ef_induced[p] = ei[3 * k + p];
			}
	}
	s->observables->energy = r.energy;

	// ---- 4. the host-side tail of pairs()/energy(), by the reference's own methods -------------------------------
	s->update_com();
	s->wrap_all();
	s->countN();
	s->observables->spin_ratio /= s->observables->N;
	s->observables->NU = s->observables->N * s->observables->energy;
	s->last_volume = s->pbc.volume;

	long call;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		call = ++g_calls;
	}
	if (both) {
		const double e_gpu = r.energy, rd = r.rd_energy, es = r.coulombic_energy, pol = r.polarization_energy;
		auto bad = [](double a, double b) { return std::fabs(a - b) > 1e-9 * std::fabs(b) + 1e-300 && !(a == b); };
		if (bad(e_gpu, e_ref) || bad(rd, ref_obs.rd_energy) || bad(es, ref_obs.coulombic_energy) || (s->polarization && bad(pol, ref_obs.polarization_energy)) ||
		    (s->polarization && r.iterator_failed != ref_failed)) {
			std::fprintf(stderr, "ref_adapter: MISMATCH at call %ld: gpu %.17g ref %.17g (rd %.17g/%.17g es %.17g/%.17g pol %.17g/%.17g)\n", call,
			             e_gpu, e_ref, rd, ref_obs.rd_energy, es, ref_obs.coulombic_energy, pol, ref_obs.polarization_energy);
			{ // diagnostics: is the difference a property of the context's state or of that one evaluation?
				mpmc_result again, fresh;
				int rc2 = mpmc_energy(c, &again);
				std::fprintf(stderr, "ref_adapter:   same context, evaluated again (rc %d): pol %.17g iterations %d (first: %d) fresh_ctx %d\n", rc2,
				             again.polarization_energy, again.polar_iterations, r.polar_iterations, (int)fresh_ctx);
				mpmc_ctx *f = nullptr;
				if (mpmc_ctx_create(device_for(s), n, &f) == MPMC_OK) {
					mpmc_set_box(f, &s->pbc.basis[0][0], &s->pbc.reciprocal_basis[0][0], s->pbc.volume, s->pbc.cutoff);
					mpmc_set_options(f, &o);
					mpmc_set_atoms(f, n, pos.data(), q.data(), alpha.data(), eps.data(), sig.data(), mol.data(), frozen.data(), disp.data(), mass.data());
					rc2 = mpmc_energy(f, &fresh);
					std::fprintf(stderr, "ref_adapter:   fresh context (rc %d): pol %.17g rd %.17g es %.17g\n", rc2, fresh.polarization_energy, fresh.rd_energy,
					             fresh.coulombic_energy);
					mpmc_ctx_destroy(f);
				}
			}
			std::abort();
		}
		*s->observables = ref_obs;
		s->iterator_failed = ref_failed;
		return e_ref;
	}
	return r.energy;
}

// summary line at exit so that a test can see how many calls were intercepted
namespace {
struct Reporter {
	~Reporter() {
		if (g_calls) std::fprintf(stderr, "ref_adapter: %ld System::energy() calls served by libmpmc_energy.so\n", g_calls);
		for (auto &kv : g_ctx)
			if (kv.second.ctx) mpmc_ctx_destroy(kv.second.ctx);
	}
} g_reporter;
} // namespace
