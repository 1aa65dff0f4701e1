// examples/pimc_nvt.cpp -- path-integral NVT Monte Carlo of a reference input file on the HIP energy path.
//   pimc_nvt INPUT.in -P 8 [-o OUTDIR] [--steps N] [--trial]
// Reads the reference's own input / PQR formats (include/mpmc_io.hpp), runs the driver of include/mpmc_pimc.hpp with one device
// context per image (images are spread round-robin over the visible GPUs) and writes OUTDIR/JOB.energy.dat (same rows as the
// reference's energy output) and OUTDIR/JOB.final-%04d.pqr; prints one JSON line with acceptance rates and throughput.
// With `parallel_restarts on` the images start from JOB.restart-%04d.pqr next to the input file, as in the reference.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "mpmc_pimc.hpp"

int main(int argc, char **argv) {
	if (argc < 4) {
		std::fprintf(stderr, "usage: %s INPUT.in -P <images> [-o OUTDIR] [--steps N]\n", argv[0]);
		return 2;
	}
	int P = 0;
	long steps_override = -1;
	bool trial = false; // per-move delta energies instead of full evaluations
	std::string outdir = ".";
	for (int k = 2; k < argc; k++) {
		if (!std::strcmp(argv[k], "-P") && k + 1 < argc) P = std::atoi(argv[++k]);
		else if (!std::strcmp(argv[k], "-o") && k + 1 < argc) outdir = argv[++k];
		else if (!std::strcmp(argv[k], "--steps") && k + 1 < argc) steps_override = std::atol(argv[++k]);
		else if (!std::strcmp(argv[k], "--trial")) trial = true;
	}
	try {
		mpmc::PathIntegralNVT<mpmc::System> mc;
		mc.cfg = mpmc::read_pimc_settings(argv[1]);
		if (steps_override > 0) mc.cfg.numsteps = (unsigned int)steps_override;
		int ndev = 1;
		if (mpmc_device_count(&ndev) != MPMC_OK || ndev < 1) {
			std::fprintf(stderr, "no HIP device: %s\n", mpmc_last_error(nullptr));
			return 1;
		}
		std::vector<std::unique_ptr<mpmc::System>> beads;
		const std::string dir = mpmc::io_detail::dirname_of(argv[1]);
		for (int b = 0; b < P; b++) {
			beads.emplace_back(new mpmc::System());
			mpmc::System &s = *beads.back();
			s.device = b % ndev; // image -> device round robin
			s.eager_dipoles = false; // nothing here reads the per-atom dipoles
			if (mc.cfg.parallel_restarts) { // one geometry per image: JOB.restart-%04d.pqr (src/SimulationControl.PathIntegral.cpp:619-621)
				char name[64];
				std::snprintf(name, sizeof name, ".restart-%04d.pqr", b);
				mpmc::read_input(argv[1], s);
				mpmc::read_pqr(dir + "/" + mc.cfg.job_name + name, s);
				s.update_pbc();
			} else {
				mpmc::load_system(argv[1], s);
			}
			mc.systems.push_back(&s);
		}
		mc.moltype_names = beads[0]->moltype_names;
		mc.init();
		mc.use_trial_moves = trial;
		const std::string base = outdir + "/" + mc.cfg.job_name;
		FILE *fp = std::fopen((base + ".energy.dat").c_str(), "w");
		if (!fp) throw 1001; // fopen_fail_write
		const auto t0 = std::chrono::steady_clock::now();
		mc.run(fp);
		const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		std::fclose(fp);
		for (int b = 0; b < P; b++) {
			char name[64];
			std::snprintf(name, sizeof name, ".final-%04d.pqr", b);
			mpmc::write_pqr(base + name, *beads[b]);
		}
		const mpmc::observables_t &o = mc.pi.sys_observables;
		long long waits[4] = {0, 0, 0, 0}; // how the host waits of all images ended (polls seen / timed out, stream syncs, yields)
		for (int b = 0; b < P; b++) {
			long long w[4] = {0, 0, 0, 0};
			if (mpmc_debug_wait_counters(beads[b]->context(), w) == 0)
				for (int k = 0; k < 4; k++) waits[k] += w[k];
		}
		std::printf("{\"wait_polls_seen\": %lld, \"wait_polls_timed_out\": %lld, \"wait_stream_syncs\": %lld, \"wait_poll_yields\": %lld, ", waits[0], waits[1],
		            waits[2], waits[3]);
		std::printf("\"P\": %d, \"natoms\": %d, \"steps\": %u, \"AR\": %.5f, \"AR_displace\": %.5f, \"AR_bead\": %.5f, \"energy\": %.17g, \"kinetic\": %.17g, "
		            "\"seconds\": %.3f, \"steps_per_s\": %.2f, \"energy_evals_per_s\": %.1f, \"devices\": %d, \"trial_moves\": %d}\n",
		            P, (int)beads[0]->atoms.size(), mc.step, mc.acceptance_rate(),
		            (mc.accept_displace + mc.reject_displace) ? (double)mc.accept_displace / (double)(mc.accept_displace + mc.reject_displace) : 0.0,
		            (mc.accept_bead + mc.reject_bead) ? (double)mc.accept_bead / (double)(mc.accept_bead + mc.reject_bead) : 0.0, o.energy, o.kinetic_energy,
		            sec, mc.cfg.numsteps / sec, mc.energy_calls / sec, ndev < P ? ndev : P, trial ? 1 : 0);
	} catch (int code) {
		std::printf("{\"error\": %d}\n", code);
		return 1;
	}
	return 0;
}
