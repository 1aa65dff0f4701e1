// examples/gibbs_nvt.cpp -- an `ensemble nvt_gibbs` input of the reference run end to end on the HIP path.
//   gibbs_nvt INPUT.in [STEPS]
// Box 0 = pqr_input on device 0, box 1 = pqr_input_B (default: the same file) on device 1 when the node shows two GPUs.  Prints the
// trajectory as one JSON object: per step the move types, the two trial energies, the Boltzmann factors, what was accepted, N and
// volume of both boxes; then the final geometries (include/mpmc_gibbs_run.hpp).
#include <cstdio>
#include <cstdlib>

#include "mpmc_gibbs_run.hpp"

int main(int argc, char **argv) {
	if (argc < 2) {
		std::fprintf(stderr, "usage: %s INPUT.in [STEPS]\n", argv[0]);
		return 2;
	}
	try {
		const mpmc::GibbsSettings cfg = mpmc::read_gibbs_settings(argv[1]);
		mpmc::System a, b;
		const std::string pqr_a = mpmc::read_input(argv[1], a);
		(void)mpmc::read_input(argv[1], b);
		std::string pqr_b = cfg.pqr_input_B.empty() ? pqr_a : cfg.pqr_input_B;
		if (pqr_b[0] != '/') pqr_b = mpmc::io_detail::dirname_of(argv[1]) + "/" + pqr_b;
		mpmc::read_pqr(pqr_a, a);
		mpmc::read_pqr(pqr_b, b);
		a.update_pbc();
		b.update_pbc();
		a.eager_dipoles = b.eager_dipoles = false;
		int ndev = 0;
		if (mpmc_device_count(&ndev) != MPMC_OK || ndev < 1) throw (int)MPMC_ERR_NO_DEVICE;
		a.device = 0;
		b.device = 1 % ndev;
		mpmc::run_gibbs_and_print(a, b, cfg, argc > 2 ? std::atoi(argv[2]) : -1);
	} catch (int code) {
		std::printf("{\"error\": %d}\n", code);
		return 1;
	}
	return 0;
}
