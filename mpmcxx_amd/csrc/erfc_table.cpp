// erfc_table.cpp -- host copy of the pair sweep's erfc table (generated: erfc_table.inc, tools/fit_erfc_table.py) and its device layout.
#include "kernels.h"
#include <cmath>

#define MPMC_ERFTAB_ROWS
namespace {
const double kRows[][6] = {
#include "erfc_table.inc"
};
} // namespace
#undef MPMC_ERFTAB_ROWS

namespace mpmc {

static_assert(sizeof(kRows) / sizeof(kRows[0]) == MPMC_ERFTAB_PIECES, "erfc table: piece count");
static_assert(kErfTableDouble2 == 3 * MPMC_ERFTAB_PIECES, "erfc table: device layout");

void erfc_table_device_layout(double2 *out) {
	for (int k = 0; k < MPMC_ERFTAB_PIECES; k++) {
		out[k] = make_double2(kRows[k][0], kRows[k][1]);
		out[MPMC_ERFTAB_PIECES + k] = make_double2(kRows[k][2], kRows[k][3]);
		out[2 * MPMC_ERFTAB_PIECES + k] = make_double2(kRows[k][4], kRows[k][5]);
	}
}

} // namespace mpmc

// measurement / test entry point (no device needed): erfc(x) and exp(-x^2) evaluated the way the kernel does, in host fp64
extern "C" int mpmc_debug_erfc_table(double x, double *erfc_out, double *gauss_out) {
	if (!(x >= 0.0) || !(x < MPMC_ERFTAB_XMAX)) return 1;
	const double xs = x * MPMC_ERFTAB_INV_H;
	const int it = (int)xs;
	const double dd = (xs - (double)it) - 0.5;
	const double *c = kRows[it];
	const double w = std::fma(std::fma(std::fma(std::fma(c[4], dd, c[3]), dd, c[2]), dd, c[1]), dd, c[0]);
	const double d = dd * (1.0 / MPMC_ERFTAB_INV_H);
	const double t = -(d * std::fma(2.0, x, -d));
	const double e[MPMC_ERFTAB_EXP_DEG + 1] = {MPMC_ERFTAB_EXP_COEFFS};
	double p = e[MPMC_ERFTAB_EXP_DEG];
	for (int k = MPMC_ERFTAB_EXP_DEG - 1; k >= 0; --k) p = std::fma(p, t, e[k]);
	const double G = c[5] * p;
	if (gauss_out) *gauss_out = G;
	if (erfc_out) *erfc_out = G * w;
	return 0;
}
