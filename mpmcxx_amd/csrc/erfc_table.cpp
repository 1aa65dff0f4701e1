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
// (one Horner pass for the interpolant p and its derivative; exp(-x^2) = -(sqrt(pi) / 2) erfc'(x))
extern "C" int mpmc_debug_erfc_table(double x, double *erfc_out, double *gauss_out) {
	if (!(x >= 0.0) || !(x < MPMC_ERFTAB_XMAX)) return 1;
	const double xs = x * MPMC_ERFTAB_INV_H;
	const int it = (int)xs;
	const double dd = xs - (double)it;
	const double *c = kRows[it];
	double b = c[5], d1 = c[5];
	for (int k = 4; k >= 1; --k) {
		b = std::fma(b, dd, c[k]);
		d1 = std::fma(d1, dd, b);
	}
	const double p = std::fma(b, dd, c[0]);
	if (gauss_out) *gauss_out = (-0.5 * 1.7724538509055160273 * MPMC_ERFTAB_INV_H) * d1;
	if (erfc_out) *erfc_out = p;
	return 0;
}

// ... and the field factor of real_term (src/System.Energy.cpp:2919-2934), erfc(x) + 2 x / sqrt(pi) exp(-x^2) = p - (x / H) p'
extern "C" int mpmc_debug_erfc_table_field(double x, double *factor_out) {
	if (!(x >= 0.0) || !(x < MPMC_ERFTAB_XMAX) || !factor_out) return 1;
	const double xs = x * MPMC_ERFTAB_INV_H;
	const int it = (int)xs;
	const double dd = xs - (double)it;
	const double *c = kRows[it];
	double b = c[5], d1 = c[5];
	for (int k = 4; k >= 1; --k) {
		b = std::fma(b, dd, c[k]);
		d1 = std::fma(d1, dd, b);
	}
	*factor_out = std::fma(-xs, d1, std::fma(b, dd, c[0]));
	return 0;
}
