/* oracle/mpmc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see mpmc_oracle.h).
 *
 * A from-scratch restatement, on flat struct-of-arrays inputs, of the arithmetic the reference
 * performs on its linked lists.  Loop nests run in the reference's pair order (i ascending,
 * j = i+1.. ascending == atom_array[i]->pairs list order, System.cpp:967-991) and every
 * expression keeps the reference's association order, so that with -ffp-contract=off this code
 * rounds the way the reference's x86-64 build does.  All citations are file:line under
 * /root/reference/src.
 */
#include "mpmc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static const double ORC_PI = 3.141592653589793238462643383279502884L; /* constants.h:13 */
static const double ORC_ONE_OVER_SQRT_PI = 0.5641895835477562869480794515607725858440506293289988; /* constants.h:48 */
static const double ORC_MAXVALUE = 1.0e40;   /* constants.h:53 */
static const double ORC_SMALL_DR = 1.0e-12;  /* constants.h:54 */
static const double ORC_MAX_ITER = 128;      /* constants.h:52 */
static const double ORC_DEBYE2SKA = 85.10597636; /* constants.h:41 */
/* Feynman-Hibbs constants, constants.h:17-37 */
static const double ORC_HBAR2 = 1.11211999e-68, ORC_HBAR4 = 1.23681087e-136, ORC_KB = 1.3806503e-23, ORC_KB2 = 1.90619525e-46;
static const double ORC_M2A2 = 1.0e20, ORC_M2A4 = 1.0e40, ORC_AMU2KG = 1.66053873e-27;

/* mass of the molecule that owns atom i (molecules are contiguous runs; Molecule::mass accumulates atom masses, System.cpp:687) */
static double molecule_mass(const orc_system *s, int i) {
	double m = 0;
	int a = i, b = i;
	while (a > 0 && s->mol_id[a - 1] == s->mol_id[i]) a--;
	while (b + 1 < s->n && s->mol_id[b + 1] == s->mol_id[i]) b++;
	for (int k = a; k <= b; k++) m += s->mass[k];
	return m;
}
static double reduced_mass(const orc_system *s, int i, int j) {
	double mi = molecule_mass(s, i), mj = molecule_mass(s, j);
	return ORC_AMU2KG * mi * mj / (mi + mj);
}
/* lj_fh_corr, System.Energy.cpp:1100-1148 */
static double lj_fh_corr(const orc_system *s, int i, int j, double epsilon, double rimg, double term12, double term6) {
	double ir = 1.0 / rimg, ir2 = ir * ir, ir3 = ir2 * ir, ir4 = ir3 * ir;
	double mu = reduced_mass(s, i, j), T = s->temperature;
	double dE = -24.0 * epsilon * (2.0 * term12 - term6) * ir;
	double d2E = 24.0 * epsilon * (26.0 * term12 - 7.0 * term6) * ir2;
	double corr = ORC_M2A2 * (ORC_HBAR2 / (24.0 * ORC_KB * T * mu)) * (d2E + 2.0 * dE / rimg);
	if (s->feynman_hibbs_order >= 4) {
		double d3E = -1344.0 * epsilon * (6.0 * term12 - term6) * ir3;
		double d4E = 12096.0 * epsilon * (10.0 * term12 - term6) * ir4;
		corr += ORC_M2A4 * (ORC_HBAR4 / (1152.0 * ORC_KB2 * T * T * mu * mu)) * (15.0 * dE * ir3 + 4.0 * d3E * ir + d4E);
	}
	return corr;
}
/* coulombic_real_FH, System.Energy.cpp:1521-1557 */
static double coulombic_real_fh(const orc_system *s, int i, int j, double r, double gaussian_term, double erfc_term) {
	double rr = r * r, ir = 1.0 / r, ir2 = ir * ir, ir3 = ir * ir2, ir4 = ir2 * ir2;
	double alpha = s->ewald_alpha, a2 = alpha * alpha, a3 = a2 * alpha, a4 = a3 * alpha;
	double mu = reduced_mass(s, i, j), T = s->temperature;
	double du = -2.0 * alpha * gaussian_term / (r * sqrt(ORC_PI)) - erfc_term * ir2;
	double d2u = (4.0 / sqrt(ORC_PI)) * gaussian_term * (a3 + 1.0 * ir2) + 2.0 * erfc_term * ir3;
	double fh2 = ORC_M2A2 * (ORC_HBAR2 / (24.0 * ORC_KB * T * mu)) * (d2u + 2.0 * du / r);
	double fh4 = 0.0;
	if (s->feynman_hibbs_order >= 4) {
		double d3u = (gaussian_term / sqrt(ORC_PI)) * (-8.0 * (a3 * a2) * r - 8.0 * a3 / r - 12.0 * alpha * ir3) - 6.0 * erfc(alpha * r) * ir4;
		double d4u = (gaussian_term / sqrt(ORC_PI)) * (8.0 * a3 * a2 + 16.0 * a3 * a4 * rr + 32.0 * a3 * ir2 + 48.0 * ir4) + 24.0 * erfc_term * (ir4 * ir);
		fh4 = ORC_M2A4 * (ORC_HBAR4 / (1152.0 * (ORC_KB * ORC_KB * T * T * mu * mu))) * (15.0 * du * ir3 + 4.0 * d3u / r + d4u);
	}
	return fh2 + fh4;
}

/* ---------------------------------------------------------------------------------------------
 * PeriodicBoundary::compute_volume :71-79, compute_cutoff :40-66, compute_reciprocal :83-101
 * ------------------------------------------------------------------------------------------- */
void orc_pbc_update(const double b[9], double R[9], double *volume, double *cutoff) {
#define B(i, j) b[3 * (i) + (j)]
	double vol;
	vol = B(0, 0) * (B(1, 1) * B(2, 2) - B(1, 2) * B(2, 1));
	vol += B(0, 1) * (B(1, 2) * B(2, 0) - B(1, 0) * B(2, 2));
	vol += B(0, 2) * (B(1, 0) * B(2, 1) - B(1, 1) * B(2, 0));
	*volume = vol;

	if (vol <= 0) {
		*cutoff = ORC_MAXVALUE; /* compute_cutoff returns MAXVALUE without touching the member; flagged invalid upstream */
	} else {
		double short_mag = ORC_MAXVALUE;
		for (int i = -15; i <= 15; i++)
			for (int j = -15; j <= 15; j++)
				for (int k = -15; k <= 15; k++) {
					if (i == 0 && j == 0 && k == 0) continue;
					double v[3];
					for (int p = 0; p < 3; p++) v[p] = i * B(0, p) + j * B(1, p) + k * B(2, p);
					double mag = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
					if (mag < short_mag) short_mag = mag;
				}
		*cutoff = 0.5 * short_mag;
	}

	double iv = 1.0 / vol;
	R[0] = iv * (B(1, 1) * B(2, 2) - B(1, 2) * B(2, 1));
	R[1] = iv * (B(0, 2) * B(2, 1) - B(0, 1) * B(2, 2));
	R[2] = iv * (B(0, 1) * B(1, 2) - B(0, 2) * B(1, 1));
	R[3] = iv * (B(1, 2) * B(2, 0) - B(1, 0) * B(2, 2));
	R[4] = iv * (B(0, 0) * B(2, 2) - B(0, 2) * B(2, 0));
	R[5] = iv * (B(0, 2) * B(1, 0) - B(0, 0) * B(1, 2));
	R[6] = iv * (B(1, 0) * B(2, 1) - B(1, 1) * B(2, 0));
	R[7] = iv * (B(0, 1) * B(2, 0) - B(0, 0) * B(2, 1));
	R[8] = iv * (B(0, 0) * B(1, 1) - B(0, 1) * B(1, 0));
#undef B
}

/* ---------------------------------------------------------------------------------------------
 * minimum_image, System.cpp:1202-1279 (stateless: the d_prev change detection only decides
 * WHETHER the same values are recomputed, SURVEY §8a note 5)
 * ------------------------------------------------------------------------------------------- */
double orc_minimum_image(const orc_system *s, int i, int j, double dimg[3], double *r_out) {
	double d[3], img[3], di[3];
	for (int p = 0; p < 3; p++) d[p] = s->pos[3 * i + p] - s->pos[3 * j + p];
	for (int p = 0; p < 3; p++) {
		img[p] = 0;
		for (int q = 0; q < 3; q++) img[p] += s->recip[3 * q + p] * d[q]; /* :1231 */
		img[p] = rint(img[p]);
	}
	for (int p = 0; p < 3; p++) {
		di[p] = 0;
		for (int q = 0; q < 3; q++) di[p] += s->basis[3 * q + p] * img[q]; /* :1241 */
	}
	for (int p = 0; p < 3; p++) di[p] = d[p] - di[p];
	double r2 = 0, ri2 = 0;
	for (int p = 0; p < 3; p++) {
		r2 += d[p] * d[p];
		ri2 += di[p] * di[p];
	}
	double r = sqrt(r2), ri = sqrt(ri2);
	if (r_out) *r_out = r;
	if (isnan(ri)) { /* :1265 */
		for (int p = 0; p < 3; p++) dimg[p] = d[p];
		return r;
	}
	for (int p = 0; p < 3; p++) dimg[p] = di[p];
	return ri;
}

/* pair_exclusions, System.cpp:1035-1197 (Lorentz-Berthelot branch :1166-1177 only) */
typedef struct {
	int rd_excluded, es_excluded, frozen, attractive_only, intra;
	double sigma, epsilon;
} pair_par;

static void pair_params(const orc_system *s, int i, int j, pair_par *pp) {
	double ei = s->epsilon[i], ej = s->epsilon[j], si = s->sigma[i], sj = s->sigma[j];
	pp->intra = (s->mol_id[i] == s->mol_id[j]);
	if (pp->intra) { /* :1042-1045 */
		pp->rd_excluded = 1;
		pp->es_excluded = 1;
	} else {
		int nodisp = !(s->has_disp && (s->has_disp[i] || s->has_disp[j]));
		pp->rd_excluded = ((ei == 0.0 || si == 0.0 || ej == 0.0 || sj == 0.0) && nodisp) ? 1 : 0; /* :1050-1056 */
		pp->es_excluded = (s->charge[i] == 0.0 || s->charge[j] == 0.0) ? 1 : 0;                    /* :1059-1062 */
	}
	pp->frozen = s->frozen[i] && s->frozen[j]; /* :1067 */
	pp->attractive_only = 0;
	if (si < 0.0 || sj < 0.0) { /* :1167-1169 -- epsilon is NOT assigned on this branch: a fresh Pair keeps 0 (Pair.h:30) */
		pp->attractive_only = 1;
		pp->sigma = 0.5 * (fabs(si) + fabs(sj));
		pp->epsilon = 0.0;
	} else if (si == 0 || sj == 0) {
		pp->sigma = 0;
		pp->epsilon = sqrt(ei * ej);
	} else {
		pp->sigma = 0.5 * (si + sj);
		pp->epsilon = sqrt(ei * ej);
	}
}

/* lj_lrc_corr :1036-1069 / lj_lrc_self :1072-1096 share this expression */
static double lrc_term(double sigma, double epsilon, double cutoff, double volume) {
	double sig_cut = fabs(sigma) / cutoff;
	double sig3 = fabs(sigma);
	sig3 *= sig3 * sig3;
	double sig_cut3 = sig_cut * sig_cut * sig_cut;
	double sig_cut9 = sig_cut3 * sig_cut3 * sig_cut3;
	return ((16.0 / 3.0) * ORC_PI * epsilon * sig3) * ((1.0 / 3.0) * sig_cut9 - sig_cut3) / volume;
}

/* ---------------------------------------------------------------------------------------------
 * lj, System.Energy.cpp:897-1032
 * ------------------------------------------------------------------------------------------- */
double orc_lj(const orc_system *s, orc_result *out) {
	double potential = 0, cutoff = s->cutoff;
	double lj_pairs = 0, lrc_pair = 0, lrc_self = 0;
	long long n_in = 0, n_pairs = 0, n_intra = 0, n_rdx = 0, n_esx = 0, n_frozen = 0;
	for (int i = 0; i < s->n; i++) {
		for (int j = i + 1; j < s->n; j++) {
			pair_par pp;
			double dimg[3], r;
			pair_params(s, i, j, &pp);
			double rimg = 0;
			if (!pp.frozen || s->polarization) rimg = orc_minimum_image(s, i, j, dimg, &r); /* System.cpp:985 */
			n_pairs++;
			n_intra += pp.intra;
			n_rdx += pp.rd_excluded;
			n_esx += pp.es_excluded;
			n_frozen += pp.frozen;

			double lrc = 0, rd = 0;
			if (s->rd_lrc && pp.epsilon != 0 && pp.sigma != 0 && !pp.frozen) /* :1047-1051 */
				lrc = lrc_term(pp.sigma, pp.epsilon, cutoff, s->volume);
			if ((rimg - ORC_SMALL_DR < cutoff) && !pp.rd_excluded && !pp.frozen) { /* :934-937 */
				double sor = fabs(pp.sigma) / rimg; /* :965-968 */
				double sor6 = sor * sor * sor;
				sor6 *= sor6;
				double sor12 = sor6 * sor6;
				double term12 = pp.attractive_only ? 0 : sor12;
				rd += 4.0 * pp.epsilon * (term12 - sor6); /* :993 */
				if (s->feynman_hibbs) rd += lj_fh_corr(s, i, j, pp.epsilon, rimg, term12, sor6); /* :998-999 */
				n_in++;
			}
			potential += rd + lrc; /* :1011 */
			lj_pairs += rd;
			lrc_pair += lrc;
		}
	}
	if (s->rd_lrc) /* :1025-1028 */
		for (int i = 0; i < s->n; i++) {
			double t = 0;
			if (s->sigma[i] != 0 && s->epsilon[i] != 0 && !s->frozen[i]) t = lrc_term(s->sigma[i], s->epsilon[i], cutoff, s->volume);
			potential += t;
			lrc_self += t;
		}
	if (out) {
		out->lj_pairs = lj_pairs;
		out->lrc_pair = lrc_pair;
		out->lrc_self = lrc_self;
		out->n_lj_in_cutoff = n_in;
		out->n_pairs = n_pairs;
		out->n_intra = n_intra;
		out->n_rd_excluded = n_rdx;
		out->n_es_excluded = n_esx;
		out->n_frozen = n_frozen;
	}
	return potential;
}

/* ---------------------------------------------------------------------------------------------
 * lj() with EXACT sums (CPU only; a measuring stick, not a restatement).  The per-pair fp64 terms are the reference's own
 * (same expressions as orc_lj above); what differs is the accumulation: the reference adds every pair's `rd + lrc` onto ONE
 * accumulator in list order (System.Energy.cpp:1011), which at 10 000 atoms means 5*10^7 additions of terms a fraction of an ulp
 * of a -5*10^6 K sum wide -- a rounding drift that grows with N.  Here every sum is carried in long double with Neumaier
 * compensation (error << 1 ulp of the fp64 result), so that "HIP path vs exact" and "reference vs exact" can be told apart.
 * out7 = { sum of rd terms, sum of pair lrc terms, sum of self lrc terms, their total (= what lj() returns, summed exactly),
 *          list-order fp64 `potential` as the reference accumulates it, list-order lj_pairs, list-order lrc_pair }
 * ------------------------------------------------------------------------------------------- */
typedef struct { long double s, c; } orc_ksum;
static void ksum_add(orc_ksum *k, double x) {
	long double t = k->s + (long double)x;
	if (fabsl(k->s) >= fabsl((long double)x)) k->c += (k->s - t) + (long double)x;
	else k->c += ((long double)x - t) + k->s;
	k->s = t;
}
void orc_lj_exact(const orc_system *s, double out7[7]) {
	const double cutoff = s->cutoff;
	orc_ksum k_rd = {0, 0}, k_lrc = {0, 0}, k_self = {0, 0};
	double potential = 0, lj_pairs = 0, lrc_pair = 0;
	for (int i = 0; i < s->n; i++) {
		for (int j = i + 1; j < s->n; j++) {
			pair_par pp;
			double dimg[3], r;
			pair_params(s, i, j, &pp);
			double rimg = 0;
			if (!pp.frozen || s->polarization) rimg = orc_minimum_image(s, i, j, dimg, &r);
			double lrc = 0, rd = 0;
			if (s->rd_lrc && pp.epsilon != 0 && pp.sigma != 0 && !pp.frozen) lrc = lrc_term(pp.sigma, pp.epsilon, cutoff, s->volume);
			if ((rimg - ORC_SMALL_DR < cutoff) && !pp.rd_excluded && !pp.frozen) {
				double sor = fabs(pp.sigma) / rimg;
				double sor6 = sor * sor * sor;
				sor6 *= sor6;
				double sor12 = sor6 * sor6;
				double term12 = pp.attractive_only ? 0 : sor12;
				rd += 4.0 * pp.epsilon * (term12 - sor6);
				if (s->feynman_hibbs) rd += lj_fh_corr(s, i, j, pp.epsilon, rimg, term12, sor6);
			}
			ksum_add(&k_rd, rd);
			ksum_add(&k_lrc, lrc);
			potential += rd + lrc;
			lj_pairs += rd;
			lrc_pair += lrc;
		}
	}
	if (s->rd_lrc)
		for (int i = 0; i < s->n; i++) {
			double t = 0;
			if (s->sigma[i] != 0 && s->epsilon[i] != 0 && !s->frozen[i]) t = lrc_term(s->sigma[i], s->epsilon[i], cutoff, s->volume);
			ksum_add(&k_self, t);
			potential += t;
		}
	const long double rd_x = k_rd.s + k_rd.c, lrc_x = k_lrc.s + k_lrc.c, self_x = k_self.s + k_self.c;
	out7[0] = (double)rd_x;
	out7[1] = (double)lrc_x;
	out7[2] = (double)self_x;
	out7[3] = (double)(rd_x + lrc_x + self_x);
	out7[4] = potential;
	out7[5] = lj_pairs;
	out7[6] = lrc_pair;
}

/* ---------------------------------------------------------------------------------------------
 * coulombic_real, System.Energy.cpp:1466-1517
 * ------------------------------------------------------------------------------------------- */
double orc_coulombic_real(const orc_system *s, orc_result *out) {
	double alpha = s->ewald_alpha, potential = 0;
	long long n_in = 0;
	for (int i = 0; i < s->n; i++) {
		for (int j = i + 1; j < s->n; j++) {
			pair_par pp;
			double dimg[3], rr = 0;
			pair_params(s, i, j, &pp);
			double es_real = 0, es_self_intra = 0;
			if (!pp.frozen) {
				double r = orc_minimum_image(s, i, j, dimg, &rr);
				if (!((r > s->cutoff) || pp.es_excluded)) { /* :1490 */
					double erfc_term = erfc(alpha * r);
					double gaussian_term = exp(-alpha * alpha * r * r);
					es_real = s->charge[i] * s->charge[j] * erfc_term / r; /* :1495 */
					/* the FH term is NOT multiplied by the charges in the reference (:1499-1500 adds the bare derivative expression) */
					if (s->feynman_hibbs) es_real += coulombic_real_fh(s, i, j, r, gaussian_term, erfc_term);
					n_in++;
				} else if (pp.es_excluded) /* :1503-1504, plain (non-image) r */
					es_self_intra = s->charge[i] * s->charge[j] * erf(alpha * rr) / rr;
			}
			potential += es_real - es_self_intra; /* :1510 */
		}
	}
	if (out) out->n_es_in_cutoff = n_in;
	return potential;
}

/* coulombic_wolf, System.Energy.cpp:1420-1462 */
double orc_coulombic_wolf(const orc_system *s) {
	double pot = 0, alpha = s->ewald_alpha, R = s->cutoff, iR = 1.0 / R, erfaRoverR = erf(alpha * R) / R;
	for (int i = 0; i < s->n; i++)
		for (int j = i + 1; j < s->n; j++) {
			pair_par pp;
			double dimg[3], rr;
			pair_params(s, i, j, &pp);
			double es = 0;
			if (!pp.frozen || s->polarization) {
				double r = orc_minimum_image(s, i, j, dimg, &rr);
				double ir = 1.0 / r;
				if (!pp.frozen && !pp.es_excluded && (r < R)) es = s->charge[i] * s->charge[j] * (ir - erfaRoverR - iR * iR * (R - r));
			}
			pot += es;
		}
	return pot;
}

/* hemisphere enumeration shared by coulombic_reciprocal :1577-1583 and recip_term :2849-2854 */
static int next_l(int l[3], int kmax, int *started) {
	/* iterates l0 in [0,kmax], l1 in [(l0?-kmax:0),kmax], l2 in [((l0||l1)?-kmax:1),kmax], skipping |l|^2 > kmax^2 */
	for (;;) {
		if (!*started) {
			l[0] = 0;
			l[1] = 0;
			l[2] = 1;
			*started = 1;
		} else {
			l[2]++;
			if (l[2] > kmax) {
				l[1]++;
				if (l[1] > kmax) {
					l[0]++;
					if (l[0] > kmax) return 0;
					l[1] = -kmax; /* l0 != 0 here */
				}
				l[2] = (!l[0] && !l[1]) ? 1 : -kmax;
			}
		}
		if (l[0] * l[0] + l[1] * l[1] + l[2] * l[2] > kmax * kmax) continue;
		return 1;
	}
}

static void kvec(const orc_system *s, const int l[3], double k[3]) {
	for (int p = 0; p < 3; p++) { /* :1586-1590 */
		k[p] = 0;
		for (int q = 0; q < 3; q++) k[p] += 2.0 * ORC_PI * s->recip[3 * p + q] * l[q];
	}
}

/* coulombic_reciprocal, System.Energy.cpp:1561-1622 */
double orc_coulombic_reciprocal(const orc_system *s) {
	double alpha = s->ewald_alpha, potential = 0;
	int l[3], started = 0;
	if (s->ewald_kmax <= 0) {
		/* kmax 0: the enumeration visits nothing */
		return 0.0 * (4.0 * ORC_PI / s->volume);
	}
	while (next_l(l, s->ewald_kmax, &started)) {
		double k[3];
		kvec(s, l, k);
		double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
		double SF_re = 0, SF_im = 0;
		for (int a = 0; a < s->n; a++) {
			if (s->frozen[a]) continue;         /* :1599 */
			if (s->charge[a] == 0.0) continue;  /* :1601 */
			const double *pos = s->pos + 3 * a;
			double pp = k[0] * pos[0] + k[1] * pos[1] + k[2] * pos[2];
			SF_re += s->charge[a] * cos(pp);
			SF_im += s->charge[a] * sin(pp);
		}
		potential += exp(-k2 / (4.0 * alpha * alpha)) / k2 * (SF_re * SF_re + SF_im * SF_im); /* :1613 */
	}
	potential *= 4.0 * ORC_PI / s->volume; /* :1619 */
	return potential;
}

/* coulombic_self, System.Energy.cpp:1626-1643 */
double orc_coulombic_self(const orc_system *s) {
	double self = 0.0;
	for (int a = 0; a < s->n; a++) {
		if (s->frozen[a]) continue;
		double e = s->ewald_alpha * s->charge[a] * s->charge[a] / sqrt(ORC_PI);
		self -= e;
	}
	return self;
}

/* ---------------------------------------------------------------------------------------------
 * static field: thole_field :3271-3296 -> thole_field_nopbc :3300-3333 | recip_term :2834-2896 + real_term :2900-2940
 * ------------------------------------------------------------------------------------------- */
static void field_nopbc(const orc_system *s, double *E) {
	for (int i = 0; i < s->n; i++)
		for (int j = i + 1; j < s->n; j++) {
			pair_par pp;
			double d[3], rr;
			pair_params(s, i, j, &pp);
			if (pp.frozen) continue;
			if (pp.intra) continue; /* :3313 */
			double r = orc_minimum_image(s, i, j, d, &rr);
			if ((r - ORC_SMALL_DR < s->cutoff) && (r != 0.)) {
				for (int p = 0; p < 3; p++) {
					E[3 * i + p] += s->charge[j] * d[p] / (r * r * r); /* :3322 */
					E[3 * j + p] -= s->charge[i] * d[p] / (r * r * r);
				}
			}
		}
}

static void field_recip(const orc_system *s, double *E) {
	double ea = s->polar_ewald_alpha;
	int l[3], started = 0;
	if (s->ewald_kmax > 0)
		while (next_l(l, s->ewald_kmax, &started)) {
			double k[3], kw[3];
			kvec(s, l, k);
			double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
			for (int p = 0; p < 3; p++) kw[p] = k[p] / k2 * exp(-k2 / (4.0 * ea * ea)); /* :2863-2865 */
			double f1 = 0, f2 = 0;
			for (int a = 0; a < s->n; a++) { /* ALL atoms, frozen included: :2868-2872 */
				const double *pos = s->pos + 3 * a;
				double kr = k[0] * pos[0] + k[1] * pos[1] + k[2] * pos[2];
				f1 += s->charge[a] * cos(kr);
				f2 += s->charge[a] * sin(kr);
			}
			for (int a = 0; a < s->n; a++) {
				const double *pos = s->pos + 3 * a;
				double kr = k[0] * pos[0] + k[1] * pos[1] + k[2] * pos[2];
				for (int p = 0; p < 3; p++) {
					E[3 * a + p] += kw[p] * sin(kr) * f1; /* :2877 */
					E[3 * a + p] -= kw[p] * cos(kr) * f2; /* :2878 */
				}
			}
		}
	for (int a = 0; a < s->n; a++)
		for (int p = 0; p < 3; p++) E[3 * a + p] *= 8.0 * ORC_PI / s->volume; /* :2890 */
}

static void field_real(const orc_system *s, double *E) {
	double a = s->polar_ewald_alpha;
	for (int i = 0; i < s->n; i++)
		for (int j = i + 1; j < s->n; j++) {
			pair_par pp;
			double d[3], rr;
			pair_params(s, i, j, &pp);
			if (pp.frozen) continue; /* :2915 */
			double r = orc_minimum_image(s, i, j, d, &rr);
			if ((r > s->cutoff) || (r == 0.0)) continue; /* :2917 */
			double r2 = r * r, factor;
			if (pp.es_excluded)
				factor = (2.0 * a * ORC_ONE_OVER_SQRT_PI * exp(-a * a * r2) * r - erf(a * r)) / (r * r2); /* :2921 */
			else
				factor = (2.0 * a * ORC_ONE_OVER_SQRT_PI * exp(-a * a * r2) * r + erfc(a * r)) / (r2 * r); /* :2929 */
			for (int p = 0; p < 3; p++) {
				E[3 * i + p] += factor * s->charge[j] * d[p];
				E[3 * j + p] -= factor * s->charge[i] * d[p];
			}
		}
}

void orc_thole_field(const orc_system *s, double *E) {
	memset(E, 0, sizeof(double) * 3 * (size_t)s->n);
	if (s->polar_ewald) {
		field_recip(s, E);
		field_real(s, E);
	} else
		field_nopbc(s, E);
}

/* ---------------------------------------------------------------------------------------------
 * thole_amatrix, System.Energy.cpp:2661-2770 (exponential damping :2731-2734)
 * ------------------------------------------------------------------------------------------- */
static void tensor_upper(const orc_system *s, int i, int j, double T[9]) {
	double d[3], rr;
	double r = orc_minimum_image(s, i, j, d, &rr);
	double r2 = r * r, ir = 0, ir3, ir5;
	double l = s->polar_damp, l2 = l * l, l3 = l2 * l;
	if (r == 0.)
		ir3 = ir5 = ORC_MAXVALUE;
	else {
		ir = 1.0 / r;
		ir3 = ir * ir * ir;
		ir5 = ir3 * ir * ir;
	}
	double explr = exp(-l * r);
	double damp1 = 1.0 - explr * (0.5 * l2 * r2 + l * r + 1.0);
	double damp2 = damp1 - explr * (l3 * r2 * r / 6.0);
	for (int p = 0; p < 3; p++)
		for (int q = 0; q < 3; q++) {
			T[3 * p + q] = -3.0 * d[p] * d[q] * damp2 * ir5; /* :2748 */
			if (p == q) T[3 * p + q] += damp1 * ir3;       /* :2754 */
		}
}

void orc_thole_amatrix_block(const orc_system *s, int i, int j, double block[9]) {
	memset(block, 0, 9 * sizeof(double));
	if (i == j) {
		for (int p = 0; p < 3; p++) block[4 * p] = (s->polarizability[i] != 0.0) ? 1.0 / s->polarizability[i] : ORC_MAXVALUE;
		return;
	}
	if (i < j)
		tensor_upper(s, i, j, block);
	else { /* lower half is a copy of the upper block, NOT its transpose: A[jj+p][ii+q] = A[ii+p][jj+q] (:2762-2764) */
		tensor_upper(s, j, i, block);
	}
}

static double **build_amatrix(const orc_system *s) {
	int n3 = 3 * s->n;
	double **A = (double **)calloc((size_t)n3, sizeof(double *));
	if (!A) return NULL;
	for (int r = 0; r < n3; r++) {
		A[r] = (double *)calloc((size_t)n3, sizeof(double));
		if (!A[r]) return NULL;
	}
	for (int i = 0; i < s->n; i++)
		for (int p = 0; p < 3; p++) A[3 * i + p][3 * i + p] = (s->polarizability[i] != 0.0) ? 1.0 / s->polarizability[i] : ORC_MAXVALUE;
	for (int i = 0; i < s->n - 1; i++)
		for (int j = i + 1; j < s->n; j++) {
			double T[9];
			tensor_upper(s, i, j, T);
			for (int p = 0; p < 3; p++)
				for (int q = 0; q < 3; q++) {
					A[3 * i + p][3 * j + q] = T[3 * p + q];
					A[3 * j + p][3 * i + q] = T[3 * p + q];
				}
		}
	return A;
}

static void free_amatrix(double **A, int n3) {
	if (!A) return;
	for (int r = 0; r < n3; r++) free(A[r]);
	free(A);
}

/* thole_iterative :3450-3543 with init_dipoles :3547-3560, contract_dipoles :3564-3598,
 * calc_dipole_rrms :3147-3177, are_we_done_yet :3215-3239.  Returns iteration count. */
static int thole_iterative(const orc_system *s, double **A, const double *E, double *mu, double *Find, double *rrms_atom,
                           int *iterator_failed) {
	int n = s->n;
	double *old_mu = (double *)calloc(3 * (size_t)n, sizeof(double));
	double *new_mu = (double *)calloc(3 * (size_t)n, sizeof(double));
	const double *alpha = s->polarizability;
	*iterator_failed = 0;
	for (int i = 0; i < n; i++)
		for (int p = 0; p < 3; p++) {
			mu[3 * i + p] = alpha[i] * (E[3 * i + p] + 0.0);
			mu[3 * i + p] *= s->polar_gamma; /* :3555-3556 (no SOR/ESOR in scope) */
		}
	int it = 0, keep = 1;
	while (keep) {
		it++;
		if (it >= ORC_MAX_ITER && s->polar_precision) { /* :3483-3494 */
			for (int i = 0; i < n; i++)
				for (int p = 0; p < 3; p++) mu[3 * i + p] = alpha[i] * (E[3 * i + p] + 0.0);
			*iterator_failed = 1;
			break;
		}
		memset(Find, 0, sizeof(double) * 3 * (size_t)n);
		if (s->polar_rrms || s->polar_precision > 0) memcpy(old_mu, mu, sizeof(double) * 3 * (size_t)n);

		for (int i = 0; i < n; i++) { /* contract_dipoles, natural order (no ranking) */
			int ii = 3 * i;
			if (alpha[i] == 0) {
				new_mu[ii] = new_mu[ii + 1] = new_mu[ii + 2] = 0;
				mu[ii] = mu[ii + 1] = mu[ii + 2] = 0;
				continue;
			}
			for (int j = 0; j < n; j++) {
				int jj = 3 * j;
				if (i != j)
					for (int p = 0; p < 3; p++) {
						const double *row = A[ii + p] + jj;
						Find[ii + p] -= row[0] * mu[jj] + row[1] * mu[jj + 1] + row[2] * mu[jj + 2]; /* :3583 */
					}
			}
			for (int p = 0; p < 3; p++) {
				new_mu[ii + p] = alpha[i] * (E[ii + p] + 0.0 + Find[ii + p]); /* :3588 */
				if (s->polar_gs) mu[ii + p] = new_mu[ii + p];                  /* :3591 Gauss-Seidel */
			}
		}

		if (s->polar_rrms || s->polar_precision > 0) /* calc_dipole_rrms */
			for (int i = 0; i < n; i++) {
				double acc = 0;
				for (int p = 0; p < 3; p++) {
					double c = new_mu[3 * i + p] - old_mu[3 * i + p];
					acc += c * c;
				}
				acc /= new_mu[3 * i] * new_mu[3 * i] + new_mu[3 * i + 1] * new_mu[3 * i + 1] + new_mu[3 * i + 2] * new_mu[3 * i + 2];
				acc = sqrt(acc);
				if (!isfinite(acc)) acc = 0;
				rrms_atom[i] = acc;
			}

		/* are_we_done_yet */
		if (s->polar_precision == 0.0) {
			keep = (it != s->polar_max_iter);
		} else {
			double allowed = s->polar_precision * s->polar_precision * ORC_DEBYE2SKA * ORC_DEBYE2SKA;
			keep = 0;
			for (int i = 0; i < n && !keep; i++)
				for (int p = 0; p < 3; p++) {
					double e = new_mu[3 * i + p] - old_mu[3 * i + p];
					if (e * e > allowed) {
						keep = 1;
						break;
					}
				}
		}
		memcpy(mu, new_mu, sizeof(double) * 3 * (size_t)n); /* :3534 */
	}
	free(old_mu);
	free(new_mu);
	return it;
}

/* polar, System.Energy.cpp:2534-2635 (iterative branch) */
double orc_polar(const orc_system *s, orc_result *out, double *ef_static, double *mu, double *ef_induced) {
	int n = s->n;
	double *E = ef_static ? ef_static : (double *)calloc(3 * (size_t)n, sizeof(double));
	double *M = mu ? mu : (double *)calloc(3 * (size_t)n, sizeof(double));
	double *F = ef_induced ? ef_induced : (double *)calloc(3 * (size_t)n, sizeof(double));
	double *rrms = (double *)calloc((size_t)n, sizeof(double));
	double **A = build_amatrix(s);
	orc_thole_field(s, E);
	int failed = 0;
	int iters = thole_iterative(s, A, E, M, F, rrms, &failed);
	double potential = 0;
	for (int i = 0; i < n; i++) potential += M[3 * i] * E[3 * i] + M[3 * i + 1] * E[3 * i + 1] + M[3 * i + 2] * E[3 * i + 2];
	potential *= -0.5;
	if (out) {
		double acc = 0;
		for (int i = 0; i < n; i++)
			if (isfinite(rrms[i])) acc += rrms[i];
		out->dipole_rrms = acc / (double)n; /* get_dipole_rrms :2639-2657 */
		out->polar_iterations = iters;
		out->iterator_failed = failed;
	}
	free_amatrix(A, 3 * n);
	free(rrms);
	if (!ef_static) free(E);
	if (!mu) free(M);
	if (!ef_induced) free(F);
	return potential;
}

/* System::energy, System.Energy.cpp:19-171 */
int orc_energy(const orc_system *s, orc_result *out, double *ef_static, double *mu, double *ef_induced) {
	memset(out, 0, sizeof(*out));
	double rd = 0, es = 0, pol = 0, vdw = 0, three = 0;
	if (!s->rd_only) {
		if (s->wolf) { /* coulombic() :1404-1405 */
			es = orc_coulombic_wolf(s);
		} else {
			out->es_real = orc_coulombic_real(s, out);
			out->es_recip = orc_coulombic_reciprocal(s);
			out->es_self = orc_coulombic_self(s);
			es = out->es_real + out->es_recip + out->es_self; /* :1412 */
		}
		if (s->polarization) pol = orc_polar(s, out, ef_static, mu, ef_induced);
	}
	rd = orc_lj(s, out);
	out->rd_energy = rd;
	out->coulombic_energy = es;
	out->polarization_energy = pol;
	out->vdw_energy = vdw;
	out->energy = rd + es + pol + vdw + three; /* :136 */
	return 0;
}

/* PI_calculate_potential, SimulationControl.PathIntegral.cpp:786-804 */
double orc_pi_aggregate(int P, const double *rd, const double *es, const double *pol, const double *vdw, double o[4]) {
	o[0] = o[1] = o[2] = o[3] = 0;
	for (int b = 0; b < P; b++) {
		o[0] += rd[b];
		o[1] += es[b];
		o[2] += pol[b];
		o[3] += vdw ? vdw[b] : 0.0;
	}
	for (int c = 0; c < 4; c++) o[c] /= P;
	return o[0] + o[1] + o[3] + o[2]; /* rd + coulombic + vdw + polarization (:803-804) */
}

/* PI_calculate_kinetic, SimulationControl.PathIntegral.cpp:806-824, with PI_chain_mass_length2_ENTIRE_SYSTEM (:851-896),
 * PI_chain_mass_length2(vector<Molecule*>&) (:908-965) and Molecule::update_COM (src/Molecule.cpp:259-281) restated on
 * flat arrays: pos [P][n][3] (image-major), mass[n], mol[n] (consecutive equal ids = one molecule), frozen[n] (the
 * molecule flag is that of its first atom, src/System.cpp:684).  Returns K in Kelvin; *chain_out = chain_mass_len2. */
double orc_pi_kinetic(int P, int n, const double *pos, const double *mass, const int *mol, const int *frozen, double T, double *chain_out) {
	const double kB = 1.3806503e-23, hBar2 = 1.11211999e-68, AMU2KG = 1.66053873e-27, A2M = 1.0e-10; /* constants.h */
	double sum = 0, N = 0;
	double *cx = (double *)malloc(sizeof(double) * 3 * (size_t)P);
	for (int a0 = 0; a0 < n;) {
		int a1 = a0;
		while (a1 < n && mol[a1] == mol[a0]) a1++;
		if (!frozen[a0]) { /* :881 (adiabatic / target molecules are outside the hot path) */
			double M = 0;
			N += 1.0; /* System::countN, src/System.cpp:909-931 */
			for (int s = 0; s < P; s++) {
				const double *p = pos + 3 * (size_t)s * n;
				double m = 0, c0 = 0, c1 = 0, c2 = 0;
				for (int a = a0; a < a1; a++) {
					m += mass[a];
					c0 += mass[a] * p[3 * a + 0];
					c1 += mass[a] * p[3 * a + 1];
					c2 += mass[a] * p[3 * a + 2];
				}
				cx[3 * s + 0] = c0 / m;
				cx[3 * s + 1] = c1 / m;
				cx[3 * s + 2] = c2 / m;
				if (s == 0) M = m;
			}
			double len2 = 0;
			for (int i = 0; i < P; i++) {
				const int j = (i + 1) % P;
				const double dx = cx[3 * i] - cx[3 * j], dy = cx[3 * i + 1] - cx[3 * j + 1], dz = cx[3 * i + 2] - cx[3 * j + 2];
				len2 += dx * dx + dy * dy + dz * dz;
			}
			len2 *= (M * AMU2KG) * (A2M * A2M);
			sum += len2;
		}
		a0 = a1;
	}
	free(cx);
	if (chain_out) *chain_out = sum;
	const double d = 3.0, Pd = (double)P;
	const double beta = 1.0 / (kB * T);
	const double omega2 = Pd / (beta * beta * hBar2);
	const double t1 = 0.5 * d * N * kB * T * Pd;
	const double t2 = 0.5 * omega2 * sum;
	return (1.0 / kB) * (t1 - t2);
}

/* ---------------------------------------------------------------------------------------------
 * CPU-baseline timing on a BOUNDED SAMPLE of one evaluation (bench.py cpu_baseline leg only).
 * Every O(N^2) stage of energy() is executed for the rows i = 0, stride, 2*stride, ... only, with exactly the
 * per-pair arithmetic of the full functions above, timed, and scaled by that stage's exact work ratio
 * (pairs in the full loop / pairs executed).  The O(K*N) reciprocal stages and the O(N) terms run in full.
 * out_sec[0..6] = estimated seconds of one FULL evaluation per stage:
 *   0 lj+lrc, 1 coulombic_real, 2 coulombic_reciprocal(+self), 3 thole_amatrix, 4 thole_field (recip+real),
 *   5 thole_iterative (polar_max_iter dense contractions), 6 total.
 * Returns the wall seconds actually spent.
 * ------------------------------------------------------------------------------------------- */
#include <time.h>
static double now_sec(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double orc_time_sample(const orc_system *s, int stride, double out_sec[7]) {
	const int n = s->n;
	const double t_begin = now_sec();
	volatile double sink = 0;
	double pairs_full = 0.5 * (double)n * (double)(n - 1), pairs_done = 0;
	int rows = 0;
	for (int i = 0; i < n; i += stride) {
		pairs_done += (double)(n - 1 - i);
		rows++;
	}
	if (pairs_done <= 0) pairs_done = 1;
	const double tri_scale = pairs_full / pairs_done; /* triangular (i<j) loops */
	const double row_scale = (double)n / (double)rows; /* full-row loops        */
	for (int k = 0; k < 7; k++) out_sec[k] = 0;

	/* 0: lj() rows */
	double t0 = now_sec();
	{
		double potential = 0, cutoff = s->cutoff;
		for (int i = 0; i < n; i += stride)
			for (int j = i + 1; j < n; j++) {
				pair_par pp;
				double dimg[3], r;
				pair_params(s, i, j, &pp);
				double rimg = 0;
				if (!pp.frozen || s->polarization) rimg = orc_minimum_image(s, i, j, dimg, &r);
				double lrc = 0, rd = 0;
				if (s->rd_lrc && pp.epsilon != 0 && pp.sigma != 0 && !pp.frozen) lrc = lrc_term(pp.sigma, pp.epsilon, cutoff, s->volume);
				if ((rimg - ORC_SMALL_DR < cutoff) && !pp.rd_excluded && !pp.frozen) {
					double sor = fabs(pp.sigma) / rimg;
					double sor6 = sor * sor * sor;
					sor6 *= sor6;
					double sor12 = sor6 * sor6;
					rd += 4.0 * pp.epsilon * ((pp.attractive_only ? 0 : sor12) - sor6);
				}
				potential += rd + lrc;
			}
		sink += potential;
	}
	out_sec[0] = (now_sec() - t0) * tri_scale;

	if (!s->rd_only) {
		/* 1: coulombic_real rows */
		t0 = now_sec();
		{
			double alpha = s->ewald_alpha, potential = 0;
			for (int i = 0; i < n; i += stride)
				for (int j = i + 1; j < n; j++) {
					pair_par pp;
					double dimg[3], rr = 0;
					pair_params(s, i, j, &pp);
					double es_real = 0, es_self_intra = 0;
					if (!pp.frozen) {
						double r = orc_minimum_image(s, i, j, dimg, &rr);
						if (!((r > s->cutoff) || pp.es_excluded)) es_real = s->charge[i] * s->charge[j] * erfc(alpha * r) / r;
						else if (pp.es_excluded) es_self_intra = s->charge[i] * s->charge[j] * erf(alpha * rr) / rr;
					}
					potential += es_real - es_self_intra;
				}
			sink += potential;
		}
		out_sec[1] = (now_sec() - t0) * tri_scale;

		/* 2: reciprocal + self, in full */
		t0 = now_sec();
		sink += orc_coulombic_reciprocal(s) + orc_coulombic_self(s);
		out_sec[2] = now_sec() - t0;

		if (s->polarization) {
			double *E = (double *)calloc(3 * (size_t)n, sizeof(double));
			double *mu = (double *)calloc(3 * (size_t)n, sizeof(double));
			double *rowbuf = (double *)calloc(9 * (size_t)n, sizeof(double)); /* one atom's 3 dense rows */

			/* 4: thole_field: reciprocal part in full, real part on sampled rows */
			t0 = now_sec();
			if (s->polar_ewald) field_recip(s, E);
			double t_recip = now_sec() - t0;
			t0 = now_sec();
			for (int i = 0; i < n; i += stride)
				for (int j = i + 1; j < n; j++) {
					pair_par pp;
					double d[3], rr;
					pair_params(s, i, j, &pp);
					if (pp.frozen) continue;
					double r = orc_minimum_image(s, i, j, d, &rr);
					double factor;
					if (s->polar_ewald) {
						if ((r > s->cutoff) || (r == 0.0)) continue;
						double a = s->polar_ewald_alpha, r2 = r * r;
						if (pp.es_excluded) factor = (2.0 * a * ORC_ONE_OVER_SQRT_PI * exp(-a * a * r2) * r - erf(a * r)) / (r * r2);
						else factor = (2.0 * a * ORC_ONE_OVER_SQRT_PI * exp(-a * a * r2) * r + erfc(a * r)) / (r2 * r);
					} else {
						if (pp.intra || !((r - ORC_SMALL_DR < s->cutoff) && (r != 0.))) continue;
						factor = 1.0 / (r * r * r);
					}
					for (int p = 0; p < 3; p++) {
						E[3 * i + p] += factor * s->charge[j] * d[p];
						E[3 * j + p] -= factor * s->charge[i] * d[p];
					}
				}
			out_sec[4] = t_recip + (now_sec() - t0) * tri_scale;
			for (int i = 0; i < 3 * n; i++) mu[i] = s->polarizability[i / 3] * E[i];

			/* 3 + 5: thole_amatrix rows of the sampled atoms, then polar_max_iter dense contractions that stream those
			 * rows from memory iteration by iteration (as the reference streams its 3N x 3N matrix) */
			free(rowbuf);
			rowbuf = (double *)calloc((size_t)rows * 9 * (size_t)n, sizeof(double));
			double t_build = 0, t_contract = 0;
			t0 = now_sec();
			{
				int k = 0;
				for (int i = 0; i < n; i += stride, k++) {
					double *rb = rowbuf + (size_t)k * 9 * n;
					for (int j = 0; j < n; j++) {
						double T[9];
						if (j == i) continue;
						tensor_upper(s, i < j ? i : j, i < j ? j : i, T);
						for (int p = 0; p < 3; p++)
							for (int q = 0; q < 3; q++) rb[(size_t)p * 3 * n + 3 * j + q] = T[3 * p + q];
					}
				}
			}
			t_build = now_sec() - t0;
			t0 = now_sec();
			for (int it = 0; it < s->polar_max_iter; it++) {
				int k = 0;
				for (int i = 0; i < n; i += stride, k++) {
					const double *rb = rowbuf + (size_t)k * 9 * n;
					double f[3] = {0, 0, 0};
					for (int j = 0; j < n; j++)
						if (j != i)
							for (int p = 0; p < 3; p++) {
								const double *row = rb + (size_t)p * 3 * n + 3 * j;
								f[p] -= row[0] * mu[3 * j] + row[1] * mu[3 * j + 1] + row[2] * mu[3 * j + 2];
							}
					sink += f[0] + f[1] + f[2];
				}
			}
			t_contract = now_sec() - t0;
			out_sec[3] = t_build * row_scale * 0.5; /* the full build fills each (i<j) block once and mirrors it */
			out_sec[5] = t_contract * row_scale;
			free(E);
			free(mu);
			free(rowbuf);
		}
	}
	for (int k = 0; k < 6; k++) out_sec[6] += out_sec[k];
	(void)sink;
	return now_sec() - t_begin;
}
