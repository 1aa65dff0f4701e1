/* oracle/mpmc_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C restatement of the reference's per-move energy path (b-tudor/mpmcxx,
 * System::energy() and what it calls).  It is the checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product library
 * (mpmcxx_amd/libmpmc_energy.so) never links, includes or calls anything in oracle/.
 *
 * Parity status: PINNED -- every function below is checked (tests/test_oracle_vs_golden.py)
 * against golden vectors generated in the build container by running the reference's own
 * object code (oracle/_ref/ref_harness, built by oracle/Makefile from /root/reference/src).
 */
#ifndef MPMC_ORACLE_H
#define MPMC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_system {
	int n;                        /* atoms, in reference atom_array order (System.cpp:881-904) */
	const double *pos;            /* [n][3] absolute (unwrapped) positions, Angstrom            */
	const double *charge;         /* [n] reduced units sqrt(K*A) (e * 408.7816, System.cpp:624) */
	const double *polarizability; /* [n] A^3                                                    */
	const double *epsilon;        /* [n] K                                                      */
	const double *sigma;          /* [n] A                                                      */
	const int *mol_id;            /* [n] molecule identity (equal ids == same Molecule)         */
	const int *frozen;            /* [n]                                                        */
	const int *has_disp;          /* [n] or NULL: nonzero iff any of c6,c8,c10 != 0             */
	double basis[9];              /* pbc.basis[q][p] row-major: rows = lattice vectors          */
	double recip[9];              /* pbc.reciprocal_basis[q][p] row-major (= inverse of basis)  */
	double volume, cutoff;
	/* options (reference keyword names, SimulationControl.cpp) */
	int rd_only, rd_lrc;
	int polarization, polar_iterative, polar_ewald, polar_max_iter, polar_gs, polar_rrms;
	int ewald_kmax;
	double polar_precision, polar_gamma, polar_damp;
	double ewald_alpha, polar_ewald_alpha;
	/* adjacent physics: Wolf electrostatics (coulombic_wolf :1420-1462) and Feynman-Hibbs corrections
	 * (lj_fh_corr :1100-1148, coulombic_real_FH :1521-1557) */
	int wolf, feynman_hibbs, feynman_hibbs_order;
	double temperature;
	const double *mass; /* [n] atom masses (amu); molecule mass = sum over the molecule (System.cpp:687) */
} orc_system;

typedef struct orc_result {
	double energy, rd_energy, coulombic_energy, polarization_energy, vdw_energy;
	double es_real, es_recip, es_self;
	double lj_pairs, lrc_pair, lrc_self;
	double dipole_rrms;
	long long n_pairs, n_intra, n_rd_excluded, n_es_excluded, n_frozen;
	long long n_lj_in_cutoff, n_es_in_cutoff;
	int polar_iterations, iterator_failed;
} orc_result;

/* PeriodicBoundary::update (PeriodicBoundary.cpp:31-101) + update_pbc alpha defaults (System.cpp:871-874) */
void orc_pbc_update(const double basis[9], double recip[9], double *volume, double *cutoff);

/* minimum_image (System.cpp:1202-1279): d = r_i - r_j ; returns rimg, fills dimg[3], *r */
double orc_minimum_image(const orc_system *s, int i, int j, double dimg[3], double *r);

/* System::energy (System.Energy.cpp:19-171).  ef_static/mu/ef_induced: [n][3] outputs or NULL. */
int orc_energy(const orc_system *s, orc_result *out, double *ef_static, double *mu, double *ef_induced);

/* component entry points (same names as the reference's public members, System.h:346-402) */
double orc_lj(const orc_system *s, orc_result *out);
/* lj() with exactly rounded sums (long double + Neumaier) next to the reference's list-order sums; see mpmc_oracle.c.  CPU only. */
void orc_lj_exact(const orc_system *s, double out7[7]);
double orc_coulombic_real(const orc_system *s, orc_result *out);
double orc_coulombic_reciprocal(const orc_system *s);
double orc_coulombic_self(const orc_system *s);
double orc_coulombic_wolf(const orc_system *s);
void orc_thole_field(const orc_system *s, double *ef_static);
/* one 3x3 block A[3i..][3j..] of thole_amatrix (System.Energy.cpp:2661-2770) */
void orc_thole_amatrix_block(const orc_system *s, int i, int j, double block[9]);
/* polar(): returns U_pol; fills ef_static, mu, ef_induced ([n][3]) */
double orc_polar(const orc_system *s, orc_result *out, double *ef_static, double *mu, double *ef_induced);

/* PI_calculate_potential aggregate (SimulationControl.PathIntegral.cpp:786-804):
 * ordered sum over beads s=0..P-1, divided by P.  out4 = {rd, coulombic, polarization, vdw}; returns V */
double orc_pi_aggregate(int P, const double *rd, const double *es, const double *pol, const double *vdw, double out4[4]);
/* PI_calculate_kinetic (PathIntegral.cpp:806-824) over P images: pos [P][n][3]; returns Kelvin, *chain_out = chain_mass_len2 */
double orc_pi_kinetic(int P, int n, const double *pos, const double *mass, const int *mol, const int *frozen, double T, double *chain_out);

/* bench.py cpu_baseline leg: time every stage of ONE evaluation on the rows i = 0, stride, 2 stride, ... and
 * scale each stage by its exact work ratio; out_sec[6] = estimated seconds of one full evaluation. */
double orc_time_sample(const orc_system *s, int stride, double out_sec[7]);

#ifdef __cplusplus
}
#endif
#endif
