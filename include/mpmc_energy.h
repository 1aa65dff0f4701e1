/* include/mpmc_energy.h -- C ABI of the MI355X-native energy hot path (libmpmc_energy.so).
 *
 * This is the drop-in boundary for ONE path of b-tudor/mpmcxx: the per-move potential-energy
 * evaluation `double System::energy()` (reference src/System.h:315, src/System.Energy.cpp:19-171) and the
 * path-integral per-bead loop around it (src/SimulationControl.PathIntegral.cpp:752-805).
 * A reference-side adapter flattens the System's Molecule->Atom lists into the arrays below, calls
 * mpmc_energy(), and copies mpmc_result into System::observables.
 *
 * Conventions
 *  - plain C, no C++/torch types; all floating point is IEEE fp64; energies in Kelvin, lengths in Angstrom,
 *    charges in reduced units sqrt(K*A) (e * 408.7816, reference src/System.cpp:624).
 *  - every entry point returns MPMC_OK (0) or a negative/positive error code; mpmc_last_error() gives text.
 *    Codes reuse the reference's throw-int values where one exists (src/constants.h:108-147).
 *  - a context is the device-side state of ONE System (one box / one PI bead): its own HIP stream and
 *    device buffers.  Re-entrant per context (the reference calls energy() concurrently from P threads on P
 *    distinct Systems, PathIntegral.cpp:772-779); never call concurrently on the same context.
 *  - host pointers unless a name says _device.
 *  - there is NO CPU fallback: without a usable HIP device mpmc_ctx_create fails with MPMC_ERR_NO_DEVICE.
 */
#ifndef MPMC_ENERGY_H
#define MPMC_ENERGY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPMC_ABI_VERSION 6

/* ---- status codes -------------------------------------------------------------------------------------- */
#define MPMC_OK 0
#define MPMC_ERR_INTERNAL 101            /* reference internal_error (constants.h:110)                    */
#define MPMC_ERR_MEMORY 2000             /* memory_request_fail                                           */
#define MPMC_ERR_INVALID_SETTING 4000    /* invalid_setting                                               */
#define MPMC_ERR_INCOMPATIBLE 4002       /* incompatible_settings                                         */
#define MPMC_ERR_UNSUPPORTED 4004        /* unsupported_setting: physics outside the hot path */
#define MPMC_ERR_INVALID_DATUM 6001      /* invalid_datum (bad atom arrays)                               */
#define MPMC_ERR_BOX 6004                /* invalid_box_dimensions                                        */
#define MPMC_ERR_INVALID_MC_MOVE 20000   /* internal_err_invalid_mc_move: the two boxes of a coordinated move disagree (constants.h:147) */
#define MPMC_ERR_INVALID_MC_MOVE_KIND 102 /* invalid_monte_carlo_move (constants.h:110)                  */
#define MPMC_ERR_NO_DEVICE (-1)          /* no HIP device / HIP runtime failure at create                 */
#define MPMC_ERR_HIP (-2)                /* a HIP call failed (text in mpmc_last_error)                   */
#define MPMC_ERR_ARG (-3)                /* NULL / out-of-range argument                                  */
#define MPMC_ERR_COMM (-4)               /* RCCL missing or an RCCL call failed (text in mpmc_comm_last_error) */

/* ---- damping (reference enum constants.h:66-70) -------------------------------------------------------- */
#define MPMC_DAMPING_OFF 0
#define MPMC_DAMPING_LINEAR 1
#define MPMC_DAMPING_EXPONENTIAL 2

/* ---- how the Thole dipole iteration is executed on the device (not a reference option) ------------------ */
#define MPMC_SOLVER_AUTO 0         /* COMPACT when its store fits, else MATRIX_FREE                         */
#define MPMC_SOLVER_MATRIX_FREE 1  /* nothing stored: the damped tensors are rebuilt from the positions in every
                                    * iteration by the same symmetric kernel (7 % slower, 20x less memory)       */
#define MPMC_SOLVER_COMPACT 2      /* store (d1/r^3, 3 d2/r^5) per unordered pair, 16 B/pair (HBM-bound)      */
#define MPMC_SOLVER_DENSE 3        /* the reference's dense 3N x 3N A matrix in device memory, contraction on the
                                    * fp64 matrix cores; on request only (HBM-bound at 9x the bytes of COMPACT)  */

/* ---- out-of-scope reference switches: pass the ones that are ON so the library can refuse them ---------- */
#define MPMC_FLAG_WOLF (1ull << 0)          /* (now supported: mpmc_options.wolf; the bit stays for ABI stability) */
#define MPMC_FLAG_FEYNMAN_HIBBS (1ull << 1) /* (now supported: mpmc_options.feynman_hibbs) */
#define MPMC_FLAG_RD_CRYSTAL (1ull << 2)
#define MPMC_FLAG_SPECTRE (1ull << 3)
#define MPMC_FLAG_GWP (1ull << 4)
#define MPMC_FLAG_USE_SG (1ull << 5)
#define MPMC_FLAG_POLARVDW (1ull << 6)
#define MPMC_FLAG_POLAR_EWALD_FULL (1ull << 7)
#define MPMC_FLAG_POLAR_WOLF (1ull << 8)
#define MPMC_FLAG_POLAR_PALMO (1ull << 9)
#define MPMC_FLAG_POLAR_GS_RANKED (1ull << 10)
#define MPMC_FLAG_POLAR_SOR (1ull << 11)
#define MPMC_FLAG_POLAR_ZODID (1ull << 12)
#define MPMC_FLAG_NON_LB_MIXING (1ull << 13) /* waldmanhagler / halgren / c6_mixing / cdvdw_* */
#define MPMC_FLAG_OTHER_RD (1ull << 14)      /* dreiding / lj_buffered_14_7 / disp_expansion / anharmonic / exp_repulsion */
#define MPMC_FLAG_AXILROD_TELLER (1ull << 15)
#define MPMC_FLAG_CAVITY_AUTOREJECT (1ull << 16)
#define MPMC_FLAG_POLAR_MATRIX_INVERSION (1ull << 17) /* polarization on with polar_iterative off */

typedef struct mpmc_ctx mpmc_ctx; /* opaque: device buffers + stream of one System */

/* Options = the reference keywords that steer energy() (src/SimulationControl.cpp line in comment) */
typedef struct mpmc_options {
	int32_t rd_only;          /* :977  skip electrostatics + polarization                                   */
	int32_t rd_lrc;           /* :1003 LJ long-range correction (default on)                                */
	int32_t polarization;     /* :653                                                                       */
	int32_t polar_iterative;  /* :1240 (required when polarization is on)                                   */
	int32_t polar_ewald;      /* :718/:1210 static field by Ewald (recip_term + real_term) instead of nopbc */
	int32_t polar_max_iter;   /* :1303 fixed iteration count when polar_precision == 0                      */
	int32_t polar_gs;         /* :1256 Gauss-Seidel: in-place sweeps in atom order (serial over 64-atom tiles)  */
	int32_t polar_rrms;       /* :1320 compute per-atom dipole RRMS every iteration                         */
	int32_t damp_type;        /* :1308 must be MPMC_DAMPING_EXPONENTIAL when polarization is on             */
	int32_t ewald_kmax;       /* :1199 (default 7)                                                          */
	int32_t solver;           /* MPMC_SOLVER_*                                                              */
	int32_t wolf;             /* :994  Wolf electrostatics instead of Ewald in coulombic() (coulombic_wolf :1420-1462) */
	double polar_precision;   /* :1298 0 => fixed count; else stop when every |d mu| < precision*DEBYE2SKA  */
	double polar_gamma;       /* :1288 pre-scaling of the initial dipoles (default 1.0)                     */
	double polar_damp;        /* :1293 Thole exponential damping length parameter                           */
	double ewald_alpha;       /* :1192 <= 0 means "unset": 3.5 / cutoff (System.cpp:871-872)                 */
	double polar_ewald_alpha; /* :1218 <= 0 means "unset": 3.5 / cutoff (System.cpp:873-874)                 */
	uint64_t unsupported_flags; /* OR of MPMC_FLAG_* that are ON in the caller's System                     */
	int32_t feynman_hibbs;       /* :1042 Feynman-Hibbs corrections to lj() and coulombic_real() (:1100-1148, :1521-1557) */
	int32_t feynman_hibbs_order; /* :1067 2 or 4 (anything else: 2, SimulationControl.cpp:2497-2500)           */
	double temperature;          /* K; needed by Feynman-Hibbs (needs atom masses in mpmc_set_atoms)           */
} mpmc_options;

/* what energy() leaves in System::observables / nodestats (src/System.h:94-113,151-185) + parity diagnostics */
typedef struct mpmc_result {
	double energy;              /* observables->energy  = rd + coulombic + polarization + vdw + three_body  */
	double rd_energy;           /* observables->rd_energy            (lj(): pairs + pair LRC + self LRC)    */
	double coulombic_energy;    /* observables->coulombic_energy     (real + reciprocal + self)             */
	double polarization_energy; /* observables->polarization_energy  (-1/2 sum mu.E0)                       */
	double vdw_energy;          /* always 0 (polarvdw is out of scope)                                      */
	double three_body_energy;   /* always 0                                                                 */
	double kinetic_energy;      /* always 0 (gwp out of scope)                                              */
	double es_real;             /* coulombic_real()                                                         */
	double es_recip;            /* coulombic_reciprocal()                                                   */
	double es_self;             /* coulombic_self()                                                         */
	double lj_pairs;            /* sum of pair->rd_energy                                                   */
	double lrc_pair;            /* sum of pair->lrc                                                         */
	double lrc_self;            /* sum of lj_lrc_self                                                       */
	double dipole_rrms;         /* observables->dipole_rrms                                                 */
	double N;                   /* observables->N  (non-frozen molecules, countN System.cpp:909-931)        */
	double NU;                  /* observables->NU = N * energy                                             */
	int64_t n_pairs;            /* N(N-1)/2                                                                 */
	int64_t n_lj_in_cutoff;     /* pairs passing  rimg-1e-12 < rc && !rd_excluded && !frozen                */
	int64_t n_es_in_cutoff;     /* pairs passing  !frozen && !(rimg > rc || es_excluded)                    */
	int64_t n_intra;            /* same-molecule pairs                                                      */
	int64_t n_rd_excluded;
	int64_t n_es_excluded;
	int64_t n_frozen;           /* pairs with both atoms frozen                                             */
	int32_t polar_iterations;   /* nodestats->polarization_iterations                                       */
	int32_t iterator_failed;    /* System::iterator_failed (=> caller rejects the move)                      */
} mpmc_result;

/* accumulated device time of the kernels of one context, measured with HIP events on the context's stream
 * (only while profiling is enabled with mpmc_set_profiling).  Index with MPMC_K_*. */
#define MPMC_K_PAIR 0        /* LJ + real-space Coulomb pair kernel                 */
#define MPMC_K_RECIP 1       /* structure factors + reciprocal energy + atom terms  */
#define MPMC_K_FIELD 2       /* static field (recip + real, or nopbc)               */
#define MPMC_K_TENSOR 3      /* dense thole_amatrix rows (mpmc_thole_amatrix)        */
#define MPMC_K_DIPOLE_ITER 4 /* Jacobi contraction, stored tensors streamed (one launch per iteration; MATRIX_FREE: its kernel) */
#define MPMC_K_REDUCE 5      /* dipole update / final reductions / polarization energy */
#define MPMC_K_DIPOLE_FAR 6  /* unused since ABI 4 (the two-kernel form of the Jacobi contraction is gone); the slot stays for layout stability */
#define MPMC_K_CLASSES 7     /* tile bounding boxes, tile-pair classes, panel table of the Jacobi contraction     */
#define MPMC_K_COUNT 8
typedef struct mpmc_timings {
	double ms[MPMC_K_COUNT];      /* summed elapsed milliseconds                                           */
	int64_t launches[MPMC_K_COUNT]; /* number of timed launches                                            */
} mpmc_timings;

/* ---- library ---------------------------------------------------------------------------------------- */
int mpmc_abi_version(void);
int mpmc_device_count(int *count);
/* (ABI 6) for host programs that carry no HIP binding of their own (bench.py's ranks import neither torch nor hip-python): a fence over
 * everything this process has enqueued on `device`, and "<marketing name> (<gcnArchName>, <n> CUs)" for the job's log. */
int mpmc_device_synchronize(int device);
int mpmc_device_name(int device, char *name, int capacity);
const char *mpmc_last_error(const mpmc_ctx *ctx); /* ctx may be NULL: last create-time error of this thread */

/* ---- PeriodicBoundary::update (src/PeriodicBoundary.cpp:31-101): volume = det(basis), reciprocal = inverse,
 *      cutoff = half the shortest lattice vector (31^3 search).  Pure host helper. */
int mpmc_pbc_compute(const double basis[9], double reciprocal[9], double *volume, double *cutoff);

/* ---- context lifetime (one per System; replaces the per-System pair lists of src/System.Pairs.cpp:21) ---- */
/*      max_atoms is a capacity hint: mpmc_set_atoms with more atoms (insertions in the uVT / Gibbs ensembles) rebuilds the device
 *      buffers behind the same handle with 25 % headroom. */
int mpmc_ctx_create(int device, int max_atoms, mpmc_ctx **out);
int mpmc_ctx_destroy(mpmc_ctx *ctx);

/* System::update_pbc (src/System.cpp:859-876): basis = pbc.basis[q][p] row-major (rows are lattice vectors).
 * reciprocal/volume/cutoff may be NULL/0 to have the library compute them exactly like PeriodicBoundary. */
int mpmc_set_box(mpmc_ctx *ctx, const double basis[9], const double *reciprocal, double volume, double cutoff);
int mpmc_set_options(mpmc_ctx *ctx, const mpmc_options *opts);
void mpmc_default_options(mpmc_options *opts); /* reference defaults (src/System.h:21-24,510-831) */

/* Flattened atom list in atom_array order (src/System.cpp:881-904).  mol_id: equal ids = same Molecule.
 * frozen: Atom::frozen.  has_disp: nonzero iff any of c6,c8,c10 != 0 (only enters the rd_excluded test,
 * src/System.cpp:1050-1056); may be NULL.  mass may be NULL (only needed by mpmc_update_com). */
int mpmc_set_atoms(mpmc_ctx *ctx, int n, const double *pos /*[n][3]*/, const double *charge, const double *polarizability,
                   const double *epsilon, const double *sigma, const int32_t *mol_id, const int32_t *frozen,
                   const int32_t *has_disp, const double *mass);
/* after a Monte Carlo move: overwrite positions of atoms [first, first+count) */
int mpmc_update_positions(mpmc_ctx *ctx, int first, int count, const double *pos /*[count][3]*/);
/* same, positions already in device memory ([n][3] fp64, e.g. a torch tensor's data_ptr).  The library reads them on the context's own
 * stream, which is not ordered against the caller's streams: the data must be complete when this is called (synchronize the stream that
 * produced it first), and it is consumed before the call returns. */
int mpmc_set_positions_device(mpmc_ctx *ctx, const double *pos_device /*[n][3]*/);

/* ---- double System::energy() -------------------------------------------------------------------------- */
int mpmc_energy(mpmc_ctx *ctx, mpmc_result *out);
/* asynchronous pair: enqueue on the context's stream / wait + fetch (lets a caller overlap beads) */
int mpmc_energy_async(mpmc_ctx *ctx);
int mpmc_energy_wait(mpmc_ctx *ctx, mpmc_result *out);
/* (ABI 5) a scheduling hint for mpmc_energy_async: how many evaluations the caller keeps in flight together with this context's (the P
 * beads of PI_calculate_potential, PathIntegral.cpp:772-779).  With four or more the evaluation runs on ONE stream -- other evaluations
 * fill the device, and the side stream's fork and join only cost (+1 to 2 % evaluations/s with 8-32 in flight) --, alone it forks the
 * side stream for the reciprocal-space work (1.5 % faster).  Never changes a result; mpmc_energy() resets it to 1, the mpmc_pi_* loops set it
 * themselves. */
int mpmc_hint_in_flight(mpmc_ctx *ctx, int n_evaluations);

/* ---- trial moves: the device-side counterpart of the reference's per-pair cache ----------------------------------------
 * The reference re-evaluates only the pairs whose displacement changed (Pair::recalculate_energy, src/System.cpp:1211-1224,
 * src/System.Energy.cpp:925,1484).  After a full mpmc_energy() of the accepted configuration:
 *   mpmc_trial_begin  : atoms [first, first+count) (original order; typically one molecule) get trial positions
 *   mpmc_trial_energy : energy of the trial configuration.  Non-polarizable boxes: O(count * N) pair terms (old and new
 *                       geometry of every pair that involves a moved atom) + O(K * count) structure-factor update, added
 *                       to the accepted totals.  Polarizable boxes: a full evaluation (the dipole solve is global).
 *   mpmc_trial_accept : the trial configuration becomes the accepted one / mpmc_trial_reject : it is discarded.
 * A full mpmc_energy() at any time re-bases the totals (the reference's flag_all_pairs, src/System.cpp:1284).
 * Trial positions equal to the accepted ones cost nothing: the trial totals are the accepted totals, no kernel runs. */
#define MPMC_TRIAL_MAX_ATOMS 256
int mpmc_trial_begin(mpmc_ctx *ctx, int first, int count, const double *new_pos /*[count][3]*/);
int mpmc_trial_energy(mpmc_ctx *ctx, mpmc_result *out);
/* the same in two halves (enqueue on the context's stream / wait): the P images of one path-integral move overlap on the device */
int mpmc_trial_energy_async(mpmc_ctx *ctx);
int mpmc_trial_energy_wait(mpmc_ctx *ctx, mpmc_result *out);
int mpmc_trial_accept(mpmc_ctx *ctx);
int mpmc_trial_reject(mpmc_ctx *ctx);

/* ---- public component entry points of the reference (src/System.h:346-402), for parity tests ----------- */
int mpmc_lj(mpmc_ctx *ctx, double *out);                  /* System::lj()                   */
int mpmc_coulombic(mpmc_ctx *ctx, double *out);           /* System::coulombic()            */
int mpmc_coulombic_real(mpmc_ctx *ctx, double *out);      /* System::coulombic_real()       */
int mpmc_coulombic_reciprocal(mpmc_ctx *ctx, double *out);/* System::coulombic_reciprocal() */
int mpmc_coulombic_self(mpmc_ctx *ctx, double *out);      /* System::coulombic_self()       */
int mpmc_polar(mpmc_ctx *ctx, double *out);               /* System::polar()                */
int mpmc_thole_field(mpmc_ctx *ctx, double *ef_static /*[n][3] host, may be NULL*/); /* System::thole_field() */
/* System::thole_amatrix(): fills rows [row0, row0+nrows) of the dense 3N x 3N matrix into `a` (host, row-major
 * nrows x 3N).  Diagonal 1/alpha (1e40 when alpha == 0), off-diagonal blocks as src/System.Energy.cpp:2744-2764. */
int mpmc_thole_amatrix(mpmc_ctx *ctx, int row0, int nrows, double *a);

/* per-atom results written back by energy() in the reference (src/Atom.h:41-47); any pointer may be NULL */
int mpmc_get_dipoles(mpmc_ctx *ctx, double *mu, double *ef_static, double *ef_induced /* each [n][3] */);
/* pairs() tail: update_com + wrap_all (src/System.cpp:1347-1425).  Host-side O(N); needs mass in set_atoms.
 * com / wrapped_com: [n_molecules][3]; wrapped_pos: [n][3]; any may be NULL. */
int mpmc_update_com(mpmc_ctx *ctx, double *com, double *wrapped_com, double *wrapped_pos, int *n_molecules);

/* ---- SimulationControl::PI_calculate_potential (PathIntegral.cpp:752-805) ------------------------------ */
/* Evaluates energy() on the n_local beads owned by this process (all enqueued before any wait, one stream per
 * bead) and returns the UN-normalised ordered sums {rd, coulombic, polarization, vdw} over those beads in
 * sums4.  per_bead (may be NULL) receives n_local mpmc_result.  The cross-rank combine (4 fp64 all-reduce over
 * RCCL / MPI_Allgather in the reference, :763-766) is the caller's; mpmc_pi_finish divides by P. */
int mpmc_pi_potential_local(mpmc_ctx **beads, int n_local, double sums4[4], mpmc_result *per_bead, int *any_iterator_failed);
/* The same, with every bead's coordinates handed over in HOST memory (pos[b]: n x 3 doubles in the caller's atom order -- what the
 * reference's bead loop holds, PathIntegral.cpp:759-775): bead b's upload is followed at once by its enqueue, so the uploads overlap
 * the evaluations of the beads in front of them instead of standing in front of the whole step. */
int mpmc_pi_potential_local_host(mpmc_ctx **beads, int n_local, const double *const *pos, double sums4[4], mpmc_result *per_bead,
                                 int *any_iterator_failed);
/* systems that shared each launch of the dipole iterations in this context's last evaluation: always 1 since ABI 4 (every bead runs
 * on its own streams; the lockstep form of rounds 1-2 was measured slower and removed).  Kept so that ABI 3 callers still link. */
int mpmc_last_batch_size(mpmc_ctx *ctx);
/* obs = sums / P ; returns V = rd + coulombic + vdw + polarization (:786-804) */
double mpmc_pi_finish(const double sums4_global[4], int P, double obs4[4]);

/* ---- the cross-GPU exchange of PI_calculate_potential on RCCL over xGMI ------------------------------------------
 * Reference: MPI_Allgather x 4 of one double per rank, then the ordered sum s = 0..P-1 (PathIntegral.cpp:763-766, :786-801).
 * Here: ONE ncclAllGather of `stride` fp64 per bead, then the same ordered sum on the host (bit-identical on every rank).
 * Bead s lives on rank s % n_ranks, local slot s / n_ranks.  RCCL is opened with dlopen at first use (mpmc_rccl_library_path);
 * without it these calls fail with MPMC_ERR_COMM and nothing else in the library is affected.
 *   one process per GPU : rank 0 calls mpmc_comm_unique_id, the host program carries the 128 bytes to the other ranks
 *                         (MPI_Bcast, a file, the torch.distributed store), every rank calls mpmc_comm_init_rank;
 *   one process, G GPUs : mpmc_comm_init_all / mpmc_pi_allreduce. */
typedef struct mpmc_comm mpmc_comm;
#define MPMC_COMM_ID_BYTES 128
int mpmc_rccl_version(int *version); /* ncclGetVersion: e.g. 22707 */
/* (ABI 6) the file the RCCL entry points were resolved from and why that copy, e.g. "/opt/rocm/lib/librccl.so.1 (next to the bound
 * libamdhip64)"; "" when RCCL could not be opened.  Order: $MPMC_RCCL_LIB, a librccl.so.1 the host program already mapped (shared, not
 * duplicated), the copy next to the HIP runtime this library is bound to, the loader's search path; always RTLD_LOCAL. */
const char *mpmc_rccl_library_path(void);
int mpmc_comm_unique_id(char id[MPMC_COMM_ID_BYTES]);
int mpmc_comm_init_rank(mpmc_comm **out, int n_ranks, int rank, const char id[MPMC_COMM_ID_BYTES], int device);
int mpmc_comm_init_all(mpmc_comm **out, int n_devices, const int *devices /* NULL: 0..n_devices-1 */);
int mpmc_comm_destroy(mpmc_comm *comm);
int mpmc_comm_info(const mpmc_comm *comm, int *n_ranks, int *rank, int *n_local_devices);
const char *mpmc_comm_last_error(const mpmc_comm *comm); /* comm may be NULL: last error of this thread */
/* all[r][0..count) = rank r's local[0..count)   (communicators of mpmc_comm_init_rank) */
int mpmc_comm_allgather_f64(mpmc_comm *comm, const double *local, int64_t count, double *all /*[n_ranks][count]*/);
/* local[n_local][stride] in local-slot order  ->  all[P][stride] in BEAD order, P = n_local * n_ranks.  stride 4 = the
 * {rd, coulombic, polarization, vdw} of PI_calculate_potential; 3 * n_molecules = the centres of mass of the kinetic estimator. */
int mpmc_pi_gather_beads(mpmc_comm *comm, const double *local, int n_local, int stride, double *all);
/* One process driving several GPUs: beads[b] may live on any device (the usual placement is b mod G).  Evaluates energy() on every
 * bead -- one host thread per device, all of a device's beads enqueued before the first wait -- gathers the per-bead values over a
 * process-wide ncclCommInitAll communicator of the devices involved, and returns the UN-normalised ordered sums over beads 0..n-1
 * (identical on every device: an all-reduce with a fixed summation order).  mpmc_pi_finish divides by P. */
int mpmc_pi_allreduce(mpmc_ctx **beads, int n_beads, double sums4[4], mpmc_result *per_bead, int *any_iterator_failed);
/* (ABI 5) what mpmc_pi_allreduce uses for these beads: the number of distinct devices they live on and the size of the process-wide
 * communicator of those devices (0 before the first mpmc_pi_allreduce on them) -- a host program's proof that RCCL saw G ranks. */
int mpmc_pi_allreduce_info(mpmc_ctx **beads, int n_beads, int *n_devices, int *comm_n_ranks);

/* ---- Gibbs ensemble: the two boxes of SimulationControl::Gibbs_mc -------------------------------------------------------
 * Reference: final_energy[0] = systems[0]->energy(); final_energy[1] = systems[1]->energy(); (src/SimulationControl.Gibbs.cpp:179-180),
 * then boltzmann_factor_NVT_Gibbs (:358-522).  The boxes are independent evaluations: box 0 -> device 0, box 1 -> device 1 of the node
 * (mpmc_ctx_create's device argument); mpmc_gibbs_energy enqueues both before it waits for either. */
int mpmc_gibbs_energy(mpmc_ctx *box_a, mpmc_ctx *box_b, mpmc_result *out_a, mpmc_result *out_b);
/* move types as the reference enumerates them (src/constants.h:87-95) */
#define MPMC_MOVETYPE_INSERT 0
#define MPMC_MOVETYPE_REMOVE 1
#define MPMC_MOVETYPE_DISPLACE 2
#define MPMC_MOVETYPE_ADIABATIC 3
#define MPMC_MOVETYPE_SPINFLIP 4
#define MPMC_MOVETYPE_VOLUME 5
#define MPMC_MOVETYPE_PERTURB_BEADS 6
typedef struct mpmc_gibbs_move {
	int32_t movetype[2];       /* sys[i]->checkpoint->movetype                                                  */
	double temperature;        /* sys[0]->temperature                                                           */
	double init_energy[2];     /* energies of the accepted configuration                                        */
	double final_energy[2];    /* systems[i]->energy() after the move (may be non-finite: bad contact)          */
	double N[2], volume[2];    /* sys[i]->observables->N / ->volume AFTER the move                              */
	double checkpoint_volume_0; /* sys[0]->checkpoint->observables->volume (the volume the move started from)   */
} mpmc_gibbs_move;
/* boltzmann_factor[i] = sys[i]->nodestats->boltzmann_factor; energy[i] (may be NULL) receives MAXVALUE where the reference overwrites
 * observables->energy on a bad contact.  Entries the reference leaves untouched are left untouched.  Returns MPMC_OK, or the
 * reference's throw codes 20000 / 102, or MPMC_ERR_UNSUPPORTED for spin-flip moves (quantum rotation is outside the path). */
int mpmc_gibbs_boltzmann_factor(const mpmc_gibbs_move *move, double boltzmann_factor[2], double energy[2]);

/* ---- SimulationControl::PI_calculate_kinetic (PathIntegral.cpp:806-824) and its chain measure (:851-965) ----
 * Host-side O(P * n_molecules); no device work.  com: centres of mass [P][n_molecules][3] of the P images of every
 * molecule (Molecule::update_COM, src/Molecule.cpp:259-281; mpmc_update_com returns one bead's block),
 * mol_mass[n_molecules] = Molecule::mass of image 0, movable[m] != 0 iff image 0 of molecule m is neither frozen,
 * adiabatic nor target (:881).  Returns sum_m M_m * AMU2KG * 1e-20 * sum_i |com_i - com_{(i+1)%P}|^2  [kg m^2]. */
double mpmc_pi_chain_mass_length2(int P, int n_molecules, const double *com, const double *mol_mass, const int32_t *movable);
/* K [Kelvin] = (1/kB) * (0.5*3*N*kB*T*P - 0.5*omega2*chain_mass_len2), omega2 = P / (beta^2 hBar2)  (Tuckerman 12.5.12);
 * N = System::countN() of image 0. */
double mpmc_pi_kinetic(double chain_mass_len2, double orient_mu_len2, double N, int P, double temperature);

/* ---- measurement ------------------------------------------------------------------------------------- */
int mpmc_set_profiling(mpmc_ctx *ctx, int enabled); /* HIP-event timing of each kernel class on ctx's stream */
int mpmc_get_timings(mpmc_ctx *ctx, mpmc_timings *out, int reset);
int mpmc_synchronize(mpmc_ctx *ctx);
/* bytes of device memory held by the context and by its Thole tensor store */
int mpmc_memory_usage(mpmc_ctx *ctx, int64_t *total_bytes, int64_t *tensor_store_bytes);

/* tile-pair statistics of the LAST evaluation (64 x 64 atom tiles; orthorhombic cells are classified by the
 * minimum-image distance between tile bounding boxes): out4 = { tile pairs, pairs whose Thole tensors are stored and
 * streamed (64 KiB each), pairs beyond the damping range (bare dipole tensor recomputed), pairs wholly beyond the cutoff } */
int mpmc_get_tile_stats(mpmc_ctx *ctx, int64_t out4[4]);

/* ---- diagnostics (no effect on results) -------------------------------------------------------------------
 * How this context's host-side waits ended since it was created: out4 = { polls of the pinned result block that saw the device's post,
 * polls that ran out of their budget (the wait then synchronised the stream), stream synchronisations, yields taken inside long polls }.
 * Short evaluations and trial moves are polled for (a few microseconds earlier than the driver's completion path); a host with fewer
 * free cores than polling threads shows up here as timeouts and yields. */
int mpmc_debug_wait_counters(mpmc_ctx *ctx, long long out4[4]);
/* Measurement / A-B switch, key = value (keys: csrc/context.cpp, struct mpmc_tuning).  ctx == NULL: the default of contexts created
 * afterwards in this process.  The library reads no environment variable for any of these. */
int mpmc_debug_configure(mpmc_ctx *ctx, const char *key, double value);

#ifdef __cplusplus
}
#endif
#endif /* MPMC_ENERGY_H */
