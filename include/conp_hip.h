/* =============================================================================================
 * conp_hip.h -- C ABI of libconp_hip.so: the MI355X-native constant-potential charge solver that
 * drops into LAMMPS behind the `fix conp` / KSpaceModule surface of srtee/lammps-USER-CONP2.
 *
 * Boundary rules
 *   - plain C: opaque handle, scalars, caller-owned pointers + sizes; no C++/torch types;
 *   - every function returns 0 on success or a negative conp_status; the message is available from
 *     conp_last_error() (thread-local).  The reference aborts through error->all(FLERR,msg)
 *     (fix_conp.cpp:86,107,127,...): the LAMMPS glue turns a non-zero status into exactly that call;
 *   - calls are synchronous with respect to the host thread unless the name ends in _async/_device;
 *     internally everything is ordered on one HIP stream (conp_fix_set_stream);
 *   - the library never falls back to a CPU path: without a usable gfx950 device every compute
 *     entry point fails with CONP_ERR_NO_DEVICE.
 *
 * Each entry point cites the reference interface it replaces (file:line in /root/reference).
 * INTEGRATION.md shows the binding a maintainer adds to fix_conp.cpp / kspacemodule.h.
 * ===========================================================================================*/
#ifndef CONP_HIP_H
#define CONP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CONP_ABI_VERSION 1

typedef enum {
  CONP_OK = 0,
  CONP_ERR_ARG = -1,        /* illegal fix command / argument (fix_conp.cpp:86,107,127,143,...) */
  CONP_ERR_STATE = -2,      /* call out of order (e.g. pre_force before setup) */
  CONP_ERR_NO_DEVICE = -3,  /* no gfx950 device / HIP runtime failure */
  CONP_ERR_NUMERIC = -4,    /* "Inversion failed!" (fix_conp.cpp:956) */
  CONP_ERR_IO = -5          /* A-matrix file problems (fix_conp.cpp:737,745) */
} conp_status;

enum { CONP_FF_NORMAL = 0, CONP_FF_FFIELD = 1, CONP_FF_NOSLAB = 2 };   /* fix_conp.cpp:68 */
enum { CONP_SOLVER_CG = 0, CONP_SOLVER_INV = 1 };                      /* fix_conp.cpp:67 */

typedef struct conp_fix conp_fix; /* opaque: one `fix conp` instance on one GPU */

/* ---- parsed `fix ID group1 conp Nevery group2 eta DV logfile [keywords]` (fix_conp.cpp:79-201) ---- */
typedef struct {
  int everynum;            /* arg[3]  :102 */
  double eta;              /* arg[5]  :110 */
  double potdiff;          /* arg[6] if numeric :116 */
  int potdiff_is_variable; /* arg[6] = v_name :112-114 (the glue evaluates the variable each step :1143) */
  int ff_flag;             /* ffield / noslab :126-133 */
  int zneutr, matout, pppm, split, qinit, lowmem, nullneutral, ehgo; /* :160-169 */
  int a_matrix_f;          /* 0 none, 1 org, 2 inv :134-145 */
  char a_matrix_file[512];
  int smartlist, eletypenum, eletypes[32]; /* etypes :146-159 */
  int minimizer, maxiter;  /* new keywords `cg [maxiter N] [tol X]`; defaults 1 / 100 (:88-90) */
  double tolerance;        /* default 1e-6 (:89) */
  char logfile[512];       /* arg[7] :119 */
  char group2[128];        /* arg[4] :104 */
  char potdiff_var[128];   /* name after "v_" when potdiff_is_variable (potdiffstr :113) */
  int conq;                /* 1 when arg[2] names the conq style (fix_conq.h:21): DV is then the prescribed charge QR */
  int cond;                /* 1 when arg[2] names the cond style (fix_cond.h): DV is the prescribed charge, potential from the cell dipole */
} conp_fix_args;

/* Parses arg[3..narg-1] of the fix command exactly like the reference constructor (same keywords, same
 * error conditions; unknown keyword -> CONP_ERR_ARG with the reference's message).  ntypes bounds etypes. */
int conp_parse_fix_args(int narg, const char *const *arg, int ntypes, conp_fix_args *out);

/* ---- constants LAMMPS supplies (force->, domain->, kspace->; SURVEY.md 8b "inputs crossing") ---- */
typedef struct {
  double qqrd2e, qqr2e, qe2f, dielectric;         /* force-> (km_ewald.cpp:79, fix_conp.cpp:412) */
  int newton_pair;                                /* force->newton_pair (fix_conp.cpp:198) */
  double g_ewald, accuracy, slab_volfactor;       /* force->kspace-> (km_ewald.cpp:66-69); accuracy ABSOLUTE */
  int slabflag;
  double xprd, yprd, zprd, boxlo_z;               /* domain-> (km_ewald.cpp:81-83, fix_conp.cpp:616-619) */
  double boxlo_x, boxlo_y;                        /* domain->boxlo (pppm_conp.cpp:146-148; only the pppm path needs x, y) */
  int ntypes;
  const double *cutsq;                            /* coulpair->cutsq flattened [(ntypes+1)*(ntypes+1)] (fix_conp.cpp:1235) */
  double cut_coul;                                /* *coulpair->extract("cut_coul") (fix_conp.cpp:1237) */
  int one_electrode;                              /* groupbit == jgroupbit (fix_conp.cpp:295) */
  int device;                                     /* HIP device ordinal of this rank; -1: rank modulo the visible devices;
                                                     -(2 + l): node-local rank l modulo the visible devices (MPI hosts) */
  int rank, nranks;                               /* shard id for the multi-GPU path (section "sharding") */
  /* `pppm` keyword (fix_conp.cpp:162, 401-404): mesh and stencil order of the pppm/conp kspace style, i.e. LAMMPS PPPM's
   * nx_pppm, ny_pppm, nz_pppm, order (pppm_conp.cpp:242, 206); ignored without the keyword */
  int pppm_nx, pppm_ny, pppm_nz, pppm_order;
  /* 1: every ghost atom is a periodic image of an owned atom and moves with it -- what LAMMPS' forward communication
   * guarantees between re-neighbourings (orthogonal box).  The host-buffer hooks then upload the owned x, q only and rebuild
   * the ghosts on the device as x_owner + n*prd (the arithmetic of Comm's pack_comm: same bits).  Checked at every
   * (setup_)post_neighbor; a ghost that is not an exact image switches the handle back to full uploads.  0: x, q of ghosts
   * are read from the caller's arrays at every hook, like the reference reads atom->x. */
  int ghost_images;
} conp_env;

/* FixConp::FixConp + FixConp::init (fix_conp.cpp:79-201, 245-300) */
int conp_fix_create(const conp_fix_args *args, const conp_env *env, conp_fix **out);
void conp_fix_destroy(conp_fix *fix);   /* FixConp::~FixConp :205-229 */
const char *conp_last_error(void);
int conp_abi_version(void);

/* FixConp::modify_param (fix_conp.cpp:1482-1515): `fix_modify ID ehgo kappa X` and `fix_modify ID ehgo coeff types eta u0|auto`
 * (arg[0] = "ehgo").  Only valid with the `ehgo` keyword; *consumed = number of arguments used (3 or 5), like the reference's
 * return value.  The per-type tables are finalised at setup (ehgo_setup_tables :1517-1559); without any coefficient the fix falls
 * back to the plain eta model, as the reference does (with its warning text in conp_last_error()). */
int conp_fix_modify_param(conp_fix *fix, int narg, const char *const *arg, int *consumed);

/* ---- LAMMPS-owned per-atom arrays, re-fetched before every hook (may be reallocated on re-neighbour) ---- */
typedef struct {
  int nlocal, nghost;
  const double *x;     /* atom->x flattened [nall][3] */
  double *q;           /* atom->q [nall]; electrode entries (owned + ghost) are overwritten by pre_force */
  const int *type;     /* atom->type [nall] */
  const int *tag;      /* atom->tag [nall] (tagint == int, fix_conp.cpp:474) */
  const int *echeck;   /* electrode_check(i) for i < nall: +1 group1, -1 group2, 0 else (fix_conp.cpp:599-605) */
} conp_atoms;

/* LAMMPS half neighbour list (NeighList inum/ilist/numneigh/firstneigh) with firstneigh flattened by the glue:
 * neighbours of i are neigh[first[i] .. first[i]+numneigh[i]); entries may carry special-bond bits (NEIGHMASK). */
typedef struct {
  int inum;
  const int *ilist;     /* [inum] */
  const int *numneigh;  /* [nall] */
  const int *first;     /* [nall] */
  const int *neigh;
  int64_t nneigh;       /* length of neigh */
} conp_neighlist;

/* FixConp::init_list (fix_conp.cpp:365-378): which = 0 alist (ele-ele, occasional), 1 blist (ele-elyte, perpetual),
 * 2 both (generic list without etypes).  The pointers must stay valid until the next hook returns. */
int conp_fix_init_list(conp_fix *fix, int which, const conp_neighlist *list);

/* ---- Fix hooks ---- */
int conp_fix_setup_post_neighbor(conp_fix *fix, const conp_atoms *atoms); /* fix_conp.cpp:382-385 (linalg_init + post_neighbor) */
int conp_fix_setup_pre_force(conp_fix *fix, const conp_atoms *atoms, int64_t ntimestep, double potdiff); /* :387-391 */
int conp_fix_post_neighbor(conp_fix *fix, const conp_atoms *atoms);       /* :468-539 */
int conp_fix_pre_force(conp_fix *fix, const conp_atoms *atoms, int64_t ntimestep, double potdiff); /* :543-573 */
double conp_fix_compute_scalar(const conp_fix *fix);                      /* :592-595 */
/* FixConp::post_force -> force_cal (fix_conp.cpp:577-580, 1163-1201) + blist_coul_cal_post_force (:1368-1444), ETA pair mode:
 * adds the real-space Gaussian-correction forces to f[nall][3] (host, accumulated like atom->f) and returns what the reference
 * adds to force->kspace->energy (Gaussian self energy) and, through Pair::ev_tally, to eng_coul and the global virial
 * (xx,yy,zz,xy,xz,yz).  The reference's arithmetic is kept as written, including `del*forcecoul` and the `eta^2 r^2 < 5.8` gate. */
int conp_fix_post_force(conp_fix *fix, const conp_atoms *atoms, double *f, double *kspace_energy_add, double *eng_coul_add,
                        double *virial_add /*[6]*/);
/* The same for a host that can say which step it is in: when `ntimestep` is the step of the last conp_fix_pre_force that wrote
 * charges and `atoms->x` is the array handed over there, positions and charges are already on the device (nothing moves between
 * pre_force and post_force of a LAMMPS step) and are not uploaded again.  Any other step (Nevery > 1, a re-neighbour) uploads. */
int conp_fix_post_force_step(conp_fix *fix, const conp_atoms *atoms, int64_t ntimestep, double *f, double *kspace_energy_add,
                             double *eng_coul_add, double *virial_add /*[6]*/);

/* finer-grained pieces of the same path (same names as the reference's methods) */
int conp_fix_linalg_setup(conp_fix *fix, const conp_atoms *atoms);        /* :426-464 a_cal, b_setq_cal, equation_solve, get_setq */
int conp_fix_a_cal(conp_fix *fix, const conp_atoms *atoms);               /* :777-861 */
int conp_fix_b_cal(conp_fix *fix, const conp_atoms *atoms);               /* :677-695 */
int conp_fix_equation_solve(conp_fix *fix);                               /* :698-718 */
int conp_fix_update_charge(conp_fix *fix, const conp_atoms *atoms, double potdiff); /* :1120-1161 */

/* ---- several MPI ranks (LAMMPS spatial decomposition) ------------------------------------------------------------------
 * The reference runs on N ranks: every rank owns the atoms of its sub-domain, and FixConp / KSpaceModuleEwald exchange through
 * MPI_Allreduce (maxtag fix_conp.cpp:415, newtonbuf :1356, structure factors km_ewald.cpp:784-785, sum q^2 :77),
 * MPI_Allgather / MPI_Allgatherv (elenum_list :492, eleall2tag :523, elebuf2eleall :535, b_comm :643, A rows :822).
 * The library makes the same exchanges through callbacks the host supplies (the glue implements them with MPI on `world`,
 * lammps_glue/fix_conp_hip.cpp), so that it needs no MPI itself.  After conp_fix_set_comm the atoms and lists handed to the
 * hooks are THIS RANK's (owned + ghost); the library
 *   - numbers the electrode atoms globally like post_neighbor does (rank-major, :492-525),
 *   - all-gathers the charged electrolyte atoms' (x, q) every update and computes its shard of the k-vectors for ALL electrode
 *     rows, adds its own real-space rows, all-reduces b, solves its rows, all-gathers q,
 *   - shards the once-per-run A build by tiles and all-reduces the matrix before the (replicated, :947-949) inverse.
 * Every callback returns 0 on success.  All ranks must enter every hook together (as with the reference's collectives). */
typedef struct {
  void *ctx;
  int rank, nranks;
  int (*allreduce_sum)(void *ctx, double *buf, int64_t n);                      /* in place, MPI_SUM, MPI_DOUBLE */
  int (*allreduce_max_int)(void *ctx, int *buf, int n);                         /* in place, MPI_MAX, MPI_INT */
  int (*allgather_int)(void *ctx, int value, int *out /*[nranks]*/);            /* MPI_Allgather of one int */
  /* MPI_Allgatherv of bytes: rank r contributes counts[r] bytes, stored at recv + displs[r] */
  int (*allgatherv)(void *ctx, const void *send, int64_t nbytes, void *recv, const int64_t *counts, const int64_t *displs);
} conp_comm;
/* call between conp_fix_create and conp_fix_setup_post_neighbor; overrides conp_env.rank / nranks */
int conp_fix_set_comm(conp_fix *fix, const conp_comm *comm);

/* Data plane on RCCL (one rank per GPU): the all-reduce of b and the all-gather of q of a device-resident update, and the
 * all-reduce of the sharded A build, run inside the library on its own stream over xGMI.  Rank 0 makes an id with
 * conp_rccl_unique_id, the host distributes the 128 bytes (MPI_Bcast, torch.distributed ...), every rank calls
 * conp_fix_comm_init_rccl before setup.  Without it a multi-rank handle uses the conp_comm callbacks (host buffers), or leaves
 * the two collectives to the caller (conp_fix_b_cal_device / _solve_device / _scatter_device). */
#define CONP_RCCL_ID_BYTES 128
int conp_rccl_unique_id(void *id_out /*[CONP_RCCL_ID_BYTES]*/);
int conp_fix_comm_init_rccl(conp_fix *fix, const void *id /*[CONP_RCCL_ID_BYTES]*/);
/* conp_rccl_available: 0 when librccl loads with every entry point the library calls -- local and cheap; ranks agree on it (MIN over
 * ranks) BEFORE the collective conp_fix_comm_init_rccl, so that a rank without RCCL cannot strand its partners inside
 * ncclCommInitRank.  conp_fix_comm_destroy_rccl: all ranks together give the communicator back (an initialisation that failed
 * somewhere: the host then makes the two exchanges itself). */
int conp_rccl_available(void);
int conp_fix_comm_destroy_rccl(conp_fix *fix);

/* ---- KSpaceModule provider surface (kspacemodule.h:30-40), Ewald provider (km_ewald.cpp) ---- */
int conp_km_conp_setup(conp_fix *fix, double qsqsum, int64_t natoms);     /* km_ewald.cpp:63-132 (qsqsum: Allreduce'd sum q^2 :72-78) */
int conp_km_a_cal(conp_fix *fix, const conp_atoms *atoms, double *aaa /*[Ne*Ne], host, overwritten: k-space part only*/); /* :147-151 */
int conp_km_b_cal(conp_fix *fix, const conp_atoms *atoms, double *bbb /*[Ne] eleall order, host*/);                          /* :153-167 */

/* ---- PPPMCONP beyond the b vector (`pppm` keyword; pppm_conp.cpp:385-534), SURVEY 8f-2 --------------------------------------
 * All three take the atoms as they are NOW (after pre_force wrote the electrode charges) and work on the handle's mesh
 * (conp_env.pppm_*).  Mesh arrays are [nz][ny][nx], periodic (one rank: LAMMPS' ghost planes folded in).
 *   conp_pppm_make_rho            : ele_make_rho (:385-426) + the make_rho override (:434-450): density = electrolyte brick +
 *                                   electrode brick, what PPPMCONP hands to PPPM::compute instead of re-spreading every atom.
 *                                   Any output may be NULL.
 *   conp_pppm_compute_group_potential (:487-534): recv[i] = - sum over the order^3 stencil of w * u_brick for owned atoms with
 *                                   sel[i] != 0.  u_brick is what PPPM::compute leaves there when per-atom energies are
 *                                   tallied (ComputePotentialAtom insists on that step, compute_potential_atom.cpp:128-130):
 *                                   the mesh potential of the TOTAL density; the library forms it from the same bricks.
 *   conp_pppm_compute_particle_potential (:452-485): the same for atom i, plus 2 g_ewald q_i / sqrt(pi).  RANK-LOCAL: reads the
 *                                   cached mesh potential (one rank: forms it on demand; several ranks: CONP_ERR_STATE unless a
 *                                   collective entry formed it since the last update). */
int conp_pppm_make_rho(conp_fix *fix, const conp_atoms *atoms, double *density, double *ele_density, double *elyte_density);
/* PPPMCONP keeps the electrolyte brick of every b_cal for its make_rho override (pppm_conp.cpp:172-228, 434-450; elyte_mapped is
 * reset by conp_pre_force, pppm_conp.h:42).  on != 0: from the next b_cal on the brick of the update stays on the device and
 * conp_pppm_make_rho adds the fresh electrode brick to it instead of spreading the electrolyte a second time (one rank; under ranks
 * the bricks are re-made from one gather).  The call itself drops whatever is cached: the glue calls it from conp_pre_force(). */
int conp_pppm_keep_density(conp_fix *fix, int on);
/* The mesh potential of the total density -- what PPPM::compute leaves in u_brick when per-atom energies are tallied.  COLLECTIVE
 * under ranks.  Afterwards conp_pppm_compute_particle_potential is a rank-local stencil gather from the cached brick, like the
 * reference's (:452-485; compute_potential_atom.cpp:168-174 calls it a different number of times on every rank), until the next
 * update or re-neighbouring.  conp_pppm_compute_group_potential and conp_compute_potential_atom leave the same cache. */
int conp_pppm_compute(conp_fix *fix, const conp_atoms *atoms);
int conp_pppm_compute_group_potential(conp_fix *fix, const conp_atoms *atoms, const int *sel /*[nlocal]*/, double *recv /*[nlocal]*/);
int conp_pppm_compute_particle_potential(conp_fix *fix, const conp_atoms *atoms, int i, double *u);

/* ---- `compute potential/atom` (compute_potential_atom.cpp:120-345), SURVEY 8f-4 --------------------------------------------
 * per-atom electrostatic potential in volts: pair part over the pair style's half list (:223-308, optional Gaussian `eta`
 * correction for atoms with etasel != 0 = eta_check :313-318), k-space part through the PPPM provider (:165-175 -> the
 * particle potential above), slab correction (:323-345), times qqr2e / qe2f (:214).
 * sel[i] = mask[i] & groupbit for i < nlocal + nghost; potential has nlocal (+ nghost when newton_pair) entries. */
typedef struct {
  int pairflag, kspaceflag, qsumflag;   /* `pair` / `kspace` / not `noqsum` (:59-88) */
  double eta;                           /* 0: no `eta` keyword */
} conp_potential_args;
int conp_compute_potential_atom(conp_fix *fix, const conp_atoms *atoms, const conp_neighlist *pairlist, const int *sel,
                                const int *etasel /*NULL without eta*/, const conp_potential_args *args, double *potential);

/* ---- state read-back for parity tests and for the glue (public members fix_conp.h:58-85) ---- */
typedef struct {
  int elenum, elenum_all, elytenum, maxtag_all, runstage;
  int kcount, kcount_flat, kcount_expand, kxmax, kymax, kzmax, kmax, kmax3d;
  int kcount_dims[7];
  int cg_iterations;
  int n_zclasses;          /* distinct electrode z values when the planar fast path of the projection is active, else 0 */
  double unitk[3], volume, gsqmx, ug_tot, totsetq, scalar_output, totinve, slabcorr;
  int64_t n_blist_pairs, n_alist_pairs, n_elyte_charged;
  int inverse_path;        /* how the last inverse (fix_conp.cpp:947-949) was formed: 0 none yet, 1 positive-definite elimination (no
                              pivot search), 2 partial pivoting */
  int inverse_retries;     /* 1: the multi-workgroup pivot panel timed out at its grid barrier and the one-workgroup panel redid it */
  int pppm_elyte_spreads;  /* `pppm`: how often the electrolyte atoms have been spread onto the mesh so far (b_cal, density and potential queries) */
  int zn_cols, zn_grid, zn_rows; /* the z-window form of the structure-factor contraction (conp_zn.hip) is in use: window columns (32 / 48), z grid points,
                              G rows this rank contracts; all 0: the full kernels */
} conp_info;
int conp_fix_info(const conp_fix *fix, conp_info *out);
/* integer tables; pass NULL for those not wanted.  Sizes: kcount / kcount_expand */
int conp_fix_get_ktables(const conp_fix *fix, int *kxvecs, int *kyvecs, int *kzvecs, double *ug, int *kxy_list, int *kz_list);
/* maps: ele2tag[elenum] ele2eleall[elenum] eleall2tag[Ne] eleall2ele[Ne+1] elecheck_eleall[Ne] elebuf2eleall[Ne] tag2eleall[maxtag+1] */
int conp_fix_get_maps(const conp_fix *fix, int *ele2tag, int *ele2eleall, int *eleall2tag, int *eleall2ele,
                      int *elecheck_eleall, int *elebuf2eleall, int *tag2eleall);
int conp_fix_get_matrix(conp_fix *fix, double *aaa_all /*[Ne*Ne]*/);   /* A, or projected A^-1 after inv() */
int conp_fix_set_matrix(conp_fix *fix, const double *aaa_all, int runstage); /* a_read 'org'/'inv' path :721-773 (file parsing is the glue's) */
int conp_fix_get_vectors(conp_fix *fix, double *bbb_all, double *eleallq, double *elesetq); /* each [Ne] or NULL */
int conp_fix_get_sfac(conp_fix *fix, double *sfacrl, double *sfacim);  /* [kcount], reference k order (km_ewald.cpp:782-786) */
int conp_fix_get_ele_trig(conp_fix *fix, double *csk, double *snk);    /* [Ne][kcount_flat] (km_ewald.cpp:261-268 lowmem) */
/* inv_project on a caller-supplied matrix (bit-exact electroneutrality projection, fix_conp.cpp:982-1067) */
int conp_inv_project(conp_fix *fix, int n, double *aaa, int nullneutral, int zneutr, const double *eleallz, double zhalf,
                     double *totinve_out);

/* On-disk matrix formats of the reference.  write: which = 0 -> "amatrix" layout (fix_conp.cpp:833-849: a tag row
 * of %20d, then rows of %20.12f), which = 1 -> "inv_a_matrix" layout (:960-977: %20d tags, %20.10f entries); the current device
 * matrix is written.  read: FixConp::a_read (:721-773) for the `org F` / `inv F` keywords -- the first Ne tokens are the tags
 * that define the permanent electrode numbering, the following Ne*Ne tokens the matrix; errors "Too many entries in A matrix
 * file" / "Too few entries in A matrix file" as in the reference.  Call it between setup_post_neighbor and setup_pre_force
 * (it stands for a_cal); the keyword decides whether the inverse is still computed (org) or taken as given (inv). */
int conp_fix_write_matrix_file(conp_fix *fix, const char *path, int which);
int conp_fix_read_matrix_file(conp_fix *fix, const conp_atoms *atoms, const char *path);

/* the LU-quality inverse that stands where the reference calls dgetrf_/dgetri_ (fix_conp.cpp:947-949), on a caller-supplied
 * row-major n x n matrix (host pointer, overwritten).  CONP_ERR_NUMERIC ("Inversion failed!") on a singular matrix. */
int conp_invert(conp_fix *fix, int n, double *aaa);

/* ---- host-only logic, callable without a GPU (CPU unit tests of the integer contracts) ---------------------------------
 * conp_host_ktables: the k-vector tables of KSpaceModuleEwald::conp_setup (km_ewald.cpp:63-132, 285-424) for the given
 * parameters.  Call once with NULL arrays to get the counts in info[16] = {kcount, kcount_flat, kcount_expand, kxmax, kymax,
 * kzmax, kmax, kmax3d, kcount_dims[0..6], n_planar}, then with arrays of those sizes.  plan_* (optional) return the GPU plan:
 * per k the planar-vector index, kz index and sign. */
int conp_host_ktables(double g_ewald, double accuracy, double slab_volfactor, int slabflag, double xprd, double yprd, double zprd,
                      double qsqsum, int64_t natoms, double qqrd2e, double dielectric, int *info /*[16]*/, int *kxvecs,
                      int *kyvecs, int *kzvecs, double *ug, int *kxy_list, int *kz_list, int *plan_p, int *plan_m, int *plan_sign);
/* conp_host_index: FixConp::post_neighbor's maps (fix_conp.cpp:468-539) for a sequence of two neighbour builds: atoms as they
 * are at the first post_neighbor (tag0/echeck0, n0 owned atoms) and at a later one (tag1/echeck1, n1).  Outputs like
 * conp_fix_get_maps, for the state after the second call; sizes[4] = {elenum, elenum_all, elytenum, maxtag_all}. */
int conp_host_index(int n0, const int *tag0, const int *echeck0, int n1, const int *tag1, const int *echeck1, int *sizes,
                    int *ele2tag, int *ele2eleall, int *eleall2tag, int *eleall2ele, int *elebuf2eleall, int *tag2eleall);
/* conp_host_pair_rows: the electrode-row regrouping of a LAMMPS half list (which = 1: blist_coul_cal membership,
 * fix_conp.cpp:1313-1353; which = 0: alist_coul_cal, :1242-1276; which = 2: post-force pairs :1411).  Returns the number of
 * pairs; with non-NULL arrays fills row_ptr[Ne+1] (which 0/1), ele_atom/oth_atom[npairs], col[npairs] (which 0). */
int64_t conp_host_pair_rows(int which, const conp_neighlist *list, const conp_atoms *atoms, int newton, int *row_ptr,
                            int *ele_atom, int *oth_atom, int *col);

/* ---- device-resident operation (bench, GPU-resident MD engines, multi-GPU) -------------------------------------
 * x/q are DEVICE pointers with the same layout as conp_atoms.x/q; nothing crosses PCIe.  One charge update =
 *   conp_fix_b_cal_device  (this rank's shard of b into the bound b buffer: its k-shard for ALL rows + its rows of
 *                           the real-space term; slab term on rank 0) -> [caller: all-reduce b over ranks] ->
 *   conp_fix_solve_device  (rows [row0,row1) of q = S b (+ dV S d) into the bound q buffer) ->
 *                          [caller: all-gather q] -> conp_fix_scatter_device (q[i] for owned+ghost electrode atoms).
 * With nranks == 1 conp_fix_pre_force_device runs all three back to back.
 * Large planar systems take the z-window form of the structure-factor contraction (conp_info.zn_cols > 0): every electrolyte atom
 * must stay within 2.5 A (in z) of its position at the last conp_fix_post_neighbor -- LAMMPS re-neighbours long before that.  The
 * device-resident entries do not synchronise, so an atom that left its window is seen one call later: that call returns
 * CONP_ERR_NUMERIC (the charges of the updates since the list build are invalid; call conp_fix_post_neighbor and repeat), and the
 * handle uses the full kernels until the next list build.  conp_fix_pre_force (host arrays) repeats the update by itself. */
int conp_fix_set_stream(conp_fix *fix, void *hip_stream);
int conp_fix_bind_device_buffers(conp_fix *fix, double *d_b /*[Ne]*/, double *d_q /*[Ne]*/);
/* this rank's electrode rows: blocks of ceil(Ne / nranks) rows, so that rank r's rows start at r * ceil(Ne / nranks) */
int conp_fix_row_range(const conp_fix *fix, int *row0, int *row1);
int conp_fix_b_cal_device(conp_fix *fix, const double *d_x, const double *d_q);
int conp_fix_solve_device(conp_fix *fix, double potdiff);
int conp_fix_scatter_device(conp_fix *fix, double *d_q_atoms, double potdiff);
int conp_fix_pre_force_device(conp_fix *fix, const double *d_x, double *d_q, double potdiff);
/* per-kernel timing of the last N updates via HIP events on the library's stream (bench.py roofline leg).
 * enable: 0 off, 1 a pair of events around every kernel, 2 around every 4th launch of the dominant kernel (sk_gemm) only --
 * cheap enough to stay on inside a timed region (an event pair drains the queue around the kernel it brackets). */
int conp_fix_profile(conp_fix *fix, int enable);
int conp_fix_profile_read(conp_fix *fix, int *nkernels, const char **names /*[16]*/, double *avg_ms /*[16]*/, int *counts /*[16]*/);
/* ---- page-locked host memory (optional; no reference counterpart: the reference has no device) ----
 * The host-buffer hooks copy x, q (and, at a re-neighbour, the flattened neighbour list) out of the host's arrays.  Out of
 * PAGEABLE memory such a copy is staged by the runtime and blocks the calling thread (about 70 us for the 1.2 MB of x, q at the
 * headline size); out of page-locked memory it is an asynchronous DMA transfer that overlaps the enqueue of the update's kernels.
 *   conp_fix_pin_host_arrays   page-locks the host's OWN arrays in place: x [3 n] and q [n], n >= nlocal + nghost of the calls
 *       that follow.  The host promises that both stay where they are, and allocated, until conp_fix_unpin_host_arrays (or the
 *       handle's destruction).  LAMMPS: atom->x / atom->q move only when the per-atom arrays grow, which happens between
 *       pre_exchange and post_neighbor of a re-neighbouring step -- the glue unpins in pre_exchange and pins again in
 *       post_neighbor.  Arrays that were not pinned (or other pointers than the pinned ones) take the staged copy as before:
 *       same results either way.  A runtime that refuses the registration returns CONP_ERR_NO_DEVICE and changes nothing.
 *   conp_host_alloc / conp_host_free   page-locked memory for arrays the host BUILDS for the hooks (the glue's flattened
 *       neighbour list: conp_neighlist.first / .neigh). */
int conp_fix_pin_host_arrays(conp_fix *fix, const double *x, const double *q, int n);
int conp_fix_unpin_host_arrays(conp_fix *fix);
void *conp_host_alloc(size_t bytes);
void conp_host_free(void *p);

/* Diagnostic (no reference counterpart): with CONP_GUARD=1 in the environment every device buffer of the library sits between two
 * 4-KB zones of a known byte pattern; this reads all zones back.  Returns the number of damaged zones (0 = no kernel has stored
 * outside its buffers so far), -1 when guard zones are off; conp_last_error() names the damaged buffers. */
int conp_debug_check_guards(void);

/* Test hooks (no reference counterpart): alternative code paths of the SAME computation, results inside the parity tolerances of
 * DESIGN.md section 2 -- what tests/ compares the default paths with (A/B inside one process).  Process-wide bit mask, read when a
 * handle is created (CONP_PATH_ROWS_HOST, CONP_PATH_PPPM_SPREAD_LAUNCH: at every call).  The first handle created with a
 * non-zero mask says so on stderr.  conp_debug_set_sk_workgroups: workgroup count of the structure-factor launch (0 = the
 * library's choice), for the tests that cover heavily split tiles on a small deck.  Not meant for production runs. */
enum {
  CONP_PATH_PARTIAL_TILES = 1 << 0,      /* planar electrodes: partial tiles + reducing launch instead of the projecting epilogue */
  CONP_PATH_A_GENERAL = 1 << 1,          /* A k-space part: the general (planar, kz) contraction instead of the z-class one */
  CONP_PATH_INV_PIVOTED = 1 << 2,        /* inverse: pivoted elimination instead of the positive-definite path */
  CONP_PATH_CG_TWO_LAUNCH = 1 << 3,      /* CG: matvec + update launches per iteration instead of one */
  CONP_PATH_GEMV_ROWS = 1 << 4,          /* solve: row-by-row product also from 2048 electrode atoms up (no packed symmetric tiles) */
  CONP_PATH_PHASE_LAUNCH = 1 << 5,       /* small systems: stand-alone phase-table launch instead of the prologue inside sk_gemm */
  CONP_PATH_PPPM_SPREAD_LAUNCH = 1 << 6, /* pppm: density brick + spreading launch also for deck-sized systems */
  CONP_PATH_ROWS_HOST = 1 << 7,          /* re-neighbour: electrode rows regrouped on the host */
  CONP_PATH_TIME_SPLIT = 1 << 8,         /* host-buffer hooks: k-space and real-space halves of b_cal in launches of their own (timing log) */
  CONP_PATH_HC_NO_WAIT = 1 << 9,         /* fused pieces + dot launch: the dot workgroups do not wait for the handed-over class table, they add the pieces themselves */
  CONP_PATH_CG_PERSIST = 1 << 11,        /* CG: ONE persistent launch per solve with a grid barrier per iteration (measured slower than a launch per iteration: not the default) */
  CONP_PATH_SK_CLASSIC = 1 << 12,        /* large planar systems: sk_gemm over all kz columns instead of the z-window contraction (conp_zn.hip) */
  CONP_PATH_HC_FUSED = 1 << 10           /* pieces' sums + pair sums + dot as ONE launch with an in-launch hand-off (measured slower than two launches: not the default) */
};
void conp_debug_set_paths(unsigned mask);
void conp_debug_set_sk_workgroups(int n);


/* ---- the fix's log file (fix_conp.cpp:119 `outf`) ----
 * The library buffers the lines the reference prints there -- "A matrix calculating ..." / "A matrix calculation time  = %g"
 * (:787, :857), the CG "Iteration %d: res = %g" / "***** Converged at iteration ..." lines (:919-928) -- and the host appends
 * them to its file.  conp_fix_write_timing adds the three lines FixConp::pre_force prints on the last step of a run
 * (:553-568: B vector / Coulomb / Kspace calculation time, seconds accumulated over the host-buffer pre_force calls, measured
 * with HIP events around the k-space and real-space halves of b_cal).  conp_fix_log_drain returns the buffered text
 * (valid until the next call on this fix) and empties the buffer. */
int conp_fix_write_timing(conp_fix *fix);
const char *conp_fix_log_drain(conp_fix *fix);
/* same for the two lines the reference sends to utils::logmesg (screen + LAMMPS log): "conp output: <e,e> = %.8g" after the
 * inverse's row sums (fix_conp.cpp:1006-1009) and "conp output: <d,d> = %.8g" at the end of linalg_setup (:458-461) */
const char *conp_fix_mesg_drain(conp_fix *fix);

#ifdef __cplusplus
}
#endif
#endif /* CONP_HIP_H */
