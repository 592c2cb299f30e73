#include "kspacemodule_hip.h"

#include <cstring>

#ifdef CONP_GLUE_MOCK
#include "mock_lammps/mpi_mock.h"
#else
#include <mpi.h>

#include "atom.h"
#include "comm.h"
#include "domain.h"
#include "error.h"
#include "fix_conp.h"
#include "force.h"
#include "kspace.h"
#endif

#include "conp_mpi_comm.h"

using namespace LAMMPS_NS;

/* the same MPI-backed callbacks as fix_conp_hip.cpp (ctx = &world): conp_mpi_comm.h */
static int km_allreduce_sum(void *ctx, double *buf, int64_t n) { return conp_glue::cb_allreduce_sum(ctx, buf, n); }
static int km_allreduce_max_int(void *ctx, int *buf, int n) { return conp_glue::cb_allreduce_max_int(ctx, buf, n); }
static int km_allgather_int(void *ctx, int value, int *out) { return conp_glue::cb_allgather_int(ctx, value, out); }
static int km_allgatherv(void *ctx, const void *send, int64_t nbytes, void *recv, const int64_t *counts, const int64_t *displs) {
  return conp_glue::cb_allgatherv(ctx, send, nbytes, recv, counts, displs);
}

KSpaceModuleHip::KSpaceModuleHip(LAMMPS *lmp) : KSpaceModule(), Pointers(lmp), h(nullptr), first(true) {}

KSpaceModuleHip::~KSpaceModuleHip() { conp_fix_destroy(h); }

void KSpaceModuleHip::fail_if(int status) {
  if (status != CONP_OK) error->all(FLERR, conp_last_error());   // the reference's only error channel (fix_conp.cpp:86,...)
}

conp_atoms KSpaceModuleHip::view() {
  const int nall = atom->nlocal + atom->nghost;
  echeck.resize(nall);
  xflat.resize(3 * (size_t)nall);
  for (int i = 0; i < nall; ++i) {
    echeck[i] = fixconp->electrode_check(i);                     // fix_conp.cpp:599-605, through the registered fix
    for (int c = 0; c < 3; ++c) xflat[3 * (size_t)i + c] = atom->x[i][c];
  }
  conp_atoms at;
  at.nlocal = atom->nlocal; at.nghost = atom->nghost; at.x = xflat.data(); at.q = atom->q; at.type = atom->type;
  at.tag = atom->tag; at.echeck = echeck.data();
  return at;
}

/* km_ewald.cpp:63-132.  The fix was registered just before (fix_conp.cpp:409), so its public members are readable here. */
void KSpaceModuleHip::conp_setup(bool lowmem) {
  lowmemflag = lowmem;           /* phases are regenerated on the fly: no csk/snk[Ne][K] table choice to make (km_ewald.cpp:261-268) */
  if (fixconp == nullptr) error->all(FLERR, "KSpaceModuleHip: register_fix() must precede conp_setup()");
  if (h == nullptr) {
    conp_fix_args fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.everynum = 1; fa.eta = fixconp->eta; fa.minimizer = CONP_SOLVER_INV; fa.maxiter = 100; fa.tolerance = 1e-6;
    fa.lowmem = lowmem ? 1 : 0; fa.nullneutral = 1;
    conp_env env;
    std::memset(&env, 0, sizeof(env));
    env.qqrd2e = force->qqrd2e; env.qqr2e = force->qqr2e; env.qe2f = force->qe2f; env.dielectric = force->dielectric;
    env.newton_pair = force->newton_pair;
    env.g_ewald = force->kspace->g_ewald; env.accuracy = force->kspace->accuracy;        /* km_ewald.cpp:66-69 */
    env.slab_volfactor = force->kspace->slab_volfactor; env.slabflag = force->kspace->slabflag;
    env.xprd = domain->xprd; env.yprd = domain->yprd; env.zprd = domain->zprd;
    env.boxlo_x = domain->boxlo[0]; env.boxlo_y = domain->boxlo[1]; env.boxlo_z = domain->boxlo[2];
    env.ntypes = atom->ntypes;
    cutsq0.assign((size_t)(atom->ntypes + 1) * (atom->ntypes + 1), 0.0);   /* the provider computes no real-space pairs */
    env.cutsq = cutsq0.data();
    env.device = comm->nprocs > 1 ? -(2 + conp_glue::node_local_rank(world)) : 0;     // ranks of a NODE spread over its GPUs
    env.rank = comm->me; env.nranks = comm->nprocs;
    fail_if(conp_fix_create(&fa, &env, &h));
    if (comm->nprocs > 1) {
      conp_comm cc;
      cc.ctx = &world; cc.rank = comm->me; cc.nranks = comm->nprocs;
      cc.allreduce_sum = km_allreduce_sum; cc.allreduce_max_int = km_allreduce_max_int;
      cc.allgather_int = km_allgather_int; cc.allgatherv = km_allgatherv;
      fail_if(conp_fix_set_comm(h, &cc));
    }
  }
  double qsqsum = 0.0;                                        /* km_ewald.cpp:72-78 */
  for (int i = 0; i < atom->nlocal; i++) qsqsum += atom->q[i] * atom->q[i];
  MPI_Allreduce(MPI_IN_PLACE, &qsqsum, 1, MPI_DOUBLE, MPI_SUM, world);
  fail_if(conp_km_conp_setup(h, qsqsum, (int64_t)atom->natoms));
}

/* km_ewald.cpp:232-275 (re)allocates the provider's tables when the fix's atom counts change; here the handle re-reads the
 * atoms and rebuilds its own index maps (same algorithm as FixConp::post_neighbor :468-539, so the same permanent numbering;
 * results are mapped through TAGS anyway, refresh_maps()) */
void KSpaceModuleHip::conp_post_neighbor(bool, bool) {
  conp_atoms at = view();
  /* the handle's hooks want a neighbour list; the provider has no pair work: an empty one (numneigh / first are per-atom arrays) */
  nolist.assign((size_t)at.nlocal + at.nghost + 1, 0);
  conp_neighlist empty;
  empty.inum = 0; empty.ilist = nolist.data(); empty.numneigh = nolist.data(); empty.first = nolist.data();
  empty.neigh = nolist.data(); empty.nneigh = 0;
  fail_if(conp_fix_init_list(h, 2, &empty));
  if (first) { fail_if(conp_fix_setup_post_neighbor(h, &at)); first = false; }
  else fail_if(conp_fix_post_neighbor(h, &at));
  refresh_maps();
}

void KSpaceModuleHip::refresh_maps() {
  conp_info info;
  fail_if(conp_fix_info(h, &info));
  lib_tag2eleall.assign((size_t)info.maxtag_all + 1, 0);
  fail_if(conp_fix_get_maps(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, lib_tag2eleall.data()));
}

void KSpaceModuleHip::a_cal(double *aaa) {
  conp_atoms at = view();
  const int ne = fixconp->elenum_all, nloc = fixconp->elenum;
  std::vector<double> full((size_t)ne * ne);
  fail_if(conp_km_a_cal(h, &at, full.data()));
  /* the library returns each unordered pair folded into the lower triangle; the reference's caller symmetrises afterwards
   * (fix_conp.cpp:826-831), so any single orientation is valid.  Rows / columns go from the library's numbering to the fix's by tag. */
  for (int i = 0; i < nloc; ++i) {
    const size_t li = (size_t)lib_tag2eleall[fixconp->ele2tag[i]];
    for (int j = 0; j < ne; ++j) aaa[(size_t)i * ne + j] += full[li * ne + (size_t)lib_tag2eleall[fixconp->eleall2tag[j]]];
  }
}

void KSpaceModuleHip::b_cal(double *bbb) {
  conp_atoms at = view();
  std::vector<double> ball(fixconp->elenum_all);
  fail_if(conp_km_b_cal(h, &at, ball.data()));
  for (int i = 0; i < fixconp->elenum; ++i) bbb[i] = ball[lib_tag2eleall[fixconp->ele2tag[i]]];   /* overwrite (km_ewald.cpp:821) */
}
