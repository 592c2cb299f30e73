#include "pppm_conp_hip.h"

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
#endif

#include "conp_mpi_comm.h"

using namespace LAMMPS_NS;

/* the same MPI-backed callbacks as fix_conp_hip.cpp / kspacemodule_hip.cpp (ctx = &world): conp_mpi_comm.h */
static int pp_allreduce_sum(void *ctx, double *buf, int64_t n) { return conp_glue::cb_allreduce_sum(ctx, buf, n); }
static int pp_allreduce_max_int(void *ctx, int *buf, int n) { return conp_glue::cb_allreduce_max_int(ctx, buf, n); }
static int pp_allgather_int(void *ctx, int value, int *out) { return conp_glue::cb_allgather_int(ctx, value, out); }
static int pp_allgatherv(void *ctx, const void *send, int64_t nbytes, void *recv, const int64_t *counts, const int64_t *displs) {
  return conp_glue::cb_allgatherv(ctx, send, nbytes, recv, counts, displs);
}

PPPMConpHip::PPPMConpHip(LAMMPS *lmp) : PPPM(lmp), KSpaceModule(), h(nullptr), first(true) {}

PPPMConpHip::~PPPMConpHip() { conp_fix_destroy(h); }   /* unlike the Ewald provider, the fix does not delete force->kspace (fix_conp.cpp:207) */

void PPPMConpHip::fail_if(int status) {
  if (status != CONP_OK) error->all(FLERR, conp_last_error());
}

conp_atoms PPPMConpHip::view() {
  const int nall = atom->nlocal + atom->nghost;
  echeck.resize(nall);
  xflat.resize(3 * (size_t)nall);
  for (int i = 0; i < nall; ++i) {
    echeck[i] = fixconp ? fixconp->electrode_check(i) : 0;
    for (int c = 0; c < 3; ++c) xflat[3 * (size_t)i + c] = atom->x[i][c];
  }
  conp_atoms at;
  at.nlocal = atom->nlocal; at.nghost = atom->nghost; at.x = xflat.data(); at.q = atom->q; at.type = atom->type;
  at.tag = atom->tag; at.echeck = echeck.data();
  return at;
}

void PPPMConpHip::conp_setup(bool lowmem) {
  lowmemflag = lowmem;
  if (fixconp == nullptr) error->all(FLERR, "pppm/conp/hip: register_fix() must precede conp_setup()");
  /* several MPI ranks: the mesh is not sharded -- rank 0 owns it; the ranks' charged electrolyte atoms are gathered per update
   * like for the Ewald provider, the electrode vector is summed over the ranks (pppm_conp.cpp:114,122 shard the mesh instead) */
  if (h == nullptr) {
    conp_fix_args fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.everynum = 1; fa.eta = fixconp->eta; fa.minimizer = CONP_SOLVER_INV; fa.maxiter = 100; fa.tolerance = 1e-6;
    fa.lowmem = lowmem ? 1 : 0; fa.nullneutral = 1; fa.pppm = 1;
    conp_env env;
    std::memset(&env, 0, sizeof(env));
    env.qqrd2e = force->qqrd2e; env.qqr2e = force->qqr2e; env.qe2f = force->qe2f; env.dielectric = force->dielectric;
    env.newton_pair = force->newton_pair;
    env.g_ewald = g_ewald; env.accuracy = accuracy; env.slab_volfactor = slab_volfactor; env.slabflag = slabflag;   /* own KSpace members */
    env.xprd = domain->xprd; env.yprd = domain->yprd; env.zprd = domain->zprd;
    env.boxlo_x = domain->boxlo[0]; env.boxlo_y = domain->boxlo[1]; env.boxlo_z = domain->boxlo[2];
    env.pppm_nx = nx_pppm; env.pppm_ny = ny_pppm; env.pppm_nz = nz_pppm; env.pppm_order = order;                      /* PPPM::set_grid_global's result */
    env.ntypes = atom->ntypes;
    cutsq0.assign((size_t)(atom->ntypes + 1) * (atom->ntypes + 1), 0.0);
    env.cutsq = cutsq0.data();
    env.device = comm->nprocs > 1 ? -(2 + conp_glue::node_local_rank(world)) : 0;     // ranks of a NODE spread over its GPUs
    env.rank = comm->me; env.nranks = comm->nprocs;
    fail_if(conp_fix_create(&fa, &env, &h));
    if (comm->nprocs > 1) {
      conp_comm cc;
      cc.ctx = &world; cc.rank = comm->me; cc.nranks = comm->nprocs;
      cc.allreduce_sum = pp_allreduce_sum; cc.allreduce_max_int = pp_allreduce_max_int;
      cc.allgather_int = pp_allgather_int; cc.allgatherv = pp_allgatherv;
      fail_if(conp_fix_set_comm(h, &cc));
    }
  }
}

void PPPMConpHip::conp_post_neighbor(bool, bool) {
  conp_atoms at = view();
  nolist.assign((size_t)at.nlocal + at.nghost + 1, 0);
  conp_neighlist empty;
  empty.inum = 0; empty.ilist = nolist.data(); empty.numneigh = nolist.data(); empty.first = nolist.data();
  empty.neigh = nolist.data(); empty.nneigh = 0;
  fail_if(conp_fix_init_list(h, 2, &empty));
  if (first) { fail_if(conp_fix_setup_post_neighbor(h, &at)); first = false; }
  else fail_if(conp_fix_post_neighbor(h, &at));
  conp_info info;
  fail_if(conp_fix_info(h, &info));
  lib_tag2eleall.assign((size_t)info.maxtag_all + 1, 0);
  fail_if(conp_fix_get_maps(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, lib_tag2eleall.data()));
}

void PPPMConpHip::a_cal(double *aaa) {          /* the Ewald matrix, like PPPMCONP::a_cal's temporary KSpaceModuleEwald (:91-101) */
  conp_atoms at = view();
  const int ne = fixconp->elenum_all, nloc = fixconp->elenum;
  std::vector<double> full((size_t)ne * ne);
  fail_if(conp_km_a_cal(h, &at, full.data()));
  for (int i = 0; i < nloc; ++i) {
    const size_t li = (size_t)lib_tag2eleall[fixconp->ele2tag[i]];
    for (int j = 0; j < ne; ++j) aaa[(size_t)i * ne + j] += full[li * ne + (size_t)lib_tag2eleall[fixconp->eleall2tag[j]]];
  }
}

void PPPMConpHip::conp_pre_force() {           /* pppm_conp.h:42: elyte_mapped = false -- a new step, the kept brick is stale */
  if (h) fail_if(conp_pppm_keep_density(h, 1));
}

void PPPMConpHip::b_cal(double *bbb) {          /* spread, Poisson solve, stencil gather on the device mesh (:269-316) */
  if (!bcal_done) fail_if(conp_pppm_keep_density(h, 1));      /* the make_rho override wants the electrolyte brick of every b_cal */
  bcal_done = true;
  conp_atoms at = view();
  std::vector<double> ball(fixconp->elenum_all);
  fail_if(conp_km_b_cal(h, &at, ball.data()));
  for (int i = 0; i < fixconp->elenum; ++i) bbb[i] = ball[lib_tag2eleall[fixconp->ele2tag[i]]];
}

void PPPMConpHip::particle_map() {              /* pppm_conp.cpp:428-432 */
  if (!bcal_done || fixconp == nullptr) PPPM::particle_map();
  /* else: the library maps the atoms on the device when it spreads them (elyte_particle_map :126-170 inside b_cal) */
}

void PPPMConpHip::make_rho() {                  /* pppm_conp.cpp:434-450 */
  if (!bcal_done || fixconp == nullptr) { PPPM::make_rho(); return; }
  conp_atoms at = view();
  dens.resize((size_t)nx_pppm * ny_pppm * nz_pppm);
  /* electrolyte brick as b_cal left it (not spread again on one rank) + the electrode brick of the CURRENT charges (ele_make_rho
   * :385-426); the library's bricks are [nz][ny][nx] with the ghost planes folded in, so the owned points are filled and the
   * ghost planes left zero: PPPM::compute's ghost sum (gc->reverse_comm) then adds nothing */
  fail_if(conp_pppm_make_rho(h, &at, dens.data(), nullptr, nullptr));
  std::memset(&(density_brick[nzlo_out][nylo_out][nxlo_out]), 0, (size_t)ngrid * sizeof(FFT_SCALAR));
  for (int iz = nzlo_in; iz <= nzhi_in; ++iz)
    for (int iy = nylo_in; iy <= nyhi_in; ++iy)
      for (int ix = nxlo_in; ix <= nxhi_in; ++ix)
        density_brick[iz][iy][ix] = (FFT_SCALAR)dens[((size_t)iz * ny_pppm + iy) * nx_pppm + ix];
}

double PPPMConpHip::compute_particle_potential(int i) {
  // RANK-LOCAL, like the reference's (pppm_conp.cpp:452-485; compute_potential_atom.cpp:168-174 calls it once per owned atom of
  // the group -- a different number of calls on every rank): a stencil gather from the mesh potential the library cached when a
  // collective entry last formed it (compute_group_potential, compute potential/atom).  Under several ranks a call without such
  // a brick is an error (error->one: only this rank is here), not a hidden collective.
  conp_atoms at = view();
  double u = 0.0;
  if (conp_pppm_compute_particle_potential(h, &at, i, &u) != CONP_OK) error->one(FLERR, conp_last_error());
  return u;
}

void PPPMConpHip::compute_group_potential(int groupbit, double *recv) {
  conp_atoms at = view();
  sel.resize(atom->nlocal);
  for (int i = 0; i < atom->nlocal; ++i) sel[i] = (atom->mask[i] & groupbit) ? 1 : 0;
  fail_if(conp_pppm_compute_group_potential(h, &at, sel.data(), recv));
}

void PPPMConpHip::total_density(double *density_brick) {
  conp_atoms at = view();
  fail_if(conp_pppm_make_rho(h, &at, density_brick, nullptr, nullptr));
}
