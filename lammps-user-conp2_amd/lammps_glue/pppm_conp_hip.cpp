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

using namespace LAMMPS_NS;

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
  if (comm->nprocs > 1) error->all(FLERR, "pppm/conp/hip: the device mesh is not sharded over MPI ranks yet (use the Ewald provider)");
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
    env.device = 0; env.rank = 0; env.nranks = 1;
    fail_if(conp_fix_create(&fa, &env, &h));
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

void PPPMConpHip::b_cal(double *bbb) {          /* spread, Poisson solve, stencil gather on the device mesh (:269-316) */
  conp_atoms at = view();
  std::vector<double> ball(fixconp->elenum_all);
  fail_if(conp_km_b_cal(h, &at, ball.data()));
  for (int i = 0; i < fixconp->elenum; ++i) bbb[i] = ball[lib_tag2eleall[fixconp->ele2tag[i]]];
}

double PPPMConpHip::compute_particle_potential(int i) {
  conp_atoms at = view();
  double u = 0.0;
  fail_if(conp_pppm_compute_particle_potential(h, &at, i, &u));
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
