#include "kspacemodule_hip.h"

#include <vector>

using namespace LAMMPS_NS;

KSpaceModuleHip::KSpaceModuleHip(LAMMPS *lmp, conp_fix *handle, int gb, int jgb, const int *const *e2ea, const int *en,
                                 const int *ena)
    : Pointers(lmp), h(handle), groupbit(gb), jgroupbit(jgb), ele2eleall(e2ea), elenum(en), elenum_all(ena) {}

void KSpaceModuleHip::fail_if(int status) {
  if (status != CONP_OK) error->all(FLERR, conp_last_error());   // the reference's only error channel (fix_conp.cpp:86,...)
}

void KSpaceModuleHip::fill_atoms(conp_atoms &at, int *&echeck_buf, double *&x_buf) {
  const int nall = atom->nlocal + atom->nghost;
  echeck_buf = new int[nall];
  x_buf = new double[3 * (size_t)nall];
  for (int i = 0; i < nall; ++i) {
    echeck_buf[i] = (atom->mask[i] & groupbit) ? 1 : ((atom->mask[i] & jgroupbit) ? -1 : 0);   // fix_conp.cpp:599-605
    for (int c = 0; c < 3; ++c) x_buf[3 * (size_t)i + c] = atom->x[i][c];
  }
  at.nlocal = atom->nlocal; at.nghost = atom->nghost; at.x = x_buf; at.q = atom->q; at.type = atom->type;
  at.tag = atom->tag; at.echeck = echeck_buf;
}

void KSpaceModuleHip::conp_setup(bool /*lowmem: the HIP provider regenerates phases on the fly, no table choice*/) {
  double qsqsum = 0.0;                                        // km_ewald.cpp:72-78; multi-rank: MPI_Allreduce here
  for (int i = 0; i < atom->nlocal; i++) qsqsum += atom->q[i] * atom->q[i];
  fail_if(conp_km_conp_setup(h, qsqsum, (int64_t)atom->natoms));
}

void KSpaceModuleHip::a_cal(double *aaa) {
  conp_atoms at; int *ec; double *xb;
  fill_atoms(at, ec, xb);
  const int ne = *elenum_all, nloc = *elenum;
  std::vector<double> full((size_t)ne * ne);
  const int rc = conp_km_a_cal(h, &at, full.data());
  delete[] ec; delete[] xb;
  fail_if(rc);
  // the library returns rows in permanent (eleall) order with each unordered pair folded into the lower triangle;
  // the reference's caller symmetrises afterwards (fix_conp.cpp:826-831), so any single orientation is valid
  for (int i = 0; i < nloc; ++i)
    for (int j = 0; j < ne; ++j) aaa[(size_t)i * ne + j] += full[(size_t)(*ele2eleall)[i] * ne + j];
}

void KSpaceModuleHip::b_cal(double *bbb) {
  conp_atoms at; int *ec; double *xb;
  fill_atoms(at, ec, xb);
  std::vector<double> ball(*elenum_all);
  const int rc = conp_km_b_cal(h, &at, ball.data());
  delete[] ec; delete[] xb;
  fail_if(rc);
  for (int i = 0; i < *elenum; ++i) bbb[i] = ball[(*ele2eleall)[i]];   // overwrite, local electrode order (km_ewald.cpp:821)
}
