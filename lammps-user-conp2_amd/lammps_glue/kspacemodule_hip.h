/* KSpaceModuleHip -- the Ewald k-space provider of USER-CONP2 (km_ewald.h:27-80) served by libconp_hip.so.
 * It implements the reference's provider interface (kspacemodule.h:26-45): a maintainer who keeps the reference's
 * FixConp unchanged replaces `new KSpaceModuleEwald(lmp)` (fix_conp.cpp:408) by `new KSpaceModuleHip(lmp, handle)`.
 * FixConpHip (fix_conp_hip.h) uses the fused hooks instead and does not need this class. */
#ifndef LMP_FIXCONP_KM_HIP_H
#define LMP_FIXCONP_KM_HIP_H

#include "conp_hip.h"
#ifdef CONP_GLUE_MOCK
#include "mock_lammps/lammps_mock.h"
#else
#include "pointers.h"
#endif

namespace LAMMPS_NS {

/* same virtual surface as the reference's KSpaceModule (kspacemodule.h:30-40) */
class KSpaceModuleIface {
 public:
  virtual ~KSpaceModuleIface() {}
  virtual void conp_setup(bool) {}
  virtual void conp_post_neighbor(bool, bool) {}
  virtual void conp_pre_force() {}
  virtual void a_cal(double *) {}
  virtual void a_read() {}
  virtual void b_cal(double *) {}
  virtual void update_charge() {}
  virtual double compute_particle_potential(int) { return 0.; }
  virtual void compute_group_potential(int, double *) {}
  virtual double return_qsum() { return 0.; }
};

class KSpaceModuleHip : public KSpaceModuleIface, protected Pointers {
 public:
  /* ele2eleall / elenum: the owning fix's public maps (fix_conp.h:58-66), read each call like the reference does */
  KSpaceModuleHip(LAMMPS *lmp, conp_fix *handle, int groupbit, int jgroupbit, const int *const *ele2eleall, const int *elenum,
                  const int *elenum_all);
  void conp_setup(bool lowmem) override;                 /* km_ewald.cpp:63-132 */
  void a_cal(double *aaa) override;                      /* km_ewald.cpp:147-151 : aaa[elenum][elenum_all], k-space part */
  void b_cal(double *bbb) override;                      /* km_ewald.cpp:153-167 : bbb[elenum], local electrode order */
  /* km_ewald.cpp:232-275 / :134-145 only (re)allocate and fill the provider's phase tables; the library sizes its device
   * tables from the atoms handed to a_cal / b_cal, so these two hooks have nothing left to do.  update_charge,
   * compute_particle_potential, compute_group_potential and return_qsum keep the base-class defaults, as in KSpaceModuleEwald. */
  void conp_post_neighbor(bool, bool) override {}
  void a_read() override {}

 private:
  conp_fix *h;
  int groupbit, jgroupbit;
  const int *const *ele2eleall;
  const int *elenum, *elenum_all;
  void fail_if(int status);
  void fill_atoms(conp_atoms &at, int *&echeck_buf, double *&x_buf);
};

}  // namespace LAMMPS_NS
#endif
