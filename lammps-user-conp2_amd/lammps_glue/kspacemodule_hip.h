/* KSpaceModuleHip -- the Ewald k-space provider of USER-CONP2 (km_ewald.h:27-80) served by libconp_hip.so.
 * It IS a KSpaceModule of the reference (kspacemodule.h:26-45): a maintainer who keeps the reference's FixConp replaces
 *     else kspmod = new KSpaceModuleEwald(lmp);          (fix_conp.cpp:408)
 * by  else kspmod = new KSpaceModuleHip(lmp);
 * and nothing else -- `kspmod->register_fix(this)` (:409) hands the fix over, and the provider reads the fix's public members
 * (elenum, elenum_all, ele2tag, eleall2tag, eta, electrode_check(): fix_conp.h:58-89) exactly as KSpaceModuleEwald does.
 * Ownership: the provider creates its own conp_fix handle in conp_setup() and destroys it in its destructor; FixConp deletes its
 * Ewald provider in ~FixConp (fix_conp.cpp:207), so the handle's life is the fix's.  FixConpHip (fix_conp_hip.h) replaces the
 * whole fix instead and does not use this class. */
#ifndef LMP_FIXCONP_KM_HIP_H
#define LMP_FIXCONP_KM_HIP_H

#include <vector>

#include "conp_hip.h"
#ifdef CONP_GLUE_MOCK
#include "mock_lammps/conp2_mock.h"
#else
#include "kspacemodule.h" /* the reference's header: KSpaceModule, FixConp */
#include "pointers.h"
#endif

namespace LAMMPS_NS {

class KSpaceModuleHip : public KSpaceModule, protected Pointers {
 public:
  explicit KSpaceModuleHip(LAMMPS *lmp);
  ~KSpaceModuleHip() override;
  void conp_setup(bool lowmem) override;                 /* km_ewald.cpp:63-132: handle creation + k tables */
  void conp_post_neighbor(bool, bool) override;          /* km_ewald.cpp:232-275: the handle follows the fix's atoms */
  void a_cal(double *aaa) override;                      /* km_ewald.cpp:147-151 : aaa[elenum][elenum_all] +=, k-space part */
  void a_read() override {}                              /* km_ewald.cpp:134-145: the phase tables are built inside a_cal / b_cal */
  void b_cal(double *bbb) override;                      /* km_ewald.cpp:153-167 : bbb[elenum] =, local electrode order */
  /* update_charge, conp_pre_force, compute_particle_potential, compute_group_potential and return_qsum keep the base-class
   * defaults, as in KSpaceModuleEwald (the Ewald provider has no mesh potential; `pppm/conp/hip` has: pppm_conp_hip.h). */

 private:
  conp_fix *h;
  bool first;
  std::vector<int> echeck, lib_tag2eleall, nolist;
  std::vector<double> xflat, cutsq0;
  void fail_if(int status);
  conp_atoms view();
  void refresh_maps();
};

}  // namespace LAMMPS_NS
#endif
