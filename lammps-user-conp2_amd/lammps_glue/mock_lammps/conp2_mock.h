// INTERFACE MOCK -- NOT USER-CONP2.  The part of the reference package's OWN public surface that a k-space provider touches:
// the KSpaceModule base (kspacemodule.h:26-45) and the public members of FixConp a provider reads (fix_conp.h:58-89), declared
// just far enough to type-check and execute kspacemodule_hip.cpp without the reference tree.  In a LAMMPS + USER-CONP2 tree the
// glue includes the reference's "kspacemodule.h" instead and this file is not used.
#pragma once
#include <functional>

#include "lammps_mock.h"

namespace LAMMPS_NS {

class KSpaceModule;

class FixConp {                       // stands for `class FixConp : public Fix` -- only what the provider reads
 public:
  int elenum = 0, elenum_all = 0, elytenum = 0;
  int *ele2tag = nullptr, *ele2eleall = nullptr, *tag2eleall = nullptr, *eleall2tag = nullptr;
  KSpaceModule *kspmod = nullptr;
  double eta = 0.0;
  bool splitflag = false;
  std::function<int(int)> check;      // test host: mask & groupbit logic of fix_conp.cpp:599-605
  int electrode_check(int i) { return check(i); }
};

class KSpaceModule {
 public:
  KSpaceModule() { fixconp = nullptr; }
  virtual ~KSpaceModule() {}
  void register_fix(class FixConp *infix) { fixconp = infix; }
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
  class FixConp *fixconp;

 protected:
  bool lowmemflag;
};

}  // namespace LAMMPS_NS
