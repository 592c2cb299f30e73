/* FixConpHip -- `fix ID group1 conp/hip Nevery group2 eta DV logfile [keywords]`: the reference's `fix conp`
 * (fix_conp.h:19-21, README.md:36-116) with its per-step charge solve and once-per-run matrix work served by
 * libconp_hip.so on an MI355X.  Same command syntax, same Fix hooks (fix_conp.cpp:233-241), same global scalar. */
#ifdef FIX_CLASS

FixStyle(conp/hip,FixConpHip)
FixStyle(conq/hip,FixConpHip)
FixStyle(cond/hip,FixConpHip)

#else

#ifndef LMP_FIX_CONP_HIP_H
#define LMP_FIX_CONP_HIP_H

#include <string>
#include <vector>

#include "conp_hip.h"
#ifdef CONP_GLUE_MOCK
#include "mock_lammps/lammps_mock.h"
#else
#include "fix.h"
#endif

namespace LAMMPS_NS {

class FixConpHip : public Fix {
 public:
  FixConpHip(class LAMMPS *, int, char **);
  ~FixConpHip() override;
  int setmask() override;
  void init() override;
  void init_list(int, class NeighList *) override;
  void setup_post_neighbor() override;
  void setup_pre_force(int) override;
  void post_neighbor() override;
  void pre_force(int) override;
  void post_force(int) override;
  void end_of_step() override;
  double compute_scalar() override;
  int modify_param(int, char **) override;

 private:
  conp_fix_args args;
  conp_fix *h;
  int jgroup, jgroupbit, potdiffvar;
  int arequest, brequest;
  class NeighList *alist, *blist;
  class Pair *coulpair;
  FILE *outf;
  bool postforceflag;
  std::vector<std::vector<std::string>> pending_modify;
  std::vector<double> cutsq_flat;
  std::vector<int> echeck, first_a, first_b, neigh_a, neigh_b;
  conp_atoms view();
  void push_list(int which, class NeighList *l, std::vector<int> &first, std::vector<int> &neigh);
  double potdiff_now();
  void fail_if(int status);
  void flush_log();
  void request_smartlist();
};

}  // namespace LAMMPS_NS
#endif
#endif
