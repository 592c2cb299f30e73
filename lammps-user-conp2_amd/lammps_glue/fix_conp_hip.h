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
  void pre_exchange() override;
  void setup_pre_exchange() override { pre_exchange(); }   /* Verlet::setup's exchange / borders grow the atom arrays without a pre_exchange() call */
  void post_run() override { pre_exchange(); }             /* between runs (create_atoms, read_dump, ...) the arrays may move: nothing stays page-locked */
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
  // the flattened neighbour lists live in page-locked memory (conp_host_alloc): the library's upload at a re-neighbour is an
  // asynchronous DMA transfer, not a staged copy
  struct PinnedInts {
    int *p = nullptr;
    size_t n = 0, cap = 0;
    ~PinnedInts() { conp_host_free(p); }
    void reserve(size_t want);
    void assign(size_t count, int v) { reserve(count); n = count; for (size_t i = 0; i < count; ++i) p[i] = v; }
    void clear() { n = 0; }
    void append(const int *b, const int *e) { const size_t k = (size_t)(e - b); reserve(n + k); for (size_t i = 0; i < k; ++i) p[n + i] = b[i]; n += k; }
    void push_back(int v) { reserve(n + 1); p[n++] = v; }
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
    int *data() { return p; }
  };
  std::vector<int> echeck;
  PinnedInts first_a, first_b, neigh_a, neigh_b;
  const double *pinned_x = nullptr, *pinned_q = nullptr;
  int pinned_n = 0;
  conp_atoms view();
  void pin_atoms();
  void push_list(int which, class NeighList *l, PinnedInts &first, PinnedInts &neigh);
  double potdiff_now();
  void fail_if(int status);
  void flush_log();
  void request_smartlist();
};

}  // namespace LAMMPS_NS
#endif
#endif
