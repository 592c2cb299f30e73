// INTERFACE MOCK -- NOT LAMMPS.  Declares just enough of the LAMMPS 27May2021 class surface for the glue in this directory:
// `make check` type-checks the glue against it, and mock_runtime.cpp gives the declarations minimal bodies so that
// glue_driver.cpp can drive FixConpHip through the same hook sequence LAMMPS' Modify / Verlet use (tests/test_gpu_glue.py).
// LAMMPS itself is not in the build image; against a real LAMMPS tree the glue includes the real headers instead
// (INTEGRATION.md).  This is a test host for OUR glue, not a way to build the reference.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "mpi_mock.h"

#define FLERR __FILE__, __LINE__
#define NEIGHMASK 0x3FFFFFFF

namespace LAMMPS_NS {
typedef int tagint;
typedef int64_t bigint;

class Error { public: [[noreturn]] void all(const char *, int, const std::string &); [[noreturn]] void one(const char *, int, const std::string &); void warning(const char *, int, const std::string &); };
class Memory {};
class Atom { public: int nlocal, nghost, ntypes, nmax = 0; bigint natoms; double **x, **f, *q; int *type, *mask, *molecule = nullptr; tagint *tag; int map(tagint);
             int *map_array = nullptr; int map_size = 0; };
class KSpace { public: double g_ewald, accuracy, slab_volfactor, energy; int slabflag; int nx_pppm = 0, ny_pppm = 0, nz_pppm = 0, order = 0;
               int compute_flag = 1, tip4pflag = 0; double qsum = 0.0;
               virtual ~KSpace() {} virtual void setup(); };
class NeighList;
class Pair { public: double **cutsq; double eng_coul, virial[6]; double cut_coul = 0.0; NeighList *list = nullptr; virtual ~Pair() {} virtual void *extract(const char *, int &); void ev_tally(int, int, int, int, double, double, double, double, double, double); };
class Force { public: double qqrd2e, qqr2e, qe2f, dielectric; int newton_pair, newton = 0; KSpace *kspace; Pair *pair; Pair *pair_match(const std::string &, int, int nsub = 0); };
class Domain { public: double xprd, yprd, zprd, zprd_half, boxlo[3]; };
class Update { public: bigint ntimestep, laststep, eflag_atom = 0; char *integrate_style; };
class Comm { public: int me, nprocs; };
class Group { public: int *bitmask; int find(const std::string &); int ngroup = 0; std::string names[32]; };
class Variable { public: int find(const char *); int equalstyle(int); double compute_equal(int);
                 std::string name; double value = 0.0; };   // the mock knows one equal-style variable
class Input { public: Variable *variable; };
class NeighRequest { public: int pair, fix, half, full, occasional, skip, intel; int *iskip, **ijskip; };
class NeighList { public: int index, inum, occasional; int *ilist, *numneigh, **firstneigh; };
class Neighbor { public: NeighRequest **requests; int nrequest = 0; int request(void *, int instance = 0); void build(int); void build_one(NeighList *, int preflag = 0); };
class Modify { public: int find_fix(const std::string &); };

class LAMMPS { public: MPI_Comm world = nullptr; Memory *memory; Error *error; Atom *atom; Force *force; Domain *domain; Update *update; Comm *comm; Group *group; Input *input; Neighbor *neighbor; Modify *modify; FILE *screen, *logfile; };

class Pointers {
 public:
  explicit Pointers(LAMMPS *ptr)
      : lmp(ptr), world(ptr->world), memory(ptr->memory), error(ptr->error), atom(ptr->atom), force(ptr->force), domain(ptr->domain),
        update(ptr->update), comm(ptr->comm), group(ptr->group), input(ptr->input), neighbor(ptr->neighbor),
        modify(ptr->modify), screen(ptr->screen), logfile(ptr->logfile) {}
  virtual ~Pointers() {}
 protected:
  LAMMPS *lmp; MPI_Comm &world; Memory *&memory; Error *&error; Atom *&atom; Force *&force; Domain *&domain; Update *&update; Comm *&comm;
  Group *&group; Input *&input; Neighbor *&neighbor; Modify *&modify; FILE *&screen; FILE *&logfile;
};

// LAMMPS' PPPM kspace style and Compute base, as far as pppm_conp_hip.* / compute_potential_atom_hip.* touch them
// PPPM::compute (pppm.cpp @ 27May2021) begins with  particle_map(); make_rho();  -- both virtual -- then sums the ghost planes into
// their owners (gc->reverse_comm), copies the owned brick to the FFT layout (brick2fft) and solves.  The mock runs the two virtuals
// on a brick that covers the rank's share of the mesh plus `order` ghost planes on every side (one rank: the whole mesh), folds the
// ghost planes back periodically and keeps the result (`density_fft`, [nz][ny][nx]): what the real class would transform.
typedef double FFT_SCALAR;
class PPPM : public KSpace, protected Pointers {
 public:
  explicit PPPM(LAMMPS *l) : Pointers(l) {}
  virtual void compute(int eflag, int vflag);
  virtual void particle_map() { ++base_particle_map_calls; }
  virtual void make_rho();                      // (the real one clears the brick and spreads every charged atom: counted here)
  int base_particle_map_calls = 0, base_make_rho_calls = 0;
  FFT_SCALAR ***density_brick = nullptr;        // [nzlo_out..nzhi_out][nylo_out..nyhi_out][nxlo_out..nxhi_out]
  int nxlo_in = 0, nxhi_in = -1, nylo_in = 0, nyhi_in = -1, nzlo_in = 0, nzhi_in = -1;
  int nxlo_out = 0, nxhi_out = -1, nylo_out = 0, nyhi_out = -1, nzlo_out = 0, nzhi_out = -1, ngrid = 0;
  std::vector<double> density_fft;
  void mock_allocate();
 private:
  std::vector<FFT_SCALAR> brick_store;
  std::vector<FFT_SCALAR *> brick_rows;
  std::vector<FFT_SCALAR **> brick_planes;
};
class Compute : protected Pointers {
 public:
  Compute(LAMMPS *l, int narg, char **arg) : Pointers(l) {
    igroup = narg > 1 ? l->group->find(arg[1]) : 0;
    groupbit = igroup >= 0 ? l->group->bitmask[igroup] : 0;
  }
  virtual ~Compute() {}
  int igroup, groupbit, peratom_flag = 0, size_peratom_cols = 0, peatomflag = 0, timeflag = 0, comm_reverse = 0;
  bigint invoked_peratom = -1;
  double *vector_atom = nullptr;
  virtual void init() {}
  virtual void setup() {}
  virtual void compute_peratom() {}
  virtual double memory_usage() { return 0.0; }
};

namespace FixConst { enum { PRE_EXCHANGE = 1 << 1, POST_NEIGHBOR = 1 << 3, PRE_FORCE = 1 << 5, POST_FORCE = 1 << 7, END_OF_STEP = 1 << 10 }; }

class Fix : protected Pointers {
 public:
  Fix(LAMMPS *l, int narg, char **arg) : Pointers(l) {      // like Fix::Fix: arg[1] is the fix group
    igroup = narg > 1 ? l->group->find(arg[1]) : 0;
    groupbit = igroup >= 0 ? l->group->bitmask[igroup] : 0;
    instance_me = 0;
  }
  virtual ~Fix() {}
  int igroup, groupbit, instance_me, scalar_flag, extscalar, global_freq, respa_level;
  virtual int setmask() = 0;
  virtual void init() {}
  virtual void init_list(int, NeighList *) {}
  virtual void setup_post_neighbor() {}
  virtual void setup_pre_force(int) {}
  virtual void pre_exchange() {}
  virtual void setup_pre_exchange() {}
  virtual void post_run() {}
  virtual void post_neighbor() {}
  virtual void pre_force(int) {}
  virtual void post_force(int) {}
  virtual void end_of_step() {}
  virtual double compute_scalar() { return 0.0; }
  virtual int modify_param(int, char **) { return 0; }
};

namespace utils {
int inumeric(const char *, int, const char *, bool, LAMMPS *);
double numeric(const char *, int, const char *, bool, LAMMPS *);
void logmesg(LAMMPS *, const std::string &);
}  // namespace utils
}  // namespace LAMMPS_NS
