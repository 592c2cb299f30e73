// INTERFACE MOCK -- NOT LAMMPS.  Declares just enough of the LAMMPS 27May2021 class surface for a syntax / type check of
// the glue in this directory (`make check`).  LAMMPS itself is not in the build image; against a real LAMMPS tree the
// glue includes the real headers instead.  Nothing here is linked or executed.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>

#define FLERR __FILE__, __LINE__
#define NEIGHMASK 0x3FFFFFFF

namespace LAMMPS_NS {
typedef int tagint;
typedef int64_t bigint;

class Error { public: [[noreturn]] void all(const char *, int, const std::string &); void warning(const char *, int, const std::string &); };
class Memory {};
class Atom { public: int nlocal, nghost, ntypes; bigint natoms; double **x, **f, *q; int *type, *mask; tagint *tag; int map(tagint); };
class KSpace { public: double g_ewald, accuracy, slab_volfactor, energy; int slabflag; virtual void setup(); };
class Pair { public: double **cutsq; double eng_coul, virial[6]; virtual void *extract(const char *, int &); void ev_tally(int, int, int, int, double, double, double, double, double, double); };
class Force { public: double qqrd2e, qqr2e, qe2f, dielectric; int newton_pair; KSpace *kspace; Pair *pair; Pair *pair_match(const std::string &, int, int nsub = 0); };
class Domain { public: double xprd, yprd, zprd, zprd_half, boxlo[3]; };
class Update { public: bigint ntimestep, laststep; char *integrate_style; };
class Comm { public: int me, nprocs; };
class Group { public: int *bitmask; int find(const std::string &); };
class Variable { public: int find(const char *); int equalstyle(int); double compute_equal(int); };
class Input { public: Variable *variable; };
class NeighRequest { public: int pair, fix, half, full, occasional, skip, intel; int *iskip, **ijskip; };
class NeighList { public: int index, inum, occasional; int *ilist, *numneigh, **firstneigh; };
class Neighbor { public: NeighRequest **requests; int request(void *, int instance = 0); void build(int); void build_one(NeighList *, int preflag = 0); };
class Modify { public: int find_fix(const std::string &); };

class LAMMPS { public: Memory *memory; Error *error; Atom *atom; Force *force; Domain *domain; Update *update; Comm *comm; Group *group; Input *input; Neighbor *neighbor; Modify *modify; FILE *screen, *logfile; };

class Pointers {
 public:
  explicit Pointers(LAMMPS *ptr)
      : lmp(ptr), memory(ptr->memory), error(ptr->error), atom(ptr->atom), force(ptr->force), domain(ptr->domain),
        update(ptr->update), comm(ptr->comm), group(ptr->group), input(ptr->input), neighbor(ptr->neighbor),
        modify(ptr->modify), screen(ptr->screen), logfile(ptr->logfile) {}
  virtual ~Pointers() {}
 protected:
  LAMMPS *lmp; Memory *&memory; Error *&error; Atom *&atom; Force *&force; Domain *&domain; Update *&update; Comm *&comm;
  Group *&group; Input *&input; Neighbor *&neighbor; Modify *&modify; FILE *&screen; FILE *&logfile;
};

namespace FixConst { enum { POST_NEIGHBOR = 1 << 3, PRE_FORCE = 1 << 5, POST_FORCE = 1 << 7, END_OF_STEP = 1 << 10 }; }

class Fix : protected Pointers {
 public:
  Fix(LAMMPS *l, int, char **) : Pointers(l) {}
  virtual ~Fix() {}
  int igroup, groupbit, instance_me, scalar_flag, extscalar, global_freq, respa_level;
  virtual int setmask() = 0;
  virtual void init() {}
  virtual void init_list(int, NeighList *) {}
  virtual void setup_post_neighbor() {}
  virtual void setup_pre_force(int) {}
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
