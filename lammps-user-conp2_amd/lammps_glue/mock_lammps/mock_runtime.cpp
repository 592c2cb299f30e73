// Minimal bodies for the interface mock (lammps_mock.h) -- a test host for the glue, NOT LAMMPS.
// Errors become C++ exceptions so that the driver can report the message the glue passed to error->all().
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "lammps_mock.h"

namespace LAMMPS_NS {

void Error::all(const char *file, int line, const std::string &msg) {
  (void)file; (void)line;
  throw std::runtime_error(msg);
}
void Error::warning(const char *, int, const std::string &msg) { std::fprintf(stderr, "WARNING: %s\n", msg.c_str()); }

int Atom::map(tagint t) { return (t >= 0 && t < map_size) ? map_array[t] : -1; }

void KSpace::setup() {}

void *Pair::extract(const char *name, int &dim) {
  dim = 0;
  if (std::strcmp(name, "cut_coul") == 0) return &cut_coul;
  return nullptr;
}
void Pair::ev_tally(int, int, int, int, double, double, double, double, double, double) {}

Pair *Force::pair_match(const std::string &word, int, int) { return word == "coul" ? pair : nullptr; }

int Group::find(const std::string &name) {
  for (int i = 0; i < ngroup; ++i) if (names[i] == name) return i;
  return -1;
}

int Variable::find(const char *n) { return name == n ? 0 : -1; }
int Variable::equalstyle(int i) { return i == 0; }
double Variable::compute_equal(int) { return value; }

int Neighbor::request(void *, int) {
  NeighRequest *r = new NeighRequest();
  std::memset(r, 0, sizeof(*r));
  r->pair = 1; r->half = 1;
  requests[nrequest] = r;
  return nrequest++;
}
void Neighbor::build(int) {}                       // the test host hands over prebuilt lists
void Neighbor::build_one(NeighList *, int) {}

int Modify::find_fix(const std::string &) { return -1; }

namespace utils {
int inumeric(const char *, int, const char *s, bool, LAMMPS *) { return std::atoi(s); }
double numeric(const char *, int, const char *s, bool, LAMMPS *) { return std::atof(s); }
void logmesg(LAMMPS *lmp, const std::string &m) {
  if (lmp->screen) std::fputs(m.c_str(), lmp->screen);
}
}  // namespace utils
}  // namespace LAMMPS_NS
