// Minimal bodies for the interface mock (lammps_mock.h) -- a test host for the glue, NOT LAMMPS.
// Errors become C++ exceptions so that the driver can report the message the glue passed to error->all().
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "lammps_mock.h"

// ---- the MPI mock: rank threads of one process meet at a generation barrier; a collective = publish, barrier, read, barrier
void MockWorld::barrier() {
  std::unique_lock<std::mutex> lk(m);
  if (failed) throw std::runtime_error("another rank failed");
  const long gen = generation;
  if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); return; }
  cv.wait(lk, [&] { return generation != gen || failed; });
  if (failed && generation == gen) throw std::runtime_error("another rank failed");
}
void MockWorld::fail() {
  std::lock_guard<std::mutex> lk(m);
  failed = true;
  cv.notify_all();
}

static size_t mock_size(MPI_Datatype t) { return t == MPI_DOUBLE ? sizeof(double) : (t == MPI_INT ? sizeof(int) : 1); }

int MPI_Comm_rank(MPI_Comm c, int *r) { *r = c ? c->rank : 0; return MPI_SUCCESS; }
int MPI_Comm_size(MPI_Comm c, int *n) { *n = c ? c->w->n : 1; return MPI_SUCCESS; }
int MPI_Barrier(MPI_Comm c) { if (c && c->w->n > 1) c->w->barrier(); return MPI_SUCCESS; }

int MPI_Allreduce(const void *send, void *recv, int count, MPI_Datatype t, MPI_Op op, MPI_Comm c) {
  const size_t bytes = (size_t)count * mock_size(t);
  if (!c || c->w->n == 1) { if (send != MPI_IN_PLACE) std::memcpy(recv, send, bytes); return MPI_SUCCESS; }
  MockWorld *w = c->w;
  std::vector<char> mine(bytes), res(bytes);
  std::memcpy(mine.data(), send == MPI_IN_PLACE ? recv : send, bytes);
  w->ptr[c->rank] = mine.data();
  w->barrier();
  for (int k = 0; k < count; ++k) {          // fixed rank order: every rank forms the same bits
    if (t == MPI_DOUBLE) {
      double a = static_cast<const double *>(w->ptr[0])[k];
      for (int r = 1; r < w->n; ++r) { const double b = static_cast<const double *>(w->ptr[r])[k]; a = op == MPI_SUM ? a + b : (b > a ? b : a); }
      reinterpret_cast<double *>(res.data())[k] = a;
    } else {
      int a = static_cast<const int *>(w->ptr[0])[k];
      for (int r = 1; r < w->n; ++r) { const int b = static_cast<const int *>(w->ptr[r])[k]; a = op == MPI_SUM ? a + b : (b > a ? b : a); }
      reinterpret_cast<int *>(res.data())[k] = a;
    }
  }
  w->barrier();
  std::memcpy(recv, res.data(), bytes);
  return MPI_SUCCESS;
}

int MPI_Allgatherv(const void *send, int scount, MPI_Datatype t, void *recv, const int *rcounts, const int *displs, MPI_Datatype, MPI_Comm c) {
  const size_t sz = mock_size(t);
  if (!c || c->w->n == 1) { std::memcpy(static_cast<char *>(recv) + (size_t)displs[0] * sz, send, (size_t)scount * sz); return MPI_SUCCESS; }
  MockWorld *w = c->w;
  w->ptr[c->rank] = send;
  w->len[c->rank] = scount;
  w->barrier();
  for (int r = 0; r < w->n; ++r)
    if (rcounts[r] > 0) std::memcpy(static_cast<char *>(recv) + (size_t)displs[r] * sz, w->ptr[r], (size_t)rcounts[r] * sz);
  w->barrier();
  return MPI_SUCCESS;
}

int MPI_Allgather(const void *send, int scount, MPI_Datatype t, void *recv, int rcount, MPI_Datatype rt, MPI_Comm c) {
  int n = 1;
  MPI_Comm_size(c, &n);
  std::vector<int> counts(n, rcount), displs(n);
  for (int r = 0; r < n; ++r) displs[r] = r * rcount;
  return MPI_Allgatherv(send, scount, t, recv, counts.data(), displs.data(), rt, c);
}

namespace LAMMPS_NS {

void PPPM::mock_allocate() {
  nxlo_in = 0; nxhi_in = nx_pppm - 1; nylo_in = 0; nyhi_in = ny_pppm - 1; nzlo_in = 0; nzhi_in = nz_pppm - 1;
  nxlo_out = -order; nxhi_out = nx_pppm - 1 + order; nylo_out = -order; nyhi_out = ny_pppm - 1 + order;
  nzlo_out = -order; nzhi_out = nz_pppm - 1 + order;
  const int ex = nxhi_out - nxlo_out + 1, ey = nyhi_out - nylo_out + 1, ez = nzhi_out - nzlo_out + 1;
  ngrid = ex * ey * ez;
  brick_store.assign((size_t)ngrid, 0.0);
  brick_rows.resize((size_t)ey * ez);
  brick_planes.resize(ez);
  // offset pointer arrays like Memory::create3d_offset: density_brick[z][y][x] with z, y, x from the *_out lower bounds
  for (int z = 0; z < ez; ++z) {
    for (int y = 0; y < ey; ++y) brick_rows[(size_t)z * ey + y] = brick_store.data() + ((size_t)z * ey + y) * ex - nxlo_out;
    brick_planes[z] = brick_rows.data() + (size_t)z * ey - nylo_out;
  }
  density_brick = brick_planes.data() - nzlo_out;
}
void PPPM::make_rho() {
  ++base_make_rho_calls;
  std::fill(brick_store.begin(), brick_store.end(), 0.0);
}
void PPPM::compute(int, int) {
  if (density_brick == nullptr) mock_allocate();
  particle_map();
  make_rho();
  // gc->reverse_comm + brick2fft on one rank: every ghost point is added to its periodic image
  density_fft.assign((size_t)nx_pppm * ny_pppm * nz_pppm, 0.0);
  auto wrap = [](int i, int n) { return ((i % n) + n) % n; };
  for (int z = nzlo_out; z <= nzhi_out; ++z)
    for (int y = nylo_out; y <= nyhi_out; ++y)
      for (int x = nxlo_out; x <= nxhi_out; ++x)
        density_fft[((size_t)wrap(z, nz_pppm) * ny_pppm + wrap(y, ny_pppm)) * nx_pppm + wrap(x, nx_pppm)] += density_brick[z][y][x];
}

void Error::all(const char *file, int line, const std::string &msg) {
  (void)file; (void)line;
  throw std::runtime_error(msg);
}
void Error::one(const char *file, int line, const std::string &msg) { all(file, line, msg); }   // (LAMMPS: MPI_Abort from the calling rank alone)
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
