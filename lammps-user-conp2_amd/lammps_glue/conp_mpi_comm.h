// conp_comm on MPI, shared by FixConpHip and KSpaceModuleHip: the collectives FixConp makes on `world` (fix_conp.cpp:415, 492,
// 523, 535, 643, 822, 1356; km_ewald.cpp:77, 784), handed to the library as callbacks.  ctx = &world.
// The library counts in int64_t (the sharded A build sums Ne*Ne doubles, the row-sharded inverse gathers Ne/N * Ne doubles per
// rank); MPI's classic interface counts in int.  Nothing is narrowed silently: sums go out in pieces of at most 2^30 elements,
// gathers switch to 8-byte units when every count and displacement allows it and REFUSE (non-zero status -> conp_last_error ->
// error->all) what still does not fit.
#pragma once
#include <climits>
#include <cstdint>
#include <vector>

#ifdef CONP_GLUE_MOCK
#include "mock_lammps/mpi_mock.h"
#else
#include <mpi.h>
#endif

namespace conp_glue {

inline int cb_allreduce_sum(void *ctx, double *buf, int64_t n) {
  MPI_Comm w = *static_cast<MPI_Comm *>(ctx);
  const int64_t piece = (int64_t)1 << 30;
  for (int64_t off = 0; off < n; off += piece) {
    const int64_t m = n - off < piece ? n - off : piece;
    if (MPI_Allreduce(MPI_IN_PLACE, buf + off, (int)m, MPI_DOUBLE, MPI_SUM, w) != MPI_SUCCESS) return 1;
  }
  return 0;
}
inline int cb_allreduce_max_int(void *ctx, int *buf, int n) {
  return MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_INT, MPI_MAX, *static_cast<MPI_Comm *>(ctx)) != MPI_SUCCESS;
}
inline int cb_allgather_int(void *ctx, int value, int *out) {
  return MPI_Allgather(&value, 1, MPI_INT, out, 1, MPI_INT, *static_cast<MPI_Comm *>(ctx)) != MPI_SUCCESS;
}
inline int cb_allgatherv(void *ctx, const void *send, int64_t nbytes, void *recv, const int64_t *counts, const int64_t *displs) {
  MPI_Comm w = *static_cast<MPI_Comm *>(ctx);
  int n = 1;
  MPI_Comm_size(w, &n);
  // bytes if they fit, else doubles (everything the library gathers is made of 8-byte items), else refuse
  int64_t unit = 1;
  for (int pass = 0; pass < 2; ++pass) {
    bool ok = nbytes % unit == 0 && nbytes / unit <= INT_MAX;
    for (int r = 0; r < n && ok; ++r)
      ok = counts[r] % unit == 0 && displs[r] % unit == 0 && counts[r] / unit <= INT_MAX && displs[r] / unit <= INT_MAX;
    if (ok) break;
    if (unit == 8) return 2;                    // would be silent corruption: say so instead
    unit = 8;
  }
  std::vector<int> c(n), d(n);
  for (int r = 0; r < n; ++r) { c[r] = (int)(counts[r] / unit); d[r] = (int)(displs[r] / unit); }
  return MPI_Allgatherv(send, (int)(nbytes / unit), unit == 8 ? MPI_DOUBLE : MPI_BYTE, recv, c.data(), d.data(),
                        unit == 8 ? MPI_DOUBLE : MPI_BYTE, w) != MPI_SUCCESS;
}

// rank of this process among the ranks of its NODE: the library maps it onto the node's GPUs (conp_env.rank keeps the global
// rank for the shard arithmetic; conp_env.device = -(2 + local rank) asks for "local rank modulo the visible devices")
inline int node_local_rank(MPI_Comm w) {
#ifdef CONP_GLUE_MOCK
  int r = 0;
  MPI_Comm_rank(w, &r);
  return r;                                     // rank threads of one process: one node
#else
  MPI_Comm node;
  int r = 0;
  if (MPI_Comm_split_type(w, MPI_COMM_TYPE_SHARED, 0, MPI_INFO_NULL, &node) != MPI_SUCCESS) { MPI_Comm_rank(w, &r); return r; }
  MPI_Comm_rank(node, &r);
  MPI_Comm_free(&node);
  return r;
#endif
}

}  // namespace conp_glue
