// Electrode-row regrouping of the LAMMPS half list ON THE DEVICE (the membership rule of blist_coul_cal, fix_conp.cpp:1326-1350).
// Input: the flattened list as uploaded for the post-force kernel (ilist, numneigh, first, neigh) and atom2eleall (eleall row of
// every owned / ghost electrode atom, -1 otherwise).  Output: the CSR rows b_real_combine walks -- row_ptr[Ne+1], ele_atom[P],
// oth_atom[P] -- with the pairs of a row in LIST ORDER, exactly what the host counting sort (conp_host.cpp build_b_rows)
// produces: that order fixes the summation tree of a row, hence the bits of b.
//   1. one wavefront per list owner counts its qualifying pairs
//   2. exclusive scan over the owners
//   3. one wavefront per owner writes (row, electrode atom, partner) of its qualifying pairs at its offset, neighbours in order
//      (ballot + popcount ranks inside a 64-neighbour chunk)  -> pairs in list order
//   4. STABLE radix sort by row (rocPRIM) -> list order survives inside a row
//   5. row_ptr[r] = first position with key >= r (binary search per row)
// Took 0.8-1.0 ms on 8 host threads (1.9 ms on one) at 2.6e5-3.5e5 listed pairs; here it is a few launches.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "conp_kernels.h"

namespace conp {

namespace {
constexpr int NEIGHMASK_DEV = 0x3FFFFFFF;

// row of the pair (owner i, neighbour j), or -1 (fix_conp.cpp:1326-1350)
__device__ __forceinline__ int pair_row(int ri, int rj, int j, int nlocal, int newton) {
  if (ri >= 0) return rj < 0 ? ri : -1;
  return (rj >= 0 && (newton || j < nlocal)) ? rj : -1;
}

__global__ __launch_bounds__(256) void rows_count_kernel(int inum, const int *__restrict__ ilist, const int *__restrict__ numneigh,
                                                         const int *__restrict__ first, const int *__restrict__ neigh,
                                                         const int *__restrict__ arow, int nlocal, int newton,
                                                         unsigned *__restrict__ count) {
  const int ii = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ii >= inum) return;
  const int i = ilist[ii], ri = arow[i];
  const int *jl = neigh + first[i];
  const int jn = numneigh[i];
  unsigned n = 0;
  for (int jj = lane; jj < jn; jj += 64) {
    const int j = jl[jj] & NEIGHMASK_DEV;
    n += pair_row(ri, arow[j], j, nlocal, newton) >= 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
  if (lane == 0) count[ii] = n;
}

__global__ __launch_bounds__(256) void rows_emit_kernel(int inum, const int *__restrict__ ilist, const int *__restrict__ numneigh,
                                                        const int *__restrict__ first, const int *__restrict__ neigh,
                                                        const int *__restrict__ arow, int nlocal, int newton,
                                                        const unsigned *__restrict__ base, unsigned *__restrict__ key,
                                                        unsigned long long *__restrict__ val) {
  const int ii = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ii >= inum) return;
  const int i = ilist[ii], ri = arow[i];
  const int *jl = neigh + first[i];
  const int jn = numneigh[i];
  unsigned pos = base[ii];
  for (int j0 = 0; j0 < jn; j0 += 64) {
    const int jj = j0 + lane;
    int row = -1, j = 0;
    if (jj < jn) { j = jl[jj] & NEIGHMASK_DEV; row = pair_row(ri, arow[j], j, nlocal, newton); }
    const unsigned long long m = __ballot(row >= 0);
    if (row >= 0) {
      const unsigned p = pos + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
      key[p] = (unsigned)row;
      const unsigned ele = ri >= 0 ? (unsigned)i : (unsigned)j, oth = ri >= 0 ? (unsigned)j : (unsigned)i;
      val[p] = ((unsigned long long)ele << 32) | oth;
    }
    pos += (unsigned)__popcll(m);
  }
}

__global__ void rows_unpack_kernel(unsigned np, const unsigned long long *__restrict__ val, int *__restrict__ ele,
                                   int *__restrict__ oth) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  const unsigned long long v = val[p];
  ele[p] = (int)(v >> 32);
  oth[p] = (int)(v & 0xffffffffull);
}

__global__ void rows_ptr_kernel(int ne, unsigned np, const unsigned *__restrict__ key, int *__restrict__ row_ptr) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > ne) return;
  unsigned lo = 0, hi = np;                       // first position with key >= r
  while (lo < hi) {
    const unsigned mid = (lo + hi) >> 1;
    if (key[mid] < (unsigned)r) lo = mid + 1; else hi = mid;
  }
  row_ptr[r] = (int)lo;
}
}  // namespace

// scratch layout (bytes), all sized for `nneigh` listed pairs and `inum` owners; returns what launch_build_b_rows needs
size_t b_rows_scratch_bytes(int inum, size_t nneigh) {
  size_t tmp_scan = 0, tmp_sort = 0;
  (void)rocprim::exclusive_scan(nullptr, tmp_scan, (unsigned *)nullptr, (unsigned *)nullptr, 0u, (size_t)inum + 1, rocprim::plus<unsigned>());
  (void)rocprim::radix_sort_pairs(nullptr, tmp_sort, (unsigned *)nullptr, (unsigned *)nullptr, (unsigned long long *)nullptr,
                                  (unsigned long long *)nullptr, nneigh, 0, 32);
  const size_t a = 256;
  auto up = [&](size_t b) { return (b + a - 1) / a * a; };
  return up(((size_t)inum + 1) * 4) * 2 + up(nneigh * 4) * 2 + up(nneigh * 8) * 2 + up(tmp_scan > tmp_sort ? tmp_scan : tmp_sort) + 4 * a;
}

// returns the number of pairs (one blocking read of the scan total); row_ptr / ele / oth must hold ne + 1 / nneigh / nneigh ints
int64_t launch_build_b_rows(hipStream_t s, int inum, size_t nneigh, const int *ilist, const int *numneigh, const int *first,
                            const int *neigh, const int *arow, int nlocal, int newton, int ne, void *scratch, size_t scratch_bytes,
                            int *row_ptr, int *ele, int *oth) {
  if (inum <= 0 || nneigh == 0) { (void)hipMemsetAsync(row_ptr, 0, ((size_t)ne + 1) * sizeof(int), s); return 0; }
  const size_t a = 256;
  auto up = [&](size_t b) { return (b + a - 1) / a * a; };
  char *p = static_cast<char *>(scratch);
  unsigned *count = reinterpret_cast<unsigned *>(p); p += up(((size_t)inum + 1) * 4);
  unsigned *base = reinterpret_cast<unsigned *>(p); p += up(((size_t)inum + 1) * 4);
  unsigned *key_in = reinterpret_cast<unsigned *>(p); p += up(nneigh * 4);
  unsigned *key_out = reinterpret_cast<unsigned *>(p); p += up(nneigh * 4);
  unsigned long long *val_in = reinterpret_cast<unsigned long long *>(p); p += up(nneigh * 8);
  unsigned long long *val_out = reinterpret_cast<unsigned long long *>(p); p += up(nneigh * 8);
  void *tmp = p;
  size_t tmp_bytes = scratch_bytes - (size_t)(p - static_cast<char *>(scratch));
  (void)hipMemsetAsync(count + inum, 0, sizeof(unsigned), s);            // the extra element makes base[inum] the total
  hipLaunchKernelGGL(rows_count_kernel, dim3((inum + 3) / 4), dim3(256), 0, s, inum, ilist, numneigh, first, neigh, arow, nlocal, newton,
                     count);
  size_t need = tmp_bytes;
  (void)rocprim::exclusive_scan(tmp, need, count, base, 0u, (size_t)inum + 1, rocprim::plus<unsigned>(), s);
  unsigned np = 0;
  (void)hipMemcpyAsync(&np, base + inum, sizeof(unsigned), hipMemcpyDeviceToHost, s);
  (void)hipStreamSynchronize(s);
  if (np == 0) { (void)hipMemsetAsync(row_ptr, 0, ((size_t)ne + 1) * sizeof(int), s); return 0; }
  hipLaunchKernelGGL(rows_emit_kernel, dim3((inum + 3) / 4), dim3(256), 0, s, inum, ilist, numneigh, first, neigh, arow, nlocal, newton,
                     base, key_in, val_in);
  int bits = 1;
  while ((1 << bits) < ne + 1 && bits < 31) ++bits;
  need = tmp_bytes;
  (void)rocprim::radix_sort_pairs(tmp, need, key_in, key_out, val_in, val_out, (size_t)np, 0, bits, s);
  hipLaunchKernelGGL(rows_unpack_kernel, dim3((np + 255) / 256), dim3(256), 0, s, np, val_out, ele, oth);
  hipLaunchKernelGGL(rows_ptr_kernel, dim3((ne + 1 + 255) / 256), dim3(256), 0, s, ne, np, key_out, row_ptr);
  return (int64_t)np;
}

}  // namespace conp
