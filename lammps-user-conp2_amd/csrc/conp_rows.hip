// Electrode-row regrouping of the LAMMPS half list ON THE DEVICE (the membership rule of blist_coul_cal, fix_conp.cpp:1326-1350).
// Input: the flattened list as uploaded for the post-force kernel (ilist, numneigh, first, neigh) and atom2eleall (eleall row of
// every owned / ghost electrode atom, -1 otherwise).  Output: the CSR rows b_real_combine walks -- row_ptr[Ne+1], ele_atom[P],
// oth_atom[P] -- with the pairs of a row in LIST ORDER (owner after owner, neighbour after neighbour), exactly what the host
// counting sort (conp_host.cpp build_b_rows) produces: that order fixes the summation tree of a row, hence the bits of b.
//
// Hand-written, four launches, no library sort (round 2 went through rocPRIM's scan + stable radix sort: ~135 us of launches and
// a blocking read-back whatever the list's size):
//   1. count    one wavefront per list owner: its qualifying pairs, one integer atomic per pair into the row histogram
//   2. scan     one workgroup: exclusive scan of the histogram -> row_ptr (and the total, copied to the host without a sync)
//   3. scatter  one wavefront per owner: every qualifying pair takes the next free slot of its row (integer atomic: the slot
//               ORDER inside a row is arbitrary here) and leaves (list position, electrode atom, partner) there
//   4. order    one wavefront per row: rank of every entry = number of entries of the row with a smaller list position (rows are
//               short: ~10^2 pairs), written to its final place.  List positions are unique, so whatever order step 3 produced,
//               the rows come out in list order -- deterministic, bit-for-bit the host's rows.
#include <hip/hip_runtime.h>

#include <cstring>

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
                                                         unsigned *__restrict__ hist /*[ne + 1], zeroed*/) {
  const int ii = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ii >= inum) return;
  const int i = ilist[ii], ri = arow[i];
  const int *jl = neigh + first[i];
  const int jn = numneigh[i];
  if (ri >= 0) {
    // the owner is the electrode atom: all its pairs share one row -- one atomic for the lot
    unsigned n = 0;
    for (int jj = lane; jj < jn; jj += 64) n += arow[jl[jj] & NEIGHMASK_DEV] < 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
    if (lane == 0 && n) atomicAdd(hist + ri, n);
    return;
  }
  for (int jj = lane; jj < jn; jj += 64) {
    const int j = jl[jj] & NEIGHMASK_DEV;
    const int row = pair_row(ri, arow[j], j, nlocal, newton);
    if (row >= 0) atomicAdd(hist + row, 1u);
  }
}

// exclusive scan of hist[0 .. n) into row_ptr[0 .. n]; fill[] = row_ptr[] (the scatter's running slot per row); one workgroup
__global__ __launch_bounds__(1024) void rows_scan_kernel(int n, const unsigned *__restrict__ hist, int *__restrict__ row_ptr,
                                                         unsigned *__restrict__ fill, unsigned *__restrict__ total) {
  __shared__ unsigned wsum[16];
  __shared__ unsigned carry;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int k = base + t;
    const unsigned v = k < n ? hist[k] : 0u;
    unsigned s = v;                                   // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = __shfl_up(s, off, 64);
      if (lane >= off) s += o;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    unsigned before = carry;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (k < n) { row_ptr[k] = (int)(before + s - v); fill[k] = before + s - v; }
    __syncthreads();
    if (t == 1023) carry = before + s;
    __syncthreads();
  }
  if (t == 0) { row_ptr[n] = (int)carry; *total = carry; }
}

__global__ __launch_bounds__(256) void rows_scatter_kernel(int inum, const int *__restrict__ ilist, const int *__restrict__ numneigh,
                                                           const int *__restrict__ first, const int *__restrict__ neigh,
                                                           const int *__restrict__ arow, int nlocal, int newton,
                                                           unsigned *__restrict__ fill, unsigned long long *__restrict__ key,
                                                           unsigned long long *__restrict__ val) {
  const int ii = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ii >= inum) return;
  const int i = ilist[ii], ri = arow[i];
  const int *jl = neigh + first[i];
  const int jn = numneigh[i];
  for (int j0 = 0; j0 < jn; j0 += 64) {
    const int jj = j0 + lane;
    int row = -1, j = 0;
    if (jj < jn) { j = jl[jj] & NEIGHMASK_DEV; row = pair_row(ri, arow[j], j, nlocal, newton); }
    unsigned slot = 0;
    if (ri >= 0) {
      // one row for the whole wavefront: one atomic, slots in lane order
      const unsigned long long m = __ballot(row >= 0);
      unsigned b = 0;
      if (lane == 0 && m) b = atomicAdd(fill + ri, (unsigned)__popcll(m));
      b = __shfl(b, 0, 64);
      slot = b + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    } else if (row >= 0) {
      slot = atomicAdd(fill + row, 1u);
    }
    if (row >= 0) {
      key[slot] = ((unsigned long long)(unsigned)ii << 32) | (unsigned)jj;        // position in the list: unique
      const unsigned ele = ri >= 0 ? (unsigned)i : (unsigned)j, oth = ri >= 0 ? (unsigned)j : (unsigned)i;
      val[slot] = ((unsigned long long)ele << 32) | oth;
    }
  }
}

// one wavefront per row: entries ranked by list position
__global__ __launch_bounds__(256) void rows_order_kernel(int ne, const int *__restrict__ row_ptr, const unsigned long long *__restrict__ key,
                                                         const unsigned long long *__restrict__ val, int *__restrict__ ele,
                                                         int *__restrict__ oth) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= ne) return;
  const int b = row_ptr[r], n = row_ptr[r + 1] - b;
  for (int e0 = 0; e0 < n; e0 += 64) {
    const int e = e0 + lane;
    const unsigned long long ke = e < n ? key[b + e] : ~0ull;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 64) {
      // 64 keys of the row per round, one per lane, handed round by shuffles (every lane compares with all of them)
      const unsigned long long kj = j0 + lane < n ? key[b + j0 + lane] : ~0ull;
      const int cnt = n - j0 < 64 ? n - j0 : 64;
      for (int u = 0; u < cnt; ++u) {
        const unsigned long long k = __shfl(kj, u, 64);
        rank += k < ke;
      }
    }
    if (e < n) {
      const unsigned long long v = val[b + e];
      ele[b + rank] = (int)(v >> 32);
      oth[b + rank] = (int)(v & 0xffffffffull);
    }
  }
}
}  // namespace

// scratch layout (bytes), sized for `nneigh` listed pairs and `ne` rows
size_t b_rows_scratch_bytes(int ne, size_t nneigh) {
  const size_t a = 256;
  auto up = [&](size_t b) { return (b + a - 1) / a * a; };
  return up(((size_t)ne + 2) * 4) * 2 + up(nneigh * 8) * 2 + 4 * a;
}

// rows from the list; no host synchronisation: the number of pairs lands in *np_pinned (page-locked host memory) once the stream
// has passed this point.  row_ptr / ele / oth must hold ne + 1 / nneigh / nneigh ints.
void launch_build_b_rows(hipStream_t s, int inum, size_t nneigh, const int *ilist, const int *numneigh, const int *first,
                         const int *neigh, const int *arow, int nlocal, int newton, int ne, void *scratch, size_t scratch_bytes,
                         int *row_ptr, int *ele, int *oth, unsigned *np_pinned) {
  (void)scratch_bytes;
  *np_pinned = 0;
  if (inum <= 0 || nneigh == 0) { (void)hipMemsetAsync(row_ptr, 0, ((size_t)ne + 1) * sizeof(int), s); return; }
  const size_t a = 256;
  auto up = [&](size_t b) { return (b + a - 1) / a * a; };
  char *p = static_cast<char *>(scratch);
  unsigned *hist = reinterpret_cast<unsigned *>(p); p += up(((size_t)ne + 2) * 4);      // [ne] rows
  unsigned *fill = reinterpret_cast<unsigned *>(p); p += up(((size_t)ne + 2) * 4);      // [ne] running slots, [ne + 1] the total
  unsigned long long *key = reinterpret_cast<unsigned long long *>(p); p += up(nneigh * 8);
  unsigned long long *val = reinterpret_cast<unsigned long long *>(p); p += up(nneigh * 8);
  (void)hipMemsetAsync(hist, 0, ((size_t)ne + 2) * sizeof(unsigned), s);
  hipLaunchKernelGGL(rows_count_kernel, dim3((inum + 3) / 4), dim3(256), 0, s, inum, ilist, numneigh, first, neigh, arow, nlocal, newton, hist);
  hipLaunchKernelGGL(rows_scan_kernel, dim3(1), dim3(1024), 0, s, ne, hist, row_ptr, fill, fill + ne + 1);
  (void)hipMemcpyAsync(np_pinned, fill + ne + 1, sizeof(unsigned), hipMemcpyDeviceToHost, s);
  hipLaunchKernelGGL(rows_scatter_kernel, dim3((inum + 3) / 4), dim3(256), 0, s, inum, ilist, numneigh, first, neigh, arow, nlocal, newton,
                     fill, key, val);
  hipLaunchKernelGGL(rows_order_kernel, dim3((ne + 3) / 4), dim3(256), 0, s, ne, row_ptr, key, val, ele, oth);
}

// atom2eleall[atom] = eleall row for the listed (atom, eleall) pairs, -1 everywhere else: built here from the pair list (a few
// thousand entries) instead of uploaded ([nall] ints, 360 KB at the headline size, filled on the host at every re-neighbour)
__global__ __launch_bounds__(256) void atom2eleall_kernel(int npairs, const int *__restrict__ pairs, int *__restrict__ atom2eleall) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < npairs) atom2eleall[pairs[2 * k]] = pairs[2 * k + 1];
}
void launch_atom2eleall(hipStream_t s, int nall, int npairs, const int *pairs, int *atom2eleall) {
  if (nall <= 0) return;
  (void)hipMemsetAsync(atom2eleall, 0xFF, (size_t)nall * sizeof(int), s);      // all bits set: -1
  if (npairs > 0) hipLaunchKernelGGL(atom2eleall_kernel, dim3((npairs + 255) / 256), dim3(256), 0, s, npairs, pairs, atom2eleall);
}

}  // namespace conp
