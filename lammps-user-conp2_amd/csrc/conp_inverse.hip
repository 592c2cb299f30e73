// In-place dense inverse of the electrode matrix on gfx950: blocked Gauss-Jordan elimination with partial (row)
// pivoting, FP64, trailing updates on the matrix cores (v_mfma_f64_16x16x4_f64).
// It stands where the reference calls LAPACK dgetrf_/dgetri_ (fix_conp.cpp:947-949); like those it handles a general
// (not necessarily positive definite) matrix.  Row-major storage, leading dimension n.
//
// One block step (NB = 64 columns K = [k0, k0+NB)):
//   1. LU-factor the panel (rows k0.., columns K) with partial pivoting -> pivots and the factored top 64 x 64 block:
//      up to 256 workgroups working together, rows resident in LDS, one grid barrier per column (1c); a one-workgroup version
//      (1a + 1b) remains as the fallback (CONP_PANEL_SINGLE, or when a barrier ever timed out)
//   2. apply the row swaps to the whole matrix
//   3. Dinv = (M[K,K])^-1 from the panel's L11, U11 (one workgroup, LDS)
//   4. Wb = Dinv * M[K,:] with the K columns zeroed;  Cct = M[:,K]^T with the K rows zeroed      (both k-major)
//   5. M -= Cct^T * Wb        (MFMA, all rows/cols; the zeroed blocks keep rows K / cols K untouched)
//   6. M[i,K] = -C[i,:] * Dinv (i not in K);  M[K,j] = Wb[:,j] (j not in K);  M[K,K] = Dinv
// After the last step the row swaps are undone as column swaps in reverse order: inv(A) = inv(P A) P.
//
// Symmetric positive definite matrices (the electrode matrix IS one: the Hessian of the Gaussian-charge Ewald energy, exactly
// symmetric after a_symmetrise -- the reference's own CG solver relies on that) take the same block steps WITHOUT steps 1 and 2:
// every diagonal block of a Schur complement of an SPD matrix is SPD, so the 64 x 64 block is inverted in place with its own
// diagonal as pivots (inv_block_spd_kernel) and no row ever moves -- no pivot search over the panel, no grid barrier, no swaps.
// A pivot that is not positive (the matrix was not SPD after all) sets info = -8: the caller restores the matrix and runs the
// pivoted elimination above.  launch_inverse_spd is tried first whenever the matrix is exactly symmetric.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "conp_kernels.h"

namespace conp {

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
constexpr int INV_NB = 64;

// ---- 1a. panel copy: P[(i - k0) * NB + c] = M[i][k0 + c]  for i >= k0 --------------------------------------------
__global__ void inv_panel_copy_kernel(int n, int k0, int nbw, const double *__restrict__ M, double *__restrict__ P,
                                      const int *__restrict__ info) {
  if (*info != 0) return;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tot = (size_t)(n - k0) * INV_NB;
  if (e >= tot) return;
  const int r = (int)(e / INV_NB), c = (int)(e % INV_NB);
  P[e] = (c < nbw) ? M[(size_t)(k0 + r) * n + k0 + c] : 0.0;
}

// ---- 1b. panel LU with partial pivoting, one workgroup of 1024 threads ------------------------------------------
// P is (m x NB) row-major, m = n - k0.  piv[j] = absolute row index swapped with row k0 + j.  info != 0 on a zero pivot.
__global__ __launch_bounds__(1024) void inv_panel_lu_kernel(int m, int k0, int nbw, double *__restrict__ P, int *__restrict__ piv,
                                                            int *__restrict__ info) {
  if (*info != 0) return;
  __shared__ double s_val[16];
  __shared__ int s_idx[16];
  __shared__ double s_row[INV_NB];
  __shared__ int s_p;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int j = 0; j < nbw; ++j) {
    // pivot search in column j over rows j..m-1 (first maximum wins, like LAPACK idamax)
    double best = -1.0;
    int bi = j;
    for (int i = j + t; i < m; i += 1024) {
      const double v = fabs(P[(size_t)i * INV_NB + j]);
      if (v > best) { best = v; bi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_down(best, off, 64);
      const int oi = __shfl_down(bi, off, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; }
    __syncthreads();
    if (t == 0) {
      double b = s_val[0];
      int p = s_idx[0];
      for (int w = 1; w < 16; ++w)
        if (s_val[w] > b || (s_val[w] == b && s_idx[w] < p)) { b = s_val[w]; p = s_idx[w]; }
      const bool bad = !(b > 0.0) || !(b <= 1.7976931348623157e308);
      s_p = bad ? -1 : p;
      piv[j] = k0 + (bad ? j : p);
      if (bad) *info = k0 + j + 1;
    }
    __syncthreads();
    if (s_p < 0) return;                       // zero / non-finite pivot: "Inversion failed!" (fix_conp.cpp:956)
    const int p = s_p;
    // swap rows j and p of the panel, keep the pivot row in LDS
    if (t < INV_NB) {
      const double a = P[(size_t)p * INV_NB + t];
      if (p != j) {
        const double b = P[(size_t)j * INV_NB + t];
        P[(size_t)p * INV_NB + t] = b;
        P[(size_t)j * INV_NB + t] = a;
      }
      s_row[t] = a;
    }
    __syncthreads();
    const double dinv = 1.0 / s_row[j];
    // scale the column below the pivot and update the trailing panel columns; 8 rows per wave in flight (the loop is
    // bound by what ONE CU can stream (~25-50 GB/s): the panel step moves m x 64 doubles per column, 3.7 ms per 4096-row
    // panel -- which is why the cooperative kernel 1c is the default)
    for (int i0 = j + 1 + 8 * (t >> 6); i0 < m; i0 += 16 * 8) {
      double lv[8], rv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u;
        const bool ok = i < m;
        const double *row = P + (size_t)(ok ? i : j) * INV_NB;
        lv[u] = row[j];
        rv[u] = row[lane];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u;
        if (i < m) {
          double *row = P + (size_t)i * INV_NB;
          const double l = lv[u] * dinv;
          if (lane > j && lane < nbw) row[lane] = rv[u] - l * s_row[lane];
          if (lane == j) row[j] = l;
        }
      }
    }
    __syncthreads();
  }
}

// ---- 1c. the same panel factorisation spread over many workgroups (cooperative launch) ---------------------------
// The single-workgroup panel above is bound by what one CU can stream (3.9 ms per 4096-row panel, 15.7 ms per 16384-row
// panel: 57 % of the inverse at Ne = 4096, 80 % at 16384).  Here workgroup w keeps panel rows [w rpw, (w+1) rpw) in LDS for
// the whole panel and the workgroups meet ONCE per column:
//   before the barrier  every workgroup publishes its pivot candidate for column j (|value|, row index, the 64-wide row);
//                       the owner of row j publishes row j
//   after the barrier   every workgroup picks the same winner (largest |value|, lowest row index on ties = LAPACK idamax),
//                       reads the winner's row, the owners of rows j / p exchange them in LDS, everybody eliminates
// Slots are double-buffered by column parity (a workgroup can be at most one barrier ahead).  Only the pivots and the top
// 64 x 64 block (L11, U11) leave the kernel -- nothing downstream reads L21.  The barrier is a monotonic agent-scope counter
// polled with sc1 loads; the exchanged slots are sc1 stores / sc1 loads, no cache-wide fences.  The spin is bounded so that
// every wave terminates even if a workgroup were not resident (info = -7).
constexpr int PC_MAXG = 256;
constexpr int PC_LD = INV_NB + 1;

__device__ __forceinline__ bool pc_better(double v, int i, double bv, int bi) { return v > bv || (v == bv && i < bi); }

__global__ __launch_bounds__(256) void inv_panel_coop_kernel(int n, int k0, int nbw, int rpw, const double *__restrict__ M,
                                                             double *__restrict__ P, int *__restrict__ piv, int *__restrict__ info,
                                                             double *__restrict__ cval, int *__restrict__ cidx,
                                                             double *__restrict__ crow, double *__restrict__ rowj,
                                                             unsigned *__restrict__ counter, unsigned spin_limit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *rows = reinterpret_cast<double *>(smem);          // [rpw][PC_LD]
  double *s_row = rows + (size_t)rpw * PC_LD;               // [64]
  double *s_rv = s_row + INV_NB;                            // [4]
  int *s_ri = reinterpret_cast<int *>(s_rv + 4);            // [4] candidate row, [4] candidate workgroup
  __shared__ int s_p, s_win, s_abort;
  __shared__ double s_best;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m = n - k0, G = gridDim.x, wg = blockIdx.x;
  if (*info != 0) return;                  // an earlier panel failed (uniform: that panel's kernel has completed)
  const int r0 = wg * rpw;
  const int nr = (m - r0 < rpw) ? m - r0 : rpw;
  if (t == 0) s_abort = 0;
  for (int e = t; e < nr * INV_NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    rows[r * PC_LD + c] = (c < nbw) ? M[(size_t)(k0 + r0 + r) * n + k0 + c] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < nbw; ++j) {
    const int par = j & 1;
    if (wave == 0) {                       // this workgroup's candidate: first largest |rows[.][j]| among rows r0 + r >= j
      double best = -1.0;
      int bi = 0x7fffffff;
      for (int r = lane; r < nr; r += 64)
        if (r0 + r >= j) {
          const double v = fabs(rows[r * PC_LD + j]);
          if (pc_better(v, r0 + r, best, bi)) { best = v; bi = r0 + r; }
        }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const int oi = __shfl_down(bi, off, 64);
        if (pc_better(ov, oi, best, bi)) { best = ov; bi = oi; }
      }
      best = __shfl(best, 0, 64); bi = __shfl(bi, 0, 64);
      // Hand-off without fences (an agent-scope release / acquire pair writes back and invalidates whole caches: 3.4 - 8 us per
      // column, i.e. most of the panel's time): every published value is a write-through (sc1) store, the storing waves drain
      // their stores (vmcnt 0) before the workgroup's ONE ticket add, and every read of a published value below is an sc1 load
      // (MI355X_MICROARCH.md, inter-workgroup visibility: "sc1 stores, vmcnt(0), barrier, agent atomic add; sc1 loads").
      if (lane == 0) {
        __hip_atomic_store(cval + par * PC_MAXG + wg, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cidx + par * PC_MAXG + wg, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (best >= 0.0)
        __hip_atomic_store(crow + ((size_t)par * PC_MAXG + wg) * INV_NB + lane, rows[(bi - r0) * PC_LD + lane], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (wave == 1 && j >= r0 && j < r0 + nr) {
      __hip_atomic_store(rowj + par * INV_NB + lane, rows[(j - r0) * PC_LD + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (t == 0) {                          // grid barrier #(j+1)
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)G * (unsigned)(j + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > spin_limit) { s_abort = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (s_abort) { if (t == 0) *info = -7; return; }
    // the same winner in every workgroup
    double v = -2.0;
    int vi = 0x7fffffff, vw = 0;
    if (t < G) {
      v = __hip_atomic_load(cval + par * PC_MAXG + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      vi = __hip_atomic_load(cidx + par * PC_MAXG + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      vw = t;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_down(v, off, 64);
      const int oi = __shfl_down(vi, off, 64), ow = __shfl_down(vw, off, 64);
      if (pc_better(ov, oi, v, vi)) { v = ov; vi = oi; vw = ow; }
    }
    if (lane == 0) { s_rv[wave] = v; s_ri[wave] = vi; s_ri[4 + wave] = vw; }
    __syncthreads();
    if (t == 0) {
      double b = s_rv[0];
      int pi = s_ri[0], pw = s_ri[4];
      for (int w = 1; w < 4; ++w)
        if (pc_better(s_rv[w], s_ri[w], b, pi)) { b = s_rv[w]; pi = s_ri[w]; pw = s_ri[4 + w]; }
      // no usable pivot in this column (exact zero, or NaN / inf from an earlier column: no candidate is ever "better" than
      // the initial -1): the reference's dgetrf_ reports it and FixConp::inv aborts with "Inversion failed!" (fix_conp.cpp:956).
      // Every workgroup sees the same candidates, takes the same decision and leaves before the next barrier.
      const bool bad = !(b > 0.0) || !(b <= 1.7976931348623157e308) || pi < j || pi >= m;
      s_best = bad ? -1.0 : b; s_p = bad ? j : pi; s_win = pw;
      if (wg == 0) {
        piv[j] = k0 + (bad ? j : pi);
        if (bad) *info = k0 + j + 1;
      }
    }
    __syncthreads();
    if (!(s_best > 0.0)) return;
    const int p = s_p;
    if (wave == 0)
      s_row[lane] = __hip_atomic_load(crow + ((size_t)par * PC_MAXG + s_win) * INV_NB + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (p != j) {                          // rows j and p change places
      if (wave == 1 && p >= r0 && p < r0 + nr)
        rows[(p - r0) * PC_LD + lane] = __hip_atomic_load(rowj + par * INV_NB + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (wave == 2 && j >= r0 && j < r0 + nr) rows[(j - r0) * PC_LD + lane] = s_row[lane];
    }
    __syncthreads();
    const double dinv = 1.0 / s_row[j];
    for (int r = wave; r < nr; r += 4) {
      if (r0 + r <= j) continue;
      double *row = rows + r * PC_LD;
      const double l = row[j] * dinv;
      const double cur = row[lane];
      if (lane > j && lane < nbw) row[lane] = cur - l * s_row[lane];
      if (lane == j) row[j] = l;
    }
    __syncthreads();
  }
  // the factored top block (rows < 64 of the panel) for inv_block_kernel
  for (int e = t; e < nr * INV_NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    if (r0 + r < INV_NB) P[(size_t)(r0 + r) * INV_NB + c] = rows[r * PC_LD + c];
  }
}

// ---- 2. row swaps on the whole matrix (each thread owns one column; swaps applied in order) ---------------------
// Every kernel downstream of a panel checks *info first: after a failed pivot the remaining pivots of that panel were never
// written and the matrix content is meaningless -- nothing may be indexed with them (the host reads info once, at the end).
__global__ void inv_row_swaps_kernel(int n, int k0, int nbw, const int *__restrict__ piv, double *__restrict__ M,
                                     const int *__restrict__ info) {
  if (*info != 0) return;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  for (int j = 0; j < nbw; ++j) {
    const int p = piv[j];
    if (p != k0 + j && p >= 0 && p < n) {
      const double a = M[(size_t)(k0 + j) * n + c], b = M[(size_t)p * n + c];
      M[(size_t)(k0 + j) * n + c] = b;
      M[(size_t)p * n + c] = a;
    }
  }
}

// ---- 3. Dinv = U11^-1 L11^-1 from the factored panel top block (one workgroup, 64 x 64 in LDS) ------------------
__global__ __launch_bounds__(256) void inv_block_kernel(int nbw, const double *__restrict__ P, double *__restrict__ Dinv,
                                                        const int *__restrict__ info) {
  if (*info != 0) return;
  __shared__ double LU[INV_NB][INV_NB];   // strict lower = L11 (unit diagonal implied), upper incl. diagonal = U11
  __shared__ double IV[INV_NB][INV_NB];   // strict lower = L11^-1 (unit diagonal implied), upper incl. diagonal = U11^-1
  const int t = threadIdx.x;
  for (int e = t; e < INV_NB * INV_NB; e += 256) {
    const int r = e / INV_NB, c = e % INV_NB;
    const bool in = r < nbw && c < nbw;
    LU[r][c] = in ? P[(size_t)r * INV_NB + c] : (r == c ? 1.0 : 0.0);   // padding block = identity
    IV[r][c] = 0.0;
  }
  __syncthreads();
  // thread c < 64 builds column c of L^-1 (forward substitution), thread 64 + c column c of U^-1 (back substitution);
  // the two triangles do not overlap in IV
  if (t < INV_NB) {
    const int c = t;
    for (int r = c + 1; r < INV_NB; ++r) {
      double s = LU[r][c];                       // k = c term: L[r][c] * Li[c][c], Li[c][c] = 1
      for (int k = c + 1; k < r; ++k) s += LU[r][k] * IV[k][c];
      IV[r][c] = -s;
    }
  } else if (t < 2 * INV_NB) {
    const int c = t - INV_NB;
    IV[c][c] = 1.0 / LU[c][c];
    for (int r = c - 1; r >= 0; --r) {
      double s = 0.0;
      for (int k = r + 1; k <= c; ++k) s += LU[r][k] * IV[k][c];
      IV[r][c] = -s / LU[r][r];
    }
  }
  __syncthreads();
  for (int e = t; e < INV_NB * INV_NB; e += 256) {
    const int r = e / INV_NB, c = e % INV_NB;
    // Dinv = U^-1 L^-1 : sum over k >= max(r, c); L^-1 has a unit diagonal
    double s = 0.0;
    for (int k = (r > c ? r : c); k < INV_NB; ++k) {
      const double li = (k == c) ? 1.0 : IV[k][c];
      s += IV[r][k] * li;
    }
    Dinv[e] = (r < nbw && c < nbw) ? s : 0.0;
  }
}


// ---- SPD path, step 3': Dinv = (M[K,K])^-1 by in-place Gauss-Jordan on the diagonal block, pivots = its own diagonal ------
// One workgroup; the block lives in REGISTERS: thread (r = t >> 2, cg = t & 3) holds B[r][16 cg .. 16 cg + 15].  Per column j the
// owners publish row j and column j through LDS (double-buffered by the parity of j: one barrier per column), then every thread
// updates its 16 values:   row j: B[j][c] = B[j][c] / p (c != j), 1 / p (c = j);   rows r != j: B[r][c] -= B[r][j] B[j][c] / p
// (c != j), B[r][j] = -B[r][j] / p.  A pivot <= 0 or not finite: info = -8 (not positive definite -> pivoted path).
// (A first version kept the block in LDS: 64 LDS accesses per thread and three barriers per column, 160 us per block, half of the
//  whole inverse.)
__global__ __launch_bounds__(256) void inv_block_spd_kernel(int n, int k0, int nbw, const double *__restrict__ M,
                                                            double *__restrict__ Dinv, int *__restrict__ info) {
  if (*info != 0) return;
  __shared__ double s_row[2][INV_NB], s_col[2][INV_NB];
  const int t = threadIdx.x, r = t >> 2, cg = t & 3;
  double bv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c = 16 * cg + k;
    bv[k] = (r < nbw && c < nbw) ? M[(size_t)(k0 + r) * n + k0 + c] : (r == c ? 1.0 : 0.0);     // padding block = identity
  }
  // publish row 0 / column 0
  if (r == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) s_row[0][16 * cg + k] = bv[k];
  }
  if (cg == 0) s_col[0][r] = bv[0];
  __syncthreads();
  bool bad = false;
  for (int j = 0; j < nbw; ++j) {
    const int par = j & 1;
    const double p = s_row[par][j];
    if (!(p > 0.0) || !(p <= 1.7976931348623157e308)) { bad = true; break; }      // (the same value in every thread: uniform)
    const double pinv = 1.0 / p;
    const double f = s_col[par][r];
    const bool prow = r == j;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int c = 16 * cg + k;
      const double rj = s_row[par][c];
      if (prow) bv[k] = (c == j) ? pinv : bv[k] * pinv;
      else bv[k] = (c == j) ? -f * pinv : bv[k] - f * (rj * pinv);
    }
    // publish row j + 1 / column j + 1 of the updated block into the other buffer
    const int jn = j + 1;
    if (jn < nbw) {
      if (r == jn) {
#pragma unroll
        for (int k = 0; k < 16; ++k) s_row[par ^ 1][16 * cg + k] = bv[k];
      }
      if (cg == (jn >> 4)) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k == (jn & 15)) v = bv[k];
        s_col[par ^ 1][r] = v;
      }
    }
    __syncthreads();
  }
  if (bad) { if (t == 0) *info = -8; return; }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c = 16 * cg + k;
    Dinv[r * INV_NB + c] = (r < nbw && c < nbw) ? bv[k] : 0.0;
  }
}

// 1 in *flag when M is exactly symmetric (flag preset to 1; any asymmetric pair clears it)
__global__ void inv_symmetry_kernel(int n, const double *__restrict__ M, int *__restrict__ flag) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)n * n) return;
  const int r = (int)(e / n), c = (int)(e % n);
  if (c > r && M[e] != M[(size_t)c * n + r]) *flag = 0;
}

// ---- 4. Wb[k][j] = sum_l Dinv[k][l] M[k0+l][j] (0 for j in K);  Cct[k][i] = M[i][k0+k] (0 for i in K) --------------
// One workgroup per 64 columns j; thread (jl, kg) forms 16 of the 64 k outputs of its column from an LDS copy of the
// 64 x 64 slab of M; the transposed copy Cct goes through LDS so that reads (along k) and writes (along i) both coalesce.
__global__ __launch_bounds__(256) void inv_prep_kernel(int n, int ld, int k0, int nbw, const double *__restrict__ M,
                                                       const double *__restrict__ Dinv, double *__restrict__ Wb,
                                                       double *__restrict__ Cct, const int *__restrict__ info) {
  if (*info != 0) return;
  extern __shared__ __attribute__((aligned(16))) char inv_smem[];
  double (*D)[INV_NB + 1] = reinterpret_cast<double (*)[INV_NB + 1]>(inv_smem);
  double (*Cc)[INV_NB + 1] = D + INV_NB;
  const int t = threadIdx.x, jl = t & 63, kg = t >> 6;
  const int j0 = blockIdx.x * INV_NB;
  for (int e = t; e < INV_NB * INV_NB; e += 256) {
    const int l = e >> 6, c = e & 63, j = j0 + c;
    D[l][c] = Dinv[e];
    const bool use = l < nbw && j < n && !(j >= k0 && j < k0 + nbw);
    Cc[l][c] = use ? M[(size_t)(k0 + l) * n + j] : 0.0;
  }
  __syncthreads();
  if (j0 + jl < ld) {
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
      const int k = kg * 16 + kk;
      double s = 0.0;
#pragma unroll 16
      for (int l = 0; l < INV_NB; ++l) s += D[k][l] * Cc[l][jl];
      Wb[(size_t)k * ld + j0 + jl] = s;
    }
  }
  __syncthreads();
  for (int e = t; e < INV_NB * INV_NB; e += 256) {       // rows i = j0 + r of M, columns K, read along k
    const int r = e >> 6, k = e & 63, i = j0 + r;
    const bool use = k < nbw && i < n && !(i >= k0 && i < k0 + nbw);
    Cc[k][r] = use ? M[(size_t)i * n + k0 + k] : 0.0;
  }
  __syncthreads();
  for (int e = t; e < INV_NB * INV_NB; e += 256) {
    const int k = e >> 6, r = e & 63;
    if (j0 + r < ld) Cct[(size_t)k * ld + j0 + r] = Cc[k][r];
  }
}

// ---- 5. M -= Cct^T * Wb on the matrix cores: workgroup 128 x 128, wave 64 x 64 (4 x 4 fragments), K = 64 ----------
__global__ __launch_bounds__(256, 2) void inv_update_kernel(int n, int ld, const double *__restrict__ Cct,
                                                            const double *__restrict__ Wb, double *__restrict__ M,
                                                            const int *__restrict__ info) {
  if (*info != 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int i0 = blockIdx.y * 128 + (wave >> 1) * 64, j0 = blockIdx.x * 128 + (wave & 1) * 64;
  d4 acc[4][4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[f][g] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
  for (int kk = 0; kk < INV_NB / 4; ++kk) {
    const double *ar = Cct + (size_t)(4 * kk + fk) * ld + i0 + fr;
    const double *br = Wb + (size_t)(4 * kk + fk) * ld + j0 + fr;
    double af[4], bf[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { af[f] = ar[16 * f]; bf[f] = br[16 * f]; }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[f][g] = MFMA_F64(af[f], bf[g], acc[f][g]);
  }
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 16 * f + fk + 4 * r, j = j0 + 16 * g + fr;
        if (i < n && j < n) M[(size_t)i * n + j] -= acc[f][g][r];
      }
}

// ---- 6. fix-up of the K rows / K columns ------------------------------------------------------------------------
// One workgroup per 64 indices j.  M[j,K] = -C[j,:] Dinv (j not in K) is formed by thread (jl, kg) for 16 of the 64 k and
// written row-wise through LDS;  M[K,j] = Wb[:,j] (j not in K), M[K,K] = Dinv.
__global__ __launch_bounds__(256) void inv_fixup_kernel(int n, int ld, int k0, int nbw, const double *__restrict__ Dinv,
                                                        const double *__restrict__ Wb, const double *__restrict__ Cct,
                                                        double *__restrict__ M, const int *__restrict__ info) {
  if (*info != 0) return;
  extern __shared__ __attribute__((aligned(16))) char inv_smem[];
  double (*D)[INV_NB + 1] = reinterpret_cast<double (*)[INV_NB + 1]>(inv_smem);
  double (*Cc)[INV_NB + 1] = D + INV_NB;
  double (*Out)[INV_NB + 1] = Cc + INV_NB;
  const int t = threadIdx.x, jl = t & 63, kg = t >> 6;
  const int j0 = blockIdx.x * INV_NB;
  for (int e = t; e < INV_NB * INV_NB; e += 256) {
    const int l = e >> 6, c = e & 63;
    D[l][c] = Dinv[e];
    Cc[l][c] = (j0 + c < ld) ? Cct[(size_t)l * ld + j0 + c] : 0.0;
  }
  __syncthreads();
#pragma unroll 4
  for (int kk = 0; kk < 16; ++kk) {
    const int k = kg * 16 + kk;
    double s = 0.0;
#pragma unroll 16
    for (int l = 0; l < INV_NB; ++l) s += Cc[l][jl] * D[l][k];
    Out[jl][k] = -s;
  }
  __syncthreads();
  for (int e = t; e < INV_NB * INV_NB; e += 256) {
    const int r = e >> 6, k = e & 63, j = j0 + r;        // row j of M, columns K
    if (j < n && k < nbw && !(j >= k0 && j < k0 + nbw)) M[(size_t)j * n + k0 + k] = Out[r][k];
  }
  for (int e = t; e < INV_NB * INV_NB; e += 256) {
    const int k = e >> 6, c = e & 63, j = j0 + c;        // rows K of M, column j
    if (j < n && k < nbw) {
      const bool inK = j >= k0 && j < k0 + nbw;
      M[(size_t)(k0 + k) * n + j] = inK ? D[k][j - k0] : Wb[(size_t)k * ld + j];
    }
  }
}

// ---- final: undo the row swaps as column swaps in reverse order (thread per row) ---------------------------------
__global__ void inv_col_swaps_kernel(int n, const int *__restrict__ piv_all, double *__restrict__ M, const int *__restrict__ info) {
  if (*info != 0) return;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double *row = M + (size_t)r * n;
  for (int k = n - 1; k >= 0; --k) {
    const int p = piv_all[k];
    if (p != k && p >= 0 && p < n) { const double a = row[k]; row[k] = row[p]; row[p] = a; }
  }
}

// workspace: P (n*64) + Wb (64*ld) + Cct (64*ld) + Dinv (64*64) doubles; piv (n) + info (1) ints
size_t inverse_workspace_doubles(int n) {
  const size_t ld = ((size_t)n + 127) / 128 * 128;
  // + the cooperative panel's exchange slots: cval[2][G], crow[2][G][64], rowj[2][64] (doubles), cidx[2][G] + counters (as doubles' worth)
  const size_t coop = 2 * PC_MAXG + 2 * (size_t)PC_MAXG * INV_NB + 2 * INV_NB + PC_MAXG + ((size_t)n / INV_NB + 2);
  return (size_t)n * INV_NB + 2 * INV_NB * ld + INV_NB * INV_NB + coop;
}

// The multi-workgroup panel is an ordinary launch of G <= (number of CUs) workgroups of 256 threads with up to 150 KB of
// LDS each: one per CU, all resident at once on a device whose earlier work on this stream has drained -- what the agent-scope
// barrier needs.  (hipLaunchCooperativeKernel, which the first version used, makes the HIP runtime create a second,
// "cooperative" HSA queue; that queue's teardown inside exit() faults under rocprofv3 -- tools/exit_probe.sh, DESIGN.md.)
// Should a workgroup ever not become resident, the bounded spin ends every wave and sets info = -7; the caller then restores
// the matrix and repeats with multi_wg = false.
bool launch_inverse(hipStream_t s, int n, double *M, double *work, int *piv_all /*[n]*/, int *info /*[1]*/, int num_cus,
                    bool multi_wg, int max_wg, unsigned spin_limit) {
  const int ld = (n + 127) / 128 * 128;
  double *P = work;
  double *Wb = P + (size_t)n * INV_NB;
  double *Cct = Wb + (size_t)INV_NB * ld;
  double *Dinv = Cct + (size_t)INV_NB * ld;
  double *cval = Dinv + (size_t)INV_NB * INV_NB;
  double *crow = cval + 2 * PC_MAXG;
  double *rowj = crow + 2 * (size_t)PC_MAXG * INV_NB;
  int *cidx = reinterpret_cast<int *>(rowj + 2 * INV_NB);            // 2 * PC_MAXG ints
  unsigned *counters = reinterpret_cast<unsigned *>(cidx + 2 * PC_MAXG);   // one per panel
  const int npanels = (n + INV_NB - 1) / INV_NB;
  (void)hipMemsetAsync(info, 0, sizeof(int), s);
  (void)hipMemsetAsync(piv_all, 0, (size_t)n * sizeof(int), s);
  (void)hipMemsetAsync(counters, 0, (size_t)npanels * sizeof(unsigned), s);
  const int ncu = num_cus > 0 ? num_cus : 1;
  const size_t lds_prep = 2 * (size_t)INV_NB * (INV_NB + 1) * sizeof(double), lds_fix = 3 * (size_t)INV_NB * (INV_NB + 1) * sizeof(double);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(inv_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prep);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(inv_fixup_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fix);
  // (the caller reads the comparison switches -- one place: conp_fix.cpp invert_device)
  const bool multi = multi_wg;
  const int maxg_env = max_wg > 0 ? max_wg : PC_MAXG;
  bool used_multi = false;
  for (int k0 = 0; k0 < n; k0 += INV_NB) {
    const int nbw = (n - k0 < INV_NB) ? n - k0 : INV_NB;
    const int m = n - k0;
    bool panel_done = false;
    if (multi) {
      int maxg = ncu < PC_MAXG ? ncu : PC_MAXG;
      if (maxg_env >= 1 && maxg_env < maxg) maxg = maxg_env;
      int rpw = (m + maxg - 1) / maxg;
      rpw = rpw < 64 ? 64 : (rpw + 15) / 16 * 16;
      const int G = (m + rpw - 1) / rpw;
      const size_t lds = ((size_t)rpw * PC_LD + INV_NB + 4) * sizeof(double) + 8 * sizeof(int);
      bool fits = lds <= 150 * 1024 && G <= ncu;
      if (fits) {
        // the grid barrier needs every workgroup resident: ask the runtime how many of THIS footprint (registers, LDS) a CU
        // takes instead of assuming one (it is one with > 80 KB of LDS; small panels pack several per CU, which is fine too)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(inv_panel_coop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(inv_panel_coop_kernel), 256, lds) != hipSuccess ||
            (long long)per_cu * ncu < G)
          fits = false;
      }
      if (fits) {
        hipLaunchKernelGGL(inv_panel_coop_kernel, dim3(G), dim3(256), lds, s, n, k0, nbw, rpw, (const double *)M, P, piv_all + k0, info,
                           cval, cidx, crow, rowj, counters + k0 / INV_NB, spin_limit);
        panel_done = true;
        used_multi = true;
      }
    }
    if (!panel_done) {
      const size_t tot = (size_t)m * INV_NB;
      hipLaunchKernelGGL(inv_panel_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, n, k0, nbw, M, P, info);
      hipLaunchKernelGGL(inv_panel_lu_kernel, dim3(1), dim3(1024), 0, s, m, k0, nbw, P, piv_all + k0, info);
    }
    hipLaunchKernelGGL(inv_row_swaps_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, k0, nbw, piv_all + k0, M, info);
    hipLaunchKernelGGL(inv_block_kernel, dim3(1), dim3(256), 0, s, nbw, P, Dinv, info);
    hipLaunchKernelGGL(inv_prep_kernel, dim3(ld / INV_NB), dim3(256), lds_prep, s, n, ld, k0, nbw, M, Dinv, Wb, Cct, info);
    hipLaunchKernelGGL(inv_update_kernel, dim3(ld / 128, ld / 128), dim3(256), 0, s, n, ld, Cct, Wb, M, info);
    hipLaunchKernelGGL(inv_fixup_kernel, dim3((n + INV_NB - 1) / INV_NB), dim3(256), lds_fix, s, n, ld, k0, nbw, Dinv, Wb, Cct, M, info);
  }
  hipLaunchKernelGGL(inv_col_swaps_kernel, dim3((n + 63) / 64), dim3(64), 0, s, n, piv_all, M, info);
  return used_multi;
}

// exact symmetry test: *flag_dev (device int) = non-zero / 0
void launch_symmetry_check(hipStream_t s, int n, const double *M, int *flag_dev) {
  // (a memset, not an asynchronous copy out of a stack variable that is gone when this returns)
  (void)hipMemsetAsync(flag_dev, 1, sizeof(int), s);
  const size_t tot = (size_t)n * n;
  hipLaunchKernelGGL(inv_symmetry_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, n, M, flag_dev);
}

// The SPD elimination (header comment): per 64-column block step the diagonal block's inverse, then steps 4-6 of the pivoted
// path unchanged.  info = -8 when a pivot was not positive; the matrix content is then meaningless (the caller keeps a copy).
void launch_inverse_spd(hipStream_t s, int n, double *M, double *work, int *info /*[1]*/) {
  const int ld = (n + 127) / 128 * 128;
  double *P = work;
  double *Wb = P + (size_t)n * INV_NB;
  double *Cct = Wb + (size_t)INV_NB * ld;
  double *Dinv = Cct + (size_t)INV_NB * ld;
  (void)hipMemsetAsync(info, 0, sizeof(int), s);
  const size_t lds_prep = 2 * (size_t)INV_NB * (INV_NB + 1) * sizeof(double), lds_fix = 3 * (size_t)INV_NB * (INV_NB + 1) * sizeof(double);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(inv_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prep);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(inv_fixup_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fix);
  for (int k0 = 0; k0 < n; k0 += INV_NB) {
    const int nbw = (n - k0 < INV_NB) ? n - k0 : INV_NB;
    hipLaunchKernelGGL(inv_block_spd_kernel, dim3(1), dim3(256), 0, s, n, k0, nbw, (const double *)M, Dinv, info);
    hipLaunchKernelGGL(inv_prep_kernel, dim3(ld / INV_NB), dim3(256), lds_prep, s, n, ld, k0, nbw, M, Dinv, Wb, Cct, info);
    hipLaunchKernelGGL(inv_update_kernel, dim3(ld / 128, ld / 128), dim3(256), 0, s, n, ld, Cct, Wb, M, info);
    hipLaunchKernelGGL(inv_fixup_kernel, dim3((n + INV_NB - 1) / INV_NB), dim3(256), lds_fix, s, n, ld, k0, nbw, Dinv, Wb, Cct, M, info);
  }
}

}  // namespace conp
