// In-place dense inverse of the electrode matrix on gfx950: blocked Gauss-Jordan elimination with partial (row)
// pivoting, FP64, trailing updates on the matrix cores (v_mfma_f64_16x16x4_f64).
// It stands where the reference calls LAPACK dgetrf_/dgetri_ (fix_conp.cpp:947-949); like those it handles a general
// (not necessarily positive definite) matrix.  Row-major storage, leading dimension n.
//
// One block step (NB = 64 columns K = [k0, k0+NB)):
//   1. copy the panel rows k0.. of columns K, LU-factor the copy with partial pivoting (one workgroup) -> pivots
//   2. apply the row swaps to the whole matrix
//   3. Dinv = (M[K,K])^-1 from the panel's L11, U11 (one workgroup, LDS)
//   4. Wb = Dinv * M[K,:] with the K columns zeroed;  Cct = M[:,K]^T with the K rows zeroed      (both k-major)
//   5. M -= Cct^T * Wb        (MFMA, all rows/cols; the zeroed blocks keep rows K / cols K untouched)
//   6. M[i,K] = -C[i,:] * Dinv (i not in K);  M[K,j] = Wb[:,j] (j not in K);  M[K,K] = Dinv
// After the last step the row swaps are undone as column swaps in reverse order: inv(A) = inv(P A) P.
#include <hip/hip_runtime.h>

#include <cmath>

#include "conp_kernels.h"

namespace conp {

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
constexpr int INV_NB = 64;

// ---- 1a. panel copy: P[(i - k0) * NB + c] = M[i][k0 + c]  for i >= k0 --------------------------------------------
__global__ void inv_panel_copy_kernel(int n, int k0, int nbw, const double *__restrict__ M, double *__restrict__ P) {
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
      s_p = p;
      piv[j] = k0 + p;
      if (!(b > 0.0)) *info = k0 + j + 1;
    }
    __syncthreads();
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
    // bound by what ONE CU can stream (~25-50 GB/s): the panel step moves m x 64 doubles per column.  Measured 3.7 ms per
    // 4096-row panel = 0.24 s of the 0.44 s setup at Ne = 4096; a multi-workgroup panel (tournament pivoting) is next-round work)
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

// ---- 2. row swaps on the whole matrix (each thread owns one column; swaps applied in order) ---------------------
__global__ void inv_row_swaps_kernel(int n, int k0, int nbw, const int *__restrict__ piv, double *__restrict__ M) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  for (int j = 0; j < nbw; ++j) {
    const int p = piv[j];
    if (p != k0 + j) {
      const double a = M[(size_t)(k0 + j) * n + c], b = M[(size_t)p * n + c];
      M[(size_t)(k0 + j) * n + c] = b;
      M[(size_t)p * n + c] = a;
    }
  }
}

// ---- 3. Dinv = U11^-1 L11^-1 from the factored panel top block (one workgroup, 64 x 64 in LDS) ------------------
__global__ __launch_bounds__(256) void inv_block_kernel(int nbw, const double *__restrict__ P, double *__restrict__ Dinv) {
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

// ---- 4. Wb[k][j] = sum_l Dinv[k][l] M[k0+l][j] (0 for j in K);  Cct[k][i] = M[i][k0+k] (0 for i in K) --------------
__global__ __launch_bounds__(256) void inv_prep_kernel(int n, int ld, int k0, int nbw, const double *__restrict__ M,
                                                       const double *__restrict__ Dinv, double *__restrict__ Wb,
                                                       double *__restrict__ Cct) {
  __shared__ double D[INV_NB][INV_NB + 1];
  for (int e = threadIdx.x; e < INV_NB * INV_NB; e += 256) D[e / INV_NB][e % INV_NB] = Dinv[e];
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;   // column for Wb, row for Cct
  if (j >= ld) return;
  const bool inK = j >= k0 && j < k0 + nbw;
  double col[INV_NB];
  if (j < n && !inK) {
#pragma unroll 8
    for (int l = 0; l < INV_NB; ++l) col[l] = (l < nbw) ? M[(size_t)(k0 + l) * n + j] : 0.0;
    for (int k = 0; k < INV_NB; ++k) {
      double s = 0.0;
#pragma unroll 8
      for (int l = 0; l < INV_NB; ++l) s += D[k][l] * col[l];
      Wb[(size_t)k * ld + j] = s;
    }
    for (int k = 0; k < INV_NB; ++k) Cct[(size_t)k * ld + j] = (k < nbw) ? M[(size_t)j * n + k0 + k] : 0.0;
  } else {
    for (int k = 0; k < INV_NB; ++k) { Wb[(size_t)k * ld + j] = 0.0; Cct[(size_t)k * ld + j] = 0.0; }
  }
}

// ---- 5. M -= Cct^T * Wb on the matrix cores: workgroup 128 x 128, wave 64 x 64 (4 x 4 fragments), K = 64 ----------
__global__ __launch_bounds__(256, 1) void inv_update_kernel(int n, int ld, const double *__restrict__ Cct,
                                                            const double *__restrict__ Wb, double *__restrict__ M) {
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
__global__ __launch_bounds__(256) void inv_fixup_kernel(int n, int ld, int k0, int nbw, const double *__restrict__ Dinv,
                                                        const double *__restrict__ Wb, const double *__restrict__ Cct,
                                                        double *__restrict__ M) {
  __shared__ double D[INV_NB][INV_NB + 1];
  for (int e = threadIdx.x; e < INV_NB * INV_NB; e += 256) D[e / INV_NB][e % INV_NB] = Dinv[e];
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const bool inK = j >= k0 && j < k0 + nbw;
  if (!inK) {
    // row j, columns K:  -C[j,:] * Dinv ;  rows K, column j: Wb[:, j]
    double c[INV_NB];
#pragma unroll 8
    for (int l = 0; l < INV_NB; ++l) c[l] = Cct[(size_t)l * ld + j];
    for (int k = 0; k < nbw; ++k) {
      double s = 0.0;
#pragma unroll 8
      for (int l = 0; l < INV_NB; ++l) s += c[l] * D[l][k];
      M[(size_t)j * n + k0 + k] = -s;
    }
    for (int k = 0; k < nbw; ++k) M[(size_t)(k0 + k) * n + j] = Wb[(size_t)k * ld + j];
  } else {
    const int c = j - k0;
    for (int k = 0; k < nbw; ++k) M[(size_t)(k0 + k) * n + j] = D[k][c];
  }
}

// ---- final: undo the row swaps as column swaps in reverse order (thread per row) ---------------------------------
__global__ void inv_col_swaps_kernel(int n, const int *__restrict__ piv_all, double *__restrict__ M) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double *row = M + (size_t)r * n;
  for (int k = n - 1; k >= 0; --k) {
    const int p = piv_all[k];
    if (p != k) { const double a = row[k]; row[k] = row[p]; row[p] = a; }
  }
}

// workspace: P (n*64) + Wb (64*ld) + Cct (64*ld) + Dinv (64*64) doubles; piv (n) + info (1) ints
size_t inverse_workspace_doubles(int n) {
  const size_t ld = ((size_t)n + 127) / 128 * 128;
  return (size_t)n * INV_NB + 2 * INV_NB * ld + INV_NB * INV_NB;
}

void launch_inverse(hipStream_t s, int n, double *M, double *work, int *piv_all /*[n]*/, int *info /*[1]*/) {
  const int ld = (n + 127) / 128 * 128;
  double *P = work;
  double *Wb = P + (size_t)n * INV_NB;
  double *Cct = Wb + (size_t)INV_NB * ld;
  double *Dinv = Cct + (size_t)INV_NB * ld;
  (void)hipMemsetAsync(info, 0, sizeof(int), s);
  for (int k0 = 0; k0 < n; k0 += INV_NB) {
    const int nbw = (n - k0 < INV_NB) ? n - k0 : INV_NB;
    const size_t tot = (size_t)(n - k0) * INV_NB;
    hipLaunchKernelGGL(inv_panel_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, n, k0, nbw, M, P);
    hipLaunchKernelGGL(inv_panel_lu_kernel, dim3(1), dim3(1024), 0, s, n - k0, k0, nbw, P, piv_all + k0, info);
    hipLaunchKernelGGL(inv_row_swaps_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, k0, nbw, piv_all + k0, M);
    hipLaunchKernelGGL(inv_block_kernel, dim3(1), dim3(256), 0, s, nbw, P, Dinv);
    hipLaunchKernelGGL(inv_prep_kernel, dim3(ld / 256 + 1), dim3(256), 0, s, n, ld, k0, nbw, M, Dinv, Wb, Cct);
    hipLaunchKernelGGL(inv_update_kernel, dim3(ld / 128, ld / 128), dim3(256), 0, s, n, ld, Cct, Wb, M);
    hipLaunchKernelGGL(inv_fixup_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, ld, k0, nbw, Dinv, Wb, Cct, M);
  }
  hipLaunchKernelGGL(inv_col_swaps_kernel, dim3((n + 63) / 64), dim3(64), 0, s, n, piv_all, M);
}

}  // namespace conp
