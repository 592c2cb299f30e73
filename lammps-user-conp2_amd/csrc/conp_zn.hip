// libconp_hip.so -- the "z-window" form of the structure-factor contraction (round 5; planar electrodes, large systems).
//
// What km_ewald.cpp:728-825 needs per update from the electrolyte is the class table
//     Hc[r][c] = sum_m w(r,m) sum_j A_rj [cos(m th_j) Tc[m][c] + sin(m th_j) Ts[m][c]]          (th_j = u_z z_j, A = the planar a / b rows)
// i.e.  Hc[r][c] = sum_j A_rj K_rc(th_j)  with K_rc a trigonometric polynomial of degree nz - 1 in th.  sk_gemm evaluates it by forming
// all 2 nz columns cos / sin(m th_j) of G (4 Nl K flop).  A band-limited periodic function is reproduced by interpolation from an
// oversampled grid with a compact window (the type-2 non-uniform FFT; window = the "exponential of semicircle" kernel
// phi(t) = exp(beta (sqrt(1 - t^2) - 1)), W = 15 taps, grid n >= 4 nz points: 5e-14 of the largest entry, tools/proto/zn_proto.py):
//     K_rc(th) = sum_g phi((g h - th) / a) P[r][c][g],      P[r][c][g] = h Re sum_m (w(r,m) (Tc - i Ts)[m][c] / phihat(m)) e^{i m g h}
// with h = 2 pi / n, a = W h / 2 and phihat the window's Fourier transform.  P is made once per run (zn_ptable_kernel).  Per update the
// atoms -- listed in the order of their z cells, so that 16 consecutive ones share a window of a few grid points -- contribute
//     acc[r][col] = sum_j A_rj phi((g0 + col) h - th_j)          a GEMM with NCOL = 32 or 48 columns instead of 2 nz = 252
// per (row tile, range of atoms), and the range's piece of the class table is  sum_col acc[r][col] P[r][c][g0 + col]  -- the same
// band-local piece sk_gemm's projecting epilogue leaves (hc_sum_kernel / b_zc_final_kernel consume it unchanged).
// MFMA work: 2 n_p x NCOL x Nl x 2 flop = 1/5 .. 1/8 of sk_gemm's.
#include <hip/hip_runtime.h>

#include "conp_kernels.h"

namespace conp {

typedef double d4 __attribute__((ext_vector_type(4)));
#define ZN_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// ---- once per run: P[(row * nzc + c) * n + g] -----------------------------------------------------------------------------------------
// one thread per (G row, grid point); cs[k] = (cos, sin)(2 pi k / n) so that the phases of m g h are exact table look-ups
__global__ __launch_bounds__(256) void zn_ptable_kernel(int R_pad, int C_pad, int nz, int kzt, int nzc, int n, const double *__restrict__ wfull,
                                                        const double *__restrict__ tzt /*[nzc][C_pad]*/, const double *__restrict__ phihat,
                                                        const double2 *__restrict__ cs, double *__restrict__ P) {
  const int g = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
  if (g >= n) return;
  const int row_a = (row >> 7) * 128 + (row & 63);             // the 'b' rows of a planar vector carry the 'a' rows' weights
  const double h = 6.283185307179586476925286766559 / n;
  for (int c = 0; c < nzc; ++c) {
    double s = 0.0;
    for (int m = 0; m < nz; ++m) {
      const int ct = m / kzt, ml = m - ct * kzt;
      const int cc = 320 * ct + 16 * (ml >> 3) + (ml & 7);     // KPlan::col_c; col_s = + 8
      const double w = wfull[(size_t)row_a * C_pad + cc];
      if (w == 0.0) continue;
      const double2 e = cs[(int)(((long long)m * g) % n)];
      s += (w / phihat[m]) * (tzt[(size_t)c * C_pad + cc] * e.x + tzt[(size_t)c * C_pad + cc + 8] * e.y);
    }
    P[((size_t)row * nzc + c) * n + g] = h * s;
  }
}

// ---- per update: the window matrix, dense per chunk: Bt[(chunk * NCOL + col) * 16 + atom] = phi((g0[chunk] + col) - u_j) -----------
// u_j = z_j n / Lz' (grid units).  A tap that would fall outside the chunk's columns raises the flag (the list order or the margins
// are stale: the host re-sorts).  One thread per (chunk, col, atom).
__global__ __launch_bounds__(256) void zn_window_kernel(int nl, int nl_pad, int ncol, int n, int W, double beta, double gscale /*n / Lz'*/,
                                                        const int *__restrict__ elyte_idx, const double *__restrict__ x,
                                                        const int *__restrict__ g0c /*[chunks]*/, double *__restrict__ Bt, int *__restrict__ flag,
                                                        int j0, int j1) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int per = ncol * 16;
  const int chunk = (int)(t / per), rem = (int)(t - (long long)chunk * per);
  const int col = rem >> 4, a = rem & 15;
  const int j = chunk * 16 + a;
  if (j >= nl_pad || j < j0 || j >= j1) return;
  double v = 0.0;
  if (j < nl) {
    const int g0 = g0c[chunk];
    // the atom's grid coordinate relative to the chunk's window origin, wrapped into (-n/2, n/2] (the grid is periodic)
    double ur = x[3 * (size_t)elyte_idx[j] + 2] * gscale - (double)g0;
    ur -= (double)n * rint(ur / (double)n);
    const double d = ((double)col - ur) * (2.0 / W);            // in units of the window's half width
    if (d > -1.0 && d < 1.0) v = exp(beta * (sqrt(1.0 - d * d) - 1.0));
    if (col == 0) {
      // this atom's taps: ceil(ur - W / 2) .. + W - 1 must lie inside [0, ncol)
      const int i0 = (int)ceil(ur - 0.5 * W);
      if (i0 < 0 || i0 + W > ncol) *reinterpret_cast<volatile int *>(flag) = 1;      // (page-locked host memory: the host sees it at its next look)
    }
  }
  Bt[t] = v;
}

// ---- per update: the contraction + the projection on P -----------------------------------------------------------------------------
// item = (row tile rt: 64 planar vectors = 128 G rows, chunk range [c0, c1), window origin g0, output slot)
// 256 threads = 4 waves; wave w owns the row fragments 2 w, 2 w + 1 (0-3: 'a' rows, 4-7: 'b' rows) x all NCF column fragments.
// LDS panel per chunk (double-buffered): [128 + 16 NCF features][16 atoms], column XOR-swizzled by feature & 15 like sk_gemm's.
constexpr int ZN_LD = 16;
template <int NCF>
__global__ __launch_bounds__(256, 3) void zn_gemm_kernel(DevPlan pl, const ZnItem *__restrict__ items, const double2 *__restrict__ Xt,
                                                         const double2 *__restrict__ Yt, const double *__restrict__ Bt,
                                                         const double *__restrict__ P, int n, int nzc, double *__restrict__ pieces,
                                                         int piece_stride) {
  constexpr int NF = 128 + 16 * NCF;                         // features per panel
  __shared__ __attribute__((aligned(16))) double panel[2][NF * ZN_LD];
  const ZnItem it = items[blockIdx.x];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int gj = t & 15, gs = t >> 4;                        // build role: atom gj of the chunk, sub-index gs 0..15
  const int fr = lane & 15, fk = lane >> 4;
  const unsigned nrx16 = (unsigned)(pl.kxmax + 2) * 16, nry16 = (unsigned)(pl.kymax + 1) * 16;
  // this thread's four planar vectors gs + 16 u of the row tile
  unsigned xo[4], yo[4];
  bool neg[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int p = it.rt * 64 + gs + 16 * u;
    xo[u] = (unsigned)pl.p_ikx[p] * 16 + gj; yo[u] = (unsigned)pl.p_iky[p] * 16 + gj;
    neg[u] = pl.p_sgn[p] < 0;                                 // (padding vectors read the all-zero X row)
  }
  const unsigned wa = (unsigned)(gs * ZN_LD + (gj ^ gs));    // (feature gs + 16 u, atom gj): + 16 u * ZN_LD; swizzle key = feature & 15 = gs
  double2 X[4], Y[4];
  double bv[NCF];
  auto load = [&](int ch) {
    const unsigned bx = (unsigned)ch * nrx16, by = (unsigned)ch * nry16;
#pragma unroll
    for (int u = 0; u < 4; ++u) { X[u] = Xt[bx + xo[u]]; Y[u] = Yt[by + yo[u]]; }
#pragma unroll
    for (int u = 0; u < NCF; ++u) bv[u] = Bt[((size_t)ch * (16 * NCF) + gs + 16 * u) * 16 + gj];
  };
  auto build = [&](double *pn) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double sy = neg[u] ? -Y[u].y : Y[u].y;
      pn[wa + 16 * u * ZN_LD] = X[u].x * Y[u].x - X[u].y * sy;                 // 'a' rows 0..63
      pn[wa + (64 + 16 * u) * ZN_LD] = X[u].x * sy + X[u].y * Y[u].x;          // 'b' rows 64..127
    }
#pragma unroll
    for (int u = 0; u < NCF; ++u) pn[wa + (128 + 16 * u) * ZN_LD] = bv[u];
  };
  d4 acc[2][NCF];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int c = 0; c < NCF; ++c) acc[f][c] = (d4){0.0, 0.0, 0.0, 0.0};
  // MFMA fragment addresses (doubles): element (feature 16 F + fr, atom 4 ks + fk) at feature * 16 + ((4 ks + fk) ^ fr)
  const unsigned fa0 = (unsigned)((16 * (2 * wave) + fr) * ZN_LD), fa1 = fa0 + 16 * ZN_LD, fb0 = (unsigned)((128 + fr) * ZN_LD);
  load(it.c0);
  build(panel[0]);
  if (it.c0 + 1 < it.c1) load(it.c0 + 1);
  __syncthreads();
  int buf = 0;
  for (int ch = it.c0; ch < it.c1; ++ch, buf ^= 1) {
    const double *pn = panel[buf];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned col = (unsigned)((4 * ks + fk) ^ fr);
      const double a0 = pn[fa0 + col], a1 = pn[fa1 + col];
      double b[NCF];
#pragma unroll
      for (int c = 0; c < NCF; ++c) b[c] = pn[fb0 + 16 * c * ZN_LD + col];
#pragma unroll
      for (int c = 0; c < NCF; ++c) { acc[0][c] = ZN_MFMA(a0, b[c], acc[0][c]); acc[1][c] = ZN_MFMA(a1, b[c], acc[1][c]); }
    }
    if (ch + 1 < it.c1) {
      build(panel[buf ^ 1]);                                  // (its inputs were requested a chunk ago)
      if (ch + 2 < it.c1) load(ch + 2);
    }
    __syncthreads();
  }
  // ---- the range's piece of the class table: piece[c * 128 + row] = sum_col acc[row][col] P[rowG][c][g0 + col]
  double *out = pieces + (size_t)it.slot * piece_stride;
  int gcol[NCF];                                              // this lane's grid columns (the grid is periodic; n is any integer)
#pragma unroll
  for (int cf = 0; cf < NCF; ++cf) { int g = (it.g0 + 16 * cf + fr) % n; gcol[cf] = g < 0 ? g + n : g; }
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int rf8 = 2 * wave + f;                             // row fragment of the tile: 0-3 'a', 4-7 'b'
    for (int c = 0; c < nzc; ++c) {
      double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int cf = 0; cf < NCF; ++cf) {
        const int g = gcol[cf];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rowl = 16 * rf8 + 4 * r + fk;
          s[r] += acc[f][cf][r] * P[((size_t)(it.rt * 128 + rowl) * nzc + c) * n + g];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = s[r];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        if (fr == 0) out[c * 128 + 16 * rf8 + 4 * r + fk] = v;
      }
    }
  }
}

void launch_zn_ptable(hipStream_t s, const DevPlan &pl, int kzt, int nzc, int n, const double *tzt, const double *phihat, const double2 *cs,
                      double *P) {
  hipLaunchKernelGGL(zn_ptable_kernel, dim3((n + 255) / 256, pl.R_pad), dim3(256), 0, s, pl.R_pad, pl.C_pad, pl.nz, kzt, nzc, n, pl.wfull, tzt,
                     phihat, cs, P);
}
void launch_zn_window(hipStream_t s, int nl, int nl_pad, int ncol, int n, int W, double beta, double gscale, const int *elyte_idx, const double *x,
                      const int *g0c, double *Bt, int *flag, int j0, int j1) {
  const long long total = (long long)(nl_pad / 16) * ncol * 16;
  hipLaunchKernelGGL(zn_window_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, nl, nl_pad, ncol, n, W, beta, gscale, elyte_idx, x,
                     g0c, Bt, flag, j0, j1);
}
void launch_zn_gemm(hipStream_t s, const DevPlan &pl, int ncf, const ZnItem *items, int nitems, const double2 *Xt, const double2 *Yt,
                    const double *Bt, const double *P, int n, int nzc, double *pieces, int piece_stride) {
  if (nitems <= 0) return;
  if (ncf == 2)
    hipLaunchKernelGGL(zn_gemm_kernel<2>, dim3(nitems), dim3(256), 0, s, pl, items, Xt, Yt, Bt, P, n, nzc, pieces, piece_stride);
  else
    hipLaunchKernelGGL(zn_gemm_kernel<3>, dim3(nitems), dim3(256), 0, s, pl, items, Xt, Yt, Bt, P, n, nzc, pieces, piece_stride);
}

}  // namespace conp
