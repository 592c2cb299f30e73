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
// Rough electrodes (no z classes): the same contraction leaves the raw windows; zn_wsum_kernel adds them on the periodic z grid and
// zn_dft_kernel applies the type-1 transform  sum_j A_j e^{i m th_j} = (h / phihat(m)) sum_g grid[g] e^{i m g h}  -- out come G and w o G
// as sk_reduce_kernel leaves them, for b_project_kernel.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "conp_kernels.h"

namespace conp {

typedef double d4 __attribute__((ext_vector_type(4)));
#define ZN_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// ---- once per run: P[((row tile * nzc + c) * n + g) * 64 + vector of the tile] -----------------------------------------------------------------------------------------
// one thread per (G row, grid point); cs[k] = (cos, sin)(2 pi k / n) so that the phases of m g h are exact table look-ups
__global__ __launch_bounds__(256) void zn_ptable_kernel(int R_pad, int C_pad, int nz, int kzt, int nzc, int n, const double *__restrict__ wfull,
                                                        const double *__restrict__ tzt /*[nzc][C_pad]*/, const double *__restrict__ phihat,
                                                        const double2 *__restrict__ cs, double *__restrict__ P) {
  const int g = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;   // row = planar vector (padded to 64 a tile)
  if (g >= n) return;
  const int row_a = (row >> 6) * 128 + (row & 63);             // the 'b' row of a planar vector carries the 'a' row's weights: one table row
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
    P[(((size_t)(row >> 6) * nzc + c) * n + g) * 64 + (row & 63)] = h * s;
  }
}

// (the window matrix of an update -- phi((g0[chunk] + col) - u_j), u_j = z_j n / Lz', stored in MFMA-fragment order
//  Bt[chunk][col >> 4][atom >> 2][col & 15][atom & 3] -- is written by elyte_phase_kernel's z-axis threads, conp_kernels.hip: one launch
//  for the phase tables and the window)

// ---- per update: the contraction + the projection on P -----------------------------------------------------------------------------
// item = (row tile rt: 64 planar vectors = 128 G rows, chunk range [c0, c1), window origin g0, output slot)
// 256 threads = 4 waves; wave w owns the row fragments w ('a' rows of 16 planar vectors) and 4 + w (their 'b' rows) x all NCF column fragments;
// the window fragment is the MFMA's A operand (accumulator rows = window columns), the feature fragment its B operand (see the epilogue).
// LDS: per wave two panels [32 rows][16 atoms] (double-buffered, private to the wave, XOR-swizzled); the window comes from memory.
constexpr int ZN_LD = 16;
#ifdef ZN_TIMELINE
// diagnostic build only (tools/zn_timeline.py): wall-clock stamps (100 MHz) of every workgroup's phases + where it ran
__device__ unsigned long long zn_tl[8192 * 6];
__device__ unsigned long long zn_tl_chunks[16 * 64];   // the workgroups that share block 0's CU: a stamp per chunk
#define ZN_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) zn_tl[blockIdx.x * 6 + (k)] = wall_clock64(); } while (0)
#else
#define ZN_STAMP(k) do { } while (0)
#endif
// (Measured on the way, with a panel per workgroup and a barrier per chunk: two or four items per workgroup under ONE barrier, so
//  that the items of a CU advance together instead of finishing one after the other -- the barrier's bubble then idles the whole CU:
//  57.3 us with two items, 53.9 with four, against 51.0 with separate workgroups, headline size; profiles/r05_zn_timeline.txt.)
template <int NCF, bool RAW>
__global__ __launch_bounds__(256, NCF == 2 ? 4 : 3) void zn_gemm_kernel(DevPlan pl, const ZnItem *__restrict__ items, int nitems,
                                                                const double2 *__restrict__ Xt, const double2 *__restrict__ Yt,
                                                                const double *__restrict__ Bt, const double *__restrict__ P, int n, int nzc,
                                                                double *__restrict__ pieces, int piece_stride) {
  constexpr int WP = 32 * ZN_LD;                             // a wave's panel: its 16 'a' rows + 16 'b' rows x 16 atoms
  extern __shared__ __attribute__((aligned(16))) double zn_lds[];
  ZN_STAMP(0);
#ifdef ZN_TIMELINE
  const long long zn_c0 = clock64();
#endif
  const int item_idx = (int)blockIdx.x;                       // (the grid is padded to a multiple of the XCD count)
  if (item_idx >= nitems) return;
  const ZnItem it = items[item_idx];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // NO workgroup barrier anywhere: a wave forms the 32 panel rows IT multiplies (two wave-private buffers in LDS; LDS operations of one
  // wave execute in order) and takes the window fragments straight from memory, in fragment order.  With a barrier per chunk a workgroup
  // that multiplies alone -- the SIMD's arbiter serves the oldest wave first, the workgroups of a CU finish one after the other -- spent
  // 0.78 us on a chunk of 0.49 us of MFMAs; a wave on its own schedule keeps the pipe busy (profiles/r05_zn_timeline.txt).
  double *const panel0 = zn_lds + (size_t)wave * 2 * WP, *const panel1 = panel0 + WP;
  const int gj = lane & 15, g4 = lane >> 4;                  // build role: atom gj of the chunk; planar vectors / pairs g4 + 4 u of the wave's
  const int fr = lane & 15, fk = lane >> 4;
  const unsigned nrx16 = (unsigned)(pl.kxmax + 2) * 16, nry16 = (unsigned)(pl.kymax + 1) * 16;
  // panel element (local row r, atom a) at rho(r) * 16 + (a ^ key(rho(r))), rho(r) = (r >> 1) + 8 (r & 1), key(q) = (q >> 1) & 7; the 'b'
  // row of a vector 16 rows further.  (rho: the two members of a pair sit 8 rows apart, so that the rows one instruction writes
  // alternate between the halves of the banks)
  auto prow = [](int r, int a) { const int q = (r >> 1) + 8 * (r & 1); return (unsigned)(q * ZN_LD + (a ^ ((q >> 1) & 7))); };
  // the MFMA's k index (k-step ks, lane group fk) is atom 4 fk + ks: a lane's four window values of a column are adjacent in memory
  unsigned rdA[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) rdA[ks] = prow(fr, 4 * fk + ks);
  // buffer loads: descriptor (item-relative base, SGPRs) + per-thread byte offset (VGPR) + chunk offset (SGPR) -- no address arithmetic
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(Xt + (size_t)it.c0 * nrx16), (short)0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2 *>(Yt + (size_t)it.c0 * nry16), (short)0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Bt + (size_t)it.c0 * (16 * NCF * 16)), (short)0, -1, 0x00020000);
  // window matrix in fragment order: Bt[chunk][column block cf][fk][column fr][4 atoms 4 fk + ks]
  const unsigned bby = (unsigned)((fk * 16 + fr) * 32);
  d4 acc[2][NCF];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int c = 0; c < NCF; ++c) acc[f][c] = (d4){0.0, 0.0, 0.0, 0.0};
  // Per chunk and wave: 8 NCF MFMAs (56 of their 64 cycles each on the SIMD's vector port, which every other VALU instruction of every
  // wave shares: profiles/r05_pipe_share.txt) and the next chunk's 32 panel values.  No VALU instruction goes into addresses (buffer
  // loads; LDS addresses are loop-invariant registers + immediates, the loop being unrolled over the two panel buffers).  k-steps 0..2
  // carry the twelve build slices between their MFMAs; the window fragments of a k-step pair are re-requested behind its last MFMA;
  // the next chunk's first fragments are read behind k-step 2, under the MFMAs of k-step 3.
  // PAIRED (row tiles of whole (+ky, -ky) pairs, KPlan::paired_lo / _hi): a lane forms BOTH members of its two pairs from one fetch of
  // the x and the y row -- half the table traffic (the texture path is the limit without it: 12 KB per wave and chunk) and 12 instead
  // of 16 FP64 operations; the other tiles (singles) take four vectors per lane.
  auto run = [&](auto pairedc) {
    constexpr bool PAIRED = decltype(pairedc)::value;
    constexpr int NV = PAIRED ? 2 : 4;                        // table fetches (x row + y row each) per lane and chunk
    unsigned xby[NV], yby[NV], sgm[NV], wa[NV], wm[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int r = PAIRED ? 2 * (g4 + 4 * u) : g4 + 4 * u, p = it.rt * 64 + 16 * wave + r;
      xby[u] = ((unsigned)pl.p_ikx[p] * 16 + gj) * 16u; yby[u] = ((unsigned)pl.p_iky[p] * 16 + gj) * 16u;
      sgm[u] = pl.p_sgn[p] < 0 ? 0x80000000u : 0u;            // (padding vectors read the all-zero X row)
      wa[u] = prow(r, gj); wm[u] = prow(r + 1, gj);           // (wm: the pair's second member)
    }
    double2 X[NV], Y[NV];
    double wB[4][NCF];                                        // window fragments of the current chunk, per k-step
    auto load_xy = [&](int u, int ch) {
      const int cr = min(ch, it.c1 - 1) - it.c0;              // (past the range's end: the last chunk again, not used)
      X[u] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rx, xby[u], cr * (int)(nrx16 * 16), 0));
      Y[u] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(ry, yby[u], cr * (int)(nry16 * 16), 0));
    };
    auto load_w = [&](int half, int ch) {                     // k-steps 2 half, 2 half + 1 of chunk ch
      const int cr = min(ch, it.c1 - 1) - it.c0;
#pragma unroll
      for (int c = 0; c < NCF; ++c) {
        const double2 v = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rb, bby + 2048 * c + 16 * half, cr * (16 * NCF * 16 * 8), 0));
        wB[2 * half][c] = v.x; wB[2 * half + 1][c] = v.y;
      }
    };
    // prologue: the first chunk's panel, then the loads in the order the loop re-issues them (its partial waits count on it)
#pragma unroll
    for (int u = 0; u < NV; ++u) load_xy(u, it.c0);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      if constexpr (PAIRED) {
        const double t2 = X[u].y * Y[u].y, t4 = X[u].y * Y[u].x;
        panel0[wa[u]] = X[u].x * Y[u].x - t2; panel0[wm[u]] = X[u].x * Y[u].x + t2;
        panel0[wa[u] + 16 * ZN_LD] = X[u].x * Y[u].y + t4; panel0[wm[u] + 16 * ZN_LD] = t4 - X[u].x * Y[u].y;
      } else {
        const double sy = __hiloint2double(__double2hiint(Y[u].y) ^ (int)sgm[u], __double2loint(Y[u].y));
        panel0[wa[u]] = X[u].x * Y[u].x - X[u].y * sy;
        panel0[wa[u] + 16 * ZN_LD] = X[u].x * sy + X[u].y * Y[u].x;
      }
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      load_xy(u, it.c0 + 1);
      if (u == NV / 2 - 1) load_w(0, it.c0);
    }
    load_w(1, it.c0);
    ZN_STAMP(1);
    double fa[2][2];                                          // two fragment sets; on entry to a chunk set 0 holds its k-step 0
    auto frag = [&](const double *pn, int ks, int set) { fa[set][0] = pn[rdA[ks]]; fa[set][1] = pn[rdA[ks] + 16 * ZN_LD]; };
    frag(panel0, 0, 0);
    auto chunk = [&](int ch, auto bufc) {
      constexpr int BUF = decltype(bufc)::value;
      const double *pn = BUF ? panel1 : panel0;
      double *pw = BUF ? panel0 : panel1;
      double sy = 0.0, va = 0.0, vb = 0.0, vc = 0.0, vd = 0.0, t2 = 0.0, t4 = 0.0;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nx = cur ^ 1;
        frag(ks < 3 ? pn : pw, ks < 3 ? ks + 1 : 0, nx);      // (k-step 3: the next chunk's k-step 0 -- its panel is complete)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2 * NCF; ++m) {
          const int f = m & 1, c = m >> 1;
          acc[f][c] = ZN_MFMA(wB[ks][c], fa[cur][f], acc[f][c]);
          const int g = 4 * ks + m;                           // gap behind this MFMA: slice g of the build (k-steps 0..2, four gaps each)
          if (ks < 3 && m < 4) {
            if constexpr (PAIRED) {                           // six slices per pair: a+, a-, b+, b-, the four panel writes, the fetches
              const int u = g / 6, part = g - 6 * u;
              if (part == 0) { t2 = X[u].y * Y[u].y; va = X[u].x * Y[u].x - t2; }
              else if (part == 1) vb = X[u].x * Y[u].x + t2;
              else if (part == 2) { t4 = X[u].y * Y[u].x; vc = X[u].x * Y[u].y + t4; }
              else if (part == 3) vd = t4 - X[u].x * Y[u].y;
              else if (part == 4) { pw[wa[u]] = va; pw[wm[u]] = vb; pw[wa[u] + 16 * ZN_LD] = vc; pw[wm[u] + 16 * ZN_LD] = vd; }
              else load_xy(u, ch + 2);
            } else {                                          // three slices per vector
              const int u = g / 3, part = g - 3 * u;
              if (part == 0) {
                sy = __hiloint2double(__double2hiint(Y[u].y) ^ (int)sgm[u], __double2loint(Y[u].y));
                va = X[u].x * Y[u].x - X[u].y * sy;
              } else if (part == 1) vb = X[u].x * sy + X[u].y * Y[u].x;
              else { pw[wa[u]] = va; pw[wa[u] + 16 * ZN_LD] = vb; load_xy(u, ch + 2); }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (ks & 1) { load_w(ks >> 1, ch + 1); __builtin_amdgcn_sched_barrier(0); }
      }
#ifdef ZN_TIMELINE
      if (t == 0 && (blockIdx.x & 255) == 0 && (blockIdx.x >> 8) < 16 && ch - it.c0 < 64) zn_tl_chunks[(blockIdx.x >> 8) * 64 + ch - it.c0] = wall_clock64();
#endif
    };
    const int nmax = it.c1 - it.c0;
#ifndef ZN_SKIP_MAIN
    for (int i = 0; i < nmax; i += 2) {
      chunk(it.c0 + i, std::integral_constant<int, 0>());
      if (i + 1 < nmax) chunk(it.c0 + i + 1, std::integral_constant<int, 1>());
    }
#endif
  };
  if (it.paired) run(std::true_type()); else run(std::false_type());
  ZN_STAMP(2);
  if constexpr (RAW) {
    // rough electrodes (no z classes): the range's window itself, raw[(slot * NCOL + col) * 128 + row] -- zn_wsum / zn_dft turn the
    // windows into the structure-factor matrix G (a lane holds columns 4 r + fk of every block for the vector 16 wave + fr)
    double *raw = pieces + (size_t)it.slot * (16 * NCF * 128) + 16 * wave + fr;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
        for (int r = 0; r < 4; ++r) raw[(16 * cf + 4 * r + fk) * 128 + 64 * f] = acc[f][cf][r];
    return;
  }
  // ---- the range's piece of the class table: piece[c * 128 + row] = sum_col acc[row][col] P[vector of row][c][g0 + col]
  // The window is the MFMA's A operand and the features its B operand, so a lane holds, for ONE planar vector (16 wave + fr) and both its
  // rows, the four columns 4 r + fk of every column block: the sum over columns is in-lane but for the four lane groups fk (two exchanges
  // per value; with the features as A operand it was a 16-lane reduction of 8 values -- ~90 vector instructions per class, and the vector
  // port is what the MFMAs of the CU's other workgroups run on).  The 'a' and the 'b' row share their weights: one fetch serves both.
  // P[((rt * nzc + c) * n + g) * 64 + v]: the 16 lanes fr read 128 contiguous bytes.  The table comes from the Infinity Cache at best
  // (13 MB a run, never twice through one L2), so the fetches of CB classes are in flight together.
  double *out = pieces + (size_t)it.slot * piece_stride + 16 * wave + fr;
  unsigned gofs[NCF][4];                                      // this lane's grid columns x 64 (the grid is periodic; n is any integer)
#pragma unroll
  for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
    for (int r = 0; r < 4; ++r) { int g = (it.g0 + 16 * cf + 4 * r + fk) % n; gofs[cf][r] = (unsigned)(g < 0 ? g + n : g) * 64u; }
  const double *Pw = P + (size_t)it.rt * nzc * n * 64 + 16 * wave + fr;
  constexpr int CB = 3;
#ifdef ZN_SKIP_EPI
  for (int cb = 0; cb < 1; cb += CB) {
#else
  for (int cb = 0; cb < nzc; cb += CB) {
#endif
    double pv[CB][NCF][4];
#pragma unroll
    for (int k = 0; k < CB; ++k) {
      const double *pc = Pw + (size_t)min(cb + k, nzc - 1) * n * 64;
#pragma unroll
      for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[k][cf][r] = pc[gofs[cf][r]];
    }
#pragma unroll
    for (int k = 0; k < CB; ++k) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0 += acc[0][cf][r] * pv[k][cf][r]; s1 += acc[1][cf][r] * pv[k][cf][r]; }
      s0 += __shfl_xor(s0, 16, 64); s1 += __shfl_xor(s1, 16, 64);
      s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64);
      if (fk == 0 && cb + k < nzc) { out[(cb + k) * 128] = s0; out[(cb + k) * 128 + 64] = s1; }
    }
  }
#ifdef ZN_TIMELINE
  __syncthreads();
  ZN_STAMP(3);
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    zn_tl[blockIdx.x * 6 + 4] = ((unsigned long long)xcc << 32) | hw;
    zn_tl[blockIdx.x * 6 + 5] = (unsigned long long)(it.c1 - it.c0) | ((unsigned long long)(clock64() - zn_c0) << 16);
  }
#endif
}

void launch_zn_ptable(hipStream_t s, const DevPlan &pl, int kzt, int nzc, int n, const double *tzt, const double *phihat, const double2 *cs,
                      double *P) {
  hipLaunchKernelGGL(zn_ptable_kernel, dim3((n + 255) / 256, pl.R_pad / 2), dim3(256), 0, s, pl.R_pad, pl.C_pad, pl.nz, kzt, nzc, n, pl.wfull, tzt,
                     phihat, cs, P);
}
template <int NCF, bool RAW>
static void zn_gemm_launch(hipStream_t s, const DevPlan &pl, const ZnItem *items, int nitems, const double2 *Xt, const double2 *Yt,
                           const double *Bt, const double *P, int n, int nzc, double *pieces, int piece_stride) {
  constexpr int lds = 4 * 2 * 32 * ZN_LD * 8;                // four waves x two panels of 32 rows
  const int nblocks = (nitems + 7) / 8 * 8;                  // (a multiple of the XCD count: item i stays on XCD i mod 8)
  hipLaunchKernelGGL((zn_gemm_kernel<NCF, RAW>), dim3(nblocks), dim3(256), lds, s, pl, items, nitems, Xt, Yt, Bt, P, n, nzc, pieces,
                     piece_stride);
}
void launch_zn_gemm(hipStream_t s, const DevPlan &pl, int ncf, const ZnItem *items, int nitems, const double2 *Xt, const double2 *Yt,
                    const double *Bt, const double *P, int n, int nzc, double *pieces, int piece_stride) {
  if (nitems <= 0) return;
  if (P == nullptr) {                                        // no class table: the raw windows (rough electrodes)
    if (ncf == 2) zn_gemm_launch<2, true>(s, pl, items, nitems, Xt, Yt, Bt, P, n, nzc, pieces, piece_stride);
    else zn_gemm_launch<3, true>(s, pl, items, nitems, Xt, Yt, Bt, P, n, nzc, pieces, piece_stride);
  } else if (ncf == 2) zn_gemm_launch<2, false>(s, pl, items, nitems, Xt, Yt, Bt, P, n, nzc, pieces, piece_stride);
  else zn_gemm_launch<3, false>(s, pl, items, nitems, Xt, Yt, Bt, P, n, nzc, pieces, piece_stride);
}

// ---- rough electrodes: from the ranges' windows to G ------------------------------------------------------------------------------------
// The windows of a row tile add up to its rows on the periodic z grid, grid[(k * n + g) * 128 + row] (k = the rank's k-th row tile):
// grid point g is covered by the few ranges whose window holds it, listed per g in the order of the ranges (cov_ptr / cov_ent =
// (range, column)): a fixed order of additions.
__global__ __launch_bounds__(128) void zn_wsum_kernel(int n, int nrg, int ncol, const int *__restrict__ cov_ptr, const int2 *__restrict__ cov_ent,
                                                      const double *__restrict__ raw, double *__restrict__ grid) {
  const int g = blockIdx.x, k = blockIdx.y, row = threadIdx.x;
  double s = 0.0;
  const int e0 = cov_ptr[g], e1 = cov_ptr[g + 1];           // (~8 ranges hold a grid point: four loads in flight at a time)
  for (int e = e0; e < e1; e += 4) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int2 en = cov_ent[min(e + u, e1 - 1)];
      v[u] = raw[((size_t)(k * nrg + en.x) * ncol + en.y) * 128 + row];
      if (e + u >= e1) v[u] = 0.0;
    }
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  grid[((size_t)k * n + g) * 128 + row] = s;
}
// once per run: the transform from the grid to the kz columns of G, Dt[g][C_pad]: column (ct, 16 b + i): kz = kzt ct + 8 b + (i & 7), the
// cosine for i < 8 and the sine for i >= 8 (KPlan::col_c / col_s), times h / phihat(kz) -- the type-1 non-uniform transform's
// deconvolution:  sum_j A_j e^{i m th_j} = (h / phihat(m)) sum_g grid[g] e^{i m g h}
__global__ __launch_bounds__(256) void zn_dtable_kernel(int C_pad, int nz, int kzt, int n, const double *__restrict__ phihat,
                                                        const double2 *__restrict__ cs, double *__restrict__ Dt) {
  const int col = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  if (col >= C_pad) return;
  const int ct = col / 320, cl = col - 320 * ct;
  const int ml = 8 * (cl >> 4) + (cl & 7), m = ct * kzt + ml;
  double v = 0.0;
  if (ml < kzt && m < nz) {
    const double2 e = cs[(int)(((long long)m * g) % n)];
    v = (6.283185307179586476925286766559 / n / phihat[m]) * ((cl & 8) ? e.y : e.x);
  }
  Dt[(size_t)g * C_pad + col] = v;
}
// per update: G = grid x Dt on the matrix cores (M = rows, K = n grid points, N = C_pad columns: 0.3 GF), written as sk_reduce_kernel
// leaves it: G row-major and w o G in MFMA-fragment-major order for b_project_kernel.
__global__ __launch_bounds__(256) void zn_dft_kernel(int n, int C_pad, int nz, int kzt, const int *__restrict__ own_rt, const double *__restrict__ grid,
                                                     const double *__restrict__ Dt, const double *__restrict__ wfull, double *__restrict__ G,
                                                     double *__restrict__ Gwf) {
  // a workgroup = one 32 x 32 tile of G (two row fragments x two column blocks: one operand fetch per MFMA instead of two -- every
  // operand comes straight from L2), its four waves = four quarters of the grid axis (the launch is short of waves and each of them
  // waits on L2 otherwise); the quarters meet in LDS
  __shared__ double part[3][64][16];
  const int lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4, kq = threadIdx.x >> 6;
  const int k = blockIdx.x >> 2, fp = blockIdx.x & 3, rt = own_rt[k], cb0 = 2 * blockIdx.y;
  {                                                          // column blocks past the last kz of their tile hold zeros (and stay zero)
    const int ct = 16 * cb0 / 320, ml0 = 8 * ((16 * cb0 - 320 * ct) >> 4);
    if (ml0 >= kzt || ct * kzt + ml0 >= nz) return;
  }
  const int nkq = n / 16, kbeg = kq * nkq, nks = kbeg + nkq;                 // this wave's k-steps [kbeg, nks)  (n is a multiple of 16)
  const double *a = grid + ((size_t)k * n + fk) * 128 + 32 * fp + fr;           // + 4 ks * 128; second fragment + 16
  const double *b = Dt + (size_t)fk * C_pad + 16 * cb0 + fr;                    // + 4 ks * C_pad; second block + 16
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  constexpr int KB = 8;
  double a0[KB][2], b0[KB][2], a1[KB][2], b1[KB][2];
  auto fetch = [&](double(*av)[2], double(*bv)[2], int k0) {
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const int ks = min(k0 + i, nks - 1);                   // (past the end: a valid address, the value is dropped below)
      av[i][0] = a[(size_t)ks * 512]; av[i][1] = a[(size_t)ks * 512 + 16];
      bv[i][0] = b[(size_t)ks * 4 * C_pad]; bv[i][1] = b[(size_t)ks * 4 * C_pad + 16];
    }
  };
  auto mult = [&](const double(*av)[2], const double(*bv)[2], int k0) {
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const bool on = k0 + i < nks;
      const double x0 = on ? av[i][0] : 0.0, x1 = on ? av[i][1] : 0.0;
      acc[0][0] = ZN_MFMA(x0, bv[i][0], acc[0][0]); acc[0][1] = ZN_MFMA(x0, bv[i][1], acc[0][1]);
      acc[1][0] = ZN_MFMA(x1, bv[i][0], acc[1][0]); acc[1][1] = ZN_MFMA(x1, bv[i][1], acc[1][1]);
    }
  };
  fetch(a0, b0, kbeg);
  for (int k0 = kbeg; k0 < nks; k0 += 2 * KB) {
    fetch(a1, b1, k0 + KB);
    mult(a0, b0, k0);
    fetch(a0, b0, k0 + 2 * KB);
    mult(a1, b1, k0 + KB);
  }
  if (kq > 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[kq - 1][lane][(2 * i + j) * 4 + r] = acc[i][j][r];
  }
  __syncthreads();
  if (kq > 0) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int f16 = 2 * fp + i, gcol = 16 * (cb0 + j) + fr;
      const size_t rf = (size_t)rt * 8 + f16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = ((acc[i][j][r] + part[0][lane][(2 * i + j) * 4 + r]) + part[1][lane][(2 * i + j) * 4 + r]) + part[2][lane][(2 * i + j) * 4 + r];
        const int row16 = 4 * r + fk;
        const size_t grow = (size_t)rt * 128 + 16 * f16 + row16;
        G[grow * C_pad + gcol] = v;
        Gwf[(rf * (C_pad / 4) + (gcol >> 2)) * 64 + (gcol & 3) * 16 + row16] = wfull[grow * C_pad + gcol] * v;
      }
    }
}
void launch_zn_dtable(hipStream_t s, const DevPlan &pl, int kzt, int n, const double *phihat, const double2 *cs, double *Dt) {
  hipLaunchKernelGGL(zn_dtable_kernel, dim3((pl.C_pad + 255) / 256, n), dim3(256), 0, s, pl.C_pad, pl.nz, kzt, n, phihat, cs, Dt);
}
void launch_zn_windows_to_g(hipStream_t s, const DevPlan &pl, int kzt, int n, int n_own, const int *own_rt, int nrg, int ncol, const int *cov_ptr,
                            const int2 *cov_ent, const double *raw, double *grid, const double *Dt, double *G, double *Gwf) {
  hipLaunchKernelGGL(zn_wsum_kernel, dim3(n, n_own), dim3(128), 0, s, n, nrg, ncol, cov_ptr, cov_ent, raw, grid);
  hipLaunchKernelGGL(zn_dft_kernel, dim3(4 * n_own, pl.C_pad / 32), dim3(256), 0, s, n, pl.C_pad, pl.nz, kzt, own_rt, grid, Dt, pl.wfull, G, Gwf);
}

}  // namespace conp

#ifdef ZN_TIMELINE
extern "C" int conp_debug_zn_timeline(unsigned long long *out, int nblocks) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(conp::zn_tl), sizeof(unsigned long long) * 6 * (size_t)nblocks);
}
extern "C" int conp_debug_zn_timeline_chunks(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(conp::zn_tl_chunks), sizeof(unsigned long long) * 16 * 64);
}
#endif
