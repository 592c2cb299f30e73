// PPPM k-space b vector on gfx950 (pppm_conp.cpp:109-316): charge spreading, Poisson solve on the mesh, stencil gather at
// the electrode atoms.  Double precision, one rank owns the whole mesh, periodic wrap instead of ghost planes
// (equivalent to GridComm's reverse/forward exchange on one rank, pppm_conp.cpp:114,122).
// The three 1-D transforms are mixed-radix (2/3/4/5) Stockham FFTs per mesh line in LDS -- LAMMPS only picks 2,3,5-smooth mesh
// sizes; any other length falls back to a plain O(n^2) DFT per line.
#include <hip/hip_runtime.h>

#include <atomic>

#include "conp_kernels.h"
#include "conp_brow.hpp"

namespace conp {

__device__ __forceinline__ int pwrap(int i, int n) { i %= n; return i < 0 ? i + n : i; }

__device__ __forceinline__ void rho1d_dev(const double *__restrict__ coeff, int order, double dx, double *w) {
  for (int k = 0; k < order; ++k) {
    double r = 0.0;
    for (int l = order - 1; l >= 0; --l) r = coeff[l * order + k] + r * dx;
    w[k] = r;
  }
}

// elyte_particle_map + elyte_make_rho (pppm_conp.cpp:126-228): order^3 threads per charged electrolyte atom, one stencil point
// (one atomic add) each, the x index running over adjacent lanes: a wavefront's atomic instruction then covers ~13 runs of
// `order` consecutive doubles instead of 64 single doubles in 64 different cache lines (the first version: order^2 threads per
// atom, `order` atomics each along x -- 13.4 us for 1280 atoms; scattered f64 atomics run ~17x below the contiguous rate,
// MI355X_MICROARCH.md).  slab_part[block] = partial sums of q z for the slab term (:301-314), taken by the first thread of every
// atom.  Orders above 6 (order^3 > 256) walk their stencil in several rounds of 256 threads.
__host__ __device__ inline int spread_threads_per_atom(int order) { const int o3 = order * order * order; return o3 < 256 ? o3 : 256; }

// Blocks beyond the spreading ones (nb_spread .. ): the real-space pair sums of the electrode rows (fix_conp.cpp:1313-1353), four
// rows per block -- they depend on x, q only and ride along here (round 4) instead of costing a launch of their own behind the mesh.
__global__ __launch_bounds__(256) void pppm_spread_kernel(PppmDev pd, int nl, const int *__restrict__ elyte_idx,
                                                          const double *__restrict__ x, const double *__restrict__ q,
                                                          double *__restrict__ rho, double *__restrict__ slab_part,
                                                          int npass, int nb_spread, BRowArgs ra, double *__restrict__ breal_out) {
  __shared__ double coeff[64];
  __shared__ double red[4];
  if ((int)blockIdx.x >= nb_spread) {
    const int row = ((int)blockIdx.x - nb_spread) * 4 + (int)(threadIdx.x >> 6);
    if (row < ra.ne) {
      const double v = b_row_pairs(ra, row, threadIdx.x & 63);
      if ((threadIdx.x & 63) == 0) breal_out[row] = v;
    }
    return;
  }
  if (threadIdx.x < pd.order * pd.order) coeff[threadIdx.x] = pd.rho_coeff[threadIdx.x];
  __syncthreads();
  const int o2 = pd.order * pd.order, o3 = o2 * pd.order;
  const int tpa = spread_threads_per_atom(pd.order);
  const int apb = blockDim.x / tpa;                 // atoms per pass
  const int ja = threadIdx.x / tpa, t0 = threadIdx.x - ja * tpa;
  double qz = 0.0;
  for (int pass = 0; pass < npass; ++pass) {
    const int j = (blockIdx.x * npass + pass) * apb + ja;
    if (ja >= apb || j >= nl) continue;
    const int i = elyte_idx[j];
    const double qq = q[i];
    int g[3];
    double w[3][8];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double xs = (x[3 * i + c] - pd.boxlo[c]) * pd.delinv[c];
      g[c] = static_cast<int>(xs + pd.shift) - 16384;
      rho1d_dev(coeff, pd.order, g[c] + pd.shiftone - xs, w[c]);
    }
    if (t0 == 0) qz += qq * x[3 * i + 2];
    for (int rem = t0; rem < o3; rem += tpa) {
      const int n = rem / o2, r2 = rem - n * o2, m = r2 / pd.order, l = r2 - m * pd.order;
      const int mz = pwrap(n + pd.nlower + g[2], pd.nz);
      const int my = pwrap(m + pd.nlower + g[1], pd.ny);
      const int mx = pwrap(l + pd.nlower + g[0], pd.nx);
      const double x0 = (pd.delvolinv * qq * w[2][n]) * w[1][m];      // the association of the row-wise version (and of :216-225)
      atomicAdd(&rho[((size_t)mz * pd.ny + my) * pd.nx + mx], x0 * w[0][l]);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) qz += __shfl_down(qz, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = qz;
  __syncthreads();
  if (threadIdx.x == 0) slab_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// 1-D DFT along `axis` of a [nz][ny][nx] complex mesh (re, im planes), in place.  One workgroup transforms XT adjacent
// lines at a time (adjacent along the fastest other index, so global accesses are coalesced for axis != 0).
// sign = -1: forward (LAMMPS FFT3d flag 1), +1: backward; no scaling.
__global__ __launch_bounds__(256) void pppm_dft_kernel(int nx, int ny, int nz, int axis, double sign,
                                                       const double *__restrict__ twid, double *__restrict__ re,
                                                       double *__restrict__ im, int XT) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = axis == 0 ? nx : (axis == 1 ? ny : nz);
  double *lr = reinterpret_cast<double *>(smem);   // [n][XT]
  double *li = lr + (size_t)n * XT;
  double *tw = li + (size_t)n * XT;                 // [n][2]
  for (int t = threadIdx.x; t < 2 * n; t += blockDim.x) tw[t] = twid[t];
  // lines: axis 0 -> (y,z) pairs, elements contiguous; axis 1 -> (x,z), stride nx; axis 2 -> (x,y), stride nx*ny
  const size_t stride = axis == 0 ? 1 : (axis == 1 ? (size_t)nx : (size_t)nx * ny);
  const int nlines = axis == 0 ? ny * nz : (axis == 1 ? nx * nz : nx * ny);
  const int line0 = blockIdx.x * XT;
  auto base_of = [&](int line) -> size_t {
    if (axis == 0) return (size_t)line * nx;
    if (axis == 1) return (size_t)(line / nx) * nx * ny + (line % nx);
    return (size_t)line;
  };
  for (int e = threadIdx.x; e < n * XT; e += blockDim.x) {
    const int t = axis == 0 ? e % n : e / XT, lx = axis == 0 ? e / n : e % XT;
    const int line = line0 + lx;
    double vr = 0.0, vi = 0.0;
    if (line < nlines) { const size_t a = base_of(line) + (size_t)t * stride; vr = re[a]; vi = im[a]; }
    lr[t * XT + lx] = vr; li[t * XT + lx] = vi;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n * XT; e += blockDim.x) {
    const int f = e / XT, lx = e % XT;
    double sr = 0.0, si = 0.0;
    int w = 0;
    for (int t = 0; t < n; ++t) {
      const double c = tw[2 * w], s = sign * tw[2 * w + 1];
      const double xr = lr[t * XT + lx], xi = li[t * XT + lx];
      sr += xr * c - xi * s;
      si += xr * s + xi * c;
      w += f; if (w >= n) w -= n;
    }
    const int line = line0 + lx;
    if (line < nlines) { const size_t a = base_of(line) + (size_t)f * stride; re[a] = sr; im[a] = si; }
  }
}

// (Measured, round 4: the lines dealt to the wavefronts -- a line per wavefront in the z pass, a run of lines per wavefront in the
//  plane passes -- so that a line's stages need no workgroup barrier between them, one barrier per axis instead of one per stage:
//  SLOWER on il_onelayer 40 x 45 x 180, 52.8 -> 55.3 us per update (z pass +2 us, plane passes +0.6): a stage spread over all 512
//  threads is one butterfly deep, a wavefront alone walks its line's butterflies with 45 of 64 lanes and nothing to overlap.)
// Mixed-radix Stockham FFT along `axis` (autosort, radices 2/3/4/5 -- LAMMPS meshes are 2,3,5-smooth), XT adjacent lines per
// workgroup, ping-pong in LDS.  Stage with radix R and sub-transform length Ns: butterfly j (0 <= j < n/R), k = j mod Ns:
//   v[r] = in[j + r n/R] * W_n^{r k n/(Ns R)} ;  out[(j - k) R + k + q Ns] = sum_r v[r] W_R^{q r}
// W_n^t = (cos, sign*sin)(2 pi t / n) from the same twiddle table the plain DFT uses; no scaling.
struct FftPlan { int nrad; int rad[12]; };

// one radix-R stage over XT = 1 << xs lines held in LDS
template <int R>
__device__ __forceinline__ void fft_stage(const double2 *__restrict__ in, double2 *__restrict__ out,
                                          const double *__restrict__ tw, int n, int Ns, int xs, double sign) {
  const int nb = n / R;                  // butterflies per line; also W_R = W_n^{nb}
  const int tstep = nb / Ns;             // twiddle index step: W_n^{r k tstep}, r k tstep < n
  const float inv_ns = 1.0f / (float)Ns;
  double wc[R], ws[R];
#pragma unroll
  for (int m = 1; m < R; ++m) { wc[m] = tw[2 * m * nb]; ws[m] = sign * tw[2 * m * nb + 1]; }
  const int XT = 1 << xs;
  for (int w = threadIdx.x; w < (nb << xs); w += blockDim.x) {
    const int j = w >> xs, lx = w & (XT - 1);
    const int k = j - Ns * (int)(((float)j + 0.5f) * inv_ns);      // j mod Ns (j < 2^20: exact in float)
    double2 v[R];
    v[0] = in[(j << xs) + lx];
    int ti = 0;
#pragma unroll
    for (int r = 1; r < R; ++r) {
      const double2 x = in[((j + r * nb) << xs) + lx];
      ti += k * tstep;
      const double c = tw[2 * ti], sn = sign * tw[2 * ti + 1];
      v[r] = make_double2(x.x * c - x.y * sn, x.x * sn + x.y * c);
    }
    const int j0 = (j - k) * R + k;
    if (R == 2) {
      out[(j0 << xs) + lx] = make_double2(v[0].x + v[1].x, v[0].y + v[1].y);
      out[((j0 + Ns) << xs) + lx] = make_double2(v[0].x - v[1].x, v[0].y - v[1].y);
    } else if (R == 4) {
      // W_4 = (0, sign): multiplying by W_4 maps (a, b) -> (-sign b, sign a)
      const double2 s02 = make_double2(v[0].x + v[2].x, v[0].y + v[2].y), d02 = make_double2(v[0].x - v[2].x, v[0].y - v[2].y);
      const double2 s13 = make_double2(v[1].x + v[3].x, v[1].y + v[3].y), d13 = make_double2(v[1].x - v[3].x, v[1].y - v[3].y);
      const double2 id13 = make_double2(-sign * d13.y, sign * d13.x);
      out[(j0 << xs) + lx] = make_double2(s02.x + s13.x, s02.y + s13.y);
      out[((j0 + Ns) << xs) + lx] = make_double2(d02.x + id13.x, d02.y + id13.y);
      out[((j0 + 2 * Ns) << xs) + lx] = make_double2(s02.x - s13.x, s02.y - s13.y);
      out[((j0 + 3 * Ns) << xs) + lx] = make_double2(d02.x - id13.x, d02.y - id13.y);
    } else {
#pragma unroll
      for (int q = 0; q < R; ++q) {
        double sr = v[0].x, si = v[0].y;
#pragma unroll
        for (int r = 1; r < R; ++r) {
          const int m = (q * r) % R;
          if (m == 0) { sr += v[r].x; si += v[r].y; }
          else { sr += v[r].x * wc[m] - v[r].y * ws[m]; si += v[r].x * ws[m] + v[r].y * wc[m]; }
        }
        out[((j0 + q * Ns) << xs) + lx] = make_double2(sr, si);
      }
    }
  }
}

// flags: 1 = input is real (im not read), 2 = only the real part of the output is stored, 4 = ... and zeros go to `im` (which then
// serves as the clean density brick of the next update), 8 = forward pass, influence function, backward pass in one go.
// gmul != nullptr: the output is
// multiplied by gscale*gmul[] (the influence function, pppm_conp.cpp:242-249) on the way out.
__global__ __launch_bounds__(512) void pppm_fft_kernel(int nx, int ny, int nz, int axis, double sign, FftPlan fp,
                                                       const double *__restrict__ twid, double *__restrict__ re,
                                                       double *__restrict__ im, int xs, int flags,
                                                       const double *__restrict__ gmul, double gscale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = axis == 0 ? nx : (axis == 1 ? ny : nz);
  const int XT = 1 << xs;
  double2 *buf0 = reinterpret_cast<double2 *>(smem);     // [n][XT]
  double2 *buf1 = buf0 + (size_t)n * XT;
  double *tw = reinterpret_cast<double *>(buf1 + (size_t)n * XT);   // [n][2]
  for (int t = threadIdx.x; t < 2 * n; t += blockDim.x) tw[t] = twid[t];
  const size_t stride = axis == 0 ? 1 : (axis == 1 ? (size_t)nx : (size_t)nx * ny);
  const int nlines = axis == 0 ? ny * nz : (axis == 1 ? nx * nz : nx * ny);
  const int line0 = blockIdx.x * XT;
  auto base_of = [&](int line) -> size_t {
    if (axis == 0) return (size_t)line * nx;
    if (axis == 1) return (size_t)(line / nx) * nx * ny + (line % nx);
    return (size_t)line;
  };
  const bool real_in = flags & 1, real_out = flags & 2;
  for (int e = threadIdx.x; e < n * XT; e += blockDim.x) {
    const int t = axis == 0 ? e % n : e >> xs, lx = axis == 0 ? e / n : e & (XT - 1);
    const int line = line0 + lx;
    double2 v = make_double2(0.0, 0.0);
    if (line < nlines) { const size_t a = base_of(line) + (size_t)t * stride; v = make_double2(re[a], real_in ? 0.0 : im[a]); }
    buf0[(t << xs) + lx] = v;
  }
  __syncthreads();
  double2 *in = buf0, *out = buf1;
  int Ns = 1;
  for (int st = 0; st < fp.nrad; ++st) {
    const int R = fp.rad[st];
    switch (R) {
      case 2: fft_stage<2>(in, out, tw, n, Ns, xs, sign); break;
      case 3: fft_stage<3>(in, out, tw, n, Ns, xs, sign); break;
      case 4: fft_stage<4>(in, out, tw, n, Ns, xs, sign); break;
      default: fft_stage<5>(in, out, tw, n, Ns, xs, sign); break;
    }
    __syncthreads();
    double2 *tmp = in; in = out; out = tmp;
    Ns *= R;
  }
  if (flags & 8) {
    // forward pass done: times the influence function, then the backward pass along the same lines before anything is stored
    // (the 3-D backward transform is separable: it may start with this axis) -- one launch and one trip through memory fewer
    double2 *io = const_cast<double2 *>(in);
    for (int e = threadIdx.x; e < n * XT; e += blockDim.x) {
      const int f = axis == 0 ? e % n : e >> xs, lx = axis == 0 ? e / n : e & (XT - 1);
      const int line = line0 + lx;
      if (line < nlines) {
        const double g = gscale * gmul[base_of(line) + (size_t)f * stride];
        double2 v = io[(f << xs) + lx];
        v.x *= g; v.y *= g;
        io[(f << xs) + lx] = v;
      }
    }
    __syncthreads();
    Ns = 1;
    for (int st = 0; st < fp.nrad; ++st) {
      const int R = fp.rad[st];
      switch (R) {
        case 2: fft_stage<2>(in, out, tw, n, Ns, xs, -sign); break;
        case 3: fft_stage<3>(in, out, tw, n, Ns, xs, -sign); break;
        case 4: fft_stage<4>(in, out, tw, n, Ns, xs, -sign); break;
        default: fft_stage<5>(in, out, tw, n, Ns, xs, -sign); break;
      }
      __syncthreads();
      double2 *tmp = in; in = out; out = tmp;
      Ns *= R;
    }
  }
  for (int e = threadIdx.x; e < n * XT; e += blockDim.x) {
    const int f = axis == 0 ? e % n : e >> xs, lx = axis == 0 ? e / n : e & (XT - 1);
    const int line = line0 + lx;
    if (line < nlines) {
      const size_t a = base_of(line) + (size_t)f * stride;
      double2 v = in[(f << xs) + lx];
      if (gmul && !(flags & 8)) { const double g = gscale * gmul[a]; v.x *= g; v.y *= g; }
      re[a] = v.x;
      if (!real_out) im[a] = v.y;
      else if (flags & 4) im[a] = 0.0;
    }
  }
}

// ---- x and y transforms of one z-plane in ONE workgroup (the plane, nx * ny complex values, fits in LDS for the decks' meshes):
// two launches fewer per 3-D transform -- the mesh path of a deck-sized system is bound by its launches.
// Generic stage: element (t, line) lives at buf[t * st_t + line * st_l]; `t_fast` picks which index runs over adjacent threads
// (the one with unit stride, so that a wavefront's LDS accesses spread over the banks).
// quotient of two non-negative ints below 2^20 through a float reciprocal (exact there: the +0.5 keeps the product away from the
// integer boundaries); an integer division costs ~40 instructions on this chip and a butterfly needs three
__device__ __forceinline__ int fdiv(int a, int b, float inv_b) { (void)b; return (int)(((float)a + 0.5f) * inv_b); }

template <int R>
__device__ __forceinline__ void fft_stage_g(const double2 *__restrict__ in, double2 *__restrict__ out, const double *__restrict__ tw,
                                            int n, int Ns, int nlines, int st_t, int st_l, bool t_fast, double sign) {
  const int nb = n / R;
  const int tstep = nb / Ns;
  const float inv_nb = 1.0f / (float)nb, inv_nl = 1.0f / (float)nlines, inv_ns = 1.0f / (float)Ns;
  double wc[R], ws[R];
#pragma unroll
  for (int m = 1; m < R; ++m) { wc[m] = tw[2 * m * nb]; ws[m] = sign * tw[2 * m * nb + 1]; }
  for (int w = threadIdx.x; w < nb * nlines; w += blockDim.x) {
    int j, lx;
    if (t_fast) { lx = fdiv(w, nb, inv_nb); j = w - lx * nb; }
    else { j = fdiv(w, nlines, inv_nl); lx = w - j * nlines; }
    const int k = j - Ns * fdiv(j, Ns, inv_ns);
    const double2 *src = in + lx * st_l;
    double2 *dst = out + lx * st_l;
    double2 v[R];
    v[0] = src[j * st_t];
    int ti = 0;
#pragma unroll
    for (int r = 1; r < R; ++r) {
      const double2 x = src[(j + r * nb) * st_t];
      ti += k * tstep;
      const double c = tw[2 * ti], sn = sign * tw[2 * ti + 1];
      v[r] = make_double2(x.x * c - x.y * sn, x.x * sn + x.y * c);
    }
    const int j0 = (j - k) * R + k;
    if (R == 2) {
      dst[j0 * st_t] = make_double2(v[0].x + v[1].x, v[0].y + v[1].y);
      dst[(j0 + Ns) * st_t] = make_double2(v[0].x - v[1].x, v[0].y - v[1].y);
    } else if (R == 4) {
      const double2 s02 = make_double2(v[0].x + v[2].x, v[0].y + v[2].y), d02 = make_double2(v[0].x - v[2].x, v[0].y - v[2].y);
      const double2 s13 = make_double2(v[1].x + v[3].x, v[1].y + v[3].y), d13 = make_double2(v[1].x - v[3].x, v[1].y - v[3].y);
      const double2 id13 = make_double2(-sign * d13.y, sign * d13.x);
      dst[j0 * st_t] = make_double2(s02.x + s13.x, s02.y + s13.y);
      dst[(j0 + Ns) * st_t] = make_double2(d02.x + id13.x, d02.y + id13.y);
      dst[(j0 + 2 * Ns) * st_t] = make_double2(s02.x - s13.x, s02.y - s13.y);
      dst[(j0 + 3 * Ns) * st_t] = make_double2(d02.x - id13.x, d02.y - id13.y);
    } else {
#pragma unroll
      for (int q = 0; q < R; ++q) {
        double sr = v[0].x, si = v[0].y;
#pragma unroll
        for (int r = 1; r < R; ++r) {
          const int m = (q * r) % R;
          if (m == 0) { sr += v[r].x; si += v[r].y; }
          else { sr += v[r].x * wc[m] - v[r].y * ws[m]; si += v[r].x * ws[m] + v[r].y * wc[m]; }
        }
        dst[(j0 + q * Ns) * st_t] = make_double2(sr, si);
      }
    }
  }
}

__device__ __forceinline__ void fft_axis_g(double2 *&in, double2 *&out, const double *tw, const FftPlan &fp, int n, int nlines,
                                           int st_t, int st_l, bool t_fast, double sign) {
  int Ns = 1;
  for (int st = 0; st < fp.nrad; ++st) {
    switch (fp.rad[st]) {
      case 2: fft_stage_g<2>(in, out, tw, n, Ns, nlines, st_t, st_l, t_fast, sign); break;
      case 3: fft_stage_g<3>(in, out, tw, n, Ns, nlines, st_t, st_l, t_fast, sign); break;
      case 4: fft_stage_g<4>(in, out, tw, n, Ns, nlines, st_t, st_l, t_fast, sign); break;
      default: fft_stage_g<5>(in, out, tw, n, Ns, nlines, st_t, st_l, t_fast, sign); break;
    }
    __syncthreads();
    double2 *tmp = in; in = out; out = tmp;
    Ns *= fp.rad[st];
  }
}

// one stencil weight (compute_rho1d, one k): the polynomial of rho1d_dev for that k alone
__device__ __forceinline__ double rho1d_one(const double *__restrict__ coeff, int order, double dx, int k) {
  double r = 0.0;
  for (int l = order - 1; l >= 0; --l) r = coeff[l * order + k] + r * dx;
  return r;
}

// real_in == 3 (round 4, deck-sized systems): the plane's workgroup spreads the charges ITSELF -- no density brick in memory, no
// spreading launch, no clearing.  Every workgroup walks the charged electrolyte atoms (1280 on the decks: 2.5 per thread), keeps
// those whose z stencil reaches its plane (~36) in a list, then all threads add the order^2 in-plane stencil points of the listed
// atoms into the LDS plane (LDS atomics; the products are elyte_make_rho's, pppm_conp.cpp:216-225 -- the order in which a mesh
// point's contributions arrive was unordered with the global atomics too).  slab_part[plane] = sum of q z over the plane's share
// of the atom list (:301-314).  Blocks beyond the planes: the real-space pair sums, eight rows per block.
struct PppmSpreadIn {
  int nl;
  const int *elyte_idx;
  const double *x, *q;
  double *slab_part;
  double *breal_out;
};

// one workgroup per z-plane: x transforms of its ny lines, then y transforms of its nx lines.  real_in: im is not read.
// real_in: 0 complex input; 1 real input in `re` (im not read); 2 real input in `im` (re not read) -- the b path spreads the charges
// into `im`, which the previous update's last backward pass left all zero, so that no clearing launch is needed; 3: see above
__global__ __launch_bounds__(512) void pppm_fft_xy_kernel(int nx, int ny, double sign, FftPlan fpx, FftPlan fpy,
                                                          const double *__restrict__ twx, const double *__restrict__ twy,
                                                          double *__restrict__ re, double *__restrict__ im, int real_in,
                                                          int out_flags /* 2: store the real part only, 4: ... and zeros into im */,
                                                          PppmDev pd, PppmSpreadIn sp, BRowArgs ra) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (real_in == 3 && (int)blockIdx.x >= pd.nz) {
    const int row = ((int)blockIdx.x - pd.nz) * 8 + (int)(threadIdx.x >> 6);
    if (row < ra.ne) {
      const double v = b_row_pairs(ra, row, threadIdx.x & 63);
      if ((threadIdx.x & 63) == 0) sp.breal_out[row] = v;
    }
    return;
  }
  const int np = nx * ny;
  double2 *b0 = reinterpret_cast<double2 *>(smem), *b1 = b0 + np;
  double *tx = reinterpret_cast<double *>(b1 + np), *ty = tx + 2 * nx;
  for (int t = threadIdx.x; t < 2 * nx; t += blockDim.x) tx[t] = twx[t];
  for (int t = threadIdx.x; t < 2 * ny; t += blockDim.x) ty[t] = twy[t];
  const size_t base = (size_t)blockIdx.x * np;
  if (real_in == 3) {
    __shared__ double coeff[64];
    __shared__ double red[8];
    __shared__ int nhit_s;
    const int p = blockIdx.x, o = pd.order, o2 = o * o;
    int *hits = reinterpret_cast<int *>(b1);          // (b1 is the first stage's output: free until then)
    const int cap = 4 * np;
    // this thread's atoms (up to 16: nl <= 8192) and their z are requested BEFORE the plane is cleared and the barrier behind it:
    // two dependent trips to memory that used to start behind that barrier
    int ai[16];
    double az[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = (int)threadIdx.x + 512 * u;
      ai[u] = j < sp.nl ? sp.elyte_idx[j] : -1;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) az[u] = ai[u] >= 0 ? sp.x[3 * ai[u] + 2] : 0.0;
    if ((int)threadIdx.x < o2) coeff[threadIdx.x] = pd.rho_coeff[threadIdx.x];
    if (threadIdx.x == 0) nhit_s = 0;
    for (int e = threadIdx.x; e < np; e += blockDim.x) b0[e] = make_double2(0.0, 0.0);
    __syncthreads();
    auto add_atom = [&](int i, int n, int m, int l) {      // stencil point (n, m, l) of atom i into the plane
      const double qq = sp.q[i];
      int g[3];
      double d[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double xs = (sp.x[3 * i + c] - pd.boxlo[c]) * pd.delinv[c];
        g[c] = static_cast<int>(xs + pd.shift) - 16384;
        d[c] = g[c] + pd.shiftone - xs;
      }
      const int my = pwrap(m + pd.nlower + g[1], pd.ny), mx = pwrap(l + pd.nlower + g[0], pd.nx);
      const double x0 = (pd.delvolinv * qq * rho1d_one(coeff, o, d[2], n)) * rho1d_one(coeff, o, d[1], m);
      atomicAdd(&b0[my * nx + mx].x, x0 * rho1d_one(coeff, o, d[0], l));
    };
    // this plane's share of the slab sum: a contiguous run of the atom list
    const int share = (sp.nl + pd.nz - 1) / pd.nz, lo = p * share, hi = lo + share < sp.nl ? lo + share : sp.nl;
    double qz = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = (int)threadIdx.x + 512 * u, i = ai[u];
      if (i < 0) continue;
      const double z = az[u];
      if (j >= lo && j < hi) qz += sp.q[i] * z;
      const double zs = (z - pd.boxlo[2]) * pd.delinv[2];
      const int gz = static_cast<int>(zs + pd.shift) - 16384;
      const int n = pwrap(p - pd.nlower - gz, pd.nz);      // the stencil index of this plane for the atom, if it has one
      if (n >= o) continue;
      const int k = atomicAdd(&nhit_s, 1);
      if (k < cap) hits[k] = j;
      else for (int r2 = 0; r2 < o2; ++r2) add_atom(i, n, r2 / o, r2 % o);      // (list full: never at the sizes this mode is used for)
    }
    __syncthreads();
    const int nh = nhit_s < cap ? nhit_s : cap;
    for (int item = threadIdx.x; item < nh * o2; item += blockDim.x) {
      const int a = item / o2, r2 = item - a * o2, m = r2 / o, l = r2 - m * o;
      const int i = sp.elyte_idx[hits[a]];
      const double zs = (sp.x[3 * i + 2] - pd.boxlo[2]) * pd.delinv[2];
      const int gz = static_cast<int>(zs + pd.shift) - 16384;
      add_atom(i, pwrap(p - pd.nlower - gz, pd.nz), m, l);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) qz += __shfl_down(qz, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = qz;
    __syncthreads();
    if (threadIdx.x == 0) sp.slab_part[p] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
  } else {
    for (int e = threadIdx.x; e < np; e += blockDim.x)
      b0[e] = real_in == 2 ? make_double2(im[base + e], 0.0) : make_double2(re[base + e], real_in ? 0.0 : im[base + e]);
    __syncthreads();
  }
  double2 *in = b0, *out = b1;
  fft_axis_g(in, out, tx, fpx, nx, ny, 1, nx, true, sign);      // x: element (t = x, line = y) at [y * nx + x]
  fft_axis_g(in, out, ty, fpy, ny, nx, nx, 1, false, sign);     // y: element (t = y, line = x)
  for (int e = threadIdx.x; e < np; e += blockDim.x) {
    re[base + e] = in[e].x;
    if (!(out_flags & 2)) im[base + e] = in[e].y;
    else if (out_flags & 4) im[base + e] = 0.0;
  }
}

static bool fft_factor(int n, FftPlan &fp) {
  fp.nrad = 0;
  while (n % 4 == 0 && fp.nrad < 12) { fp.rad[fp.nrad++] = 4; n /= 4; }
  for (int r : {2, 3, 5})
    while (n % r == 0 && fp.nrad < 12) { fp.rad[fp.nrad++] = r; n /= r; }
  return n == 1;
}

// rho(k) -> V(k): scale by 1/N and the influence function (pppm_conp.cpp:242-249)
__global__ void pppm_greens_kernel(int nfft, double scaleinv, const double *__restrict__ greensfn, double *__restrict__ re,
                                   double *__restrict__ im) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nfft) return;
  const double gsc = scaleinv * greensfn[i];
  re[i] *= gsc; im[i] *= gsc;
}

// b_i = - sum over the order^3 stencil of w_x w_y w_z u (pppm_conp.cpp:278-299); weights/indices cached per electrode atom
// like ele2rho / part2grid (aaa_map_rho :318-344).  One wavefront per electrode atom: lanes take (z, y) stencil rows, then a
// shuffle reduction; slots 1..3 of bk are cleared here.
// finish != 0 (round 4; one rank): the row of b is COMPLETED here -- slab term and the real-space pair sum (ra.breal, formed by the
// spread launch's spare blocks) with b_row's operations -- instead of by b_real_combine in a launch of its own.
__global__ __launch_bounds__(256) void pppm_gather_kernel(PppmDev pd, int ne, int ne_pad, const int *__restrict__ egrid /*[ne][3]*/,
                                                          const double *__restrict__ ew /*[ne][3][8]*/,
                                                          const double *__restrict__ u, double *bk /*not restrict: with `finish` b_row reads back what this kernel stored (ra.bk is the same buffer)*/, int finish, BRowArgs ra) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= ne) return;
  // (every wave derives the slab scalar with the same summation tree, like b_real_combine_kernel)
  const double sc = (finish && ra.slab) ? b_slab_scalar(ra, lane) : 0.0;
  const int gx = egrid[3 * i], gy = egrid[3 * i + 1], gz = egrid[3 * i + 2];
  const double *w = ew + (size_t)i * 24;
  const int o2 = pd.order * pd.order;
  double acc = 0.0;
  for (int row = lane; row < o2; row += 64) {
    const int n = row / pd.order, m = row - n * pd.order;
    const int mz = pwrap(n + pd.nlower + gz, pd.nz), my = pwrap(m + pd.nlower + gy, pd.ny);
    const double y0 = w[16 + n] * w[8 + m];
    const double *line = u + ((size_t)mz * pd.ny + my) * pd.nx;
    for (int l = 0; l < pd.order; ++l) acc -= (y0 * w[l]) * line[pwrap(l + pd.nlower + gx, pd.nx)];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) { bk[i] = acc; bk[ne_pad + i] = 0.0; bk[2 * ne_pad + i] = 0.0; bk[3 * ne_pad + i] = 0.0; }
  if (finish) b_row(ra, i, lane, sc);          // reads bk[i] .. bk[3 ne_pad + i] back as written by this lane 0: (k0 + 0) + (0 + 0) = k0
}

__global__ void zero_kernel(size_t n, double *p) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0;
}

static bool mesh_smooth(const PppmDev &pd) {
  FftPlan fp;
  return fft_factor(pd.nx, fp) && fft_factor(pd.ny, fp) && fft_factor(pd.nz, fp);
}

// forward (sign -1) or backward 3-D transform.  fused (radix path only): the forward transform takes real input and applies
// gscale*greensfn on its last pass; the backward one stores only the real part.
// raise a kernel's dynamic-LDS limit only when a launch needs more than it was last given (per device; see conp_kernels.hip)
struct LdsGrant { std::atomic<size_t> granted[64]; };
template <typename K>
static void grant_lds(K kernel, size_t bytes, LdsGrant &g) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<size_t> &v = g.granted[dev & 63];
  if (bytes <= v.load(std::memory_order_relaxed)) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  v.store(bytes, std::memory_order_relaxed);
}

// forward (sign -1) or backward 3-D transform.  fused (radix path only): the forward transform takes real input (from `re`, or
// from `im` when rho_in_im) and applies gscale*greensfn on its last pass; the backward one stores only the real part (and zeros
// into `im` when zero_im).
static void dft3(hipStream_t s, const PppmDev &pd, double sign, double *re, double *im, bool fused, double gscale,
                 bool rho_in_im = false, bool zero_im = false) {
  const int dims[3] = {pd.nx, pd.ny, pd.nz};
  int axis0 = 0;
  {
    FftPlan fpx, fpy;
    const size_t lds = (size_t)2 * pd.nx * pd.ny * sizeof(double2) + (size_t)2 * (pd.nx + pd.ny) * sizeof(double);
    if (fused && fft_factor(pd.nx, fpx) && fft_factor(pd.ny, fpy) && lds <= 128 * 1024) {
      // the x and y passes of every z-plane in one workgroup
      static LdsGrant g{};
      grant_lds(pppm_fft_xy_kernel, lds, g);
      hipLaunchKernelGGL(pppm_fft_xy_kernel, dim3(pd.nz), dim3(512), lds, s, pd.nx, pd.ny, sign, fpx, fpy, pd.twid[0], pd.twid[1], re, im,
                         sign < 0 ? (rho_in_im ? 2 : 1) : 0, 0, pd, PppmSpreadIn{}, BRowArgs{});
      axis0 = 2;
    } else if (rho_in_im) {
      // (the caller only asks for this with a fused xy pass; keep the contract honest if a mesh ever does not qualify)
      (void)hipMemcpyAsync(re, im, sizeof(double) * (size_t)pd.nfft, hipMemcpyDeviceToDevice, s);
    }
  }
  for (int axis = axis0; axis < 3; ++axis) {
    const int n = dims[axis];
    FftPlan fp;
    if (fft_factor(n, fp)) {              // 2,3,5-smooth length (every mesh LAMMPS picks): radix FFT
      int xs = 0;
      while (xs < 4 && (size_t)n * 32 * (2u << xs) <= 64 * 1024) ++xs;
      const int XT = 1 << xs;
      const int nlines = axis == 0 ? pd.ny * pd.nz : (axis == 1 ? pd.nx * pd.nz : pd.nx * pd.ny);
      const size_t lds = ((size_t)2 * n * XT) * sizeof(double2) + (size_t)2 * n * sizeof(double);
      int flags = 0;
      const double *gmul = nullptr;
      if (fused && sign < 0 && axis == 0) flags |= 1;
      if (fused && sign < 0 && axis == 2) gmul = pd.greensfn;
      if (fused && sign > 0 && axis == 2) flags |= zero_im ? 6 : 2;
      static LdsGrant g{};
      grant_lds(pppm_fft_kernel, lds, g);
      hipLaunchKernelGGL(pppm_fft_kernel, dim3((nlines + XT - 1) / XT), dim3(512), lds, s, pd.nx, pd.ny, pd.nz, axis, sign, fp,
                         pd.twid[axis], re, im, xs, flags, gmul, gscale);
      continue;
    }
    int XT = (int)(48 * 1024 / ((size_t)n * 16));
    XT = XT < 1 ? 1 : (XT > 16 ? 16 : XT);
    const int nlines = axis == 0 ? pd.ny * pd.nz : (axis == 1 ? pd.nx * pd.nz : pd.nx * pd.ny);
    const size_t lds = ((size_t)2 * n * XT + 2 * n) * sizeof(double);
    static LdsGrant g{};
    grant_lds(pppm_dft_kernel, lds, g);
    hipLaunchKernelGGL(pppm_dft_kernel, dim3((nlines + XT - 1) / XT), dim3(256), lds, s, pd.nx, pd.ny, pd.nz, axis, sign,
                       pd.twid[axis], re, im, XT);
  }
}

// rho -> u_brick in three launches when the mesh qualifies (2,3,5-smooth, a z-plane fits in LDS): x and y forward per plane (real
// input from `re`, or from `im` when rho_in_im), z forward * greensfn / N * z backward per bundle of lines, x and y backward per
// plane (real part only; zeros into `im` when zero_im).  Returns false (nothing launched) when the mesh does not qualify.
// sp != nullptr: no density brick at all -- the forward xy pass spreads the charges itself (real_in 3; pair rows in its spare blocks
// when ra holds any)
static bool poisson_three_launches(hipStream_t s, const PppmDev &pd, double *re, double *im, bool rho_in_im, bool zero_im,
                                   const PppmSpreadIn *sp = nullptr, const BRowArgs *ra = nullptr) {
  FftPlan fpx, fpy, fpz;
  const size_t lds_xy = (size_t)2 * pd.nx * pd.ny * sizeof(double2) + (size_t)2 * (pd.nx + pd.ny) * sizeof(double);
  if (!(fft_factor(pd.nx, fpx) && fft_factor(pd.ny, fpy) && fft_factor(pd.nz, fpz) && lds_xy <= 128 * 1024)) return false;
  const double gscale = 1.0 / ((double)pd.nx * pd.ny * pd.nz);
  static LdsGrant gxy{}, gz{};
  grant_lds(pppm_fft_xy_kernel, lds_xy, gxy);
  if (sp) {
    const BRowArgs rows = ra ? *ra : BRowArgs{};
    const int nrb = ra ? (rows.ne + 7) / 8 : 0;
    hipLaunchKernelGGL(pppm_fft_xy_kernel, dim3(pd.nz + nrb), dim3(512), lds_xy, s, pd.nx, pd.ny, -1.0, fpx, fpy, pd.twid[0], pd.twid[1], re,
                       im, 3, 0, pd, *sp, rows);
  } else
    hipLaunchKernelGGL(pppm_fft_xy_kernel, dim3(pd.nz), dim3(512), lds_xy, s, pd.nx, pd.ny, -1.0, fpx, fpy, pd.twid[0], pd.twid[1], re, im,
                       rho_in_im ? 2 : 1, 0, pd, PppmSpreadIn{}, BRowArgs{});
  int xs = 0;
  while (xs < 4 && (size_t)pd.nz * 32 * (2u << xs) <= 64 * 1024) ++xs;
  const int XT = 1 << xs, nlines = pd.nx * pd.ny;
  const size_t lds_z = ((size_t)2 * pd.nz * XT) * sizeof(double2) + (size_t)2 * pd.nz * sizeof(double);
  grant_lds(pppm_fft_kernel, lds_z, gz);
  hipLaunchKernelGGL(pppm_fft_kernel, dim3((nlines + XT - 1) / XT), dim3(512), lds_z, s, pd.nx, pd.ny, pd.nz, 2, -1.0, fpz, pd.twid[2], re, im,
                     xs, 8, pd.greensfn, gscale);
  hipLaunchKernelGGL(pppm_fft_xy_kernel, dim3(pd.nz), dim3(512), lds_xy, s, pd.nx, pd.ny, +1.0, fpx, fpy, pd.twid[0], pd.twid[1], re, im, 0,
                     zero_im ? 6 : 2, pd, PppmSpreadIn{}, BRowArgs{});
  return true;
}

// this rank's k-space b through the mesh: bk[0..ne) = PPPM b (slot 0), slots 1..3 zeroed
void launch_pppm_b(hipStream_t s, const PppmDev &pd, int nl, const int *elyte_idx, const double *x, const double *q, int ne,
                   int ne_pad, const int *egrid, const double *ew, double *re, double *im, double *slab_part, int *n_slab_part,
                   double *bk, bool *im_clean, const BRowArgs *pairs, double *breal_out, BRowArgs *fin, double *keep_rho) {
  const bool fused = mesh_smooth(pd);
  FftPlan fx, fy;
  // the density brick: `im` when the last backward pass left it all zero (*im_clean) -- one launch fewer per update
  const bool xy_fused = fused && fft_factor(pd.nx, fx) && fft_factor(pd.ny, fy) &&
                        (size_t)2 * pd.nx * pd.ny * sizeof(double2) + (size_t)2 * (pd.nx + pd.ny) * sizeof(double) <= 128 * 1024;
  const double gscale = 1.0 / ((double)pd.nx * pd.ny * pd.nz);
  // deck-sized systems: the forward xy pass spreads the charges itself (CONP_PPPM_SPREAD_LAUNCH: comparison switch, the launch of its own)
  // keep_rho: somebody wants the electrolyte density brick of this update (the make_rho override, pppm_conp.cpp:434-450) -- the brick
  // exists only on the path that spreads in a launch of its own; it is copied out before the forward transform overwrites it
  const bool spread_launch = path_on(CONP_PATH_PPPM_SPREAD_LAUNCH) || keep_rho != nullptr;      // (read per call: the test flips it between two handles)
  if (xy_fused && !spread_launch && nl <= 8192 && pd.nz <= 1024 && pd.order * pd.order <= 64) {
    BRowArgs ra{};
    const bool rows = pairs && breal_out;
    if (rows) ra = *pairs;
    const PppmSpreadIn sp{nl, elyte_idx, x, q, slab_part, breal_out};
    if (poisson_three_launches(s, pd, re, im, false, false, &sp, rows ? &ra : nullptr)) {
      *n_slab_part = pd.nz;
      if (im_clean) *im_clean = false;           // (`im` holds the z pass's imaginary parts now)
      BRowArgs fa{};
      if (fin) { fin->n_slab_part = *n_slab_part; fa = *fin; }
      hipLaunchKernelGGL(pppm_gather_kernel, dim3((ne + 3) / 4), dim3(256), 0, s, pd, ne, ne_pad, egrid, ew, re, bk, fin ? 1 : 0, fa);
      return;
    }
  }
  const bool use_im = xy_fused && im_clean != nullptr;
  double *rho = use_im ? im : re;
  if (!use_im || !*im_clean) hipLaunchKernelGGL(zero_kernel, dim3(512), dim3(256), 0, s, (size_t)pd.nfft, rho);
  if (!fused) hipLaunchKernelGGL(zero_kernel, dim3(512), dim3(256), 0, s, (size_t)pd.nfft, im);
  const int apb = 256 / spread_threads_per_atom(pd.order);
  const int ngroups = (nl + apb - 1) / apb > 0 ? (nl + apb - 1) / apb : 1;
  const int npass = (ngroups + 1023) / 1024;       // at most 1024 slab partial sums (launch_b_real_combine reads them per wave)
  const int nb = (ngroups + npass - 1) / npass;
  *n_slab_part = nb;
  // pairs / fin != NULL: the real-space pair sums ride in the spread launch and the gather completes the rows of b (one rank)
  BRowArgs ra{};
  int nrb = 0;
  if (pairs && breal_out) { ra = *pairs; nrb = (ra.ne + 3) / 4; }
  hipLaunchKernelGGL(pppm_spread_kernel, dim3(nb + nrb), dim3(256), 0, s, pd, nl, elyte_idx, x, q, rho, slab_part, npass, nb, ra, breal_out);
  if (keep_rho) (void)hipMemcpyAsync(keep_rho, rho, (size_t)pd.nfft * sizeof(double), hipMemcpyDeviceToDevice, s);
  if (!(xy_fused && poisson_three_launches(s, pd, re, im, use_im, use_im))) {
    dft3(s, pd, -1.0, re, im, fused, gscale, use_im, false);
    if (!fused)
      hipLaunchKernelGGL(pppm_greens_kernel, dim3((pd.nfft + 255) / 256), dim3(256), 0, s, pd.nfft, gscale, pd.greensfn, re, im);
    dft3(s, pd, +1.0, re, im, fused, gscale, false, use_im);
  }
  if (im_clean) *im_clean = use_im;
  BRowArgs fa{};
  if (fin) { fin->n_slab_part = *n_slab_part; fa = *fin; }
  hipLaunchKernelGGL(pppm_gather_kernel, dim3((ne + 3) / 4), dim3(256), 0, s, pd, ne, ne_pad, egrid, ew, re, bk, fin ? 1 : 0, fa);
}

// ---- PPPM coupling beyond b (pppm_conp.cpp:385-534) ---------------------------------------------------------------------------
// density brick of the atoms in `idx` (ele_make_rho :385-426 for the electrode atoms, elyte_make_rho for the electrolyte; their
// sum is what the make_rho override hands to PPPM::compute, :434-450).  rho is zeroed here.
void launch_pppm_density(hipStream_t s, const PppmDev &pd, int n, const int *idx, const double *x, const double *q, double *rho,
                         double *slab_scratch) {
  hipLaunchKernelGGL(zero_kernel, dim3(512), dim3(256), 0, s, (size_t)pd.nfft, rho);
  if (n <= 0) return;
  const int apb = 256 / spread_threads_per_atom(pd.order);
  const int ngroups = (n + apb - 1) / apb;
  const int npass = (ngroups + 1023) / 1024;
  const int nb = (ngroups + npass - 1) / npass;
  hipLaunchKernelGGL(pppm_spread_kernel, dim3(nb), dim3(256), 0, s, pd, n, idx, x, q, rho, slab_scratch, npass, nb, BRowArgs{}, nullptr);
}

// rho (in `re`) -> u_brick (in `re`): forward transform, greensfn / N, backward transform (elyte_poisson :230-267; the same
// arithmetic PPPM::poisson leaves in u_brick for per-atom energies)
void launch_pppm_poisson(hipStream_t s, const PppmDev &pd, double *re, double *im) {
  const bool fused = mesh_smooth(pd);
  const double gscale = 1.0 / ((double)pd.nx * pd.ny * pd.nz);
  if (fused && poisson_three_launches(s, pd, re, im, false, false)) return;
  if (!fused) hipLaunchKernelGGL(zero_kernel, dim3(512), dim3(256), 0, s, (size_t)pd.nfft, im);
  dft3(s, pd, -1.0, re, im, fused, gscale);
  if (!fused)
    hipLaunchKernelGGL(pppm_greens_kernel, dim3((pd.nfft + 255) / 256), dim3(256), 0, s, pd.nfft, gscale, pd.greensfn, re, im);
  dft3(s, pd, +1.0, re, im, fused, gscale);
}

// compute_group_potential / compute_particle_potential (:452-534): out[idx[k]] = - sum over the order^3 stencil of w u_brick
// (+ self * q_i: the 2 g / sqrt(pi) q_i of the particle flavour); stencil weights from the atom's current position
// (particle_map + compute_rho1d), one wavefront per atom
__global__ __launch_bounds__(256) void pppm_probe_kernel(PppmDev pd, int n, const int *__restrict__ idx, const double *__restrict__ x,
                                                         const double *__restrict__ q, const double *__restrict__ u, double self,
                                                         double *__restrict__ out) {
  __shared__ double coeff[64];
  if (threadIdx.x < pd.order * pd.order) coeff[threadIdx.x] = pd.rho_coeff[threadIdx.x];
  __syncthreads();
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= n) return;
  const int i = idx[k];
  int g[3];
  double w[3][8];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double xs = (x[3 * i + c] - pd.boxlo[c]) * pd.delinv[c];
    g[c] = static_cast<int>(xs + pd.shift) - 16384;
    rho1d_dev(coeff, pd.order, g[c] + pd.shiftone - xs, w[c]);
  }
  const int o2 = pd.order * pd.order;
  double acc = 0.0;
  for (int row = lane; row < o2; row += 64) {
    const int nn = row / pd.order, m = row - nn * pd.order;
    const int mz = pwrap(nn + pd.nlower + g[2], pd.nz), my = pwrap(m + pd.nlower + g[1], pd.ny);
    const double y0 = w[2][nn] * w[1][m];
    const double *line = u + ((size_t)mz * pd.ny + my) * pd.nx;
    for (int l = 0; l < pd.order; ++l) acc -= (y0 * w[0][l]) * line[pwrap(l + pd.nlower + g[0], pd.nx)];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) out[i] = acc + self * q[i];
}

void launch_pppm_probe(hipStream_t s, const PppmDev &pd, int n, const int *idx, const double *x, const double *q, const double *u,
                       double self, double *out) {
  if (n <= 0) return;
  hipLaunchKernelGGL(pppm_probe_kernel, dim3((n + 3) / 4), dim3(256), 0, s, pd, n, idx, x, q, u, self, out);
}

// ---- compute potential/atom, pair part (compute_potential_atom.cpp:223-308): one wavefront per list owner -------------------
__global__ __launch_bounds__(256) void potential_pair_kernel(int inum, const int *__restrict__ ilist, const int *__restrict__ numneigh,
                                                             const int *__restrict__ first, const int *__restrict__ neigh, int nlocal,
                                                             int newton, const double *__restrict__ x, const double *__restrict__ q,
                                                             const int *__restrict__ type, const int *__restrict__ sel,
                                                             const int *__restrict__ etasel, int ntypes,
                                                             const double *__restrict__ cutsq, double cut_coulsq, double g_ewald,
                                                             double eta, double *__restrict__ potential) {
#pragma clang fp contract(off)
  const int ii = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ii >= inum) return;
  const int i = ilist[ii];
  const int gci = sel[i];
  const double qi = q[i];
  double mine = 0.0;
  for (int jj = lane; jj < numneigh[i]; jj += 64) {
    const int j = neigh[first[i] + jj] & 0x3FFFFFFF;
    const int gcj = sel[j];
    if (!((gci || gcj) && (qi != 0.0 || q[j] != 0.0) && (newton || gci || j < nlocal))) continue;
    const double delx = x[3 * i] - x[3 * j], dely = x[3 * i + 1] - x[3 * j + 1], delz = x[3 * i + 2] - x[3 * j + 2];
    double rsq = delx * delx + dely * dely + delz * delz;
    if (rsq < 1e-10) rsq = 1e-10;
    if (!(rsq < cutsq[type[i] * (ntypes + 1) + type[j]] && rsq < cut_coulsq)) continue;
    const double r = sqrt(rsq), grij = g_ewald * r;
    double expm2 = exp(-grij * grij), t = 1.0 / (1.0 + 0.3275911 * grij);
    double erfc_ = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429)))) * expm2;
    double dudq = erfc_ / r;
    if (eta != 0.0) {
      const int ne2 = etasel[i] + etasel[j];
      if (ne2) {
        const double etarij = ne2 == 2 ? eta * r / sqrt(2.0) : eta * r;
        if (etarij < 5.8) {
          expm2 = exp(-etarij * etarij);
          t = 1.0 / (1.0 + 0.3275911 * etarij);
          erfc_ = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429)))) * expm2;
          dudq -= erfc_ / r;
        }
      }
    }
    if (gci) mine += q[j] * dudq;
    if (j < nlocal || newton) atomicAdd(&potential[j], qi * dudq);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
  if (lane == 0 && gci) atomicAdd(&potential[i], mine);
}

void launch_potential_pair(hipStream_t s, int inum, const int *ilist, const int *numneigh, const int *first, const int *neigh,
                           int nlocal, int newton, const double *x, const double *q, const int *type, const int *sel,
                           const int *etasel, int ntypes, const double *cutsq, double cut_coulsq, double g_ewald, double eta,
                           double *potential) {
  if (inum <= 0) return;
  hipLaunchKernelGGL(potential_pair_kernel, dim3((inum + 3) / 4), dim3(256), 0, s, inum, ilist, numneigh, first, neigh, nlocal, newton,
                     x, q, type, sel, etasel, ntypes, cutsq, cut_coulsq, g_ewald, eta, potential);
}

}  // namespace conp
