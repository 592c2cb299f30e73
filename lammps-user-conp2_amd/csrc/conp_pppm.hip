// PPPM k-space b vector on gfx950 (pppm_conp.cpp:109-316): charge spreading, Poisson solve on the mesh, stencil gather at
// the electrode atoms.  Double precision, one rank owns the whole mesh, periodic wrap instead of ghost planes
// (equivalent to GridComm's reverse/forward exchange on one rank, pppm_conp.cpp:114,122).
// The three 1-D transforms are plain O(n^2) DFTs per mesh line staged through LDS: correct for any mesh size (LAMMPS
// picks sizes with factors 2, 3, 5), fast enough for the decks' meshes (27x24x144); a radix FFT is the next step for large ones.
#include <hip/hip_runtime.h>

#include "conp_kernels.h"

namespace conp {

__device__ __forceinline__ int pwrap(int i, int n) { i %= n; return i < 0 ? i + n : i; }

__device__ __forceinline__ void rho1d_dev(const double *__restrict__ coeff, int order, double dx, double *w) {
  for (int k = 0; k < order; ++k) {
    double r = 0.0;
    for (int l = order - 1; l >= 0; --l) r = coeff[l * order + k] + r * dx;
    w[k] = r;
  }
}

// elyte_particle_map + elyte_make_rho (pppm_conp.cpp:126-228): one thread per charged electrolyte atom, order^3 atomic adds.
// Also the per-block partial sums of q z for the slab term (:301-314).
__global__ __launch_bounds__(256) void pppm_spread_kernel(PppmDev pd, int nl, const int *__restrict__ elyte_idx,
                                                          const double *__restrict__ x, const double *__restrict__ q,
                                                          double *__restrict__ rho, double *__restrict__ slab_part) {
  __shared__ double coeff[64];
  __shared__ double red[4];
  if (threadIdx.x < pd.order * pd.order) coeff[threadIdx.x] = pd.rho_coeff[threadIdx.x];
  __syncthreads();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  double qz = 0.0;
  if (j < nl) {
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
    qz = qq * x[3 * i + 2];
    const double z0 = pd.delvolinv * qq;
    for (int n = 0; n < pd.order; ++n) {
      const int mz = pwrap(n + pd.nlower + g[2], pd.nz);
      const double y0 = z0 * w[2][n];
      for (int m = 0; m < pd.order; ++m) {
        const int my = pwrap(m + pd.nlower + g[1], pd.ny);
        const double x0 = y0 * w[1][m];
        for (int l = 0; l < pd.order; ++l) {
          const int mx = pwrap(l + pd.nlower + g[0], pd.nx);
          atomicAdd(&rho[((size_t)mz * pd.ny + my) * pd.nx + mx], x0 * w[0][l]);
        }
      }
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

// rho(k) -> V(k): scale by 1/N and the influence function (pppm_conp.cpp:242-249)
__global__ void pppm_greens_kernel(int nfft, double scaleinv, const double *__restrict__ greensfn, double *__restrict__ re,
                                   double *__restrict__ im) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nfft) return;
  const double gsc = scaleinv * greensfn[i];
  re[i] *= gsc; im[i] *= gsc;
}

// b_i = - sum over the order^3 stencil of w_x w_y w_z u (pppm_conp.cpp:278-299); weights/indices cached per electrode atom
// like ele2rho / part2grid (aaa_map_rho :318-344)
__global__ void pppm_gather_kernel(PppmDev pd, int ne, const int *__restrict__ egrid /*[ne][3]*/,
                                   const double *__restrict__ ew /*[ne][3][8]*/, const double *__restrict__ u,
                                   double *__restrict__ bk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ne) return;
  const int gx = egrid[3 * i], gy = egrid[3 * i + 1], gz = egrid[3 * i + 2];
  const double *w = ew + (size_t)i * 24;
  double bbbtmp = 0.0;
  for (int n = 0; n < pd.order; ++n) {
    const int mz = pwrap(n + pd.nlower + gz, pd.nz);
    const double z0 = w[16 + n];
    for (int m = 0; m < pd.order; ++m) {
      const int my = pwrap(m + pd.nlower + gy, pd.ny);
      const double y0 = z0 * w[8 + m];
      for (int l = 0; l < pd.order; ++l) {
        const int mx = pwrap(l + pd.nlower + gx, pd.nx);
        const double x0 = y0 * w[l];
        bbbtmp -= x0 * u[((size_t)mz * pd.ny + my) * pd.nx + mx];
      }
    }
  }
  bk[i] = bbbtmp;
}

__global__ void zero_kernel(size_t n, double *p) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0;
}

static void dft3(hipStream_t s, const PppmDev &pd, double sign, double *re, double *im) {
  const int dims[3] = {pd.nx, pd.ny, pd.nz};
  for (int axis = 0; axis < 3; ++axis) {
    const int n = dims[axis];
    int XT = (int)(48 * 1024 / ((size_t)n * 16));
    XT = XT < 1 ? 1 : (XT > 16 ? 16 : XT);
    const int nlines = axis == 0 ? pd.ny * pd.nz : (axis == 1 ? pd.nx * pd.nz : pd.nx * pd.ny);
    const size_t lds = ((size_t)2 * n * XT + 2 * n) * sizeof(double);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pppm_dft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(pppm_dft_kernel, dim3((nlines + XT - 1) / XT), dim3(256), lds, s, pd.nx, pd.ny, pd.nz, axis, sign,
                       pd.twid[axis], re, im, XT);
  }
}

// this rank's k-space b through the mesh: bk[0..ne) = PPPM b (slot 0), slots 1..3 zeroed
void launch_pppm_b(hipStream_t s, const PppmDev &pd, int nl, const int *elyte_idx, const double *x, const double *q, int ne,
                   int ne_pad, const int *egrid, const double *ew, double *re, double *im, double *slab_part, int *n_slab_part,
                   double *bk) {
  hipLaunchKernelGGL(zero_kernel, dim3(512), dim3(256), 0, s, (size_t)pd.nfft, re);
  hipLaunchKernelGGL(zero_kernel, dim3(512), dim3(256), 0, s, (size_t)pd.nfft, im);
  hipLaunchKernelGGL(zero_kernel, dim3(64), dim3(256), 0, s, (size_t)4 * ne_pad, bk);
  const int nb = (nl + 255) / 256 > 0 ? (nl + 255) / 256 : 1;
  *n_slab_part = nb;
  hipLaunchKernelGGL(pppm_spread_kernel, dim3(nb), dim3(256), 0, s, pd, nl, elyte_idx, x, q, re, slab_part);
  dft3(s, pd, -1.0, re, im);
  hipLaunchKernelGGL(pppm_greens_kernel, dim3((pd.nfft + 255) / 256), dim3(256), 0, s, pd.nfft,
                     1.0 / ((double)pd.nx * pd.ny * pd.nz), pd.greensfn, re, im);
  dft3(s, pd, +1.0, re, im);
  hipLaunchKernelGGL(pppm_gather_kernel, dim3((ne + 255) / 256), dim3(256), 0, s, pd, ne, egrid, ew, re, bk);
}

}  // namespace conp
