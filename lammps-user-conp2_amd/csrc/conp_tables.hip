// gfx950 kernels for the once-per-run electrode phase tables (KSpaceModuleEwald::sincos_a_ele / sincos_a_comm_eleall,
// km_ewald.cpp:426-531: csk / snk of every electrode atom for every flat k entry).
//
// The reference fills csk[i][kflat], snk[i][kflat] per atom: libm cos / sin of unitk * x per axis (:441-442), the angle-addition
// recurrence along each axis (:445-449), then the (kx, +-ky) products (:464-477).  Here the tables are written straight into the
// layouts the other kernels read -- Xe / Ye (axis phases, k-major), Tz (z phases in G's column layout), Rp (planar phases in G's
// row layout) -- by two launches; nothing of size Ne x kflat exists on the host any more (it was 33 of the 61 ms of the setup at
// Ne = 4096 and 0.57 of 1.1 s at Ne = 16384, plus a 47-MB upload).
// What stays on the host: the 3 Ne seed pairs (cos, sin)(unitk_c * x_ic) by libm -- the reference's own calls, so that every table
// entry keeps the reference's bits (device sincos is not bit-identical to glibc's); everything that is O(Ne * kflat) runs here,
// with FMA contraction off: the same multiplies and adds in the same order as km_ewald.cpp.
#include "conp_kernels.h"

namespace conp {

// one thread per (atom, axis).  seeds: [6][ne] = cx, sx, cy, sy, cz, sz.
//   x: Xe[k][i] = (c_k, s_k), k = 0 .. kxmax (row 0 = (1, 0)); row kxmax + 1 stays zero (padding planar vectors point there)
//   y: Ye[k][i], k = 0 .. kymax
//   z: Tz[col_c(m)][i] = c_m, Tz[col_s(m)][i] = s_m, m = 0 .. nz - 1   (KPlan::col_c: 320 ct + 16 (ml >> 3) + (ml & 7))
__global__ __launch_bounds__(128) void ele_axis_kernel(int ne, int ne_pad, const double *__restrict__ seeds, int kxmax, int kymax,
                                                       int nz, int kzt, double2 *__restrict__ Xe, double2 *__restrict__ Ye,
                                                       double *__restrict__ Tz) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (i >= ne) return;
  const double c1 = seeds[(size_t)(2 * c) * ne + i], s1 = seeds[(size_t)(2 * c + 1) * ne + i];
  if (c < 2) {
    double2 *t = (c == 0 ? Xe : Ye) + i;
    const int nrow = c == 0 ? kxmax : kymax;
    t[0] = make_double2(1.0, 0.0);
    double cm = c1, sm = s1;
    if (nrow >= 1) t[ne_pad] = make_double2(c1, s1);
    for (int m = 2; m <= nrow; ++m) {
      const double cn = cm * c1 - sm * s1;          // km_ewald.cpp:446-447
      const double sn = sm * c1 + cm * s1;
      cm = cn; sm = sn;
      t[(size_t)m * ne_pad] = make_double2(cm, sm);
    }
  } else {
    auto col_c = [kzt](int m) { const int ct = m / kzt, ml = m - ct * kzt; return 320 * ct + 16 * (ml >> 3) + (ml & 7); };
    Tz[(size_t)col_c(0) * ne_pad + i] = 1.0;        // m = 0: (1, 0); the sin row stays zero
    double cm = c1, sm = s1;
    for (int m = 1; m < nz; ++m) {
      if (m > 1) {
        const double cn = cm * c1 - sm * s1;
        const double sn = sm * c1 + cm * s1;
        cm = cn; sm = sn;
      }
      const size_t cc = (size_t)col_c(m);
      Tz[cc * ne_pad + i] = cm;
      Tz[(cc + 8) * ne_pad + i] = sm;
    }
  }
}

// one thread per (atom, planar vector p < np): Rp[row_a(p)][i], Rp[row_b(p)][i] from the axis phases, the products of
// km_ewald.cpp:464-477 (b_zc_final_kernel rebuilds the same rows per update with the same expression):
//   (kx, +ky): c = cx cy - sx sy,  s = cx sy + sx cy        (kx, -ky): c = cx cy + sx sy,  s = -cx sy + sx cy
// axis vectors read row 0 = (1, 0) of the other axis: c = cx * 1 - sx * 0 = cx exactly.
__global__ __launch_bounds__(256) void ele_planar_kernel(int ne, int ne_pad, int np, const int *__restrict__ p_ikx,
                                                         const int *__restrict__ p_iky, const int *__restrict__ p_sgn,
                                                         const double2 *__restrict__ Xe, const double2 *__restrict__ Ye,
                                                         double *__restrict__ Rp) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ne) return;
  for (int p = blockIdx.y; p < np; p += gridDim.y) {
    const double2 X = Xe[(size_t)p_ikx[p] * ne_pad + i], Y = Ye[(size_t)p_iky[p] * ne_pad + i];
    const double sy = p_sgn[p] < 0 ? -Y.y : Y.y;
    const size_t ra = (size_t)((p >> 6) * 128 + (p & 63));
    Rp[ra * ne_pad + i] = X.x * Y.x - X.y * sy;
    Rp[(ra + 64) * ne_pad + i] = X.x * sy + X.y * Y.x;
  }
}

// z-class phase tables: Tzc[t][c] = Tz[t][rep[c]] ([C_pad][64]) and the class-major copy TzcT[c][t] ([nzc][C_pad])
__global__ __launch_bounds__(256) void ele_zclass_kernel(int C_pad, int ne_pad, int nzc, const int *__restrict__ rep,
                                                         const double *__restrict__ Tz, double *__restrict__ Tzc,
                                                         double *__restrict__ TzcT) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= C_pad * nzc) return;
  const int c = e / C_pad, t = e - c * C_pad;
  const double v = Tz[(size_t)t * ne_pad + rep[c]];
  Tzc[(size_t)t * 64 + c] = v;
  TzcT[(size_t)c * C_pad + t] = v;
}

void launch_ele_tables(hipStream_t s, const DevPlan &pl, int kzt, int ne, int ne_pad, const double *seeds, double2 *Xe, double2 *Ye,
                       double *Tz, double *Rp) {
  if (ne <= 0) return;
  hipLaunchKernelGGL(ele_axis_kernel, dim3((ne + 127) / 128, 3), dim3(128), 0, s, ne, ne_pad, seeds, pl.kxmax, pl.kymax, pl.nz, kzt, Xe,
                     Ye, Tz);
  const int py = pl.np < 256 ? pl.np : 256;
  hipLaunchKernelGGL(ele_planar_kernel, dim3((ne + 255) / 256, py), dim3(256), 0, s, ne, ne_pad, pl.np, pl.p_ikx, pl.p_iky, pl.p_sgn,
                     (const double2 *)Xe, (const double2 *)Ye, Rp);
}

void launch_ele_zclass(hipStream_t s, int C_pad, int ne_pad, int nzc, const int *rep, const double *Tz, double *Tzc, double *TzcT) {
  if (nzc <= 0) return;
  hipLaunchKernelGGL(ele_zclass_kernel, dim3((C_pad * nzc + 255) / 256), dim3(256), 0, s, C_pad, ne_pad, nzc, rep, Tz, Tzc, TzcT);
}

}  // namespace conp
