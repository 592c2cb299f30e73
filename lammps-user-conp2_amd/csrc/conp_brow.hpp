// Device helpers shared by conp_kernels.hip and conp_pppm.hip: the wave sum, the reference's erfc polynomial and pair potentials,
// and the assembly of one electrode row of b (k-space partials + slab term + real-space pair sum).  One definition, so that every
// kernel that finishes a row of b does it with the same operations in the same order.
#pragma once
#include <hip/hip_runtime.h>

#include "conp_kernels.h"

namespace conp {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---- erfc(x)/r through the reference's 5-term polynomial (fix_conp.cpp:53-60, 1446-1454) and the pair potentials -------------
__device__ __forceinline__ double erfcr_sqrt_dev(double a2_r2) {
#pragma clang fp contract(off)
  if (a2_r2 < 5.8 * 5.8) {
    const double a_r = sqrt(a2_r2);
    const double expm2 = exp(-a2_r2);
    const double t = 1.0 / (1.0 + 0.3275911 * a_r);
    return t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429)))) * expm2 / a_r;
  }
  return 0.0;
}

// pair_potential of the reference (fix_conp.cpp:1467-1475 eta_potential_A / eta_potential, :1561-1566 ehgo_potential)
__device__ __forceinline__ double pair_potential_dev(const RealParams &rp, double rsq, int ti, int tj, bool for_a) {
#pragma clang fp contract(off)
  if (rp.ehgo) {
    const double etaij = rp.eta_ij[ti * (rp.ntypes + 1) + tj], foij = rp.fo_ij[ti * (rp.ntypes + 1) + tj];
    const double etarij2 = etaij * etaij * rsq;
    return foij * exp(-0.5 * etarij2) - erfcr_sqrt_dev(etarij2) * etaij;
  }
  if (for_a) {
    const double etarij2 = rp.eta * rp.eta * rsq / 2;
    return -erfcr_sqrt_dev(etarij2) * rp.eta / sqrt(2.0);
  }
  return -erfcr_sqrt_dev(rp.eta * rp.eta * rsq) * rp.eta;
}

// One electrode row of b, by one wave (all 64 lanes return the same values):
//   b[row] = (bk0 + bk1) + (bk2 + bk3)                           (k-space shard, km_ewald.cpp:789-825)
//          - z_row * sum_j 4 pi q_j z_j / V                      (slab, km_ewald.cpp:827-847; rank 0 only)
//          - sum_pairs q_j [erfc(g r) - erfc(eta r)] / r         (rows row0..row1 only; fix_conp.cpp:1313-1353)
__device__ __forceinline__ double b_slab_scalar(const BRowArgs &a, int lane) {
#pragma clang fp contract(off)
  double sp = 0.0;
  for (int k = lane; k < a.n_slab_part; k += 64) sp += a.slab_part[k];
  sp = wave_sum(sp);
  return a.slab_pref * __shfl(sp, 0, 64);
}
// the real-space pair sum of one row by one wave (all lanes return it)
__device__ __forceinline__ double b_row_pairs(const BRowArgs &a, int row, int lane) {
#pragma clang fp contract(off)
  const int nt1 = a.rp.ntypes + 1;
  double sum = 0.0;
  if (row >= a.row0 && row < a.row1) {
    for (int p = a.row_ptr[row] + lane; p < a.row_ptr[row + 1]; p += 64) {
      const int ie = a.ele_atom[p], jo = a.oth_atom[p];
      const double dx = a.x[3 * ie] - a.x[3 * jo], dy = a.x[3 * ie + 1] - a.x[3 * jo + 1], dz = a.x[3 * ie + 2] - a.x[3 * jo + 2];
      const double rsq = dx * dx + dy * dy + dz * dz;
      if (rsq < a.rp.cutsq[a.type[ie] * nt1 + a.type[jo]] && rsq < a.rp.cut_coulsq) {
        double dudq = erfcr_sqrt_dev(a.rp.g_ewald * a.rp.g_ewald * rsq) * a.rp.g_ewald;
        dudq += pair_potential_dev(a.rp, rsq, a.type[ie], a.type[jo], false);
        sum -= a.q[jo] * dudq;
      }
    }
  }
  sum = wave_sum(sum);
  return __shfl(sum, 0, 64);
}
__device__ __forceinline__ void b_row(const BRowArgs &a, int row, int lane, double sc) {
#pragma clang fp contract(off)
  // a.breal != NULL: the pair sums were formed earlier in this update (by the spare blocks of elyte_phase_kernel)
  const double sum = a.breal ? a.breal[row] : b_row_pairs(a, row, lane);
  if (lane == 0) {
    double v = 0.0;
    if (a.add_k) {
      const double k0 = a.bk[row], k1 = a.bk[a.ne_pad + row], k2 = a.bk[2 * (size_t)a.ne_pad + row], k3 = a.bk[3 * (size_t)a.ne_pad + row];
      v = (k0 + k1) + (k2 + k3);
    }
    if (a.slab) v -= a.ele_z[row] * sc;
    v += sum;
    a.b_out[row] = v;
    if (a.slab && row == 0 && a.slab_out) *a.slab_out = sc;
  }
}

}  // namespace conp
