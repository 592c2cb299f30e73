// gfx950 (MI355X, CDNA4) kernels of the constant-potential charge solver.  wave = 64 lanes.
//
// Data model: every half-space Ewald vector of the reference list (km_ewald.cpp:285-361) is
// (planar vector p, kz index m, sign).  With a_pj = q_j cos(theta_pj), b_pj = q_j sin(theta_pj),
// c_mj = cos(m uz z_j), s_mj = sin(m uz z_j) the four real products
//        G[(p,a|b)][(m,c|s)] = sum_j {a,b}_pj {c,s}_mj
// hold all structure factors (km_ewald.cpp:668-780): S_re(p,+-m) = CC -+ SS,  S_im(p,+-m) = SC +- CS.
// That contraction over the electrolyte atoms is the per-step hot spot and runs on the FP64 matrix cores
// (v_mfma_f64_16x16x4_f64).  Fragment layout used throughout (verified on hardware, tools/microbench):
//   A operand: lane l holds A[row = l & 15][k = l >> 4];  B operand: lane l holds B[k = l >> 4][col = l & 15];
//   C/D: register r of lane l is C[row = (l >> 4) + 4 r][col = l & 15].
#include "conp_kernels.h"
#include "conp_brow.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace conp {

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

const char *env_knob(const char *name) {
  const char *v = getenv(name);
  if (v) {
    static std::atomic<unsigned long long> said{0};
    unsigned h = 0;
    for (const char *c = name; *c; ++c) h = h * 31u + (unsigned char)*c;
    const unsigned long long bit = 1ull << (h & 63u);
    if (!(said.fetch_or(bit) & bit))
      fprintf(stderr, "libconp_hip: %s=%s is set -- a non-default code path is in use\n", name, v);
  }
  return v;
}
static std::atomic<unsigned> g_paths{0};
static std::atomic<int> g_sk_nwg{0};
bool path_on(unsigned bit) {
  const unsigned m = g_paths.load(std::memory_order_relaxed);
  if (m & bit) {
    static std::atomic<unsigned> said{0};
    if (!(said.fetch_or(bit) & bit)) fprintf(stderr, "libconp_hip: test path 0x%x selected (conp_debug_set_paths) -- a non-default code path is in use\n", bit);
    return true;
  }
  return false;
}
int debug_sk_workgroups() { return g_sk_nwg.load(std::memory_order_relaxed); }
extern "C" void conp_debug_set_paths(unsigned mask) { g_paths.store(mask, std::memory_order_relaxed); }
extern "C" void conp_debug_set_sk_workgroups(int n) { g_sk_nwg.store(n > 0 ? n : 0, std::memory_order_relaxed); }

__device__ double block_sum_1024(double v, double *red);

// cache-policy experiments (comparison builds only; the product leaves both at 0): non-temporal partial-tile stores / loads,
// non-temporal phase-table stores
#ifndef SK_PART_NT
#define SK_PART_NT 0
#endif
#ifndef EP_TABLE_NT
#define EP_TABLE_NT 0
#endif
__device__ __forceinline__ void st_d2(double2 *p, double2 v, bool nt) {
  if (nt) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
  else *p = v;
}
__device__ __forceinline__ double2 ld_d2(const double2 *p, bool nt) {
  if (nt) { double2 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
  return *p;
}

// ================================================================================================
// 1. electrolyte phase tables  (km_ewald.cpp:685-724: libm cos/sin of unitk*x, then the angle-addition
//    recurrence c_m = c_{m-1} c_1 - s_{m-1} s_1, s_m = s_{m-1} c_1 + c_{m-1} s_1)
//    Xt[kx][j], Yt[ky][j], Zt[m][j] as (cos, sin), k-major / atom-contiguous like the reference's cs/sn.
//    Row 0 of each table is (1, 0).  Also compacts q and emits per-block partial sums of q*z (slab, :835-841).
// ================================================================================================
// One thread per (atom, axis): blockIdx.y = 0 (x), 1 (y), 2 (z) -- the three recurrences are independent, the z one is the
// long one (nz steps), so splitting them puts 3x as many wavefronts on the chip for the same serial depth.
// Blocks [0, 3 nb): the phase tables (axis c = block / nb).  Blocks beyond: the real-space pair sums of the electrode rows
// (fix_conp.cpp:1313-1353), two rows per block -- they depend on x, q only, so they ride along in this launch (the chip is far
// from full with the 3 nb phase blocks) instead of costing gather latency in a launch of their own after the structure factors.
constexpr int EP_THREADS = 128;
constexpr int ZN_SPLIT = 4;
__global__ __launch_bounds__(EP_THREADS) void elyte_phase_kernel(int nb, int nl, int nl_pad, const int *__restrict__ elyte_idx,
                                                          const double *__restrict__ x, const double *__restrict__ q,
                                                          double ux, double uy, double uz, int kxmax, int kymax, int nz,
                                                          int kzt, int nrz, double2 *__restrict__ Xt, double2 *__restrict__ Yt,
                                                          double2 *__restrict__ Zs, double *__restrict__ qc,
                                                          double *__restrict__ slab_part, BRowArgs ra, double *__restrict__ breal_out,
                                                          int j0, int j1 /* atoms whose tables are wanted (a rank's share) */,
                                                          ZnWindow zw /* .Bt != null: the z-window form -- no z phase table, the window matrix instead */) {
#pragma clang fp contract(off)
  // z-window form: ZN_SPLIT threads per atom on the z axis (a quarter of the window columns each: the exponentials are the launch's
  // longest chain otherwise), so 2 nb + ZN_SPLIT nb table blocks; else 3 nb
  const int zsplit = zw.Bt ? ZN_SPLIT : 1, ntab = (2 + zsplit) * nb;
  if ((int)blockIdx.x >= ntab) {
    const int row = ((int)blockIdx.x - ntab) * (EP_THREADS / 64) + (threadIdx.x >> 6);
    if (row < ra.ne) {
      const double v = b_row_pairs(ra, row, threadIdx.x & 63);
      if ((threadIdx.x & 63) == 0) breal_out[row] = v;
    }
    return;
  }
  const int c = (int)blockIdx.x < 2 * nb ? (int)blockIdx.x / nb : 2, bx = (int)blockIdx.x - c * nb;
  const int jt = bx * blockDim.x + threadIdx.x;
  const int j = c == 2 ? jt / zsplit : jt, part = c == 2 ? jt - j * zsplit : 0;
  double qz = 0.0;
  // several ranks: tables only for this rank's atoms [j0, j1) (whole wavefronts of the others leave here); the compact charges and
  // the slab sum's q z (z axis) cover all atoms on every rank
  const bool tab = j >= j0 && j < j1;
  if (j < nl_pad && !tab && c == 2 && part == 0) {
    double xc = 0, qq = 0;
    if (j < nl) {
      const int i = elyte_idx[j];
      xc = x[3 * i + c]; qq = q[i];
    }
    qc[j] = qq; qz = qq * xc;
  }
  if (j < nl_pad && tab) {
    double xc = 0, qq = 0;
    if (j < nl) {
      const int i = elyte_idx[j];
      xc = x[3 * i + c]; qq = q[i];
    }
    const double ang = (c == 0 ? ux : (c == 1 ? uy : uz)) * xc;
    const int nrow = c == 0 ? kxmax + 1 : (c == 1 ? kymax + 1 : nz);
    // tables are blocked by 16 atoms (one sk_gemm chunk): element (row, atom j) at ((j >> 4) * NR + row) * 16 + (j & 15), so that
    // everything a workgroup loads for a chunk sits in a few contiguous KB instead of one 256-byte piece per 512-KB row
    const int NR = c == 0 ? kxmax + 2 : (c == 1 ? kymax + 1 : nrz);
    double2 *t = (c == 0 ? Xt : (c == 1 ? Yt : Zs)) + ((size_t)(j >> 4) * NR) * 16 + (j & 15);
    // X and Y: every row.  Z: sk_gemm's thread (gs = 8 q + r, atom) generates the five kz values 40 q + r + 8 u (u < 5) of a
    // column tile (160 kz) from ONE seed by repeated rotation with the 8-kz step: row 0 of Zs is that step, (cos, sin)(8 uz z);
    // row 1 + 32 ct + gs is the seed, the phase at m = kzt ct + 40 q + r (kzt = kz values per column tile, KPlan::kzt).  Same recurrence as the reference's, re-associated.
    // the x table carries the charge (q cos, q sin): one multiply here instead of two per planar vector in sk_gemm
    const double sc = (c == 0) ? qq : 1.0;
    double c1 = 1.0, s1 = 0.0;
    if (c != 2 || !zw.Bt) sincos(ang, &s1, &c1);            // (the z-window form needs no z phases: four threads per atom would pay for them)
    auto rot = [](double &cr, double &sr, double cw, double sw) {       // (cr, sr) *= (cw, sw), the reference's angle addition
      const double cn = cr * cw - sr * sw;
      const double sn = sr * cw + cr * sw;
      cr = cn; sr = sn;
    };
    if (c != 2) {
      if (c == 0) t[(size_t)(kxmax + 1) * 16] = make_double2(0.0, 0.0);   // row for padding planar vectors (no contribution)
      t[0] = make_double2(sc, 0.0);
      double cm = c1, sm = s1;
      if (nrow > 1) t[16] = make_double2(sc * c1, sc * s1);
      for (int m = 2; m < nrow; ++m) {
        rot(cm, sm, c1, s1);
        st_d2(t + (size_t)m * 16, make_double2(sc * cm, sc * sm), EP_TABLE_NT);
      }
    } else if (zw.Bt) {
      // the z-window form (conp_zn.hip): this atom's row of the window matrix, dense over its chunk's columns:
      // Bt[(chunk * ncol + col) * 16 + atom] = phi((g0[chunk] + col) - u), u = z n / Lz' wrapped relative to the window origin
      if (part == 0) { qc[j] = qq; qz = qq * xc; }
      // fragment order: Bt[chunk][column block][atom group a >> 2][column & 15][a & 3]  (zn_gemm_kernel's lane (fr, fk) reads 4 atoms)
      double *bt = zw.Bt + ((size_t)(j >> 4) * zw.ncol) * 16 + ((j & 15) >> 2) * 64 + (j & 3);
      const int cw = zw.ncol / ZN_SPLIT, col0 = part * cw;                 // this thread's columns (ncol = 32 or 48)
      if (j < nl) {
        double ur = xc * zw.gscale - (double)zw.g0c[j >> 4];
        ur -= (double)zw.n * rint(ur / (double)zw.n);
        const int i0 = (int)ceil(ur - 0.5 * zw.W);
        if (part == 0 && (i0 < 0 || i0 + zw.W > zw.ncol)) *reinterpret_cast<volatile int *>(zw.flag) = 1;   // (page-locked host word: a tap left its window)
        for (int col = col0; col < col0 + cw; ++col) {
          const double d = ((double)col - ur) * (2.0 / zw.W);            // in units of the window's half width
          double v = 0.0;
          if (d > -1.0 && d < 1.0) v = exp(zw.beta * (sqrt(1.0 - d * d) - 1.0));
          bt[(size_t)(col >> 4) * 256 + (col & 15) * 4] = v;
        }
      } else
        for (int col = col0; col < col0 + cw; ++col) bt[(size_t)(col >> 4) * 256 + (col & 15) * 4] = 0.0;
    } else {
      qc[j] = qq; qz = qq * xc;
      const int nct = (nrz - 1) / 32;                 // column tiles
      // eight seeds in flight (r = 0 .. 7), advanced together by 40 kz per step inside a column tile, and their copies at the
      // tile's start by the tile's kz count kzt per tile: the dependent chain is 7 unit steps, a handful to the 40-kz and kzt
      // rotations, then one step per column tile + 3 inside it (a walk in unit steps would be 160 per tile: the launch's longest)
      double bc8[8], bs8[8];
      bc8[0] = 1.0; bs8[0] = 0.0;
#pragma unroll
      for (int r = 1; r < 8; ++r) { bc8[r] = bc8[r - 1]; bs8[r] = bs8[r - 1]; rot(bc8[r], bs8[r], c1, s1); }
      double c8 = bc8[7], s8 = bs8[7];
      rot(c8, s8, c1, s1);                            // the 8-kz rotation
      t[0] = make_double2(c8, s8);
      double c40 = c8, s40 = s8;
      rot(c40, s40, c8, s8);                          // 16
      rot(c40, s40, c40, s40);                        // 32
      rot(c40, s40, c8, s8);                          // 40
      double ct_c = 1.0, ct_s = 0.0;                  // the kzt-kz rotation: (8-kz rotation)^(kzt / 8), square and multiply
      {
        double pc = c8, ps = s8;
        for (int e = kzt >> 3; e; e >>= 1) {
          if (e & 1) rot(ct_c, ct_s, pc, ps);
          rot(pc, ps, pc, ps);
        }
      }
      for (int ct = 0; ct < nct; ++ct) {
        double sc8[8], ss8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { sc8[r] = bc8[r]; ss8[r] = bs8[r]; }
#pragma unroll
        for (int q40 = 0; q40 < 4; ++q40) {
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            st_d2(t + (size_t)(1 + 32 * ct + 8 * q40 + r) * 16, make_double2(sc8[r], ss8[r]), EP_TABLE_NT);
            if (q40 < 3) rot(sc8[r], ss8[r], c40, s40);
          }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) rot(bc8[r], bs8[r], ct_c, ct_s);
      }
    }
  }
  if (c != 2) return;
  // block partial of sum q z
  __shared__ double red[EP_THREADS / 64];
  double v = wave_sum(qz);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < EP_THREADS / 64; ++w) tot += red[w];
    slab_part[bx] = tot;
  }
}

// ghost atoms as periodic images of owned atoms: x_g = x_owner + n * prd, q_g = q_owner (conp_env.ghost_images).
// `n * prd` is formed first and then added, like Comm::pack_comm adds `pbc * prd` -- the same bits as LAMMPS' own ghosts.
__global__ void ghost_fill_kernel(int nlocal, int nghost, const int *__restrict__ owner, const int *__restrict__ img,
                                  double px, double py, double pz, double *__restrict__ x, double *__restrict__ q) {
#pragma clang fp contract(off)
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nghost) return;
  const int o = owner[g], i = nlocal + g;
  const double dx = img[3 * g] * px, dy = img[3 * g + 1] * py, dz = img[3 * g + 2] * pz;
  x[3 * i] = x[3 * o] + dx; x[3 * i + 1] = x[3 * o + 1] + dy; x[3 * i + 2] = x[3 * o + 2] + dz;
  q[i] = q[o];
}

void launch_ghost_fill(hipStream_t s, int nlocal, int nghost, const int *owner, const int *img, double px, double py, double pz,
                       double *x, double *q) {
  if (nghost <= 0) return;
  hipLaunchKernelGGL(ghost_fill_kernel, dim3((nghost + 255) / 256), dim3(256), 0, s, nlocal, nghost, owner, img, px, py, pz, x, q);
}

void launch_elyte_phase(hipStream_t s, int nl, int nl_pad, const int *elyte_idx, const double *x, const double *q,
                        double ux, double uy, double uz, int kxmax, int kymax, int nz, int kzt, int nrz, double2 *Xt,
                        double2 *Yt, double2 *Zs, double *qc, double *slab_part, int *n_slab_part, const BRowArgs *rows,
                        double *breal_out, int j0, int j1, const ZnWindow *zw) {
  const int nb = (nl_pad + EP_THREADS - 1) / EP_THREADS, zsplit = zw ? ZN_SPLIT : 1;
  *n_slab_part = zsplit * nb;                              // (the z blocks leave the partial sums of q z)
  BRowArgs ra{};
  int nrb = 0;
  if (rows && breal_out) { ra = *rows; nrb = (ra.ne + EP_THREADS / 64 - 1) / (EP_THREADS / 64); }
  hipLaunchKernelGGL(elyte_phase_kernel, dim3((2 + zsplit) * nb + nrb), dim3(EP_THREADS), 0, s, nb, nl, nl_pad, elyte_idx, x, q, ux, uy, uz, kxmax,
                     kymax, nz, kzt, nrz, Xt, Yt, Zs, qc, slab_part, ra, breal_out, j0, j1, zw ? *zw : ZnWindow{});
}

// ================================================================================================
// 2. structure-factor contraction on the FP64 matrix cores.
//    Work item = (row tile rt: 64 planar vectors = 128 G rows, col tile ct: up to 20 column fragments = 160 kz = 320 G cols,
//    atom split).  A column fragment = 8 kz: 8 cos + 8 sin columns.  Only the leading column fragments of a row fragment hold
//    listed k vectors (sphere cut-off) and only those are computed; they are dealt round-robin to the 4 column groups of the
//    workgroup.  Workgroup = 512 threads = 8 waves = 2 row halves x 4 column groups; wave tile = 4 x (<=5) fragments.
//    Atoms are walked in chunks of J = 16 through a double-buffered LDS operand panel, panel[feature][atom]:
//      features   0..127 : a_pj / b_pj  = q_j (cos,sin)(theta_pj)  from the X / Y phase tables (one complex product)
//      features 128..447 : c_mj / s_mj  regenerated from a z-phase seed (one per thread: kz = 40 q + r, then steps of 8) with the
//                          reference's angle-addition recurrence (km_ewald.cpp:709-719)
//    Rows are NOT padded: the atom column is XOR-swizzled with the low four bits of the feature index (SK_LD below), which makes
//    every MFMA-fragment ds_read_b64 and every 16-lane ds_write_b64 conflict-free.
//    Software pipeline per chunk: issue global loads for chunk c+1 -> MFMA on chunk c -> build panel c+1 -> barrier (waves 4-7:
//    build first, multiply second).  Partial tiles go to part[segment] in fragment-major order (sk_part_off); sk_reduce sums a
//    tile's segments in a fixed order (deterministic).
// ================================================================================================
#ifndef SK_LATE_MODE
#define SK_LATE_MODE 1
#endif
// raise a kernel's dynamic-LDS limit, only when a launch needs more than it was last given: per-update launches must not
// pay a runtime call each (the decks' updates are bound by host launch cost)
// The attribute is per device: the cache is indexed by the calling thread's current device (handles on several GPUs in one
// process are supported, conp_env.device), and atomic because hosts may drive handles from different threads (a lost race
// only sets the attribute twice).
struct DynLdsCache { std::atomic<size_t> granted[64]; };
template <typename K>
static void ensure_dyn_lds(K kernel, size_t bytes, DynLdsCache &cache) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<size_t> &g = cache.granted[dev & 63];
  if (bytes <= g.load(std::memory_order_relaxed)) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  g.store(bytes, std::memory_order_relaxed);
}

constexpr int SK_J = 16;
// LDS operand panel: panel[feature][16 atoms], NO padding -- the atom column is XOR-swizzled with the low four bits of the
// feature index: element (feature, atom j) lives at feature * 16 + (j ^ (feature & 15)).  An MFMA fragment read (16 consecutive
// features x the 2 atoms 4 ks + fk of a 32-lane group) then touches all 64 banks exactly once as ds_read_b64 AND all 32 banks
// once per 16-lane group should the compiler pair two reads into ds_read2st64_b64; the 16 atoms of one feature a 16-lane group
// writes are a permutation of one 128-byte row: conflict-free ds_write_b64.  (The first version padded rows to 17 doubles:
// clean for the paired form only -- every single ds_read_b64 had a 2-way conflict, SQ_LDS_BANK_CONFLICT ~ 1 per LDS instruction.)
constexpr int SK_LD = SK_J;
// rows of a panel: 32 RF operand rows of the band's planar vectors (RF = 4 or 5 row fragments per half: 'a' rows then 'b' rows), then
// the z features, 16 per column fragment.  A band of RF = 4 takes up to 20 column fragments (128 + 320 rows), a band of RF = 5 up to
// 16 (160 + 256); the z build writes whole groups of five fragments per thread, so the RF = 5 layout may spill into rows 416..479 --
// inside the buffer, read by nobody.
constexpr int SK_NF = 160 + 320;
constexpr int SK_PANEL = SK_NF * SK_LD;          // doubles per buffer (61,440 bytes)
constexpr unsigned SK_BUF1 = 65536;              // byte offset of the second buffer: switching buffers is one XOR of a byte address
constexpr size_t SK_LDS_BYTES = SK_BUF1 + (size_t)SK_PANEL * sizeof(double);
// Measured and left off (round 5, tools/ab_libs.sh on one box, `make variant_nobar VDEF=-DSK_NOBAR=1`): the chunk loop WITHOUT the
// workgroup barrier -- sk_gemm 244.6-252.9 us against 234.5-236.6 with the barrier (headline size; parity tests green in both forms).
// Why it loses (tools/microbench/pipe_share*_bench.hip, profiles/r05_pipe_share.txt): the SIMD's issue arbiter serves the OLDER wave
// first -- a wave that streams MFMAs keeps its rate (64.0 cycles per MFMA) whatever its younger partner does, and the partner gets what
// is left (its own MFMAs: 13 % of the pipe; VALU / LDS instructions: one per ~30 cycles instead of ~10).  Waves 0-3 are the older
// ones: let loose, they run a chunk ahead and take the pipe from the late waves' multiply of the previous chunk, then wait for it in
// front of their next build -- while both roles build at the same time (nobody multiplies); the barrier is what keeps the two
// roles' multiply phases apart, back to back.  Also left off: SK_LOAD_AHEAD (the early waves' table loads in front of the barrier
// instead of behind it): 237.1-237.5 vs 234.5-236.6 us.
// The form without the barrier: four counters in the gap between the two panel buffers -- built[b] counts the
// wave-builds written into buffer b, done[b] the wave-multiplies that have finished reading it (both monotonic inside a segment,
// zeroed at its start).  A wave waits for the panel it needs (8 builds per chunk) or for the buffer it is about to overwrite (8
// multiplies per chunk), not for the slowest wave of the workgroup: the early and the late waves of a SIMD may drift by up to a
// chunk against each other.  LDS operations of one wave complete in order, so the ds_add behind a wave's panel writes is its release;
// the waiting side has the counter's value back (it branches on it) before it requests operands.
#ifndef SK_NOBAR
#define SK_NOBAR 0
#endif
// SK_LOAD_AHEAD: the early waves request the table rows of chunk c + 2 right behind their build of chunk c + 1 -- in front of the
// chunk's barrier, where they wait for the late waves anyway -- instead of at the top of the next chunk, between the barrier and
// their first multiply (the address arithmetic and issue of 6-8 loads: ~150 cycles of every chunk on the workgroup's critical path)
#ifndef SK_LOAD_AHEAD
#define SK_LOAD_AHEAD 0
#endif
constexpr unsigned SK_CNT = (unsigned)SK_PANEL * sizeof(double);     // byte 61440: built[0], built[1], done[0], done[1]
static_assert(SK_CNT + 16 <= SK_BUF1, "counters sit in the gap between the panel buffers");
typedef __attribute__((address_space(3))) char *sk_lds_ptr;
// (inline assembly on purpose: through a generic pointer the compiler reads the counter with a FLAT load and waits for vmcnt(0) --
//  every table load in flight -- at each poll; and its atomic optimiser wraps a one-lane add in a wave reduction)
__device__ __forceinline__ void sk_cnt_signal(char *smem, unsigned idx) {
  const unsigned a = (unsigned)(unsigned long)(sk_lds_ptr)smem + SK_CNT + 4u * idx;
  unsigned long long save;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0"
               : "=&s"(save) : "v"(a), "v"(1u) : "memory");
}
__device__ __forceinline__ void sk_cnt_wait(char *smem, unsigned idx, unsigned target) {
  const unsigned a = (unsigned)(unsigned long)(sk_lds_ptr)smem + SK_CNT + 4u * idx;
  for (;;) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= target) break;
    __builtin_amdgcn_s_sleep(1);
  }
}

// one step of the angle-addition recurrence without FMA contraction (same arithmetic as elyte_phase_kernel)
__device__ __forceinline__ double2 zstep(double2 z, double2 st) {
  // contraction allowed here (2 mul + 2 fma instead of 4 mul + 2 add): at most 4 steps from a seed that elyte_phase computed
  // with the reference's arithmetic, and FP64 VALU time is MFMA time on this chip (shared pipe)
  double2 r;
  r.x = fma(z.x, st.x, -(z.y * st.y));
  r.y = fma(z.y, st.x, z.x * st.y);
  return r;
}

// Partial tiles (sk_gemm -> sk_reduce) are stored MFMA-fragment-major: the 16 x 16 fragment (row fragment f16 of 8, column
// fragment fi of 20) is one run of 256 doubles, inside it the accumulator register pair (r, r + 1), r even, of lane (fk, fr) is one
// 16-byte unit (two ADJACENT registers: the store needs no register shuffling):
//   element (row = 16 f16 + 4 r + fk, col = 16 fi + fr)  at  (f16 * 20 + fi) * 256 + (r >> 1) * 128 + (16 fk + fr) * 2 + (r & 1)
__device__ __forceinline__ unsigned sk_part_off(int rowl, int col) {
  const int r = (rowl >> 2) & 3;
  return (unsigned)((((rowl >> 4) * 20 + (col >> 4)) << 8) + ((r >> 1) << 7) + ((((rowl & 3) << 4) + (col & 15)) << 1) + (r & 1));
}

// the thread index WITHOUT the hardware's v0: wave index (a scalar the kernel derives once) and the lane's position in the wave.
// v0 kept alive across the chunk loops for the segment's epilogue costs a register the 20-fragment bodies do not have (the
// allocator spilled it and reloaded it once per chunk).
// (volatile assembly: formed where it is asked for -- as a plain expression it is loop-invariant, hoisted to the top of the kernel
//  and spilled like v0 was)
__device__ __forceinline__ unsigned sk_tid(int wave) {
  unsigned lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  return ((unsigned)wave << 6) | lane;
}
struct SkRaw {        // raw inputs of one thread for one chunk
  double2 X0, Y0, X1, Y1, Zseed, Zst;     // Zseed: phase at the thread's first kz of the column tile, Zst: the 8-kz rotation
  double2 X2, Y2;                         // RF = 5 bands: the fifth row fragment's vector of the threads gs < 16 (the early waves)
};

struct SkCtx {        // per-thread constants of one work item
  SkItem it;
  int nl_pad, nz;
  int gj, gs;                       // generation role: atom gj of the chunk, sub-index gs 0..31
  unsigned xy0, xy1, xy2, zoff;     // (row * 16 + gj) inside a chunk block of the phase tables: X in the low, Y in the high half-word
  unsigned nrx16, nry16, nrz16;                // rows per chunk block * 16 (X, Y, Z tables)
  bool neg0, neg1, neg2;            // ky < 0: flips the sign of sin(ky y)
  bool zact;                        // this thread's kz values reach into an active column fragment
  unsigned za;                      // doubles offset of (cos feature of the thread's first kz, atom gj) in a panel (swizzled)
  int dsin;                         // from a cos feature to its sin feature
  int fr, fk, rh, cg;
  int wave;                         // rh + 2 cg (wave-uniform)
  unsigned wa;                      // generation: doubles offset of (feature gs, atom gj) in a panel (swizzled)
  // MFMA fragments: this lane's element of A / B fragment 0 at k-step ks sits at byte  base + ((ks ^ p) << 5),  p = fr >> 2
  // (atom 4 ks + fk, swizzled by the row: (4 ks + fk) ^ fr = ((ks ^ p) << 2) | ((fk ^ fr) & 3))
  unsigned base_a, base_b, pq;      // pq = p << 5
  int f0;                           // row fragments f >= f0 skip the wave's last column fragment (sphere culling), wave-uniform
  const double2 *Xt, *Yt, *Zs;
  const double *qc;
  const SkProj *proj;               // projected output (planar electrodes, sk_project_out) when not null
  int dbg;
  int nfr;                          // row fragments (16 planar vectors each) the plan has: 4 n_row_tiles (a band's padding lies beyond)
};

template <bool THIRD>
__device__ __forceinline__ void sk_load_raw(const SkCtx &c, int ch, SkRaw &r) {
  // blocked tables: element (row, atom) of chunk ch at (ch * NR + row) * 16 + (atom & 15); the row offsets already hold row * 16 + gj
  const unsigned bx = (unsigned)ch * c.nrx16, by = (unsigned)ch * c.nry16, bz = (unsigned)ch * c.nrz16;
  r.X0 = c.Xt[bx + (c.xy0 & 0xffffu)]; r.Y0 = c.Yt[by + (c.xy0 >> 16)];
  r.X1 = c.Xt[bx + (c.xy1 & 0xffffu)]; r.Y1 = c.Yt[by + (c.xy1 >> 16)];
  if constexpr (THIRD) { r.X2 = c.Xt[bx + (c.xy2 & 0xffffu)]; r.Y2 = c.Yt[by + (c.xy2 >> 16)]; }
  if (c.zact) { r.Zst = c.Zs[bz + c.gj]; r.Zseed = c.Zs[bz + c.zoff]; }
}

// RF row fragments per half: 'a' rows 0 .. 16 RF - 1, 'b' rows 16 RF .. 32 RF - 1, z features from row 32 RF.  THIRD: this thread
// (gs < 16 of an RF = 5 band) also builds planar vector 64 + gs, the fifth row fragment.
template <int RF, bool THIRD>
__device__ __forceinline__ void sk_build_panel(const SkCtx &c, const SkRaw &r, double *pn) {
  // (kx, sg*ky): q cos = (q cx) cy - (q sx)(sg sy) ; q sin = (q cx)(sg sy) + (q sx) cy   (km_ewald.cpp:739-747); the X table
  // already carries q, padding rows read an all-zero X row
  // sg is +1 or -1 (0 only on padding rows, whose X row is all zero anyway): a sign flip, not an FP64 multiply
  // features gs, 32 + gs, 64 + gs (a) and 16 RF + the same (b) share their low four bits: one swizzled column for all of them
  {
    const double sy = c.neg0 ? -r.Y0.y : r.Y0.y;
    pn[c.wa] = r.X0.x * r.Y0.x - r.X0.y * sy;
    pn[c.wa + 16 * RF * SK_LD] = r.X0.x * sy + r.X0.y * r.Y0.x;
  }
  {
    const double sy = c.neg1 ? -r.Y1.y : r.Y1.y;
    pn[c.wa + 32 * SK_LD] = r.X1.x * r.Y1.x - r.X1.y * sy;
    pn[c.wa + (32 + 16 * RF) * SK_LD] = r.X1.x * sy + r.X1.y * r.Y1.x;
  }
  if constexpr (THIRD) {
    const double sy = c.neg2 ? -r.Y2.y : r.Y2.y;
    pn[c.wa + 64 * SK_LD] = r.X2.x * r.Y2.x - r.X2.y * sy;
    pn[c.wa + (64 + 16 * RF) * SK_LD] = r.X2.x * sy + r.X2.y * r.Y2.x;
  }
  if (c.zact) {
    // thread (gs = 8 q + r, gj) owns the kz values 40 q + r + 8 u (u < 5) of the column tile: column fragments 5 q + u, position r.
    // Column fragment kz >> 3 = 8 cos features then 8 sin features (KPlan::col_c / col_s): the thread's cos features are 16 rows
    // apart with the SAME swizzle column (gj ^ r) -- one address register and immediate offsets; the sin feature sits 8 rows below
    // its cos feature, its swizzle key has bit 3 set: column (gj ^ r) ^ 8, i.e. 8 doubles to the right or to the left depending
    // on bit 3 of gj alone.  Threads whose first kz lies beyond the tile's sphere cut skip the lot (zact) -- with 40 kz per 8
    // sub-indices those are whole wavefronts, the high ones: the late waves (4-7) of a tile that is cut short have little to
    // build before they multiply.  (kz beyond nz - 1: the recurrence just runs on -- finite values in G columns that carry zero
    // weight and no listed k.)
    // What this replaced, measured: ten separately computed offsets (the first form of the 8-kz layout) cost either ten
    // registers -- spills in the NFW = 5 bodies -- or per-chunk integer arithmetic on the build's critical path (3 % at the
    // headline size, 7 % at 16384 / 262144); kz values strided by 32 (every thread busy on every tile) lost the short build of
    // the late waves: 12 % at the headline size.
    // Z_1 = Z_0 W by one complex product, then the three-term recurrence of the angle addition, Z_{u+1} = 2 cos(d) Z_u - Z_{u-1}
    // (d = the 8-kz step): two FMAs per value instead of two multiplies and two FMAs -- 11 FP64 operations for the five values
    // instead of 16, on the pipe the MFMAs run on
    double2 Zp = r.Zseed, Z = zstep(r.Zseed, r.Zst);
    const double tc = r.Zst.x + r.Zst.x;
    double *pz = pn + c.za;
    pz[0] = Zp.x;
    pz[c.dsin] = Zp.y;
#pragma unroll
    for (int u = 1; u < 5; ++u) {
      pz[u * 16 * SK_LD] = Z.x;
      pz[u * 16 * SK_LD + c.dsin] = Z.y;
      if (u < 4) {
        const double2 Zn = make_double2(fma(tc, Z.x, -Zp.x), fma(tc, Z.y, -Zp.y));
        Zp = Z; Z = Zn;
      }
    }
  }
}

// MFMA phase of one chunk for a wave that owns NFW column fragments (fi = 4 g + cg, g < NFW) x 4 row fragments f.
// Row-fragment-major.  Fragment registers: the NFW B values of the current k-step, one A value in use and the NEXT A value
// already on its way (issued before the <= NFW MFMAs of the current row fragment); during the last row fragment of a k-step
// every B value is re-read for the next k-step right after its last use.  So each LDS read has about NFW MFMAs (64 cycles
// apiece) between issue and first use, with 2 NFW + 4 fragment registers instead of 2 NFW + 8 -- a wave that multiplies alone
// (its SIMD partner is building the next panel) no longer stalls on LDS at the top of every k-step (tools/sk_stamp.py: that
// was ~1900 of the ~7900 cycles of a chunk).  After the last k-step the "next" reads wrap to k-step 0 of the same buffer:
// six reads nobody uses instead of a branch.
// Sphere culling: row fragments f >= f0 skip the wave's LAST column fragment (the host orders planar vectors by |k_p|, so the
// kz cut never grows with f; a fragment whose cut is shorter still than NFW - 1 multiplies a few zero-weight columns: G entries
// no listed k reads).  One wave-uniform branch per row fragment instead of one per MFMA.
#define SK_LDS_F64(byte_addr) (*reinterpret_cast<const double *>(smem + (byte_addr)))
template <int RF, int NFW>
__device__ __forceinline__ void sk_mfma_chunk(const SkCtx &c, const char *smem, unsigned buf, d4 (&acc)[RF][NFW > 0 ? NFW : 1]) {
  if constexpr (NFW > 0) {
    constexpr unsigned FA = 16 * SK_LD * 8, FB = 64 * SK_LD * 8;        // bytes between row fragments / between this wave's column fragments
    const unsigned ba = c.base_a ^ buf, bb = c.base_b ^ buf;
    double bf[NFW], a0, a1;
    unsigned q = c.pq;                                                  // (ks ^ p) << 5 for ks = 0: the 32-byte group of k-step 0
    // Issue order matters: LDS reads retire in order and the compiler's s_waitcnt at the top of the k-step loop must cover the
    // loop entry AND the back edge.  A first, then the B fragments one instruction each (a paired ds_read2st64 here would merge
    // two counter slots), exactly the order the loop body re-reads them in: the waits become lgkmcnt(NFW), (NFW-1), ... -- every
    // fragment gets ~NFW MFMAs of cover.  (B first / A last made the loop head wait for the read issued JUST before it:
    // one full LDS round trip per k-step with the matrix pipe idle.)
    a0 = SK_LDS_F64(ba + q);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NFW; ++g) { bf[g] = SK_LDS_F64(bb + q + g * FB); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll 1
    for (int ks = 0; ks < SK_J / 4; ++ks) {
      const unsigned qn = (unsigned)(((ks + 1) & 3) << 5) ^ c.pq;       // the group of the next k-step (wraps after the last)
      const unsigned an = ba + qn, bn = bb + qn, ac = ba + q;
#pragma unroll
      for (int f = 0; f < RF; ++f) {
        a1 = f < RF - 1 ? SK_LDS_F64(ac + (f + 1) * FA) : SK_LDS_F64(an);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g + 1 < NFW; ++g) {
          acc[f][g] = MFMA_F64(a0, bf[g], acc[f][g]);
          if (f == RF - 1) { bf[g] = SK_LDS_F64(bn + g * FB); __builtin_amdgcn_sched_barrier(0); }
        }
        if (f < c.f0) acc[f][NFW - 1] = MFMA_F64(a0, bf[NFW - 1], acc[f][NFW - 1]);      // wave-uniform
        if (f == RF - 1) bf[NFW - 1] = SK_LDS_F64(bn + (NFW - 1) * FB);
        __builtin_amdgcn_sched_barrier(0);
        a0 = a1;
      }
      q = qn;
    }
  }
}

// Straight-line form of the same phase: all four k-steps unrolled, the sphere cut F0 a template parameter -- no loop, no branch,
// no scalar bookkeeping between MFMAs.  The (k-step, lane)-dependent 32-byte group of the swizzle costs one address register per
// k-step and operand side (8 in all, XORed with the buffer bit once per chunk); fragment / column-group offsets are immediates.
// The first reads of a chunk's MFMA phase -- the A value of row fragment 0 and the wave's NFW B values of k-step 0 -- can be
// requested apart from the phase itself (SkPre): a LATE wave requests them BEFORE it builds the next panel, so that they queue
// ahead of the build's fourteen LDS writes (LDS operations of a wave complete in order) and the first MFMAs do not wait for the
// write drain behind the build.
template <int NFW>
struct SkPre { double a0; double bf[NFW > 0 ? NFW : 1]; };
template <int NFW>
__device__ __forceinline__ void sk_mfma_prefetch(const SkCtx &c, const char *smem, unsigned buf, SkPre<NFW> &pre) {
  if constexpr (NFW > 0) {
    constexpr unsigned FB = 64 * SK_LD * 8;
    const unsigned q = c.pq;
    pre.a0 = SK_LDS_F64((c.base_a ^ buf) + q);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NFW; ++g) { pre.bf[g] = SK_LDS_F64((c.base_b ^ buf) + q + g * FB); __builtin_amdgcn_sched_barrier(0); }
  }
}
template <int RF, int NFW, int F0>
__device__ __forceinline__ void sk_mfma_chunk_u(const SkCtx &c, const char *smem, unsigned buf, d4 (&acc)[RF][NFW > 0 ? NFW : 1],
                                                const SkPre<NFW> *pre = nullptr) {
  if constexpr (NFW > 0) {
    constexpr unsigned FA = 16 * SK_LD * 8, FB = 64 * SK_LD * 8;
    unsigned aa[4], ab[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned q = (unsigned)(ks << 5) ^ c.pq;
      aa[ks] = (c.base_a ^ buf) + q;
      ab[ks] = (c.base_b ^ buf) + q;
    }
    double bf[NFW], a0, a1;
    if (pre) {
      a0 = pre->a0;
#pragma unroll
      for (int g = 0; g < NFW; ++g) bf[g] = pre->bf[g];
    } else {
      a0 = SK_LDS_F64(aa[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < NFW; ++g) { bf[g] = SK_LDS_F64(ab[0] + g * FB); __builtin_amdgcn_sched_barrier(0); }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kn = (ks + 1) & 3;
#pragma unroll
      for (int f = 0; f < RF; ++f) {
        // the last row fragment of the last k-step has nothing left to fetch
        if (!(ks == 3 && f == RF - 1)) a1 = f < RF - 1 ? SK_LDS_F64(aa[ks] + (f + 1) * FA) : SK_LDS_F64(aa[kn]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g + 1 < NFW; ++g) {
          acc[f][g] = MFMA_F64(a0, bf[g], acc[f][g]);
          if (f == RF - 1 && ks < 3) { bf[g] = SK_LDS_F64(ab[kn] + g * FB); __builtin_amdgcn_sched_barrier(0); }
        }
        if (f < F0) acc[f][NFW - 1] = MFMA_F64(a0, bf[NFW - 1], acc[f][NFW - 1]);
        if (f == RF - 1 && ks < 3) bf[NFW - 1] = SK_LDS_F64(ab[kn] + (NFW - 1) * FB);
        __builtin_amdgcn_sched_barrier(0);
        a0 = a1;
      }
    }
  }
}

// Round 5: the same MFMAs and the same registers (NFW B values, two A values), the operand reads gathered into FEWER interruptions
// of the MFMA stream.  tools/microbench/sk_reads_bench.hip (profiles/r05_sk_reads.txt): a single wave's stream loses ~6.6 % against
// constant operand registers, and loses the same when the reads go into registers nobody multiplies -- it is not the waiting, it is
// every PLACE where a non-MFMA instruction sits between two MFMAs (~10 cycles each, whether one read is issued there or four).  The
// form above has RF + NFW such places per k-step (an A read in front of every row-fragment group, a B re-read behind every MFMA of
// the last group); this one has RF: the last group re-reads the wave's B values in two bursts -- half of them and the next k-step's
// first A value behind its middle MFMA, the rest and the next k-step's second A value behind its last -- and the groups 1 .. RF-2
// read the A value of the group behind them as before.  Microbenchmark, one wave per SIMD: 0.715 -> 0.737 of peak (constant
// operands: 0.764).
template <int RF, int NFW, int F0>
__device__ __forceinline__ void sk_mfma_chunk_v(const SkCtx &c, const char *smem, unsigned buf, d4 (&acc)[RF][NFW > 0 ? NFW : 1]) {
  if constexpr (NFW > 0) {
    constexpr unsigned FA = 16 * SK_LD * 8, FB = 64 * SK_LD * 8;
    unsigned aa[4], ab[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned q = (unsigned)(ks << 5) ^ c.pq;
      aa[ks] = (c.base_a ^ buf) + q;
      ab[ks] = (c.base_b ^ buf) + q;
    }
    double bf[NFW], av[2];                 // av[p]: the A value of the current group, av[p ^ 1]: the next one's (p flips per group)
    av[0] = SK_LDS_F64(aa[0]);
    if (RF > 1) av[1] = SK_LDS_F64(aa[0] + FA);
#pragma unroll
    for (int g = 0; g < NFW; ++g) bf[g] = SK_LDS_F64(ab[0] + g * FB);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int H = (NFW + 1) / 2;        // B values re-read in the first burst of the last group
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kn = (ks + 1) & 3;
#pragma unroll
      for (int f = 0; f < RF; ++f) {
        const int p = (ks * RF + f) & 1;    // (compile-time: everything is unrolled)
        // in front of the groups 1 .. RF-2: the A value of the group behind this one, into the register the previous group freed
        if (f >= 1 && f + 1 < RF) { av[p ^ 1] = SK_LDS_F64(aa[ks] + (f + 1) * FA); __builtin_amdgcn_sched_barrier(0); }
        const int nm = f < F0 ? NFW : NFW - 1;              // MFMAs of this group (the sphere cut drops the last column fragment)
        if (f < RF - 1 || ks == 3) {
#pragma unroll
          for (int g = 0; g < NFW; ++g) if (g < nm) acc[f][g] = MFMA_F64(av[p], bf[g], acc[f][g]);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          // the last group of a k-step that has a successor: two bursts
#pragma unroll
          for (int g = 0; g < NFW; ++g) if (g < H && g < nm) acc[f][g] = MFMA_F64(av[p], bf[g], acc[f][g]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < H; ++g) bf[g] = SK_LDS_F64(ab[kn] + g * FB);
          if (RF > 1) av[p ^ 1] = SK_LDS_F64(aa[kn]);       // the next k-step's first A value (this group has no successor in ITS k-step)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < NFW; ++g) if (g >= H && g < nm) acc[f][g] = MFMA_F64(av[p], bf[g], acc[f][g]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = H; g < NFW; ++g) bf[g] = SK_LDS_F64(ab[kn] + g * FB);
          if (RF > 1) av[p] = SK_LDS_F64(aa[kn] + FA);      // ... and its second, into the register this group is done with
          else av[p ^ 1] = SK_LDS_F64(aa[kn]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
}

// The product multiplies with the straight-line form; -DSK_MFMA_LOOP=1 (`make loopform`) builds the round-2 loop form for A/B runs.
#ifndef SK_MFMA_LOOP
#define SK_MFMA_LOOP 0
#endif
#if SK_MFMA_LOOP
#define SK_MFMA_PHASE(RF, NFW, F0, c, smem, buf, acc) sk_mfma_chunk<RF, NFW>(c, smem, buf, acc)
#define SK_LATE_PREFETCH 0
#else
// Measured in the kernel and left off (tools/ab_libs.sh, one box): sk_gemm 237.1-237.7 us with the burst form against 234.8-235.3 (headline),
// 676.4-679.8 against 673.9-676.5 (slab geometry) -- what a single wave gains in the microbenchmark its SIMD partner was already filling
// in the kernel, and the next k-step's first A value has two MFMAs of cover instead of a group's.
#ifndef SK_BURST_READS
#define SK_BURST_READS 0
#endif
#if SK_BURST_READS
#define SK_MFMA_PHASE(RF, NFW, F0, c, smem, buf, acc) sk_mfma_chunk_v<RF, NFW, F0>(c, smem, buf, acc)
#else
#define SK_MFMA_PHASE(RF, NFW, F0, c, smem, buf, acc) sk_mfma_chunk_u<RF, NFW, F0>(c, smem, buf, acc)
#endif
// Measured and left off (round 4, one box, `make variant_pre VDEF=-DSK_LATE_PREFETCH=1`): the late waves' first operand reads requested
// ahead of their panel build -- sk_gemm 243.5 vs 238.6 us at the headline size, 734 vs 734 us in the slab geometry.  The reads
// queue ahead of the build's LDS writes as intended, but NFW + 1 more live registers through the build and a wait for them in
// front of it cost more than the write drain they skip.
#ifndef SK_LATE_PREFETCH
#define SK_LATE_PREFETCH 0
#endif
// Early waves: first operand reads ahead of the next chunk's table loads.  Measured (tools/ab_libs.sh, one box each): in every body,
// sk_gemm 235.9 -> 234.2 us at the headline size, 677 us either way in the slab geometry -- but four chunk loops of the bodies that
// hold all 20 accumulator fragments then spill (tests/test_host_logic.py); restricted to the bodies with headroom: 235.2 vs 235.0 us,
// and 17.24 vs 17.15 ms at 16384 / 262144.  Left off.
#ifndef SK_EARLY_PREFETCH
#define SK_EARLY_PREFETCH 0
#endif
// issue priority of the LATE waves (build first, multiply second: they arrive last at every barrier, the early waves wait ~a
// quarter of a chunk for them): SK_PRIO > 0 raises them above their SIMD partner -- 3 through their build, SK_PRIO through their
// multiply phase -- the early waves stay at 0
#ifndef SK_PRIO
#define SK_PRIO 0
#endif
// the same for the EARLY waves (multiply first): 1 = raised through their multiply phase only, 2 = through multiply and build.
// Measured on one box (tools/ab_libs.sh, sk_gemm us, headline / slab geometry): late waves raised (SK_PRIO 1 / 2) 243.5 / 243.6 vs
// 237.9 and 709 / 709 vs 680 -- worse; early waves raised through the multiply phase (SK_PRIO_E 1) 238.2 vs 237.6 and 675.7 vs 678.4;
// through multiply and build (2) 236.5 and 673.4: the product's setting (3, the highest priority, reads the same: 235.5 / 674 either way).
#ifndef SK_PRIO_E
#define SK_PRIO_E 2
#endif
#endif
// Ablation switches (phases of the kernel turned off, TIMING ONLY, results are garbage) exist in the diagnostic build
// -DSK_ABLATE only (`make ablate`, tools/sk_ablate.sh); in the product build the tests below fold to constants.
#ifdef SK_ABLATE
#define SK_DBG(c, bit) ((c).dbg & (bit))
#else
#define SK_DBG(c, bit) 0
#endif

// ---- diagnostic build only (-DSK_STAMP, `make stamp`, tools/sk_stamp.py): s_memtime stamps around the phases of a chunk,
// summed per wave in scalar registers and stored once per segment into a buffer nothing else reads.  In the product build no
// stamp executes.  Its fences forbid overlaps the real kernel has: read the SHARES, never the length (guide, "In-kernel stamps").
#ifdef SK_STAMP
__device__ unsigned long long sk_stamp_buf[1024 * 8 * 8];   // [workgroup][wave][prologue, load issue, mfma, build, barrier, epilogue, chunks, late]
__device__ unsigned long long sk_clock_buf[1024 * 4];       // [workgroup][s_memtime start, end, s_memrealtime (100 MHz) start, end]
__device__ unsigned long long sk_seg_buf[4096 * 4];         // [segment][workgroup << 32 | rt, nbf, chunks, s_memrealtime ticks]: cost-model fit
#define SK_STAMP_T(t)                                                          \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");  \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#define SK_STAMP_ADD(sum, a, b) sum += (b) - (a)
#else
#define SK_STAMP_T(t) do { } while (0)
#define SK_STAMP_ADD(sum, a, b) do { } while (0)
#endif

// Planar electrodes: every electrode atom's z phase is one of a few columns ("z classes", b_hc_kernel), and all the update needs
// from G is  Hc[r][zc] = sum_t w[r][t] G[r][t] Tzc[t][zc]  -- linear in G, so a segment projects ITS partial tile before it leaves
// the registers and writes 128 x nzc doubles instead of 128 x 320: the 59 MB of partial tiles per update (headline size), the
// launch that summed and projected them (15 us) and their write-back go away; the segments' Hc pieces (8 KB each) are added in a
// fixed order by whoever consumes them (hc_sum_kernel / b_zc_final_kernel).
//   lane (fk, fr) of wave (rh, cg) holds G[16 (4 rh + f) + fk + 4 r][16 (4 g + cg) + fr]: it forms, per class, the sum over ITS
//   columns (g) of w G Tzc for its 16 rows; the 16 lanes fr of a row are added by DPP exchanges inside the register file (a
//   transposing butterfly, row16_sum4), the four column groups cg through 4 KB of LDS per class.  Two barriers per segment -- a first version
//   went through LDS per row fragment (eight barriers, a dozen serialised LDS / memory waits per fragment): 8 us per segment.
constexpr int SK_HC = 8;          // classes a segment's piece has room for
constexpr int SK_HC_ROWS = 160;   // rows it has room for: 32 RF of the widest band
constexpr int SK_HC_MAX = 8;
// the value the lane's partner under DPP control CTRL holds (row_mirror 0x140: lane ^ 15, row_half_mirror 0x141: lane ^ 7 inside
// its half row, quad_perm [3,2,1,0] 0x1B: lane ^ 3, quad_perm [1,0,3,2] 0xB1: lane ^ 1)
template <int CTRL>
__device__ __forceinline__ double dpp_get(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// four values per lane, each to be added over the 16 lanes of its row: a transposing butterfly -- a lane keeps the values whose
// index has ITS lane bit and adds the partner's copy of them (2, then 1 exchange), then two plain exchange-adds: five instead of
// 4 x 4.  Every lane of quad-group q = fr >> 2 ends up with the total of value q.  Fixed association: bitwise reproducible.
__device__ __forceinline__ double row16_sum4(const double (&s)[4], unsigned fr) {
  const bool b3 = fr & 8, b2 = fr & 4;
  double k2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) k2[i] = (b3 ? s[2 + i] : s[i]) + dpp_get<0x140>(b3 ? s[i] : s[2 + i]);
  double k = (b2 ? k2[1] : k2[0]) + dpp_get<0x141>(b2 ? k2[0] : k2[1]);
  k += dpp_get<0x1B>(k);
  k += dpp_get<0xB1>(k);
  return k;
}
// The piece of a segment is BAND-LOCAL: [nzc][32 RF rows] (the band's 'a' rows, then its 'b' rows), class stride 32 RF -- for a
// band that is a row tile of the plan (RF = 4, g0 a multiple of 4) exactly the [class][128] piece the dot kernel adds itself;
// hc_sum_kernel maps a band's rows to the plan's.
template <int RF, int NFW>
__device__ __forceinline__ void sk_project_out(const SkCtx &c, char *smem, d4 (&acc)[RF][NFW > 0 ? NFW : 1], double *__restrict__ hout) {
  double *L = reinterpret_cast<double *>(smem);
  // everything this needs is derived HERE, behind opaque copies of the thread index and the parameter block's address: left to
  // itself the compiler forms the lane's offsets and loads the parameters at the top of the kernel and carries them through the
  // chunk loop -- 32 more SGPRs and spills in a kernel that has no register to spare
  unsigned t = sk_tid(c.wave);
  asm volatile("" : "+v"(t));
  const SkProj *pp = c.proj;
  asm volatile("" : "+s"(pp));
  // (constant address space spelled out: scalar loads, wave-uniform values)
  const __attribute__((address_space(4))) SkProj *cp = (const __attribute__((address_space(4))) SkProj *)pp;
  const int nzc = cp->nzc;
  const unsigned cpad = (unsigned)cp->cpad;
  const double *wfull = cp->wfull, *tzt = cp->tzt;
  const unsigned fr = t & 15, fk = (t >> 4) & 3, rh = (t >> 6) & 1, cg = t >> 7;
  const unsigned colb = (unsigned)c.it.ct * 320 + 16 * cg + fr;                                  // + 64 g
  // (global address space spelled out: the pointers come out of memory behind an asm barrier, the compiler would use FLAT loads,
  //  which count on the LDS counter too -- every wait for an LDS read then waits for the weights in flight)
  typedef __attribute__((address_space(1))) const double *gcd_t;
  // (the 'a' and 'b' rows of a planar vector carry the same weight, KPlan::wfull: both row halves read the 'a' rows -- the second
  //  wave of a column group finds the lines in L2)
  // row fragment f of the band is fragment g0 + f of the plan: rows (g >> 2) * 128 + (g & 3) * 16 .. of wfull (a band's padding
  // fragments lie beyond the plan: they read the last fragment's weights and hold zero accumulators)
  const unsigned g0 = (unsigned)c.it.g0, glast = (unsigned)c.nfr - 1;
  auto wrow = [&](unsigned f) { const unsigned g = g0 + f < glast ? g0 + f : glast; return (g >> 2) * 128 + (g & 3) * 16; };
  gcd_t wp = (gcd_t)(wfull + fk * cpad + colb);                                                   // + (wrow(f) + 4 r) * cpad
  gcd_t tzg = (gcd_t)tzt;
  // the weights of row fragment f + 1 are requested as soon as those of f have been used; the tile's z-class phases go through LDS
  double wv[NFW > 0 ? NFW : 1][4];
  if constexpr (NFW > 0) {
#pragma unroll
    for (int g = 0; g < NFW; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) wv[g][r] = wp[(wrow(0) + (unsigned)(4 * r)) * cpad + 64 * g];
  }
  double *Tl = L;                                                                                 // [nzc][320]
  double *Lp = L + SK_HC_MAX * 320;                                                               // [nzc][32 RF rows][4 cg]
  for (unsigned e = t; e < (unsigned)nzc * 320; e += 512) {
    const unsigned zc = e / 320, cc = e - zc * 320;
    Tl[e] = tzg[zc * cpad + (unsigned)c.it.ct * 320 + cc];
  }
  const double *tl = Tl + 16 * cg + fr;                                                           // + zc * 320 + 64 g
  double *lp = Lp + (16 * RF * rh + fk) * 4 + cg;                                                 // + (zc * 32 RF + 16 f + 4 r) * 4
  __syncthreads();
#pragma unroll
  for (int f = 0; f < RF; ++f) {
    // (w G) in place: the accumulators are done with
    if constexpr (NFW > 0) {
#pragma unroll
      for (int g = 0; g < NFW; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[f][g][r] *= wv[g][r];
    }
    for (int zc = 0; zc < nzc; ++zc) {
      double s[4] = {0.0, 0.0, 0.0, 0.0};
      if constexpr (NFW > 0) {
#pragma unroll
        for (int g = 0; g < NFW; ++g) {
          const double tv = tl[zc * 320 + 64 * g];
#pragma unroll
          for (int r = 0; r < 4; ++r) s[r] += acc[f][g][r] * tv;
        }
      }
      const double tot = row16_sum4(s, fr);            // every lane of quad-group fr >> 2: the row sum of value r = fr >> 2
      if ((fr & 3) == 0) lp[(zc * 32 * RF + 16 * f + 4 * (fr >> 2)) * 4] = tot;
    }
    if constexpr (NFW > 0) {
      if (f < RF - 1) {
        // not earlier: the registers are the ones this row fragment's accumulators free.  (Requesting them right after the
        // multiplies above, to have the class loop in between, makes the allocator spill hundreds of registers; pulling the
        // wave's weights into L2 with throw-away loads during the segment's last chunk changed nothing: A/B 240.8 vs 240.6 us;
        // parking the weights of fragments 2 and 3 in the dead accumulators of 0 and 1: 450 spilled registers.)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int g = 0; g < NFW; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) wv[g][r] = wp[(wrow(f + 1) + (unsigned)(4 * r)) * cpad + 64 * g];
      }
    }
  }
  __syncthreads();
  for (unsigned o = t; o < (unsigned)nzc * 32 * RF; o += 512) {
    const double2 *src = reinterpret_cast<const double2 *>(Lp + 4 * o);
    const double2 v0 = src[0], v1 = src[1];
    hout[o] = (v0.x + v0.y) + (v1.x + v1.y);
  }
  __syncthreads();          // (the next segment's prologue writes panels into this memory)
}

// One segment = (tile, chunk range).  Between two barriers the workgroup multiplies chunk c (panel buffer c&1) and
// builds chunk c+1 (other buffer).  The two waves of a SIMD (w and w+4) do this in OPPOSITE order -- waves 0-3
// multiply first, waves 4-7 build first -- so one wave's operand generation overlaps its partner's MFMAs.
template <int RF, int NFW, bool late, int F0>
__device__ __forceinline__ void sk_body(const SkCtx &c, char *smem, double *outA, double *outB, double *outP) {
  // the fifth row fragment's planar vectors (64 + gs, gs < 16) are built by the threads gs < 16: the EARLY waves, which wait at the
  // barrier for the late ones anyway
  constexpr bool THIRD = RF == 5 && !late;
  d4 acc[RF][NFW > 0 ? NFW : 1];
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int g = 0; g < (NFW > 0 ? NFW : 1); ++g) acc[f][g] = (d4){0.0, 0.0, 0.0, 0.0};
  SkRaw raw;
#ifdef SK_STAMP
  unsigned long long st_a = 0, st_b = 0, s_pro = 0, s_load = 0, s_mfma = 0, s_build = 0, s_bar = 0, s_epi = 0;
#endif
  SK_STAMP_T(st_a);
  sk_load_raw<THIRD>(c, c.it.c0, raw);
#if SK_NOBAR
  if (threadIdx.x < 4) reinterpret_cast<unsigned *>(smem + SK_CNT)[threadIdx.x] = 0u;
#endif
  sk_build_panel<RF, THIRD>(c, raw, reinterpret_cast<double *>(smem));
  if ((late || SK_LOAD_AHEAD) && c.it.c0 + 1 < c.it.c1) sk_load_raw<THIRD>(c, c.it.c0 + 1, raw);
  __syncthreads();
  SK_STAMP_T(st_b); SK_STAMP_ADD(s_pro, st_a, st_b);
  unsigned buf = 0;                                    // byte offset of the panel being multiplied: 0 or SK_BUF1
  for (int ch = c.it.c0; ch < c.it.c1; ++ch, buf ^= SK_BUF1) {
    double *nxt = reinterpret_cast<double *>(smem + (buf ^ SK_BUF1));
    const bool more = ch + 1 < c.it.c1;
#if SK_NOBAR
    // chunk i of the segment lives in buffer i & 1.  Its panel is complete when built[i & 1] has reached 8 ((i + 1) >> 1) (chunk 0
    // was built in front of the segment's barrier); buffer (i + 1) & 1 may be overwritten with chunk i + 1 when done[(i + 1) & 1]
    // has reached 8 ((i + 1) >> 1): all eight waves have multiplied chunk i - 1
    const unsigned bi = buf ? 1u : 0u;
    const unsigned tgt = 8u * (unsigned)((ch - c.it.c0 + 1) >> 1);
#define SK_WAIT_PANEL() sk_cnt_wait(smem, bi, tgt)
#define SK_DONE_PANEL() sk_cnt_signal(smem, 2u + bi)
#define SK_WAIT_FREE() sk_cnt_wait(smem, 2u + (bi ^ 1u), tgt)
#define SK_BUILT_NEXT() sk_cnt_signal(smem, bi ^ 1u)
#else
#define SK_WAIT_PANEL() do { } while (0)
#define SK_DONE_PANEL() do { } while (0)
#define SK_WAIT_FREE() do { } while (0)
#define SK_BUILT_NEXT() do { } while (0)
#endif
    if (!late) {
      SK_STAMP_T(st_a);
#if SK_PRIO_E
      __builtin_amdgcn_s_setprio(SK_PRIO_E);
#endif
      // the phase's first operand reads go out BEFORE the next chunk's table loads are issued: the loads' address arithmetic and
      // issue (6-8 instructions per thread) then run in the shadow of the LDS round trip instead of in front of it.  Not in the
      // bodies that hold all 20 accumulator fragments: NFW + 1 more live registers there mean spills inside the chunk loop
      constexpr bool EPRE = SK_EARLY_PREFETCH && !SK_MFMA_LOOP && RF * NFW < 20;
      SkPre<NFW> pre;
      if constexpr (EPRE) { if (!(SK_DBG(c, 2))) sk_mfma_prefetch<NFW>(c, smem, buf, pre); }
      if (!SK_LOAD_AHEAD && more && !(SK_DBG(c, 4))) sk_load_raw<THIRD>(c, ch + 1, raw);
      SK_STAMP_T(st_b); SK_STAMP_ADD(s_load, st_a, st_b);
      SK_WAIT_PANEL();
      if constexpr (EPRE) { if (!(SK_DBG(c, 2))) sk_mfma_chunk_u<RF, NFW, F0>(c, smem, buf, acc, &pre); }
      else { if (!(SK_DBG(c, 2))) SK_MFMA_PHASE(RF, NFW, F0, c, smem, buf, acc); }
      SK_DONE_PANEL();
      SK_STAMP_T(st_a); SK_STAMP_ADD(s_mfma, st_b, st_a);
#if SK_PRIO_E == 1
      __builtin_amdgcn_s_setprio(0);            // (variant 1: the early waves' build at the base priority; 2 / 3: raised throughout)
#endif
      if (more) {
        SK_WAIT_FREE();
        if (!(SK_DBG(c, 1))) sk_build_panel<RF, THIRD>(c, raw, nxt);
        SK_BUILT_NEXT();
      }
      if (SK_LOAD_AHEAD && ch + 2 < c.it.c1 && !(SK_DBG(c, 4))) sk_load_raw<THIRD>(c, ch + 2, raw);
      SK_STAMP_T(st_b); SK_STAMP_ADD(s_build, st_a, st_b);
    } else {
#if SK_LATE_MODE == 1
      // full stagger: build first, multiply second
      SK_STAMP_T(st_a);
#if SK_PRIO
      __builtin_amdgcn_s_setprio(3);
#endif
#if SK_LATE_PREFETCH
      // (the phase's first operand reads go out ahead of the build's LDS writes: sk_mfma_prefetch)
      SkPre<NFW> pre;
      if (!(SK_DBG(c, 2))) sk_mfma_prefetch<NFW>(c, smem, buf, pre);
#endif
      if (more) {
        SK_WAIT_FREE();
        if (!(SK_DBG(c, 1))) sk_build_panel<RF, THIRD>(c, raw, nxt);
        SK_BUILT_NEXT();
      }
      SK_STAMP_T(st_b); SK_STAMP_ADD(s_build, st_a, st_b);
      if (ch + 2 < c.it.c1 && !(SK_DBG(c, 4))) sk_load_raw<THIRD>(c, ch + 2, raw);
      SK_STAMP_T(st_a); SK_STAMP_ADD(s_load, st_b, st_a);
#if SK_PRIO
      __builtin_amdgcn_s_setprio(SK_PRIO);
#endif
      SK_WAIT_PANEL();
#if SK_LATE_PREFETCH
      if (!(SK_DBG(c, 2))) sk_mfma_chunk_u<RF, NFW, F0>(c, smem, buf, acc, &pre);
#else
      if (!(SK_DBG(c, 2))) SK_MFMA_PHASE(RF, NFW, F0, c, smem, buf, acc);
#endif
      SK_DONE_PANEL();
      SK_STAMP_T(st_b); SK_STAMP_ADD(s_mfma, st_a, st_b);
#else
#error "only the full stagger (SK_LATE_MODE 1) is kept: half stagger measured 264 vs 260 us (DESIGN.md)"
#endif
    }
#if !SK_NOBAR
    __syncthreads();
#endif
    SK_STAMP_T(st_a); SK_STAMP_ADD(s_bar, st_b, st_a);
#undef SK_WAIT_PANEL
#undef SK_DONE_PANEL
#undef SK_WAIT_FREE
#undef SK_BUILT_NEXT
  }
#if SK_NOBAR
  __syncthreads();          // the epilogue re-uses the panels' memory: every wave has multiplied its last chunk
#endif
  // ---- partial tile out (only the active fragments), fragment-major: sk_part_off().  Two 16-byte stores per fragment, each
  //      wave-instruction one contiguous KB (the row-major layout took four 8-byte stores per fragment, each four 128-byte pieces:
  //      the store tail of a segment is issue-bound)
  if (SK_DBG(c, 16)) return;
  SK_STAMP_T(st_a);
  if (c.proj) {
    sk_project_out<RF, NFW>(c, smem, acc, outP);
  } else {
    // one lane-dependent offset, made opaque per segment: left to itself the compiler hoists all store addresses out of the
    // segment loop and spills them.  Row fragment f of the band is fragment (g0 + f) & 3 of row tile (g0 + f) >> 2: the partial
    // tile of the first row tile the band touches (outA) or of the next one (outB; null: the band's padding, nothing to store)
    unsigned lane_off = (unsigned)(((4 * c.rh * 20 + c.cg) << 8) + 2 * (16 * c.fk + c.fr));
    asm volatile("" : "+v"(lane_off));
    const unsigned g0 = (unsigned)c.it.g0;
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const unsigned g = g0 + (unsigned)f;
      double *ob = (g >> 2) == (g0 >> 2) ? outA : outB;
      if (!ob) continue;                                        // (uniform)
      double2 *o = reinterpret_cast<double2 *>(ob + lane_off) + (((g & 3) * 20) << 7);
#pragma unroll
      for (int gg = 0; gg < NFW; ++gg) {
        st_d2(o + ((4 * gg) << 7), make_double2(acc[f][gg][0], acc[f][gg][1]), SK_PART_NT);
        st_d2(o + ((4 * gg) << 7) + 64, make_double2(acc[f][gg][2], acc[f][gg][3]), SK_PART_NT);
      }
    }
  }
#ifdef SK_STAMP
  SK_STAMP_T(st_b); SK_STAMP_ADD(s_epi, st_a, st_b);
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) {
    unsigned long long *o = sk_stamp_buf + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8;
    o[0] += s_pro; o[1] += s_load; o[2] += s_mfma; o[3] += s_build; o[4] += s_bar; o[5] += s_epi;
    o[6] += (unsigned long long)(c.it.c1 - c.it.c0); o[7] = late ? 1 : 0;
  }
#endif
}


// ---- small systems: the phase tables of a segment's atoms, computed by the segment's own workgroup (SkFuse) --------------------
// The arithmetic is elyte_phase_kernel's, value for value (libm-grade sincos, the reference's angle-addition recurrence without
// FMA contraction, the same rotation sequence for the z seeds); the decomposition differs: one thread per (atom, x axis), (atom, y
// axis) and (atom, z axis, r) -- the eight z seeds r of an atom, which the stand-alone kernel keeps in flight in ONE thread, go to
// eight threads here (a segment has ~48 atoms for 512 threads).  Several bands run over the same atoms: each writes the same values
// to the same addresses.  The workgroup reads what IT wrote (one workgroup per CU, the CU's L1 was invalidated at the kernel's
// start, the barrier behind this makes the stores visible inside the workgroup): no dependency on any other workgroup.
// NOT inlined: a real call, once per segment.  Inlined, its live ranges (libm-grade sincos) are allocated together with the chunk
// loops' and the allocator spilled inside them (35 VGPRs, 31 SGPRs); as a function it has a register file of its own.
struct SkPlanDims { int kxmax, kymax, n_col_tiles; };
__device__ __attribute__((noinline)) void sk_phase_prologue(SkPlanDims pl, const SkFuse *fzp, int c0, int c1, int slab_slot, char *smem) {
#pragma clang fp contract(off)
  // (arguments of a real call arrive in vector registers: made wave-uniform again here, so that the block is read by scalar loads)
  {
    const unsigned long long a = (unsigned long long)fzp;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    fzp = (const SkFuse *)(((unsigned long long)hi << 32) | lo);
  }
  c0 = __builtin_amdgcn_readfirstlane(c0); c1 = __builtin_amdgcn_readfirstlane(c1);
  slab_slot = __builtin_amdgcn_readfirstlane(slab_slot);
  pl.kxmax = __builtin_amdgcn_readfirstlane(pl.kxmax); pl.kymax = __builtin_amdgcn_readfirstlane(pl.kymax);
  pl.n_col_tiles = __builtin_amdgcn_readfirstlane(pl.n_col_tiles);
  const __attribute__((address_space(4))) SkFuse &fz = *(const __attribute__((address_space(4))) SkFuse *)fzp;
  const int natoms = 16 * (c1 - c0);
  const int NA = (natoms + 63) & ~63;               // items are kind-major in runs of NA: a wavefront has one kind
  const int nrz = 1 + pl.n_col_tiles * 32;
  double *qzs = reinterpret_cast<double *>(smem);   // [natoms] q z of the segment's atoms (slab term), summed by thread 0 in order
  auto rot = [](double &cr, double &sr, double cw, double sw) {       // (cr, sr) *= (cw, sw), the reference's angle addition
    const double cn = cr * cw - sr * sw;
    const double sn = sr * cw + cr * sw;
    cr = cn; sr = sn;
  };
  for (int it = threadIdx.x; it < 10 * NA; it += 512) {
    const int kind = it / NA, al = it - kind * NA;   // kind 0: x axis, 1: y axis, 2 + r: z axis, seed r
    if (al >= natoms) continue;
    const int j = 16 * c0 + al;
    const int c = kind < 2 ? kind : 2;
    double xc = 0, qq = 0;
    if (j < fz.nl) {
      const int i = fz.elyte_idx[j];
      xc = fz.x[3 * i + c]; qq = fz.q[i];
    }
    const double ang = (c == 0 ? fz.ux : (c == 1 ? fz.uy : fz.uz)) * xc;
    double c1_, s1_;
    sincos(ang, &s1_, &c1_);
    if (c != 2) {
      const int nrow = c == 0 ? pl.kxmax + 1 : pl.kymax + 1;
      const int NR = c == 0 ? pl.kxmax + 2 : pl.kymax + 1;
      double2 *t = (c == 0 ? fz.Xt : fz.Yt) + ((size_t)(j >> 4) * NR) * 16 + (j & 15);
      const double sc = (c == 0) ? qq : 1.0;
      if (c == 0) t[(size_t)(pl.kxmax + 1) * 16] = make_double2(0.0, 0.0);
      t[0] = make_double2(sc, 0.0);
      double cm = c1_, sm = s1_;
      if (nrow > 1) t[16] = make_double2(sc * c1_, sc * s1_);
      for (int m = 2; m < nrow; ++m) {
        rot(cm, sm, c1_, s1_);
        t[(size_t)m * 16] = make_double2(sc * cm, sc * sm);
      }
    } else {
      const int r = kind - 2;
      double2 *t = fz.Zs + ((size_t)(j >> 4) * nrz) * 16 + (j & 15);
      // the chain 1, w, w^2, ... w^7 (this thread keeps w^r), w^8, w^40, w^kzt: the stand-alone kernel's sequence
      double bc = 1.0, bs = 0.0, mc = 1.0, ms = 0.0;
      for (int k = 1; k < 8; ++k) { rot(bc, bs, c1_, s1_); if (k == r) { mc = bc; ms = bs; } }
      double c8 = bc, s8 = bs;
      rot(c8, s8, c1_, s1_);
      double c40 = c8, s40 = s8;
      rot(c40, s40, c8, s8);
      rot(c40, s40, c40, s40);
      rot(c40, s40, c8, s8);
      double ct_c = 1.0, ct_s = 0.0;
      {
        double pc = c8, ps = s8;
        for (int e = fz.kzt >> 3; e; e >>= 1) {
          if (e & 1) rot(ct_c, ct_s, pc, ps);
          rot(pc, ps, pc, ps);
        }
      }
      if (r == 0) {
        t[0] = make_double2(c8, s8);
        fz.qc[j] = qq;
        qzs[al] = qq * xc;
      }
      for (int ct = 0; ct < pl.n_col_tiles; ++ct) {
        double sc_ = mc, ss_ = ms;
#pragma unroll
        for (int q40 = 0; q40 < 4; ++q40) {
          t[(size_t)(1 + 32 * ct + 8 * q40 + r) * 16] = make_double2(sc_, ss_);
          if (q40 < 3) rot(sc_, ss_, c40, s40);
        }
        rot(mc, ms, ct_c, ct_s);
      }
    }
  }
  __syncthreads();
  if (slab_slot >= 0 && threadIdx.x == 0) {
    double tot = 0.0;
    for (int a = 0; a < natoms; ++a) tot += qzs[a];
    fz.slab_part[slab_slot] = tot;
  }
  __syncthreads();          // (the panels are built into the memory the q z values sat in)
}

// Persistent-style launch: workgroup w runs the segments of row w of the work list (SkWItem; host: equal cost per workgroup,
// a segment boundary may fall inside a tile -- "stream-K" over the atom chunks).
// FUSE: the small-system variant (SkFuse: phase tables and pair sums inside this launch).  A variant of its own, so that the
// other keeps the register allocation it was tuned to: with the prologue in the same function the allocator reloads one spilled
// lane constant per chunk-loop iteration -- nothing for segments of <= 4 chunks, 15 % at the 16384 / 262144 size.
template <bool FUSE>
__global__ __launch_bounds__(512, 2) void sk_gemm_kernel(DevPlan pl, const SkWItem *__restrict__ witems, int maxseg, int nl_pad,
                                                         const double2 *__restrict__ Xt, const double2 *__restrict__ Yt,
                                                         const double2 *__restrict__ Zs, const double *__restrict__ qc,
                                                         double *__restrict__ part, const SkProj *__restrict__ proj, int dbg,
                                                         const SkFuse *__restrict__ fzp /*device copy, or null*/, int nwg_sk,
                                                         unsigned *__restrict__ ticket /*b_zc_fused_kernel's hand-off word, zeroed here; or null*/) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // panel buffer 0 at byte 0, buffer 1 at byte SK_BUF1
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  if (ticket && blockIdx.x == 0 && t == 0) *ticket = 0u;
  if (FUSE && (int)blockIdx.x >= nwg_sk) {
    // spare workgroups of a small system's launch: the real-space pair sums of the electrode rows, one wavefront per row
    const BRowArgs ra = fzp->rows;
    const int row = ((int)blockIdx.x - nwg_sk) * 8 + wave;
    if (row < ra.ne) {
      const double v = b_row_pairs(ra, row, t & 63);
      if ((t & 63) == 0) fzp->breal_out[row] = v;
    }
    return;
  }
  SkCtx c;
  c.dbg = dbg;
  c.nl_pad = nl_pad; c.nz = pl.nz;
  c.rh = wave & 1; c.cg = wave >> 1; c.wave = wave;
  c.Xt = Xt; c.Yt = Yt; c.Zs = Zs; c.qc = qc;
  c.proj = proj;
  c.nrx16 = (unsigned)(pl.kxmax + 2) * 16; c.nry16 = (unsigned)(pl.kymax + 1) * 16; c.nrz16 = (unsigned)(1 + pl.n_col_tiles * 32) * 16;
  const bool late = wave >= 4 && !(SK_DBG(c, 8));
#ifdef SK_STAMP
  unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  const SkWItem *wi = witems + (size_t)blockIdx.x * maxseg;
  SkWItem cur = wi[0];
  const int nseg = cur.nseg;
  for (int sgi = 0; sgi < nseg; ++sgi) {
    if (sgi) cur = wi[sgi];
    c.it = SkItem{cur.g0, cur.rf, cur.ct, cur.c0, cur.c1, cur.nbf};
    const int rf = __builtin_amdgcn_readfirstlane(cur.rf);      // row fragments per half of this band: 4 or 5
    // (small systems) the segment's phase tables first: before any per-lane constant of the chunk loop exists
    if (FUSE) sk_phase_prologue(SkPlanDims{pl.kxmax, pl.kymax, pl.n_col_tiles}, fzp, cur.c0, cur.c1, cur.slab_slot, smem);
    c.nfr = 4 * pl.n_row_tiles;
    // the lane's constants are formed per segment from an opaque copy of the thread index: formed once at the top of the kernel
    // they would be live across the epilogue, which has no register to spare for them (they were spilled around it)
    {
      unsigned tt = sk_tid(wave);
      asm volatile("" : "+v"(tt));
      c.gj = tt & 15; c.gs = tt >> 4;
      c.fr = tt & 15; c.fk = (tt >> 4) & 3;
      c.wa = (unsigned)(c.gs * SK_LD + (c.gj ^ (c.gs & 15)));
      c.za = (unsigned)((32 * rf + 16 * 5 * (c.gs >> 3) + (c.gs & 7)) * SK_LD + (c.gj ^ (c.gs & 7)));
      c.dsin = 8 * SK_LD + ((c.gj & 8) ? -8 : 8);
      c.base_a = ((unsigned)(16 * rf * c.rh + c.fr) * SK_LD + (unsigned)((c.fk ^ c.fr) & 3)) * 8u;
      c.base_b = ((unsigned)(32 * rf + 16 * c.cg + c.fr) * SK_LD + (unsigned)((c.fk ^ c.fr) & 3)) * 8u;
      c.pq = (unsigned)(c.fr >> 2) << 5;
    }
    // per row fragment f (16 planar vectors) only the leading nff_f column fragments (8 kz each) are inside the cut-off sphere;
    // the band's count is the largest of them (an odd count leaves the last fragment of the last kz block unwritten:
    // the partial buffer is zeroed when allocated, those columns carry no listed k and zero weight)
    int nfrag = 0;
#pragma unroll
    for (int f = 0; f < 5; ++f) { const int nff = f < rf ? (int)((c.it.nbf >> (8 * f)) & 255u) : 0; nfrag = nff > nfrag ? nff : nfrag; }
    const int nfw = (nfrag - c.cg + 3) >> 2;            // fragments of this wave: fi = 4 g + cg < nfrag   (wave-uniform)
    // f0 = the first row fragment whose cut leaves out this wave's last column fragment (rf: none)
    int f0 = rf;
#pragma unroll
    for (int f = 4; f >= 0; --f) {
      if (f >= rf) continue;
      const int nff = (int)((c.it.nbf >> (8 * f)) & 255u);
      const int n = (nff - c.cg + 3) >> 2;
      if (n < nfw) f0 = f;
    }
    // (never-increasing counts are the host's ordering; if a later fragment had MORE columns than an earlier one, f0 would
    //  point at the earlier one and the later would lose its last column: guard by taking the cut of the LAST offender only when
    //  all following fragments are short too -- which the loop above does: f0 is the smallest f with a short cut, so any longer
    //  cut behind it would be wrong; fall back to no culling then)
#pragma unroll
    for (int f = 0; f < 5; ++f) {
      if (f >= rf) continue;
      const int nff = (int)((c.it.nbf >> (8 * f)) & 255u);
      if (f >= f0 && ((nff - c.cg + 3) >> 2) >= nfw) f0 = rf;
    }
    c.f0 = __builtin_amdgcn_readfirstlane(f0);
    // planar vectors of this thread: 16 g0 + gs, + 32, and -- fifth row fragment, threads gs < 16 -- + 64 (the host pads the
    // vector tables past the plan's end with vectors that read the all-zero X row)
    const int p0 = c.it.g0 * 16 + c.gs, p1 = p0 + 32, p2 = p0 + 64;
    // (a table has at most kmax + 2 < 4096 rows of 16: the offsets fit in 16 bits)
    c.xy0 = ((unsigned)pl.p_ikx[p0] * 16 + c.gj) | (((unsigned)pl.p_iky[p0] * 16 + c.gj) << 16);
    c.xy1 = ((unsigned)pl.p_ikx[p1] * 16 + c.gj) | (((unsigned)pl.p_iky[p1] * 16 + c.gj) << 16);
    c.neg0 = pl.p_sgn[p0] < 0; c.neg1 = pl.p_sgn[p1] < 0;             // (padding vectors read the all-zero X row)
    c.xy2 = 0; c.neg2 = false;
    if (rf == 5 && c.gs < 16) {
      c.xy2 = ((unsigned)pl.p_ikx[p2] * 16 + c.gj) | (((unsigned)pl.p_iky[p2] * 16 + c.gj) << 16);
      c.neg2 = pl.p_sgn[p2] < 0;
    }
    c.zact = 40 * (c.gs >> 3) + (c.gs & 7) < 8 * nfrag;     // the thread's first kz lies in an active column fragment
    c.zoff = (unsigned)(1 + c.it.ct * 32 + c.gs) * 16 + c.gj;
    // output: projecting -- ONE band-local piece per segment (slot = the segment's index cur.sg); else partial tiles, one per row
    // tile of the plan the band touches (cur.sga: row tile g0 >> 2; cur.sgb: the next one, -1 when the band ends inside the first
    // or its tail is padding)
    double *outA = nullptr, *outB = nullptr, *outP = nullptr;
    if (proj) outP = part + (size_t)cur.sg * (SK_HC * SK_HC_ROWS);            // projecting: the segment's band-local piece
    else {
      outA = part + (size_t)cur.sga * (128 * 320);
      if (cur.sgb >= 0) outB = part + (size_t)cur.sgb * (128 * 320);
    }
#ifdef SK_STAMP
    const unsigned long long sg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    // one body per (row fragments of the band, column fragments of this wave, stagger role, sphere cut): the MFMA phase is
    // straight-line code.  f0 = rf (no culling) is always correct -- culled fragments only hold G entries that no listed k reads --
    // and serves as the default.  RF = 4 bands take up to 5 column fragments per wave, RF = 5 bands up to 4 (20 accumulator
    // fragments either way).
#define SK_BODY_F0_4(N, L)                                            \
  switch (c.f0) {                                                     \
    case 1: sk_body<4, N, L, 1>(c, smem, outA, outB, outP); break;          \
    case 2: sk_body<4, N, L, 2>(c, smem, outA, outB, outP); break;          \
    case 3: sk_body<4, N, L, 3>(c, smem, outA, outB, outP); break;          \
    default: sk_body<4, N, L, 4>(c, smem, outA, outB, outP); break;         \
  }
#define SK_BODY_F0_5(N, L)                                            \
  switch (c.f0) {                                                     \
    case 1: sk_body<5, N, L, 1>(c, smem, outA, outB, outP); break;          \
    case 2: sk_body<5, N, L, 2>(c, smem, outA, outB, outP); break;          \
    case 3: sk_body<5, N, L, 3>(c, smem, outA, outB, outP); break;          \
    case 4: sk_body<5, N, L, 4>(c, smem, outA, outB, outP); break;          \
    default: sk_body<5, N, L, 5>(c, smem, outA, outB, outP); break;         \
  }
#define SK_BODY_NFW(L)                                                \
  if (rf == 5) {                                                      \
    switch (nfw) {                                                    \
      case 4: SK_BODY_F0_5(4, L) break;                               \
      case 3: SK_BODY_F0_5(3, L) break;                               \
      case 2: SK_BODY_F0_5(2, L) break;                               \
      case 1: SK_BODY_F0_5(1, L) break;                               \
      default: sk_body<5, 0, L, 5>(c, smem, outA, outB, outP); break;       \
    }                                                                 \
  } else {                                                            \
    switch (nfw) {                                                    \
      case 5: SK_BODY_F0_4(5, L) break;                               \
      case 4: SK_BODY_F0_4(4, L) break;                               \
      case 3: SK_BODY_F0_4(3, L) break;                               \
      case 2: SK_BODY_F0_4(2, L) break;                               \
      case 1: SK_BODY_F0_4(1, L) break;                               \
      default: sk_body<4, 0, L, 4>(c, smem, outA, outB, outP); break;       \
    }                                                                 \
  }
    if (late) { SK_BODY_NFW(true) } else { SK_BODY_NFW(false) }
#undef SK_BODY_NFW
#undef SK_BODY_F0_4
#undef SK_BODY_F0_5
#ifdef SK_STAMP
    if (t == 0 && sg < 4096) {
      unsigned long long *o = sk_seg_buf + (size_t)sg * 4;
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));      // which XCD this workgroup landed on
      o[0] = ((unsigned long long)blockIdx.x << 32) | ((xcc & 15u) << 16) | (unsigned)c.it.g0; o[1] = c.it.nbf; o[2] = (unsigned long long)(unsigned)(c.it.c1 - c.it.c0) | ((unsigned long long)c.it.rf << 32);
      o[3] = __builtin_amdgcn_s_memrealtime() - sg_t0;
    }
#endif
  }
#ifdef SK_STAMP
  if (t == 0 && blockIdx.x < 1024) {
    unsigned long long *o = sk_clock_buf + (size_t)blockIdx.x * 4;
    o[0] = ck0; o[1] = __builtin_amdgcn_s_memtime(); o[2] = rt0; o[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

#ifdef SK_STAMP
extern "C" int conp_debug_sk_clock(unsigned long long *out /*[1024*4]*/) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sk_clock_buf), sizeof(unsigned long long) * 1024 * 4) == hipSuccess ? 0 : -1;
}
extern "C" int conp_debug_sk_segs(unsigned long long *out /*[4096*4]*/) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sk_seg_buf), sizeof(unsigned long long) * 4096 * 4) == hipSuccess ? 0 : -1;
}
extern "C" int conp_debug_sk_stamps(unsigned long long *out /*[1024*8*8]*/, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(sk_stamp_buf), sizeof(unsigned long long) * 1024 * 8 * 8) != hipSuccess) return -1;
  if (reset) {
    static unsigned long long zeros[1024 * 8 * 8];
    if (hipMemcpyToSymbol(HIP_SYMBOL(sk_stamp_buf), zeros, sizeof(zeros)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

int sk_hc_stride() { return SK_HC * SK_HC_ROWS; }
int sk_hc_max_classes() { return SK_HC_MAX; }

// proj == nullptr: partial tiles [segment][128 x 320] into `part`; otherwise (planar electrodes, at most sk_hc_max_classes() z
// classes; proj = device copy of the parameter block) the segments' projected pieces [segment][sk_hc_stride()]
void launch_sk_gemm(hipStream_t s, const DevPlan &pl, const SkWItem *witems, int maxseg, int nwg, int nl_pad,
                    const double2 *Xt, const double2 *Yt, const double2 *Zs, const double *qc, double *part, const SkProj *proj,
                    const SkFuse *fuse, int fuse_rows, unsigned *ticket) {
  if (nwg <= 0) return;
  // fuse: DEVICE copy of the parameter block (conp_fix.cpp uploads it when its content changes); fuse_rows: its rows.ne
  const int extra = fuse ? (fuse_rows + 7) / 8 : 0;
  const size_t lds = SK_LDS_BYTES;
  static DynLdsCache granted{}, granted_f{};
  if (fuse) ensure_dyn_lds(sk_gemm_kernel<true>, lds, granted_f);
  else ensure_dyn_lds(sk_gemm_kernel<false>, lds, granted);
#ifdef SK_ABLATE
  static const int dbg = getenv("CONP_SK_DBG") ? atoi(getenv("CONP_SK_DBG")) : 0;   // diagnostic build only
#else
  const int dbg = 0;
#endif
  if (fuse)
    hipLaunchKernelGGL(sk_gemm_kernel<true>, dim3(nwg + extra), dim3(512), lds, s, pl, witems, maxseg, nl_pad, Xt, Yt, Zs, qc, part, proj,
                       dbg, fuse, nwg, ticket);
  else
    hipLaunchKernelGGL(sk_gemm_kernel<false>, dim3(nwg), dim3(512), lds, s, pl, witems, maxseg, nl_pad, Xt, Yt, Zs, qc, part, proj, dbg,
                       fuse, nwg, ticket);
}

// G = sum over a tile's splits (fixed order).  Gwf = w * G in MFMA-fragment-major order for b_project:
//   Gwf[((rf * (C_pad/4)) + ts) * 64 + fk * 16 + fr]  = (w G)[16 rf + fr][4 ts + fk]
// One block per (tile, 16-row fragment, 80-column quarter): its Gwf output is one contiguous run of 1280 doubles,
// transposed through LDS so that partial reads, G writes and Gwf writes are all coalesced.
// A thread walks its splits serially (8 loads in flight), so heavily split tiles (multi-GPU shards: one tile cut 256 ways)
// are summed in two levels:  level 1 (blockIdx.y = group g) adds splits [16 g, 16 g + 16) into slot 16 g in place,
// level 2 adds the group slots (stride 16) and writes G / Gwf.  level 0 = everything in one pass.
// Sum of one element's partials over `count` splits `step` doubles apart.  8 (16) loads in flight; the association
// ((s0+s1)+(s2+s3))+((s4+s5)+(s6+s7)) is fixed -> bitwise reproducible.  part_sum2: the same for both halves of a 16-byte unit
// (rows r and r + 2 of a fragment: the two elements a thread of the 80-column slices owns), each with exactly that order.
__device__ __forceinline__ double part_sum1(const double *__restrict__ src, size_t step, int count) {
  double s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int sp = 0;
  for (; sp + 16 <= count; sp += 16) {       // 16 partials in flight per element; additions in the order of the 8-wide loop
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(sp + u) * step];
#pragma unroll
    for (int u = 0; u < 8; ++u) s8[u] += v[u];
#pragma unroll
    for (int u = 0; u < 8; ++u) s8[u] += v[8 + u];
  }
  for (; sp + 8 <= count; sp += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s8[u] += src[(size_t)(sp + u) * step];
  }
  for (int u = 0; sp < count; ++sp, ++u) s8[u] += src[(size_t)sp * step];
  return ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
}
__device__ __forceinline__ double2 part_sum2(const double *__restrict__ src, size_t step, int count) {
  double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int sp = 0;
  for (; sp + 16 <= count; sp += 16) {
    double2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = ld_d2(reinterpret_cast<const double2 *>(src + (size_t)(sp + u) * step), SK_PART_NT);
#pragma unroll
    for (int u = 0; u < 8; ++u) { a8[u] += v[u].x; b8[u] += v[u].y; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a8[u] += v[8 + u].x; b8[u] += v[8 + u].y; }
  }
  for (; sp + 8 <= count; sp += 8) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld_d2(reinterpret_cast<const double2 *>(src + (size_t)(sp + u) * step), SK_PART_NT);
#pragma unroll
    for (int u = 0; u < 8; ++u) { a8[u] += v[u].x; b8[u] += v[u].y; }
  }
  for (int u = 0; sp < count; ++sp, ++u) {
    const double2 v = ld_d2(reinterpret_cast<const double2 *>(src + (size_t)sp * step), SK_PART_NT);
    a8[u] += v.x; b8[u] += v.y;
  }
  return make_double2(((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7])),
                      ((b8[0] + b8[1]) + (b8[2] + b8[3])) + ((b8[4] + b8[5]) + (b8[6] + b8[7])));
}

// first row of a reducing thread: with one element per thread the 16 rows in order; with two (a 16-byte unit = rows 4 r + fk and
// 4 (r + 1) + fk, r even) the unit index u = 0..7 -> rows {0..3, 8..11}, the partner 4 rows below
template <int K> __device__ __forceinline__ int skr_row0(int u) { return K == 1 ? u : (u & 3) + 8 * (u >> 2); }
constexpr int SKR_GROUP = 16;
// a tile's 320 columns are cut into SKR_SL slices: one block per (tile, 16-row fragment, slice).  Eight slices of 40 columns
// (round 1 and most of round 2: four of 80, two elements per thread): twice the blocks pulling on the partial tiles -- the
// reduction is a pure stream, and 32 blocks per tile left most CUs idle for small systems and for one rank's one or two tiles
// SL = 4 slices of 80 columns (two elements per thread) when the plan has enough tiles to fill the chip that way (headline: 8
// tiles x 32 blocks; eight slices cost 4.7 us there: twice the Hc slots for the dot kernel to stage), 8 slices of 40 otherwise
constexpr int SKR_T = 640;                         // threads per block
// two levels (one more launch) once the most-split tile has more than this many partials: measured break-even on the headline
// box -- 40 partials (one GPU) 4 us faster in one level, 57 (two ranks) equal, 113 (four ranks) 12 us faster in two
static int skr_two_level_from() { static const int v = diag_switch("CONP_SKR_TWO") ? atoi(diag_switch("CONP_SKR_TWO")) : 4 * SKR_GROUP; return v; }
template <int SKR_SL>
__global__ __launch_bounds__(SKR_T) void sk_reduce_kernel(int C_pad, const SkTile *__restrict__ tiles,
                                                        double *__restrict__ part, const double *__restrict__ wfull,
                                                        double *__restrict__ G, double *__restrict__ Gwf, int level) {
  constexpr int SKR_W = 320 / SKR_SL, SKR_K = 16 * SKR_W / SKR_T;      // slice width; elements per thread (2 or 1)
  __shared__ double tr[16 * SKR_W];
  const SkTile tl = tiles[blockIdx.x / (8 * SKR_SL)];
  const int f16 = (blockIdx.x / SKR_SL) & 7;   // 16-row fragment inside the 128-row tile
  const int q = blockIdx.x % SKR_SL;           // column slice of the 320-column tile
  const size_t plane = 128 * 320;
  int first = 0, count = tl.nsplit, stride = 1;
  if (level == 1) {
    first = SKR_GROUP * blockIdx.y;
    if (first >= tl.nsplit) return;
    count = tl.nsplit - first < SKR_GROUP ? tl.nsplit - first : SKR_GROUP;
  } else if (level == 2) {
    stride = SKR_GROUP;
    count = (tl.nsplit + SKR_GROUP - 1) / SKR_GROUP;
  }
  const size_t step = (size_t)stride * plane;
  // this thread's element(s): (row0, cl) and, in the 80-column slices, (row0 + 4, cl) -- the other half of the same 16-byte unit
  const int cl = threadIdx.x % SKR_W;
  const int row0 = skr_row0<SKR_K>(threadIdx.x / SKR_W);
  const int col = SKR_W * q + cl;
  static_assert(SKR_K == 1 || SKR_T / SKR_W == 8, "the 80-column slices: 8 row pairs per block");
  double sums[SKR_K];
#pragma unroll
  for (int k = 0; k < SKR_K; ++k) sums[k] = 0.0;
  double *src = part + (size_t)(tl.item0 + first) * plane + sk_part_off(16 * f16 + row0, col);
  if (col < 32 * tl.nba) {
    if constexpr (SKR_K == 2) { const double2 v = part_sum2(src, step, count); sums[0] = v.x; sums[1] = v.y; }
    else sums[0] = part_sum1(src, step, count);
    if (level == 1) {                            // this thread is the only reader and writer of the unit
      if constexpr (SKR_K == 2) *reinterpret_cast<double2 *>(src) = make_double2(sums[0], sums[1]);
      else src[0] = sums[0];
    }
  }
  if (level == 1) return;
#pragma unroll
  for (int k = 0; k < SKR_K; ++k) {
    const int row = row0 + 4 * k, rowl = 16 * f16 + row;
    const size_t grow = (size_t)tl.rt * 128 + rowl, gcol = (size_t)tl.ct * 320 + col;
    G[grow * C_pad + gcol] = sums[k];
    tr[(cl >> 2) * 64 + (cl & 3) * 16 + row] = wfull[grow * C_pad + gcol] * sums[k];
  }
  __syncthreads();
  const size_t rf = (size_t)tl.rt * 8 + f16;
  double *dst = Gwf + (rf * (C_pad / 4) + (size_t)tl.ct * 80 + (SKR_W / 4) * q) * 64;
#pragma unroll
  for (int k = 0; k < SKR_K; ++k) dst[threadIdx.x + SKR_T * k] = tr[threadIdx.x + SKR_T * k];
}

// slices per tile: enough blocks to fill the chip
static int skr_slices(int ntiles) { return ntiles * 32 >= 200 ? 4 : 8; }

template <int SL>
static void launch_sk_reduce_sl(hipStream_t s, const DevPlan &pl, const SkTile *tiles, int ntiles, int max_nsplit, double *part, double *G,
                                double *Gwf) {
  if (max_nsplit > skr_two_level_from()) {
    const int ngroups = (max_nsplit + SKR_GROUP - 1) / SKR_GROUP;
    hipLaunchKernelGGL(sk_reduce_kernel<SL>, dim3(ntiles * 8 * SL, ngroups), dim3(SKR_T), 0, s, pl.C_pad, tiles, part, pl.wfull, G, Gwf, 1);
    hipLaunchKernelGGL(sk_reduce_kernel<SL>, dim3(ntiles * 8 * SL), dim3(SKR_T), 0, s, pl.C_pad, tiles, part, pl.wfull, G, Gwf, 2);
  } else {
    hipLaunchKernelGGL(sk_reduce_kernel<SL>, dim3(ntiles * 8 * SL), dim3(SKR_T), 0, s, pl.C_pad, tiles, part, pl.wfull, G, Gwf, 0);
  }
}
void launch_sk_reduce(hipStream_t s, const DevPlan &pl, const SkTile *tiles, int ntiles, int max_nsplit, double *part, double *G,
                      double *Gwf) {
  if (ntiles <= 0) return;
  if (skr_slices(ntiles) == 4) launch_sk_reduce_sl<4>(s, pl, tiles, ntiles, max_nsplit, part, G, Gwf);
  else launch_sk_reduce_sl<8>(s, pl, tiles, ntiles, max_nsplit, part, G, Gwf);
}

// structure factors in the reference's k order (parity read-back; km_ewald.cpp sfacrl_all / sfacim_all)
__global__ void sfac_gather_kernel(int kcount, int C_pad, int PT, const int *__restrict__ row_a,
                                   const int *__restrict__ col_c, const int *__restrict__ k_sign,
                                   const double *__restrict__ G, double *__restrict__ sr, double *__restrict__ si) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= kcount) return;
  const size_t ra = (size_t)row_a[k] * C_pad, rb = (size_t)(row_a[k] + PT) * C_pad;
  const int cc = col_c[k], cs = col_c[k] + 8;
  const double CC = G[ra + cc], CS = G[ra + cs], SC = G[rb + cc], SS = G[rb + cs];
  const double sg = (double)k_sign[k];
  // (p,+m): Sr = CC - SS, Si = CS + SC ; (p,-m): Sr = CC + SS, Si = SC - CS   (km_ewald.cpp:768-773)
  sr[k] = CC - sg * SS;
  si[k] = SC + sg * CS;
}

void launch_sfac_gather(hipStream_t s, int kcount, int C_pad, int PT, const int *sf_row_a, const int *sf_col_c,
                        const int *k_sign, const double *G, double *sfacrl, double *sfacim) {
  hipLaunchKernelGGL(sfac_gather_kernel, dim3((kcount + 255) / 256), dim3(256), 0, s, kcount, C_pad, PT, sf_row_a,
                     sf_col_c, k_sign, G, sfacrl, sfacim);
}

// ================================================================================================
// 3. k-space b vector (km_ewald.cpp:789-825):  b_i = - sum_{r,t} Rp[r][i] * (w G)[r][t] * Tz[t][i]
//    Workgroup = 16 waves, 64 electrode atoms (4 column fragments) x the units of one of four `parts`
//    (the parts' sums are added by b_real_combine as (k0 + k1) + (k2 + k3): a fixed order).  H = (w G)(16-row fragment) x Tz(slice in LDS) on MFMA, Hadamard
//    with Rp and column sum in the epilogue.  (w G) arrives fragment-major: every A operand is one contiguous
//    512-byte load.  Only the leading 8*nba k-steps of a row tile carry weight.
// ================================================================================================
// Round 4: 64 electrode atoms per workgroup (four column fragments: every A operand load feeds four MFMAs -- the 32-atom form re-read
// (w G) from L2 once per 32 atoms, 333 MB per launch at the headline size) and four PARTS instead of two halves (the units of a
// column tile -- (row tile, row fragment) pairs -- are dealt to the parts one by one: 256 workgroups at Ne = 4096; all four slots of
// bk_part are written).  The Tz slice of 64 atoms is staged in two halves of 160 columns (80 KB), as four [160][16] blocks -- the
// same conflict-free fragment reads as before; a unit's product with each half is multiplied with Rp and added on its own (the
// projection is linear in H).
__global__ __launch_bounds__(1024) void b_project_kernel(int C_pad, int ne_pad, int n_col_tiles, const int *__restrict__ ct_ptr,
                                                         const SkTile *__restrict__ tiles, const double *__restrict__ Gwf,
                                                         const double *__restrict__ Rp, const double *__restrict__ Tz,
                                                         double *__restrict__ bk_part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double *tz = reinterpret_cast<double *>(smem);          // [4 atom fragments][160 columns][16 atoms]
  double *red = tz + 4 * 160 * 16;                         // [16 waves][64]
  const int part = blockIdx.y;
  const int i0 = blockIdx.x * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  double psum[4] = {0.0, 0.0, 0.0, 0.0};
  for (int ct = 0; ct < n_col_tiles; ++ct) {
    const int tb = ct_ptr[ct], te = ct_ptr[ct + 1];
    if (te <= tb) continue;
    int nba_max = 0;
    for (int k = tb; k < te; ++k) nba_max = tiles[k].nba > nba_max ? tiles[k].nba : nba_max;      // (uniform: a handful of tiles)
    for (int h = 0; h < 2; ++h) {                        // columns [160 h, 160 h + 160) of the tile = k-steps [40 h, 40 h + 40)
      if (40 * h >= 8 * nba_max) break;
      __syncthreads();
      for (int e = t; e < 160 * 64; e += 1024) {
        const int col = e >> 6, a = e & 63;
        tz[((a >> 4) * 160 + col) * 16 + (a & 15)] = Tz[(size_t)(ct * 320 + 160 * h + col) * ne_pad + i0 + a];
      }
      __syncthreads();
      // units of this part: u_all = part, part + 4, ... over the 8 (te - tb) (tile, row fragment) pairs; dealt to the 16 waves
      // (18 units for 16 waves at the headline plan: two rounds, the second for two waves.  Measured, round 4: units cut into two
      //  or four runs of k-steps so that the rounds are shorter -- halves 46.0-47.3 against 47.4-47.6 us, within the spread of one
      //  box; quarters 52.2 us: every item pays the epilogue's sixteen Rp loads.  Also measured: a host-made schedule that gives
      //  every SIMD of every part about the same number of k-steps (longest unit first; a table of unit indices per wave) --
      //  50.8-53.0 against 47.6-50.1 us.  The balance is not what this kernel waits for.)
      const int nu = 8 * (te - tb);
      for (int u = part + 4 * wave; u < nu; u += 64) {
        const SkTile tl = tiles[tb + (u >> 3)];
        const int rf = tl.rt * 8 + (u & 7);
        const int ks0 = 40 * h, ks1 = 8 * tl.nba < 40 * (h + 1) ? 8 * tl.nba : 40 * (h + 1);
        if (ks1 <= ks0) continue;
        d4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = (d4){0.0, 0.0, 0.0, 0.0};
        const double *ap = Gwf + ((size_t)rf * (C_pad / 4) + (size_t)ct * 80) * 64 + lane;
        const double *bp = tz + fk * 16 + fr;
        // the A operands (one double per lane and k-step, straight from L2) are requested a group of eight k-steps ahead of the
        // MFMAs that use them: the counters say the waves of this kernel wait for instructions' operands 44 % of their cycles with
        // the matrix pipe 60 % busy, and the LDS is not what they wait for (SQ_WAIT_INST_LDS 0.2 %)
        // (k-step counts are multiples of eight: 8 nba, phases of 40)
        double an[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) an[i] = ap[(size_t)(ks0 + i) * 64];
#pragma unroll 1
        for (int tg = ks0; tg < ks1; tg += 8) {
          double ac[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) ac[i] = an[i];
          if (tg + 8 < ks1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) an[i] = ap[(size_t)(tg + 8 + i) * 64];
          }
          const double *bq = bp + 64 * (tg - ks0);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = MFMA_F64(ac[i], bq[64 * i + c * 160 * 16], acc[c]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double *rp = Rp + (size_t)(16 * rf + fk + 4 * r) * ne_pad + i0 + fr;
#pragma unroll
          for (int c = 0; c < 4; ++c) psum[c] += rp[16 * c] * acc[c][r];
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) { psum[c] += __shfl_xor(psum[c], 16, 64); psum[c] += __shfl_xor(psum[c], 32, 64); }
  __syncthreads();
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wave * 64 + 16 * c + lane] = psum[c];
  }
  __syncthreads();
  if (t < 64) {
    double sum = 0.0;
    for (int w = 0; w < 16; ++w) sum += red[w * 64 + t];
    bk_part[(size_t)part * ne_pad + i0 + t] = -sum;
  }
}

void launch_b_project(hipStream_t s, const DevPlan &pl, int ne_pad, const int *ct_ptr, const SkTile *tiles, const double *Gwf,
                      const double *Rp, const double *Tz, double *bk_part) {
  const size_t lds = ((size_t)4 * 160 * 16 + 16 * 64) * sizeof(double);
  static DynLdsCache granted{};
  ensure_dyn_lds(b_project_kernel, lds, granted);
  hipLaunchKernelGGL(b_project_kernel, dim3(ne_pad / 64, 4), dim3(1024), lds, s, pl.C_pad, ne_pad, pl.n_col_tiles, ct_ptr, tiles,
                     Gwf, Rp, Tz, bk_part);
}

// ---- planar-electrode fast path ------------------------------------------------------------------------------------
// Electrode atoms that share a z coordinate (graphene sheets: every atom of a layer) share their z-phase column Tz[:, i].
// With at most 64 distinct z values ("z classes") the projection factorises:
//     Hc[r][zc] = sum_t (w G)[r][t] * Tzc[t][zc]            R_pad x 64   (MFMA, 1/64 of the general work)
//     b_i      = - sum_r Rp[r][i] * Hc[r][zclass(i)]        one 2*n_p-term dot product per atom
// Same arithmetic per term as the general kernel; only the grouping of equal columns changes.
// grid = (R_pad/16 row fragments, 4 k-quarters); one wave per (fragment, quarter, 16-class group)
__global__ __launch_bounds__(256) void b_hc_kernel(int C_pad, int n_col_tiles, const int *__restrict__ rt_mine, int nzc16,
                                                   const int *__restrict__ nb_act, const double *__restrict__ Gwf,
                                                   const double *__restrict__ Tzc /*[C_pad][64]*/,
                                                   double *__restrict__ Hc4 /*[8 slots, 4 written here][64][R_pad]*/, int R_pad) {
  const int rf = blockIdx.x, rt = rf >> 3, kq = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  if (wave >= nzc16) return;
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  if (rt_mine[rt]) {
    for (int ct = 0; ct < n_col_tiles; ++ct) {
      const int nba = nb_act[ct * (R_pad >> 7) + rt];
      const int nks = 8 * nba, t0 = nks * kq / 4, t1 = nks * (kq + 1) / 4;
      const double *ap = Gwf + ((size_t)rf * (C_pad / 4) + (size_t)ct * 80) * 64 + lane;
      const double *bp = Tzc + (size_t)(ct * 320 + fk) * 64 + 16 * wave + fr;
#pragma unroll 8
      for (int ts = t0; ts < t1; ++ts) acc = MFMA_F64(ap[(size_t)ts * 64], bp[(size_t)ts * 256], acc);
    }
  }
  double *out = Hc4 + (size_t)kq * R_pad * 64 + (size_t)(16 * wave + fr) * R_pad;      // class-major: see b_zc_dot_kernel
#pragma unroll
  for (int r = 0; r < 4; ++r) out[16 * rf + fk + 4 * r] = acc[r];
}

// sum of one Hc element over the segments slot_idx[s0 .. s1) that worked on its row tile (sk_project_out's pieces, `stride` doubles
// apart), eight loads in flight, fixed association: bitwise reproducible
// (Measured, round 4: sixteen in flight for b_zc_final_kernel on the decks -- 32 pieces per row tile, two round trips instead of
//  four -- is SLOWER: 11.0 -> 11.4 us for hc + dot on il_onelayer, 13.4 -> 14.5 on il_twolayer; the kernel holds its first eight
//  phases in flight meanwhile and 128 registers do not take both, 7 values spill.)
__device__ __forceinline__ double hc_slot_sum(const double *__restrict__ h, const int *__restrict__ slot_idx, int s0, int s1, size_t stride) {
  double acc = 0.0;
  int s = s0;
  for (; s + 8 <= s1; s += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = h[(size_t)slot_idx[s + u] * stride];
    acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  double v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = s + u < s1 ? h[(size_t)slot_idx[s + u] * stride] : 0.0;
  return acc + (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
}

// the segments' Hc pieces of every owned row tile added into slot 0 of Hc4 (class-major, [64][R_pad]) for the consumers that read
// that table (b_zc_dot_kernel: row quarters of ALL tiles per block); one block per owned row tile.
// (Built, measured and removed: the same sum by the LAST segment to finish on a row tile, inside sk_gemm -- write-through stores of
//  the pieces, vmcnt(0), barrier, one agent-scope ticket add per segment, sc1 loads by the last arriver.  This launch (5.3 us + a
//  kernel boundary) went away and sk_gemm grew by 5.7-6.4 us: every workgroup pays the atomic's round trip behind its epilogue and
//  eight late ones add 39 pieces each on the kernel's critical path.  0.2884 vs 0.2890 ms per update: nothing.  The same for
//  sym_finish inside sym_gemv: 25.6 us instead of 23.1.)
// Eight threads per element (round 4): thread u of an element's octet loads the pieces 8 g + u of every group g of eight at once
// (39 pieces at the headline size: five loads in flight, ONE round trip -- the one-thread form walked five dependent rounds), the
// octet's lanes add a group by a butterfly -- ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)), hc_slot_sum's tree, additions
// commute -- and the groups are added in list order: the same bits as hc_slot_sum.
// One row fragment g of the plan (16 planar vectors: rows 16 (g & 3) + i and 64 + the same of row tile g >> 2) per blockIdx.y.  In
// every column tile the fragment belongs to one band of sk_gemm's schedule -- not the same band from tile to tile, the cut differs --
// and every segment of that band left a band-local piece [class][32 rf rows]: the host lists them per fragment (frag_ptr / ents: the
// offset of the fragment's first 'a' row in the piece, and the band's rf), column tile after column tile, segment after segment.
constexpr int HCS_MAXG = 16;                 // groups of eight pieces a thread keeps in flight per pass
// Block rows beyond the plan's fragments (blockIdx.y >= nfrag; round 4): the real-space pair sums of the electrode rows,
// four rows per block -- they depend on x, q only and may ride in any launch ahead of the dot kernel.
__global__ __launch_bounds__(256) void hc_sum_kernel(const int *__restrict__ frag_ptr, const int2 *__restrict__ ents, int R_pad, int nzc,
                                                     const double *__restrict__ Hp, double *__restrict__ Hc4, int nfrag, BRowArgs ra,
                                                     double *__restrict__ breal_out) {
  if ((int)blockIdx.y >= nfrag) {
    const int row = (((int)blockIdx.y - nfrag) * (int)gridDim.x + (int)blockIdx.x) * 4 + (int)(threadIdx.x >> 6);
    if (row < ra.ne) {
      const double v = b_row_pairs(ra, row, threadIdx.x & 63);
      if ((threadIdx.x & 63) == 0) breal_out[row] = v;
    }
    return;
  }
  const int g = blockIdx.y;
  const int s0 = frag_ptr[g], s1 = frag_ptr[g + 1];
  const int u = threadIdx.x & 7;
  const int e = blockIdx.x * 32 + (threadIdx.x >> 3);          // element of the fragment: (class, half, i)
  const bool live = e < 32 * nzc;
  const int cls = live ? e >> 5 : 0, half = (e >> 4) & 1, i = e & 15;
  double acc = 0.0;
  const int ng = (s1 - s0) / 8 + 1;                  // hc_slot_sum: the full groups, then one zero-padded group (possibly all zero)
  for (int gb = 0; gb < ng; gb += HCS_MAXG) {
    double v[HCS_MAXG];
#pragma unroll
    for (int k = 0; k < HCS_MAXG; ++k) {
      const int sidx = s0 + 8 * (gb + k) + u;
      v[k] = 0.0;
      if (live && sidx < s1) {
        const int2 en = ents[sidx];                  // x: offset of the fragment's first 'a' row in its piece, y: the band's rf
        v[k] = Hp[(size_t)en.x + (size_t)(cls * 32 * en.y + half * 16 * en.y + i)];
      }
    }
#pragma unroll
    for (int k = 0; k < HCS_MAXG; ++k) {
      if (gb + k >= ng) break;                        // (uniform over the block)
      double t = v[k];
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      acc += t;
    }
  }
  if (live && u == 0) Hc4[(size_t)cls * R_pad + (size_t)(g >> 2) * 128 + 64 * half + 16 * (g & 3) + i] = acc;
}

// grid = (ne_pad/64 atom blocks, 4 row quarters) -> partial slot blockIdx.y of bk; 16 waves: wave w takes rows r = w mod 16.
// The block first sums the 4 k-quarter slots of Hc for ITS rows and the nzc classes in use into LDS (one pass of coalesced
// loads) -- reading them per thread and per row from global cost more than the 42 MB Rp stream itself (20 -> 11 us).
__global__ __launch_bounds__(1024) void b_zc_dot_kernel(int n_own, const int *__restrict__ own_rt, int R_pad, int ne_pad, int nzc,
                                                        const double *__restrict__ Rp, const double *__restrict__ Hc4,
                                                        const int *__restrict__ zclass, double *__restrict__ bk, int nslot) {
  extern __shared__ __attribute__((aligned(16))) char zc_smem[];
  double *H = reinterpret_cast<double *>(zc_smem);          // [n_own * 32][nzc]
  __shared__ double red[16][64];
  const int a = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + a;
  const int zc = zclass[i];
  const size_t hp = (size_t)R_pad * 64;
  // Hc4 is class-major ([slot][class][R_pad]): the rows of one class are one contiguous run, and only the nzc classes in use are read
  const int nrow = n_own * 32;
  for (int e = threadIdx.x; e < nrow * nzc; e += 1024) {
    const int cls = e / nrow, rowl = e - cls * nrow;
    const double *h = Hc4 + (size_t)cls * R_pad + (size_t)own_rt[rowl >> 5] * 128 + blockIdx.y * 32 + (rowl & 31);
    if (nslot == 1) { H[rowl * nzc + cls] = h[0]; continue; }          // already summed (hc_sum_kernel)
    const double h03 = (h[0] + h[hp]) + (h[2 * hp] + h[3 * hp]);       // slots = column slices of the reduction (4 or 8)
    H[rowl * nzc + cls] = nslot == 4 ? h03 : h03 + ((h[4 * hp] + h[5 * hp]) + (h[6 * hp] + h[7 * hp]));
  }
  __syncthreads();
  double sum = 0.0;
  // 4 row tiles at a time: 8 Rp rows in flight per thread
  for (int k0 = 0; k0 < n_own; k0 += 4) {        // this rank's row tiles
    double rp[8], hc[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = k0 + u < n_own;
      const int kk = ok ? k0 + u : k0;
      const int rt = own_rt[kk];
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const size_t r = (size_t)rt * 128 + blockIdx.y * 32 + w + 16 * v;
        rp[2 * u + v] = ok ? Rp[r * ne_pad + i] : 0.0;
        hc[2 * u + v] = H[(kk * 32 + w + 16 * v) * nzc + zc];
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += rp[k] * hc[k];
  }
  red[w][a] = sum;
  __syncthreads();
  if (w == 0) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += red[k][a];
    bk[(size_t)blockIdx.y * ne_pad + i] = -tot;
  }
}

// ---- the same dot, finished in one launch: grid = ne_pad / 16 atom blocks; thread = (atom a of 16, planar vector w of 64).
// Every thread takes planar vector 64 tile + w of each of this rank's row tiles, i.e. G rows w ('a') and 64 + w ('b').  Their
// electrode phases are NOT streamed from the [R_pad][Ne] table (34 MB at the headline size, once per update) but rebuilt from
// the electrode atoms' axis tables Xe[kx][i], Ye[ky][i] (3 MB, L2-resident) with the products the host used to fill that table
// (electrode_trig, km_ewald.cpp:464-477; no FMA contraction: the same bits).  The four k-quarter slots of Hc are summed into
// LDS for ALL 128 rows of a tile; the 64 row lanes of an atom are added in a fixed order; then slab term and real-space sum
// (already formed by elyte_phase's spare blocks): b is complete, no b_real_combine launch.  Used when the whole Hc table of
// this rank fits in 64 KB of LDS (planar electrodes: 2-6 z classes).
__global__ __launch_bounds__(1024) void b_zc_final_kernel(int n_own, const int *__restrict__ own_rt, int R_pad, int ne_pad, int nzc,
                                                          const double2 *__restrict__ Xe, const double2 *__restrict__ Ye,
                                                          const int *__restrict__ own_pv, const double *__restrict__ Hc4,
                                                          const int *__restrict__ zclass, BRowArgs ra, int nslot,
                                                          const int *__restrict__ slot_ptr, const int *__restrict__ slot_idx) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) char zf_smem[];
  double *H = reinterpret_cast<double *>(zf_smem);          // [n_own * 128][nzc]
  __shared__ double red[64][17];
  const int a = threadIdx.x & 15, w = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + a;
  const int zc = zclass[i];
  const size_t hp = (size_t)R_pad * 64;
  // the slab scalar by the first wave, with b_real_combine's summation tree (the 16 finishing threads are lanes of that wave)
  const double sc = (ra.slab && threadIdx.x < 64) ? b_slab_scalar(ra, threadIdx.x) : 0.0;
  // what the finishing threads add at the very end is requested now (it was one more dependent round trip behind the last barrier)
  double fin_z = 0.0, fin_r = 0.0;
  if (w == 0 && i < ra.ne) { if (ra.slab) fin_z = ra.ele_z[i]; fin_r = ra.breal[i]; }
  // the first 8 row tiles' phases are requested BEFORE the Hc table is staged: the two latencies overlap instead of adding up --
  // at the headline size that is everything this thread reads
  double2 xe0[8], ye0[8];
  int sg0[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const bool ok = u < n_own;                   // an absent tile gets a zero X phase: it adds 0 * H
    const int pk = own_pv[(ok ? u : 0) * 64 + w];      // |kx| | |ky| << 12 | (ky < 0) << 24
    sg0[u] = (pk >> 24) & 1;
    xe0[u] = ok ? Xe[(size_t)(pk & 4095) * ne_pad + i] : make_double2(0.0, 0.0);
    ye0[u] = Ye[(size_t)((pk >> 12) & 4095) * ne_pad + i];
  }
  const int nrow = n_own * 128;
  if (slot_ptr) {
    // Hc4 = the segments' projected pieces (sk_project_out): the pieces of a row tile are added here, in list order
    for (int e = threadIdx.x; e < nrow * nzc; e += 1024) {
      const int cls = e / nrow, rowl = e - cls * nrow, k = rowl >> 7;
      H[rowl * nzc + cls] = hc_slot_sum(Hc4 + cls * 128 + (rowl & 127), slot_idx, slot_ptr[k], slot_ptr[k + 1], (size_t)SK_HC * SK_HC_ROWS);
    }
  } else
  for (int e = threadIdx.x; e < nrow * nzc; e += 1024) {      // class-major Hc4: coalesced runs of rows, classes in use only
    const int cls = e / nrow, rowl = e - cls * nrow;
    const double *h = Hc4 + (size_t)cls * R_pad + (size_t)own_rt[rowl >> 7] * 128 + (rowl & 127);
    if (nslot == 1) { H[rowl * nzc + cls] = h[0]; continue; }          // already summed (hc_sum_kernel)
    const double h03 = (h[0] + h[hp]) + (h[2 * hp] + h[3 * hp]);       // slots = column slices of the reduction (4 or 8)
    H[rowl * nzc + cls] = nslot == 4 ? h03 : h03 + ((h[4 * hp] + h[5 * hp]) + (h[6 * hp] + h[7 * hp]));
  }
  __syncthreads();
  double sum = 0.0;
  // (cos, sin)(kx x +- ky y) of a planar vector for atom i: neg flips the sign of sin(ky y); padding vectors read the all-zero
  // row kxmax + 1 of Xe (like the electrolyte's X table) and come out as zero
  auto phase = [](double2 X, double2 Y, int neg, double &pa, double &pb) {
    const double sy = neg ? -Y.y : Y.y;
    pa = X.x * Y.x - X.y * sy;
    pb = X.x * sy + X.y * Y.x;
  };
#pragma unroll
  for (int u = 0; u < 8; ++u) {                  // tiles 0..7 (absent tiles add 0 * H)
    const int kk = u < n_own ? u : 0;
    double pa, pb;
    phase(xe0[u], ye0[u], sg0[u], pa, pb);
    sum += pa * H[(kk * 128 + w) * nzc + zc];
    sum += pb * H[(kk * 128 + w + 64) * nzc + zc];
  }
  for (int k0 = 8; k0 < n_own; k0 += 4) {        // 4 row tiles at a time
    double2 xe[4], ye[4];
    int sg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = k0 + u < n_own;
      const int pk = own_pv[(ok ? k0 + u : k0) * 64 + w];
      sg[u] = (pk >> 24) & 1;
      xe[u] = ok ? Xe[(size_t)(pk & 4095) * ne_pad + i] : make_double2(0.0, 0.0);
      ye[u] = Ye[(size_t)((pk >> 12) & 4095) * ne_pad + i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k0 + u < n_own ? k0 + u : k0;
      double pa, pb;
      phase(xe[u], ye[u], sg[u], pa, pb);
      sum += pa * H[(kk * 128 + w) * nzc + zc];
      sum += pb * H[(kk * 128 + w + 64) * nzc + zc];
    }
  }
  red[w][a] = sum;
  __syncthreads();
  if (w == 0 && i < ra.ne) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 64; k += 4) tot += (red[k][a] + red[k + 1][a]) + (red[k + 2][a] + red[k + 3][a]);
    double v = -tot;
    if (ra.slab) v -= fin_z * sc;
    v += fin_r;
    ra.b_out[i] = v;
    if (ra.slab && i == 0 && ra.slab_out) *ra.slab_out = sc;
  }
}

// Many pieces per row fragment (the z-window form: one per range, 113 at the headline size): 32 threads per element -- octet o of the
// four takes the groups g = o, o + 4, ... (eight pieces each, thread u of the octet the piece 8 g + u: at most a handful of loads per
// thread instead of fifteen), adds a group by the same butterfly, and the first lane of the element collects the groups' sums and adds
// them in list order: the same additions in the same order as hc_sum_kernel, the same bits.  Pair rows in the block rows behind the
// fragments as there.
constexpr int HCW_MAXK = 4;                  // groups per octet and pass (16 or 32 groups = 128 / 256 pieces a pass)
template <int NO /*octets per element: 4 or 8*/>
__global__ __launch_bounds__(256) void hc_sum_wide_kernel(const int *__restrict__ frag_ptr, const int2 *__restrict__ ents, int R_pad, int nzc,
                                                          const double *__restrict__ Hp, double *__restrict__ Hc4, int nfrag, BRowArgs ra,
                                                          double *__restrict__ breal_out) {
  if ((int)blockIdx.y >= nfrag) {
    const int row = (((int)blockIdx.y - nfrag) * (int)gridDim.x + (int)blockIdx.x) * 4 + (int)(threadIdx.x >> 6);
    if (row < ra.ne) {
      const double v = b_row_pairs(ra, row, threadIdx.x & 63);
      if ((threadIdx.x & 63) == 0) breal_out[row] = v;
    }
    return;
  }
  const int g = blockIdx.y;
  const int s0 = frag_ptr[g], s1 = frag_ptr[g + 1];
  const int u = threadIdx.x & 7, o = (threadIdx.x >> 3) & (NO - 1);
  const int e = blockIdx.x * (32 / NO) + threadIdx.x / (8 * NO);      // element of the fragment: (class, half, i)
  const bool live = e < 32 * nzc;
  const int cls = live ? e >> 5 : 0, half = (e >> 4) & 1, i = e & 15;
  const int lane = threadIdx.x & 63, lane0 = lane & ~(8 * NO - 1);  // (first lane of the element in its wave)
  double acc = 0.0;
  const int ng = (s1 - s0) / 8 + 1;                  // hc_slot_sum: the full groups, then one zero-padded group (possibly all zero)
  for (int gb = 0; gb < ng; gb += NO * HCW_MAXK) {
    double t[HCW_MAXK];
#pragma unroll
    for (int k = 0; k < HCW_MAXK; ++k) {
      const int sidx = s0 + 8 * (gb + NO * k + o) + u;
      double v = 0.0;
      if (live && sidx < s1) {
        const int2 en = ents[sidx];                  // x: offset of the fragment's first 'a' row in its piece, y: the band's rf
        v = Hp[(size_t)en.x + (size_t)(cls * 32 * en.y + half * 16 * en.y + i)];
      }
      t[k] = v;
    }
#pragma unroll
    for (int k = 0; k < HCW_MAXK; ++k) {
      double v = t[k];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      t[k] = v;
    }
#pragma unroll
    for (int k = 0; k < HCW_MAXK; ++k)
#pragma unroll
      for (int oo = 0; oo < NO; ++oo) {
        const double tv = __shfl(t[k], lane0 + 8 * oo, 64);   // group gb + NO k + oo, from its octet
        if (gb + NO * k + oo < ng) acc += tv;                 // (uniform over the block)
      }
  }
  if (live && (threadIdx.x & (8 * NO - 1)) == 0) Hc4[(size_t)cls * R_pad + (size_t)(g >> 2) * 128 + 64 * half + 16 * (g & 3) + i] = acc;
}

// ---- round 5: hc_sum_kernel and b_zc_final_kernel as ONE launch (headline and slab plans: many pieces per row fragment) ----------
// The dot kernel needs the whole class table H of the rank, the pieces' sums are a few workgroups' work: they used to be two launches
// with the real-space pair rows riding in the first.  Here the first P workgroups add the pieces (hc_sum_kernel's arithmetic: eight
// lanes per element, groups of eight by a butterfly, groups in list order) and PUBLISH the table -- write-through stores (agent-scope
// relaxed atomics = sc1), the storing waves' vmcnt(0), the workgroup barrier, ONE relaxed agent-scope ticket add per workgroup: the
// fence-free hand-off of the inverse's panel (conp_inverse.hip; a device-scope fence writes back and invalidates an XCD's whole L2:
// the hand-offs of rounds 2-4 that lost were built on those, or put the waiting on sk_gemm's critical path).  The dot workgroups do
// everything that does not need H first -- request their first eight tiles' phases, and form the pair sums of their 16 electrode
// rows, one wavefront per row (the rows that rode along in hc_sum's launch) -- then ONE thread polls the ticket, and H comes out of
// memory by sc1 loads.  Co-residency: the grid is P + ne_pad / 16 workgroups of 1024 threads (273 at the headline size, two fit a
// CU), producers first in dispatch order; a dot workgroup whose wait runs out (bounded spin) adds the pieces it needs ITSELF --
// the same bits, frag_sum_serial -- so a co-scheduled kernel can slow this one down but not hang it.
// The ticket word is zeroed by sk_gemm's launch (workgroup 0), which is always in front of this one on the stream.
__device__ __forceinline__ double frag_sum_serial(const int *__restrict__ frag_ptr, const int2 *__restrict__ ents,
                                                  const double *__restrict__ Hp, int g, int cls, int half, int i) {
  const int s0 = frag_ptr[g], s1 = frag_ptr[g + 1];
  double acc = 0.0;
  const int ng = (s1 - s0) / 8 + 1;
  for (int gb = 0; gb < ng; ++gb) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int sidx = s0 + 8 * gb + u;
      v[u] = 0.0;
      if (sidx < s1) {
        const int2 en = ents[sidx];
        v[u] = Hp[(size_t)en.x + (size_t)(cls * 32 * en.y + half * 16 * en.y + i)];
      }
    }
    acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  return acc;
}
struct HcFused {
  const int *frag_ptr; const int2 *ents; const double *Hp; double *Hc; unsigned *ticket;
  int nfrag, P; unsigned spin_limit;
};
__global__ __launch_bounds__(1024) void b_zc_fused_kernel(HcFused hf, int n_own, const int *__restrict__ own_rt, int R_pad, int ne_pad,
                                                          int nzc, const double2 *__restrict__ Xe, const double2 *__restrict__ Ye,
                                                          const int *__restrict__ own_pv, const int *__restrict__ zclass, BRowArgs ra,
                                                          BRowArgs pairs) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) char zf_smem[];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < hf.P) {
    // ---- producer: 128 elements of the class table per workgroup, eight lanes each
    const int u = tid & 7, e = (int)blockIdx.x * 128 + (tid >> 3);
    const int per = 32 * nzc;
    const bool live = e < hf.nfrag * per;
    const int g = live ? e / per : 0, rem = live ? e - g * per : 0;
    const int cls = rem >> 5, half = (rem >> 4) & 1, i = rem & 15;
    const int s0 = hf.frag_ptr[g], s1 = live ? hf.frag_ptr[g + 1] : s0;
    double acc = 0.0;
    const int ng = (s1 - s0) / 8 + 1;                  // (uniform over the wave: its eight elements belong to one fragment)
    for (int gb = 0; gb < ng; gb += HCS_MAXG) {
      double v[HCS_MAXG];
#pragma unroll
      for (int k = 0; k < HCS_MAXG; ++k) {
        const int sidx = s0 + 8 * (gb + k) + u;
        v[k] = 0.0;
        if (sidx < s1) {
          const int2 en = hf.ents[sidx];
          v[k] = hf.Hp[(size_t)en.x + (size_t)(cls * 32 * en.y + half * 16 * en.y + i)];
        }
      }
#pragma unroll
      for (int k = 0; k < HCS_MAXG; ++k) {
        if (gb + k >= ng) break;
        double t = v[k];
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        acc += t;
      }
    }
    if (live && u == 0)
      __hip_atomic_store(hf.Hc + (size_t)cls * R_pad + (size_t)(g >> 2) * 128 + 64 * half + 16 * (g & 3) + i, acc, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(hf.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // ---- dot workgroup: b_zc_final_kernel with the pair sums and the wait in it
  double *H = reinterpret_cast<double *>(zf_smem);          // [n_own * 128][nzc]
  __shared__ double red[64][17];
  __shared__ double prl[16];
  __shared__ int s_ready;
  const int blk = (int)blockIdx.x - hf.P;
  const int a = tid & 15, w = tid >> 4;
  const int i = blk * 16 + a;
  const int zc = zclass[i];
  const double sc = (ra.slab && tid < 64) ? b_slab_scalar(ra, tid) : 0.0;
  double fin_z = 0.0;
  if (w == 0 && i < ra.ne && ra.slab) fin_z = ra.ele_z[i];
  double2 xe0[8], ye0[8];
  int sg0[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const bool ok = u < n_own;
    const int pk = own_pv[(ok ? u : 0) * 64 + w];
    sg0[u] = (pk >> 24) & 1;
    xe0[u] = ok ? Xe[(size_t)(pk & 4095) * ne_pad + i] : make_double2(0.0, 0.0);
    ye0[u] = Ye[(size_t)((pk >> 12) & 4095) * ne_pad + i];
  }
  {
    // the real-space pair sum of electrode row 16 blk + wave, by this wavefront
    const int row = blk * 16 + (tid >> 6);
    const double v = row < pairs.ne ? b_row_pairs(pairs, row, tid & 63) : 0.0;
    if ((tid & 63) == 0) prl[tid >> 6] = v;
  }
  if (tid == 0) {
    unsigned spins = 0;
    int ok = 1;
    while (__hip_atomic_load(hf.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)hf.P) {
      if (++spins > hf.spin_limit) { ok = 0; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    s_ready = ok;
  }
  __syncthreads();
  const int nrow = n_own * 128;
  if (s_ready) {
    // class-major table: runs of rows, classes in use only; sc1 loads, up to four per thread in flight (64 KB of table at most)
    for (int e0 = tid; e0 < nrow * nzc; e0 += 4 * 1024) {
      double hv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + 1024 * u;
        hv[u] = 0.0;
        if (e < nrow * nzc) {
          const int cls = e / nrow, rowl = e - cls * nrow;
          const double *src = hf.Hc + (size_t)cls * R_pad + (size_t)own_rt[rowl >> 7] * 128 + (rowl & 127);
          asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(hv[u]) : "v"(src) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + 1024 * u;
        if (e < nrow * nzc) { const int cls = e / nrow, rowl = e - cls * nrow; H[rowl * nzc + cls] = hv[u]; }
      }
    }
  } else {
    for (int e = tid; e < nrow * nzc; e += 1024) {      // the wait ran out: the same sums, by this workgroup
      const int cls = e / nrow, rowl = e - cls * nrow, r = rowl & 127;
      H[rowl * nzc + cls] = frag_sum_serial(hf.frag_ptr, hf.ents, hf.Hp, 4 * own_rt[rowl >> 7] + ((r & 63) >> 4), cls, r >> 6, r & 15);
    }
  }
  __syncthreads();
  double sum = 0.0;
  auto phase = [](double2 X, double2 Y, int neg, double &pa, double &pb) {
    const double sy = neg ? -Y.y : Y.y;
    pa = X.x * Y.x - X.y * sy;
    pb = X.x * sy + X.y * Y.x;
  };
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int kk = u < n_own ? u : 0;
    double pa, pb;
    phase(xe0[u], ye0[u], sg0[u], pa, pb);
    sum += pa * H[(kk * 128 + w) * nzc + zc];
    sum += pb * H[(kk * 128 + w + 64) * nzc + zc];
  }
  for (int k0 = 8; k0 < n_own; k0 += 4) {
    double2 xe[4], ye[4];
    int sg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = k0 + u < n_own;
      const int pk = own_pv[(ok ? k0 + u : k0) * 64 + w];
      sg[u] = (pk >> 24) & 1;
      xe[u] = ok ? Xe[(size_t)(pk & 4095) * ne_pad + i] : make_double2(0.0, 0.0);
      ye[u] = Ye[(size_t)((pk >> 12) & 4095) * ne_pad + i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k0 + u < n_own ? k0 + u : k0;
      double pa, pb;
      phase(xe[u], ye[u], sg[u], pa, pb);
      sum += pa * H[(kk * 128 + w) * nzc + zc];
      sum += pb * H[(kk * 128 + w + 64) * nzc + zc];
    }
  }
  red[w][a] = sum;
  __syncthreads();
  if (w == 0 && i < ra.ne) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 64; k += 4) tot += (red[k][a] + red[k + 1][a]) + (red[k + 2][a] + red[k + 3][a]);
    double v = -tot;
    if (ra.slab) v -= fin_z * sc;
    v += prl[a];
    ra.b_out[i] = v;
    if (ra.slab && i == 0 && ra.slab_out) *ra.slab_out = sc;
  }
}

// (The four row-quarter partials of an atom are added, with the slab and real-space terms, by b_real_combine_kernel.  Letting the
//  last-arriving quarter block of each atom block do that here -- sc1 hand-off, ticket -- was measured: 32 -> 47 us for the pair
//  at the headline size, 22 -> 33 us on il_onelayer: 64 late workgroups do serially what 4096 waves of their own launch do at once.)
// true when b_zc_final_kernel can take the place of b_zc_dot + b_real_combine
bool zc_final_fits(int n_own, int nzc) { return n_own > 0 && (size_t)n_own * 128 * nzc * sizeof(double) <= 64 * 1024; }

static void launch_b_zc_final(hipStream_t s, const DevPlan &pl, int n_own, const int *own_rt, int ne_pad, int nzc, const double2 *Xe,
                              const double2 *Ye, const int *own_pv, const double *Hc, const int *zclass, const BRowArgs &fin, int nslot,
                              const int *slot_ptr = nullptr, const int *slot_idx = nullptr) {
  const size_t lds = (size_t)n_own * 128 * nzc * sizeof(double);
  static DynLdsCache granted{};
  ensure_dyn_lds(b_zc_final_kernel, lds, granted);
  hipLaunchKernelGGL(b_zc_final_kernel, dim3(ne_pad / 16), dim3(1024), lds, s, n_own, own_rt, pl.R_pad, ne_pad, nzc, Xe, Ye, own_pv, Hc,
                     zclass, fin, nslot, slot_ptr, slot_idx);
}

static void launch_b_zc_dot(hipStream_t s, int n_own, const int *own_rt, int R_pad, int ne_pad, int nzc, const double *Rp,
                            const double *Hc, const int *zclass, double *bk_part, int nslot) {
  const size_t lds = (size_t)(n_own > 0 ? n_own : 1) * 32 * nzc * sizeof(double);      // the host keeps this <= 96 KB (conp_fix.cpp)
  static DynLdsCache granted{};
  ensure_dyn_lds(b_zc_dot_kernel, lds, granted);
  hipLaunchKernelGGL(b_zc_dot_kernel, dim3(ne_pad / 64, 4), dim3(1024), lds, s, n_own, own_rt, R_pad, ne_pad, nzc, Rp, Hc, zclass, bk_part, nslot);
}

// Planar electrodes with one column tile (nz <= 160): the last partial-tile sum and the Hc product in ONE kernel -- a block
// of sk_reduce already holds its 16 x 80 piece of (w G) in LDS in MFMA-fragment order, exactly the A operand b_hc needs, so the
// Gwf round trip and one launch go away (the decks' updates are launch-bound).  Same sums in the same order as the two
// kernels; Hc4 slot = the block's 80-column quarter.
template <int SKR_SL>
__global__ __launch_bounds__(SKR_T) void sk_reduce_hc_kernel(int C_pad, const SkTile *__restrict__ tiles, const double *__restrict__ part,
                                                           const double *__restrict__ wfull, double *__restrict__ G,
                                                           const double *__restrict__ Tzc /*[C_pad][64]*/,
                                                           double *__restrict__ Hc4 /*[SKR_SL slots][64][R_pad]*/, int R_pad, int nzc16,
                                                           int level) {
  constexpr int SKR_W = 320 / SKR_SL, SKR_K = 16 * SKR_W / SKR_T;
  __shared__ double tr[16 * SKR_W];
  const SkTile tl = tiles[blockIdx.x / (8 * SKR_SL)];
  const int f16 = (blockIdx.x / SKR_SL) & 7;
  const int q = blockIdx.x % SKR_SL;
  const size_t plane = 128 * 320;
  const int stride = level == 2 ? SKR_GROUP : 1;
  const int count = level == 2 ? (tl.nsplit + SKR_GROUP - 1) / SKR_GROUP : tl.nsplit;
  const size_t step = (size_t)stride * plane;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int nks = 8 * tl.nba;
  constexpr int NKS = SKR_W / 4;               // k-steps (4 columns each) of this block's slice
  // the z-class operand of the product below depends on the tile only: the waves that will multiply request their values
  // now, so that the loads are in flight while the partial tiles are summed (they were dependent-looking groups of four
  // behind the barrier: 4 us of the 9.4 this kernel took on il_onelayer)
  double bz[NKS];
  if (wave < nzc16) {
    const double *bp = Tzc + (size_t)(tl.ct * 320 + SKR_W * q + fk) * 64 + 16 * wave + fr;
#pragma unroll
    for (int tsl = 0; tsl < NKS; ++tsl) bz[tsl] = NKS * q + tsl < nks ? bp[(size_t)tsl * 256] : 0.0;
  }
  const int cl = threadIdx.x % SKR_W;                                   // element(s) of this thread: see sk_reduce_kernel
  const int row0 = skr_row0<SKR_K>(threadIdx.x / SKR_W);
  const int col = SKR_W * q + cl;
  double sums[SKR_K];
#pragma unroll
  for (int k = 0; k < SKR_K; ++k) sums[k] = 0.0;
  if (col < 32 * tl.nba) {
    const double *src = part + (size_t)tl.item0 * plane + sk_part_off(16 * f16 + row0, col);
    if constexpr (SKR_K == 2) { const double2 v = part_sum2(src, step, count); sums[0] = v.x; sums[1] = v.y; }
    else sums[0] = part_sum1(src, step, count);
  }
#pragma unroll
  for (int k = 0; k < SKR_K; ++k) {
    const int row = row0 + 4 * k, rowl = 16 * f16 + row;
    const size_t grow = (size_t)tl.rt * 128 + rowl, gcol = (size_t)tl.ct * 320 + col;
    G[grow * C_pad + gcol] = sums[k];
    tr[(cl >> 2) * 64 + (cl & 3) * 16 + row] = wfull[grow * C_pad + gcol] * sums[k];
  }
  __syncthreads();
  if (wave >= nzc16) return;
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int tsl = 0; tsl < NKS; ++tsl)
    if (NKS * q + tsl < nks) acc = MFMA_F64(tr[tsl * 64 + lane], bz[tsl], acc);      // (wave-uniform condition)
  const int rf = tl.rt * 8 + f16;
  double *out = Hc4 + (size_t)q * R_pad * 64 + (size_t)(16 * wave + fr) * R_pad;       // slot = this block's column slice
#pragma unroll
  for (int r = 0; r < 4; ++r) out[16 * rf + fk + 4 * r] = acc[r];
}

// sk_reduce (+ level 1 when tiles are heavily split) with the Hc product fused in, then the per-atom dot
template <int SL>
static void launch_reduce_hc_sl(hipStream_t s, const DevPlan &pl, const SkTile *tiles, int ntiles, int max_nsplit, double *part, double *G,
                                const double *Tzc, double *Hc, int nzc16) {
  int level = 0;
  if (max_nsplit > skr_two_level_from()) {
    const int ngroups = (max_nsplit + SKR_GROUP - 1) / SKR_GROUP;
    hipLaunchKernelGGL(sk_reduce_kernel<SL>, dim3(ntiles * 8 * SL, ngroups), dim3(SKR_T), 0, s, pl.C_pad, tiles, part, pl.wfull, G, nullptr, 1);
    level = 2;
  }
  hipLaunchKernelGGL(sk_reduce_hc_kernel<SL>, dim3(ntiles * 8 * SL), dim3(SKR_T), 0, s, pl.C_pad, tiles, part, pl.wfull, G, Tzc, Hc, pl.R_pad,
                     nzc16, level);
}
void launch_reduce_project_zclass(hipStream_t s, const DevPlan &pl, const SkTile *tiles, int ntiles, int max_nsplit, double *part,
                                  double *G, int ne_pad, int n_own, const int *own_rt, int nzc, const double *Tzc, const double *Rp,
                                  const double2 *Xe, const double2 *Ye, const int *own_pv, const int *zclass, double *Hc,
                                  double *bk_part, const BRowArgs *fin) {
  if (ntiles <= 0) return;
  const int nzc16 = (nzc + 15) / 16;
  const int nslot = skr_slices(ntiles);
  if (nslot == 4) launch_reduce_hc_sl<4>(s, pl, tiles, ntiles, max_nsplit, part, G, Tzc, Hc, nzc16);
  else launch_reduce_hc_sl<8>(s, pl, tiles, ntiles, max_nsplit, part, G, Tzc, Hc, nzc16);
  if (fin) launch_b_zc_final(s, pl, n_own, own_rt, ne_pad, nzc, Xe, Ye, own_pv, Hc, zclass, *fin, nslot);
  else launch_b_zc_dot(s, n_own, own_rt, pl.R_pad, ne_pad, nzc, Rp, Hc, zclass, bk_part, nslot);
}

void launch_b_project_zclass(hipStream_t s, const DevPlan &pl, int ne_pad, const int *rt_mine, int n_own, const int *own_rt, int nzc, const double *Gwf,
                             const double *Tzc, const double *Rp, const double2 *Xe, const double2 *Ye, const int *own_pv, const int *zclass,
                             double *Hc, double *bk_part, const BRowArgs *fin) {
  const int nzc16 = (nzc + 15) / 16;
  hipLaunchKernelGGL(b_hc_kernel, dim3(pl.R_pad / 16, 4), dim3(256), 0, s, pl.C_pad, pl.n_col_tiles, rt_mine, nzc16,
                     pl.nb_act, Gwf, Tzc, Hc, pl.R_pad);
  if (fin) launch_b_zc_final(s, pl, n_own, own_rt, ne_pad, nzc, Xe, Ye, own_pv, Hc, zclass, *fin, 4);      // b_hc writes four k-quarter slots
  else launch_b_zc_dot(s, n_own, own_rt, pl.R_pad, ne_pad, nzc, Rp, Hc, zclass, bk_part, 4);
}

// planar electrodes, sk_gemm in projecting mode: Hp = the segments' pieces; slot lists per owned row tile.  fin: the dot kernel
// adds the pieces itself (presum: hc_sum first -- many pieces per tile, every block of the dot kernel would re-add them all);
// otherwise hc_sum -> slot 0 of Hc -> b_zc_dot.
void launch_project_zclass_pieces(hipStream_t s, const DevPlan &pl, int ne_pad, int n_own, const int *own_rt, int nzc, const double *Hp,
                                  const int *slot_ptr, const int *slot_idx, bool presum, const int *frag_ptr, const int2 *frag_ents, int nfrag,
                                  const double *Rp, const double2 *Xe,
                                  const double2 *Ye, const int *own_pv, const int *zclass, double *Hc, double *bk_part, const BRowArgs *fin,
                                  const BRowArgs *pairs, double *breal_out, unsigned *ticket, unsigned spin_limit, bool wide) {
  if (n_own <= 0) return;
  if (fin && !presum) { launch_b_zc_final(s, pl, n_own, own_rt, ne_pad, nzc, Xe, Ye, own_pv, Hp, zclass, *fin, 0, slot_ptr, slot_idx); return; }
  if (nfrag > 0 && fin && pairs && ticket) {
    // ONE launch (round 5): the pieces' sums published through the fence-free hand-off, the pair sums and the dot behind them
    HcFused hf{frag_ptr, frag_ents, Hp, Hc, ticket, nfrag, (nfrag * 32 * nzc + 127) / 128, spin_limit};
    const BRowArgs fa = *fin;
    const size_t lds = (size_t)n_own * 128 * nzc * sizeof(double);
    static DynLdsCache granted{};
    ensure_dyn_lds(b_zc_fused_kernel, lds, granted);
    hipLaunchKernelGGL(b_zc_fused_kernel, dim3(hf.P + ne_pad / 16), dim3(1024), lds, s, hf, n_own, own_rt, pl.R_pad, ne_pad, nzc, Xe, Ye,
                       own_pv, zclass, fa, *pairs);
    return;
  }
  if (nfrag > 0) {
    // pairs: the real-space pair sums ride in this launch (block rows behind the fragments)
    const BRowArgs ra = pairs ? *pairs : BRowArgs{};
    if (wide) {                                        // many pieces per fragment: 32 threads per element (hc_sum_wide_kernel)
      // (eight octets per element: 19.7 us for the sum + the dot at the headline size against 18.7 with four and 21.8 with hc_sum_kernel)
      constexpr int NO = 4;
      const int extra = pairs ? ((ra.ne + 3) / 4 + NO * nzc - 1) / (NO * nzc) : 0;
      hipLaunchKernelGGL(hc_sum_wide_kernel<NO>, dim3(NO * nzc, nfrag + extra), dim3(256), 0, s, frag_ptr, frag_ents, pl.R_pad, nzc, Hp, Hc, nfrag, ra,
                         breal_out);
    } else {
      const int extra = pairs ? ((ra.ne + 3) / 4 + nzc - 1) / nzc : 0;
      hipLaunchKernelGGL(hc_sum_kernel, dim3(nzc, nfrag + extra), dim3(256), 0, s, frag_ptr, frag_ents, pl.R_pad, nzc, Hp, Hc, nfrag, ra, breal_out);
    }
  }
  if (fin) launch_b_zc_final(s, pl, n_own, own_rt, ne_pad, nzc, Xe, Ye, own_pv, Hc, zclass, *fin, 1);
  else launch_b_zc_dot(s, n_own, own_rt, pl.R_pad, ne_pad, nzc, Rp, Hc, zclass, bk_part, 1);
}

// ================================================================================================
// 4. real-space kernels.  erfc(x)/r through the reference's 5-term polynomial (fix_conp.cpp:53-60, 1446-1454)
// ================================================================================================
// one wave per electrode row (global eleall index), fused with the assembly of this rank's b contribution:
//   b[row] = bk_half0[row] + bk_half1[row]                       (k-space shard, km_ewald.cpp:789-825)
//          - z_row * sum_j 4 pi q_j z_j / V                      (slab, km_ewald.cpp:827-847; rank 0 only)
//          - sum_pairs q_j [erfc(g r) - erfc(eta r)] / r         (rows row0..row1 only; fix_conp.cpp:1313-1353)
__global__ __launch_bounds__(256) void b_real_combine_kernel(BRowArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.ne) return;
  const int lane = threadIdx.x & 63;
  const double sc = a.slab ? b_slab_scalar(a, lane) : 0.0;     // every wave derives the same scalar with the same summation tree
  b_row(a, row, lane, sc);
}

void launch_b_real_combine(hipStream_t s, int ne, int ne_pad, int row0, int row1, const int *row_ptr, const int *ele_atom,
                           const int *oth_atom, const double *x, const double *q, const int *type, RealParams rp, int add_k,
                           const double *bk, int slab, const double *ele_z, const double *slab_part, int n_slab_part,
                           double slab_pref, double *b_out, double *slab_out, const double *breal) {
  BRowArgs a = make_brow(ne, ne_pad, row0, row1, row_ptr, ele_atom, oth_atom, x, q, type, rp, add_k, bk, slab, ele_z, slab_part,
                         n_slab_part, slab_pref, b_out, slab_out);
  a.breal = breal;
  hipLaunchKernelGGL(b_real_combine_kernel, dim3((ne + 3) / 4), dim3(256), 0, s, a);
}

// ================================================================================================
// 5. dense solve pieces.  GEMV: one wave per row, 16-byte loads (fix_conp.cpp:1135-1139 ddot_ per row)
// ================================================================================================
// Cache policy of the matrix stream.  The matrix is read once per update, every update.  A small matrix (the decks: 5 - 22 MB)
// stays in L2 / the Infinity Cache between updates with plain loads.  A large one is streamed with non-temporal loads so that it
// does not evict the phase tables and partial tiles the next update reads.  Measured at the headline size (134 MB of matrix beside
// ~110 MB of other per-update traffic, nominally inside the 256 MiB Infinity Cache): plain loads made the GEMV 1.5 us faster and
// the phase and reduction kernels 3.5 us slower -- the matrix does not stay resident beside that much written data.
constexpr size_t GEMV_RESIDENT_BYTES = (size_t)64 << 20;
static bool gemv_nt(size_t matrix_bytes) {
  static const char *e = diag_switch("CONP_GEMV_NT");      // comparison switch: 0 / 1 forces the policy
  if (e) return atoi(e) != 0;
  return matrix_bytes > GEMV_RESIDENT_BYTES;
}
template <bool NT>
__device__ __forceinline__ double2 nt_load(const double2 *p) {
  if constexpr (!NT) return *p;
  double2 v;
  v.x = __builtin_nontemporal_load(&p->x);
  v.y = __builtin_nontemporal_load(&p->y);
  return v;
}

// this lane's share of one row's dot product S[row,:] . b -- ONE function for every GEMV kernel, so that they agree to the bit.
// Eight 16-byte row loads in flight per lane (two groups of four, multiplied in the order a 4-wide loop would use them): with
// 16 resident waves per CU that is 128 KB on its way per CU, what ~6 TB/s at HBM latency asks for.
template <bool NT>
__device__ __forceinline__ double gemv_row_dot(int n, const double *__restrict__ srow, const double *__restrict__ b, int lane) {
  double s0 = 0.0, s1 = 0.0;
  if ((n & 1) == 0) {
    const double2 *s2 = reinterpret_cast<const double2 *>(srow);
    const double2 *b2 = reinterpret_cast<const double2 *>(b);
    double t0 = 0.0, t1 = 0.0;      // the two accumulator pairs are combined in a fixed order
    int j = lane;
    for (; j + 448 < n / 2; j += 512) {
      double2 a[8], c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = nt_load<NT>(s2 + j + 64 * u);
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = b2[j + 64 * u];
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        s0 = fma(a[u].x, c[u].x, s0); s1 = fma(a[u].y, c[u].y, s1);
        t0 = fma(a[u + 1].x, c[u + 1].x, t0); t1 = fma(a[u + 1].y, c[u + 1].y, t1);
      }
    }
    for (; j + 192 < n / 2; j += 256) {
      const double2 a0 = nt_load<NT>(s2 + j), a1 = nt_load<NT>(s2 + j + 64), a2 = nt_load<NT>(s2 + j + 128), a3 = nt_load<NT>(s2 + j + 192);
      const double2 b0 = b2[j], b1 = b2[j + 64], b2v = b2[j + 128], b3 = b2[j + 192];
      s0 = fma(a0.x, b0.x, s0); s1 = fma(a0.y, b0.y, s1);
      t0 = fma(a1.x, b1.x, t0); t1 = fma(a1.y, b1.y, t1);
      s0 = fma(a2.x, b2v.x, s0); s1 = fma(a2.y, b2v.y, s1);
      t0 = fma(a3.x, b3.x, t0); t1 = fma(a3.y, b3.y, t1);
    }
    for (; j < n / 2; j += 64) {
      const double2 a = s2[j], bb = b2[j];
      s0 = fma(a.x, bb.x, s0);
      s1 = fma(a.y, bb.y, s1);
    }
    s0 += t0; s1 += t1;
  } else {
    for (int j = lane; j < n; j += 64) s0 = fma(srow[j], b[j], s0);
  }
  return s0 + s1;
}

template <bool NT>
__global__ __launch_bounds__(256) void gemv_rows_kernel(int n, int row0, int row1, const double *__restrict__ S,
                                                        const double *__restrict__ b, double *__restrict__ y) {
  const int row = row0 + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= row1) return;
  const int lane = threadIdx.x & 63;
  const double r = wave_sum(gemv_row_dot<NT>(n, S + (size_t)row * n, b, lane));
  if (lane == 0) y[row] = r;
}

// GEMV with the charge write in its tail (fix_conp.cpp:1135-1158 in ONE launch; plain `fix conp`, all rows on this rank):
//   every wave: y[row] = S[row,:] . b  (same loop, same association as gemv_rows_kernel), then q = y + dV * setq (+ qinit) into
//   q_ele[row] and into every owned / ghost copy of that electrode atom (CSR row -> atoms).
// Bit-identical to gemv_rows_kernel + charge_finish_kernel's charge part; one launch and one dependent-launch gap fewer per
// update.  The fix scalar's group-1 sum of y (:1150, :1159) is formed when somebody asks for it (left_sum_kernel): a first
// version made the last block to finish compute it in this launch -- 1024 ticket adds on one address plus the fences cost
// more than the launch they saved (gemv 23 + finish 7 us -> 38 us fused).
template <bool NT>
__global__ __launch_bounds__(256) void gemv_finish_kernel(int n, const double *__restrict__ S, const double *__restrict__ b,
                                                          double *__restrict__ y, const double *__restrict__ elesetq,
                                                          const double *__restrict__ eleinitq, double potdiff,
                                                          const int *__restrict__ atoms_ptr, const int *__restrict__ atoms_of,
                                                          double *__restrict__ q_ele, double *__restrict__ q_atoms) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  double r = wave_sum(gemv_row_dot<NT>(n, S + (size_t)row * n, b, lane));
  r = __shfl(r, 0, 64);
  double v;
  {
#pragma clang fp contract(off)
    v = r + potdiff * elesetq[row];
    if (eleinitq) v += eleinitq[row];
  }
  if (lane == 0) { y[row] = r; q_ele[row] = v; }
  if (q_atoms)
    for (int k = atoms_ptr[row] + lane; k < atoms_ptr[row + 1]; k += 64) q_atoms[atoms_of[k]] = v;
}

void launch_gemv_finish(hipStream_t s, int n, const double *S, const double *b, double *y, const double *elesetq,
                        const double *eleinitq, double potdiff, const int *atoms_ptr, const int *atoms_of, double *q_ele,
                        double *q_atoms) {
  if (n <= 0) return;
  if (gemv_nt((size_t)n * n * sizeof(double)))
    hipLaunchKernelGGL(gemv_finish_kernel<true>, dim3((n + 3) / 4), dim3(256), 0, s, n, S, b, y, elesetq, eleinitq, potdiff, atoms_ptr,
                       atoms_of, q_ele, q_atoms);
  else
    hipLaunchKernelGGL(gemv_finish_kernel<false>, dim3((n + 3) / 4), dim3(256), 0, s, n, S, b, y, elesetq, eleinitq, potdiff, atoms_ptr,
                       atoms_of, q_ele, q_atoms);
}


// ---- GEMV with the projected inverse as a SYMMETRIC matrix: half the bytes ---------------------------------------------------
// S = A^-1 - (A^-1 e)(A^-1 e)^T / (e^T A^-1 e) is symmetric (A is; fix_conp.cpp:826-831, :982-1067); as computed it is symmetric up
// to rounding (~1e-16 relative).  The product below uses its LOWER triangle for both halves -- the matrix (S_lower + S_lower^T -
// diag), equal to S within that rounding -- packed once per run as 128 x 128 tiles (tile (bi, bj <= bi) at bi (bi + 1) / 2 + bj,
// row-major inside; the diagonal tiles hold both mirrored halves): 4 Ne^2 bytes per update instead of 8 Ne^2.
// The packing pass also measures how symmetric the matrix IS (max |S_ij - S_ji| against max |S_ij|): a matrix that came from a file
// or from conp_fix_set_matrix need not be, and the reference multiplies full rows (ddot_, fix_conp.cpp:1135-1139) -- the caller
// keeps the row-by-row product for such a matrix instead of symmetrising it silently.
//
// Round 4: the tile never touches LDS.  One workgroup per tile, four wavefronts of 32 rows; lane l of a wavefront loads the columns
// (2 l, 2 l + 1) of each of its rows -- 32 coalesced 16-byte loads per lane, all requested at once, the whole tile in flight:
//   transposed product (rows of block bj): acc_t += S[r][cols] * b_i[r] in registers, b_i[r] wavefront-uniform; the four
//     wavefronts' partial columns are added through 4 KB of LDS in a fixed order;
//   direct product (rows of block bi): p_r = S[r][2l] b_j[2l] + S[r][2l+1] b_j[2l+1], then the 32 row sums over the 64 lanes by a
//     transposing butterfly (a lane keeps the half of its values that carries ITS lane bit and adds the partner's copy: 16 + 8 + 4
//     + 2 + 1 + 1 = 32 exchange-adds instead of 32 x 6) -- lane l ends up with the sum of row l >> 1.
// Round 3 staged the tile through LDS in four 32-row quarters: its 16-byte writes into odd-stride rows cost 3.1 bank-conflict cycles
// per LDS instruction (SQ_LDS_BANK_CONFLICT 1 081 344 on 348 736 instructions) and eight barriers per tile: 16.2 us, 4.3 TB/s.
// Every (row block, source tile) pair owns one slot of yp[nb][ne_pad]: nothing is added across workgroups, the finishing kernel sums
// a row's nb slots in a fixed order -> bitwise reproducible.
// (Tried in round 4: the finish inside this launch -- every workgroup announces itself at a counter per row block behind a
//  __threadfence(), the last arrival of a block sums its slots.  Correct, and 82 us instead of 17.6 for the pair of launches: a
//  device-scope release / acquire on gfx950 writes back and invalidates the XCD's whole L2, 528 workgroups do it one after the
//  other.  A kernel boundary is the cheap device-wide fence here.)
constexpr int SG_T = 128;                 // tile edge
__device__ __forceinline__ unsigned long long f64_bits_abs(double v) { return (unsigned long long)__double_as_longlong(fabs(v)); }
__global__ __launch_bounds__(256) void sym_pack_kernel(int ne, const double *__restrict__ S, double *__restrict__ Spk,
                                                       unsigned long long *__restrict__ stat /*[2]: bits of max |S_ij|, max |S_ij - S_ji|*/) {
  int t = blockIdx.x, bi = 0;
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  const int bj = t - bi * (bi + 1) / 2;
  double *out = Spk + (size_t)t * SG_T * SG_T;
  double mx = 0.0, md = 0.0;
  for (int e = threadIdx.x; e < SG_T * SG_T; e += 256) {
    const int r = e >> 7, c = e & 127;
    int i = bi * SG_T + r, j = bj * SG_T + c;
    if (j > i) { const int k = i; i = j; j = k; }            // diagonal tile, upper half: the mirrored lower element
    double v = 0.0;
    if (i < ne && j < ne) {
      v = S[(size_t)i * ne + j];
      const double u = S[(size_t)j * ne + i];
      mx = fmax(mx, fmax(fabs(v), fabs(u)));
      md = fmax(md, fabs(v - u));
    }
    out[e] = v;
  }
  // (non-negative doubles order like their bit patterns: an integer atomicMax is exact and order-independent)
  for (int off = 32; off > 0; off >>= 1) { mx = fmax(mx, __shfl_down(mx, off, 64)); md = fmax(md, __shfl_down(md, off, 64)); }
  if ((threadIdx.x & 63) == 0) { atomicMax(stat, f64_bits_abs(mx)); atomicMax(stat + 1, f64_bits_abs(md)); }
}

// sum over the 64 lanes of each of the 16 values of `p`: lane l returns the total of p[l >> 2] (the four lanes of a quad hold it).
// Fixed association: bitwise reproducible.
__device__ __forceinline__ double wave_sum16_transposed(double (&p)[16], unsigned lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bool hi = lane & 32;
    const double keep = hi ? p[8 + i] : p[i], send = hi ? p[i] : p[8 + i];
    p[i] = keep + __shfl_xor(send, 32, 64);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool hi = lane & 16;
    const double keep = hi ? p[4 + i] : p[i], send = hi ? p[i] : p[4 + i];
    p[i] = keep + __shfl_xor(send, 16, 64);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const bool hi = lane & 8;
    const double keep = hi ? p[2 + i] : p[i], send = hi ? p[i] : p[2 + i];
    p[i] = keep + __shfl_xor(send, 8, 64);
  }
  {
    const bool hi = lane & 4;
    const double keep = hi ? p[1] : p[0], send = hi ? p[0] : p[1];
    p[0] = keep + __shfl_xor(send, 4, 64);
  }
  p[0] += __shfl_xor(p[0], 2, 64);
  return p[0] + __shfl_xor(p[0], 1, 64);
}

// (Cache policy, measured in round 4: non-temporal loads of the packed tiles make the pair of launches 19.6 instead of 18.1 us --
//  67 MB of packed matrix stay partly resident in the Infinity Cache between updates, unlike the 134 MB of the full one above.)
// (two passes of 16 rows per wavefront, the second pass's loads requested when the first pass's values have been used: ~110
//  registers, four workgroups per CU -- all 528 tiles of the headline size resident at once.  With all 32 rows of a wavefront in flight
//  the kernel needs 204 registers: two workgroups per CU, 512 slots for 528 tiles, a second round for sixteen of them.)
__global__ __launch_bounds__(256, 4) void sym_gemv_kernel(int ne, int ne_pad, const double *__restrict__ Spk, const double *__restrict__ b,
                                                          double *__restrict__ yp /*[nb][ne_pad]*/) {
  __shared__ double bi_s[SG_T], bj_s[SG_T];
  __shared__ double tr[4][SG_T];
  int t = blockIdx.x, bi = 0;
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  const int bj = t - bi * (bi + 1) / 2;
  const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double2 *tile = reinterpret_cast<const double2 *>(Spk + (size_t)t * SG_T * SG_T) + (size_t)(32 * wave) * 64 + lane;
  double2 v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = tile[r * 64];
  // b beyond Ne reads as zero: a bound b vector holds Ne entries only (conp_fix_bind_device_buffers), the packed tiles are zero
  // there, and 0 * (whatever lies behind the host's buffer) must not become NaN
  {
    const int i = (tid < SG_T ? bi : bj) * SG_T + (int)(tid & (SG_T - 1));
    (tid < SG_T ? bi_s : bj_s)[tid & (SG_T - 1)] = i < ne ? b[i] : 0.0;
  }
  __syncthreads();
  // (both b blocks come out of LDS behind the barrier: with b_j in registers the compiler forms the direct products above the barrier
  //  and has to carry the loaded rows across it for the transposed ones -- spills)
  const double bj0 = bj_s[2 * lane], bj1 = bj_s[2 * lane + 1];
  double at0 = 0.0, at1 = 0.0;
  double *ydir = yp + (size_t)bj * ne_pad + bi * SG_T + 32 * wave + (lane >> 2);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    double p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = fma(v[r].y, bj1, v[r].x * bj0);
      const double br = bi_s[32 * wave + 16 * h + r];      // wavefront-uniform: one broadcast read
      at0 = fma(v[r].x, br, at0);
      at1 = fma(v[r].y, br, at1);
    }
    if (h == 0) {
      // the second pass's loads reuse the first pass's registers: the offset is made opaque BEHIND the first pass's last use of
      // them (left alone the compiler requests all 32 rows at the top and spills)
      unsigned off2 = 16 * 64;
      asm volatile("" : "+v"(off2) : "v"(at0), "v"(at1));
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = tile[off2 + r * 64];
    }
    const double rs = wave_sum16_transposed(p, lane);
    if ((lane & 3) == 0) ydir[16 * h] = rs;
  }
  if (bi != bj) {
    tr[wave][2 * lane] = at0; tr[wave][2 * lane + 1] = at1;
    __syncthreads();
    if (tid < SG_T) yp[(size_t)bi * ne_pad + bj * SG_T + tid] = (tr[0][tid] + tr[1][tid]) + (tr[2][tid] + tr[3][tid]);
  }
}

// y[row] = sum of the row's nb slots (fixed order), q = y + dV setq (+ qinit); then the charge write of gemv_finish_kernel's tail:
// the block's rows own a contiguous run of the CSR atom list, walked by all threads (row of an entry: atoms_row).
// Four threads per row: thread g adds the slots g, g + 4, g + 8, ... in order -- exactly the partial sum s4[g] of the 4-wide loop the
// one-thread form ran (round 3), combined as (s0 + s1) + (s2 + s3): the same bits, a quarter of the dependent loads per thread.
constexpr int SF_R = 32;                  // rows per finishing block: 128 blocks of 128 threads at Ne = 4096
__global__ __launch_bounds__(4 * SF_R) void sym_finish_kernel(int n, int ne_pad, int nb, const double *__restrict__ yp, double *__restrict__ y,
                                                         const double *__restrict__ elesetq, const double *__restrict__ eleinitq,
                                                         double potdiff, const int *__restrict__ atoms_ptr,
                                                         const int *__restrict__ atoms_of, const int *__restrict__ atoms_row,
                                                         double *__restrict__ q_ele, double *__restrict__ q_atoms) {
  __shared__ double part[4][SF_R];
  __shared__ double vq[SF_R];
  const int rl = threadIdx.x & (SF_R - 1), g = threadIdx.x / SF_R;
  const int row0 = blockIdx.x * SF_R, row = row0 + rl;
  const int rend = row0 + SF_R < n ? row0 + SF_R : n;
  // the block's run of the atom list and this row's constants are requested before the slots: their latency hides behind the sums
  int k0 = 0, k1 = 0;
  if (q_atoms) { k0 = atoms_ptr[row0]; k1 = atoms_ptr[rend]; }
  double sq = 0.0, iq = 0.0;
  if (g == 0 && row < n) { sq = elesetq[row]; if (eleinitq) iq = eleinitq[row]; }
  double s = 0.0;
  if (row < n) {
    for (int kb = g; kb < nb; kb += 32) {              // eight slots in flight per thread
      double w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = kb + 4 * u < nb ? yp[(size_t)(kb + 4 * u) * ne_pad + row] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += w[u];
    }
  }
  part[g][rl] = s;
  __syncthreads();
  if (g == 0 && row < n) {
    const double r = (part[0][rl] + part[1][rl]) + (part[2][rl] + part[3][rl]);
    double v;
    {
#pragma clang fp contract(off)
      v = r + potdiff * sq;
      if (eleinitq) v += iq;
    }
    y[row] = r; q_ele[row] = v;
    vq[rl] = v;
  }
  __syncthreads();
  if (!q_atoms) return;
  for (int k = k0 + (int)threadIdx.x; k < k1; k += 4 * SF_R) q_atoms[atoms_of[k]] = vq[atoms_row[k] - row0];
}

size_t sym_packed_doubles(int ne_pad) { const size_t nb = ne_pad / SG_T; return nb * (nb + 1) / 2 * SG_T * SG_T; }
void launch_sym_pack(hipStream_t s, int ne, int ne_pad, const double *S, double *Spk, unsigned long long *stat) {
  const int nb = ne_pad / SG_T;
  (void)hipMemsetAsync(stat, 0, 2 * sizeof(unsigned long long), s);
  hipLaunchKernelGGL(sym_pack_kernel, dim3(nb * (nb + 1) / 2), dim3(256), 0, s, ne, S, Spk, stat);
}
void launch_sym_gemv_finish(hipStream_t s, int n, int ne_pad, const double *Spk, const double *b, double *yp, double *y,
                            const double *elesetq, const double *eleinitq, double potdiff, const int *atoms_ptr, const int *atoms_of,
                            const int *atoms_row, double *q_ele, double *q_atoms) {
  const int nb = ne_pad / SG_T;
  hipLaunchKernelGGL(sym_gemv_kernel, dim3(nb * (nb + 1) / 2), dim3(256), 0, s, n, ne_pad, Spk, b, yp);
  hipLaunchKernelGGL(sym_finish_kernel, dim3((n + SF_R - 1) / SF_R), dim3(4 * SF_R), 0, s, n, ne_pad, nb, (const double *)yp, y, elesetq, eleinitq,
                     potdiff, atoms_ptr, atoms_of, atoms_row, q_ele, q_atoms);
}

void launch_gemv_rows(hipStream_t s, int n, int row0, int row1, const double *S, const double *b, double *y) {
  if (row1 <= row0) return;
  if (gemv_nt((size_t)(row1 - row0) * n * sizeof(double)))
    hipLaunchKernelGGL(gemv_rows_kernel<true>, dim3((row1 - row0 + 3) / 4), dim3(256), 0, s, n, row0, row1, S, b, y);
  else
    hipLaunchKernelGGL(gemv_rows_kernel<false>, dim3((row1 - row0 + 3) / 4), dim3(256), 0, s, n, row0, row1, S, b, y);
}

// fix_conp.cpp:1149-1159 in one launch:
//   blocks 0 .. nb-2 : q_ele[e] = eleallq[e] + dV * elesetq[e] (+ eleinitq[e]) for e < ne, and the same value written to
//                      every owned or ghost electrode atom: a compact list of (atom index, eleall index) pairs, so that the
//                      launch covers the electrode atoms, not all nall atoms
//   last block       : netcharge_left = sum of eleallq over the group-1 atoms (fixed tree)
__global__ __launch_bounds__(256) void charge_finish_kernel(int ne, int nall, const int *__restrict__ atom2eleall,
                                                            const int *__restrict__ elecheck, const double *__restrict__ eleallq,
                                                            const double *__restrict__ elesetq, const double *__restrict__ eleinitq,
                                                            double potdiff, const double *__restrict__ d_potdiff,
                                                            double *__restrict__ q_ele,
                                                            double *__restrict__ q_atoms, double *__restrict__ left_out) {
#pragma clang fp contract(off)
  if (d_potdiff) potdiff = *d_potdiff;      // fix conq: the potential difference was derived on the device
  if (blockIdx.x == gridDim.x - 1) {
    if (!left_out) return;
    __shared__ double red[4];
    // this one block is the kernel's critical path: 16 independent loads in flight per thread (Ne = 4096: one round);
    // fixed association -> reproducible
    double s16[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) s16[u] = 0.0;
    for (int i0 = threadIdx.x; i0 < ne; i0 += 4096) {
      int ec[16];
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = i0 + 256 * u;
        ec[u] = i < ne ? elecheck[i] : 0;
        v[u] = i < ne ? eleallq[i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s16[u] += (ec[u] == 1) ? v[u] : 0.0;
    }
    double s4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) s4[u] = (s16[u] + s16[u + 4]) + (s16[u + 8] + s16[u + 12]);
    double s = wave_sum((s4[0] + s4[1]) + (s4[2] + s4[3]));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *left_out = (red[0] + red[1]) + (red[2] + red[3]);
    return;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < ne) {
    double v = eleallq[i] + potdiff * elesetq[i];
    if (eleinitq) v += eleinitq[i];
    q_ele[i] = v;
  }
  if (q_atoms && i < nall) {               // nall = number of owned + ghost ELECTRODE atoms; atom2eleall = (atom, eleall) pairs
    const int a = atom2eleall[2 * i], e = atom2eleall[2 * i + 1];
    double v = eleallq[e] + potdiff * elesetq[e];
    if (eleinitq) v += eleinitq[e];
    q_atoms[a] = v;
  }
}

void launch_charge_finish(hipStream_t s, int ne, int nall, const int *atom2eleall, const int *elecheck, const double *eleallq,
                          const double *elesetq, const double *eleinitq, double potdiff, const double *d_potdiff, double *q_ele,
                          double *q_atoms, double *left_out) {
  const int n = q_atoms ? (nall > ne ? nall : ne) : ne;
  hipLaunchKernelGGL(charge_finish_kernel, dim3((n + 255) / 256 + 1), dim3(256), 0, s, ne, nall, atom2eleall, elecheck, eleallq,
                     elesetq, eleinitq, potdiff, d_potdiff, q_ele, q_atoms, left_out);
}

// fix conq (fix_conq.cpp:74-80): potential difference that yields total charge -Q / +Q:  dV = -(Q + sum_left q) / totsetq
__global__ void conq_potdiff_kernel(const double *__restrict__ left, double rightcharge, double totsetq, int one_electrode,
                                    double *__restrict__ out) {
#pragma clang fp contract(off)
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double netcharge_right = -(*left);
  double v = -(rightcharge - netcharge_right) / totsetq;
  if (one_electrode) v += 2 * rightcharge / totsetq;
  *out = v;
}

// fix cond (fix_cond.cpp:99-116): dV = (Q - dipole/lz - setz . eleallq) * vmult, dipole = - sum_{electrolyte} q z
// (slab_part holds the per-block partial sums of q z written by the phase / spread kernel); one workgroup, fixed trees
__global__ __launch_bounds__(1024) void cond_potdiff_kernel(int ne, const double *__restrict__ setzvec,
                                                            const double *__restrict__ eleallq,
                                                            const double *__restrict__ slab_part, int n_slab_part, double lz,
                                                            double rightcharge, double vmult, double *__restrict__ out) {
#pragma clang fp contract(off)
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < ne; i += 1024) s += setzvec[i] * eleallq[i];
  s = block_sum_1024(s, red);
  double qz = 0.0;
  for (int k = threadIdx.x; k < n_slab_part; k += 1024) qz += slab_part[k];
  qz = block_sum_1024(qz, red);
  if (threadIdx.x == 0) *out = (rightcharge + qz / lz - s) * vmult;     // dipole = -qz
}

void launch_cond_potdiff(hipStream_t s, int ne, const double *setzvec, const double *eleallq, const double *slab_part,
                         int n_slab_part, double lz, double rightcharge, double vmult, double *out) {
  hipLaunchKernelGGL(cond_potdiff_kernel, dim3(1), dim3(1024), 0, s, ne, setzvec, eleallq, slab_part, n_slab_part, lz, rightcharge,
                     vmult, out);
}

void launch_conq_potdiff(hipStream_t s, const double *left, double rightcharge, double totsetq, int one_electrode, double *out) {
  hipLaunchKernelGGL(conq_potdiff_kernel, dim3(1), dim3(64), 0, s, left, rightcharge, totsetq, one_electrode, out);
}

// ---- post-force real-space correction (fix_conp.cpp:1368-1444) ----------------------------------------------------
__device__ __forceinline__ double ferfcr_sqrt_dev(double a2_r2) {
#pragma clang fp contract(off)
  if (a2_r2 < 5.8 * 5.8) {
    const double a_r = sqrt(a2_r2);
    const double expm2 = exp(-a2_r2);
    const double t = 1.0 / (1.0 + 0.3275911 * a_r);
    const double erfcr = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429)))) * expm2 / a_r;
    return erfcr + 1.12837917 * expm2;
  }
  return 0.0;
}

// one thread per listed (owner i, neighbour j) pair with exactly one electrode member; acc[0] eng_coul, acc[1..6] virial
__device__ __forceinline__ void post_force_pair(int i, int j, bool ei, int nlocal, int newton, const double *__restrict__ x,
                                                const double *__restrict__ q, const int *__restrict__ type,
                                                const int *__restrict__ atom2eleall, const RealParams &rp, double qqrd2e,
                                                double *__restrict__ f, double *__restrict__ acc) {
#pragma clang fp contract(off)
  if (ei == (atom2eleall[j] >= 0)) return;                         // exactly one electrode member
  const double delx = x[3 * i] - x[3 * j], dely = x[3 * i + 1] - x[3 * j + 1], delz = x[3 * i + 2] - x[3 * j + 2];
  const double rsq = delx * delx + dely * dely + delz * delz;
  if (!(rsq < rp.cutsq[type[i] * (rp.ntypes + 1) + type[j]])) return;
  const double etarij2 = rp.eta * rp.eta * rsq;
  if (!(etarij2 < 5.8)) return;                                  // :1419 (ERFC_MAX, not its square -- kept as written)
  atomicAdd(&acc[8], 1.0);                                         // pairs inside the Gaussian overlap range: usually none
  const bool eleilocal = atom2eleall[i] >= 0;
  const double prefactor = qqrd2e * q[i] * q[j];
  double fterm;
  if (rp.ehgo) {                                                  // ehgo_force :1568-1573
    const double etaij = rp.eta_ij[type[i] * (rp.ntypes + 1) + type[j]], foij = rp.fo_ij[type[i] * (rp.ntypes + 1) + type[j]];
    const double e2 = etaij * etaij * rsq;
    fterm = e2 * foij * exp(-0.5 * e2) - ferfcr_sqrt_dev(e2) * etaij;
  } else fterm = -ferfcr_sqrt_dev(etarij2) * rp.eta;             // eta_force :1477-1480
  const double forcecoul = prefactor * fterm;
  const double fpair = forcecoul / rsq;
  if (!eleilocal) {
    atomicAdd(&f[3 * i], delx * forcecoul); atomicAdd(&f[3 * i + 1], dely * forcecoul); atomicAdd(&f[3 * i + 2], delz * forcecoul);
  } else if (newton || j < nlocal) {
    atomicAdd(&f[3 * j], -(delx * forcecoul)); atomicAdd(&f[3 * j + 1], -(dely * forcecoul)); atomicAdd(&f[3 * j + 2], -(delz * forcecoul));
  }
  const double ecoul = prefactor * pair_potential_dev(rp, rsq, type[i], type[j], false);
  double w = 0.0;
  if (newton) w = 1.0;
  else { if (i < nlocal) w += 0.5; if (j < nlocal) w += 0.5; }
  atomicAdd(&acc[0], w * ecoul);
  atomicAdd(&acc[1], w * (delx * delx * fpair)); atomicAdd(&acc[2], w * (dely * dely * fpair)); atomicAdd(&acc[3], w * (delz * delz * fpair));
  atomicAdd(&acc[4], w * (delx * dely * fpair)); atomicAdd(&acc[5], w * (delx * delz * fpair)); atomicAdd(&acc[6], w * (dely * delz * fpair));
}


// One wavefront per list owner walks that atom's neighbours straight from the (flattened) LAMMPS half list: every listed pair
// with exactly one electrode member takes part (blist_coul_cal_post_force has no newton / ghost filter, fix_conp.cpp:1411), so
// no pair list has to be compacted on the host at a re-neighbour.
__global__ __launch_bounds__(256) void post_force_kernel(int inum, const int *__restrict__ ilist, const int *__restrict__ numneigh,
                                                         const int *__restrict__ first, const int *__restrict__ neigh,
                                                         int nlocal, int newton, const double *__restrict__ x,
                                                         const double *__restrict__ q, const int *__restrict__ type,
                                                         const int *__restrict__ atom2eleall, RealParams rp, double qqrd2e,
                                                         double *__restrict__ f, double *__restrict__ acc) {
#pragma clang fp contract(off)
  const int ii = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ii >= inum) return;
  const int i = ilist[ii];
  const int *jl = neigh + first[i];
  const int jn = numneigh[i];
  const bool ei = atom2eleall[i] >= 0;
  for (int jj = threadIdx.x & 63; jj < jn; jj += 64) post_force_pair(i, jl[jj] & 0x3FFFFFFF, ei, nlocal, newton, x, q, type, atom2eleall, rp, qqrd2e, f, acc);
}

// Gaussian self energy sum over owned electrode atoms of q^2 (fix_conp.cpp:1167-1181), one workgroup, fixed tree
__global__ __launch_bounds__(1024) void ele_qsq_kernel(int nlocal, const int *__restrict__ atom2eleall, const double *__restrict__ q,
                                                       const int *__restrict__ type, const double *__restrict__ u0_i,
                                                       double *__restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < nlocal; i += 1024)
    if (atom2eleall[i] >= 0) s += u0_i ? u0_i[type[i]] * q[i] * q[i] : q[i] * q[i];       // EHGO weights by u0 of the type (:1189)
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0.0; for (int k = 0; k < 16; ++k) t += red[k]; *out = t; }
}

void launch_post_force(hipStream_t s, int inum, const int *ilist, const int *numneigh, const int *first, const int *neigh, int nlocal,
                       int nall, int newton, const double *x,
                       const double *q, const int *type, const int *atom2eleall, RealParams rp, double qqrd2e, double *f,
                       double *acc /*[9]: eng_coul, virial[6], qsq, number of contributing pairs*/, bool clear_f) {
  if (clear_f) (void)hipMemsetAsync(f, 0, (size_t)nall * 3 * sizeof(double), s);
  (void)hipMemsetAsync(acc, 0, 9 * sizeof(double), s);
  if (inum > 0)
    hipLaunchKernelGGL(post_force_kernel, dim3((inum + 3) / 4), dim3(256), 0, s, inum, ilist, numneigh, first, neigh, nlocal, newton,
                       x, q, type, atom2eleall, rp, qqrd2e, f, acc);
  hipLaunchKernelGGL(ele_qsq_kernel, dim3(1), dim3(1024), 0, s, nlocal, atom2eleall, q, type, rp.ehgo ? rp.u0_i : nullptr, acc + 7);
}

// sum of v over the group-1 ("left") electrode atoms (totsetq :1098-1104); one workgroup
__global__ __launch_bounds__(1024) void left_sum_kernel(int ne, const int *__restrict__ elecheck,
                                                        const double *__restrict__ v, double *__restrict__ out) {
  __shared__ double red[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < ne; i += 1024) if (elecheck[i] == 1) s += v[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int k = 0; k < 16; ++k) tot += red[k];
    *out = tot;
  }
}

void launch_left_sum(hipStream_t s, int ne, const int *elecheck, const double *v, double *out) {
  hipLaunchKernelGGL(left_sum_kernel, dim3(1), dim3(1024), 0, s, ne, elecheck, v, out);
}

// The end of a host-buffer update: the group-1 sum (left_sum_kernel's, same tree) AND the results on their way to the host --
// the electrode charges and the four scalars are stored straight into page-locked host memory by this kernel (posted PCIe writes,
// 32 KB at Ne = 4096) instead of two copy-engine transfers behind it (each ~10 us of latency on this runtime).
__global__ __launch_bounds__(1024) void results_out_kernel(int ne, const int *__restrict__ elecheck, const double *__restrict__ v,
                                                           double *__restrict__ scal /*[4] device*/, int do_left,
                                                           const double *__restrict__ qele, double *__restrict__ host_q,
                                                           double *__restrict__ host_scal /*[4]*/) {
  __shared__ double red[16];
  if (do_left) {
    double s = 0.0;
    for (int i = threadIdx.x; i < ne; i += 1024) if (elecheck[i] == 1) s += v[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0;
      for (int k = 0; k < 16; ++k) tot += red[k];
      scal[1] = tot;
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) host_scal[threadIdx.x] = scal[threadIdx.x];
  for (int i = threadIdx.x; i < ne; i += 1024) host_q[i] = qele[i];
}
void launch_results_out(hipStream_t s, int ne, const int *elecheck, const double *v, double *scal, bool do_left, const double *qele,
                        double *host_q, double *host_scal) {
  hipLaunchKernelGGL(results_out_kernel, dim3(1), dim3(1024), 0, s, ne, elecheck, v, scal, do_left ? 1 : 0, qele, host_q, host_scal);
}

// ================================================================================================
// 6. once-per-run: Ewald A matrix.
//    k-space (km_ewald.cpp:584-645):  A_ij = sum_{r,t} w(r,t) Rp[r][i] Tz[t][i] Rp[r][j] Tz[t][j]   for i > j
//    -- a SYRK over the (r,t) index on the FP64 matrix cores.  Workgroup = 4 waves = 128 x 128 tile, wave = 64 x 64
//    (4 x 4 fragments); operands are formed in registers from Rp (global) and Tz (LDS): A = w * Rp_i * Tz_i, B = Rp_j * Tz_j.
//    Only tiles with (row block >= col block) run; the strict upper triangle is left for a_symmetrise.
// ================================================================================================
// The Tz operand is staged through LDS.  (A first version formed both operands from global loads: the tile's Tz slab, 320 x
// 256 doubles, does not fit in LDS and was re-read from L2 once per G row -- 9 global loads per 16 MFMAs, MFMA pipe 49 % busy,
// 103 ms at Ne = 4096.)  The (r, t) loop nest is turned inside out instead: a 32-t chunk of Tz (= one 16-kz block = 8
// k-steps) for the tile's 128 + 128 atoms stays in LDS while r runs over every G row whose sphere cut reaches that block
// (row tiles are rings sorted by |k_p|: in practice a prefix of the rows); per r: 8 Rp + 8 w global loads for
// 128 MFMAs.  LDS row stride 272 doubles: the two t rows a half-wave reads land in disjoint bank halves.
constexpr int AK_TC = 32;          // t rows per chunk
constexpr int AK_LD = 272;         // 256 atoms + 16 pad
__global__ __launch_bounds__(256, 2) void a_kspace_lds_kernel(int R_pad, int C_pad, int ne, int ne_pad, int n_row_tiles,
                                                              const int *__restrict__ nb_act, const double *__restrict__ wfull,
                                                              const double *__restrict__ Rp, const double *__restrict__ Tz,
                                                              double *__restrict__ A, int nsplit,
                                                              const int *__restrict__ chunk_group, size_t part_stride,
                                                              int tile_first, int tile_stride) {
  extern __shared__ __attribute__((aligned(16))) char ak_smem[];
  double *L = reinterpret_cast<double *>(ak_smem);      // [AK_TC][AK_LD]
  // unit = (tile, split s): the kz chunks are dealt to `nsplit` groups of about equal work on the host; split s sums its group
  // into its own copy of the tile (A + s * part_stride), a_parts_sum adds the copies in a fixed order.  More, smaller units:
  // 528 tiles on 512 workgroup slots (Ne = 4096) leave a 30 % tail otherwise.
  const int split = blockIdx.x % nsplit;
  A += (size_t)split * part_stride;
  // several ranks: the lower-triangle tiles are dealt cyclically (rank r takes tiles r, r + nranks, ...; equal cost apart from
  // the half-empty diagonal tiles), everybody's zero-initialised matrix is then summed (conp_fix.cpp allreduce_matrix)
  int tidx = (blockIdx.x / nsplit) * tile_stride + tile_first, bi = 0;
  while ((bi + 1) * (bi + 2) / 2 <= tidx) ++bi;
  const int bj = tidx - bi * (bi + 1) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  const int ibase = bi * 128 + wi * 64 + fr, jbase = bj * 128 + wj * 64 + fr;
  const bool idle = (bi == bj && wj > wi);       // wave tile strictly above the diagonal: helps with the staging only
  d4 acc[4][4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[f][g] = (d4){0.0, 0.0, 0.0, 0.0};
  const int nchunk = C_pad / AK_TC;        // 10 per column tile; a chunk nobody reaches belongs to no group (chunk_group = -1)
  const double *li = L + fk * AK_LD + wi * 64 + fr, *lj = L + fk * AK_LD + 128 + wj * 64 + fr;
  for (int c = 0; c < nchunk; ++c) {
    if (chunk_group[c] != split) continue;
    __syncthreads();
    for (int e = threadIdx.x; e < AK_TC * 256; e += 256) {
      const int t = e >> 8, a = e & 255;
      const int atom = (a < 128) ? bi * 128 + a : bj * 128 + (a - 128);
      L[t * AK_LD + a] = Tz[(size_t)(AK_TC * c + t) * ne_pad + atom];
    }
    __syncthreads();
    if (idle) continue;
    const int *nbc = nb_act + (c / 10) * n_row_tiles;
    const int cl = c % 10;
    for (int r = 0; r < R_pad; ++r) {
      if (nbc[r >> 7] <= cl) { r |= 127; continue; }        // this row tile's sphere cut ends before kz block cl of the column tile
      double ri[4], rj[4], ww[8];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        ri[f] = Rp[(size_t)r * ne_pad + ibase + 16 * f];
        rj[f] = Rp[(size_t)r * ne_pad + jbase + 16 * f];
      }
      const double *wrow = wfull + (size_t)r * C_pad + AK_TC * c + fk;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) ww[ks] = wrow[4 * ks];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        double af[4], bf[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          af[f] = ww[ks] * ri[f] * li[(4 * ks) * AK_LD + 16 * f];
          bf[f] = rj[f] * lj[(4 * ks) * AK_LD + 16 * f];
        }
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[f][g] = MFMA_F64(af[f], bf[g], acc[f][g]);
      }
    }
  }
  if (idle) return;
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = bi * 128 + wi * 64 + 16 * f + fk + 4 * r;
        const int j = bj * 128 + wj * 64 + 16 * g + fr;
        if (i < ne && j < i) A[(size_t)i * ne + j] = acc[f][g][r];
      }
}

// ---- planar electrodes (few distinct electrode z values, the z classes of the projection's fast path): the k-space part of A
// factorises the same way b does.  Tz[t][i] depends on the atom's class only, so
//     A_ij = sum_r Rp[r][i] Rp[r][j] W_r[c(i)][c(j)],      W_r[c][c'] = sum_t w(r,t) Tzc[t][c] Tzc[t][c']       (R_pad x nzc^2, tiny)
// -- a contraction over the 2 n_p planar rows instead of over all (planar, kz) pairs: at the headline size 3.4e10 instead of
// 3.9e12 flop.  As a GEMM: k index = (r, c'), A operand Rp[r][i] W_r[c(i)][c'], B operand Rp[r][j] [c(j) == c'].
// Same tile / rank conventions and the same output (strict lower triangle) as a_kspace_lds_kernel.
__global__ __launch_bounds__(256) void a_wz_kernel(int R_pad, int C_pad, int nzc, const double *__restrict__ wfull,
                                                   const double *__restrict__ Tzc /*[C_pad][64]*/, double *__restrict__ Wz /*[R_pad][nzc][nzc]*/) {
  // one wavefront per (row r, class pair): fixed-order sum over the columns
  const int lane = threadIdx.x & 63;
  const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= R_pad * nzc * nzc) return;
  const int r = u / (nzc * nzc), cc = u - r * nzc * nzc, c = cc / nzc, cp = cc - c * nzc;
  double s = 0.0;
  for (int t = lane; t < C_pad; t += 64) {
    const double w = wfull[(size_t)r * C_pad + t];
    if (w != 0.0) s += w * Tzc[(size_t)t * 64 + c] * Tzc[(size_t)t * 64 + cp];
  }
  s = wave_sum(s);
  if (lane == 0) Wz[u] = s;
}

__global__ __launch_bounds__(256, 2) void a_kspace_zc_kernel(int R_pad, int ne, int ne_pad, int nzc, const double *__restrict__ Wz,
                                                             const double *__restrict__ Rp, const int *__restrict__ zclass,
                                                             double *__restrict__ A, int tile_first, int tile_stride) {
  int tidx = blockIdx.x * tile_stride + tile_first, bi = 0;
  while ((bi + 1) * (bi + 2) / 2 <= tidx) ++bi;
  const int bj = tidx - bi * (bi + 1) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int fr = lane & 15, fk = lane >> 4;
  if (bi == bj && wj > wi) return;               // wave tile strictly above the diagonal
  const int ibase = bi * 128 + wi * 64 + fr, jbase = bj * 128 + wj * 64 + fr;
  int ci[4], cj[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) { ci[f] = zclass[ibase + 16 * f]; cj[f] = zclass[jbase + 16 * f]; }
  d4 acc[4][4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[f][g] = (d4){0.0, 0.0, 0.0, 0.0};
  const int nk = R_pad * nzc;                    // k = r * nzc + c'   (R_pad is a multiple of 128: nk of 4)
#pragma unroll 2
  for (int k0 = 0; k0 < nk; k0 += 4) {
    const int k = k0 + fk, r = k / nzc, cp = k - r * nzc;
    const double *rp = Rp + (size_t)r * ne_pad;
    const double *wz = Wz + (size_t)r * nzc * nzc + cp;
    double af[4], bf[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      af[f] = rp[ibase + 16 * f] * wz[ci[f] * nzc];
      bf[f] = cj[f] == cp ? rp[jbase + 16 * f] : 0.0;
    }
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
        const int i = bi * 128 + wi * 64 + 16 * f + fk + 4 * r;
        const int j = bj * 128 + wj * 64 + 16 * g + fr;
        if (i < ne && j < i) A[(size_t)i * ne + j] = acc[f][g][r];
      }
}

void launch_a_kspace_zclass(hipStream_t s, const DevPlan &pl, int ne, int ne_pad, int nzc, const double *Rp, const double *Tzc,
                            const int *zclass, double *Wz /*[R_pad * nzc * nzc] scratch*/, double *A, int rank, int nranks) {
  const int nb = ne_pad / 128;
  const int ntiles_all = nb * (nb + 1) / 2;
  const int ntiles = ntiles_all > rank ? (ntiles_all - rank + nranks - 1) / nranks : 0;
  const int nu = pl.R_pad * nzc * nzc;
  hipLaunchKernelGGL(a_wz_kernel, dim3((nu + 3) / 4), dim3(256), 0, s, pl.R_pad, pl.C_pad, nzc, pl.wfull, Tzc, Wz);
  if (ntiles > 0)
    hipLaunchKernelGGL(a_kspace_zc_kernel, dim3(ntiles), dim3(256), 0, s, pl.R_pad, ne, ne_pad, nzc, (const double *)Wz, Rp, zclass, A, rank,
                       nranks);
}

// A (strict lower triangle) = fixed-order sum of the split copies: ((A0 + A1) + (A2 + A3)), copies 1.. in `parts`
__global__ void a_parts_sum_kernel(int ne, int nsplit, double *__restrict__ A, const double *__restrict__ parts) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n2 = (size_t)ne * ne;
  if (e >= n2) return;
  const int i = (int)(e / ne), j = (int)(e % ne);
  if (j >= i) return;
  double v = A[e];
  if (nsplit == 2) v += parts[e];
  else if (nsplit == 4) v = (v + parts[e]) + (parts[n2 + e] + parts[2 * n2 + e]);
  A[e] = v;
}

// how many ways the kz chunks of a tile are split: none when there are many more tiles than workgroup slots
int a_kspace_nsplit(int ne_pad, int num_cus, int nchunk, int nranks) {
  const int nb = ne_pad / 128;
  const long ntiles = ((long)nb * (nb + 1) / 2 + nranks - 1) / nranks, slots = 2L * num_cus;
  if (ntiles >= 6 * slots || nchunk < 2) return 1;
  return nchunk >= 4 ? 4 : 2;
}

// A must have room for nsplit copies of ne * ne doubles; copy s of a tile goes to A + s * ne * ne, the sum lands in copy 0
void launch_a_kspace(hipStream_t s, const DevPlan &pl, int ne, int ne_pad, const double *Rp, const double *Tz, double *A, int nsplit,
                     const int *chunk_group, int rank, int nranks) {
  const int nb = ne_pad / 128;
  const int ntiles_all = nb * (nb + 1) / 2;
  const int ntiles = ntiles_all > rank ? (ntiles_all - rank + nranks - 1) / nranks : 0;
  const size_t lds = (size_t)AK_TC * AK_LD * sizeof(double);
  static DynLdsCache granted{};
  ensure_dyn_lds(a_kspace_lds_kernel, lds, granted);
  const size_t n2 = (size_t)ne * ne;
  if (ntiles > 0)
    hipLaunchKernelGGL(a_kspace_lds_kernel, dim3(ntiles * nsplit), dim3(256), lds, s, pl.R_pad, pl.C_pad, ne, ne_pad, pl.n_row_tiles,
                       pl.nb_act, pl.wfull, Rp, Tz, A, nsplit, chunk_group, n2, rank, nranks);
  if (nsplit > 1) hipLaunchKernelGGL(a_parts_sum_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, ne, nsplit, A, A + n2);
}

// diagonal ug_tot - 2g/sqrt(pi) + sqrt(2) eta/sqrt(pi) (km_ewald.cpp:631-634, fix_conp.cpp:796-801) and the slab
// term 4 pi z_i z_j / V on j <= i (km_ewald.cpp:647-665)
__global__ void a_diag_slab_kernel(int ne, double diag_k, double diag_self, const double *__restrict__ diag_self_atom, int slab,
                                   double pref, const double *__restrict__ ele_z, double *__restrict__ A) {
#pragma clang fp contract(off)
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)ne * ne) return;
  const int i = (int)(e / ne), j = (int)(e % ne);
  if (j > i) return;
  double v = A[e];
  if (i == j) v = diag_k;
  if (slab) v += pref * ele_z[i] * ele_z[j];
  if (i == j) v += diag_self_atom ? diag_self_atom[i] : diag_self;     // sqrt(2) eta / sqrt(pi), or u0 of the atom's type (EHGO :803-810)
  A[e] = v;
}

// one thread per electrode row, pairs in list order (deterministic): A[row][col] += [erfc(g r) - erfc(eta r / sqrt 2)] / r
// (fix_conp.cpp:1242-1276, eta_potential_A :1467-1470)
__global__ void a_real_kernel(int ne, int row0, int row1, const int *__restrict__ row_ptr, const int *__restrict__ ele_atom,
                              const int *__restrict__ oth_atom, const int *__restrict__ col, const double *__restrict__ x,
                              const int *__restrict__ type, RealParams rp, double *__restrict__ A) {
#pragma clang fp contract(off)
  const int row = row0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= row1 || row >= ne) return;
  const int nt1 = rp.ntypes + 1;
  for (int p = row_ptr[row]; p < row_ptr[row + 1]; ++p) {
    const int ie = ele_atom[p], jo = oth_atom[p];
    const double dx = x[3 * ie] - x[3 * jo], dy = x[3 * ie + 1] - x[3 * jo + 1], dz = x[3 * ie + 2] - x[3 * jo + 2];
    const double rsq = dx * dx + dy * dy + dz * dz;
    if (rsq < rp.cutsq[type[ie] * nt1 + type[jo]] && rsq < rp.cut_coulsq) {
      double dudq = erfcr_sqrt_dev(rp.g_ewald * rp.g_ewald * rsq) * rp.g_ewald;
      dudq += pair_potential_dev(rp, rsq, type[ie], type[jo], true);
      A[(size_t)row * ne + col[p]] += dudq;
    }
  }
}

void launch_a_real(hipStream_t s, int ne, int row0, int row1, const int *row_ptr, const int *ele_atom, const int *oth_atom,
                   const int *col, const double *x, const int *type, RealParams rp, double *A) {
  if (row1 <= row0) return;
  hipLaunchKernelGGL(a_real_kernel, dim3((row1 - row0 + 63) / 64), dim3(64), 0, s, ne, row0, row1, row_ptr, ele_atom, oth_atom, col,
                     x, type, rp, A);
}

// a_ij += a_ji ; a_ji = a_ij  for i > j   (fix_conp.cpp:826-831)
__global__ void a_symmetrise_kernel(int ne, double *__restrict__ A) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)ne * ne) return;
  const int i = (int)(e / ne), j = (int)(e % ne);
  if (j >= i) return;
  const double v = A[(size_t)i * ne + j] + A[(size_t)j * ne + i];
  A[(size_t)i * ne + j] = v;
  A[(size_t)j * ne + i] = v;
}

void launch_a_symmetrise(hipStream_t s, int ne, double *A) {
  const size_t n2 = (size_t)ne * ne;
  hipLaunchKernelGGL(a_symmetrise_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, ne, A);
}

void launch_a_diag_slab(hipStream_t s, int ne, double diag_k, double diag_self, const double *diag_self_atom, int slab, double pref,
                        const double *ele_z, double *A) {
  const size_t n2 = (size_t)ne * ne;
  hipLaunchKernelGGL(a_diag_slab_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, ne, diag_k, diag_self, diag_self_atom,
                     slab, pref, ele_z, A);
}

// ================================================================================================
// 7. electroneutrality projection, BIT-EXACT with the reference's operation order (fix_conp.cpp:996-1060):
//    ainve_i = sum_j A_ij in j order (one thread per row, sequential), totinve = sum_i ainve_i in i order (one thread),
//    A_ij -= ainve_i * ainve_j / totinve  evaluated as (ainve_i*ainve_j)/totinve, only if totinve^2 > 1e-8.
//    mask != NULL: the zneutr variant (e restricted to atoms with z > zhalf).
// ================================================================================================
__global__ void inv_rowsum_kernel(int n, const double *__restrict__ A, const unsigned char *__restrict__ mask,
                                  double *__restrict__ ainve) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double *row = A + (size_t)i * n;
  double acc = 0;
  if (mask) { for (int j = 0; j < n; ++j) if (mask[j]) acc += row[j]; }
  else { for (int j = 0; j < n; ++j) acc += row[j]; }
  ainve[i] = acc;
}

__global__ void inv_total_kernel(int n, const double *__restrict__ ainve, const unsigned char *__restrict__ mask,
                                 double *__restrict__ totinve) {
#pragma clang fp contract(off)
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double tot = 0;
  for (int i = 0; i < n; ++i) if (!mask || mask[i]) tot += ainve[i];
  *totinve = tot;
}

__global__ void inv_apply_kernel(int n, double *__restrict__ A, const double *__restrict__ ainve,
                                 const double *__restrict__ totinve) {
#pragma clang fp contract(off)
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)n * n) return;
  const double tot = *totinve;
  if (!(tot * tot > 1e-8)) return;
  const int i = (int)(e / n), j = (int)(e % n);
  const double prod = ainve[i] * ainve[j];
  const double quot = prod / tot;
  A[e] = A[e] - quot;
}

void launch_inv_project(hipStream_t s, int n, double *A, int use_mask, const unsigned char *mask, double *ainve,
                        double *totinve, int apply) {
  const unsigned char *m = use_mask ? mask : nullptr;
  hipLaunchKernelGGL(inv_rowsum_kernel, dim3((n + 63) / 64), dim3(64), 0, s, n, A, m, ainve);
  hipLaunchKernelGGL(inv_total_kernel, dim3(1), dim3(64), 0, s, n, ainve, m, totinve);
  if (apply) launch_inv_project_apply(s, n, A, ainve, totinve);
}

void launch_inv_project_apply(hipStream_t s, int n, double *A, const double *ainve, const double *totinve) {
  const size_t n2 = (size_t)n * n;
  hipLaunchKernelGGL(inv_apply_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, n, A, ainve, totinve);
}

// ================================================================================================
// 8. conjugate gradient with the neutrality constraint (fix_conp.cpp:864-930).  The matvec is gemv_rows; the
//    vector updates + scalars run in ONE workgroup so that every reduction has a fixed order.
//    scal: [0] lresnorm [1] lgamma [2] netr [3] ptap [4] alpha [5] beta [6] converged-iteration [7] net charge at
//          convergence [8] converged flag (0/1), followed in the same buffer by hist[iter] = lresnorm of every iteration
// ================================================================================================
// two sums at once, each with block_sum_1024's tree (same bits), one pair of barriers instead of two
__device__ __forceinline__ void block_sum2_1024(double &a, double &b, double *red /*[32]*/) {
  a = wave_sum(a); b = wave_sum(b);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[16 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  double ta = 0.0, tb = 0.0;
  for (int k = 0; k < 16; ++k) { ta += red[k]; tb += red[16 + k]; }
  a = ta; b = tb;
}
__device__ double block_sum_1024(double v, double *red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double tot = 0.0;
  for (int k = 0; k < 16; ++k) tot += red[k];
  return tot;
}

__global__ __launch_bounds__(1024) void cg_init_kernel(int n, const double *__restrict__ b, double *__restrict__ q,
                                                       double *__restrict__ res, double *__restrict__ p,
                                                       double *__restrict__ scal) {
  __shared__ double red[16];
  double netr = 0, l2 = 0;
  for (int i = threadIdx.x; i < n; i += 1024) { q[i] = 0.0; const double r = b[i]; res[i] = r; netr += r; l2 += r * r; }
  netr = block_sum_1024(netr, red);
  l2 = block_sum_1024(l2, red);
  const double ave = netr / n;
  for (int i = threadIdx.x; i < n; i += 1024) p[i] = res[i] - ave;
  if (threadIdx.x == 0) { const double lres = l2 - netr * ave; scal[0] = lres; scal[1] = lres; scal[2] = netr; scal[6] = 0.0; scal[7] = 0.0; scal[8] = 0.0; }
}

void launch_cg_init(hipStream_t s, int n, const double *A, const double *b, double *q, double *res, double *p, double *scal) {
  (void)A;
  hipLaunchKernelGGL(cg_init_kernel, dim3(1), dim3(1024), 0, s, n, b, q, res, p, scal);
}

__global__ __launch_bounds__(1024) void cg_update_kernel(int n, double *__restrict__ q, double *__restrict__ res,
                                                         double *__restrict__ p, const double *__restrict__ ap,
                                                         double *__restrict__ scal, double tolerance, int *__restrict__ done,
                                                         int iter, double *__restrict__ hist) {
  __shared__ double red[32];
  if (*done) return;
  double ptap = 0;
  for (int i = threadIdx.x; i < n; i += 1024) ptap += p[i] * ap[i];
  ptap = block_sum_1024(ptap, red);
  const double lresnorm = scal[0], gamma = scal[1];
  const double alpha = lresnorm / ptap;
  double lg = 0, netr = 0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    q[i] = q[i] + alpha * p[i];
    const double r = res[i] - alpha * ap[i];
    res[i] = r; lg += r * r; netr += r;
  }
  block_sum2_1024(lg, netr, red);
  const double ave = netr / n;
  lg -= netr * ave;
  const double beta = lg / gamma;
  double lr = 0;
  for (int i = threadIdx.x; i < n; i += 1024) { const double pn = beta * p[i] + res[i] - ave; p[i] = pn; lr += res[i] * pn; }
  lr = block_sum_1024(lr, red);
  if (threadIdx.x == 0) {
    scal[0] = lr; scal[1] = lg; scal[2] = netr; scal[3] = ptap; scal[4] = alpha; scal[5] = beta;
    hist[iter] = lr;
  }
  if (lr / n < tolerance) {                 // block-uniform: every thread holds the same lr
    double qs = 0.0;                        // net charge of the solution for the "Converged" log line (fix_conp.cpp:917-918)
    for (int i = threadIdx.x; i < n; i += 1024) qs += q[i];
    qs = block_sum_1024(qs, red);
    if (threadIdx.x == 0) { *done = 1; scal[6] = (double)iter; scal[7] = qs; scal[8] = 1.0; }
  }
}

// this lane's share of A[row,:] . p for the CG kernels: ONE accumulator in column order (what the two-launch and the one-launch
// form must agree on to the bit); the loads of eight steps are requested together, the additions stay in order
__device__ __forceinline__ double cg_row_dot(int n, const double *__restrict__ srow, const double *p, int lane) {
  double s0 = 0.0;
  int j = lane;
  // (Measured and removed, round 4: ALL of a deck-sized row's elements requested at once -- 26 loads per lane, one round trip
  //  instead of three -- made the iteration SLOWER, 84.5 vs 79.9 us per solve on il_twolayer: the matrix comes out of L2, the loads
  //  of a later batch overlap the products of an earlier one, and one 26-deep batch serialises them.  Also measured, no effect:
  //  the row's first sixteen elements requested at the top of cg_step_kernel, ahead of the vector update it repeats -- 79.4 vs
  //  79.4-80.8 us: the launch is not waiting for the matrix.  A third: the update's vectors and the two scalars of the iteration
  //  requested together with the convergence flag at the head of cg_step_kernel instead of behind it -- 78.7-79.4 against
  //  77.2-77.5 us: not the head's dependent loads either.)
  for (; j + 960 < n; j += 1024) {           // sixteen steps' loads in flight: a deck-sized row (Ne = 1664) is two round trips
    double a[16], x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { a[u] = srow[j + 64 * u]; x[u] = p[j + 64 * u]; }
#pragma unroll
    for (int u = 0; u < 16; ++u) s0 = fma(a[u], x[u], s0);
  }
  for (; j + 448 < n; j += 512) {
    double a[8], x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = srow[j + 64 * u]; x[u] = p[j + 64 * u]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) s0 = fma(a[u], x[u], s0);
  }
  for (; j < n; j += 64) s0 = fma(srow[j], p[j], s0);
  return s0;
}

__global__ __launch_bounds__(256) void gemv_rows_guarded_kernel(int n, const double *__restrict__ S, const double *__restrict__ b,
                                                                double *__restrict__ y, const int *__restrict__ done) {
  if (*done) return;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const double s0 = wave_sum(cg_row_dot(n, S + (size_t)row * n, b, lane));
  if (lane == 0) y[row] = s0;
}

void launch_cg_iter(hipStream_t s, int n, const double *A, double *q, double *res, double *p, double *ap, double *scal,
                    double tolerance, int *done, int iter, double *hist) {
  hipLaunchKernelGGL(gemv_rows_guarded_kernel, dim3((n + 3) / 4), dim3(256), 0, s, n, A, p, ap, done);
  hipLaunchKernelGGL(cg_update_kernel, dim3(1), dim3(1024), 0, s, n, q, res, p, ap, scal, tolerance, done, iter, hist);
}

// ---- one launch per iteration.  Every workgroup first repeats the vector update of iteration iter - 1 for itself -- the same
// code, thread -> element mapping and reduction trees as cg_update_kernel, so every workgroup holds the same bits -- with the new
// search direction landing in LDS, then multiplies ITS 16 rows of A with it (iteration iter's matvec, gemv_rows_guarded's
// arithmetic).  Workgroup 0 alone writes the vectors and scalars back.  State is double-buffered by iteration parity (set k & 1
// holds p_k, A p_k and the residual entering iteration k): a fast workgroup's outputs never overwrite what a slow one still reads.
// Redundant work per workgroup: 3 vectors of n doubles from L2 and five block reductions -- against one kernel boundary and one
// single-workgroup kernel per iteration (il_twolayer, 6 iterations: 13 -> 7 launches per solve).
//   mode: 1 = start (residual and direction from b, fix_conp.cpp:870-884) + matvec 1;  2 = update(iter - 1) + matvec(iter);
//         3 = matvec(iter) only (the state was completed by a mode-4 launch);  4 = update(iter) only, one workgroup
__global__ __launch_bounds__(1024) void cg_step_kernel(int n, const double *__restrict__ A, const double *__restrict__ b,
                                                       double *__restrict__ q, double *__restrict__ res2, double *__restrict__ p2,
                                                       double *__restrict__ ap2, double *__restrict__ scal, double tolerance,
                                                       int *__restrict__ done, int iter, double *__restrict__ hist, int mode,
                                                       int rows_per_block, double *__restrict__ host_ctl, int n_ctl) {
  extern __shared__ __attribute__((aligned(16))) char cg_smem[];
  double *pl = reinterpret_cast<double *>(cg_smem);          // the direction of the matvec: [n]
  __shared__ double red[32];
  // mode 4 ends a batch: its one workgroup stores the control block (scalars, flag, net charge, residual history -- n_ctl doubles
  // from scal) straight into page-locked host memory: the host's read-back needs no copy-engine transfer behind this launch
  auto to_host = [&]() {
    if (mode != 4 || !host_ctl) return;
    __syncthreads();                                         // (scal[] entries written by thread 0 of this workgroup above)
    for (int i = threadIdx.x; i < n_ctl; i += 1024) host_ctl[i] = scal[i];
  };
  if (mode != 1 && *done) { to_host(); return; }             // (the start launch clears the flag of the previous solve itself)
  const bool writer = blockIdx.x == 0;
  const int k_upd = mode == 4 ? iter : iter - 1;             // the iteration whose update this launch applies (modes 2, 4)
  const int so = mode == 4 ? (iter + 1) & 1 : iter & 1;      // set written: the state entering iteration k_upd + 1
  double *res_o = res2 + (size_t)so * n, *p_o = p2 + (size_t)so * n;
  if (mode == 1) {
    double netr = 0, l2 = 0;
    for (int i = threadIdx.x; i < n; i += 1024) { const double r = b[i]; netr += r; l2 += r * r; }
    netr = block_sum_1024(netr, red);
    l2 = block_sum_1024(l2, red);
    const double ave = netr / n;
    for (int i = threadIdx.x; i < n; i += 1024) {
      const double r = b[i], pn = r - ave;
      pl[i] = pn;
      if (writer) { q[i] = 0.0; res_o[i] = r; p_o[i] = pn; }
    }
    if (writer && threadIdx.x == 0) {
      const double lres = l2 - netr * ave;
      scal[0] = lres; scal[1] = lres; scal[2] = netr; scal[6] = 0.0; scal[7] = 0.0; scal[8] = 0.0;
      scal[9 + 2 * so] = lres; scal[10 + 2 * so] = lres;
      *done = 0;
    }
  } else if (mode == 2 || mode == 4) {
    const int si = k_upd & 1;
    const double *res_i = res2 + (size_t)si * n, *p_i = p2 + (size_t)si * n, *ap_i = ap2 + (size_t)si * n;
    // up to four elements per thread (n <= 4096) stay in registers through the three passes -- one trip to L2 instead of three
    // dependent ones; longer vectors re-read them.  Same operations in the same order either way.
    const bool keep = n <= 4096;
    double kp[4], kap[4], kr[4];
    if (keep) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = threadIdx.x + 1024 * u;
        kp[u] = i < n ? p_i[i] : 0.0; kap[u] = i < n ? ap_i[i] : 0.0; kr[u] = i < n ? res_i[i] : 0.0;
      }
    }
    double ptap = 0;
    if (keep) {
#pragma unroll
      for (int u = 0; u < 4; ++u) if (threadIdx.x + 1024 * u < n) ptap += kp[u] * kap[u];
    } else {
      for (int i = threadIdx.x; i < n; i += 1024) ptap += p_i[i] * ap_i[i];
    }
    ptap = block_sum_1024(ptap, red);
    // (residual norm, gamma) entering iteration k live in the parity slot scal[9 + 2 (k & 1)], scal[10 + 2 (k & 1)]: every
    // workgroup of this launch reads slot k_upd & 1 while workgroup 0 writes the other one
    const double lresnorm = scal[9 + 2 * si], gamma = scal[10 + 2 * si];
    const double alpha = lresnorm / ptap;
    double lg = 0, netr = 0;
    if (keep) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = threadIdx.x + 1024 * u;
        if (i < n) {
          if (writer) q[i] = q[i] + alpha * kp[u];
          const double r = kr[u] - alpha * kap[u];
          if (writer) res_o[i] = r;
          kr[u] = r;
          lg += r * r; netr += r;
        }
      }
    } else {
      for (int i = threadIdx.x; i < n; i += 1024) {
        if (writer) q[i] = q[i] + alpha * p_i[i];
        const double r = res_i[i] - alpha * ap_i[i];
        if (writer) res_o[i] = r;
        lg += r * r; netr += r;
      }
    }
    block_sum2_1024(lg, netr, red);
    const double ave = netr / n;
    lg -= netr * ave;
    const double beta = lg / gamma;
    double lr = 0;
    if (keep) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = threadIdx.x + 1024 * u;
        if (i < n) {
          const double r = kr[u];
          const double pn = beta * kp[u] + r - ave;
          pl[i] = pn;
          if (writer) p_o[i] = pn;
          lr += r * pn;
        }
      }
    } else {
      for (int i = threadIdx.x; i < n; i += 1024) {
        const double r = res_i[i] - alpha * ap_i[i];           // the same value as above, recomputed instead of kept
        const double pn = beta * p_i[i] + r - ave;
        pl[i] = pn;
        if (writer) p_o[i] = pn;
        lr += r * pn;
      }
    }
    lr = block_sum_1024(lr, red);
    if (writer && threadIdx.x == 0) {
      scal[0] = lr; scal[1] = lg; scal[2] = netr; scal[3] = ptap; scal[4] = alpha; scal[5] = beta;      // what cg_update_kernel reports
      scal[9 + 2 * so] = lr; scal[10 + 2 * so] = lg;
      hist[k_upd] = lr;
    }
    if (lr / n < tolerance) {                                // uniform over the whole grid: every workgroup holds the same lr
      if (writer) {
        double qs = 0.0;
        for (int i = threadIdx.x; i < n; i += 1024) qs += q[i];      // this workgroup wrote q above: visible after the block barriers
        qs = block_sum_1024(qs, red);
        if (threadIdx.x == 0) { *done = 1; scal[6] = (double)k_upd; scal[7] = qs; scal[8] = 1.0; }
      }
      to_host();
      return;
    }
  } else {
    const double *p_i = p2 + (size_t)(iter & 1) * n;
    for (int i = threadIdx.x; i < n; i += 1024) pl[i] = p_i[i];
  }
  if (mode == 4) { to_host(); return; }
  __syncthreads();
  // 16 waves: up to 16 rows per workgroup; small matrices take 8 so that more CUs pull on the matrix (Ne = 1664: 208 instead of 104)
  const int wv = threadIdx.x >> 6;
  const int row = blockIdx.x * rows_per_block + wv;
  if (wv >= rows_per_block || row >= n) return;
  const int lane = threadIdx.x & 63;
  const double s0 = wave_sum(cg_row_dot(n, A + (size_t)row * n, pl, lane));
  if (lane == 0) ap2[(size_t)(iter & 1) * n + row] = s0;
}

// ---- round 5: the whole solve as ONE launch (n <= 4096: the vectors live in registers, four elements per thread).  MEASURED SLOWER than a
// launch per iteration (il_twolayer: 92-94 against 76-78 us per solve, profiles/r05_cg_persist_ab.txt): a test path, not the default.
// A kernel boundary invalidates the XCDs' L2s (the counters say so: cg_step_kernel fetches the whole matrix from the memory side at
// every launch, FETCH_SIZE = 8 n^2 bytes), so a launch per iteration re-reads a deck-sized matrix that would fit the aggregate L2
// (22 MB at Ne = 1664) seven times.  Here the workgroups stay: each keeps multiplying ITS rows (plain loads: from its XCD's L2 from
// the second iteration on), publishes the products by write-through stores, passes ONE grid barrier per iteration -- the fence-free
// ticket of the inverse's panel -- and reads the whole product vector back by sc1 loads; every workgroup then repeats the vector
// update for itself (cg_step_kernel's code, reduction trees and element mapping: the same bits in every workgroup and the same bits
// as the launch-per-iteration forms), with p, r and q never leaving its registers.  All workgroups must be resident (the host asks
// the occupancy API); a barrier wait that runs out stores -7 in the control block and the host repeats the solve with a launch per
// iteration.  Two ticket words alternate between solves: this solve zeroes the other one.
__global__ __launch_bounds__(1024) void cg_persist_kernel(int n, const double *__restrict__ A, const double *__restrict__ b,
                                                          double *__restrict__ q, double *ap2 /*[2][n]: the products, by iteration parity*/,
                                                          double *__restrict__ scal, double tolerance, int maxiter, double *__restrict__ hist,
                                                          int rows_per_block, double *__restrict__ host_ctl, unsigned *ticket,
                                                          unsigned *ticket_next, unsigned spin_limit) {
  extern __shared__ __attribute__((aligned(16))) char cg_smem[];
  double *pl = reinterpret_cast<double *>(cg_smem);          // the direction of the matvec: [n]
  __shared__ double red[32];
  __shared__ int s_abort;
  const bool writer = blockIdx.x == 0;
  const unsigned G = gridDim.x;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  if (writer && tid == 0) *ticket_next = 0u;
  if (tid == 0) s_abort = 0;
  double kp[4], kr[4], kq[4], kap[4];
  // ---- start (cg_step_kernel mode 1)
  double lresnorm, gamma;
  {
    double netr = 0, l2 = 0;
    for (int i = tid; i < n; i += 1024) { const double r = b[i]; netr += r; l2 += r * r; }
    netr = block_sum_1024(netr, red);
    l2 = block_sum_1024(l2, red);
    const double ave = netr / n;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 1024 * u;
      const double r = i < n ? b[i] : 0.0;
      kr[u] = r; kp[u] = i < n ? r - ave : 0.0; kq[u] = 0.0;
    }
    const double lres = l2 - netr * ave;
    lresnorm = lres; gamma = lres;
    if (writer && tid == 0) { scal[0] = lres; scal[1] = lres; scal[2] = netr; scal[6] = 0.0; scal[7] = 0.0; scal[8] = 0.0; }
  }
  const int row = blockIdx.x * rows_per_block + wv;
  const bool has_row = wv < rows_per_block && row < n;
  auto finish = [&](int it, double done_flag, double lr, double lg, double netr, double ptap, double alpha, double beta) {
    // the writer workgroup leaves the solution and the control block (device copy and, when asked, the page-locked host copy)
    if (!writer) return;
    double qs = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = tid + 1024 * u; if (i < n) { q[i] = kq[u]; qs += kq[u]; } }
    qs = block_sum_1024(qs, red);
    if (tid == 0) {
      scal[0] = lr; scal[1] = lg; scal[2] = netr; scal[3] = ptap; scal[4] = alpha; scal[5] = beta;
      scal[6] = (double)it; scal[7] = qs; scal[8] = done_flag;
    }
    __syncthreads();
    if (host_ctl) for (int i = tid; i < 16 + it + 1; i += 1024) host_ctl[i] = i < 16 ? scal[i] : hist[i - 16];
  };
  for (int it = 1;; ++it) {
    // ---- matvec of iteration it with the direction every workgroup holds
    __syncthreads();                                         // (the previous iteration's readers of pl are done)
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = tid + 1024 * u; if (i < n) pl[i] = kp[u]; }
    __syncthreads();
    double *apw = ap2 + (size_t)(it & 1) * n;
    if (has_row) {
      const double s0 = wave_sum(cg_row_dot(n, A + (size_t)row * n, pl, lane));
      if (lane == 0) __hip_atomic_store(apw + row, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {                                          // grid barrier #it
      __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = G * (unsigned)it;
      unsigned spins = 0;
      while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > spin_limit) { s_abort = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (s_abort) {
      if (writer && tid == 0) { scal[8] = -7.0; if (host_ctl) host_ctl[8] = -7.0; }
      return;
    }
    // (sc1 loads requested together: as relaxed atomics the compiler waits for each before it issues the next)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 1024 * u;
      kap[u] = 0.0;
      if (i < n) { const double *src = apw + i; asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(kap[u]) : "v"(src) : "memory"); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(kap[u]));
    // ---- the update of iteration it (cg_step_kernel modes 2 / 4, the `keep` branch)
    double ptap = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) if (tid + 1024 * u < n) ptap += kp[u] * kap[u];
    ptap = block_sum_1024(ptap, red);
    const double alpha = lresnorm / ptap;
    double lg = 0, netr = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 1024 * u;
      if (i < n) {
        kq[u] = kq[u] + alpha * kp[u];
        const double r = kr[u] - alpha * kap[u];
        kr[u] = r;
        lg += r * r; netr += r;
      }
    }
    block_sum2_1024(lg, netr, red);
    const double ave = netr / n;
    lg -= netr * ave;
    const double beta = lg / gamma;
    double lr = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 1024 * u;
      if (i < n) {
        const double r = kr[u];
        const double pn = beta * kp[u] + r - ave;
        kp[u] = pn;
        lr += r * pn;
      }
    }
    lr = block_sum_1024(lr, red);
    if (writer && tid == 0) hist[it] = lr;
    lresnorm = lr; gamma = lg;
    const bool converged = lr / n < tolerance;               // uniform over the whole grid: every workgroup holds the same lr
    if (converged || it + 1 >= maxiter) { finish(it, converged ? 1.0 : 0.0, lr, lg, netr, ptap, alpha, beta); return; }
  }
}
bool cg_persist_fits(int n) { return n > 0 && n <= 4096; }
// returns false when the workgroups cannot all be resident on `num_cus` compute units (the caller takes the launch-per-iteration form)
bool launch_cg_persist(hipStream_t s, int num_cus, int n, const double *A, const double *b, double *q, double *ap2, double *scal,
                       double tolerance, int maxiter, double *hist, double *host_ctl, unsigned *ticket, unsigned *ticket_next,
                       unsigned spin_limit) {
  const size_t lds = (size_t)n * sizeof(double);
  static DynLdsCache granted{};
  ensure_dyn_lds(cg_persist_kernel, lds, granted);
  const int rpb = 8;
  const int nwg = (n + rpb - 1) / rpb;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cg_persist_kernel, 1024, lds) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (per_cu <= 0 || nwg > per_cu * num_cus) return false;
  hipLaunchKernelGGL(cg_persist_kernel, dim3(nwg), dim3(1024), lds, s, n, A, b, q, ap2, scal, tolerance, maxiter, hist, rpb, host_ctl, ticket,
                     ticket_next, spin_limit);
  return true;
}

bool cg_step_fits(int n) { return n > 0 && (size_t)n * sizeof(double) <= 128 * 1024; }

void launch_cg_step(hipStream_t s, int n, const double *A, const double *b, double *q, double *res2, double *p2, double *ap2,
                    double *scal, double tolerance, int *done, int iter, double *hist, int mode, double *host_ctl, int n_ctl) {
  const size_t lds = (size_t)n * sizeof(double);
  static DynLdsCache granted{};
  ensure_dyn_lds(cg_step_kernel, lds, granted);
  const int rpb = n <= 4096 ? 8 : 16;
  hipLaunchKernelGGL(cg_step_kernel, dim3(mode == 4 ? 1 : (n + rpb - 1) / rpb), dim3(1024), lds, s, n, A, b, q, res2, p2, ap2, scal, tolerance,
                     done, iter, hist, mode, rpb, mode == 4 ? host_ctl : nullptr, n_ctl);
}

}  // namespace conp
