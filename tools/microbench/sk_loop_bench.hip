// Microbenchmark: the MFMA phase of sk_gemm on its own -- over a static LDS panel, one workgroup per CU -- in several forms:
//   loop     sk_mfma_chunk    (k-step loop, one wave-uniform culling branch per row fragment: the round-2 product form)
//   unrolled sk_mfma_chunk_u  (straight line, the sphere cut a template parameter)
//   regs     the same MFMA sequence with constant operand registers, no LDS read at all (what the loop form alone costs)
// each with / without the per-chunk barrier and with 2 or 1 waves per SIMD.  What fraction of the FP64 matrix peak does the
// instruction mix reach when nothing else (panel build, table loads, epilogue) is in the way?
// Build (from the repo root): hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/microbench/sk_loop_bench.hip -o tools/microbench/sk_loop_bench
#include "../../lammps-user-conp2_amd/csrc/conp_kernels.hip"

#include <cstdio>

using namespace conp;

// FORM 0 loop, 1 unrolled, 2 registers only (loop form), 3 registers only (straight line)
template <int NFW, int F0, int FORM, int BAR, int NT>
__global__ __launch_bounds__(NT, 2) void loop_kernel(int iters, int f0, double *out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (unsigned i = threadIdx.x; i < SK_LDS_BYTES / 8; i += NT) reinterpret_cast<double *>(smem)[i] = 1e-3 * (double)(i % 97);
  __syncthreads();
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  SkCtx c;
  c.rh = wave & 1; c.cg = wave >> 1;
  c.fr = lane & 15; c.fk = lane >> 4;
  c.base_a = ((unsigned)(64 * c.rh + c.fr) * SK_LD + (unsigned)((c.fk ^ c.fr) & 3)) * 8u;
  c.base_b = ((unsigned)(128 + 16 * c.cg + c.fr) * SK_LD + (unsigned)((c.fk ^ c.fr) & 3)) * 8u;
  c.pq = (unsigned)(c.fr >> 2) << 5;
  c.f0 = __builtin_amdgcn_readfirstlane(f0);
  d4 acc[4][NFW];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < NFW; ++g) acc[f][g] = (d4){0.0, 0.0, 0.0, 0.0};
  unsigned buf = 0;
  double ra = 1e-3 * lane, rb[NFW];
#pragma unroll
  for (int g = 0; g < NFW; ++g) rb[g] = 1e-4 * (lane + g);
  for (int it = 0; it < iters; ++it, buf ^= SK_BUF1) {
    if constexpr (FORM == 0) sk_mfma_chunk<4, NFW>(c, smem, buf, acc);
    else if constexpr (FORM == 1) sk_mfma_chunk_u<4, NFW, F0>(c, smem, buf, acc);
    else if constexpr (FORM == 2) {
#pragma unroll 1
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
#pragma unroll
          for (int g = 0; g + 1 < NFW; ++g) acc[f][g] = MFMA_F64(ra, rb[g], acc[f][g]);
          if (f < c.f0) acc[f][NFW - 1] = MFMA_F64(ra, rb[NFW - 1], acc[f][NFW - 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
#pragma unroll
          for (int g = 0; g + 1 < NFW; ++g) acc[f][g] = MFMA_F64(ra, rb[g], acc[f][g]);
          if (f < F0) acc[f][NFW - 1] = MFMA_F64(ra, rb[NFW - 1], acc[f][NFW - 1]);
        }
    }
    if (BAR) __syncthreads();
  }
  double s = 0.0;
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < NFW; ++g) s += acc[f][g][0] + acc[f][g][1] + acc[f][g][2] + acc[f][g][3];
  out[(size_t)blockIdx.x * NT + t] = s;
}

template <int NFW, int F0, int FORM, int BAR, int NT>
static void run(int ncu, double *d_out) {
  static const char *forms[] = {"loop", "unrolled", "regs/loop", "regs/straight"};
  const int iters = 4000;
  auto k = loop_kernel<NFW, F0, FORM, BAR, NT>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SK_LDS_BYTES);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(ncu), dim3(NT), SK_LDS_BYTES, 0, 200, F0, d_out);     // warm-up
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(ncu), dim3(NT), SK_LDS_BYTES, 0, iters, F0, d_out);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  // MFMAs per wave per chunk: 4 k-steps x (4 row fragments x (NFW - 1) + F0 unculled last ones)
  const double mf = 4.0 * (4.0 * (NFW - 1) + F0);
  const double flops = (double)ncu * (NT / 64) * iters * mf * 2048.0;
  printf("%-14s %s  %d waves/SIMD  NFW %d f0 %d: %8.3f ms  %6.2f TFLOP/s  %.3f of 78.6\n", forms[FORM], BAR ? "barrier/chunk" : "no barrier   ",
         NT / 256, NFW, F0, best, flops / (best * 1e-3) / 1e12, flops / (best * 1e-3) / 78.6e12);
  fflush(stdout);
}

template <int NFW, int F0>
static void family(int ncu, double *d_out) {
  run<NFW, F0, 0, 1, 512>(ncu, d_out);
  run<NFW, F0, 1, 1, 512>(ncu, d_out);
  run<NFW, F0, 0, 0, 512>(ncu, d_out);
  run<NFW, F0, 1, 0, 512>(ncu, d_out);
  run<NFW, F0, 2, 1, 512>(ncu, d_out);
  run<NFW, F0, 3, 1, 512>(ncu, d_out);
  run<NFW, F0, 2, 0, 512>(ncu, d_out);
  run<NFW, F0, 3, 0, 512>(ncu, d_out);
  run<NFW, F0, 0, 0, 256>(ncu, d_out);
  run<NFW, F0, 1, 0, 256>(ncu, d_out);
  run<NFW, F0, 3, 0, 256>(ncu, d_out);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  double *d_out;
  (void)hipMalloc(&d_out, sizeof(double) * (size_t)ncu * 512);
  printf("device %s, %d CUs\n", p.name, ncu);
  family<4, 4>(ncu, d_out);
  family<4, 2>(ncu, d_out);
  family<5, 4>(ncu, d_out);
  family<3, 4>(ncu, d_out);
  family<2, 4>(ncu, d_out);
  (void)hipFree(d_out);
  return 0;
}
