// Microbenchmark: the MFMA phase of sk_gemm on its own -- sk_mfma_chunk over a static LDS panel, two waves per SIMD, one
// workgroup per CU -- with and without the per-chunk barrier, with and without the culling branch.  What fraction of the FP64
// matrix peak does this instruction mix reach when nothing else (panel build, table loads, epilogue) is in the way?
// Build (from the repo root): hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/microbench/sk_loop_bench.hip -o tools/microbench/sk_loop_bench
#include "../../lammps-user-conp2_amd/csrc/conp_kernels.hip"

#include <cstdio>

using namespace conp;

template <int NFW, int MODE>
__global__ __launch_bounds__(512, 2) void loop_kernel(int iters, int f0, double *out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (unsigned i = threadIdx.x; i < SK_LDS_BYTES / 8; i += 512) reinterpret_cast<double *>(smem)[i] = 1e-3 * (double)(i % 97);
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
  for (int it = 0; it < iters; ++it, buf ^= SK_BUF1) {
    sk_mfma_chunk<NFW>(c, smem, buf, acc);
    if (MODE & 1) __syncthreads();
  }
  double s = 0.0;
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < NFW; ++g) s += acc[f][g][0] + acc[f][g][1] + acc[f][g][2] + acc[f][g][3];
  out[(size_t)blockIdx.x * 512 + t] = s;
}

template <int NFW, int MODE>
static void run(const char *what, int f0, int ncu, double *d_out) {
  const int iters = 4000;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(loop_kernel<NFW, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SK_LDS_BYTES);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((loop_kernel<NFW, MODE>), dim3(ncu), dim3(512), SK_LDS_BYTES, 0, 200, f0, d_out);     // warm-up
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a, 0);
  hipLaunchKernelGGL((loop_kernel<NFW, MODE>), dim3(ncu), dim3(512), SK_LDS_BYTES, 0, iters, f0, d_out);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  // MFMAs per wave per chunk: 4 k-steps x (4 row fragments x (NFW - 1) + f0 culled ones)
  const double mf = 4.0 * (4.0 * (NFW - 1) + f0);
  const double flops = (double)ncu * 8 * iters * mf * 2048.0;
  printf("%-46s NFW %d f0 %d: %8.3f ms  %6.2f TFLOP/s  %.3f of 78.6\n", what, NFW, f0, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 78.6e12);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  double *d_out;
  (void)hipMalloc(&d_out, sizeof(double) * (size_t)ncu * 512);
  printf("device %s, %d CUs\n", p.name, ncu);
  run<4, 0>("no barrier", 4, ncu, d_out);
  run<4, 1>("barrier per chunk (4 k-steps)", 4, ncu, d_out);
  run<4, 1>("barrier per chunk, last column culled from f = 2", 2, ncu, d_out);
  run<5, 0>("no barrier", 4, ncu, d_out);
  run<5, 1>("barrier per chunk", 4, ncu, d_out);
  run<3, 1>("barrier per chunk", 4, ncu, d_out);
  run<2, 1>("barrier per chunk", 4, ncu, d_out);
  (void)hipFree(d_out);
  return 0;
}
