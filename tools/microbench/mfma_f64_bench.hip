// Microbenchmark: FP64 MFMA / FP64 VALU issue rates and the f64 16x16x4 fragment layout on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_bench.hip -o mfma_f64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// ---- layout check: D = A(16x4) * B(4x16) --------------------------------------------------------
__global__ void layout_kernel(const double* A, const double* B, double* D) {
  int l = threadIdx.x;                 // one wave
  double a = A[(l & 15) * 4 + (l >> 4)];   // A[row=l&15][k=l>>4]
  double b = B[(l >> 4) * 16 + (l & 15)];  // B[k=l>>4][col=l&15]
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) + 4 * r, col = l & 15;
    D[row * 16 + col] = c[r];
  }
}

// ---- throughput: NACC independent accumulators, ITERS iterations ---------------------------------
template <int NACC>
__global__ void mfma_rate_kernel(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void fma_rate_kernel(double* out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = i * 1e-3;
  double a = a0 + threadIdx.x * 1e-12, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MFMA and VALU fma interleaved in one wave: do they overlap?
template <int NACC, int NV>
__global__ void mixed_rate_kernel(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  double v[NV];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  for (int i = 0; i < NV; ++i) v[i] = i * 1e-3;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV / NACC; ++j) v[i * (NV / NACC) + j] = __builtin_fma(v[i * (NV / NACC) + j], a, b);
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NV; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d kHz  L2=%d  gcn=%s\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.l2CacheSize, prop.gcnArchName);
  // layout check
  {
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = 1 + i * 0.5 + (i % 3); B[i] = 2 - i * 0.25 + (i % 5) * 3; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s = fma(A[i * 4 + k], B[k * 16 + j], s); R[i * 16 + j] = s; }
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    double maxerr = 0; int exact = 0;
    for (int i = 0; i < 256; ++i) { maxerr = fmax(maxerr, fabs(D[i] - R[i])); exact += (D[i] == R[i]); }
    printf("layout check f64 16x16x4: max |D-ref| = %g, bit-exact vs k-ordered fma chain: %d/256\n", maxerr, exact);
  }
  const int CU = prop.multiProcessorCount;
  double* out; CK(hipMalloc(&out, (size_t)CU * 8 * 1024 * 8));
  const int iters = 20000;
  for (int wpb : {256, 512, 1024}) {       // 1, 2, 4 waves per SIMD
    for (int nacc : {1, 2, 4, 8}) {
      double ms;
      auto run = [&](auto kern) { return time_ms([&] { hipLaunchKernelGGL(kern, dim3(CU), dim3(wpb), 0, 0, out, iters, 1.0000001, 0.5); }); };
      if (nacc == 1) ms = run(mfma_rate_kernel<1>); else if (nacc == 2) ms = run(mfma_rate_kernel<2>); else if (nacc == 4) ms = run(mfma_rate_kernel<4>); else ms = run(mfma_rate_kernel<8>);
      double nm = (double)CU * (wpb / 64) * iters * nacc;
      double flops = nm * 16 * 16 * 4 * 2;
      printf("MFMA f64 16x16x4: %4d thr/blk nacc=%d  %8.3f ms  %7.2f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", wpb, nacc, ms, flops / ms * 1e-9,
             ms * 1e6 / ((double)iters * nacc * (wpb / 256.0)));
    }
  }
  for (int wpb : {256, 512, 1024}) {
    for (int nacc : {4, 8, 16}) {
      double ms;
      auto run = [&](auto kern) { return time_ms([&] { hipLaunchKernelGGL(kern, dim3(CU), dim3(wpb), 0, 0, out, iters, 0.9999999, 1e-7); }); };
      if (nacc == 4) ms = run(fma_rate_kernel<4>); else if (nacc == 8) ms = run(fma_rate_kernel<8>); else ms = run(fma_rate_kernel<16>);
      double flops = (double)CU * wpb * iters * nacc * 2;
      printf("VALU fma f64:     %4d thr/blk nacc=%2d %8.3f ms  %7.2f TFLOP/s\n", wpb, nacc, ms, flops / ms * 1e-9);
    }
  }
  for (int wpb : {256, 512}) {
    double ms = time_ms([&] { hipLaunchKernelGGL((mixed_rate_kernel<4, 8>), dim3(CU), dim3(wpb), 0, 0, out, iters, 0.9999999, 1e-7); });
    double fm = (double)CU * (wpb / 64) * iters * 4 * 2048, fv = (double)CU * wpb * iters * 8 * 2;
    printf("mixed 4 MFMA + 8 fma/iter: %4d thr/blk %8.3f ms  MFMA %7.2f TF + VALU %7.2f TF\n", wpb, ms, fm / ms * 1e-9, fv / ms * 1e-9);
    ms = time_ms([&] { hipLaunchKernelGGL((mixed_rate_kernel<4, 16>), dim3(CU), dim3(wpb), 0, 0, out, iters, 0.9999999, 1e-7); });
    fv = (double)CU * wpb * iters * 16 * 2;
    printf("mixed 4 MFMA + 16 fma/iter: %4d thr/blk %8.3f ms  MFMA %7.2f TF + VALU %7.2f TF\n", wpb, ms, fm / ms * 1e-9, fv / ms * 1e-9);
  }
  printf("done\n");
  return 0;
}
