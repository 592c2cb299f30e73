// Microbenchmark (round 5): does sustained FP64 VALU / LDS work of one wave slow the FP64 MFMA stream of its SIMD partner?
// Workgroup of 512 threads per CU: waves 0-3 stream v_mfma_f64_16x16x4 (16 independent accumulators), waves 4-7 run, for the whole
// duration, a loop of:  0 nothing  1 independent v_fma_f64  2 dependent v_fma_f64  3 v_fma_f32  4 ds_write_b64  5 ds_read_b64
// 6: v_fma_f64 at priority 3.  Reports cycles per MFMA of the MFMA waves and cycles per op of the partner.
// Build: hipcc -O3 --offload-arch=gfx950 pipe_share2_bench.hip -o pipe_share2_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(512) void k(int mode, int mfma_iters, int op_iters, unsigned long long *out, double *sink, int both) {
  __shared__ double lds[512 * 2];
  const int wave = threadIdx.x >> 6;
  unsigned long long t0 = 0, t1 = 0;
  double s = 0;
  if (wave < 4 || both) {
    d4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else {
    double y[8];
    float yf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { y[i] = 1.0 + threadIdx.x * 1e-6 + i; yf[i] = (float)y[i]; }
    const double ca = 0.999999, cb = 1e-7;
    const unsigned la = (unsigned)(threadIdx.x * 8);
    if (mode == 6) __builtin_amdgcn_s_setprio(3);
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < op_iters; ++it) {
      if (mode == 1 || mode == 6) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(y[i & 7]) : "v"(ca), "v"(cb));
      } else if (mode == 2) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(y[0]) : "v"(ca), "v"(cb));
      } else if (mode == 3) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(yf[i & 7]) : "v"(0.99999f), "v"(1e-7f));
      } else if (mode == 4) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("ds_write_b64 %0, %1" ::"v"(la), "v"(y[0]) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (mode == 5) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("ds_read_b64 %0, %1" : "=v"(y[i & 7]) : "v"(la) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 8; ++i) s += y[i] + yf[i];
    s += lds[threadIdx.x];
  }
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  int ncu = 256;
  unsigned long long *d_out; double *d_sink;
  CK(hipMalloc(&d_out, sizeof(unsigned long long) * ncu * 8));
  CK(hipMalloc(&d_sink, sizeof(double) * ncu * 512));
  std::vector<unsigned long long> h(ncu * 8);
  const char *names[] = {"partner idle", "partner: independent v_fma_f64", "partner: dependent v_fma_f64", "partner: v_fma_f32", "partner: ds_write_b64",
                         "partner: ds_read_b64", "partner: v_fma_f64 at prio 3", "partner ALSO streams MFMA"};
  const int mfma_iters = 2000;
  for (int mode = 0; mode < 8; ++mode) {
    // partner op count chosen so that it runs about as long as the MFMA stream (2000 * 16 * 64 = 2.05 M cycles)
    const int op_iters = mode == 0 ? 0 : 2000000 / (32 * 12);
    CK(hipMemset(d_out, 0, sizeof(unsigned long long) * ncu * 8));
    hipLaunchKernelGGL(k, dim3(ncu), dim3(512), 0, 0, mode, mfma_iters, op_iters, d_out, d_sink, mode == 7 ? 1 : 0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * ncu * 8, hipMemcpyDeviceToHost));
    double sm = 0, st = 0;
    for (int b = 0; b < ncu; ++b) { for (int w = 0; w < 4; ++w) sm += (double)h[b * 8 + w]; for (int w = 4; w < 8; ++w) st += (double)h[b * 8 + w]; }
    printf("%-34s MFMA waves: %7.2f cycles per MFMA   partner: %8.2f cycles per op (total %.0f vs %.0f cycles)\n", names[mode], sm / (ncu * 4.0) / (mfma_iters * 16.0),
           mode == 7 ? st / (ncu * 4.0) / (mfma_iters * 16.0) : (op_iters ? st / (ncu * 4.0) / (op_iters * 32.0) : 0.0), st / (ncu * 4.0), sm / (ncu * 4.0));
  }
  return 0;
}
