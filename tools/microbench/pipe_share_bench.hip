// Microbenchmark (round 5): what does an instruction of wave B cost while its SIMD partner A streams FP64 MFMAs?
// One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) issue independent v_mfma_f64_16x16x4 back to back (or idle),
// waves 4-7 run a test sequence between two s_memtime stamps:
//   0: a dependent chain of v_fma_f64            1: eight independent chains of v_fma_f64
//   2: ds_write_b64 x N, then lgkmcnt(0)          3: ds_read_b64 + lgkmcnt(0), N times
//   4: a dependent chain of v_fma_f32             5: a dependent chain of v_mul_f64
//   6: a dependent chain of v_add_f64             7: ds_write_b64 + lgkmcnt(0), N times (dependent round trips)
// Build: hipcc -O3 --offload-arch=gfx950 pipe_share_bench.hip -o pipe_share_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int NOPS = 64;      // test operations between the stamps, repeated REP times
constexpr int REP = 8;

__global__ __launch_bounds__(512) void share_kernel(int mode, int mfma_on, int mfma_iters, unsigned long long *out, double *sink, int prio, int swap) {
  __shared__ double lds[512 * 8];
  const int wave = threadIdx.x >> 6;
  if ((wave < 4) != (swap != 0)) {
    d4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mfma_on) {
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 8 + (wave & 3)) * 2] = t1 - t0;
    return;
  }
  // test waves: give the MFMA waves a head start
  __builtin_amdgcn_s_sleep(100);
#ifndef V1
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  if (prio == 2) __builtin_amdgcn_s_setprio(2);
  if (prio == 3) __builtin_amdgcn_s_setprio(3);
#endif
  double x = 1.0 + threadIdx.x * 1e-6, y[8];
  float xf = 1.0f + threadIdx.x * 1e-3f;
  const double ca = 0.999999, cb = 1e-7;
#pragma unroll
  for (int i = 0; i < 8; ++i) y[i] = x + i;
  const unsigned la = (unsigned)(threadIdx.x * 8);
  unsigned long long tot = 0;
  for (int r = 0; r < REP; ++r) {
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (mode == 0) {
#pragma unroll
      for (int i = 0; i < NOPS; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(ca), "v"(cb));
    } else if (mode == 1) {
#pragma unroll
      for (int i = 0; i < NOPS; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(y[i & 7]) : "v"(ca), "v"(cb));
    } else if (mode == 2) {
#pragma unroll
      for (int i = 0; i < NOPS; ++i) asm volatile("ds_write_b64 %0, %1" ::"v"(la), "v"(x) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (mode == 3) {
#pragma unroll
      for (int i = 0; i < NOPS; ++i) asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(la) : "memory");
    } else if (mode == 4) {
#pragma unroll
      for (int i = 0; i < NOPS; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xf) : "v"(0.99999f), "v"(1e-7f));
    } else if (mode == 5) {
#pragma unroll
      for (int i = 0; i < NOPS; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(ca));
    } else if (mode == 6) {
#pragma unroll
      for (int i = 0; i < NOPS; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(cb));
    } else if (mode == 7) {
#pragma unroll
      for (int i = 0; i < NOPS; ++i) asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(la), "v"(x) : "memory");
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    tot += t1 - t0;
  }
  double s = x + xf;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += y[i];
  sink[blockIdx.x * 512 + threadIdx.x] = s + lds[threadIdx.x];
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 8 + 4 + (wave & 3)) * 2] = tot;
}

int main() {
  int ncu = 256;
  unsigned long long *d_out; double *d_sink;
  CK(hipMalloc(&d_out, sizeof(unsigned long long) * ncu * 16));
  CK(hipMalloc(&d_sink, sizeof(double) * ncu * 512));
  std::vector<unsigned long long> h(ncu * 16);
  const char *names[] = {"dep v_fma_f64", "8 indep v_fma_f64", "ds_write_b64 stream", "ds_read_b64 round trips", "dep v_fma_f32",
                         "dep v_mul_f64", "dep v_add_f64", "ds_write_b64 round trips"};
  const int mfma_iters = 4000;      // 64000 MFMAs per wave: ~4 M cycles, far longer than any test sequence
  for (int swap = 0; swap < 2; ++swap)
  for (int prio = 0; prio < 4; prio += 3) {
  printf("== test waves are waves %s, at priority %d (MFMA waves at 0)\n", swap ? "0-3 (older)" : "4-7 (younger)", prio);
  for (int mode = 0; mode < 8; ++mode) {
    double per_op[2], per_mfma[2];
    for (int on = 0; on < 2; ++on) {
      CK(hipMemset(d_out, 0, sizeof(unsigned long long) * ncu * 16));
      hipLaunchKernelGGL(share_kernel, dim3(ncu), dim3(512), 0, 0, mode, on, mfma_iters, d_out, d_sink, prio, swap);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * ncu * 16, hipMemcpyDeviceToHost));
      double st = 0, sm = 0;
      for (int b = 0; b < ncu; ++b) {
        for (int w = 0; w < 4; ++w) sm += (double)h[(b * 8 + w) * 2];
        for (int w = 4; w < 8; ++w) st += (double)h[(b * 8 + w) * 2];
      }
      per_op[on] = st / (ncu * 4.0) / (REP * NOPS);
      per_mfma[on] = sm / (ncu * 4.0) / (mfma_iters * 16.0);
    }
    printf("%-28s ticks/op: partner idle %8.2f   partner streaming MFMA %8.2f   (MFMA wave: %6.2f ticks per MFMA)\n", names[mode], per_op[0],
           per_op[1], per_mfma[1]);
  }
  }
  // reference: MFMA wave alone (test waves in mode 4 = FP32, cheap)
  return 0;
}
