// Microbenchmark (round 5): where does a SINGLE wave's MFMA stream lose its 5-6 % when its operands come out of LDS?
// (sk_loop_bench: 0.989 of peak with constant operand registers, 0.939 with sk_mfma_chunk_u's LDS reads, one wave per SIMD.)
// One wave per SIMD (256 threads), RF = 4 row fragments x NFW = 4 column fragments, four k-steps per "chunk", no barrier:
//   0  sk_mfma_chunk_u as it is
//   1  constant operands, no LDS instruction (the ceiling of the instruction stream)
//   2  constant operands for the MFMAs, the SAME ds_read_b64 sequence issued into registers nobody multiplies (the reads' return
//      traffic alone)
//   3  as 2 with every second read left out
//   4  operands from LDS, the WHOLE next k-step's operands (4 A + 4 B values) requested one k-step ahead (2 x 8 registers)
//   5  as 0 but the A value of row fragment f + 2 requested (two groups ahead instead of one)
// Build (repo root): hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/microbench/sk_reads_bench.hip -o tools/microbench/sk_reads_bench
#include "../../lammps-user-conp2_amd/csrc/conp_kernels.hip"

#include <cstdio>

using namespace conp;

constexpr int RF = 4, NFW = 4;

template <int FORM>
__global__ __launch_bounds__(256, 1) void k(int iters, double *out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (unsigned i = threadIdx.x; i < SK_LDS_BYTES / 8; i += 256) reinterpret_cast<double *>(smem)[i] = 1e-3 * (double)(i % 97);
  __syncthreads();
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  SkCtx c;
  c.rh = wave & 1; c.cg = wave >> 1;
  c.fr = lane & 15; c.fk = lane >> 4;
  c.base_a = ((unsigned)(64 * c.rh + c.fr) * SK_LD + (unsigned)((c.fk ^ c.fr) & 3)) * 8u;
  c.base_b = ((unsigned)(128 + 16 * c.cg + c.fr) * SK_LD + (unsigned)((c.fk ^ c.fr) & 3)) * 8u;
  c.pq = (unsigned)(c.fr >> 2) << 5;
  c.f0 = RF;
  d4 acc[RF][NFW];
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int g = 0; g < NFW; ++g) acc[f][g] = (d4){0.0, 0.0, 0.0, 0.0};
  constexpr unsigned FA = 16 * SK_LD * 8, FB = 64 * SK_LD * 8;
  double ra = 1e-3 * lane, rb[NFW];
#pragma unroll
  for (int g = 0; g < NFW; ++g) rb[g] = 1e-4 * (lane + g);
  double sink = 0.0;
  unsigned buf = 0;
  for (int it = 0; it < iters; ++it, buf ^= SK_BUF1) {
    unsigned aa[4], ab[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { const unsigned q = (unsigned)(ks << 5) ^ c.pq; aa[ks] = (c.base_a ^ buf) + q; ab[ks] = (c.base_b ^ buf) + q; }
    if constexpr (FORM == 0) sk_mfma_chunk_u<RF, NFW, RF>(c, smem, buf, acc);
    else if constexpr (FORM == 1) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int f = 0; f < RF; ++f) {
#pragma unroll
          for (int g = 0; g < NFW; ++g) acc[f][g] = MFMA_F64(ra, rb[g], acc[f][g]);
          __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (FORM == 2 || FORM == 3) {
      double d0 = 0.0, d1 = 0.0;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int f = 0; f < RF; ++f) {
          if (FORM == 2 || (f & 1) == 0) { d0 = SK_LDS_F64(aa[ks] + f * FA); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
          for (int g = 0; g < NFW; ++g) {
            acc[f][g] = MFMA_F64(ra, rb[g], acc[f][g]);
            if (f == RF - 1 && (FORM == 2 || (g & 1) == 0)) { d1 = SK_LDS_F64(ab[ks] + g * FB); __builtin_amdgcn_sched_barrier(0); }
          }
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("" : "+v"(d0), "+v"(d1));
        }
      sink += d0 + d1;
    } else if constexpr (FORM == 4) {
      double ca[RF], cb[NFW], na[RF], nb[NFW];
#pragma unroll
      for (int f = 0; f < RF; ++f) ca[f] = SK_LDS_F64(aa[0] + f * FA);
#pragma unroll
      for (int g = 0; g < NFW; ++g) cb[g] = SK_LDS_F64(ab[0] + g * FB);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int kn = (ks + 1) & 3;
#pragma unroll
        for (int f = 0; f < RF; ++f) {
          // the next k-step's operands: one A and one B value requested per row-fragment group
          na[f] = SK_LDS_F64(aa[kn] + f * FA);
          nb[f] = SK_LDS_F64(ab[kn] + f * FB);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < NFW; ++g) acc[f][g] = MFMA_F64(ca[f], cb[g], acc[f][g]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int f = 0; f < RF; ++f) { ca[f] = na[f]; cb[f] = nb[f]; }
      }
    } else if constexpr (FORM == 6 || FORM == 7) {
      // accumulators in AccVGPRs (inline assembly: the builtin leaves the choice to the allocator, which takes plain VGPRs):
      // 6 the product's read sequence, 7 constant operands
#define MFMA_A(av, bv, cc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(cc) : "v"(av), "v"(bv))
      if constexpr (FORM == 7) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int f = 0; f < RF; ++f)
#pragma unroll
            for (int g = 0; g < NFW; ++g) MFMA_A(ra, rb[g], acc[f][g]);
      } else {
        double bf[NFW], a0, a1 = 0.0;
        a0 = SK_LDS_F64(aa[0]);
#pragma unroll
        for (int g = 0; g < NFW; ++g) bf[g] = SK_LDS_F64(ab[0] + g * FB);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int kn = (ks + 1) & 3;
#pragma unroll
          for (int f = 0; f < RF; ++f) {
            if (!(ks == 3 && f == RF - 1)) a1 = f < RF - 1 ? SK_LDS_F64(aa[ks] + (f + 1) * FA) : SK_LDS_F64(aa[kn]);
            asm volatile("" : "+v"(a1));
#pragma unroll
            for (int g = 0; g < NFW; ++g) {
              MFMA_A(a0, bf[g], acc[f][g]);
              if (f == RF - 1 && ks < 3) { bf[g] = SK_LDS_F64(ab[kn] + g * FB); asm volatile("" : "+v"(bf[g])); }
            }
            a0 = a1;
          }
        }
      }
    } else if constexpr (FORM == 8 || FORM == 9) {
      // the product's registers (NFW B values, two A values; form 9: three A values), the reads gathered into FEWER interruptions of
      // the MFMA stream: the last row-fragment group re-reads the B values in two bursts -- two behind its second MFMA, two behind
      // its last together with the next group's A value -- instead of one behind each MFMA
      double bf[NFW], a0, a1 = 0.0, a2 = 0.0;
      a0 = SK_LDS_F64(aa[0]);
      if (FORM == 9) a1 = SK_LDS_F64(aa[0] + FA);
#pragma unroll
      for (int g = 0; g < NFW; ++g) bf[g] = SK_LDS_F64(ab[0] + g * FB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int kn = (ks + 1) & 3;
#pragma unroll
        for (int f = 0; f < RF; ++f) {
          if (FORM == 8) {
            // A of the next group: at the start of groups 1 .. RF-1 for the group behind; the one for group 0 of the next k-step rides
            // in the burst at the end of the last group (below), the one for group 1 is read at group 0's start as before
            if (f < RF - 1) { a1 = SK_LDS_F64(aa[ks] + (f + 1) * FA); __builtin_amdgcn_sched_barrier(0); }
          } else {
            // form 9: A values in pairs, two groups ahead (f even: read A of f + 2 and f + 3)
            if ((f & 1) == 0) {
              const int f2 = f + 2;
              if (f2 < RF) { a2 = SK_LDS_F64(aa[ks] + f2 * FA); }
              else { a2 = SK_LDS_F64(aa[kn] + (f2 - RF) * FA); }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (f < RF - 1) {
#pragma unroll
            for (int g = 0; g < NFW; ++g) acc[f][g] = MFMA_F64(a0, bf[g], acc[f][g]);
            __builtin_amdgcn_sched_barrier(0);
          } else {
            acc[f][0] = MFMA_F64(a0, bf[0], acc[f][0]);
            acc[f][1] = MFMA_F64(a0, bf[1], acc[f][1]);
            __builtin_amdgcn_sched_barrier(0);
            bf[0] = SK_LDS_F64(ab[kn]); bf[1] = SK_LDS_F64(ab[kn] + FB);
            __builtin_amdgcn_sched_barrier(0);
            acc[f][2] = MFMA_F64(a0, bf[2], acc[f][2]);
            acc[f][3] = MFMA_F64(a0, bf[3], acc[f][3]);
            __builtin_amdgcn_sched_barrier(0);
            bf[2] = SK_LDS_F64(ab[kn] + 2 * FB); bf[3] = SK_LDS_F64(ab[kn] + 3 * FB);
            if (FORM == 8) a1 = SK_LDS_F64(aa[kn]);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (FORM == 8) a0 = a1;
          else { a0 = a1; a1 = a2; if ((f & 1) == 0) { const int f3 = f + 3; a2 = f3 < RF ? SK_LDS_F64(aa[ks] + f3 * FA) : SK_LDS_F64(aa[kn] + (f3 - RF) * FA); } }
        }
      }
    } else {
      // FORM 5: A two groups ahead
      double bf[NFW], a0, a1, a2;
      a0 = SK_LDS_F64(aa[0]); a1 = SK_LDS_F64(aa[0] + FA);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < NFW; ++g) { bf[g] = SK_LDS_F64(ab[0] + g * FB); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int kn = (ks + 1) & 3;
#pragma unroll
        for (int f = 0; f < RF; ++f) {
          const int f2 = f + 2;
          a2 = f2 < RF ? SK_LDS_F64(aa[ks] + f2 * FA) : SK_LDS_F64(aa[kn] + (f2 - RF) * FA);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < NFW; ++g) {
            acc[f][g] = MFMA_F64(a0, bf[g], acc[f][g]);
            if (f == RF - 1) { bf[g] = SK_LDS_F64(ab[kn] + g * FB); __builtin_amdgcn_sched_barrier(0); }
          }
          __builtin_amdgcn_sched_barrier(0);
          a0 = a1; a1 = a2;
        }
      }
    }
  }
  double s = sink;
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int g = 0; g < NFW; ++g) s += acc[f][g][0] + acc[f][g][1] + acc[f][g][2] + acc[f][g][3];
  out[(size_t)blockIdx.x * 256 + t] = s;
}

template <int FORM>
static void run(int ncu, double *d_out, const char *name) {
  const int iters = 4000;
  auto kk = k<FORM>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SK_LDS_BYTES);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(kk, dim3(ncu), dim3(256), SK_LDS_BYTES, 0, 200, d_out);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(kk, dim3(ncu), dim3(256), SK_LDS_BYTES, 0, iters, d_out);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double flops = (double)ncu * 4 * iters * (4.0 * RF * NFW) * 2048.0;
  printf("%-72s %8.3f ms  %6.2f TFLOP/s  %.3f of 78.6\n", name, best, flops / (best * 1e-3) / 1e12, flops / (best * 1e-3) / 78.6e12);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  double *d_out;
  (void)hipMalloc(&d_out, sizeof(double) * (size_t)ncu * 256);
  printf("device %s, %d CUs; one wave per SIMD, 4 x 4 fragments per wave, 64 MFMAs per chunk, no barrier\n", p.name, ncu);
  run<0>(ncu, d_out, "0 sk_mfma_chunk_u (product)");
  run<1>(ncu, d_out, "1 constant operands, no LDS instruction");
  run<2>(ncu, d_out, "2 constant operands + the product's ds_read sequence into dead registers");
  run<3>(ncu, d_out, "3 constant operands + every second ds_read");
  run<4>(ncu, d_out, "4 operands from LDS, the whole next k-step requested a k-step ahead");
  run<5>(ncu, d_out, "5 operands from LDS, A two row-fragment groups ahead");
  run<6>(ncu, d_out, "6 the product's read sequence, accumulators in AccVGPRs");
  run<7>(ncu, d_out, "7 constant operands, accumulators in AccVGPRs");
  run<8>(ncu, d_out, "8 the product's registers, B re-reads in two bursts, A(next k-step) in the second");
  run<1>(ncu, d_out, "1 again");
  run<0>(ncu, d_out, "0 again");
  (void)hipFree(d_out);
  return 0;
}
