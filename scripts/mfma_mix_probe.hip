// What slows v_mfma_f64_16x16x4_f64 (accumulators in VGPRs: 77 TFLOP/s alone, scripts/mfma_agpr_probe.hip) down to the ~50 TFLOP/s
// the solve kernel sees? One ingredient at a time, 1024-thread workgroups (4 waves per SIMD), 8 independent accumulators.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/mfma_mix_probe.hip -o xivo_amd/csrc/build/mfma_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA_V(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// MODE 0: plain; 1: one v_xor_b32 per MFMA (operand negation); 2: A operand from LDS (ds_read_b64 issued one MFMA group ahead);
// 3: A operand from LDS, waited for right before use (what the compiler does under register pressure); 4: 2 + xor;
// 5: single accumulator (dependent chain); 6: two accumulators; 7: LDS operand with a v_add_u32 address per read;
// 8: B operand = freshly written accumulator half (the substitution's t[s])
template <int MODE>
__global__ __launch_bounds__(1024) void k(double* sink, int iters) {
  __shared__ double lds[8192];
  d4 acc[8];
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int i = threadIdx.x; i < 8192; i += 1024) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  const double* p = lds + (threadIdx.x & 63);
  unsigned off = (threadIdx.x & 63) * 8;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) MFMA_V(acc[i], a, b);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_xor_b32 %0, 0x80000000, %0" : "+v"(reinterpret_cast<int*>(&a)[1]));
        MFMA_V(acc[i], a, b);
      }
    } else if (MODE == 2 || MODE == 4) {
      double av[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) av[i] = p[64 * i + 512 * (it & 7)];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 4) asm volatile("v_xor_b32 %0, 0x80000000, %0" : "+v"(reinterpret_cast<int*>(&av[i])[1]));
        MFMA_V(acc[i], av[i], b);
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        double a0, a1;
        asm volatile("ds_read_b64 %0, %2 offset:0\n\tds_read_b64 %1, %2 offset:512\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a0), "=&v"(a1) : "v"(off + 1024 * i));
        MFMA_V(acc[i], a0, b);
        MFMA_V(acc[i + 1], a1, b);
      }
    } else if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 8; ++i) MFMA_V(acc[0], a, b);
    } else if (MODE == 6) {
#pragma unroll
      for (int i = 0; i < 8; ++i) MFMA_V(acc[i & 1], a, b);
    } else if (MODE == 7) {
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        double a0, a1; unsigned o0, o1;
        asm volatile("v_add_u32 %2, %4, %5\n\tv_add_u32 %3, %4, %6\n\tds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a0), "=&v"(a1), "=&v"(o0), "=&v"(o1) : "v"(off), "s"(1024 * i), "s"(1024 * i + 512));
        MFMA_V(acc[i], a0, b);
        MFMA_V(acc[i + 1], a1, b);
      }
    } else if (MODE == 8) {
      // t = mfma(.., acc0) then 7 MFMAs whose B operand is a component of t (RAW on the MFMA result)
      MFMA_V(acc[0], a, b);
#pragma unroll
      for (int i = 1; i < 8; ++i) MFMA_V(acc[i], a, acc[0][i & 3]);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) sink[0] = s;
}

template <class K>
static double run(K kern, int blocks, int threads, int iters, double* sink) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, sink, iters / 10);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, sink, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3;
}

int main() {
  double* sink;
  (void)hipMalloc(&sink, 64);
  const int iters = 4000, blocks = 256;
  const double waves = 16.0 * blocks;
  double t;
#define REPORT(name) printf("%-72s %8.2f TFLOP/s\n", name, 8 * 2048.0 * iters * waves / t / 1e12)
  t = run(k<0>, blocks, 1024, iters, sink); REPORT("0 plain, 8 independent accumulators");
  t = run(k<1>, blocks, 1024, iters, sink); REPORT("1 + one v_xor_b32 per MFMA");
  t = run(k<2>, blocks, 1024, iters, sink); REPORT("2 A operand from LDS, 8 reads issued ahead (compiler-scheduled)");
  t = run(k<3>, blocks, 1024, iters, sink); REPORT("3 A operand from LDS, 2 reads + s_waitcnt right before 2 MFMAs");
  t = run(k<4>, blocks, 1024, iters, sink); REPORT("4 = 2 + xor");
  t = run(k<5>, blocks, 1024, iters, sink); REPORT("5 one accumulator (dependent chain)");
  t = run(k<6>, blocks, 1024, iters, sink); REPORT("6 two accumulators");
  t = run(k<7>, blocks, 1024, iters, sink); REPORT("7 = 3 + one v_add_u32 address per read");
  t = run(k<8>, blocks, 1024, iters, sink); REPORT("8 B operand = component of a just-written accumulator");
  (void)hipFree(sink);
  return 0;
}
