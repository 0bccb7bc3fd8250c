// Does v_mfma_f64_16x16x4_f64 issue faster with its accumulator (srcC / vDst) in AGPRs than in VGPRs on gfx950?
// Build: hipcc -O3 --offload-arch=gfx950 scripts/mfma_agpr_probe.hip -o xivo_amd/csrc/build/mfma_agpr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA_A(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define MFMA_V(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
// A / B operands from AGPRs as well
#define MFMA_AA(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b))

template <int NACC, int MODE, int TPB>
__global__ __launch_bounds__(TPB) void k(double* sink, int iters) {
  d4 acc[NACC];
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (MODE == 0) MFMA_V(acc[i], a, b);
      else if (MODE == 1) MFMA_A(acc[i], a, b);
      else MFMA_AA(acc[i], a, b);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) sink[0] = s;
}

// builtin (compiler's choice) with distinct operands per MFMA: a chain like the solve kernel's (B operand = another accumulator)
template <int NACC, int TPB>
__global__ __launch_bounds__(TPB) void kchain(double* sink, int iters) {
  d4 acc[NACC];
  double a = 1.0 + 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{1e-3 * i, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 1; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[0][i & 3], acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) sink[0] = s;
}

template <class K>
static double run(K kern, int blocks, int threads, int iters, double* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, sink, iters / 10);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, sink, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3;
}

int main() {
  double* sink;
  hipMalloc(&sink, 64);
  const int iters = 4000, CU = 256;
  for (int w : {1, 2, 4}) {
    const int blocks = CU * w;
    const double waves = 4.0 * blocks;
    double t;
#define REPORT(name, nmf) printf("%-40s waves/SIMD %d  %8.2f TFLOP/s\n", name, w, (nmf) * 2048.0 * iters * waves / t / 1e12)
    t = run(k<4, 0, 256>, blocks, 256, iters, sink); REPORT("acc VGPR nacc4", 4);
    t = run(k<8, 0, 256>, blocks, 256, iters, sink); REPORT("acc VGPR nacc8", 8);
    t = run(k<16, 0, 256>, blocks, 256, iters, sink); REPORT("acc VGPR nacc16", 16);
    t = run(k<4, 1, 256>, blocks, 256, iters, sink); REPORT("acc AGPR nacc4", 4);
    t = run(k<8, 1, 256>, blocks, 256, iters, sink); REPORT("acc AGPR nacc8", 8);
    t = run(k<16, 1, 256>, blocks, 256, iters, sink); REPORT("acc AGPR nacc16", 16);
    t = run(k<8, 2, 256>, blocks, 256, iters, sink); REPORT("acc AGPR + A operand AGPR nacc8", 8);
    t = run(kchain<9, 256>, blocks, 256, iters, sink); REPORT("builtin chain (B = accumulator) nacc8", 8);
  }
  // the solve kernel's shape: 1024-thread workgroups, 4 waves per SIMD, 128-register budget
  {
    const int w = 4; const int blocks = CU; const double waves = 16.0 * blocks; double t;
    t = run(k<8, 0, 1024>, blocks, 1024, iters, sink); REPORT("1024-thr WG: acc VGPR nacc8", 8);
    t = run(k<8, 1, 1024>, blocks, 1024, iters, sink); REPORT("1024-thr WG: acc AGPR nacc8", 8);
  }
  hipFree(sink);
  return 0;
}
