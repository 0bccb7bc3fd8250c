// Issue-rate probes for the fp64 pipes of gfx950 (MI355X): what the matrix core and the vector ALU sustain for the
// shapes the EKF kernels use. Build: hipcc -O3 --offload-arch=gfx950 scripts/mfma_probe.hip -o xivo_amd/csrc/build/mfma_probe
// Prints one line per probe: name, waves per SIMD, TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_f64_16(double* sink, int iters) {
  d4 acc[NACC];
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) sink[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_f64_4(double* sink, int iters) {
  double acc[NACC];
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s == 12345.678) sink[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_f32_16(double* sink, int iters) {
  f4 acc[NACC];
  const float a = 1.0f + 1e-6f * threadIdx.x, b = 1.0f - 1e-6f * threadIdx.x;
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) sink[0] = s;
}

// vector ALU: NACC independent v_fma_f64 chains per lane
template <int NACC>
__global__ __launch_bounds__(256) void k_valu_f64(double* sink, int iters) {
  double acc[NACC];
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.1 * i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  if (s == 12345.678) sink[0] = s;
}

// both pipes from one wave: 8 MFMA + NV v_fma_f64 per trip (do they overlap?)
template <int NV>
__global__ __launch_bounds__(256) void k_mix_f64(double* sink, int iters) {
  d4 acc[8];
  double v[NV];
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 0.1 * i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV / 8; ++j) v[i * (NV / 8) + j] = __builtin_fma(v[i * (NV / 8) + j], a, b);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i];
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
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms * 1e-3;
}

int main() {
  double* sink;
  hipMalloc(&sink, 64);
  const int iters = 4000;
  const int CU = 256;
  // waves per SIMD w: blocks of 256 threads (4 waves, one per SIMD) x w per CU
  for (int w : {1, 2, 4, 8}) {
    const int blocks = CU * w;
    const double waves = 4.0 * blocks;
    double t;
#define REPORT(name, flop_per_wave_iter)                                                                  \
    printf("%-28s waves/SIMD %d  %8.2f TFLOP/s\n", name, w, (flop_per_wave_iter) * iters * waves / t / 1e12)
    t = run(k_mfma_f64_16<4>, blocks, 256, iters, sink);  REPORT("mfma_f64_16x16x4 nacc4", 4 * 2048.0);
    t = run(k_mfma_f64_16<8>, blocks, 256, iters, sink);  REPORT("mfma_f64_16x16x4 nacc8", 8 * 2048.0);
    t = run(k_mfma_f64_16<16>, blocks, 256, iters, sink); REPORT("mfma_f64_16x16x4 nacc16", 16 * 2048.0);
    t = run(k_mfma_f64_4<8>, blocks, 256, iters, sink);   REPORT("mfma_f64_4x4x4 nacc8", 8 * 512.0);
    t = run(k_mfma_f32_16<8>, blocks, 256, iters, sink);  REPORT("mfma_f32_16x16x4 nacc8", 8 * 2048.0);
    t = run(k_valu_f64<8>, blocks, 256, iters, sink);     REPORT("valu_fma_f64 nacc8", 8 * 128.0);
    t = run(k_valu_f64<16>, blocks, 256, iters, sink);    REPORT("valu_fma_f64 nacc16", 16 * 128.0);
    t = run(k_mix_f64<16>, blocks, 256, iters, sink);     REPORT("mix 8 mfma + 16 vfma", 8 * 2048.0 + 16 * 128.0);
    t = run(k_mix_f64<64>, blocks, 256, iters, sink);     REPORT("mix 8 mfma + 64 vfma", 8 * 2048.0 + 64 * 128.0);
    t = run(k_mix_f64<128>, blocks, 256, iters, sink);    REPORT("mix 8 mfma + 128 vfma", 8 * 2048.0 + 128 * 128.0);
  }
  hipFree(sink);
  return 0;
}
