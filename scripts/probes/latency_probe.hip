// dependent-issue latencies on gfx950 (one wave per SIMD, nothing else on the CU): cycles per operation of a chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(double* out, long long* cyc, double seed) {
  double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-9;
  d4 acc = d4{a, a, a, a};
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) a = __builtin_fma(a, b, c);
      else if (MODE == 1) a = a * b;
      else if (MODE == 2) a = __builtin_amdgcn_rsq(a) + 1.0;
      else if (MODE == 3) { int lo = __builtin_amdgcn_readlane(__double2loint(a), 5); a = a + __hiloint2double(0x3ff00000, lo & 1); }
      else if (MODE == 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      else if (MODE == 5) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); a = acc[0] * b; }
      else if (MODE == 6) { float f = (float)a; f = __builtin_fmaf(f, 1.0000001f, 1e-9f); a = (double)f; }
      else if (MODE == 7) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); int lo = __builtin_amdgcn_readlane(__double2loint(acc[0]), 5); a = __hiloint2double(0x3ff00000, lo); }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = a + acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 64 * 8 * 8); hipMalloc(&cyc, 64);
  const char* names[] = {"v_fma_f64 chain", "v_mul_f64 chain", "v_rsq_f64 + v_add_f64 chain", "v_readlane + v_add chain", "mfma_f64_16x16x4 chain (acc dependent)", "mfma -> v_mul_f64 -> mfma chain", "cvt f64->f32, v_fma_f32, cvt back chain", "mfma -> readlane -> mfma(A operand) chain"};
#define RUN(M) { hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-45s %.1f cycles per step\n", names[M], c / 1024.0); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
  return 0;
}
