// Empirical operand / result layout of v_mfma_f64_4x4x4_4b_f64 on gfx950: for every pair (la, lb) with A = e_la, B = e_lb
// (unit vectors over the 64 lanes) print which result lanes are non-zero.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}
int main() {
  double *dA, *dB, *dD, hA[64], hB[64], hD[64];
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      for (int i = 0; i < 64; ++i) { hA[i] = i == la; hB[i] = i == lb; }
      hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
      hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
      for (int i = 0; i < 64; ++i) if (hD[i] != 0.0) printf("%d %d %d\n", la, lb, i);
    }
  return 0;
}
