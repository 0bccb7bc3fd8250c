// accuracy of v_rsq_f64 and of one / two Newton steps on it (gfx950): max relative error over random inputs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
__global__ void k(const double* in, double* o0, double* o1, double* o2, int n) {
#pragma clang fp contract(off)
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  double p = in[i];
  double rd = __builtin_amdgcn_rsq(p);
  o0[i] = rd;
  const double hx = 0.5 * p;
  rd = rd * __builtin_fma(-(hx * rd), rd, 1.5);
  o1[i] = rd;
  rd = rd * __builtin_fma(-(hx * rd), rd, 1.5);
  o2[i] = rd;
}
int main() {
  const int n = 1 << 20;
  double* h = (double*)malloc(n * 8); srand(1);
  for (int i = 0; i < n; ++i) h[i] = exp(((double)rand() / RAND_MAX - 0.5) * 60.0);
  double *d, *o[3]; hipMalloc(&d, n * 8); for (int j = 0; j < 3; ++j) hipMalloc(&o[j], n * 8);
  hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o[0], o[1], o[2], n);
  double* r = (double*)malloc(n * 8);
  for (int j = 0; j < 3; ++j) {
    hipMemcpy(r, o[j], n * 8, hipMemcpyDeviceToHost);
    long double worst = 0;
    for (int i = 0; i < n; ++i) { long double t = 1.0L / sqrtl((long double)h[i]); long double e = fabsl(((long double)r[i] - t) / t); if (e > worst) worst = e; }
    printf("newton steps %d: max rel err %.3Le (%.2Lf ulp)\n", j, worst, worst / 1.1102230246251565e-16L);
  }
  return 0;
}
