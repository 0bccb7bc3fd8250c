// The 16 x 16 factor-and-invert step of the one-kernel update, alone on a CU: cycles per block for the sixteen-pivot chain
// (factor_invert_diag_chain) and the four-column blocked form (factor_invert_diag_blocked), and their error against a host
// Cholesky. Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I xivo_amd/csrc -I include scripts/probes/diag_chain_probe.hip -o /tmp/diag_probe
#include "chol_device.h"
#include <stdio.h>
#include <math.h>
#include <vector>
using namespace xivo_hip;
template <int MODE>
__global__ void k(const double* A, double* Lout, double* Yout, long long* cyc, int* badout) {
  const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
  d4 x0;
  for (int r = 0; r < 4; ++r) x0[r] = A[(lg + 4 * r) * 16 + li];
  d4 x, y; int bad = 0;
  double chk = 0.0;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 32; ++it) {
    x = x0;
    for (int r = 0; r < 4; ++r) x[r] += chk * 1e-300;
    if (MODE == 0) factor_invert_diag(x, y, bad, 0, li, lg);
    else if (MODE == 1) factor_invert_diag_chain(x, y, bad, 0, li, lg);
    else if (MODE < 16) factor_invert_diag_blocked<MODE - 2>(x, y, bad, 0, li, lg);
    else factor_invert_diag_blocked2<MODE - 16>(x, y, bad, 0, li, lg);
    chk += x[0] + y[3];
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < 4; ++r) { Lout[(lg + 4 * r) * 16 + li] = x[r]; Yout[(lg + 4 * r) * 16 + li] = y[r]; }   // x[r] = L[li][lg+4r] -> stored transposed: Lout[c*16+li]
  if (lane == 0) { cyc[0] = t1 - t0; badout[0] = bad; }
}
int main() {
  double A[256], L[256] = {0}, G[256];
  unsigned s = 12345;
  for (int i = 0; i < 256; ++i) { s = s * 1664525u + 1013904223u; G[i] = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double a = (i == j) ? 2.0 : 0.0; for (int k = 0; k < 16; ++k) a += G[i * 16 + k] * G[j * 16 + k]; A[i * 16 + j] = a; }
  for (int j = 0; j < 16; ++j) {
    double d = A[j * 16 + j]; for (int k = 0; k < j; ++k) d -= L[j * 16 + k] * L[j * 16 + k];
    L[j * 16 + j] = sqrt(d);
    for (int i = j + 1; i < 16; ++i) { double v = A[i * 16 + j]; for (int k = 0; k < j; ++k) v -= L[i * 16 + k] * L[j * 16 + k]; L[i * 16 + j] = v / L[j * 16 + j]; }
  }
  double *dA, *dL, *dY; long long* dc; int* db;
  hipMalloc(&dA, 2048); hipMalloc(&dL, 2048); hipMalloc(&dY, 2048); hipMalloc(&dc, 8); hipMalloc(&db, 4);
  hipMemcpy(dA, A, 2048, hipMemcpyHostToDevice);
  const char* names[] = {"factor_invert_diag (loop, rounds 2-5)", "factor_invert_diag_chain (16 pivots)", "factor_invert_diag_blocked (4 x 4 columns)", "  blocked, cubic rsqrt step", "  blocked, pivots in pairs", "  blocked, pairs + cubic", "  blocked, raw v_rsq (timing only)", "", "  blocked, pairs + raw v_rsq (timing only)", "","","","","","","",
  "blocked2 (column-wise inverse, lazy bad)", "  blocked2 + cubic rsqrt", "","","  blocked2 raw v_rsq (timing only)","","","",
  "  blocked2 + 4x4x4 panels", "  blocked2 + 4x4x4 panels + cubic", "  blocked2 + 4x4x4 + pairs", "  blocked2 + 4x4x4 + pairs + cubic"};
#define RUN(M) { for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, dA, dL, dY, dc, db); \
    long long c; int b; double Lg[256], Yg[256]; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); hipMemcpy(&b, db, 4, hipMemcpyDeviceToHost); \
    hipMemcpy(Lg, dL, 2048, hipMemcpyDeviceToHost); hipMemcpy(Yg, dY, 2048, hipMemcpyDeviceToHost); \
    double eL = 0, eI = 0; \
    for (int i = 0; i < 16; ++i) for (int j = 0; j <= i; ++j) eL = fmax(eL, fabs(Lg[j * 16 + i] - L[i * 16 + j]) / fabs(L[i * 16 + i])); \
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double v = 0; for (int k2 = 0; k2 < 16; ++k2) v += (k2 <= i ? L[i * 16 + k2] : 0.0) * Yg[k2 * 16 + j]; eI = fmax(eI, fabs(v - (i == j))); } \
    printf("%-45s %7.0f cycles per block   max |L - L_host| / L_ii = %.2e   max |L_host Y - I| = %.2e   bad = %d\n", names[M], c / 32.0, eL, eI, b); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(16) RUN(17) RUN(20) RUN(24) RUN(25) RUN(26) RUN(27)
  return 0;
}
