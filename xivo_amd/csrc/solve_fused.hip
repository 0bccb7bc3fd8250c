// The whitened Joseph update with the Cholesky factorisation of S inside the solve kernel (trsm_lds_kernel.h, CHOL = true):
// `K_.transpose() = S_.ldlt().solve(H_ * P_)` ... `P_ += K_ * K_.transpose()` (/root/reference/src/estimator.cpp:1265-1287)
// in ONE kernel per filter once S is formed. A translation unit of its own: the instantiations of the solve kernel take
// minutes to compile (chol_trsm.hip holds eighteen of them).
#include "trsm_lds_kernel.h"

namespace xivo_hip {

namespace {

template <int NBM>
int launch_fused_t(const TrsmArgs& g_in, hipStream_t stream) {
  TrsmArgs g = g_in;
  const int nb = g.Mp / 16;
  const int grid = ((g.batch + 7) / 8) * 8;
  const size_t lds = 160 * 1024;
  g.t_jbp = (int)(lds / 2 / ((size_t)nb * 4 * 64 * sizeof(double)));   // two operand buffers in the product phase
  if (g.t_jbp > 16) g.t_jbp = 16;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_lds_f64_kernel<NBM, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((trsm_lds_f64_kernel<NBM, 4, true>), dim3(grid), dim3(1024), lds, stream, g);
  return (int)hipGetLastError();
}

// Short factors on narrow states: NWV-wave workgroups, MINB of them per CU (trsm_lds_kernel.h)
// (MINB = waves per SIMD the register budget is cut for - HIP's second __launch_bounds__ argument; WGS = workgroups per CU
//  that makes: 4 MINB / NWV)
template <int NBM, int NWV, int MINB>
int launch_narrow_t(const TrsmArgs& g_in, hipStream_t stream) {
  TrsmArgs g = g_in;
  const int nb = g.Mp / 16;
  const int grid = ((g.batch + 7) / 8) * 8;
  constexpr int WGS = 4 * MINB / NWV;
  // LDS per workgroup: the factor (diagonal blocks in slots of their own) or the two operand buffers of the product
  // phase, whichever is larger - within the CU's 160 KiB / WGS
  const size_t cap = (size_t)(160 * 1024 / WGS) & ~(size_t)1023;
  const size_t factor = ((size_t)nb * (nb + 1) / 2 + nb) * 16 * 17 * sizeof(double) + 64;
  g.t_jbp = (int)(cap / 2 / ((size_t)nb * 4 * 64 * sizeof(double)));
  if (g.t_jbp > 16) g.t_jbp = 16;
  if (g.t_jbp < 1 || factor > cap) return (int)hipErrorInvalidValue;
  const size_t prod = (size_t)2 * g.t_jbp * nb * 4 * 64 * sizeof(double);
  const size_t lds = factor > prod ? factor : prod;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_lds_f64_kernel<NBM, 4, false, NWV, MINB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
    attr_set = true;
  }
  hipLaunchKernelGGL((trsm_lds_f64_kernel<NBM, 4, false, NWV, MINB>), dim3(grid), dim3(64 * NWV), lds, stream, g);
  return (int)hipGetLastError();
}

}  // namespace

// shapes the narrow instantiations hold: every column of the state in one NWV-wave workgroup, a factor of seven block rows.
// Round 5, two forms (profiles/r05_narrow_solve_ab.json; BASELINE config 2: N = 150, M = 100, 16384 filters; the sixteen-wave
// kernel <10,4> - ten live waves, stash of W through HBM - takes 4.10 ms there):
//   mode 2 (DEFAULT): ONE ten-wave (N <= 160) or twelve-wave (N <= 192) workgroup per CU on 170 VGPRs per wave: seven block
//     rows of right-hand sides (56 VGPRs) AND of W (56) stay in registers - no stash, no read-back, no operand DMA (the KEEPW
//     form the kernel already has for factors of at most six block rows on 128 VGPRs): 3.72 ms, config 2 1.96 -> 2.06 M updates/s.
//   mode 1 (XIVO_HIP_NARROW_SOLVE=1): TWO ten-wave workgroups per CU on 96 VGPRs, stash as before - the memory phases of one
//     filter under the matrix phases of another: 4.32 ms (that shape is bound by HBM, the second workgroup adds requests,
//     not bandwidth; 80 KB of LDS per workgroup cut the product phase to two column blocks per pass).
//   XIVO_HIP_NARROW_SOLVE=0: the sixteen-wave kernel. Same bits in every mode (tests/test_update_gpu.py).
static int narrow_mode() {
  static const int m = [] { const char* e = getenv("XIVO_HIP_NARROW_SOLVE"); return e ? (atoi(e) == 2 ? 2 : (atoi(e) == 1 ? 1 : 0)) : 2; }();
  return m;
}
bool trsm_narrow_supported(int Mp, int Np) {
  const int m = narrow_mode();
  return m != 0 && Np % 16 == 0 && Mp / 16 == 7 && Np <= (m == 2 ? 192 : 160);
}
int launch_trsm_narrow(const TrsmArgs& g, hipStream_t stream) {
  if (g.batch <= 0) return 0;
  if (narrow_mode() == 1) return launch_narrow_t<7, 10, 5>(g, stream);
  return g.Np <= 160 ? launch_narrow_t<7, 10, 3>(g, stream) : launch_narrow_t<7, 12, 3>(g, stream);
}
void trsm_narrow_label(int Mp, int Np, char* buf, size_t n) {
  if (narrow_mode() == 1) snprintf(buf, n, "trsm_lds_f64_kernel<7,4,false,10,5>");
  else snprintf(buf, n, "trsm_lds_f64_kernel<7,4,false,%d,3>", Np <= 160 ? 10 : 12);
}

// shapes the fused kernel holds: the whole state in one 16-wave workgroup, a factor of at most ten block rows
// OPT-IN (XIVO_HIP_FUSED_CHOL=1), measured and not adopted: per 16384 filters at (250, 160) the solve grows from 9.6 to 12.2 ms
// while the stand-alone Cholesky it replaces costs 1.7 ms (config 2: 4.09 -> 5.83 against 0.80; TUM-VI size: 1.69 -> 2.12
// against 0.16). The factorisation is a serial chain (ten diagonal blocks, sixteen dependent columns each) that one
// workgroup per CU has nothing to hide behind - ~85 k cycles per filter - whereas the stand-alone kernel keeps three factors in
// flight per CU (26.7 us of CU time per factor). Same bits either way (tests/test_update_gpu.py).
bool trsm_chol_fused_supported(int Mp, int Np) {
  static const bool on = getenv("XIVO_HIP_FUSED_CHOL") != nullptr;
  return on && Mp / 16 <= 10 && Np <= 256 && Np % 16 == 0;
}

// g.LU = S (lower triangle + diagonal blocks), g.invD unused, g.T = the covariance (updated in place), g.joseph = 2,
// g.chol_status = per-filter factorisation status (out)
int launch_trsm_chol_fused(const TrsmArgs& g, hipStream_t stream) {
  if (g.batch <= 0) return 0;
  const int nb = g.Mp / 16;
  if (nb <= 6) return launch_fused_t<6>(g, stream);
  if (nb <= 10) return launch_fused_t<10>(g, stream);
  return (int)hipErrorInvalidValue;
}

void trsm_chol_fused_label(int Mp, char* buf, size_t n) { snprintf(buf, n, "trsm_lds_f64_kernel<%d,4,true>", Mp / 16 <= 6 ? 6 : 10); }

}  // namespace xivo_hip
