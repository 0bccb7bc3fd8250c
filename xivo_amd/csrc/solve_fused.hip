// The ten- / twelve-wave instantiations of the whitened in-solve Joseph update (trsm_lds_kernel.h) for a factor of seven block
// rows on a narrow state (BASELINE config 2 on the multi-kernel pipeline: N = 150, M = 100):
// `K_.transpose() = S_.ldlt().solve(H_ * P_)` ... `P_ += K_ * K_.transpose()` (/root/reference/src/estimator.cpp:1265-1287) once S is
// factored. ONE ten-wave (N <= 160) or twelve-wave (N <= 192) workgroup per CU on 170 VGPRs per wave: seven block rows of
// right-hand sides (56 VGPRs) AND of W (56) stay in registers - no stash, no read-back, no operand DMA (the KEEPW form the
// kernel has for factors of at most six block rows on 128 VGPRs): 3.72 against 4.10 ms per 16384 filters for the sixteen-wave
// kernel (round 5; two ten-wave workgroups per CU with the stash: 4.32 ms - that shape is bound by HBM; the factorisation
// inside this kernel: 5.83 against 4.09 + 0.80 ms - both measured, recorded in docs/HISTORY.md and removed in round 6).
// A translation unit of its own: the instantiations of the solve kernel take minutes to compile (chol_trsm.hip holds the rest).
// Since round 6 these shapes default to the one-kernel update (fused_update.hip); this is what XIVO_HIP_FLAG_MULTI_KERNEL runs.
#include "trsm_lds_kernel.h"

namespace xivo_hip {

namespace {

template <int NBM, int NWV, int MINB>
int launch_narrow_t(const TrsmArgs& g_in, hipStream_t stream) {
  TrsmArgs g = g_in;
  const int nb = g.Mp / 16;
  const int grid = ((g.batch + 7) / 8) * 8;
  constexpr int WGS = 4 * MINB / NWV;              // workgroups per CU the register budget allows (MINB = waves per SIMD)
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
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_lds_f64_kernel<NBM, 4, NWV, MINB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
    attr_set = true;
  }
  hipLaunchKernelGGL((trsm_lds_f64_kernel<NBM, 4, NWV, MINB>), dim3(grid), dim3(64 * NWV), lds, stream, g);
  return (int)hipGetLastError();
}

}  // namespace

// shapes the narrow instantiations hold: every column of the state in one NWV-wave workgroup, a factor of seven block rows
bool trsm_narrow_supported(int Mp, int Np) { return Np % 16 == 0 && Mp / 16 == 7 && Np <= 192; }
int launch_trsm_narrow(const TrsmArgs& g, hipStream_t stream) {
  if (g.batch <= 0) return 0;
  return g.Np <= 160 ? launch_narrow_t<7, 10, 3>(g, stream) : launch_narrow_t<7, 12, 3>(g, stream);
}
void trsm_narrow_label(int Mp, int Np, char* buf, size_t n) { snprintf(buf, n, "trsm_lds_f64_kernel<7,4,%d,3>", Np <= 160 ? 10 : 12); }

}  // namespace xivo_hip
