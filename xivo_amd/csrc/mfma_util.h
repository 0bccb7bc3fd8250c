// Small device helpers shared by the factorisation (chol_f64.hip) and the solve kernels (chol_trsm.hip).
#pragma once
#include "common.h"
#include <utility>

namespace xivo_hip {

namespace {

__device__ __forceinline__ double readlane_d(double v, int srclane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

template <class F, int... Js>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Js...>) {
  (f(std::integral_constant<int, Js>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// Barrier for exchanges that go through LDS only: __syncthreads() also drains the vector-memory counter, i.e. waits
// until every global store issued so far is acknowledged - microseconds per barrier that nothing here depends on.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace

}  // namespace xivo_hip
