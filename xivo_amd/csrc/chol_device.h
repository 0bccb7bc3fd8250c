// Device routines shared by the Cholesky kernels (chol_f64.hip) and the solve kernel that factors S itself
// (trsm_lds_kernel.h, CHOL = true): the pivot arithmetic and the 16 x 16 diagonal-block factor-and-invert step. Every kernel
// that factors S uses exactly these, with the same operand order around them, so that they all produce the SAME bits.
#pragma once
#include "mfma_util.h"

#ifndef XIVO_CHAIN_STAMP
#define XIVO_CHAIN_STAMP(slot) do {} while (0)   // (trace builds of the one-kernel update: shader-clock stamp per pivot)
#endif
#ifndef XIVO_CHOL_UNROLL16
#define XIVO_CHOL_UNROLL16 0
#endif

namespace xivo_hip {

namespace {

// d = sqrt(p) and rd = 1 / sqrt(p) of a pivot: hardware estimate + two Newton steps, d = p * rd with one correction - no
// sqrt / divide in the serial chain. BOTH Cholesky kernels use this routine and the same operand order everywhere else
// (two accumulators over the k-slices of a block product, inverse rows scaled by rd), so that they produce the SAME bits:
// which of them a node runs faster (chol_pick in capi.hip) then changes the time, never the result.
__device__ __forceinline__ void pivot_scale(double p, double& d, double& rd) {
#pragma clang fp contract(off)
  rd = __builtin_amdgcn_rsq(p);
  const double hx = 0.5 * p;
  rd = rd * __builtin_fma(-(hx * rd), rd, 1.5);
  rd = rd * __builtin_fma(-(hx * rd), rd, 1.5);
  d = p * rd;
  d = __builtin_fma(__builtin_fma(-d, d, p), 0.5 * rd, d);
}

// Factor AND invert one 16x16 diagonal block, held by one wave in the C/D layout of v_mfma_f64_16x16x4_f64:
// x[r] of lane (li, lg) is X[li][lg + 4 r] on entry (X symmetric up to rounding; only X[i][c], i >= c, is consumed - the
// same elements the round-2 routines read). On return x[r] = L[li][lg + 4 r] (for lg + 4 r <= li) and
// y[r] = inv(L)[lg + 4 r][li].
// Column c of the right-looking factorisation, and column c of the forward substitution L Y = I, are RANK-ONE updates:
//   X[i][p] -= L[i][c] L[p][c]   (i, p > c)            Y[i][j] -= L[i][c] Y[c][j]   (i > c)
// and both run on the matrix pipe with no cross-lane traffic at all: in the C/D layout "column c of L" is the register
// x[c >> 2] of the lanes lg == (c & 3), indexed by li - which is the A (and B) operand of k-slice (c & 3) as it stands - and
// row c of Y is y[c >> 2] of the same lanes. The three other k-slices are fed exact zeros, so every element receives exactly
// one fused multiply-add per column, in ascending column order: the arithmetic of the round-2 routines (v_readlane /
// ds_bpermute broadcasts + v_fma, ~2200 instructions and 7.8 us per block on a lone wave) in ~40 instructions per column.
// What is left in the serial chain per column: one v_readlane pair (the pivot), pivot_scale, one multiply, one MFMA.
__device__ __forceinline__ void factor_invert_diag(d4& x, d4& y, int& bad, const int row0, const int li, const int lg) {
#pragma unroll
  for (int r = 0; r < 4; ++r) y[r] = (lg + 4 * r == li) ? 1.0 : 0.0;
  // columns c = 4 rc + lgc: the register index rc is unrolled, the k-slice lgc is a run-time loop (the lane selects of
  // v_readlane and the lane masks are scalar values anyway) - a quarter of the code of sixteen unrolled columns. The register
  // kernel is straight-line code executed once per factor; its size is what the instruction fetch of a CU pair sees
  // (64 KB of instruction cache): ten block columns of sixteen unrolled columns each were 76 KB and ran 1.6x slower on
  // some nodes of the pool than on others.
#pragma unroll
  for (int rc = 0; rc < 4; ++rc) {
#if XIVO_CHOL_UNROLL16     // A/B build (scripts/ab_chol.sh): all sixteen columns unrolled
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int lgc = 0; lgc < 4; ++lgc) {
      const int c = 4 * rc + lgc;
      double dcc = readlane_d(x[rc], c + 16 * lgc);
      if (!(dcc > 0.0)) {
        if (!bad) bad = 1 + row0 + c;
        dcc = 1.0;
      }
      double d, rd;
      pivot_scale(dcc, d, rd);
      const bool own = (lg == lgc);
      const double lc = x[rc] * rd;            // L[li][c] in the lanes lg == lgc
      const double yc = y[rc] * rd;            // row c of inv(L): final
      if (own) { x[rc] = (li == c) ? d : lc; y[rc] = yc; }
      // (column 15 has no rows below it: its two products are exact zeros)
      const bool below = own && li > c;
      const double bl = below ? lc : 0.0;
      const double al = -bl;
      const double by = own ? yc : 0.0;
      x = mfma(al, bl, x);
      y = mfma(al, by, y);
    }
  }
}

// factor_invert_diag for a chain that runs ALONE on its SIMD (round 6, the one-kernel update: one workgroup per CU, nothing
// else to issue while the diagonal block is factored). There the sixteen pivots cost what the wave's own in-order
// instruction stream costs: factor_invert_diag's loop body is ~50 instructions and three taken branches per pivot, and the
// next pivot is read (v_readlane, ~30 cycles to a scalar register) from the result of the rank-one MFMA (~94 cycles from
// issue to a VALU consumer) before pivot_scale can even start - 460 cycles per pivot measured (scripts/probes/latency_probe.hip:
// dependent v_fma_f64 7.6, v_rsq_f64 ~18, v_readlane -> VALU 38, MFMA -> VALU -> MFMA 93 cycles per step).
// This variant is straight-line code for all sixteen columns (compile-time lane selects, no branches), software-pipelined
// by one pivot:
//  * the next pivot does not wait for the matrix pipe's update of the whole block and a second pass: the two elements column
//    c's MFMA would combine into it - X[c + 1][c] and X[c + 1][c + 1] as columns 0 .. c - 1 left them - are read once, and the
//    pivot is formed by the same single fused multiply-add the matrix pipe applies to that element (its other k-slices are
//    exact zeros): p' = fma(-l, l, X[c + 1][c + 1]), l = X[c + 1][c] rd;
//  * v_rsq_f64 and the two refinement steps of pivot c + 1 are issued in front of column c's vector work (scaling, lane
//    selects, the two MFMAs), which then fills the latency slots of that dependent chain;
//  * the diagonal entry is L[c][c] = p rd like every other entry of the column (factor_invert_diag corrects it to the
//    rounded square root with four more dependent operations; p rd is within 2 ulp of it, and the inverse block is built
//    from the same rd, so L inv(L) = I holds to the same rounding either way).
// Not bit-identical to factor_invert_diag (the diagonal entries differ in the last place): the one-kernel route is its only
// user. One copy per kernel (its caller loops over the block columns at run time): ~700 instructions.
__device__ __forceinline__ double chain_rsqrt(double p) {     // v_rsq_f64 (2^-24) + two Newton steps: 2 ulp (scripts/probes/rsq_probe.hip)
#pragma clang fp contract(off)
  double rd = __builtin_amdgcn_rsq(p);
  const double hx = 0.5 * p;
  rd = rd * __builtin_fma(-(hx * rd), rd, 1.5);
  rd = rd * __builtin_fma(-(hx * rd), rd, 1.5);
  return rd;
}
__device__ __forceinline__ void factor_invert_diag_chain(d4& x, d4& y, int& bad, const int row0, const int li, const int lg) {
#pragma clang fp contract(off)
#pragma unroll
  for (int r = 0; r < 4; ++r) y[r] = (lg + 4 * r == li) ? 1.0 : 0.0;
  double p = readlane_d(x[0], 0);
  if (!(p > 0.0)) { bad = 1 + row0; p = 1.0; }
  double rd = chain_rsqrt(p);
  static_for<16>([&](auto cc) {
    constexpr int c = decltype(cc)::value, rc = c >> 2, lgc = c & 3;
    XIVO_CHAIN_STAMP(row0 == 0 ? 16 + c : 99);
    double rn = 0.0;
    if constexpr (c < 15) {
      // X[c + 1][c] lives in lane (c + 1, lg = c & 3) of register c >> 2, X[c + 1][c + 1] in lane (c + 1, (c + 1) & 3) of register (c + 1) >> 2
      const double x10 = readlane_d(x[rc], (c + 1) + 16 * lgc);
      const double x11 = readlane_d(x[(c + 1) >> 2], (c + 1) + 16 * ((c + 1) & 3));
      const double l1 = x10 * rd;
      double pn = __builtin_fma(-l1, l1, x11);
      const bool ok = pn > 0.0;
      if (!ok && !bad) bad = 1 + row0 + c + 1;
      pn = ok ? pn : 1.0;
      rn = chain_rsqrt(pn);
    }
    const bool own = (lg == lgc);
    const double lc = x[rc] * rd;            // L[li][c] in the lanes lg == lgc (li == c: p rd)
    const double yc = y[rc] * rd;            // row c of inv(L): final
    if (own) { x[rc] = lc; y[rc] = yc; }
    const bool below = own && li > c;
    const double bl = below ? lc : 0.0;
    const double al = -bl;
    const double by = own ? yc : 0.0;
    x = mfma(al, bl, x);
    y = mfma(al, by, y);
    rd = rn;
  });
}

}  // namespace

}  // namespace xivo_hip
