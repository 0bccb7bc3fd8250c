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

// factor_invert_diag in FOUR steps of four columns (round 6). The sixteen-pivot chain above pays one matrix-pipe round trip
// per pivot (rank-one MFMA -> v_readlane of the next pivot's two elements -> v_rsq_f64 + two Newton steps -> scale -> next
// MFMA: ~400 cycles per pivot measured inside the one-kernel update, 6.5 k cycles per block). Here the matrix pipe is entered
// twice per FOUR columns:
//   1. the 4 x 4 diagonal sub-block D (ten elements of register x[q], lanes (c + a) + 16 b) is read into scalar registers;
//   2. every lane factors it and inverts the factor with plain VALU arithmetic on those uniform values (four v_rsq_f64 chains,
//      no cross-lane traffic): L44, M = inv(L44);
//   3. the 16 x 4 panel is ONE MFMA: out[n][i] = sum_k M[n][k] X[c + k][i] - the A operand is M placed in the ten lanes
//      (li = n < 4, lg = k <= n), the B operand is the lane's own x[q] - and the result lands in register 0 of lane (i, n):
//      the layout of x[q] itself. The same A operand applied to y[q] gives the four new rows of inv(L);
//   4. the trailing update X -= Lp Lp^T (and Y -= Lp Ytop) is ONE rank-four MFMA each, operands again the lane's own registers.
// Per four columns the chain holds ten v_readlane pairs, the 4 x 4 factorisation (~4 x 75 cycles), the lane selects of M and
// two dependent MFMAs. Same mathematics as factor_invert_diag, different summation order inside a 4-column step (sub-block
// terms first): results agree to rounding, not bit for bit. The one-kernel route is its only user.
// one cubic step on v_rsq_f64's 2^-24 estimate: e = 1 - p rd^2 ~ 1e-7, rd (1 + e/2 + 3 e^2 / 8) leaves e^3 ~ 1e-22 - a shorter dependent
// chain (mul, fma, fma, fma) than two Newton steps (mul, fma, mul, mul, fma, mul)
__device__ __forceinline__ double cubic_rsqrt(double p) {
#pragma clang fp contract(off)
  const double rd = __builtin_amdgcn_rsq(p);
  const double e = __builtin_fma(-(p * rd), rd, 1.0);
  const double t = __builtin_fma(0.375, e, 0.5);
  return __builtin_fma(rd * e, t, rd);
}
template <int VAR>
__device__ __forceinline__ double var_rsqrt(double p) {
  if (VAR & 1) return cubic_rsqrt(p);
  if (VAR & 4) return __builtin_amdgcn_rsq(p);     // (timing experiment only)
  return chain_rsqrt(p);
}
template <int VAR = 0>
__device__ __forceinline__ void factor_invert_diag_blocked(d4& x, d4& y, int& bad, const int row0, const int li, const int lg) {
#pragma clang fp contract(off)
#pragma unroll
  for (int r = 0; r < 4; ++r) y[r] = (lg + 4 * r == li) ? 1.0 : 0.0;
  const d4 zero = d4{0.0, 0.0, 0.0, 0.0};
  static_for<4>([&](auto qc) {
    constexpr int q = decltype(qc)::value, c = 4 * q;
    XIVO_CHAIN_STAMP(row0 == 0 ? 16 + 4 * q : 99);
    // D[a][b] = X[c + a][c + b]: register q of lane li = c + b, lg = a
    const double d00 = readlane_d(x[q], c + 0), d10 = readlane_d(x[q], c + 16), d20 = readlane_d(x[q], c + 32), d30 = readlane_d(x[q], c + 48);
    const double d11 = readlane_d(x[q], c + 1 + 16), d21 = readlane_d(x[q], c + 1 + 32), d31 = readlane_d(x[q], c + 1 + 48);
    const double d22 = readlane_d(x[q], c + 2 + 32), d32 = readlane_d(x[q], c + 2 + 48), d33 = readlane_d(x[q], c + 3 + 48);
    double p0 = d00;
    if (!(p0 > 0.0)) { if (!bad) bad = 1 + row0 + c; p0 = 1.0; }
    double r0, r1, r2, r3, l10, l20, l30, l21, l31, l32;
    if constexpr (VAR & 2) {
      // pivots in pairs: p1 = d11 - d10^2 / p0 = q1 / p0 with q1 = d11 p0 - d10^2 (the same cancellation as the subtraction it
      // replaces), so 1 / sqrt(p1) = rsqrt(q1) sqrt(p0) = rsqrt(q1) p0 r0: the two v_rsq_f64 chains run side by side
      double q1 = __builtin_fma(d11, p0, -(d10 * d10));
      if (!(q1 > 0.0)) { if (!bad) bad = 1 + row0 + c + 1; q1 = 1.0; }
      r0 = var_rsqrt<VAR>(p0);
      const double s1 = var_rsqrt<VAR>(q1);
      r1 = s1 * (p0 * r0);
      l10 = d10 * r0; l20 = d20 * r0; l30 = d30 * r0;
      const double t21 = __builtin_fma(-l20, l10, d21), t31 = __builtin_fma(-l30, l10, d31);
      const double t22 = __builtin_fma(-l20, l20, d22), t32 = __builtin_fma(-l30, l20, d32), t33 = __builtin_fma(-l30, l30, d33);
      l21 = t21 * r1; l31 = t31 * r1;
      double p2 = __builtin_fma(-l21, l21, t22);
      if (!(p2 > 0.0)) { if (!bad) bad = 1 + row0 + c + 2; p2 = 1.0; }
      const double u32 = __builtin_fma(-l31, l21, t32), u33 = __builtin_fma(-l31, l31, t33);
      double q3 = __builtin_fma(u33, p2, -(u32 * u32));
      if (!(q3 > 0.0)) { if (!bad) bad = 1 + row0 + c + 3; q3 = 1.0; }
      r2 = var_rsqrt<VAR>(p2);
      const double s3 = var_rsqrt<VAR>(q3);
      r3 = s3 * (p2 * r2);
      l32 = u32 * r2;
    } else {
      r0 = var_rsqrt<VAR>(p0);
      l10 = d10 * r0; l20 = d20 * r0; l30 = d30 * r0;
      double p1 = __builtin_fma(-l10, l10, d11);
      if (!(p1 > 0.0)) { if (!bad) bad = 1 + row0 + c + 1; p1 = 1.0; }
      const double t21 = __builtin_fma(-l20, l10, d21), t31 = __builtin_fma(-l30, l10, d31);
      const double t22 = __builtin_fma(-l20, l20, d22), t32 = __builtin_fma(-l30, l20, d32), t33 = __builtin_fma(-l30, l30, d33);
      r1 = var_rsqrt<VAR>(p1);
      l21 = t21 * r1; l31 = t31 * r1;
      double p2 = __builtin_fma(-l21, l21, t22);
      if (!(p2 > 0.0)) { if (!bad) bad = 1 + row0 + c + 2; p2 = 1.0; }
      const double u32 = __builtin_fma(-l31, l21, t32), u33 = __builtin_fma(-l31, l31, t33);
      r2 = var_rsqrt<VAR>(p2);
      l32 = u32 * r2;
      double p3 = __builtin_fma(-l32, l32, u33);
      if (!(p3 > 0.0)) { if (!bad) bad = 1 + row0 + c + 3; p3 = 1.0; }
      r3 = var_rsqrt<VAR>(p3);
    }
    // M = inv(L44), row by row (forward substitution on the identity)
    const double m10 = -(l10 * r0) * r1;
    const double m20 = -__builtin_fma(l21, m10, l20 * r0) * r2, m21 = -(l21 * r1) * r2;
    const double m30 = -__builtin_fma(l32, m20, __builtin_fma(l31, m10, l30 * r0)) * r3, m31 = -__builtin_fma(l32, m21, l31 * r1) * r3, m32 = -(l32 * r2) * r3;
    // the A operand: M[li][lg] in the ten lanes li < 4, lg <= li
    double mt = 0.0;
    mt = (li == 0 && lg == 0) ? r0 : mt;
    mt = (li == 1 && lg == 0) ? m10 : mt;
    mt = (li == 1 && lg == 1) ? r1 : mt;
    mt = (li == 2 && lg == 0) ? m20 : mt;
    mt = (li == 2 && lg == 1) ? m21 : mt;
    mt = (li == 2 && lg == 2) ? r2 : mt;
    mt = (li == 3 && lg == 0) ? m30 : mt;
    mt = (li == 3 && lg == 1) ? m31 : mt;
    mt = (li == 3 && lg == 2) ? m32 : mt;
    mt = (li == 3 && lg == 3) ? r3 : mt;
    const d4 pan = mfma(mt, x[q], zero);           // pan[0] of lane (i, n): L[i][c + n] (rows i >= c + n meaningful)
    const d4 ytp = mfma(mt, y[q], zero);           // ytp[0] of lane (j, n): inv(L)[c + n][j]
    const double lp = (li >= c + lg) ? pan[0] : 0.0;
    x[q] = lp;
    y[q] = ytp[0];
    if constexpr (q < 3) {
      const double lb = (li > c + 3) ? lp : 0.0;   // rows below the sub-block
      const double al = -lb;
      x = mfma(al, lb, x);
      y = mfma(al, ytp[0], y);
    }
  });
}

// Second cut of the four-column form, trimmed for the wave's own instruction stream (the chain is issue-bound: ~170
// instructions per step at 4 - 8 cycles each, v_rsq_f64's refinement is only an eighth of it):
//  * inv(L44) column by column in the lanes that need it: lane group lg solves L44 y = e_lg with per-lane constants e (ten
//    operations, the same for every lane) and picks y[li & 3] - instead of sixteen uniform operations and ten lane selects;
//  * the two panel products run on v_mfma_f64_4x4x4_4b_f64 (four passes instead of sixteen: operands A lane 16 k + 4 blk + i,
//    B lane 16 k + 4 blk + j, result lane 16 i + 4 blk + j - scripts/mfma44_layout.hip - i.e. M[li & 3][lg] against the lane's
//    own x[q] gives L[li][c + lg] in place);
//  * a non-positive pivot is only recorded (first one wins); the arithmetic runs on (NaNs at worst: the caller discards the
//    factor of such a filter).
template <int VAR = 0>
__device__ __forceinline__ void factor_invert_diag_blocked2(d4& x, d4& y, int& bad, const int row0, const int li, const int lg) {
#pragma clang fp contract(off)
#pragma unroll
  for (int r = 0; r < 4; ++r) y[r] = (lg + 4 * r == li) ? 1.0 : 0.0;
  const d4 zero = d4{0.0, 0.0, 0.0, 0.0};
  const double e0 = lg == 0 ? 1.0 : 0.0, e1 = lg == 1 ? 1.0 : 0.0, e2 = lg == 2 ? 1.0 : 0.0, e3 = lg == 3 ? 1.0 : 0.0;
  const int n = li & 3;
  static_for<4>([&](auto qc) {
    constexpr int q = decltype(qc)::value, c = 4 * q;
    XIVO_CHAIN_STAMP(row0 == 0 ? 16 + 4 * q : 99);
    const double d00 = readlane_d(x[q], c + 0), d10 = readlane_d(x[q], c + 16), d20 = readlane_d(x[q], c + 32), d30 = readlane_d(x[q], c + 48);
    const double d11 = readlane_d(x[q], c + 1 + 16), d21 = readlane_d(x[q], c + 1 + 32), d31 = readlane_d(x[q], c + 1 + 48);
    const double d22 = readlane_d(x[q], c + 2 + 32), d32 = readlane_d(x[q], c + 2 + 48), d33 = readlane_d(x[q], c + 3 + 48);
    const double p0 = d00;
    double r0, r1, r2, r3, p1, p2, p3, l10, l20, l30, l21, l31, l32, y0, y1, y2, b3;
    if constexpr (VAR & 2) {
      // pivots in pairs: 1 / sqrt(p1) = rsqrt(q1) p0 r0 with q1 = d11 p0 - d10^2 = p1 p0: two v_rsq_f64 chains side by side
      const double q1 = __builtin_fma(d11, p0, -(d10 * d10));
      r0 = var_rsqrt<VAR>(p0);
      const double s1 = var_rsqrt<VAR>(q1);
      p1 = q1;                                             // (sign test only)
      const double h0 = p0 * r0;
      l10 = d10 * r0; l20 = d20 * r0; l30 = d30 * r0;
      r1 = s1 * h0;
      const double t21 = __builtin_fma(-l20, l10, d21), t31 = __builtin_fma(-l30, l10, d31);
      const double t22 = __builtin_fma(-l20, l20, d22), t32 = __builtin_fma(-l30, l20, d32), t33 = __builtin_fma(-l30, l30, d33);
      y0 = e0 * r0;
      l21 = t21 * r1; l31 = t31 * r1;
      p2 = __builtin_fma(-l21, l21, t22);
      const double u32 = __builtin_fma(-l31, l21, t32), u33 = __builtin_fma(-l31, l31, t33);
      const double q3 = __builtin_fma(u33, p2, -(u32 * u32));
      r2 = var_rsqrt<VAR>(p2);
      const double s3 = var_rsqrt<VAR>(q3);
      p3 = q3;
      y1 = __builtin_fma(-l10, y0, e1) * r1;
      const double a2 = __builtin_fma(-l21, y1, __builtin_fma(-l20, y0, e2));
      const double a3 = __builtin_fma(-l31, y1, __builtin_fma(-l30, y0, e3));
      const double h2 = p2 * r2;
      l32 = u32 * r2;
      r3 = s3 * h2;
      y2 = a2 * r2;
      b3 = __builtin_fma(-l32, y2, a3);
    } else {
      r0 = var_rsqrt<VAR>(p0);
      l10 = d10 * r0; l20 = d20 * r0; l30 = d30 * r0;
      p1 = __builtin_fma(-l10, l10, d11);
      const double t21 = __builtin_fma(-l20, l10, d21), t31 = __builtin_fma(-l30, l10, d31);
      const double t22 = __builtin_fma(-l20, l20, d22), t32 = __builtin_fma(-l30, l20, d32), t33 = __builtin_fma(-l30, l30, d33);
      r1 = var_rsqrt<VAR>(p1);
      y0 = e0 * r0;
      l21 = t21 * r1; l31 = t31 * r1;
      p2 = __builtin_fma(-l21, l21, t22);
      const double u32 = __builtin_fma(-l31, l21, t32), u33 = __builtin_fma(-l31, l31, t33);
      r2 = var_rsqrt<VAR>(p2);
      y1 = __builtin_fma(-l10, y0, e1) * r1;
      const double a2 = __builtin_fma(-l21, y1, __builtin_fma(-l20, y0, e2));
      const double a3 = __builtin_fma(-l31, y1, __builtin_fma(-l30, y0, e3));
      l32 = u32 * r2;
      p3 = __builtin_fma(-l32, l32, u33);
      r3 = var_rsqrt<VAR>(p3);
      y2 = a2 * r2;
      b3 = __builtin_fma(-l32, y2, a3);
    }
    const double pre = n == 0 ? y0 : (n == 1 ? y1 : y2);
    const double mt = n == 3 ? b3 * r3 : pre;             // M[li & 3][lg] (0 above the diagonal: e is)
    const bool ok = (p0 > 0.0) && (p1 > 0.0) && (p2 > 0.0) && (p3 > 0.0);
    if (!ok && !bad) bad = 1 + row0 + c + (!(p0 > 0.0) ? 0 : !(p1 > 0.0) ? 1 : !(p2 > 0.0) ? 2 : 3);
    double pan0, ytp0;
    if constexpr (VAR & 8) {
      pan0 = __builtin_amdgcn_mfma_f64_4x4x4f64(mt, x[q], 0.0, 0, 0, 0);
      ytp0 = __builtin_amdgcn_mfma_f64_4x4x4f64(mt, y[q], 0.0, 0, 0, 0);
    } else {
      const double mz = li < 4 ? mt : 0.0;
      pan0 = mfma(mz, x[q], zero)[0];
      ytp0 = mfma(mz, y[q], zero)[0];
    }
    const double lp = (li >= c + lg) ? pan0 : 0.0;
    x[q] = lp;
    y[q] = ytp0;
    if constexpr (q < 3) {
      const double lb = (li > c + 3) ? lp : 0.0;   // rows below the sub-block
      const double al = -lb;
      x = mfma(al, lb, x);
      y = mfma(al, ytp0, y);
    }
  });
}

}  // namespace

}  // namespace xivo_hip
