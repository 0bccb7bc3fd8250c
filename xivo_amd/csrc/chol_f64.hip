// Batched SPD factorisation S = L L^T for the EKF measurement update (gfx950 / MI355X): the factor behind
// `K_.transpose() = S_.ldlt().solve(H_ * P_)` (/root/reference/src/estimator.cpp:1265-1267). The reference uses Eigen's
// pivoted LDL^T; S = HPH^T + R is SPD, so an un-pivoted blocked Cholesky is the same linear map up to rounding (a filter
// whose S is not SPD is flagged and answered by ldlt_fallback.hip). The solve kernels are in chol_trsm.hip.
//
// chol_f64_kernel : one wave64 per filter, left-looking, 16x16 blocks (the v_mfma_f64_16x16x4_f64 tile); its explicit
//   inverse of every diagonal block is kept so every panel / triangular-solve step is an MFMA.
//   Output: L in the lower triangle, L^T mirrored into the upper triangle.
// chol_reg_f64_kernel : four waves per filter, the factor in registers (M <= 192).
// Both factor and invert a diagonal block with the SAME routine (factor_invert_diag) and use the same operand order
// everywhere else, so they produce the same bits: which of them a node runs faster changes the time, never the result.
#include <stdlib.h>
#include <stdio.h>

#include "mfma_util.h"
#include "chol_device.h"

// The diagonal-block step of both Cholesky kernels: the four-column form (chol_device.h, round 6; chol_S 1.95 -> 1.86 ms per 16384 at
// M = 160, 2.65 -> 2.48 per 4096 at M = 300, 0.27 -> 0.25 at M = 120) - or, -DXIVO_CHOL_BLOCKED=0, the sixteen-column loop of rounds 3-5
#ifndef XIVO_CHOL_BLOCKED
#define XIVO_CHOL_BLOCKED 1
#endif
#if XIVO_CHOL_BLOCKED
#define XIVO_CHOL_DIAG factor_invert_diag_blocked2<9>
#else
#define XIVO_CHOL_DIAG factor_invert_diag
#endif
#include "gate_device.h"

namespace xivo_hip {

namespace {


// One wave64 per filter (no cross-wave barriers; many filters resident per CU so
// the serial 16x16 diagonal factorisations of different filters overlap).
__global__ __launch_bounds__(64, 4) void chol_f64_kernel(CholArgs g) {
  const int filt = blockIdx.x;
  if (filt >= g.batch) return;
  double* S = g.S + (long)filt * g.strideS;
  double* invD = g.invD + (long)filt * g.strideInvD;
  const long ld = g.lds;
  const int nb = g.Mp / 16;
  const int lane = threadIdx.x;
  const int li = lane & 15, lg = lane >> 4;

  __shared__ double sInv[2][256];
  int bad = 0;

  for (int j = 0; j < nb; ++j) {
    // ---- 1. diagonal block update: sum_{k<j} L_jk L_jk^T (two accumulators for ILP)
    // ---- 2. factor + invert the 16x16 diagonal block in the accumulator layout (factor_invert_diag)
    {
      d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
      for (int k = 0; k < j; ++k) {
        const double a0 = S[(16 * j + li) + (long)(16 * k + 0 + lg) * ld];
        const double a1 = S[(16 * j + li) + (long)(16 * k + 4 + lg) * ld];
        const double a2 = S[(16 * j + li) + (long)(16 * k + 8 + lg) * ld];
        const double a3 = S[(16 * j + li) + (long)(16 * k + 12 + lg) * ld];
        acc0 = mfma(a0, a0, acc0);
        acc1 = mfma(a1, a1, acc1);
        acc0 = mfma(a2, a2, acc0);
        acc1 = mfma(a3, a3, acc1);
      }
      d4 x, y;
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = S[(16 * j + li) + (long)(16 * j + lg + 4 * r) * ld] - (acc0[r] + acc1[r]);
      XIVO_CHOL_DIAG(x, y, bad, 16 * j, li, lg);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = lg + 4 * r;
        if (c <= li) {
          S[(16 * j + li) + (long)(16 * j + c) * ld] = x[r];
          S[(16 * j + c) + (long)(16 * j + li) * ld] = x[r];
        }
        sInv[0][c + li * 16] = y[r];   // inv(L)(c, li)
        sInv[1][li + c * 16] = y[r];   // inv(L)^T(li, c)
      }
    }
    __syncthreads();
    for (int e = lane; e < 512; e += 64) invD[(long)j * 512 + e] = (&sInv[0][0])[e];

    // ---- 3. panel: L_ij^T = inv(L_jj) * (S_ij^T - sum_k L_jk L_ik^T), i > j; two rows in flight
    for (int i = j + 1; i < nb; i += 2) {
      const bool two = (i + 1 < nb);
      const int i2 = two ? i + 1 : i;
      // (k-slices 0, 2 and 1, 3 in separate accumulators, summed at the end: the register kernel's order)
      d4 accA = d4{0.0, 0.0, 0.0, 0.0}, accB = d4{0.0, 0.0, 0.0, 0.0};
      d4 accA1 = d4{0.0, 0.0, 0.0, 0.0}, accB1 = d4{0.0, 0.0, 0.0, 0.0};
      for (int k = 0; k < j; ++k) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const long col = (long)(16 * k + 4 * s + lg) * ld;
          const double a = S[(16 * j + li) + col];
          const double b1 = S[(16 * i + li) + col];
          const double b2 = S[(16 * i2 + li) + col];
          if (s & 1) { accA1 = mfma(a, b1, accA1); accB1 = mfma(a, b2, accB1); }
          else { accA = mfma(a, b1, accA); accB = mfma(a, b2, accB); }
        }
      }
      d4 rhsA, rhsB;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        rhsA[r] = S[(16 * i + li) + (long)(16 * j + lg + 4 * r) * ld] - (accA[r] + accA1[r]);
        rhsB[r] = S[(16 * i2 + li) + (long)(16 * j + lg + 4 * r) * ld] - (accB[r] + accB1[r]);
      }
      d4 outA = d4{0.0, 0.0, 0.0, 0.0}, outB = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double iv = sInv[0][li + (4 * s + lg) * 16];
        outA = mfma(iv, rhsA[s], outA);
        outB = mfma(iv, rhsB[s], outB);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        S[(16 * i + li) + (long)(16 * j + lg + 4 * r) * ld] = outA[r];   // L(i-block, j-block)
        S[(16 * j + lg + 4 * r) + (long)(16 * i + li) * ld] = outA[r];   // L^T mirrored to upper
        if (two) {
          S[(16 * i2 + li) + (long)(16 * j + lg + 4 * r) * ld] = outB[r];
          S[(16 * j + lg + 4 * r) + (long)(16 * i2 + li) * ld] = outB[r];
        }
      }
    }
    __syncthreads();
  }
  if (lane == 0) g.status[filt] = bad;
}

// Register-resident variant for factors of at most NB <= 12 block rows (M <= 192): one workgroup of
// four waves per filter, wave w owns the block rows i = w (mod 4) and keeps every L_ik it has produced
// in registers, in the MFMA operand layout (the C layout of L_ik^T, see above, IS that layout). S is
// read from HBM exactly once and L written once; the left-looking kernel above re-reads L_ik from
// global for every later column (3x the bytes of S with thousands of filters in flight: it was
// HBM-bound at 4.8 TB/s) and runs its whole dependency chain in one wave (178 us for one filter).
// Per block column j (owner wave = j mod 4):
//   A  owner : diagonal update sum_k L_jk L_jk^T from its registers; factor + invert the 16x16 block
//              spread over all 64 lanes (column broadcasts by ds_bpermute, no sqrt / divide in the
//              chain) -> LDS / global; publishes the row panel L_jk, k < j, in LDS
//   C  all   : own rows i > j:  L_ij^T = inv(L_jj) (S_ij^T - sum_k L_jk L_ik^T), kept + stored
// with one barrier between A and C and one after C.
// MINB = workgroups per CU the register budget is cut for: 3 (168 VGPRs, a few spills) is faster for thousands of
// factors (0.67 vs 0.75 ms / 4096 at M = 160), 2 (212 VGPRs) for a single one (76 vs 82 us)

// GATE (round 5): Estimator::MHGating (src/update.cpp:60-96) in the prologue of the factorisation - the chi-square distances
// from the 2 x 2 diagonal blocks of S (the compact copy ell<S> leaves), the relaxation loop, the mask; the rows / columns of
// the rejected pairs are then decoupled WHERE S IS LOADED (0, unit diagonal - what gate_ell_kernel writes into S), their
// inn / diagR / compressed values / P H^T columns neutralised from here. One launch and one pass over S fewer per update;
// with three factors in flight per CU the gate's two dependent round trips hide behind the other workgroups. Same values as
// gate_ell_kernel bit for bit (same expressions), so the factor and everything behind it are unchanged.
template <int NB, int MINB, bool PRE, bool UPFRONT, int NW = 4, bool GATE = false>
__global__ __launch_bounds__(64 * NW, MINB) void chol_reg_f64_kernel(CholArgs g, int mirror, CholGateArgs gt) {
  constexpr int RW = (NB + NW - 1) / NW;   // NW waves per factor: 4 (M <= 192), 8 (M <= 320: one workgroup per CU)
  const int filt = blockIdx.x;
  if (filt >= g.batch) return;
  double* S = g.S + (long)filt * g.strideS;
  double* invD = g.invD + (long)filt * g.strideInvD;
  const long ld = g.lds;
  const int nb = g.Mp / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;

  __shared__ double sInv[2][256];
  __shared__ __attribute__((aligned(16))) double sRow[(NB - 1) * 256];   // [k][lane][4]
  __shared__ int sBad;
  if (tid == 0) sBad = 0;
  __shared__ unsigned char sRej[GATE ? 16 * NB : 1];    // 1: the row belongs to a rejected feature
  if constexpr (GATE) {
    static_assert(NW == 4 && !UPFRONT, "the gate rides in the many-factors instantiations");
    double* sdist = sRow;                                // F <= 8 NB distances + the threshold: scratch until the first panel is published
    const int F = gt.F;
    double* inn = gt.inn + (long)filt * gt.strideInn;
    double* dr = gt.diagR + (long)filt * gt.strideR;
    const double* sd = gt.Sdiag + (long)filt * gt.strideSdiag;
    for (int f = tid; f < F; f += 64 * NW) {             // (gate_ell_body, from_S with the compact diagonal blocks)
      const double s00 = sd[4 * f] - dr[2 * f] + gt.R;
      const double s10 = sd[4 * f + 2];
      const double s11 = sd[4 * f + 3] - dr[2 * f + 1] + gt.R;
      sdist[f] = mh_dist_2x2(s00, s10, s11, inn[2 * f], inn[2 * f + 1]);
    }
    for (int m = tid; m < 16 * NB; m += 64 * NW) sRej[m] = 0;
    __syncthreads();
    if (wave == 0) {
      const double th = relax_threshold(sdist, F, gt.thresh, gt.mult, gt.min_inliers, lane);
      if (lane == 0) sdist[F] = th;
    }
    __syncthreads();
    const double th = sdist[F];
    for (int f = tid; f < F; f += 64 * NW) {
      const bool in = sdist[f] < th;
      gt.mask[(long)filt * F + f] = in ? 1 : 0;
      gt.dist[(long)filt * F + f] = sdist[f];
      if (!in) {
        sRej[2 * f] = 1; sRej[2 * f + 1] = 1;
        inn[2 * f] = 0.0; inn[2 * f + 1] = 0.0;
        dr[2 * f] = 1.0; dr[2 * f + 1] = 1.0;
        double* val = gt.ellval + (long)filt * gt.strideVal + (long)f * gt.ell_w * 2;
        for (int t = 0; t < 2 * gt.ell_w; ++t) val[t] = 0.0;
      }
    }
    __syncthreads();
    double* PHT = gt.PHT + (long)filt * gt.stridePHT;    // the solve's right-hand sides: columns of the rejected pairs
    for (int f = 0; f < F; ++f) {
      if (!sRej[2 * f]) continue;
      for (int n = tid; n < gt.Np; n += 64 * NW) { PHT[n + (long)(2 * f) * gt.ldpht] = 0.0; PHT[n + (long)(2 * f + 1) * gt.ldpht] = 0.0; }
    }
    __syncthreads();                                      // sRow is scratch no longer
  }
  // element (row, col) of S as the gate leaves it: rows / columns of rejected pairs decoupled (gate_ell_body's in-place edit)
  auto gated = [&](double v, int row, int col) -> double {
    if constexpr (GATE) return (sRej[row] | sRej[col]) ? (row == col ? 1.0 : 0.0) : v;
    else return v;
  };

  d4 L[RW][NB];   // L[ii][k] = block (i = wave + 4 ii, k); only k < i is ever touched
  // the diagonal blocks this wave will factor, fetched up front (their latency would otherwise sit in
  // the serial chain of every column): x-layout, element r = S[row li][col lg + 4 r] of block (jd, jd)
  d4 sdiag[RW];
#pragma unroll
  for (int ii = 0; ii < RW; ++ii) {
    const int jd = wave + NW * ii;
    if (NW == 4 && jd < nb) {   // (eight waves, up to 19 block rows: no registers to spare - requested per block column below)
#pragma unroll
      for (int r = 0; r < 4; ++r) sdiag[ii][r] = gated(S[(16 * jd + li) + (long)(16 * jd + lg + 4 * r) * ld], 16 * jd + li, 16 * jd + lg + 4 * r);
    }
  }

  // UPFRONT: every block of S this wave will turn into a block of L is requested here, into the register block that will
  // hold the result - one memory latency per factor instead of one per block column (on this ISA loads and stores
  // share vmcnt and retire in order, so a load issued in column j also waits for the stores of column j - 1)
  if (UPFRONT) {
#pragma unroll
    for (int ii = 0; ii < RW; ++ii) {
      const int i = wave + NW * ii;
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        if (k < NW * ii + NW - 1 && k < i && i < nb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) L[ii][k][r] = S[(16 * i + li) + (long)(16 * k + lg + 4 * r) * ld];
        }
      }
    }
  }
  d4 pre0 = d4{0.0, 0.0, 0.0, 0.0}, pre1 = d4{0.0, 0.0, 0.0, 0.0};   // partial diagonal update of the NEXT column's owner
  // compile-time column index: every L[][] subscript below is a constant, so the factor stays in registers
  static_for<NB>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if (j < nb) {
    constexpr int owner = j % NW, jj = j / NW;
    // S_ij for the rows this wave will finish in phase C (in flight across phases A and B)
    d4 sreg[RW];
#pragma unroll
    for (int ii = 0; ii < RW; ++ii) {
      const int i = wave + NW * ii;
      if (!UPFRONT && NW * ii + NW - 1 > j && i > j && i < nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sreg[ii][r] = gated(S[(16 * i + li) + (long)(16 * j + lg + 4 * r) * ld], 16 * i + li, 16 * j + lg + 4 * r);
      }
    }
    // diagonal block in the 64-lane layout x[r] = X[row li][col lg + 4 r] - which is what the MFMA
    // accumulators of the (symmetric) update already are, so no transpose through LDS
    d4 x;
    if (NW != 4 && wave == owner) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sdiag[jj][r] = S[(16 * j + li) + (long)(16 * j + lg + 4 * r) * ld];
    }
    if (wave == owner) {
      // the terms k < j - 1 of the diagonal update were formed while column j - 1 was being factored (below): only the
      // block row produced by column j - 1 itself is still in the serial chain (same operands, same order)
      d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = d4{0.0, 0.0, 0.0, 0.0};
      if (PRE) { acc0 = pre0; acc1 = pre1; }
#pragma unroll
      for (int k = (PRE && j > 0) ? j - 1 : 0; k < j; ++k) {
        const d4 a = L[jj][k];
        acc0 = mfma(a[0], a[0], acc0);
        acc1 = mfma(a[1], a[1], acc1);
        acc0 = mfma(a[2], a[2], acc0);
        acc1 = mfma(a[3], a[3], acc1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = sdiag[jj][r] - (acc0[r] + acc1[r]);
    }
    if constexpr (PRE && j + 1 < NB) if (j + 1 < nb && wave == ((j + 1) % NW)) {   // next column's owner, idle until the barrier: sum_{k<j} L_{j+1,k} L_{j+1,k}^T
      pre0 = d4{0.0, 0.0, 0.0, 0.0}; pre1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < j; ++k) {
        const d4 a = L[(j + 1) / NW][k];
        pre0 = mfma(a[0], a[0], pre0);
        pre1 = mfma(a[1], a[1], pre1);
        pre0 = mfma(a[2], a[2], pre0);
        pre1 = mfma(a[3], a[3], pre1);
      }
    }
    if (wave == owner) {
      int bad = 0;
      d4 y;
      XIVO_CHOL_DIAG(x, y, bad, 16 * j, li, lg);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = lg + 4 * r;
        if (c <= li) {
          S[(16 * j + li) + (long)(16 * j + c) * ld] = x[r];
          S[(16 * j + c) + (long)(16 * j + li) * ld] = x[r];
        }
        sInv[0][c + li * 16] = y[r];   // inv(L)(c, li)
        sInv[1][li + c * 16] = y[r];   // inv(L)^T(li, c)
      }
      if (bad && lane == 0 && sBad == 0) sBad = bad;
      // publish the row panel of block row j
#pragma unroll
      for (int k = 0; k < j; ++k) *reinterpret_cast<d4*>(&sRow[(k * 64 + lane) * 4]) = L[jj][k];
    }
    lds_barrier();
    if (wave == owner)
      for (int e = lane; e < 512; e += 64) invD[(long)j * 512 + e] = (&sInv[0][0])[e];
#pragma unroll
    for (int ii = 0; ii < RW; ++ii) {
      const int i = wave + NW * ii;
      if (NW * ii + NW - 1 > j && i > j && i < nb) {
        d4 accA = d4{0.0, 0.0, 0.0, 0.0}, accB = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < j; ++k) {
          if (k < NW * ii + NW - 1) {
            const d4 a = *reinterpret_cast<const d4*>(&sRow[(k * 64 + lane) * 4]);
            const d4 bb = L[ii][k];
            accA = mfma(a[0], bb[0], accA);
            accB = mfma(a[1], bb[1], accB);
            accA = mfma(a[2], bb[2], accA);
            accB = mfma(a[3], bb[3], accB);
          }
        }
        d4 rhs;
#pragma unroll
        for (int r = 0; r < 4; ++r) rhs[r] = (UPFRONT ? L[ii][j][r] : sreg[ii][r]) - (accA[r] + accB[r]);
        d4 out = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) out = mfma(sInv[0][li + (4 * s4 + lg) * 16], rhs[s4], out);
        L[ii][j] = out;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          S[(16 * i + li) + (long)(16 * j + lg + 4 * r) * ld] = out[r];                 // L(i-block, j-block)
          if (mirror) S[(16 * j + lg + 4 * r) + (long)(16 * i + li) * ld] = out[r];     // L^T for the streamed solve
        }
      }
    }
    lds_barrier();     // sInv / sRow are rewritten by the next owner
    }
  });
  if (tid == 0) g.status[filt] = sBad;
}

}  // namespace

// XIVO_HIP_CHOL_WAVE (A/B, and the bit-identity test of the two factorisation kernels): the one-wave kernel for every size
static bool chol_one_wave_forced() { static const bool on = getenv("XIVO_HIP_CHOL_WAVE") != nullptr; return on; }
// the instantiation of the four-wave register kernel a (size, batch) takes - ONE place decides it, for the launcher, the gate
// and the label alike: mode 0 = many factors (three workgroups per CU at >= nine block rows, blocks of S requested per block
// column, no look-ahead: the only instantiations that carry the gate), mode 2 = few factors (two workgroups per CU, every
// block of S requested up front, look-ahead on the diagonal update)
static int chol_reg_mode(int batch) { return batch >= 512 ? 0 : 2; }

// whether launch_chol_f64 would run an instantiation that carries the gate for this factor size / batch
bool chol_gate_supported(int Mp, int batch) {
  return !chol_one_wave_forced() && Mp / 16 <= 12 && chol_reg_mode(batch) == 0;
}

int launch_chol_f64(const CholArgs& g, hipStream_t stream, const CholGateArgs* gate) {
  if (g.batch <= 0) return 0;
  const CholGateArgs nogate{};
  if (gate && !chol_gate_supported(g.Mp, g.batch)) return (int)hipErrorInvalidValue;
  const int nb = g.Mp / 16;
  // Round 3 (factor_invert_diag on the matrix pipe): the register kernel is ~11 500 instructions at ten block rows (it
  // was ~45 000: 170-200 KB of straight-line code that lost 2.6x on nodes with slow instruction fetch) and is the
  // default for every batch size it holds; the k-slice loop of factor_invert_diag then brought it to 42 KB. Measured at
  // M = 160: one factor 48-52 us (round 2: 80), 16384 factors 1.6-1.7 ms (register kernel, three workgroups per CU, no
  // look-ahead) against 2.4-2.6 ms for the one-wave kernel and 2.3-2.4 ms for either kernel before. Thirteen to nineteen
  // block rows run the same kernel on eight waves (below); the one-wave kernel serves what is left (M > 304).
  // Every instantiation produces the same bits (tests/test_update_gpu.py::test_cholesky_kernels_are_bit_identical).
  if (!chol_one_wave_forced() && nb <= 12) {
    const int mirror = (nb > 10 || g.latency) ? 1 : 0;   // the streamed solve reads the mirrored upper triangle: nb >= 12, and nb = 11 when the
                                                         // whitened outputs leave the kernel (launch_trsm_f64)
    const bool many = g.batch >= 512;
    const int mode = chol_reg_mode(g.batch);
    if (gate && mode != 0) return (int)hipErrorInvalidValue;   // (only the mode-0 instantiations carry the gate)
#define CHOL_REG_LAUNCH(NB_, MINB_)                                                                                                     \
  do {                                                                                                                                  \
    if (mode == 0 && gate) hipLaunchKernelGGL((chol_reg_f64_kernel<NB_, MINB_, false, false, 4, true>), dim3(g.batch), dim3(256), 0, stream, g, mirror, *gate); \
    else if (mode == 0) hipLaunchKernelGGL((chol_reg_f64_kernel<NB_, MINB_, false, false>), dim3(g.batch), dim3(256), 0, stream, g, mirror, nogate);   \
    else hipLaunchKernelGGL((chol_reg_f64_kernel<NB_, MINB_, true, true>), dim3(g.batch), dim3(256), 0, stream, g, mirror, nogate);             \
  } while (0)
    if (nb <= 4) CHOL_REG_LAUNCH(4, 2);
    else if (nb <= 8) CHOL_REG_LAUNCH(8, 2);
    else if (nb <= 10 && many) CHOL_REG_LAUNCH(10, 3);
    else if (nb <= 10) CHOL_REG_LAUNCH(10, 2);
    else if (many) CHOL_REG_LAUNCH(12, 3);
    else CHOL_REG_LAUNCH(12, 2);
#undef CHOL_REG_LAUNCH
    return (int)hipGetLastError();
  }
  if (gate) return (int)hipErrorInvalidValue;
  // 13..19 block rows (M <= 304: BASELINE config 4, the 125-feature build): the register kernel on eight waves, one
  // workgroup per CU (two waves per SIMD at up to 256 VGPRs) - S read once, L written once, where the one-wave kernel
  // re-reads L_ik for every later block column (15.7 GB per 4096 factors at M = 300, HBM-bound)
  if (!chol_one_wave_forced() && nb > 12 && nb <= 19) {
    const int mirror = 1;
    const bool pre = g.batch < 512;
    if (nb <= 16) {
      if (pre) hipLaunchKernelGGL((chol_reg_f64_kernel<16, 1, true, false, 8>), dim3(g.batch), dim3(512), 0, stream, g, mirror, nogate);
      else hipLaunchKernelGGL((chol_reg_f64_kernel<16, 1, false, false, 8>), dim3(g.batch), dim3(512), 0, stream, g, mirror, nogate);
    } else {
      if (pre) hipLaunchKernelGGL((chol_reg_f64_kernel<19, 1, true, false, 8>), dim3(g.batch), dim3(512), 0, stream, g, mirror, nogate);
      else hipLaunchKernelGGL((chol_reg_f64_kernel<19, 1, false, false, 8>), dim3(g.batch), dim3(512), 0, stream, g, mirror, nogate);
    }
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(chol_f64_kernel, dim3(g.batch), dim3(64), 0, stream, g);
  return (int)hipGetLastError();
}

void chol_kernel_label(int Mp, int batch, char* buf, size_t n) {
  const int nb = Mp / 16;
  const bool reg8 = nb > 12 && nb <= 19;
  if (chol_one_wave_forced() || (nb > 12 && !reg8)) snprintf(buf, n, "chol_f64_kernel");
  else if (reg8) snprintf(buf, n, "chol_reg_f64_kernel<%d,1,8 waves>", nb <= 16 ? 16 : 19);
  else snprintf(buf, n, "chol_reg_f64_kernel<%d,%d>", nb <= 4 ? 4 : (nb <= 8 ? 8 : (nb <= 10 ? 10 : 12)), (nb > 8 && batch >= 512) ? 3 : 2);
}

}  // namespace xivo_hip
