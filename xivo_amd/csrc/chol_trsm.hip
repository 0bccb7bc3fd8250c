// Batched gain solve (and, inside it, the covariance update) for the EKF measurement update.
//
// Replaces `K_.transpose() = S_.ldlt().solve(H_ * P_)` and `err_ = K_ * inn_`
// (/root/reference/src/estimator.cpp:1265-1267) once S = L L^T is factored (chol_f64.hip: the reference uses Eigen's
// pivoted LDL^T; S = HPH^T + R is SPD, so an un-pivoted blocked Cholesky is the same linear map up to rounding - tests
// assert 1e-8 on dx, 1e-6 on P).
//
// Every kernel here has a wave64 own 16 right-hand-side columns and keep the whole Mp x 16 solution in accumulator registers:
// the f64 MFMA C/D layout (row = (lane>>4) + 4*reg, col = lane&15) is exactly the B-operand layout of the four k-slices of the
// next MFMA, so forward and backward substitution chain with no data movement at all; dx = K*inn falls out as a lane reduce.
// trsm_lds_f64_kernel<NBM, TF> : one workgroup of 16 waves per filter, the whole factor in LDS; beyond the solve it
//   carries, on the gain still in its registers (which is also the A-operand layout),
//     TF = 1  T = K (HP) - P                                   (estimator.cpp:1276-1280, re-associated pipeline)
//     TF = 2  P+ = P - W^T W after the forward substitution    (symmetric form)
//     TF = 4  the whole Joseph update in its whitened form     P+ = P - (W - D)^T (W + D)   (round 3: the default up to M = 176)
//     TF = 5  the whitened outputs V^T, Y^T for a tiled product outside the kernel (N > 256)
//   through sym_tiles_from_regs: symmetric N x N x M products with one operand in registers and the other arriving
//   in LDS by global_load_lds (or from the owner waves' registers), every unordered pair of 16-row blocks once.
// pnew_reg_f64_kernel : P+ = G K^T - T by the same walk with the rows of G loaded into the registers.
// trsm_stream_f64_kernel<NB, WH> : factors beyond the LDS (M > 176), panel by panel (DMA panels); WH = 1: whitened outputs.
#include <stdlib.h>

#include "mfma_util.h"
#include <stdio.h>

#include "trsm_lds_kernel.h"

namespace xivo_hip {

namespace {

// y = L^-1 inn, one wave per filter: block row k after block row k - 1; the lanes split the dot products of the 16
// rows over the columns already solved (4 lanes per row, strided), then inv(L_kk) (16 x 16, kept by the factorisation)
// finishes the block. 160^2 / 2 multiply-adds per filter - negligible next to the matrix solve.
__global__ __launch_bounds__(256) void fwd_vec_kernel(const double* __restrict__ LUall, long strideLU, int ldlu,
                                                     const double* __restrict__ invDall, long strideInvD,
                                                     const double* __restrict__ innall, long strideInn, double* __restrict__ yall,
                                                     long strideY, int Mp, int batch) {
  __shared__ double sy[4][384];                          // y of the wave's filter (LDS: same-wave write -> read is ordered)
  const int wv = threadIdx.x >> 6;
  const int filt = blockIdx.x * (blockDim.x >> 6) + wv;
  if (filt >= batch) return;
  const int lane = threadIdx.x & 63, r = lane & 15, part = lane >> 4;
  const double* L = LUall + (long)filt * strideLU;
  const double* invD = invDall + (long)filt * strideInvD;
  const double* inn = innall + (long)filt * strideInn;
  double* y = yall + (long)filt * strideY;
  double* ys = sy[wv];
  const int nb = Mp / 16;
  for (int k = 0; k < nb; ++k) {
    double acc = 0.0;
    for (int c = part; c < 16 * k; c += 4) acc = fma(L[(16 * k + r) + (long)c * ldlu], ys[c], acc);
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    const double rhs = inn[16 * k + r] - acc;             // valid in every lane (all four parts hold the sum)
    // y_k = inv(L_kk) rhs: row r of the inverse (column-major 16 x 16) against the 16 right-hand sides
    double out = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) out = fma(invD[(long)k * 512 + r + 16 * c], __shfl(rhs, c), out);
    if (part == 0) { ys[16 * k + r] = out; y[16 * k + r] = out; }
  }
}


// LDS-resident variant: the workgroup (16 waves = 256 right-hand-side columns) first
// copies the whole factor into LDS - the 45..55 lower 16x16 blocks of L, with
// inv(L_kk) sitting in the diagonal slots - and then every A operand of the
// forward AND the backward substitution is a conflict-free ds_read_b64 (blocks are
// stored column-major with a leading dimension of 17 so that both L_ik and its
// transpose read without bank conflicts). No barrier after the initial copy.
// Symmetric [16 nwl x 16 nwl] result  Out = A Src^T - Minit  formed tile by tile from register-resident rows:
// wave w holds rows 16w..16w+15 of A as X[mb][r] = A[16w + li, 16 mb + lg + 4 r] - the MFMA operand layout - and the
// rows of Src [16 nwl x 16 nb, leading dimension ldsrc over the row index] arrive in LDS, jbp row blocks per phase,
// double-buffered. Wave w forms the tiles (w, j) for the blocks j cyclically below it ((w - j) mod nwl in 0..nwl/2,
// the antipodal pair going to the higher wave: every unordered pair of blocks once, 8 or 9 tiles per wave at nwl = 16),
// accumulators starting at -Minit, and writes each tile to its lower-triangle position and the mirror image.
// All 16 waves of the workgroup must call it (barriers); `sL` = the whole 160 KB of LDS.
// SRC_REGS: Src = A itself (a symmetric rank-k product, P - W^T W): the wave that owns a block copies its registers
// into the LDS buffer - nothing is read back from memory. NEG_OUT: Out = Minit - A Src^T.
// FIXUP: Src arrives by DMA as usual and the wave that owns a block then replaces it in LDS by 2 Src - A (its registers):
// the whitened Joseph form needs (W - D)^T (W + D) with only W in memory (see TF == 4 below).
// YREGS (with SRC_REGS): the LDS operand is 2 Wr - A from the owner waves' registers (short factors keep W in registers:
// no stash, no DMA, no fix-up pass).
// P+ = G K^T - T (estimator.cpp:1280-1287 re-associated) with the rows of G in registers: the same tile walk as the T
// phase above, one workgroup per filter, wave w holding rows 16w..16w+15 of G and the blocks of K arriving in LDS. For a
// pair of blocks (a, b) the wave that owns it forms EITHER entry (a, b) = G_a K_b^T OR (b, a) = G_b K_a^T of the as-coded
// product - they differ by the rounding of the solve - and writes it to both positions. Replaces the tiled GEMM
// (HBM-bound: every panel read 1.5 times) for the all-fp64 correction product.
template <int NBM>
__global__ __launch_bounds__(1024) void pnew_reg_f64_kernel(PnewRegArgs g) {
  extern __shared__ __attribute__((aligned(16))) double sL[];
  const int bidx = blockIdx.x;
  const int filt = (bidx >> 3) * 8 + (bidx & 7);
  if (filt >= g.batch) return;
  if (g.skip_status && g.skip_status[filt] != 0) return;   // S was not positive definite: P stays the prior
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int nb = g.Mp / 16;
  const int c0 = wave * 16;
  const bool live = c0 < g.Np;
  const double* __restrict__ G = g.G + (long)filt * g.strideG;
  d4 X[NBM];
#pragma unroll
  for (int i = 0; i < NBM; ++i) {
    X[i] = d4{0.0, 0.0, 0.0, 0.0};
    if (live && i < nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) X[i][r] = G[(c0 + li) + (long)(16 * i + lg + 4 * r) * g.ldg];
    }
  }
  sym_tiles_from_regs<NBM>(X, X, sL, g.K + (long)filt * g.strideK, g.ldk, g.T + (long)filt * g.strideT, g.ldt,
                           g.P + (long)filt * g.strideP, g.ldp, nb, g.Np / 16, g.jbp, live, wave, wave, lane);
}

// Streaming variant for factors that do not fit LDS (M > 176) and for states wider than one workgroup (N > 256): the
// workgroup (8 waves = 128 right-hand-side columns, X in registers as above) walks the block columns of L. The column
// panel of step k+1 - inv(L_kk) plus the blocks L_ik, i > k (forward) or L_ki^T, i < k (backward, read from the mirrored
// upper triangle) - travels global -> LDS by DMA (global_load_lds: no registers; round 2 staged it through 20 VGPRs per
// thread and the <19> instantiation spilled 1227 of them) while step k's MFMAs run, into the other half of a
// double-buffered panel; one barrier per step. Blocks sit in LDS as the DMA delivers them - column-major 16 x 16,
// unpadded - which is conflict-free for every read here (the backward pass reads the mirrored copy, never a transpose).
// WH = 1: whitened outputs for the covariance update outside the kernel (tiled product P - V^T Y):
//   after the forward pass W^T goes to the stash (g.K), the backward pass reads W_k back one step before it needs it,
//   forms D_k = B_k - L_kk^T K^T_k in place and leaves Y_k = W_k + D_k in g.Yout, V_k = W_k - D_k in g.K; dx as usual.
// NWS (round 5): waves per workgroup = 16 NWS right-hand-side columns. 8 is the kernel of rounds 2-4; 4 serves the latency route
// (<= 64 filters): one wave per SIMD, so that a wave's chain of MFMAs has the matrix pipe to itself (two waves per SIMD take
// turns on it: the solve of ONE filter at (250, 160) issues 480 MFMAs per wave = 14 us alone, 28 us shared) and twice as
// many CUs work on the few filters there are.
template <int NBM, int WH, int NWS = 8>
// (second launch bound = waves per SIMD the register budget is cut for: 8-wave workgroups -> 3 / 2 / 1 workgroups per CU)
__global__ __launch_bounds__(64 * NWS, NWS == 8 ? (NBM <= 4 ? 6 : (NBM <= 8 ? 4 : 2)) : 2) void trsm_stream_f64_kernel(TrsmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double sL[];   // [2][NBM + 1][256]
  constexpr int PSZ = (NBM + 1) * 256;
  constexpr int WCOLS = 16 * NWS;
  const int chunks = (g.Np + WCOLS - 1) / WCOLS;
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;
  const int filt = (slot / chunks) * 8 + xcd;
  const int chunk = slot % chunks;
  if (filt >= g.batch) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int nb = g.Mp / 16;
  const double* __restrict__ LU = g.LU + (long)filt * g.strideLU;
  const double* __restrict__ invD = g.invD + (long)filt * g.strideInvD;
  const long ld = g.ldlu;
  const int c0 = chunk * WCOLS + wave * 16;
  const bool live = c0 < g.Np;
  const __amdgpu_buffer_rsrc_t rPHT = buf_rsrc(g.PHT + (long)filt * g.stridePHT), rK = buf_rsrc(g.K + (long)filt * g.strideK),
                               rInn = buf_rsrc(g.fwd_only ? g.y + (long)filt * g.strideY : g.inn + (long)filt * g.strideInn),
                               rY = buf_rsrc(WH ? g.Yout + (long)filt * g.strideY2 : g.K);
  const unsigned vPHT = (unsigned)((live ? c0 + li : 0) + lg * g.ldpht) * 8u;
  const unsigned vK = (unsigned)((live ? c0 + li : 0) + lg * g.ldk) * 8u;
  const unsigned vY = WH ? (unsigned)((live ? c0 + li : 0) + lg * g.ldy2) * 8u : 0u;

  d4 X[NBM];
#pragma unroll
  for (int i = 0; i < NBM; ++i) {
    X[i] = d4{0.0, 0.0, 0.0, 0.0};
    if (live && i < nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) X[i][r] = buf_ld(rPHT, vPHT, (unsigned)((16 * i + 4 * r) * g.ldpht) * 8u);
    }
  }

  // panel of step k: slot j holds block (j, k) of the stored factor (lower triangle: L_jk, j > k; mirrored upper triangle:
  // L_kj^T, j < k); slot k holds inv(L_kk) (forward) or its transpose (backward); backward + WH: slot nb holds the diagonal
  // block L_kk itself (stored symmetric: L below, L^T above - masked on read)
  auto issue_panel = [&](int k, bool fwd, double* buf) {
    const int j0 = fwd ? k : 0, nj = (fwd ? nb - k : k + 1) + ((WH && !fwd) ? 1 : 0);
    for (int q = wave; q < nj * 2; q += NWS) {
      int j = j0 + (q >> 1);
      const int h = q & 1;
      const bool extra = WH && !fwd && (q >> 1) == nj - 1;
      const double* src;
      if (extra) { j = nb; src = LU + (16 * k + 2 * (lane & 7)) + (long)(16 * k + 8 * h + (lane >> 3)) * ld; }
      else if (j == k) src = invD + (long)k * 512 + (fwd ? 0 : 256) + h * 128 + 2 * lane;
      else src = LU + (16 * j + 2 * (lane & 7)) + (long)(16 * k + 8 * h + (lane >> 3)) * ld;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + j * 256 + h * 128), 16, 0, 0);
    }
  };
  double* buf0 = sL;
  double* buf1 = sL + PSZ;

  // ---- forward: L Y = HP
  issue_panel(0, true, buf0);
#pragma unroll
  for (int k = 0; k < NBM; ++k) {
    if (k < nb) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                              // panel k landed for everybody; the other buffer is free again
      const double* cur = (k & 1) ? buf1 : buf0;
      if (k + 1 < nb) issue_panel(k + 1, true, (k & 1) ? buf0 : buf1);
      if (live) {
        d4 t = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) t = mfma(cur[k * 256 + li + 16 * (4 * s + lg)], X[k][s], t);
        X[k] = t;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int i = k + 1; i < NBM; ++i) {
            if (i < nb) X[i] = mfma(-cur[i * 256 + li + 16 * (4 * s + lg)], t[s], X[i]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double part = 0.0;
  if (WH && live) {     // W^T to the stash: the backward pass destroys it and the covariance update needs it again
#pragma unroll
    for (int i = 0; i < NBM; ++i) {
      if (i < nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) buf_st(X[i][r], rK, vK, (unsigned)((16 * i + 4 * r) * g.ldk) * 8u);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- backward: L^T K^T = Y  (panels of column k of the upper triangle = rows of L^T)
  __syncthreads();
  if (!g.fwd_only) issue_panel(nb - 1, false, buf0);
  int par = 0;
#pragma unroll
  for (int k = NBM - 1; k >= 0; --k) {
    if (k < nb && !g.fwd_only) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const double* cur = par ? buf1 : buf0;
      if (k > 0) issue_panel(k - 1, false, par ? buf0 : buf1);
      par ^= 1;
      if (live) {
        d4 wk = d4{0.0, 0.0, 0.0, 0.0};
        if (WH) {
          if (k == nb - 1) wk = X[k];
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) wk[r] = buf_ld(rK, vK, (unsigned)((16 * k + 4 * r) * g.ldk) * 8u);
          }
        }
        d4 t = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) t = mfma(cur[k * 256 + li + 16 * (4 * s + lg)], X[k][s], t);
        if (WH) {
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {          // D_k = B_k - L_kk^T K^T_k; (L_kk)^T element (li, kk) = L_kk(kk, li), zero for kk < li
            const int kk = 4 * s2 + lg;
            const double a = cur[nb * 256 + li + 16 * kk];   // the diagonal block is stored symmetric: (li, kk) = (kk, li)
            X[k] = mfma(kk >= li ? -a : 0.0, t[s2], X[k]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) part = fma(t[r], buf_ld(rInn, (unsigned)lg * 8u, (unsigned)(16 * k + 4 * r) * 8u), part);
        } else {
          X[k] = t;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int i = 0; i < k; ++i) X[i] = mfma(-cur[i * 256 + li + 16 * (4 * s + lg)], t[s], X[i]);
        }
        if (WH) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {                                                                              // Y_k = W_k + D_k
            if (g.out_f32) buf_st_f32((float)(wk[r] + X[k][r]), rY, vY >> 1, (unsigned)((16 * k + 4 * r) * g.ldy2) * 4u);
            else buf_st(wk[r] + X[k][r], rY, vY, (unsigned)((16 * k + 4 * r) * g.ldy2) * 8u);
          }
          X[k] = wk - X[k];                                                                                          // V_k = W_k - D_k
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // (keeps the scheduler from hoisting every step's stash / inn loads to the top)
    }
  }
  if (!live) return;
#pragma unroll
  for (int i = 0; i < NBM; ++i) {
    if (i < nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (float outputs: V^T goes behind Y^T in the Yout buffer - the K buffer still holds the fp64 stash other column
        //  chunks of this filter may be reading back)
        if (WH && g.out_f32) buf_st_f32((float)X[i][r], rY, vY >> 1, (unsigned)(g.Np * g.Mp + (16 * i + 4 * r) * g.ldy2) * 4u);
        else buf_st(X[i][r], rK, vK, (unsigned)((16 * i + 4 * r) * g.ldk) * 8u);
        if (!WH) part = fma(X[i][r], buf_ld(rInn, (unsigned)lg * 8u, (unsigned)(16 * i + 4 * r) * 8u), part);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  part += __shfl_xor(part, 16);
  part += __shfl_xor(part, 32);
  if (lg == 0) g.err[(long)filt * g.strideErr + c0 + li] = part;
}

template <int NBM, int WH, int NWS = 8>
int launch_trsm_stream_t(const TrsmArgs& g, hipStream_t stream) {
  const int chunks = (g.Np + 16 * NWS - 1) / (16 * NWS);
  const int grid = ((g.batch + 7) / 8) * 8 * chunks;
  const size_t lds = (size_t)2 * (NBM + 1) * 256 * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_stream_f64_kernel<NBM, WH, NWS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((trsm_stream_f64_kernel<NBM, WH, NWS>), dim3(grid), dim3(64 * NWS), lds, stream, g);
  return (int)hipGetLastError();
}
// Block-row capacities the streamed kernel is instantiated for. Register allocation of the fully unrolled substitution
// is erratic from one capacity to the next (spilled VGPRs, hipcc 7.2: plain 14: 0, 16: 0, 18: 4801, 19: 6353, 20: 5019,
// 22: 10, 24: 26; whitened 14: 0, 16: 35, 19: 12, 20: 47, 22: 74, 24: 806 - scripts/resource_usage.sh), so each variant
// uses the capacities that compile clean.
static int stream_capacity(int nb, bool whitened) {
  if (whitened && nb <= 8) return nb <= 4 ? 4 : (nb <= 6 ? 6 : 8);   // small factors (latency route, eight block rows on a wide state)
  if (nb <= 14) return 14;
  if (whitened) return nb <= 19 ? 19 : (nb <= 22 ? 22 : 24);
  return nb <= 16 ? 16 : (nb <= 22 ? 22 : 24);
}

template <int NBM, int TF>
int launch_trsm_lds_tf(const TrsmArgs& g_in, hipStream_t stream) {
  TrsmArgs g = g_in;
  const int nb = g.Mp / 16;
  const int chunks = (g.Np + 255) / 256;
  const int grid = ((g.batch + 7) / 8) * 8 * chunks;
  size_t lds = (size_t)nb * (nb + 1) / 2 * 16 * 17 * sizeof(double);
  if (TF) {   // the T phase re-uses all of the LDS for its B operands: nb * 2 KB per column block
    lds = 160 * 1024;
    g.t_jbp = (int)(lds / 2 / ((size_t)nb * 4 * 64 * sizeof(double)));   // two buffers
    if (g.t_jbp > 16) g.t_jbp = 16;
    if (XIVO_ABL == 10) g.t_jbp = 2;
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_lds_f64_kernel<NBM, TF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((trsm_lds_f64_kernel<NBM, TF>), dim3(grid), dim3(1024), lds, stream, g);
  return (int)hipGetLastError();
}
template <int NBM>
int launch_trsm_lds_t(const TrsmArgs& g, hipStream_t stream) {
  if (g.Yout && !g.fwd_only) return launch_trsm_lds_tf<NBM, 5>(g, stream);   // whitened outputs, any number of column chunks
  {
    if (g.T && trsm_forms_T(g.Mp, g.Np))
      return g.fwd_only ? launch_trsm_lds_tf<NBM, 2>(g, stream)
                        : (g.joseph ? launch_trsm_lds_tf<NBM, 4>(g, stream) : launch_trsm_lds_tf<NBM, 1>(g, stream));
  }
  return launch_trsm_lds_tf<NBM, 0>(g, stream);
}

}  // namespace

bool pnew_reg_supported(int Mp, int Np) {
  return Mp / 16 <= 10 && Np <= 256 && Np % 16 == 0;
}

template <int NBM>
static int launch_pnew_reg_t(const PnewRegArgs& g_in, hipStream_t stream) {
  PnewRegArgs g = g_in;
  const int nb = g.Mp / 16;
  const size_t lds = 160 * 1024;
  g.jbp = (int)(lds / 2 / ((size_t)nb * 4 * 64 * sizeof(double)));
  if (g.jbp > 16) g.jbp = 16;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pnew_reg_f64_kernel<NBM>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((pnew_reg_f64_kernel<NBM>), dim3(((g.batch + 7) / 8) * 8), dim3(1024), lds, stream, g);
  return (int)hipGetLastError();
}

int launch_pnew_reg_f64(const PnewRegArgs& g, hipStream_t stream) {
  if (g.batch <= 0) return 0;
  const int nb = g.Mp / 16;
  if (nb <= 6) return launch_pnew_reg_t<6>(g, stream);
  if (nb <= 10) return launch_pnew_reg_t<10>(g, stream);
  return (int)hipErrorInvalidValue;
}

void pnew_reg_kernel_label(int Mp, char* buf, size_t n) { snprintf(buf, n, "pnew_reg_f64_kernel<%d>", Mp / 16 <= 6 ? 6 : 10); }

// Few filters (fewer one-per-filter workgroups than a quarter of the CUs): the solve spreads over 128-column workgroups of
// the streamed kernel and the covariance product over the tiles of the stand-alone GEMM, instead of one CU per filter doing
// all 13 000 MFMAs of an update by itself (87 us at the issue rate for N = 250, M = 160). The caller (capi.hip) combines this
// with the pipeline's own conditions and XIVO_HIP_FLAG_THROUGHPUT_ROUTE.
bool trsm_latency_route(int Mp, int batch) { return Mp / 16 <= 14 && batch <= 64; }

bool trsm_forms_T(int Mp, int Np) {
  return Mp / 16 <= 11 && Np <= 256 && Np % 16 == 0;   // the factor fits the LDS and one workgroup covers every column
}

int launch_trsm_f64(const TrsmArgs& g, hipStream_t stream) {
  if (g.batch <= 0) return 0;
  const int nb = g.Mp / 16;
  if (g.latency && g.Yout && !g.fwd_only && nb <= 14) {      // few filters: four-wave workgroups of 64 columns (one wave per SIMD)
    if (nb <= 4) return launch_trsm_stream_t<4, 1, 4>(g, stream);
    if (nb <= 6) return launch_trsm_stream_t<6, 1, 4>(g, stream);
    if (nb <= 8) return launch_trsm_stream_t<8, 1, 4>(g, stream);
    return launch_trsm_stream_t<14, 1, 4>(g, stream);
  }
  if (g.stream8 && g.Yout && !g.fwd_only && nb <= 8) {
    if (nb <= 4) return launch_trsm_stream_t<4, 1>(g, stream);
    if (nb <= 6) return launch_trsm_stream_t<6, 1>(g, stream);
    return launch_trsm_stream_t<8, 1>(g, stream);
  }
  if (g.out_f32 && g.Yout && !g.fwd_only && nb <= 14) return launch_trsm_stream_t<14, 1>(g, stream);   // (only the streamed kernel writes float outputs)
  // whole factor in LDS (nb(nb+1)/2 blocks of 16x17 doubles) when it fits 160 KiB
  // (a four-block-row instantiation of the whitened form for the TUM-VI build's 30 features spills 1232 VGPRs - hipcc 7.2;
  //  six block rows serve M <= 96)
  if (nb <= 6) return launch_trsm_lds_t<6>(g, stream);
  if (nb <= 10) return launch_trsm_lds_t<10>(g, stream);
  // (eleven block rows with the whitened outputs leaving the kernel: that instantiation spills 700 VGPRs - streamed instead)
  if (nb <= 11 && !(g.Yout && !g.fwd_only)) return launch_trsm_lds_t<11>(g, stream);
  // larger factors: stream the factor through a double-buffered LDS panel
  const bool wh = g.Yout && !g.fwd_only;
  if (nb <= 24) {
    switch (stream_capacity(nb, wh) * 2 + (wh ? 1 : 0)) {
      case 14 * 2: return launch_trsm_stream_t<14, 0>(g, stream);
      case 14 * 2 + 1: return launch_trsm_stream_t<14, 1>(g, stream);
      case 16 * 2: return launch_trsm_stream_t<16, 0>(g, stream);
      case 19 * 2 + 1: return launch_trsm_stream_t<19, 1>(g, stream);
      case 22 * 2: return launch_trsm_stream_t<22, 0>(g, stream);
      case 22 * 2 + 1: return launch_trsm_stream_t<22, 1>(g, stream);
      case 24 * 2: return launch_trsm_stream_t<24, 0>(g, stream);
      default: return launch_trsm_stream_t<24, 1>(g, stream);
    }
  }
  return (int)hipErrorInvalidValue;
}

int launch_fwd_vec(const double* LU, long strideLU, int ldlu, const double* invD, long strideInvD, const double* inn, long strideInn,
                   double* y, long strideY, int Mp, int batch, hipStream_t stream) {
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(fwd_vec_kernel, dim3((batch + 3) / 4), dim3(256), 0, stream, LU, strideLU, ldlu, invD, strideInvD, inn, strideInn,
                     y, strideY, Mp, batch);
  return (int)hipGetLastError();
}

void trsm_kernel_label(int Mp, char* buf, size_t n, int forms_T, bool latency, bool stream8) {
  const int nb = Mp / 16;
  if (latency && forms_T == 5 && nb <= 14) snprintf(buf, n, "trsm_stream_f64_kernel<%d,1,4>", stream_capacity(nb, true));
  else if (stream8 && forms_T >= 4 && nb <= 8) snprintf(buf, n, "trsm_stream_f64_kernel<%d,1>", stream_capacity(nb, true));
  else if (nb <= 11) snprintf(buf, n, "trsm_lds_f64_kernel<%d,%d>", nb <= 6 ? 6 : (nb <= 10 ? 10 : 11), forms_T);
  else snprintf(buf, n, "trsm_stream_f64_kernel<%d,%d>", stream_capacity(nb, forms_T >= 4), forms_T >= 4 ? 1 : 0);
}

}  // namespace xivo_hip
