// Batched fp64 MFMA GEMM for the EKF measurement update (gfx950 / MI355X).
//
// One kernel serves every dense contraction of Estimator::UpdateJosephForm
// (/root/reference/src/estimator.cpp:1257-1288) and the dense covariance
// propagation Phi P Phi^T + Q:
//     HP   = H * P   (and its transpose PH^T)   (estimator.cpp:1259 / :1266)
//     S    = HP * H^T + diag(R)                 (estimator.cpp:1259-1263)
//     A    = K * H - I                          (estimator.cpp:1276-1279)
//     T    = A * P = K * (HP) - P               (estimator.cpp:1280, first product)
//     P+   = T * A^T + K diag(R) K^T            (estimator.cpp:1280-1287, fused)
// All of them are written as C = A_op * B_op^T with the output index contiguous
// in both operands ("NT", column-major) - P is symmetric so H*P reads P by rows,
// and H^T is kept next to H - so there is exactly ONE LDS layout and no
// transposing stage.
//
// Tiling (CDNA4): 256 threads = 4 wave64; each wave owns WM x WN tiles of
// v_mfma_f64_16x16x4_f64 (4 accumulator VGPR pairs each). A/B k-panels (BK = 16)
// are staged global -> registers -> LDS (the register leg is issued before the
// MFMA block of the previous panel so HBM/L2 latency hides under it), LDS rows
// padded by 16 doubles so the two k-rows a ds_read_b64 lane group touches fall
// in different bank halves. The accumulator is computed TRANSPOSED (mfma(b, a))
// so that each store instruction writes four 128-byte runs of the column-major
// output; mirrored / transposed outputs go through a per-wave LDS transpose for
// the same reason.
//
// Symmetric outputs (S, P+) use 128x128 tiles, skip tiles above the diagonal,
// and on DIAGONAL tiles switch the wave -> block mapping to "strips": wave w owns
// block rows {w, 7-w} of the 8x8 block grid, i.e. exactly 9 of the 36
// lower-triangle blocks each, so the triangle costs 9/16 of a full tile instead
// of the 16/16 its busiest wave would otherwise pay.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace xivo_hip {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// Compute-type traits: fp64 (default, the reference's arithmetic) or fp32 MFMA with fp32 accumulation
// (XIVO_HIP_FLAG_FP32_WHITENED: BASELINE.json config 4, the whitened correction product only). Operands stay fp64 in HBM
// and are rounded to fp32 when they are written to LDS; results are widened on store.
template <typename CT> struct Cx;
// Optional (-DXIVO_MFMA44=1, off by default): fp64 products on v_mfma_f64_4x4x4_4b_f64. In a pure issue-rate probe
// (scripts/mfma_probe.hip) it sustains 72 TFLOP/s on the MI355X against 49 for v_mfma_f64_16x16x4_f64, but in this
// kernel it measured SLOWER (dense (KH-I)P 2.96 vs 2.20 ms per 4096 filters, T 6.93 vs 6.64 ms per 16384): 2.5x the
// LDS fragment reads per flop with their latency exposed at 2 waves per SIMD, and 185 instead of 168 VGPRs for the
// 128x64 tile (2 instead of 3 workgroups per CU). Kept for a future software-pipelined version. One 16x16x4 step becomes four 4x4x4
// instructions, m = 0..3: lane group blk (= (lane & 15) >> 2) multiplies the B column block blk (the operand register
// the 16x16x4 form uses, unchanged) with the A row block (blk + m) & 3 - the A fragment read from LDS rotated by 4 m
// lanes within its 16-lane row. Operand layout (measured, scripts/mfma44_layout.hip): A / B lane = 16 k + 4 blk + i|j,
// D lane = 16 i + 4 blk + j. Accumulator component m of lane (lg, li) then holds
//   C[I0 + 4 (((li >> 2) + m) & 3) + (li & 3)][J0 + 4 (li >> 2) + lg];
// tiles are converted from / to the 16x16x4 accumulator layout through a per-wave LDS pad around the main loop, so
// prologue and epilogue are shared. -DXIVO_MFMA44=0 builds the 16x16x4 form (A/B).
#ifndef XIVO_MFMA44
#define XIVO_MFMA44 0
#endif
#ifndef XIVO_NT_STORE
#define XIVO_NT_STORE 0     // A/B: non-temporal stores of the output tile
#endif
template <> struct Cx<double> {
  typedef d4 acc_t; typedef d2 pair_t;
  static __device__ __forceinline__ acc_t zero() { return d4{0.0, 0.0, 0.0, 0.0}; }
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ double mfma44(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
  // C/D row of accumulator register r for lane group lg (cdna_hip_programming.md section 3: f64 differs)
  static __device__ __forceinline__ int crow(int lg, int r) { return lg + 4 * r; }
  static __device__ __forceinline__ pair_t cvt(d2 v) { return v; }
};
template <> struct Cx<float> {
  typedef f4 acc_t; typedef f2 pair_t;
  static __device__ __forceinline__ float mfma44(float, float, float c) { return c; }   // never used: fp32 keeps 16x16x4
  static __device__ __forceinline__ acc_t zero() { return f4{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int crow(int lg, int r) { return 4 * lg + r; }
  static __device__ __forceinline__ pair_t cvt(d2 v) { return pair_t{(float)v[0], (float)v[1]}; }
};

// wave tiles whose 4x4x4 form fits the 256-VGPR budget without spilling (measured with -Rpass-analysis): up to 16
// accumulator slots, except the 3x5 / 5x3 shapes
constexpr bool mf44_tile(int wm, int wn) { return XIVO_MFMA44 && wm * wn <= 16 && wm * wn != 15; }

// FAST: the tile is interior (every 16x16 block in range and wanted), so slot activity is a
// compile-time property - for strip tiles together with the compile-time wave index WAVE - and
// the MFMA loop is straight-line code. Otherwise each slot is guarded by a wave-uniform bit
// (hipcc turns those guards into a chain of scalar branches: ~2 taken branches per MFMA,
// which is what made half-empty diagonal tiles SLOWER than full ones).
template <typename CT, int WM, int WN, int BK, bool STRIP, bool FAST, int WAVE>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int filt, const int m0, const int n0,
                                          double* smem_raw) {
  typedef typename Cx<CT>::acc_t acc_t;
  typedef typename Cx<CT>::pair_t pair_t;
  CT* smem = reinterpret_cast<CT*>(smem_raw);
  static_assert(!STRIP || (WM == 4 && WN == 4), "strip mapping is defined for 128x128 tiles");
  constexpr int BM = 32 * WM, BN = 32 * WN;
  constexpr int LDAS = BM + 16, LDBS = BN + 16;
  constexpr int NS = WM * WN;                 // accumulator slots per wave
  constexpr int NA = STRIP ? 2 : WM;          // A fragments per k-slice
  constexpr int NB = STRIP ? 2 * WN : WN;     // B fragments per k-slice
  CT* As = smem;
  CT* Bs = smem + BK * LDAS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // readfirstlane: tells the compiler the wave index (and everything derived from it: the slot
  // activity mask, strip rows) is wave-uniform -> SGPRs and scalar branches instead of exec masking
  const int wave = (STRIP && FAST) ? WAVE : __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int wr = wave & 1, wc = wave >> 1;

  // block-row / block-col (units of 16 within the tile) of A fragment a / B fragment b
  auto arow = [&](int a) -> int { return STRIP ? (a == 0 ? wave : 7 - wave) : wr * WM + a; };
  auto bcol = [&](int b) -> int { return STRIP ? b : wc * WN + b; };
  // slot q -> (a, b)
  auto slot_a = [&](int q) -> int { return STRIP ? q / NB : q / WN; };
  auto slot_b = [&](int q) -> int { return STRIP ? q % NB : q % WN; };

  // wave-uniform activity mask per slot
  unsigned active = 0;
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const int I0 = m0 + 16 * arow(slot_a(q)), J0 = n0 + 16 * bcol(slot_b(q));
    bool on = (I0 < g.Mp) && (J0 < g.Np);
    if (g.lower_only && J0 > I0) on = false;   // block strictly above the diagonal
    if (on) active |= 1u << q;
  }
  active = __builtin_amdgcn_readfirstlane(active);
  // compile-time activity of slot q on the FAST path
  auto fast_on = [&](int q) -> bool { return !STRIP || slot_b(q) <= arow(slot_a(q)); };
  auto is_on = [&](int q) -> bool { return FAST ? fast_on(q) : (active & (1u << q)) != 0; };

  // Accumulators start at 0, or at -/+Msub for C = acc -/+ Msub (T = K(HP) - P): the
  // loads are issued here, ahead of the first k-panel, instead of serialising
  // behind the stores of the epilogue.
  // fp32 products keep Msub in fp64: it is added to the widened accumulators in the epilogue
  // (loaded after the main loop, when the staging registers are dead) instead of being rounded
  // to fp32 here.
#ifndef XIVO_LATE_MSUB_F64
#define XIVO_LATE_MSUB_F64 0   // A/B: fp64 products also fetch Msub through the epilogue ring instead of initialising the accumulators
#endif
  constexpr bool LATE_MSUB = sizeof(CT) == 4 || XIVO_LATE_MSUB_F64;
  constexpr bool MF44 = mf44_tile(WM, WN) && sizeof(CT) == 8;
  // 16x16x4 accumulator layout <-> 4x4x4 accumulator layout of one 16x16 block through this wave's LDS pad
  // (pad element (i, j) of the block at j * 17 + i; same-wave LDS writes and reads are ordered)
  double* Tcv = smem_raw + (threadIdx.x >> 6) * (16 * 17);
  auto to_mf44 = [&](acc_t v) -> acc_t {      // v[r] = C[li][lg + 4 r]  ->  component m as documented above
    acc_t o;
#pragma unroll
    for (int r = 0; r < 4; ++r) Tcv[(lg + 4 * r) * 17 + li] = (double)v[r];
#pragma unroll
    for (int m = 0; m < 4; ++m) o[m] = (CT)Tcv[(4 * (li >> 2) + lg) * 17 + 4 * (((li >> 2) + m) & 3) + (li & 3)];
    return o;
  };
  auto from_mf44 = [&](acc_t v) -> acc_t {
    acc_t o;
#pragma unroll
    for (int m = 0; m < 4; ++m) Tcv[(4 * (li >> 2) + lg) * 17 + 4 * (((li >> 2) + m) & 3) + (li & 3)] = (double)v[m];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (CT)Tcv[(lg + 4 * r) * 17 + li];
    return o;
  };
  acc_t acc[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    acc[q] = Cx<CT>::zero();
    if (!LATE_MSUB && (g.epilogue == EPI_SUB_MAT || g.epilogue == EPI_ADD_MAT || g.epilogue == EPI_RSUB_MAT) && is_on(q)) {
      const double* Ms = g.Msub + (long)filt * g.strideMsub;
      const int i = m0 + 16 * arow(slot_a(q)) + li, J0 = n0 + 16 * bcol(slot_b(q));
      const double sgn = g.epilogue == EPI_ADD_MAT ? 1.0 : -1.0;      // RSUB: acc - Msub here, negated in the epilogue
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = J0 + Cx<CT>::crow(lg, r);
        double v = sgn * Ms[i + (long)j * g.ldmsub];
        if (g.McolScale) v *= g.McolScale[(long)filt * g.strideMcol + j];
        acc[q][r] = (CT)v;
      }
      if (MF44) acc[q] = to_mf44(acc[q]);
    }
  }
  if (MF44 && !LATE_MSUB && (g.epilogue == EPI_SUB_MAT || g.epilogue == EPI_ADD_MAT || g.epilogue == EPI_RSUB_MAT)) __syncthreads();   // the pads overlap the k-panels

  const int steps0 = (g.seg[0].K + BK - 1) / BK;
  const int steps1 = g.nseg > 1 ? (g.seg[1].K + BK - 1) / BK : 0;
  const int nsteps = steps0 + steps1;

  constexpr int RA = WM * BK / 16, RB = WN * BK / 16;   // 16-byte loads per thread per panel
  d2 ra[RA], rb[RB];

  // Branch-free panel loads: every per-thread quantity that does not change from one
  // k-step to the next (element offset inside the panel, row validity) is computed once
  // per K-segment; out-of-range rows / k-columns are read from a clamped (always valid)
  // address and zeroed with a select, so the loop body is load + select, no exec masking.
  int eoffA[RA], eoffB[RB];
  unsigned okA = 0, okB = 0;
  const double* segA = nullptr;
  const double* segB = nullptr;
  const double* segS = nullptr;
  int seg_lda = 0, seg_ldb = 0, seg_K = 0, seg_af32 = 0, seg_bf32 = 0;
  auto setup_seg = [&](int sidx) {
    const GemmSeg& sg = g.seg[sidx];
    segA = sg.A + (long)filt * sg.strideA;
    segB = sg.B + (long)filt * sg.strideB;
    segS = sg.scale ? sg.scale + (long)filt * sg.strideScale : nullptr;
    seg_lda = sg.lda; seg_ldb = sg.ldb; seg_K = sg.K; seg_af32 = sg.a_f32; seg_bf32 = sg.b_f32;
    if (sizeof(CT) == 4 && sg.a_f32) segA = reinterpret_cast<const double*>(reinterpret_cast<const float*>(sg.A) + (long)filt * sg.strideA);
    if (sizeof(CT) == 4 && sg.b_f32) segB = reinterpret_cast<const double*>(reinterpret_cast<const float*>(sg.B) + (long)filt * sg.strideB);
    okA = 0; okB = 0;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int idx = tid + 256 * r;
      const int k = idx / (16 * WM), p = idx % (16 * WM);
      const int row = m0 + 2 * p;
      const bool ok = row < g.Mp;
      eoffA[r] = (ok ? row : 0) + k * sg.lda;
      if (ok) okA |= 1u << r;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int idx = tid + 256 * r;
      const int k = idx / (16 * WN), p = idx % (16 * WN);
      const int col = n0 + 2 * p;
      const bool ok = col < g.Np;
      eoffB[r] = (ok ? col : 0) + k * sg.ldb;
      if (ok) okB |= 1u << r;
    }
  };

  // Issue only: nothing below consumes the loaded registers, so the loads stay in flight
  // under the MFMA block; masking (row validity, k-tail) and the optional per-k scale are
  // applied in store_lds, one k-step later.
  int pend_k0 = 0, pend_K = 0, pend_af32 = 0, pend_bf32 = 0;
  const double* pend_S = nullptr;
  auto load_global = [&](int t) {
    if (t == 0) setup_seg(0);
    else if (t == steps0) setup_seg(1);
    const int k0 = (t < steps0 ? t : t - steps0) * BK;
    const bool tail = k0 + BK > seg_K;            // only the last, partial k-step of a segment
    const double* Ab = segA + (long)k0 * seg_lda;
    const double* Bb = segB + (long)k0 * seg_ldb;
    pend_k0 = k0; pend_K = seg_K; pend_S = segS; pend_af32 = seg_af32; pend_bf32 = seg_bf32;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int k = (tid + 256 * r) / (16 * WM);
      const bool kin = !tail || (k0 + k < seg_K);
      const long eo = kin ? eoffA[r] : eoffA[r] - k * seg_lda;
      if (sizeof(CT) == 4 && seg_af32) {   // A kept in HBM as float (it only ever feeds this fp32 product):
        // issue-only 8-byte load of the float pair, bits parked in ra[r][0]; unpacked in store_lds
        ra[r][0] = *reinterpret_cast<const double*>(reinterpret_cast<const float*>(segA) + (long)k0 * seg_lda + eo);
      } else {
        ra[r] = *reinterpret_cast<const d2*>(Ab + eo);
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int k = (tid + 256 * r) / (16 * WN);
      const bool kin = !tail || (k0 + k < seg_K);
      const long eo = kin ? eoffB[r] : eoffB[r] - k * seg_ldb;
      if (sizeof(CT) == 4 && seg_bf32) rb[r][0] = *reinterpret_cast<const double*>(reinterpret_cast<const float*>(segB) + (long)k0 * seg_ldb + eo);
      else rb[r] = *reinterpret_cast<const d2*>(Bb + eo);
    }
  };

  auto store_lds = [&]() {
    const bool tail = pend_k0 + BK > pend_K;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int idx = tid + 256 * r;
      const int k = idx / (16 * WM), p = idx % (16 * WM);
      const bool keep = ((okA >> r) & 1u) && (!tail || pend_k0 + k < pend_K);
      if (sizeof(CT) == 4 && pend_af32) {
        const double bits = keep ? ra[r][0] : 0.0;      // +0.0 is also two float zeros
        *reinterpret_cast<double*>(As + k * LDAS + 2 * p) = bits;
      } else {
        *reinterpret_cast<pair_t*>(As + k * LDAS + 2 * p) = Cx<CT>::cvt(keep ? ra[r] : d2{0.0, 0.0});
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int idx = tid + 256 * r;
      const int k = idx / (16 * WN), p = idx % (16 * WN);
      const bool keep = ((okB >> r) & 1u) && (!tail || pend_k0 + k < pend_K);
      if (sizeof(CT) == 4 && pend_bf32) {                // B kept in HBM as float: the pair's bits as they are (no per-k scale in this mode)
        *reinterpret_cast<double*>(Bs + k * LDBS + 2 * p) = keep ? rb[r][0] : 0.0;
        continue;
      }
      d2 v = keep ? rb[r] : d2{0.0, 0.0};
      if (pend_S) v *= pend_S[pend_k0 + k < pend_K ? pend_k0 + k : pend_K - 1];
      *reinterpret_cast<pair_t*>(Bs + k * LDBS + 2 * p) = Cx<CT>::cvt(v);
    }
  };

  if (nsteps > 0) load_global(0);
  for (int t = 0; t < nsteps; ++t) {
    store_lds();
    __syncthreads();
    if (t + 1 < nsteps) load_global(t + 1);
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      if constexpr (MF44) {
        CT bb[NB];
#pragma unroll
        for (int x = 0; x < NB; ++x) bb[x] = Bs[(4 * s + lg) * LDBS + 16 * bcol(x) + li];
#pragma unroll
        for (int m = 0; m < 4; ++m) {       // one rotation of the A fragments at a time: NA live registers, not 4 NA
          CT a[NA];
#pragma unroll
          for (int x = 0; x < NA; ++x) a[x] = As[(4 * s + lg) * LDAS + 16 * arow(x) + ((li + 4 * m) & 15)];
#pragma unroll
          for (int q = 0; q < NS; ++q) {
            if (is_on(q)) acc[q][m] = Cx<CT>::mfma44(bb[slot_b(q)], a[slot_a(q)], acc[q][m]);
          }
        }
      } else {
        CT a[NA], bb[NB];
#pragma unroll
        for (int x = 0; x < NA; ++x) a[x] = As[(4 * s + lg) * LDAS + 16 * arow(x) + li];
#pragma unroll
        for (int x = 0; x < NB; ++x) bb[x] = Bs[(4 * s + lg) * LDBS + 16 * bcol(x) + li];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          if (is_on(q))
            acc[q] = Cx<CT>::mfma(bb[slot_b(q)], a[slot_a(q)], acc[q]);
        }
      }
    }
    __syncthreads();
  }

  // fp32 products: Msub (fp64) is added in the epilogue, fetched through a small ring a few slots ahead of
  // the stores (holding all of it would cost 128 VGPRs and the third workgroup per CU)
  constexpr int MSD = 4;
  double msring[LATE_MSUB ? MSD : 1][4];
  const bool late_ms = LATE_MSUB && (g.epilogue == EPI_SUB_MAT || g.epilogue == EPI_ADD_MAT || g.epilogue == EPI_RSUB_MAT);
  auto load_ms = [&](int q) {
    if (q >= NS || !is_on(q)) return;
    const double* Ms = g.Msub + (long)filt * g.strideMsub;
    const double sgn = g.epilogue == EPI_SUB_MAT ? -1.0 : 1.0;       // RSUB: + Msub, the accumulator is negated below
    const int i = m0 + 16 * arow(slot_a(q)) + li, J0 = n0 + 16 * bcol(slot_b(q));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = J0 + Cx<CT>::crow(lg, r);
      double v = sgn * Ms[i + (long)j * g.ldmsub];
      if (g.McolScale) v *= g.McolScale[(long)filt * g.strideMcol + j];
      msring[LATE_MSUB ? q % MSD : 0][r] = v;
    }
  };
  if (late_ms) {
#pragma unroll
    for (int q = 0; q < MSD - 1; ++q) load_ms(q);
  }

  // Epilogue. acc[q][r] = C[i = I0 + li][j = J0 + lg + 4r]
  double* Cb = g.C + (long)filt * g.strideC;
  double* C2b = g.C2 ? g.C2 + (long)filt * g.strideC2 : nullptr;
  const double* dg = g.diag ? g.diag + (long)filt * g.strideDiag : nullptr;
  double* Tw = smem_raw + wave * (16 * 17);  // per-wave 16x16 transpose pad (main-loop LDS is free now)
  const bool need_t = (g.lower_only && !g.no_mirror) || C2b;
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    if (late_ms) load_ms(q + MSD - 1);
    if (!is_on(q)) continue;
    const int I0 = m0 + 16 * arow(slot_a(q)), J0 = n0 + 16 * bcol(slot_b(q));
    const int i = I0 + li;
    if (MF44) acc[q] = from_mf44(acc[q]);
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = J0 + Cx<CT>::crow(lg, r);
      v[r] = (double)acc[q][r];
      if (g.epilogue == EPI_RSUB_MAT) v[r] = -v[r];
      if (late_ms) v[r] += msring[LATE_MSUB ? q % MSD : 0][r];
      if (g.epilogue == EPI_ADD_DIAG) {
        if (i == j) v[r] += dg[i];
      } else if (g.epilogue == EPI_SUB_IDENT) {
        if (i == j) v[r] -= 1.0;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jl = Cx<CT>::crow(lg, r);
      const int j = J0 + jl;
      if (!g.lower_only || i >= j) {
#if XIVO_NT_STORE
        __builtin_nontemporal_store(v[r], &Cb[i + (long)j * g.ldc]);     // the output is not read again by this kernel
#else
        Cb[i + (long)j * g.ldc] = v[r];
#endif
      }
      if (need_t) Tw[jl * 17 + li] = v[r];   // T[j_local][i_local]
    }
    if (need_t) {
      // The lower triangle is authoritative and is mirrored, so a symmetric result is
      // exactly symmetric (the reference never re-symmetrises P, estimator.cpp:1280;
      // its asymmetry is rounding noise). Going through the LDS transpose, the mirror
      // (and the optional transposed second output) is also written in 128-byte runs.
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double t = Tw[li * 17 + lg + 4 * r];   // element (i2 = I0 + lg + 4r, j2 = J0 + li)
        const int i2 = I0 + lg + 4 * r, j2 = J0 + li;
#if XIVO_NT_STORE
        if (g.lower_only && !g.no_mirror && i2 > j2) __builtin_nontemporal_store(t, &Cb[j2 + (long)i2 * g.ldc]);
#else
        if (g.lower_only && !g.no_mirror && i2 > j2) Cb[j2 + (long)i2 * g.ldc] = t;
#endif
        if (C2b && (g.c2_rows == 0 || i2 < g.c2_rows)) C2b[j2 + (long)i2 * g.ldc2] = t;
      }
    }
  }
}

#ifndef XIVO_GEMM22_BK
#define XIVO_GEMM22_BK 16     // A/B: k-panel depth of the <2,2> instantiation (64 x 64 tiles: the latency route's products)
#endif
#ifndef XIVO_GEMM33_BK
#define XIVO_GEMM33_BK 16     // A/B: k-panel depth of the <3,3> instantiation
#endif
template <int WM, int WN>
constexpr int pick_bk() {
  return (WM == 3 && WN == 3) ? XIVO_GEMM33_BK : ((WM == 2 && WN == 2) ? XIVO_GEMM22_BK : 16);   // 32 measured neutral on MI355X and costs 32 more staging VGPRs
}

// waves per SIMD (= workgroups per CU) the register budget of an instantiation is cut for (A/B: -DXIVO_GEMM33_WAVES etc.)
#ifndef XIVO_GEMM33_WAVES
#define XIVO_GEMM33_WAVES 4   // 128 VGPRs (29 spilled) -> four workgroups per CU: P - V^T Y at N = 276 2.87 -> 2.45 ms per 4096 filters (5: 342 spilled)
#endif
#ifndef XIVO_GEMM44F_WAVES
#define XIVO_GEMM44F_WAVES 0
#endif
#ifndef XIVO_GEMM34D_WAVES
#define XIVO_GEMM34D_WAVES 0
#endif
#ifndef XIVO_GEMM44D_WAVES
#define XIVO_GEMM44D_WAVES 0
#endif
template <int WM, int WN, typename CT>
constexpr int gemm_waves() {
  if (XIVO_GEMM33_WAVES > 0 && WM == 3 && WN == 3 && sizeof(CT) == 8) return XIVO_GEMM33_WAVES;
  if (XIVO_GEMM44F_WAVES > 0 && WM == 4 && WN == 4 && sizeof(CT) == 4) return XIVO_GEMM44F_WAVES;
  if (XIVO_GEMM44D_WAVES > 0 && WM == 4 && WN == 4 && sizeof(CT) == 8) return XIVO_GEMM44D_WAVES;
  if (XIVO_GEMM34D_WAVES > 0 && WM == 3 && WN == 4 && sizeof(CT) == 8) return XIVO_GEMM34D_WAVES;
  return (sizeof(CT) == 4 && WM * WN <= 16) ? 3 : 2;
}
template <int WM, int WN, typename CT>
__global__ __launch_bounds__(256, (gemm_waves<WM, WN, CT>())) void gemm_nt_f64_kernel(GemmArgs g) {
  constexpr int BK = pick_bk<WM, WN>();
  constexpr int BM = 32 * WM, BN = 32 * WN;
  constexpr int LDAS = BM + 16, LDBS = BN + 16;
  __shared__ __attribute__((aligned(16))) double smem[BK * (LDAS + LDBS)];

  // XCD-aware block -> (filter, tile): blocks b and b+8 share an XCD (and its
  // L2), so all tiles of one filter are kept on one XCD, adjacent in dispatch.
  const int nt = g.tiles_m * g.tiles_n;
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;
  const int filt = (slot / nt) * 8 + xcd;
  // Rotate the tile order from filter to filter: the dispatcher hands consecutive workgroups of an
  // XCD to its CUs round-robin, so a fixed order pins every "tile 1" to the same quarter of the CUs -
  // with unequal tiles (symmetric outputs: full, strip and skipped tiles) that serialised the heavy
  // tiles on 8 of 32 CUs (measured: one off-diagonal tile per filter cost as much as all four).
  const int tile = (slot + slot / nt) % nt;
  if (filt >= g.batch) return;
  if (g.skip_status && g.skip_status[filt]) return;
  const int tm = tile % g.tiles_m, tn = tile / g.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  if (g.lower_only && n0 >= m0 + BM) return;  // tile strictly above the diagonal
  const bool inside = (m0 + BM <= g.Mp) && (n0 + BN <= g.Np);
  if constexpr (WM == 4 && WN == 4) {
    if (g.lower_only == 1 && m0 == n0) {
      if (inside) {
        const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        if (w == 0) gemm_tile<CT, WM, WN, BK, true, true, 0>(g, filt, m0, n0, smem);
        else if (w == 1) gemm_tile<CT, WM, WN, BK, true, true, 1>(g, filt, m0, n0, smem);
        else if (w == 2) gemm_tile<CT, WM, WN, BK, true, true, 2>(g, filt, m0, n0, smem);
        else gemm_tile<CT, WM, WN, BK, true, true, 3>(g, filt, m0, n0, smem);
      } else {
        gemm_tile<CT, WM, WN, BK, true, false, -1>(g, filt, m0, n0, smem);
      }
      return;
    }
  }
  // interior tile with every block wanted (for lower_only: strictly below the diagonal)
  // lower_only == 2: diagonal tiles are computed in full (dense FAST path) and only stored as
  // lower triangle + mirror
  if (inside && (!g.lower_only || n0 + BN <= m0 || g.lower_only == 2)) gemm_tile<CT, WM, WN, BK, false, true, -1>(g, filt, m0, n0, smem);
  else gemm_tile<CT, WM, WN, BK, false, false, -1>(g, filt, m0, n0, smem);
}

template <int WM, int WN>
int launch_t(const GemmArgs& a, hipStream_t stream) {
  GemmArgs g = a;
  g.tiles_m = (g.Mp + 32 * WM - 1) / (32 * WM);
  g.tiles_n = (g.Np + 32 * WN - 1) / (32 * WN);
  const int groups = (g.batch + 7) / 8;
  const int grid = groups * 8 * g.tiles_m * g.tiles_n;
  if (grid <= 0) return 0;
  if (g.fp32) hipLaunchKernelGGL((gemm_nt_f64_kernel<WM, WN, float>), dim3(grid), dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((gemm_nt_f64_kernel<WM, WN, double>), dim3(grid), dim3(256), 0, stream, g);
  return (int)hipGetLastError();
}

int pick_w(int dim) {
  // choose the per-dimension wave tile count (2..5, i.e. block 64..160) that
  // minimises padded work; ties go to the larger tile (fewer L2 re-reads).
  int best = 2, best_pad = 1 << 30;
  for (int w = 2; w <= 5; ++w) {
    const int blk = 32 * w;
    const int padded = ((dim + blk - 1) / blk) * blk;
    if (padded < best_pad || (padded == best_pad && w > best)) {
      best = w;
      best_pad = padded;
    }
  }
  return best;
}

}  // namespace

void gemm_pick_tile(int Mp, int Np, int lower_only, int* WM, int* WN) {
  if (lower_only && Mp > 64) {  // symmetric output: square 128x128 tiles (strip-balanced diagonals)
    *WM = 4;
    *WN = 4;
    return;
  }
  // among the instantiated wave tiles minimise (padded MFMA work) x (1 + 0.5 (1/WM + 1/WN)) - the
  // second factor stands for operand bytes and per-k-step overhead per flop, which fall with the
  // tile size; ties go to the larger tile. <4,5> is not built (spills), <5,4> is.
  static const int cand[][2] = {{2, 2}, {2, 3}, {2, 4}, {2, 5}, {3, 2}, {3, 3}, {3, 4}, {3, 5},
                                {4, 2}, {4, 3}, {4, 4}, {5, 2}, {5, 3}, {5, 4}};
  double best_cost = -1;
  int bw = 2, bn = 2;
  for (const auto& c : cand) {
    const long pr = ((Mp + 32 * c[0] - 1) / (32 * c[0])) * 32L * c[0];
    const long pc = ((Np + 32 * c[1] - 1) / (32 * c[1])) * 32L * c[1];
    // tiles that run the 4x4x4 form issue 1.45x the flops per cycle (72 vs 49 TFLOP/s)
    const double cost = (double)pr * pc * (1.0 + 0.5 * (1.0 / c[0] + 1.0 / c[1])) * (mf44_tile(c[0], c[1]) ? 1.0 : 1.45);
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && c[0] * c[1] > bw * bn)) {
      best_cost = cost; bw = c[0]; bn = c[1];
    }
  }
  *WM = bw;
  *WN = bn;
}

static void pick_for(const GemmArgs& a, int* wm_out, int* wn_out);

void gemm_kernel_label(const GemmArgs& a, char* buf, size_t n) {
  int wm, wn;
  pick_for(a, &wm, &wn);
  snprintf(buf, n, "gemm_nt_f64_kernel<%d,%d,%s>", wm, wn, a.fp32 ? "float" : "double");
}

int launch_gemm_nt_f64(const GemmArgs& a_in, hipStream_t stream) {
  GemmArgs a = a_in;
  int wm, wn;
  pick_for(a, &wm, &wn);
  if (wm == 3 && wn == 3 && a.lower_only == 1) a.lower_only = 2;   // diagonal tiles in full on the straight-line path, stored as lower triangle + mirror
#define XIVO_GEMM_CASE(M_, N_) \
  if (wm == M_ && wn == N_) return launch_t<M_, N_>(a, stream);
  XIVO_GEMM_CASE(2, 2) XIVO_GEMM_CASE(2, 3) XIVO_GEMM_CASE(2, 4) XIVO_GEMM_CASE(2, 5)
  XIVO_GEMM_CASE(3, 2) XIVO_GEMM_CASE(3, 3) XIVO_GEMM_CASE(3, 4) XIVO_GEMM_CASE(3, 5)
  XIVO_GEMM_CASE(4, 2) XIVO_GEMM_CASE(4, 3) XIVO_GEMM_CASE(4, 4)
  XIVO_GEMM_CASE(5, 2) XIVO_GEMM_CASE(5, 3) XIVO_GEMM_CASE(5, 4)
#undef XIVO_GEMM_CASE
  return (int)hipErrorInvalidValue;
}

static void pick_for(const GemmArgs& a, int* wm_out, int* wn_out) {
  int wm, wn;
  gemm_pick_tile(a.Mp, a.Np, a.lower_only, &wm, &wn);
  // accumulators initialised from memory (T = K(HP) - P): the 64 extra loads per lane sit in the
  // tile prologue; the narrower 128x64 tile (3 instead of 2 workgroups per CU) hides them
  // (measured 0.54 vs 0.72 ms per 1024 filters at N=250)
  if ((a.epilogue == EPI_SUB_MAT || a.epilogue == EPI_ADD_MAT || a.epilogue == EPI_RSUB_MAT) && !a.lower_only && wm == 4 && wn == 4)
    wn = 2;
  // short contractions on a non-square output (the lead product of an online-calibration stacking: K = 48, three k-panels):
  // all prologue (the accumulators start from the output) and epilogue - small tiles, more workgroups in flight
  // (N = 276: <3,4> 1.20 -> <3,2> 0.85 ms per 4096 filters; <3,3> 1.0, <5,2> 1.35)
  {
    const int ktot = a.seg[0].K + (a.nseg > 1 ? a.seg[1].K : 0);
    if (!a.lower_only && !a.fp32 && a.Mp != a.Np && ktot <= 64) {
      if (a.Mp % 96 == 0) { wm = 3; wn = 2; }
      else if (a.Mp % 64 == 0) { wm = 2; wn = 2; }
    }
  }
  // few filters (the latency route of the update, chol_trsm.hip): 64 x 64 tiles - ten workgroups per symmetric 256 x 256
  // output instead of three, on a chip that is empty anyway
  if (a.small_tiles && a.lower_only && a.Mp > 64 && !a.fp32) { wm = 2; wn = 2; }
  // fp64 symmetric outputs whose side is a multiple of 96 but not of 128 (N = 276 -> 288: the online-calibration build: 6
  // tiles of 96^2 instead of 6 of 128^2 over a 384-wide grid), or beyond 256 and not a multiple of 128 (N = 400: config 4):
  // 96 x 96 tiles on the instantiation that keeps four workgroups per CU (these products are short in K - M <= 304 - and
  // bound by the latency of their panel loads, not by MFMA or HBM: P - V^T Y 4.05 -> 2.45 ms per 4096 filters at N = 276,
  // 7.35 -> 6.9 ms at N = 400; the fp32 product of config 4 measured the same on either tile and keeps 128 x 128)
  else if (a.lower_only && !a.fp32 && a.Mp == a.Np && (a.Mp % 96 == 0 || a.Mp > 256) && a.Mp % 128 != 0) { wm = 3; wn = 3; }
  *wm_out = wm; *wn_out = wn;
}

}  // namespace xivo_hip
