// The whole measurement update of one filter in ONE kernel, for the shapes a CU holds (TUM-VI: N = 203, M <= 60; BASELINE
// config 2: N = 150, M = 100): `S_ = H_ * P_ * H_^T + diagR_; K_^T = S_.ldlt().solve(H_ * P_); err_ = K_ * inn_;
// P_ = (K H - I) P (K H - I)^T + K R K^T` (/root/reference/src/estimator.cpp:1257-1288) and, in front of it, the numeric core
// of Estimator::MHGating (/root/reference/src/update.cpp:60-96) on the 2 x 2 diagonal blocks of S.
//
// The five-kernel sparse pipeline (capi.hip, update_sparse_range) moves ~2.2 MB per filter at these sizes against 0.45 MB of
// compulsory traffic: P H^T, S, the factor and its inverse blocks, the gated right-hand sides all cross HBM between kernels.
// Here they never leave the CU - one workgroup per filter, wave w owns state rows [16 w, 16 w + 16):
//   1  H P (rows of H x the wave's 16 state columns) straight into the MFMA accumulator layout the solve wants:
//      the 12 common columns of the row-pair compressed H (ell.h) as a (16 x 12)(12 x 16) product on the matrix pipe, the 9 private
//      columns as gathers of P's columns (16 contiguous doubles = one 128-byte line per column and wave) + FMAs.
//      P is symmetric by contract, so "column k, rows 16 w .." is read as row k of the wave's own column block.
//   2  the waves park their H P in LDS as the slab ell<S> would have loaded (state index x measurement column) and walk the
//      row pairs over it: S = H (P H^T) + diag(R), lower triangle + diagonal blocks, into the padded 16 x 17 block slots the
//      factorisation works on - in XC-column passes when S and the slab do not fit together
//   3  gate: chi-square distances from S's diagonal, threshold relaxation, rejected pairs decoupled in S (0, unit diagonal),
//      their right-hand sides zeroed in the registers, inn = 0, R = 1
//   4  S = L L^T in LDS (chol_device.h: the same routines and operand order as every other factorisation here)
//   5  W = L^-1 (H P), K^T = L^-T W, dx = K inn, D = W - L^T K^T  - all in registers (trsm_lds_kernel.h, KEEPW form)
//   6  P+ = P - (W - D)^T (W + D), lower triangle + mirror, in place (sym_tiles_from_regs)
// HBM traffic per filter: the columns of P that H names (<= N^2), P's lower triangle, P+ out, the compressed rows.
#include <hip/hip_runtime.h>
#ifndef XIVO_FUSED_TRACE
#define XIVO_FUSED_TRACE 0
#endif
#if XIVO_FUSED_TRACE
__device__ unsigned long long xivo_fused_trace2_buf[128 * 16 * 32];
#define XIVO_CHAIN_STAMP(slot) do { if ((threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < 128 && (slot) < 32) \
    xivo_fused_trace2_buf[((blockIdx.x >> 6) * 16 + (threadIdx.x >> 6)) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#include "trsm_lds_kernel.h"
#include "ell.h"
#include "gate_device.h"
#include "fused_update.h"

// XIVO_FUSED_TRACE (scripts/build_variant.sh): shader-clock stamps of wave 0 at the phase boundaries of every 64th workgroup
#ifndef XIVO_FUSED_TRACE
#define XIVO_FUSED_TRACE 0
#endif
#if XIVO_FUSED_TRACE
__device__ unsigned long long xivo_fused_trace_buf[512 * 16];
#define FTR(i) do { if (threadIdx.x == 0 && (blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < 512) \
    xivo_fused_trace_buf[(blockIdx.x >> 6) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int xivo_hip_debug_read_fused_trace(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(xivo_fused_trace_buf), (size_t)n * sizeof(unsigned long long));
}
// second level: every wave, 16 slots (the factorisation: slot 2 j / 2 j + 1 = entry / exit of column j's work of that wave)
#define FTR2(slot) do { if ((threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < 128 && (slot) < 32) \
    xivo_fused_trace2_buf[((blockIdx.x >> 6) * 16 + (threadIdx.x >> 6)) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int xivo_hip_debug_read_fused_trace2(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(xivo_fused_trace2_buf), (size_t)n * sizeof(unsigned long long));
}
#else
#define FTR(i) do {} while (0)
#define FTR2(slot) do {} while (0)
#endif

// XIVO_FUSED_ABL (timing-only ablations, scripts/build_variant.sh; results are WRONG for any value but 0): 1 stop behind the
// gathers of phase 1, 2 stop behind the factorisation, 3 no stores of P+, 4 no mirror stores, 5 no loads of the P tiles,
// 6 stop in front of the factorisation, 8 no forward substitution next to the factorisation and stop behind it
#ifndef XIVO_FUSED_ABL
#define XIVO_FUSED_ABL 0
#endif
#ifndef XIVO_FUSED_DIAG_CHAIN
#define XIVO_FUSED_DIAG_CHAIN 0   // A/B: the sixteen-pivot chain (factor_invert_diag_chain) instead of the four-column form
#endif
#ifndef XIVO_FUSED_BALANCE
#define XIVO_FUSED_BALANCE 1      // product tiles oriented by SIMD load at 10 / 13 column blocks (0: the cyclic rule everywhere)
#endif
#ifndef XIVO_FUSED_TU
#define XIVO_FUSED_TU 4
#endif
#ifndef XIVO_FUSED_FWD_LATE
#define XIVO_FUSED_FWD_LATE 0   // A/B: the forward substitution behind the factorisation instead of next to it
#endif
#ifndef XIVO_FUSED_GVAR
#define XIVO_FUSED_GVAR 0   // timing-only variants of the gather's address pattern (with XIVO_FUSED_ABL=1)
#endif

namespace xivo_hip {

namespace {

constexpr int FU_CWU = 12, FU_PWU = 9, FU_NSLOT = FU_CWU + FU_PWU;

// LDS map (doubles): slab | factor slots | diagonal blocks | coefficients | slot indices | inn | diagR | distances | flags
struct FusedLds {
  int slab, fac, diag, ops, pidx, inn, dR, dist, rej, total;
};
__host__ __device__ inline FusedLds fused_lds_map(int Np, int Mp, int XC) {
  const int nb = Mp / 16, pairs = Mp / 2;
  FusedLds m;
  m.slab = 0;
  m.fac = Np * XC;
  m.diag = m.fac + nb * (nb + 1) / 2 * 272;
  m.ops = m.diag + nb * 272;
  m.pidx = m.ops + pairs * FU_NSLOT * 2;
  m.inn = m.pidx + (pairs * ELL_PIW + ELL_CW + 3) / 4;   // 16-bit indices
  m.dR = m.inn + Mp;
  m.dist = m.dR + Mp;
  m.rej = m.dist + pairs + 2;
  m.total = m.rej + Mp / 8 + 2;                          // one byte per row + the breakdown flag
  return m;
}

// hand-placed vector-memory requests and waits of the gather phase (see the kernel, phase 1)
typedef int fu_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fu_gather(double& dst, unsigned off, const fu_v4i& rs) {
  asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(off), "s"(rs) : "memory");
}
template <int N>
__device__ __forceinline__ void fu_wait3(double& r0, double& r1, double& r2, int& tok) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(tok) : "n"(N) : "memory");
}
__device__ __forceinline__ void fu_anchor(unsigned& a, int& b) { asm volatile("" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void fu_anchor(unsigned& a, int& b, double x, double y) { asm volatile("" : "+v"(a), "+v"(b) : "v"(x), "v"(y)); }

// The covariance product of the one-kernel update when every column block's operand is in LDS at once (one phase) and every
// wave owns exactly NTU tiles: tile t of wave w is block pair (w, w - t mod nwl) - for an even nwl the last one is a dummy
// for half the waves (computed, not stored). Straight-line code: the -P tile of step t is requested PD steps ahead, and
// nothing between a request and its use is conditional on anything but the wave index - so the compiler's wait counts are
// exact (vmcnt(n) = the younger requests and stores). In the general loop below the requests and stores sit inside
// `if (t < nt)` / per-lane `if` blocks: its bookkeeping gives up at every join, every wait is vmcnt(0), and each tile waits
// for the loads just requested AND for the acknowledgement of the previous tile's stores - a full memory round trip per
// tile, 46 k cycles for seven tiles of sixteen MFMAs. To keep the stores unconditional the diagonal tile writes ALL its
// lanes twice: the direct pass stores the whole block (its upper half is then overwritten), the mirror pass stores the
// transposed lower half into the upper half and, in the lower half, the direct pass's own value again.
// Which wave forms which tile when the waves do not spread evenly over the four SIMDs (wave w runs on SIMD w % 4): ten column
// blocks are 3 + 3 + 2 + 2 waves, thirteen 4 + 3 + 3 + 3, and with the cyclic rule above every wave forms the same number of
// tiles - the SIMDs with one wave more set the time of the phase (18 against 12 tile slots; 28 against 21). Tile {a, b} can be
// formed by wave a or by wave b (each holds its own row block of W - D in registers, W + D of every block is in LDS): these
// tables orient the 45 / 78 pairs so that every SIMD forms 14 + 14 + 14 + 13 of the 55 tiles, or 23 + 23 + 23 + 22 of the 91
// (a maximum-flow orientation, computed offline; the wave's own diagonal tile first, -1 = no tile).
#define XIVO_FUSED_TILES10 \
    {0, 1, 3, 4, 5, -1, -1, -1}, {1, 2, 3, 4, 5, -1, -1, -1}, {2, 0, 3, 4, 5, 6, 8, -1}, {3, 4, 5, 6, 7, 8, 9, -1}, {4, 5, 6, 8, 9, -1, -1, -1}, \
    {5, 7, 8, 9, -1, -1, -1, -1}, {6, 0, 1, 5, 7, 8, 9, -1}, {7, 0, 1, 2, 4, 8, 9, -1}, {8, 0, 1, 9, -1, -1, -1, -1}, {9, 0, 1, 2, -1, -1, -1, -1}
#define XIVO_FUSED_TILES13 \
    {0, 1, 2, 4, 6, 7, -1, -1}, {1, 2, 3, 4, 5, 6, 7, 9}, {2, 3, 4, 5, 6, 7, 8, 10}, {3, 0, 4, 5, 6, 7, 8, 10}, {4, 5, 8, 9, 10, 12, -1, -1}, \
    {5, 0, 6, 7, 8, 9, 11, 12}, {6, 4, 7, 8, 9, 10, 11, 12}, {7, 4, 8, 9, 10, 11, 12, -1}, {8, 0, 1, 9, 11, 12, -1, -1}, {9, 0, 2, 3, 10, 11, 12, -1}, \
    {10, 0, 1, 5, 8, 11, 12, -1}, {11, 0, 1, 2, 3, 4, 12, -1}, {12, 0, 1, 2, 3, -1, -1, -1}
__device__ const signed char kFusedTiles10[10][8] = {XIVO_FUSED_TILES10};
__device__ const signed char kFusedTiles13[13][8] = {XIVO_FUSED_TILES13};
// (host copies for xivo_hip_selftest_fused_tiles: every block pair exactly once, the counts the kernel's dispatch assumes)
static const signed char kFusedTiles10Host[10][8] = {XIVO_FUSED_TILES10};
static const signed char kFusedTiles13Host[13][8] = {XIVO_FUSED_TILES13};

// TAB = 0: the cyclic rule (tile t of wave w is the pair (w, w - t mod nwl), NTU = nwl / 2 + 1 tiles per wave, the last one a dummy
// for half the waves of an even nwl); TAB = 10 / 13: the wave's list of the tables above, NTU of them, all real.
template <int NBM, int NTU, int TAB = 0>
__device__ __forceinline__ void fused_product_one_phase(const d4 (&X)[NBM], const double* ybuf, double* tsc, double* Pio, int ldp, int nb, int nwl,
                                                        int wave, int lane) {
  constexpr int PD = 4;
  static_assert(NTU >= PD, "the prefetch ring is primed with PD tiles");
  const int li = lane & 15, lg = lane >> 4;
  const __amdgpu_buffer_rsrc_t rO = buf_rsrc(Pio);
  const unsigned vM = (unsigned)(li + lg * ldp) * 8u;
  auto block_of = [&](int t, int& jb, bool& real) {
    if constexpr (TAB == 10) { jb = __builtin_amdgcn_readfirstlane(kFusedTiles10[wave][t]); real = true; }
    else if constexpr (TAB == 13) { jb = __builtin_amdgcn_readfirstlane(kFusedTiles13[wave][t]); real = true; }
    else {
      jb = wave - t; if (jb < 0) jb += nwl;
      real = 2 * t < nwl || (2 * t == nwl && wave > jb);
    }
  };
  d4 ring[PD];
  auto request = [&](auto tc) {
    constexpr int t = decltype(tc)::value;
    int jb; bool real; block_of(t, jb, real);
    const int ba = jb <= wave ? wave : jb, bb = jb <= wave ? jb : wave;       // block (ba, bb), ba >= bb: P's lower triangle
#pragma unroll
    for (int r = 0; r < 4; ++r) ring[t % PD][r] = XIVO_FUSED_ABL == 5 ? 1.0 : buf_ld_once(rO, vM, (unsigned)(16 * ba + (16 * bb + 4 * r) * ldp) * 8u);
  };
  static_for<(PD < NTU ? PD : NTU)>([&](auto tc) { request(tc); });
  lds_barrier();                                   // every wave's operand W + D is in place
  static_for<NTU>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    int jb; bool real; block_of(t, jb, real);
    d4 acc = -ring[t % PD];
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    if constexpr (t + PD < NTU) request(std::integral_constant<int, t + PD>{});
    const double* Bop = ybuf + jb * nb * 256 + lane;
    if (jb <= wave) {
#pragma unroll
      for (int mb = 0; mb < NBM; ++mb) {
        if (mb < nb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = mfma(Bop[(mb * 4 + r) * 64], X[mb][r], acc);
        }
      }
    } else {
#pragma unroll
      for (int mb = 0; mb < NBM; ++mb) {
        if (mb < nb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = mfma(X[mb][r], Bop[(mb * 4 + r) * 64], acc);
        }
      }
    }
    // the tile (block (ba, bbk), ba >= bbk: rows li, columns lg + 4 r) and its mirror image through the 2 KB transpose in LDS
    // (four full 128-byte lines per store instruction instead of sixteen 32-byte pieces)
    const int ba = jb <= wave ? wave : jb, bbk = jb <= wave ? jb : wave;
#pragma unroll
    for (int r = 0; r < 4; ++r) tsc[(lg + 4 * r) * 16 + (li ^ (lg + 4 * r))] = -acc[r];      // element (a = li, b = lg + 4 r)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double mv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double v = tsc[li * 16 + ((lg + 4 * r) ^ li)];                                        // element (a = lg + 4 r, b = li)
      mv[r] = (jb != wave || lg + 4 * r > li) ? v : -acc[r];                                      // diagonal tile: the lower triangle is authoritative
    }
    if (t + 1 < NTU || real) {                     // (only the last step can be a dummy: nothing is counted behind it)
      if (XIVO_FUSED_ABL != 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) buf_st_out(-acc[r], rO, vM, (unsigned)(16 * ba + (16 * bbk + 4 * r) * ldp) * 8u);
      }
      if (XIVO_FUSED_ABL != 3 && XIVO_FUSED_ABL != 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) buf_st_out(mv[r], rO, vM, (unsigned)(16 * bbk + (16 * ba + 4 * r) * ldp) * 8u);
      }
    }
  });
}

// PW = private slots per row pair the kernel walks: 9 (group anchor 6 + feature 3: the in-state rows FillJacobianBlock stacks) or 6
// for batches whose widest pair uses no more (the gather then needs 24 instead of 36 column pieces per unit - three 1 KB requests
// instead of five - and the S walk six gathers per pair instead of nine). The LDS map keeps the 9-slot sizes either way.
template <int NBM, int NWV, int XC, int GD, int PW>
__global__ __launch_bounds__(64 * NWV, 1) void fused_update_f64_kernel(FusedArgs g) {
  static_assert(PW == 9 || PW == 6, "private slots walked: a multiple of three (the S walk takes them three at a time)");
  constexpr int BLK = 16 * 17, CWU = FU_CWU, PWU = PW, NSLOT = FU_CWU + PW, KS = CWU / 4, NC = XC / 16;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int filt = blockIdx.x;
  if (filt >= g.batch) return;
  const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = NT >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int Np = g.Np, Mp = g.Mp, nb = Mp / 16, pairs = Mp / 2, nblk = nb * (nb + 1) / 2;
  const FusedLds map = fused_lds_map(Np, Mp, XC);
  double* tile = sm + map.slab;
  double* sL = sm + map.fac;
  double* sD = sm + map.diag;
  double* ops = sm + map.ops;
  unsigned short* pidx = reinterpret_cast<unsigned short*>(sm + map.pidx);
  double* sInn = sm + map.inn;
  double* sR = sm + map.dR;
  double* sdist = sm + map.dist;
  unsigned char* sRej = reinterpret_cast<unsigned char*>(sm + map.rej);
  int& sBad = *reinterpret_cast<int*>(sRej + Mp);
  int& sAny = *reinterpret_cast<int*>(sRej + Mp + 4);          // the gate rejected at least one pair
  double* Pio = g.P + (long)filt * g.strideP;
  double* innG = g.inn + (long)filt * g.strideInn;
  double* dRG = g.diagR + (long)filt * g.strideR;
  FTR(0);

  // ---- 0  coefficients and slot indices of every row pair -> LDS (ell_tile_kernel's stage_ops), inn / diagR next to them
  {
    const d2* __restrict__ gv = reinterpret_cast<const d2*>(g.ell.val + (long)filt * g.ell.stride_val());
    for (int e = tid; e < pairs * NSLOT; e += NT) {
      const int p = e / NSLOT, t = e % NSLOT;
      *reinterpret_cast<d2*>(ops + 2 * e) = gv[p * ELL_W + (t < CWU ? t : ELL_CW + (t - CWU))];
    }
    const int* __restrict__ gi = g.ell.idx + (long)filt * g.ell.stride_idx();
    for (int e = tid; e < pairs * ELL_PIW; e += NT) {
      const int p = e / ELL_PIW, t = e % ELL_PIW;
      pidx[e] = t < PWU ? (unsigned short)gi[p * ELL_W + ELL_CW + t] : (unsigned short)0;
    }
    if (tid < ELL_CW) pidx[pairs * ELL_PIW + tid] = (unsigned short)gi[tid];   // the common slots (the same in every pair)
    for (int m = tid; m < Mp; m += NT) { sInn[m] = innG[m]; sR[m] = dRG[m]; sRej[m] = 0; }
    if (tid == 0) { sBad = 0; sAny = 0; }
  }
  __syncthreads();
  FTR(1);
  FTR2(28);

  // ---- 1  X = H P, the wave's 16 state columns: X[i][r] of lane (li, lg) = (H P)[16 i + 4 r + lg, c0 + li]
  const int c0 = 16 * wave;
  const __amdgpu_buffer_rsrc_t rP = buf_rsrc(Pio);
  const unsigned vP = (unsigned)(c0 + li) * 8u;
  d4 X[NBM];
  {
    const unsigned short* cidx = pidx + pairs * ELL_PIW;
    double cb[KS];                                                 // B operand: row ck = common slot 4 s + lg of P, the wave's columns
                                                                   // (requested behind the last unit of the walk below)
    // private slots, in the row order of ell_tile_kernel's walk: register r of block row i stands for row (r & 1) of pair
    // 8 i + lg + 4 (r >> 1) - both rows of a pair in one lane, so one gather serves both (the natural order, row 4 r + lg,
    // needs every gather twice). The rows return to the natural order on their way through the slab of phase 2.
    // A "unit" u = 2 i + q is the PWU columns of the four pairs 8 i + lg + 4 q, lg = 0..3: 36 pieces of 128 bytes (a column
    // of P, the wave's 16 rows). They arrive by LDS DMA, 16 bytes per lane: eight lanes per piece, eight pieces = 1 KB per
    // instruction, five instructions per unit (the last four pieces of the fifth are repeats) - a CU keeps about 64 vector-
    // memory INSTRUCTIONS in flight whatever their width (the 8-byte register gathers of the first version took 40 cycles
    // apiece with every wave's queue full: 40 k cycles for this phase), so the width is what buys bandwidth. GD units per
    // wave are in flight in the (still unused) slab / factor region of the LDS; the lanes then pick their values with
    // ds_read_b64. Requests and waits are hand-placed: the DMA instructions and the empty asm statements that carry the
    // sums of the unit consumed before keep their order, s_waitcnt vmcnt(n) counts exactly the younger requests.
    constexpr int NU = 2 * NBM, UI = (4 * PWU + 7) / 8, USZ = UI * 128;   // units; DMA instructions per unit (eight pieces each); doubles per unit slot
    const unsigned long long pbase = reinterpret_cast<unsigned long long>(Pio);
    const fu_v4i rs = fu_v4i{(int)(unsigned)pbase, (int)(unsigned)(pbase >> 32), 0x7FFFFFFF, 0x00020000};   // (= buf_rsrc(Pio))
    const unsigned ld8 = (unsigned)g.ldp * 8u;
    double* stg = sm + wave * (GD * USZ);
    // piece e = 8 n + (lane >> 3) of instruction n: pair lg_e = e / 9 of the unit, slot t = e % 9 (e >= 36: piece 35 again)
    int pe[UI];
#pragma unroll
    for (int n = 0; n < UI; ++n) { const int e = min(8 * n + (lane >> 3), 4 * PWU - 1); pe[n] = (e / PWU) * ELL_PIW + e % PWU; }
    const unsigned vrow = (unsigned)(c0 + 2 * (lane & 7)) * 8u;      // rows 2 j, 2 j + 1 of the wave's block
    auto request = [&](auto uc, int tok) {
      constexpr int u = decltype(uc)::value, i = u >> 1, q = u & 1;
      const int ib = i < nb ? i : nb - 1;          // (a block row past the factor re-reads the last one; its sums are never used)
      const unsigned short* pk = pidx + (8 * ib + 4 * q) * ELL_PIW + tok;
#pragma unroll
      for (int n = 0; n < UI; ++n) {
        const char* src = reinterpret_cast<const char*>(Pio) + (size_t)(__umul24((unsigned)pk[pe[n]], ld8) + vrow);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stg + (u % GD) * USZ + n * 128), 16, 0, 0);
      }
    };
#pragma unroll
    for (int i = 0; i < NBM; ++i) X[i] = d4{0.0, 0.0, 0.0, 0.0};
    static_for<(GD < NU ? GD : NU)>([&](auto uc) {
      int tok = 0;
      asm volatile("" : "+v"(tok));
      request(uc, tok);
    });
    static_for<NU>([&](auto uc) {
      constexpr int u = decltype(uc)::value, i = u >> 1, q = u & 1;
      constexpr int younger = ((u + GD - 1 < NU - 1 ? u + GD - 1 : NU - 1) - u) * UI + (u + GD >= NU ? KS : 0);   // requests behind this unit's
      const int ib = i < nb ? i : nb - 1;
      int lgq = lg;
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(lgq) : "n"(younger) : "memory");
      const double* sv = stg + (u % GD) * USZ + lgq * (PWU * 16) + li;
      const d2* pv = reinterpret_cast<const d2*>(ops) + (long)(8 * ib + lgq + 4 * q) * NSLOT + CWU;
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int t = 0; t < PWU; ++t) { const double v = sv[t * 16]; const d2 c = pv[t]; a0 = fma(c[0], v, a0); a1 = fma(c[1], v, a1); }
      asm volatile("" : "+v"(a0), "+v"(a1));       // (the sums are formed HERE: the slot is free for the next request behind this point)
      X[i][2 * q] = a0; X[i][2 * q + 1] = a1;
      if constexpr (u + GD < NU) {
        int tok = 0;
        asm volatile("" : "+v"(tok) : "v"(a0), "v"(a1));
        request(std::integral_constant<int, u + GD>{}, tok);
      }
      if constexpr (u + GD == NU - 1 || (NU <= GD && u == 0)) {       // the last unit is on its way: the common rows behind it (registers)
        unsigned vp = vP; int lgc = lg;
        fu_anchor(vp, lgc);
#pragma unroll
        for (int s = 0; s < KS; ++s) fu_gather(cb[s], __umul24((unsigned)cidx[4 * s + lgc], ld8) + vp, rs);
      }
    });
    {
      int tok = 0;
      static_assert(KS == 3, "three common k-steps");
      fu_wait3<0>(cb[0], cb[1], cb[2], tok);
    }
    // common slots on the matrix pipe: MFMA row li = pair 8 i + pa, row ra of it (the same row order), k = slot 4 s + lg
    const int pa = (li & 3) + 4 * (li >> 3), ra = (li >> 2) & 1;
#pragma unroll
    for (int i = 0; i < NBM; ++i) {
      if (i < nb) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const double av = ops[((long)(8 * i + pa) * NSLOT + 4 * s + lg) * 2 + ra];
          X[i] = mfma(av, cb[s], X[i]);
        }
      }
    }
  }
  FTR(2);
  FTR2(29);
  if (XIVO_FUSED_ABL == 1) {
    double chk = 0.0;
#pragma unroll
    for (int i = 0; i < NBM; ++i) chk += X[i][0] + X[i][1] + X[i][2] + X[i][3];
    if (chk == 12345.678) g.err[(long)filt * g.strideErr + c0 + li] = chk;
    return;
  }

  // ---- 2  S = H (P H^T) + diag(R): the waves park their columns of H P as the slab of ell<S> (row k = state index,
  //         column = measurement row of the pass, XOR swizzle as in ell_tile_kernel) and walk the row pairs over it
  lds_barrier();                                   // every wave is done with its gather staging: the slab takes that LDS
  FTR(9);
  // (the lane coordinates of this phase pass through an empty asm: its LDS addresses depend on nothing but the lane, so the
  //  optimiser computes them at the top of the kernel and the register allocator then spills them across the gather -
  //  twelve dependent scratch reloads in front of the slab stores, 8 k cycles)
  int li2 = li, lg2 = lg;
  asm volatile("" : "+v"(li2), "+v"(lg2));
  for (int x0 = 0; x0 < Mp; x0 += XC) {
#pragma unroll
    for (int i = 0; i < NBM; ++i) {
      if (i < nb && 16 * i >= x0 && 16 * i < x0 + XC) {
        // register r holds row 2 (lg + 4 (r >> 1)) + (r & 1) of the block (phase 1); it comes back as row 4 r + lg - the
        // accumulator layout of the substitutions. A wave reads only what it wrote itself: no barrier in between.
        const int k = c0 + li2;
        double* trow = tile + k * XC;
#pragma unroll
        for (int r = 0; r < 4; ++r) trow[(16 * i - x0 + 2 * (lg2 + 4 * (r >> 1)) + (r & 1)) ^ (k & 15)] = X[i][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r) X[i][r] = trow[(16 * i - x0 + 4 * r + lg2) ^ (k & 15)];
      }
    }
    if (x0 == 0) FTR(10);
    lds_barrier();
    if (x0 == 0) FTR(11);
    // tasks: (row-pair block rb, 16-column block c of this pass) on or below the diagonal of S
    const int cb0 = x0 / 16, ncb = min(NC, nb - cb0);
    const unsigned short* cidx = pidx + pairs * ELL_PIW;
    int ck[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) ck[s] = cidx[4 * s + lg];
    const int pa = (li & 3) + 4 * (li >> 3), ra = (li >> 2) & 1;     // A operand: MFMA row li = pair pa, row ra of it
    for (int task = wave; task < nb * ncb; task += nw) {
      const int rb = task / ncb, c = task - rb * ncb, cb = cb0 + c;
      if (cb > rb) continue;
      const int p0 = 8 * rb;
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int p = p0 + lg + 4 * q;
        const d2* pv = reinterpret_cast<const d2*>(ops) + (long)p * NSLOT + CWU;
        // all slot indices of the pair up front (packed 16-bit: three 8-byte reads): a step then waits for ONE LDS round trip -
        // its gathers - instead of two (index, then gather)
        uint2 iq[ELL_PIW / 4];
        {
          const uint2* pq = reinterpret_cast<const uint2*>(pidx + p * ELL_PIW);
#pragma unroll
          for (int u = 0; u < ELL_PIW / 4; ++u) iq[u] = pq[u];
        }
#pragma unroll
        for (int t0 = 0; t0 < PWU; t0 += 3) {        // three slots per step (all nine at once: 54 registers, spills around the walk)
          d2 v[3]; double gg[3];
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            v[t] = pv[t0 + t];
            const unsigned w = ((t0 + t) & 2) ? iq[(t0 + t) >> 2].y : iq[(t0 + t) >> 2].x;
            const int k = (int)(((t0 + t) & 1) ? (w >> 16) : (w & 0xffffu));
            gg[t] = tile[k * XC + ((16 * c + li) ^ (k & 15))];
          }
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            acc[2 * q] = fma(v[t][0], gg[t], acc[2 * q]);
            acc[2 * q + 1] = fma(v[t][1], gg[t], acc[2 * q + 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double av = ops[((long)(p0 + pa) * NSLOT + 4 * s + lg) * 2 + ra];
        const double bv = tile[ck[s] * XC + ((16 * c + li) ^ (ck[s] & 15))];
        acc = mfma(av, bv, acc);
      }
      // acc[r] = S[m, x]: m = 2 (p0 + lg + 4 (r >> 1)) + (r & 1), x = 16 cb + li
      const int xl = li;
      double* dst = cb == rb ? sD + rb * BLK : sL + (rb * (rb + 1) / 2 + cb) * BLK;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ml = 2 * (lg + 4 * (r >> 1)) + (r & 1);          // row inside the block
        double v = acc[r];
        if (cb == rb && ml == xl) v += sR[16 * rb + ml];
        dst[ml + 17 * xl] = v;
      }
    }
    if (x0 == 0) FTR(12);
    lds_barrier();
  }
  FTR(3);

  // ---- 3  Estimator::MHGating on the diagonal of S (gate_ell_body, from_S): same expressions, same bits
  if (g.gate) {
    const int F = g.F;
    for (int f = tid; f < F; f += NT) {
      const double* d = sD + ((2 * f) >> 4) * BLK;
      const int o = (2 * f) & 15;
      const double s00 = d[o + 17 * o] - sR[2 * f] + g.R;
      const double s10 = d[o + 1 + 17 * o];
      const double s11 = d[o + 1 + 17 * (o + 1)] - sR[2 * f + 1] + g.R;
      sdist[f] = mh_dist_2x2(s00, s10, s11, sInn[2 * f], sInn[2 * f + 1]);
    }
    lds_barrier();
    // (every wave runs the relaxation loop on the distances in LDS and arrives at the same threshold: one barrier fewer
    //  than a single wave publishing it)
    const double th = relax_threshold(sdist, F, g.thresh, g.mult, g.min_inliers, lane);
    for (int f = tid; f < F; f += NT) {
      const bool in = sdist[f] < th;
      g.mask[(long)filt * F + f] = in ? 1 : 0;
      g.dist[(long)filt * F + f] = sdist[f];
      if (!in) {
        sAny = 1;
        sRej[2 * f] = 1; sRej[2 * f + 1] = 1;
        sInn[2 * f] = 0.0; sInn[2 * f + 1] = 0.0;
        innG[2 * f] = 0.0; innG[2 * f + 1] = 0.0;
        dRG[2 * f] = 1.0; dRG[2 * f + 1] = 1.0;
        double* val = g.ell.val + (long)filt * g.ell.stride_val() + (long)f * ELL_W * 2;
        for (int t = 0; t < 2 * ELL_W; ++t) val[t] = 0.0;
      }
    }
    lds_barrier();
    if (sAny) {                                    // (the common case - every candidate passes - ends here)
      if (g.H) {   // dense copies of the stacked rows are alive (xivo_hip_get_H, the dense fallback rows): keep them consistent
        double* H = g.H + (long)filt * g.strideH;
        double* HT = g.HT + (long)filt * g.strideHT;
        for (int f = 0; f < F; ++f) {
          if (!sRej[2 * f]) continue;
          for (int n = tid; n < Np; n += NT) {
            H[2 * f + (long)n * g.ldh] = 0.0; H[2 * f + 1 + (long)n * g.ldh] = 0.0;
            HT[n + (long)(2 * f) * g.ldht] = 0.0; HT[n + (long)(2 * f + 1) * g.ldht] = 0.0;
          }
        }
      }
      // rows / columns of the rejected pairs decoupled in S (0, unit diagonal): one block per wave and trip, four elements per lane
      for (int t = wave; t < nblk + nb; t += nw) {
        int i, k;
        if (t < nblk) { i = 0; while ((i + 1) * (i + 2) / 2 <= t) ++i; k = t - i * (i + 1) / 2; if (i == k) continue; }   // (a diagonal slot of the factor: receives inv(L_kk))
        else { i = k = t - nblk; }
        double* blk = t < nblk ? sL + t * BLK : sD + (t - nblk) * BLK;
        const bool rr = sRej[16 * i + li] != 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = lg + 4 * r;
          if (rr | (sRej[16 * k + c] != 0)) blk[li + 17 * c] = (i == k && li == c) ? 1.0 : 0.0;
        }
      }
      // ... and their right-hand sides zeroed
#pragma unroll
      for (int i = 0; i < NBM; ++i) {
        if (i < nb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (sRej[16 * i + 4 * r + lg]) X[i][r] = 0.0;
        }
      }
      lds_barrier();
    }
  }
  FTR(4);
  if (XIVO_FUSED_ABL == 6) {                                   // stop in front of the factorisation
    double chk = 0.0;
#pragma unroll
    for (int i = 0; i < NBM; ++i) chk += X[i][0] + X[i][1] + X[i][2] + X[i][3];
    if (chk == 12345.678) g.err[(long)filt * g.strideErr + c0 + li] = chk;
    return;
  }

  // ---- 4  S = L L^T in LDS (trsm_lds_kernel.h, CHOL): block row i belongs to wave i. The forward substitution rides along:
  //         step k needs inv(L_kk) and the blocks L_ik below it - complete behind the barrier of column k + 1 - so every wave
  //         runs step j - 1 next to column j's panel while the next owner factors (which defers its own step by one column:
  //         the chain of diagonal blocks is the critical path). Each X[i] still receives its terms in ascending k: same bits.
  auto forward = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const double* Dk = sL + (k * (k + 1) / 2 + k) * BLK;
    d4 t = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) t = mfma(Dk[li + 17 * (4 * s + lg)], X[k][s], t);
    X[k] = t;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = k + 1; i < NBM; ++i) {
        if (i < nb) {
          const double a = sL[(i * (i + 1) / 2 + k) * BLK + li + 17 * (4 * s + lg)];
          X[i] = mfma(-a, t[s], X[i]);
        }
      }
    }
  };
  {
    const int lo = li + 17 * lg;
    // look-ahead: while the owner of column j factors its diagonal block the other waves of the factor are idle - they form
    // the partial sums that do not depend on column j: the panel's sum_{k<j} L_jk L_ik^T, and (next owner) the diagonal
    // update's sum_{k<j} L_{j+1,k} L_{j+1,k}^T. Behind the barrier only the terms of column j itself are left in the chain.
    // The two accumulators receive their terms in the same (ascending k) order as before: same bits.
    d4 pre0 = d4{0.0, 0.0, 0.0, 0.0}, pre1 = d4{0.0, 0.0, 0.0, 0.0};
    // (the block-column loop runs at run time: ONE copy of the straight-line diagonal factorisation in the kernel; the
    //  forward steps behind it need compile-time block indices - dispatched by static_for on the run-time column)
#pragma unroll 1
    for (int j = 0; j < nb; ++j) {
      {
        const int dj = (j * (j + 1) / 2 + j) * BLK;                         // diagonal slot: inv(L_jj)
        FTR2(4 * j);
        d4 accA = d4{0.0, 0.0, 0.0, 0.0}, accB = d4{0.0, 0.0, 0.0, 0.0};
        if (wave == j) {
          __builtin_amdgcn_s_setprio(3);             // the chain of diagonal blocks is the critical path: ahead of the forward steps of the other waves
          const d4 acc0 = pre0, acc1 = pre1;         // terms k < j - 1 formed while column j - 1 was factored, term j - 1 behind its panel step
          d4 x, y;
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = sD[j * BLK + lo + 68 * r] - (acc0[r] + acc1[r]);
          int bad = 0;
          FTR2(4 * j + 1);
#if XIVO_FUSED_DIAG_CHAIN
          factor_invert_diag_chain(x, y, bad, 16 * j, li, lg);
#else
          factor_invert_diag_blocked2<9>(x, y, bad, 16 * j, li, lg);
#endif
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = lg + 4 * r;
            sD[j * BLK + li + 17 * c] = c <= li ? x[r] : 0.0;               // L_jj, upper triangle zero
            sL[dj + c + 17 * li] = y[r];                                    // inv(L_jj)(c, li)
          }
          if (bad && lane == 0 && sBad == 0) sBad = bad;
          __builtin_amdgcn_s_setprio(0);
        } else if (wave > j && wave < nb) {
          const double* Lj = sL + (j * (j + 1) / 2) * BLK + lo;
          const double* Li = sL + (wave * (wave + 1) / 2) * BLK + lo;
          // (block column j - 1 of row j was written behind the previous barrier by wave j: not before this one)
#pragma unroll 1
          for (int k = 0; k < j - 1; ++k) {
            const double a0 = Lj[k * BLK], b0 = Li[k * BLK], a1 = Lj[k * BLK + 68], b1 = Li[k * BLK + 68];
            accA = mfma(a0, b0, accA);
            accB = mfma(a1, b1, accB);
            const double a2 = Lj[k * BLK + 136], b2 = Li[k * BLK + 136], a3 = Lj[k * BLK + 204], b3 = Li[k * BLK + 204];
            accA = mfma(a2, b2, accA);
            accB = mfma(a3, b3, accB);
          }
          if (wave == j + 1) {                       // next owner: its diagonal update over the block columns already final
            pre0 = d4{0.0, 0.0, 0.0, 0.0}; pre1 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
            for (int k = 0; k < j; ++k) {
              const double a0 = Li[k * BLK], a1 = Li[k * BLK + 68], a2 = Li[k * BLK + 136], a3 = Li[k * BLK + 204];
              pre0 = mfma(a0, a0, pre0);
              pre1 = mfma(a1, a1, pre1);
              pre0 = mfma(a2, a2, pre0);
              pre1 = mfma(a3, a3, pre1);
            }
          }
        }
        FTR2(4 * j + 2);
        lds_barrier();
        FTR2(4 * j + 3);
        if (wave > j && wave < nb) {
          if (wave == j + 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
          double* Li = sL + (wave * (wave + 1) / 2) * BLK + lo;
          if (j >= 1) {
            const int k = j - 1;
            const double* Lj = sL + (j * (j + 1) / 2) * BLK + lo;
            const double a0 = Lj[k * BLK], b0 = Li[k * BLK], a1 = Lj[k * BLK + 68], b1 = Li[k * BLK + 68];
            accA = mfma(a0, b0, accA);
            accB = mfma(a1, b1, accB);
            const double a2 = Lj[k * BLK + 136], b2 = Li[k * BLK + 136], a3 = Lj[k * BLK + 204], b3 = Li[k * BLK + 204];
            accA = mfma(a2, b2, accA);
            accB = mfma(a3, b3, accB);
          }
          d4 rhs;
#pragma unroll
          for (int r = 0; r < 4; ++r) rhs[r] = Li[j * BLK + 68 * r] - (accA[r] + accB[r]);
          d4 out = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) out = mfma(sL[dj + li + 17 * (4 * s4 + lg)], rhs[s4], out);
#pragma unroll
          for (int r = 0; r < 4; ++r) Li[j * BLK + 68 * r] = out[r];
          if (wave != j + 1) __builtin_amdgcn_s_setprio(0);     // (the next owner keeps its priority into its factorisation)
          else {
            // the last term of the next diagonal update, L_{j+1,j} L_{j+1,j}^T, straight from the registers it was just formed in
            // (out[s] is element (li, lg + 4 s) of the block: the A and the B operand of k-slice s as it stands) - no LDS round trip
            pre0 = mfma(out[0], out[0], pre0);
            pre1 = mfma(out[1], out[1], pre1);
            pre0 = mfma(out[2], out[2], pre0);
            pre1 = mfma(out[3], out[3], pre1);
          }
        }
        // (a broken factor leaves NaNs at worst: nothing of X is used then)
        if (!XIVO_FUSED_FWD_LATE && XIVO_FUSED_ABL != 8) {
          if (wave == j) static_for<NBM>([&](auto kc) { constexpr int k = decltype(kc)::value; if (k == j - 2) forward(kc); });
          if (wave != j + 1 || j + 1 >= nb) static_for<NBM>([&](auto kc) { constexpr int k = decltype(kc)::value; if (k == j - 1) forward(kc); });
        }
      }
    }
    lds_barrier();
    if (XIVO_FUSED_FWD_LATE) static_for<NBM>([&](auto kc) { constexpr int k = decltype(kc)::value; if (k < nb - 1) forward(kc); });
  }
  const int chol_bad = sBad;
  if (tid == 0) g.status[filt] = chol_bad;
  FTR(5);
  if (XIVO_FUSED_ABL == 2 || XIVO_FUSED_ABL == 8) {
    double chk = 0.0;
#pragma unroll
    for (int i = 0; i < NBM; ++i) chk += X[i][0] + X[i][1] + X[i][2] + X[i][3];
    if (chk == 12345.678) g.err[(long)filt * g.strideErr + c0 + li] = chk;
    return;
  }
  if (chol_bad) {
    // S is not positive definite: the pivoted L D L^T fallback (ldlt_fallback.hip) takes this filter from the prior. What it
    // reads from the pipeline is the gated P H^T - rebuilt here from the coefficients still in LDS (the registers are part-way
    // through the forward substitution); a rare path, plain loops
    double* PHT = g.PHT + (long)filt * g.stridePHT;
    const unsigned short* cidx = pidx + pairs * ELL_PIW;
    for (int e = tid; e < Np * Mp; e += NT) {
      const int n = e % Np, m = e / Np, pp = m >> 1, h = m & 1;
      double acc = 0.0;
      if (!sRej[m]) {
        for (int t = 0; t < CWU; ++t) acc = fma(ops[((long)pp * NSLOT + t) * 2 + h], Pio[n + (long)cidx[t] * g.ldp], acc);
        for (int t = 0; t < PWU; ++t) acc = fma(ops[((long)pp * NSLOT + CWU + t) * 2 + h], Pio[n + (long)pidx[pp * ELL_PIW + t] * g.ldp], acc);
      }
      PHT[n + (long)m * g.ldpht] = acc;
    }
    return;
  }

  // ---- 5  the last forward step, then the backward substitution; W stays in registers next to the working copy
  //         (trsm_lds_kernel.h, T4 + KEEPW)
  static_for<NBM>([&](auto kc) { constexpr int k = decltype(kc)::value; if (k == nb - 1) forward(kc); });
  d4 Wk[NBM];
#pragma unroll
  for (int i = 0; i < NBM; ++i) Wk[i] = X[i];
  FTR(6);
  double part = 0.0;
#pragma unroll
  for (int k = NBM - 1; k >= 0; --k) {
    if (k < nb) {
      const double* Dk = sL + (k * (k + 1) / 2 + k) * BLK;
      d4 t = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) t = mfma(Dk[(4 * s + lg) + 17 * li], X[k][s], t);
      const double* Lk = sD + k * BLK;
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) X[k] = mfma(-Lk[(4 * s2 + lg) + 17 * li], t[s2], X[k]);
#pragma unroll
      for (int r = 0; r < 4; ++r) part = fma(t[r], sInn[16 * k + 4 * r + lg], part);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < k; ++i) {
          const double a = sL[(k * (k + 1) / 2 + i) * BLK + (4 * s + lg) + 17 * li];   // (L_ki)^T
          X[i] = mfma(-a, t[s], X[i]);
        }
      }
      X[k] = Wk[k] - X[k];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  part += __shfl_xor(part, 16);
  part += __shfl_xor(part, 32);
  if (lg == 0) g.err[(long)filt * g.strideErr + c0 + li] = part;
  FTR(7);
  FTR2(30);

  // ---- 6  P+ = P - (W - D)^T (W + D) in place, lower triangle + mirror. Wave w forms the tiles (w, j) for the column blocks j
  //         cyclically below it (every unordered pair once); W + D = 2 W - V of every wave goes through LDS - into the dead
  //         slab when all of it fits there (one phase, written before any barrier), else over the dead factor in two buffers.
  //         The -P tiles of a wave are requested PD tiles ahead (sym_tiles_from_regs requests one: at these sizes a tile's
  //         MFMA chain is a third of a memory round trip).
  {
    const int nwl = Np / 16, jbp = g.jbp;
    const int nph = (nwl + jbp - 1) / jbp, bufsz = jbp * nb * 256;
    const bool early = nph == 1 && bufsz <= Np * XC;
    auto write_y = [&](int p) {
      const int jb0 = p * jbp;
      if (wave >= jb0 && wave < jb0 + min(jbp, nwl - jb0)) {
        double* dst = sm + (p & 1) * bufsz + (wave - jb0) * nb * 256 + lane;
#pragma unroll
        for (int mb = 0; mb < NBM; ++mb) {
          if (mb < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(mb * 4 + r) * 64] = fma(2.0, Wk[mb][r], -X[mb][r]);
          }
        }
      }
    };
    auto my_tiles = [&](int p) -> unsigned {
      unsigned todo = 0;
      if (p >= nph) return todo;
      const int jb0 = p * jbp, nj = min(jbp, nwl - jb0);
      for (int jl = 0; jl < nj; ++jl) {
        int d = wave - (jb0 + jl);
        if (d < 0) d += nwl;
        if (2 * d < nwl || (2 * d == nwl && wave > jb0 + jl)) todo |= 1u << jl;
      }
      return todo;
    };
    constexpr int PD = 4, MAXT = 9;                                   // tiles in flight; tiles of one wave in one phase (<= 16 / 2 + 1)
    const __amdgpu_buffer_rsrc_t rO = buf_rsrc(Pio);
    const unsigned vM = (unsigned)(li + lg * g.ldp) * 8u;
    double* tsc = sm + g.tsc_off + wave * 256;                         // this wave's transpose scratch (16 x 16, XOR swizzle)
    d4 ring[PD];
    int jl_of[MAXT];
    auto list_tiles = [&](unsigned todo) -> int {
      int n = 0;
#pragma unroll
      for (int t = 0; t < MAXT; ++t) { jl_of[t] = todo ? __builtin_ctz(todo) : 0; if (todo) { ++n; todo &= todo - 1; } }
      return n;
    };
    auto request = [&](int jb, d4& dst) {
      const int ba = jb <= wave ? wave : jb, bb = jb <= wave ? jb : wave;       // block (ba, bb), ba >= bb: P's lower triangle
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[r] = XIVO_FUSED_ABL == 5 ? 1.0 : buf_ld_once(rO, vM, (unsigned)(16 * ba + (16 * bb + 4 * r) * g.ldp) * 8u);
    };
    if (!early) __syncthreads();                   // the factor is dead: the LDS takes the operands
    write_y(0);
    if (nph == 1 && (nwl / 2 + 1 == 7 || nwl / 2 + 1 == 6)) {   // the two target shapes (13 / 10 column blocks): hand-counted waits
      double* tscw = sm + g.tsc_off + wave * 256;
      if (XIVO_FUSED_BALANCE && nwl == 10) {         // tiles oriented for 3 + 3 + 2 + 2 waves on the four SIMDs (tables above)
        const int ntw = (wave == 2 || wave == 3 || wave == 6 || wave == 7) ? 7 : ((wave == 0 || wave == 1 || wave == 4) ? 5 : 4);
        if (ntw == 7) fused_product_one_phase<NBM, 7, 10>(X, sm, tscw, Pio, g.ldp, nb, nwl, wave, lane);
        else if (ntw == 5) fused_product_one_phase<NBM, 5, 10>(X, sm, tscw, Pio, g.ldp, nb, nwl, wave, lane);
        else fused_product_one_phase<NBM, 4, 10>(X, sm, tscw, Pio, g.ldp, nb, nwl, wave, lane);
      } else if (XIVO_FUSED_BALANCE && nwl == 13) {  // 4 + 3 + 3 + 3 waves
        const int ntw = (wave == 12) ? 5 : ((wave & 3) == 0 ? 6 : ((wave == 7 || wave >= 9) ? 7 : 8));
        if (ntw == 8) fused_product_one_phase<NBM, 8, 13>(X, sm, tscw, Pio, g.ldp, nb, nwl, wave, lane);
        else if (ntw == 7) fused_product_one_phase<NBM, 7, 13>(X, sm, tscw, Pio, g.ldp, nb, nwl, wave, lane);
        else if (ntw == 6) fused_product_one_phase<NBM, 6, 13>(X, sm, tscw, Pio, g.ldp, nb, nwl, wave, lane);
        else fused_product_one_phase<NBM, 5, 13>(X, sm, tscw, Pio, g.ldp, nb, nwl, wave, lane);
      } else if (nwl / 2 + 1 == 7) fused_product_one_phase<NBM, 7>(X, sm, tscw, Pio, g.ldp, nb, nwl, wave, lane);
      else fused_product_one_phase<NBM, 6>(X, sm, tscw, Pio, g.ldp, nb, nwl, wave, lane);
      FTR(8);
      FTR2(31);
      return;
    }
    int nt = list_tiles(my_tiles(0));
    static_for<PD>([&](auto tc) { constexpr int t = decltype(tc)::value; if (t < nt) request(jl_of[t], ring[t]); });
    for (int p = 0; p < nph; ++p) {
      lds_barrier();                               // the operands of phase p are in place (and the other buffer is free again)
      if (p + 1 < nph) write_y(p + 1);
      const int jb0 = p * jbp;
      const double* buf = sm + (p & 1) * bufsz;
      static_for<MAXT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if (t < nt) {
          const int jl = jl_of[t], jb = jb0 + jl;
          d4 acc = -ring[t % PD];
          asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
          if constexpr (t + PD < MAXT) { if (t + PD < nt) request(jb0 + jl_of[t + PD], ring[t % PD]); }
          const double* Bop = buf + jl * nb * 256 + lane;
          if (jb <= wave) {
#pragma unroll
            for (int mb = 0; mb < NBM; ++mb) {
              if (mb < nb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = mfma(Bop[(mb * 4 + r) * 64], X[mb][r], acc);
              }
            }
          } else {
#pragma unroll
            for (int mb = 0; mb < NBM; ++mb) {
              if (mb < nb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = mfma(X[mb][r], Bop[(mb * 4 + r) * 64], acc);
              }
            }
          }
          // the tile (block (ba, bbk), ba >= bbk: rows li, columns lg + 4 r) and its mirror image. The mirror goes through a
          // 2 KB transpose in LDS so that it, too, leaves with the lanes along its rows - four full 128-byte lines per store
          // instruction; stored straight from the accumulators it is sixteen 32-byte pieces per instruction, and the stores
          // of this phase are what it waits for (back-pressure of the texture path: ~300 cycles per scattered store)
          const int ba = jb <= wave ? wave : jb, bbk = jb <= wave ? jb : wave;
#pragma unroll
          for (int r = 0; r < 4; ++r) tsc[(lg + 4 * r) * 16 + (li ^ (lg + 4 * r))] = -acc[r];      // element (a = li, b = lg + 4 r)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if ((jb != wave || li >= lg + 4 * r) && XIVO_FUSED_ABL != 3)                              // diagonal tile: the lower triangle is authoritative
              buf_st_out(-acc[r], rO, vM, (unsigned)(16 * ba + (16 * bbk + 4 * r) * g.ldp) * 8u);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double v = tsc[li * 16 + ((lg + 4 * r) ^ li)];                                        // element (a = lg + 4 r, b = li)
            if ((jb != wave || lg + 4 * r > li) && XIVO_FUSED_ABL != 3 && XIVO_FUSED_ABL != 4)
              buf_st_out(v, rO, vM, (unsigned)(16 * bbk + (16 * ba + 4 * r) * g.ldp) * 8u);
          }
        }
      });
      if (p + 1 < nph) {
        nt = list_tiles(my_tiles(p + 1));
        const int jn0 = (p + 1) * jbp;
        static_for<PD>([&](auto tc) { constexpr int t = decltype(tc)::value; if (t < nt) request(jn0 + jl_of[t], ring[t]); });
      }
    }
  }
  FTR(8);
  FTR2(31);
}

template <int NBM, int NWV, int XC, int GD, int PW>
int launch_fused_update_g(const FusedArgs& g_in, hipStream_t stream) {
  FusedArgs g = g_in;
  const int nb = g.Mp / 16, nwl = g.Np / 16;
  const FusedLds map = fused_lds_map(g.Np, g.Mp, XC);
  const size_t cap = 160 * 1024;
  size_t lds = (size_t)map.total * sizeof(double);
  if (lds > cap) return (int)hipErrorInvalidValue;
  // product phase: every column block's operand at once when it fits (one phase, one buffer), else two buffers
  const size_t per = (size_t)nb * 256 * sizeof(double), tsc = (size_t)nwl * 256 * sizeof(double);   // + one 16 x 16 transpose scratch per wave
  size_t opbytes;
  if ((size_t)nwl * per + tsc <= cap) { g.jbp = nwl; opbytes = (size_t)nwl * per; }
  else { g.jbp = (int)((cap - tsc) / 2 / per); opbytes = (size_t)2 * g.jbp * per; }
  if (g.jbp < 1) return (int)hipErrorInvalidValue;
  g.tsc_off = (int)(opbytes / sizeof(double));
  if (opbytes + tsc > lds) lds = opbytes + tsc;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_update_f64_kernel<NBM, NWV, XC, GD, PW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
    attr_set = true;
  }
  hipLaunchKernelGGL((fused_update_f64_kernel<NBM, NWV, XC, GD, PW>), dim3(g.batch), dim3(64 * nwl), lds, stream, g);
  return (int)hipGetLastError();
}
// gather staging of phase 1: two units (10 KB) per wave in flight where the slab / factor region of the LDS holds them, else one
static int fused_private_slots(int pw, int XC) { return (pw <= 6 && (XC == 64 || XC == 48)) ? 6 : 9; }
static int fused_gather_depth(int Np, int Mp, int XC, int pws) {
  const int room = fused_lds_map(Np, Mp, XC).ops;   // doubles in front of the coefficients (live during the gather)
  const int usz = (4 * pws + 7) / 8 * 128;
  return (Np / 16) * 2 * usz <= room ? 2 : 1;
}
// (slots, staging depth) of the instantiation launch_fused_update_t picks
static void fused_variant(int Np, int Mp, int XC, int pw, int& pws, int& gd) {
  pws = fused_private_slots(pw, XC);
  if (pws == 6 && fused_gather_depth(Np, Mp, XC, 6) != 2) pws = 9;
  gd = fused_gather_depth(Np, Mp, XC, pws);
}
// (the six-slot form is instantiated where it is used: the widest slab of either factor size with the two-unit staging)
template <int NBM, int NWV, int XC>
int launch_fused_update_t(const FusedArgs& g, hipStream_t stream) {
  int pws, gd;
  fused_variant(g.Np, g.Mp, XC, g.pw, pws, gd);
  if constexpr ((NBM == 4 && XC == 64) || (NBM == 7 && XC == 48)) {
    if (pws == 6) return launch_fused_update_g<NBM, NWV, XC, 2, 6>(g, stream);
  }
  return gd == 2 ? launch_fused_update_g<NBM, NWV, XC, 2, 9>(g, stream) : launch_fused_update_g<NBM, NWV, XC, 1, 9>(g, stream);
}

}  // namespace

// instantiations (NBM, NWV, XC; x two gather depths): <4, 16, 64> M <= 64 on any state one workgroup holds (128 VGPRs; <4, 16, 32>: where the 64-wide slab does not fit next to S); <7, 12, 48> (slab in three passes; <7, 12, 32>: four, where 48 columns do not fit) M <= 112, N <= 192 (168 VGPRs)
static int fused_pick(int Mp, int Np) {
  if (Np % 16 || Mp % 16 || Np < 16 || Mp < 16) return 0;   // (A/B against the multi-kernel pipeline: XIVO_HIP_FLAG_MULTI_KERNEL)
  const int nb = Mp / 16, nwl = Np / 16;
  if (nb <= 4 && nwl <= 16 && (size_t)fused_lds_map(Np, Mp, 64).total * 8 <= 160 * 1024) return 1;
  if (nb <= 4 && nwl <= 16 && (size_t)fused_lds_map(Np, Mp, 32).total * 8 <= 160 * 1024) return 3;
  if (nb <= 7 && nwl <= 12 && (size_t)fused_lds_map(Np, Mp, 48).total * 8 <= 160 * 1024) return 4;
  if (nb <= 7 && nwl <= 12 && (size_t)fused_lds_map(Np, Mp, 32).total * 8 <= 160 * 1024) return 2;
  return 0;
}
#if XIVO_FUSED_TU != 7
bool fused_update_supported(int Mp, int Np) { return fused_pick(Mp, Np) != 0; }
#endif
// Two translation units (the instantiations take minutes to compile): this file holds the M <= 64 kernels and the dispatch,
// fused_update7.hip - which includes this file with XIVO_FUSED_TU = 7 - the M <= 112 kernels. A trace build
// (-DXIVO_FUSED_TRACE=1) keeps everything here: its stamp buffers are device globals of ONE unit.
#if XIVO_FUSED_TU == 7 || XIVO_FUSED_TRACE
int launch_fused_update_k2(const FusedArgs& g, hipStream_t stream) { return launch_fused_update_t<7, 12, 32>(g, stream); }
int launch_fused_update_k4(const FusedArgs& g, hipStream_t stream) { return launch_fused_update_t<7, 12, 48>(g, stream); }
#else
int launch_fused_update_k2(const FusedArgs& g, hipStream_t stream);
int launch_fused_update_k4(const FusedArgs& g, hipStream_t stream);
#endif
#if XIVO_FUSED_TU != 7
int launch_fused_update(const FusedArgs& g, hipStream_t stream) {
  if (g.batch <= 0) return 0;
  switch (fused_pick(g.Mp, g.Np)) {
    case 1: return launch_fused_update_t<4, 16, 64>(g, stream);
    case 2: return launch_fused_update_k2(g, stream);
    case 3: return launch_fused_update_t<4, 16, 32>(g, stream);
    case 4: return launch_fused_update_k4(g, stream);
  }
  return (int)hipErrorInvalidValue;
}
// The orientation tables of the product phase, checked on the host: every unordered block pair {a, b} (a == b: the diagonal tile)
// is formed by exactly one wave, a wave's own diagonal tile comes first, the per-wave counts are the ones the kernel's dispatch
// hard-codes, and no SIMD (wave % 4) forms more than ceil(tiles / 4) of them. Returns 0, or the number of the failed check.
int fused_tiles_selftest(int nwl, int* per_simd /* [4] */) {
  if (nwl != 10 && nwl != 13) return -1;
  int seen[13][13] = {}, cnt[13] = {}, simd[4] = {};
  for (int w = 0; w < nwl; ++w) {
    const signed char* row = nwl == 10 ? kFusedTiles10Host[w] : kFusedTiles13Host[w];
    if (row[0] != w) return 1;
    bool ended = false;
    for (int t = 0; t < 8; ++t) {
      const int jb = row[t];
      if (jb < 0) { ended = true; continue; }
      if (ended || jb >= nwl) return 2;
      const int a = w > jb ? w : jb, b = w > jb ? jb : w;
      if (seen[a][b]++) return 3;
      ++cnt[w]; ++simd[w & 3];
    }
  }
  for (int a = 0; a < nwl; ++a) for (int b = 0; b <= a; ++b) if (seen[a][b] != 1) return 4;
  for (int w = 0; w < nwl; ++w) {
    const int want = nwl == 10 ? ((w == 2 || w == 3 || w == 6 || w == 7) ? 7 : ((w == 0 || w == 1 || w == 4) ? 5 : 4))
                               : ((w == 12) ? 5 : ((w & 3) == 0 ? 6 : ((w == 7 || w >= 9) ? 7 : 8)));
    if (cnt[w] != want) return 5;
  }
  const int tiles = nwl * (nwl + 1) / 2;
  for (int q = 0; q < 4; ++q) { if (simd[q] > (tiles + 3) / 4) return 6; if (per_simd) per_simd[q] = simd[q]; }
  return 0;
}
void fused_update_label(int Mp, int Np, int pw, char* buf, size_t n) {
  const int k = fused_pick(Mp, Np);
  const int xc = (k == 1) ? 64 : (k == 4 ? 48 : 32);
  int pws, gd;
  fused_variant(Np, Mp, xc, pw, pws, gd);
  snprintf(buf, n, "fused_update_f64_kernel<%s,%d,%d,%d>", (k == 1 || k == 3) ? "4,16" : "7,12", xc, gd, pws);
}
#endif

}  // namespace xivo_hip
