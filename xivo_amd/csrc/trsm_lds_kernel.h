// The one-workgroup-per-filter solve kernel (trsm_lds_f64_kernel) and its symmetric tile walk, shared by chol_trsm.hip (every
// TF variant on a factor that chol_f64.hip left in memory), solve_fused.hip (its ten- / twelve-wave instantiations) and
// fused_update.hip (the buffer helpers). See chol_trsm.hip for the algebra.
#pragma once
#include <stdlib.h>
#include <stdio.h>

#include "mfma_util.h"
#include "chol_device.h"

// XIVO_ABL: timing-only ablations of trsm_lds_f64_kernel<., 4> (scripts/ablate_solve.sh builds one library per value; the
// results are WRONG for any value but 0): 1 stop after the substitutions, 2 skip the substitutions, 3 no fix-up pass /
// barrier, 4 no stores of P+, 5 no dx accumulation in the backward loop, 6 no stash write / read-back, 7 no loads of the P
// tiles, 8 no operand DMA, 9 LDS-only barrier at the phase start (no vmcnt drain), 10 two row blocks per phase,
// 11 = 2 + no MFMA in the product phase: the memory floor of the product's access pattern (round 5)
#ifndef XIVO_ABL
#define XIVO_ABL 0
#endif

// XIVO_TRACE (scripts/build_variant.sh trace "-DXIVO_TRACE=1"; scripts/trace_solve.py reads it back): shader-clock stamps inside
// trsm_lds_f64_kernel<., 4> of every 64th workgroup - wave 0 at the phase boundaries of the kernel (XTR / XTRP), every wave
// inside the product phases (XTR2: behind the fix-up, behind each tile's MFMA chain, behind each tile's stores). Timing
// only, results unchanged; compiled out by default. Where the kernel's time goes: DESIGN.md 3.0.
#ifndef XIVO_TRACE
#define XIVO_TRACE 0
#endif
#if XIVO_TRACE
__device__ unsigned long long xivo_trace_buf[512 * 32];
__device__ unsigned long long xivo_trace2_buf[128 * 16 * 4 * 16];   // [workgroup][wave][phase][slot]
#define XTR(i) do { if (T4 && !WOUT && threadIdx.x == 0 && (blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < 512) \
    xivo_trace_buf[(blockIdx.x >> 6) * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define XTRP(i) do { if (FIXUP && threadIdx.x == 0 && (blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < 512) \
    xivo_trace_buf[(blockIdx.x >> 6) * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define XTR2(ph, slot) do { if (FIXUP && (threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < 128 && (ph) < 4 && (slot) < 16) \
    xivo_trace2_buf[(((blockIdx.x >> 6) * 16 + (threadIdx.x >> 6)) * 4 + (ph)) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int xivo_hip_debug_read_trace(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(xivo_trace_buf), (size_t)n * sizeof(unsigned long long));
}
extern "C" int xivo_hip_debug_read_trace2(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(xivo_trace2_buf), (size_t)n * sizeof(unsigned long long));
}
#else
#define XTR(i) do {} while (0)
#define XTRP(i) do {} while (0)
#define XTR2(ph, slot) do {} while (0)
#endif

namespace xivo_hip {

namespace {

// Buffer addressing for the per-filter matrices of the one-workgroup-per-filter kernels: a 128-bit resource per matrix in
// SGPRs, ONE 32-bit per-lane byte offset that every access of that matrix shares, and the block / column part of the
// address as a wave-uniform scalar offset - instead of a 64-bit address pair per access in VGPRs (the solve kernel lives
// on exactly 128 VGPRs) and 64-bit vector arithmetic in the MFMA stream.
typedef unsigned int bufu2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
// (loads the kernel reads exactly once - right-hand sides, the tiles of P: a non-temporal policy measured neutral, round 5)
__device__ __forceinline__ double buf_ld_once(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st_f32(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_st(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bufu2, v), r, voff, soff, 0);
}
// (stores of P+ in the product phase: default cache policy - nt / sc1 measured 4-11 % slower, round 5: the 32-byte pieces of the
//  mirror stores merge into full lines in the L2 under the default policy)
__device__ __forceinline__ void buf_st_out(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bufu2, v), r, voff, soff, 0);
}

template <int NBM, bool SRC_REGS = false, bool NEG_OUT = false, bool FIXUP = false, bool YREGS = false, int NWV = 16>
__device__ __forceinline__ void sym_tiles_from_regs(const d4 (&X)[NBM], const d4 (&Wr)[NBM], double* sL, const double* __restrict__ Src, int ldsrc,
                                                    const double* __restrict__ Minit, int ldm, double* __restrict__ Out, int ldo,
                                                    int nb, int nwl, int jbp, bool live, int w, int wave, int lane) {
  const int li = lane & 15, lg = lane >> 4;
  const int nph = (nwl + jbp - 1) / jbp;
  const int bufsz = jbp * nb * 256;                // doubles per LDS buffer (two of them)
  // Operands of phase p straight from global memory into LDS (no registers, asynchronous): one instruction moves
  // 8 columns m x 16 rows j of Src (16 bytes per lane) to 128 consecutive doubles, so that block (jl, m, j) sits at
  // jl nb 256 + 16 m + j - for the MFMA step (mb, r) that is (4 mb + r) 64 + lane: lane-contiguous reads.
  auto issue = [&](int p) {
    const int jb0 = p * jbp, nj = min(jbp, nwl - jb0);
    double* buf = sL + (p & 1) * bufsz;
    if (SRC_REGS) {
      if (live && w >= jb0 && w < jb0 + nj) {
        double* dst = buf + (w - jb0) * nb * 256 + lane;
#pragma unroll
        for (int mb = 0; mb < NBM; ++mb) {
          if (mb < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(mb * 4 + r) * 64] = YREGS ? fma(2.0, Wr[mb][r], -X[mb][r]) : X[mb][r];
          }
        }
      }
      return;
    }
    for (int q = wave; q < (XIVO_ABL == 8 ? 0 : nj * 2 * nb); q += NWV) {
      const int jl = q / (2 * nb), t = q - jl * 2 * nb;
      const double* src = Src + (16 * (jb0 + jl) + 2 * (lane & 7)) + (long)(8 * t + (lane >> 3)) * ldsrc;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + jl * nb * 256 + t * 128), 16, 0, 0);
    }
  };
  auto my_tiles = [&](int p) -> unsigned {
    unsigned todo = 0;
    if (!live || p >= nph) return todo;
    const int jb0 = p * jbp, nj = min(jbp, nwl - jb0);
    for (int jl = 0; jl < nj; ++jl) {
      int d = w - (jb0 + jl);
      if (d < 0) d += nwl;
      if (2 * d < nwl || (2 * d == nwl && w > jb0 + jl)) todo |= 1u << jl;
    }
    return todo;
  };
  // A tile is formed in the orientation that makes its lower-triangle position (a, b), a >= b, lane-contiguous in a:
  // blocks below the diagonal (and the diagonal one) swap the two MFMA operands - the tile comes out transposed,
  // lanes along the row index - blocks above it stand for their mirror image. Minit is read there (its lower
  // triangle, as the stand-alone product does), Out(a, b) and its mirror Out(b, a) are written.
  const __amdgpu_buffer_rsrc_t rM = buf_rsrc(Minit), rO = buf_rsrc(Out);
  const unsigned vM = (unsigned)(li + lg * ldm) * 8u;              // element (li, lg) of a 16 x 16 block of Minit
  const unsigned vO = (unsigned)(li + lg * ldo) * 8u, vOt = (unsigned)(lg + li * ldo) * 8u;   // ... of Out, and of its mirror image
  auto load_m = [&](int jb, d4& acc) {
    const int ba = jb <= w ? w : jb, bb = jb <= w ? jb : w;       // block (ba, bb), ba >= bb
#pragma unroll
    for (int r = 0; r < 4; ++r)
      acc[r] = XIVO_ABL == 7 ? 1.0 : buf_ld_once(rM, vM, (unsigned)(16 * ba + (16 * bb + 4 * r) * ldm) * 8u);   // (negated where it is consumed: no wait here)
  };
  issue(0);
  unsigned todo = my_tiles(0);
  d4 nxt = d4{0.0, 0.0, 0.0, 0.0};
  if (todo) load_m(__builtin_ctz(todo), nxt);
  for (int p = 0; p < nph; ++p) {
    XTRP(8 + 3 * p);
    if (XIVO_ABL == 9) lds_barrier();
    else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XTRP(9 + 3 * p);
    __syncthreads();                               // phase p landed for every wave; the other buffer is free again
    }
    XTRP(10 + 3 * p);
    XTR2(p, 0);
    int tslot = 2;
    const int jb0 = p * jbp;
    const double* buf = sL + (p & 1) * bufsz;
    if (FIXUP && XIVO_ABL != 3) {
      if (live && w >= jb0 && w < jb0 + min(jbp, nwl - jb0)) {
        double* dst = sL + (p & 1) * bufsz + (w - jb0) * nb * 256 + lane;
#pragma unroll
        for (int mb = 0; mb < NBM; ++mb) {
          if (mb < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(mb * 4 + r) * 64] = fma(2.0, dst[(mb * 4 + r) * 64], -X[mb][r]);
          }
        }
      }
      lds_barrier();
    }
    XTR2(p, 1);
    bool fetch = p + 1 < nph;                      // phase p + 1 is requested once the first tile has its -Minit (so that
    while (todo) {                                 // the wait on those loads does not sit behind the new requests)
      const int jl = __builtin_ctz(todo);
      todo &= todo - 1;
      const int jb = jb0 + jl;
      d4 acc = -nxt;
      if (fetch) { asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])); issue(p + 1); fetch = false; }
      if (todo) load_m(jb0 + __builtin_ctz(todo), nxt);
      const double* Bop = buf + jl * nb * 256 + lane;
      if (XIVO_ABL == 11) {   // the product phase's memory traffic alone: tile loads, operand DMA, stores - no MFMA (one LDS read per tile)
        acc[0] += 0.0 * Bop[0];   // (P+ = P: every step of the timing loop sees the same, factorisable, covariance)
      } else if (jb <= w) {
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
      XTR2(p, tslot); ++tslot;
      const int ba = jb <= w ? w : jb, bbk = jb <= w ? jb : w;
      const int a = 16 * ba + li, b = 16 * bbk + lg;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int bb = b + 4 * r;
        if ((jb != w || a >= bb) && !(XIVO_ABL == 4 && acc[r] != 12345.678)) {   // diagonal tile: the lower triangle is authoritative
          const double v = NEG_OUT ? -acc[r] : acc[r];
          buf_st_out(v, rO, vO, (unsigned)(16 * ba + (16 * bbk + 4 * r) * ldo) * 8u);
          if (a != bb) buf_st_out(v, rO, vOt, (unsigned)(16 * bbk + 4 * r + 16 * ba * ldo) * 8u);
        }
      }
      XTR2(p, tslot); ++tslot;
    }
    XTR2(p, 14);
    if (fetch) issue(p + 1);
    todo = my_tiles(p + 1);
    if (todo) load_m((p + 1) * jbp + __builtin_ctz(todo), nxt);
  }
  XTRP(8 + 3 * nph);
}

// TF: the workgroup goes on to form T = K (HP) - P (estimator.cpp:1280, the left product distributed over
// the H P already at hand) while K^T sits in its registers in exactly the A-operand layout of the MFMA: wave w
// owns state rows 16w..16w+15 of K. Once the factor is dead the LDS takes the B operands - P H^T again, 8
// column blocks at a time, each block stored as the 4 nb registers a wave would hold of it, lane-contiguous
// (conflict-free ds_read_b64) - and wave w forms the 16 x 16 tiles (w, j) for the j cyclically below it
// (every unordered pair of blocks once: 8 or 9 tiles per wave), accumulators starting at -P, and writes each
// tile and its mirror. K is never read back and H P is read once more instead of 1.5 times by the tiled GEMM.
// NWV / MINB (round 5): waves per workgroup and workgroups per CU the register budget is cut for. 16 / 1 is the kernel of
// rounds 1-4 (one workgroup owns the CU). A state of at most 16 NWV columns with a short factor leaves room for MORE THAN ONE
// workgroup per CU (BASELINE config 2: N = 150 -> ten waves, seven block rows: 76 KB of LDS, 96 VGPRs): the memory phases
// of one filter (right-hand sides in, covariance tiles in and out) then run under the matrix phases of another - the
// overlap a single workgroup cannot have, because one filter's working set fills the CU at the metric point.
template <int NBM, int TF, int NWV = 16, int MINB = 1>
__global__ __launch_bounds__(64 * NWV, MINB) void trsm_lds_f64_kernel(TrsmArgs g) {
  constexpr int BLK = 16 * 17;
  constexpr int NT = 64 * NWV;
  static_assert(TF != 3, "TF == 3 (the round-2 expanded Joseph form) was removed in round 6");
  // the whitened form needs the diagonal blocks L_kk next to their inverses: in slots of their own while the LDS has room (<= 10
  // block rows), else packed into the unused upper triangle + pad row of the inverse's slot (a few selects per read)
  constexpr bool T4 = TF == 4 || TF == 5;   // whitened Joseph form; TF == 5: its outputs V^T, Y^T for a product outside the kernel
  constexpr bool WOUT = TF == 5;
  // short factors (M <= 96): W stays in registers next to the working copy - no stash, no read-back, no DMA of the operand
  // (also seven block rows on a ten- or twelve-wave workgroup that has the CU to itself: three waves per SIMD, 170 VGPRs
  //  each - BASELINE config 2)
  constexpr bool KEEPW = T4 && (NBM <= 6 || (NBM == 7 && NWV <= 12 && MINB <= 3));
  constexpr bool PACK = T4 && NBM > 10;
  extern __shared__ __attribute__((aligned(16))) double sL[];   // [nb(nb+1)/2][16 x 17]
  const int chunks = (g.Np + 16 * NWV - 1) / (16 * NWV);
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
  const double* __restrict__ PHT = g.PHT + (long)filt * g.stridePHT;
  const long ld = g.ldlu;

  XTR(0);
  // the right-hand sides first: their loads are in flight while the factor is copied (one workgroup per CU - nothing else
  // would hide the latency of either)
  const int c0 = chunk * 16 * NWV + wave * 16;
  const bool live = c0 < g.Np;
  const __amdgpu_buffer_rsrc_t rPHT = buf_rsrc(PHT), rK = buf_rsrc(g.K + (long)filt * g.strideK),
                               rInn = buf_rsrc(g.fwd_only ? g.y + (long)filt * g.strideY : g.inn + (long)filt * g.strideInn);
  const unsigned vPHT = (unsigned)((c0 + li) + lg * g.ldpht) * 8u;   // element (c0 + li, lg) of P H^T; + (16 i + 4 r) ldpht as a scalar offset
  const unsigned vK = (unsigned)((c0 + li) + lg * g.ldk) * 8u;
  d4 X[NBM];
  auto load_rhs = [&](int i0, int i1) {
#pragma unroll
    for (int i = 0; i < NBM; ++i) {
      if (i >= i0 && i < i1) {
        X[i] = d4{0.0, 0.0, 0.0, 0.0};
        if (live && i < nb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) X[i][r] = buf_ld_once(rPHT, vPHT, (unsigned)((16 * i + 4 * r) * g.ldpht) * 8u);
        }
      }
    }
  };
  load_rhs(0, NBM);
  d4 Wk[KEEPW ? NBM : 1];                           // (KEEPW) the forward-substituted columns, kept next to the working copy
#pragma unroll
  for (int i = 0; i < (KEEPW ? NBM : 1); ++i) Wk[i] = d4{0.0, 0.0, 0.0, 0.0};

  // cooperative copy: block (i,k), i >= k at slot i(i+1)/2 + k. The loads of a thread are requested four at a time before
  // the first of them is consumed (compile-time trip count): the loop used to wait for each of its ~7 round trips in turn,
  // on a CU that has nothing else to run meanwhile. (All of them at once would spill: the right-hand sides are in flight.)
  const int nblk = nb * (nb + 1) / 2;
  constexpr bool DIAG = !PACK && T4;                                  // the diagonal blocks L_kk in slots of their own
  constexpr int CPY = (NBM * (NBM + 1) / 2 * 128 + NT - 1) / NT;     // d2 loads per thread: factor ...
  constexpr int CPD = DIAG ? (NBM * 128 + NT - 1) / NT : 0;         // ... + diagonal blocks
  constexpr int CPB = 4;                                              // loads in flight per thread
  double* sD = sL + nblk * BLK;                                       // (upper triangle zeroed)
#pragma unroll
  for (int u0 = 0; u0 < CPY + CPD; u0 += CPB) {
    d2 cv[CPB], cu[CPB];
#pragma unroll
    for (int q = 0; q < CPB; ++q) {
      const int u = u0 + q;
      cv[q] = d2{0.0, 0.0}; cu[q] = d2{0.0, 0.0};
      if (u < CPY) {
        const int e = tid + NT * u;
        if (e < nblk * 128) {
          const int t = e >> 7, w = e & 127;
          int i = 0;
          while ((i + 1) * (i + 2) / 2 <= t) ++i;            // block row of slot t (nb <= 24: a few iterations)
          const int k = t - i * (i + 1) / 2;
          const int r = (w & 7) * 2, c = w >> 3;             // rows r, r+1 of column c
          cv[q] = *reinterpret_cast<const d2*>(i != k ? LU + (16 * i + r) + (long)(16 * k + c) * ld : invD + (long)k * 512 + r + 16 * c);
          if (PACK && i == k) { cu[q][0] = LU[(16 * k + c) + (long)(16 * k + r) * ld]; cu[q][1] = LU[(16 * k + c) + (long)(16 * k + r + 1) * ld]; }
        }
      } else if (u < CPY + CPD) {
        const int e = tid + NT * (u - CPY);
        if (e < nb * 128) {
          const int k = e >> 7, w = e & 127;
          cv[q] = *reinterpret_cast<const d2*>(LU + (16 * k + (w & 7) * 2) + (long)(16 * k + (w >> 3)) * ld);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < CPB; ++q) {
      const int u = u0 + q;
      if (u < CPY) {
        const int e = tid + NT * u;
        if (e < nblk * 128) {
          const int t = e >> 7, w = e & 127;
          int i = 0;
          while ((i + 1) * (i + 2) / 2 <= t) ++i;
          const int k = t - i * (i + 1) / 2;
          const int r = (w & 7) * 2, c = w >> 3;
          d2 v = cv[q];
          // diagonal slot: inv(L_kk) in the lower triangle; packed: L_kk^T above it (the strictly lower part of L_kk read
          // transposed: the factorisation need not have mirrored it) and the diagonal of L_kk in the pad row
          if (PACK && i == k) { v[0] = r >= c ? v[0] : cu[q][0]; v[1] = r + 1 >= c ? v[1] : cu[q][1]; }
          sL[t * BLK + r + 17 * c] = v[0];
          sL[t * BLK + r + 1 + 17 * c] = v[1];
        }
      } else if (u < CPY + CPD) {
        const int e = tid + NT * (u - CPY);
        if (e < nb * 128) {
          const int k = e >> 7, w = e & 127;
          const int r = (w & 7) * 2, c = w >> 3;
          sD[k * BLK + r + 17 * c] = r >= c ? cv[q][0] : 0.0;
          sD[k * BLK + r + 1 + 17 * c] = r + 1 >= c ? cv[q][1] : 0.0;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (PACK) {
    for (int e = tid; e < nb * 16; e += NT) {
      const int k = e >> 4, c = e & 15;
      sL[(k * (k + 1) / 2 + k) * BLK + 16 + 17 * c] = LU[(16 * k + c) + (long)(16 * k + c) * ld];
    }
  }
  XTR(1);
  __syncthreads();
  XTR(2);
  if (!TF && !live) return;
  if (live) {
  // forward: L Y = HP
#pragma unroll
  for (int k = 0; k < NBM; ++k) {
    if (k < nb && !(T4 && (XIVO_ABL == 2 || XIVO_ABL == 11))) {
      const double* Dk = sL + (k * (k + 1) / 2 + k) * BLK;
      d4 t = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        double a = Dk[li + 17 * (4 * s + lg)];
        if (PACK) a = li >= 4 * s + lg ? a : 0.0;         // (the slot's upper triangle belongs to L_kk^T)
        t = mfma(a, X[k][s], t);
      }
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
    }
  }
  XTR(3);
  double part = 0.0;
  if (KEEPW) {
#pragma unroll
    for (int i = 0; i < NBM; ++i) Wk[i] = X[i];
  }
  if (T4 && !KEEPW && XIVO_ABL != 6) {
    // the forward-substituted columns W^T = (L^-1 H P)^T leave for the stash (the K buffer: the gain itself is never
    // stored by this variant) - the backward substitution below destroys them and the covariance update needs them again
#pragma unroll
    for (int i = 0; i < NBM; ++i) {
      if (i < nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) buf_st(X[i][r], rK, vK, (unsigned)((16 * i + 4 * r) * g.ldk) * 8u);
      }
    }
  }
  XTR(4);
  // backward: L^T K^T = Y
#pragma unroll
  for (int k = NBM - 1; k >= 0; --k) {
    if (k < nb && !g.fwd_only && !(T4 && (XIVO_ABL == 2 || XIVO_ABL == 11))) {
      const double* Dk = sL + (k * (k + 1) / 2 + k) * BLK;
      // TF == 4: W_k comes back from the stash while this step's MFMAs run (requested here, used at the end of the step;
      // the last block row has not been touched yet: it is still in X)
      d4 wk = d4{0.0, 0.0, 0.0, 0.0};
      if (KEEPW) wk = Wk[k];
      else if (T4) {
        if (k == nb - 1) wk = X[k];
        else if (XIVO_ABL != 6) {
#pragma unroll
          for (int r = 0; r < 4; ++r) wk[r] = buf_ld(rK, vK, (unsigned)((16 * k + 4 * r) * g.ldk) * 8u);
        }
      }
      d4 t = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        double a = Dk[(4 * s + lg) + 17 * li];
        if (PACK) a = 4 * s + lg >= li ? a : 0.0;
        t = mfma(a, X[k][s], t);
      }
      if (T4) {
        // D_k = B_k - L_kk^T K^T_k, B_k = W_k - sum_{i>k} L_ik^T K^T_i the right-hand side this step just consumed: the
        // residual of the backward substitution, i.e. W_k - (L^T K^T)_k evaluated with the partial sums already at hand
        // (both evaluations of L^T K^T carry the same rounding bound); it replaces the gain block, whose last uses -
        // the updates of the rows above and dx - are right here
        const double* Lk = PACK ? Dk : sD + k * BLK;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          const int kk = 4 * s2 + lg;                                        // (L_kk)^T element (li, kk) = L_kk(kk, li)
          if (PACK) {
            const double a = Lk[kk > li ? li + 17 * kk : 16 + 17 * li];
            X[k] = mfma(kk >= li ? -a : 0.0, t[s2], X[k]);
          } else {
            X[k] = mfma(-Lk[kk + 17 * li], t[s2], X[k]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) part = fma(t[r], XIVO_ABL == 5 ? 1.0 : buf_ld(rInn, (unsigned)lg * 8u, (unsigned)(16 * k + 4 * r) * 8u), part);
      } else {
        X[k] = t;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < k; ++i) {
          const double a = sL[(k * (k + 1) / 2 + i) * BLK + (4 * s + lg) + 17 * li];   // (L_ki)^T
          X[i] = mfma(-a, t[s], X[i]);
        }
      }
      // TF == 4: V_k = W_k - D_k, the row block of V^T = (W - D)^T - the register operand of the covariance product below
      // (or, for states wider than one workgroup, of the tiled product outside: then Y_k = W_k + D_k leaves for g.Yout here)
      if (T4) {
        if (WOUT) {
          const __amdgpu_buffer_rsrc_t rY = buf_rsrc(g.Yout + (long)filt * g.strideY2);
#pragma unroll
          for (int r = 0; r < 4; ++r) buf_st(wk[r] + X[k][r], rY, (unsigned)((c0 + li) + lg * g.ldy2) * 8u, (unsigned)((16 * k + 4 * r) * g.ldy2) * 8u);
        }
        X[k] = wk - X[k];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  if (!T4) {
#pragma unroll
    for (int i = 0; i < NBM; ++i) {
      if (i < nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (TF != 2) buf_st(X[i][r], rK, vK, (unsigned)((16 * i + 4 * r) * g.ldk) * 8u);   // (the symmetric form keeps W^T in registers: nothing reads it back)
          part = fma(X[i][r], buf_ld(rInn, (unsigned)lg * 8u, (unsigned)(16 * i + 4 * r) * 8u), part);
        }
      }
    }
  }
  part += __shfl_xor(part, 16);
  part += __shfl_xor(part, 32);
  if (lg == 0) g.err[(long)filt * g.strideErr + c0 + li] = part;

  // TF == 4: X = V now. The covariance update is the Joseph expression for the gain just computed, in the whitened
  // coordinates of the factor (S = L L^T, H P = L W, V = L^T K^T = W - D):
  //   P+ = P - K(HP) - (K(HP))^T + K S K^T = P - V^T W - W^T V + V^T V = P - (W - D)^T (W + D) + (W^T D - D^T W),
  // whose antisymmetric last term drops out of the lower-triangle + mirror evaluation every pipeline here uses. The rows
  // of V^T = (W - D)^T are the register operand, W + D = 2 W - V the LDS operand (below).

  }
  if (!TF) return;

  XTR(5);
  __syncthreads();                                 // the factor is dead: the LDS takes the operands
  XTR(6);
  if (T4) {
    if (WOUT) {   // whitened outputs only: V^T replaces the stash, the covariance product runs outside (tiled)
      if (live) {
#pragma unroll
        for (int i = 0; i < NBM; ++i) {
          if (i < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) buf_st(X[i][r], rK, vK, (unsigned)((16 * i + 4 * r) * g.ldk) * 8u);
          }
        }
      }
      return;
    }
    // ---- P+ = P - (W - D)^T (W + D) in place: W arrives from the stash by DMA, the owner waves turn it into W + D
    if (g.skip_status && g.skip_status[filt] != 0) return;   // S not positive definite: P stays the prior
    if (XIVO_ABL == 1) return;
    double* Pio = g.T + (long)filt * g.strideT;
    if constexpr (KEEPW) sym_tiles_from_regs<NBM, true, true, false, true, NWV>(X, Wk, sL, nullptr, 0, Pio, g.ldt, Pio, g.ldt, nb, g.Np / 16, g.t_jbp,
                                                                 live, c0 >> 4, wave, lane);
    else sym_tiles_from_regs<NBM, false, true, true, false, NWV>(X, X, sL, g.K + (long)filt * g.strideK, g.ldk, Pio, g.ldt, Pio, g.ldt, nb, g.Np / 16, g.t_jbp,
                                                     live, c0 >> 4, wave, lane);
    return;
  }
  if (TF == 2) {
    // ---- symmetric form: P+ = P - W^T W in place, W^T = the forward-substituted columns still in registers
    if (g.skip_status && g.skip_status[filt] != 0) return;   // S not positive definite: P stays the prior
    double* Pio = g.T + (long)filt * g.strideT;
    sym_tiles_from_regs<NBM, true, true, false, false, NWV>(X, X, sL, nullptr, 0, Pio, g.ldt, Pio, g.ldt, nb, g.Np / 16, g.t_jbp, live, c0 >> 4, wave, lane);
    return;
  }
  // ---- T = K (HP) - P
  sym_tiles_from_regs<NBM, false, false, false, false, NWV>(X, X, sL, PHT, g.ldpht, g.Pm + (long)filt * g.stridePm, g.ldpm, g.T + (long)filt * g.strideT, g.ldt,
                           nb, g.Np / 16, g.t_jbp, live, c0 >> 4, wave, lane);
}

}  // namespace

}  // namespace xivo_hip
