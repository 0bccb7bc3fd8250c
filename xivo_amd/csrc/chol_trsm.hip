// Batched gain solve (and, inside it, the covariance update) for the EKF measurement update.
//
// Replaces `K_.transpose() = S_.ldlt().solve(H_ * P_)` and `err_ = K_ * inn_`
// (/root/reference/src/estimator.cpp:1265-1267) once S = L L^T is factored (chol_f64.hip: the reference uses Eigen's
// pivoted LDL^T; S = HPH^T + R is SPD, so an un-pivoted blocked Cholesky is the same linear map up to rounding - tests
// assert 1e-8 on dx, 1e-6 on P).
//
// trsm_f64_kernel : each wave64 owns 16 right-hand-side columns and keeps the
//   whole Mp x 16 solution in accumulator registers. The f64 MFMA C/D layout
//   (row = (lane>>4) + 4*reg, col = lane&15) is exactly the B-operand layout of
//   the four k-slices of the next MFMA, so forward and backward substitution
//   chain with no data movement at all; dx = K*inn falls out as a lane reduce.
// trsm_lds_f64_kernel<NBM, TF> : one workgroup of 16 waves per filter, the whole factor in LDS; beyond the solve it
//   carries, on the gain still in its registers (which is also the A-operand layout),
//     TF = 1  T = K (HP) - P                                   (estimator.cpp:1276-1280, re-associated pipeline)
//     TF = 2  P+ = P - W^T W after the forward substitution    (symmetric form)
//     TF = 3  the whole Joseph update in its expanded form     (estimator.cpp:1276-1287; round-2 default, XIVO_HIP_FLAG_EXPANDED_JOSEPH)
//     TF = 4  the whole Joseph update in its whitened form     P+ = P - (W - D)^T (W + D)   (round 3: the default up to M = 176)
//     TF = 5  the whitened outputs V^T, Y^T for a tiled product outside the kernel (N > 256)
//   through sym_tiles_from_regs: symmetric N x N x M products with one operand in registers and the other arriving
//   in LDS by global_load_lds (or from the owner waves' registers), every unordered pair of 16-row blocks once.
// pnew_reg_f64_kernel : P+ = G K^T - T by the same walk with the rows of G loaded into the registers.
// trsm_stream_f64_kernel<NB, WH> : factors beyond the LDS (M > 176), panel by panel (DMA panels); WH = 1: whitened outputs.
#include <stdlib.h>

#include "mfma_util.h"
#include <stdio.h>

// XIVO_ABL: timing-only ablations of trsm_lds_f64_kernel<., 4> (scripts/ablate_solve.sh builds one library per value; the
// results are WRONG for any value but 0): 1 stop after the substitutions, 2 skip the substitutions, 3 no fix-up pass /
// barrier, 4 no stores of P+, 5 no dx accumulation in the backward loop, 6 no stash write / read-back, 7 no loads of the P
// tiles, 8 no operand DMA, 9 LDS-only barrier at the phase start (no vmcnt drain), 10 two row blocks per phase
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
__device__ __forceinline__ void buf_st_f32(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_st(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bufu2, v), r, voff, soff, 0);
}

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

template <int NBM, int WPE>
__global__ __launch_bounds__(256, WPE) void trsm_f64_kernel(TrsmArgs g) {
  const int chunks = (g.Np + 63) / 64;
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;
  const int filt = (slot / chunks) * 8 + xcd;
  const int chunk = slot % chunks;
  if (filt >= g.batch) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int c0 = chunk * 64 + wave * 16;
  if (c0 >= g.Np) return;
  const int nb = g.Mp / 16;

  const double* __restrict__ LU = g.LU + (long)filt * g.strideLU;
  const double* __restrict__ invD = g.invD + (long)filt * g.strideInvD;
  const double* __restrict__ PHT = g.PHT + (long)filt * g.stridePHT;
  const long ld = g.ldlu;

  d4 X[NBM];
#pragma unroll
  for (int i = 0; i < NBM; ++i) {
    X[i] = d4{0.0, 0.0, 0.0, 0.0};
    if (i < nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) X[i][r] = PHT[(c0 + li) + (long)(16 * i + lg + 4 * r) * g.ldpht];
    }
  }

  // forward: L Y = HP
#pragma unroll
  for (int k = 0; k < NBM; ++k) {
    if (k < nb) {
      d4 t = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) t = mfma(invD[(long)k * 512 + li + (4 * s + lg) * 16], X[k][s], t);
      X[k] = t;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = k + 1; i < NBM; ++i) {
          if (i < nb) {
            const double a = LU[(16 * i + li) + (long)(16 * k + 4 * s + lg) * ld];
            X[i] = mfma(-a, t[s], X[i]);
          }
        }
      }
    }
  }
  // backward: L^T K^T = Y
#pragma unroll
  for (int k = NBM - 1; k >= 0; --k) {
    if (k < nb && !g.fwd_only) {
      d4 t = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) t = mfma(invD[(long)k * 512 + 256 + li + (4 * s + lg) * 16], X[k][s], t);
      X[k] = t;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < k; ++i) {
          const double a = LU[(16 * i + li) + (long)(16 * k + 4 * s + lg) * ld];
          X[i] = mfma(-a, t[s], X[i]);
        }
      }
    }
  }

  // K[(c0+li), m] = X[m-block][..];  dx[c0+li] = sum_m K * inn
  double* __restrict__ K = g.K + (long)filt * g.strideK;
  const double* __restrict__ inn = g.fwd_only ? g.y + (long)filt * g.strideY : g.inn + (long)filt * g.strideInn;
  double part = 0.0;
#pragma unroll
  for (int i = 0; i < NBM; ++i) {
    if (i < nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * i + lg + 4 * r;
        K[(c0 + li) + (long)m * g.ldk] = X[i][r];
        part = fma(X[i][r], inn[m], part);
      }
    }
  }
  part += __shfl_xor(part, 16);
  part += __shfl_xor(part, 32);
  if (lg == 0) g.err[(long)filt * g.strideErr + c0 + li] = part;
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
template <int NBM, bool SRC_REGS = false, bool NEG_OUT = false, bool FIXUP = false, bool YREGS = false>
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
    for (int q = wave; q < (XIVO_ABL == 8 ? 0 : nj * 2 * nb); q += 16) {
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
      acc[r] = XIVO_ABL == 7 ? 1.0 : buf_ld(rM, vM, (unsigned)(16 * ba + (16 * bb + 4 * r) * ldm) * 8u);   // (negated where it is consumed: no wait here)
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
      if (jb <= w) {
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
          buf_st(v, rO, vO, (unsigned)(16 * ba + (16 * bbk + 4 * r) * ldo) * 8u);
          if (a != bb) buf_st(v, rO, vOt, (unsigned)(16 * bbk + 4 * r + 16 * ba * ldo) * 8u);
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
template <int NBM, int TF>
__global__ __launch_bounds__(1024) void trsm_lds_f64_kernel(TrsmArgs g) {
  constexpr int BLK = 16 * 17;
  // TF == 3 needs the diagonal blocks L_kk next to their inverses: in slots of their own while the LDS has room (<= 10
  // block rows), else packed into the unused upper triangle + pad row of the inverse's slot (a few selects per read)
  constexpr bool T4 = TF == 4 || TF == 5;   // whitened Joseph form; TF == 5: its outputs V^T, Y^T for a product outside the kernel
  constexpr bool WOUT = TF == 5;
  // short factors (M <= 96): W stays in registers next to the working copy - no stash, no read-back, no DMA of the operand
  constexpr bool KEEPW = T4 && NBM <= 6;
  constexpr bool PACK = (TF == 3 || T4) && NBM > 10;
  extern __shared__ __attribute__((aligned(16))) double sL[];   // [nb(nb+1)/2][16 x 17]
  const int chunks = (g.Np + 255) / 256;
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
  const int c0 = chunk * 256 + wave * 16;
  const bool live = c0 < g.Np;
  const __amdgpu_buffer_rsrc_t rPHT = buf_rsrc(PHT), rK = buf_rsrc(g.K + (long)filt * g.strideK),
                               rInn = buf_rsrc(g.fwd_only ? g.y + (long)filt * g.strideY : g.inn + (long)filt * g.strideInn);
  const unsigned vPHT = (unsigned)((c0 + li) + lg * g.ldpht) * 8u;   // element (c0 + li, lg) of P H^T; + (16 i + 4 r) ldpht as a scalar offset
  const unsigned vK = (unsigned)((c0 + li) + lg * g.ldk) * 8u;
  d4 X[NBM];
#pragma unroll
  for (int i = 0; i < NBM; ++i) {
    X[i] = d4{0.0, 0.0, 0.0, 0.0};
    if (live && i < nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) X[i][r] = buf_ld(rPHT, vPHT, (unsigned)((16 * i + 4 * r) * g.ldpht) * 8u);
    }
  }
  d4 Wk[KEEPW ? NBM : 1];                           // (KEEPW) the forward-substituted columns, kept next to the working copy
#pragma unroll
  for (int i = 0; i < (KEEPW ? NBM : 1); ++i) Wk[i] = d4{0.0, 0.0, 0.0, 0.0};

  // cooperative copy: block (i,k), i >= k at slot i(i+1)/2 + k. The loads of a thread are requested four at a time before
  // the first of them is consumed (compile-time trip count): the loop used to wait for each of its ~7 round trips in turn,
  // on a CU that has nothing else to run meanwhile. (All of them at once would spill: the right-hand sides are in flight.)
  const int nblk = nb * (nb + 1) / 2;
  constexpr bool DIAG = !PACK && (TF == 3 || T4);                    // the diagonal blocks L_kk in slots of their own
  constexpr int CPY = (NBM * (NBM + 1) / 2 * 128 + 1023) / 1024;     // d2 loads per thread: factor ...
  constexpr int CPD = DIAG ? (NBM * 128 + 1023) / 1024 : 0;         // ... + diagonal blocks
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
        const int e = tid + 1024 * u;
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
        const int e = tid + 1024 * (u - CPY);
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
        const int e = tid + 1024 * u;
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
        const int e = tid + 1024 * (u - CPY);
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
    for (int e = tid; e < nb * 16; e += 1024) {
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
    if (k < nb && !(T4 && XIVO_ABL == 2)) {
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
    if (k < nb && !g.fwd_only && !(T4 && XIVO_ABL == 2)) {
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

  if (TF == 3) {
    // ---- the whole covariance update on the gain in registers (expanded Joseph form, see the launcher's comment):
    //   V^T = L^T K^T (in place, ascending block rows), then L V^T (in place, descending) = (K L L^T)^T,
    //   Z^T = 2 P H^T - K L L^T  [= P H^T - G,  G = K (L L^T) - P H^T the residual of the gain equation]
#pragma unroll
    for (int j = 0; j < NBM; ++j) {
      if (j < nb) {
        const double* Dj = PACK ? sL + (j * (j + 1) / 2 + j) * BLK : sD + j * BLK;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {                                                                      // (L_jj)^T:
          const int kk = 4 * s2 + lg;                                                                         // element (li, kk) = L_jj(kk, li)
          if (PACK) {
            const double a = Dj[kk > li ? li + 17 * kk : 16 + 17 * li];
            acc = mfma(kk >= li ? a : 0.0, X[j][s2], acc);
          } else {
            acc = mfma(Dj[kk + 17 * li], X[j][s2], acc);
          }
        }
#pragma unroll
        for (int i = j + 1; i < NBM; ++i) {
          if (i < nb) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
              acc = mfma(sL[(i * (i + 1) / 2 + j) * BLK + (4 * s2 + lg) + 17 * li], X[i][s2], acc);            // (L_ij)^T
          }
        }
        X[j] = acc;
      }
    }
#pragma unroll
    for (int i = NBM - 1; i >= 0; --i) {
      if (i < nb) {
        const double* Di = PACK ? sL + (i * (i + 1) / 2 + i) * BLK : sD + i * BLK;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {                                                                      // L_ii:
          const int kk = 4 * s2 + lg;                                                                         // element (li, kk), stored transposed when packed
          if (PACK) {
            const double a = Di[li > kk ? kk + 17 * li : 16 + 17 * li];
            acc = mfma(li >= kk ? a : 0.0, X[i][s2], acc);
          } else {
            acc = mfma(Di[li + 17 * kk], X[i][s2], acc);
          }
        }
#pragma unroll
        for (int k = 0; k < i; ++k) {
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2)
            acc = mfma(sL[(i * (i + 1) / 2 + k) * BLK + li + 17 * (4 * s2 + lg)], X[k][s2], acc);              // L_ik
        }
        X[i] = acc;
      }
    }
#pragma unroll
    for (int i = 0; i < NBM; ++i) {
      if (i < nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) X[i][r] = fma(2.0, buf_ld(rPHT, vPHT, (unsigned)((16 * i + 4 * r) * g.ldpht) * 8u), -X[i][r]);
      }
    }
  }
  }
  if (!TF) return;

  XTR(5);
  __syncthreads();                                 // the factor is dead: the LDS takes the operands
  XTR(6);
  if (TF == 3) {
    // ---- P+ = P - Z^T K^T in place: rows of Z^T in registers, blocks of the gain just written arrive through LDS
    if (g.skip_status && g.skip_status[filt] != 0) return;   // S not positive definite: P stays the prior
    double* Pio = g.T + (long)filt * g.strideT;
    sym_tiles_from_regs<NBM, false, true>(X, X, sL, g.K + (long)filt * g.strideK, g.ldk, Pio, g.ldt, Pio, g.ldt, nb, g.Np / 16, g.t_jbp,
                                          live, c0 >> 4, wave, lane);
    return;
  }
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
    if constexpr (KEEPW) sym_tiles_from_regs<NBM, true, true, false, true>(X, Wk, sL, nullptr, 0, Pio, g.ldt, Pio, g.ldt, nb, g.Np / 16, g.t_jbp,
                                                                 live, c0 >> 4, wave, lane);
    else sym_tiles_from_regs<NBM, false, true, true>(X, X, sL, g.K + (long)filt * g.strideK, g.ldk, Pio, g.ldt, Pio, g.ldt, nb, g.Np / 16, g.t_jbp,
                                                     live, c0 >> 4, wave, lane);
    return;
  }
  if (TF == 2) {
    // ---- symmetric form: P+ = P - W^T W in place, W^T = the forward-substituted columns still in registers
    if (g.skip_status && g.skip_status[filt] != 0) return;   // S not positive definite: P stays the prior
    double* Pio = g.T + (long)filt * g.strideT;
    sym_tiles_from_regs<NBM, true, true>(X, X, sL, nullptr, 0, Pio, g.ldt, Pio, g.ldt, nb, g.Np / 16, g.t_jbp, live, c0 >> 4, wave, lane);
    return;
  }
  // ---- T = K (HP) - P
  sym_tiles_from_regs<NBM>(X, X, sL, PHT, g.ldpht, g.Pm + (long)filt * g.stridePm, g.ldpm, g.T + (long)filt * g.strideT, g.ldt,
                           nb, g.Np / 16, g.t_jbp, live, c0 >> 4, wave, lane);
}

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
template <int NBM, int WH>
// (second launch bound = waves per SIMD the register budget is cut for: 8-wave workgroups -> 3 / 2 / 1 workgroups per CU)
__global__ __launch_bounds__(512, NBM <= 4 ? 6 : (NBM <= 8 ? 4 : 2)) void trsm_stream_f64_kernel(TrsmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double sL[];   // [2][NBM + 1][256]
  constexpr int PSZ = (NBM + 1) * 256;
  const int chunks = (g.Np + 127) / 128;
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
  const int c0 = chunk * 128 + wave * 16;
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
    for (int q = wave; q < nj * 2; q += 8) {
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

template <int NBM, int WH>
int launch_trsm_stream_t(const TrsmArgs& g, hipStream_t stream) {
  const int chunks = (g.Np + 127) / 128;
  const int grid = ((g.batch + 7) / 8) * 8 * chunks;
  const size_t lds = (size_t)2 * (NBM + 1) * 256 * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&trsm_stream_f64_kernel<NBM, WH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((trsm_stream_f64_kernel<NBM, WH>), dim3(grid), dim3(512), lds, stream, g);
  return (int)hipGetLastError();
}
// Block-row capacities the streamed kernel is instantiated for. Register allocation of the fully unrolled substitution
// is erratic from one capacity to the next (spilled VGPRs, hipcc 7.2: plain 14: 0, 16: 0, 18: 4801, 19: 6353, 20: 5019,
// 22: 10, 24: 26; whitened 14: 0, 16: 35, 19: 12, 20: 47, 22: 74, 24: 806 - scripts/resource_usage.sh), so each variant
// uses the capacities that compile clean.
static int stream_capacity(int nb, bool whitened) {
  if (whitened && nb <= 8) return nb <= 4 ? 4 : (nb <= 6 ? 6 : 8);   // small factors: two workgroups per CU (XIVO_HIP_SMALL_STREAM)
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
                        : (g.joseph == 2 ? launch_trsm_lds_tf<NBM, 4>(g, stream)
                                         : (g.joseph ? launch_trsm_lds_tf<NBM, 3>(g, stream) : launch_trsm_lds_tf<NBM, 1>(g, stream)));
  }
  return launch_trsm_lds_tf<NBM, 0>(g, stream);
}

template <int NBM>
int launch_trsm_t(const TrsmArgs& g, hipStream_t stream) {
  const int chunks = (g.Np + 63) / 64;
  const int grid = ((g.batch + 7) / 8) * 8 * chunks;
  constexpr int WPE = NBM <= 6 ? 6 : (NBM <= 10 ? 4 : (NBM <= 14 ? 3 : (NBM <= 20 ? 2 : 1)));
  hipLaunchKernelGGL((trsm_f64_kernel<NBM, WPE>), dim3(grid), dim3(256), 0, stream, g);
  return (int)hipGetLastError();
}

}  // namespace

bool pnew_reg_supported(int Mp, int Np) {
  static const bool off = getenv("XIVO_HIP_NO_PNEW_REG") != nullptr;   // A/B knob: the tiled GEMM instead
  return !off && Mp / 16 <= 10 && Np <= 256 && Np % 16 == 0;
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
// with the pipeline's own conditions and XIVO_HIP_FLAG_THROUGHPUT_ROUTE; XIVO_HIP_NO_LATENCY_ROUTE: A/B knob.
bool trsm_latency_route(int Mp, int batch) {
  static const bool off = getenv("XIVO_HIP_NO_LATENCY_ROUTE") != nullptr;
  return !off && Mp / 16 <= 14 && batch <= 64;
}

bool trsm_forms_T(int Mp, int Np) {
  static const bool off = getenv("XIVO_HIP_NO_TRSM_T") != nullptr;   // A/B knob: T as a stand-alone product
  static const bool small_stream = getenv("XIVO_HIP_SMALL_STREAM") != nullptr;   // A/B knob: small factors take the whitened-outputs path
  if (small_stream && Mp / 16 <= 8) return false;
  return !off && Mp / 16 <= 11 && Np <= 256 && Np % 16 == 0;   // the factor fits the LDS and one workgroup covers every column
}

int launch_trsm_f64(const TrsmArgs& g, hipStream_t stream) {
  if (g.batch <= 0) return 0;
  const int nb = g.Mp / 16;
  // A/B knob (round 3 experiment): small factors with the whitened outputs through the streamed kernel - 8-wave workgroups of
  // 128 columns, two or more per CU, instead of one 16-wave workgroup per filter
  static const bool small_stream = getenv("XIVO_HIP_SMALL_STREAM") != nullptr;
  if ((small_stream || g.latency) && g.Yout && !g.fwd_only && nb <= 8) {
    if (nb <= 4) return launch_trsm_stream_t<4, 1>(g, stream);
    if (nb <= 6) return launch_trsm_stream_t<6, 1>(g, stream);
    return launch_trsm_stream_t<8, 1>(g, stream);
  }
  if (g.latency && g.Yout && !g.fwd_only && nb <= 14) return launch_trsm_stream_t<14, 1>(g, stream);
  if (g.out_f32 && g.Yout && !g.fwd_only && nb <= 14) return launch_trsm_stream_t<14, 1>(g, stream);   // (only the streamed kernel writes float outputs)
  // whole factor in LDS (nb(nb+1)/2 blocks of 16x17 doubles) when it fits 160 KiB
  // (a four-block-row instantiation of the whitened form for the TUM-VI build's 30 features spills 1232 VGPRs - hipcc 7.2;
  //  six block rows serve M <= 96)
  if (nb <= 6) return launch_trsm_lds_t<6>(g, stream);
  if (nb <= 10) return launch_trsm_lds_t<10>(g, stream);
  // (eleven block rows with the whitened outputs leaving the kernel: that instantiation spills 700 VGPRs - streamed instead)
  if (nb <= 11 && !(g.Yout && !g.fwd_only)) return launch_trsm_lds_t<11>(g, stream);
  // larger factors: stream the factor through a double-buffered LDS panel
  // A/B knob (per-wave L2 reads instead of the streamed panel). Ignored when the whitened outputs are wanted: only the streamed
  // kernel writes them, and a caller that then forms P - V^T Y from an unwritten Y would be silently wrong.
  static const bool no_stream = getenv("XIVO_HIP_TRSM_NOSTREAM") != nullptr;
  const bool wh = g.Yout && !g.fwd_only;
  if ((!no_stream || wh) && nb <= 24) {
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
  if (nb <= 4) return launch_trsm_t<4>(g, stream);
  if (nb <= 7) return launch_trsm_t<7>(g, stream);
  if (nb <= 10) return launch_trsm_t<10>(g, stream);
  if (nb <= 14) return launch_trsm_t<14>(g, stream);
  if (nb <= 19) return launch_trsm_t<19>(g, stream);
  if (nb <= 24) return launch_trsm_t<24>(g, stream);
  return (int)hipErrorInvalidValue;
}

int launch_fwd_vec(const double* LU, long strideLU, int ldlu, const double* invD, long strideInvD, const double* inn, long strideInn,
                   double* y, long strideY, int Mp, int batch, hipStream_t stream) {
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(fwd_vec_kernel, dim3((batch + 3) / 4), dim3(256), 0, stream, LU, strideLU, ldlu, invD, strideInvD, inn, strideInn,
                     y, strideY, Mp, batch);
  return (int)hipGetLastError();
}

void trsm_kernel_label(int Mp, char* buf, size_t n, int forms_T, bool latency) {
  const int nb = Mp / 16;
  const bool no_stream = getenv("XIVO_HIP_TRSM_NOSTREAM") != nullptr;
  static const bool small_stream = getenv("XIVO_HIP_SMALL_STREAM") != nullptr;
  if (latency && forms_T == 5 && nb <= 14) snprintf(buf, n, "trsm_stream_f64_kernel<%d,1>", stream_capacity(nb, true));
  else if (small_stream && forms_T >= 4 && nb <= 8) snprintf(buf, n, "trsm_stream_f64_kernel<%d,1>", stream_capacity(nb, true));
  else if (nb <= 11) snprintf(buf, n, "trsm_lds_f64_kernel<%d,%d>", nb <= 6 ? 6 : (nb <= 10 ? 10 : 11), forms_T);
  else if (!no_stream || forms_T >= 4) snprintf(buf, n, "trsm_stream_f64_kernel<%d,%d>", stream_capacity(nb, forms_T >= 4), forms_T >= 4 ? 1 : 0);
  else snprintf(buf, n, "trsm_f64_kernel");
}

}  // namespace xivo_hip
