// Symmetric-output variant of the batched fp64 MFMA GEMM (gfx950 / MI355X):
//     S  = HP * H^T + diag(R)            (estimator.cpp:1259-1263)
//     P+ = T * A^T + K diag(R) K^T       (estimator.cpp:1280-1287)
// Only the lower triangle is computed (and mirrored, so the result is exactly
// symmetric). The unit of work is the 16x16 MFMA block, not a rectangular tile:
// the block rows of the triangle are cut into groups of <= 72 blocks, one
// 256-thread workgroup per group, and inside a workgroup the blocks are dealt
// round-robin to the four waves ("slots"), so every wave carries the same number
// of MFMAs whatever the triangle's shape. Each slot reads its own A and B
// fragment from LDS (2 ds_read_b64 per 64-cycle MFMA - LDS bandwidth is nowhere
// near a limit for fp64), which is what makes the free-form assignment possible.
// A workgroup stages only the A rows of its group and the B rows up to its last
// block row, so bytes fetched per MFMA stay below the ~10 B/clk/CU a CU can pull
// from HBM/Infinity Cache - the actual limiter of the rectangular-tile version
// on triangle (half-empty) tiles.
#include <stdlib.h>

#include "common.h"

namespace xivo_hip {

namespace {

constexpr int NS_MAX = 18;    // slots per wave -> up to 72 blocks per workgroup
constexpr int ROWS_MAX = 12;  // block rows per group
constexpr int COLS_MAX = 16;  // block cols (output <= 256)

template <int BK, int NS>
__global__ __launch_bounds__(256, 2) void gemm_sym_f64_kernel(GemmArgs g, SymGroups sg) {
  constexpr int RA_MAX = ROWS_MAX * BK / 32, RB_MAX = COLS_MAX * BK / 32;   // 16-byte loads per thread per panel
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int ng = sg.n;
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;
  const int filt = (slot / ng) * 8 + xcd;
  const int grp = (slot + slot / ng) % ng;   // rotate so unequal groups spread over all CUs
  if (filt >= g.batch) return;
  if (g.skip_status && g.skip_status[filt]) return;
  const int r0 = sg.r0[grp], r1 = sg.r1[grp];
  const int nrows = r1 - r0, ncols = r1;               // in 16-blocks
  const int LDAS = 16 * (nrows + 1 + (nrows & 1));     // LDAS/16 odd -> k-rows alternate bank halves
  const int LDBS = 16 * (ncols + 1 + (ncols & 1));
  double* As = smem;
  double* Bs = smem + BK * LDAS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;

  // slot q of wave w <-> t = w + 4q in the row-major enumeration of the group's blocks
  const int first = r0 * (r0 + 1) / 2;
  const int total = r1 * (r1 + 1) / 2 - first;
  int srow[NS], scol[NS];
  unsigned active = 0;
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const int t = wave + 4 * q;
    int r = r0, base = 0;
    if (t < total) {
      while (base + r + 1 <= t) { base += r + 1; ++r; }
      active |= 1u << q;
    }
    srow[q] = __builtin_amdgcn_readfirstlane(t < total ? r : r0);
    scol[q] = __builtin_amdgcn_readfirstlane(t < total ? t - base : 0);
  }

  active = __builtin_amdgcn_readfirstlane(active);

  d4 acc[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};

  const int steps0 = (g.seg[0].K + BK - 1) / BK;
  const int steps1 = g.nseg > 1 ? (g.seg[1].K + BK - 1) / BK : 0;
  const int nsteps = steps0 + steps1;

  // thread -> (k, pair) of the panels; PW = pairs per k-column. k = idx / PW via an
  // exact reciprocal (idx < BK*PW <= 2048, PW <= 128, 20-bit shift: exact, checked exhaustively).
  const int PWA = 8 * nrows, PWB = 8 * ncols;
  const unsigned invA = ((1u << 20) + PWA - 1) / PWA, invB = ((1u << 20) + PWB - 1) / PWB;
  d2 ra[RA_MAX], rb[RB_MAX];

  auto load_global = [&](int t) {
    const int s = t < steps0 ? 0 : 1;
    const GemmSeg& seg = g.seg[s];
    const int k0 = (s ? t - steps0 : t) * BK;
    const double* Ab = seg.A + (long)filt * seg.strideA + 16 * r0;
    const double* Bb = seg.B + (long)filt * seg.strideB;
#pragma unroll
    for (int r = 0; r < RA_MAX; ++r) {
      const int idx = tid + 256 * r;
      const int k = (int)((idx * invA) >> 20), p = idx - k * PWA;
      d2 v = d2{0.0, 0.0};
      if (idx < BK * PWA && k0 + k < seg.K) v = *reinterpret_cast<const d2*>(Ab + 2 * p + (long)(k0 + k) * seg.lda);
      ra[r] = v;
    }
#pragma unroll
    for (int r = 0; r < RB_MAX; ++r) {
      const int idx = tid + 256 * r;
      const int k = (int)((idx * invB) >> 20), p = idx - k * PWB;
      d2 v = d2{0.0, 0.0};
      if (idx < BK * PWB && k0 + k < seg.K) {
        v = *reinterpret_cast<const d2*>(Bb + 2 * p + (long)(k0 + k) * seg.ldb);
        if (seg.scale) v *= seg.scale[(long)filt * seg.strideScale + k0 + k];
      }
      rb[r] = v;
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int r = 0; r < RA_MAX; ++r) {
      const int idx = tid + 256 * r;
      const int k = (int)((idx * invA) >> 20), p = idx - k * PWA;
      if (idx < BK * PWA) *reinterpret_cast<d2*>(As + k * LDAS + 2 * p) = ra[r];
    }
#pragma unroll
    for (int r = 0; r < RB_MAX; ++r) {
      const int idx = tid + 256 * r;
      const int k = (int)((idx * invB) >> 20), p = idx - k * PWB;
      if (idx < BK * PWB) *reinterpret_cast<d2*>(Bs + k * LDBS + 2 * p) = rb[r];
    }
  };

  if (nsteps > 0) load_global(0);
  for (int t = 0; t < nsteps; ++t) {
    store_lds();
    __syncthreads();
    if (t + 1 < nsteps) load_global(t + 1);
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      const double* ap = As + (4 * s + lg) * LDAS + li;
      const double* bp = Bs + (4 * s + lg) * LDBS + li;
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        // (a branch-free body lets hipcc hoist all 2*NS fragment reads and spill; keep the guard)
        if (active & (1u << q)) {
          const double a = ap[16 * (srow[q] - r0)];
          const double bb = bp[16 * scol[q]];
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(bb, a, acc[q], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // Epilogue: acc[q][r] = C[i = 16 srow + li][j = 16 scol + lg + 4r]; direct store of
  // the lower part, mirrored store through a per-wave LDS transpose (128-byte runs).
  double* Cb = g.C + (long)filt * g.strideC;
  const double* dg = g.diag ? g.diag + (long)filt * g.strideDiag : nullptr;
  double* Tw = smem + wave * (16 * 17);
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    if (!(active & (1u << q))) continue;
    const int I0 = 16 * srow[q], J0 = 16 * scol[q];
    const int i = I0 + li;
    d4 v = acc[q];
    if (g.epilogue == EPI_ADD_DIAG) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (i == J0 + lg + 4 * r) v[r] += dg[i];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = J0 + lg + 4 * r;
      if (i >= j) Cb[i + (long)j * g.ldc] = v[r];
      Tw[(lg + 4 * r) * 17 + li] = v[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double t = Tw[li * 17 + lg + 4 * r];
      const int i2 = I0 + lg + 4 * r, j2 = J0 + li;
      if (i2 > j2) Cb[j2 + (long)i2 * g.ldc] = t;
    }
  }
}

}  // namespace

// One workgroup must hold the whole triangle (<= 72 blocks, i.e. <= 11 block rows): with more
// than one row group the rectangular strip-tile kernel measured faster (P+ at 256: 1.09 vs 1.17 ms).
bool gemm_sym_supported(int Mp) { return Mp >= 16 && Mp <= 176; }

int launch_gemm_sym_f64(const GemmArgs& a, hipStream_t stream) {
  const int nb = a.Mp / 16;
  if (!gemm_sym_supported(a.Mp) || a.Mp != a.Np) return (int)hipErrorInvalidValue;
  // cut block rows into groups of <= 4*NS blocks and <= 2*RA_MAX rows, as evenly as possible
  const int cap = 4 * NS_MAX, total = nb * (nb + 1) / 2;
  int ngroups = (total + cap - 1) / cap;
  SymGroups sg;
  for (;; ++ngroups) {
    if (ngroups > 8) return (int)hipErrorInvalidValue;
    const double target = (double)total / ngroups;
    int r = 0, gi = 0;
    bool ok = true;
    while (r < nb && gi < ngroups) {
      int cnt = 0, r0 = r;
      while (r < nb && cnt + r + 1 <= cap && (r - r0) < ROWS_MAX && (cnt < target - 0.5 * (r + 1) || gi == ngroups - 1)) {
        cnt += r + 1;
        ++r;
      }
      if (r == r0) { ok = false; break; }
      sg.r0[gi] = r0; sg.r1[gi] = r; ++gi;
    }
    if (ok && r == nb) { sg.n = gi; break; }
  }
  // panel depth: 16 when <= 14 slots per wave suffice (252 VGPRs, no spills; 0.258 vs 0.273 ms for the
  // 160x160 S), else 8 (with 18 slots BK = 16 spills)
  int max_blocks0 = 0;
  for (int gi = 0; gi < sg.n; ++gi) {
    const int cnt = sg.r1[gi] * (sg.r1[gi] + 1) / 2 - sg.r0[gi] * (sg.r0[gi] + 1) / 2;
    if (cnt > max_blocks0) max_blocks0 = cnt;
  }
  const int bk = max_blocks0 <= 56 ? 16 : 8;
  int max_lds = 0;
  for (int gi = 0; gi < sg.n; ++gi) {
    const int nr = sg.r1[gi] - sg.r0[gi], nc = sg.r1[gi];
    const int lda = 16 * (nr + 1 + (nr & 1)), ldb = 16 * (nc + 1 + (nc & 1));
    int bytes = bk * (lda + ldb) * (int)sizeof(double);
    if (bytes < 4 * 16 * 17 * 8) bytes = 4 * 16 * 17 * 8;   // epilogue transpose pads
    if (bytes > max_lds) max_lds = bytes;
  }
  const int grid = ((a.batch + 7) / 8) * 8 * sg.n;
  // slots per wave actually needed by the largest group: 14 (<= 56 blocks, e.g. the 160x160 S) frees
  // 32 accumulator VGPRs, which is what lets the BK = 16 panel depth run without spills
  int max_blocks = 0;
  for (int gi = 0; gi < sg.n; ++gi) {
    const int cnt = sg.r1[gi] * (sg.r1[gi] + 1) / 2 - sg.r0[gi] * (sg.r0[gi] + 1) / 2;
    if (cnt > max_blocks) max_blocks = cnt;
  }
  const bool small = max_blocks <= 56;
  if (small && bk == 16) hipLaunchKernelGGL((gemm_sym_f64_kernel<16, 14>), dim3(grid), dim3(256), max_lds, stream, a, sg);
  else if (small) hipLaunchKernelGGL((gemm_sym_f64_kernel<8, 14>), dim3(grid), dim3(256), max_lds, stream, a, sg);
  else if (bk == 16) hipLaunchKernelGGL((gemm_sym_f64_kernel<16, 18>), dim3(grid), dim3(256), max_lds, stream, a, sg);
  else hipLaunchKernelGGL((gemm_sym_f64_kernel<8, 18>), dim3(grid), dim3(256), max_lds, stream, a, sg);
  return (int)hipGetLastError();
}

}  // namespace xivo_hip
