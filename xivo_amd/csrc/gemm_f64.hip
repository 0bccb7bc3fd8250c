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
#include "common.h"

namespace xivo_hip {

namespace {

template <int WM, int WN, int BK, bool STRIP>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int filt, const int m0, const int n0,
                                          double* smem) {
  static_assert(!STRIP || (WM == 4 && WN == 4), "strip mapping is defined for 128x128 tiles");
  constexpr int BM = 32 * WM, BN = 32 * WN;
  constexpr int LDAS = BM + 16, LDBS = BN + 16;
  constexpr int NS = WM * WN;                 // accumulator slots per wave
  constexpr int NA = STRIP ? 2 : WM;          // A fragments per k-slice
  constexpr int NB = STRIP ? 2 * WN : WN;     // B fragments per k-slice
  double* As = smem;
  double* Bs = smem + BK * LDAS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
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

  // Accumulators start at 0, or at -/+Msub for C = acc -/+ Msub (T = K(HP) - P): the
  // loads are issued here, ahead of the first k-panel, instead of serialising
  // behind the stores of the epilogue.
  d4 acc[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    acc[q] = d4{0.0, 0.0, 0.0, 0.0};
    if ((g.epilogue == EPI_SUB_MAT || g.epilogue == EPI_ADD_MAT) && (active & (1u << q))) {
      const double* Ms = g.Msub + (long)filt * g.strideMsub;
      const int i = m0 + 16 * arow(slot_a(q)) + li, J0 = n0 + 16 * bcol(slot_b(q));
      const double sgn = g.epilogue == EPI_SUB_MAT ? -1.0 : 1.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] = sgn * Ms[i + (long)(J0 + lg + 4 * r) * g.ldmsub];
    }
  }

  const int steps0 = (g.seg[0].K + BK - 1) / BK;
  const int steps1 = g.nseg > 1 ? (g.seg[1].K + BK - 1) / BK : 0;
  const int nsteps = steps0 + steps1;

  constexpr int RA = WM * BK / 16, RB = WN * BK / 16;   // 16-byte loads per thread per panel
  d2 ra[RA], rb[RB];

  auto load_global = [&](int t) {
    const int s = t < steps0 ? 0 : 1;
    const GemmSeg& sg = g.seg[s];
    const int k0 = (s ? t - steps0 : t) * BK;
    const double* Ab = sg.A + (long)filt * sg.strideA;
    const double* Bb = sg.B + (long)filt * sg.strideB;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int idx = tid + 256 * r;
      const int k = idx / (16 * WM), p = idx % (16 * WM);
      const int row = m0 + 2 * p;
      d2 v = d2{0.0, 0.0};
      if (row < g.Mp && k0 + k < sg.K) v = *reinterpret_cast<const d2*>(Ab + row + (long)(k0 + k) * sg.lda);
      ra[r] = v;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int idx = tid + 256 * r;
      const int k = idx / (16 * WN), p = idx % (16 * WN);
      const int col = n0 + 2 * p;
      d2 v = d2{0.0, 0.0};
      if (col < g.Np && k0 + k < sg.K) v = *reinterpret_cast<const d2*>(Bb + col + (long)(k0 + k) * sg.ldb);
      if (sg.scale && k0 + k < sg.K) {
        const double sc = sg.scale[(long)filt * sg.strideScale + k0 + k];
        v *= sc;
      }
      rb[r] = v;
    }
  };

  auto store_lds = [&]() {
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int idx = tid + 256 * r;
      const int k = idx / (16 * WM), p = idx % (16 * WM);
      *reinterpret_cast<d2*>(As + k * LDAS + 2 * p) = ra[r];
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int idx = tid + 256 * r;
      const int k = idx / (16 * WN), p = idx % (16 * WN);
      *reinterpret_cast<d2*>(Bs + k * LDBS + 2 * p) = rb[r];
    }
  };

  if (nsteps > 0) load_global(0);
  for (int t = 0; t < nsteps; ++t) {
    store_lds();
    __syncthreads();
    if (t + 1 < nsteps) load_global(t + 1);
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      double a[NA], bb[NB];
#pragma unroll
      for (int x = 0; x < NA; ++x) a[x] = As[(4 * s + lg) * LDAS + 16 * arow(x) + li];
#pragma unroll
      for (int x = 0; x < NB; ++x) bb[x] = Bs[(4 * s + lg) * LDBS + 16 * bcol(x) + li];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        if (active & (1u << q))
          acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(bb[slot_b(q)], a[slot_a(q)], acc[q], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // Epilogue. acc[q][r] = C[i = I0 + li][j = J0 + lg + 4r]
  double* Cb = g.C + (long)filt * g.strideC;
  double* C2b = g.C2 ? g.C2 + (long)filt * g.strideC2 : nullptr;
  const double* dg = g.diag ? g.diag + (long)filt * g.strideDiag : nullptr;
  double* Tw = smem + wave * (16 * 17);  // per-wave 16x16 transpose pad (main-loop LDS is free now)
  const bool need_t = g.lower_only || C2b;
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    if (!(active & (1u << q))) continue;
    const int I0 = m0 + 16 * arow(slot_a(q)), J0 = n0 + 16 * bcol(slot_b(q));
    const int i = I0 + li;
    d4 v = acc[q];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = J0 + lg + 4 * r;
      if (g.epilogue == EPI_ADD_DIAG) {
        if (i == j) v[r] += dg[i];
      } else if (g.epilogue == EPI_SUB_IDENT) {
        if (i == j) v[r] -= 1.0;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = J0 + lg + 4 * r;
      if (!g.lower_only || i >= j) Cb[i + (long)j * g.ldc] = v[r];
      if (need_t) Tw[(lg + 4 * r) * 17 + li] = v[r];   // T[j_local][i_local]
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
        if (g.lower_only && i2 > j2) Cb[j2 + (long)i2 * g.ldc] = t;
        if (C2b) C2b[j2 + (long)i2 * g.ldc2] = t;
      }
    }
  }
}

// k-panel depth: 32 where two workgroups' LDS (2 x BK x (BM+BN+32) doubles) still fit
// the CU's 160 KiB - twice the bytes in flight per workgroup and half the barriers -
// else 16. K segments that are not a multiple of 32 are handled by zero-filling.
template <int WM, int WN>
constexpr int pick_bk() {
  return (WM * WN <= 16 && 2 * 32 * (32 * WM + 32 * WN + 32) * 8 <= 160 * 1024) ? 32 : 16;   // larger tiles: registers
}

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_kernel(GemmArgs g) {
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
  const int tile = slot % nt;
  if (filt >= g.batch) return;
  const int tm = tile % g.tiles_m, tn = tile / g.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  if (g.lower_only && n0 >= m0 + BM) return;  // tile strictly above the diagonal
  if constexpr (WM == 4 && WN == 4) {
    if (g.lower_only && m0 == n0) {
      gemm_tile<WM, WN, BK, true>(g, filt, m0, n0, smem);
      return;
    }
  }
  gemm_tile<WM, WN, BK, false>(g, filt, m0, n0, smem);
}

template <int WM, int WN>
int launch_t(const GemmArgs& a, hipStream_t stream) {
  GemmArgs g = a;
  g.tiles_m = (g.Mp + 32 * WM - 1) / (32 * WM);
  g.tiles_n = (g.Np + 32 * WN - 1) / (32 * WN);
  const int groups = (g.batch + 7) / 8;
  const int grid = groups * 8 * g.tiles_m * g.tiles_n;
  if (grid <= 0) return 0;
  hipLaunchKernelGGL((gemm_nt_f64_kernel<WM, WN>), dim3(grid), dim3(256), 0, stream, g);
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
  int wm = pick_w(Mp), wn = pick_w(Np);
  // accumulator budget: WM*WN <= 20 (160 VGPRs of accumulators)
  while (wm * wn > 20 || (wm == 4 && wn == 5)) {   // <4,5> spills; <5,4> does not
    if (wn >= wm) --wn; else --wm;
  }
  *WM = wm;
  *WN = wn;
}

int launch_gemm_nt_f64(const GemmArgs& a, hipStream_t stream) {
  int wm, wn;
  gemm_pick_tile(a.Mp, a.Np, a.lower_only, &wm, &wn);
#define XIVO_GEMM_CASE(M_, N_) \
  if (wm == M_ && wn == N_) return launch_t<M_, N_>(a, stream);
  XIVO_GEMM_CASE(2, 2) XIVO_GEMM_CASE(2, 3) XIVO_GEMM_CASE(2, 4) XIVO_GEMM_CASE(2, 5)
  XIVO_GEMM_CASE(3, 2) XIVO_GEMM_CASE(3, 3) XIVO_GEMM_CASE(3, 4) XIVO_GEMM_CASE(3, 5)
  XIVO_GEMM_CASE(4, 2) XIVO_GEMM_CASE(4, 3) XIVO_GEMM_CASE(4, 4)
  XIVO_GEMM_CASE(5, 2) XIVO_GEMM_CASE(5, 3) XIVO_GEMM_CASE(5, 4)
#undef XIVO_GEMM_CASE
  return (int)hipErrorInvalidValue;
}

}  // namespace xivo_hip
