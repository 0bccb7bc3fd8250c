// Batched fp64 MFMA GEMM for the EKF measurement update (gfx950 / MI355X).
//
// One kernel serves every dense contraction of Estimator::UpdateJosephForm
// (/root/reference/src/estimator.cpp:1257-1288) and the covariance-propagation
// cross blocks (/root/reference/src/rk4.cpp:98-102):
//     HP   = H * P                          (estimator.cpp:1259 / :1266)
//     S    = HP * H^T + diag(R)             (estimator.cpp:1259-1263)
//     A    = K * H - I                      (estimator.cpp:1276-1279)
//     T    = A * P                          (estimator.cpp:1280, first product)
//     P+   = T * A^T + K diag(R) K^T        (estimator.cpp:1280-1287, fused)
// All of them are written as C = A_op * B_op^T with the output index contiguous
// in both operands ("NT", column-major) - P is symmetric so H*P reads P by rows,
// and H^T is kept next to H - so there is exactly ONE LDS layout and no
// transposing stage.
//
// Tiling (CDNA4): 256 threads = 4 wave64 in a 2x2 grid; each wave owns
// WM x WN tiles of v_mfma_f64_16x16x4_f64 (4 accumulator VGPR pairs each).
// A/B k-panels (BK = 16) are staged global -> registers -> LDS (the register
// leg is issued before the MFMA block of the previous panel so HBM/L2 latency
// hides under ~80 MFMAs x 64 cycles), LDS rows padded by 16 doubles so the two
// k-rows a ds_read_b64 lane group touches fall in different bank halves.
// The accumulator is computed TRANSPOSED (mfma(b, a)) so that each store
// instruction writes four 128-byte runs of the column-major output.
#include "common.h"

namespace xivo_hip {

namespace {

constexpr int BK = 16;

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_kernel(GemmArgs g) {
  constexpr int BM = 32 * WM, BN = 32 * WN;
  constexpr int LDAS = BM + 16, LDBS = BN + 16;
  __shared__ __attribute__((aligned(16))) double smem[BK * (LDAS + LDBS)];
  double* As = smem;
  double* Bs = smem + BK * LDAS;

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

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wr = wave & 1, wc = wave >> 1;
  const int wrow0 = wr * 16 * WM, wcol0 = wc * 16 * WN;

  // number of valid 16-blocks for this wave (wave-uniform)
  int mvalid = (g.Mp - m0 - wrow0) / 16;
  mvalid = mvalid < 0 ? 0 : (mvalid > WM ? WM : mvalid);
  int nvalid = (g.Np - n0 - wcol0) / 16;
  nvalid = nvalid < 0 ? 0 : (nvalid > WN ? WN : nvalid);

  d4 acc[WM][WN];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int n = 0; n < WN; ++n) acc[m][n] = d4{0.0, 0.0, 0.0, 0.0};

  const int steps0 = g.seg[0].K / BK;
  const int steps1 = g.nseg > 1 ? g.seg[1].K / BK : 0;
  const int nsteps = steps0 + steps1;

  d2 ra[WM], rb[WN];

  auto load_global = [&](int t) {
    const int s = t < steps0 ? 0 : 1;
    const GemmSeg& sg = g.seg[s];
    const int k0 = (s ? t - steps0 : t) * BK;
    const double* Ab = sg.A + (long)filt * sg.strideA;
    const double* Bb = sg.B + (long)filt * sg.strideB;
#pragma unroll
    for (int r = 0; r < WM; ++r) {
      const int idx = tid + 256 * r;
      const int k = idx / (16 * WM), p = idx % (16 * WM);
      const int row = m0 + 2 * p;
      d2 v = d2{0.0, 0.0};
      if (row < g.Mp) v = *reinterpret_cast<const d2*>(Ab + row + (long)(k0 + k) * sg.lda);
      ra[r] = v;
    }
#pragma unroll
    for (int r = 0; r < WN; ++r) {
      const int idx = tid + 256 * r;
      const int k = idx / (16 * WN), p = idx % (16 * WN);
      const int col = n0 + 2 * p;
      d2 v = d2{0.0, 0.0};
      if (col < g.Np) v = *reinterpret_cast<const d2*>(Bb + col + (long)(k0 + k) * sg.ldb);
      if (sg.scale) {
        const double sc = sg.scale[(long)filt * sg.strideScale + k0 + k];
        v *= sc;
      }
      rb[r] = v;
    }
  };

  auto store_lds = [&]() {
#pragma unroll
    for (int r = 0; r < WM; ++r) {
      const int idx = tid + 256 * r;
      const int k = idx / (16 * WM), p = idx % (16 * WM);
      *reinterpret_cast<d2*>(As + k * LDAS + 2 * p) = ra[r];
    }
#pragma unroll
    for (int r = 0; r < WN; ++r) {
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
      double a[WM], bb[WN];
      const double* ap = As + (4 * s + lg) * LDAS + wrow0 + li;
      const double* bp = Bs + (4 * s + lg) * LDBS + wcol0 + li;
#pragma unroll
      for (int m = 0; m < WM; ++m) a[m] = ap[16 * m];
#pragma unroll
      for (int n = 0; n < WN; ++n) bb[n] = bp[16 * n];
#pragma unroll
      for (int m = 0; m < WM; ++m) {
        if (m < mvalid) {
#pragma unroll
          for (int n = 0; n < WN; ++n) {
            if (n < nvalid)
              acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(bb[n], a[m], acc[m][n], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }

  // Epilogue. acc[m][n][r] = C[i = .. + li][j = .. + lg + 4r]
  double* Cb = g.C + (long)filt * g.strideC;
  const double* dg = g.diag ? g.diag + (long)filt * g.strideDiag : nullptr;
#pragma unroll
  for (int m = 0; m < WM; ++m) {
    if (m >= mvalid) continue;
    const int i = m0 + wrow0 + 16 * m + li;
#pragma unroll
    for (int n = 0; n < WN; ++n) {
      if (n >= nvalid) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = n0 + wcol0 + 16 * n + lg + 4 * r;
        double v = acc[m][n][r];
        if (g.epilogue == EPI_ADD_DIAG) {
          if (i == j) v += dg[i];
        } else if (g.epilogue == EPI_SUB_IDENT) {
          if (i == j) v -= 1.0;
        }
        if (g.lower_only) {
          // keep the lower triangle authoritative and mirror it, so the result
          // is exactly symmetric (the reference never re-symmetrises P,
          // estimator.cpp:1280 - the asymmetry there is rounding noise)
          if (i >= j) {
            Cb[i + (long)j * g.ldc] = v;
            if (i != j) Cb[j + (long)i * g.ldc] = v;
          }
        } else {
          Cb[i + (long)j * g.ldc] = v;
        }
      }
    }
  }
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

void gemm_pick_tile(int Mp, int Np, int* WM, int* WN) {
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
  gemm_pick_tile(a.Mp, a.Np, &wm, &wn);
#define XIVO_GEMM_CASE(M_, N_) \
  if (wm == M_ && wn == N_) return launch_t<M_, N_>(a, stream);
  XIVO_GEMM_CASE(2, 2) XIVO_GEMM_CASE(2, 3) XIVO_GEMM_CASE(2, 4) XIVO_GEMM_CASE(2, 5)
  XIVO_GEMM_CASE(3, 2) XIVO_GEMM_CASE(3, 3) XIVO_GEMM_CASE(3, 4) XIVO_GEMM_CASE(3, 5)
  XIVO_GEMM_CASE(4, 2) XIVO_GEMM_CASE(4, 3) XIVO_GEMM_CASE(4, 4) XIVO_GEMM_CASE(4, 5)
  XIVO_GEMM_CASE(5, 2) XIVO_GEMM_CASE(5, 3) XIVO_GEMM_CASE(5, 4)
#undef XIVO_GEMM_CASE
  return (int)hipErrorInvalidValue;
}

}  // namespace xivo_hip
