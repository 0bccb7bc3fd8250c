// C ABI of the MI355X EKF measurement-update path (include/xivo_hip.h).
// Host-side orchestration only: owns the device buffers of a batch of filters,
// sequences the kernels of gemm_f64.hip / chol_trsm.hip / ekf_kernels.hip on one
// HIP stream, never throws and never aborts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include "../../include/xivo_hip.h"
#include "common.h"
#include "ekf_kernels.h"
#include "ell.h"
#include "fused_update.h"

using namespace xivo_hip;

namespace {

enum Stage : int {
  ST_JAC = 0, ST_GATE, ST_STACK, ST_HP, ST_S, ST_CHOL, ST_TRSM, ST_KH, ST_AP, ST_PNEW, ST_OTHER, ST_PROP_STATE, ST_PROP_TAIL, ST_COUNT
};
const char* kStageNames[ST_COUNT] = {"jac_instate", "mh_gate", "stack_H", "gemm_HP", "gemm_S", "chol_S",
                                     "trsm_gain", "gemm_KH_I", "gemm_AP", "gemm_Pnew", "other", "propagate_state",
                                     "propagate_tail"};

struct EventPair { hipEvent_t a, b; int stage; };

}  // namespace

struct xivo_hip_ctx {
  int device = 0;
  int N = 0, Np = 0, Mmax = 0, Mpmax = 0, Bmax = 0;
  unsigned flags = 0;
  hipStream_t stream = nullptr;
  // per-filter device buffers
  double *P = nullptr, *Psnap = nullptr, *H = nullptr, *HT = nullptr, *HP = nullptr, *PHT = nullptr, *S = nullptr;
  double *K = nullptr, *A = nullptr, *T = nullptr, *invD = nullptr, *inn = nullptr, *diagR = nullptr;
  double *err = nullptr, *staging = nullptr, *scratch = nullptr;
  double *neg1 = nullptr, *yvec = nullptr;   // symmetric form: a vector of -1 (operand scale), y = L^-1 inn per filter
  int* status = nullptr;
  // row-pair compressed H (ell.h) + host mirror of the per-filter "does not fit" flag
  EllBuffers ell{};
  std::vector<int> ell_over_h, ell_nc_h, ell_pw_h;
  int* ell_flags_h = nullptr;   // pinned, device-mapped [Bmax][3]: over / nc / pw as the hand-over kernel leaves them
  int* ell_flags_d = nullptr;   // its device alias
  int last_path = 0;
  int last_route = 0;   // UpdateRoute of the last pass (xivo_hip_last_route)
  // dense H / H^T of the stacked rows: written eagerly by set_measurements, lazily after xivo_hip_stack
  bool dense_valid = true;
  bool dense_from_ell = false;   // the stacked rows came in through set_measurements (compressed rows are the source)
  bool ht_valid = true;          // the transposed dense copy H^T matches H (false after a producer skipped it: mixed stacking)
  // mixed stacking (round 3): in-state rows [0, mixed_row0) exist in the row-pair compressed form only, the OOS rows
  // appended by xivo_hip_oos_project from row mixed_row0 on in the dense buffer only; -1: not in that mode
  int mixed_row0 = -1;
  bool h_clean = true;           // every row of the dense H buffer the mixed mode has not written itself is zero
  double stack_R = 0.0; int stack_B = 0;
  size_t staging_elems = 0;
  long sP = 0, sH = 0, sHT = 0, sS = 0, sK = 0, sInvD = 0, sA = 0;   // sA: A buffer, max(N x N, N x M)
  int M = 0, Mp = 0;  // rows currently staged
  int chunk = 0;      // filters per pipeline pass (0 = whole batch)
  int call_batch = 0; // filters of the whole update call being walked in chunks (0: not chunked)
  int* ldlt_used = nullptr;     // per filter: 1 = the last update went through the pivoted L D L^T fallback
  // G-level
  xivo_layout lay{};
  xivo_cam cam{};
  bool have_layout = false;
  // online-calibration builds, measurement side (xivo_hip_set_calib): extra Jacobian blocks, dense stacking
  bool calib_on = false;       // measurement side of an online-calibration build (td / Cg / bg / intrinsics blocks)
  bool calib_motion = false;   // motion side: kMotionSize > 23 (xivo_hip_propagate_calib)
  xivo_calib_layout cl{-1, -1, 0, 0};
  xivo_calib_in* calib = nullptr;   // [Bmax]
  double* Jc = nullptr;             // [Bmax x Fmax x 44]
  int Fmax = 0, F = 0;
  xivo_pose_in* poses = nullptr;
  int* absorb_count = nullptr;   // State::counter of every filter (src/core.h:120-122)
  // OnePointRANSAC scratch (allocated on first use): BackupState copies, selection results
  double* Prs = nullptr; xivo_pose_in* poses_rs = nullptr; xivo_group_in* groups_rs = nullptr;
  unsigned char *rs_low = nullptr, *rs_lowkeep = nullptr, *rs_keep = nullptr;
  unsigned long long *rs_zg = nullptr, *rs_gmask = nullptr;
  int *rs_state = nullptr, *rs_gauge = nullptr, *rs_nrej = nullptr;
  double* rs_chi = nullptr;
  int rs_Fmax = 0;
  xivo_group_in* groups = nullptr;
  xivo_feat_in* feats = nullptr;
  double *J = nullptr, *finn = nullptr, *dist = nullptr;
  unsigned char* mask = nullptr;
  int gate_sparse_last = 0;   // mask/dist row stride: Fmax after the layout-faithful gate, F after the dense one
  int* rows_instate = nullptr;
  xivo_oos_in* oos = nullptr;
  int oos_cap = 0;
  int oos_row0 = -1;   // first row of the OOS block of the last xivo_hip_oos_project (-1: none since the last stacking)
  double oos_R = 0.0;
  double* pd_h = nullptr; double pd_h0 = 0.0;   // step-size-controlled Dormand-Prince: the step each filter carries (xivo_hip_propagate)
  int oos_nb = 0, oos_n = 0, oos_max_rows = 0, oos_whole = 0;   // shape of the resident OOS list (xivo_hip_oos_project with feats == NULL)
  int* oos_rows = nullptr;
  xivo_calib_in* calib_rs = nullptr;            // BackupState of the calibration state (OnePointRANSAC, online-calibration builds)
  // online-calibration builds on the sparse pipeline (round 5): the calibration columns of the stacked rows as a dense
  // [Mpmax x LEAD_K] block per filter next to the row-pair compressed rows; lead_valid: the current stacking has one
  double* Hlead = nullptr; bool lead_valid = false;
  void* lc_buf = nullptr; size_t lc_cap = 0;   // xivo_hip_close_loop_stack: matches | dense rows | inn | diagR
  xivo_subfilter_feat* sub = nullptr;   // staging of xivo_hip_subfilter_update
  std::vector<char> hstage;                        // host staging of d2h_rows
  void* edit_buf = nullptr; size_t edit_cap = 0;   // device copy of the ops of xivo_hip_edit_batch
  // one-filter plumbing call (xivo_hip_update_joseph_host): page-locked, device-mapped staging block owned by the context,
  // scratch of the host-side row compression
  char* pin_h = nullptr; char* pin_d = nullptr; size_t pin_bytes = 0;
  struct HostCompressScratch { std::vector<int> cnt, occ, cslot, n; std::vector<double> v; } hc;
  size_t sub_cap = 0;
  // timing
  hipEvent_t t0 = nullptr, t1 = nullptr;
  std::vector<EventPair> pool;
  size_t pool_used = 0;
  float stage_ms[ST_COUNT] = {0};
  int stage_launches[ST_COUNT] = {0};
  double stage_flops[ST_COUNT] = {0};
  double stage_bytes[ST_COUNT] = {0};         // algorithmic HBM bytes of the stage's last launch (inputs once + outputs once)
  char stage_kernel[ST_COUNT][64] = {{0}};   // kernel instantiation of the stage's last launch (as rocprofv3 names it)
};

namespace {

// XIVO_HIP_DEBUG=1: name the failing runtime call on stderr (the C ABI itself only returns a status)
static bool debug_on() { static const bool on = getenv("XIVO_HIP_DEBUG") != nullptr; return on; }
#define HIP_TRY(expr)                              \
  do {                                             \
    hipError_t e_ = (expr);                        \
    if (e_ != hipSuccess) {                        \
      if (debug_on()) fprintf(stderr, "xivo_hip: %s -> %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return XIVO_HIP_ERR_HIP;                     \
    }                                              \
  } while (0)

template <class T>
int dev_alloc(T** p, size_t n) {
  if (n == 0) { *p = nullptr; return XIVO_HIP_OK; }
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) return XIVO_HIP_ERR_NOMEM;
  e = hipMemset(*p, 0, n * sizeof(T));
  return e == hipSuccess ? XIVO_HIP_OK : XIVO_HIP_ERR_HIP;
}

struct StageTimer {
  xivo_hip_ctx* c; EventPair* ep = nullptr;
  StageTimer(xivo_hip_ctx* ctx, int stage, double flops, const char* kernel = nullptr, double bytes = 0.0) : c(ctx) {
    if (!(c->flags & XIVO_HIP_FLAG_PROFILE)) return;
    c->stage_bytes[stage] = bytes;
    if (kernel) { strncpy(c->stage_kernel[stage], kernel, 63); c->stage_kernel[stage][63] = 0; }
    if (c->pool_used >= c->pool.size()) {
      EventPair np; np.stage = stage;
      if (hipEventCreate(&np.a) != hipSuccess || hipEventCreate(&np.b) != hipSuccess) return;
      c->pool.push_back(np);
    }
    ep = &c->pool[c->pool_used++];
    ep->stage = stage;
    c->stage_launches[stage]++;
    c->stage_flops[stage] = flops;
    hipEventRecord(ep->a, c->stream);
  }
  ~StageTimer() { if (ep) hipEventRecord(ep->b, c->stream); }
};

int collect_profile(xivo_hip_ctx* c) {
  if (c->pool_used == 0) return XIVO_HIP_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < c->pool_used; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->pool[i].a, c->pool[i].b) == hipSuccess) c->stage_ms[c->pool[i].stage] += ms;
  }
  c->pool_used = 0;
  return XIVO_HIP_OK;
}

MeasBuffers meas_buffers(xivo_hip_ctx* c) {
  MeasBuffers mb;
  mb.H = c->H; mb.strideH = c->sH; mb.ldh = c->Mpmax;
  mb.HT = c->HT; mb.strideHT = c->sHT; mb.ldht = c->Np;
  mb.inn = c->inn; mb.strideInn = c->Mpmax;
  mb.diagR = c->diagR; mb.strideR = c->Mpmax;
  return mb;
}

// leading state columns the calibration blocks live in: td 23, Cg 24..32, (Ca 33..38,) bg 9..11, intrinsics up to 39..47
constexpr int LEAD_K = 48;
static bool calib_sparse(const xivo_hip_ctx* c) {   // (XIVO_HIP_FLAG_DENSE_H keeps the round-4 dense stacking of these builds)
  return c->calib_on && c->Hlead && c->cl.cam_begin + 9 <= LEAD_K && c->Np >= LEAD_K &&
         !(c->flags & (XIVO_HIP_FLAG_DENSE_H | XIVO_HIP_FLAG_SYMMETRIC_FORM | XIVO_HIP_FLAG_STANDALONE_TAIL));
}

SceneBuffers scene_buffers(xivo_hip_ctx* c) {
  SceneBuffers sb;
  sb.poses = c->poses; sb.groups = c->groups; sb.feats = c->feats;
  sb.J = c->J; sb.finn = c->finn; sb.mask = c->mask; sb.dist = c->dist;
  sb.Fmax = c->Fmax; sb.F = c->F;
  sb.calib = c->calib_on ? c->calib : nullptr; sb.Jc = c->calib_on ? c->Jc : nullptr; sb.cl = c->cl;
  if (!c->calib_on) sb.cl = xivo_calib_layout{-1, -1, 0, 0};
  sb.invdepth = (c->flags & XIVO_HIP_FLAG_INVDEPTH) ? 1 : 0;
  return sb;
}

// Device -> host copy of `rows` rows of `width` bytes (device pitch dpitch, host pitch hpitch). hipMemcpy2D issues one
// small DMA per row - 13 ms for 2048 rows of 30 bytes - so short rows come over as one contiguous block and are
// repacked on the host.
int d2h_rows(xivo_hip_ctx* c, void* dst, size_t hpitch, const void* src, size_t dpitch, size_t width, size_t rows) {
  if (rows == 0 || width == 0) return XIVO_HIP_OK;
  if (width == dpitch && width == hpitch) {
    HIP_TRY(hipMemcpyAsync(dst, src, width * rows, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return XIVO_HIP_OK;
  }
  const size_t span = dpitch * (rows - 1) + width;
  if (c->hstage.size() < span) c->hstage.resize(span);
  HIP_TRY(hipMemcpyAsync(c->hstage.data(), src, span, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (size_t r = 0; r < rows; ++r) memcpy((char*)dst + r * hpitch, c->hstage.data() + r * dpitch, width);
  return XIVO_HIP_OK;
}

int ensure_staging(xivo_hip_ctx* c, size_t elems) {
  if (elems <= c->staging_elems) return XIVO_HIP_OK;
  if (c->staging) hipFree(c->staging);
  c->staging = nullptr; c->staging_elems = 0;
  if (hipMalloc((void**)&c->staging, elems * sizeof(double)) != hipSuccess) return XIVO_HIP_ERR_NOMEM;
  c->staging_elems = elems;
  return XIVO_HIP_OK;
}

// copy nb host matrices (rows x cols, leading dim ld, `stride` elements apart)
// into a packed device staging area (ld = rows)
int h2d_packed(xivo_hip_ctx* c, double* dst, const double* src, int nb, int rows, int cols, long stride, int ld) {
  if (ld == rows) {
    HIP_TRY(hipMemcpy2DAsync(dst, (size_t)rows * cols * sizeof(double), src, (size_t)stride * sizeof(double),
                             (size_t)rows * cols * sizeof(double), nb, hipMemcpyHostToDevice, c->stream));
  } else {
    for (int b = 0; b < nb; ++b)
      HIP_TRY(hipMemcpy2DAsync(dst + (size_t)b * rows * cols, (size_t)rows * sizeof(double), src + (size_t)b * stride,
                               (size_t)ld * sizeof(double), (size_t)rows * sizeof(double), cols,
                               hipMemcpyHostToDevice, c->stream));
  }
  return XIVO_HIP_OK;
}

int d2h_packed(xivo_hip_ctx* c, double* dst, const double* src, int nb, int rows, int cols, long stride, int ld) {
  if (ld == rows) {
    HIP_TRY(hipMemcpy2DAsync(dst, (size_t)stride * sizeof(double), src, (size_t)rows * cols * sizeof(double),
                             (size_t)rows * cols * sizeof(double), nb, hipMemcpyDeviceToHost, c->stream));
  } else {
    for (int b = 0; b < nb; ++b)
      HIP_TRY(hipMemcpy2DAsync(dst + (size_t)b * stride, (size_t)ld * sizeof(double), src + (size_t)b * rows * cols,
                               (size_t)rows * sizeof(double), (size_t)rows * sizeof(double), cols,
                               hipMemcpyDeviceToHost, c->stream));
  }
  return XIVO_HIP_OK;
}

bool bad_range(xivo_hip_ctx* c, int b0, int nb) { return !c || b0 < 0 || nb < 0 || b0 + nb > c->Bmax; }

struct GemmExtra {
  int epi = EPI_NONE;
  const double* diag = nullptr; long sDiag = 0;
  const double* msub = nullptr; long sMsub = 0; int ldmsub = 0;
  const double* mcol = nullptr; long sMcol = 0;
  double* C2 = nullptr; long sC2 = 0; int ldc2 = 0;
  int c2_rows = 0;   // > 0: the transposed copy only of the leading c2_rows rows of C (the columns of C2 a consumer reads)
  int lower_only = 0;
  int fp32 = 0;
  int a_f32 = 0;   // first operand stored as float
  int b_f32 = 0;   // second operand stored as float
  int no_mirror = 0;
  const int* skip = nullptr;   // per-filter status: non-zero = leave the output of that filter untouched
  const double* scale0 = nullptr;   // per-k scale of the first segment's B operand (same vector for every filter)
  int small_tiles = 0;   // symmetric output on 64 x 64 tiles (latency route)
};

int gemm(xivo_hip_ctx* c, int stage, int B, int rows, int cols, const double* A0, long sA0, int lda0,
         const double* B0, long sB0, int ldb0, int K0, const double* A1, long sA1, int lda1, const double* B1,
         long sB1, int ldb1, int K1, const double* scale1, long sScale1, double* C, long sC, int ldc,
         const GemmExtra& x) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.seg[0] = GemmSeg{A0, B0, x.scale0, sA0, sB0, 0, lda0, ldb0, K0, x.a_f32, x.b_f32};
  g.nseg = 1;
  if (A1) {
    g.seg[1] = GemmSeg{A1, B1, scale1, sA1, sB1, sScale1, lda1, ldb1, K1, 0, 0};
    g.nseg = 2;
  }
  g.C = C; g.strideC = sC; g.ldc = ldc; g.Mp = rows; g.Np = cols;
  g.C2 = x.C2; g.strideC2 = x.sC2; g.ldc2 = x.ldc2; g.c2_rows = x.c2_rows;
  g.diag = x.diag; g.strideDiag = x.sDiag; g.Msub = x.msub; g.strideMsub = x.sMsub; g.ldmsub = x.ldmsub;
  g.McolScale = x.mcol; g.strideMcol = x.sMcol;
  g.epilogue = x.epi; g.lower_only = x.lower_only; g.no_mirror = x.no_mirror; g.batch = B; g.fp32 = x.fp32;
  g.skip_status = x.skip; g.small_tiles = x.small_tiles;
  // algorithmic flops of the product: a symmetric output needs its lower triangle only
  const double outs = x.lower_only ? 0.5 * rows * (cols + 1.0) : (double)rows * cols;
  const double flops = 2.0 * outs * (double)(K0 + (A1 ? K1 : 0)) * B;
  const bool sym = g.lower_only && rows == cols && gemm_sym_supported(rows) && !g.C2 && !g.fp32 &&
                   (g.epilogue == EPI_NONE || g.epilogue == EPI_ADD_DIAG);
  char label[64] = "gemm_sym_f64_kernel";
  if (!sym) gemm_kernel_label(g, label, sizeof(label));
  const double bytes = 8.0 * B * ((double)rows * K0 * (x.a_f32 ? 0.5 : 1.0) + (double)cols * K0 * (x.b_f32 ? 0.5 : 1.0) + (A1 ? ((double)rows + cols) * K1 : 0.0) +
                                  (x.msub ? outs : 0.0) + (double)rows * cols + (x.C2 ? (double)rows * cols : 0.0))
                       - (A0 == B0 ? 8.0 * B * (double)cols * K0 : 0.0);   // a symmetric product reads its one operand once
  StageTimer st(c, stage, flops, label, bytes);
  const int rc = sym ? launch_gemm_sym_f64(g, c->stream) : launch_gemm_nt_f64(g, c->stream);
  if (rc != 0 && debug_on()) fprintf(stderr, "xivo_hip: gemm launch (%s) -> %s\n", label, hipGetErrorString((hipError_t)rc));
  return rc == 0 ? XIVO_HIP_OK : XIVO_HIP_ERR_HIP;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Route selection of Estimator::UpdateJosephForm (src/estimator.cpp:1257-1288), in ONE place. Every pass of the update
// (update_joseph_range) asks plan_update() once; the pipelines below only execute what the plan says, and
// tests/test_update_gpu.py::test_every_route_of_the_plan enumerates the routes of this table against the oracle.
//
//   route              | rows of H                   | gain + covariance                                   | when
//   -------------------+-----------------------------+-----------------------------------------------------+--------------------------
//   FUSED              | row-pair compressed         | one kernel per filter (fused_update.hip)            | M <= 64 / N <= 256 or M <= 112 / N <= 192, default form
//   SPARSE_IN_SOLVE    | compressed (+ OOS / lead)   | whitened Joseph form inside the solve kernel        | N <= 256, M <= 176, > 64 filters
//   SPARSE_WHITENED    | compressed (+ OOS / lead)   | whitened outputs V^T, Y^T + tiled P - V^T Y         | wider shapes; <= 64 filters (latency route)
//   SPARSE_SYMMETRIC   | compressed                  | P - W^T W, forward substitution only                | XIVO_HIP_FLAG_SYMMETRIC_FORM
//   SPARSE_TAIL        | compressed                  | T = K(HP) - P, G = T H^T + K R, P+ = G K^T - T      | XIVO_HIP_FLAG_STANDALONE_TAIL
//   DENSE_ASCODED      | dense                       | A = KH - I, T = A P, P+ = T A^T + K R K^T           | XIVO_HIP_FLAG_DENSE_H
//   DENSE_WHITENED     | dense (H does not compress) | dense H P and S, then as SPARSE_IN_SOLVE / _WHITENED | an H without XIVO's row structure
//   DENSE_SYMMETRIC    | dense                       | as SPARSE_SYMMETRIC                                 | SYMMETRIC_FORM on dense rows
enum UpdateRoute : int { ROUTE_FUSED = 0, ROUTE_SPARSE_IN_SOLVE, ROUTE_SPARSE_WHITENED, ROUTE_SPARSE_SYMMETRIC, ROUTE_SPARSE_TAIL,
                         ROUTE_DENSE_ASCODED, ROUTE_DENSE_WHITENED, ROUTE_DENSE_SYMMETRIC, ROUTE_COUNT };
static const char* kRouteNames[ROUTE_COUNT] = {"fused", "sparse_in_solve", "sparse_whitened", "sparse_symmetric", "sparse_tail",
                                               "dense_ascoded", "dense_whitened", "dense_symmetric"};
struct UpdatePlan {
  int route;
  bool sparse;        // the rows are used in their compressed form
  bool in_solve;      // the covariance update runs inside the solve kernel (one workgroup per filter)
  bool latency;       // few filters: streamed solve on four-wave workgroups + the product on 64 x 64 tiles
  bool stream8;       // N > 256 with a short factor: the streamed solve on eight-wave workgroups
  bool f32_whitened;  // XIVO_HIP_FLAG_FP32_WHITENED applies (the product runs outside the solve kernel because of the SHAPE)
};

// (B = the filters of this pass; with the batch walked in chunks - XIVO_HIP_CHUNK - the few-filter decision is made on the
//  WHOLE call's batch, c->call_batch: chunks of <= 64 filters of a large batch must not take the few-filter kernels)
static UpdatePlan plan_update(const xivo_hip_ctx* c, int b0, int B, bool gate) {
  const int Np = c->Np, Mp = c->Mp;
  const unsigned f = c->flags;
  UpdatePlan p{};
  bool sparse = !(f & XIVO_HIP_FLAG_DENSE_H);
  int nc_max = 0, pw_max = 1;
  for (int b = b0; b < b0 + B; ++b) {
    sparse = sparse && c->ell_over_h[b] == 0;
    nc_max = std::max(nc_max, c->ell_nc_h[b]); pw_max = std::max(pw_max, c->ell_pw_h[b]);
  }
  const bool extra_rows = c->mixed_row0 >= 0 || c->lead_valid;     // dense OOS rows / the leading calibration block next to the compressed rows
  // the stand-alone tail's G = T H^T walks compressed rows of ALL of H, and the compact gate of a calibration stacking reads whole rows
  if (sparse && extra_rows && ((f & XIVO_HIP_FLAG_STANDALONE_TAIL) || (c->lead_valid && gate))) sparse = false;
  if (sparse && c->lead_valid && (f & XIVO_HIP_FLAG_SYMMETRIC_FORM)) sparse = false;
  p.sparse = sparse;
  const bool holds = trsm_forms_T(Mp, Np);                          // one workgroup per filter holds the factor and every column of the state
  const int Ball = c->call_batch > B ? c->call_batch : B;
  p.latency = !(f & (XIVO_HIP_FLAG_THROUGHPUT_ROUTE | XIVO_HIP_FLAG_STANDALONE_TAIL | XIVO_HIP_FLAG_SYMMETRIC_FORM)) &&
              trsm_latency_route(Mp, Ball) && (sparse || !(f & XIVO_HIP_FLAG_DENSE_H));
  p.stream8 = !p.latency && Np > 256 && Mp / 16 <= 8;               // (N = 276, M = 120: 2.01 -> 1.39 ms per 4096 filters)
  if (f & XIVO_HIP_FLAG_SYMMETRIC_FORM) { p.route = sparse ? ROUTE_SPARSE_SYMMETRIC : ROUTE_DENSE_SYMMETRIC; p.in_solve = holds; return p; }
  if (!sparse && (f & XIVO_HIP_FLAG_DENSE_H)) { p.route = ROUTE_DENSE_ASCODED; return p; }
  if (sparse && (f & XIVO_HIP_FLAG_STANDALONE_TAIL)) { p.route = ROUTE_SPARSE_TAIL; return p; }
  if (sparse && !extra_rows && !(f & XIVO_HIP_FLAG_MULTI_KERNEL) && nc_max <= 12 && pw_max <= 9 && fused_update_supported(Mp, Np)) {
    p.route = ROUTE_FUSED; p.latency = false; return p;
  }
  p.in_solve = holds && !p.latency;
  p.f32_whitened = (f & XIVO_HIP_FLAG_FP32_WHITENED) && !holds;
  p.route = sparse ? (p.in_solve ? ROUTE_SPARSE_IN_SOLVE : ROUTE_SPARSE_WHITENED) : ROUTE_DENSE_WHITENED;
  return p;
}

static int ensure_gate_buffers(xivo_hip_ctx* c, int F);
static int ensure_dense(xivo_hip_ctx* c);
static int ensure_HT(xivo_hip_ctx* c);

extern "C" {

const char* xivo_hip_strerror(int s) {
  switch (s) {
    case XIVO_HIP_OK: return "ok";
    case XIVO_HIP_ERR_INVALID: return "invalid argument";
    case XIVO_HIP_ERR_HIP: return "HIP runtime error";
    case XIVO_HIP_ERR_NOT_SPD: return "innovation covariance S is not positive definite";
    case XIVO_HIP_ERR_NOMEM: return "out of device memory";
    case XIVO_HIP_ERR_UNSUPPORTED: return "size not supported by the compiled kernels";
    default: return "unknown status";
  }
}

void xivo_hip_destroy(xivo_hip_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  void* ptrs[] = {c->P, c->Psnap, c->H, c->HT, c->HP, c->PHT, c->S, c->K, c->A, c->T, c->invD, c->inn, c->diagR, c->err,
                  c->staging, c->scratch, c->neg1, c->yvec, c->status, c->poses, c->groups, c->feats, c->J, c->finn, c->dist,
                  c->mask, c->rows_instate, c->absorb_count, c->Prs, c->poses_rs, c->groups_rs, c->rs_low, c->rs_lowkeep, c->rs_keep,
                  c->rs_zg, c->rs_gmask, c->rs_state, c->rs_gauge, c->rs_nrej, c->rs_chi, c->oos, c->oos_rows, c->ell.idx, c->ell.val, c->ell.nc, c->ell.pw, c->ell.over, c->sub, c->edit_buf, c->lc_buf, c->calib_rs, c->Hlead, c->ldlt_used, c->calib, c->Jc, c->pd_h};
  for (void* p : ptrs) if (p) hipFree(p);
  if (c->ell_flags_h) hipHostFree(c->ell_flags_h);
  if (c->pin_h) hipHostFree(c->pin_h);
  for (auto& ep : c->pool) { hipEventDestroy(ep.a); hipEventDestroy(ep.b); }
  if (c->t0) hipEventDestroy(c->t0);
  if (c->t1) hipEventDestroy(c->t1);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int xivo_hip_create(xivo_hip_ctx** out, int device, int N, int M_max, int batch_max, unsigned flags) {
  if (!out || N <= 0 || M_max <= 0 || batch_max <= 0) return XIVO_HIP_ERR_INVALID;
  *out = nullptr;
  if (round_up16(M_max) / 16 > 24) return XIVO_HIP_ERR_UNSUPPORTED;  // trsm register budget
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return XIVO_HIP_ERR_HIP;
  HIP_TRY(hipSetDevice(device));
  xivo_hip_ctx* c = new (std::nothrow) xivo_hip_ctx();
  if (!c) return XIVO_HIP_ERR_NOMEM;
  c->device = device; c->N = N; c->Np = round_up16(N); c->Mmax = M_max; c->Mpmax = round_up16(M_max);
  c->Bmax = batch_max; c->flags = flags;
  if (const char* e = getenv("XIVO_HIP_CHUNK")) c->chunk = atoi(e);
  const size_t B = batch_max;
  const size_t Np = c->Np, Mp = c->Mpmax;
  c->sP = (long)(Np * Np); c->sH = (long)(Mp * Np); c->sHT = (long)(Np * Mp); c->sS = (long)(Mp * Mp);
  c->sK = (long)(Np * Mp); c->sInvD = (long)(Mp / 16 * 512); c->sA = c->sP > c->sK ? c->sP : c->sK;
  int rc = XIVO_HIP_OK;
  auto A = [&](auto** p, size_t n) { if (rc == XIVO_HIP_OK) rc = dev_alloc(p, n); };
  if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return XIVO_HIP_ERR_HIP; }
  A(&c->P, B * c->sP); A(&c->H, B * c->sH); A(&c->HT, B * c->sHT); A(&c->HP, B * c->sH); A(&c->PHT, B * c->sK);
  A(&c->S, B * c->sS); A(&c->K, B * c->sK); A(&c->A, B * c->sA); A(&c->T, B * c->sP);
  A(&c->invD, B * c->sInvD); A(&c->inn, B * Mp); A(&c->diagR, B * Mp); A(&c->err, B * Np);
  A(&c->status, B); A(&c->ldlt_used, B); A(&c->scratch, B * Np);
  A(&c->neg1, Mp); A(&c->yvec, B * Mp);
  if (rc == XIVO_HIP_OK) {
    std::vector<double> m1(Mp, -1.0);
    if (hipMemcpy(c->neg1, m1.data(), Mp * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = XIVO_HIP_ERR_HIP;
  }
  c->ell.pairs_max = (int)(Mp / 2);
  A(&c->ell.idx, B * c->ell.stride_idx()); A(&c->ell.val, B * c->ell.stride_val()); A(&c->ell.nc, B); A(&c->ell.pw, B); A(&c->ell.over, B);
  c->ell_over_h.assign(B, 1); c->ell_nc_h.assign(B, ELL_CW); c->ell_pw_h.assign(B, ELL_PW);
  // the hand-over kernel mirrors its three per-filter flags into host-mapped pinned memory: the host picks the kernel
  // instantiations from them after one stream synchronisation. (Three device-to-host copies into pageable vectors cost
  // 85 us of idle GPU per call - a third of a B = 1 step.) If the mapping is refused the copies are used.
  if (rc == XIVO_HIP_OK && hipHostMalloc(reinterpret_cast<void**>(&c->ell_flags_h), (size_t)B * 3 * sizeof(int), hipHostMallocMapped) == hipSuccess) {
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->ell_flags_d), c->ell_flags_h, 0) != hipSuccess) {
      hipHostFree(c->ell_flags_h); c->ell_flags_h = nullptr; c->ell_flags_d = nullptr;
    }
  } else c->ell_flags_h = nullptr;
  if (rc == XIVO_HIP_OK && hipEventCreate(&c->t0) != hipSuccess) rc = XIVO_HIP_ERR_HIP;
  if (rc == XIVO_HIP_OK && hipEventCreate(&c->t1) != hipSuccess) rc = XIVO_HIP_ERR_HIP;
  if (rc != XIVO_HIP_OK) { xivo_hip_destroy(c); return rc; }
  *out = c;
  return XIVO_HIP_OK;
}

int xivo_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// NUMA node of the host memory / cores next to `device` (sysfs of its PCI function), -1 when unknown: the launcher binds
// one rank per GPU to that node's cores (xivo_amd/shard.py) - eight ranks of an 8-GPU node must not pile up on socket 0
int xivo_hip_device_numa_node(int device) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) return -1;
  for (char* q = bdf; *q; ++q) if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a');
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node;
}

int xivo_hip_sync(xivo_hip_ctx* c) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c) return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XIVO_HIP_OK;
}

int xivo_hip_set_flags(xivo_hip_ctx* c, unsigned flags) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c) return XIVO_HIP_ERR_INVALID;
  c->flags = flags;
  return XIVO_HIP_OK;
}

// ------------------------------------------------------------------ P residency
int xivo_hip_upload_P(xivo_hip_ctx* c, int b0, int nb, const double* P, long stride, int ld) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !P || ld < c->N) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  HIP_TRY(hipSetDevice(c->device));
  const int N = c->N;
  int rc = ensure_staging(c, (size_t)nb * N * N);
  if (rc) return rc;
  rc = h2d_packed(c, c->staging, P, nb, N, N, stride, ld);
  if (rc) return rc;
  HIP_TRY((hipError_t)launch_unpack_P(c->staging, c->P + (long)b0 * c->sP, N, c->Np, c->Np, c->sP, nb, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // host buffer is only borrowed for the call
  return XIVO_HIP_OK;
}

int xivo_hip_download_P(xivo_hip_ctx* c, int b0, int nb, double* P, long stride, int ld) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !P || ld < c->N) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  HIP_TRY(hipSetDevice(c->device));
  const int N = c->N;
  int rc = ensure_staging(c, (size_t)nb * N * N);
  if (rc) return rc;
  HIP_TRY((hipError_t)launch_pack_P(c->P + (long)b0 * c->sP, c->staging, N, c->Np, c->sP, nb, c->stream));
  rc = d2h_packed(c, P, c->staging, nb, N, N, stride, ld);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XIVO_HIP_OK;
}

int xivo_hip_snapshot_P(xivo_hip_ctx* c) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c) return XIVO_HIP_ERR_INVALID;
  if (!c->Psnap) {
    if (hipMalloc((void**)&c->Psnap, (size_t)c->Bmax * c->sP * sizeof(double)) != hipSuccess) return XIVO_HIP_ERR_NOMEM;
  }
  HIP_TRY(hipMemcpyAsync(c->Psnap, c->P, (size_t)c->Bmax * c->sP * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return XIVO_HIP_OK;
}

int xivo_hip_restore_P(xivo_hip_ctx* c) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !c->Psnap) return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipMemcpyAsync(c->P, c->Psnap, (size_t)c->Bmax * c->sP * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return XIVO_HIP_OK;
}

int xivo_hip_p_zero_rc(xivo_hip_ctx* c, int b, int off, int len) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b, 1) || off < 0 || len < 0 || off + len > c->N) return XIVO_HIP_ERR_INVALID;
  return launch_p_zero_rc(c->P + (long)b * c->sP, c->Np, c->Np, off, len, c->stream) ? XIVO_HIP_ERR_HIP : XIVO_HIP_OK;
}

int xivo_hip_p_copy_rc(xivo_hip_ctx* c, int b, int dst, int src, int len) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b, 1) || dst < 0 || src < 0 || len < 0 || dst + len > c->N || src + len > c->N) return XIVO_HIP_ERR_INVALID;
  return launch_p_copy_rc(c->P + (long)b * c->sP, c->Np, c->Np, dst, src, len, c->stream) ? XIVO_HIP_ERR_HIP : XIVO_HIP_OK;
}

int xivo_hip_p_set_block3(xivo_hip_ctx* c, int b, int off, const double* P3) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b, 1) || !P3 || off < 0 || off + 3 > c->N) return XIVO_HIP_ERR_INVALID;
  double* dst = c->P + (long)b * c->sP + off + (long)off * c->Np;
  HIP_TRY(hipMemcpy2DAsync(dst, (size_t)c->Np * sizeof(double), P3, 3 * sizeof(double), 3 * sizeof(double), 3,
                           hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XIVO_HIP_OK;
}

int xivo_hip_p_diag(xivo_hip_ctx* c, int b, double* out) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b, 1) || !out) return XIVO_HIP_ERR_INVALID;
  HIP_TRY((hipError_t)launch_p_diag(c->P + (long)b * c->sP, c->Np, c->N, c->scratch, c->stream));
  HIP_TRY(hipMemcpyAsync(out, c->scratch, (size_t)c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XIVO_HIP_OK;
}

// ------------------------------------------------------------------ S-level
// Hand-over of dense measurements that already live in device memory (dH: M x N column-major per filter): ONE
// launch builds the row-pair compressed rows of the whole range; the padded dense copies are written only for
// the filters that do not fit it (they take the dense pipeline) and otherwise rebuilt from the compressed rows
// on demand (ensure_dense).
static int stage_measurements(xivo_hip_ctx* c, int b0, int nb, int M, const double* dH, long strideH, int ldh,
                              const double* dInn, long strideInn, const double* dR, long strideR) {
  const int N = c->N;
  MeasBuffers mb = meas_buffers(c);
  mb.H += (long)b0 * mb.strideH; mb.HT += (long)b0 * mb.strideHT;
  mb.inn += (long)b0 * mb.strideInn; mb.diagR += (long)b0 * mb.strideR;
  c->M = M; c->Mp = round_up16(M);
  EllBuffers e = c->ell;
  e.idx += (long)b0 * e.stride_idx(); e.val += (long)b0 * e.stride_val(); e.nc += b0; e.pw += b0; e.over += b0;
  c->lead_valid = false;
  // (XIVO_HIP_NO_COMPRESS: test hook for the branch very wide states take - the shape limit itself is N > ~2800 at M = 384)
  static const bool no_compress = getenv("XIVO_HIP_NO_COMPRESS") != nullptr;
  if (!meas_compress_fits(c->Mpmax, c->Np) || no_compress) {
    // the compression kernel's LDS lists do not fit this shape: every filter keeps its dense rows and takes the dense pipeline
    StageTimer st(c, ST_STACK, 0.0, "unpack_meas_kernel", 8.0 * nb * (3.0 * M * N + 4.0 * M));
    HIP_TRY((hipError_t)launch_meas_vectors(dInn, strideInn, dR, strideR, M, c->Mpmax, e, mb.inn, mb.strideInn, mb.diagR, mb.strideR, nb, c->stream));
    HIP_TRY((hipError_t)launch_unpack_meas(dH, strideH, ldh, nullptr, mb, M, c->Mpmax, N, c->Np, nb, c->stream));
    for (int b = b0; b < b0 + nb; ++b) { c->ell_over_h[b] = 1; c->ell_nc_h[b] = ELL_CW; c->ell_pw_h[b] = ELL_PW + 1; }
    c->dense_valid = true; c->dense_from_ell = true; c->ht_valid = true; c->mixed_row0 = -1; c->h_clean = false;
    return XIVO_HIP_OK;
  }
  {
    StageTimer st(c, ST_STACK, 0.0, "meas_compress_kernel", 8.0 * nb * ((double)M * N + 4.0 * M) + (double)nb * c->ell.pairs_max * ELL_W * 20.0);
    // clear up to the allocated row count so stale rows of a previous, larger M vanish
    HIP_TRY((hipError_t)launch_meas_compress(dH, strideH, ldh, dInn, strideInn, dR, strideR, M, N, c->Np, c->Mpmax, e, mb.inn,
                                             mb.strideInn, mb.diagR, mb.strideR, nb, c->stream,
                                             c->ell_flags_d ? c->ell_flags_d + 3 * (long)b0 : nullptr));
  }
  if (c->ell_flags_h) {
    HIP_TRY(hipStreamSynchronize(c->stream));     // kernel end = system-scope release: the mirrored flags are in host memory
    const int* f = c->ell_flags_h + 3 * (long)b0;
    for (int b = 0; b < nb; ++b) { c->ell_over_h[b0 + b] = f[3 * b]; c->ell_nc_h[b0 + b] = f[3 * b + 1]; c->ell_pw_h[b0 + b] = f[3 * b + 2]; }
  } else {
    HIP_TRY(hipMemcpyAsync(c->ell_over_h.data() + b0, e.over, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ell_nc_h.data() + b0, e.nc, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->ell_pw_h.data() + b0, e.pw, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  bool any_over = false;
  for (int b = b0; b < b0 + nb && !any_over; ++b) any_over = c->ell_over_h[b] != 0;
  if (debug_on()) fprintf(stderr, "xivo_hip: hand-over b0=%d nb=%d M=%d any_over=%d nc0=%d pw0=%d\n", b0, nb, M, (int)any_over, c->ell_nc_h[b0], c->ell_pw_h[b0]);
  if (any_over) HIP_TRY((hipError_t)launch_unpack_meas(dH, strideH, ldh, e.over, mb, M, c->Mpmax, N, c->Np, nb, c->stream));
  c->dense_valid = false; c->dense_from_ell = true; c->ht_valid = true;   // (ensure_dense rebuilds H and H^T together)
  c->mixed_row0 = -1; if (any_over) c->h_clean = false;
  return XIVO_HIP_OK;
}

int xivo_hip_set_measurements(xivo_hip_ctx* c, int b0, int nb, int M, const double* H, long strideH, int ldh,
                              const double* inn, long strideInn, const double* diagR, long strideR) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !H || !inn || !diagR || M <= 0 || M > c->Mmax || ldh < M) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  HIP_TRY(hipSetDevice(c->device));
  const int N = c->N;
  const size_t per = (size_t)M * N + 2 * (size_t)M;
  int rc = ensure_staging(c, (size_t)nb * per);
  if (rc) return rc;
  double* sH = c->staging;
  double* sInn = sH + (size_t)nb * M * N;
  double* sR = sInn + (size_t)nb * M;
  rc = h2d_packed(c, sH, H, nb, M, N, strideH, ldh);
  if (rc) return rc;
  rc = h2d_packed(c, sInn, inn, nb, M, 1, strideInn, M);
  if (rc) return rc;
  rc = h2d_packed(c, sR, diagR, nb, M, 1, strideR, M);
  if (rc) return rc;
  rc = stage_measurements(c, b0, nb, M, sH, (long)M * N, M, sInn, M, sR, M);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));   // host buffers are only borrowed for the call
  return XIVO_HIP_OK;
}

int xivo_hip_set_measurements_device(xivo_hip_ctx* c, int b0, int nb, int M, const double* dH, long strideH, int ldh,
                                     const double* dInn, long strideInn, const double* dR, long strideR) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !dH || !dInn || !dR || M <= 0 || M > c->Mmax || ldh < M || strideH < 0) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  HIP_TRY(hipSetDevice(c->device));
  return stage_measurements(c, b0, nb, M, dH, strideH, ldh, dInn, strideInn, dR, strideR);
}

// One pass of the update pipeline over filters [b0, b0 + B).
struct GateParams { int F; double R, thresh, mult; int min_inliers; };

// XIVO_HIP_FLAG_SYMMETRIC_FORM: gain and covariance in the symmetric "square-root" form. With S = L L^T and
// W = L^-1 (H P) (forward substitution only):  K (H P) = W^T W,  dx = K inn = W^T (L^-1 inn),  P+ = P - W^T W.
// This is the covariance the Joseph form of src/estimator.cpp:1276-1287 evaluates to for the optimal gain (the Joseph
// correction term vanishes identically), computed without the backward substitution, the residual G and the second
// N x N x M product; its rounding error grows with cond(L) = sqrt(cond(S)), not cond(S). Opt-in: the reference codes
// the Joseph form, which stays the default.
static int finish_symmetric(xivo_hip_ctx* c, const UpdatePlan& plan, int b0, int B, double* S, int lds, double* invD, double* PHT, double* K,
                            double* P, const double* inn, int Mp, int Np) {
  double* y = c->yvec + (long)b0 * c->Mpmax;
  {
    StageTimer st(c, ST_OTHER, 0.0, "fwd_vec_kernel");
    HIP_TRY((hipError_t)launch_fwd_vec(S, c->sS, lds, invD, c->sInvD, inn, c->Mpmax, y, c->Mpmax, Mp, B, c->stream));
  }
  {
    TrsmArgs a{}; a.LU = S; a.strideLU = c->sS; a.ldlu = lds; a.invD = invD; a.strideInvD = c->sInvD;
    a.PHT = PHT; a.stridePHT = c->sK; a.ldpht = Np; a.K = K; a.strideK = c->sK; a.ldk = Np;
    a.inn = inn; a.strideInn = c->Mpmax; a.err = c->err + (long)b0 * Np; a.strideErr = Np; a.Mp = Mp; a.Np = Np;
    a.batch = B; a.fwd_only = 1; a.y = y; a.strideY = c->Mpmax;
    // the solve kernel goes on to P+ = P - W^T W in place, W^T still in its registers (blocks exchanged through LDS)
    const bool p_here = plan.in_solve;
    if (p_here) { a.T = P; a.strideT = c->sP; a.ldt = Np; a.skip_status = c->status + b0; }
    char label[64]; trsm_kernel_label(Mp, label, sizeof(label), p_here ? 2 : 0);
    const double outs = 0.5 * Np * (Np + 1.0), Nf = c->N, Mf = c->M;
    StageTimer st(c, ST_TRSM, (1.0 * Mf * Mf * Nf + (p_here ? Nf * (Nf + 1.0) * Mf : 0.0)) * B, label,
                  8.0 * B * (0.5 * Mp * (Mp + 1) + Mp / 16 * 512.0 + (p_here ? 1.0 : 2.0) * Np * Mp + (p_here ? outs + (double)Np * Np : 0.0)));
    HIP_TRY((hipError_t)launch_trsm_f64(a, c->stream));
    if (p_here) return XIVO_HIP_OK;
  }
  // P+ = P - W^T W in place: the accumulators start at -P (every tile reads its part of P before it stores anything)
  // and the result is negated on the way out
  GemmExtra x; x.epi = EPI_RSUB_MAT; x.msub = P; x.sMsub = c->sP; x.ldmsub = Np; x.lower_only = 1;
  x.skip = c->status + b0;
  return gemm(c, ST_PNEW, B, Np, Np, K, c->sK, Np, K, c->sK, Np, Mp, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0, P, c->sP, Np, x);
}

// The factorisation, the gain, dx and the covariance update once P H^T and S are formed - shared by the sparse and the dense
// whitened pipelines (they differ in how H P and S are built, not behind them):
//   S = L L^T (gate folded into its prologue when `cg` is given)                 estimator.cpp:1266
//   in_solve : W = L^-1 (HP), K^T = L^-T W, dx, P+ = P - (W - D)^T (W + D) inside the solve kernel  estimator.cpp:1265-1287
//   else     : V^T, Y^T leave the (chunked / streamed) solve, P+ = P - V^T Y as one tiled symmetric product
static int finish_whitened(xivo_hip_ctx* c, const UpdatePlan& plan, int b0, int B, double* S, int lds, double* invD, double* PHT, double* K,
                           double* G, double* P, const double* inn, int Mp, int Np, const CholGateArgs* cg) {
  const double Nf = c->N, Mf = c->M;
  {
    CholArgs a{}; a.S = S; a.strideS = c->sS; a.lds = lds; a.Mp = Mp; a.invD = invD; a.strideInvD = c->sInvD;
    a.status = c->status + b0; a.batch = B; a.latency = plan.latency || plan.stream8;   // (the streamed solve reads the mirrored upper triangle)
    char clabel[64]; chol_kernel_label(Mp, B, clabel, sizeof(clabel));
    if (cg) { const size_t n = strlen(clabel); snprintf(clabel + n, sizeof(clabel) - n, "+gate"); }
    StageTimer st(c, ST_CHOL, Mf * Mf * Mf / 3.0 * B, clabel, 8.0 * B * ((double)Mp * (Mp + 1) + Mp / 16 * 512.0));
    HIP_TRY((hipError_t)launch_chol_f64(a, c->stream, cg));
  }
  {
    TrsmArgs a{}; a.LU = S; a.strideLU = c->sS; a.ldlu = lds; a.invD = invD; a.strideInvD = c->sInvD;
    a.PHT = PHT; a.stridePHT = c->sK; a.ldpht = Np; a.K = K; a.strideK = c->sK; a.ldk = Np;
    a.inn = inn; a.strideInn = c->Mpmax; a.err = c->err + (long)b0 * Np; a.strideErr = Np; a.Mp = Mp; a.Np = Np; a.batch = B;
    if (plan.in_solve) { a.T = P; a.strideT = c->sP; a.ldt = Np; a.joseph = 2; a.skip_status = c->status + b0; }
    else { a.Yout = G; a.strideY2 = c->sA; a.ldy2 = Np; a.latency = plan.latency; a.stream8 = plan.stream8 ? 1 : 0; a.out_f32 = plan.f32_whitened ? 1 : 0; }
    char label[64]; trsm_kernel_label(Mp, label, sizeof(label), plan.in_solve ? 4 : 5, plan.latency, plan.stream8);
    // seven block rows on a narrow state: the ten- / twelve-wave instantiation with W in registers (solve_fused.hip)
    const bool narrow = plan.in_solve && trsm_narrow_supported(Mp, Np);
    if (narrow) trsm_narrow_label(Mp, Np, label, sizeof(label));
    const double t_outs = 0.5 * Np * (Np + 1.0), t_outs_f = 0.5 * Nf * (Nf + 1.0);
    // algorithmic flops (true N, M): the two triangular solves (M^2 N each), the residual blocks of the whitened form
    // (2 * 16 * M * N) and, in the solve kernel, the symmetric N x N x M product (lower triangle). Algorithmic bytes: the
    // factor, P H^T once, P's lower triangle in, P out (the gain is not stored)
    StageTimer st(c, ST_TRSM, (2.0 * Mf * Mf * Nf + 32.0 * Mf * Nf + (plan.in_solve ? 2.0 * t_outs_f * Mf : 0.0)) * B, label,
                  8.0 * B * (0.5 * Mp * (Mp + 1) + Mp / 16 * 512.0 + (plan.in_solve ? 1.0 : 3.0) * Np * Mp + (plan.in_solve ? t_outs + (double)Np * Np : 0.0)));
    HIP_TRY((hipError_t)(narrow ? launch_trsm_narrow(a, c->stream) : launch_trsm_f64(a, c->stream)));
    if (plan.in_solve) return XIVO_HIP_OK;
  }
  // P+ = P - V^T Y in place (V^T in the K buffer, Y^T in the G buffer), lower triangle + mirror
  GemmExtra x; x.epi = EPI_RSUB_MAT; x.msub = P; x.sMsub = c->sP; x.ldmsub = Np; x.lower_only = 1; x.skip = c->status + b0;
  x.small_tiles = plan.latency;
  if (plan.f32_whitened) {   // XIVO_HIP_FLAG_FP32_WHITENED: both operands left the solve as float (Y^T at float 0, V^T at float Np Mp of G)
    x.fp32 = 1; x.a_f32 = 1; x.b_f32 = 1;
    const double* Vf = reinterpret_cast<const double*>(reinterpret_cast<const float*>(G) + (long)Np * Mp);
    return gemm(c, ST_PNEW, B, Np, Np, Vf, 2 * c->sA, Np, G, 2 * c->sA, Np, Mp, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0, P, c->sP, Np, x);
  }
  return gemm(c, ST_PNEW, B, Np, Np, K, c->sK, Np, G, c->sA, Np, Mp, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0, P, c->sP, Np, x);
}

// Sparse-H pipelines (ell.h): H P, S and T H^T skip the structural zeros of H; the factorisation, the gain and the
// N x N x M covariance products stay on the MFMA kernels.
//   FUSED            everything in one kernel per filter                                     fused_update.hip
//   otherwise        HP = H P (+ P H^T)            ell_mul<HP>                               estimator.cpp:1259
//                    S = (HP) H^T + R              ell_mul<S>                                estimator.cpp:1259-1263
//                    [MH gating]                   in the prologue of the factorisation / gate_ell    update.cpp:60-96
//                    then finish_whitened / finish_symmetric, or (SPARSE_TAIL)
//                    T = K (HP) - P, G = T H^T + K R, P+ = G K^T - T                          estimator.cpp:1276-1287 re-associated
static int update_sparse_range(xivo_hip_ctx* c, const UpdatePlan& plan, int b0, int B, const GateParams* gate) {
  const int Np = c->Np, Mp = c->Mp, ldh = c->Mpmax, lds = c->Mpmax;
  double* P = c->P + (long)b0 * c->sP;
  double* HP = c->HP + (long)b0 * c->sH;
  double* PHT = c->PHT + (long)b0 * c->sK;
  double* S = c->S + (long)b0 * c->sS;
  double* K = c->K + (long)b0 * c->sK;
  double* G = c->A + (long)b0 * c->sA;
  double* T = c->T + (long)b0 * c->sP;
  double* invD = c->invD + (long)b0 * c->sInvD;
  double* inn = c->inn + (long)b0 * c->Mpmax;
  double* diagR = c->diagR + (long)b0 * c->Mpmax;
  EllBuffers e = c->ell;
  e.idx += (long)b0 * e.stride_idx(); e.val += (long)b0 * e.stride_val(); e.nc += b0; e.pw += b0; e.over += b0;
  int nc_max = 0, pw_max = 1;
  for (int b = b0; b < b0 + B; ++b) { nc_max = std::max(nc_max, c->ell_nc_h[b]); pw_max = std::max(pw_max, c->ell_pw_h[b]); }
  // algorithmic flops are counted on the TRUE sizes N, M (the padded Np, Mp only size the launches and the bytes)
  const double Nf = c->N, Mf = c->M;
  const double nnz_flops = 2.0 * Mf * 21.0;   // per contiguous-index value: 21 structural non-zeros per row
  int rc;
  if (plan.route == ROUTE_FUSED) {
    // Round 6: the shapes a CU holds (TUM-VI 203 / 60, BASELINE config 2 150 / 100) take ONE kernel for the whole update -
    // P H^T, S, the gate, the factor, both substitutions and the covariance product stay in the registers and the LDS of the
    // workgroup that owns the filter; nothing but P, P+ and the compressed rows crosses HBM.
    FusedArgs a{};
    a.P = P; a.strideP = c->sP; a.ldp = Np; a.ell = e; a.inn = inn; a.strideInn = c->Mpmax; a.diagR = diagR; a.strideR = c->Mpmax;
    a.err = c->err + (long)b0 * Np; a.strideErr = Np; a.PHT = PHT; a.stridePHT = c->sK; a.ldpht = Np; a.status = c->status + b0;
    a.Np = Np; a.Mp = Mp; a.batch = B; a.pw = pw_max;
    if (gate) {
      a.gate = 1; a.F = gate->F; a.R = gate->R; a.thresh = gate->thresh; a.mult = gate->mult; a.min_inliers = gate->min_inliers;
      a.mask = c->mask + (long)b0 * gate->F; a.dist = c->dist + (long)b0 * gate->F;
      if (c->dense_valid) { a.H = c->H + (long)b0 * c->sH; a.strideH = c->sH; a.ldh = ldh; a.HT = c->HT + (long)b0 * c->sHT; a.strideHT = c->sHT; a.ldht = Np; }
      c->gate_sparse_last = 0;
    }
    char label[64]; fused_update_label(Mp, Np, pw_max, label, sizeof(label));
    const double t_outs_f = 0.5 * Nf * (Nf + 1.0);
    StageTimer st(c, ST_TRSM, (nnz_flops * (Nf + Mf) + Mf * Mf * Mf / 3.0 + 2.0 * Mf * Mf * Nf + 32.0 * Mf * Nf + 2.0 * t_outs_f * Mf) * B, label,
                  B * (16.0 * Np * Np + (Mp / 2) * ELL_W * 20.0));
    HIP_TRY((hipError_t)launch_fused_update(a, c->stream));
    return XIVO_HIP_OK;
  }
  // mixed stacking: rows [0, mr0) of H are the compressed in-state rows, rows [mr0, M) the dense OOS rows appended by
  // xivo_hip_oos_project (non-zero over the extrinsics + group columns only: src/oos.cpp:74-88). The in-state rows keep the
  // sparse walk below; the OOS block goes through two small MFMA products (rows padded to 16 from mr0 on).
  const int mr0 = c->mixed_row0;
  const int Mp_ell = mr0 >= 0 ? round_up16(mr0) : Mp;
  const int oos_pad = mr0 >= 0 ? round_up16(Mp - mr0) : 0;
  // the OOS rows are zero beyond the extrinsics and group columns (the mode clears and writes nothing else there): the two
  // products of the OOS block contract over the leading oos_k state columns only
  bool walk_tiled = false;
  const int oos_k = (mr0 >= 0 && c->have_layout) ? std::min(Np, round_up16(c->lay.group_begin + 6 * c->lay.n_groups)) : Np;
  {
    EllMulArgs a{}; a.ell = e; a.Src = P; a.strideSrc = c->sP; a.ldsrc = Np; a.out = PHT; a.strideOut = c->sK; a.ldo = Np;
    a.out2 = HP; a.strideOut2 = c->sH; a.ldo2 = ldh; a.X = Np; a.Mp = Mp_ell; a.batch = B; a.nc_max = nc_max; a.pw_max = pw_max; a.cols = Np;
    char label[64]; ell_kernel_label(ELL_HP, a, label, sizeof(label));
    walk_tiled = ell_uses_slab_form(a);   // (the same decision for ell<S> below: it depends on the shape and slot counts only)
    StageTimer st(c, ST_HP, nnz_flops * Nf * B, label, 8.0 * B * ((double)Np * Np + (double)Np * Mp));
    HIP_TRY((hipError_t)launch_ell_mul(ELL_HP, a, c->stream));
  }
  if (mr0 >= 0) {   // (H P)_oos = H_oos P, with its transpose into the P H^T columns behind the in-state ones
    const double* Hd = c->H + (long)b0 * c->sH + mr0;
    GemmExtra x; x.C2 = PHT + (long)mr0 * Np; x.sC2 = c->sK; x.ldc2 = Np;
    rc = gemm(c, ST_HP, B, oos_pad, Np, Hd, c->sH, ldh, P, c->sP, Np, oos_k, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0, HP + mr0, c->sH, ldh, x);
    if (rc) return rc;
  }
  // online-calibration stacking on the sparse pipeline: the calibration columns of H live in the leading dense block
  // L [Mp x LEAD_K] (stack_kernel): P H^T += P[:, 0:LEAD_K] L^T on the MFMA product
  const bool lead = c->lead_valid && mr0 < 0;
  const double* Ld = lead ? c->Hlead + (long)b0 * c->Mpmax * LEAD_K : nullptr;
  const long sLd = (long)c->Mpmax * LEAD_K;
  const int ldl = c->Mpmax;   // (stack_kernel lays the block out on the allocated row count)
  if (lead) {
    // (the tiled walk writes P H^T only: this product completes it in place and leaves H P as its transposed copy - the
    //  leading LEAD_K columns the S product below reads, or all of it where ell<S> takes the gather form, which reads H P)
    GemmExtra x; x.epi = EPI_ADD_MAT; x.msub = PHT; x.sMsub = c->sK; x.ldmsub = Np; x.C2 = HP; x.sC2 = c->sH; x.ldc2 = ldh;
    x.c2_rows = walk_tiled ? LEAD_K : 0;
    rc = gemm(c, ST_HP, B, Np, Mp, P, c->sP, Np, Ld, sLd, ldl, LEAD_K, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0, PHT, c->sK, Np, x);
    if (rc) return rc;
  }
  GateEllArgs ga{};
  if (gate) {
    ga.ell = e;
    ga.H = c->dense_valid ? c->H + (long)b0 * c->sH : nullptr; ga.strideH = c->sH; ga.ldh = ldh;
    ga.HT = c->dense_valid ? c->HT + (long)b0 * c->sHT : nullptr; ga.strideHT = c->sHT; ga.ldht = Np; ga.PHT = PHT;
    ga.HP = nullptr;   // H P [Mp x Np] has no reader behind this point (S is formed already, the solve reads P H^T)
    ga.inn = inn; ga.strideInn = c->Mpmax; ga.diagR = diagR; ga.strideR = c->Mpmax;
    ga.mask = c->mask + (long)b0 * gate->F; ga.dist = c->dist + (long)b0 * gate->F;
    ga.F = gate->F; ga.Np = Np; ga.batch = B;
    ga.S = S; ga.strideS = c->sS; ga.lds = lds; ga.Mp = Mp; ga.from_S = 1;   // distances from the diagonal blocks of S
    ga.R = gate->R; ga.thresh = gate->thresh; ga.mult = gate->mult; ga.min_inliers = gate->min_inliers;
  }
  int diag_done = 0;
  {
    EllMulArgs a{}; a.ell = e; a.Src = PHT; a.strideSrc = c->sK; a.ldsrc = Np; a.SrcAlt = HP; a.strideSrcAlt = c->sH; a.ldsrcAlt = ldh;
    a.out = S; a.strideOut = c->sS; a.ldo = lds; a.cols = Np;
    a.diagR = diagR; a.strideR = c->Mpmax; a.X = Mp; a.Mp = Mp_ell; a.batch = B; a.nc_max = nc_max; a.pw_max = pw_max;
    // the 2 x 2 diagonal blocks of S once more, compact (the T buffer is free until the solve): what the gate reads
    if (gate && mr0 < 0 && !lead && (long)2 * Mp <= c->sP) { a.diag_out = c->T + (long)b0 * c->sP; a.strideDiag = c->sP; a.diag_done = &diag_done; }
    char label[64]; ell_kernel_label(ELL_S, a, label, sizeof(label));
    StageTimer st(c, ST_S, nnz_flops * Mf * B, label, 8.0 * B * ((double)Np * Mp + (double)Mp * Mp));
    HIP_TRY((hipError_t)launch_ell_mul(ELL_S, a, c->stream));
  }
  if (mr0 >= 0) {   // the OOS x OOS block of S (the OOS x in-state block came out of the walk above: rows of S run over all M)
    const double* Hd = c->H + (long)b0 * c->sH + mr0;
    GemmExtra x; x.epi = EPI_ADD_DIAG; x.diag = diagR + mr0; x.sDiag = c->Mpmax; x.lower_only = 1;
    rc = gemm(c, ST_S, B, oos_pad, oos_pad, HP + mr0, c->sH, ldh, Hd, c->sH, ldh, oos_k, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0,
              S + mr0 + (long)mr0 * lds, c->sS, lds, x);
    if (rc) return rc;
  }
  if (lead) {
    // S += (H P)[:, 0:LEAD_K] L^T. The walk above left, in the lower triangle, S[i, j] = sum over the COMPRESSED columns k of
    // row j of (H P)[i, k] H[j, k] with the complete H P: what is missing is the same sum over row j's calibration columns
    // (the order of the operands matters - L (H P)^T is the transpose, and neither term is symmetric on its own)
    GemmExtra x; x.epi = EPI_ADD_MAT; x.msub = S; x.sMsub = c->sS; x.ldmsub = lds; x.lower_only = 1;
    rc = gemm(c, ST_S, B, Mp, Mp, HP, c->sH, ldh, Ld, sLd, ldl, LEAD_K, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0, S, c->sS, lds, x);
    if (rc) return rc;
  }
  if (gate) c->gate_sparse_last = 0;
  // With thousands of factors the gate rides in the prologue of the factorisation (chol_f64.hip, GATE): the distances come
  // from the compact diagonal blocks ell<S> just left, the rejected pairs are decoupled where the factor loads S - no gate
  // launch, no extra pass over S. (Few filters, dense copies of H alive, mixed stacking: the gate kernel.)
  CholGateArgs cg{};
  bool gate_folded = false;
  if (gate) {
    if (diag_done) { ga.Sdiag = c->T + (long)b0 * c->sP; ga.strideSdiag = c->sP; }
    gate_folded = diag_done && !c->dense_valid && mr0 < 0 && !plan.latency && chol_gate_supported(Mp, B);
    if (gate_folded) {
      cg.Sdiag = ga.Sdiag; cg.strideSdiag = ga.strideSdiag; cg.inn = inn; cg.strideInn = c->Mpmax; cg.diagR = diagR; cg.strideR = c->Mpmax;
      cg.ellval = e.val; cg.strideVal = e.stride_val(); cg.ell_w = ELL_W; cg.PHT = PHT; cg.stridePHT = c->sK; cg.ldpht = Np; cg.Np = Np;
      cg.mask = ga.mask; cg.dist = ga.dist; cg.F = gate->F; cg.R = gate->R; cg.thresh = gate->thresh; cg.mult = gate->mult;
      cg.min_inliers = gate->min_inliers;
    } else {
      StageTimer st(c, ST_GATE, 0.0, "gate_ell_kernel");
      HIP_TRY((hipError_t)launch_gate_ell(ga, c->stream));
    }
  }
  if (plan.route != ROUTE_SPARSE_TAIL && plan.route != ROUTE_SPARSE_SYMMETRIC)
    return finish_whitened(c, plan, b0, B, S, lds, invD, PHT, K, G, P, inn, Mp, Np, gate_folded ? &cg : nullptr);
  {
    CholArgs a{}; a.S = S; a.strideS = c->sS; a.lds = lds; a.Mp = Mp; a.invD = invD; a.strideInvD = c->sInvD;
    a.status = c->status + b0; a.batch = B; a.latency = 0;
    char clabel[64]; chol_kernel_label(Mp, B, clabel, sizeof(clabel));
    if (gate_folded) { const size_t n = strlen(clabel); snprintf(clabel + n, sizeof(clabel) - n, "+gate"); }
    StageTimer st(c, ST_CHOL, Mf * Mf * Mf / 3.0 * B, clabel, 8.0 * B * ((double)Mp * (Mp + 1) + Mp / 16 * 512.0));
    HIP_TRY((hipError_t)launch_chol_f64(a, c->stream, gate_folded ? &cg : nullptr));
  }
  if (plan.route == ROUTE_SPARSE_SYMMETRIC) return finish_symmetric(c, plan, b0, B, S, lds, invD, PHT, K, P, inn, Mp, Np);
  // ---- SPARSE_TAIL (XIVO_HIP_FLAG_STANDALONE_TAIL): K^T = S^-1 (HP), dx; T = K (HP) - P; G = T H^T + K R; P+ = G K^T - T
  const bool t_here = trsm_forms_T(Mp, Np);      // the solve forms T on the gain still in its registers
  {
    TrsmArgs a{}; a.LU = S; a.strideLU = c->sS; a.ldlu = lds; a.invD = invD; a.strideInvD = c->sInvD;
    a.PHT = PHT; a.stridePHT = c->sK; a.ldpht = Np; a.K = K; a.strideK = c->sK; a.ldk = Np;
    a.inn = inn; a.strideInn = c->Mpmax; a.err = c->err + (long)b0 * Np; a.strideErr = Np; a.Mp = Mp; a.Np = Np; a.batch = B;
    if (t_here) { a.T = T; a.strideT = c->sP; a.ldt = Np; a.Pm = P; a.stridePm = c->sP; a.ldpm = Np; }
    char label[64]; trsm_kernel_label(Mp, label, sizeof(label), t_here ? 1 : 0);
    const double t_outs = 0.5 * Np * (Np + 1.0), t_outs_f = 0.5 * Nf * (Nf + 1.0);
    StageTimer st(c, ST_TRSM, (2.0 * Mf * Mf * Nf + (t_here ? 2.0 * t_outs_f * Mf : 0.0)) * B, label,
                  8.0 * B * (0.5 * Mp * (Mp + 1) + Mp / 16 * 512.0 + 2.0 * Np * Mp + (t_here ? t_outs + (double)Np * Np : 0.0)));
    HIP_TRY((hipError_t)launch_trsm_f64(a, c->stream));
  }
  if (!t_here) {  // T = K (HP) - P = (HP)^T S^-1 (HP) - P: symmetric up to the rounding of the solve, so the lower
                  // triangle is computed and mirrored
    GemmExtra x; x.epi = EPI_SUB_MAT; x.msub = P; x.sMsub = c->sP; x.ldmsub = Np; x.lower_only = 1;
    rc = gemm(c, ST_AP, B, Np, Np, K, c->sK, Np, PHT, c->sK, Np, Mp, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0, T, c->sP, Np, x);
    if (rc) return rc;
  }
  {  // G = T H^T + K diag(R)   [Np x Mp, in the A buffer]
    EllMulArgs a{}; a.ell = e; a.Src = T; a.strideSrc = c->sP; a.ldsrc = Np; a.out = G; a.strideOut = c->sA; a.ldo = Np;
    a.diagR = diagR; a.strideR = c->Mpmax; a.K = K; a.strideK = c->sK; a.ldk = Np; a.X = Np; a.Mp = Mp; a.batch = B; a.nc_max = nc_max; a.pw_max = pw_max; a.cols = Np;
    char label[64]; ell_kernel_label(ELL_G, a, label, sizeof(label));
    StageTimer st(c, ST_KH, nnz_flops * Np * B, label, 8.0 * B * ((double)Np * Np + 2.0 * Np * Mp));
    HIP_TRY((hipError_t)launch_ell_mul(ELL_G, a, c->stream));
  }
  if (pnew_reg_supported(Mp, Np)) {
    // P+ = G K^T - T, all fp64: rows of G in registers, blocks of K through LDS, one workgroup per filter
    PnewRegArgs a{}; a.G = G; a.strideG = c->sA; a.ldg = Np; a.K = K; a.strideK = c->sK; a.ldk = Np;
    a.T = T; a.strideT = c->sP; a.ldt = Np; a.P = P; a.strideP = c->sP; a.ldp = Np;
    a.skip_status = c->status + b0; a.Mp = Mp; a.Np = Np; a.batch = B;
    char label[64]; pnew_reg_kernel_label(Mp, label, sizeof(label));
    const double outs = 0.5 * Np * (Np + 1.0);
    StageTimer st(c, ST_PNEW, 2.0 * outs * Mp * B, label, 8.0 * B * (2.0 * Np * Mp + outs + (double)Np * Np));
    HIP_TRY((hipError_t)launch_pnew_reg_f64(a, c->stream));
    return XIVO_HIP_OK;
  }
  // P+ = G K^T - T   (lower triangle + mirror)
  GemmExtra x; x.epi = EPI_SUB_MAT; x.msub = T; x.sMsub = c->sP; x.ldmsub = Np; x.lower_only = 1;
  x.skip = c->status + b0;   // S not positive definite: P of that filter stays the prior (reported through xivo_hip_get_status)
  return gemm(c, ST_PNEW, B, Np, Np, G, c->sA, Np, K, c->sK, Np, Mp, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0, P, c->sP, Np, x);
}

static int update_joseph_range_impl(xivo_hip_ctx* c, int b0, int B, const GateParams* gate);

// One pass of the update over filters [b0, b0 + B), then the device answer to a filter whose S the un-pivoted Cholesky
// could not factor: Eigen's diagonally pivoted L D L^T (what src/estimator.cpp:1266 runs for EVERY filter) and the
// as-coded Joseph update, on exactly those filters (ldlt_fallback.hip). Every pipeline leaves the covariance of such a
// filter untouched and its status set, so the fallback starts from the prior.
static int update_joseph_range(xivo_hip_ctx* c, int b0, int B, const GateParams* gate = nullptr) {
  int rc = update_joseph_range_impl(c, b0, B, gate);
  if (rc || (c->flags & XIVO_HIP_FLAG_NO_LDLT_FALLBACK)) {   // no fallback launch: clear the flags of this call here
    HIP_TRY(hipMemsetAsync(c->ldlt_used + b0, 0, (size_t)B * sizeof(int), c->stream));
    return rc;
  }
  // (the fallback kernel writes ldlt_used of EVERY filter of the range: 0 where the Cholesky succeeded, 1 where it stepped in)
  LdltFallbackArgs a{};
  a.status = c->status + b0; a.used = c->ldlt_used + b0;
  a.ell = c->ell; a.ell.idx += (long)b0 * a.ell.stride_idx(); a.ell.val += (long)b0 * a.ell.stride_val();
  a.ell.nc += b0; a.ell.pw += b0; a.ell.over += b0;
  a.H = c->H + (long)b0 * c->sH; a.strideH = c->sH; a.ldh = c->Mpmax; a.use_dense = c->last_path == 0 ? 1 : 0;
  a.mixed_row0 = c->last_path == 1 ? c->mixed_row0 : -1;
  if (c->last_path == 1 && c->lead_valid) { a.lead = c->Hlead + (long)b0 * c->Mpmax * LEAD_K; a.strideLead = (long)c->Mpmax * LEAD_K; a.ldlead = c->Mpmax; a.lead_k = LEAD_K; }
  a.PHT = c->PHT + (long)b0 * c->sK; a.stridePHT = c->sK; a.ldpht = c->Np;
  a.S = c->S + (long)b0 * c->sS; a.strideS = c->sS; a.lds = c->Mpmax;
  a.K = c->K + (long)b0 * c->sK; a.strideK = c->sK; a.ldk = c->Np;
  a.A = c->A + (long)b0 * c->sA; a.strideA = c->sA; a.lda = c->Np;
  a.T = c->T + (long)b0 * c->sP; a.strideT = c->sP; a.ldt = c->Np;
  a.P = c->P + (long)b0 * c->sP; a.strideP = c->sP; a.ldp = c->Np;
  a.inn = c->inn + (long)b0 * c->Mpmax; a.strideInn = c->Mpmax; a.diagR = c->diagR + (long)b0 * c->Mpmax; a.strideR = c->Mpmax;
  a.err = c->err + (long)b0 * c->Np; a.strideErr = c->Np;
  a.N = c->N; a.M = c->M; a.batch = B;
  StageTimer st(c, ST_OTHER, 0.0, "ldlt_fallback_kernel");
  HIP_TRY((hipError_t)launch_ldlt_fallback(a, c->stream));
  return XIVO_HIP_OK;
}

static int update_joseph_range_impl(xivo_hip_ctx* c, int b0, int B, const GateParams* gate) {
  const int Np = c->Np, Mp = c->Mp, ldh = c->Mpmax, lds = c->Mpmax;
  const UpdatePlan plan = plan_update(c, b0, B, gate != nullptr);
  c->last_path = plan.sparse ? 1 : 0;
  c->last_route = plan.route;
  if (plan.sparse) return update_sparse_range(c, plan, b0, B, gate);
  const double* H = c->H + (long)b0 * c->sH;
  const double* HT = c->HT + (long)b0 * c->sHT;
  double* P = c->P + (long)b0 * c->sP;
  double* HP = c->HP + (long)b0 * c->sH;
  double* PHT = c->PHT + (long)b0 * c->sK;
  double* S = c->S + (long)b0 * c->sS;
  double* K = c->K + (long)b0 * c->sK;
  double* A = c->A + (long)b0 * c->sP;
  double* T = c->T + (long)b0 * c->sP;
  double* invD = c->invD + (long)b0 * c->sInvD;
  const double* inn = c->inn + (long)b0 * c->Mpmax;
  const double* diagR = c->diagR + (long)b0 * c->Mpmax;
  int rc = ensure_dense(c);   // (mixed stacking / a leading block: the in-state rows are rebuilt densely next to the rows already in place)
  if (rc) return rc;
  {  // HP = H * P and its transpose PH^T (estimator.cpp:1259 first product; P symmetric => B operand = P rows)
    GemmExtra x; x.C2 = PHT; x.sC2 = c->sK; x.ldc2 = Np;
    rc = gemm(c, ST_HP, B, Mp, Np, H, c->sH, ldh, P, c->sP, Np, Np, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0,
              HP, c->sH, ldh, x);
    if (rc) return rc;
  }
  if (gate) {  // Estimator::MHGating on the rows just multiplied (update.cpp:60-96): S_f = (HP)_f H_f^T + R
    rc = ensure_HT(c);   // the gate reads (and neutralises) the transposed rows
    if (rc) return rc;
    GateDenseArgs a{};
    a.H = H; a.strideH = c->sH; a.ldh = ldh; a.HP = HP; a.strideHP = c->sH; a.ldhp = ldh;
    a.Hw = c->H + (long)b0 * c->sH; a.HTw = c->HT + (long)b0 * c->sHT; a.strideHT = c->sHT; a.ldht = Np;
    a.HPw = HP; a.PHTw = PHT; a.PHTr = PHT;
    a.inn = c->inn + (long)b0 * c->Mpmax; a.strideInn = c->Mpmax; a.diagR = c->diagR + (long)b0 * c->Mpmax;
    a.strideR = c->Mpmax; a.mask = c->mask + (long)b0 * gate->F; a.dist = c->dist + (long)b0 * gate->F;
    a.F = gate->F; a.Np = Np; a.batch = B;
    a.R = gate->R; a.thresh = gate->thresh; a.mult = gate->mult; a.min_inliers = gate->min_inliers;
    a.ell = c->ell; a.have_ell = 0;
    StageTimer st(c, ST_GATE, 0.0, "gate_dense_kernel");
    c->gate_sparse_last = 0;
    HIP_TRY((hipError_t)launch_gate_dense(a, c->stream));
  }
  {  // S = HP * H^T + diag(R)  (estimator.cpp:1259-1263); lower triangle + mirror
    GemmExtra x; x.epi = EPI_ADD_DIAG; x.diag = diagR; x.sDiag = c->Mpmax; x.lower_only = 1;
    rc = gemm(c, ST_S, B, Mp, Mp, HP, c->sH, ldh, H, c->sH, ldh, Np, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0,
              S, c->sS, lds, x);
    if (rc) return rc;
  }
  if (plan.route == ROUTE_DENSE_WHITENED)      // an H without XIVO's row structure: everything behind S as on the sparse pipeline
    return finish_whitened(c, plan, b0, B, S, lds, invD, PHT, K, c->A + (long)b0 * c->sA, P, inn, Mp, Np, nullptr);
  {  // S = L L^T
    CholArgs a{}; a.S = S; a.strideS = c->sS; a.lds = lds; a.Mp = Mp; a.invD = invD; a.strideInvD = c->sInvD;
    a.status = c->status + b0; a.batch = B; a.latency = 0;
    char clabel[64]; chol_kernel_label(Mp, B, clabel, sizeof(clabel));
    StageTimer st(c, ST_CHOL, (double)Mp * Mp * Mp / 3.0 * B, clabel, 8.0 * B * ((double)Mp * (Mp + 1) + Mp / 16 * 512.0));
    HIP_TRY((hipError_t)launch_chol_f64(a, c->stream));
  }
  if (plan.route == ROUTE_DENSE_SYMMETRIC) return finish_symmetric(c, plan, b0, B, S, lds, invD, PHT, K, P, inn, Mp, Np);
  // ---- DENSE_ASCODED (XIVO_HIP_FLAG_DENSE_H): the products of estimator.cpp:1265-1287 as they are written
  {  // K^T = S^-1 HP ; dx = K inn  (estimator.cpp:1265-1267)
    TrsmArgs a{}; a.LU = S; a.strideLU = c->sS; a.ldlu = lds; a.invD = invD; a.strideInvD = c->sInvD;
    a.PHT = PHT; a.stridePHT = c->sK; a.ldpht = Np; a.K = K; a.strideK = c->sK; a.ldk = Np;
    a.inn = inn; a.strideInn = c->Mpmax; a.err = c->err + (long)b0 * Np; a.strideErr = Np; a.Mp = Mp; a.Np = Np;
    a.batch = B;
    char label[64]; trsm_kernel_label(Mp, label, sizeof(label), 0);
    StageTimer st(c, ST_TRSM, 2.0 * Mp * Mp * Np * B, label, 8.0 * B * (0.5 * Mp * (Mp + 1) + Mp / 16 * 512.0 + 2.0 * Np * Mp));
    HIP_TRY((hipError_t)launch_trsm_f64(a, c->stream));
  }
  rc = ensure_HT(c);
  if (rc) return rc;
  {  // A = K * H - I  (estimator.cpp:1276-1279)
    GemmExtra x; x.epi = EPI_SUB_IDENT;
    rc = gemm(c, ST_KH, B, Np, Np, K, c->sK, Np, HT, c->sHT, Np, Mp, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0,
              A, c->sP, Np, x);
    if (rc) return rc;
  }
  {  // T = A * P = K * (HP) - P  (estimator.cpp:1280, left product; distributes over the already
     // formed HP, 2MN^2 instead of 2N^3 flops, same value up to rounding)
    GemmExtra x; x.epi = EPI_SUB_MAT; x.msub = P; x.sMsub = c->sP; x.ldmsub = Np;
    rc = gemm(c, ST_AP, B, Np, Np, K, c->sK, Np, PHT, c->sK, Np, Mp, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0,
              T, c->sP, Np, x);
    if (rc) return rc;
  }
  {  // P = T * A^T + K diag(R) K^T  (estimator.cpp:1280-1287, fused; lower triangle + mirror)
    GemmExtra x; x.lower_only = 1;
    x.skip = c->status + b0;
    rc = gemm(c, ST_PNEW, B, Np, Np, T, c->sP, Np, A, c->sP, Np, Np, K, c->sK, Np, K, c->sK, Np, Mp, diagR,
              c->Mpmax, P, c->sP, Np, x);
  }
  return rc;
}

int xivo_hip_update_joseph(xivo_hip_ctx* c, int B) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || B <= 0 || B > c->Bmax || c->Mp <= 0) return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  // Filters are independent, so the batch is walked in chunks whose intermediates
  // (HP, PH^T, S, K, A, T: ~2.7 MB per filter at N=250/M=160) stay resident in the
  // 256 MiB Infinity Cache between consecutive kernels instead of round-tripping HBM.
  const int chunk = c->chunk > 0 ? c->chunk : B;
  c->call_batch = chunk < B ? B : 0;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    int rc = update_joseph_range(c, b0, nb);
    if (rc) { c->call_batch = 0; return rc; }
  }
  c->call_batch = 0;
  return XIVO_HIP_OK;
}

int xivo_hip_update_dense_gated(xivo_hip_ctx* c, int B, int F, double R, double mh_thresh, double mh_mult,
                                int min_inliers) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || B <= 0 || B > c->Bmax || c->Mp <= 0 || F <= 0 || 2 * F > c->M) return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  int rc = ensure_gate_buffers(c, F);
  if (rc) return rc;
  GateParams gp{F, R, mh_thresh, mh_mult, min_inliers};
  // Estimator::OutlierRejection only gates when F > min_required_inliers_ (src/manager.cpp:635)
  const GateParams* g = F > min_inliers ? &gp : nullptr;
  const int chunk = c->chunk > 0 ? c->chunk : B;
  c->call_batch = chunk < B ? B : 0;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    rc = update_joseph_range(c, b0, nb, g);
    if (rc) { c->call_batch = 0; return rc; }
  }
  c->call_batch = 0;
  return XIVO_HIP_OK;
}

int xivo_hip_last_path(xivo_hip_ctx* c) { return c ? c->last_path : -1; }
int xivo_hip_last_route(xivo_hip_ctx* c) { return c ? c->last_route : -1; }
const char* xivo_hip_route_name(int route) { return route >= 0 && route < ROUTE_COUNT ? kRouteNames[route] : ""; }

double xivo_hip_stage_bytes(xivo_hip_ctx* c, int stage) {
  return (c && stage >= 0 && stage < ST_COUNT) ? c->stage_bytes[stage] : 0.0;
}

const char* xivo_hip_stage_kernel(xivo_hip_ctx* c, int stage) {
  return (c && stage >= 0 && stage < ST_COUNT) ? c->stage_kernel[stage] : "";
}

int xivo_hip_get_gate(xivo_hip_ctx* c, int B, int F, unsigned char* mask_out, double* dist_out) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || B <= 0 || B > c->Bmax || F <= 0 || !c->mask) return XIVO_HIP_ERR_INVALID;
  // the dense gate packs [B][F]; the layout-faithful gate (xivo_hip_mh_gate / filter_update) strides by Fmax
  const size_t ld = c->gate_sparse_last ? (size_t)c->Fmax : (size_t)F;
  if (c->gate_sparse_last && F != c->F) return XIVO_HIP_ERR_INVALID;
  if (mask_out) { int rc = d2h_rows(c, mask_out, F, c->mask, ld, F, B); if (rc) return rc; }
  if (dist_out) {
    int rc = d2h_rows(c, dist_out, F * sizeof(double), c->dist, ld * sizeof(double), F * sizeof(double), B);
    if (rc) return rc;
  }
  return XIVO_HIP_OK;
}

int xivo_hip_get_err(xivo_hip_ctx* c, int b0, int nb, double* err, long stride) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !err || stride < c->N) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  return d2h_rows(c, err, (size_t)stride * sizeof(double), c->err + (long)b0 * c->Np, (size_t)c->Np * sizeof(double),
                  (size_t)c->N * sizeof(double), nb);
}

int xivo_hip_get_ldlt_used(xivo_hip_ctx* c, int b0, int nb, int* used) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !used) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  HIP_TRY(hipMemcpyAsync(used, c->ldlt_used + b0, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XIVO_HIP_OK;
}

int xivo_hip_get_status(xivo_hip_ctx* c, int b0, int nb, int* status) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !status) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  HIP_TRY(hipMemcpyAsync(status, c->status + b0, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int i = 0; i < nb; ++i) if (status[i]) return XIVO_HIP_ERR_NOT_SPD;
  return XIVO_HIP_OK;
}

// ------------------------------------------------------------------ one-filter plumbing call
// Estimator::UpdateJosephForm() as the reference calls it (src/update.cpp:141, :332): members in host memory in, members in
// host memory out, ONE call, ONE host synchronisation. See include/xivo_hip.h.
// Row-pair compression of ONE dense H_ on the host: the arithmetic-free format conversion meas_compress_kernel does for a
// batch (ell_kernels.hip - same lists, same common-column rule, same slot order, so the rows are those the device would have
// built, bit for bit), done while the matrix is staged: the host has to touch every byte of H_ once anyway, and the
// compressed rows are 1/7 of it. Returns over (1: the rows do not fit the compressed form).
static int host_compress(xivo_hip_ctx::HostCompressScratch& sc, const double* H, int ldh, int M, int N, int pairs_clear, int* idx,
                         double* val, int* nc_out, int* pw_out) {
  const int pairs = (M + 1) / 2;
  sc.cnt.assign(pairs_clear, 0); sc.occ.assign(N, 0); sc.cslot.assign(N, 0);
  sc.n.resize((size_t)pairs_clear * ELL_W); sc.v.resize((size_t)pairs_clear * ELL_W * 2);
  int* cnt = sc.cnt.data(); int* occ = sc.occ.data(); int* cslot = sc.cslot.data();
  int* ln = sc.n.data(); double* lv = sc.v.data();
  const int Me = M & ~1;                            // rows covered by complete pairs
  for (int n = 0; n < N; ++n) {
    const double* col = H + (size_t)n * ldh;
    const uint64_t* cb = reinterpret_cast<const uint64_t*>(col);
    int m = 0;
    for (; m + 8 <= Me; m += 8) {                   // four pairs at a time: all-zero runs (most of H_) cost one test
      const uint64_t any = cb[m] | cb[m + 1] | cb[m + 2] | cb[m + 3] | cb[m + 4] | cb[m + 5] | cb[m + 6] | cb[m + 7];
      if ((any << 1) == 0) continue;                // +0.0 / -0.0 only
      for (int q = m; q < m + 8; q += 2) {
        const double v0 = col[q], v1 = col[q + 1];
        if (v0 != 0.0 || v1 != 0.0) {
          const int p = q >> 1;
          if (cnt[p] < ELL_W) { ln[p * ELL_W + cnt[p]] = n; lv[2 * (p * ELL_W + cnt[p])] = v0; lv[2 * (p * ELL_W + cnt[p]) + 1] = v1; }
          ++cnt[p]; ++occ[n];
        }
      }
    }
    for (; m < M; m += 2) {
      const double v0 = col[m], v1 = m + 1 < M ? col[m + 1] : 0.0;
      if (v0 != 0.0 || v1 != 0.0) {
        const int p = m >> 1;
        if (cnt[p] < ELL_W) { ln[p * ELL_W + cnt[p]] = n; lv[2 * (p * ELL_W + cnt[p])] = v0; lv[2 * (p * ELL_W + cnt[p]) + 1] = v1; }
        ++cnt[p]; ++occ[n];
      }
    }
  }
  int ne = 0;
  for (int p = 0; p < pairs; ++p) ne += cnt[p] > 0;
  int ccols[ELL_CW] = {0};
  int flagged = 0;
  for (int n = 0; n < N; ++n) {                     // columns used by more than half of the non-empty pairs, ascending
    if (ne > 0 && 2 * occ[n] > ne) {
      if (flagged < ELL_CW) { cslot[n] = flagged + 1; ccols[flagged] = n; }
      ++flagged;
    }
  }
  const int nc = flagged < ELL_CW ? flagged : ELL_CW;
  int pw = 0, over = 0;
  for (int p = 0; p < pairs_clear; ++p) {
    int* pi = idx + (size_t)p * ELL_W;
    double* pv = val + (size_t)p * ELL_W * 2;
    for (int t = 0; t < ELL_W; ++t) { pi[t] = t < nc ? ccols[t] : 0; pv[2 * t] = 0.0; pv[2 * t + 1] = 0.0; }
    int pos = 0;
    const int walk = cnt[p] < ELL_W ? cnt[p] : ELL_W;
    for (int k = 0; k < walk; ++k) {
      const int n = ln[p * ELL_W + k];
      const double v0 = lv[2 * (p * ELL_W + k)], v1 = lv[2 * (p * ELL_W + k) + 1];
      const int cs = cslot[n];
      if (cs) { pv[2 * (cs - 1)] = v0; pv[2 * (cs - 1) + 1] = v1; }
      else {
        if (pos < ELL_PW) { pi[ELL_CW + pos] = n; pv[2 * (ELL_CW + pos)] = v0; pv[2 * (ELL_CW + pos) + 1] = v1; }
        ++pos;
      }
    }
    if (cnt[p] > ELL_W) pos = ELL_PW + 1;           // more than 28 non-zero columns cannot fit
    if (pos > ELL_PW) over = 1;
    if (pos > pw) pw = pos;
  }
  *nc_out = nc; *pw_out = pw;
  return over;
}

// test hook (no device, no context): the host-side row compression on its own, for the CPU test that pins it to the format
// of ell.h / meas_compress_kernel. idx [pairs_clear][28], val [pairs_clear][28][2]; returns over.
int xivo_hip_selftest_fused_tiles(int column_blocks, int* per_simd) { return fused_tiles_selftest(column_blocks, per_simd); }

int xivo_hip_selftest_host_compress(const double* H, int ldh, int M, int N, int pairs_clear, int* idx, double* val, int* nc, int* pw) {
  if (!H || !idx || !val || !nc || !pw || M <= 0 || N <= 0 || ldh < M || 2 * pairs_clear < M) return XIVO_HIP_ERR_INVALID;
  xivo_hip_ctx::HostCompressScratch sc;
  return host_compress(sc, H, ldh, M, N, pairs_clear, idx, val, nc, pw);
}

int xivo_hip_update_joseph_host(xivo_hip_ctx* c, int b, int M, const double* H, int ldh, const double* inn,
                                const double* diagR, double* P, int ldp, double* err_out, unsigned mode) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  const bool p_up = !(mode & XIVO_HIP_HOST_P_RESIDENT), p_down = !(mode & XIVO_HIP_HOST_KEEP_P);
  if (bad_range(c, b, 1) || !H || !inn || !diagR || !err_out || M <= 0 || M > c->Mmax || ldh < M ||
      ((p_up || p_down) && (!P || ldp < c->N)))
    return XIVO_HIP_ERR_INVALID;
  const int N = c->N, Np = c->Np, pairs_clear = c->Mpmax / 2;
  // the staged block: compressed rows | inn | diagR | flags | P in | P out | err | status
  auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
  const size_t o_idx = 0, o_val = al(o_idx + (size_t)pairs_clear * ELL_W * sizeof(int)),
               o_inn = al(o_val + (size_t)pairs_clear * ELL_W * 2 * sizeof(double)), o_R = al(o_inn + (size_t)c->Mpmax * sizeof(double)),
               o_flags = al(o_R + (size_t)c->Mpmax * sizeof(double)), o_Pin = al(o_flags + 4 * sizeof(int)),
               o_Pout = al(o_Pin + (size_t)N * N * sizeof(double)), o_err = al(o_Pout + (size_t)N * N * sizeof(double)),
               o_st = al(o_err + (size_t)N * sizeof(double)), total = al(o_st + 4 * sizeof(int));
  if (!c->pin_h) {
    if (hipHostMalloc(reinterpret_cast<void**>(&c->pin_h), total, hipHostMallocMapped) != hipSuccess) { c->pin_h = nullptr; (void)hipGetLastError(); return XIVO_HIP_ERR_NOMEM; }
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->pin_d), c->pin_h, 0) != hipSuccess || !c->pin_d) {
      hipHostFree(c->pin_h); c->pin_h = nullptr; c->pin_d = nullptr; (void)hipGetLastError(); return XIVO_HIP_ERR_HIP;
    }
    c->pin_bytes = total;
  }
  // the row-pair compressed rows, built while H_ is staged; an H_ that does not fit them (dense rows, stacked OOS rows) or a
  // context pinned to the dense / fp32 pipelines takes the general entry points - same results, more crossings
  int nc = 0, pw = 0, over = 1;
  const bool want_ell = !(c->flags & XIVO_HIP_FLAG_DENSE_H) && meas_compress_fits(c->Mpmax, Np);
  if (want_ell)
    over = host_compress(c->hc, H, ldh, M, N, pairs_clear, reinterpret_cast<int*>(c->pin_h + o_idx),
                         reinterpret_cast<double*>(c->pin_h + o_val), &nc, &pw);
  if (over) {
    int rc = XIVO_HIP_OK;
    if (p_up) rc = xivo_hip_upload_P(c, b, 1, P, (long)ldp * N, ldp);
    if (!rc) rc = xivo_hip_set_measurements(c, b, 1, M, H, (long)ldh * N, ldh, inn, M, diagR, M);
    if (!rc) rc = update_joseph_range(c, b, 1);
    if (rc) return rc;
    int st = 0;
    rc = xivo_hip_get_status(c, b, 1, &st);
    if (rc) return rc;
    rc = xivo_hip_get_err(c, b, 1, err_out, N);
    if (!rc && p_down) rc = xivo_hip_download_P(c, b, 1, P, (long)ldp * N, ldp);
    return rc;
  }
  double* s_inn = reinterpret_cast<double*>(c->pin_h + o_inn);
  double* s_R = reinterpret_cast<double*>(c->pin_h + o_R);
  for (int m = 0; m < c->Mpmax; ++m) { s_inn[m] = m < M ? inn[m] : 0.0; s_R[m] = m < M ? diagR[m] : 1.0; }
  int* s_flags = reinterpret_cast<int*>(c->pin_h + o_flags);
  s_flags[0] = nc; s_flags[1] = pw; s_flags[2] = 0;
  // P_ crosses through the context's page-locked block: one host copy each way (~9 us per 500 KB), the boundary kernels
  // read / write the block over PCIe. (Page-locking the caller's own P_ in place - hipHostRegister - saved 18 us per call
  // and was dropped: with large pageable copies elsewhere in the process the runtime's own pinning of recycled heap
  // addresses left the device faulting on the registered pages, scripts/register_stress.py)
  DropinInArgs ia{};
  if (p_up) {
    double* sp = reinterpret_cast<double*>(c->pin_h + o_Pin);   // (the lower triangle is all the device reads: p_unpack_device.h)
    for (int j = 0; j < N; ++j) memcpy(sp + (size_t)j * N + j, P + (size_t)j * ldp + j, (size_t)(N - j) * sizeof(double));
    ia.Psrc = reinterpret_cast<const double*>(c->pin_d + o_Pin); ia.ldps = N;
  }
  ia.P = c->P + (long)b * c->sP; ia.N = N; ia.Np = Np; ia.ldp = Np;
  ia.block = c->pin_d; ia.off_idx = (int)o_idx; ia.off_val = (int)o_val; ia.off_inn = (int)o_inn; ia.off_R = (int)o_R; ia.off_flags = (int)o_flags;
  ia.pairs_clear = pairs_clear; ia.Mpmax = c->Mpmax;
  ia.idx = c->ell.idx + (long)b * c->ell.stride_idx(); ia.val = c->ell.val + (long)b * c->ell.stride_val();
  ia.inn = c->inn + (long)b * c->Mpmax; ia.diagR = c->diagR + (long)b * c->Mpmax;
  ia.nc = c->ell.nc + b; ia.pw = c->ell.pw + b; ia.over = c->ell.over + b;
  {
    StageTimer st(c, ST_STACK, 0.0, "dropin_in_kernel", (p_up ? 8.0 * N * N : 0.0) + (double)pairs_clear * ELL_W * 20.0 + 16.0 * c->Mpmax);
    HIP_TRY((hipError_t)launch_dropin_in(ia, c->stream));
  }
  // what stage_measurements leaves behind for the pipeline
  c->M = M; c->Mp = round_up16(M);
  c->ell_over_h[b] = 0; c->ell_nc_h[b] = nc; c->ell_pw_h[b] = pw;
  c->dense_valid = false; c->dense_from_ell = true; c->ht_valid = true; c->mixed_row0 = -1;
  // (from here on kernels that read the context's pinned block may be in flight: an early return drains the stream first,
  //  the next call overwrites that block)
  int rc = update_joseph_range(c, b, 1);
  if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
  DropinOutArgs oa{};
  oa.P = c->P + (long)b * c->sP; oa.N = N; oa.ldp = Np;
  if (p_down) { oa.Pdst = reinterpret_cast<double*>(c->pin_d + o_Pout); oa.ldpd = N; }
  oa.err = c->err + (long)b * Np; oa.err_dst = reinterpret_cast<double*>(c->pin_d + o_err);
  oa.status = c->status + b; oa.ldlt_used = c->ldlt_used + b; oa.flags_dst = reinterpret_cast<int*>(c->pin_d + o_st);
  {
    StageTimer st(c, ST_OTHER, 0.0, "dropin_out_kernel", (p_down ? 8.0 * N * N : 0.0) + 8.0 * N);
    if (launch_dropin_out(oa, c->stream) != 0) { (void)hipStreamSynchronize(c->stream); return XIVO_HIP_ERR_HIP; }
  }
  HIP_TRY(hipStreamSynchronize(c->stream));        // the one synchronisation of the call: kernel end = system-scope release
  const int* s_st = reinterpret_cast<const int*>(c->pin_h + o_st);
  memcpy(err_out, c->pin_h + o_err, (size_t)N * sizeof(double));
  if (p_down) {
    const double* sp = reinterpret_cast<const double*>(c->pin_h + o_Pout);
    if (ldp == N) memcpy(P, sp, (size_t)N * N * sizeof(double));
    else for (int j = 0; j < N; ++j) memcpy(P + (size_t)j * ldp, sp + (size_t)j * N, (size_t)N * sizeof(double));
  }
  return s_st[0] ? XIVO_HIP_ERR_NOT_SPD : XIVO_HIP_OK;
}

static int ensure_gate_buffers(xivo_hip_ctx* c, int F) {
  if (F <= c->Fmax && c->mask) return XIVO_HIP_OK;
  const int Fm = F > c->Mpmax / 2 ? F : c->Mpmax / 2;
  void* olds[] = {c->feats, c->J, c->finn, c->dist, c->mask, c->Jc};
  for (void* p : olds) if (p) hipFree(p);
  c->feats = nullptr; c->J = nullptr; c->finn = nullptr; c->dist = nullptr; c->mask = nullptr; c->Jc = nullptr;
  const size_t B = c->Bmax;
  int rc = dev_alloc(&c->feats, B * Fm);
  if (!rc) rc = dev_alloc(&c->J, B * Fm * 42);
  if (!rc) rc = dev_alloc(&c->finn, B * Fm * 2);
  if (!rc) rc = dev_alloc(&c->dist, B * Fm);
  if (!rc) rc = dev_alloc(&c->mask, B * Fm);
  if (!rc && c->calib_on) rc = dev_alloc(&c->Jc, B * Fm * 44);
  if (!rc && !c->rows_instate) rc = dev_alloc(&c->rows_instate, B);
  // every entry starts absent (sind = -1) and masked out until a scene / edit writes it
  if (!rc && hipMemsetAsync(c->feats, 0xFF, B * Fm * sizeof(xivo_feat_in), c->stream) != hipSuccess) rc = XIVO_HIP_ERR_HIP;
  if (!rc && hipMemsetAsync(c->mask, 0, B * Fm, c->stream) != hipSuccess) rc = XIVO_HIP_ERR_HIP;
  if (!rc) c->Fmax = Fm;
  return rc;
}

int xivo_hip_mh_gate_dense(xivo_hip_ctx* c, int B, int F, double R, double mh_thresh, double mh_mult,
                           int min_inliers, unsigned char* mask_out, double* dist_out) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || B <= 0 || B > c->Bmax || F <= 0 || 2 * F > c->M) return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  int rc = ensure_gate_buffers(c, F);
  if (rc) return rc;
  rc = ensure_dense(c);
  if (rc) return rc;
  rc = ensure_HT(c);
  if (rc) return rc;
  const int Np = c->Np, Mp = c->Mp, ldh = c->Mpmax;
  {
    GemmExtra x; x.C2 = c->PHT; x.sC2 = c->sK; x.ldc2 = Np;
    rc = gemm(c, ST_HP, B, Mp, Np, c->H, c->sH, ldh, c->P, c->sP, Np, Np, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0,
              c->HP, c->sH, ldh, x);
  }
  if (rc) return rc;
  GateDenseArgs a{};
  a.H = c->H; a.strideH = c->sH; a.ldh = ldh; a.HP = c->HP; a.strideHP = c->sH; a.ldhp = ldh;
  a.Hw = c->H; a.HTw = c->HT; a.strideHT = c->sHT; a.ldht = Np; a.HPw = nullptr; a.PHTw = nullptr; a.PHTr = c->PHT;
  a.inn = c->inn; a.strideInn = c->Mpmax; a.diagR = c->diagR; a.strideR = c->Mpmax;
  a.mask = c->mask; a.dist = c->dist; a.F = F; a.Np = Np; a.batch = B;
  a.R = R; a.thresh = mh_thresh; a.mult = mh_mult; a.min_inliers = min_inliers;
  a.ell = c->ell; a.have_ell = 1;
  {
    StageTimer st(c, ST_GATE, 0.0, "gate_dense_kernel");
    c->gate_sparse_last = 0;
    HIP_TRY((hipError_t)launch_gate_dense(a, c->stream));
  }
  if (mask_out) HIP_TRY(hipMemcpyAsync(mask_out, c->mask, (size_t)B * F, hipMemcpyDeviceToHost, c->stream));
  if (dist_out) HIP_TRY(hipMemcpyAsync(dist_out, c->dist, (size_t)B * F * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (mask_out || dist_out) HIP_TRY(hipStreamSynchronize(c->stream));
  return XIVO_HIP_OK;
}

// ------------------------------------------------------------------ G-level
int xivo_hip_set_layout(xivo_hip_ctx* c, const xivo_layout* lay, const xivo_cam* cam) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !lay || !cam) return XIVO_HIP_ERR_INVALID;
  if (lay->N != c->N || lay->group_begin < 21 || lay->n_groups <= 0 || lay->n_features <= 0 ||
      lay->feature_begin < lay->group_begin + 6 * lay->n_groups ||
      lay->feature_begin + 3 * lay->n_features > lay->N)
    return XIVO_HIP_ERR_INVALID;
  if (cam->model < XIVO_CAM_PINHOLE || cam->model > XIVO_CAM_EQUI) return XIVO_HIP_ERR_INVALID;
  c->lay = *lay; c->cam = *cam; c->have_layout = true;
  if (!c->poses) {
    int rc = dev_alloc(&c->poses, (size_t)c->Bmax);
    if (!rc) rc = dev_alloc(&c->absorb_count, (size_t)c->Bmax);
    if (!rc) rc = dev_alloc(&c->groups, (size_t)c->Bmax * lay->n_groups);
    if (rc) return rc;
  }
  return XIVO_HIP_OK;
}

int xivo_hip_set_scene(xivo_hip_ctx* c, int b0, int nb, int F, const xivo_pose_in* poses,
                       const xivo_group_in* groups, const xivo_feat_in* feats) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !c->have_layout || F <= 0 || 2 * F > c->Mmax || !poses || !groups || !feats)
    return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  int rc = ensure_gate_buffers(c, F);
  if (rc) return rc;
  for (long i = 0; i < (long)nb * F; ++i) {
    const xivo_feat_in& f = feats[i];
    if (f.sind == -1) continue;   // absent entry
    if (f.ref_sind < 0 || f.ref_sind >= c->lay.n_groups || f.sind < 0 || f.sind >= c->lay.n_features)
      return XIVO_HIP_ERR_INVALID;
  }
  c->F = F;
  HIP_TRY(hipMemcpyAsync(c->poses + b0, poses, (size_t)nb * sizeof(xivo_pose_in), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->groups + (size_t)b0 * c->lay.n_groups, groups,
                         (size_t)nb * c->lay.n_groups * sizeof(xivo_group_in), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpy2DAsync(c->feats + (size_t)b0 * c->Fmax, (size_t)c->Fmax * sizeof(xivo_feat_in), feats,
                           (size_t)F * sizeof(xivo_feat_in), (size_t)F * sizeof(xivo_feat_in), nb,
                           hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XIVO_HIP_OK;
}

int xivo_hip_jacobians_instate(xivo_hip_ctx* c, int B) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !c->have_layout || B <= 0 || B > c->Bmax || c->F <= 0) return XIVO_HIP_ERR_INVALID;
  StageTimer st(c, ST_JAC, 0.0, "jac_instate_kernel");
  return launch_jac_instate(scene_buffers(c), c->lay, c->cam, B, c->stream) ? XIVO_HIP_ERR_HIP : XIVO_HIP_OK;
}

int xivo_hip_get_jacobians(xivo_hip_ctx* c, int b0, int nb, double* J, double* inn) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || c->F <= 0) return XIVO_HIP_ERR_INVALID;
  const size_t F = c->F, Fm = c->Fmax;
  if (J) {
    int rc = d2h_rows(c, J, F * 42 * sizeof(double), c->J + (size_t)b0 * Fm * 42, Fm * 42 * sizeof(double),
                      F * 42 * sizeof(double), nb);
    if (rc) return rc;
  }
  if (inn) {
    int rc = d2h_rows(c, inn, F * 2 * sizeof(double), c->finn + (size_t)b0 * Fm * 2, Fm * 2 * sizeof(double),
                      F * 2 * sizeof(double), nb);
    if (rc) return rc;
  }
  return XIVO_HIP_OK;
}

int xivo_hip_set_calib(xivo_hip_ctx* c, const xivo_calib_layout* layout) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !c->have_layout) return XIVO_HIP_ERR_INVALID;
  if (!layout) { c->calib_on = false; c->calib_motion = false; c->cl = xivo_calib_layout{-1, -1, 0, 0}; return XIVO_HIP_OK; }
  const xivo_calib_layout& l = *layout;
  const int N = c->N;
  if (l.td >= N || (l.Cg >= 0 && l.Cg + 15 > N) || l.cam_dim < 0 || l.cam_dim > 9 ||
      (l.cam_dim > 0 && (l.cam_begin < 0 || l.cam_begin + l.cam_dim > N)))
    return XIVO_HIP_ERR_INVALID;
  // slots as src/core.h:40-75 numbers them: td right behind Wsg, Cg behind td (or Wsg), the intrinsics behind the motion block
  if ((l.td >= 0 && l.td != 23) || (l.Cg >= 0 && l.Cg != (l.td >= 0 ? 24 : 23))) return XIVO_HIP_ERR_INVALID;
  if (!c->calib) { int rc = dev_alloc(&c->calib, (size_t)c->Bmax); if (rc) return rc; }
  if (!c->Jc && c->Fmax > 0) { int rc = dev_alloc(&c->Jc, (size_t)c->Bmax * c->Fmax * 44); if (rc) return rc; }
  if (!c->Hlead) { int rc = dev_alloc(&c->Hlead, (size_t)c->Bmax * c->Mpmax * LEAD_K); if (rc) return rc; }
  c->lead_valid = false;
  c->cl = l;
  c->calib_on = l.td >= 0 || l.cam_dim > 0;       // measurement side: blocks beyond the default build's (the Cg / bg blocks sit inside the td block)
  c->calib_motion = l.td >= 0 || l.Cg >= 0;       // motion side: kMotionSize > 23
  return XIVO_HIP_OK;
}

int xivo_hip_set_calib_state(xivo_hip_ctx* c, int b0, int nb, const xivo_calib_in* calib) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !calib || !c->calib) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  HIP_TRY(hipMemcpyAsync(c->calib + b0, calib, (size_t)nb * sizeof(xivo_calib_in), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));    // host buffer is only borrowed for the call
  return XIVO_HIP_OK;
}

int xivo_hip_set_calib_gyro(xivo_hip_ctx* c, int b0, int nb, const double* gyro3) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !gyro3 || !c->calib) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  static_assert(offsetof(xivo_calib_in, gyro) == 0, "gyro leads xivo_calib_in");
  HIP_TRY(hipMemcpy2DAsync(c->calib + b0, sizeof(xivo_calib_in), gyro3, 3 * sizeof(double), 3 * sizeof(double), (size_t)nb,
                           hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XIVO_HIP_OK;
}

int xivo_hip_get_calib_state(xivo_hip_ctx* c, int b0, int nb, xivo_calib_in* calib) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !calib || !c->calib) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  HIP_TRY(hipMemcpyAsync(calib, c->calib + b0, (size_t)nb * sizeof(xivo_calib_in), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XIVO_HIP_OK;
}

int xivo_hip_get_jacobians_calib(xivo_hip_ctx* c, int b0, int nb, double* Jc) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || c->F <= 0 || !Jc || !c->calib_on || !c->Jc) return XIVO_HIP_ERR_INVALID;
  const size_t F = c->F, Fm = c->Fmax;
  return d2h_rows(c, Jc, F * 44 * sizeof(double), c->Jc + (size_t)b0 * Fm * 44, Fm * 44 * sizeof(double), F * 44 * sizeof(double), nb);
}

static int gate_impl(xivo_hip_ctx* c, int B, double R, double th, double mult, int min_inl, int use_gating) {
  GateArgs a{};
  a.sb = scene_buffers(c); a.lay = c->lay; a.P = c->P; a.strideP = c->sP; a.ldp = c->Np;
  a.R = R; a.thresh = th; a.mult = mult; a.min_inliers = min_inl; a.batch = B; a.use_gating = use_gating;
  StageTimer st(c, ST_GATE, 0.0, "gate_sparse_kernel");
  c->gate_sparse_last = 1;
  return launch_gate_sparse(a, c->stream) ? XIVO_HIP_ERR_HIP : XIVO_HIP_OK;
}

// Estimator::MHGating of an online-calibration build: the gate needs the WHOLE row J() incl. the td / Cg / bg / intrinsics
// blocks (update.cpp:60-70), which is not the row FillJacobianBlock stacks (the :675-676 overwrite): every present feature is
// stacked once as its full J() (dense rows) and gated on (J P) J^T + R by the dense-row gate. gate = 0: every present feature
// is an inlier (Estimator::OutlierRejection does not gate F <= min_required_inliers_, src/manager.cpp:635).
static int stack_impl(xivo_hip_ctx* c, int B, double R, int write_dense, unsigned char* mask_override, int full_rows);
static int calib_gate(xivo_hip_ctx* c, int B, double R, double mh_thresh, double mh_mult, int min_inliers, int gate) {
  int rc = gate_impl(c, B, R, mh_thresh, mh_mult, min_inliers, 0);
  if (rc) return rc;
  if (gate) {
    c->M = 2 * c->F; c->Mp = round_up16(c->M);
    c->dense_valid = true; c->dense_from_ell = false; c->stack_R = R; c->stack_B = B; c->oos_row0 = -1; c->mixed_row0 = -1; c->h_clean = false;
    rc = stack_impl(c, B, R, 1, nullptr, /*full_rows=*/1);
    if (rc) return rc;
    rc = ensure_HT(c);
    if (rc) return rc;
    const int Np = c->Np, Mp = c->Mp, ldh = c->Mpmax;
    {
      GemmExtra x; x.C2 = c->PHT; x.sC2 = c->sK; x.ldc2 = Np;
      rc = gemm(c, ST_HP, B, Mp, Np, c->H, c->sH, ldh, c->P, c->sP, Np, Np, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0, c->HP, c->sH, ldh, x);
      if (rc) return rc;
    }
    GateDenseArgs a{};
    a.H = c->H; a.strideH = c->sH; a.ldh = ldh; a.HP = c->HP; a.strideHP = c->sH; a.ldhp = ldh;
    a.Hw = c->H; a.HTw = c->HT; a.strideHT = c->sHT; a.ldht = Np; a.HPw = nullptr; a.PHTw = nullptr; a.PHTr = c->PHT;
    a.inn = c->inn; a.strideInn = c->Mpmax; a.diagR = c->diagR; a.strideR = c->Mpmax;
    a.mask = c->mask; a.dist = c->dist; a.F = c->F; a.Np = Np; a.batch = B; a.mask_ld = c->Fmax;   // (the stride xivo_hip_stack reads the mask with)
    a.R = R; a.thresh = mh_thresh; a.mult = mh_mult; a.min_inliers = min_inliers;
    a.ell = c->ell; a.have_ell = 0;
    a.feats = c->feats; a.Fmax = c->Fmax;        // absent entries of ragged batches are no candidates (per-filter present count)
    StageTimer st(c, ST_GATE, 0.0, "gate_dense_kernel");
    c->gate_sparse_last = 1;
    HIP_TRY((hipError_t)launch_gate_dense(a, c->stream));
  }
  return XIVO_HIP_OK;
}

int xivo_hip_mh_gate(xivo_hip_ctx* c, int B, double R, double mh_thresh, double mh_mult, int min_inliers,
                     unsigned char* mask_out, double* dist_out) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !c->have_layout || B <= 0 || B > c->Bmax || c->F <= 0) return XIVO_HIP_ERR_INVALID;
  // (online-calibration builds: the compact gate works on the whole row too - 43 columns, gate_sparse_kernel's wide form;
  //  with XIVO_HIP_FLAG_DENSE_H the dense-row gate of round 4)
  int rc = (c->calib_on && !calib_sparse(c)) ? calib_gate(c, B, R, mh_thresh, mh_mult, min_inliers, 1)
                                             : gate_impl(c, B, R, mh_thresh, mh_mult, min_inliers, 1);
  if (rc) return rc;
  const size_t F = c->F, Fm = c->Fmax;
  if (mask_out) { rc = d2h_rows(c, mask_out, F, c->mask, Fm, F, B); if (rc) return rc; }
  if (dist_out) {
    rc = d2h_rows(c, dist_out, F * sizeof(double), c->dist, Fm * sizeof(double), F * sizeof(double), B);
    if (rc) return rc;
  }
  return XIVO_HIP_OK;
}

static int stack_impl(xivo_hip_ctx* c, int B, double R, int write_dense, unsigned char* mask_override = nullptr, int full_rows = 0) {
  StackArgs a{};
  a.sb = scene_buffers(c); a.lay = c->lay; a.mb = meas_buffers(c);
  if (write_dense) c->ht_valid = true;
  if (mask_override) a.sb.mask = mask_override;
  a.Mp = c->Mpmax; a.Np = c->Np; a.batch = B; a.R = R;
  a.fix_group_block = (full_rows || (c->flags & XIVO_HIP_FLAG_FIX_GROUP_BLOCK)) ? 1 : 0;
  a.rows_instate = c->rows_instate;
  a.ell = c->ell; a.emit_ell = 1; a.write_dense = write_dense;
  // as-coded stacking of an online-calibration build on the sparse pipeline: compressed rows + the leading dense block
  // (full_rows - the whole-row stackings of the gate / RANSAC - stay dense rows)
  if (!full_rows && !write_dense && calib_sparse(c)) { a.lead = c->Hlead; a.strideLead = (long)c->Mpmax * LEAD_K; a.lead_k = LEAD_K; }
  StageTimer st(c, ST_STACK, 0.0, "stack_kernel");
  return launch_stack(a, c->stream) ? XIVO_HIP_ERR_HIP : XIVO_HIP_OK;
}

// the transposed dense copy: a G-level producer may have skipped it (mixed stacking); a consumer that needs it - the dense-row gate, the as-coded K H - I - rebuilds it from H here
static int ensure_HT(xivo_hip_ctx* c) {
  if (c->ht_valid) return XIVO_HIP_OK;
  StageTimer st(c, ST_STACK, 0.0, "transpose_H_kernel");
  if (launch_transpose_H(c->H, c->sH, c->Mpmax, c->HT, c->sHT, c->Np, c->Mpmax, c->Np, c->Bmax, c->stream)) return XIVO_HIP_ERR_HIP;
  c->ht_valid = true;
  return XIVO_HIP_OK;
}

// the dense copies of the stacked rows, for the consumers that need them (dense pipeline, OOS rows, get_H)
static int ensure_dense(xivo_hip_ctx* c) {
  if (c->dense_valid) return XIVO_HIP_OK;
  c->dense_valid = true;
  if (c->mixed_row0 >= 0) {   // mixed stacking: the in-state rows come from the compressed form, the OOS rows are in place
    c->h_clean = false; c->ht_valid = false;
    StageTimer st(c, ST_STACK, 0.0, "ell_to_dense_kernel");
    return launch_ell_to_dense(c->ell, c->H, c->sH, c->Mpmax, nullptr, c->sHT, c->Np, c->Mpmax, c->Np, c->Bmax, c->stream, c->mixed_row0)
               ? XIVO_HIP_ERR_HIP : XIVO_HIP_OK;
  }
  c->h_clean = false;
  if (c->dense_from_ell) {   // S-level hand-over: the compressed rows are the source (filters that do not fit hold dense rows already)
    StageTimer st(c, ST_STACK, 0.0, "ell_to_dense_kernel");
    return launch_ell_to_dense(c->ell, c->H, c->sH, c->Mpmax, c->HT, c->sHT, c->Np, c->Mpmax, c->Np, c->Bmax, c->stream)
               ? XIVO_HIP_ERR_HIP : XIVO_HIP_OK;
  }
  if (c->lead_valid) {   // online-calibration stacking on the sparse pipeline: dense rows carry the calibration columns themselves
    c->lead_valid = false;
    for (int b = 0; b < c->stack_B; ++b) c->ell_over_h[b] = 1;
  }
  return stack_impl(c, c->stack_B, c->stack_R, 1);
}

int xivo_hip_stack(xivo_hip_ctx* c, int B, double R) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !c->have_layout || B <= 0 || B > c->Bmax || c->F <= 0) return XIVO_HIP_ERR_INVALID;
  c->M = 2 * c->F; c->Mp = round_up16(c->M);
  const bool csp = calib_sparse(c);
  for (int b = 0; b < B; ++b) {
    // (calibration blocks: up to 34 shared columns - dense rows, or compressed rows + the leading dense block)
    c->ell_over_h[b] = (c->calib_on && !csp) ? 1 : 0; c->ell_nc_h[b] = 12;
    c->ell_pw_h[b] = (c->flags & XIVO_HIP_FLAG_FIX_GROUP_BLOCK) ? 9 : 6;   // group block(s) + feature block
  }
  c->lead_valid = csp;
  // the sparse-H pipeline reads only the compressed rows: skip the 2 x Mp x Np dense zero-fill + scatter
  const int dense = ((c->flags & XIVO_HIP_FLAG_DENSE_H) || (c->calib_on && !csp)) ? 1 : 0;
  c->dense_valid = dense != 0; c->dense_from_ell = false; c->stack_R = R; c->stack_B = B; c->oos_row0 = -1;
  c->mixed_row0 = -1; if (dense) c->h_clean = false;
  return stack_impl(c, B, R, dense);
}

int xivo_hip_oos_project(xivo_hip_ctx* c, int b0, int nb, int n_oos, const xivo_oos_in* feats, double Roos,
                         int* rows_out) {
  return xivo_hip_oos_project_ex(c, b0, nb, n_oos, feats, Roos, rows_out, 0u);
}

int xivo_hip_oos_project_ex(xivo_hip_ctx* c, int b0, int nb, int n_oos, const xivo_oos_in* feats, double Roos,
                            int* rows_out, unsigned options) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !c->have_layout || n_oos <= 0 || b0 != 0) return XIVO_HIP_ERR_INVALID;
  if (options & ~XIVO_HIP_OOS_WHOLE_BUFFER) return XIVO_HIP_ERR_INVALID;
  // XIVO_HIP_OOS_WHOLE_BUFFER (src/oos.cpp:28 as coded): SlowGivens sees the whole 2 kMaxGroup-row buffers of the feature, so
  // every feature contributes 2 kMaxGroup - 3 rows; the rows behind its 2 k observations are zero (include/xivo_hip.h)
  const int whole = (options & XIVO_HIP_OOS_WHOLE_BUFFER) ? 2 * c->lay.n_groups : 0;
  // feats == NULL: the list uploaded by the previous call is still resident (same nb, n_oos) - project it again
  if (!feats && (!c->oos || c->oos_nb != nb || c->oos_n != n_oos || c->oos_whole != whole)) return XIVO_HIP_ERR_INVALID;
  int max_rows = feats ? 0 : c->oos_max_rows;
  for (int b = 0; feats && b < nb; ++b) {
    int rows = 0;
    for (int o = 0; o < n_oos; ++o) {
      const xivo_oos_in& f = feats[(size_t)b * n_oos + o];
      if (f.n_obs < 2 || f.n_obs > XIVO_OOS_MAX_OBS) return XIVO_HIP_ERR_INVALID;
      for (int q = 0; q < f.n_obs; ++q)
        if (f.group_sind[q] < 0 || f.group_sind[q] >= c->lay.n_groups) return XIVO_HIP_ERR_INVALID;
      if (whole && 2 * f.n_obs > whole) return XIVO_HIP_ERR_INVALID;     // (more observations than the reference's buffer has rows)
      rows += whole ? whole - 3 : 2 * f.n_obs - 3;
    }
    if (rows > max_rows) max_rows = rows;
  }
  if (c->M + max_rows > c->Mmax) return XIVO_HIP_ERR_INVALID;
  // Mixed stacking (round 3, default whenever the in-state rows were stacked in the compressed form only and nothing
  // forces the dense pipeline): the OOS rows go to the dense buffer behind the in-state rows and the update keeps the
  // sparse walk for the in-state rows - only the OOS block takes the MFMA products (update_sparse_range). Needs a
  // 16-row-padded OOS block inside the allocation; otherwise (and with XIVO_HIP_FLAG_DENSE_H) every row
  // becomes dense as before.
  const bool mixed = !c->calib_on && !c->dense_valid && !c->dense_from_ell && c->oos_row0 < 0 && b0 == 0 &&
                     !(c->flags & XIVO_HIP_FLAG_DENSE_H) && (c->M % 2 == 0) &&
                     c->M + round_up16(max_rows + 16) <= c->Mpmax && c->Np <= 512;
  if (!mixed) { int rcd = ensure_dense(c); if (rcd) return rcd; c->mixed_row0 = -1; }
  else {
    // the OOS rows must start from zero: only the extrinsics and group columns are ever written there in this mode, so
    // those are cleared (whole rows once, if anything else has used the dense buffer since it was allocated)
    const int nz = round_up16(max_rows + 16) < c->Mpmax - c->M ? round_up16(max_rows + 16) : c->Mpmax - c->M;
    if (!c->h_clean) {
      HIP_TRY((hipError_t)launch_zero_rows(c->H, c->sH, c->Mpmax, 0, c->Mpmax, 0, c->Np, c->Bmax, c->stream));
      c->h_clean = true;
    } else {
      HIP_TRY((hipError_t)launch_zero_rows(c->H, c->sH, c->Mpmax, c->M, nz, 15, 21, nb, c->stream));
      HIP_TRY((hipError_t)launch_zero_rows(c->H, c->sH, c->Mpmax, c->M, nz, c->lay.group_begin, c->lay.group_begin + 6 * c->lay.n_groups, nb, c->stream));
    }
    c->mixed_row0 = c->M;
  }
  if (n_oos * nb > c->oos_cap) {
    if (c->oos) hipFree(c->oos);
    c->oos = nullptr; c->oos_cap = 0;
    if (!feats) return XIVO_HIP_ERR_INVALID;
    int rc = dev_alloc(&c->oos, (size_t)n_oos * c->Bmax);
    if (rc) return rc;
    c->oos_cap = n_oos * c->Bmax;
  }
  if (!c->oos_rows) { int rc = dev_alloc(&c->oos_rows, (size_t)c->Bmax); if (rc) return rc; }
  if (feats) {
    HIP_TRY(hipMemcpyAsync(c->oos, feats, (size_t)nb * n_oos * sizeof(xivo_oos_in), hipMemcpyHostToDevice, c->stream));
    c->oos_nb = nb; c->oos_n = n_oos; c->oos_max_rows = max_rows; c->oos_whole = whole;
  }
  OosArgs a{};
  a.feats = c->oos; a.n_oos = n_oos; a.poses = c->poses; a.groups = c->groups; a.lay = c->lay; a.cam = c->cam;
  a.calib = c->calib_on ? c->calib : nullptr; a.cam_dim = c->calib_on ? c->cl.cam_dim : 0;
  a.mb = meas_buffers(c); a.row0 = c->M; a.Mp = c->Mpmax; a.Np = c->Np; a.batch = nb; a.Roos = Roos; a.whole = whole;
  if (mixed) { a.mb.HT = nullptr; c->ht_valid = false; }
  c->oos_row0 = c->M; c->oos_R = Roos;
  a.rows_out = c->oos_rows;
  {
    StageTimer st(c, ST_OTHER, 0.0);
    HIP_TRY((hipError_t)launch_oos(a, c->stream));
  }
  if (rows_out) HIP_TRY(hipMemcpyAsync(rows_out, c->oos_rows, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->M += max_rows; c->Mp = round_up16(c->M);
  if (!mixed) for (int b = b0; b < b0 + nb; ++b) c->ell_over_h[b] = 1;   // OOS rows are dense over the group blocks: dense path
  return XIVO_HIP_OK;
}

// Estimator::OnePointRANSAC (src/update.cpp:213-393) for filters [0,B) on the resident state, after
// xivo_hip_jacobians_instate + xivo_hip_mh_gate (the MH inlier mask is the input set):
//   select (low-innovation set, temporary reference group)                       :238-301   ransac_select_kernel
//   BackupState: P, nominal state, groups                                        :283       device-to-device copies
//   zero P rows / cols of non-members                                            :299-316   ransac_zero_kernel
//   partial update on the FULL rows J() of the low-innovation set + AbsorbError  :320-333   stack (full rows) + update + absorb
//   re-Jacobians at the updated state, chi-square rescue                         :343-369   jac_instate + ransac_rescue_kernel
//   RestoreState, re-Jacobians at the original state                             :383-387
// The resulting inlier set replaces the MH mask (what xivo_hip_stack / xivo_hip_absorb_error read afterwards).
int xivo_hip_one_point_ransac(xivo_hip_ctx* c, int B, double R, double ransac_thresh, double ransac_chi2,
                              const int* gauge_group, const unsigned long long* absorb_groups,
                              unsigned char* inlier_mask_out, double* chi2_out, int* n_rejected_out) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !c->have_layout || B <= 0 || B > c->Bmax || c->F <= 0 || !c->mask || !c->poses) return XIVO_HIP_ERR_INVALID;
  if (c->lay.n_groups > 64) return XIVO_HIP_ERR_UNSUPPORTED;
  const size_t Bm = c->Bmax, ng = c->lay.n_groups;
  // online-calibration builds: the calibration state is backed up / restored with X_ (imu_.BackupState, Camera::BackupState,
  // src/estimator.cpp:1421-1427), the partial update stacks the whole rows J() as dense rows, AbsorbError retracts td / Cg / Ca /
  // the intrinsics too, and the rescue test uses the whole-row distances of the dense-row gate
  const bool cal = c->calib_on;
  if (cal && !c->calib_rs) { int rc = dev_alloc(&c->calib_rs, Bm); if (rc) return rc; }
  if (!c->Prs || c->rs_Fmax != c->Fmax) {
    void* olds[] = {c->rs_low, c->rs_lowkeep, c->rs_keep, c->rs_chi};
    for (void* p : olds) if (p) hipFree(p);
    c->rs_low = c->rs_lowkeep = c->rs_keep = nullptr; c->rs_chi = nullptr;
    int rc = XIVO_HIP_OK;
    auto A = [&](auto** p, size_t n) { if (rc == XIVO_HIP_OK && !*p) rc = dev_alloc(p, n); };
    A(&c->Prs, Bm * c->sP); A(&c->poses_rs, Bm); A(&c->groups_rs, Bm * ng);
    A(&c->rs_low, Bm * c->Fmax); A(&c->rs_lowkeep, Bm * c->Fmax); A(&c->rs_keep, Bm * c->Fmax); A(&c->rs_chi, Bm * c->Fmax);
    A(&c->rs_zg, Bm); A(&c->rs_gmask, Bm); A(&c->rs_state, Bm); A(&c->rs_gauge, Bm); A(&c->rs_nrej, Bm);
    if (rc) return rc;
    c->rs_Fmax = c->Fmax;
  }
  if (gauge_group) HIP_TRY(hipMemcpyAsync(c->rs_gauge, gauge_group, (size_t)B * sizeof(int), hipMemcpyHostToDevice, c->stream));
  else HIP_TRY(hipMemsetAsync(c->rs_gauge, 0xFF, (size_t)B * sizeof(int), c->stream));
  if (absorb_groups) HIP_TRY(hipMemcpyAsync(c->rs_gmask, absorb_groups, (size_t)B * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
  RansacArgs a{};
  a.sb = scene_buffers(c); a.lay = c->lay; a.P = c->P; a.strideP = c->sP; a.ldp = c->Np; a.Np = c->Np;
  a.R = R; a.thresh = ransac_thresh; a.chi2 = ransac_chi2; a.gauge = c->rs_gauge;
  a.low = c->rs_low; a.low_keep = c->rs_lowkeep; a.zero_groups = c->rs_zg; a.state = c->rs_state;
  a.keep = c->rs_keep; a.chi = c->rs_chi; a.n_rejected = c->rs_nrej; a.batch = B;
  {
    StageTimer st(c, ST_OTHER, 0.0, "ransac_select_kernel");
    HIP_TRY((hipError_t)launch_ransac_select(a, c->stream));
  }
  // the low-innovation set as select found it (filters with nothing to update get an all-neutral stacking mask)
  HIP_TRY(hipMemcpyAsync(c->rs_lowkeep, c->rs_low, (size_t)B * c->Fmax, hipMemcpyDeviceToDevice, c->stream));
  // BackupState (src/estimator.cpp:1410-1428)
  HIP_TRY(hipMemcpyAsync(c->Prs, c->P, (size_t)B * c->sP * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->poses_rs, c->poses, (size_t)B * sizeof(xivo_pose_in), hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->groups_rs, c->groups, (size_t)B * ng * sizeof(xivo_group_in), hipMemcpyDeviceToDevice, c->stream));
  if (cal) HIP_TRY(hipMemcpyAsync(c->calib_rs, c->calib, (size_t)B * sizeof(xivo_calib_in), hipMemcpyDeviceToDevice, c->stream));
  {
    StageTimer st(c, ST_OTHER, 0.0, "ransac_zero_kernel");
    HIP_TRY((hipError_t)launch_ransac_zero(a, c->P, c->stream));
  }
  // partial update: H_ rows = the full J() of the low-innovation inliers (:326 - no FillJacobianBlock), R_ on the diagonal
  c->M = 2 * c->F; c->Mp = round_up16(c->M);
  for (int b = 0; b < B; ++b) { c->ell_over_h[b] = cal ? 1 : 0; c->ell_nc_h[b] = 12; c->ell_pw_h[b] = 9; }
  c->lead_valid = false;
  const int dense = ((c->flags & XIVO_HIP_FLAG_DENSE_H) || cal) ? 1 : 0;
  c->dense_valid = dense != 0; c->dense_from_ell = !cal; c->stack_R = R; c->stack_B = B;
  c->oos_row0 = -1;   // the partial stacking replaces the rows of any earlier xivo_hip_oos_project (as xivo_hip_stack does)
  c->mixed_row0 = -1; if (dense) c->h_clean = false;
  int rc = stack_impl(c, B, R, dense, c->rs_low, 1);
  if (rc) return rc;
  rc = xivo_hip_update_joseph(c, B);
  if (rc) return rc;
  {  // AbsorbError (:333): in_current_ekf_update_ is empty at this point of Estimator::UpdateStep (cleared at
     // src/manager.cpp:28, filled after OutlierRejection), so no feature state moves; State::counter is restored with X_
    AbsorbArgs ab{};
    ab.poses = c->poses; ab.groups = c->groups; ab.feats = c->feats; ab.mask = nullptr; ab.err = c->err; ab.strideErr = c->Np;
    ab.lay = c->lay; ab.F = c->F; ab.Fmax = c->Fmax; ab.batch = B; ab.counter = nullptr; ab.status = c->status;
    ab.group_mask = absorb_groups ? c->rs_gmask : nullptr;
    ab.calib = (c->calib_on || c->calib_motion) ? c->calib : nullptr; ab.cl = c->cl;
    StageTimer st(c, ST_OTHER, 0.0, "absorb_error_kernel");
    HIP_TRY((hipError_t)launch_absorb_error(ab, c->stream));
  }
  rc = xivo_hip_jacobians_instate(c, B);                                   // :348 at the updated state
  if (rc) return rc;
  if (!cal) {
    StageTimer st(c, ST_OTHER, 0.0, "ransac_rescue_kernel");
    HIP_TRY((hipError_t)launch_ransac_rescue(a, c->stream));
  } else {
    // S = J P J^T + R of every MH inlier on its WHOLE row at the updated state against the partially updated P (:350-356):
    // the rows stacked once more in full (scratch: xivo_hip_stack re-stacks the final inlier set), H P, the dense-row distances
    c->dense_valid = true; c->dense_from_ell = false; c->h_clean = false;
    rc = stack_impl(c, B, R, 1, nullptr, /*full_rows=*/1);
    if (rc) return rc;
    rc = ensure_HT(c);
    if (rc) return rc;
    const int Np = c->Np, Mp = c->Mp, ldh = c->Mpmax;
    {
      GemmExtra x; x.C2 = c->PHT; x.sC2 = c->sK; x.ldc2 = Np;
      rc = gemm(c, ST_HP, B, Mp, Np, c->H, c->sH, ldh, c->P, c->sP, Np, Np, nullptr, 0, 0, nullptr, 0, 0, 0, nullptr, 0, c->HP, c->sH, ldh, x);
      if (rc) return rc;
    }
    GateDenseArgs ga{};
    ga.H = c->H; ga.strideH = c->sH; ga.ldh = ldh; ga.HP = c->HP; ga.strideHP = c->sH; ga.ldhp = ldh;
    ga.Hw = c->H; ga.HTw = c->HT; ga.strideHT = c->sHT; ga.ldht = Np; ga.HPw = nullptr; ga.PHTw = nullptr; ga.PHTr = c->PHT;
    ga.inn = c->inn; ga.strideInn = c->Mpmax; ga.diagR = c->diagR; ga.strideR = c->Mpmax;
    // (scratch outputs: the mask goes to rs_low - dead once the partial update is stacked -, the distances to rs_chi, where the
    //  decision kernel below reads them and leaves chi2 per tested feature; c->dist keeps the MH distances)
    ga.mask = c->rs_low; ga.dist = c->rs_chi; ga.F = c->F; ga.Np = Np; ga.batch = B; ga.mask_ld = c->Fmax;
    ga.R = R; ga.thresh = ransac_chi2; ga.mult = 1.0; ga.min_inliers = -1; ga.no_relax = 1;
    ga.ell = c->ell; ga.have_ell = 0; ga.feats = c->feats; ga.Fmax = c->Fmax;
    {
      StageTimer st(c, ST_GATE, 0.0, "gate_dense_kernel");
      HIP_TRY((hipError_t)launch_gate_dense(ga, c->stream));
    }
    StageTimer st(c, ST_OTHER, 0.0, "ransac_rescue_dist_kernel");
    HIP_TRY((hipError_t)launch_ransac_rescue_dist(a, c->rs_chi, c->Fmax, c->stream));
  }
  // RestoreState + Jacobians at the original state (:383-387)
  HIP_TRY(hipMemcpyAsync(c->P, c->Prs, (size_t)B * c->sP * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->poses, c->poses_rs, (size_t)B * sizeof(xivo_pose_in), hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->groups, c->groups_rs, (size_t)B * ng * sizeof(xivo_group_in), hipMemcpyDeviceToDevice, c->stream));
  if (cal) HIP_TRY(hipMemcpyAsync(c->calib, c->calib_rs, (size_t)B * sizeof(xivo_calib_in), hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->mask, c->rs_keep, (size_t)B * c->Fmax, hipMemcpyDeviceToDevice, c->stream));
  rc = xivo_hip_jacobians_instate(c, B);
  if (rc) return rc;
  c->gate_sparse_last = 1;
  const size_t F = c->F, Fm = c->Fmax;
  if (inlier_mask_out) { rc = d2h_rows(c, inlier_mask_out, F, c->mask, Fm, F, B); if (rc) return rc; }
  if (chi2_out) { rc = d2h_rows(c, chi2_out, F * sizeof(double), c->rs_chi, Fm * sizeof(double), F * sizeof(double), B); if (rc) return rc; }
  if (n_rejected_out) HIP_TRY(hipMemcpyAsync(n_rejected_out, c->rs_nrej, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));   // gauge_group / absorb_groups are borrowed host memory
  return XIVO_HIP_OK;
}

// Estimator::CloseLoopInternal's stacking (src/update.cpp:183-196) with Feature::ComputeLCJacobian (src/oos.cpp:92-145) on the
// resident scene: the 2n rows of every filter are built dense in a scratch block (lc_rows_kernel) and handed over like any
// device-resident H_ (stage_measurements: row-pair compressed where they fit - group block private, extrinsics [+ intrinsics]
// common -, so the update that follows takes the sparse pipeline). The inlier mask of the last gating pass is left alone:
// AbsorbError after a loop closure updates in_current_ekf_update_ as the last FilterUpdate left it (src/estimator.cpp:906-912).
int xivo_hip_close_loop_stack(xivo_hip_ctx* c, int b0, int nb, int n, const xivo_lc_match* matches, double Rlc) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !c->have_layout || !c->poses || !c->feats || n <= 0 || 2 * n > c->Mmax || !matches || !(Rlc > 0.0))
    return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  for (size_t i = 0; i < (size_t)nb * n; ++i) {
    const xivo_lc_match& m = matches[i];
    if (m.feat >= c->F || (m.feat >= 0 && (m.group_sind < 0 || m.group_sind >= c->lay.n_groups))) return XIVO_HIP_ERR_INVALID;
  }
  const int M = 2 * n, N = c->N;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_m = 0, o_H = al((size_t)nb * n * sizeof(xivo_lc_match)), o_inn = al(o_H + (size_t)nb * M * N * sizeof(double)),
               o_R = al(o_inn + (size_t)nb * M * sizeof(double)), total = al(o_R + (size_t)nb * M * sizeof(double));
  if (total > c->lc_cap) {
    if (c->lc_buf) hipFree(c->lc_buf);
    c->lc_buf = nullptr; c->lc_cap = 0;
    if (hipMalloc(&c->lc_buf, total) != hipSuccess) { (void)hipGetLastError(); return XIVO_HIP_ERR_NOMEM; }
    c->lc_cap = total;
  }
  char* base = static_cast<char*>(c->lc_buf);
  HIP_TRY(hipMemcpyAsync(base + o_m, matches, (size_t)nb * n * sizeof(xivo_lc_match), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(base + o_H, 0, (size_t)nb * M * N * sizeof(double), c->stream));     // H_.setZero(2n, N) (update.cpp:184)
  LcArgs a{};
  a.matches = reinterpret_cast<const xivo_lc_match*>(base + o_m); a.n = n;
  a.poses = c->poses + b0; a.groups = c->groups + (size_t)b0 * c->lay.n_groups; a.feats = c->feats + (size_t)b0 * c->Fmax; a.Fmax = c->Fmax;
  a.lay = c->lay; a.cam = c->cam; a.calib = c->calib_on ? c->calib + b0 : nullptr;
  a.cl = c->calib_on ? c->cl : xivo_calib_layout{-1, -1, 0, 0}; a.invdepth = (c->flags & XIVO_HIP_FLAG_INVDEPTH) ? 1 : 0;
  a.H = reinterpret_cast<double*>(base + o_H); a.strideH = (long)M * N; a.ldh = M;
  a.inn = reinterpret_cast<double*>(base + o_inn); a.diagR = reinterpret_cast<double*>(base + o_R); a.strideV = M;
  a.Rlc = Rlc; a.batch = nb;
  {
    StageTimer st(c, ST_OTHER, 0.0, "lc_rows_kernel");
    HIP_TRY((hipError_t)launch_lc_rows(a, c->stream));
  }
  c->oos_row0 = -1;
  return stage_measurements(c, b0, nb, M, a.H, a.strideH, a.ldh, a.inn, a.strideV, a.diagR, a.strideV);
}

// Measurement compression of the OOS rows appended by the last xivo_hip_oos_project (use_compression_ /
// compression_trigger_ratio_, src/estimator.h:399-402; xivo::QR, src/helpers.cpp:77-101): per filter, when the block
// has more than trigger_ratio times as many rows as non-zero columns, it is replaced by the triangular factor of its QR
// decomposition (oos_compress_kernel) and the row count of the stacked measurement shrinks accordingly.
int xivo_hip_compress_oos(xivo_hip_ctx* c, int B, double trigger_ratio, int* rows_out) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !c->have_layout || B <= 0 || B > c->Bmax || c->oos_row0 < 0 || !c->oos_rows || B != c->oos_nb || !(trigger_ratio >= 1.0))
    return XIVO_HIP_ERR_INVALID;
  OosCompressArgs a{};
  a.lay = c->lay; a.mb = meas_buffers(c); a.row0 = c->oos_row0; a.rows = c->oos_rows; a.rows_out = c->oos_rows;
  if (c->mixed_row0 >= 0) { a.mb.HT = nullptr; c->ht_valid = false; }
  a.ratio = trigger_ratio; a.Roos = c->oos_R; a.batch = B;
  int rc;
  {
    StageTimer st(c, ST_OTHER, 0.0, "oos_compress_kernel");
    rc = launch_oos_compress(a, c->oos_max_rows, c->stream);
  }
  if (rc > 0) return XIVO_HIP_ERR_HIP;
  std::vector<int> rows(B);
  HIP_TRY(hipMemcpyAsync(rows.data(), c->oos_rows, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  int mx = 0;
  for (int b = 0; b < B; ++b) mx = rows[b] > mx ? rows[b] : mx;
  // (rc == -1: block larger than the built kernels - rows are left as they are, which is always valid)
  c->M = c->oos_row0 + mx; c->Mp = round_up16(c->M); c->oos_max_rows = mx;
  if (rows_out) memcpy(rows_out, rows.data(), (size_t)B * sizeof(int));
  return XIVO_HIP_OK;
}

int xivo_hip_filter_update(xivo_hip_ctx* c, int B, double R, double mh_thresh, double mh_mult, int min_inliers,
                           int use_gating) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !c->have_layout || B <= 0 || B > c->Bmax || c->F <= 0) return XIVO_HIP_ERR_INVALID;
  int rc = xivo_hip_jacobians_instate(c, B);
  if (rc) return rc;
  // Estimator::OutlierRejection only gates when F > min_required_inliers_ (src/manager.cpp:635)
  const int gate = use_gating && c->F > min_inliers;
  if (c->calib_on && !calib_sparse(c)) {
    // online-calibration builds on dense rows (XIVO_HIP_FLAG_DENSE_H): the gate needs the WHOLE row J() incl. the td / Cg / bg / intrinsics blocks (update.cpp:60-70),
    // which is not the row FillJacobianBlock stacks (the :675-676 overwrite): every present feature is stacked once as its
    // full J() (dense rows), gated on (J P) J^T + R by the dense-row gate, then the inliers are stacked as coded and updated
    rc = calib_gate(c, B, R, mh_thresh, mh_mult, min_inliers, gate);
    if (rc) return rc;
    rc = xivo_hip_stack(c, B, R);
    if (rc) return rc;
    return xivo_hip_update_joseph(c, B);
  }
  rc = gate_impl(c, B, R, mh_thresh, mh_mult, min_inliers, gate);
  if (rc) return rc;
  rc = xivo_hip_stack(c, B, R);
  if (rc) return rc;
  return xivo_hip_update_joseph(c, B);
}

static int givens_impl(xivo_hip_ctx* c, int nb, int rows, int nx, int nf, double* x, double* Hx, double* Hf,
                       int effective_rows, int* rows_out, int qr) {
  if (!c || nb <= 0 || rows < 2 || nx <= 0 || !x || !Hx || (!qr && (!Hf || nf <= 0 || nf > 64)) || (qr && nx > 512))
    return XIVO_HIP_ERR_INVALID;
  const int eff = effective_rows < 0 ? rows : effective_rows;
  // the reference CHECKs these (helpers.cpp:49-53, 79-84); here they are an error code
  if (eff > rows || eff < 2 || (qr ? eff <= nx : eff < nf)) return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  const size_t ex = (size_t)nb * rows, ehx = (size_t)nb * rows * nx, ehf = qr ? 0 : (size_t)nb * rows * nf;
  int rc = ensure_staging(c, ex + ehx + ehf);
  if (rc) return rc;
  double* dx = c->staging; double* dHx = dx + ex; double* dHf = dHx + ehx;
  HIP_TRY(hipMemcpyAsync(dx, x, ex * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(dHx, Hx, ehx * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (!qr) HIP_TRY(hipMemcpyAsync(dHf, Hf, ehf * sizeof(double), hipMemcpyHostToDevice, c->stream));
  GivensArgs a{}; a.x = dx; a.Hx = dHx; a.Hf = qr ? nullptr : dHf; a.rows = rows; a.nx = nx; a.nf = nf; a.eff = effective_rows;
  a.batch = nb; a.qr = qr;
  {
    StageTimer st(c, ST_OTHER, 0.0, "givens_kernel");
    HIP_TRY((hipError_t)launch_givens(a, c->stream));
  }
  HIP_TRY(hipMemcpyAsync(x, dx, ex * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(Hx, dHx, ehx * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (!qr) HIP_TRY(hipMemcpyAsync(Hf, dHf, ehf * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (rows_out) for (int b = 0; b < nb; ++b) rows_out[b] = qr ? eff : eff - nf;
  return XIVO_HIP_OK;
}

int xivo_hip_givens(xivo_hip_ctx* c, int nb, int rows, int nx, int nf, double* x, double* Hx, double* Hf,
                    int effective_rows, int* rows_out) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  return givens_impl(c, nb, rows, nx, nf, x, Hx, Hf, effective_rows, rows_out, 0);
}

int xivo_hip_qr(xivo_hip_ctx* c, int nb, int rows, int nx, double* x, double* Hx, int effective_rows, int* rows_out) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  return givens_impl(c, nb, rows, nx, 0, x, Hx, nullptr, effective_rows, rows_out, 1);
}

int xivo_hip_subfilter_update(xivo_hip_ctx* c, int b0, int nb, int n, xivo_subfilter_feat* feats,
                              const xivo_subfilter_opts* opts) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !c->have_layout || !c->poses || n <= 0 || !feats || !opts) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  for (size_t i = 0; i < (size_t)nb * n; ++i)
    if (feats[i].ref_sind < 0 || feats[i].ref_sind >= c->lay.n_groups) return XIVO_HIP_ERR_INVALID;
  const size_t bytes = (size_t)nb * n * sizeof(xivo_subfilter_feat);
  if (bytes > c->sub_cap) {
    if (c->sub) hipFree(c->sub);
    c->sub = nullptr; c->sub_cap = 0;
    if (hipMalloc((void**)&c->sub, bytes) != hipSuccess) return XIVO_HIP_ERR_NOMEM;
    c->sub_cap = bytes;
  }
  HIP_TRY(hipMemcpyAsync(c->sub, feats, bytes, hipMemcpyHostToDevice, c->stream));
  {
    StageTimer st(c, ST_OTHER, 0.0, "subfilter_kernel");
    if (launch_subfilter(c->sub, n, c->poses + b0, c->groups + (size_t)b0 * c->lay.n_groups, c->lay.n_groups, c->cam,
                         *opts, nb, c->stream, c->calib_on ? c->calib + b0 : nullptr, c->calib_on ? c->cl.cam_dim : 0,
                         (c->flags & XIVO_HIP_FLAG_INVDEPTH) ? 1 : 0))
      return XIVO_HIP_ERR_HIP;
  }
  HIP_TRY(hipMemcpyAsync(feats, c->sub, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XIVO_HIP_OK;
}

// Criteria::CandidateComparison (src/options.cpp:34-61) and the selection order of SelectAndAddNewFeatures /
// AddFeaturesToState-style loops (src/manager.cpp:364-376,417-421): host arithmetic on the array
// xivo_hip_subfilter_update returned - no device work.
int xivo_hip_candidate_order(const xivo_subfilter_feat* feats, int nb, int n, int strict, int score_type, int* order_out,
                             int* n_out, double* score_out) {
  if (!feats || nb < 0 || n <= 0 || !order_out || !n_out || score_type < 0 || score_type > 2) return XIVO_HIP_ERR_INVALID;
  for (int b = 0; b < nb; ++b) {
    const xivo_subfilter_feat* f = feats + (size_t)b * n;
    std::vector<int> idx;
    for (int i = 0; i < n; ++i) {
      if (score_out) {
        const double dn = sqrt(f[i].P[0] * f[i].P[0] + f[i].P[4] * f[i].P[4] + f[i].P[8] * f[i].P[8]);   // P().diagonal().norm()
        score_out[(size_t)b * n + i] = score_type == 0 ? -1.0 * f[i].P[8] : (score_type == 1 ? -1.0 * dn : -1.0 * (dn + f[i].outlier_counter));
      }
      if (f[i].candidate & (strict ? 2 : 1)) idx.push_back(i);
    }
    // as coded, the comparison ignores the score it has just computed from comparison_score_type and orders by
    // status, then Feature::score() = -P(2,2) (options.cpp:60); FeatureStatus READY = 2 > INITIALIZING = 1 (core.h:190-199).
    // std::sort leaves the order of equivalent elements unspecified: here ties keep the list order.
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c2) {
      const int s1 = f[a].status == XIVO_FEAT_READY ? 2 : 1, s2 = f[c2].status == XIVO_FEAT_READY ? 2 : 1;
      return (s1 > s2) || (s1 == s2 && -f[a].P[8] > -f[c2].P[8]);
    });
    n_out[b] = (int)idx.size();
    for (int i = 0; i < n; ++i) order_out[(size_t)b * n + i] = i < (int)idx.size() ? idx[i] : -1;
  }
  return XIVO_HIP_OK;
}

int xivo_hip_absorb_error(xivo_hip_ctx* c, int B) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !c->have_layout || B <= 0 || B > c->Bmax || c->F <= 0 || !c->mask) return XIVO_HIP_ERR_INVALID;
  AbsorbArgs a{};
  a.poses = c->poses; a.groups = c->groups; a.feats = c->feats; a.mask = c->mask; a.err = c->err; a.strideErr = c->Np;
  a.lay = c->lay; a.F = c->F; a.Fmax = c->Fmax; a.batch = B; a.counter = c->absorb_count; a.status = c->status;
  a.calib = (c->calib_on || c->calib_motion) ? c->calib : nullptr; a.cl = c->cl;
  StageTimer st(c, ST_OTHER, 0.0);
  return launch_absorb_error(a, c->stream) ? XIVO_HIP_ERR_HIP : XIVO_HIP_OK;
}

int xivo_hip_edit_batch(xivo_hip_ctx* c, int F, int n_ops, const xivo_edit_op* ops) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !c->have_layout || !c->poses || F <= 0 || 2 * F > c->Mmax || n_ops < 0 || (n_ops > 0 && !ops))
    return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  int rc = ensure_gate_buffers(c, F);
  if (rc) return rc;
  const xivo_layout& L = c->lay;
  std::vector<int> wg_filter, wg_begin;
  for (int o = 0; o < n_ops; ++o) {
    const xivo_edit_op& e = ops[o];
    if (e.b < 0 || e.b >= c->Bmax || (o > 0 && e.b < ops[o - 1].b)) return XIVO_HIP_ERR_INVALID;
    bool ok = false;
    switch (e.kind) {
      case XIVO_EDIT_P_ZERO_RC: ok = e.i0 >= 0 && e.i1 >= 0 && e.i0 + e.i1 <= c->N; break;
      case XIVO_EDIT_P_COPY_RC: ok = e.i0 >= 0 && e.i1 >= 0 && e.i2 >= 0 && e.i0 + e.i2 <= c->N && e.i1 + e.i2 <= c->N; break;
      case XIVO_EDIT_P_SET_BLOCK3: ok = e.i0 >= 0 && e.i0 + 3 <= c->N; break;
      case XIVO_EDIT_ADD_GROUP: case XIVO_EDIT_REMOVE_GROUP: ok = e.i0 >= 0 && e.i0 < L.n_groups; break;
      case XIVO_EDIT_ADD_FEATURE:
        ok = e.i0 >= 0 && e.i0 < F && e.i1 >= 0 && e.i1 < L.n_features && e.i2 >= 0 && e.i2 < L.n_groups; break;
      case XIVO_EDIT_REMOVE_FEATURE: case XIVO_EDIT_SET_XP: ok = e.i0 >= 0 && e.i0 < F; break;
      default: ok = false;
    }
    if (!ok) return XIVO_HIP_ERR_INVALID;
    if (o == 0 || e.b != ops[o - 1].b) { wg_filter.push_back(e.b); wg_begin.push_back(o); }
  }
  c->F = F;
  if (n_ops == 0) return XIVO_HIP_OK;
  wg_begin.push_back(n_ops);
  const int n_wg = (int)wg_filter.size();
  const size_t bytes_ops = (size_t)n_ops * sizeof(xivo_edit_op);
  const size_t bytes = bytes_ops + (size_t)(2 * n_wg + 1) * sizeof(int);
  if (bytes > c->edit_cap) {
    if (c->edit_buf) hipFree(c->edit_buf);
    c->edit_buf = nullptr; c->edit_cap = 0;
    const size_t cap = bytes * 2;
    if (hipMalloc(&c->edit_buf, cap) != hipSuccess) return XIVO_HIP_ERR_NOMEM;
    c->edit_cap = cap;
  }
  char* d = (char*)c->edit_buf;
  HIP_TRY(hipMemcpyAsync(d, ops, bytes_ops, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d + bytes_ops, wg_filter.data(), (size_t)n_wg * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d + bytes_ops + (size_t)n_wg * sizeof(int), wg_begin.data(), (size_t)(n_wg + 1) * sizeof(int),
                         hipMemcpyHostToDevice, c->stream));
  EditArgs a{};
  a.ops = (const xivo_edit_op*)d; a.wg_filter = (const int*)(d + bytes_ops); a.wg_begin = a.wg_filter + n_wg;
  a.P = c->P; a.strideP = c->sP; a.ldp = c->Np; a.Np = c->Np; a.lay = L;
  a.poses = c->poses; a.groups = c->groups; a.feats = c->feats; a.Fmax = c->Fmax;
  {
    StageTimer st(c, ST_OTHER, 0.0, "edit_batch_kernel");
    HIP_TRY((hipError_t)launch_edit_batch(a, n_wg, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));   // the host vectors above are pageable staging
  return XIVO_HIP_OK;
}

int xivo_hip_set_pixels(xivo_hip_ctx* c, int b0, int nb, int F, const double* xp) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !c->have_layout || !c->poses || F <= 0 || 2 * F > c->Mmax || !xp) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  HIP_TRY(hipSetDevice(c->device));
  int rc = ensure_gate_buffers(c, F);
  if (rc) return rc;
  rc = ensure_staging(c, (size_t)nb * F * 2);
  if (rc) return rc;
  c->F = F;
  HIP_TRY(hipMemcpyAsync(c->staging, xp, (size_t)nb * F * 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY((hipError_t)launch_set_pixels(c->feats + (size_t)b0 * c->Fmax, c->Fmax, F, c->staging, nb, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));   // xp is borrowed host memory
  return XIVO_HIP_OK;
}

int xivo_hip_get_scene(xivo_hip_ctx* c, int b0, int nb, xivo_pose_in* poses, xivo_group_in* groups, xivo_feat_in* feats) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !c->have_layout || !c->poses) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  if (poses) HIP_TRY(hipMemcpyAsync(poses, c->poses + b0, (size_t)nb * sizeof(xivo_pose_in), hipMemcpyDeviceToHost, c->stream));
  if (groups) HIP_TRY(hipMemcpyAsync(groups, c->groups + (size_t)b0 * c->lay.n_groups,
                                     (size_t)nb * c->lay.n_groups * sizeof(xivo_group_in), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (feats && c->F > 0) {
    int rc = d2h_rows(c, feats, (size_t)c->F * sizeof(xivo_feat_in), c->feats + (size_t)b0 * c->Fmax,
                      (size_t)c->Fmax * sizeof(xivo_feat_in), (size_t)c->F * sizeof(xivo_feat_in), nb);
    if (rc) return rc;
  }
  return XIVO_HIP_OK;
}

int xivo_hip_get_H(xivo_hip_ctx* c, int b, int* M_out, double* H, int ldh, double* inn, double* diagR) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b, 1) || c->M <= 0) return XIVO_HIP_ERR_INVALID;
  const int M = c->M;
  if (M_out) *M_out = M;
  if (H) {
    if (ldh < M) return XIVO_HIP_ERR_INVALID;
    { int rcd = ensure_dense(c); if (rcd) return rcd; }
    HIP_TRY(hipMemcpy2DAsync(H, (size_t)ldh * sizeof(double), c->H + (long)b * c->sH, (size_t)c->Mpmax * sizeof(double),
                             (size_t)M * sizeof(double), c->N, hipMemcpyDeviceToHost, c->stream));
  }
  if (inn) HIP_TRY(hipMemcpyAsync(inn, c->inn + (long)b * c->Mpmax, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (diagR) HIP_TRY(hipMemcpyAsync(diagR, c->diagR + (long)b * c->Mpmax, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XIVO_HIP_OK;
}

// ------------------------------------------------------------------ propagation tail
int xivo_hip_propagate_cov(xivo_hip_ctx* c, int b0, int nb, int nm, const double* Phi, const double* Pmm) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || nm <= 0 || nm > 40 || nm > c->N || !Phi || !Pmm) return XIVO_HIP_ERR_INVALID;
  if (nb == 0) return XIVO_HIP_OK;
  const size_t per = (size_t)nm * nm;
  int rc = ensure_staging(c, 2 * per * nb);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->staging, Phi, per * nb * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->staging + per * nb, Pmm, per * nb * sizeof(double), hipMemcpyHostToDevice, c->stream));
  {
    StageTimer st(c, ST_OTHER, 0.0);
    if (launch_propagate_cov(c->P, c->sP, c->Np, c->N, c->Np, nm, c->staging, c->staging + per * nb, b0, nb, c->stream))
      return XIVO_HIP_ERR_HIP;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XIVO_HIP_OK;
}

int xivo_hip_propagate(xivo_hip_ctx* c, int b0, int nb, int n_imu, const xivo_imu_in* imu, const xivo_prop_opts* o) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !c->have_layout || !c->poses || !imu || !o || n_imu <= 0 || c->N < 23 || c->lay.group_begin < 23)
    return XIVO_HIP_ERR_INVALID;
  if (c->calib_motion) return XIVO_HIP_ERR_UNSUPPORTED;   // kMotionSize > 23: xivo_hip_propagate_calib
  if (nb == 0) return XIVO_HIP_OK;
  for (size_t b = 0; b < (size_t)nb * n_imu; ++b)
    if (!(imu[b].dt > 0.0) || (o->stepsize >= 0 && o->stepsize < 1e-6)) return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  // control_stepsize (src/princedormand.cpp:26-60): Dormand-Prince with a positive cfg step and a growth factor only
  if (o->control_stepsize && (o->method != 1 || !(o->stepsize > 0) || !(o->max_scale_factor > 0))) return XIVO_HIP_ERR_INVALID;
  if (o->control_stepsize && (!c->pd_h || c->pd_h0 != o->stepsize)) {
    // the reference's function-local static `h` starts at the cfg step (:23): one per filter here
    if (!c->pd_h) { int rcd = dev_alloc(&c->pd_h, (size_t)c->Bmax); if (rcd) return rcd; }
    std::vector<double> h0((size_t)c->Bmax, o->stepsize);
    HIP_TRY(hipMemcpy(c->pd_h, h0.data(), h0.size() * sizeof(double), hipMemcpyHostToDevice));
    c->pd_h0 = o->stepsize;
  }
  const size_t per = 529;
  const size_t imu_d = ((size_t)nb * n_imu * sizeof(xivo_imu_in) + 7) / 8;      // in doubles
  int rc = ensure_staging(c, 2 * per * nb + 144 + 529 + imu_d);
  if (rc) return rc;
  double* dPhi = c->staging; double* dPmm = dPhi + per * nb; double* dQi = dPmm + per * nb; double* dQm = dQi + 144;
  xivo_imu_in* dImu = reinterpret_cast<xivo_imu_in*>(dQm + 529);
  HIP_TRY(hipMemcpyAsync(dQi, o->Qimu, 144 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(dQm, o->Qmodel, 529 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(dImu, imu, (size_t)nb * n_imu * sizeof(xivo_imu_in), hipMemcpyHostToDevice, c->stream));
  PropStateArgs a{};
  a.poses = c->poses + b0; a.imu = dImu; a.n_imu = n_imu; a.Qimu = dQi; a.Qmodel = dQm;
  a.g[0] = o->g[0]; a.g[1] = o->g[1]; a.g[2] = o->g[2]; a.method = o->method; a.stepsize = o->stepsize;
  a.P = c->P + (long)b0 * c->sP; a.strideP = c->sP; a.ldp = c->Np; a.Phi_out = dPhi; a.Pmm_out = dPmm; a.batch = nb;
  if (o->control_stepsize) {
    a.pd_h = c->pd_h + b0; a.pd_tol = o->tolerance; a.pd_min_scale = o->min_scale_factor; a.pd_max_scale = o->max_scale_factor;
  }
  {
    char plabel[64];
    snprintf(plabel, sizeof(plabel), "propagate_state_wave_kernel<%d>", a.method ? 7 : 4);
    // algorithmic flops: per integrator sub-step and stage the 23 x 23 Lyapunov right-hand side
    // F P + P F^T (2 * 2 * 23^3) and the transition recursion F + c F FK (2 * 23^3), as the reference codes them (dense);
    // sub-steps as src/rk4.cpp:19-31 cuts a sample: ceil(dt / stepsize), the sample's own dt when stepsize <= 0
    double substeps = 0.0;
    for (int s = 0; s < n_imu; ++s) substeps += o->stepsize > 0 ? std::ceil(imu[s].dt / o->stepsize) : 1.0;
    const double stage_flops = 6.0 * 23.0 * 23.0 * 23.0 + 2.0 * 23.0 * 12.0 * (12.0 + 23.0);
    StageTimer st(c, ST_PROP_STATE, (double)nb * substeps * (a.method ? 7.0 : 4.0) * stage_flops, plabel,
                  (double)nb * (3.0 * 529 + n_imu * sizeof(xivo_imu_in) / 8.0 + 60.0) * sizeof(double));
    HIP_TRY((hipError_t)launch_propagate_state(a, c->stream));
  }
  {
    // tail: reads and writes the 23 rows and 23 columns of P that change (+ Phi, P_mm)
    StageTimer st(c, ST_PROP_TAIL, (double)nb * 2.0 * (2.0 * 23.0 * 23.0 * (c->N - 23)), "propagate_cov_fixed_kernel<23>",
                  (double)nb * (4.0 * 23 * c->N + 2.0 * 529) * sizeof(double));
    HIP_TRY((hipError_t)launch_propagate_cov(c->P, c->sP, c->Np, c->N, c->Np, 23, dPhi, dPmm, b0, nb, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));   // imu / opts are borrowed host memory
  return XIVO_HIP_OK;
}

// Estimator::Propagate of an online-calibration build (kMotionSize = 24 / 38 / 39): propagate_state_calib_kernel + the
// run-time-nm tail
int xivo_hip_propagate_calib(xivo_hip_ctx* c, int b0, int nb, int n_imu, const xivo_imu_in* imu, const xivo_prop_opts* o,
                             const double* Qmodel) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (bad_range(c, b0, nb) || !c->have_layout || !c->poses || !imu || !o || !Qmodel || n_imu <= 0 || !c->calib_motion || !c->calib)
    return XIVO_HIP_ERR_INVALID;
  const int nm = c->cl.Cg >= 0 ? c->cl.Cg + 15 : c->cl.td + 1;
  if (nm > 40 || c->N < nm || c->lay.group_begin < nm) return XIVO_HIP_ERR_INVALID;
  if (o->control_stepsize && (o->method != 1 || !(o->stepsize > 0) || !(o->max_scale_factor > 0))) return XIVO_HIP_ERR_INVALID;
  if (o->control_stepsize && (!c->pd_h || c->pd_h0 != o->stepsize)) {   // (as in xivo_hip_propagate: the step every filter carries)
    if (!c->pd_h) { int rcd = dev_alloc(&c->pd_h, (size_t)c->Bmax); if (rcd) return rcd; }
    std::vector<double> h0((size_t)c->Bmax, o->stepsize);
    HIP_TRY(hipMemcpy(c->pd_h, h0.data(), h0.size() * sizeof(double), hipMemcpyHostToDevice));
    c->pd_h0 = o->stepsize;
  }
  if (nb == 0) return XIVO_HIP_OK;
  for (size_t b = 0; b < (size_t)nb * n_imu; ++b)
    if (!(imu[b].dt > 0.0) || (o->stepsize >= 0 && o->stepsize < 1e-6)) return XIVO_HIP_ERR_INVALID;
  const size_t per = (size_t)nm * nm;
  const size_t imu_d = ((size_t)nb * n_imu * sizeof(xivo_imu_in) + 7) / 8;      // in doubles
  int rc = ensure_staging(c, 2 * per * nb + 144 + per + imu_d);
  if (rc) return rc;
  double* dPhi = c->staging; double* dPmm = dPhi + per * nb; double* dQi = dPmm + per * nb; double* dQm = dQi + 144;
  xivo_imu_in* dImu = reinterpret_cast<xivo_imu_in*>(dQm + per);
  HIP_TRY(hipMemcpyAsync(dQi, o->Qimu, 144 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(dQm, Qmodel, per * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(dImu, imu, (size_t)nb * n_imu * sizeof(xivo_imu_in), hipMemcpyHostToDevice, c->stream));
  PropStateArgs a{};
  a.poses = c->poses + b0; a.imu = dImu; a.n_imu = n_imu; a.Qimu = dQi; a.Qmodel = dQm;
  a.g[0] = o->g[0]; a.g[1] = o->g[1]; a.g[2] = o->g[2]; a.method = o->method; a.stepsize = o->stepsize;
  if (o->control_stepsize) {
    a.pd_h = c->pd_h + b0; a.pd_tol = o->tolerance; a.pd_min_scale = o->min_scale_factor; a.pd_max_scale = o->max_scale_factor;
  }
  a.P = c->P + (long)b0 * c->sP; a.strideP = c->sP; a.ldp = c->Np; a.Phi_out = dPhi; a.Pmm_out = dPmm; a.batch = nb;
  a.nm = nm; a.iCg = c->cl.Cg; a.calib = c->calib + b0;
  {
    char plabel[64];
    snprintf(plabel, sizeof(plabel), "propagate_state_calib_kernel<%d>", a.method ? 7 : 4);
    StageTimer st(c, ST_PROP_STATE, 0.0, plabel);
    HIP_TRY((hipError_t)launch_propagate_state_calib(a, c->stream));
  }
  {
    StageTimer st(c, ST_PROP_TAIL, 0.0, "propagate_cov_kernel", (double)nb * (4.0 * nm * c->N + 2.0 * per) * sizeof(double));
    HIP_TRY((hipError_t)launch_propagate_cov(c->P, c->sP, c->Np, c->N, c->Np, nm, dPhi, dPmm, b0, nb, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));   // imu / opts / Qmodel are borrowed host memory
  return XIVO_HIP_OK;
}

// ------------------------------------------------------------------ device buffers for resident inputs (bench / tests)
int xivo_hip_dev_alloc(xivo_hip_ctx* c, size_t bytes, void** out) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !out || bytes == 0) return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  *out = nullptr;
  return hipMalloc(out, bytes) == hipSuccess ? XIVO_HIP_OK : XIVO_HIP_ERR_NOMEM;
}

int xivo_hip_dev_free(xivo_hip_ctx* c, void* p) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c) return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (p) HIP_TRY(hipFree(p));
  return XIVO_HIP_OK;
}

int xivo_hip_dev_upload(xivo_hip_ctx* c, void* dst, const void* src, size_t bytes, size_t total_bytes) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !dst || !src || bytes == 0 || total_bytes < bytes) return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  // replicate the uploaded block over the rest of the buffer (doubling device-to-device copies)
  for (size_t have = bytes; have < total_bytes;) {
    const size_t n = have < total_bytes - have ? have : total_bytes - have;
    HIP_TRY(hipMemcpyAsync((char*)dst + have, dst, n, hipMemcpyDeviceToDevice, c->stream));
    have += n;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return XIVO_HIP_OK;
}

// ------------------------------------------------------------------ timing
int xivo_hip_timer_begin(xivo_hip_ctx* c) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c) return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipEventRecord(c->t0, c->stream));
  return XIVO_HIP_OK;
}

int xivo_hip_timer_end(xivo_hip_ctx* c, float* ms) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !ms) return XIVO_HIP_ERR_INVALID;
  HIP_TRY(hipEventRecord(c->t1, c->stream));
  HIP_TRY(hipEventSynchronize(c->t1));
  HIP_TRY(hipEventElapsedTime(ms, c->t0, c->t1));
  return XIVO_HIP_OK;
}

int xivo_hip_profile_reset(xivo_hip_ctx* c) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c) return XIVO_HIP_ERR_INVALID;
  int rc = collect_profile(c);
  for (int i = 0; i < ST_COUNT; ++i) { c->stage_ms[i] = 0.f; c->stage_launches[i] = 0; c->stage_flops[i] = 0.0; }
  return rc;
}

int xivo_hip_profile_get(xivo_hip_ctx* c, int* n, const char** names, float* ms, int* launches, double* flops) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  if (!c || !n) return XIVO_HIP_ERR_INVALID;
  int rc = collect_profile(c);
  if (rc) return rc;
  *n = ST_COUNT;
  for (int i = 0; i < ST_COUNT; ++i) {
    if (names) names[i] = kStageNames[i];
    if (ms) ms[i] = c->stage_ms[i];
    if (launches) launches[i] = c->stage_launches[i];
    if (flops) flops[i] = c->stage_flops[i];
  }
  return XIVO_HIP_OK;
}

int xivo_hip_bench_mfma_peak(xivo_hip_ctx* c, double* out4) {
  if (c && hipSetDevice(c->device) != hipSuccess) return XIVO_HIP_ERR_HIP;
  // out4[0] = TFLOP/s with the chip full (8 workgroups / CU)
  // out4[1] = shader cycles per MFMA per SIMD, one wave per SIMD (256 workgroups)
  // out4[2] = sustained shader clock (GHz) during the full-chip run
  // out4[3] = TFLOP/s with one wave per SIMD
  if (!c || !out4) return XIVO_HIP_ERR_INVALID;
  const int iters = 2000;
  double res[2][2];
  for (int mode = 0; mode < 2; ++mode) {
    const int blocks = mode == 0 ? 256 * 8 : 256;
    HIP_TRY((hipError_t)launch_mfma_peak(c->scratch, 10, blocks, c->stream));
    HIP_TRY(hipEventRecord(c->t0, c->stream));
    HIP_TRY((hipError_t)launch_mfma_peak(c->scratch, iters, blocks, c->stream));
    HIP_TRY(hipEventRecord(c->t1, c->stream));
    HIP_TRY(hipEventSynchronize(c->t1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->t0, c->t1));
    double cyc = 0.0;
    HIP_TRY(hipMemcpy(&cyc, c->scratch + 1, sizeof(double), hipMemcpyDeviceToHost));
    const double flops = 2.0 * 16 * 16 * 4 * 8.0 * iters * 4.0 /*waves*/ * blocks;
    res[mode][0] = flops / (ms * 1e-3) / 1e12;
    res[mode][1] = cyc;
    if (mode == 0) out4[2] = cyc / (ms * 1e-3) / 1e9;  // cycles of one resident wave / wall time (lower bound)
  }
  out4[0] = res[0][0];
  out4[3] = res[1][0];
  out4[1] = res[1][1] / (8.0 * iters);
  return XIVO_HIP_OK;
}

void xivo_hip_gemm_tile(int rows, int cols, int symmetric, int* bm, int* bn) {
  int wm, wn;
  gemm_pick_tile(round_up16(rows), round_up16(cols), symmetric, &wm, &wn);
  if (bm) *bm = 32 * wm;
  if (bn) *bn = 32 * wn;
}

}  // extern "C"
