// EKF-specific kernels around the batched GEMM / Cholesky core (gfx950).
//
//  jac_instate_kernel   Feature::ComputeJacobian           src/feature.cpp:542-656
//  gate_sparse_kernel   Estimator::MHGating                src/update.cpp:50-116
//  gate_dense_kernel    same numeric core on dense J rows  src/update.cpp:60-96
//  stack_kernel         FilterUpdate stacking + Feature::FillJacobianBlock
//                                                          src/update.cpp:129-138, src/feature.cpp:658-684
//  oos_kernel           ComputeOOSJacobian(+Internal) + SlowGivens
//                                                          src/oos.cpp:8-89, src/helpers.cpp:13-23
//  propagate_cov_kernel covariance cross-block tail        src/rk4.cpp:92-102
//  subfilter_kernel     Feature::SubfilterUpdate + Criteria::Candidate(Strict) + Feature::score
//                                                          src/feature.cpp:246-297,133-142, src/options.cpp:10-33
//  absorb_error_kernel  Estimator::AbsorbError             src/estimator.cpp:875-921
//  givens_kernel        xivo::Givens / xivo::QR            src/helpers.cpp:27-101
//  p_* kernels          host edits of P_
// (all paths relative to /root/reference). These are HBM/L2-bound byte movers
// or tiny per-feature 3x3 chains: one thread / one wave64 per feature, wave
// reductions for the chi-square gating, no MFMA.
#include "ekf_kernels.h"
#include "camera_device.h"
#include "gate_device.h"
#include "p_unpack_device.h"

namespace xivo_hip {

namespace {

// ---------------------------------------------------------------- pack/unpack
// (lower triangle of the host matrix authoritative: p_unpack_device.h)
__global__ __launch_bounds__(256) void unpack_P_kernel(const double* __restrict__ raw, double* __restrict__ P, int N, int Np,
                                                      int ldp, long strideP) {
  __shared__ double tile[kPUnpackTile][kPUnpackTile + 1];
  const int f = blockIdx.y;
  p_unpack_tile_pair(raw + (long)f * N * N, N, N, P + (long)f * strideP, ldp, Np, blockIdx.x, tile);
}

__global__ void pack_P_kernel(const double* __restrict__ P, double* __restrict__ raw, int N, int ldp,
                              long strideP) {
  const int f = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)N * N) return;
  const int i = (int)(e % N), j = (int)(e / N);
  raw[(long)f * N * N + e] = P[(long)f * strideP + i + (long)j * ldp];
}

// dense padded H / H^T of the filters that do NOT fit the row-pair compressed form (only_if[f] != 0; null: all)
__global__ void unpack_meas_kernel(const double* __restrict__ rawH, long strideRaw, int ldraw,
                                   const int* __restrict__ only_if, MeasBuffers mb, int M, int Mp, int N, int Np) {
  const int f = blockIdx.y;
  if (only_if && !only_if[f]) return;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long tot = (long)Mp * Np;
  if (e < tot) {
    {  // H: m fastest
      const int m = (int)(e % Mp), n = (int)(e / Mp);
      double v = 0.0;
      if (m < M && n < N) v = rawH[(long)f * strideRaw + m + (long)n * ldraw];
      mb.H[(long)f * mb.strideH + m + (long)n * mb.ldh] = v;
    }
    {  // H^T: n fastest
      const int n = (int)(e % Np), m = (int)(e / Np);
      double v = 0.0;
      if (m < M && n < N) v = rawH[(long)f * strideRaw + m + (long)n * ldraw];
      mb.HT[(long)f * mb.strideHT + n + (long)m * mb.ldht] = v;
    }
  }
}

// ---------------------------------------------------------------- P edits
__global__ void p_zero_rc_kernel(double* P, int ldp, int Np, int off, int len) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Np) return;
  for (int r = 0; r < len; ++r) {
    P[(off + r) + (long)t * ldp] = 0.0;
    P[t + (long)(off + r) * ldp] = 0.0;
  }
}
// rows first, then columns - the order of Estimator::AddGroupToState
// (src/estimator.cpp:808-816); phase selects which.
__global__ void p_copy_rc_kernel(double* P, int ldp, int Np, int dst, int src, int len, int phase) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Np) return;
  for (int r = 0; r < len; ++r) {
    if (phase == 0) P[(dst + r) + (long)t * ldp] = P[(src + r) + (long)t * ldp];
    else P[t + (long)(dst + r) * ldp] = P[t + (long)(src + r) * ldp];
  }
}
__global__ void p_diag_kernel(const double* P, int ldp, int N, double* out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < N) out[t] = P[t + (long)t * ldp];
}

__global__ __launch_bounds__(256) void gate_dense_kernel(GateDenseArgs a) {
  const int filt = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  extern __shared__ double sdist[];  // F doubles + 1
  const double* H = a.H + (long)filt * a.strideH;
  const double* HP = a.HP + (long)filt * a.strideHP;
  double* inn = a.inn + (long)filt * a.strideInn;
  // S_f = (HP)_f H_f^T + R I2 from the TRANSPOSED copies (P H^T and H^T are [Np x Mp] with
  // the state index contiguous), so every lane streams four contiguous columns.
  const double* PHT = a.PHTr + (long)filt * a.strideHT;
  const double* HT = a.HTw + (long)filt * a.strideHT;
  // ragged batches (a.feats given): absent entries (sind < 0) were stacked as zero rows; they are no candidates - distance
  // +inf, never an inlier, not counted - and a filter gates only with more than min_inliers present entries, exactly as
  // gate_sparse_kernel does (src/manager.cpp:635)
  __shared__ int s_present;
  if (tid == 0) s_present = 0;
  __syncthreads();
  if (a.feats) {
    int cnt = 0;
    for (int f = tid; f < a.F; f += 256) cnt += a.feats[(long)filt * a.Fmax + f].sind >= 0 ? 1 : 0;
    if (cnt) atomicAdd(&s_present, cnt);
  }
  __syncthreads();
  const int present = a.feats ? s_present : a.F;
  const bool gating = !a.feats || present > a.min_inliers;
  for (int f = wave; f < a.F; f += 4) {
    const bool here = !a.feats || a.feats[(long)filt * a.Fmax + f].sind >= 0;
    if (!here || !gating) { if (lane == 0) sdist[f] = here ? 0.0 : __builtin_inf(); continue; }
    double s00 = 0, s10 = 0, s11 = 0;
    const double* p0 = PHT + (long)(2 * f) * a.ldht;
    const double* p1 = p0 + a.ldht;
    const double* h0 = HT + (long)(2 * f) * a.ldht;
    const double* h1 = h0 + a.ldht;
    for (int n = lane; n < a.Np; n += 64) {
      const double hp0 = p0[n], hp1 = p1[n], hh0 = h0[n], hh1 = h1[n];
      s00 = fma(hp0, hh0, s00);
      s10 = fma(hp1, hh0, s10);
      s11 = fma(hp1, hh1, s11);
    }
    s00 = wave_sum(s00) + a.R;
    s10 = wave_sum(s10);
    s11 = wave_sum(s11) + a.R;
    if (lane == 0) sdist[f] = mh_dist_2x2(s00, s10, s11, inn[2 * f], inn[2 * f + 1]);
  }
  __syncthreads();
  if (wave == 0) {
    // (not gating: present entries carry 0, absent ones +inf - any positive threshold keeps exactly the present ones)
    // (no_relax: a plain chi-square test against thresh - the rescue pass of OnePointRANSAC, update.cpp:352-356)
    const double th = gating ? (a.no_relax ? a.thresh : relax_threshold(sdist, a.F, a.thresh, a.mult, a.min_inliers, lane, present)) : 1.0;
    if (lane == 0) sdist[a.F] = th;
  }
  __syncthreads();
  const double th = sdist[a.F];
  for (int f = tid; f < a.F; f += 256) {
    const bool in = sdist[f] < th;
    a.mask[(long)filt * (a.mask_ld ? a.mask_ld : a.F) + f] = in ? 1 : 0;
    a.dist[(long)filt * (a.mask_ld ? a.mask_ld : a.F) + f] = (gating && sdist[f] != __builtin_inf()) ? sdist[f] : 0.0;
    if (!in) {
      inn[2 * f] = 0.0; inn[2 * f + 1] = 0.0;
      double* dr = a.diagR + (long)filt * a.strideR;
      dr[2 * f] = 1.0; dr[2 * f + 1] = 1.0;
      if (a.have_ell) {
        double* ev = a.ell.val + (long)filt * a.ell.stride_val() + (long)f * ELL_W * 2;
        for (int t = 0; t < 2 * ELL_W; ++t) ev[t] = 0.0;
      }
    }
  }
  // neutralise rejected rows of H / H^T
  double* Hw = a.Hw + (long)filt * a.strideH;
  double* HTw = a.HTw + (long)filt * a.strideHT;
  for (int f = 0; f < a.F; ++f) {
    if (sdist[f] < th) continue;
    for (int n = tid; n < a.Np; n += 256) {
      Hw[2 * f + (long)n * a.ldh] = 0.0;
      Hw[2 * f + 1 + (long)n * a.ldh] = 0.0;
      HTw[n + (long)(2 * f) * a.ldht] = 0.0;
      HTw[n + (long)(2 * f + 1) * a.ldht] = 0.0;
      if (a.HPw) {
        double* HPw = a.HPw + (long)filt * a.strideHP;
        double* PHTw = a.PHTw + (long)filt * a.strideHT;
        HPw[2 * f + (long)n * a.ldhp] = 0.0;
        HPw[2 * f + 1 + (long)n * a.ldhp] = 0.0;
        PHTw[n + (long)(2 * f) * a.ldht] = 0.0;
        PHTw[n + (long)(2 * f + 1) * a.ldht] = 0.0;
      }
    }
  }
}

// ---------------------------------------------------------------- 3x3 helpers (row-major m[i][j])
struct M3 { double m[3][3]; };
struct V3 { double v[3]; };

__device__ __forceinline__ M3 m3_from_colmajor(const double* p) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = p[i + 3 * j];
  return r;
}
__device__ __forceinline__ M3 m3_t(const M3& a) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
  return r;
}
__device__ __forceinline__ M3 m3_mul(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
__device__ __forceinline__ M3 m3_neg(const M3& a) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = -a.m[i][j];
  return r;
}
__device__ __forceinline__ M3 m3_add(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j];
  return r;
}
__device__ __forceinline__ V3 m3_mulv(const M3& a, const V3& x) {
  V3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i) r.v[i] = a.m[i][0] * x.v[0] + a.m[i][1] * x.v[1] + a.m[i][2] * x.v[2];
  return r;
}
// SO3::hat (sophus/so3.hpp): [0 -z y; z 0 -x; -y x 0]
__device__ __forceinline__ M3 hat(const V3& w) {
  M3 r;
  r.m[0][0] = 0; r.m[0][1] = -w.v[2]; r.m[0][2] = w.v[1];
  r.m[1][0] = w.v[2]; r.m[1][1] = 0; r.m[1][2] = -w.v[0];
  r.m[2][0] = -w.v[1]; r.m[2][1] = w.v[0]; r.m[2][2] = 0;
  return r;
}
// 2x3 = (2x3) * (3x3)
__device__ __forceinline__ void m23_mul(const double a[2][3], const M3& b, double out[2][3]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) out[i][j] = a[i][0] * b.m[0][j] + a[i][1] * b.m[1][j] + a[i][2] * b.m[2][j];
}

// project(Xcn) with Jacobian (common/project.h:11-24) then Camera::Project;
// returns dxp_dXcn = dxp_dxcn * dxcn_dXcn (feature.cpp:611-620, oos.cpp:66-70)
__device__ __forceinline__ void project_pixel(const xivo_cam& cam, const V3& Xcn, double xp[2],
                                              double dxp_dXcn[2][3]) {
  const double X = Xcn.v[0], Y = Xcn.v[1], Z = Xcn.v[2];
  const double xcn0 = X / Z, xcn1 = Y / Z;
  const double d[2][3] = {{1 / Z, 0, -X / (Z * Z)}, {0, 1 / Z, -Y / (Z * Z)}};
  double Jc[2][2];
  camera_project(cam, xcn0, xcn1, xp, Jc);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) dxp_dXcn[i][j] = Jc[i][0] * d[0][j] + Jc[i][1] * d[1][j];
}

// Online camera calibration (USE_ONLINE_CAMERA_CALIB): the intrinsics are state, one set per filter, resident in
// xivo_calib_in::intr (fx fy cx cy d[0..4] - the order of the state slots); the context's xivo_cam names the model
__device__ __forceinline__ xivo_cam filter_cam(const xivo_cam& cam, const xivo_calib_in* calib, int cam_dim, int filt) {
  xivo_cam c = cam;
  if (calib && cam_dim > 0) {
    const double* p = calib[filt].intr;
    c.fx = p[0]; c.fy = p[1]; c.cx = p[2]; c.cy = p[3];
#pragma unroll
    for (int k = 0; k < 5; ++k) c.d[k] = p[4 + k];
  }
  return c;
}

// Feature::Xc (feature.cpp:98-105): Xc and dXc/dx from the feature's local state x = (X/Z, Y/Z, log Z) through
// unproject_logz (project.h:79-95) or, in the USE_INVDEPTH build (XIVO_HIP_FLAG_INVDEPTH), x = (X/Z, Y/Z, 1/Z) through
// unproject_invz = project_invz (project.h:31-56). Feature::z (feature.cpp:120-126) for the depth tests.
__device__ __forceinline__ V3 feature_unproject(const double* x, int invdepth, M3& dXc_dx) {
  V3 Xc;
  if (invdepth) {
    const double r = x[2];
    Xc.v[0] = x[0] / r; Xc.v[1] = x[1] / r; Xc.v[2] = 1.0 / r;
    dXc_dx.m[0][0] = 1 / r; dXc_dx.m[0][1] = 0; dXc_dx.m[0][2] = -x[0] / (r * r);
    dXc_dx.m[1][0] = 0; dXc_dx.m[1][1] = 1 / r; dXc_dx.m[1][2] = -x[1] / (r * r);
    dXc_dx.m[2][0] = 0; dXc_dx.m[2][1] = 0; dXc_dx.m[2][2] = -1 / (r * r);
  } else {
    const double z = exp(x[2]);
    Xc.v[0] = x[0] * z; Xc.v[1] = x[1] * z; Xc.v[2] = z;
    dXc_dx.m[0][0] = z; dXc_dx.m[0][1] = 0; dXc_dx.m[0][2] = x[0] * z;
    dXc_dx.m[1][0] = 0; dXc_dx.m[1][1] = z; dXc_dx.m[1][2] = x[1] * z;
    dXc_dx.m[2][0] = 0; dXc_dx.m[2][1] = 0; dXc_dx.m[2][2] = z;
  }
  return Xc;
}
__device__ __forceinline__ double feature_depth(double x2, int invdepth) { return invdepth ? 1.0 / x2 : exp(x2); }

// ---------------------------------------------------------------- in-state Jacobian
// One thread per (filter, feature). Output J is 2 x 21 row-major with block
// order [Wsb Tsb Wbc Tbc Wsbr Tsbr x] (the 7 structural non-zero blocks of
// Feature::J_, feature.cpp:623-645).
__global__ void jac_instate_kernel(SceneBuffers sb, xivo_layout lay, xivo_cam cam_ctx, int batch) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= batch * sb.F) return;
  const int filt = t / sb.F, f = t % sb.F;
  const xivo_cam cam = filter_cam(cam_ctx, sb.calib, sb.cl.cam_dim, filt);
  const xivo_pose_in& pose = sb.poses[filt];
  const xivo_feat_in& ft = sb.feats[(long)filt * sb.Fmax + f];
  if (ft.sind < 0) {   // absent entry (ragged batches): no Jacobian, no innovation
    double* J0 = sb.J + ((long)filt * sb.Fmax + f) * 42;
    for (int i = 0; i < 42; ++i) J0[i] = 0.0;
    sb.finn[((long)filt * sb.Fmax + f) * 2] = 0.0;
    sb.finn[((long)filt * sb.Fmax + f) * 2 + 1] = 0.0;
    if (sb.Jc) { double* Jc0 = sb.Jc + ((long)filt * sb.Fmax + f) * 44; for (int i = 0; i < 44; ++i) Jc0[i] = 0.0; }
    return;
  }
  const xivo_group_in& grp = sb.groups[(long)filt * lay.n_groups + ft.ref_sind];

  const M3 Rsb = m3_from_colmajor(pose.Rsb), Rbc = m3_from_colmajor(pose.Rbc);
  const M3 Rsb_t = m3_t(Rsb), Rbc_t = m3_t(Rbc);
  const M3 Rsbr = m3_from_colmajor(grp.Rsb);
  const V3 Tsb{{pose.Tsb[0], pose.Tsb[1], pose.Tsb[2]}}, Tbc{{pose.Tbc[0], pose.Tbc[1], pose.Tbc[2]}};
  const V3 Tsbr{{grp.Tsb[0], grp.Tsb[1], grp.Tsb[2]}};

  // Xc = this->Xc(&dXc_dx) (feature.cpp:98-105, :555)
  M3 dXc_dx;
  const V3 Xc = feature_unproject(ft.x, sb.invdepth, dXc_dx);

  // feature.cpp:556-560
  V3 Xbr = m3_mulv(Rbc, Xc);
#pragma unroll
  for (int i = 0; i < 3; ++i) Xbr.v[i] += Tbc.v[i];
  V3 Xs = m3_mulv(Rsbr, Xbr);
#pragma unroll
  for (int i = 0; i < 3; ++i) Xs.v[i] += Tsbr.v[i];
  V3 dXs;
#pragma unroll
  for (int i = 0; i < 3; ++i) dXs.v[i] = Xs.v[i] - Tsb.v[i];
  const V3 Xb = m3_mulv(Rsb_t, dXs);
  V3 dXb;
#pragma unroll
  for (int i = 0; i < 3; ++i) dXb.v[i] = Xb.v[i] - Tbc.v[i];
  const V3 Xcn = m3_mulv(Rbc_t, dXb);

  // feature.cpp:563-590 (products associated left to right as Eigen does)
  const M3 dXbr_dWbc = m3_mul(m3_neg(Rbc), hat(Xc));
  const M3 dXs_dWsbr = m3_mul(m3_neg(Rsbr), hat(Xbr));
  const M3 dXb_dWsb = hat(Xb);
  const M3 dXcn_dXs = m3_mul(Rbc_t, Rsb_t);                 // dXcn_dXb * dXb_dXs
  const M3 dXcn_dXbr = m3_mul(dXcn_dXs, Rsbr);              // ... * dXs_dXbr
  const M3 dXcn_dTbc = m3_add(m3_neg(Rbc_t), dXcn_dXbr);    // :579-580 (dXbr_dTbc = I)
  const M3 dXcn_dWbc = m3_add(hat(Xcn), m3_mul(dXcn_dXbr, dXbr_dWbc));  // :581-582
  const M3 dXcn_dTsb = m3_mul(Rbc_t, m3_neg(Rsb_t));        // :585
  const M3 dXcn_dWsb = m3_mul(Rbc_t, dXb_dWsb);             // :586
  const M3 dXcn_dTsbr = dXcn_dXs;                           // :587 (dXs_dTsbr = I)
  const M3 dXcn_dWsbr = m3_mul(dXcn_dXs, dXs_dWsbr);        // :588
  const M3 dXcn_dx = m3_mul(m3_mul(dXcn_dXbr, Rbc), dXc_dx);  // :590

  double xp[2], dxp_dXcn[2][3];
  project_pixel(cam, Xcn, xp, dxp_dXcn);

  if (sb.Jc) {   // online-calibration builds: the td / Cg / bg / intrinsics blocks (feature.cpp:592-609, :611-618, :632-651)
    double* Jc = sb.Jc + ((long)filt * sb.Fmax + f) * 44;
    for (int i = 0; i < 44; ++i) Jc[i] = 0.0;
    if (sb.cl.td >= 0) {
      const xivo_calib_in& cb = sb.calib[filt];
      const M3 Cg = m3_from_colmajor(cb.Cg);
      const V3 gyro{{cb.gyro[0], cb.gyro[1], cb.gyro[2]}};
      V3 gyro_calib = m3_mulv(Cg, gyro);                                   // :593  Cg * gyro - bg
#pragma unroll
      for (int i = 0; i < 3; ++i) gyro_calib.v[i] -= pose.bg[i];
      const V3 Vsb{{pose.Vsb[0], pose.Vsb[1], pose.Vsb[2]}};
      // dXcn_dtd = -Rbc_t * (hat(gyro_calib) * Rsb_t * (Xs - Tsb) + Rsb_t * Vsb)          :594-595
      const V3 u1 = m3_mulv(m3_mul(hat(gyro_calib), Rsb_t), dXs);
      const V3 u2 = m3_mulv(Rsb_t, Vsb);
      V3 u;
#pragma unroll
      for (int i = 0; i < 3; ++i) u.v[i] = u1.v[i] + u2.v[i];
      const V3 dXcn_dtd = m3_mulv(m3_neg(Rbc_t), u);
      // dXcn_dW = dAB_dB<3,1>(Rbc_t * hat(Rsb_t * (Xs - Tsb)) * td) = that 3 x 3 matrix     :598-599
      M3 dXcn_dW = m3_mul(Rbc_t, hat(m3_mulv(Rsb_t, dXs)));
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) dXcn_dW.m[i][j] *= cb.td;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        Jc[i * 22 + 0] = dxp_dXcn[i][0] * dXcn_dtd.v[0] + dxp_dXcn[i][1] * dXcn_dtd.v[1] + dxp_dXcn[i][2] * dXcn_dtd.v[2];   // :632
        double jw[3];                                                      // dxp_dXcn * dXcn_dW
#pragma unroll
        for (int j = 0; j < 3; ++j) jw[j] = dxp_dXcn[i][0] * dXcn_dW.m[0][j] + dxp_dXcn[i][1] * dXcn_dW.m[1][j] + dxp_dXcn[i][2] * dXcn_dW.m[2][j];
        // dXcn_dCg = dXcn_dW * dW_dCg, dW_dCg row k = gyro at columns 3k..3k+2 (:601-605): column 3k + j = dXcn_dW[:, k] * gyro[j]
        if (sb.cl.Cg >= 0)
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              double acc = 0.0;                                            // (the product as Eigen forms it: dxp_dXcn * (dXcn_dW * dW_dCg))
#pragma unroll
              for (int q = 0; q < 3; ++q) acc += dxp_dXcn[i][q] * (dXcn_dW.m[q][k] * gyro.v[j]);
              Jc[i * 22 + 1 + 3 * k + j] = acc;
            }
#pragma unroll
        for (int j = 0; j < 3; ++j) Jc[i * 22 + 10 + j] = -jw[j];          // dXcn_dbg = -dXcn_dW (:607, :636)
      }
    }
    if (sb.cl.cam_dim > 0) {                                               // :611-618, :647-651
      double xq[2], Jq[2][2], jacc[2][9];
      const double xcn0 = Xcn.v[0] / Xcn.v[2], xcn1 = Xcn.v[1] / Xcn.v[2];
      camera_project_jacc(cam, xcn0, xcn1, xq, Jq, jacc);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < sb.cl.cam_dim && j < 9; ++j) Jc[i * 22 + 13 + j] = jacc[i][j];
    }
  }

  double blk[7][2][3];
  m23_mul(dxp_dXcn, dXcn_dWsb, blk[0]);
  m23_mul(dxp_dXcn, dXcn_dTsb, blk[1]);
  m23_mul(dxp_dXcn, dXcn_dWbc, blk[2]);
  m23_mul(dxp_dXcn, dXcn_dTbc, blk[3]);
  m23_mul(dxp_dXcn, dXcn_dWsbr, blk[4]);
  m23_mul(dxp_dXcn, dXcn_dTsbr, blk[5]);
  m23_mul(dxp_dXcn, dXcn_dx, blk[6]);

  double* J = sb.J + ((long)filt * sb.Fmax + f) * 42;
#pragma unroll
  for (int b = 0; b < 7; ++b)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) J[i * 21 + 3 * b + j] = blk[b][i][j];
  double* inn = sb.finn + ((long)filt * sb.Fmax + f) * 2;
  inn[0] = ft.xp[0] - xp[0];   // feature.cpp:654-655
  inn[1] = ft.xp[1] - xp[1];
}

// column of the error state that compact-J column c (0..20) maps to
__device__ __forceinline__ int jcol(const xivo_layout& lay, const xivo_feat_in& ft, int c) {
  const int b = c / 3, o = c % 3;
  switch (b) {
    case 0: return 0 + o;    // Index::Wsb  (core.h:41)
    case 1: return 3 + o;    // Index::Tsb
    case 2: return 15 + o;   // Index::Wbc
    case 3: return 18 + o;   // Index::Tbc
    case 4: return lay.group_begin + 6 * ft.ref_sind + o;
    case 5: return lay.group_begin + 6 * ft.ref_sind + 3 + o;
    default: return lay.feature_begin + 3 * ft.sind + o;
  }
}

// res^T (J P J^T + R I2)^-1 res of one feature by one wave64 (src/update.cpp:60-70, :352-356): the 21 x 21 sub-block of
// P the full row J touches, lane (a, c) forming (P J^T)(a, c) in ascending b, reduced across lanes, 2x2 LLT. The value is
// valid in every lane. The gathers of P are issued as seven wave-wide loads: the wave parks the sub-block (element
// e = a + 21 b in lane e mod 64) and the two rows of J in its own LDS scratch and reads its operands from there. (The first
// version had every lane (a, c) gather its 21 elements itself - 42 gather instructions per feature, half of them duplicates
// between the c = 0 and c = 1 lanes - and the gate kernel was bound by the number of 8-byte gathers a CU's address unit
// retires, not by memory latency: 0.40 -> 0.28 ms per 4096 filters x 60 features, same bits.) scratch: 441 + 42 doubles.
__device__ __forceinline__ double feature_chi2_lds(const double* P, int ldp, const xivo_layout& lay, const xivo_feat_in& ft,
                                                   const double* J, const double* inn, double R, int lane, double* scratch) {
  double* sP = scratch;          // [a + 21 b]
  double* sJ = scratch + 441;    // [c * 21 + b]
  double pv[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int e = lane + 64 * k;
    const int ea = e < 441 ? e % 21 : 0, eb = e < 441 ? e / 21 : 0;
    pv[k] = P[jcol(lay, ft, ea) + (long)jcol(lay, ft, eb) * ldp];
  }
  const double jmine = lane < 42 ? J[lane] : 0.0;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int e = lane + 64 * k;
    if (e < 441) sP[e] = pv[k];
  }
  if (lane < 42) sJ[lane] = jmine;
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes are done (one wave, no barrier needed)
  __builtin_amdgcn_wave_barrier();
  double v = 0.0;
  const int ra = lane % 21, rc = lane / 21;
  if (lane < 42) {
#pragma unroll
    for (int b = 0; b < 21; ++b) v = fma(sP[ra + 21 * b], sJ[rc * 21 + b], v);
  }
  const double j0 = lane < 42 ? sJ[ra] : 0.0, j1 = lane < 42 ? sJ[21 + ra] : 0.0;
  __builtin_amdgcn_wave_barrier();      // (the next feature's writes stay behind these reads: program order within the wave)
  double s00 = (lane < 21) ? j0 * v : 0.0;
  double s10 = (lane < 21) ? j1 * v : 0.0;
  double s11 = (lane >= 21 && lane < 42) ? j1 * v : 0.0;
  s00 = wave_sum(s00) + R;
  s10 = wave_sum(s10);
  s11 = wave_sum(s11) + R;
  return mh_dist_2x2(s00, s10, s11, inn[0], inn[1]);
}

// One workgroup per filter (4 waves; 16 for fewer than 256 filters - latency); a wave64 per feature computes S = J P J^T + R I2
// from the 21 x 21 sub-block of P the feature touches (J is structurally
// sparse), reduces it across lanes, and the 2x2 LLT gives the Mahalanobis
// distance. Then one wave runs the threshold-relaxation loop.
// Online-calibration builds: J() has 22 more columns EVERY feature shares - td, Cg (9), bg (3), the intrinsics (9 slots) - next
// to the 21 of the default build (src/feature.cpp:623-651); 12 of those 21 (Wsb, Tsb, Wbc, Tbc) are shared as well. The 34 x 34
// block of P on the shared columns is therefore the same for all features of a filter: the workgroup parks it in LDS once,
// and a feature gathers only its 9 private columns (group, feature) against the shared ones and themselves - 387 elements,
// seven wave-wide loads as in the default build, instead of the 43 x 43 = 1849 of a gather per feature (2.0 -> 0.46 ms per 4096
// filters x 60 features at N = 276). The sums run over the 43 columns in the order of the whole row, as before.
constexpr int WIDE_NS = 34, WIDE_NP = 9, WIDE_NC = 43;
constexpr int WIDE_X = WIDE_NS * WIDE_NP, WIDE_Y = WIDE_NP * WIDE_NP;         // P[shared, private] | P[private, private]
constexpr int WIDE_SCR = WIDE_X + WIDE_Y + 2 * WIDE_NC + 1;                   // doubles of LDS scratch per wave (474)
constexpr int WIDE_PSS = WIDE_NS * WIDE_NS;                                   // doubles of the per-filter shared block
__device__ __forceinline__ int wide_scol(const xivo_calib_layout& cl, int s) {   // state column of shared slot s
  if (s < 6) return s;                                    // Index::Wsb, Tsb
  if (s < 12) return 15 + (s - 6);                        // Index::Wbc, Tbc
  const int k = s - 12;                                   // the layout of Jc: td | Cg 9 | bg 3 | intrinsics 9
  if (k == 0) return cl.td >= 0 ? cl.td : 0;              // (a block that is switched off carries zeros in Jc: any valid column will do)
  if (k < 10) return cl.Cg >= 0 ? cl.Cg + (k - 1) : 0;
  if (k < 13) return 9 + (k - 10);                        // Index::bg
  return (k - 13) < cl.cam_dim ? cl.cam_begin + (k - 13) : 0;
}
__device__ __forceinline__ double feature_chi2_wide(const double* P, int ldp, const xivo_layout& lay, const xivo_calib_layout& cl,
                                                    const xivo_feat_in& ft, const double* J, const double* Jc, const double* inn, double R,
                                                    int lane, const double* sPss, double* scratch) {
  double* X = scratch;                        // [s + 34 p] = P[shared s, private p]
  double* Y = scratch + WIDE_X;               // [p + 9 q]
  double* sJ = scratch + WIDE_X + WIDE_Y;     // [row * 43 + w], w in the order of the whole row: 12 common | 9 private | 22 calibration
  auto pcol = [&](int q) -> int { return q < 6 ? lay.group_begin + 6 * ft.ref_sind + q : lay.feature_begin + 3 * ft.sind + (q - 6); };
  double pv[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int e = lane + 64 * k;
    int row = 0, col = 0;
    if (e < WIDE_X) { row = wide_scol(cl, e % WIDE_NS); col = pcol(e / WIDE_NS); }
    else if (e < WIDE_X + WIDE_Y) { const int q = e - WIDE_X; row = pcol(q % WIDE_NP); col = pcol(q / WIDE_NP); }
    pv[k] = P[row + (long)col * ldp];
  }
  double j0 = 0.0, j1 = 0.0;
  if (lane < WIDE_NC) { j0 = lane < 21 ? J[lane] : Jc[lane - 21]; j1 = lane < 21 ? J[21 + lane] : Jc[22 + lane - 21]; }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int e = lane + 64 * k;
    if (e < WIDE_X + WIDE_Y) scratch[e] = pv[k];
  }
  if (lane < WIDE_NC) { sJ[lane] = j0; sJ[WIDE_NC + lane] = j1; }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  double v0 = 0.0, v1 = 0.0;
  if (lane < WIDE_NC) {
    // lane a of the whole row: shared (a < 12 or a >= 21: slot a or a - 9) or private (slot a - 12)
    const bool ash = lane < 12 || lane >= 21;
    const int as = lane < 12 ? lane : lane - 9, ap = lane - 12;
    const double* ps = ash ? sPss + as : X + WIDE_NS * ap;   // P[a, shared b]: step over b
    const int ss = ash ? WIDE_NS : 1;
    const double* pp = ash ? X + as : Y + ap;                // P[a, private b]
    const int sp = ash ? WIDE_NS : WIDE_NP;
    for (int b = 0; b < 12; ++b) { const double p = ps[b * ss]; v0 = fma(p, sJ[b], v0); v1 = fma(p, sJ[WIDE_NC + b], v1); }
    for (int b = 12; b < 21; ++b) { const double p = pp[(b - 12) * sp]; v0 = fma(p, sJ[b], v0); v1 = fma(p, sJ[WIDE_NC + b], v1); }
    for (int b = 21; b < WIDE_NC; ++b) { const double p = ps[(b - 9) * ss]; v0 = fma(p, sJ[b], v0); v1 = fma(p, sJ[WIDE_NC + b], v1); }
  }
  __builtin_amdgcn_wave_barrier();
  const double s00 = wave_sum(j0 * v0) + R;
  const double s10 = wave_sum(j1 * v0);
  const double s11 = wave_sum(j1 * v1) + R;
  return mh_dist_2x2(s00, s10, s11, inn[0], inn[1]);
}

__global__ __launch_bounds__(1024) void gate_sparse_kernel(GateArgs a) {
  const int filt = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = blockDim.x, nw = nt >> 6;   // 4 waves per filter for a big batch, 16 when few filters must finish fast
  extern __shared__ double sdist[];
  const SceneBuffers& sb = a.sb;
  const double* P = a.P + (long)filt * a.strideP;
  // entries present in this filter (sind >= 0); Estimator::OutlierRejection gates only when there are more
  // than min_required_inliers_ of them (src/manager.cpp:635)
  __shared__ int s_present;
  if (tid == 0) s_present = 0;
  __syncthreads();
  // the two slot indices of every entry, parked in LDS by this pass: the per-feature loop below then starts its gathers of
  // P and J right away instead of behind a load of the entry (two dependent memory round trips per feature, fifteen
  // features per wave one after the other, were what the kernel's time was)
  int* s_slot = reinterpret_cast<int*>(sdist + sb.F + 1);   // [2 F]: sind, ref_sind
  double* s_pss = sdist + sb.F + 1 + (2 * sb.F + 1) / 2;    // online-calibration builds: P on the 34 shared columns
  double* s_scr = s_pss + (sb.Jc ? WIDE_PSS : 0) + (long)wave * (sb.Jc ? WIDE_SCR : 484);   // per wave: feature_chi2_lds / _wide scratch
  if (sb.Jc && a.use_gating) {
    for (int e = tid; e < WIDE_PSS; e += nt) s_pss[e] = P[wide_scol(sb.cl, e % WIDE_NS) + (long)wide_scol(sb.cl, e / WIDE_NS) * a.ldp];
  }
  {
    int cnt = 0;
    for (int f = tid; f < sb.F; f += nt) {
      const xivo_feat_in& ft = sb.feats[(long)filt * sb.Fmax + f];
      const int si = ft.sind, rs = ft.ref_sind;
      s_slot[2 * f] = si; s_slot[2 * f + 1] = rs;
      cnt += si >= 0 ? 1 : 0;
    }
    if (cnt) atomicAdd(&s_present, cnt);
  }
  __syncthreads();
  const int present = s_present;
  const bool gating = a.use_gating && present > a.min_inliers;
  if (gating) {
    for (int f = wave; f < sb.F; f += nw) {
      xivo_feat_in ft;                      // only the slots are read by feature_chi2 / jcol
      ft.sind = s_slot[2 * f]; ft.ref_sind = s_slot[2 * f + 1];
      if (ft.sind < 0) { if (lane == 0) sdist[f] = __builtin_inf(); continue; }
      const double* J = sb.J + ((long)filt * sb.Fmax + f) * 42;
      const double* inn = sb.finn + ((long)filt * sb.Fmax + f) * 2;
      const double d = sb.Jc ? feature_chi2_wide(P, a.ldp, a.lay, sb.cl, ft, J, sb.Jc + ((long)filt * sb.Fmax + f) * 44, inn, a.R, lane, s_pss, s_scr)
                             : feature_chi2_lds(P, a.ldp, a.lay, ft, J, inn, a.R, lane, s_scr);
      if (lane == 0) sdist[f] = d;
    }
    __syncthreads();
    if (wave == 0) {
      const double th = relax_threshold(sdist, sb.F, a.thresh, a.mult, a.min_inliers, lane, present);
      if (lane == 0) sdist[sb.F] = th;
    }
    __syncthreads();
  }
  const double th = gating ? sdist[sb.F] : 0.0;
  for (int f = tid; f < sb.F; f += nt) {
    const bool here = s_slot[2 * f] >= 0;
    const bool in = gating ? (sdist[f] < th) : here;
    sb.mask[(long)filt * sb.Fmax + f] = in ? 1 : 0;
    sb.dist[(long)filt * sb.Fmax + f] = (gating && here) ? sdist[f] : 0.0;
  }
}

// Stack H (and H^T), inn, diagR for one filter: rows 2f, 2f+1 belong to feature
// f; a rejected feature keeps its two rows but they are neutral (H row = 0,
// inn = 0, diagR = 1), which is algebraically the reference's "row not stacked".
__global__ __launch_bounds__(256) void stack_kernel(StackArgs a) {
  const int filt = blockIdx.x, tid = threadIdx.x;
  const SceneBuffers& sb = a.sb;
  double* H = a.mb.H + (long)filt * a.mb.strideH;
  double* HT = a.mb.HT + (long)filt * a.mb.strideHT;
  // H_.setZero(total_size, N) (update.cpp:130)
  if (a.write_dense) {
    // (a.mb.HT == nullptr: the consumer never reads the transposed copy - the re-associated dense pipeline)
    if (a.mb.ldh == a.Mp && a.mb.ldht == a.Np) {
      // both copies are contiguous Mp x Np blocks (multiples of 16 doubles): one flat pass of 16-byte stores each
      d2* h2 = reinterpret_cast<d2*>(H);
      d2* t2 = reinterpret_cast<d2*>(HT);
      const long n2 = (long)a.Mp * a.Np / 2;
      if (a.mb.HT) for (long e = tid; e < n2; e += 256) { h2[e] = d2{0.0, 0.0}; t2[e] = d2{0.0, 0.0}; }
      else for (long e = tid; e < n2; e += 256) h2[e] = d2{0.0, 0.0};
    } else {
      for (int n = 0; n < a.Np; ++n)
        for (int m = tid; m < a.Mp; m += 256) H[m + (long)n * a.mb.ldh] = 0.0;
      if (a.mb.HT)
        for (int m = 0; m < a.Mp; ++m)
          for (int n = tid; n < a.Np; n += 256) HT[n + (long)m * a.mb.ldht] = 0.0;
    }
  }
  double* inn = a.mb.inn + (long)filt * a.mb.strideInn;
  double* dR = a.mb.diagR + (long)filt * a.mb.strideR;
  for (int m = tid; m < a.Mp; m += 256) { inn[m] = 0.0; dR[m] = 1.0; }
  __syncthreads();
  for (int f = tid; f < sb.F; f += 256) {
    if (!sb.mask[(long)filt * sb.Fmax + f]) continue;
    const xivo_feat_in& ft = sb.feats[(long)filt * sb.Fmax + f];
    const double* J = sb.J + ((long)filt * sb.Fmax + f) * 42;
    const double* fi = sb.finn + ((long)filt * sb.Fmax + f) * 2;
    for (int b = 0; b < 7; ++b) {
      // Feature::FillJacobianBlock: the group-rotation block is overwritten by the
      // group-translation block and goff+3.. stays zero (feature.cpp:675-676)
      int src = b;
      if (!a.fix_group_block) {
        if (b == 4) src = 5;
        else if (b == 5) continue;
      }
      for (int o = 0; o < 3; ++o) {
        const int col = jcol(a.lay, ft, 3 * b + o);
        for (int i = 0; i < 2; ++i) {
          const double v = J[i * 21 + 3 * src + o];
          if (!a.write_dense) continue;
          H[(2 * f + i) + (long)col * a.mb.ldh] = v;
          if (a.mb.HT) HT[col + (long)(2 * f + i) * a.mb.ldht] = v;
        }
      }
    }
    if (sb.Jc && a.write_dense) {   // online-calibration builds: Feature::FillJacobianBlock :664-670, :679-683
      const double* Jc = sb.Jc + ((long)filt * sb.Fmax + f) * 44;
      auto put = [&](int col, int i, double v) {
        H[(2 * f + i) + (long)col * a.mb.ldh] = v;
        if (a.mb.HT) HT[col + (long)(2 * f + i) * a.mb.ldht] = v;
      };
      for (int i = 0; i < 2; ++i) {
        if (sb.cl.td >= 0) {
          put(sb.cl.td, i, Jc[i * 22]);
          if (sb.cl.Cg >= 0) for (int j = 0; j < 9; ++j) put(sb.cl.Cg + j, i, Jc[i * 22 + 1 + j]);
          for (int j = 0; j < 3; ++j) put(9 + j, i, Jc[i * 22 + 10 + j]);          // Index::bg
        }
        for (int j = 0; j < sb.cl.cam_dim && j < 9; ++j) put(sb.cl.cam_begin + j, i, Jc[i * 22 + 13 + j]);
      }
    }
    inn[2 * f] = fi[0]; inn[2 * f + 1] = fi[1];      // update.cpp:136
    dR[2 * f] = a.R; dR[2 * f + 1] = a.R;            // update.cpp:137
  }
  if (tid == 0 && a.rows_instate) a.rows_instate[filt] = 2 * sb.F;
  if (!a.emit_ell) return;
  // row-pair compressed form: the 12 sensor pose / extrinsics columns are the common slots, the
  // group and feature blocks the private ones (ell.h)
  int* eidx = a.ell.idx + (long)filt * a.ell.stride_idx();
  double* eval = a.ell.val + (long)filt * a.ell.stride_val();
  // (calibration blocks: dense rows - or, with a.lead, the "leading dense block" next to compressed rows)
  if (tid == 0) { a.ell.nc[filt] = 12; a.ell.over[filt] = (sb.Jc && !a.lead) ? 1 : 0; a.ell.pw[filt] = a.fix_group_block ? 9 : 6; }
  if (sb.Jc && a.lead) {
    // Online-calibration builds on the sparse pipeline: the td / Cg / bg / intrinsics blocks of FillJacobianBlock
    // (feature.cpp:664-670, :679-683) are columns EVERY row pair shares - more of them than the compressed form has common
    // slots - and all lie in the leading lead_k state columns: they go into a dense [Mp x lead_k] block of their own (zero
    // wherever the compressed rows hold the column: Wsb, Tsb, Wbc, Tbc), which the update multiplies by two skinny GEMMs
    double* L = a.lead + (long)filt * a.strideLead;
    for (int e = tid; e < a.Mp * a.lead_k; e += 256) {
      const int m = e % a.Mp, k = e / a.Mp, f = m >> 1, i = m & 1;
      double v = 0.0;
      if (f < sb.F && sb.mask[(long)filt * sb.Fmax + f]) {
        const double* Jc = sb.Jc + ((long)filt * sb.Fmax + f) * 44 + i * 22;
        if (sb.cl.td >= 0) {
          if (k == sb.cl.td) v = Jc[0];
          else if (sb.cl.Cg >= 0 && k >= sb.cl.Cg && k < sb.cl.Cg + 9) v = Jc[1 + k - sb.cl.Cg];
          else if (k >= 9 && k < 12) v = Jc[10 + k - 9];
        }
        if (k >= sb.cl.cam_begin && k < sb.cl.cam_begin + sb.cl.cam_dim) v = Jc[13 + k - sb.cl.cam_begin];
      }
      L[m + (long)k * a.Mp] = v;
    }
  }
  // one thread per (pair, slot): consecutive threads write consecutive 16-byte value slots / 4-byte index slots (a thread
  // per pair wrote 84 scalars 448 bytes apart from its neighbour's: 0.32 ms per 4096 filters, bound by the store count)
  const d2 zero2 = d2{0.0, 0.0};
  for (int e = tid; e < (a.Mp / 2) * ELL_W; e += 256) {
    const int p = e / ELL_W, t = e % ELL_W;
    const bool on = p < sb.F && sb.mask[(long)filt * sb.Fmax + p];
    const xivo_feat_in& ft = sb.feats[(long)filt * sb.Fmax + (p < sb.F ? p : 0)];
    const double* J = sb.J + ((long)filt * sb.Fmax + (p < sb.F ? p : 0)) * 42;
    int idx = 0, c = -1;        // c: compact-J column whose two values fill the slot
    if (t < 12) { idx = jcol(a.lay, ft, t); if (on) c = t; }
    else if (t >= ELL_CW && on) {
      const int k = t - ELL_CW;
      if (a.fix_group_block) { if (k < 9) { idx = jcol(a.lay, ft, 12 + k); c = 12 + k; } }
      // Feature::FillJacobianBlock as coded: the group-rotation block is overwritten by the group-translation block
      // (feature.cpp:675-676): columns of block 4 carry the values of block 5, block 5 contributes no slots
      else if (k < 3) { idx = jcol(a.lay, ft, 12 + k); c = 15 + k; }
      else if (k < 6) { idx = jcol(a.lay, ft, 15 + k); c = 15 + k; }
    }
    eidx[e] = idx;
    reinterpret_cast<d2*>(eval)[e] = c >= 0 ? d2{J[c], J[21 + c]} : zero2;
  }
}

// ---------------------------------------------------------------- depth sub-filter
// One thread per (filter, feature): every product below is 3x3 / 2x3 / 2x2 and is written in the
// reference's association order (feature.cpp:246-297).
__global__ void subfilter_kernel(xivo_subfilter_feat* feats, int n, const xivo_pose_in* poses,
                                 const xivo_group_in* groups, int n_groups, xivo_cam cam_ctx, xivo_subfilter_opts o,
                                 int batch, const xivo_calib_in* calib, int cam_dim, int invdepth) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= batch * n) return;
  const int filt = t / n;
  const xivo_cam cam = filter_cam(cam_ctx, calib, cam_dim, filt);
  xivo_subfilter_feat& f = feats[t];
  const xivo_pose_in& pose = poses[filt];
  const xivo_group_in& grp = groups[(long)filt * n_groups + f.ref_sind];
  const M3 Rsb = m3_from_colmajor(pose.Rsb), Rbc = m3_from_colmajor(pose.Rbc), Rsbr = m3_from_colmajor(grp.Rsb);
  const V3 Tsb{{pose.Tsb[0], pose.Tsb[1], pose.Tsb[2]}}, Tbc{{pose.Tbc[0], pose.Tbc[1], pose.Tbc[2]}};
  const V3 Tsbr{{grp.Tsb[0], grp.Tsb[1], grp.Tsb[2]}};
  const int init_counter = f.init_counter + 1;                                   // :256
  // Xc(&dXc_dx) (:258; feature.cpp:98-105)
  M3 dXc_dx;
  const V3 Xc = feature_unproject(f.x, invdepth, dXc_dx);
  // gtot = (gsb * gbc)^-1 * ref.gsb * gbc   (:260)
  const M3 Rsc = m3_mul(Rsb, Rbc), Rrc = m3_mul(Rsbr, Rbc);
  V3 Tsc = m3_mulv(Rsb, Tbc), Trc = m3_mulv(Rsbr, Tbc);
  V3 dT;
#pragma unroll
  for (int i = 0; i < 3; ++i) { Tsc.v[i] += Tsb.v[i]; Trc.v[i] += Tsbr.v[i]; dT.v[i] = Trc.v[i] - Tsc.v[i]; }
  const M3 Rsc_t = m3_t(Rsc);
  const M3 Rtot = m3_mul(Rsc_t, Rrc);
  const V3 Ttot = m3_mulv(Rsc_t, dT);
  V3 Xcn = m3_mulv(Rtot, Xc);
#pragma unroll
  for (int i = 0; i < 3; ++i) Xcn.v[i] += Ttot.v[i];                              // :261
  double xp[2], dxp_dXcn[2][3];
  project_pixel(cam, Xcn, xp, dxp_dXcn);                                          // :263-267 (dxp_dxcn * dxcn_dXcn)
  double tmp[2][3], H[2][3];
  m23_mul(dxp_dXcn, Rtot, tmp);
  m23_mul(tmp, dXc_dx, H);                                                        // :269
  const double inn0 = f.xp[0] - xp[0], inn1 = f.xp[1] - xp[1];
  M3 P = m3_from_colmajor(f.P);
  // S = H P H^T + Rtri I  (:272-275)
  double HPm[2][3];
  m23_mul(H, P, HPm);
  double S[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) S[i][j] = HPm[i][0] * H[j][0] + HPm[i][1] * H[j][1] + HPm[i][2] * H[j][2];
  S[0][0] += o.Rtri; S[1][1] += o.Rtri;
  // ratio = inn . S^-1 inn / MH_thresh  (:277; 2x2 LDL^T without the pivot search, same value to rounding)
  double outlier = f.outlier_counter;
  {
    const double l10 = S[1][0] / S[0][0], d1 = S[1][1] - l10 * S[0][1];
    const double y1 = inn1 - l10 * inn0;
    const double s1 = y1 / d1, s0 = (inn0 - S[0][1] * s1) / S[0][0];
    const double ratio = (inn0 * s0 + inn1 * s1) / o.MH_thresh;
    if (ratio > 1) {                                                              // :279-285
      S[0][0] += o.Rtri * (ratio - 1); S[1][1] += o.Rtri * (ratio - 1);
      outlier += sqrt(ratio);
    } else {
      outlier = 0.0;
    }
  }
  // K = P H^T S^-1 (:287; Eigen's 2x2 inverse = adjugate / determinant)
  const double det = S[0][0] * S[1][1] - S[0][1] * S[1][0], idet = 1.0 / det;
  const double Si[2][2] = {{S[1][1] * idet, -S[0][1] * idet}, {-S[1][0] * idet, S[0][0] * idet}};
  double PHt[3][2], K[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) PHt[i][j] = P.m[i][0] * H[j][0] + P.m[i][1] * H[j][1] + P.m[i][2] * H[j][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) K[i][j] = PHt[i][0] * Si[0][j] + PHt[i][1] * Si[1][j];
  double xn[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) xn[i] = f.x[i] + (K[i][0] * inn0 + K[i][1] * inn1);   // :289
  M3 A;                                                                           // I - K H (:290)
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) A.m[i][j] = (i == j ? 1.0 : 0.0) - (K[i][0] * H[0][j] + K[i][1] * H[1][j]);
  const M3 AP = m3_mul(A, P);
  M3 Pn;                                                                          // :291
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      Pn.m[i][j] = (AP.m[i][0] * A.m[j][0] + AP.m[i][1] * A.m[j][1] + AP.m[i][2] * A.m[j][2]) +
                   ((K[i][0] * o.Rtri) * K[j][0] + (K[i][1] * o.Rtri) * K[j][1]);
  const int status = init_counter > o.ready_steps ? XIVO_FEAT_READY : XIVO_FEAT_INITIALIZING;   // :293-297
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    f.x[i] = xn[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) f.P[i + 3 * j] = Pn.m[i][j];
  }
  f.outlier_counter = outlier; f.init_counter = init_counter; f.status = status;
  // Criteria::Candidate / CandidateStrict (options.cpp:10-33), Feature::score (feature.cpp:133-142)
  const double zed = feature_depth(xn[2], invdepth);                              // Feature::z (feature.cpp:120-126)
  const bool ok = outlier < o.max_subfilter_outlier && zed > o.min_depth && zed < o.max_depth;
  f.candidate = (ok ? 1 : 0) | ((ok && status == XIVO_FEAT_READY) ? 2 : 0);
  f.score = -Pn.m[2][2];
}

// ---------------------------------------------------------------- Givens / QR
// givens(a, b), helpers.cpp:27-46 (G&VL Alg. 5.1.3; eps = 1e-4f, common/alias.h:80): G = [c s; -s c]
__device__ __forceinline__ void givens_cs(double a, double b, double& c, double& s) {
  const double eps = (double)1e-4f;
  if (fabs(b) < eps) { c = 1.0; s = 0.0; return; }
  if (fabs(b) > fabs(a)) { const double t = -a / b; s = 1.0 / sqrt(1.0 + t * t); c = s * t; }
  else { const double t = -b / a; c = 1.0 / sqrt(1.0 + t * t); s = c * t; }
}

// One wave per problem. The rotations of one column sweep run bottom-up and each touches rows (r, r+1): every
// lane owns the matrix columns j = lane (mod 64) and carries the current row r+1 of its columns in registers,
// so within a sweep each element is loaded once and stored once, and the pivot pair (a, b) of the sweep's
// column comes from its owner lane by a shuffle - no lane ever reads what another lane wrote.
//   qr = 0  xivo::Givens: pivots from Hf [rows x nf]; rotated: Hf (all nf columns), Hx (only its first nf
//           columns - helpers.cpp:64 as coded), x; then rows 0.. are replaced by rows nf.. (helpers.cpp:69-73)
//   qr = 1  xivo::QR: pivots from Hx; rotated: Hx (all nx columns), x (helpers.cpp:78-101)
__global__ __launch_bounds__(64) void givens_kernel(GivensArgs a) {
  constexpr int MAXC = 8;                                  // column chunks of 64: nx <= 512 for QR
  const int prob = blockIdx.x, lane = threadIdx.x;
  double* x = a.x + (long)prob * a.rows;
  double* Hx = a.Hx + (long)prob * a.rows * a.nx;
  double* P = a.qr ? Hx : a.Hf + (long)prob * a.rows * a.nf;   // pivot matrix
  const int pc = a.qr ? a.nx : a.nf;                            // pivot / elimination columns
  const int rows = a.eff < 0 ? a.rows : a.eff;
  const long ld = a.rows;
  const int nchunk = (pc + 63) / 64;
  for (int c = 0; c < pc && c < rows - 1; ++c) {
    const int owner = c & 63, och = c >> 6;
    // carries: row r+1 of the columns this lane owns, in the pivot matrix and (Givens) in Hx, and of x
    double cp[MAXC], chx = 0.0, cx = 0.0;
#pragma unroll
    for (int q = 0; q < MAXC; ++q) { const int j = lane + 64 * q; cp[q] = (q < nchunk && j < pc) ? P[(rows - 1) + ld * j] : 0.0; }
    if (!a.qr && lane < a.nf && lane < a.nx) chx = Hx[(rows - 1) + ld * lane];
    if (lane == 0) cx = x[rows - 1];
    for (int r = rows - 2; r >= c; --r) {
      double pa = 0.0, pb = 0.0;
#pragma unroll
      for (int q = 0; q < MAXC; ++q) if (q == och) { pa = P[r + ld * c]; pb = cp[q]; }   // meaningful on the owner lane only
      pa = __shfl(pa, owner); pb = __shfl(pb, owner);
      double cs, sn;
      givens_cs(pa, pb, cs, sn);
      // Gt = givens(a, b)^T = [c -s; s c]:  row_r <- c row_r - s row_r+1 ;  row_r+1 <- s row_r + c row_r+1
#pragma unroll
      for (int q = 0; q < MAXC; ++q) {
        const int j = lane + 64 * q;
        if (q < nchunk && j < pc) {
          const double top = P[r + ld * j], bot = cp[q];
          P[(r + 1) + ld * j] = sn * top + cs * bot;
          cp[q] = cs * top - sn * bot;
        }
      }
      if (!a.qr && lane < a.nf && lane < a.nx) {
        const double top = Hx[r + ld * lane], bot = chx;
        Hx[(r + 1) + ld * lane] = sn * top + cs * bot;
        chx = cs * top - sn * bot;
      }
      if (lane == 0) {
        const double top = x[r], bot = cx;
        x[r + 1] = sn * top + cs * bot;
        cx = cs * top - sn * bot;
      }
    }
#pragma unroll
    for (int q = 0; q < MAXC; ++q) { const int j = lane + 64 * q; if (q < nchunk && j < pc) P[c + ld * j] = cp[q]; }
    if (!a.qr && lane < a.nf && lane < a.nx) Hx[c + ld * lane] = chx;
    if (lane == 0) x[c] = cx;
  }
  if (!a.qr) {   // strip the first nf rows (helpers.cpp:69-73); increasing r reads rows not yet overwritten
    for (int r = 0; r < rows - a.nf; ++r) {
      for (int j = lane; j < a.nx; j += 64) Hx[r + ld * j] = Hx[(r + a.nf) + ld * j];
      if (lane < a.nf) P[r + ld * lane] = P[(r + a.nf) + ld * lane];
      if (lane == 0) x[r] = x[r + a.nf];
    }
  }
}

// ---------------------------------------------------------------- AbsorbError
// SO3::exp (Rodrigues), as SO3_from_rotvec (src/helpers.cpp:374-378)
__device__ __forceinline__ M3 so3_exp_dev(double wx, double wy, double wz) {
  const double th = sqrt(wx * wx + wy * wy + wz * wz);
  const V3 w{{wx, wy, wz}};
  const M3 W = hat(w), W2 = m3_mul(W, W);
  const double a = th < 1e-10 ? 1.0 : sin(th) / th, b = th < 1e-10 ? 0.5 : (1.0 - cos(th)) / (th * th);
  M3 R;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R.m[i][j] = (i == j ? 1.0 : 0.0) + a * W.m[i][j] + b * W2.m[i][j];
  return R;
}
__device__ __forceinline__ void rot_retract(double* Rcm, double wx, double wy, double wz) {   // R <- R exp(w), column-major storage
  const M3 R = m3_mul(m3_from_colmajor(Rcm), so3_exp_dev(wx, wy, wz));
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Rcm[i + 3 * j] = R.m[i][j];
}
// The periodic re-normalisation of State::operator+= (src/core.h:154-162, every kEnforceSO3Freq = 50 absorbs):
// Sophus SO3::normalize() on Rsb / Rbc (unit quaternion; here: matrix -> quaternion -> normalise -> matrix, which
// also re-orthonormalises the stored matrix) and Rsg <- exp(log(Rsg) with its z component zeroed).
__device__ __forceinline__ void rot_to_quat(const M3& R, double q[4]) {   // (w, x, y, z), Shepperd's branch on the largest diagonal term
  const double t = R.m[0][0] + R.m[1][1] + R.m[2][2];
  if (t > 0.0) {
    const double s = sqrt(t + 1.0) * 2.0;
    q[0] = 0.25 * s; q[1] = (R.m[2][1] - R.m[1][2]) / s; q[2] = (R.m[0][2] - R.m[2][0]) / s; q[3] = (R.m[1][0] - R.m[0][1]) / s;
  } else if (R.m[0][0] > R.m[1][1] && R.m[0][0] > R.m[2][2]) {
    const double s = sqrt(1.0 + R.m[0][0] - R.m[1][1] - R.m[2][2]) * 2.0;
    q[0] = (R.m[2][1] - R.m[1][2]) / s; q[1] = 0.25 * s; q[2] = (R.m[0][1] + R.m[1][0]) / s; q[3] = (R.m[0][2] + R.m[2][0]) / s;
  } else if (R.m[1][1] > R.m[2][2]) {
    const double s = sqrt(1.0 + R.m[1][1] - R.m[0][0] - R.m[2][2]) * 2.0;
    q[0] = (R.m[0][2] - R.m[2][0]) / s; q[1] = (R.m[0][1] + R.m[1][0]) / s; q[2] = 0.25 * s; q[3] = (R.m[1][2] + R.m[2][1]) / s;
  } else {
    const double s = sqrt(1.0 + R.m[2][2] - R.m[0][0] - R.m[1][1]) * 2.0;
    q[0] = (R.m[1][0] - R.m[0][1]) / s; q[1] = (R.m[0][2] + R.m[2][0]) / s; q[2] = (R.m[1][2] + R.m[2][1]) / s; q[3] = 0.25 * s;
  }
  const double n = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] *= n; q[1] *= n; q[2] *= n; q[3] *= n;
}
__device__ __forceinline__ void quat_to_colmajor(const double q[4], double* Rcm) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  Rcm[0] = 1.0 - 2.0 * (y * y + z * z); Rcm[3] = 2.0 * (x * y - w * z);       Rcm[6] = 2.0 * (x * z + w * y);
  Rcm[1] = 2.0 * (x * y + w * z);       Rcm[4] = 1.0 - 2.0 * (x * x + z * z); Rcm[7] = 2.0 * (y * z - w * x);
  Rcm[2] = 2.0 * (x * z - w * y);       Rcm[5] = 2.0 * (y * z + w * x);       Rcm[8] = 1.0 - 2.0 * (x * x + y * y);
}
__device__ __forceinline__ void rot_normalize(double* Rcm) {
  double q[4];
  rot_to_quat(m3_from_colmajor(Rcm), q);
  quat_to_colmajor(q, Rcm);
}
__device__ __forceinline__ void rot_zero_log_z(double* Rcm) {   // Sophus SO3::log on the unit quaternion, z <- 0, exp
  double q[4];
  rot_to_quat(m3_from_colmajor(Rcm), q);
  const double n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3], w = q[0];
  double k;
  if (n2 < 1e-20) k = 2.0 / w - 2.0 / 3.0 * n2 / (w * w * w);
  else {
    const double n = sqrt(n2);
    k = fabs(w) < 1e-10 ? (w > 0.0 ? 3.141592653589793 / n : -3.141592653589793 / n) : 2.0 * atan(n / w) / n;
  }
  const M3 R = so3_exp_dev(k * q[1], k * q[2], 0.0);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Rcm[i + 3 * j] = R.m[i][j];
}
// ---------------------------------------------------------------- batched resident edits (xivo_hip_edit_batch)
__device__ __forceinline__ void edit_zero_rc(double* P, int ldp, int Np, int off, int len, int tid) {
  for (int t = tid; t < Np; t += 256)
    for (int r = 0; r < len; ++r) {
      P[(off + r) + (long)t * ldp] = 0.0;
      P[t + (long)(off + r) * ldp] = 0.0;
    }
  __syncthreads();
}
// rows, then columns (which re-read the rows just written): the order of src/estimator.cpp:808-816
__device__ __forceinline__ void edit_copy_rc(double* P, int ldp, int Np, int dst, int src, int len, int tid) {
  for (int t = tid; t < Np; t += 256)
    for (int r = 0; r < len; ++r) P[(dst + r) + (long)t * ldp] = P[(src + r) + (long)t * ldp];
  __syncthreads();
  for (int t = tid; t < Np; t += 256)
    for (int r = 0; r < len; ++r) P[t + (long)(dst + r) * ldp] = P[t + (long)(src + r) * ldp];
  __syncthreads();
}
// xivo_hip_set_pixels: one thread per (filter, list entry)
__global__ void set_pixels_kernel(xivo_feat_in* feats, int Fmax, int F, const double* xp, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const double u = xp[2 * t], v = xp[2 * t + 1];
  if (u != u || v != v) return;   // NaN: not tracked in this frame
  xivo_feat_in& f = feats[(long)(t / F) * Fmax + (t % F)];
  f.xp[0] = u; f.xp[1] = v;
}
// One workgroup per filter that has ops; its ops run in array order.
__global__ __launch_bounds__(256) void edit_batch_kernel(EditArgs a) {
  const int w = blockIdx.x, tid = threadIdx.x;
  const int filt = a.wg_filter[w];
  double* P = a.P + (long)filt * a.strideP;
  xivo_feat_in* feats = a.feats + (long)filt * a.Fmax;
  xivo_group_in* groups = a.groups + (long)filt * a.lay.n_groups;
  const xivo_pose_in& X = a.poses[filt];
  for (int o = a.wg_begin[w]; o < a.wg_begin[w + 1]; ++o) {
    const xivo_edit_op& op = a.ops[o];
    switch (op.kind) {
      case XIVO_EDIT_P_ZERO_RC: edit_zero_rc(P, a.ldp, a.Np, op.i0, op.i1, tid); break;
      case XIVO_EDIT_P_COPY_RC: edit_copy_rc(P, a.ldp, a.Np, op.i0, op.i1, op.i2, tid); break;
      case XIVO_EDIT_P_SET_BLOCK3:
        if (tid < 9) P[(op.i0 + tid % 3) + (long)(op.i0 + tid / 3) * a.ldp] = op.v[tid];
        __syncthreads();
        break;
      case XIVO_EDIT_ADD_GROUP: {
        if (tid < 9) groups[op.i0].Rsb[tid] = X.Rsb[tid];
        else if (tid < 12) groups[op.i0].Tsb[tid - 9] = X.Tsb[tid - 9];
        const int off = a.lay.group_begin + 6 * op.i0;
        edit_copy_rc(P, a.ldp, a.Np, off, 0, 3, tid);       // Index::Wsb
        edit_copy_rc(P, a.ldp, a.Np, off + 3, 3, 3, tid);   // Index::Tsb
        break;
      }
      case XIVO_EDIT_REMOVE_GROUP: edit_zero_rc(P, a.ldp, a.Np, a.lay.group_begin + 6 * op.i0, 6, tid); break;
      case XIVO_EDIT_ADD_FEATURE: {
        if (tid == 0) {
          xivo_feat_in& f = feats[op.i0];
          f.x[0] = op.v[0]; f.x[1] = op.v[1]; f.x[2] = op.v[2];
          f.xp[0] = op.v[3]; f.xp[1] = op.v[4];
          f.sind = op.i1; f.ref_sind = op.i2;
        }
        const int off = a.lay.feature_begin + 3 * op.i1;
        edit_zero_rc(P, a.ldp, a.Np, off, 3, tid);
        if (tid < 9) P[(off + tid % 3) + (long)(off + tid / 3) * a.ldp] = op.v[5 + tid];
        __syncthreads();
        break;
      }
      case XIVO_EDIT_REMOVE_FEATURE: {
        const int sind = feats[op.i0].sind;
        __syncthreads();
        if (sind >= 0) {
          edit_zero_rc(P, a.ldp, a.Np, a.lay.feature_begin + 3 * sind, 3, tid);
          if (tid == 0) feats[op.i0].sind = -1;
          __syncthreads();
        }
        break;
      }
      case XIVO_EDIT_SET_XP:
        if (tid < 2) feats[op.i0].xp[tid] = op.v[tid];
        __syncthreads();
        break;
      default: break;
    }
  }
}

// One workgroup per filter; thread 0 retracts the motion state, threads 1.. the group slots and features.
__global__ __launch_bounds__(256) void absorb_error_kernel(AbsorbArgs a) {
  const int filt = blockIdx.x, tid = threadIdx.x;
  double* err = a.err + (long)filt * a.strideErr;
  if (a.status && a.status[filt]) {                      // S was not positive definite: K / dx of that filter are meaningless
    for (int n = tid; n < a.lay.N; n += 256) err[n] = 0.0;
    return;
  }
  if (tid == 0) {                                        // State::operator+= (core.h:135-165)
    xivo_pose_in& X = a.poses[filt];
    rot_retract(X.Rsb, err[0], err[1], err[2]);
    rot_retract(X.Rbc, err[15], err[16], err[17]);
    rot_retract(X.Rsg, err[21], err[22], 0.0);
    for (int i = 0; i < 3; ++i) {
      X.Tsb[i] += err[3 + i]; X.Vsb[i] += err[6 + i]; X.bg[i] += err[9 + i]; X.ba[i] += err[12 + i]; X.Tbc[i] += err[18 + i];
    }
    if (a.calib) {                                       // online-calibration builds
      xivo_calib_in& cb = a.calib[filt];
      if (a.cl.td >= 0) cb.td += err[a.cl.td];           // core.h:150-152
      if (a.cl.Cg >= 0) {                                // estimator.cpp:879-884 -> IMUState::operator+= (imu.cpp:7-21): Ca's upper
        int k = a.cl.Cg + 9;                             // triangle row by row, then Cg row by row (both stored column-major)
        for (int i = 0; i < 3; ++i)
          for (int j = i; j < 3; ++j) cb.Ca[i + 3 * j] += err[k++];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) cb.Cg[i + 3 * j] += err[a.cl.Cg + 3 * i + j];
      }
      for (int k = 0; k < a.cl.cam_dim && k < 9; ++k) cb.intr[k] += err[a.cl.cam_begin + k];   // estimator.cpp:886-890
    }
    if (a.counter && ++a.counter[filt] % 50 == 0) {      // kEnforceSO3Freq (core.h:111,154-162)
      rot_normalize(X.Rsb);
      rot_normalize(X.Rbc);
      rot_zero_log_z(X.Rsg);
    }
  }
  const unsigned long long gmask = a.group_mask ? a.group_mask[filt] : ~0ull;   // instate_groups_ (estimator.cpp:897)
  for (int g = tid; g < a.lay.n_groups; g += 256) {      // SO3xR3::operator+= (group.h:25-29); empty slots have dx = 0
    if (g < 64 && !((gmask >> g) & 1ull)) continue;
    xivo_group_in& G = a.groups[(long)filt * a.lay.n_groups + g];
    const int off = a.lay.group_begin + 6 * g;
    rot_retract(G.Rsb, err[off], err[off + 1], err[off + 2]);
    for (int i = 0; i < 3; ++i) G.Tsb[i] += err[off + 3 + i];
  }
  for (int f = tid; f < (a.mask ? a.F : 0); f += 256) {  // Feature::UpdateState for in_current_ekf_update_ (estimator.cpp:906-912)
    if (!a.mask[(long)filt * a.Fmax + f]) continue;
    xivo_feat_in& ft = a.feats[(long)filt * a.Fmax + f];
    if (ft.sind < 0) continue;
    const int off = a.lay.feature_begin + 3 * ft.sind;
    for (int i = 0; i < 3; ++i) ft.x[i] += err[off + i];
  }
  __syncthreads();
  for (int n = tid; n < a.lay.N; n += 256) err[n] = 0.0;  // err_.setZero() (estimator.cpp:920)
}

// ---------------------------------------------------------------- OOS / MSCKF rows
// One wave64 per (filter, OOS feature). Lane 0 runs Eigen's FullPivLU on the
// 3 x 2k matrix Hf^T exactly as FullPivLU::computeInPlace / kernel() do
// (thirdparty/eigen/Eigen/src/LU/FullPivLU.h:490-580, 619-699) so that the
// null-space basis A - which is NOT orthonormal - matches SlowGivens
// (helpers.cpp:13-23) and not merely its span; lanes then form A^T Hx, A^T r.
constexpr int OOS_R = 2 * XIVO_OOS_MAX_OBS;  // max rows 2k

__global__ __launch_bounds__(64) void oos_kernel(OosArgs a) {
  const int filt = blockIdx.y, o = blockIdx.x, lane = threadIdx.x;
  const xivo_oos_in& ft = a.feats[(long)filt * a.n_oos + o];
  const xivo_pose_in& pose = a.poses[filt];
  const int k = ft.n_obs, R2 = 2 * k;
  __shared__ double sHf[OOS_R][3];
  __shared__ double sHx[OOS_R][12];   // per row: [Wg(3) Tg(3) Wbc(3) Tbc(3)]
  __shared__ double sInn[OOS_R];
  __shared__ double sA[OOS_R][OOS_R]; // kernel basis, 2k x dimker
  __shared__ int sQ[OOS_R];
  __shared__ int sRank;
  __shared__ int sRow0;

  // row offset of this feature = row0 + sum_{o' < o} (2 k_o' - 3)  (rank 3 assumed for the
  // reservation; rows beyond the actual kernel dimension stay neutral)
  if (lane == 0) {
    int r = a.row0;
    for (int q = 0; q < o; ++q) {
      const int kq = a.feats[(long)filt * a.n_oos + q].n_obs;
      r += kq >= 2 ? (a.whole ? a.whole - 3 : 2 * kq - 3) : 0;
    }
    sRow0 = r;
    if (o == a.n_oos - 1 && a.rows_out) a.rows_out[filt] = r + (k >= 2 ? (a.whole ? a.whole - 3 : 2 * k - 3) : 0) - a.row0;
  }
  // per-observation Jacobians (oos.cpp:39-89), one lane per observation
  if (lane < k) {
    const xivo_group_in& g = a.groups[(long)filt * a.lay.n_groups + ft.group_sind[lane]];
    const M3 Rsb = m3_from_colmajor(g.Rsb), Rbc = m3_from_colmajor(pose.Rbc);
    const M3 Rsb_t = m3_t(Rsb), Rbc_t = m3_t(Rbc);
    V3 d;
#pragma unroll
    for (int i = 0; i < 3; ++i) d.v[i] = ft.Xs[i] - g.Tsb[i];
    const V3 Xb = m3_mulv(Rsb_t, d);
#pragma unroll
    for (int i = 0; i < 3; ++i) d.v[i] = Xb.v[i] - pose.Tbc[i];
    const V3 Xcn = m3_mulv(Rbc_t, d);
    double xp[2], dxp_dXcn[2][3];
    project_pixel(filter_cam(a.cam, a.calib, a.cam_dim, filt), Xcn, xp, dxp_dXcn);
    double t1[2][3], out[2][3];
    m23_mul(dxp_dXcn, Rbc_t, t1);                 // dxp_dXcn * dXcn_dXb
    m23_mul(t1, Rsb_t, out);                      // * dXb_dXs -> Hf        (oos.cpp:74-75)
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) sHf[2 * lane + i][j] = out[i][j];
    m23_mul(t1, hat(Xb), out);                    // * dXb_dWsb -> goff     (oos.cpp:78-79)
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) sHx[2 * lane + i][j] = out[i][j];
    m23_mul(t1, m3_neg(Rsb_t), out);              // * dXb_dTsb -> goff + 3 (oos.cpp:80-81)
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) sHx[2 * lane + i][3 + j] = out[i][j];
    m23_mul(dxp_dXcn, hat(Xcn), out);             // dXcn_dWbc              (oos.cpp:82-83)
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) sHx[2 * lane + i][6 + j] = out[i][j];
    m23_mul(dxp_dXcn, m3_neg(Rbc_t), out);        // dXcn_dTbc              (oos.cpp:84-85)
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) sHx[2 * lane + i][9 + j] = out[i][j];
    sInn[2 * lane] = ft.xp[lane][0] - xp[0];      // oos.cpp:72
    sInn[2 * lane + 1] = ft.xp[lane][1] - xp[1];
  }
  __syncthreads();

  // FullPivLU of Hf^T (3 x 2k) and its kernel (Eigen FullPivLU.h:446-534, 619-699; helpers.cpp:15-16), one lane per
  // column of Hf^T with the column's three entries in registers: the pivot search is a wave arg-max that keeps the
  // FIRST maximum of Eigen's column-major scan (smaller column, then smaller row), row swaps are register selects,
  // column swaps and the broadcasts of the pivot column are lane shuffles. Every arithmetic operation is the one the
  // serial algorithm performs on that element, so the basis equals Eigen's to rounding.
  {
    const int cols = R2;
    const bool incol = lane < cols;
    double c0 = incol ? sHf[lane][0] : 0.0, c1 = incol ? sHf[lane][1] : 0.0, c2 = incol ? sHf[lane][2] : 0.0;
    int q = lane;                     // m_q: position -> original column, built from the column transpositions
    int nonzero = 3;
    double maxpivot = 0.0;
    int colsT0 = 0, colsT1 = 1, colsT2 = 2;
    bool live = true;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      if (!live) continue;
      // biggest |.| of the bottom-right corner: per lane over rows kk..2 (first maximum), then over lanes j >= kk
      double best = -1.0; int bi = kk;
      if (incol && lane >= kk) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (i < kk) continue;
          const double sv = fabs(i == 0 ? c0 : (i == 1 ? c1 : c2));
          if (sv > best) { best = sv; bi = i; }
        }
      }
      int bj = lane;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const double ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o), oj = __shfl_xor(bj, o);
        if (ob > best || (ob == best && oj < bj)) { best = ob; bi = oi; bj = oj; }
      }
      const double big = best; const int br = bi, bc = bj;     // wave-uniform
      if (big == 0.0) { nonzero = kk; live = false; continue; }   // the rest of the corner is exactly zero (FullPivLU.h:486-494)
      if (big > maxpivot) maxpivot = big;
      if (kk == 0) colsT0 = bc; else if (kk == 1) colsT1 = bc; else colsT2 = bc;
      // rows kk <-> br in every column
      if (br != kk) {
        double& a_ = kk == 0 ? c0 : (kk == 1 ? c1 : c2);
        if (br == 1) { const double t = a_; a_ = c1; c1 = t; }
        else if (br == 2) { const double t = a_; a_ = c2; c2 = t; }
      }
      // columns kk <-> bc
      if (bc != kk) {
        const int src = lane == kk ? bc : (lane == bc ? kk : lane);
        c0 = __shfl(c0, src); c1 = __shfl(c1, src); c2 = __shfl(c2, src);
        q = __shfl(q, src);
      }
      // multipliers in column kk, then the rank-1 update of the corner
      const double pk0 = __shfl(c0, kk), pk1 = __shfl(c1, kk), pk2 = __shfl(c2, kk);    // column kk as it is now
      const double piv = kk == 0 ? pk0 : (kk == 1 ? pk1 : pk2);
      double l1 = 0.0, l2 = 0.0;        // multipliers of rows 1, 2 (those below kk)
      if (kk == 0) { l1 = pk1 / piv; l2 = pk2 / piv; if (lane == 0) { c1 = l1; c2 = l2; } }
      else if (kk == 1) { l2 = pk2 / piv; if (lane == 1) c2 = l2; }
      if (lane > kk) {
        const double top = kk == 0 ? c0 : (kk == 1 ? c1 : c2);     // sLU[kk][j]
        if (kk == 0) { c1 -= l1 * top; c2 -= l2 * top; }
        else if (kk == 1) { c2 -= l2 * top; }
      }
    }
    (void)colsT0; (void)colsT1; (void)colsT2;
    if (incol) sQ[lane] = q;
    // rank with Eigen's default threshold eps * diagonalSize (FullPivLU.h threshold())
    const double thr = maxpivot * (2.220446049250313e-16 * 3);
    const double d0 = __shfl(c0, 0), d1 = __shfl(c1, 1), d2 = __shfl(c2, 2);
    int piv[3]; int np = 0;
    if (0 < nonzero && fabs(d0) > thr) piv[np++] = 0;
    if (1 < nonzero && fabs(d1) > thr) piv[np++] = 1;
    if (2 < nonzero && fabs(d2) > thr) piv[np++] = 2;
    const int rank = np, dimker = cols - rank;
    const int p0 = rank > 0 ? piv[0] : 0, p1 = rank > 1 ? piv[1] : 0, p2 = rank > 2 ? piv[2] : 0;
    if (lane == 0) sRank = rank;
    // trapezoid m (rank x cols): row i = row piv[i] of the LU with its strictly lower part zeroed (FullPivLU.h:660-672)
    auto rowsel = [&](int r) -> double { return r == 0 ? c0 : (r == 1 ? c1 : c2); };
    double m0 = rank > 0 ? rowsel(p0) : 0.0;
    double m1 = rank > 1 ? (lane < 1 ? 0.0 : rowsel(p1)) : 0.0;
    double m2 = rank > 2 ? (lane < 2 ? 0.0 : rowsel(p2)) : 0.0;
    auto swap_cols = [&](int ca, int cb) {
      if (ca == cb) return;
      const int src = lane == ca ? cb : (lane == cb ? ca : lane);
      m0 = __shfl(m0, src); m1 = __shfl(m1, src); m2 = __shfl(m2, src);
    };
    if (rank > 0) swap_cols(0, p0);
    if (rank > 1) swap_cols(1, p1);
    if (rank > 2) swap_cols(2, p2);
    // upper-triangular solve m[:, :rank] X = m[:, rank:], one lane per right-hand side
    const double u00 = __shfl(m0, 0), u01 = __shfl(m0, 1), u02 = __shfl(m0, 2), u11 = __shfl(m1, 1), u12 = __shfl(m1, 2), u22 = __shfl(m2, 2);
    if (lane >= rank && incol) {
      if (rank == 3) { m2 = m2 / u22; m1 = (m1 - u12 * m2) / u11; m0 = ((m0 - u01 * m1) - u02 * m2) / u00; }
      else if (rank == 2) { m1 = m1 / u11; m0 = (m0 - u01 * m1) / u00; }
      else if (rank == 1) { m0 = m0 / u00; }
    }
    if (rank > 2) swap_cols(2, p2);
    if (rank > 1) swap_cols(1, p1);
    if (rank > 0) swap_cols(0, p0);
    __syncthreads();                              // sQ complete
    // dst.row(q[i]) = -m.row(i).tail(dimker), rows q[rank..] zero, then the identity block (FullPivLU.h:690-697)
    if (incol && lane >= rank) {
      const int c = lane - rank;
      for (int j = 0; j < cols; ++j) sA[j][c] = 0.0;
      if (rank > 0) sA[sQ[0]][c] = -m0;
      if (rank > 1) sA[sQ[1]][c] = -m1;
      if (rank > 2) sA[sQ[2]][c] = -m2;
      sA[sQ[lane]][c] = 1.0;
    }
    (void)dimker;
  }
  __syncthreads();

  // Hx <- A^T Hx, inn <- A^T inn (helpers.cpp:20, oos.cpp:29); lane r owns output row r
  const int dimker = R2 - sRank;
  const int nrows_res = k >= 2 ? 2 * k - 3 : 0;  // rows reserved for this feature
  if (lane < nrows_res) {
    const int row = sRow0 + lane;
    double* H = a.mb.H + (long)filt * a.mb.strideH;
    double* HT = a.mb.HT + (long)filt * a.mb.strideHT;
    double* inn = a.mb.inn + (long)filt * a.mb.strideInn;
    double* dR = a.mb.diagR + (long)filt * a.mb.strideR;
    if (lane < dimker && row < a.Mp) {
      double rr = 0.0;
      for (int j = 0; j < R2; ++j) rr += sA[j][lane] * sInn[j];
      // the rows arrive zero-filled (stack). The camera-extrinsics columns collect a term from every observation: summed
      // in registers, stored once. A group block belongs to one observation - a plain store - unless the feature was
      // seen twice from the same group; only then the read-modify-write that a general accumulation needs.
      bool dup = false;
      for (int o1 = 0; o1 < k; ++o1)
        for (int o2 = o1 + 1; o2 < k; ++o2) dup = dup || ft.group_sind[o1] == ft.group_sind[o2];
      double ex[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      for (int ob = 0; ob < k; ++ob) {
        const int goff = a.lay.group_begin + 6 * ft.group_sind[ob];
        const double a0 = sA[2 * ob][lane], a1 = sA[2 * ob + 1][lane];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double v = a0 * sHx[2 * ob][c] + a1 * sHx[2 * ob + 1][c];
          const int col = goff + c;
          if (dup) v += H[row + (long)col * a.mb.ldh];
          H[row + (long)col * a.mb.ldh] = v;
          if (a.mb.HT) HT[col + (long)row * a.mb.ldht] = v;
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) ex[c] += a0 * sHx[2 * ob][6 + c] + a1 * sHx[2 * ob + 1][6 + c];
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int col = 15 + c;                       // Index::Wbc (15..17), Index::Tbc (18..20)
        H[row + (long)col * a.mb.ldh] = ex[c];
        if (a.mb.HT) HT[col + (long)row * a.mb.ldht] = ex[c];
      }
      inn[row] = rr;
      dR[row] = a.Roos;
    }
  }
  // src/oos.cpp:28 as coded hands SlowGivens the whole 2 kMaxGroup-row buffers: the rows behind the 2 k filled ones are zero
  // columns of Hf^T - never a pivot, never moved by a column transposition of the three pivot steps - so FullPivLU::kernel
  // appends one unit vector per such row behind the 2 k - 3 basis vectors above (checked against the oracle's
  // restatement on the padded buffers, tests/test_oos_gpu.py): zero rows of H, inn = 0, diagR = Roos
  if (a.whole && k >= 2) {
    double* inn = a.mb.inn + (long)filt * a.mb.strideInn;
    double* dR = a.mb.diagR + (long)filt * a.mb.strideR;
    for (int r = nrows_res + lane; r < a.whole - 3; r += 64) {
      const int row = sRow0 + r;
      if (row < a.Mp) { inn[row] = 0.0; dR[row] = a.Roos; }
    }
  }
}

// ---------------------------------------------------------------- loop-closure rows
// Feature::ComputeLCJacobian (oos.cpp:92-145) as Estimator::CloseLoopInternal drives it (update.cpp:183-196): one thread per
// (filter, match). The OLD in-state feature's world position Xs = Feature::Xs(gbc) (feature.cpp:107-118: its own state and
// anchor group) is re-observed as pixel xp by the group in slot group_sind: row pair 2m, 2m+1 of a zeroed H gets
// d xp / d (Wsb_g, Tsb_g, Wbc, Tbc) [+ the intrinsics block under USE_ONLINE_CAMERA_CALIB, :125-142], inn = obs.xp - xp,
// diagR = Rlc. No block for the old feature's own state or its anchor group: as coded. feat < 0: an absent match (ragged
// batches) - a neutral row pair (H = 0, inn = 0, R = 1).
__global__ void lc_rows_kernel(LcArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.batch * a.n) return;
  const int filt = t / a.n, m = t % a.n;
  const xivo_lc_match& mt = a.matches[t];
  double* H = a.H + (long)filt * a.strideH;          // 2n x N column-major, ld = ldh, zero-filled by the caller
  double* inn = a.inn + (long)filt * a.strideV;
  double* dR = a.diagR + (long)filt * a.strideV;
  const int r0 = 2 * m;
  if (mt.feat < 0) { inn[r0] = 0.0; inn[r0 + 1] = 0.0; dR[r0] = 1.0; dR[r0 + 1] = 1.0; return; }
  const xivo_pose_in& pose = a.poses[filt];
  const xivo_feat_in& ft = a.feats[(long)filt * a.Fmax + mt.feat];
  // a matched feature that is not in the state (a free slot of a ragged batch: sind < 0) or whose anchor slot is out of range is
  // the caller's error, not an address: the pair stays neutral (the host checked feat / group_sind against the layout)
  if (ft.sind < 0 || ft.ref_sind < 0 || ft.ref_sind >= a.lay.n_groups) { inn[r0] = 0.0; inn[r0 + 1] = 0.0; dR[r0] = 1.0; dR[r0 + 1] = 1.0; return; }
  const xivo_group_in& gref = a.groups[(long)filt * a.lay.n_groups + ft.ref_sind];
  const xivo_group_in& g = a.groups[(long)filt * a.lay.n_groups + mt.group_sind];
  const xivo_cam cam = filter_cam(a.cam, a.calib, a.cl.cam_dim, filt);
  const M3 Rbc = m3_from_colmajor(pose.Rbc), Rsbr = m3_from_colmajor(gref.Rsb), Rsb = m3_from_colmajor(g.Rsb);
  const M3 Rsb_t = m3_t(Rsb), Rbc_t = m3_t(Rbc);
  // Xs(gbc) = ref_->gsb() * gbc * Xc (feature.cpp:112-113)
  M3 dXc_dx;
  const V3 Xc = feature_unproject(ft.x, a.invdepth, dXc_dx);
  V3 Xbr = m3_mulv(Rbc, Xc);
#pragma unroll
  for (int i = 0; i < 3; ++i) Xbr.v[i] += pose.Tbc[i];
  V3 Xs = m3_mulv(Rsbr, Xbr);
  V3 d;
#pragma unroll
  for (int i = 0; i < 3; ++i) { Xs.v[i] += gref.Tsb[i]; d.v[i] = Xs.v[i] - g.Tsb[i]; }
  const V3 Xb = m3_mulv(Rsb_t, d);                              // :107
#pragma unroll
  for (int i = 0; i < 3; ++i) d.v[i] = Xb.v[i] - pose.Tbc[i];
  const V3 Xcn = m3_mulv(Rbc_t, d);                             // :113
  double xp[2], dxp_dXcn[2][3];
  project_pixel(cam, Xcn, xp, dxp_dXcn);                        // :123-133
  const M3 dXcn_dTsb = m3_mul(Rbc_t, m3_neg(Rsb_t));            // :120  dXcn_dXb * dXb_dTsb
  const M3 dXcn_dWsb = m3_mul(Rbc_t, hat(Xb));                  // :121  dXcn_dXb * dXb_dWsb
  double blk[2][3];
  const int goff = a.lay.group_begin + 6 * mt.group_sind;
  auto put = [&](int col) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) H[r0 + i + (long)(col + j) * a.ldh] = blk[i][j];
  };
  m23_mul(dxp_dXcn, dXcn_dWsb, blk); put(goff);                 // :136
  m23_mul(dxp_dXcn, dXcn_dTsb, blk); put(goff + 3);             // :137
  m23_mul(dxp_dXcn, hat(Xcn), blk); put(15);                    // :138  Index::Wbc
  m23_mul(dxp_dXcn, m3_neg(Rbc_t), blk); put(18);               // :139  Index::Tbc
  if (a.cl.cam_dim > 0) {                                       // :141-144
    double xq[2], Jq[2][2], jacc[2][9];
    camera_project_jacc(cam, Xcn.v[0] / Xcn.v[2], Xcn.v[1] / Xcn.v[2], xq, Jq, jacc);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < a.cl.cam_dim && j < 9; ++j) H[r0 + i + (long)(a.cl.cam_begin + j) * a.ldh] = jacc[i][j];
  }
  inn[r0] = mt.xp[0] - xp[0]; inn[r0 + 1] = mt.xp[1] - xp[1];   // :146
  dR[r0] = a.Rlc; dR[r0 + 1] = a.Rlc;                           // update.cpp:193
}

// ---------------------------------------------------------------- propagation tail
// P_mm <- Pmm_new ; P_ms <- Phi P_ms ; P_sm <- P_sm Phi^T (rk4.cpp:92-102). One
// workgroup per filter, Phi in LDS; thread j owns structure column / row j.
__global__ __launch_bounds__(256) void propagate_cov_kernel(double* Pall, long strideP, int ldp, int N, int nm,
                                                            const double* Phi_all, const double* Pmm_all,
                                                            int b0) {
  const int filt = b0 + blockIdx.x, tid = threadIdx.x;
  double* P = Pall + (long)filt * strideP;
  const double* Phi = Phi_all + (long)blockIdx.x * nm * nm;
  const double* Pmm = Pmm_all + (long)blockIdx.x * nm * nm;
  extern __shared__ double sPhi[];  // nm*nm, column-major
  for (int e = tid; e < nm * nm; e += 256) sPhi[e] = Phi[e];
  __syncthreads();
  constexpr int MAXM = 40;
  for (int j = nm + tid; j < N; j += 256) {
    double col[MAXM], row[MAXM];
    for (int k = 0; k < nm; ++k) { col[k] = P[k + (long)j * ldp]; row[k] = P[j + (long)k * ldp]; }
    for (int i = 0; i < nm; ++i) {
      double s = 0.0, t = 0.0;
      for (int k = 0; k < nm; ++k) {
        s = fma(sPhi[i + k * nm], col[k], s);   // (Phi P_ms)(i, j)
        t = fma(row[k], sPhi[i + k * nm], t);   // (P_sm Phi^T)(j, i)
      }
      P[i + (long)j * ldp] = s;
      P[j + (long)i * ldp] = t;
    }
  }
  for (int e = tid; e < nm * nm; e += 256) P[(e % nm) + (long)(e / nm) * ldp] = Pmm[e];
}

// Compile-time motion size (the default build's 23): col / row stay in registers (with a run-time nm the two
// arrays are indexed dynamically and live in scratch). One workgroup per filter, one thread per state column j >= NM.
// The row block P_ms (NM x (N - NM): NM contiguous doubles per column, columns ldp apart) goes through LDS so that
// HBM sees each 8 NM-byte run once, in lane order, on the way in and on the way out; the column block P_sm is
// coalesced as it lies. Phi sits in LDS with an even leading dimension so that one 16-byte broadcast read feeds two
// output rows (4 FMAs). Bound: HBM - 4 x 8 NM (N - NM) bytes per filter.
template <int NM>
__global__ __launch_bounds__(256) void propagate_cov_fixed_kernel(double* Pall, long strideP, int ldp, int N,
                                                                  const double* Phi_all, const double* Pmm_all, int b0) {
  constexpr int LP = NM + 1;        // even: rows (i, i + 1), i even, of one Phi column are 16-byte aligned
  static_assert(LP % 2 == 0, "NM must be odd");
  const int filt = b0 + blockIdx.x, tid = threadIdx.x;
  double* P = Pall + (long)filt * strideP;
  const double* Phi = Phi_all + (long)blockIdx.x * NM * NM;
  const double* Pmm = Pmm_all + (long)blockIdx.x * NM * NM;
  __shared__ __attribute__((aligned(16))) double sPhi[LP * NM];   // column-major, row NM = 0
  __shared__ double sBlk[NM * 256];                                // [k + NM * (j - j0)]
  for (int e = tid; e < NM * NM; e += 256) sPhi[(e % NM) + LP * (e / NM)] = Phi[e];
  if (tid < NM) sPhi[NM + LP * tid] = 0.0;
  for (int j0 = NM; j0 < N; j0 += 256) {
    const int nc = N - j0 < 256 ? N - j0 : 256;
    __syncthreads();                                // sPhi ready / previous chunk written back
    {
      int k = tid % NM, jj = tid / NM;              // element e = tid + 256 m  <->  (k, jj)
      for (int e = tid; e < NM * nc; e += 256) {
        sBlk[e] = P[k + (long)(j0 + jj) * ldp];
        k += 256 % NM; jj += 256 / NM;
        if (k >= NM) { k -= NM; ++jj; }
      }
    }
    __syncthreads();
    if (tid < nc) {
      const int j = j0 + tid;
      double col[NM], row[NM];
#pragma unroll
      for (int k = 0; k < NM; ++k) { col[k] = sBlk[k + NM * tid]; row[k] = P[j + (long)k * ldp]; }
#pragma unroll
      for (int i = 0; i < NM; i += 2) {
        double s0 = 0.0, s1 = 0.0, t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int k = 0; k < NM; ++k) {
          const double2 ph = *reinterpret_cast<const double2*>(&sPhi[i + LP * k]);
          s0 = fma(ph.x, col[k], s0);   // (Phi P_ms)(i, j)
          t0 = fma(row[k], ph.x, t0);   // (P_sm Phi^T)(j, i)
          s1 = fma(ph.y, col[k], s1);
          t1 = fma(row[k], ph.y, t1);
        }
        sBlk[i + NM * tid] = s0;
        P[j + (long)i * ldp] = t0;
        if (i + 1 < NM) {
          sBlk[i + 1 + NM * tid] = s1;
          P[j + (long)(i + 1) * ldp] = t1;
        }
      }
    }
    __syncthreads();
    {
      int k = tid % NM, jj = tid / NM;
      for (int e = tid; e < NM * nc; e += 256) {
        P[k + (long)(j0 + jj) * ldp] = sBlk[e];
        k += 256 % NM; jj += 256 / NM;
        if (k >= NM) { k -= NM; ++jj; }
      }
    }
  }
  for (int e = tid; e < NM * NM; e += 256) P[(e % NM) + (long)(e / NM) * ldp] = Pmm[e];
}

// ---------------------------------------------------------------- propagation: state + covariance stages
// Runge-Kutta tableaus as the reference codes them: RK4Step (rk4.cpp:35-103; the 4th stage re-uses the half-step
// IMU sample, :77) and PrinceDormandStep (princedormand.cpp:85-221, weights :195-200).
struct RkTableau { int ns; double a[7][6]; double c_step[7]; double c_imu[7]; double b[7]; };
__constant__ RkTableau kTableau[2] = {
    {4,
     {{0}, {0.5}, {0.0, 0.5}, {0.0, 0.0, 1.0}},
     {0.0, 0.5, 0.5, 1.0},
     {0.0, 0.5, 0.5, 0.5},
     {1 / 6.0, 2 / 6.0, 2 / 6.0, 1 / 6.0}},
    {7,
     {{0},
      {2 / 9.0},
      {1 / 12.0, 3 / 12.0},
      {55 / 324.0, -75 / 324.0, 200 / 324.0},
      {83 / 330.0, -195 / 330.0, 305 / 330.0, 27 / 330.0},
      {-19 / 28.0, 63 / 28.0, 4 / 28.0, -108 / 28.0, 88 / 28.0},
      {38 / 400.0, 0.0, 240 / 400.0, -243 / 400.0, 330 / 400.0, 35 / 400.0}},
     {0.0, 2 / 9.0, 3 / 9.0, 5 / 9.0, 6 / 9.0, 1.0, 1.0},
     {0.0, 2 / 9.0, 3 / 9.0, 5 / 9.0, 6 / 9.0, 1.0, 1.0},
     {0.0862, 0.0, 0.6660, -0.7857, 0.9570, 0.0965, -0.0200}}};

// exp(hat(w)) for the per-stage rotation increments (|w| = |gyro| * step, a few mrad): sin(t)/t and (1 - cos t)/t^2 as
// even Taylor series in t^2 - for |w| <= 0.25 the truncation is < 1e-20, below the rounding of the sin / cos route -
// which removes sqrt, sin, cos and two divisions (and their registers) from the chain every stage waits for. A larger
// increment (a single 0.1 s step of a fast spin) is halved until it is small and the result squared back:
// exp(w) = exp(w / 2^n)^(2^n), each squaring costing one rounding of a rotation matrix.
__device__ __forceinline__ M3 so3_exp_small(double wx, double wy, double wz) {
  double t2 = wx * wx + wy * wy + wz * wz;
  int halvings = 0;                         // scaling and squaring for the (unusual) large increment
  while (t2 > 0.0625 && halvings < 64) { wx *= 0.5; wy *= 0.5; wz *= 0.5; t2 *= 0.25; ++halvings; }
  const double a = fma(t2, fma(t2, fma(t2, fma(t2, fma(t2, fma(t2, 1.0 / 6227020800.0, -1.0 / 39916800.0), 1.0 / 362880.0),
                                                 -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
  const double b = fma(t2, fma(t2, fma(t2, fma(t2, fma(t2, fma(t2, 1.0 / 87178291200.0, -1.0 / 479001600.0), 1.0 / 3628800.0),
                                                 -1.0 / 40320.0), 1.0 / 720.0), -1.0 / 24.0), 0.5);
  const V3 w{{wx, wy, wz}};
  const M3 W = hat(w), W2 = m3_mul(W, W);
  M3 R;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R.m[i][j] = (i == j ? 1.0 : 0.0) + a * W.m[i][j] + b * W2.m[i][j];
  for (; halvings > 0; --halvings) R = m3_mul(R, R);
  return R;
}

struct MotionRegs { M3 Rsb; V3 Tsb, Vsb, bg, ba; };   // Rsg is a constant of Propagate: its product Rsg g is passed separately

// ComposeMotion, estimator.cpp:598-613 (default build: Cg = Ca = I)
__device__ __forceinline__ void compose_motion_dev(MotionRegs& X, const V3& V, const V3& gyro, const V3& accel, double dt,
                                                   const V3& Rg) {
  V3 gc, ac;
#pragma unroll
  for (int i = 0; i < 3; ++i) { gc.v[i] = gyro.v[i] - X.bg.v[i]; ac.v[i] = accel.v[i] - X.ba.v[i]; }
  const V3 Ra = m3_mulv(X.Rsb, ac);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    X.Tsb.v[i] += V.v[i] * dt;                                   // :608
    X.Vsb.v[i] += (Ra.v[i] + Rg.v[i]) * dt;                      // :609
  }
  X.Rsb = m3_mul(X.Rsb, so3_exp_small(gc.v[0] * dt, gc.v[1] * dt, gc.v[2] * dt));   // :610
}

// Structure the propagation kernels exploit (the online-calibration kernel below: one workgroup of 256 threads per filter; the default
// build: one wave per filter, further down - same arithmetic per element):
//  * The nominal state of stage st - ComposeMotion of the sub-step's start state with the interpolated IMU sample
//    (rk4.cpp:49-88) - feeds the covariance stages only through Rsb(st), the bias-corrected gyro / accel and the stage
//    velocity K_st, and none of these depends on another stage (only Tsb does, through the a_ij-weighted velocities,
//    and Tsb enters no Jacobian). So the serial chain of ns ComposeMotion + ComputeMotionJacobianAt evaluations
//    (estimator.cpp:598-704) collapses to a pre-pass in which wave w evaluates stages w, w + 4 and publishes the
//    Jacobian blocks (36 numbers) and K_st in LDS, followed by one ComposeMotion for the sub-step itself.
//  * F = dX'/dX (23 x 23) has non-zero rows only for Wsb, Tsb, Vsb and at most 8 non-zeros in a row - dW/dW,
//    dW/dbg = -I, dT/dV = I, dV/dW, dV/dba = -Rsb, dV/dWsg - and G (23 x 12) is four 3 x 3 blocks (-I, -Rsb, I, I).
//    Neither is materialised: F M, M F^T and G Q G^T are formed from register copies of the 36 numbers with the
//    structural zeros skipped (exact: the skipped terms are 0 * x, the surviving ones are summed in the same
//    ascending-k order as the dense product).
//  * NS (stages) is a template parameter: the tableau-weighted sums are unrolled, all LDS loads of a sum are in
//    flight together, and coefficients of stages not yet computed are the tableau's zeros times finite stale values.
// The 23 x 23 matrices live in LDS (column-major, ld 23); FK keeps its 9 non-zero rows only ([i + 9 j]).

// ---------------------------------------------------------------- propagation, online-calibration builds
// The reference's USE_ONLINE_TEMPORAL_CALIB / USE_ONLINE_IMU_CALIB builds (src/core.h:49-75) carry td, Cg (9) and Ca (6) in the
// motion block: kMotionSize = 24 / 38 / 39, ComposeMotion uses imu_.Cg() / imu_.Ca() (estimator.cpp:603-604) and
// ComputeMotionJacobianAt adds dWsb/dCg (:626-631, :674-679) and dVsb/dCa (:633-636, :680-684). The rows of F that are not
// identically zero are still the nine of Wsb / Tsb / Vsb, so the products keep the shape of the default-build kernel above -
// F P0 is 9 x nm, P0 F^T is nm x 9, FK_q has nine rows, G Q G^T the same 12 x 12 support - but the nine rows are held DENSE
// (nm columns each, structural zeros multiplied through: 0 * x adds +0.0 in the same ascending-k sums) and nm is a run-time
// value. One workgroup of 256 threads per filter, everything in LDS (153 KB for nm = 39 with the seven Dormand-Prince
// stages: one workgroup per CU). Not a tuned kernel: these builds are off the metric path.
__device__ __forceinline__ void compose_motion_calib_dev(MotionRegs& X, const V3& V, const V3& gyro, const V3& accel, double dt,
                                                         const V3& Rg, const M3& Cg, const M3& Ca) {
  const V3 cg = m3_mulv(Cg, gyro), ca = m3_mulv(Ca, accel);
  V3 gc, ac;
#pragma unroll
  for (int i = 0; i < 3; ++i) { gc.v[i] = cg.v[i] - X.bg.v[i]; ac.v[i] = ca.v[i] - X.ba.v[i]; }   // :603-604
  const V3 Ra = m3_mulv(X.Rsb, ac);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    X.Tsb.v[i] += V.v[i] * dt;                                   // :608
    X.Vsb.v[i] += (Ra.v[i] + Rg.v[i]) * dt;                      // :609
  }
  X.Rsb = m3_mul(X.Rsb, so3_exp_small(gc.v[0] * dt, gc.v[1] * dt, gc.v[2] * dt));   // :610
}

template <int NS>
__global__ __launch_bounds__(256) void propagate_state_calib_kernel(PropStateArgs a) {
  constexpr int NT = 256, FR = 9, JS = 60;   // JS: doubles per stage of published Jacobian blocks
  const int nm = a.nm, NN = nm * nm, NF = FR * nm, iCg = a.iCg, iCa = iCg >= 0 ? iCg + 9 : -1;
  extern __shared__ double sm[];
  const int lane = threadIdx.x, filt = blockIdx.x;
  const RkTableau& tab = kTableau[NS == 4 ? 0 : 1];
  double* Pmm = sm;              // P_mm at the start of the sub-step
  double* P0 = Pmm + NN;
  double* PKs = P0 + NN;         // [NS][nm x nm]
  double* PhiA = PKs + NS * NN;  // rows < 9 of the accumulated transition ([i + 9 j]; the other rows stay identity rows)
  double* PhiB = PhiA + NF;
  double* S1 = PhiB + NF;        // [9 x nm] sum a_q FK_q, later rows < 9 of I + FK h
  double* F9 = S1 + NF;          // [9 x nm] the non-zero rows of F of the current stage, dense
  double* FPs = F9 + NF;         // [9 x nm]  F P0
  double* PFs = FPs + NF;        // [nm x 9]  P0 F^T ([i + nm j])
  double* FKs = PFs + NF;        // [NS][9 x nm]
  double* GQG = FKs + NS * NF;   // [12 x 12] support of G Q G^T: rows / cols (Wsb, Vsb, bg, ba)
  double* Q = GQG + 144;
  double* GQc = Q + 144;         // [12 x 12] the non-zero rows of G Q
  double* nom = GQc + 144;       // Rsb[9] row-major, Tsb, Vsb, bg, ba, Rsg g (3 each: 9..23), gyro, accel, slope_gyro, slope_accel
                                 // (24..35), Cg[9], Ca[9] row-major (36..53)
  double* sKs = nom + 64;        // [NS][3] stage velocities
  double* Jms = sKs + 24;        // [NS][JS]: dW/dW, dV/dW, -Rsb, dV/dWsg (3 x 3 row-major each), raw gyro (3), dV/dCa (3 x 6)

  const double* Pg = a.P + (long)filt * a.strideP;
  for (int e = lane; e < NN; e += NT) Pmm[e] = Pg[(e % nm) + (long)(e / nm) * a.ldp];
  for (int e = lane; e < NF; e += NT) PhiA[e] = (e % FR) == (e / FR) ? 1.0 : 0.0;
  for (int e = lane; e < NS * NF; e += NT) FKs[e] = 0.0;        // finite values under the tableau's zero coefficients
  for (int e = lane; e < NS * NN; e += NT) PKs[e] = 0.0;
  double* Phi = PhiA;
  double* PhiN = PhiB;
  for (int e = lane; e < 144; e += NT) {
    const double q = a.Qimu[e];
    Q[e] = q;
    const int r = e % 12;     // rows of G Q that do not depend on the state: Wsb = -Q[0:3,:], bg = Q[6:9,:], ba = Q[9:12,:]
    if (r < 3) GQc[e] = -q;
    else if (r >= 6) GQc[e] = q;
  }
  xivo_pose_in& pose = a.poses[filt];
  const V3 gv{{a.g[0], a.g[1], a.g[2]}};
  if (lane == 0) {
    const V3 Rg0 = m3_mulv(m3_from_colmajor(pose.Rsg), gv);
    const xivo_calib_in& cb = a.calib[filt];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        nom[3 * i + j] = pose.Rsb[i + 3 * j];
        nom[36 + 3 * i + j] = iCg >= 0 ? cb.Cg[i + 3 * j] : (i == j ? 1.0 : 0.0);
        nom[45 + 3 * i + j] = iCg >= 0 ? cb.Ca[i + 3 * j] : (i == j ? 1.0 : 0.0);
      }
      nom[9 + i] = pose.Tsb[i]; nom[12 + i] = pose.Vsb[i]; nom[15 + i] = pose.bg[i]; nom[18 + i] = pose.ba[i];
      nom[21 + i] = Rg0.v[i];
    }
  }
  auto load_nominal = [&](MotionRegs& X, V3& Rg, M3& Cg, M3& Ca) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) { X.Rsb.m[i][j] = nom[3 * i + j]; Cg.m[i][j] = nom[36 + 3 * i + j]; Ca.m[i][j] = nom[45 + 3 * i + j]; }
      X.Tsb.v[i] = nom[9 + i]; X.Vsb.v[i] = nom[12 + i]; X.bg.v[i] = nom[15 + i]; X.ba.v[i] = nom[18 + i];
      Rg.v[i] = nom[21 + i];
    }
  };
  __syncthreads();

  const xivo_imu_in* imu_f = a.imu + (long)filt * a.n_imu;
  // step-size-controlled Dormand-Prince (princedormand.cpp:26-60, as in propagate_state_wave_kernel): every thread carries the step
  const bool ctl = NS == 7 && a.pd_h != nullptr;
  double hs = ctl ? a.pd_h[filt] : 0.0;
  for (int smp = 0; smp < a.n_imu; ++smp) {
    if (lane < 3) {
      nom[24 + lane] = imu_f[smp].gyro[lane]; nom[27 + lane] = imu_f[smp].accel[lane];
      nom[30 + lane] = imu_f[smp].slope_gyro[lane]; nom[33 + lane] = imu_f[smp].slope_accel[lane];
    }
    const double dt = imu_f[smp].dt;
    if (ctl) {
      if (hs < 1e-6) hs = a.stepsize;        // :30-32
      hs = fmin(hs, dt);                     // :34
    }
    __syncthreads();
    double total = 0.0;
    while (total < dt || (!ctl && a.stepsize < 0)) {     // rk4.cpp:13-32, princedormand.cpp:62-81
      double h = a.stepsize;
      if (ctl) h = hs;
      else if (a.stepsize < 0) h = dt;
      else if (total + h > dt) h = dt - total;
      else if (total + h + 0.5 * h > dt) h = 0.5 * h;

      // -- nominal pre-pass: thread st evaluates stage st (ComposeMotion + ComputeMotionJacobianAt, estimator.cpp:598-704)
      if (lane < NS) {
        const int st = lane;
        MotionRegs X0; V3 Rg; M3 Cg, Ca;
        load_nominal(X0, Rg, Cg, Ca);
        const double ti = tab.c_imu[st] * h;
        V3 gi, ai;
#pragma unroll
        for (int i = 0; i < 3; ++i) { gi.v[i] = nom[24 + i] + nom[30 + i] * ti; ai.v[i] = nom[27 + i] + nom[33 + i] * ti; }
        if (st > 0) {
          const V3 V0{{0, 0, 0}};   // the a_ij-weighted velocities only move Tsb, which no Jacobian reads
          compose_motion_calib_dev(X0, V0, gi, ai, tab.c_step[st] * h, Rg, Cg, Ca);
        }
        const V3 cg = m3_mulv(Cg, gi), ca = m3_mulv(Ca, ai);
        V3 gc, ac;
#pragma unroll
        for (int i = 0; i < 3; ++i) { gc.v[i] = cg.v[i] - X0.bg.v[i]; ac.v[i] = ca.v[i] - X0.ba.v[i]; }
        const M3 w_dW_dW = m3_neg(hat(gc));
        const M3 w_dV_dW = m3_neg(m3_mul(X0.Rsb, hat(ac)));
        const M3 w_dV_dWsg = m3_neg(m3_mul(X0.Rsb, hat(gv)));
        const M3 w_nR = m3_neg(X0.Rsb);
        double* Jm = Jms + st * JS;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          sKs[3 * st + i] = X0.Vsb.v[i];
          Jm[36 + i] = gi.v[i];                                  // :626-631: the RAW gyro sample fills dWsb/dCg
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            Jm[3 * i + j] = w_dW_dW.m[i][j]; Jm[9 + 3 * i + j] = w_dV_dW.m[i][j];
            Jm[18 + 3 * i + j] = w_nR.m[i][j]; Jm[27 + 3 * i + j] = w_dV_dWsg.m[i][j];
          }
        }
        // :633-636 dV_dCa = dAB_dA<3,3>(accel) dAB_dB<3,3>(Rsb) dA_dAu<3>() with the index conventions of common/rodrigues.h
        // (:143-165 row index p N + n against :208-227 row index p N + n of a COLUMN-major vec): what survives is
        // dV_dCa(n, u(m, n)) = (Rsb^T accel)(m) for m <= n, u = the upper-triangle counter of dA_dAu (row by row)
        V3 w;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          double v = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) v = fma(ai.v[k], X0.Rsb.m[k][m], v);
          w.v[m] = v;
        }
#pragma unroll
        for (int e = 0; e < 18; ++e) Jm[39 + e] = 0.0;
        {
          int u = 0;
#pragma unroll
          for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int n = m; n < 3; ++n) { Jm[39 + 6 * n + u] = w.v[m]; ++u; }
        }
      }
      __syncthreads();
      // the sub-step of the nominal state itself
      if (lane == 0) {
        MotionRegs X; V3 Rg; M3 Cg, Ca;
        load_nominal(X, Rg, Cg, Ca);
        V3 ge, ae, Kt{{0, 0, 0}};
#pragma unroll
        for (int i = 0; i < 3; ++i) { ge.v[i] = nom[24 + i] + nom[30 + i] * h; ae.v[i] = nom[27 + i] + nom[33 + i] * h; }
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
          for (int i = 0; i < 3; ++i) Kt.v[i] += tab.b[q] * sKs[3 * q + i];
        compose_motion_calib_dev(X, Kt, ge, ae, h, Rg, Cg, Ca);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j) nom[3 * i + j] = X.Rsb.m[i][j];
          nom[9 + i] = X.Tsb.v[i]; nom[12 + i] = X.Vsb.v[i];
          nom[24 + i] = ge.v[i]; nom[27 + i] = ae.v[i];   // rk4.cpp:27-28: the next sub-step starts from the interpolated sample
        }
      }
      auto phase_a = [&](int st) {   // P0 = Pmm + (sum a_q PK_q) h, S = sum a_q FK_q, this stage's F rows and G Q rows
        const double* Jm = Jms + st * JS;
        for (int e = lane; e < NN; e += NT) {
          double sp = 0.0;
#pragma unroll
          for (int q = 0; q < NS - 1; ++q) sp += tab.a[st][q] * PKs[q * NN + e];
          P0[e] = Pmm[e] + sp * h;
        }
        for (int e = lane; e < NF; e += NT) {
          double sf = 0.0;
#pragma unroll
          for (int q = 0; q < NS - 1; ++q) sf += tab.a[st][q] * FKs[q * NF + e];
          S1[e] = sf;
          const int i = e % FR, j = e / FR;
          double f = 0.0;
          if (i < 3) {                                            // Wsb rows
            if (j < 3) f = Jm[3 * i + j];
            else if (j == 9 + i) f = -1.0;
            else if (iCg >= 0 && j >= iCg + 3 * i && j < iCg + 3 * i + 3) f = Jm[36 + (j - iCg - 3 * i)];
          } else if (i < 6) {                                     // Tsb rows
            if (j == 3 + i) f = 1.0;
          } else {                                                // Vsb rows
            const int r = i - 6;
            if (j < 3) f = Jm[9 + 3 * r + j];
            else if (j >= 12 && j < 15) f = Jm[18 + 3 * r + (j - 12)];
            else if (j == 21 || j == 22) f = Jm[27 + 3 * r + (j - 21)];
            else if (iCa >= 0 && j >= iCa && j < iCa + 6) f = Jm[39 + 6 * r + (j - iCa)];
          }
          F9[e] = f;
        }
        if (lane >= 224 && lane < 236) {                          // (G Q)[Vsb_i, l] = sum_k -Rsb[i][k] Q[3 + k, l]
          const int l = lane - 224;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) v = fma(Jm[18 + 3 * i + k], Q[(3 + k) + 12 * l], v);
            GQc[(3 + i) + 12 * l] = v;
          }
        }
      };
      phase_a(0);
      __syncthreads();
      for (int st = 0; st < NS; ++st) {
        const double* Jm = Jms + st * JS;
        // -- phase B: F P0, FK_st = F + F S h, P0 F^T, G Q G^T
        for (int e = lane; e < NF; e += NT) {
          const int i = e % FR, j = e / FR;
          double fp = 0.0, fs = 0.0;
          for (int k = 0; k < nm; ++k) fp = fma(F9[i + FR * k], P0[k + nm * j], fp);
#pragma unroll
          for (int k = 0; k < FR; ++k) fs = fma(F9[i + FR * k], S1[k + FR * j], fs);   // rows >= 9 of S are zero
          FPs[e] = fp;
          FKs[st * NF + e] = F9[e] + fs * h;
        }
        for (int e = lane; e < NF; e += NT) {
          const int i = e % nm, j = e / nm;                       // (P0 F^T)[i, j < 9]
          double pf = 0.0;
          for (int k = 0; k < nm; ++k) pf = fma(P0[i + nm * k], F9[j + FR * k], pf);
          PFs[e] = pf;
        }
        if (lane < 144) {
          const int r = lane % 12, cidx = lane / 12;
          double v;
          if (cidx < 3) v = fma(GQc[r + 12 * cidx], -1.0, 0.0);
          else if (cidx < 6) {
            v = fma(GQc[r + 12 * 3], Jm[18 + 3 * (cidx - 3) + 0], 0.0);
            v = fma(GQc[r + 12 * 4], Jm[18 + 3 * (cidx - 3) + 1], v);
            v = fma(GQc[r + 12 * 5], Jm[18 + 3 * (cidx - 3) + 2], v);
          } else v = GQc[r + 12 * cidx];
          GQG[r + 12 * cidx] = v;
        }
        __syncthreads();
        // -- phase C: PK_st = F P0 + P0 F^T + G Q G^T, then phase A of the next stage
        for (int e = lane; e < NN; e += NT) {
          const int i = e % nm, j = e / nm;
          const int ci = i < 3 ? i : ((i >= 6 && i < 15) ? i - 3 : -1), cj = j < 3 ? j : ((j >= 6 && j < 15) ? j - 3 : -1);
          const double fp = i < FR ? FPs[i + FR * j] : 0.0, pf = j < FR ? PFs[i + nm * j] : 0.0;
          const double gq = (ci >= 0 && cj >= 0) ? GQG[ci + 12 * cj] : 0.0;
          PKs[st * NN + e] = (fp + pf) + gq;
        }
        __syncthreads();
        if (st + 1 < NS) { phase_a(st + 1); __syncthreads(); }
      }
      // combine the stages
      for (int e = lane; e < NN; e += NT) {
        double pk = 0.0;
#pragma unroll
        for (int q = 0; q < NS; ++q) pk += tab.b[q] * PKs[q * NN + e];
        Pmm[e] += pk * h;                              // rk4.cpp:92-93
      }
      for (int e = lane; e < NF; e += NT) {
        double fk = 0.0;
#pragma unroll
        for (int q = 0; q < NS; ++q) fk += tab.b[q] * FKs[q * NF + e];
        S1[e] = ((e % FR) == (e / FR) ? 1.0 : 0.0) + fk * h;    // rows < 9 of Phi_step = I + FK h
      }
      __syncthreads();
      for (int e = lane; e < NF; e += NT) {            // Phi <- Phi_step Phi (rows >= 9 of both are identity rows)
        const int i = e % FR, j = e / FR;
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < FR; ++k) v = fma(S1[i + FR * k], Phi[k + FR * j], v);
        if (j >= FR) v = fma(S1[i + FR * j], 1.0, v);
        PhiN[e] = v;
      }
      { double* t = Phi; Phi = PhiN; PhiN = t; }
      __syncthreads();
      total += h;
      if (ctl) {
        const double err = 0.0;                // PrinceDormandStep returns 0 (:216-220)
        double scale;
        if (err == 0.0) scale = a.pd_max_scale;                                                     // :42-43
        else scale = fmin(fmax(0.8 * sqrt(sqrt(a.pd_tol * h / err)), a.pd_min_scale), a.pd_max_scale);   // :45-47
        hs = h * scale;                                                                             // :51
        if (total < dt) {                                                                           // :52-58
          if (total + hs > dt) hs = dt - total;
          else if (total + hs + 0.5 * hs > dt) hs = 0.5 * hs;
        }
        // the next step starts from gyro0 + slope * total_step (:38-39)
        if (lane < 3) {
          nom[24 + lane] = imu_f[smp].gyro[lane] + imu_f[smp].slope_gyro[lane] * total;
          nom[27 + lane] = imu_f[smp].accel[lane] + imu_f[smp].slope_accel[lane] * total;
        }
        __syncthreads();
      } else if (a.stepsize < 0) break;
    }
    for (int e = lane; e < NN; e += NT) Pmm[e] += a.Qmodel[e];   // estimator.cpp:590, per Propagate
    __syncthreads();
  }
  for (int e = lane; e < NN; e += NT) {
    const int i = e % nm, j = e / nm;
    a.Pmm_out[(long)filt * NN + e] = Pmm[e];
    a.Phi_out[(long)filt * NN + e] = i < FR ? Phi[i + FR * j] : (i == j ? 1.0 : 0.0);
  }
  if (ctl && lane == 0) a.pd_h[filt] = hs;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      pose.Tsb[i] = nom[9 + i]; pose.Vsb[i] = nom[12 + i];
#pragma unroll
      for (int j = 0; j < 3; ++j) pose.Rsb[i + 3 * j] = nom[3 * i + j];
    }
  }
}

// ---------------------------------------------------------------- propagation, one wave per filter
// The workgroup kernel above spends its time in barriers between phases that keep a few dozen lanes busy. Here ONE
// wave owns a filter: nothing waits on another wave, the phases follow each other in program order, and what only
// ever belongs to one lane leaves the LDS:
//  * element e = lane + 64 m (m < 9) of every 23 x 23 matrix is handled by the same lane in the tableau sums, the
//    stage combination and the + Qmodel step, so P_mm and the stage derivatives PK_q live in registers (the stage
//    loop is unrolled: PK_q is a named register set and the tableau a compile-time constant whose zeros cost nothing);
//  * LDS (28 KB RK4 / 34 KB Dormand-Prince: 5 / 4 filters per CU) keeps what crosses lanes: P0, the F products,
//    the FK_q, the transition;
//  * the nominal pre-pass is vectorised over the stages: lane q < NS composes stage q, lane NS the sub-step itself
//    (the same ComposeMotion code with its own sample time, step and velocity; a stage of step 0 composes with the
//    identity, exactly), each lane forming the tableau-weighted velocity from the stage velocities it recomputes.
// Arithmetic per element is that of the workgroup kernel (same ascending-k / ascending-q sums).
template <int NS> struct RkConst;
template <> struct RkConst<4> {
  static constexpr double a[4][3] = {{0, 0, 0}, {0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1.0}};
  static constexpr double c_step[4] = {0.0, 0.5, 0.5, 1.0};
  static constexpr double c_imu[4] = {0.0, 0.5, 0.5, 0.5};
  static constexpr double b[4] = {1 / 6.0, 2 / 6.0, 2 / 6.0, 1 / 6.0};
};
template <> struct RkConst<7> {
  static constexpr double a[7][6] = {{0},
                                     {2 / 9.0},
                                     {1 / 12.0, 3 / 12.0},
                                     {55 / 324.0, -75 / 324.0, 200 / 324.0},
                                     {83 / 330.0, -195 / 330.0, 305 / 330.0, 27 / 330.0},
                                     {-19 / 28.0, 63 / 28.0, 4 / 28.0, -108 / 28.0, 88 / 28.0},
                                     {38 / 400.0, 0.0, 240 / 400.0, -243 / 400.0, 330 / 400.0, 35 / 400.0}};
  static constexpr double c_step[7] = {0.0, 2 / 9.0, 3 / 9.0, 5 / 9.0, 6 / 9.0, 1.0, 1.0};
  static constexpr double c_imu[7] = {0.0, 2 / 9.0, 3 / 9.0, 5 / 9.0, 6 / 9.0, 1.0, 1.0};
  static constexpr double b[7] = {0.0862, 0.0, 0.6660, -0.7857, 0.9570, 0.0965, -0.0200};
};

template <int NS>
__global__ __launch_bounds__(64) void propagate_state_wave_kernel(PropStateArgs a) {
  constexpr int NM = 23, NN = NM * NM, FR = 9, NF = FR * NM, EL = 9, FL = 4;   // EL / FL: elements of a 23 x 23 / 9 x 23 matrix per lane
  using TB = RkConst<NS>;
  extern __shared__ double sm[];
  const int lane = threadIdx.x, filt = blockIdx.x;
  const RkTableau& tab = kTableau[NS == 4 ? 0 : 1];
  double* P0 = sm;
  double* S1 = P0 + NN;        // 23 x 23 scratch whose rows >= 9 stay zero: sum a_q FK_q, later I + FK h
  double* FPs = S1 + NN;       // [9 x 23]  F P0   ([i + 9 j])
  double* GQG = FPs + NF;      // [12 x 12] support of G Q G^T
  double* Q = GQG + 144;
  double* GQc = Q + 144;       // [12 x 12] the non-zero rows of G Q
  double* zero = GQc + 144;    // one 0.0 + pad
  double* nom = zero + 2;      // nominal state + IMU sample (layout of the workgroup kernel)
  double* Jms = nom + 36;      // [NS][4][3 x 3]
  double* F9 = Jms + NS * 36;  // [9 x 23] dense non-zero rows of F of the current stage
  double* FKs = F9 + NF;       // [NS][9 x 23]
  double* PhiA = FKs + NS * NF;

  const double* Pg = a.P + (long)filt * a.strideP;
  // Lanes past the end of a matrix repeat its last element (clamped index): no predicates in the loops, the copies
  // hold identical values and only the owner stores at the end.
  double Pmm[EL], PK[NS][EL];
  int off[EL];                 // the three terms of PK(e) as packed 8-bit offsets into FPs / FPs (transposed entry) / GQG (255: absent)
#pragma unroll
  for (int m = 0; m < EL; ++m) {
    const int e = min(lane + 64 * m, NN - 1), i = e % NM, j = e / NM;
    const int ci = i < 3 ? i : ((i >= 6 && i < 15) ? i - 3 : -1), cj = j < 3 ? j : ((j >= 6 && j < 15) ? j - 3 : -1);
    Pmm[m] = Pg[i + (long)j * a.ldp];
    off[m] = (i < FR ? i + FR * j : 255) | ((j < FR ? j + FR * i : 255) << 8) | (((ci >= 0 && cj >= 0) ? ci + 12 * cj : 255) << 16);   // F P0, (F P0)^T, G Q G^T
    S1[e] = 0.0;
#pragma unroll
    for (int q = 0; q < NS; ++q) PK[q][m] = 0.0;
  }
#pragma unroll
  for (int u = 0; u < FL; ++u) {
    const int f = min(lane + 64 * u, NF - 1);
    const int i = f % FR, j = f / FR;
    PhiA[f] = i == j ? 1.0 : 0.0;
    F9[f] = (i < 3 && j == 9 + i) ? -1.0 : ((i >= 3 && i < 6 && j == 3 + i) ? 1.0 : 0.0);   // dWsb/dbg = -I, dTsb/dVsb = I
#pragma unroll
    for (int q = 0; q < NS; ++q) FKs[q * NF + f] = 0.0;
  }
  if (lane == 0) zero[0] = 0.0;
  for (int e = lane; e < 144; e += 64) {
    const double q = a.Qimu[e];
    Q[e] = q;
    const int r = e % 12;      // rows of G Q that do not depend on the state: Wsb rows = -Q[0:3,:], bg / ba rows = Q[6:12,:]
    if (r < 3) GQc[e] = -q;
    else if (r >= 6) GQc[e] = q;
  }
  double* Phi = PhiA;
  xivo_pose_in& pose = a.poses[filt];
  const V3 gv{{a.g[0], a.g[1], a.g[2]}};
  if (lane == 0) {
    const V3 Rg0 = m3_mulv(m3_from_colmajor(pose.Rsg), gv);   // Rsg g (estimator.cpp:609)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) nom[3 * i + j] = pose.Rsb[i + 3 * j];
      nom[9 + i] = pose.Tsb[i]; nom[12 + i] = pose.Vsb[i]; nom[15 + i] = pose.bg[i]; nom[18 + i] = pose.ba[i];
      nom[21 + i] = Rg0.v[i];
    }
  }
  __syncthreads();   // (one wave: orders the LDS traffic, no waiting)

  const xivo_imu_in* imu_f = a.imu + (long)filt * a.n_imu;
  const int c3 = lane < 3 ? lane : 0;
  double n_g = imu_f[0].gyro[c3], n_a = imu_f[0].accel[c3], n_sg = imu_f[0].slope_gyro[c3], n_sa = imu_f[0].slope_accel[c3];
  double n_dt = imu_f[0].dt;
  // step-size-controlled Dormand-Prince (princedormand.cpp:26-60): the step the last sample - or the last call - left behind
  const bool ctl = NS == 7 && a.pd_h != nullptr;
  double hs = ctl ? a.pd_h[filt] : 0.0;
  for (int smp = 0; smp < a.n_imu; ++smp) {
    if (lane < 3) { nom[24 + lane] = n_g; nom[27 + lane] = n_a; nom[30 + lane] = n_sg; nom[33 + lane] = n_sa; }
    const double dt = n_dt;
    const double cur_g = n_g, cur_a = n_a, cur_sg = n_sg, cur_sa = n_sa;     // (lanes 0-2: this sample as it arrived)
    if (ctl) {
      if (hs < 1e-6) hs = a.stepsize;        // :30-32
      hs = fmin(hs, dt);                     // :34
    }
    if (smp + 1 < a.n_imu) {
      const xivo_imu_in& nx = imu_f[smp + 1];
      n_g = nx.gyro[c3]; n_a = nx.accel[c3]; n_sg = nx.slope_gyro[c3]; n_sa = nx.slope_accel[c3]; n_dt = nx.dt;
    }
    __syncthreads();
    double total = 0.0;
    // fixed sub-stepping with the half-step tail trick (rk4.cpp:13-32, princedormand.cpp:62-81)
    while (total < dt || (!ctl && a.stepsize < 0)) {
      double h = a.stepsize;
      if (ctl) h = hs;
      else if (a.stepsize < 0) h = dt;
      else if (total + h > dt) h = dt - total;
      else if (total + h + 0.5 * h > dt) h = 0.5 * h;

      {  // -- nominal pre-pass: lane q < NS = stage q, lane NS = the sub-step itself
        MotionRegs X0; V3 Rg;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j) X0.Rsb.m[i][j] = nom[3 * i + j];
          X0.Tsb.v[i] = nom[9 + i]; X0.Vsb.v[i] = nom[12 + i]; X0.bg.v[i] = nom[15 + i]; X0.ba.v[i] = nom[18 + i];
          Rg.v[i] = nom[21 + i];
        }
        V3 g0, a0, sg, sa;
#pragma unroll
        for (int i = 0; i < 3; ++i) { g0.v[i] = nom[24 + i]; a0.v[i] = nom[27 + i]; sg.v[i] = nom[30 + i]; sa.v[i] = nom[33 + i]; }
        // the tableau-weighted velocity: K_q = Vsb of stage q's ComposeMotion (estimator.cpp:609)
        V3 Kt{{0, 0, 0}};
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const double tq = TB::c_imu[q] * h, dq = TB::c_step[q] * h;
          V3 ac;
#pragma unroll
          for (int i = 0; i < 3; ++i) ac.v[i] = (a0.v[i] + sa.v[i] * tq) - X0.ba.v[i];
          const V3 Ra = m3_mulv(X0.Rsb, ac);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            double kq = X0.Vsb.v[i];
            if (q > 0) kq += (Ra.v[i] + Rg.v[i]) * dq;
            Kt.v[i] += TB::b[q] * kq;
          }
        }
        const bool is_step = lane == NS;
        const int st = lane < NS ? lane : 0;
        const double ti = is_step ? h : tab.c_imu[st] * h;
        const double ds = is_step ? h : tab.c_step[st] * h;
        V3 gi, ai, V;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          gi.v[i] = g0.v[i] + sg.v[i] * ti; ai.v[i] = a0.v[i] + sa.v[i] * ti;
          V.v[i] = is_step ? Kt.v[i] : 0.0;   // the a_ij-weighted velocities of a stage only move Tsb, which no Jacobian reads
        }
        compose_motion_dev(X0, V, gi, ai, ds, Rg);
        // ComputeMotionJacobianAt (estimator.cpp:615-704): the blocks of F and G
        V3 gc, ac;
#pragma unroll
        for (int i = 0; i < 3; ++i) { gc.v[i] = gi.v[i] - X0.bg.v[i]; ac.v[i] = ai.v[i] - X0.ba.v[i]; }
        const M3 w_dW_dW = m3_neg(hat(gc));
        const M3 w_dV_dW = m3_neg(m3_mul(X0.Rsb, hat(ac)));
        const M3 w_dV_dWsg = m3_neg(m3_mul(X0.Rsb, hat(gv)));
        const M3 w_nR = m3_neg(X0.Rsb);
        if (lane < NS) {
          double* Jm = Jms + st * 36;
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              Jm[3 * i + j] = w_dW_dW.m[i][j]; Jm[9 + 3 * i + j] = w_dV_dW.m[i][j];
              Jm[18 + 3 * i + j] = w_nR.m[i][j]; Jm[27 + 3 * i + j] = w_dV_dWsg.m[i][j];
            }
        } else if (is_step) {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) nom[3 * i + j] = X0.Rsb.m[i][j];
            nom[9 + i] = X0.Tsb.v[i]; nom[12 + i] = X0.Vsb.v[i];
            nom[24 + i] = gi.v[i]; nom[27 + i] = ai.v[i];   // rk4.cpp:27-28: the next sub-step starts from the interpolated sample
          }
        }
      }
      __syncthreads();

#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const double* Jm = Jms + st * 36;
        // -- phase A: P0 = Pmm + (sum_q a_q PK_q) h, S = sum_q a_q FK_q (rows < 9), the stage's entries of F, Vsb rows of G Q
#pragma unroll
        for (int m = 0; m < EL; ++m) {
          const int e = min(lane + 64 * m, NN - 1);
          double sp = 0.0;
#pragma unroll
          for (int q = 0; q < st; ++q)
            if (TB::a[st][q] != 0.0) sp += TB::a[st][q] * PK[q][m];
          P0[e] = Pmm[m] + sp * h;
        }
#pragma unroll
        for (int u = 0; u < FL; ++u) {
          const int f = min(lane + 64 * u, NF - 1);
          double sf = 0.0;
#pragma unroll
          for (int q = 0; q < st; ++q)
            if (TB::a[st][q] != 0.0) sf += TB::a[st][q] * FKs[q * NF + f];
          S1[(f % FR) + NM * (f / FR)] = sf;
        }
        if (lane < 33) {                                            // the stage's 33 state-dependent entries of F
          const int l = lane, blk = l < 27 ? l / 9 : 3, m = l - 9 * blk;
          const int i = blk < 3 ? m / 3 : m / 2, j = blk < 3 ? m % 3 : m % 2;
          const int row = blk == 0 ? i : 6 + i, col = blk < 2 ? j : (blk == 2 ? 12 + j : 21 + j);
          F9[row + FR * col] = Jm[9 * blk + 3 * i + j];
        } else if (lane < 45) {                                     // (G Q)[Vsb_i, l] = sum_k -Rsb[i][k] Q[3 + k, l]
          const int l = lane - 33;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) v = fma(Jm[18 + 3 * i + k], Q[(3 + k) + 12 * l], v);
            GQc[(3 + i) + 12 * l] = v;
          }
        }
        __syncthreads();

        // -- phase B: the structured products (the published 3 x 3 blocks once into registers)
        M3 dW_dW, dV_dW, nR, dV_dWsg;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            dW_dW.m[i][j] = Jm[3 * i + j]; dV_dW.m[i][j] = Jm[9 + 3 * i + j];
            nR.m[i][j] = Jm[18 + 3 * i + j]; dV_dWsg.m[i][j] = Jm[27 + 3 * i + j];
          }
        {
          // lanes 0..22: column j of F P0, lanes 32..54: column j of F S and from it FK of the stage
          const bool fk_task = lane >= 32;
          const int j = fk_task ? lane - 32 : lane;
          if (j < NM) {
            const double* M = (fk_task ? S1 : P0) + NM * j;
            double o[9];
            const double m0 = M[0], m1 = M[1], m2 = M[2], m12 = M[12], m13 = M[13], m14 = M[14], m21 = M[21], m22 = M[22];
#pragma unroll
            for (int i = 0; i < 3; ++i) {                           // Wsb rows: k = 0..2 (dW/dW), k = 9 + i (-1)
              double v = fma(dW_dW.m[i][0], m0, 0.0);
              v = fma(dW_dW.m[i][1], m1, v);
              v = fma(dW_dW.m[i][2], m2, v);
              o[i] = fma(-1.0, M[9 + i], v);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) o[3 + i] = fma(1.0, M[6 + i], 0.0);   // Tsb rows: k = 6 + i (1)
#pragma unroll
            for (int i = 0; i < 3; ++i) {                           // Vsb rows: k = 0..2, 12..14, 21..22
              double v = fma(dV_dW.m[i][0], m0, 0.0);
              v = fma(dV_dW.m[i][1], m1, v);
              v = fma(dV_dW.m[i][2], m2, v);
              v = fma(nR.m[i][0], m12, v);
              v = fma(nR.m[i][1], m13, v);
              v = fma(nR.m[i][2], m14, v);
              v = fma(dV_dWsg.m[i][0], m21, v);
              o[6 + i] = fma(dV_dWsg.m[i][1], m22, v);
            }
            if (fk_task) {                                          // FK_st = F + F S h
#pragma unroll
              for (int i = 0; i < FR; ++i) FKs[st * NF + i + FR * j] = F9[i + FR * j] + o[i] * h;
            } else {
#pragma unroll
              for (int i = 0; i < FR; ++i) FPs[i + FR * j] = o[i];
            }
          }
        }
        // (P0 F^T is not formed: P0 is symmetric - P_mm and every stage derivative are, bit for bit but for the
        //  rounding-level asymmetry of G Q G^T - so (P0 F^T)[i][j] = (F P0)[j][i], the same products summed in the same order)
        if (lane >= 32 && lane < 44) {                              // (G Q G^T)[r, :] on the 12 x 12 support
          const int r = lane - 32;
          const double g3 = GQc[r + 12 * 3], g4 = GQc[r + 12 * 4], g5 = GQc[r + 12 * 5];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            GQG[r + 12 * j] = fma(GQc[r + 12 * j], -1.0, 0.0);     // Wsb columns: G[Wsb_j, j] = -1
            double v = fma(g3, nR.m[j][0], 0.0);                    // Vsb columns: G[Vsb_j, 3..5] = -Rsb[j][:]
            v = fma(g4, nR.m[j][1], v);
            GQG[r + 12 * (3 + j)] = fma(g5, nR.m[j][2], v);
            GQG[r + 12 * (6 + j)] = GQc[r + 12 * (6 + j)];          // bg, ba columns: +1
            GQG[r + 12 * (9 + j)] = GQc[r + 12 * (9 + j)];
          }
        }
        __syncthreads();

        // -- phase C: PK_st = F P0 + P0 F^T + G Q G^T
#pragma unroll
        for (int m = 0; m < EL; ++m) {
          const int o1 = off[m] & 255, o2 = (off[m] >> 8) & 255, o3 = (off[m] >> 16) & 255;
          const double t1 = o1 == 255 ? 0.0 : FPs[o1], t2 = o2 == 255 ? 0.0 : FPs[o2], t3 = o3 == 255 ? 0.0 : GQG[o3];
          PK[st][m] = (t1 + t2) + t3;
        }
        // (phase A of the next stage writes P0 / S1 / F9 / GQc, which phase B above has finished reading; FPs / PFs / GQG are
        //  rewritten by the next phase B only, after the reads just issued - program order within the one wave)
      }
      // combine the stages
#pragma unroll
      for (int m = 0; m < EL; ++m) {
        double pk = 0.0;
#pragma unroll
        for (int q = 0; q < NS; ++q)
          if (TB::b[q] != 0.0) pk += TB::b[q] * PK[q][m];
        Pmm[m] += pk * h;                              // rk4.cpp:92-93
      }
#pragma unroll
      for (int u = 0; u < FL; ++u) {
        const int f = min(lane + 64 * u, NF - 1);
        const int i = f % FR, j = f / FR;
        double fk = 0.0;
#pragma unroll
        for (int q = 0; q < NS; ++q)
          if (TB::b[q] != 0.0) fk += TB::b[q] * FKs[q * NF + f];
        S1[i + NM * j] = (i == j ? 1.0 : 0.0) + fk * h;     // rows < 9 of Phi_step = I + FK h
      }
      __syncthreads();
      {                                                // Phi <- Phi_step Phi (rows >= 9 of both are identity rows), in place:
        double pv[FL];                                 // every lane has read its column before any lane writes
#pragma unroll
        for (int u = 0; u < FL; ++u) {
          const int f = min(lane + 64 * u, NF - 1);
          const int i = f % FR, j = f / FR;
          double v = 0.0;
#pragma unroll
          for (int k = 0; k < FR; ++k) v = fma(S1[i + NM * k], Phi[k + FR * j], v);
          const double tail = fma(S1[i + NM * j], 1.0, v);
          pv[u] = j >= FR ? tail : v;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < FL; ++u) Phi[min(lane + 64 * u, NF - 1)] = pv[u];
      }
      __syncthreads();
      total += h;
      if (ctl) {
        const double err = 0.0;                // PrinceDormandStep returns 0: its error estimate is commented out (:216-220)
        double scale;
        if (err == 0.0) scale = a.pd_max_scale;                                                     // :42-43
        else scale = fmin(fmax(0.8 * sqrt(sqrt(a.pd_tol * h / err)), a.pd_min_scale), a.pd_max_scale);   // :45-47
        hs = h * scale;                                                                             // :51
        if (total < dt) {                                                                           // :52-58
          if (total + hs > dt) hs = dt - total;
          else if (total + hs + 0.5 * hs > dt) hs = 0.5 * hs;
        }
        // the next step starts from gyro0 + slope * total_step (:38-39), not from the running sum of the fixed-step branch
        if (lane < 3) { nom[24 + lane] = cur_g + cur_sg * total; nom[27 + lane] = cur_a + cur_sa * total; }
        __syncthreads();
      } else if (a.stepsize < 0) break;
    }
#pragma unroll
    for (int m = 0; m < EL; ++m) Pmm[m] += a.Qmodel[min(lane + 64 * m, NN - 1)];   // P_mm += Qmodel (estimator.cpp:590), per Propagate
  }
  __syncthreads();
  if (ctl && lane == 0) a.pd_h[filt] = hs;
#pragma unroll
  for (int m = 0; m < EL; ++m) {
    const int e = lane + 64 * m;
    if (e < NN) a.Pmm_out[(long)filt * NN + e] = Pmm[m];
  }
  for (int e = lane; e < NN; e += 64) {
    const int i = e % NM, j = e / NM;
    a.Phi_out[(long)filt * NN + e] = i < FR ? Phi[i + FR * j] : (i == j ? 1.0 : 0.0);
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      pose.Tsb[i] = nom[9 + i]; pose.Vsb[i] = nom[12 + i];
#pragma unroll
      for (int j = 0; j < 3; ++j) pose.Rsb[i + 3 * j] = nom[3 * i + j];
    }
  }
}

// ---------------------------------------------------------------- measurement compression of the OOS rows
// use_compression_ / compression_trigger_ratio_ (src/estimator.h:399-402) and xivo::QR (src/helpers.cpp:77-101, "QR-based
// measurement compression") are parsed / defined but never run by the reference's pipeline. Here the block of
// null-space-projected OOS rows appended under the in-state rows (rows [row0, row0 + rows_b) of the stacked H) is
// replaced by the triangular factor of its QR decomposition whenever it has more than `ratio` times as many rows as
// non-zero columns: the rows only touch the camera-extrinsics and group columns, so 7 rows per feature collapse to at
// most 6 + 6 * n_groups rows for the whole block. Orthogonal row operations with isotropic noise Roos leave the update
// (S, K, dx, P+) unchanged to rounding (tests compare against the uncompressed oracle update and check
// Hc^T Hc = H^T H, Hc^T rc = H^T r).
// One workgroup of four waves per filter: lane = candidate column (CPL columns per lane: [Wbc Tbc | group slots], the
// residual rides along as one more column), wave w keeps rows [w RW, (w + 1) RW) of its columns in registers.
// Householder reflections column by column (a column that is zero from the pivot row down is skipped), v broadcast
// from the pivot lane with v_readlane, the per-column dot products reduced over the four waves through LDS: two
// barriers per column, no dynamic register indexing (the row loops are unrolled and predicated on r >= pivot row).
template <int RW, int CPL>
__global__ __launch_bounds__(256) void oos_compress_kernel(OosCompressArgs a) {
  const int filt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rows = a.rows[filt];
  const int ncand = 6 + 6 * a.lay.n_groups;            // candidate columns; column index ncand = the residual
  double* H = a.mb.H + (long)filt * a.mb.strideH;
  double* HT = a.mb.HT + (long)filt * a.mb.strideHT;
  double* inn = a.mb.inn + (long)filt * a.mb.strideInn;
  double* dR = a.mb.diagR + (long)filt * a.mb.strideR;
  __shared__ double sdot[4][64 * CPL];
  __shared__ double ssum[4], spiv[4];
  auto col_of = [&](int jj) -> int { return jj < 6 ? 15 + jj : a.lay.group_begin + (jj - 6); };
  double v[CPL][RW];
  // rows this wave holds, as a VECTOR value (a scalar one makes the compiler keep RW row predicates in SGPRs)
  int nloc;
  asm volatile("v_mov_b32 %0, %1" : "=v"(nloc) : "s"(rows - wave * RW));
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int jj = lane + 64 * q;
    // column jj of the block: contiguous over the rows (H is column-major); the residual is one more column
    const double* src = jj < ncand ? H + (a.row0 + wave * RW) + (long)col_of(jj) * a.mb.ldh : inn + (a.row0 + wave * RW);
    const bool have = jj <= ncand;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) v[q][rr] = (have && rr < nloc) ? src[rr] : 0.0;
  }
  // trigger: rows > ratio * (non-zero candidate columns)
  __shared__ unsigned long long smask[4][CPL];
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    bool any = false;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) any = any || v[q][rr] != 0.0;
    const unsigned long long m = __ballot(any && lane + 64 * q < ncand);
    if (lane == 0) smask[wave][q] = m;
  }
  __syncthreads();
  int nzc = 0;
#pragma unroll
  for (int q = 0; q < CPL; ++q) nzc += __popcll(smask[0][q] | smask[1][q] | smask[2][q] | smask[3][q]);
  if (!((double)rows > a.ratio * (double)nzc) || rows <= 1) {
    if (tid == 0) a.rows_out[filt] = rows;
    return;
  }
  // Per column (= per reflection) the work of a wave is two passes over its RW register rows: the dot products
  // v^T A[:, c] and the rank-1 update. Everything that depends on the row index relative to the pivot row is folded
  // into the broadcast vector the owner lane publishes in LDS (zero above the pivot row, x_p - alpha at it), so both
  // passes are ds_read (broadcast) + v_fma per row, nothing else: the kernel is bound by VALU issue.
  __shared__ double scol[4][RW];
  __shared__ double salpha[64 * CPL];
  __shared__ int spivrow[64 * CPL];
  for (int c = tid; c < 64 * CPL; c += 256) spivrow[c] = -1;
  __syncthreads();
  int p = 0;   // pivot row = number of reflections applied so far
  for (int j = 0; j < ncand && p < rows; ++j) {
    const int jl = j & 63, jq = j >> 6;
    // pivot row relative to this wave's first row, in a VECTOR register on purpose (a scalar one makes the compiler
    // keep the RW row predicates of the unrolled loop as 64-bit lane masks in SGPRs, which spill)
    int pl;
    asm volatile("v_mov_b32 %0, %1" : "=v"(pl) : "s"(p - wave * RW));
    if (lane == jl) {    // the owner lane publishes column j masked to the rows >= p, its squared norm, the pivot element
      double s = 0.0, x0 = 0.0;
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        if (q != jq) continue;
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
          const double e = rr >= pl ? v[q][rr] : 0.0;
          scol[wave][rr] = e;
          s = fma(e, e, s);
          if (rr == pl) x0 = e;
        }
      }
      ssum[wave] = s; spiv[wave] = x0;
    }
    __syncthreads();
    const double stot = ssum[0] + ssum[1] + ssum[2] + ssum[3];
    const double xp = spiv[p / RW];       // from the wave that owns row p
    if (stot == 0.0) { __syncthreads(); continue; }       // nothing from the pivot row down in this column: no reflection
    const double nrm = sqrt(stot);
    const double alpha = xp > 0.0 ? -nrm : nrm;
    const double beta = 1.0 / (stot - xp * alpha);        // H = I - beta v v^T, v = x - alpha e_p
    if (wave == p / RW && lane == jl) { scol[wave][p % RW] = xp - alpha; salpha[j] = alpha; spivrow[j] = p; }
    __syncthreads();
    double dot[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) dot[q] = 0.0;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const double vr = scol[wave][rr];
#pragma unroll
      for (int q = 0; q < CPL; ++q) dot[q] = fma(vr, v[q][rr], dot[q]);
    }
#pragma unroll
    for (int q = 0; q < CPL; ++q) sdot[wave][lane + 64 * q] = dot[q];
    __syncthreads();
    double w[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int c = lane + 64 * q;
      w[q] = beta * (sdot[0][c] + sdot[1][c] + sdot[2][c] + sdot[3][c]);
    }
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const double vr = scol[wave][rr];
#pragma unroll
      for (int q = 0; q < CPL; ++q) v[q][rr] = fma(-vr, w[q], v[q][rr]);
    }
    ++p;
  }
  __syncthreads();
  // rows [0, p): the triangular factor - the reflected columns exactly (alpha on the pivot, zero below it; the
  // arithmetic leaves rounding-level residue there); rows [p, rows): exactly neutral
  int pl;
  asm volatile("v_mov_b32 %0, %1" : "=v"(pl) : "s"(p - wave * RW));
#pragma unroll
  for (int q = 0; q < CPL; ++q) {
    const int jj = lane + 64 * q;
    if (jj < ncand) {
      const int col = col_of(jj);
      const int prow = spivrow[jj];                       // -1: never a pivot column (zero from its turn on)
      const double al = prow >= 0 ? salpha[jj] : 0.0;
      int prl;                                            // pivot row of this column relative to the wave's rows
      asm volatile("v_mov_b32 %0, %1" : "=v"(prl) : "v"(prow - wave * RW));
      double* hd = H + (a.row0 + wave * RW) + (long)col * a.mb.ldh;
      double* ht = HT + col + (long)(a.row0 + wave * RW) * a.mb.ldht;
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        if (rr < nloc) {
          double x = rr < pl ? v[q][rr] : 0.0;
          if (prow >= 0 && rr >= prl) x = rr == prl ? al : 0.0;
          hd[rr] = x;
          if (a.mb.HT) *ht = x;
        }
        ht += a.mb.ldht;
      }
    } else if (jj == ncand) {
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        if (rr < nloc) {
          inn[a.row0 + wave * RW + rr] = rr < pl ? v[q][rr] : 0.0;
          dR[a.row0 + wave * RW + rr] = rr < pl ? a.Roos : 1.0;
        }
      }
    }
  }
  if (tid == 0) a.rows_out[filt] = p;
}

// ---------------------------------------------------------------- Estimator::OnePointRANSAC (src/update.cpp:213-393)
// select: the low-innovation set among the MH inliers (:238-258 - the hypothesis index k is drawn but never used, so
// the maximal set is {f : |xp - Predict| < ransac_thresh_}; xp - Predict is the innovation of the Jacobian pass), the
// groups that hold one, the temporary reference group when gauge_group_ptr_ holds none (FindNewRefGroup,
// src/estimator.cpp:1394-1407: smallest summed 6 diagonal entries of P, ascending slot order) and what has to be
// zeroed in P (:299-316). state: 0 = every MH inlier is low-innovation (or there is none): nothing to do (:263-265);
// 1 = partial update + rescue; 2 = no low-innovation inlier: rescue against the prior (:287 guard).
__global__ __launch_bounds__(64) void ransac_select_kernel(RansacArgs a) {
  const int filt = blockIdx.x, lane = threadIdx.x;
  const SceneBuffers& sb = a.sb;
  const double* P = a.P + (long)filt * a.strideP;
  unsigned long long active = 0, withlow = 0;
  int n_mh = 0, n_low = 0;
  for (int f0 = 0; f0 < sb.F; f0 += 64) {
    const int f = f0 + lane;
    bool mh = false, low = false;
    int ref = 0;
    if (f < sb.F) {
      const xivo_feat_in& ft = sb.feats[(long)filt * sb.Fmax + f];
      mh = sb.mask[(long)filt * sb.Fmax + f] && ft.sind >= 0;
      ref = mh ? ft.ref_sind : 0;
      const double r0 = sb.finn[((long)filt * sb.Fmax + f) * 2], r1 = sb.finn[((long)filt * sb.Fmax + f) * 2 + 1];
      low = mh && sqrt(r0 * r0 + r1 * r1) < a.thresh;
      a.low[(long)filt * sb.Fmax + f] = low ? 1 : 0;
    }
    n_mh += __popcll(__ballot(mh));
    n_low += __popcll(__ballot(low));
    unsigned long long ma = mh ? 1ull << ref : 0ull, ml = low ? 1ull << ref : 0ull;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      ma |= __shfl_xor(ma, o);
      ml |= __shfl_xor(ml, o);
    }
    active |= ma; withlow |= ml;
  }
  int state = 1;
  unsigned long long zg = 0;
  if (n_mh == 0 || n_low == n_mh) state = 0;
  else if (n_low == 0) state = 2;
  if (state == 1) {
    const int gauge = a.gauge ? a.gauge[filt] : -1;
    if (gauge < 0 || gauge >= 64 || !((withlow >> gauge) & 1ull)) {
      double best = __builtin_inf();
      int arg = -1;
      for (int g = 0; g < a.lay.n_groups; ++g) {                    // wave-uniform loop
        if (!((withlow >> g) & 1ull)) continue;
        const int off = a.lay.group_begin + 6 * g;
        double cov = 0.0;
        for (int i = 0; i < 6; ++i) cov += P[(off + i) + (long)(off + i) * a.ldp];
        if (cov < best) { best = cov; arg = g; }
      }
      if (arg >= 0) zg |= 1ull << arg;
    }
    zg |= active & ~withlow;
  } else {
    // nothing is updated for this filter: an all-neutral measurement set leaves P and the state untouched
    for (int f = lane; f < sb.F; f += 64) a.low[(long)filt * sb.Fmax + f] = 0;
  }
  if (lane == 0) { a.state[filt] = state; a.zero_groups[filt] = zg; }
}

// P rows / columns of the MH inliers outside the low-innovation set and of the groups named by select (:299-316)
__global__ __launch_bounds__(256) void ransac_zero_kernel(RansacArgs a, double* Pall) {
  const int filt = blockIdx.x, tid = threadIdx.x;
  if (a.state[filt] != 1) return;
  const SceneBuffers& sb = a.sb;
  double* P = Pall + (long)filt * a.strideP;
  const unsigned long long zg = a.zero_groups[filt];
  for (int g = 0; g < a.lay.n_groups; ++g) {
    if (!((zg >> g) & 1ull)) continue;
    const int off = a.lay.group_begin + 6 * g;
    for (int t = tid; t < a.Np; t += 256)
      for (int r = 0; r < 6; ++r) { P[(off + r) + (long)t * a.ldp] = 0.0; P[t + (long)(off + r) * a.ldp] = 0.0; }
  }
  for (int f = 0; f < sb.F; ++f) {
    const xivo_feat_in& ft = sb.feats[(long)filt * sb.Fmax + f];
    if (!(sb.mask[(long)filt * sb.Fmax + f] && ft.sind >= 0) || a.low[(long)filt * sb.Fmax + f]) continue;
    const int off = a.lay.feature_begin + 3 * ft.sind;
    for (int t = tid; t < a.Np; t += 256)
      for (int r = 0; r < 3; ++r) { P[(off + r) + (long)t * a.ldp] = 0.0; P[t + (long)(off + r) * a.ldp] = 0.0; }
  }
}

// rescue (:343-369): chi-square test of every MH inlier outside the low-innovation set with the Jacobians re-taken at
// the partially updated state against the partially updated P; the final inlier mask replaces the MH mask.
__global__ __launch_bounds__(256) void ransac_rescue_kernel(RansacArgs a) {
  const int filt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const SceneBuffers& sb = a.sb;
  const double* P = a.P + (long)filt * a.strideP;
  const int state = a.state[filt];
  __shared__ int s_rej;
  __shared__ double s_scr[4][484];   // feature_chi2_lds: the 21 x 21 sub-block of P and the two rows of J, per wave
  if (tid == 0) s_rej = 0;
  __syncthreads();
  for (int f = wave; f < sb.F; f += 4) {
    const long e = (long)filt * sb.Fmax + f;
    const xivo_feat_in& ft = sb.feats[e];
    const bool mh = sb.mask[e] && ft.sind >= 0;
    double d = 0.0;
    bool keep = mh;
    if (mh && state != 0 && !a.low_keep[e]) {
      d = feature_chi2_lds(P, a.ldp, a.lay, ft, sb.J + e * 42, sb.finn + e * 2, a.R, lane, s_scr[wave]);
      keep = d < a.chi2;
      if (!keep && lane == 0) atomicAdd(&s_rej, 1);
    }
    if (lane == 0) { a.keep[e] = keep ? 1 : 0; a.chi[e] = d; }
  }
  __syncthreads();
  if (tid == 0) a.n_rejected[filt] = s_rej;
}

// the same decision from distances that are already formed (online-calibration builds: the whole-row distances of the
// dense-row gate - J() carries the td / Cg / bg / intrinsics blocks there, which the compact 21-column form does not hold)
__global__ __launch_bounds__(256) void ransac_rescue_dist_kernel(RansacArgs a, const double* dist, int ld) {
  const int filt = blockIdx.x, tid = threadIdx.x;
  const SceneBuffers& sb = a.sb;
  const int state = a.state[filt];
  __shared__ int s_rej;
  if (tid == 0) s_rej = 0;
  __syncthreads();
  for (int f = tid; f < sb.F; f += 256) {
    const long e = (long)filt * sb.Fmax + f;
    const bool mh = sb.mask[e] && sb.feats[e].sind >= 0;
    double d = 0.0;
    bool keep = mh;
    if (mh && state != 0 && !a.low_keep[e]) {
      d = dist[(long)filt * ld + f];
      keep = d < a.chi2;
      if (!keep) atomicAdd(&s_rej, 1);
    }
    a.keep[e] = keep ? 1 : 0; a.chi[e] = d;
  }
  __syncthreads();
  if (tid == 0) a.n_rejected[filt] = s_rej;
}

// ---------------------------------------------------------------- fp64 MFMA issue-rate probe
// Every wave issues `iters` x 8 independent v_mfma_f64_16x16x4_f64; wave 0 of
// block 0 also reports the shader-clock cycles it spent (s_memtime), so the
// host can separate "cycles per MFMA" from "sustained clock".
__global__ __launch_bounds__(256) void mfma_peak_kernel(double* sink, int iters) {
  d4 acc[8];
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    // accumulators pinned to VGPRs: left to itself the compiler parks the accumulators of a small kernel like this one in
    // AGPRs, and v_mfma_f64_16x16x4_f64 with an AGPR destination issues at 50 TFLOP/s instead of 77 on this part
    // (scripts/mfma_agpr_probe.hip) - rounds 1 and 2 took that for the instruction's ceiling. The update kernels keep
    // their accumulators in VGPRs (checked in the ISA), so 77 TFLOP/s = 98 % of the datasheet is the ceiling that applies.
    for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  const long long t1 = clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) sink[1] = (double)(t1 - t0);
  if (s == 12345.678) sink[0] = s;
}

}  // namespace

#define CHECK_LAUNCH() return (int)hipGetLastError()

int launch_unpack_P(const double* raw, double* P, int N, int Np, int ldp, long strideP, int batch,
                    hipStream_t s) {
  dim3 grid((unsigned)p_unpack_pairs(Np), batch);
  hipLaunchKernelGGL(unpack_P_kernel, grid, dim3(256), 0, s, raw, P, N, Np, ldp, strideP);
  CHECK_LAUNCH();
}
int launch_pack_P(const double* P, double* raw, int N, int ldp, long strideP, int batch, hipStream_t s) {
  dim3 grid((unsigned)(((long)N * N + 255) / 256), batch);
  hipLaunchKernelGGL(pack_P_kernel, grid, dim3(256), 0, s, P, raw, N, ldp, strideP);
  CHECK_LAUNCH();
}
int launch_unpack_meas(const double* rawH, long strideRaw, int ldraw, const int* only_if, MeasBuffers mb, int M,
                       int Mp, int N, int Np, int batch, hipStream_t s) {
  dim3 grid((unsigned)(((long)Mp * Np + 255) / 256), batch);
  hipLaunchKernelGGL(unpack_meas_kernel, grid, dim3(256), 0, s, rawH, strideRaw, ldraw, only_if, mb, M, Mp, N, Np);
  CHECK_LAUNCH();
}
// H^T [Np x Mp, ldht] from the dense H [Mp x Np, ldh] of every filter: the transposed copy is optional for the G-level
// producers (capi.hip: skip_HT) and rebuilt here when a consumer turns up after all. 32 x 32 tiles through LDS so that
// both sides move 256-byte runs.
__global__ __launch_bounds__(256) void transpose_H_kernel(const double* __restrict__ Hall, long strideH, int ldh,
                                                         double* __restrict__ HTall, long strideHT, int ldht, int Mp, int Np) {
  __shared__ double t[32][33];
  const double* H = Hall + (long)blockIdx.z * strideH;
  double* HT = HTall + (long)blockIdx.z * strideHT;
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
    t[j][tx] = (m0 + tx < Mp && n0 + j < Np) ? H[(m0 + tx) + (long)(n0 + j) * ldh] : 0.0;
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (n0 + tx < Np && m0 + j < Mp) HT[(n0 + tx) + (long)(m0 + j) * ldht] = t[tx][j];
}
int launch_transpose_H(const double* H, long strideH, int ldh, double* HT, long strideHT, int ldht, int Mp, int Np, int batch,
                       hipStream_t s) {
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(transpose_H_kernel, dim3((Mp + 31) / 32, (Np + 31) / 32, batch), dim3(256), 0, s, H, strideH, ldh, HT, strideHT, ldht, Mp, Np);
  CHECK_LAUNCH();
}
int launch_p_zero_rc(double* P, int ldp, int Np, int off, int len, hipStream_t s) {
  hipLaunchKernelGGL(p_zero_rc_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, P, ldp, Np, off, len);
  CHECK_LAUNCH();
}
int launch_p_copy_rc(double* P, int ldp, int Np, int dst, int src, int len, hipStream_t s) {
  hipLaunchKernelGGL(p_copy_rc_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, P, ldp, Np, dst, src, len, 0);
  hipLaunchKernelGGL(p_copy_rc_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, P, ldp, Np, dst, src, len, 1);
  CHECK_LAUNCH();
}
int launch_p_diag(const double* P, int ldp, int N, double* out, hipStream_t s) {
  hipLaunchKernelGGL(p_diag_kernel, dim3((N + 255) / 256), dim3(256), 0, s, P, ldp, N, out);
  CHECK_LAUNCH();
}
int launch_gate_dense(const GateDenseArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(gate_dense_kernel, dim3(a.batch), dim3(256), (a.F + 1) * sizeof(double), s, a);
  CHECK_LAUNCH();
}
int launch_jac_instate(const SceneBuffers& sb, const xivo_layout& lay, const xivo_cam& cam, int batch,
                       hipStream_t s) {
  const int tot = batch * sb.F;
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(jac_instate_kernel, dim3((tot + 127) / 128), dim3(128), 0, s, sb, lay, cam, batch);
  CHECK_LAUNCH();
}
int launch_gate_sparse(const GateArgs& a, hipStream_t s) {
  int nt = a.batch < 256 ? 1024 : 256;
  // distances + threshold | slot indices | per-wave scratch of feature_chi2_lds (64 KB without an opt-in: fewer waves if F is large)
  const size_t scr = a.sb.Jc ? WIDE_SCR : 484, pss = a.sb.Jc ? WIDE_PSS : 0;
  auto lds_of = [&](int t) { return ((size_t)(a.sb.F + 1) + (2 * a.sb.F + 1) / 2 + pss + (size_t)(t / 64) * scr) * sizeof(double); };
  while (nt > 64 && lds_of(nt) > 65536) nt /= 2;
  const size_t lds = lds_of(nt);
  hipLaunchKernelGGL(gate_sparse_kernel, dim3(a.batch), dim3(nt), lds, s, a);
  CHECK_LAUNCH();
}
int launch_stack(const StackArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(stack_kernel, dim3(a.batch), dim3(256), 0, s, a);
  CHECK_LAUNCH();
}
int launch_subfilter(xivo_subfilter_feat* feats, int n, const xivo_pose_in* poses, const xivo_group_in* groups,
                     int n_groups, xivo_cam cam, xivo_subfilter_opts o, int batch, hipStream_t s, const xivo_calib_in* calib,
                     int cam_dim, int invdepth) {
  const int tot = batch * n;
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(subfilter_kernel, dim3((tot + 127) / 128), dim3(128), 0, s, feats, n, poses, groups, n_groups, cam,
                     o, batch, calib, cam_dim, invdepth);
  CHECK_LAUNCH();
}
int launch_givens(const GivensArgs& a, hipStream_t s) {
  if (a.batch <= 0) return 0;
  hipLaunchKernelGGL(givens_kernel, dim3(a.batch), dim3(64), 0, s, a);
  CHECK_LAUNCH();
}
int launch_set_pixels(xivo_feat_in* feats, int Fmax, int F, const double* xp, int nb, hipStream_t s) {
  const int n = nb * F;
  hipLaunchKernelGGL(set_pixels_kernel, dim3((n + 255) / 256), dim3(256), 0, s, feats, Fmax, F, xp, n);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
int launch_edit_batch(const EditArgs& a, int n_wg, hipStream_t s) {
  hipLaunchKernelGGL(edit_batch_kernel, dim3(n_wg), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
int launch_oos_compress(const OosCompressArgs& a, int rows_max, hipStream_t s) {
  const int ncols = 6 + 6 * a.lay.n_groups + 1;      // candidates + the residual column
  if (ncols <= 64 && rows_max <= 144) hipLaunchKernelGGL((oos_compress_kernel<36, 1>), dim3(a.batch), dim3(256), 0, s, a);
  else if (ncols <= 64 && rows_max <= 256) hipLaunchKernelGGL((oos_compress_kernel<64, 1>), dim3(a.batch), dim3(256), 0, s, a);
  else if (ncols <= 128 && rows_max <= 144) hipLaunchKernelGGL((oos_compress_kernel<36, 2>), dim3(a.batch), dim3(256), 0, s, a);
  else return -1;                                     // not built for this size: the caller leaves the rows as they are
  CHECK_LAUNCH();
}
int launch_ransac_select(const RansacArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(ransac_select_kernel, dim3(a.batch), dim3(64), 0, s, a);
  CHECK_LAUNCH();
}
int launch_ransac_zero(const RansacArgs& a, double* P, hipStream_t s) {
  hipLaunchKernelGGL(ransac_zero_kernel, dim3(a.batch), dim3(256), 0, s, a, P);
  CHECK_LAUNCH();
}
int launch_ransac_rescue_dist(const RansacArgs& a, const double* dist, int ld, hipStream_t s) {
  hipLaunchKernelGGL(ransac_rescue_dist_kernel, dim3(a.batch), dim3(256), 0, s, a, dist, ld);
  CHECK_LAUNCH();
}
int launch_ransac_rescue(const RansacArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(ransac_rescue_kernel, dim3(a.batch), dim3(256), 0, s, a);
  CHECK_LAUNCH();
}
int launch_absorb_error(const AbsorbArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(absorb_error_kernel, dim3(a.batch), dim3(256), 0, s, a);
  CHECK_LAUNCH();
}
int launch_lc_rows(const LcArgs& a, hipStream_t s) {
  const int tot = a.batch * a.n;
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(lc_rows_kernel, dim3((tot + 63) / 64), dim3(64), 0, s, a);
  CHECK_LAUNCH();
}
int launch_oos(const OosArgs& a, hipStream_t s) {
  if (a.n_oos <= 0) return 0;
  hipLaunchKernelGGL(oos_kernel, dim3(a.n_oos, a.batch), dim3(64), 0, s, a);
  CHECK_LAUNCH();
}
int launch_propagate_cov(double* P, long strideP, int ldp, int N, int Np, int nm, const double* Phi,
                         const double* Pmm, int b0, int nb, hipStream_t s) {
  (void)Np;
  if (nm > 40) return (int)hipErrorInvalidValue;
  if (nm == 23) {
    hipLaunchKernelGGL(propagate_cov_fixed_kernel<23>, dim3(nb), dim3(256), 0, s, P, strideP, ldp, N, Phi, Pmm, b0);
    CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(propagate_cov_kernel, dim3(nb), dim3(256), nm * nm * sizeof(double), s, P, strideP, ldp, N,
                     nm, Phi, Pmm, b0);
  CHECK_LAUNCH();
}
template <int NS>
static int launch_propagate_state_wave(const PropStateArgs& a, hipStream_t s) {
  // LDS: P0, S1, F P0 / P0 F^T scratch, Q / GQ / GQG^T supports, nominal state, F9, two transition buffers, per stage 36
  // Jacobian entries + FK: RK4 28 KB (5 filters per CU), Dormand-Prince 34 KB (4)
  const size_t lds = (size_t)(2 * 529 + 3 * 207 + 3 * 144 + 2 + 36 + NS * (36 + 207)) * sizeof(double);
  hipLaunchKernelGGL(propagate_state_wave_kernel<NS>, dim3(a.batch), dim3(64), lds, s, a);
  CHECK_LAUNCH();
}
size_t propagate_calib_lds(int nm, int ns) {
  const size_t NN = (size_t)nm * nm, NF = 9 * (size_t)nm;
  return ((2 + ns) * NN + (6 + ns) * NF + 432 + 64 + 24 + (size_t)ns * 60) * sizeof(double);
}
template <int NS>
static int launch_propagate_state_calib_ns(const PropStateArgs& a, hipStream_t s) {
  const size_t lds = propagate_calib_lds(a.nm, NS);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&propagate_state_calib_kernel<NS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(propagate_state_calib_kernel<NS>, dim3(a.batch), dim3(256), lds, s, a);
  return (int)hipGetLastError();
}
int launch_propagate_state_calib(const PropStateArgs& a, hipStream_t s) {
  if (a.batch <= 0) return 0;
  if (a.nm < 23 || a.nm > 40 || !a.calib) return (int)hipErrorInvalidValue;
  return a.method ? launch_propagate_state_calib_ns<7>(a, s) : launch_propagate_state_calib_ns<4>(a, s);
}

int launch_propagate_state(const PropStateArgs& a, hipStream_t s) {
  if (a.batch <= 0) return 0;
  return a.method ? launch_propagate_state_wave<7>(a, s) : launch_propagate_state_wave<4>(a, s);
}
int launch_mfma_peak(double* sink, int iters, int blocks, hipStream_t s) {
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, s, sink, iters);
  CHECK_LAUNCH();
}

}  // namespace xivo_hip
