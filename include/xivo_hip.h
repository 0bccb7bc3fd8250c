/* xivo_hip.h - C ABI of the MI355X-native EKF measurement-update path for XIVO.
 *
 * Drop-in boundary for the reference's private hot path (the reference has no
 * FFI of its own - SURVEY.md section 8b): each entry point below names the
 * reference member function / lines it replaces, relative to /root/reference.
 *
 * Conventions
 *  - plain C, no exceptions, no aborts: every call returns XIVO_HIP_OK (0) or a
 *    negative status (the reference LOG(FATAL)s / throws instead,
 *    src/estimator.cpp:121,587,821,844 - the C++ adapter in
 *    xivo_amd/host/estimator_hip.h converts a non-zero status back to that).
 *  - all matrices are column-major double (common/alias.h:11, Eigen default),
 *    with an explicit leading dimension where the caller owns the buffer.
 *  - batch-first: a context holds `batch_max` independent filters (the
 *    reference is one singleton filter per process, src/estimator.cpp:26);
 *    "b0, nb" = first filter and number of filters a call touches.
 *  - the covariance P of every filter is device resident; host edits of P go
 *    through the xivo_hip_p_* calls (mirrors of the host edits listed in
 *    SURVEY.md a17) or through upload/download.
 *  - a context is single threaded (like the reference's estimator); different
 *    contexts may be driven from different host threads / processes (one per GPU).
 *  - host pointers are borrowed for the duration of the call only.
 */
#ifndef XIVO_HIP_H_
#define XIVO_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xivo_hip_ctx xivo_hip_ctx;

enum {
  XIVO_HIP_OK = 0,
  XIVO_HIP_ERR_INVALID = -1,     /* bad argument / size                         */
  XIVO_HIP_ERR_HIP = -2,         /* a HIP runtime call failed                   */
  XIVO_HIP_ERR_NOT_SPD = -3,     /* S = HPH^T + R not positive definite         */
  XIVO_HIP_ERR_NOMEM = -4,
  XIVO_HIP_ERR_UNSUPPORTED = -5  /* size outside what the kernels are built for */
};

/* flags for xivo_hip_create / stacking */
enum {
  XIVO_HIP_FLAG_NONE = 0u,
  /* Feature::FillJacobianBlock writes the group-translation block over the
   * group-rotation block (src/feature.cpp:675-676). Default = reproduce it;
   * this flag gives the evidently intended full row (as src/update.cpp:326). */
  XIVO_HIP_FLAG_FIX_GROUP_BLOCK = 1u,
  /* record HIP events around every kernel launch (per-stage timing) */
  XIVO_HIP_FLAG_PROFILE = 2u,
  /* By default the update exploits the row sparsity of H: when every row pair of every filter of the
   * call has at most 16 columns shared by most pairs + 12 private non-zero columns (true for the stacked
   * in-state Jacobians of src/update.cpp:129-138: 21 per pair), H P, S and the H-products of the
   * covariance stage skip the structural zeros (exact: the skipped terms are 0 * x), and the covariance stage
   * evaluates the Joseph expression in its whitened form, P+ = P - (W - D)^T (W + D). An H that does not
   * compress (dense rows, arbitrary input) keeps the same evaluation on dense products for H P and S. This flag forces the
   * AS-CODED sequence of src/estimator.cpp:1259-1287 on dense products for any H: A = K H - I, T = A P,
   * P+ = T A^T + K R K^T (the pure-GEMM variant; 3 x the flops). */
  XIVO_HIP_FLAG_DENSE_H = 64u,
  /* Symmetric ("square-root") form of the gain and covariance: S = L L^T, W = L^-1 (H P) by forward substitution only,
   *   dx = W^T (L^-1 inn),   P+ = P - W^T W
   * - what the Joseph expression of src/estimator.cpp:1276-1287 evaluates to for the optimal gain K = P H^T S^-1 (its
   * correction term (K S - P H^T) K^T vanishes identically), without the backward substitution, the gain residual and
   * the second N x N x M product: about 80 % of the device time of the default. The result is symmetric by
   * construction and its rounding error scales with cond(L) = sqrt(cond(S)). Opt-in: the reference codes the Joseph
   * form and the default reproduces that expression; parity of this mode against the reference is tested to the same
   * tolerances (1e-6 on P, 1e-8 on dx), including an ill-conditioned S. */
  XIVO_HIP_FLAG_SYMMETRIC_FORM = 256u,
  /* Sparse-H pipeline: by default the covariance update is the whitened Joseph expression on the gain still in the solve
   * kernel's registers (or, for wider shapes, on the whitened outputs of the solve) - T and G never reach memory. This flag
   * selects the re-associated tail from stand-alone kernels instead: T = K(HP) - P, G = T H^T + K R, P+ = G K^T - T
   * (= T (KH - I)^T + K R K^T, exact for any gain like the Joseph form it is) - 0.8 x the speed, 5 x closer to the as-coded
   * fp64 result (both lose digits in proportion to cond(S) and meet the 1e-6 / 1e-8 tolerances by orders of magnitude). */
  XIVO_HIP_FLAG_STANDALONE_TAIL = 512u,
  /* Opt-in (round 4; BASELINE config 4 "fp32 MFMA with stated tolerance"), shapes whose covariance product runs outside the
   * solve kernel (N > 256 or M > 176): the whitened outputs V^T = (W - D)^T and Y^T = (W + D)^T leave the fp64 solve as
   * FLOAT and P+ = P - V^T Y runs on v_mfma_f32_16x16x4_f32 (fp32 accumulation over M, subtracted from P in fp64): half
   * the operand bytes of a product that is HBM-bound, twice the matrix rate. Everything else (S, the factorisation, both
   * substitutions, dx) stays fp64: dx is unchanged, P+ carries the rounding of the float operands, <= 5e-5 relative
   * Frobenius (measured ~1e-7 .. 3e-6 per update). Shapes the in-solve update holds are not affected, at any batch size.
   * Over a CHAIN of updates on the resident covariance (tests/test_variants_gpu.py::test_fp32_whitened_chain, 25 updates):
   * P stays within 5e-5 of the all-fp64 chain (measured 2.6e-5); dx of each update is bit-identical to the fp64 path given
   * the same prior, but against the fp64 chain a later dx deviates by up to 5e-2 relative (measured 2.3e-2; <= 0.05 posterior
   * standard deviations): P - V^T Y cancels in the directions earlier measurements shrank, and floats resolve 6e-8 |P| there.
   * Use it where that is acceptable; the default stays all fp64. */
  XIVO_HIP_FLAG_FP32_WHITENED = 16384u,
  /* By default a filter whose innovation covariance the un-pivoted Cholesky cannot factor (S indefinite / not positive
   * definite) is updated the reference's way after all: Eigen's diagonally pivoted L D L^T solve (src/estimator.cpp:1266)
   * and the as-coded Joseph form, on that filter only (ldlt_fallback.hip); xivo_hip_get_ldlt_used tells which filters took
   * that route and their status reads 0. With this flag such a filter keeps its prior covariance bit for bit, absorbs
   * nothing, and xivo_hip_get_status reports it (the behaviour of rounds 1-2). */
  XIVO_HIP_FLAG_NO_LDLT_FALLBACK = 4096u,
  /* By default an update of at most 64 filters (one estimator is the reference's own use) takes the latency route of the
   * default pipeline: the solve on 128-column workgroups of the streamed kernel, the covariance product P - V^T Y on
   * 64 x 64 tiles - the same whitened Joseph evaluation spread over tens of CUs instead of one CU per filter (B = 1,
   * N = 250, M = 160: solve + product 0.07 instead of 0.13 ms). With this flag every batch size runs the kernels sized for
   * thousands of filters (one workgroup per filter, whole update inside the solve kernel). */
  XIVO_HIP_FLAG_THROUGHPUT_ROUTE = 8192u,
  /* The reference's USE_INVDEPTH build (src/CMakeLists.txt:10): a feature's local state is (X/Z, Y/Z, 1/Z) instead of
   * (X/Z, Y/Z, log Z) - Feature::Xc goes through unproject_invz (src/feature.cpp:98-105, common/project.h:31-56) and
   * Feature::z is 1 / x(2) (:120-126). Every kernel that unprojects a feature (in-state Jacobians, depth sub-filter and its
   * candidate depth test, loop-closure rows) follows the flag; the Jacobian block d/dx changes accordingly. */
  XIVO_HIP_FLAG_INVDEPTH = 32768u,
  /* Round 6: shapes one workgroup holds end to end (M <= 64 with N <= 256 - the TUM-VI build; M <= 112 with N <= 192 -
   * BASELINE config 2) run the WHOLE update - P H^T, S, the MH gate, the factorisation, both substitutions, dx and the
   * covariance product - in one kernel per filter (fused_update.hip): nothing but P, P+ and the compressed rows crosses HBM.
   * Same algebra as the multi-kernel pipeline (whitened Joseph evaluation, same gate expressions). This flag keeps such a
   * shape on the multi-kernel pipeline (A/B, and the route-against-route parity tests). */
  XIVO_HIP_FLAG_MULTI_KERNEL = 65536u
};

/* camera models implemented on device (common/camera_pinhole.h,
 * common/camera_equidist.h, common/camera_radtan.h, common/camera_atan.h) */
enum { XIVO_CAM_PINHOLE = 0, XIVO_CAM_ATAN = 1, XIVO_CAM_RADTAN = 2, XIVO_CAM_EQUI = 3 };

/* Error-state layout (src/core.h:40-105). N is a run-time value here
 * (kFullSize is a compile-time constant in the reference). */
typedef struct {
  int N;              /* kFullSize                                           */
  int group_begin;    /* kGroupBegin  (23 in the default build)              */
  int n_groups;       /* kMaxGroup                                           */
  int feature_begin;  /* kFeatureBegin = group_begin + 6*n_groups            */
  int n_features;     /* kMaxFeature                                         */
} xivo_layout;

typedef struct {
  int model;          /* XIVO_CAM_*                                          */
  int rows, cols;
  double fx, fy, cx, cy;
  double d[5];        /* EQUI: k0..k3 ; RADTAN: p1,p2,k1,k2,k3(order of the
                         reference ctor) ; ATAN: w                            */
} xivo_cam;

/* Nominal poses one filter needs for Feature::ComputeJacobian
 * (src/update.cpp:24-32 passes X_.Rsb, X_.Tsb, X_.Rbc, X_.Tbc). 3x3 matrices
 * are column-major. */
typedef struct {
  double Rsb[9], Tsb[3];
  double Rbc[9], Tbc[3];
  /* rest of the nominal motion state (src/core.h:117-130): not read by the Jacobians, but
   * retracted by xivo_hip_absorb_error so the whole State can stay device resident */
  double Vsb[3], bg[3], ba[3];
  double Rsg[9];
} xivo_pose_in;

/* One group anchor (src/group.h:41-107): pose + state slot `sind`. */
typedef struct {
  double Rsb[9], Tsb[3];
} xivo_group_in;

/* One in-state feature (src/feature.h:74-232). */
typedef struct {
  double x[3];        /* (X/Z, Y/Z, log Z) in the reference camera frame, feature.h:258-262 */
  double xp[2];       /* last tracked pixel, Feature::back()                 */
  int ref_sind;       /* ref_->sind(): slot of the reference group           */
  int sind;           /* feature slot; -1 = absent entry: contributes no rows, is never an inlier
                         (filters of one batch may hold different numbers of features)      */
} xivo_feat_in;

/* One out-of-state (MSCKF) feature with k observations from in-state groups
 * (src/oos.cpp:8-89). */
#define XIVO_OOS_MAX_OBS 16
typedef struct {
  double Xs[3];                       /* cache_.Xs, src/oos.cpp:17           */
  int n_obs;
  int group_sind[XIVO_OOS_MAX_OBS];   /* obs.g->sind()                       */
  double xp[XIVO_OOS_MAX_OBS][2];     /* obs.xp                              */
} xivo_oos_in;

/* ---- lifetime -------------------------------------------------------- */
int xivo_hip_create(xivo_hip_ctx** out, int device, int N, int M_max, int batch_max, unsigned flags);
void xivo_hip_destroy(xivo_hip_ctx* ctx);
const char* xivo_hip_strerror(int status);
int xivo_hip_sync(xivo_hip_ctx* ctx);
/* number of visible HIP devices (0 if none / on error) */
int xivo_hip_device_count(void);
/* NUMA node of the host cores / memory closest to `device` (from the sysfs entry of its PCI function), -1 if unknown.
 * The reference runs one estimator per process (src/estimator.cpp:26); with one process per GPU the launcher uses this to
 * keep each rank's host side (hand-over staging, BatchEstimator's OpenMP team) on its GPU's socket. */
int xivo_hip_device_numa_node(int device);
int xivo_hip_set_flags(xivo_hip_ctx* ctx, unsigned flags);

/* ---- covariance residency (Estimator::P_, src/estimator.h:423; a17) --- */
/* P: nb column-major N x N matrices, `stride` elements apart, leading dimension ld. The LOWER triangle of each uploaded
 * matrix is authoritative: the device state is its exact mirror (every pipeline treats P as symmetric; the reference never
 * re-symmetrises P_, src/estimator.cpp:1280-1287, so its triangles differ by rounding - a symmetric matrix round-trips bit
 * for bit, a rounding-level asymmetry stays inside the parity tolerances, tests/test_robustness_gpu.py). */
int xivo_hip_upload_P(xivo_hip_ctx* ctx, int b0, int nb, const double* P, long stride, int ld);
int xivo_hip_download_P(xivo_hip_ctx* ctx, int b0, int nb, double* P, long stride, int ld);
/* BackupState / RestoreState P part (src/estimator.cpp:1413-1414,1434-1435) */
int xivo_hip_snapshot_P(xivo_hip_ctx* ctx);
int xivo_hip_restore_P(xivo_hip_ctx* ctx);
/* P.block(off,0,len,N)=0; P.block(0,off,N,len)=0 (src/estimator.cpp:757-759,781-783,1476-1477; src/update.cpp:299-315) */
int xivo_hip_p_zero_rc(xivo_hip_ctx* ctx, int b, int off, int len);
/* copy rows+cols [src,src+len) onto [dst,dst+len) (AddGroupToState, src/estimator.cpp:808-816) */
int xivo_hip_p_copy_rc(xivo_hip_ctx* ctx, int b, int dst, int src, int len);
/* P.block<3,3>(off,off) = P3 (Feature::FillCovarianceBlock, src/feature.cpp:753-760) */
int xivo_hip_p_set_block3(xivo_hip_ctx* ctx, int b, int off, const double* P3);
/* diag(P) (FindNewRefGroup reads it, src/estimator.cpp:1394-1407) */
int xivo_hip_p_diag(xivo_hip_ctx* ctx, int b, double* diag_out);

/* ---- S-level: dense H / inn / diagR given (Estimator::H_, inn_, diagR_) -- */
/* stage measurements of filters [b0,b0+nb): H is M x N (ldh), inn and diagR
 * have M entries. M may differ between calls, M <= M_max. */
int xivo_hip_set_measurements(xivo_hip_ctx* ctx, int b0, int nb, int M,
                              const double* H, long strideH, int ldh,
                              const double* inn, long strideInn,
                              const double* diagR, long strideR);
/* The same hand-over for measurements that are already in device memory (a device-side producer, or a caller that
 * keeps its Eigen buffers in pinned / managed memory): dH is M x N column-major per filter (leading dimension ldh,
 * filters strideH elements apart), dInn / dR have M entries. This is the per-frame device work of the S-level
 * boundary - H_ changes with every camera frame (src/update.cpp:129-138): one batched launch builds the row-pair
 * compressed rows (each element of H read once); padded dense copies are materialised only for filters whose rows
 * do not fit the compressed form. The device buffers are read during the call only. */
int xivo_hip_set_measurements_device(xivo_hip_ctx* ctx, int b0, int nb, int M,
                                     const double* dH, long strideH, int ldh,
                                     const double* dInn, long strideInn,
                                     const double* dR, long strideR);
/* Estimator::UpdateJosephForm (src/estimator.cpp:1257-1288) for filters [0,B):
 * S = HPH^T + R, K^T = S^-1 HP, dx = K inn, P <- (KH-I)P(KH-I)^T + K R K^T,
 * on the resident P with the staged measurements. Asynchronous on the
 * context's stream. */
int xivo_hip_update_joseph(xivo_hip_ctx* ctx, int B);
/* err_ (dx) of filters [b0,b0+nb) after the update (before AbsorbError) */
int xivo_hip_get_err(xivo_hip_ctx* ctx, int b0, int nb, double* err, long stride);
/* per-filter factorisation status of the last update (0 = ok) ; returns
 * XIVO_HIP_ERR_NOT_SPD if any is non-zero */
int xivo_hip_get_status(xivo_hip_ctx* ctx, int b0, int nb, int* status);
/* used[i] = 1 if the last update of filter b0 + i ran the pivoted L D L^T fallback (S was not positive definite; the
 * reference's S.ldlt().solve, src/estimator.cpp:1266, handles that case silently), 0 otherwise */
int xivo_hip_get_ldlt_used(xivo_hip_ctx* ctx, int b0, int nb, int* used);
/* Estimator::UpdateJosephForm() AS THE REFERENCE CALLS IT (src/estimator.cpp:1257-1288; callers src/update.cpp:141 and
 * :332): one filter whose members live in pageable host memory - P_ (N x N, in-out), H_ (M x N), inn_, diagR_ (M) in,
 * err_ (N) out - in ONE call with ONE host synchronisation. Replaces the sequence upload_P / set_measurements /
 * update_joseph / get_status / get_err / download_P (four synchronisations, three staged copies) for the drop-in binding
 * of INTEGRATION.md section 3. The dense H_ is scanned once on the host while it is staged and crosses PCIe as row-pair
 * compressed rows (the same rows the batched hand-over builds on the device, bit for bit); an H_ without XIVO's row
 * structure takes the general entry points inside this call (same results). P_ crosses through the context's page-locked
 * block (one host copy each way; the boundary kernels address the block directly). Returns XIVO_HIP_ERR_NOT_SPD exactly when
 * xivo_hip_get_status would: with XIVO_HIP_FLAG_NO_LDLT_FALLBACK for any S the Cholesky cannot factor, and without it when the
 * pivoted L D L^T fallback met non-finite arithmetic (a NaN / Inf in P_, H_ or diagR_); P is then the prior, err_out zero.
 * The context's staged row count becomes M.
 *   mode: XIVO_HIP_HOST_P_RESIDENT  the device copy of P (filter b) is current - P_ is not uploaded (no host edit since
 *                                   the last upload / download; P may be NULL when KEEP_P is set too)
 *         XIVO_HIP_HOST_KEEP_P      P+ stays on the device only (resident mode through the same call; read it back later
 *                                   with xivo_hip_download_P) */
#define XIVO_HIP_HOST_P_RESIDENT 1u
#define XIVO_HIP_HOST_KEEP_P 2u
int xivo_hip_update_joseph_host(xivo_hip_ctx* ctx, int b, int M, const double* H, int ldh, const double* inn,
                                const double* diagR, double* P, int ldp, double* err_out, unsigned mode);
/* Test hook (needs neither a device nor a context): the host-side row-pair compression xivo_hip_update_joseph_host applies
 * to H_ while staging it - idx [pairs_clear][28] / val [pairs_clear][28][2] in the layout of the batched hand-over
 * (xivo_hip_set_measurements_device), *nc common slots, *pw private slots; returns 1 when the rows do not fit, else 0. */
/* Host-only check of the tables that say which wave of the one-kernel update (csrc/fused_update.hip) forms which tile of
 * P+ at 10 / 13 column blocks: every block pair exactly once, the counts the kernel's dispatch assumes, per_simd[4] (optional)
 * the tiles per SIMD. 0 = consistent, -1 = no table for that size. No GPU needed. */
int xivo_hip_selftest_fused_tiles(int column_blocks, int* per_simd);
int xivo_hip_selftest_host_compress(const double* H, int ldh, int M, int N, int pairs_clear, int* idx, double* val, int* nc, int* pw);
/* Estimator::MHGating numeric core on dense rows (src/update.cpp:60-96):
 * rows 2f,2f+1 of the staged H are feature f's J. Writes the inlier mask and
 * Mahalanobis distances; rejected rows are then neutralised in the staged
 * measurements so that a following xivo_hip_update_joseph equals the
 * reference's FilterUpdate over the inliers only. */
int xivo_hip_mh_gate_dense(xivo_hip_ctx* ctx, int B, int F, double R, double mh_thresh,
                           double mh_mult, int min_inliers,
                           unsigned char* inlier_mask_out, double* mh_dist_out);

/* FilterUpdate on dense candidate rows in one pass: HP = H P once, Mahalanobis gating of
 * features 0..F-1 from (HP)_f H_f^T + R (skipped when F <= min_inliers, src/manager.cpp:635),
 * rejected rows neutralised, then UpdateJosephForm. xivo_hip_get_gate returns the mask /
 * distances of that pass ([B x F]). */
int xivo_hip_update_dense_gated(xivo_hip_ctx* ctx, int B, int F, double R, double mh_thresh,
                                double mh_mult, int min_inliers);
int xivo_hip_get_gate(xivo_hip_ctx* ctx, int B, int F, unsigned char* inlier_mask_out, double* mh_dist_out);

/* ---- G-level: features + poses given, Jacobians built on device -------- */
int xivo_hip_set_layout(xivo_hip_ctx* ctx, const xivo_layout* layout, const xivo_cam* cam);
int xivo_hip_set_scene(xivo_hip_ctx* ctx, int b0, int nb, int F,
                       const xivo_pose_in* poses, const xivo_group_in* groups /* nb x n_groups */,
                       const xivo_feat_in* feats /* nb x F */);
/* Estimator::ComputeInstateJacobians (src/update.cpp:24-32) =
 * Feature::ComputeJacobian x F (src/feature.cpp:542-656) */
int xivo_hip_jacobians_instate(xivo_hip_ctx* ctx, int B);
/* compact per-feature result: J blocks 2x21 ([Wsb Tsb Wbc Tbc Wg Tg x], row-major 2 x 21) + inn (2) */
int xivo_hip_get_jacobians(xivo_hip_ctx* ctx, int b0, int nb, double* J2x21, double* inn2);
/* ---- online-calibration builds (measurement side) ---------------------------------------------------------------
 * The reference's USE_ONLINE_TEMPORAL_CALIB / USE_ONLINE_IMU_CALIB / USE_ONLINE_CAMERA_CALIB builds (src/CMakeLists.txt:13-15)
 * put a camera-IMU time offset td, the gyro calibration Cg (9) / accel calibration Ca (6) and up to 9 camera intrinsics
 * into the state (src/core.h:49-83) and give Feature::ComputeJacobian four more blocks (src/feature.cpp:592-609, :611-618,
 * :632-651): d/dtd (2 x 1), d/dCg (2 x 9), d/dbg (2 x 3, at Index::bg) and d/d(intrinsics) (2 x Camera::dim()), which
 * Feature::FillJacobianBlock stacks as well (:664-670, :679-683). xivo_hip_set_calib switches those blocks on for the
 * context: xivo_hip_jacobians_instate then also fills them (xivo_hip_get_jacobians_calib), xivo_hip_stack /
 * xivo_hip_filter_update stack them. Such a row pair has up to 34 columns that every feature shares - more than the 16
 * common slots of the row-pair compressed form - but all of them lie in the leading 48 state columns: the stacking is the
 * default build's compressed rows + a dense [M x 48] block of the calibration columns, and the update takes the sparse
 * pipeline with two skinny products on top (round 5; xivo_hip_last_path 1). MH gating uses the whole row as the reference's
 * f->J() does (43 columns in the compact gate). Where the calibration columns do not fit the leading 48 (cam_begin + 9 > 48),
 * under XIVO_HIP_FLAG_DENSE_H / _SYMMETRIC_FORM / _STANDALONE_TAIL, and whenever dense rows are needed after all (OOS rows appended, xivo_hip_update_dense_gated,
 * xivo_hip_get_H - which therefore sends the NEXT update of that stacking down the dense pipeline) the rows are (re-)stacked
 * as dense rows and gate / update through the dense pipeline (round 4). Same results within the stated tolerances either way.
 * Slots as the reference's Index enum / kCameraBegin would number them
 * (the caller's xivo_layout already counts them in N, group_begin, feature_begin); -1 / 0 = that block is not in the build.
 * Motion side of those builds: xivo_hip_propagate_calib (below, next to xivo_hip_propagate) integrates the
 * kMotionSize = 24 / 38 / 39-dimensional motion block with the Cg / Ca columns of ComputeMotionJacobianAt
 * (src/estimator.cpp:626-638, :674-688) and imu_.Cg() / imu_.Ca() in ComposeMotion (:603-604); xivo_hip_absorb_error
 * then also retracts td, Cg, Ca and the intrinsics (src/core.h:150-152, src/estimator.cpp:879-890, src/imu.cpp:7-21,
 * common/camera_autocalib.h) of the resident per-filter calibration state below. */
typedef struct {
  int td;         /* Index::td, or -1 (no USE_ONLINE_TEMPORAL_CALIB: then neither the td, nor the Cg, nor the bg block exists) */
  int Cg;         /* Index::Cg (9 columns; Ca's 6 follow at Cg + 9), or -1 (no USE_ONLINE_IMU_CALIB); the measurement-side
                     Cg / bg blocks need td >= 0 (they are nested in the temporal block, src/feature.cpp:592-609)              */
  int cam_begin;  /* kCameraBegin                                                                                              */
  int cam_dim;    /* Camera::dim(): 4 pinhole (fx fy cx cy), 5 atan (+ w), 9 radtan (+ p1 p2 k1 k2 k3), 8 equidistant
                     (+ k0..k3); 0 = no USE_ONLINE_CAMERA_CALIB                                                               */
} xivo_calib_layout;
typedef struct {   /* per filter: what Estimator::ComputeInstateJacobians hands down (src/update.cpp:27-28) + the rest of the
                      calibration state that AbsorbError retracts */
  double gyro[3];  /* last_gyro_ (raw measurement)        */
  double Cg[9];    /* imu_.Cg(), column-major             */
  double td;       /* X_.td                               */
  double Ca[9];    /* imu_.Ca(), column-major (upper triangular, src/imu.cpp:24); identity when the build has no Ca */
  double intr[9];  /* camera intrinsics in the order of the state slots: fx fy cx cy, then xivo_cam.d[0..4]. With
                      cam_dim > 0 every kernel that projects (Jacobians, OOS rows, depth sub-filter) takes the filter's
                      intrinsics from here - the context's xivo_cam only names the model; cam_dim = 0: not read */
} xivo_calib_in;
/* layout == NULL switches the calibration blocks off again (default build) */
int xivo_hip_set_calib(xivo_hip_ctx* ctx, const xivo_calib_layout* layout);
int xivo_hip_set_calib_state(xivo_hip_ctx* ctx, int b0, int nb, const xivo_calib_in* calib /* nb */);
int xivo_hip_get_calib_state(xivo_hip_ctx* ctx, int b0, int nb, xivo_calib_in* calib_out /* nb */);
/* only last_gyro_ of every filter (what changes from frame to frame; td / Cg / Ca / intr are state and stay as AbsorbError
 * left them): gyro3 = nb x 3 doubles */
int xivo_hip_set_calib_gyro(xivo_hip_ctx* ctx, int b0, int nb, const double* gyro3);
/* the calibration blocks of the last xivo_hip_jacobians_instate, per feature 2 x 22 row-major:
 * [ td (1) | Cg (9) | bg (3) | intrinsics (9, the first cam_dim in use) ]; blocks that are switched off read 0 */
int xivo_hip_get_jacobians_calib(xivo_hip_ctx* ctx, int b0, int nb, double* Jc2x22);
/* Estimator::MHGating (src/update.cpp:50-116) on the compact Jacobians (full
 * J row, as the reference gates with f->J()). */
int xivo_hip_mh_gate(xivo_hip_ctx* ctx, int B, double R, double mh_thresh, double mh_mult,
                     int min_inliers, unsigned char* inlier_mask_out, double* mh_dist_out);
/* Estimator::FilterUpdate stacking (src/update.cpp:129-138) through
 * Feature::FillJacobianBlock (src/feature.cpp:658-684): in-state inlier rows
 * first, then any OOS rows appended by xivo_hip_oos_project. */
int xivo_hip_stack(xivo_hip_ctx* ctx, int B, double R);
/* Feature::ComputeOOSJacobian (src/oos.cpp:8-89) + SlowGivens
 * (src/helpers.cpp:13-23): per feature (2k-3) projected rows appended after
 * the in-state rows with diagR = Roos. rows_out[b] = total OOS rows. feats == NULL projects the list of the
 * previous call again (it stays resident; same nb and n_oos) - for a caller that re-linearises without new tracks.
 * Camera-calibration builds (cam_dim > 0): the observations are projected with the filter's own intrinsics, and - as the
 * reference codes ComputeOOSJacobianInternal (src/oos.cpp:39-89 writes the group / Wbc / Tbc blocks only) - the rows carry NO
 * intrinsics block; of the row builders in that file only ComputeLCJacobian has one (:125-142, xivo_hip_close_loop_stack). */
int xivo_hip_oos_project(xivo_hip_ctx* ctx, int b0, int nb, int n_oos, const xivo_oos_in* feats,
                         double Roos, int* rows_out);
/* The same with options. XIVO_HIP_OOS_WHOLE_BUFFER reproduces src/oos.cpp:28 AS CODED: SlowGivens is handed the whole
 * 2 * kMaxGroup-row buffers of the feature (src/jac.h:12-16), not the 2k rows ComputeOOSJacobianInternal filled, so every
 * feature contributes 2 * kMaxGroup - 3 rows (kMaxGroup = the layout's n_groups) however many groups saw it. What the
 * reference leaves in the rows behind 2k is unspecified (the buffers are Eigen::resize'd, never cleared); this mode defines
 * them as zero, for which FullPivLU::kernel appends one unit vector per such row: the first 2k - 3 rows are those of the
 * default call, the others are H = 0, inn = 0, diagR = Roos - they change M and S, not K, dx or P+. options = 0 is
 * xivo_hip_oos_project (SURVEY 8 a9's specification: the top 2k rows). feats == NULL re-projects with the options of the
 * call that uploaded the list. */
#define XIVO_HIP_OOS_WHOLE_BUFFER 1u
int xivo_hip_oos_project_ex(xivo_hip_ctx* ctx, int b0, int nb, int n_oos, const xivo_oos_in* feats,
                            double Roos, int* rows_out, unsigned options);
/* One loop-closure match (Estimator::CloseLoopInternal, src/update.cpp:171-212; the mapper that FINDS matches is out of
 * scope): an in-state feature ("old_feature") re-observed by the group in slot group_sind (Graph::LastAddedGroup) at pixel xp
 * (the observation of the new feature that was matched to it). */
typedef struct {
  int feat;        /* position of the old feature in the resident feature list (xivo_hip_set_scene / xivo_hip_edit_batch);
                      its state and anchor group give Xs = Feature::Xs(gbc), src/feature.cpp:107-118; -1 = absent match */
  int group_sind;  /* obs.g->sind()                                                                                      */
  double xp[2];    /* obs.xp                                                                                             */
} xivo_lc_match;
/* Feature::ComputeLCJacobian (src/oos.cpp:92-145) for the n matches of each filter in [b0, b0 + nb) + the stacking of
 * CloseLoopInternal (src/update.cpp:183-196: H_.setZero(2n, N), row pair 2i from match i, diagR_ = Rlc): the staged
 * measurement of those filters becomes the 2n loop-closure rows - d xp / d (group pose, Wbc, Tbc) and, for a context with
 * camera calibration on (xivo_hip_set_calib, cam_dim > 0), the intrinsics block of :125-142. xivo_hip_update_joseph +
 * xivo_hip_absorb_error then complete CloseLoopInternal (:207-208). matches: host, [nb x n]. */
int xivo_hip_close_loop_stack(xivo_hip_ctx* ctx, int b0, int nb, int n, const xivo_lc_match* matches, double Rlc);
/* Measurement compression (use_compression_ / compression_trigger_ratio_, src/estimator.h:399-402 - parsed by the
 * reference, src/estimator.cpp:115-117, but never acted on; xivo::QR, src/helpers.cpp:77-101 "QR-based measurement
 * compression"): call after xivo_hip_oos_project. For every filter whose OOS block has more than trigger_ratio (>= 1;
 * reference default 1.5) times as many rows as non-zero columns, the block is replaced by the triangular factor of its
 * QR decomposition (Householder reflections on the device; the rows only touch the extrinsics and group columns, so
 * the 2k-3 rows of every OOS feature collapse to at most 6 + 6 n_groups rows in total). Orthogonal row operations with
 * isotropic noise leave S^-1-weighted quantities - K, dx, P+ - unchanged to rounding. rows_out[b] (host, may be NULL) =
 * OOS rows of filter b afterwards; the stacked row count M shrinks to in-state rows + the largest of them. */
int xivo_hip_compress_oos(xivo_hip_ctx* ctx, int B, double trigger_ratio, int* rows_out);
/* Estimator::OnePointRANSAC (src/update.cpp:213-393) for filters [0,B) on the resident state; call after
 * xivo_hip_jacobians_instate + xivo_hip_mh_gate (the MH inliers are the input set, as OutlierRejection hands them over,
 * src/manager.cpp:629-650). Low-innovation set {|inn| < ransac_thresh} (the hypothesis loop of :238-258 never uses its
 * random index), BackupState on the device, P rows/cols of non-members zeroed (+ a temporary reference group when
 * gauge_group[b] holds no low-innovation inlier, FindNewRefGroup), partial UpdateJosephForm on the full rows J() +
 * AbsorbError, Jacobians at the updated state, chi-square rescue with ransac_chi2, RestoreState, Jacobians at the
 * original state. The resulting inlier set REPLACES the MH mask: a following xivo_hip_stack / xivo_hip_update_joseph /
 * xivo_hip_absorb_error runs on it.
 *   gauge_group   host [B]: slot of gauge_group_ptr_, -1 = none (NULL: none for every filter)
 *   absorb_groups host [B]: bit g = group slot g is in instate_groups_ when the partial update is absorbed (that list is
 *                 the previous frame's, src/manager.cpp:103); NULL = every slot
 *   inlier_mask_out / chi2_out [B x F], n_rejected_out [B]: host, any may be NULL. chi2 is 0 for features not tested.
 * Online-calibration builds (xivo_hip_set_calib): the calibration state is backed up and restored with X_
 * (src/estimator.cpp:1421-1427, :1442-1448), the partial update runs on the whole rows J() with their td / Cg / bg / intrinsics
 * blocks, AbsorbError retracts td / Cg / Ca / the intrinsics, and the rescue test is the whole-row chi-square of :350-356. */
int xivo_hip_one_point_ransac(xivo_hip_ctx* ctx, int B, double R, double ransac_thresh, double ransac_chi2,
                              const int* gauge_group, const unsigned long long* absorb_groups,
                              unsigned char* inlier_mask_out, double* chi2_out, int* n_rejected_out);
/* jac -> gate -> stack -> UpdateJosephForm in one call (Estimator::UpdateStep's
 * numeric core, src/manager.cpp:72-104) */
int xivo_hip_filter_update(xivo_hip_ctx* ctx, int B, double R, double mh_thresh, double mh_mult,
                           int min_inliers, int use_gating);
int xivo_hip_get_H(xivo_hip_ctx* ctx, int b, int* M_out, double* H, int ldh, double* inn, double* diagR);
/* ---- the step just before the path - depth sub-filter + candidate scoring ----
 * Feature::SubfilterUpdate (src/feature.cpp:246-297): one 3x3 EKF step per feature that is tracked but not
 * in the state yet, against the resident sensor pose / anchor groups (xivo_hip_set_scene), followed by
 * Criteria::Candidate / CandidateStrict (src/options.cpp:10-33) and Feature::score (src/feature.cpp:133-142),
 * which decide who is moved into the state. Embarrassingly parallel: one thread per (filter, feature). */
enum { XIVO_FEAT_INITIALIZING = 0, XIVO_FEAT_READY = 1 };
typedef struct {
  double x[3];            /* in/out: (X/Z, Y/Z, log Z) in the anchor camera frame (src/feature.h:258-262) */
  double P[9];            /* in/out: 3x3 covariance, column-major */
  double xp[2];           /* in: tracked pixel in the current frame */
  double outlier_counter; /* in/out */
  double score;           /* out: -P(2,2) */
  int ref_sind;           /* in: anchor group slot */
  int status;             /* in/out: XIVO_FEAT_INITIALIZING / XIVO_FEAT_READY */
  int init_counter;       /* in/out */
  int candidate;          /* out: bit 0 Criteria::Candidate, bit 1 Criteria::CandidateStrict */
} xivo_subfilter_feat;
typedef struct {
  double Rtri, MH_thresh;           /* SubfilterOptions (src/options.h:25-32): 3.5, 5.991 */
  int ready_steps;                  /* 5 */
  double min_depth, max_depth;      /* cfg min_depth / max_depth: 0.05, 5.0 */
  double max_subfilter_outlier;     /* 0.01 */
} xivo_subfilter_opts;
/* feats: host array [nb x n] (filter-major), updated in place */
int xivo_hip_subfilter_update(xivo_hip_ctx* ctx, int b0, int nb, int n, xivo_subfilter_feat* feats,
                              const xivo_subfilter_opts* opts);

/* Criteria::CandidateComparison (src/options.cpp:34-61): the order in which candidates enter the state
 * (std::sort of src/manager.cpp:375-376,420-421,499-500). feats [nb x n] as returned by xivo_hip_subfilter_update;
 * strict = 0: Criteria::Candidate passes, 1: Criteria::CandidateStrict. order_out [nb x n]: indices of the passing
 * candidates of each filter, best first, padded with -1; n_out [nb] their number. The comparison is reproduced AS
 * CODED: FeatureStatus first (READY before INITIALIZING), then Feature::score() = -P(2,2) - the value it computes from
 * `comparison_score_type` is never used (:41-60). score_out (optional, [nb x n]) returns that value anyway:
 * score_type 0 "DepthUncertainty" -P(2,2); 1 "CovarianceDiagNorm" -|diag P|; 2 "CovarianceDiagNormPlusOutlierCount"
 * -(|diag P| + outlier_counter). Host arithmetic only. */
int xivo_hip_candidate_order(const xivo_subfilter_feat* feats, int nb, int n, int strict, int score_type,
                             int* order_out, int* n_out, double* score_out);

/* ---- Estimator::Propagate on the device-resident state ----
 * For filters [b0, b0 + nb): integrates the nominal motion state (Rsb, Tsb, Vsb, bg, ba, Rsg of the resident
 * xivo_pose_in, xivo_hip_set_scene) over dt with RK4Step (src/rk4.cpp:35-103) or PrinceDormandStep
 * (src/princedormand.cpp:85-221) under the fixed sub-stepping of src/rk4.cpp:13-32 (stepsize < 0: one step),
 * ComposeMotion (src/estimator.cpp:598-613) and ComputeMotionJacobianAt (src/estimator.cpp:615-704, default build:
 * no online IMU calibration), accumulates the sub-step transitions, then applies the covariance tail
 * (src/rk4.cpp:92-102) and P_mm += Qmodel (src/estimator.cpp:590) to the resident P. The IMU sample is modelled as
 * the reference does between two messages: value at the start + slope * t (src/estimator.cpp:556-575). */
typedef struct {
  double gyro[3], accel[3];              /* last_gyro_, last_accel_ */
  double slope_gyro[3], slope_accel[3];  /* (curr - last) / dt */
  double dt;
} xivo_imu_in;
typedef struct {
  double Qimu[144];    /* 12 x 12, column-major (gyro, accel, gyro-bias, accel-bias noise) */
  double Qmodel[529];  /* 23 x 23, column-major */
  double g[3];         /* gravity in the spatial frame before Rsg */
  int method;          /* 0: RK4, 1: PrinceDormand */
  double stepsize;     /* cfg integration stepsize (0.002); < 0: a single step of length dt */
  /* cfg_["PrinceDormand"] of the step-size-controlled branch (src/princedormand.cpp:17-22, :26-60; off in every shipped
   * configuration). control_stepsize != 0 (method 1 only, stepsize > 0) runs that branch AS CODED: PrinceDormandStep returns
   * 0 - its error estimate is commented out (:216-220) - so every step is followed by h *= max_scale_factor, clipped to the
   * end of the sample with the half-step rule (:53-58); a step starts from gyro0 + slope * total_step (:38-39), and the
   * current step h is carried from one sample - and one call - to the next as the reference's function-local static does
   * (:13; per filter here, (re)started at `stepsize` by the first controlled call of a context or a call with another
   * stepsize). tolerance / min_scale_factor only enter through the dead err != 0 arm; attempts is read and never used (:19).
   * All zero (the value-initialised struct of a caller that predates them): the fixed-step branch. */
  int control_stepsize;
  int attempts;
  double tolerance, min_scale_factor, max_scale_factor;
} xivo_prop_opts;
/* imu: [nb][n_imu] - the n_imu samples of each filter since its last call (one Estimator::Propagate each, in order);
 * their transitions are accumulated on chip and the O(23 N) cross-covariance tail is applied once. */
int xivo_hip_propagate(xivo_hip_ctx* ctx, int b0, int nb, int n_imu, const xivo_imu_in* imu, const xivo_prop_opts* opts);
/* The same for an online-calibration build (xivo_hip_set_calib with td >= 0 or Cg >= 0): kMotionSize = Index::End
 * (src/core.h:40-75) = Cg + 15, or td + 1 without the IMU calibration; ComposeMotion with the resident imu_.Cg() / imu_.Ca()
 * (xivo_calib_in), the motion Jacobian with the dWsb/dCg and dVsb/dCa blocks (src/estimator.cpp:626-638, :674-688), the tail
 * over motion_size rows / columns. Qmodel: motion_size x motion_size, column-major (opts->Qmodel is not read).
 * opts->control_stepsize works as in xivo_hip_propagate (the step a filter carries is shared by the two entry points).
 * xivo_hip_propagate itself refuses such a context (XIVO_HIP_ERR_UNSUPPORTED): its motion block is the default build's 23. */
int xivo_hip_propagate_calib(xivo_hip_ctx* ctx, int b0, int nb, int n_imu, const xivo_imu_in* imu, const xivo_prop_opts* opts,
                             const double* Qmodel);

/* ---- orthonormal Givens elimination and QR measurement compression ----
 * Batched xivo::Givens (src/helpers.cpp:48-75) and xivo::QR (src/helpers.cpp:78-101) on host arrays, nb
 * independent problems of identical shape, column-major, leading dimension = rows.
 *  Givens: eliminates Hf [rows x nf] with the rotations of G&VL Alg. 5.1.3 (the reference's givens(), eps guard
 *          1e-4f), rotates x and - as the reference codes it, helpers.cpp:64 - only the first nf columns of
 *          Hx [rows x nx]; then strips the first nf rows. rows_out[b] = rows - nf.
 *  QR:     triangularises Hx [rows x nx] (all columns rotated), rotates x; rows_out[b] = rows (the caller keeps
 *          the top block). effective_rows = -1: all rows. */
int xivo_hip_givens(xivo_hip_ctx* ctx, int nb, int rows, int nx, int nf, double* x, double* Hx, double* Hf,
                    int effective_rows, int* rows_out);
int xivo_hip_qr(xivo_hip_ctx* ctx, int nb, int rows, int nx, double* x, double* Hx, int effective_rows, int* rows_out);

/* Estimator::AbsorbError (src/estimator.cpp:875-921) on the device-resident nominal state of filters
 * [0,B): X += dx via State::operator+= (src/core.h:135-165: SO3 exp on Rsb, Rbc, Rsg), every group slot
 * += dx segment (src/group.h:25-29), every feature that was an inlier of the last gating / stacking pass
 * x += dx segment (src/feature.h:220); then dx = 0. (no host round trip of the state.) */
int xivo_hip_absorb_error(xivo_hip_ctx* ctx, int B);
/* download the resident scene (any pointer may be NULL) */
int xivo_hip_get_scene(xivo_hip_ctx* ctx, int b0, int nb, xivo_pose_in* poses, xivo_group_in* groups,
                       xivo_feat_in* feats);

/* ---- resident state edits between updates, batched over filters ----
 * The reference edits X_/P_ one filter at a time on the host (the functions cited per kind). A sequence driver that
 * keeps thousands of filters resident cannot afford one launch per edit, so a whole frame's edits of all filters go
 * down in one call: ops must be grouped by filter with non-decreasing `b`; the ops of one filter are applied in
 * array order by one workgroup, different filters run concurrently. Offsets are error-state indices. */
enum {
  XIVO_EDIT_P_ZERO_RC = 0,     /* i0 = off, i1 = len             : as xivo_hip_p_zero_rc                              */
  XIVO_EDIT_P_COPY_RC = 1,     /* i0 = dst, i1 = src, i2 = len   : as xivo_hip_p_copy_rc                              */
  XIVO_EDIT_P_SET_BLOCK3 = 2,  /* i0 = off, v[0..8] = P3         : as xivo_hip_p_set_block3                           */
  /* Estimator::AddGroupToState (src/estimator.cpp:801-816), i0 = group slot: resident group[i0] <- current (Rsb,Tsb),
   * P rows then columns of the slot <- those of Wsb, then of Tsb */
  XIVO_EDIT_ADD_GROUP = 3,
  /* Estimator::RemoveGroupFromState (src/estimator.cpp:745-759), i0 = group slot */
  XIVO_EDIT_REMOVE_GROUP = 4,
  /* Estimator::AddFeatureToState (src/estimator.cpp:820-846) + Feature::FillCovarianceBlock (src/feature.cpp:753-760):
   * i0 = position j in the resident feature list, i1 = feature slot sind, i2 = anchor group slot,
   * v[0..2] = x, v[3..4] = xp, v[5..13] = the feature's own 3x3 covariance (column-major) */
  XIVO_EDIT_ADD_FEATURE = 5,
  /* Estimator::RemoveFeatureFromState (src/estimator.cpp:762-783), i0 = position j: its slot's rows/cols are zeroed
   * and the entry becomes absent (sind = -1) */
  XIVO_EDIT_REMOVE_FEATURE = 6,
  /* new tracked pixel of the feature at position i0 (Feature::back()), v[0..1] = xp */
  XIVO_EDIT_SET_XP = 7
};
typedef struct {
  int b;             /* filter */
  int kind;          /* XIVO_EDIT_* */
  int i0, i1, i2;
  int reserved;
  double v[14];
} xivo_edit_op;
/* F = length of the resident feature list the ops index (positions 0..F-1; entries never written are absent).
 * F <= M_max / 2. Sets the list length used by the following Jacobian / gating / update calls. */
int xivo_hip_edit_batch(xivo_hip_ctx* ctx, int F, int n_ops, const xivo_edit_op* ops);

/* New tracked pixels of a whole frame in one dense array (what the tracker hands over per camera frame; replaces one
 * XIVO_EDIT_SET_XP op per feature): xp is [nb][F][2]; a NaN pair leaves that entry's pixel untouched (feature not
 * tracked in this frame / absent entry). Sets the list length F like xivo_hip_edit_batch. */
int xivo_hip_set_pixels(xivo_hip_ctx* ctx, int b0, int nb, int F, const double* xp);

/* ---- covariance propagation tail (src/rk4.cpp:92-102, src/estimator.cpp:590) */
/* P_mm <- Pmm_new ; P_ms <- Phi P_ms ; P_sm <- P_sm Phi^T. Phi and Pmm_new
 * are nm x nm (nm = kMotionSize: 23, or up to 40 for the online-calibration builds), one pair per filter. */
int xivo_hip_propagate_cov(xivo_hip_ctx* ctx, int b0, int nb, int nm, const double* Phi,
                           const double* Pmm_new);

/* ---- measurement helpers (bench.py / tests only) ----------------------- */
/* device buffers on the context's GPU for inputs that are "already resident" (what xivo_hip_set_measurements_device is
 * handed): allocate, upload `bytes` from the host and replicate that block until `total_bytes` are filled, free. */
int xivo_hip_dev_alloc(xivo_hip_ctx* ctx, size_t bytes, void** out);
int xivo_hip_dev_free(xivo_hip_ctx* ctx, void* p);
int xivo_hip_dev_upload(xivo_hip_ctx* ctx, void* dst, const void* src, size_t bytes, size_t total_bytes);
int xivo_hip_timer_begin(xivo_hip_ctx* ctx);
int xivo_hip_timer_end(xivo_hip_ctx* ctx, float* ms_out);
/* per-stage accumulated GPU time (ms) and launch counts since the last reset;
 * needs XIVO_HIP_FLAG_PROFILE. names_out[i] points to static strings. */
#define XIVO_HIP_MAX_STAGES 16
int xivo_hip_profile_reset(xivo_hip_ctx* ctx);
int xivo_hip_profile_get(xivo_hip_ctx* ctx, int* n_stages, const char** names_out,
                         float* ms_out, int* launches_out, double* flops_per_launch_out);
/* fp64 MFMA issue-rate microbenchmark of v_mfma_f64_16x16x4_f64; out4 = {TFLOP/s full chip,
 * cycles per MFMA (1 wave/SIMD), sustained clock GHz (lower bound), TFLOP/s 1 wave/SIMD} */
int xivo_hip_bench_mfma_peak(xivo_hip_ctx* ctx, double* out4);
/* tile the batched GEMM picks for an (rows x cols) output (symmetric = lower
 * triangle + mirror mode), for DESIGN.md/tests */
void xivo_hip_gemm_tile(int rows, int cols, int symmetric, int* bm, int* bn);
/* which rows the last update call used: 0 = dense, 1 = row-pair compressed (sparse-H) */
int xivo_hip_last_path(xivo_hip_ctx* ctx);
/* the route the last update pass took (round 6: the one table in capi.hip, plan_update): 0 fused (one kernel), 1 sparse rows +
 * in-solve whitened update, 2 sparse rows + whitened outputs + tiled product, 3 sparse symmetric form, 4 sparse stand-alone tail,
 * 5 dense as-coded, 6 dense rows + whitened update, 7 dense symmetric form; xivo_hip_route_name gives the table's name */
int xivo_hip_last_route(xivo_hip_ctx* ctx);
const char* xivo_hip_route_name(int route);
/* kernel instantiation the last launch of profile stage `stage` ran (index as in xivo_hip_profile_get; needs
 * XIVO_HIP_FLAG_PROFILE), spelled as rocprofv3 --kernel-trace prints it minus spaces; "" if none */
const char* xivo_hip_stage_kernel(xivo_hip_ctx* ctx, int stage);
/* algorithmic HBM bytes of that launch: every input and every output of the stage once */
double xivo_hip_stage_bytes(xivo_hip_ctx* ctx, int stage);

#ifdef __cplusplus
}
#endif
#endif /* XIVO_HIP_H_ */
