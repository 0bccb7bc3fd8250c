// Launchers of the EKF-specific (non-GEMM) kernels. See ekf_kernels.hip.
#pragma once
#include "common.h"
#include "ell.h"
#include "../../include/xivo_hip.h"

namespace xivo_hip {

// raw (n x n, ld = n, contiguous per filter) <-> padded (ld = ldp) covariance
int launch_unpack_P(const double* raw, double* P, int N, int Np, int ldp, long strideP, int batch,
                    hipStream_t s);
int launch_pack_P(const double* P, double* raw, int N, int ldp, long strideP, int batch, hipStream_t s);

// One-filter plumbing call (dropin.hip, capi.hip: xivo_hip_update_joseph_host): the boundary kernels address page-locked
// host memory directly. `block` = the staged compressed rows of the filter: ints idx[pairs_clear][ELL_W] at off_idx,
// doubles val[pairs_clear][ELL_W][2] at off_val, inn[Mpmax] at off_inn, diagR[Mpmax] at off_R, ints {nc, pw, over} at off_flags.
struct DropinInArgs {
  const double* Psrc; int ldps;          // host-mapped N x N covariance (null: the device copy is current)
  double* P; int N, Np, ldp;             // the filter's padded device covariance
  const void* block; int off_idx, off_val, off_inn, off_R, off_flags;
  int pairs_clear, Mpmax;
  int* idx; double* val; double* inn; double* diagR; int* nc; int* pw; int* over;   // the filter's device buffers
};
struct DropinOutArgs {
  const double* P; int N, ldp;           // the filter's padded device covariance
  double* Pdst; int ldpd;                // host-mapped destination (null: P stays on the device)
  const double* err; double* err_dst;    // dx [N]
  const int* status; const int* ldlt_used; int* flags_dst;   // -> {status, ldlt_used}
};
int launch_dropin_in(const DropinInArgs& a, hipStream_t s);
int launch_dropin_out(const DropinOutArgs& a, hipStream_t s);

// raw H (M x N, ld = M), inn (M), diagR (M)  ->  padded H (Mp x Np, ld ldh),
// H^T (Np x Mp, ld ldht), inn (Mp, zero pad), diagR (Mp, pad = 1)
struct MeasBuffers {
  double* H; long strideH; int ldh;
  double* HT; long strideHT; int ldht;
  double* inn; long strideInn;
  double* diagR; long strideR;
};
int launch_unpack_meas(const double* rawH, long strideRaw, int ldraw, const int* only_if, MeasBuffers mb,
                       int M, int Mp, int N, int Np, int batch, hipStream_t s);

// P edits
// H^T rebuilt from the dense H of every filter (the G-level producers may skip writing it: capi.hip skip_HT)
int launch_transpose_H(const double* H, long strideH, int ldh, double* HT, long strideHT, int ldht, int Mp, int Np, int batch,
                       hipStream_t s);
int launch_p_zero_rc(double* P, int ldp, int Np, int off, int len, hipStream_t s);
int launch_p_copy_rc(double* P, int ldp, int Np, int dst, int src, int len, hipStream_t s);
int launch_p_diag(const double* P, int ldp, int N, double* out, hipStream_t s);

// dense-row Mahalanobis gating (update.cpp:60-96) + neutralising rejected rows
struct GateDenseArgs {
  const double* H; long strideH; int ldh;       // candidate rows (J of feature f = rows 2f, 2f+1)
  const double* HP; long strideHP; int ldhp;    // H * P of the same rows
  double* Hw; double* HTw; long strideHT; int ldht;  // H / H^T to neutralise
  double* HPw; double* PHTw;                         // optional: HP / (HP)^T rows to neutralise too
  const double* PHTr;                                // (HP)^T = P H^T of the candidate rows [Np x Mp]
  double* inn; long strideInn;
  double* diagR; long strideR;
  unsigned char* mask; double* dist;            // [batch x F]  (row stride mask_ld when it is set)
  int F, Np, batch;
  double R, thresh, mult; int min_inliers;
  EllBuffers ell; int have_ell;                 // also zero the rejected pairs of the compressed form (ell.h)
  int mask_ld;                                  // 0: rows of mask / dist are F entries apart
  const xivo_feat_in* feats; int Fmax;          // optional [batch x Fmax]: entries with sind < 0 are absent (ragged batches)
  int no_relax;                                 // 1: inlier <=> distance < thresh, no relaxation loop (min_inliers = -1: every filter tested)
};
int launch_gate_dense(const GateDenseArgs& a, hipStream_t s);

// G-level kernels
struct SceneBuffers {
  const xivo_pose_in* poses;     // [batch]
  const xivo_group_in* groups;   // [batch x n_groups]
  const xivo_feat_in* feats;     // [batch x Fmax]
  double* J;                     // [batch x Fmax x 42]  (2 x 21 row-major)
  double* finn;                  // [batch x Fmax x 2]
  unsigned char* mask;           // [batch x Fmax]
  double* dist;                  // [batch x Fmax]
  int Fmax, F;
  // online-calibration builds (include/xivo_hip.h, xivo_hip_set_calib): null / td = -1, cam_dim = 0 when switched off
  const xivo_calib_in* calib;    // [batch]
  double* Jc;                    // [batch x Fmax x 44]  (2 x 22 row-major: td | Cg 9 | bg 3 | intrinsics 9)
  xivo_calib_layout cl;
  int invdepth;                  // XIVO_HIP_FLAG_INVDEPTH: features are (X/Z, Y/Z, 1/Z) (USE_INVDEPTH build, feature.cpp:98-105)
};
int launch_jac_instate(const SceneBuffers& sb, const xivo_layout& lay, const xivo_cam& cam, int batch,
                       hipStream_t s);
struct GateArgs {
  SceneBuffers sb; xivo_layout lay;
  const double* P; long strideP; int ldp;
  double R, thresh, mult; int min_inliers; int batch; int use_gating;
};
int launch_gate_sparse(const GateArgs& a, hipStream_t s);
struct StackArgs {
  SceneBuffers sb; xivo_layout lay; MeasBuffers mb;
  int Mp, Np, batch; double R; int fix_group_block;
  int* rows_instate;   // [batch] out: 2 * F (rows reserved for in-state features)
  EllBuffers ell; int emit_ell;   // also emit the row-pair compressed form (ell.h)
  int write_dense;                // 0: only inn / diagR / ELL (the dense H, H^T are materialised on demand)
  // online-calibration builds on the sparse pipeline: [batch] dense blocks [Mp x lead_k] (leading dimension Mp) that take the
  // calibration columns of every row pair; null: those builds stack dense rows
  double* lead; long strideLead; int lead_k;
};
int launch_stack(const StackArgs& a, hipStream_t s);

// Feature::SubfilterUpdate + candidate tests (feature.cpp:246-297, options.cpp:10-33)
int launch_subfilter(xivo_subfilter_feat* feats, int n, const xivo_pose_in* poses, const xivo_group_in* groups,
                     int n_groups, xivo_cam cam, xivo_subfilter_opts o, int batch, hipStream_t s,
                     const xivo_calib_in* calib = nullptr, int cam_dim = 0, int invdepth = 0);

// Estimator::Propagate state + covariance stages (rk4.cpp, princedormand.cpp, estimator.cpp:598-704): one wave per
// filter; writes the accumulated transition Phi and the new P_mm (23 x 23 each, column-major) for the tail kernel
struct PropStateArgs {
  xivo_pose_in* poses; const xivo_imu_in* imu;      // [nb] / [nb][n_imu] (poses already offset to b0)
  int n_imu;
  const double* Qimu; const double* Qmodel;         // device copies
  double g[3]; int method; double stepsize;
  const double* P; long strideP; int ldp;           // resident covariance (offset to b0)
  double* Phi_out; double* Pmm_out;                 // [nb][529] ([nb][nm * nm] for the calibration kernel)
  int batch;
  // online-calibration builds (launch_propagate_state_calib): motion size, slot of Cg (-1: none; Ca follows at + 9), the
  // resident calibration state (offset to b0); Qmodel is nm x nm
  int nm, iCg; const xivo_calib_in* calib;
  // step-size-controlled Dormand-Prince (princedormand.cpp:26-60; default-build kernel only): the per-filter step carried between
  // samples and calls (the reference's function-local static h), null = fixed steps
  double* pd_h; double pd_tol, pd_min_scale, pd_max_scale;
};
int launch_propagate_state(const PropStateArgs& a, hipStream_t s);
int launch_propagate_state_calib(const PropStateArgs& a, hipStream_t s);

// xivo::Givens / xivo::QR (helpers.cpp:27-101), one wave per problem, in place
struct GivensArgs {
  double* x; double* Hx; double* Hf;   // per problem: x [rows], Hx [rows x nx], Hf [rows x nf] (null for QR)
  int rows, nx, nf, eff, batch, qr;
};
int launch_givens(const GivensArgs& a, hipStream_t s);

// AbsorbError on the resident scene (estimator.cpp:875-921)
struct AbsorbArgs {
  xivo_pose_in* poses; xivo_group_in* groups; xivo_feat_in* feats; const unsigned char* mask;
  double* err; long strideErr; xivo_layout lay; int F, Fmax, batch;
  const int* status;   // [batch] factorisation status of the update that produced err: non-zero -> nothing is absorbed, err <- 0
  const unsigned long long* group_mask;   // optional [batch]: bit g = group slot g is in instate_groups_ (null: every slot)
  int* counter;   // [batch] State::counter (core.h:120-122): absorbs so far, drives the periodic SO3 re-normalisation
  xivo_calib_in* calib; xivo_calib_layout cl;   // online-calibration builds: td / Cg / Ca / intrinsics retracted too (null: default build)
};
int launch_absorb_error(const AbsorbArgs& a, hipStream_t s);

// Estimator::OnePointRANSAC on the resident state (update.cpp:213-393): selection, P zeroing, rescue test
struct RansacArgs {
  SceneBuffers sb; xivo_layout lay;
  const double* P; long strideP; int ldp; int Np;
  double R, thresh, chi2;
  const int* gauge;                 // [batch] slot of gauge_group_ptr_ (-1: none), may be null
  unsigned char* low;               // [batch x Fmax] out: low-innovation set (all 0 for filters in state 0 / 2)
  const unsigned char* low_keep;    // rescue: the set select found (a copy - `low` doubles as the stacking mask)
  unsigned long long* zero_groups;  // [batch] out
  int* state;                       // [batch] out: 0 nothing to do, 1 partial update + rescue, 2 rescue against the prior
  unsigned char* keep; double* chi; int* n_rejected;   // rescue outputs
  int batch;
};
int launch_ransac_select(const RansacArgs& a, hipStream_t s);
int launch_ransac_zero(const RansacArgs& a, double* P, hipStream_t s);
int launch_ransac_rescue(const RansacArgs& a, hipStream_t s);
int launch_ransac_rescue_dist(const RansacArgs& a, const double* dist /* [batch x ld] */, int ld, hipStream_t s);

// batched resident edits (xivo_hip_edit_batch): wg_filter[w] = filter of workgroup w, its ops are
// ops[wg_begin[w] .. wg_begin[w + 1])
struct EditArgs {
  const xivo_edit_op* ops; const int* wg_filter; const int* wg_begin;
  double* P; long strideP; int ldp, Np; xivo_layout lay;
  xivo_pose_in* poses; xivo_group_in* groups; xivo_feat_in* feats; int Fmax;
};
int launch_edit_batch(const EditArgs& a, int n_wg, hipStream_t s);
int launch_set_pixels(xivo_feat_in* feats /* already offset to b0 */, int Fmax, int F, const double* xp, int nb, hipStream_t s);

// OOS (MSCKF) rows: oos.cpp:39-89 + helpers.cpp:13-23
struct OosArgs {
  const xivo_oos_in* feats; int n_oos;          // [batch x n_oos]
  const xivo_pose_in* poses; const xivo_group_in* groups;
  xivo_layout lay; xivo_cam cam; MeasBuffers mb;
  const xivo_calib_in* calib; int cam_dim;   // online camera calibration: per-filter intrinsics (null / 0: the context's camera)
  int row0;            // first free row (after the in-state rows)
  int Mp, Np, batch; double Roos;
  int* rows_out;       // [batch]
  int whole;           // 0, or the 2 kMaxGroup rows of the reference's per-feature buffers (XIVO_HIP_OOS_WHOLE_BUFFER)
};
int launch_oos(const OosArgs& a, hipStream_t s);

// loop-closure rows: oos.cpp:92-145 + the stacking of update.cpp:183-196
struct LcArgs {
  const xivo_lc_match* matches; int n;          // [batch x n]
  const xivo_pose_in* poses; const xivo_group_in* groups; const xivo_feat_in* feats; int Fmax;
  xivo_layout lay; xivo_cam cam; const xivo_calib_in* calib; xivo_calib_layout cl; int invdepth;
  double* H; long strideH; int ldh;             // [batch] 2n x N column-major, zero-filled
  double* inn; double* diagR; long strideV;     // [batch] 2n each
  double Rlc; int batch;
};
int launch_lc_rows(const LcArgs& a, hipStream_t s);

// measurement compression of the appended OOS rows (estimator.h:399-402, helpers.cpp:77-101)
struct OosCompressArgs {
  xivo_layout lay; MeasBuffers mb;
  int row0;              // first OOS row
  const int* rows;       // [batch] OOS rows of each filter (as written by launch_oos)
  int* rows_out;         // [batch] rows after compression (may alias rows)
  double ratio, Roos;
  int batch;
};
// returns -1 (nothing launched) when the block is larger than the built instantiations
int launch_oos_compress(const OosCompressArgs& a, int rows_max, hipStream_t s);

// propagation tail (rk4.cpp:92-102)
int launch_propagate_cov(double* P, long strideP, int ldp, int N, int Np, int nm, const double* Phi,
                         const double* Pmm, int b0, int nb, hipStream_t s);

int launch_mfma_peak(double* sink, int iters, int blocks, hipStream_t s);

// Diagonally pivoted L D L^T fallback (Eigen's S.ldlt().solve of src/estimator.cpp:1266) + the as-coded Joseph update for
// the filters whose status is non-zero (ldlt_fallback.hip); clears their status and sets used[filt]
}  // namespace xivo_hip
#include "ell.h"
namespace xivo_hip {
struct LdltFallbackArgs {
  int* status; int* used;
  EllBuffers ell; const double* H; long strideH; int ldh; int use_dense;
  int mixed_row0;   // >= 0: rows below it are row-pair compressed, rows from it on are dense (OOS rows behind in-state rows)
  const double* lead; long strideLead; int ldlead, lead_k;   // leading dense block next to the compressed rows (online-calibration stacking), or null
  const double* PHT; long stridePHT; int ldpht;
  double* S; long strideS; int lds;
  double* K; long strideK; int ldk;
  double* A; long strideA; int lda;     // scratch: I - K H
  double* T; long strideT; int ldt;     // scratch: (I - K H) P
  double* P; long strideP; int ldp;
  const double* inn; long strideInn;
  const double* diagR; long strideR;
  double* err; long strideErr;
  int N, M, batch;
};
int launch_ldlt_fallback(const LdltFallbackArgs& a, hipStream_t s);

}  // namespace xivo_hip
