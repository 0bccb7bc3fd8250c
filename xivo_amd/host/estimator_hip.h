// C++ host adapter: the reference's Estimator / Feature hot-path surface on top
// of the C ABI (include/xivo_hip.h).
//
// The reference exposes this path only as member functions of its singleton
// `Estimator` and of `Feature` (src/estimator.h:261-313, src/feature.h:134-188).
// This header keeps those names, argument meanings and error behaviour so that
// the calling code of src/manager.cpp:72-104 (ComputeInstateJacobians ->
// OutlierRejection/MHGating -> FilterUpdate) reads the same. It runs in the
// "plumbing" mode of SURVEY.md 8b: the host members P_, H_, inn_, diagR_, err_
// stay authoritative, every call uploads what the device needs and downloads
// what the reference would have left in its members. (The throughput path is
// the batched C ABI itself; this adapter is the drop-in for one filter.)
//
// Matrix types: minimal column-major `MatX` / `VecX` / `Vec2` / `Vec3` / `Mat3` with Eigen's memory layout and the
// subset of Eigen's interface the adapter uses (operator(), rows(), cols(), data(), setZero, resize). With
// -DXIVO_HIP_USE_EIGEN they are the reference's own aliases (common/alias.h) instead - the build a maintainer uses
// inside the reference tree; tests/test_boundary_cpu.py compiles both.
#pragma once
#include <cmath>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/xivo_hip.h"

// The reference's online-calibration builds (src/CMakeLists.txt:13-15): compile the adapter with the same defines and it
// carries the extra state (X_.td, imu_.Cg() / Ca(), the camera intrinsics), the extra Jacobian blocks of Feature::J_ /
// FillJacobianBlock (src/feature.cpp:592-651, :664-683), their retraction in AbsorbError (src/estimator.cpp:875-890) and
// switches the context with xivo_hip_set_calib. (xivo_amd/host/Makefile builds libxivo_host_calib.so that way.)
#if defined(USE_ONLINE_TEMPORAL_CALIB) || defined(USE_ONLINE_IMU_CALIB) || defined(USE_ONLINE_CAMERA_CALIB)
#define XIVO_HIP_ONLINE_CALIB 1
#else
#define XIVO_HIP_ONLINE_CALIB 0
#endif

namespace xivo {
namespace hip {

#ifdef XIVO_HIP_USE_EIGEN
}  // namespace hip
}  // namespace xivo
// Inside the reference tree: the reference's own matrix aliases (common/alias.h: Eigen 3.3.9 + Sophus, column-major
// double; the tree is built with -DEIGEN_INITIALIZE_MATRICES_BY_ZERO, CMakeLists.txt:43). tests/test_boundary_cpu.py
// compiles this adapter that way against /root/reference's headers.
#include "alias.h"
namespace xivo {
namespace hip {
using ::xivo::number_t;
using ::xivo::MatX;
using ::xivo::VecX;
using ::xivo::Vec2;
using ::xivo::Vec3;
using ::xivo::Mat3;
#else
using number_t = double;  // common/alias.h:11


struct VecX {
  std::vector<number_t> v;
  VecX() = default;
  explicit VecX(int n) : v(n, 0.0) {}
  int size() const { return (int)v.size(); }
  int rows() const { return (int)v.size(); }
  void setZero(int n) { v.assign(n, 0.0); }
  void resize(int n) { v.resize(n); }
  number_t& operator()(int i) { return v[i]; }
  number_t operator()(int i) const { return v[i]; }
  number_t* data() { return v.data(); }
  const number_t* data() const { return v.data(); }
};

struct MatX {  // column-major, like Eigen's default (CMakeLists.txt:42)
  std::vector<number_t> v;
  int r = 0, c = 0;
  MatX() = default;
  MatX(int rows, int cols) : v((size_t)rows * cols, 0.0), r(rows), c(cols) {}
  int rows() const { return r; }
  int cols() const { return c; }
  void setZero(int rows, int cols) { r = rows; c = cols; v.assign((size_t)rows * cols, 0.0); }
  number_t& operator()(int i, int j) { return v[(size_t)j * r + i]; }
  number_t operator()(int i, int j) const { return v[(size_t)j * r + i]; }
  number_t* data() { return v.data(); }
  const number_t* data() const { return v.data(); }
};

// fixed-size types: zero-initialised like the reference's Eigen build (EIGEN_INITIALIZE_MATRICES_BY_ZERO)
struct Vec2 { number_t v[2] = {0, 0}; number_t& operator()(int i) { return v[i]; } number_t operator()(int i) const { return v[i]; }
              number_t* data() { return v; } const number_t* data() const { return v; } };
struct Vec3 { number_t v[3] = {0, 0, 0}; number_t& operator()(int i) { return v[i]; } number_t operator()(int i) const { return v[i]; }
              number_t* data() { return v; } const number_t* data() const { return v; } };
struct Mat3 {  // column-major 3x3
  number_t v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  number_t& operator()(int i, int j) { return v[3 * j + i]; }
  number_t operator()(int i, int j) const { return v[3 * j + i]; }
  number_t* data() { return v; }
  const number_t* data() const { return v; }
};
#endif
inline Mat3 Identity3() { Mat3 I; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) I(i, j) = i == j ? 1.0 : 0.0; return I; }

enum class FeatureStatus { INSTATE, GAUGE, REJECTED_BY_FILTER };  // subset of src/core.h used on this path

// Group anchor (src/group.h:41-107): pose + state slot.
struct Group {
  Mat3 Rsb_ = Identity3(); Vec3 Tsb_; int sind_ = -1;
  const Mat3& Rsb() const { return Rsb_; }
  const Vec3& Tsb() const { return Tsb_; }
  int sind() const { return sind_; }
};
using GroupPtr = Group*;

class Estimator;

// Feature (src/feature.h:74-232): the members the hot path reads and writes.
class Feature {
 public:
  Vec3 x_;                 // (X/Z, Y/Z, log Z), feature.h:258-262
  Vec2 back_;              // last tracked pixel
  GroupPtr ref_ = nullptr;
  int sind_ = -1;
  FeatureStatus status_ = FeatureStatus::INSTATE;

  const Vec2& back() const { return back_; }
  GroupPtr ref() const { return ref_; }
  int sind() const { return sind_; }
  FeatureStatus status() const { return status_; }
  void SetStatus(FeatureStatus s) { status_ = s; }
  // J(): 2 x kFullSize row pair, inn(): innovation (feature.h:187-188 accessors)
  const MatX& J() const { return J_; }
  const Vec2& inn() const { return inn_; }
  // Feature::FillJacobianBlock (src/feature.cpp:658-684), including the
  // group-block overwrite of :675-676 unless the estimator was created with
  // XIVO_HIP_FLAG_FIX_GROUP_BLOCK.
  void FillJacobianBlock(MatX& H, int offset) const;

 private:
  friend class Estimator;
  MatX J_;
  Vec2 inn_;
  const Estimator* owner_ = nullptr;
};
using FeaturePtr = Feature*;

// Estimator: the hot-path members and methods of src/estimator.h:261-313,423-509.
class Estimator {
 public:
  Estimator(const xivo_layout& layout, const xivo_cam& cam, int max_features, unsigned flags = 0, int device = 0);
  ~Estimator();
  Estimator(const Estimator&) = delete;
  Estimator& operator=(const Estimator&) = delete;

  // ---- members with the reference's names (src/estimator.h:423-509) ----
  MatX P_, H_, K_;         // K_ holds K*sqrt(R) after UpdateJosephForm, as in the reference (estimator.cpp:1282-1286)
  VecX inn_, diagR_, err_;
  number_t R_ = 2.25;      // visual_meas_std^2 (estimator.cpp:334-335)
  number_t MH_thresh_ = 5.991, MH_thresh_multipler_ = 1.1;  // estimator.cpp:366-369
  int min_required_inliers_ = 5;
  bool use_MH_gating_ = true;
  int num_mh_rejected_ = 0;
  // nominal state pieces ComputeInstateJacobians passes down (update.cpp:27-28)
  Mat3 Rsb_ = Identity3(), Rbc_ = Identity3(); Vec3 Tsb_, Tbc_;
  std::vector<FeaturePtr> instate_features_;
  std::vector<FeaturePtr> in_current_ekf_update_;
  std::vector<GroupPtr> groups_;   // indexed by slot `sind`

  // ---- propagation members (src/estimator.h: X_, g_, Qimu_, Qmodel_, slope_*, last_/curr_ IMU) ----
  Vec3 Vsb_, bg_, ba_;     // X_.Vsb, X_.bg, X_.ba (Rsb_/Tsb_/Rbc_/Tbc_ above are the rest of X_)
  Mat3 Rsg_ = Identity3(); // X_.Rsg
  Vec3 g_;                 // gravity, src/estimator.cpp:g_
  MatX Qimu_, Qmodel_;     // 12x12, 23x23 (src/estimator.cpp:590)
  Vec3 slope_accel_, slope_gyro_, last_accel_, last_gyro_, curr_accel_, curr_gyro_;
  std::string integration_method_ = "PrinceDormand";   // cfg/tumvi_cam0.json:14
  number_t stepsize_ = 0.002;                          // RK4.stepsize / PrinceDormand.stepsize

  // ---- methods with the reference's names ----
  // Estimator::Propagate (src/estimator.cpp:539-592). `dt` = curr_time_ - last_time_, which the
  // reference takes from its message timestamps. The 23x23 stage arithmetic (RK4Step,
  // PrinceDormandStep, ComputeMotionJacobianAt, ComposeMotion) runs on the host exactly as in
  // the reference; the O(23 x N) cross-covariance tail (src/rk4.cpp:92-102) and P_mm += Qmodel
  // run on the device, once per call, with the sub-step transitions accumulated.
  void Propagate(bool visual_meas, number_t dt);
  void UpdateJosephForm();                 // src/estimator.cpp:1257-1288
  void ComputeInstateJacobians();          // src/update.cpp:24-32
  std::vector<FeaturePtr> MHGating();      // src/update.cpp:50-116
  void FilterUpdate();                     // src/update.cpp:120-153 (AbsorbError is left to the caller)
  void AbsorbError();                      // src/estimator.cpp:875-921 (host, O(N))
  // Estimator::OnePointRANSAC (src/update.cpp:213-393): a composition of the pieces above -
  // low-innovation set, BackupState, P row/col zeroing, partial UpdateJosephForm on the full J
  // rows, AbsorbError, re-Jacobians + chi-square rescue on the device, RestoreState.
  std::vector<FeaturePtr> OnePointRANSAC(const std::vector<FeaturePtr>& mh_inliers);
  number_t ransac_thresh_ = 5, ransac_Chi2_ = 5.89;        // src/estimator.cpp:132-134
#if XIVO_HIP_ONLINE_CALIB
  // ---- online-calibration state (src/core.h:117-130 X_.td; src/imu.h:12-27 imu_.X_; the camera's own parameters) ----
  number_t td_ = 0;                        // X_.td
  Mat3 Cg_ = Identity3(), Ca_ = Identity3();   // imu_.Cg(), imu_.Ca()
  number_t intr_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // fx fy cx cy + the model's distortion parameters (the state slots' order)
  const xivo_calib_layout& calib_layout() const { return cl_; }
  int motion_size() const { return motion_; }   // kMotionSize of this build (24 / 38 / 39)
#endif
  GroupPtr gauge_group_ptr_ = nullptr;
  std::vector<GroupPtr> instate_groups_;   // groups AbsorbError retracts (src/manager.cpp:103)
  int num_oneptransac_rejected_ = 0;
  int absorb_counter_ = 0;                 // State::counter (src/core.h:120-122)
  int num_not_spd_ = 0;                    // updates dropped because S was not positive definite (P_ kept, err_ = 0)
  bool last_update_ok_ = true;
  std::vector<number_t> ransac_chi2_;      // chi-square distances of the rescue step (for tests/diagnostics)
  // host edits of P_ stay plain host code on the authoritative host copy

  // UpdateJosephForm crosses PCIe with P_ twice per call. trust_device_P_ = true lets it skip the upload when the device
  // copy is known to be current (the last adapter call that moved P_ was an upload or a download and nothing has edited
  // P_ on the host since); host code that edits P_ (AddGroupToState, RemoveFeatureFromState, ... - SURVEY a17) then calls
  // InvalidateDeviceP(). Off by default: P_ is a public member, as in the reference, and the adapter cannot see writes to it.
  bool trust_device_P_ = false;
  void InvalidateDeviceP() { device_P_current_ = false; }
  void SyncDeviceP();                      // uploads P_ unless the device copy is known to be current
  bool legacy_plumbing_ = false;           // UpdateJosephForm through the six general calls of rounds 1-3 (A/B timing)

  const xivo_layout& layout() const { return lay_; }
  unsigned flags() const { return flags_; }

 private:
  void Check(int status, const char* what) const;
  bool device_P_current_ = false;
#if XIVO_HIP_ONLINE_CALIB
  xivo_calib_layout cl_{-1, -1, 0, 0};
  int motion_ = 23;
#endif
  xivo_hip_ctx* ctx_ = nullptr;
  xivo_layout lay_;
  xivo_cam cam_;
  unsigned flags_;
  int max_features_;
};

}  // namespace hip
}  // namespace xivo
