// oracle/_ref "extracted" driver - TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Compiles the REFERENCE'S OWN TEXT of the hot-path member functions: oracle/ref/extract_reference.py cuts
//   Estimator::UpdateJosephForm        src/estimator.cpp:1257-1288
//   Estimator::MHGating                src/update.cpp:50-116          (the whole function, bookkeeping included)
//   Estimator::FilterUpdate            src/update.cpp:120-153
//   Feature::FillJacobianBlock         src/feature.cpp:658-684
//   Estimator::ComposeMotion / ComputeMotionJacobianAt   src/estimator.cpp:598-704
//   Estimator::RK4Step                 src/rk4.cpp:35-103
//   Estimator::PrinceDormandStep       src/princedormand.cpp:85-221
//   Estimator::AbsorbError (both)      src/estimator.cpp:875-921
//   enum Index ... struct State        src/core.h:40-180 ;  enum class FeatureStatus  src/core.h:190-199 ;  struct SO3xR3  src/group.h:17-30
// out of /root/reference at build time into oracle/_ref/extracted/*.inc (git-ignored), and this file includes them as member
// functions of shim classes that declare exactly the members those bodies touch, with the reference's member types
// (F_ / G_ are Eigen::SparseMatrix, Feature::J_ is a fixed-size 2 x kFullSize matrix, src/estimator.h:467-509,
// src/feature.h:281) - so Eigen picks the same evaluation paths as in the reference build. Everything else the bodies
// need is the reference's own as well: Eigen 3.3.9, Sophus, common/alias.h, common/rodrigues.h, src/helpers.cpp (verbatim).
// kFullSize is a compile-time constant in the reference (src/core.h:87-105): one library per state size,
// -DEKF_MAX_GROUPS / -DEKF_MAX_FEATURES as the reference's own build knobs (src/CMakeLists.txt:27-28).
// What is NOT the reference's text here: the class shells (member declarations), the no-op timer / DestroyFeatures, and the
// extern "C" wrappers that copy arrays in and out.
#include "helpers.cpp"  // /root/reference/src/helpers.cpp (SO3_from_rotvec, ...)

#include <set>
#include <unordered_set>

#include "Eigen/Cholesky"
#include "Eigen/Sparse"
#include "camera_atan.h"
#include "camera_equidist.h"
#include "camera_pinhole.h"
#include "camera_radtan.h"
#include "project.h"     // project, unproject_logz (common/project.h)
#include "rodrigues.h"   // dAB_dA, dAB_dB, dA_dAu (common/rodrigues.h)

namespace xivo {

// stand-in for CameraManager (src/camera_manager.h:25-118 needs jsoncpp + glog to construct): the same dispatch onto the
// reference's own camera classes, configured by the extern "C" wrappers below
struct RefCamCfg { int model; int rows, cols; double fx, fy, cx, cy; double d[5]; };
class Camera {
 public:
  static Camera* instance() { static Camera c; return &c; }
  RefCamCfg cfg{0, 480, 640, 500, 500, 320, 240, {0, 0, 0, 0, 0}};
  int dim() const { return cfg.model == 0 ? 4 : (cfg.model == 1 ? 5 : (cfg.model == 2 ? 9 : 8)); }
  template <typename Derived>
  Eigen::Matrix<typename Derived::Scalar, 2, 1> Project(const Eigen::MatrixBase<Derived>& xc, Eigen::Matrix<typename Derived::Scalar, 2, 2>* jac = nullptr,
                                                        Eigen::Matrix<typename Derived::Scalar, 2, -1>* jacc = nullptr) const {
    const RefCamCfg& c = cfg;
    switch (c.model) {   // src/camera_manager.h:33-49
      case 0: return PinholeCamera<double>(c.rows, c.cols, c.fx, c.fy, c.cx, c.cy).Project(xc, jac, jacc);
      case 1: return ATANCamera<double>(c.rows, c.cols, c.fx, c.fy, c.cx, c.cy, c.d[0]).Project(xc, jac, jacc);
      case 2: return RadialTangentialCamera<double>(c.rows, c.cols, c.fx, c.fy, c.cx, c.cy, c.d[0], c.d[1], c.d[2], c.d[3], c.d[4]).Project(xc, jac, jacc);
      default: return EquidistantCamera<double>(c.rows, c.cols, c.fx, c.fy, c.cx, c.cy, c.d[0], c.d[1], c.d[2], c.d[3]).Project(xc, jac, jacc);
    }
  }
  // CameraManager::UpdateState (src/camera_manager.h:91-101) -> A_*Camera::UpdateState (common/camera_autocalib.h:34-47, :78-84,
  // :107-116, :145-157): every parameter += its component, in the order fx fy cx cy + the model's distortion parameters
  // (= cfg.d's order). Restated here (those classes pull in component.h / jsoncpp).
  template <class V> void UpdateState(const V& dX) {
    cfg.fx += dX(0); cfg.fy += dX(1); cfg.cx += dX(2); cfg.cy += dX(3);
    for (int k = 4; k < dim(); ++k) cfg.d[k - 4] += dX(k);
  }
};

#include "extracted/core_index_state.inc"
#include "extracted/feature_status.inc"
#include "extracted/so3xr3.inc"
#include "extracted/jacobian_cache.inc"

using Vec6 = Eigen::Matrix<number_t, 6, 1>;

class RefGroup {   // the members of Group (src/group.h:41-107) the extracted bodies touch
 public:
  int id() const { return id_; }
  int sind() const { return sind_; }
  const SO3& Rsb() const { return X_.Rsb; }        // src/group.h:66-67
  const Vec3& Tsb() const { return X_.Tsb; }
  void UpdateState(const Vec6& dX) { X_ += dX; }   // src/group.h:75
  int id_ = 0, sind_ = -1;
  SO3xR3 X_;
};
using GroupPtr = RefGroup*;

class RefFeature {   // the members of Feature (src/feature.h:74-284) the extracted bodies touch
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  const Eigen::Matrix<number_t, 2, kFullSize>& J() const { return J_; }   // src/feature.h:165
  const Vec2& inn() const { return inn_; }                                // :166
  FeatureStatus status() const { return status_; }
  void SetStatus(FeatureStatus s) { status_ = s; }
  int id() const { return id_; }
  int sind() const { return sind_; }
  GroupPtr ref() const { return ref_; }
  void UpdateState(const Vec3& dx) { x_ += dx; }                          // src/feature.h:220
  void FillJacobianBlock(MatX& H, int offset);                            // src/feature.cpp:658-684 (extracted below)
  void ComputeJacobian(const Mat3& Rsb, const Vec3& Tsb, const Mat3& Rbc, const Vec3& Tbc, const Vec3& gyro, const Mat3& Cg,
                       const Vec3& bg, const Vec3& Vsb, number_t td);    // src/feature.cpp:542-656 (extracted below)
  Vec3 Xc(Mat3* J = nullptr);                                             // src/feature.cpp:98-105 (extracted below)
  const Vec2& back() const { return back_; }                              // (the last tracked pixel, src/feature.h)
  static JacobianCache cache_;                                            // src/feature.h / feature.cpp:20: process-wide static
  Eigen::Matrix<number_t, 2, kFullSize> J_;                               // src/feature.h:281
  Vec2 inn_, back_;
  Vec3 x_, Xc_;
  FeatureStatus status_ = FeatureStatus::INSTATE;
  int id_ = 0, sind_ = -1;
  GroupPtr ref_ = nullptr;
};
using FeaturePtr = RefFeature*;

JacobianCache RefFeature::cache_;
// IMUState (src/imu.h:12-27: Ca, Cg and the 15-dimensional Tangent) with the reference's own operator+= (src/imu.cpp:7-21,
// extracted below); the IMU object around it (src/imu.h:29-46) reduced to the accessors the extracted bodies call
#ifndef CHECK
#define CHECK(c) if (!(c)) abort()
#endif
struct IMUState {
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  Mat3 Ca = Mat3::Identity(), Cg = Mat3::Identity();
  using Tangent = Eigen::Matrix<number_t, 15, 1>;
  void operator+=(const Tangent& dX);
};
#include "extracted/imu_state_plus.inc"
struct RefImu {
  IMUState X_;
  const Mat3& Ca() const { return X_.Ca; }
  const Mat3& Cg() const { return X_.Cg; }
  void UpdateState(const IMUState::Tangent& dX) { X_ += dX; }     // src/imu.h:34
};
struct RefTimer { void Tick(const char*) {} void Tock(const char*) {} };

class RefEstimator {   // the members of Estimator (src/estimator.h:387-575) the extracted bodies touch, same names and types
 public:
  EIGEN_MAKE_ALIGNED_OPERATOR_NEW
  RefEstimator() : F_(kMotionSize, kMotionSize), G_(kMotionSize, 12) {}
  void UpdateJosephForm();
  std::vector<FeaturePtr> MHGating();
  void FilterUpdate();
  void AbsorbError(const VecX& err);
  void AbsorbError();
  void ComposeMotion(State& X, const Vec3& V, const Eigen::Matrix<number_t, 6, 1>& gyro_accel, number_t dt);
  void ComputeMotionJacobianAt(const State& X, const Eigen::Matrix<number_t, 6, 1>& gyro_accel);
  void RK4Step(const Vec3& gyro0, const Vec3& accel0, number_t dt);
  number_t PrinceDormandStep(const Vec3& gyro0, const Vec3& accel0, number_t dt);
  void UpdateState(const State::Tangent& dX) { X_ += dX; }        // src/estimator.h:234
  void DestroyFeatures(const std::vector<FeaturePtr>& v) { destroyed_ = v; }   // (src/graph bookkeeping: recorded, not acted on)

  RefTimer timer_;
  State X_;
  VecX err_;
  RefImu imu_;
  Vec3 g_;
  Eigen::SparseMatrix<number_t> F_, G_;                             // src/estimator.h:467-470
  MatX P_, Qmodel_, Qimu_, S_, K_, H_, I_KH_;
  VecX inn_, diagR_;
  number_t R_ = 0, MH_thresh_ = 0, MH_thresh_multipler_ = 0;
  int min_required_inliers_ = 0, num_mh_rejected_ = 0;
  Vec3 slope_accel_, slope_gyro_;
  std::vector<FeaturePtr> instate_features_, in_current_ekf_update_, destroyed_;
  std::vector<GroupPtr> instate_groups_, needs_new_gauge_features_;
  std::unordered_set<GroupPtr> affected_groups_;
};

#define Estimator RefEstimator
#define Feature RefFeature
#include "extracted/update_joseph_form.inc"
#include "extracted/mh_gating.inc"
#include "extracted/fill_jacobian_block.inc"
#include "extracted/feature_xc.inc"
#include "extracted/compute_jacobian.inc"
#include "extracted/filter_update.inc"
#include "extracted/absorb_error_vec.inc"
#include "extracted/absorb_error.inc"
#include "extracted/compose_motion.inc"
#include "extracted/compute_motion_jacobian_at.inc"
#include "extracted/rk4_step.inc"
#include "extracted/prince_dormand_step.inc"
#undef Feature
#undef Estimator

}  // namespace xivo

using namespace xivo;
namespace {
using MapMat = Eigen::Map<const MatX>;
using MapMatW = Eigen::Map<MatX>;
using MapVec = Eigen::Map<const VecX>;
using MapVecW = Eigen::Map<VecX>;

// state30 = [Rsb(9, column-major) Tsb(3) Vsb(3) bg(3) ba(3) Rsg(9)] - the layout oracle/ref_binding.py uses for xivo_ref.cpp
void load_state(State& X, const double* s) {
  X.Rsb = SO3(Eigen::Quaterniond(Mat3(Eigen::Map<const Mat3>(s))));
  X.Tsb = Eigen::Map<const Vec3>(s + 9); X.Vsb = Eigen::Map<const Vec3>(s + 12);
  X.bg = Eigen::Map<const Vec3>(s + 15); X.ba = Eigen::Map<const Vec3>(s + 18);
  X.Rsg = SO3(Eigen::Quaterniond(Mat3(Eigen::Map<const Mat3>(s + 21))));
}
void store_state(const State& X, double* s) {
  (Eigen::Map<Mat3>(s)) = X.Rsb.matrix(); (Eigen::Map<Vec3>(s + 9)) = X.Tsb; (Eigen::Map<Vec3>(s + 12)) = X.Vsb;
  (Eigen::Map<Vec3>(s + 15)) = X.bg; (Eigen::Map<Vec3>(s + 18)) = X.ba; (Eigen::Map<Mat3>(s + 21)) = X.Rsg.matrix();
}
}  // namespace

extern "C" {

int refx_full_size(void) { return kFullSize; }
int refx_motion_size(void) { return kMotionSize; }
// slots of the online-calibration builds as the extracted enum Index numbers them (-1 / 0: not in this build)
int refx_index_td(void) {
#ifdef USE_ONLINE_TEMPORAL_CALIB
  return Index::td;
#else
  return -1;
#endif
}
int refx_index_Cg(void) {
#ifdef USE_ONLINE_IMU_CALIB
  return Index::Cg;
#else
  return -1;
#endif
}
int refx_camera_begin(void) { return kCameraBegin; }
int refx_max_camera_intrinsics(void) { return kMaxCameraIntrinsics; }

// Feature::ComputeJacobian + Feature::FillJacobianBlock as extracted, in whatever build this library is (default, or the
// online-calibration defines). 3 x 3 inputs column-major. Outputs: J (2 x kFullSize column-major), inn (2), and the two rows
// FillJacobianBlock stacks from it (Hrow, 2 x kFullSize column-major).
void refx_compute_jacobian(const double* x, const double* xp_meas, const double* Rsbr, const double* Tsbr, const double* Rsb,
                           const double* Tsb, const double* Rbc, const double* Tbc, const double* gyro, const double* Cg,
                           const double* bg, const double* Vsb, double td, const RefCamCfg* cam, int ref_sind, int sind,
                           double* J_out, double* inn_out, double* Hrow_out) {
  Camera::instance()->cfg = *cam;
  RefGroup g; g.sind_ = ref_sind;
  g.X_.Rsb = SO3(Eigen::Quaterniond(Mat3(Eigen::Map<const Mat3>(Rsbr)))); g.X_.Tsb = Eigen::Map<const Vec3>(Tsbr);
  RefFeature f; f.ref_ = &g; f.sind_ = sind;
  f.x_ = Eigen::Map<const Vec3>(x); f.back_ = Eigen::Map<const Vec2>(xp_meas);
  const Mat3 Rsb_ = Eigen::Map<const Mat3>(Rsb), Rbc_ = Eigen::Map<const Mat3>(Rbc), Cg_ = Eigen::Map<const Mat3>(Cg);
  f.ComputeJacobian(Rsb_, Eigen::Map<const Vec3>(Tsb), Rbc_, Eigen::Map<const Vec3>(Tbc), Eigen::Map<const Vec3>(gyro), Cg_,
                    Eigen::Map<const Vec3>(bg), Eigen::Map<const Vec3>(Vsb), td);
  (Eigen::Map<Eigen::Matrix<number_t, 2, kFullSize>>(J_out)) = f.J();
  inn_out[0] = f.inn()(0); inn_out[1] = f.inn()(1);
  MatX H = MatX::Zero(2, kFullSize);
  f.FillJacobianBlock(H, 0);
  (MapMatW(Hrow_out, 2, kFullSize)) = H;
}
int refx_group_begin(void) { return kGroupBegin; }
int refx_feature_begin(void) { return kFeatureBegin; }

// Estimator::UpdateJosephForm as extracted; any N (the members are dynamic)
void refx_update_joseph(int N, int M, const double* H, const double* P, const double* inn, const double* diagR,
                        double* err_out, double* P_out) {
  RefEstimator e;
  e.H_ = MapMat(H, M, N); e.P_ = MapMat(P, N, N); e.inn_ = MapVec(inn, M); e.diagR_ = MapVec(diagR, M); e.err_ = VecX::Zero(N);
  e.UpdateJosephForm();
  (MapVecW(err_out, N)) = e.err_;
  (MapMatW(P_out, N, N)) = e.P_;
}

// Estimator::MHGating as extracted (N = kFullSize). J: F blocks of 2 x N column-major. status_io: FeatureStatus per feature
// (3 = INSTATE, 7 = GAUGE in, 4 = REJECTED_BY_FILTER out). Returns the number of inliers; inlier_idx lists them in order.
int refx_mh_gating(int F, const double* J, const double* inn, const double* P, double R, double thresh, double mult, int min_inliers,
                   int* status_io, int* inlier_idx, int* num_rejected, int* n_destroyed) {
  const int N = kFullSize;
  RefEstimator e;
  e.P_ = MapMat(P, N, N); e.R_ = R; e.MH_thresh_ = thresh; e.MH_thresh_multipler_ = mult; e.min_required_inliers_ = min_inliers;
  std::vector<RefGroup> gs(1);
  std::vector<RefFeature, Eigen::aligned_allocator<RefFeature>> fs(F);
  for (int f = 0; f < F; ++f) {
    fs[f].J_ = Eigen::Map<const Eigen::Matrix<number_t, 2, kFullSize>>(J + (size_t)f * 2 * N);
    fs[f].inn_ = Vec2(inn[2 * f], inn[2 * f + 1]); fs[f].id_ = f; fs[f].ref_ = &gs[0];
    fs[f].status_ = static_cast<FeatureStatus>(status_io[f]);
    e.instate_features_.push_back(&fs[f]);
  }
  std::vector<FeaturePtr> inl = e.MHGating();
  for (size_t i = 0; i < inl.size(); ++i) inlier_idx[i] = inl[i]->id();
  for (int f = 0; f < F; ++f) status_io[f] = static_cast<int>(fs[f].status());
  *num_rejected = e.num_mh_rejected_; *n_destroyed = (int)e.destroyed_.size();
  return (int)inl.size();
}

// Estimator::FilterUpdate as extracted (FillJacobianBlock incl. the :675-676 overwrite, UpdateJosephForm, AbsorbError) for the
// motion state + features: J blocks as above, ref_sind / sind per feature; state30 in-out, x (3 per feature) in-out,
// P in-out; H_out (2F x N column-major, may be null), err_before_absorb (N).
void refx_filter_update(int F, const double* J, const double* inn, const int* ref_sind, const int* sind, double R, double* P_io,
                        double* state30_io, const double* Rbc, const double* Tbc, double* x_io, double* H_out, double* err_before_absorb) {
  const int N = kFullSize;
  RefEstimator e;
  e.P_ = MapMat(P_io, N, N); e.R_ = R; e.err_ = VecX::Zero(N);
  load_state(e.X_, state30_io);
  e.X_.Rbc = SO3(Eigen::Quaterniond(Mat3(Eigen::Map<const Mat3>(Rbc)))); e.X_.Tbc = Eigen::Map<const Vec3>(Tbc);
  std::vector<RefGroup> gs(kMaxGroup);
  for (int g = 0; g < kMaxGroup; ++g) gs[g].sind_ = g;
  std::vector<RefFeature, Eigen::aligned_allocator<RefFeature>> fs(F);
  for (int f = 0; f < F; ++f) {
    fs[f].J_ = Eigen::Map<const Eigen::Matrix<number_t, 2, kFullSize>>(J + (size_t)f * 2 * N);
    fs[f].inn_ = Vec2(inn[2 * f], inn[2 * f + 1]); fs[f].ref_ = &gs[ref_sind[f]]; fs[f].sind_ = sind[f];
    fs[f].x_ = Eigen::Map<const Vec3>(x_io + 3 * f);
    e.in_current_ekf_update_.push_back(&fs[f]);
  }
  // FilterUpdate = stacking + UpdateJosephForm + AbsorbError; err_ is zeroed by AbsorbError, so the stacking + update are run
  // first on a copy to report H_ and err_ before the absorb, then the real thing
  {
    RefEstimator c = e;
    const int total = 2 * F;
    c.H_.setZero(total, N); c.inn_.setZero(total); c.diagR_.resize(total);
    for (int i = 0; i < F; ++i) { c.in_current_ekf_update_[i]->FillJacobianBlock(c.H_, 2 * i); c.inn_.segment<2>(2 * i) = fs[i].inn(); c.diagR_.segment<2>(2 * i) << R, R; }
    c.UpdateJosephForm();
    if (H_out) (MapMatW(H_out, total, N)) = c.H_;
    (MapVecW(err_before_absorb, N)) = c.err_;
  }
  e.FilterUpdate();
  (MapMatW(P_io, N, N)) = e.P_;
  store_state(e.X_, state30_io);
  for (int f = 0; f < F; ++f) (Eigen::Map<Vec3>(x_io + 3 * f)) = fs[f].x_;
}

// Estimator::RK4Step / PrinceDormandStep as extracted (with ComposeMotion, ComputeMotionJacobianAt), N = kFullSize
// Cg / Ca: imu_.Cg() / imu_.Ca(), column-major, may be null (identity)
void refx_integrator_step_calib(int use_rk4, double* state30_io, double* P_io, const double* gyro0, const double* accel0,
                                const double* slope_gyro, const double* slope_accel, double dt, const double* Qimu, const double* g,
                                const double* Cg, const double* Ca) {
  const int N = kFullSize;
  RefEstimator e;
  load_state(e.X_, state30_io);
  if (Cg) e.imu_.X_.Cg = Eigen::Map<const Mat3>(Cg);
  if (Ca) e.imu_.X_.Ca = Eigen::Map<const Mat3>(Ca);
  e.P_ = MapMat(P_io, N, N); e.Qimu_ = MapMat(Qimu, 12, 12); e.g_ = Eigen::Map<const Vec3>(g);
  e.slope_gyro_ = Eigen::Map<const Vec3>(slope_gyro); e.slope_accel_ = Eigen::Map<const Vec3>(slope_accel);
  const Vec3 gy = Eigen::Map<const Vec3>(gyro0), ac = Eigen::Map<const Vec3>(accel0);
  if (use_rk4) e.RK4Step(gy, ac, dt); else e.PrinceDormandStep(gy, ac, dt);
  store_state(e.X_, state30_io);
  (MapMatW(P_io, N, N)) = e.P_;
}
void refx_integrator_step(int use_rk4, double* state30_io, double* P_io, const double* gyro0, const double* accel0,
                          const double* slope_gyro, const double* slope_accel, double dt, const double* Qimu, const double* g) {
  refx_integrator_step_calib(use_rk4, state30_io, P_io, gyro0, accel0, slope_gyro, slope_accel, dt, Qimu, g, nullptr, nullptr);
}

int refx_index_Ca(void) {
#ifdef USE_ONLINE_IMU_CALIB
  return Index::Ca;
#else
  return -1;
#endif
}

// Estimator::AbsorbError(err) as extracted, motion part only (no groups / features in the lists): State::operator+= incl. td,
// IMUState::operator+= (as extracted) for Ca / Cg, the camera intrinsics. td / Cg / Ca (column-major) / cam are in-out.
void refx_absorb_motion_calib(double* state30_io, const double* Rbc_in, const double* Tbc_in, double* Rbc_out, double* Tbc_out,
                              double* td_io, double* Cg_io, double* Ca_io, RefCamCfg* cam_io, const double* err) {
  RefEstimator e;
  load_state(e.X_, state30_io);
  e.X_.Rbc = SO3(Eigen::Quaterniond(Mat3(Eigen::Map<const Mat3>(Rbc_in)))); e.X_.Tbc = Eigen::Map<const Vec3>(Tbc_in);
  e.X_.td = *td_io;
  e.imu_.X_.Cg = Eigen::Map<const Mat3>(Cg_io); e.imu_.X_.Ca = Eigen::Map<const Mat3>(Ca_io);
  Camera::instance()->cfg = *cam_io;
  const VecX ev = MapVec(err, kFullSize);
  e.AbsorbError(ev);
  store_state(e.X_, state30_io);
  (Eigen::Map<Mat3>(Rbc_out)) = e.X_.Rbc.matrix(); (Eigen::Map<Vec3>(Tbc_out)) = e.X_.Tbc;
  *td_io = e.X_.td;
  (Eigen::Map<Mat3>(Cg_io)) = e.imu_.Cg(); (Eigen::Map<Mat3>(Ca_io)) = e.imu_.Ca();
  *cam_io = Camera::instance()->cfg;
}

}  // extern "C"
