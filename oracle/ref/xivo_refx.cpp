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

#include <chrono>
#include <map>
#include <memory>
#include <random>
#include <set>
#include <string>
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
  // CameraManager::BackupState / RestoreState (src/camera_manager.h:148-175 -> common/camera_autocalib.h): the parameters
  RefCamCfg cfg0{};
  void BackupState() { cfg0 = cfg; }
  void RestoreState() { cfg = cfg0; }
};

#include "extracted/core_index_state.inc"
#include "extracted/feature_status.inc"
#include "extracted/so3xr3.inc"
#include "extracted/jacobian_cache.inc"
#include "extracted/oos_jacobian_struct.inc"
#include "extracted/subfilter_options.inc"

using Vec6 = Eigen::Matrix<number_t, 6, 1>;

class RefGroup {   // the members of Group (src/group.h:41-107) the extracted bodies touch
 public:
  int id() const { return id_; }
  int sind() const { return sind_; }
  const SO3& Rsb() const { return X_.Rsb; }        // src/group.h:66-67
  const Vec3& Tsb() const { return X_.Tsb; }
  void UpdateState(const Vec6& dX) { X_ += dX; }   // src/group.h:75
  SE3 gsb() const { return SE3{X_.Rsb, X_.Tsb}; }   // src/group.h:67
  bool instate() const { return instate_; }        // (src/group.h: status_ == GroupStatus::INSTATE || GAUGE)
  void BackupState() { X0_ = X_; }                 // src/group.h:58-59
  void RestoreState() { X_ = X0_; }
  int id_ = 0, sind_ = -1;
  bool instate_ = true;
  SO3xR3 X_, X0_;
};
using GroupPtr = RefGroup*;
#include "extracted/observation.inc"
using Obs = Observation;   // src/core.h:235
class RefEstimator;

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
  Vec3 Xs(const SE3& gbc, Mat3* J = nullptr);                             // src/feature.cpp:107-118 (extracted below)
  int ComputeOOSJacobian(const std::vector<Observation>& vobs, const Mat3& Rbc, const Vec3& Tbc);       // src/oos.cpp:8-37
  void ComputeOOSJacobianInternal(const Observation& obs, const Mat3& Rbc, const Vec3& Tbc);            // src/oos.cpp:39-89
  void ComputeLCJacobian(const Obs& obs, const Mat3& Rbc, const Vec3& Tbc, int match_counter, MatX& H, VecX& inn);   // src/oos.cpp:92-145
  void SubfilterUpdate(const SE3& gsb, const SE3& gbc, const SubfilterOptions& options);                // src/feature.cpp:246-297
  const Vec2& xp() const { return back(); }                               // src/feature.h:169
  const Vec2& Predict(const SE3& gsb, const SE3& gbc) {                   // src/feature.h:175-179 (header-inline; restated)
    Vec3 Xc = (gsb * gbc).inverse() * this->Xs(gbc);
    pred_ = Camera::instance()->Project(project(Xc));
    return pred_;
  }
  void BackupState() { x0_ = x_; }                                        // src/feature.h:108-109
  void RestoreState() { x_ = x0_; }
  OOSJacobian oos_;                                                       // src/feature.h (oos_, oos_jac_counter_)
  int oos_jac_counter_ = 0;
  Mat3 P_;                                                                // the feature's own 3 x 3 covariance (sub-filter)
  int init_counter_ = 0;
  number_t outlier_counter_ = 0;
  Vec3 Xs_, x0_;
  Vec2 pred_;
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
  IMUState X_backup_;
  void BackupState() { X_backup_ = X_; }                          // src/imu.h:35-36
  void RestoreState() { X_ = X_backup_; }
};
// stand-in for the Json::Value lookups of Estimator::RK4 / PrinceDormand (cfg_["RK4"].get("stepsize", 0.002).asDouble(),
// src/rk4.cpp:9, src/princedormand.cpp:17-24): a two-level map of numbers; a missing key yields the caller's default
struct RefJson {
  std::map<std::string, RefJson> kids;
  double num = 0;
  RefJson operator[](const char* k) const { auto it = kids.find(k); return it == kids.end() ? RefJson{} : it->second; }
  RefJson get(const char* k, double dflt) const { auto it = kids.find(k); RefJson r; r.num = it == kids.end() ? dflt : it->second.num; return r; }
  double asDouble() const { return num; }
  bool asBool() const { return num != 0; }
  int asInt() const { return (int)num; }
};
template <typename... Args> std::string StrFormat(const char* format, Args... args) {   // common/utils.h:299-306 (needs OpenCV there)
  char buf[512];
  snprintf(buf, 512, format, args...);
  return buf;
}
using timestamp_t = std::chrono::nanoseconds;   // src/core.h:30
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
  // round 5: integrator outer loops, Propagate, 1-pt RANSAC and what they call
  void RK4(const Vec3& gyro0, const Vec3& accel0, number_t dt);                                   // src/rk4.cpp:5-33
  void PrinceDormand(const Vec3& gyro0, const Vec3& accel0, number_t dt);                         // src/princedormand.cpp:7-83
  void Fehlberg(const Vec3&, const Vec3&, number_t) { throw std::runtime_error("NotImplemented"); }   // src/estimator.cpp:594-596
  void Propagate(bool visual_meas);                                                               // src/estimator.cpp:539-592
  std::vector<FeaturePtr> OnePointRANSAC(const std::vector<FeaturePtr>& mh_inliers);              // src/update.cpp:213-393
  GroupPtr FindNewRefGroup(std::vector<GroupPtr>& candidates);                                    // src/estimator.cpp:1394-1407
  void BackupState(std::unordered_set<FeaturePtr>& features, std::unordered_set<GroupPtr>& groups);   // :1410-1428
  void RestoreState(std::unordered_set<FeaturePtr>& features, std::unordered_set<GroupPtr>& groups);  // :1431-1449
  SE3 gbc() const { return SE3{X_.Rbc, X_.Tbc}; }                 // src/estimator.h:153-154
  SE3 gsb() const { return SE3{X_.Rsb, X_.Tsb}; }
  static RefEstimator* instance() { static RefEstimator* e = new RefEstimator; return e; }   // (only OOS_update_min_observations is read)
  int OOS_update_min_observations() const { return OOS_update_min_observations_; }
  int OOS_update_min_observations_ = 5;                            // src/estimator.cpp:118-119
  RefJson cfg_;
  timestamp_t curr_time_{0}, last_time_{0};
  bool simulation_ = true;
  Vec3 curr_accel_, curr_gyro_, last_accel_, last_gyro_;
  std::string integration_method_ = "RK4";
  // (std::unique_ptr in the reference, src/estimator.h:575; shared here so that the shell stays copyable; default seed as src/estimator.cpp:410-411)
  std::shared_ptr<std::default_random_engine> rng_{new std::default_random_engine};
  number_t ransac_thresh_ = 0, ransac_prob_ = 0.99, ransac_Chi2_ = 0;
  GroupPtr gauge_group_ptr_ = nullptr;
  State X0_;
  MatX P0_;
  int num_oneptransac_rejected_ = 0;

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
#include "extracted/feature_xs.inc"
#include "extracted/compute_oos_jacobian.inc"
#include "extracted/compute_oos_jacobian_internal.inc"
#include "extracted/compute_lc_jacobian.inc"
#include "extracted/subfilter_update.inc"
#include "extracted/rk4.inc"
#include "extracted/prince_dormand.inc"
#include "extracted/propagate.inc"
std::vector<FeaturePtr>      // (the return type of the definition below sits on its own line in the reference, src/update.cpp:213)
#include "extracted/one_point_ransac.inc"
#include "extracted/find_new_ref_group.inc"
#include "extracted/backup_state.inc"
#include "extracted/restore_state.inc"
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


// ---- round 5: the remaining rows of the path, all on the reference's own text -------------------------------------------
namespace {
SO3 so3_of(const double* R9) { return SO3(Eigen::Quaterniond(Mat3(Eigen::Map<const Mat3>(R9)))); }
void set_group(RefGroup& g, const double* R9, const double* T3, int sind) {
  g.X_.Rsb = so3_of(R9); g.X_.Tsb = Eigen::Map<const Vec3>(T3); g.sind_ = sind;
}
double g_stepsize_fixed = -12345.0;   // Estimator::RK4 / PrinceDormand read their step size ONCE per process into a function-local static
std::map<std::string, double> g_pd_cfg;    // cfg_["PrinceDormand"] beyond "stepsize" (refx_pd_control)
}  // namespace

int refx_max_group(void) { return kMaxGroup; }
int refx_use_invdepth(void) {
#ifdef USE_INVDEPTH
  return 1;
#else
  return 0;
#endif
}

// Feature::Xs(gbc) (src/feature.cpp:107-118, with Feature::Xc :98-105) as extracted
void refx_feature_xs(const double* x, const double* Rsbr, const double* Tsbr, const double* Rbc, const double* Tbc, double* Xs_out) {
  RefGroup g; set_group(g, Rsbr, Tsbr, 0);
  RefFeature f; f.ref_ = &g; f.x_ = Eigen::Map<const Vec3>(x);
  (Eigen::Map<Vec3>(Xs_out)) = f.Xs(SE3(so3_of(Rbc), Eigen::Map<const Vec3>(Tbc)));
}

// Feature::ComputeOOSJacobian + ComputeOOSJacobianInternal (src/oos.cpp:8-89) as extracted - INCLUDING the reference's use of
// the whole 2 * kMaxGroup-row buffers in SlowGivens (:28): rows beyond the 2k live ones are zero (a fresh Feature's oos_), so
// the result carries 2 * kMaxGroup - 2k extra kernel columns that are unit vectors, i.e. exactly-zero rows of Hx / inn next
// to the rows this repo produces from the live rows only (tests/test_oracle_pinned.py compares them).
// obs_*: n_obs observations (group pose, slot, in-state flag, pixel). Returns oos_jac_counter_ (= rows of Hx_out; 0 when
// fewer than min_obs observations come from in-state groups). Hx_out: rows x kFullSize column-major (ld = rows), capacity
// 2 * kMaxGroup rows; inn_out likewise; Xs_out = cache_.Xs.
int refx_compute_oos_jacobian(const double* x, const double* Rsbr, const double* Tsbr, int n_obs, const double* obs_R, const double* obs_T,
                              const int* obs_sind, const int* obs_instate, const double* obs_xp, const double* Rbc, const double* Tbc,
                              const RefCamCfg* cam, int min_obs, double* Xs_out, double* Hx_out, double* inn_out) {
  Camera::instance()->cfg = *cam;
  RefEstimator::instance()->OOS_update_min_observations_ = min_obs;
  RefGroup gref; set_group(gref, Rsbr, Tsbr, 0);
  std::vector<RefGroup> gs(n_obs);
  std::vector<Observation, Eigen::aligned_allocator<Observation>> vobs_a(n_obs);
  for (int i = 0; i < n_obs; ++i) {
    set_group(gs[i], obs_R + 9 * i, obs_T + 3 * i, obs_sind[i]); gs[i].instate_ = obs_instate[i] != 0;
    vobs_a[i].g = &gs[i]; vobs_a[i].xp = Eigen::Map<const Vec2>(obs_xp + 2 * i);
  }
  std::vector<Observation> vobs(vobs_a.begin(), vobs_a.end());
  RefFeature f; f.ref_ = &gref; f.x_ = Eigen::Map<const Vec3>(x);
  const Mat3 Rbc_ = Eigen::Map<const Mat3>(Rbc);
  const int rows = f.ComputeOOSJacobian(vobs, Rbc_, Eigen::Map<const Vec3>(Tbc));
  (Eigen::Map<Vec3>(Xs_out)) = RefFeature::cache_.Xs;
  if (rows > 0) {
    (MapMatW(Hx_out, f.oos_.Hx.rows(), kFullSize)) = f.oos_.Hx;
    (MapVecW(inn_out, f.oos_.inn.size())) = f.oos_.inn;
  }
  return rows;
}

// Feature::ComputeLCJacobian (src/oos.cpp:92-145) as extracted, driven as Estimator::CloseLoopInternal drives it
// (src/update.cpp:183-196): H_.setZero(2n, N), inn_.setZero(2n), one call per match. Per match: the OLD feature's state x and
// the pose of ITS reference group, the group that observes the new feature (pose, slot) and the observed pixel.
void refx_compute_lc_jacobian(int n, const double* x, const double* Rsbr, const double* Tsbr, const double* obs_R, const double* obs_T,
                              const int* obs_sind, const double* obs_xp, const double* Rbc, const double* Tbc, const RefCamCfg* cam,
                              double* H_out, double* inn_out) {
  Camera::instance()->cfg = *cam;
  MatX H_; VecX inn_;
  H_.setZero(2 * n, kFullSize); inn_.setZero(2 * n);
  const Mat3 Rbc_ = Eigen::Map<const Mat3>(Rbc);
  for (int i = 0; i < n; ++i) {
    RefGroup gref, gobs; set_group(gref, Rsbr + 9 * i, Tsbr + 3 * i, 0); set_group(gobs, obs_R + 9 * i, obs_T + 3 * i, obs_sind[i]);
    RefFeature old_feature; old_feature.ref_ = &gref; old_feature.x_ = Eigen::Map<const Vec3>(x + 3 * i);
    Observation obs; obs.g = &gobs; obs.xp = Eigen::Map<const Vec2>(obs_xp + 2 * i);
    old_feature.ComputeLCJacobian(obs, Rbc_, Eigen::Map<const Vec3>(Tbc), i, H_, inn_);
  }
  (MapMatW(H_out, 2 * n, kFullSize)) = H_;
  (MapVecW(inn_out, 2 * n)) = inn_;
}

// Feature::SubfilterUpdate (src/feature.cpp:246-297) as extracted; same argument order as the retyped ref_subfilter_update.
// Returns 1 when the feature leaves READY, 0 INITIALIZING.
int refx_subfilter_update(double* x, double* P, const double* xp_meas, const double* Rsb, const double* Tsb, const double* Rbc,
                          const double* Tbc, const double* Rsbr, const double* Tsbr, const RefCamCfg* cam, double Rtri, double MH_thresh,
                          int ready_steps, int* init_counter, double* outlier_counter) {
  Camera::instance()->cfg = *cam;
  RefGroup gref; set_group(gref, Rsbr, Tsbr, 0);
  RefFeature f; f.ref_ = &gref; f.x_ = Eigen::Map<const Vec3>(x); f.P_ = Eigen::Map<const Mat3>(P);
  f.back_ = Eigen::Map<const Vec2>(xp_meas); f.init_counter_ = *init_counter; f.outlier_counter_ = *outlier_counter;
  f.status_ = FeatureStatus::INITIALIZING;
  SubfilterOptions opt; opt.Rtri = Rtri; opt.MH_thresh = MH_thresh; opt.ready_steps = ready_steps;
  f.SubfilterUpdate(SE3(so3_of(Rsb), Eigen::Map<const Vec3>(Tsb)), SE3(so3_of(Rbc), Eigen::Map<const Vec3>(Tbc)), opt);
  (Eigen::Map<Vec3>(x)) = f.x_; (Eigen::Map<Mat3>(P)) = f.P_;
  *init_counter = f.init_counter_; *outlier_counter = f.outlier_counter_;
  return f.status() == FeatureStatus::READY ? 1 : 0;
}

// Estimator::Propagate (src/estimator.cpp:539-592) as extracted, with Estimator::RK4 (src/rk4.cpp:5-33) / PrinceDormand
// (src/princedormand.cpp:7-83) and their steps as extracted. dt = curr_time_ - last_time_ = dt_ns nanoseconds. The integrators
// keep their step size in a function-local static that is read once per process: returns -1 if this process already fixed
// another one. visual_meas = 0: slopes from (curr - last) / dt (:559-568); 1: slopes as handed in (:569-575).
// The keys of cfg_["PrinceDormand"] that switch on the step-size-controlled branch of Estimator::PrinceDormand
// (src/princedormand.cpp:17-22, :26-60). The extracted function reads them ONCE per process into function-local statics (and
// keeps its current step `h` in another one): call this before the first refx_propagate of a process - the tests load a
// private copy of this library for it.
void refx_pd_control(int control_stepsize, double tolerance, int attempts, double min_scale_factor, double max_scale_factor) {
  g_pd_cfg["control_stepsize"] = control_stepsize; g_pd_cfg["tolerance"] = tolerance; g_pd_cfg["attempts"] = attempts;
  g_pd_cfg["min_scale_factor"] = min_scale_factor; g_pd_cfg["max_scale_factor"] = max_scale_factor;
}

int refx_propagate(int method, int visual_meas, double* state30_io, double* P_io, double* last_gyro_io, double* last_accel_io,
                   const double* curr_gyro, const double* curr_accel, double* slope_gyro_io, double* slope_accel_io, long long dt_ns,
                   const double* Qimu, const double* Qmodel, const double* g, const double* Cg, const double* Ca, double stepsize) {
  if (g_stepsize_fixed != -12345.0 && g_stepsize_fixed != stepsize) return -1;
  g_stepsize_fixed = stepsize;
  const int N = kFullSize;
  RefEstimator e;
  RefJson ss; ss.num = stepsize;
  e.cfg_.kids["RK4"].kids["stepsize"] = ss; e.cfg_.kids["PrinceDormand"].kids["stepsize"] = ss;
  // cfg_["PrinceDormand"]["control_stepsize"] etc. (src/princedormand.cpp:17-22), set once per process by refx_pd_control below
  for (const auto& kv : g_pd_cfg) { RefJson v; v.num = kv.second; e.cfg_.kids["PrinceDormand"].kids[kv.first] = v; }
  e.integration_method_ = method == 0 ? "RK4" : "PrinceDormand";
  load_state(e.X_, state30_io);
  if (Cg) e.imu_.X_.Cg = Eigen::Map<const Mat3>(Cg);
  if (Ca) e.imu_.X_.Ca = Eigen::Map<const Mat3>(Ca);
  e.P_ = MapMat(P_io, N, N); e.Qimu_ = MapMat(Qimu, 12, 12); e.Qmodel_ = MapMat(Qmodel, kMotionSize, kMotionSize);
  e.g_ = Eigen::Map<const Vec3>(g);
  e.last_gyro_ = Eigen::Map<const Vec3>(last_gyro_io); e.last_accel_ = Eigen::Map<const Vec3>(last_accel_io);
  e.curr_gyro_ = Eigen::Map<const Vec3>(curr_gyro); e.curr_accel_ = Eigen::Map<const Vec3>(curr_accel);
  e.slope_gyro_ = Eigen::Map<const Vec3>(slope_gyro_io); e.slope_accel_ = Eigen::Map<const Vec3>(slope_accel_io);
  e.last_time_ = timestamp_t(0); e.curr_time_ = timestamp_t(dt_ns);
  e.Propagate(visual_meas != 0);
  store_state(e.X_, state30_io);
  (MapMatW(P_io, N, N)) = e.P_;
  (Eigen::Map<Vec3>(last_gyro_io)) = e.last_gyro_; (Eigen::Map<Vec3>(last_accel_io)) = e.last_accel_;
  (Eigen::Map<Vec3>(slope_gyro_io)) = e.slope_gyro_; (Eigen::Map<Vec3>(slope_accel_io)) = e.slope_accel_;
  return 0;
}

// Estimator::OnePointRANSAC (src/update.cpp:213-393) as extracted - the WHOLE function: hypothesis loop on rng_, BackupState,
// FindNewRefGroup, zeroing of P_, partial UpdateJosephForm + AbsorbError, Jacobians at the updated state, chi-square rescue,
// RestoreState, Jacobians at the original state (all of them extracted). All F features are the MH inliers handed in; their
// J_ / inn_ are computed first by Feature::ComputeJacobian as ComputeInstateJacobians does. groups: kMaxGroup poses by slot.
//   gauge_sind: slot of gauge_group_ptr_ (-1: none); absorb_groups: bit g = slot g in instate_groups_ ; in_update[F]: feature is in
//   in_current_ekf_update_ (both lists are the previous frame's when AbsorbError runs inside, src/manager.cpp:103)
// Outputs: keep[F] 1 = in the returned inlier set; status_io (FeatureStatus per feature); *n_rejected = num_oneptransac_rejected_;
// P_io / state30_io / x_io as the function leaves them (= restored); J_after [F][2 x N] = J_ after the final re-computation.
int refx_one_point_ransac(int F, double* x_io, const double* xp, const int* ref_sind, const int* sind, int* status_io, const double* gR,
                          const double* gT, double* state30_io, const double* Rbc, const double* Tbc, double* P_io, double R,
                          double ransac_thresh, double ransac_prob, double ransac_chi2, int gauge_sind, unsigned long long absorb_groups,
                          const int* in_update, const RefCamCfg* cam, int* keep, int* n_rejected, double* J_after) {
  const int N = kFullSize;
  Camera::instance()->cfg = *cam;
  RefEstimator e;
  e.P_ = MapMat(P_io, N, N); e.R_ = R; e.err_ = VecX::Zero(N);
  e.ransac_thresh_ = ransac_thresh; e.ransac_prob_ = ransac_prob; e.ransac_Chi2_ = ransac_chi2;
  load_state(e.X_, state30_io);
  e.X_.Rbc = so3_of(Rbc); e.X_.Tbc = Eigen::Map<const Vec3>(Tbc);
  e.last_gyro_.setZero();
  std::vector<RefGroup> gs(kMaxGroup);
  for (int g = 0; g < kMaxGroup; ++g) { set_group(gs[g], gR + 9 * g, gT + 3 * g, g); gs[g].id_ = g; }
  for (int g = 0; g < kMaxGroup; ++g) if ((absorb_groups >> g) & 1ull) e.instate_groups_.push_back(&gs[g]);
  e.gauge_group_ptr_ = gauge_sind >= 0 ? &gs[gauge_sind] : nullptr;
  std::vector<RefFeature, Eigen::aligned_allocator<RefFeature>> fs(F);
  std::vector<FeaturePtr> mh;
  for (int f = 0; f < F; ++f) {
    fs[f].x_ = Eigen::Map<const Vec3>(x_io + 3 * f); fs[f].back_ = Eigen::Map<const Vec2>(xp + 2 * f);
    fs[f].ref_ = &gs[ref_sind[f]]; fs[f].sind_ = sind[f]; fs[f].id_ = f; fs[f].status_ = static_cast<FeatureStatus>(status_io[f]);
    fs[f].ComputeJacobian(e.X_.Rsb.matrix(), e.X_.Tsb, e.X_.Rbc.matrix(), e.X_.Tbc, e.last_gyro_, e.imu_.Cg(), e.X_.bg, e.X_.Vsb, e.X_.td);
    mh.push_back(&fs[f]);
    if (in_update && in_update[f]) e.in_current_ekf_update_.push_back(&fs[f]);
  }
  std::vector<FeaturePtr> out = e.OnePointRANSAC(mh);
  for (int f = 0; f < F; ++f) keep[f] = 0;
  for (FeaturePtr p : out) keep[p->id()] = 1;
  for (int f = 0; f < F; ++f) {
    status_io[f] = static_cast<int>(fs[f].status());
    (Eigen::Map<Vec3>(x_io + 3 * f)) = fs[f].x_;
    if (J_after) (Eigen::Map<Eigen::Matrix<number_t, 2, kFullSize>>(J_after + (size_t)f * 2 * N)) = fs[f].J();
  }
  *n_rejected = e.num_oneptransac_rejected_;
  (MapMatW(P_io, N, N)) = e.P_;
  store_state(e.X_, state30_io);
  return (int)out.size();
}

}  // extern "C"
