// oracle/_ref driver - TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Built by oracle/ref/Makefile against the reference's OWN sources where they
// lie under /root/reference: vendored Eigen 3.3.9 (LDLT, LLT, FullPivLU, dense
// products), Sophus (SO3::hat/exp), src/helpers.cpp (included verbatim, the way
// the reference's own unit test does - src/test/unittest_givens.cpp:1 - so the
// file-static givens() is reachable), common/project.h and the four
// common/camera_*.h models. The Estimator / Feature member functions themselves
// cannot be compiled here (their TUs pull OpenCV via common/utils.h:17), so each
// is re-stated below against the same Eigen types with the reference's exact
// expression sequence, citing the lines it follows.
#include "helpers.cpp"  // /root/reference/src/helpers.cpp

#include "Eigen/Cholesky"
#include "Eigen/LU"
#include "camera_atan.h"
#include "camera_equidist.h"
#include "camera_pinhole.h"
#include "camera_radtan.h"
#include "project.h"

using namespace xivo;

namespace {
using MapMat = Eigen::Map<const MatX>;
using MapMatW = Eigen::Map<MatX>;
using MapVec = Eigen::Map<const VecX>;
using MapVecW = Eigen::Map<VecX>;

struct CamCfg { int model; int rows, cols; double fx, fy, cx, cy; double d[5]; };

Vec2 cam_project(const CamCfg& c, const Vec2& xc, Mat2* jac) {
  switch (c.model) {
    case 0: return PinholeCamera<double>(c.rows, c.cols, c.fx, c.fy, c.cx, c.cy).Project(xc, jac);
    case 1: return ATANCamera<double>(c.rows, c.cols, c.fx, c.fy, c.cx, c.cy, c.d[0]).Project(xc, jac);
    case 2: return RadialTangentialCamera<double>(c.rows, c.cols, c.fx, c.fy, c.cx, c.cy, c.d[0], c.d[1], c.d[2], c.d[3], c.d[4]).Project(xc, jac);
    default: return EquidistantCamera<double>(c.rows, c.cols, c.fx, c.fy, c.cx, c.cy, c.d[0], c.d[1], c.d[2], c.d[3]).Project(xc, jac);
  }
}
}  // namespace

extern "C" {

// Estimator::UpdateJosephForm, src/estimator.cpp:1257-1288 (members -> locals)
void ref_update_joseph(int N, int M, const double* H, const double* P, const double* inn, const double* diagR,
                       double* err_out, double* P_out) {
  MatX H_ = MapMat(H, M, N), P_ = MapMat(P, N, N);
  VecX inn_ = MapVec(inn, M), diagR_ = MapVec(diagR, M), err_ = VecX::Zero(N);
  MatX S_, K_, I_KH_;
  S_ = H_ * P_ * H_.transpose();
  for (int i = 0; i < diagR_.size(); ++i) S_(i, i) += diagR_(i);
  K_.setZero(err_.size(), H_.rows());
  K_.transpose() = S_.ldlt().solve(H_ * P_);
  err_ = K_ * inn_;
  I_KH_ = K_ * H_;
  for (int i = 0; i < err_.size(); ++i) I_KH_(i, i) -= 1;
  P_ = I_KH_ * P_ * I_KH_.transpose();
  int kr = K_.rows(), kc = K_.cols();
  for (int i = 0; i < kc; ++i) K_.block(0, i, kr, 1) *= sqrt(diagR_(i));
  P_.noalias() += K_ * K_.transpose();
  (MapVecW(err_out, N)) = err_;
  (MapMatW(P_out, N, N)) = P_;
}

// Estimator::MHGating distances, src/update.cpp:60-70. J: F blocks of 2 x N col-major
void ref_mh_distances(int F, int N, const double* J, const double* P, const double* inn, double R, double* dist) {
  MatX P_ = MapMat(P, N, N);
  for (int f = 0; f < F; ++f) {
    Eigen::Matrix<double, 2, Eigen::Dynamic> Jf = Eigen::Map<const Eigen::Matrix<double, 2, Eigen::Dynamic>>(J + (size_t)f * 2 * N, 2, N);
    Vec2 res(inn[2 * f], inn[2 * f + 1]);
    Mat2 S = Jf * P_ * Jf.transpose();
    S(0, 0) += R;
    S(1, 1) += R;
    dist[f] = res.dot(S.llt().solve(res));
  }
}

// Feature::ComputeJacobian, src/feature.cpp:542-656 (default build). 3x3 inputs col-major.
// Outputs: J (2 x N col-major), inn (2), cache = [Xc Xs Xcn | dXcn_d{Wsb,Tsb,Wbc,Tbc,Wsbr,Tsbr,Xs}] = 9 + 7*9
void ref_compute_jacobian(const double* x, const double* xp_meas, const double* Rsbr_, const double* Tsbr_,
                          const double* Rsb_, const double* Tsb_, const double* Rbc_, const double* Tbc_,
                          const CamCfg* cam, int N, int group_begin, int feature_begin, int ref_sind, int sind,
                          double* J_out, double* inn_out, double* cache_out) {
  Mat3 Rsb = Eigen::Map<const Mat3>(Rsb_), Rbc = Eigen::Map<const Mat3>(Rbc_), Rsbr = Eigen::Map<const Mat3>(Rsbr_);
  Vec3 Tsb = Eigen::Map<const Vec3>(Tsb_), Tbc = Eigen::Map<const Vec3>(Tbc_), Tsbr = Eigen::Map<const Vec3>(Tsbr_);
  Vec3 x_ = Eigen::Map<const Vec3>(x);
  Mat3 Rsb_t = Rsb.transpose(), Rbc_t = Rbc.transpose();
  Mat3 dXc_dx;
  Vec3 Xc = unproject_logz(x_, &dXc_dx);
  Vec3 Xbr = Rbc * Xc + Tbc;
  Vec3 Xs = Rsbr * Xbr + Tsbr;
  Vec3 Xb = Rsb_t * (Xs - Tsb);
  Vec3 Xcn = Rbc_t * (Xb - Tbc);
  Mat3 dXbr_dXc = Rbc, dXbr_dTbc = Mat3::Identity(), dXbr_dWbc = -Rbc * SO3::hat(Xc);
  Mat3 dXs_dXbr = Rsbr, dXs_dTsbr = Mat3::Identity(), dXs_dWsbr = -Rsbr * SO3::hat(Xbr);
  Mat3 dXb_dXs = Rsb_t, dXb_dTsb = -Rsb_t, dXb_dWsb = SO3::hat(Xb);
  Mat3 dXcn_dXb = Rbc_t;
  Mat3 dXcn_dTbc = -Rbc_t + dXcn_dXb * dXb_dXs * dXs_dXbr * dXbr_dTbc;
  Mat3 dXcn_dWbc = SO3::hat(Xcn) + dXcn_dXb * dXb_dXs * dXs_dXbr * dXbr_dWbc;
  Mat3 dXcn_dTsb = dXcn_dXb * dXb_dTsb;
  Mat3 dXcn_dWsb = dXcn_dXb * dXb_dWsb;
  Mat3 dXcn_dTsbr = dXcn_dXb * dXb_dXs * dXs_dTsbr;
  Mat3 dXcn_dWsbr = dXcn_dXb * dXb_dXs * dXs_dWsbr;
  Mat3 dXcn_dXs = dXcn_dXb * dXb_dXs;
  Mat3 dXcn_dx = dXcn_dXs * dXs_dXbr * dXbr_dXc * dXc_dx;
  Mat23 dxcn_dXcn;
  Vec2 xcn = project(Xcn, &dxcn_dXcn);
  Mat2 dxp_dxcn;
  Vec2 xp = cam_project(*cam, xcn, &dxp_dxcn);
  Mat23 dxp_dXcn = dxp_dxcn * dxcn_dXcn;
  Eigen::Matrix<double, 2, Eigen::Dynamic> J = Eigen::Matrix<double, 2, Eigen::Dynamic>::Zero(2, N);
  J.block<2, 3>(0, 0) = dxp_dXcn * dXcn_dWsb;
  J.block<2, 3>(0, 3) = dxp_dXcn * dXcn_dTsb;
  J.block<2, 3>(0, 15) = dxp_dXcn * dXcn_dWbc;
  J.block<2, 3>(0, 18) = dxp_dXcn * dXcn_dTbc;
  int goff = group_begin + 6 * ref_sind, foff = feature_begin + 3 * sind;
  J.block<2, 3>(0, goff) = dxp_dXcn * dXcn_dWsbr;
  J.block<2, 3>(0, goff + 3) = dxp_dXcn * dXcn_dTsbr;
  J.block<2, 3>(0, foff) = dxp_dXcn * dXcn_dx;
  Eigen::Map<Eigen::Matrix<double, 2, Eigen::Dynamic>>(J_out, 2, N) = J;
  inn_out[0] = xp_meas[0] - xp(0);
  inn_out[1] = xp_meas[1] - xp(1);
  if (cache_out) {
    (Eigen::Map<Vec3>(cache_out + 0)) = Xc; (Eigen::Map<Vec3>(cache_out + 3)) = Xs; (Eigen::Map<Vec3>(cache_out + 6)) = Xcn;
    const Mat3* ms[7] = {&dXcn_dWsb, &dXcn_dTsb, &dXcn_dWbc, &dXcn_dTbc, &dXcn_dWsbr, &dXcn_dTsbr, &dXcn_dXs};
    for (int i = 0; i < 7; ++i) Eigen::Map<Mat3>(cache_out + 9 + 9 * i) = *ms[i];
  }
}

// Feature::FillJacobianBlock, src/feature.cpp:658-684 on a col-major H (ldh rows)
void ref_fill_jacobian_block(double* H, int M, int N, int offset, const double* J, int group_begin, int feature_begin,
                             int ref_sind, int sind) {
  MapMatW Hm(H, M, N);
  Eigen::Map<const Eigen::Matrix<double, 2, Eigen::Dynamic>> J_(J, 2, N);
  Hm.block<2, 3>(offset, 0) = J_.block<2, 3>(0, 0);
  Hm.block<2, 3>(offset, 3) = J_.block<2, 3>(0, 3);
  Hm.block<2, 3>(offset, 15) = J_.block<2, 3>(0, 15);
  Hm.block<2, 3>(offset, 18) = J_.block<2, 3>(0, 18);
  int goff = group_begin + 6 * ref_sind, foff = feature_begin + 3 * sind;
  Hm.block<2, 3>(offset, goff) = J_.block<2, 3>(0, goff);
  Hm.block<2, 3>(offset, goff) = J_.block<2, 3>(0, goff + 3);
  Hm.block<2, 3>(offset, foff) = J_.block<2, 3>(0, foff);
}

// Feature::ComputeOOSJacobianInternal, src/oos.cpp:39-89 for one observation.
void ref_oos_internal(const double* Xs_, const double* Rsb_, const double* Tsb_, const double* Rbc_, const double* Tbc_,
                      const double* xp_obs, const CamCfg* cam, int N, int group_begin, int g_sind, double* Hf_out /*2x3 col-major*/,
                      double* Hx_out /*2xN col-major*/, double* inn_out) {
  Vec3 Xs = Eigen::Map<const Vec3>(Xs_), Tsb = Eigen::Map<const Vec3>(Tsb_), Tbc = Eigen::Map<const Vec3>(Tbc_);
  Mat3 Rsb = Eigen::Map<const Mat3>(Rsb_), Rbc = Eigen::Map<const Mat3>(Rbc_);
  int goff = group_begin + 6 * g_sind;
  Mat3 Rsb_t = Rsb.transpose(), Rbc_t = Rbc.transpose();
  Vec3 Xb = Rsb_t * (Xs - Tsb);
  Mat3 dXb_dXs = Rsb_t, dXb_dTsb = -Rsb_t, dXb_dWsb = SO3::hat(Xb);
  Vec3 Xcn = Rbc_t * (Xb - Tbc);
  Mat3 dXcn_dXb = Rbc_t, dXcn_dWbc = SO3::hat(Xcn), dXcn_dTbc = -Rbc_t;
  Mat23 dxcn_dXcn;
  Vec2 xcn = project(Xcn, &dxcn_dXcn);
  Mat2 dxp_dxcn;
  Vec2 xp = cam_project(*cam, xcn, &dxp_dxcn);
  Mat23 dxp_dXcn = dxp_dxcn * dxcn_dXcn;
  inn_out[0] = xp_obs[0] - xp(0);
  inn_out[1] = xp_obs[1] - xp(1);
  (Eigen::Map<Mat23>(Hf_out)) = dxp_dXcn * dXcn_dXb * dXb_dXs;
  Eigen::Matrix<double, 2, Eigen::Dynamic> Hx = Eigen::Matrix<double, 2, Eigen::Dynamic>::Zero(2, N);
  Hx.block<2, 3>(0, goff) = dxp_dXcn * dXcn_dXb * dXb_dWsb;
  Hx.block<2, 3>(0, goff + 3) = dxp_dXcn * dXcn_dXb * dXb_dTsb;
  Hx.block<2, 3>(0, 15) = dxp_dXcn * dXcn_dWbc;
  Hx.block<2, 3>(0, 18) = dxp_dXcn * dXcn_dTbc;
  Eigen::Map<Eigen::Matrix<double, 2, Eigen::Dynamic>>(Hx_out, 2, N) = Hx;
}

// xivo::SlowGivens (the real one, src/helpers.cpp:13-23) + inn <- A^T inn (src/oos.cpp:29).
// Hf: R x 3, Hx: R x N, inn: R (col-major). Returns rows of the projected system.
int ref_slow_givens(int R, int N, const double* Hf, const double* Hx, const double* inn, double* Hx_out, double* inn_out,
                    double* A_out, int* A_cols) {
  MatX Hf_ = MapMat(Hf, R, 3), Hx_ = MapMat(Hx, R, N), A;
  VecX inn_ = MapVec(inn, R);
  int rows = SlowGivens(Hf_, Hx_, A);
  VecX r = A.transpose() * inn_;
  (MapMatW(Hx_out, rows, N)) = Hx_;
  (MapVecW(inn_out, rows)) = r;
  if (A_out) MapMatW(A_out, A.rows(), A.cols()) = A;
  if (A_cols) *A_cols = (int)A.cols();
  return rows;
}

// file-static xivo::givens (src/helpers.cpp:27-46); G col-major 2x2
void ref_givens(double a, double b, double* G) { (Eigen::Map<Mat2>(G)) = givens(a, b); }

// xivo::Givens (the real one, src/helpers.cpp:48-75)
int ref_Givens(int R, int N, double* x, double* Hx, double* Hf, int effective_rows) {
  VecX x_ = MapVec(x, R);
  MatX Hx_ = MapMat(Hx, R, N), Hf_ = MapMat(Hf, R, 3);
  int rows = Givens(x_, Hx_, Hf_, effective_rows);
  (MapVecW(x, R)) = x_; (MapMatW(Hx, R, N)) = Hx_; (MapMatW(Hf, R, 3)) = Hf_;
  return rows;
}

// xivo::QR (src/helpers.cpp:78-101)
int ref_QR(int R, int N, double* x, double* Hx, int effective_rows) {
  VecX x_ = MapVec(x, R);
  MatX Hx_ = MapMat(Hx, R, N);
  int rows = QR(x_, Hx_, effective_rows);
  (MapVecW(x, R)) = x_; (MapMatW(Hx, R, N)) = Hx_;
  return rows;
}

// Eigen::FullPivLU<MatX>(A).kernel() for an r x c matrix; returns kernel columns
int ref_fullpivlu_kernel(int r, int c, const double* A, double* ker_out, int* rank_out) {
  MatX A_ = MapMat(A, r, c);
  Eigen::FullPivLU<MatX> lu(A_);
  MatX k = lu.kernel();
  (MapMatW(ker_out, k.rows(), k.cols())) = k;
  if (rank_out) *rank_out = (int)lu.rank();
  return (int)k.cols();
}

// camera Project through the reference's own model classes
void ref_camera_project(const CamCfg* cam, const double* xc, double* xp, double* jac /*2x2 col-major*/) {
  Mat2 J = Mat2::Zero();
  Vec2 p = cam_project(*cam, Vec2(xc[0], xc[1]), &J);
  xp[0] = p(0); xp[1] = p(1);
  (Eigen::Map<Mat2>(jac)) = J;
}

// covariance tail of RK4Step, src/rk4.cpp:89-102 (+ Qmodel, src/estimator.cpp:590)
void ref_rk4_cov_tail(int N, int nm, double* P, const double* FK, const double* PK, double dt, const double* Qmodel) {
  MapMatW P_(P, N, N);
  MatX F_ = MatX::Identity(nm, nm);
  F_ = F_ + MapMat(FK, nm, nm) * dt;
  P_.block(0, 0, nm, nm) = P_.block(0, 0, nm, nm) + MapMat(PK, nm, nm) * dt;
  P_.block(0, nm, nm, N - nm) = F_ * P_.block(0, nm, nm, N - nm);
  P_.block(nm, 0, N - nm, nm) = P_.block(nm, 0, N - nm, nm) * F_.transpose();
  if (Qmodel) P_.block(0, 0, nm, nm).noalias() += MapMat(Qmodel, nm, nm);
}

// Sophus::SO3::exp
// Feature::SubfilterUpdate, src/feature.cpp:246-297 (default build: log-depth), restated on the same Eigen /
// Sophus types with the identical expression sequence. x (3), P (3x3 col-major), counters in/out.
// Returns the new status (0 INITIALIZING, 1 READY).
int ref_subfilter_update(double* x, double* P, const double* xp_meas, const double* Rsb_, const double* Tsb_,
                         const double* Rbc_, const double* Tbc_, const double* Rsbr_, const double* Tsbr_,
                         const CamCfg* cam, double Rtri, double MH_thresh, int ready_steps, int* init_counter,
                         double* outlier_counter) {
  const Mat3 Rsb = Eigen::Map<const Mat3>(Rsb_), Rbc = Eigen::Map<const Mat3>(Rbc_), Rsbr = Eigen::Map<const Mat3>(Rsbr_);
  const Vec3 Tsb = Eigen::Map<const Vec3>(Tsb_), Tbc = Eigen::Map<const Vec3>(Tbc_), Tsbr = Eigen::Map<const Vec3>(Tsbr_);
  const SE3 gsb = SE3(SO3(Rsb), Tsb), gbc = SE3(SO3(Rbc), Tbc), gref = SE3(SO3(Rsbr), Tsbr);
  Vec3 x_ = Eigen::Map<const Vec3>(x);
  Mat3 P_ = Eigen::Map<const Mat3>(P);
  (*init_counter)++;
  Mat3 dXc_dx;
  Vec3 Xc = unproject_logz(x_, &dXc_dx);
  SE3 gtot = (gsb * gbc).inverse() * gref * gbc;
  Vec3 Xcn = gtot * Xc;
  Mat3 dXcn_dXc = gtot.so3().matrix();
  Mat23 dxcn_dXcn;
  Vec2 xcn = project(Xcn, &dxcn_dXcn);
  Mat2 dxp_dxcn;
  Vec2 xp = cam_project(*cam, xcn, &dxp_dxcn);
  Mat23 H = dxp_dxcn * dxcn_dXcn * dXcn_dXc * dXc_dx;
  const Vec2 xpm = Eigen::Map<const Vec2>(xp_meas);
  Vec2 inn = xpm - xp;
  Mat2 S = H * P_ * H.transpose();
  S(0, 0) += Rtri;
  S(1, 1) += Rtri;
  double ratio{inn.dot(S.ldlt().solve(inn)) / MH_thresh};
  if (ratio > 1) {
    S(0, 0) += Rtri * (ratio - 1);
    S(1, 1) += Rtri * (ratio - 1);
    *outlier_counter += sqrt(ratio);
  } else {
    *outlier_counter = 0;
  }
  Eigen::Matrix<double, 3, 2> K = P_ * H.transpose() * S.inverse();
  x_ += K * inn;
  Mat3 I_KH = Mat3::Identity() - K * H;
  P_ = I_KH * P_ * I_KH.transpose() + K * Rtri * K.transpose();
  (Eigen::Map<Vec3>(x)) = x_;
  (Eigen::Map<Mat3>(P)) = P_;
  return *init_counter > ready_steps ? 1 : 0;
}

void ref_so3_exp(const double* w, double* R) { (Eigen::Map<Mat3>(R)) = SO3::exp(Eigen::Map<const Vec3>(w)).matrix(); }

}  // extern "C"

// ---------------------------------------------------------------------------
// Estimator::RK4Step (src/rk4.cpp:35-103) with ComposeMotion (src/estimator.cpp:598-613)
// and ComputeMotionJacobianAt (src/estimator.cpp:615-704), default build, restated line
// by line on the reference's own types (Sophus SO3 incl. normalize(), Eigen dense).
// state = [Rsb(9, col-major) Tsb(3) Vsb(3) bg(3) ba(3) Rsg(9)] in/out; P is N x N in/out.
// ---------------------------------------------------------------------------
namespace {
struct RefState { SO3 Rsb; Vec3 Tsb, Vsb, bg, ba; SO3 Rsg; };
struct RefProp {
  RefState X_;
  MatX P_, F_, G_, Qimu_;
  Vec3 g_, slope_gyro_, slope_accel_;
  Mat3 Cg, Ca;
  void ComposeMotion(RefState& X, const Vec3& V, const Eigen::Matrix<double, 6, 1>& gyro_accel, double dt) {
    Vec3 gyro = gyro_accel.head<3>();
    Vec3 accel = gyro_accel.tail<3>();
    Vec3 gyro_calib = Cg * gyro - X.bg;
    Vec3 accel_calib = Ca * accel - X.ba;
    X.Tsb += V * dt;
    X.Vsb += (X.Rsb * accel_calib + X.Rsg * g_) * dt;
    X.Rsb *= SO3::exp(gyro_calib * dt);
    X.Rsb.normalize();
  }
  void ComputeMotionJacobianAt(const RefState& X, const Eigen::Matrix<double, 6, 1>& gyro_accel) {
    Vec3 gyro = gyro_accel.head<3>();
    Vec3 accel = gyro_accel.tail<3>();
    Vec3 gyro_calib = Cg * gyro - X.bg;
    Vec3 accel_calib = Ca * accel - X.ba;
    Mat3 Rsb = X.Rsb.matrix();
    Mat3 dWsb_dWsb = -SO3::hat(gyro_calib);
    Mat3 dV_dWsb = -Rsb * SO3::hat(accel_calib);
    Mat3 dV_dba = -Rsb;
    Mat3 dV_dWsg = -Rsb * SO3::hat(g_);
    F_.setZero(23, 23);
    for (int j = 0; j < 3; ++j) {
      F_(0 + j, 9 + j) = -1;
      F_(3 + j, 6 + j) = 1;
      for (int i = 0; i < 3; ++i) {
        F_(0 + i, 0 + j) = dWsb_dWsb(i, j);
        F_(6 + i, 0 + j) = dV_dWsb(i, j);
        F_(6 + i, 12 + j) = dV_dba(i, j);
        if (j < 2) F_(6 + i, 21 + j) = dV_dWsg(i, j);
      }
    }
    G_.setZero(23, 12);
    for (int j = 0; j < 3; ++j) {
      G_(0 + j, j) = -1;
      G_(9 + j, 6 + j) = 1;
      G_(12 + j, 9 + j) = 1;
      for (int i = 0; i < 3; ++i) G_(6 + i, 3 + j) = -Rsb(i, j);
    }
  }
  void RK4Step(const Vec3& gyro0, const Vec3& accel0, double dt) {
    const int kMotionSize = 23;
    const int kFullSize = (int)P_.rows();
    double halfstep = 0.5 * dt;
    RefState X0;
    Vec3 K1, K2, K3, K4;
    MatX FK1, FK2, FK3, FK4, PK1, PK2, PK3, PK4;
    Eigen::Matrix<double, 6, 1> slope;
    slope << slope_gyro_, slope_accel_;
    Eigen::Matrix<double, 6, 1> gyro_accel, gyro_accel0;
    gyro_accel0 << gyro0, accel0;
    X0 = X_;
    K1 = X0.Vsb;
    ComputeMotionJacobianAt(X0, gyro_accel0);
    FK1 = F_;
    MatX P0 = P_.block(0, 0, kMotionSize, kMotionSize);
    PK1 = F_ * P0 + P0 * F_.transpose() + G_ * Qimu_ * G_.transpose();
    X0 = X_;
    gyro_accel = gyro_accel0 + halfstep * slope;
    ComposeMotion(X0, 0.5 * K1, gyro_accel, halfstep);
    K2 = X0.Vsb;
    ComputeMotionJacobianAt(X0, gyro_accel);
    FK2 = F_ + F_ * FK1 * halfstep;
    P0 = P_.block(0, 0, kMotionSize, kMotionSize) + halfstep * PK1;
    PK2 = F_ * P0 + P0 * F_.transpose() + G_ * Qimu_ * G_.transpose();
    X0 = X_;
    gyro_accel = gyro_accel0 + halfstep * slope;
    ComposeMotion(X0, 0.5 * K2, gyro_accel, halfstep);
    K3 = X0.Vsb;
    ComputeMotionJacobianAt(X0, gyro_accel);
    FK3 = F_ + F_ * FK2 * halfstep;
    P0 = P_.block(0, 0, kMotionSize, kMotionSize) + halfstep * PK2;
    PK3 = F_ * P0 + P0 * F_.transpose() + G_ * Qimu_ * G_.transpose();
    X0 = X_;
    gyro_accel = gyro_accel0 + halfstep * slope;
    ComposeMotion(X0, K3, gyro_accel, dt);
    K4 = X0.Vsb;
    ComputeMotionJacobianAt(X0, gyro_accel);
    FK4 = F_ + F_ * FK3 * dt;
    P0 = P_.block(0, 0, kMotionSize, kMotionSize) + dt * PK3;
    PK4 = F_ * P0 + P0 * F_.transpose() + G_ * Qimu_ * G_.transpose();
    Vec3 Ktot = (K1 + 2.0 * (K2 + K3) + K4) / 6.0;
    MatX FK = (FK1 + 2.0 * (FK2 + FK3) + FK4) / 6.0;
    MatX PK = (PK1 + 2.0 * (PK2 + PK3) + PK4) / 6.0;
    gyro_accel = gyro_accel0 + dt * slope;
    ComposeMotion(X_, Ktot, gyro_accel, dt);
    F_.setIdentity(kMotionSize, kMotionSize);
    F_ = F_ + FK * dt;
    P_.block(0, 0, kMotionSize, kMotionSize) = P_.block(0, 0, kMotionSize, kMotionSize) + PK * dt;
    P_.block(0, kMotionSize, kMotionSize, kFullSize - kMotionSize) =
        F_ * P_.block(0, kMotionSize, kMotionSize, kFullSize - kMotionSize);
    P_.block(kMotionSize, 0, kFullSize - kMotionSize, kMotionSize) =
        P_.block(kMotionSize, 0, kFullSize - kMotionSize, kMotionSize) * F_.transpose();
  }
  // Estimator::PrinceDormandStep, src/princedormand.cpp:85-221 (same members; the function-local statics are locals)
  void PrinceDormandStep(const Vec3& gyro0, const Vec3& accel0, double dt) {
    const int kMotionSize = 23;
    const int kFullSize = (int)P_.rows();
    const double r_9 = 1.0 / 9.0, r_2_9 = 2.0 / 9.0, r_12 = 1.0 / 12.0, r_324 = 1.0 / 324.0, r_330 = 1.0 / 330.0,
                 r_28 = 1.0 / 28.0, r_400 = 1.0 / 400.0;
    RefState X0;
    Vec3 K1, K2, K3, K4, K5, K6, K7;
    MatX FK1, FK2, FK3, FK4, FK5, FK6, FK7;
    MatX PK1, PK2, PK3, PK4, PK5, PK6, PK7;
    double step;
    Eigen::Matrix<double, 6, 1> slope;
    slope << slope_gyro_, slope_accel_;
    Eigen::Matrix<double, 6, 1> gyro_accel0, gyro_accel;
    gyro_accel0 << gyro0, accel0;

    X0 = X_;
    K1 = X0.Vsb;
    ComputeMotionJacobianAt(X0, gyro_accel0);
    FK1 = F_;
    MatX P0 = P_.block(0, 0, kMotionSize, kMotionSize);
    PK1 = F_ * P0 + P0 * F_.transpose() + G_ * Qimu_ * G_.transpose();

    X0 = X_;
    step = r_2_9 * dt;
    gyro_accel = gyro_accel0 + slope * step;
    ComposeMotion(X0, r_2_9 * (K1), gyro_accel, step);
    ComputeMotionJacobianAt(X0, gyro_accel);
    K2 = X0.Vsb;
    FK2 = F_ + F_ * r_2_9 * (FK1)*dt;
    P0 = P_.block(0, 0, kMotionSize, kMotionSize) + r_2_9 * (PK1)*dt;
    PK2 = F_ * P0 + P0 * F_.transpose() + G_ * Qimu_ * G_.transpose();

    X0 = X_;
    step = 3.0 * r_9 * dt;
    gyro_accel = gyro_accel0 + slope * step;
    ComposeMotion(X0, r_12 * (K1 + 3.0 * K2), gyro_accel, step);
    ComputeMotionJacobianAt(X0, gyro_accel);
    K3 = X0.Vsb;
    FK3 = F_ + F_ * r_12 * (FK1 + 3.0 * FK2) * dt;
    P0 = P_.block(0, 0, kMotionSize, kMotionSize) + r_12 * (PK1 + 3.0 * PK2) * dt;
    PK3 = F_ * P0 + P0 * F_.transpose() + G_ * Qimu_ * G_.transpose();

    X0 = X_;
    step = 5.0 * r_9 * dt;
    gyro_accel = gyro_accel0 + slope * step;
    ComposeMotion(X0, r_324 * (55.0 * K1 - 75.0 * K2 + 200.0 * K3), gyro_accel, step);
    ComputeMotionJacobianAt(X0, gyro_accel);
    K4 = X0.Vsb;
    FK4 = F_ + F_ * r_324 * (55.0 * FK1 - 75.0 * FK2 + 200.0 * FK3) * dt;
    P0 = P_.block(0, 0, kMotionSize, kMotionSize) + r_324 * (55.0 * PK1 - 75.0 * PK2 + 200.0 * PK3) * dt;
    PK4 = F_ * P0 + P0 * F_.transpose() + G_ * Qimu_ * G_.transpose();

    X0 = X_;
    step = 6.0 * r_9 * dt;
    gyro_accel = gyro_accel0 + slope * step;
    ComposeMotion(X0, r_330 * (83.0 * K1 - 195.0 * K2 + 305.0 * K3 + 27.0 * K4), gyro_accel, step);
    ComputeMotionJacobianAt(X0, gyro_accel);
    K5 = X0.Vsb;
    FK5 = F_ + F_ * r_330 * (83.0 * FK1 - 195.0 * FK2 + 305.0 * FK3 + 27.0 * FK4) * dt;
    P0 = P_.block(0, 0, kMotionSize, kMotionSize) + r_330 * (83.0 * PK1 - 195.0 * PK2 + 305.0 * PK3 + 27.0 * PK4) * dt;
    PK5 = F_ * P0 + P0 * F_.transpose() + G_ * Qimu_ * G_.transpose();

    X0 = X_;
    step = dt;
    gyro_accel = gyro_accel0 + slope * step;
    ComposeMotion(X0, r_28 * (-19.0 * K1 + 63.0 * K2 + 4.0 * K3 - 108.0 * K4 + 88.0 * K5), gyro_accel, step);
    ComputeMotionJacobianAt(X0, gyro_accel);
    K6 = X0.Vsb;
    FK6 = F_ + F_ * r_28 * (-19.0 * FK1 + 63.0 * FK2 + 4.0 * FK3 - 108.0 * FK4 + 88.0 * FK5) * dt;
    P0 = P_.block(0, 0, kMotionSize, kMotionSize) +
         r_28 * (-19.0 * PK1 + 63.0 * PK2 + 4.0 * PK3 - 108.0 * PK4 + 88.0 * PK5) * dt;
    PK6 = F_ * P0 + P0 * F_.transpose() + G_ * Qimu_ * G_.transpose();

    X0 = X_;
    step = dt;
    gyro_accel = gyro_accel0 + slope * step;
    ComposeMotion(X0, r_400 * (38.0 * K1 + 240.0 * K3 - 243.0 * K4 + 330.0 * K5 + 35.0 * K6), gyro_accel, step);
    ComputeMotionJacobianAt(X0, gyro_accel);
    K7 = X0.Vsb;
    FK7 = F_ + F_ * r_400 * (38.0 * FK1 + 240.0 * FK3 - 243.0 * FK4 + 330.0 * FK5 + 35.0 * FK6) * dt;
    P0 = P_.block(0, 0, kMotionSize, kMotionSize) +
         r_400 * (38.0 * PK1 + 240.0 * PK3 - 243.0 * PK4 + 330.0 * PK5 + 35.0 * PK6) * dt;
    PK7 = F_ * P0 + P0 * F_.transpose() + G_ * Qimu_ * G_.transpose();

    MatX K, FK, PK;
    K = 0.0862 * K1 + 0.6660 * K3 - 0.7857 * K4 + 0.9570 * K5 + 0.0965 * K6 - 0.0200 * K7;
    FK = 0.0862 * FK1 + 0.6660 * FK3 - 0.7857 * FK4 + 0.9570 * FK5 + 0.0965 * FK6 - 0.0200 * FK7;
    PK = 0.0862 * PK1 + 0.6660 * PK3 - 0.7857 * PK4 + 0.9570 * PK5 + 0.0965 * PK6 - 0.0200 * PK7;

    gyro_accel = gyro_accel0 + slope * dt;
    ComposeMotion(X_, K, gyro_accel, dt);

    F_.setIdentity(kMotionSize, kMotionSize);
    F_ = F_ + FK * dt;

    P_.block(0, 0, kMotionSize, kMotionSize).noalias() += PK * dt;
    P_.block(0, kMotionSize, kMotionSize, kFullSize - kMotionSize) =
        F_ * P_.block(0, kMotionSize, kMotionSize, kFullSize - kMotionSize);
    P_.block(kMotionSize, 0, kFullSize - kMotionSize, kMotionSize) =
        P_.block(kMotionSize, 0, kFullSize - kMotionSize, kMotionSize) * F_.transpose();
  }
};

void load_prop(RefProp& r, int N, const double* state30, const double* P, const double* slope_gyro, const double* slope_accel,
               const double* Qimu, const double* g_vec) {
  Mat3 R0 = Eigen::Map<const Mat3>(state30), Rg = Eigen::Map<const Mat3>(state30 + 21);
  r.X_.Rsb = SO3(Eigen::Quaterniond(R0));
  r.X_.Rsg = SO3(Eigen::Quaterniond(Rg));
  r.X_.Tsb = Eigen::Map<const Vec3>(state30 + 9); r.X_.Vsb = Eigen::Map<const Vec3>(state30 + 12);
  r.X_.bg = Eigen::Map<const Vec3>(state30 + 15); r.X_.ba = Eigen::Map<const Vec3>(state30 + 18);
  r.P_ = MapMat(P, N, N);
  r.Qimu_ = MapMat(Qimu, 12, 12);
  r.g_ = Eigen::Map<const Vec3>(g_vec);
  r.slope_gyro_ = Eigen::Map<const Vec3>(slope_gyro); r.slope_accel_ = Eigen::Map<const Vec3>(slope_accel);
  r.Cg.setIdentity(); r.Ca.setIdentity();
}
void store_prop(const RefProp& r, int N, double* state30, double* P) {
  Eigen::Map<Mat3> Ro(state30);
  Ro = r.X_.Rsb.matrix();
  (Eigen::Map<Vec3>(state30 + 9)) = r.X_.Tsb; (Eigen::Map<Vec3>(state30 + 12)) = r.X_.Vsb;
  (MapMatW(P, N, N)) = r.P_;
}
}  // namespace

extern "C" void ref_pd_step(int N, double* state30, double* P, const double* gyro0, const double* accel0,
                            const double* slope_gyro, const double* slope_accel, double dt, const double* Qimu,
                            const double* g_vec) {
  RefProp r;
  load_prop(r, N, state30, P, slope_gyro, slope_accel, Qimu, g_vec);
  r.PrinceDormandStep(Eigen::Map<const Vec3>(gyro0), Eigen::Map<const Vec3>(accel0), dt);
  store_prop(r, N, state30, P);
}

// Estimator::OnePointRANSAC, src/update.cpp:213-393, numeric core up to and including the partial update (:238-332):
// the low-innovation set (the hypothesis index k of :240-244 is drawn but never used, so the set is
// {f : |xp - Predict| < ransac_thresh_}; xp - Predict == inn at the state the Jacobians were taken at), the temporary
// reference group (FindNewRefGroup, src/estimator.cpp:1394-1407), the zeroing of P (:299-316), the stacking of the
// FULL rows J() (:320-330) and UpdateJosephForm (:332). The caller continues with AbsorbError / ComputeJacobian /
// the chi-square rescue through the functions above.
//   J: F blocks of 2 x N (col-major), inn: F x 2, sind / ref: feature and anchor-group slots, gauge: slot of
//   gauge_group_ptr_ (-1: none in the state). Outputs: low[F] (0/1), err[N], P_out (P after zeroing + update; = P if
//   no update ran). Returns the number of low-innovation inliers, or -1 when all are (early return of :263-265).
extern "C" int ref_one_point_ransac_core(int N, int F, const double* J, const double* inn, const double* P, const int* sind,
                                         const int* ref, int gauge, int group_begin, int feature_begin, double R,
                                         double ransac_thresh, int* low, double* err_out, double* P_out) {
  const int kGroupBegin = group_begin, kFeatureBegin = feature_begin, kGroupSize = 6, kFeatureSize = 3;
  MatX P_ = MapMat(P, N, N);
  const int size = N;
  std::vector<bool> is_low_innovation_inlier;
  std::vector<int> groups_with_low_inn_inlier, active_groups;
  int n_low = 0;
  for (int i = 0; i < F; ++i) {
    Vec2 res(inn[2 * i], inn[2 * i + 1]);
    const bool in = res.norm() < ransac_thresh;                                      // :246-249
    is_low_innovation_inlier.push_back(in);
    low[i] = in ? 1 : 0;
    n_low += in;
    if (std::find(active_groups.begin(), active_groups.end(), ref[i]) == active_groups.end()) active_groups.push_back(ref[i]);
    if (in && std::find(groups_with_low_inn_inlier.begin(), groups_with_low_inn_inlier.end(), ref[i]) == groups_with_low_inn_inlier.end())
      groups_with_low_inn_inlier.push_back(ref[i]);
  }
  (MapMatW(P_out, N, N)) = P_;
  for (int i = 0; i < N; ++i) err_out[i] = 0.0;
  if (n_low == F) return -1;                                                         // :263-265
  if (n_low > 0) {
    if (std::find(groups_with_low_inn_inlier.begin(), groups_with_low_inn_inlier.end(), gauge) == groups_with_low_inn_inlier.end()) {
      // FindNewRefGroup: std::min_element over the candidates by the summed 6 diagonal entries (first minimum wins).
      // The reference iterates an unordered_set of pointers; here the candidates are visited in ascending slot order.
      std::vector<int> candidates = groups_with_low_inn_inlier;
      std::sort(candidates.begin(), candidates.end());
      auto git = std::min_element(candidates.begin(), candidates.end(), [&](int g1, int g2) -> bool {
        int offset1 = kGroupBegin + 6 * g1, offset2 = kGroupBegin + 6 * g2;
        number_t cov1{0}, cov2{0};
        for (int i = 0; i < 6; ++i) { cov1 += P_(offset1 + i, offset1 + i); cov2 += P_(offset2 + i, offset2 + i); }
        return cov1 < cov2;
      });
      int offset = kGroupBegin + kGroupSize * (*git);
      P_.block(offset, 0, kGroupSize, size).setZero();
      P_.block(0, offset, size, kGroupSize).setZero();
    }
    for (int i = 0; i < F; ++i) {
      if (!is_low_innovation_inlier[i]) {
        int offset = kFeatureBegin + kFeatureSize * sind[i];
        P_.block(offset, 0, kFeatureSize, size).setZero();
        P_.block(0, offset, size, kFeatureSize).setZero();
      }
    }
    for (int g : active_groups) {
      if (std::find(groups_with_low_inn_inlier.begin(), groups_with_low_inn_inlier.end(), g) == groups_with_low_inn_inlier.end()) {
        int offset = kGroupBegin + kGroupSize * g;
        P_.block(offset, 0, kGroupSize, size).setZero();
        P_.block(0, offset, size, kGroupSize).setZero();
      }
    }
    MatX H_; VecX inn_, diagR_;
    H_.setZero(2 * n_low, size);
    inn_.setZero(2 * n_low);
    diagR_.resize(2 * n_low);
    int f_cnt = 0;
    for (int i = 0; i < F; ++i) {
      if (is_low_innovation_inlier[i]) {
        H_.block(2 * f_cnt, 0, 2, size) = Eigen::Map<const Eigen::Matrix<double, 2, Eigen::Dynamic>>(J + (size_t)i * 2 * N, 2, N);
        inn_.segment<2>(2 * f_cnt) = Vec2(inn[2 * i], inn[2 * i + 1]);
        diagR_.segment<2>(2 * f_cnt) << R, R;
        f_cnt++;
      }
    }
    std::vector<double> Pin(P_.data(), P_.data() + (size_t)N * N);
    ref_update_joseph(N, 2 * n_low, H_.data(), Pin.data(), inn_.data(), diagR_.data(), err_out, P_out);
  }
  return n_low;
}

extern "C" void ref_rk4_step(int N, double* state30, double* P, const double* gyro0, const double* accel0,
                             const double* slope_gyro, const double* slope_accel, double dt, const double* Qimu,
                             const double* g_vec) {
  RefProp r;
  load_prop(r, N, state30, P, slope_gyro, slope_accel, Qimu, g_vec);
  r.RK4Step(Eigen::Map<const Vec3>(gyro0), Eigen::Map<const Vec3>(accel0), dt);
  store_prop(r, N, state30, P);
}
