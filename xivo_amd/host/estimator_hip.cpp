// Implementation of the reference-shaped adapter (estimator_hip.h) over the C ABI.
#include "estimator_hip.h"

#include <chrono>
#include <cmath>
#include <cstring>

namespace xivo {
namespace hip {

void Estimator::Check(int status, const char* what) const {
  // The reference aborts (LOG(FATAL), src/estimator.cpp:121,587) or throws
  // (std::runtime_error, src/estimator.cpp:821,844) on failure; the C ABI only
  // returns codes, so the adapter restores the throwing behaviour.
  if (status != XIVO_HIP_OK)
    throw std::runtime_error(std::string(what) + ": " + xivo_hip_strerror(status));
}

Estimator::Estimator(const xivo_layout& layout, const xivo_cam& cam, int max_features, unsigned flags, int device)
    : lay_(layout), cam_(cam), flags_(flags), max_features_(max_features) {
  Check(xivo_hip_create(&ctx_, device, layout.N, 2 * max_features, 1, flags), "xivo_hip_create");
  Check(xivo_hip_set_layout(ctx_, &lay_, &cam_), "xivo_hip_set_layout");
#if XIVO_HIP_ONLINE_CALIB
  {  // enum Index / kMotionSize / kCameraBegin / kGroupBegin as src/core.h:40-105 numbers them under this build's defines
    int nxt = 23;
#ifdef USE_ONLINE_TEMPORAL_CALIB
    cl_.td = nxt++;
#endif
#ifdef USE_ONLINE_IMU_CALIB
    cl_.Cg = nxt; nxt += 15;               // Cg (9) then Ca (6)
#endif
    motion_ = nxt;
    cl_.cam_begin = motion_;
#ifdef USE_ONLINE_CAMERA_CALIB
    cl_.cam_dim = cam.model == XIVO_CAM_PINHOLE ? 4 : (cam.model == XIVO_CAM_ATAN ? 5 : (cam.model == XIVO_CAM_RADTAN ? 9 : 8));
    const int group_begin = motion_ + 9;   // kMaxCameraIntrinsics slots
#else
    const int group_begin = motion_;
#endif
    if (layout.group_begin != group_begin)
      throw std::invalid_argument("Estimator: layout.group_begin does not match this build's kGroupBegin");
    Check(xivo_hip_set_calib(ctx_, &cl_), "xivo_hip_set_calib");
    intr_[0] = cam.fx; intr_[1] = cam.fy; intr_[2] = cam.cx; intr_[3] = cam.cy;
    for (int k = 0; k < 5; ++k) intr_[4 + k] = cam.d[k];
  }
#endif
  P_.setZero(layout.N, layout.N);
  err_.setZero(layout.N);
  groups_.assign(layout.n_groups, nullptr);
}

Estimator::~Estimator() { xivo_hip_destroy(ctx_); }

void Feature::FillJacobianBlock(MatX& H, int offset) const {
  const xivo_layout& lay = owner_->layout();
  auto copy3 = [&](int dst, int src) {
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) H(offset + i, dst + j) = J_(i, src + j);
  };
  copy3(0, 0);    // Index::Wsb   feature.cpp:659
  copy3(3, 3);    // Index::Tsb   :660
  copy3(15, 15);  // Index::Wbc   :661
  copy3(18, 18);  // Index::Tbc   :662
  const int goff = lay.group_begin + 6 * ref_->sind();
  const int foff = lay.feature_begin + 3 * sind();
  copy3(goff, goff);                                   // :675
  if (owner_->flags() & XIVO_HIP_FLAG_FIX_GROUP_BLOCK) copy3(goff + 3, goff + 3);
  else copy3(goff, goff + 3);                          // :676 overwrites, goff+3.. stays zero
  copy3(foff, foff);                                   // :677
#if XIVO_HIP_ONLINE_CALIB
  const xivo_calib_layout& cl = owner_->calib_layout();
  auto copyn = [&](int col, int n) { for (int i = 0; i < 2; ++i) for (int j = 0; j < n; ++j) H(offset + i, col + j) = J_(i, col + j); };
#ifdef USE_ONLINE_TEMPORAL_CALIB
  copyn(cl.td, 1);                                     // :665
#ifdef USE_ONLINE_IMU_CALIB
  copyn(cl.Cg, 9);                                     // :667
#endif
  copyn(9, 3);                                         // Index::bg :669
#endif
#ifdef USE_ONLINE_CAMERA_CALIB
  copyn(cl.cam_begin, cl.cam_dim);                     // :679-683
#endif
#endif
}

// uploads P_ if the device copy is not known to be current (resident flows call it after editing P_ on the host)
void Estimator::SyncDeviceP() {
  if (device_P_current_) return;
  const int N = lay_.N;
  Check(xivo_hip_upload_P(ctx_, 0, 1, P_.data(), (long)N * N, N), "upload_P");
  device_P_current_ = true;
}

void Estimator::UpdateJosephForm() {
  const int N = lay_.N, M = H_.rows();
  if (H_.cols() != N || inn_.size() != M || diagR_.size() != M || P_.rows() != N)
    throw std::invalid_argument("UpdateJosephForm: inconsistent sizes");
  err_.setZero(N);
  if (legacy_plumbing_) {   // the six-call sequence of rounds 1-3 (kept for A/B timing: tests, bench.py `dropin`)
    Check(xivo_hip_upload_P(ctx_, 0, 1, P_.data(), (long)N * N, N), "upload_P");
    Check(xivo_hip_set_measurements(ctx_, 0, 1, M, H_.data(), (long)M * N, M, inn_.data(), M, diagR_.data(), M),
          "set_measurements");
    Check(xivo_hip_update_joseph(ctx_, 1), "update_joseph");
    int st = 0;
    const int rc = xivo_hip_get_status(ctx_, 0, 1, &st);
    if (rc == XIVO_HIP_ERR_NOT_SPD) { ++num_not_spd_; last_update_ok_ = false; return; }
    Check(rc, "get_status");
    last_update_ok_ = true;
    Check(xivo_hip_get_err(ctx_, 0, 1, err_.data(), N), "get_err");
    Check(xivo_hip_download_P(ctx_, 0, 1, P_.data(), (long)N * N, N), "download_P");
    device_P_current_ = true;
    return;
  }
  // ONE call, one host synchronisation: P_ through the context's page-locked block, H_ scanned once on the host and
  // handed over as row-pair compressed rows, err_ and the status come back with P_
  const unsigned mode = (trust_device_P_ && device_P_current_) ? XIVO_HIP_HOST_P_RESIDENT : 0u;
  const int rc = xivo_hip_update_joseph_host(ctx_, 0, M, H_.data(), M, inn_.data(), diagR_.data(), P_.data(), N, err_.data(), mode);
  if (rc == XIVO_HIP_ERR_NOT_SPD) {
    // only with XIVO_HIP_FLAG_NO_LDLT_FALLBACK (by default the device runs the reference's pivoted L D L^T for such a
    // filter and the status reads OK): S = HPH^T + R was not positive definite, the device left its P - and so P_ - the
    // prior; the measurement is dropped (err_ = 0), counted, and the filter carries on
    err_.setZero(N);
    ++num_not_spd_;
    last_update_ok_ = false;
    device_P_current_ = true;
    return;
  }
  Check(rc, "update_joseph_host");
  last_update_ok_ = true;
  device_P_current_ = true;
}

void Estimator::ComputeInstateJacobians() {
  const int F = (int)instate_features_.size();
  if (F == 0) return;
  if (F > max_features_) throw std::runtime_error("more in-state features than the context was created for");
  xivo_pose_in pose;
  std::memcpy(pose.Rsb, Rsb_.data(), sizeof(pose.Rsb)); std::memcpy(pose.Tsb, Tsb_.data(), sizeof(pose.Tsb));
  std::memcpy(pose.Rbc, Rbc_.data(), sizeof(pose.Rbc)); std::memcpy(pose.Tbc, Tbc_.data(), sizeof(pose.Tbc));
  std::memcpy(pose.Vsb, Vsb_.data(), sizeof(pose.Vsb)); std::memcpy(pose.bg, bg_.data(), sizeof(pose.bg));
  std::memcpy(pose.ba, ba_.data(), sizeof(pose.ba)); std::memcpy(pose.Rsg, Rsg_.data(), sizeof(pose.Rsg));
  std::vector<xivo_group_in> gs(lay_.n_groups);
  for (int g = 0; g < lay_.n_groups; ++g) {
    const Mat3 I = Identity3(); const Vec3 z;
    const Group* gp = groups_[g];
    std::memcpy(gs[g].Rsb, gp ? gp->Rsb_.data() : I.data(), sizeof(gs[g].Rsb));
    std::memcpy(gs[g].Tsb, gp ? gp->Tsb_.data() : z.data(), sizeof(gs[g].Tsb));
  }
  std::vector<xivo_feat_in> fs(F);
  for (int i = 0; i < F; ++i) {
    const Feature* f = instate_features_[i];
    std::memcpy(fs[i].x, f->x_.data(), sizeof(fs[i].x));
    std::memcpy(fs[i].xp, f->back_.data(), sizeof(fs[i].xp));
    fs[i].ref_sind = f->ref_->sind();
    fs[i].sind = f->sind_;
  }
  Check(xivo_hip_set_scene(ctx_, 0, 1, F, &pose, gs.data(), fs.data()), "set_scene");
#if XIVO_HIP_ONLINE_CALIB
  {  // what ComputeInstateJacobians hands down besides the poses: last_gyro_, imu_.Cg(), X_.td (src/update.cpp:27-28) + the
     // camera's current intrinsics
    xivo_calib_in cs{};
    std::memcpy(cs.gyro, last_gyro_.data(), sizeof(cs.gyro)); std::memcpy(cs.Cg, Cg_.data(), sizeof(cs.Cg));
    cs.td = td_; std::memcpy(cs.Ca, Ca_.data(), sizeof(cs.Ca)); std::memcpy(cs.intr, intr_, sizeof(cs.intr));
    Check(xivo_hip_set_calib_state(ctx_, 0, 1, &cs), "set_calib_state");
  }
#endif
  Check(xivo_hip_jacobians_instate(ctx_, 1), "jacobians_instate");
  std::vector<double> J((size_t)F * 42), inn((size_t)F * 2);
  Check(xivo_hip_get_jacobians(ctx_, 0, 1, J.data(), inn.data()), "get_jacobians");
#if XIVO_HIP_ONLINE_CALIB
  std::vector<double> Jc((size_t)F * 44);              // per feature 2 x 22: td | Cg 9 | bg 3 | intrinsics 9
  Check(xivo_hip_get_jacobians_calib(ctx_, 0, 1, Jc.data()), "get_jacobians_calib");
#endif
  for (int i = 0; i < F; ++i) {
    Feature* f = instate_features_[i];
    f->owner_ = this;
    f->J_.setZero(2, lay_.N);          // J_.setZero() feature.cpp:622
    const int goff = lay_.group_begin + 6 * f->ref_->sind(), foff = lay_.feature_begin + 3 * f->sind_;
    const int offs[7] = {0, 3, 15, 18, goff, goff + 3, foff};
    for (int b = 0; b < 7; ++b)
      for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c) f->J_(r, offs[b] + c) = J[(size_t)i * 42 + r * 21 + 3 * b + c];
    f->inn_(0) = inn[2 * i]; f->inn_(1) = inn[2 * i + 1];
#if XIVO_HIP_ONLINE_CALIB
    for (int r = 0; r < 2; ++r) {                        // J_ blocks of src/feature.cpp:632-651
      const double* q = &Jc[(size_t)i * 44 + r * 22];
      if (cl_.td >= 0) {
        f->J_(r, cl_.td) = q[0];
        if (cl_.Cg >= 0) for (int c = 0; c < 9; ++c) f->J_(r, cl_.Cg + c) = q[1 + c];
        for (int c = 0; c < 3; ++c) f->J_(r, 9 + c) = q[10 + c];                       // Index::bg
      }
      for (int c = 0; c < cl_.cam_dim; ++c) f->J_(r, cl_.cam_begin + c) = q[13 + c];
    }
#endif
  }
}

std::vector<FeaturePtr> Estimator::MHGating() {
  const int F = (int)instate_features_.size(), N = lay_.N;
  std::vector<FeaturePtr> inliers;
  num_mh_rejected_ = 0;
  if (F == 0) return inliers;
  if (!(trust_device_P_ && device_P_current_)) Check(xivo_hip_upload_P(ctx_, 0, 1, P_.data(), (long)N * N, N), "upload_P");
  device_P_current_ = true;
  std::vector<unsigned char> mask(F);
  std::vector<double> dist(F);
  Check(xivo_hip_mh_gate(ctx_, 1, R_, MH_thresh_, MH_thresh_multipler_, min_required_inliers_, mask.data(), dist.data()),
        "mh_gate");
  // num_mh_rejected_ accumulates over the relaxation rounds (src/update.cpp:87): replay the
  // loop on the distances to keep the counter's value identical.
  {
    number_t th = MH_thresh_;
    int n_in = 0, guard = 0;
    while (n_in < min_required_inliers_ && guard++ < 4096) {
      n_in = 0;
      for (int i = 0; i < F; ++i) {
        if (dist[i] < th) ++n_in; else ++num_mh_rejected_;
      }
      if (n_in == F) break;
      th *= MH_thresh_multipler_;
    }
  }
  for (int i = 0; i < F; ++i) {
    Feature* f = instate_features_[i];
    if (f->status() != FeatureStatus::GAUGE) f->SetStatus(FeatureStatus::INSTATE);   // :76-80
    if (mask[i]) inliers.push_back(f);
    else f->SetStatus(FeatureStatus::REJECTED_BY_FILTER);                               // :110
  }
  return inliers;
}

void Estimator::FilterUpdate() {
  const int total_size = 2 * (int)in_current_ekf_update_.size();
  H_.setZero(total_size, err_.size());       // update.cpp:130
  inn_.setZero(total_size);                  // :131
  diagR_.resize(total_size);                 // :132
  for (int i = 0; i < (int)in_current_ekf_update_.size(); ++i) {
    in_current_ekf_update_[i]->FillJacobianBlock(H_, 2 * i);   // :135
    inn_(2 * i) = in_current_ekf_update_[i]->inn()(0);          // :136
    inn_(2 * i + 1) = in_current_ekf_update_[i]->inn()(1);
    diagR_(2 * i) = R_; diagR_(2 * i + 1) = R_;                 // :137
  }
  UpdateJosephForm();                                           // :141
}

// ---------------------------------------------------------------------------
// AbsorbError and OnePointRANSAC (compositions; the heavy parts run on the device)
// ---------------------------------------------------------------------------
namespace {
Mat3 mul3(const Mat3& a, const Mat3& b) { Mat3 c; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j); return c; }
Mat3 exp3(double wx, double wy, double wz) {   // SO3_from_rotvec / SO3::exp (src/helpers.cpp:374-378)
  const double th = std::sqrt(wx * wx + wy * wy + wz * wz);
  Mat3 W; W(0,0)=0; W(0,1)=-wz; W(0,2)=wy; W(1,0)=wz; W(1,1)=0; W(1,2)=-wx; W(2,0)=-wy; W(2,1)=wx; W(2,2)=0;
  const Mat3 W2 = mul3(W, W);
  const double a = th < 1e-10 ? 1.0 : std::sin(th) / th, b = th < 1e-10 ? 0.5 : (1 - std::cos(th)) / (th * th);
  Mat3 R;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = (i == j) + a * W(i, j) + b * W2(i, j);
  return R;
}
void zero_rc(MatX& P, int off, int len) {
  const int n = P.rows();
  for (int r = 0; r < len; ++r) for (int t = 0; t < n; ++t) { P(off + r, t) = 0.0; P(t, off + r) = 0.0; }
}
}  // namespace

namespace {
// unit quaternion (w, x, y, z) of a rotation matrix (Shepperd), and back: Sophus SO3::normalize() on the
// reference's quaternion storage = this round trip on the matrix storage used here
void rot_to_quat(const Mat3& R, number_t q[4]) {
  const number_t t = R(0, 0) + R(1, 1) + R(2, 2);
  if (t > 0.0) {
    const number_t s = std::sqrt(t + 1.0) * 2.0;
    q[0] = 0.25 * s; q[1] = (R(2, 1) - R(1, 2)) / s; q[2] = (R(0, 2) - R(2, 0)) / s; q[3] = (R(1, 0) - R(0, 1)) / s;
  } else if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) {
    const number_t s = std::sqrt(1.0 + R(0, 0) - R(1, 1) - R(2, 2)) * 2.0;
    q[0] = (R(2, 1) - R(1, 2)) / s; q[1] = 0.25 * s; q[2] = (R(0, 1) + R(1, 0)) / s; q[3] = (R(0, 2) + R(2, 0)) / s;
  } else if (R(1, 1) > R(2, 2)) {
    const number_t s = std::sqrt(1.0 + R(1, 1) - R(0, 0) - R(2, 2)) * 2.0;
    q[0] = (R(0, 2) - R(2, 0)) / s; q[1] = (R(0, 1) + R(1, 0)) / s; q[2] = 0.25 * s; q[3] = (R(1, 2) + R(2, 1)) / s;
  } else {
    const number_t s = std::sqrt(1.0 + R(2, 2) - R(0, 0) - R(1, 1)) * 2.0;
    q[0] = (R(1, 0) - R(0, 1)) / s; q[1] = (R(0, 2) + R(2, 0)) / s; q[2] = (R(1, 2) + R(2, 1)) / s; q[3] = 0.25 * s;
  }
  const number_t n = 1.0 / std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] *= n;
}
Mat3 quat_to_rot(const number_t q[4]) {
  const number_t w = q[0], x = q[1], y = q[2], z = q[3];
  Mat3 R;
  R(0, 0) = 1 - 2 * (y * y + z * z); R(0, 1) = 2 * (x * y - w * z);     R(0, 2) = 2 * (x * z + w * y);
  R(1, 0) = 2 * (x * y + w * z);     R(1, 1) = 1 - 2 * (x * x + z * z); R(1, 2) = 2 * (y * z - w * x);
  R(2, 0) = 2 * (x * z - w * y);     R(2, 1) = 2 * (y * z + w * x);     R(2, 2) = 1 - 2 * (x * x + y * y);
  return R;
}
}  // namespace

void Estimator::AbsorbError() {
  // State::operator+= (src/core.h:135-165)
  Rsb_ = mul3(Rsb_, exp3(err_(0), err_(1), err_(2)));
  for (int i = 0; i < 3; ++i) { Tsb_(i) += err_(3 + i); Vsb_(i) += err_(6 + i); bg_(i) += err_(9 + i); ba_(i) += err_(12 + i); Tbc_(i) += err_(18 + i); }
  Rbc_ = mul3(Rbc_, exp3(err_(15), err_(16), err_(17)));
  Rsg_ = mul3(Rsg_, exp3(err_(21), err_(22), 0.0));
#if XIVO_HIP_ONLINE_CALIB
  if (cl_.td >= 0) td_ += err_(cl_.td);                                  // src/core.h:150-152
  if (cl_.Cg >= 0) {                                                     // estimator.cpp:879-884 -> IMUState::operator+= (src/imu.cpp:7-21)
    int k = cl_.Cg + 9;                                                  // Index::Ca: the upper triangle row by row
    for (int i = 0; i < 3; ++i) for (int j = i; j < 3; ++j) Ca_(i, j) += err_(k++);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Cg_(i, j) += err_(cl_.Cg + 3 * i + j);
  }
  for (int k = 0; k < cl_.cam_dim; ++k) intr_[k] += err_(cl_.cam_begin + k);   // estimator.cpp:886-890 -> A_*Camera::UpdateState
#endif
  if (++absorb_counter_ % 50 == 0) {   // kEnforceSO3Freq (src/core.h:111,154-162)
    number_t q[4];
    rot_to_quat(Rsb_, q); Rsb_ = quat_to_rot(q);
    rot_to_quat(Rbc_, q); Rbc_ = quat_to_rot(q);
    rot_to_quat(Rsg_, q);              // Wsg = Rsg.log(); Wsg(2) = 0; Rsg = exp(Wsg)   (Sophus SO3::log on the quaternion)
    const number_t n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3], w = q[0];
    number_t k;
    if (n2 < 1e-20) k = 2.0 / w - 2.0 / 3.0 * n2 / (w * w * w);
    else {
      const number_t n = std::sqrt(n2);
      k = std::fabs(w) < 1e-10 ? (w > 0.0 ? M_PI / n : -M_PI / n) : 2.0 * std::atan(n / w) / n;
    }
    Rsg_ = exp3(k * q[1], k * q[2], 0.0);
  }
  for (Group* g : instate_groups_) {                                   // estimator.cpp:897-905
    const int off = lay_.group_begin + 6 * g->sind();
    g->Rsb_ = mul3(g->Rsb_, exp3(err_(off), err_(off + 1), err_(off + 2)));
    for (int i = 0; i < 3; ++i) g->Tsb_(i) += err_(off + 3 + i);
  }
  for (Feature* f : in_current_ekf_update_) {                          // :906-912
    const int off = lay_.feature_begin + 3 * f->sind();
    for (int i = 0; i < 3; ++i) f->x_(i) += err_(off + i);
  }
  err_.setZero(err_.size());                                            // :920
}

std::vector<FeaturePtr> Estimator::OnePointRANSAC(const std::vector<FeaturePtr>& mh_inliers) {
  if (mh_inliers.empty()) return mh_inliers;
  const int n = (int)mh_inliers.size(), size = lay_.N;
  // update.cpp:238-258: the hypothesis index k is drawn but never used, so the maximal
  // low-innovation set is {f : |xp - Predict| < ransac_thresh_}; inn() == xp - Predict at this state.
  std::vector<bool> low(n);
  int n_low = 0;
  for (int i = 0; i < n; ++i) {
    const Vec2& r = mh_inliers[i]->inn();
    low[i] = std::sqrt(r(0) * r(0) + r(1) * r(1)) < ransac_thresh_;
    n_low += low[i];
  }
  ransac_chi2_.assign(n, -1.0);
  num_oneptransac_rejected_ = 0;
  if (n_low == n) return mh_inliers;                                    // :263-265
  // BackupState (estimator.cpp:1410-1449)
  const MatX P0 = P_;
  const Mat3 Rsb0 = Rsb_, Rbc0 = Rbc_, Rsg0 = Rsg_;
  const Vec3 Tsb0 = Tsb_, Vsb0 = Vsb_, bg0 = bg_, ba0 = ba_, Tbc0 = Tbc_;
  std::vector<Group> g0; for (Group* g : groups_) g0.push_back(g ? *g : Group());
  std::vector<Vec3> x0; for (Feature* f : mh_inliers) x0.push_back(f->x_);
  const std::vector<FeaturePtr> instate0 = instate_features_;
#if XIVO_HIP_ONLINE_CALIB
  const number_t td0 = td_; const Mat3 Cg0 = Cg_, Ca0 = Ca_;           // imu_.BackupState / Camera::BackupState (estimator.cpp:1421-1427)
  number_t intr0[9]; std::memcpy(intr0, intr_, sizeof(intr0));
#endif

  std::vector<Group*> groups_low, active;
  auto has = [](const std::vector<Group*>& v, Group* g) { for (Group* q : v) if (q == g) return true; return false; };
  for (int i = 0; i < n; ++i) {
    if (!has(active, mh_inliers[i]->ref())) active.push_back(mh_inliers[i]->ref());
    if (low[i] && !has(groups_low, mh_inliers[i]->ref())) groups_low.push_back(mh_inliers[i]->ref());
  }
  if (n_low > 0) {
    if (!has(groups_low, gauge_group_ptr_)) {                           // :292-301, FindNewRefGroup estimator.cpp:1394-1407
      Group* best = nullptr; double bestc = 0;
      for (Group* g : groups_low) {
        double c = 0; const int off = lay_.group_begin + 6 * g->sind();
        for (int d = 0; d < 6; ++d) c += P_(off + d, off + d);
        if (!best || c < bestc) { best = g; bestc = c; }
      }
      zero_rc(P_, lay_.group_begin + 6 * best->sind(), 6);
    }
    for (int i = 0; i < n; ++i)                                         // :304-310
      if (!low[i]) zero_rc(P_, lay_.feature_begin + 3 * mh_inliers[i]->sind(), 3);
    for (Group* g : active)                                             // :311-317
      if (!has(groups_low, g)) zero_rc(P_, lay_.group_begin + 6 * g->sind(), 6);
    device_P_current_ = false;                                          // P_ was edited on the host just above
    H_.setZero(2 * n_low, size); inn_.setZero(2 * n_low); diagR_.resize(2 * n_low);
    int c = 0;
    for (int i = 0; i < n; ++i) {
      if (!low[i]) continue;
      const MatX& J = mh_inliers[i]->J();
      for (int j = 0; j < size; ++j) { H_(2 * c, j) = J(0, j); H_(2 * c + 1, j) = J(1, j); }   // :326 full row
      inn_(2 * c) = mh_inliers[i]->inn()(0); inn_(2 * c + 1) = mh_inliers[i]->inn()(1);
      diagR_(2 * c) = R_; diagR_(2 * c + 1) = R_;
      ++c;
    }
    UpdateJosephForm();                                                 // :332
    AbsorbError();                                                      // :333
  }
  // rescue high-innovation measurements (:338-377): Jacobians at the updated state + chi-square, on the device
  std::vector<FeaturePtr> hi;
  for (int i = 0; i < n; ++i) if (!low[i]) hi.push_back(mh_inliers[i]);
  instate_features_ = hi;
  ComputeInstateJacobians();
  if (!(trust_device_P_ && device_P_current_)) Check(xivo_hip_upload_P(ctx_, 0, 1, P_.data(), (long)size * size, size), "upload_P");
  device_P_current_ = true;
  std::vector<unsigned char> mask(hi.size());
  std::vector<double> dist(hi.size());
  Check(xivo_hip_mh_gate(ctx_, 1, R_, ransac_Chi2_, 1.0, 0, mask.data(), dist.data()), "mh_gate");
  std::vector<FeaturePtr> output;
  for (int i = 0; i < n; ++i) if (low[i]) output.push_back(mh_inliers[i]);
  int h = 0;
  for (int i = 0; i < n; ++i) {
    if (low[i]) continue;
    ransac_chi2_[i] = dist[h];
    if (dist[h] < ransac_Chi2_) output.push_back(mh_inliers[i]);        // :357-358
    else { mh_inliers[i]->SetStatus(FeatureStatus::REJECTED_BY_FILTER); ++num_oneptransac_rejected_; }   // :364-367
    ++h;
  }
  // RestoreState + re-compute Jacobians at the original state (:383-387)
  device_P_current_ = false;
  P_ = P0; Rsb_ = Rsb0; Rbc_ = Rbc0; Rsg_ = Rsg0; Tsb_ = Tsb0; Vsb_ = Vsb0; bg_ = bg0; ba_ = ba0; Tbc_ = Tbc0;
  for (size_t g = 0; g < groups_.size(); ++g) if (groups_[g]) *groups_[g] = g0[g];
#if XIVO_HIP_ONLINE_CALIB
  td_ = td0; Cg_ = Cg0; Ca_ = Ca0; std::memcpy(intr_, intr0, sizeof(intr0));
#endif
  for (int i = 0; i < n; ++i) mh_inliers[i]->x_ = x0[i];
  instate_features_ = mh_inliers;
  ComputeInstateJacobians();
  instate_features_ = instate0;
  return output;
}

// ---------------------------------------------------------------------------
// Propagation: host stages + device tail
// ---------------------------------------------------------------------------
namespace {
constexpr int NM = 23;   // kMotionSize (src/core.h)
struct M23 {             // column-major 23x23
  double v[NM * NM];
  M23() { std::memset(v, 0, sizeof(v)); }
  double& operator()(int i, int j) { return v[j * NM + i]; }
  double operator()(int i, int j) const { return v[j * NM + i]; }
};
M23 mul(const M23& a, const M23& b) {
  M23 c;
  for (int j = 0; j < NM; ++j)
    for (int k = 0; k < NM; ++k) {
      const double bkj = b(k, j);
      if (bkj == 0.0) continue;
      for (int i = 0; i < NM; ++i) c(i, j) += a(i, k) * bkj;
    }
  return c;
}
M23 transpose(const M23& a) { M23 c; for (int i = 0; i < NM; ++i) for (int j = 0; j < NM; ++j) c(i, j) = a(j, i); return c; }
M23 axpby(double alpha, const M23& a, double beta, const M23& b) { M23 c; for (int e = 0; e < NM * NM; ++e) c.v[e] = alpha * a.v[e] + beta * b.v[e]; return c; }
Vec3 mulv(const Mat3& R, const Vec3& x) { Vec3 r; for (int i = 0; i < 3; ++i) r(i) = R(i, 0) * x(0) + R(i, 1) * x(1) + R(i, 2) * x(2); return r; }
Mat3 mulm(const Mat3& a, const Mat3& b) { Mat3 c; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j); return c; }
Mat3 hat3(const Vec3& w) { Mat3 m; m(0,0)=0; m(0,1)=-w(2); m(0,2)=w(1); m(1,0)=w(2); m(1,1)=0; m(1,2)=-w(0); m(2,0)=-w(1); m(2,1)=w(0); m(2,2)=0; return m; }
Mat3 so3_exp(const Vec3& w) {   // Sophus::SO3::exp (Rodrigues)
  const double th = std::sqrt(w(0) * w(0) + w(1) * w(1) + w(2) * w(2));
  Mat3 W = hat3(w), W2 = mulm(W, W), R;
  const double a = th < 1e-10 ? 1.0 : std::sin(th) / th, b = th < 1e-10 ? 0.5 : (1 - std::cos(th)) / (th * th);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = (i == j) + a * W(i, j) + b * W2(i, j);
  return R;
}
struct MState { Mat3 Rsb; Vec3 Tsb, Vsb, bg, ba; Mat3 Rsg; };
struct Tableau { int S; double a[7][7]; double c_step[7], c_imu[7], b[7]; };
const Tableau kRK4 = {4, {{0}, {0.5}, {0, 0.5}, {0, 0, 1.0}}, {0, 0.5, 0.5, 1.0}, {0, 0.5, 0.5, 0.5},
                      {1 / 6.0, 2 / 6.0, 2 / 6.0, 1 / 6.0}};                                   // src/rk4.cpp:35-96
const Tableau kPD = {7,
                     {{0}, {2 / 9.0}, {1 / 12.0, 3 / 12.0}, {55 / 324.0, -75 / 324.0, 200 / 324.0},
                      {83 / 330.0, -195 / 330.0, 305 / 330.0, 27 / 330.0},
                      {-19 / 28.0, 63 / 28.0, 4 / 28.0, -108 / 28.0, 88 / 28.0},
                      {38 / 400.0, 0, 240 / 400.0, -243 / 400.0, 330 / 400.0, 35 / 400.0}},
                     {0, 2 / 9.0, 3 / 9.0, 5 / 9.0, 6 / 9.0, 1, 1}, {0, 2 / 9.0, 3 / 9.0, 5 / 9.0, 6 / 9.0, 1, 1},
                     {0.0862, 0, 0.6660, -0.7857, 0.9570, 0.0965, -0.0200}};                   // src/princedormand.cpp:85-200

// Estimator::ComposeMotion (src/estimator.cpp:598-613), Cg = Ca = I (default build)
void ComposeMotion(MState& X, const Vec3& V, const Vec3& gyro, const Vec3& accel, double dt, const Vec3& g) {
  Vec3 gc, ac;
  for (int i = 0; i < 3; ++i) { gc(i) = gyro(i) - X.bg(i); ac(i) = accel(i) - X.ba(i); }
  const Vec3 Ra = mulv(X.Rsb, ac), Rg = mulv(X.Rsg, g);
  Vec3 w;
  for (int i = 0; i < 3; ++i) { X.Tsb(i) += V(i) * dt; X.Vsb(i) += (Ra(i) + Rg(i)) * dt; w(i) = gc(i) * dt; }
  X.Rsb = mulm(X.Rsb, so3_exp(w));
}
// Estimator::ComputeMotionJacobianAt (src/estimator.cpp:615-704) -> dense F (23x23), G (23x12)
void MotionJacobian(const MState& X, const Vec3& gyro, const Vec3& accel, const Vec3& g, M23& F, double G[NM][12]) {
  Vec3 gc, ac;
  for (int i = 0; i < 3; ++i) { gc(i) = gyro(i) - X.bg(i); ac(i) = accel(i) - X.ba(i); }
  const Mat3 hg = hat3(gc), ha = hat3(ac), hgr = hat3(g);
  const Mat3 dV_dW = mulm(X.Rsb, ha), dV_dWsg = mulm(X.Rsb, hgr);
  F = M23();
  std::memset(G, 0, sizeof(double) * NM * 12);
  for (int j = 0; j < 3; ++j) {
    F(0 + j, 9 + j) = -1; F(3 + j, 6 + j) = 1;
    for (int i = 0; i < 3; ++i) {
      F(0 + i, 0 + j) = -hg(i, j);
      F(6 + i, 0 + j) = -dV_dW(i, j);
      F(6 + i, 12 + j) = -X.Rsb(i, j);
      if (j < 2) F(6 + i, 21 + j) = -dV_dWsg(i, j);
    }
    G[0 + j][j] = -1; G[9 + j][6 + j] = 1; G[12 + j][9 + j] = 1;
    for (int i = 0; i < 3; ++i) G[6 + i][3 + j] = -X.Rsb(i, j);
  }
}
// one RK4Step / PrinceDormandStep on the motion block; Phi_out = I + FK dt
void IntegratorStep(const Tableau& tb, MState& X, M23& Pmm, M23& Phi_out, const Vec3& gyro0, const Vec3& accel0,
                    const Vec3& sg, const Vec3& sa, double dt, const MatX& Qimu, const Vec3& g) {
  Vec3 Ks[7]; M23 FKs[7], PKs[7];
  for (int i = 0; i < tb.S; ++i) {
    MState X0 = X;
    Vec3 gy, ac;
    for (int c = 0; c < 3; ++c) { gy(c) = gyro0(c) + sg(c) * tb.c_imu[i] * dt; ac(c) = accel0(c) + sa(c) * tb.c_imu[i] * dt; }
    M23 Fs, Ps;
    if (i > 0) {
      Vec3 V;
      for (int j = 0; j < i; ++j) {
        for (int c = 0; c < 3; ++c) V(c) += tb.a[i][j] * Ks[j](c);
        Fs = axpby(1.0, Fs, tb.a[i][j], FKs[j]);
        Ps = axpby(1.0, Ps, tb.a[i][j], PKs[j]);
      }
      ComposeMotion(X0, V, gy, ac, tb.c_step[i] * dt, g);
    }
    M23 F; double G[NM][12];
    MotionJacobian(X0, gy, ac, g, F, G);
    Ks[i] = X0.Vsb;
    FKs[i] = i == 0 ? F : axpby(1.0, F, dt, mul(F, Fs));
    const M23 P0 = i == 0 ? Pmm : axpby(1.0, Pmm, dt, Ps);
    const M23 FP = mul(F, P0);
    M23 GQG;
    for (int r = 0; r < NM; ++r)
      for (int c = 0; c < NM; ++c) {
        double s = 0;
        for (int p = 0; p < 12; ++p) {
          if (G[r][p] == 0.0) continue;
          for (int q = 0; q < 12; ++q) s += G[r][p] * Qimu(p, q) * G[c][q];
        }
        GQG(r, c) = s;
      }
    PKs[i] = axpby(1.0, axpby(1.0, FP, 1.0, mul(P0, transpose(F))), 1.0, GQG);
  }
  Vec3 K; M23 FK, PK;
  for (int i = 0; i < tb.S; ++i) {
    for (int c = 0; c < 3; ++c) K(c) += tb.b[i] * Ks[i](c);
    FK = axpby(1.0, FK, tb.b[i], FKs[i]);
    PK = axpby(1.0, PK, tb.b[i], PKs[i]);
  }
  Vec3 gy, ac;
  for (int c = 0; c < 3; ++c) { gy(c) = gyro0(c) + sg(c) * dt; ac(c) = accel0(c) + sa(c) * dt; }
  ComposeMotion(X, K, gy, ac, dt, g);
  Pmm = axpby(1.0, Pmm, dt, PK);
  Phi_out = M23();
  for (int i = 0; i < NM; ++i) Phi_out(i, i) = 1.0;
  Phi_out = axpby(1.0, Phi_out, dt, FK);
}
}  // namespace

void Estimator::Propagate(bool visual_meas, number_t dt) {
  if (dt == 0) return;                                    // estimator.cpp:551-556
#if XIVO_HIP_ONLINE_CALIB
  // Online-calibration builds: kMotionSize is 24 / 38 / 39 and ComputeMotionJacobianAt carries the dWsb/dCg and dVsb/dCa columns
  // (src/estimator.cpp:626-638, :674-688); the adapter hands the whole call to the device-native integrator
  // (xivo_hip_propagate_calib: RK4 / Dormand-Prince with the sub-stepping of src/rk4.cpp:13-32 on the resident state).
  {
    xivo_imu_in s{};
    if (!visual_meas) {                                   // :558-568
      for (int i = 0; i < 3; ++i) { slope_accel_(i) = (curr_accel_(i) - last_accel_(i)) / dt; slope_gyro_(i) = (curr_gyro_(i) - last_gyro_(i)) / dt; }
    }
    std::memcpy(s.gyro, last_gyro_.data(), 24); std::memcpy(s.accel, last_accel_.data(), 24);
    std::memcpy(s.slope_gyro, slope_gyro_.data(), 24); std::memcpy(s.slope_accel, slope_accel_.data(), 24); s.dt = dt;
    if (!visual_meas) { last_accel_ = curr_accel_; last_gyro_ = curr_gyro_; }
    else for (int i = 0; i < 3; ++i) { last_accel_(i) += slope_accel_(i) * dt; last_gyro_(i) += slope_gyro_(i) * dt; }   // :569-575
    xivo_prop_opts o{};
    if (Qimu_.rows() != 12 || Qmodel_.rows() != motion_) throw std::invalid_argument("Propagate: Qimu_ 12 x 12, Qmodel_ kMotionSize x kMotionSize");
    std::memcpy(o.Qimu, Qimu_.data(), sizeof(o.Qimu)); std::memcpy(o.g, g_.data(), sizeof(o.g));
    if (integration_method_ == "PrinceDormand") o.method = 1;
    else if (integration_method_ == "RK4") o.method = 0;
    else throw std::runtime_error("Unknown integration method");
    o.stepsize = stepsize_;
    xivo_pose_in pose;
    std::memcpy(pose.Rsb, Rsb_.data(), 72); std::memcpy(pose.Tsb, Tsb_.data(), 24); std::memcpy(pose.Rbc, Rbc_.data(), 72);
    std::memcpy(pose.Tbc, Tbc_.data(), 24); std::memcpy(pose.Vsb, Vsb_.data(), 24); std::memcpy(pose.bg, bg_.data(), 24);
    std::memcpy(pose.ba, ba_.data(), 24); std::memcpy(pose.Rsg, Rsg_.data(), 72);
    std::vector<xivo_group_in> gs(lay_.n_groups);
    for (auto& g : gs) { const Mat3 I = Identity3(); std::memcpy(g.Rsb, I.data(), 72); g.Tsb[0] = g.Tsb[1] = g.Tsb[2] = 0; }
    xivo_feat_in none{}; none.sind = -1; none.ref_sind = 0;
    Check(xivo_hip_set_scene(ctx_, 0, 1, 1, &pose, gs.data(), &none), "set_scene");
    xivo_calib_in cs{};
    std::memcpy(cs.gyro, s.gyro, 24); std::memcpy(cs.Cg, Cg_.data(), 72); cs.td = td_; std::memcpy(cs.Ca, Ca_.data(), 72);
    std::memcpy(cs.intr, intr_, sizeof(cs.intr));
    Check(xivo_hip_set_calib_state(ctx_, 0, 1, &cs), "set_calib_state");
    const int N = lay_.N;
    Check(xivo_hip_upload_P(ctx_, 0, 1, P_.data(), (long)N * N, N), "upload_P");
    Check(xivo_hip_propagate_calib(ctx_, 0, 1, 1, &s, &o, Qmodel_.data()), "propagate_calib");
    Check(xivo_hip_get_scene(ctx_, 0, 1, &pose, nullptr, nullptr), "get_scene");
    std::memcpy(Rsb_.data(), pose.Rsb, 72); std::memcpy(Tsb_.data(), pose.Tsb, 24); std::memcpy(Vsb_.data(), pose.Vsb, 24);
    Check(xivo_hip_download_P(ctx_, 0, 1, P_.data(), (long)N * N, N), "download_P");
    device_P_current_ = true;
    return;
  }
#endif
  Vec3 accel0, gyro0;
  if (!visual_meas) {                                     // :558-568
    for (int i = 0; i < 3; ++i) {
      slope_accel_(i) = (curr_accel_(i) - last_accel_(i)) / dt;
      slope_gyro_(i) = (curr_gyro_(i) - last_gyro_(i)) / dt;
    }
    accel0 = last_accel_; gyro0 = last_gyro_;
    last_accel_ = curr_accel_; last_gyro_ = curr_gyro_;
  } else {                                                // :569-575
    accel0 = last_accel_; gyro0 = last_gyro_;
    for (int i = 0; i < 3; ++i) {
      last_accel_(i) = accel0(i) + slope_accel_(i) * dt;
      last_gyro_(i) = gyro0(i) + slope_gyro_(i) * dt;
    }
  }
  const Tableau* tb;
  if (integration_method_ == "PrinceDormand") tb = &kPD;
  else if (integration_method_ == "RK4") tb = &kRK4;
  else throw std::runtime_error("Unknown integration method");        // LOG(FATAL), estimator.cpp:587
  const int N = lay_.N;
  MState X{Rsb_, Tsb_, Vsb_, bg_, ba_, Rsg_};
  M23 Pmm, PhiAcc;
  for (int i = 0; i < NM; ++i) { PhiAcc(i, i) = 1.0; for (int j = 0; j < NM; ++j) Pmm(i, j) = P_(i, j); }
  auto one = [&](const Vec3& gy, const Vec3& ac, double h) {
    M23 Phi;
    IntegratorStep(*tb, X, Pmm, Phi, gy, ac, slope_gyro_, slope_accel_, h, Qimu_, g_);
    PhiAcc = mul(Phi, PhiAcc);
  };
  if (stepsize_ < 0) {
    one(gyro0, accel0, dt);
  } else {                                                // rk4.cpp:16-31 / princedormand.cpp:62-81
    number_t total = 0;
    Vec3 gy = gyro0, ac = accel0;
    while (total < dt) {
      number_t h = stepsize_;
      if (total + h > dt) h = dt - total;
      else if (total + h + 0.5 * h > dt) h = 0.5 * h;     // half step trick
      one(gy, ac, h);
      for (int i = 0; i < 3; ++i) { gy(i) += slope_gyro_(i) * h; ac(i) += slope_accel_(i) * h; }
      total += h;
    }
  }
  for (int i = 0; i < NM; ++i) for (int j = 0; j < NM; ++j) Pmm(i, j) += Qmodel_(i, j);   // estimator.cpp:590
  Rsb_ = X.Rsb; Tsb_ = X.Tsb; Vsb_ = X.Vsb;
  // device: P_mm <- Pmm ; P_ms <- Phi P_ms ; P_sm <- P_sm Phi^T  (src/rk4.cpp:92-102, accumulated)
  Check(xivo_hip_upload_P(ctx_, 0, 1, P_.data(), (long)N * N, N), "upload_P");
  Check(xivo_hip_propagate_cov(ctx_, 0, 1, NM, PhiAcc.v, Pmm.v), "propagate_cov");
  Check(xivo_hip_download_P(ctx_, 0, 1, P_.data(), (long)N * N, N), "download_P");
  device_P_current_ = true;
}

}  // namespace hip
}  // namespace xivo

// ---------------------------------------------------------------------------
// C shim so the adapter can be exercised from pytest (tests/test_host_adapter_gpu.py)
// ---------------------------------------------------------------------------
extern "C" int xivo_host_selftest_update_step(const xivo_layout* lay, const xivo_cam* cam, unsigned flags, int F,
                                              const xivo_pose_in* pose, const xivo_group_in* groups,
                                              const xivo_feat_in* feats, double* P_inout, double R, double mh_thresh,
                                              double mh_mult, int min_inliers, double* err_out,
                                              unsigned char* inlier_mask_out, int* num_mh_rejected_out, char* msg,
                                              int msg_len) {
  using namespace xivo::hip;
  try {
    Estimator est(*lay, *cam, F, flags);
    const int N = lay->N;
    std::memcpy(est.P_.data(), P_inout, sizeof(double) * N * N);
    std::memcpy(est.Rsb_.data(), pose->Rsb, 72); std::memcpy(est.Tsb_.data(), pose->Tsb, 24);
    std::memcpy(est.Rbc_.data(), pose->Rbc, 72); std::memcpy(est.Tbc_.data(), pose->Tbc, 24);
    est.R_ = R; est.MH_thresh_ = mh_thresh; est.MH_thresh_multipler_ = mh_mult; est.min_required_inliers_ = min_inliers;
    std::vector<Group> gs(lay->n_groups);
    for (int g = 0; g < lay->n_groups; ++g) {
      std::memcpy(gs[g].Rsb_.data(), groups[g].Rsb, 72); std::memcpy(gs[g].Tsb_.data(), groups[g].Tsb, 24);
      gs[g].sind_ = g; est.groups_[g] = &gs[g];
    }
    std::vector<Feature> fs(F);
    for (int i = 0; i < F; ++i) {
      std::memcpy(fs[i].x_.data(), feats[i].x, 24); std::memcpy(fs[i].back_.data(), feats[i].xp, 16);
      fs[i].ref_ = &gs[feats[i].ref_sind]; fs[i].sind_ = feats[i].sind;
      est.instate_features_.push_back(&fs[i]);
    }
    // the numeric core of Estimator::UpdateStep (src/manager.cpp:72-104)
    est.ComputeInstateJacobians();
    std::vector<FeaturePtr> inliers;
    if (est.use_MH_gating_ && (int)est.instate_features_.size() > est.min_required_inliers_)  // manager.cpp:635
      inliers = est.MHGating();
    else
      inliers = est.instate_features_;
    est.in_current_ekf_update_ = inliers;
    est.FilterUpdate();
    std::memcpy(P_inout, est.P_.data(), sizeof(double) * N * N);
    std::memcpy(err_out, est.err_.data(), sizeof(double) * N);
    for (int i = 0; i < F; ++i) inlier_mask_out[i] = fs[i].status() != FeatureStatus::REJECTED_BY_FILTER;
    *num_mh_rejected_out = est.num_mh_rejected_;
    return 0;
  } catch (const std::exception& e) {
    if (msg && msg_len > 0) { std::strncpy(msg, e.what(), msg_len - 1); msg[msg_len - 1] = 0; }
    return -1;
  }
}

#if XIVO_HIP_ONLINE_CALIB
// The same step in an online-calibration build (libxivo_host_calib.so): calib in / out = the calibration state before the
// step and after AbsorbError; also Propagate over one IMU sample first when n_imu = 1 (imu, Qimu 12 x 12, Qmodel kMotionSize^2).
extern "C" int xivo_host_selftest_update_step_calib(const xivo_layout* lay, const xivo_cam* cam, unsigned flags, int F,
                                                    xivo_pose_in* pose, const xivo_group_in* groups, const xivo_feat_in* feats,
                                                    xivo_calib_in* calib, double* P_inout, double R, double mh_thresh, double mh_mult,
                                                    int min_inliers, int use_ransac, double ransac_thresh, double ransac_chi2,
                                                    double* err_out, unsigned char* inlier_mask_out, int absorb,
                                                    const xivo_imu_in* imu, const double* Qimu, const double* Qmodel, const double* g_vec,
                                                    int method, int* slots_out, char* msg, int msg_len) {
  using namespace xivo::hip;
  try {
    Estimator est(*lay, *cam, F, flags);
    const int N = lay->N;
    slots_out[0] = est.calib_layout().td; slots_out[1] = est.calib_layout().Cg; slots_out[2] = est.calib_layout().cam_begin;
    slots_out[3] = est.calib_layout().cam_dim; slots_out[4] = est.motion_size();
    std::memcpy(est.P_.data(), P_inout, sizeof(double) * N * N);
    std::memcpy(est.Rsb_.data(), pose->Rsb, 72); std::memcpy(est.Tsb_.data(), pose->Tsb, 24);
    std::memcpy(est.Rbc_.data(), pose->Rbc, 72); std::memcpy(est.Tbc_.data(), pose->Tbc, 24);
    std::memcpy(est.Vsb_.data(), pose->Vsb, 24); std::memcpy(est.bg_.data(), pose->bg, 24); std::memcpy(est.ba_.data(), pose->ba, 24);
    std::memcpy(est.Rsg_.data(), pose->Rsg, 72);
    std::memcpy(est.last_gyro_.data(), calib->gyro, 24); std::memcpy(est.Cg_.data(), calib->Cg, 72); est.td_ = calib->td;
    std::memcpy(est.Ca_.data(), calib->Ca, 72); std::memcpy(est.intr_, calib->intr, sizeof(est.intr_));
    est.R_ = R; est.MH_thresh_ = mh_thresh; est.MH_thresh_multipler_ = mh_mult; est.min_required_inliers_ = min_inliers;
    est.ransac_thresh_ = ransac_thresh; est.ransac_Chi2_ = ransac_chi2;
    if (imu) {
      std::memcpy(est.last_gyro_.data(), imu->gyro, 24); std::memcpy(est.last_accel_.data(), imu->accel, 24);
      std::memcpy(est.slope_gyro_.data(), imu->slope_gyro, 24); std::memcpy(est.slope_accel_.data(), imu->slope_accel, 24);
      std::memcpy(est.g_.data(), g_vec, 24);
      est.Qimu_.setZero(12, 12); std::memcpy(est.Qimu_.data(), Qimu, sizeof(double) * 144);
      const int nm = est.motion_size();
      est.Qmodel_.setZero(nm, nm); std::memcpy(est.Qmodel_.data(), Qmodel, sizeof(double) * nm * nm);
      est.integration_method_ = method == 0 ? "RK4" : "PrinceDormand";
      est.Propagate(true, imu->dt);
    }
    std::vector<Group> gs(lay->n_groups);
    for (int g = 0; g < lay->n_groups; ++g) {
      std::memcpy(gs[g].Rsb_.data(), groups[g].Rsb, 72); std::memcpy(gs[g].Tsb_.data(), groups[g].Tsb, 24);
      gs[g].sind_ = g; est.groups_[g] = &gs[g]; est.instate_groups_.push_back(&gs[g]);
    }
    std::vector<Feature> fs(F);
    for (int i = 0; i < F; ++i) {
      std::memcpy(fs[i].x_.data(), feats[i].x, 24); std::memcpy(fs[i].back_.data(), feats[i].xp, 16);
      fs[i].ref_ = &gs[feats[i].ref_sind]; fs[i].sind_ = feats[i].sind;
      est.instate_features_.push_back(&fs[i]);
    }
    est.ComputeInstateJacobians();
    std::vector<FeaturePtr> inliers;
    if (est.use_MH_gating_ && (int)est.instate_features_.size() > est.min_required_inliers_) inliers = est.MHGating();
    else inliers = est.instate_features_;
    if (use_ransac) { est.gauge_group_ptr_ = &gs[0]; inliers = est.OnePointRANSAC(inliers); }
    est.in_current_ekf_update_ = inliers;
    est.FilterUpdate();
    std::memcpy(err_out, est.err_.data(), sizeof(double) * N);
    if (absorb) est.AbsorbError();
    std::memcpy(P_inout, est.P_.data(), sizeof(double) * N * N);
    for (int i = 0; i < F; ++i) inlier_mask_out[i] = 0;
    for (FeaturePtr f : inliers) inlier_mask_out[f - &fs[0]] = 1;
    std::memcpy(pose->Rsb, est.Rsb_.data(), 72); std::memcpy(pose->Tsb, est.Tsb_.data(), 24); std::memcpy(pose->Vsb, est.Vsb_.data(), 24);
    std::memcpy(pose->bg, est.bg_.data(), 24); std::memcpy(pose->Rbc, est.Rbc_.data(), 72); std::memcpy(pose->Tbc, est.Tbc_.data(), 24);
    std::memcpy(calib->Cg, est.Cg_.data(), 72); calib->td = est.td_; std::memcpy(calib->Ca, est.Ca_.data(), 72);
    std::memcpy(calib->intr, est.intr_, sizeof(est.intr_));
    return 0;
  } catch (const std::exception& e) {
    if (msg && msg_len > 0) { std::strncpy(msg, e.what(), msg_len - 1); msg[msg_len - 1] = 0; }
    return -1;
  }
}
#endif

// state30 = [Rsb(9 col-major) Tsb Vsb bg ba Rsg(9)] in/out; imu18 = [last_gyro last_accel curr_gyro curr_accel slope_gyro slope_accel]
extern "C" int xivo_host_selftest_propagate(int N, int use_rk4, int visual_meas, double dt, double stepsize, double* state30,
                                            double* P_inout, double* imu18, const double* Qimu, const double* Qmodel,
                                            const double* g_vec, char* msg, int msg_len) {
  using namespace xivo::hip;
  try {
    xivo_layout lay{N, 23, 1, 29, (N - 29) / 3 > 0 ? (N - 29) / 3 : 1};
    xivo_cam cam{}; cam.model = XIVO_CAM_PINHOLE; cam.fx = cam.fy = 500; cam.cx = cam.cy = 250; cam.rows = cam.cols = 500;
    Estimator est(lay, cam, 1, 0);
    std::memcpy(est.P_.data(), P_inout, sizeof(double) * N * N);
    std::memcpy(est.Rsb_.data(), state30, 72); std::memcpy(est.Tsb_.data(), state30 + 9, 24); std::memcpy(est.Vsb_.data(), state30 + 12, 24);
    std::memcpy(est.bg_.data(), state30 + 15, 24); std::memcpy(est.ba_.data(), state30 + 18, 24); std::memcpy(est.Rsg_.data(), state30 + 21, 72);
    std::memcpy(est.last_gyro_.data(), imu18, 24); std::memcpy(est.last_accel_.data(), imu18 + 3, 24);
    std::memcpy(est.curr_gyro_.data(), imu18 + 6, 24); std::memcpy(est.curr_accel_.data(), imu18 + 9, 24);
    std::memcpy(est.slope_gyro_.data(), imu18 + 12, 24); std::memcpy(est.slope_accel_.data(), imu18 + 15, 24);
    std::memcpy(est.g_.data(), g_vec, 24);
    est.Qimu_.setZero(12, 12); std::memcpy(est.Qimu_.data(), Qimu, sizeof(double) * 144);
    est.Qmodel_.setZero(23, 23); std::memcpy(est.Qmodel_.data(), Qmodel, sizeof(double) * 529);
    est.integration_method_ = use_rk4 ? "RK4" : "PrinceDormand";
    est.stepsize_ = stepsize;
    est.Propagate(visual_meas != 0, dt);
    std::memcpy(P_inout, est.P_.data(), sizeof(double) * N * N);
    std::memcpy(state30, est.Rsb_.data(), 72); std::memcpy(state30 + 9, est.Tsb_.data(), 24); std::memcpy(state30 + 12, est.Vsb_.data(), 24);
    std::memcpy(imu18, est.last_gyro_.data(), 24); std::memcpy(imu18 + 3, est.last_accel_.data(), 24);
    std::memcpy(imu18 + 12, est.slope_gyro_.data(), 24); std::memcpy(imu18 + 15, est.slope_accel_.data(), 24);
    return 0;
  } catch (const std::exception& e) {
    if (msg && msg_len > 0) { std::strncpy(msg, e.what(), msg_len - 1); msg[msg_len - 1] = 0; }
    return -1;
  }
}

extern "C" int xivo_host_selftest_ransac(const xivo_layout* lay, const xivo_cam* cam, int F, const xivo_pose_in* pose,
                                         const xivo_group_in* groups, const xivo_feat_in* feats, const double* P_in, double R,
                                         double ransac_thresh, double ransac_chi2, int gauge_group,
                                         unsigned char* kept_out, double* chi2_out, int* num_rejected_out,
                                         double* restore_err_out, char* msg, int msg_len) {
  using namespace xivo::hip;
  try {
    Estimator est(*lay, *cam, F, 0);
    const int N = lay->N;
    std::memcpy(est.P_.data(), P_in, sizeof(double) * N * N);
    std::memcpy(est.Rsb_.data(), pose->Rsb, 72); std::memcpy(est.Tsb_.data(), pose->Tsb, 24);
    std::memcpy(est.Rbc_.data(), pose->Rbc, 72); std::memcpy(est.Tbc_.data(), pose->Tbc, 24);
    est.R_ = R; est.ransac_thresh_ = ransac_thresh; est.ransac_Chi2_ = ransac_chi2;
    std::vector<Group> gs(lay->n_groups);
    for (int g = 0; g < lay->n_groups; ++g) {
      std::memcpy(gs[g].Rsb_.data(), groups[g].Rsb, 72); std::memcpy(gs[g].Tsb_.data(), groups[g].Tsb, 24);
      gs[g].sind_ = g; est.groups_[g] = &gs[g]; est.instate_groups_.push_back(&gs[g]);
    }
    est.gauge_group_ptr_ = &gs[gauge_group];
    std::vector<Feature> fs(F);
    std::vector<FeaturePtr> all;
    for (int i = 0; i < F; ++i) {
      std::memcpy(fs[i].x_.data(), feats[i].x, 24); std::memcpy(fs[i].back_.data(), feats[i].xp, 16);
      fs[i].ref_ = &gs[feats[i].ref_sind]; fs[i].sind_ = feats[i].sind;
      all.push_back(&fs[i]);
    }
    est.instate_features_ = all;
    est.ComputeInstateJacobians();
    std::vector<FeaturePtr> out = est.OnePointRANSAC(all);
    for (int i = 0; i < F; ++i) { kept_out[i] = 0; chi2_out[i] = est.ransac_chi2_.empty() ? -1.0 : est.ransac_chi2_[i]; }
    for (FeaturePtr f : out) kept_out[f - &fs[0]] = 1;
    *num_rejected_out = est.num_oneptransac_rejected_;
    double e = 0;
    for (int i = 0; i < N * N; ++i) e = std::fmax(e, std::fabs(est.P_.data()[i] - P_in[i]));
    for (int i = 0; i < 9; ++i) e = std::fmax(e, std::fabs(est.Rsb_.data()[i] - pose->Rsb[i]));
    for (int g = 0; g < lay->n_groups; ++g) for (int i = 0; i < 3; ++i) e = std::fmax(e, std::fabs(gs[g].Tsb_.data()[i] - groups[g].Tsb[i]));
    *restore_err_out = e;
    return 0;
  } catch (const std::exception& e) {
    if (msg && msg_len > 0) { std::strncpy(msg, e.what(), msg_len - 1); msg[msg_len - 1] = 0; }
    return -1;
  }
}

// Multi-frame replay of the per-frame numeric flow of the reference for ONE filter:
//   per frame: `n_imu` x Propagate(false) + Propagate(true) (src/estimator.cpp:475-592,1122),
//   ComputeInstateJacobians -> MHGating -> FilterUpdate -> AbsorbError (src/manager.cpp:72-104, update.cpp:145).
// imu: [n_frames][n_imu][6] (gyro, accel) samples, dt_imu between them; pixels: [n_frames][F][2].
// Outputs the final P, the final nominal motion state (30), group poses and feature states.
extern "C" int xivo_host_selftest_sequence(const xivo_layout* lay, const xivo_cam* cam, int F, int n_frames, int n_imu,
                                           double dt_imu, const double* imu, const double* pixels, double* state30,
                                           xivo_group_in* groups_io, xivo_feat_in* feats_io, const double* Rbc,
                                           const double* Tbc, double* P_inout, const double* Qimu, const double* Qmodel,
                                           const double* g_vec, int use_rk4, double R, int* inliers_per_frame, char* msg,
                                           int msg_len) {
  using namespace xivo::hip;
  try {
    Estimator est(*lay, *cam, F, 0);
    const int N = lay->N;
    std::memcpy(est.P_.data(), P_inout, sizeof(double) * N * N);
    std::memcpy(est.Rsb_.data(), state30, 72); std::memcpy(est.Tsb_.data(), state30 + 9, 24); std::memcpy(est.Vsb_.data(), state30 + 12, 24);
    std::memcpy(est.bg_.data(), state30 + 15, 24); std::memcpy(est.ba_.data(), state30 + 18, 24); std::memcpy(est.Rsg_.data(), state30 + 21, 72);
    std::memcpy(est.Rbc_.data(), Rbc, 72); std::memcpy(est.Tbc_.data(), Tbc, 24);
    std::memcpy(est.g_.data(), g_vec, 24);
    est.Qimu_.setZero(12, 12); std::memcpy(est.Qimu_.data(), Qimu, sizeof(double) * 144);
    est.Qmodel_.setZero(23, 23); std::memcpy(est.Qmodel_.data(), Qmodel, sizeof(double) * 529);
    est.integration_method_ = use_rk4 ? "RK4" : "PrinceDormand";
    est.R_ = R;
    std::vector<Group> gs(lay->n_groups);
    for (int g = 0; g < lay->n_groups; ++g) {
      std::memcpy(gs[g].Rsb_.data(), groups_io[g].Rsb, 72); std::memcpy(gs[g].Tsb_.data(), groups_io[g].Tsb, 24);
      gs[g].sind_ = g; est.groups_[g] = &gs[g]; est.instate_groups_.push_back(&gs[g]);
    }
    std::vector<Feature> fs(F);
    for (int i = 0; i < F; ++i) {
      std::memcpy(fs[i].x_.data(), feats_io[i].x, 24);
      fs[i].ref_ = &gs[feats_io[i].ref_sind]; fs[i].sind_ = feats_io[i].sind;
    }
    std::memcpy(est.last_gyro_.data(), imu, 24); std::memcpy(est.last_accel_.data(), imu + 3, 24);
    for (int t = 0; t < n_frames; ++t) {
      for (int k = 0; k < n_imu; ++k) {
        const double* s = imu + ((size_t)t * n_imu + k) * 6;
        std::memcpy(est.curr_gyro_.data(), s, 24); std::memcpy(est.curr_accel_.data(), s + 3, 24);
        if (t == 0 && k == 0) continue;                 // the first sample only seeds last_*
        est.Propagate(false, dt_imu);
      }
      est.Propagate(true, 0.5 * dt_imu);                // the image arrives half an IMU period later
      est.instate_features_.clear();
      for (int i = 0; i < F; ++i) {
        if (fs[i].status() == FeatureStatus::REJECTED_BY_FILTER) continue;   // removed from the state by the caller
        std::memcpy(fs[i].back_.data(), pixels + ((size_t)t * F + i) * 2, 16);
        est.instate_features_.push_back(&fs[i]);
      }
      est.ComputeInstateJacobians();
      std::vector<FeaturePtr> inl = ((int)est.instate_features_.size() > est.min_required_inliers_)
                                        ? est.MHGating() : est.instate_features_;
      inliers_per_frame[t] = (int)inl.size();
      est.in_current_ekf_update_ = inl;
      est.FilterUpdate();
      est.AbsorbError();
      // RemoveFeatureFromState for rejected ones (src/estimator.cpp:762-783): host edit of the authoritative P_
      for (int i = 0; i < F; ++i)
        if (fs[i].status() == FeatureStatus::REJECTED_BY_FILTER) {
          const int off = lay->feature_begin + 3 * fs[i].sind();
          for (int r = 0; r < 3; ++r) for (int q = 0; q < N; ++q) { est.P_(off + r, q) = 0.0; est.P_(q, off + r) = 0.0; }
        }
    }
    std::memcpy(P_inout, est.P_.data(), sizeof(double) * N * N);
    std::memcpy(state30, est.Rsb_.data(), 72); std::memcpy(state30 + 9, est.Tsb_.data(), 24); std::memcpy(state30 + 12, est.Vsb_.data(), 24);
    std::memcpy(state30 + 15, est.bg_.data(), 24); std::memcpy(state30 + 18, est.ba_.data(), 24); std::memcpy(state30 + 21, est.Rsg_.data(), 72);
    for (int g = 0; g < lay->n_groups; ++g) { std::memcpy(groups_io[g].Rsb, gs[g].Rsb_.data(), 72); std::memcpy(groups_io[g].Tsb, gs[g].Tsb_.data(), 24); }
    for (int i = 0; i < F; ++i) std::memcpy(feats_io[i].x, fs[i].x_.data(), 24);
    return 0;
  } catch (const std::exception& e) {
    if (msg && msg_len > 0) { std::strncpy(msg, e.what(), msg_len - 1); msg[msg_len - 1] = 0; }
    return -1;
  }
}

// Wall time of the literal drop-in call: xivo::hip::Estimator::UpdateJosephForm() with P_, H_, inn_, diagR_ in pageable
// host memory (what a caller at src/update.cpp:141 sees), one estimator, `n_calls` calls each from the same prior
// (P_ is restored on the host between calls, outside the timed region; a copy into P_ is a host edit, so the call uploads
// it again). mode: 0 = the one-call path (default of the adapter), 1 = the six-call sequence of rounds 1-3, 2 = one-call
// path with trust_device_P_ and KEEP-style residency (no upload when the device copy is current: the prior is restored
// with an explicit upload outside the timing). ms_out[n_calls] = per-call wall time; P_out / err_out = result of the last call.
extern "C" int xivo_host_time_update_joseph(int N, int M, const double* P0, const double* H, const double* inn, const double* diagR,
                                            int n_calls, int mode, unsigned flags, double* ms_out, double* P_out, double* err_out,
                                            char* msg, int msg_len) {
  using namespace xivo::hip;
  try {
    xivo_layout lay{N, 23, 1, 29, (N - 29) / 3 > 0 ? (N - 29) / 3 : 1};
    xivo_cam cam{}; cam.model = XIVO_CAM_PINHOLE; cam.fx = cam.fy = 500; cam.cx = cam.cy = 250; cam.rows = cam.cols = 500;
    Estimator est(lay, cam, (M + 1) / 2, flags);
    est.legacy_plumbing_ = mode == 1;
    est.trust_device_P_ = mode == 2;
    est.H_.setZero(M, N); std::memcpy(est.H_.data(), H, sizeof(double) * (size_t)M * N);
    est.inn_.setZero(M); std::memcpy(est.inn_.data(), inn, sizeof(double) * M);
    est.diagR_.setZero(M); std::memcpy(est.diagR_.data(), diagR, sizeof(double) * M);
    for (int it = 0; it < n_calls; ++it) {
      std::memcpy(est.P_.data(), P0, sizeof(double) * (size_t)N * N);
      est.InvalidateDeviceP();
      if (mode == 2) est.SyncDeviceP();               // resident flow: the prior is on the device already
      const auto t0 = std::chrono::steady_clock::now();
      est.UpdateJosephForm();
      const auto t1 = std::chrono::steady_clock::now();
      ms_out[it] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    }
    std::memcpy(P_out, est.P_.data(), sizeof(double) * (size_t)N * N);
    std::memcpy(err_out, est.err_.data(), sizeof(double) * N);
    return 0;
  } catch (const std::exception& e) {
    if (msg && msg_len > 0) { std::strncpy(msg, e.what(), msg_len - 1); msg[msg_len - 1] = 0; }
    return -1;
  }
}
