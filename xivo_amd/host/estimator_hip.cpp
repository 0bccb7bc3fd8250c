// Implementation of the reference-shaped adapter (estimator_hip.h) over the C ABI.
#include "estimator_hip.h"

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
}

void Estimator::UpdateJosephForm() {
  const int N = lay_.N, M = H_.rows();
  if (H_.cols() != N || inn_.size() != M || diagR_.size() != M || P_.rows() != N)
    throw std::invalid_argument("UpdateJosephForm: inconsistent sizes");
  Check(xivo_hip_upload_P(ctx_, 0, 1, P_.data(), (long)N * N, N), "upload_P");
  Check(xivo_hip_set_measurements(ctx_, 0, 1, M, H_.data(), (long)M * N, M, inn_.data(), M, diagR_.data(), M),
        "set_measurements");
  Check(xivo_hip_update_joseph(ctx_, 1), "update_joseph");
  err_.setZero(N);
  Check(xivo_hip_get_err(ctx_, 0, 1, err_.data(), N), "get_err");
  Check(xivo_hip_download_P(ctx_, 0, 1, P_.data(), (long)N * N, N), "download_P");
  int st = 0;
  Check(xivo_hip_get_status(ctx_, 0, 1, &st), "UpdateJosephForm (S = HPH^T + R)");
}

void Estimator::ComputeInstateJacobians() {
  const int F = (int)instate_features_.size();
  if (F == 0) return;
  if (F > max_features_) throw std::runtime_error("more in-state features than the context was created for");
  xivo_pose_in pose;
  std::memcpy(pose.Rsb, Rsb_.v, sizeof(pose.Rsb)); std::memcpy(pose.Tsb, Tsb_.v, sizeof(pose.Tsb));
  std::memcpy(pose.Rbc, Rbc_.v, sizeof(pose.Rbc)); std::memcpy(pose.Tbc, Tbc_.v, sizeof(pose.Tbc));
  std::vector<xivo_group_in> gs(lay_.n_groups);
  for (int g = 0; g < lay_.n_groups; ++g) {
    Mat3 I; Vec3 z;
    const Group* gp = groups_[g];
    std::memcpy(gs[g].Rsb, gp ? gp->Rsb_.v : I.v, sizeof(gs[g].Rsb));
    std::memcpy(gs[g].Tsb, gp ? gp->Tsb_.v : z.v, sizeof(gs[g].Tsb));
  }
  std::vector<xivo_feat_in> fs(F);
  for (int i = 0; i < F; ++i) {
    const Feature* f = instate_features_[i];
    std::memcpy(fs[i].x, f->x_.v, sizeof(fs[i].x));
    std::memcpy(fs[i].xp, f->back_.v, sizeof(fs[i].xp));
    fs[i].ref_sind = f->ref_->sind();
    fs[i].sind = f->sind_;
  }
  Check(xivo_hip_set_scene(ctx_, 0, 1, F, &pose, gs.data(), fs.data()), "set_scene");
  Check(xivo_hip_jacobians_instate(ctx_, 1), "jacobians_instate");
  std::vector<double> J((size_t)F * 42), inn((size_t)F * 2);
  Check(xivo_hip_get_jacobians(ctx_, 0, 1, J.data(), inn.data()), "get_jacobians");
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
  }
}

std::vector<FeaturePtr> Estimator::MHGating() {
  const int F = (int)instate_features_.size(), N = lay_.N;
  std::vector<FeaturePtr> inliers;
  num_mh_rejected_ = 0;
  if (F == 0) return inliers;
  Check(xivo_hip_upload_P(ctx_, 0, 1, P_.data(), (long)N * N, N), "upload_P");
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
    std::memcpy(est.Rsb_.v, pose->Rsb, 72); std::memcpy(est.Tsb_.v, pose->Tsb, 24);
    std::memcpy(est.Rbc_.v, pose->Rbc, 72); std::memcpy(est.Tbc_.v, pose->Tbc, 24);
    est.R_ = R; est.MH_thresh_ = mh_thresh; est.MH_thresh_multipler_ = mh_mult; est.min_required_inliers_ = min_inliers;
    std::vector<Group> gs(lay->n_groups);
    for (int g = 0; g < lay->n_groups; ++g) {
      std::memcpy(gs[g].Rsb_.v, groups[g].Rsb, 72); std::memcpy(gs[g].Tsb_.v, groups[g].Tsb, 24);
      gs[g].sind_ = g; est.groups_[g] = &gs[g];
    }
    std::vector<Feature> fs(F);
    for (int i = 0; i < F; ++i) {
      std::memcpy(fs[i].x_.v, feats[i].x, 24); std::memcpy(fs[i].back_.v, feats[i].xp, 16);
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
