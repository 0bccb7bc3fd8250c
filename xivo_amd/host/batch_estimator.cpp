// See batch_estimator.h. Decision for decision the frame loop of xivo_amd/sequence.py (SequenceRunner.frame, ImuFeeder).
#include "batch_estimator.h"
#include <stdio.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <cstdlib>
#include <numeric>
#include <stdexcept>
#include <string>

namespace xivo {
namespace hip {

namespace {
// a microsecond of book-keeping per filter: a handful of threads is all the loop can use (256 made it 8x slower) - and
// with one process per GPU never more than this rank's share of the host cores: the launcher (xivo_amd/shard.py:bind_rank)
// sets OMP_NUM_THREADS = min(8, cores / ranks)
static int team_size() {
  static const int n = [] {
    const char* e = std::getenv("OMP_NUM_THREADS");
    const int v = e ? std::atoi(e) : 8;
    return v < 1 ? 1 : (v > 8 ? 8 : v);
  }();
  return n;
}
#define kThreads team_size()
double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
xivo_edit_op make_op(int b, int kind, int i0 = 0, int i1 = 0, int i2 = 0) {
  xivo_edit_op o;
  std::memset(&o, 0, sizeof(o));
  o.b = b; o.kind = kind; o.i0 = i0; o.i1 = i1; o.i2 = i2;
  return o;
}
}  // namespace

void BatchEstimator::Check(int rc, const char* what) {
  // the reference LOG(FATAL)s / throws on these conditions (src/estimator.cpp:121,587,821,844)
  if (rc != XIVO_HIP_OK) throw std::runtime_error(std::string(what) + ": " + xivo_hip_strerror(rc));
}

BatchEstimator::BatchEstimator(const BatchConfig& cfg, int B, int device, const xivo_pose_in* poses0, const double* P0)
    : cfg_(cfg), B_(B) {
  if (cfg.cam.model != XIVO_CAM_PINHOLE)
    throw std::invalid_argument("point-cloud input initialises features with a pinhole un-projection");
  const int N = cfg.N(), F = cfg.n_features;
  Check(xivo_hip_create(&ctx_, device, N, 2 * F, B, (cfg.fix_group_block ? XIVO_HIP_FLAG_FIX_GROUP_BLOCK : 0u) |
                                                     (cfg.use_invdepth ? XIVO_HIP_FLAG_INVDEPTH : 0u)), "create");
  xivo_layout lay{N, 23, cfg.n_groups, 23 + 6 * cfg.n_groups, F};
  Check(xivo_hip_set_layout(ctx_, &lay, &cfg.cam), "set_layout");
  for (int b = 0; b < B; ++b) Check(xivo_hip_upload_P(ctx_, b, 1, P0, (long)N * N, N), "upload_P");
  std::vector<xivo_group_in> groups((size_t)B * cfg.n_groups);
  for (auto& g : groups) { std::memset(&g, 0, sizeof(g)); g.Rsb[0] = g.Rsb[4] = g.Rsb[8] = 1.0; }
  std::vector<xivo_feat_in> feats((size_t)B * F);
  for (auto& f : feats) { std::memset(&f, 0, sizeof(f)); f.sind = -1; }
  Check(xivo_hip_set_scene(ctx_, 0, B, F, poses0, groups.data(), feats.data()), "set_scene");
  books_.resize(B);
  for (auto& bk : books_) {
    bk.group_refs.assign(cfg.n_groups, -1);
    bk.feat_id.assign(F, -1);
    bk.feat_ref.assign(F, -1);
  }
  last_gyro_.assign((size_t)B * 3, 0.0); last_accel_ = last_gyro_; slope_gyro_ = last_gyro_; slope_accel_ = last_gyro_;
  mask_.assign((size_t)B * F, 0);
  xp_.assign((size_t)B * F * 2, 0.0);
  slot_track_all_.assign((size_t)B * F, -1);
}

BatchEstimator::~BatchEstimator() {
  if (ctx_) xivo_hip_destroy(ctx_);
}

// Estimator::Propagate, visual_meas == false (src/estimator.cpp:558-567)
void BatchEstimator::InertialMeas(double t, const double* gyro, const double* accel) {
  const double t0 = now_s();
  if (!have_imu_) {
    std::copy(gyro, gyro + (size_t)B_ * 3, last_gyro_.begin());
    std::copy(accel, accel + (size_t)B_ * 3, last_accel_.begin());
    have_imu_ = true; t_ = t;
    host_s_ += now_s() - t0;
    return;
  }
  const double dt = t - t_;
  if (!(dt > 0.0)) {
    // Estimator::Propagate returns on dt == 0 with last_ / slope_ untouched (src/estimator.cpp:550-555); a message from
    // the past is skipped the same way
    if (dt < 0.0) fprintf(stderr, "xivo::hip::BatchEstimator: IMU message older than the filter time skipped\n");
    host_s_ += now_s() - t0;
    return;
  }
  std::vector<xivo_imu_in> rec(B_);
  for (int b = 0; b < B_; ++b) {
    xivo_imu_in& r = rec[b];
    for (int i = 0; i < 3; ++i) {
      const size_t k = (size_t)b * 3 + i;
      slope_gyro_[k] = (gyro[k] - last_gyro_[k]) / dt;
      slope_accel_[k] = (accel[k] - last_accel_[k]) / dt;
      r.gyro[i] = last_gyro_[k]; r.accel[i] = last_accel_[k];
      r.slope_gyro[i] = slope_gyro_[k]; r.slope_accel[i] = slope_accel_[k];
      last_gyro_[k] = gyro[k]; last_accel_[k] = accel[k];
    }
    r.dt = dt;
  }
  pending_.push_back(std::move(rec));
  t_ = t;
  host_s_ += now_s() - t0;
}

void BatchEstimator::DropFeature(Book& bk, int j) {
  bk.id2slot.erase(bk.feat_id[j]);
  bk.group_refs[bk.feat_ref[j]] -= 1;
  bk.feat_id[j] = -1; bk.feat_ref[j] = -1;
}

// Estimator::DiscardAffectedGroups, simplified: a group leaves the state with its last feature
void BatchEstimator::DiscardEmptyGroups(int b, std::vector<xivo_edit_op>& ops) {
  Book& bk = books_[b];
  for (int g = 0; g < cfg_.n_groups; ++g)
    if (bk.group_refs[g] == 0) {
      ops.push_back(make_op(b, XIVO_EDIT_REMOVE_GROUP, g));
      bk.group_refs[g] = -1;
    }
}

void BatchEstimator::VisualMeasPointCloud(double t, const int* off, const int64_t* ids, const double* meas,
                                          unsigned char* mask_out) {
  double t0 = now_s();
  const int F = cfg_.n_features;
  // Estimator::Propagate, visual_meas == true (src/estimator.cpp:568-575): extrapolate along the last slope; dt == 0
  // (IMU and camera stamps coincide, the simulation case) propagates nothing (:550-555)
  if (have_imu_ && t != t_) {
    const double dt = t - t_;
    std::vector<xivo_imu_in> rec(B_);
    for (int b = 0; b < B_; ++b) {
      xivo_imu_in& r = rec[b];
      for (int i = 0; i < 3; ++i) {
        const size_t k = (size_t)b * 3 + i;
        r.gyro[i] = last_gyro_[k]; r.accel[i] = last_accel_[k];
        r.slope_gyro[i] = slope_gyro_[k]; r.slope_accel[i] = slope_accel_[k];
        last_gyro_[k] = last_gyro_[k] + slope_gyro_[k] * dt;
        last_accel_[k] = last_accel_[k] + slope_accel_[k] * dt;
      }
      r.dt = dt;
    }
    pending_.push_back(std::move(rec));
    t_ = t;
  }
  if (!pending_.empty()) {
    const int K = (int)pending_.size();
    std::vector<xivo_imu_in> imu((size_t)B_ * K);
    for (int k = 0; k < K; ++k)
      for (int b = 0; b < B_; ++b) imu[(size_t)b * K + k] = pending_[k][b];
    pending_.clear();
    host_s_ += now_s() - t0;
    Check(xivo_hip_propagate(ctx_, 0, B_, K, imu.data(), &cfg_.prop), "propagate");
    t0 = now_s();
  }
  // --- before the update: tracker-dropped features leave (ProcessTracks, src/manager.cpp:152-169), tracked ones get
  // their new pixel
  const double nan = std::numeric_limits<double>::quiet_NaN();
  std::fill(xp_.begin(), xp_.end(), nan);
  // which track carries each in-state feature: the <= kMaxFeature in-state ids sorted once per filter, one binary
  // search per track (a frame brings ~10^2 tracks per filter, most of them not in the state)
  in_state_.assign((size_t)off[B_], 0);
  per_.resize(B_);
  // (filters are independent: the per-filter book-keeping runs over the host cores, every filter filling its own op
  // list; the lists are then joined in filter order, which xivo_hip_edit_batch requires anyway)
#pragma omp parallel for schedule(static) num_threads(kThreads) if (B_ >= 512)
  for (int b = 0; b < B_; ++b) {
    std::vector<xivo_edit_op>& ops = per_[b];
    ops.clear();
    std::vector<std::pair<int64_t, int>> instate;      // (id, slot), ascending id
    std::vector<int> slot_track(F);
    Book& bk = books_[b];
    for (int j = 0; j < F; ++j) if (bk.feat_id[j] >= 0) instate.emplace_back(bk.feat_id[j], j);
    std::sort(instate.begin(), instate.end());
    std::fill(slot_track.begin(), slot_track.end(), -1);
    for (int k = off[b]; k < off[b + 1]; ++k) {
      auto it = std::lower_bound(instate.begin(), instate.end(), std::make_pair(ids[k], -1));
      if (it != instate.end() && it->first == ids[k]) { slot_track[it->second] = k; in_state_[k] = 1; }
    }
    for (int j = 0; j < F; ++j) {
      if (bk.feat_id[j] < 0) continue;
      const int k = slot_track[j];
      if (k >= 0) {
        xp_[((size_t)b * F + j) * 2] = meas[(size_t)k * 3];
        xp_[((size_t)b * F + j) * 2 + 1] = meas[(size_t)k * 3 + 1];
      } else {
        ops.push_back(make_op(b, XIVO_EDIT_REMOVE_FEATURE, j));
        DropFeature(bk, j);
      }
    }
    DiscardEmptyGroups(b, ops);
    for (int j = 0; j < F; ++j) slot_track_all_[(size_t)b * F + j] = slot_track[j];
  }
  std::vector<xivo_edit_op> ops;
  for (int b = 0; b < B_; ++b) ops.insert(ops.end(), per_[b].begin(), per_[b].end());
  host_s_ += now_s() - t0;
  Check(xivo_hip_edit_batch(ctx_, F, (int)ops.size(), ops.empty() ? nullptr : ops.data()), "edit_batch");
  Check(xivo_hip_set_pixels(ctx_, 0, B_, F, xp_.data()), "set_pixels");
  // --- measurement update on the tracked in-state features (src/manager.cpp:72-104), ragged over the filters
  const double R = cfg_.visual_meas_std * cfg_.visual_meas_std;
  if (cfg_.use_1pt_RANSAC) {
    // Estimator::OutlierRejection with use_1pt_RANSAC (src/manager.cpp:629-650): MH gating, OnePointRANSAC on its inliers,
    // the update on what it keeps. (No gauge group / previous-frame group list in this simplified life cycle.)
    Check(xivo_hip_jacobians_instate(ctx_, B_), "jacobians_instate");
    Check(xivo_hip_mh_gate(ctx_, B_, R, cfg_.MH_thresh, cfg_.MH_adjust_factor, cfg_.use_MH_gating ? cfg_.min_inliers : (1 << 30), nullptr, nullptr), "mh_gate");
    Check(xivo_hip_one_point_ransac(ctx_, B_, R, cfg_.ransac_thresh, cfg_.ransac_Chi2, nullptr, nullptr, nullptr, nullptr, nullptr), "one_point_ransac");
    Check(xivo_hip_stack(ctx_, B_, R), "stack");
    Check(xivo_hip_update_joseph(ctx_, B_), "update_joseph");
  } else {
    Check(xivo_hip_filter_update(ctx_, B_, R, cfg_.MH_thresh, cfg_.MH_adjust_factor, cfg_.min_inliers, cfg_.use_MH_gating), "filter_update");
  }
  Check(xivo_hip_get_gate(ctx_, B_, F, mask_.data(), nullptr), "get_gate");
  // a filter whose S was not positive definite keeps its prior P and absorbs nothing (the device skips both); it is
  // counted and reported here - the reference's pivoted LDL^T cannot fail, so there is no reference behaviour to mirror
  status_.resize(B_);
  const int st = xivo_hip_get_status(ctx_, 0, B_, status_.data());
  if (st == XIVO_HIP_ERR_NOT_SPD) { for (int b = 0; b < B_; ++b) n_not_spd_ += status_[b] != 0; }
  else Check(st, "get_status");
  Check(xivo_hip_absorb_error(ctx_, B_), "absorb_error");
  t0 = now_s();
  for (int b = 0; b < B_; ++b) n_updates_ += books_[b].id2slot.empty() ? 0 : 1;
  // --- after the update: MH-rejected features leave (src/update.cpp:105-113), new ones enter with a new group
  const double fx = cfg_.cam.fx, fy = cfg_.cam.fy, cx = cfg_.cam.cx, cy = cfg_.cam.cy;
  // Camera::GetFocalLength() = 0.5 sqrt(fx^2 + fy^2) (src/camera_manager.cpp:56, src/estimator.cpp:351-352)
  const double fl = 0.5 * std::sqrt(fx * fx + fy * fy);
  const double sd[3] = {cfg_.initial_std_x / fl, cfg_.initial_std_y / fl, cfg_.initial_std_z};
  long rejected = 0;
#pragma omp parallel for schedule(static) reduction(+ : rejected) num_threads(kThreads) if (B_ >= 512)
  for (int b = 0; b < B_; ++b) {
    std::vector<xivo_edit_op>& ops = per_[b];
    ops.clear();
    std::vector<int> free_slots, order;
    Book& bk = books_[b];
    for (int j = 0; j < F; ++j)
      if (bk.feat_id[j] >= 0 && !mask_[(size_t)b * F + j]) {
        ops.push_back(make_op(b, XIVO_EDIT_REMOVE_FEATURE, j));
        in_state_[slot_track_all_[(size_t)b * F + j]] = 0;      // its track is a candidate again right away
        DropFeature(bk, j);
        ++rejected;
      }
    DiscardEmptyGroups(b, ops);
    free_slots.clear();
    for (int j = 0; j < F; ++j) if (bk.feat_id[j] < 0) free_slots.push_back(j);
    int g = -1;
    for (int q = 0; q < cfg_.n_groups; ++q) if (bk.group_refs[q] < 0) { g = q; break; }
    if (g < 0 || ((int)free_slots.size() < cfg_.min_new_features && !bk.id2slot.empty())) continue;
    // candidates: tracks not in the state, inside the depth range, by ascending id
    order.clear();
    for (int k = off[b]; k < off[b + 1]; ++k)
      if (!in_state_[k] && cfg_.min_depth < meas[(size_t)k * 3 + 2] && meas[(size_t)k * 3 + 2] < cfg_.max_depth)
        order.push_back(k);
    std::stable_sort(order.begin(), order.end(), [&](int a_, int b_) { return ids[a_] < ids[b_]; });
    if (order.empty()) continue;
    ops.push_back(make_op(b, XIVO_EDIT_ADD_GROUP, g));          // Estimator::AddGroupToState (src/estimator.cpp:786-819)
    bk.group_refs[g] = 0;
    const size_t n_new = std::min(free_slots.size(), order.size());
    for (size_t q = 0; q < n_new; ++q) {
      const int j = free_slots[q], k = order[q];
      const double u = meas[(size_t)k * 3], v = meas[(size_t)k * 3 + 1], z = meas[(size_t)k * 3 + 2];
      xivo_edit_op o = make_op(b, XIVO_EDIT_ADD_FEATURE, j, j, g);   // AddFeatureToState + FillCovarianceBlock
      o.v[0] = (u - cx) / fx; o.v[1] = (v - cy) / fy; o.v[2] = cfg_.use_invdepth ? 1.0 / z : std::log(z);   // Feature::Initialize (src/feature.cpp:144-150)
      o.v[3] = u; o.v[4] = v;
      o.v[5] = sd[0] * sd[0]; o.v[9] = sd[1] * sd[1]; o.v[13] = sd[2] * sd[2];   // P_ = diag(std)^2 (:158-159)
      ops.push_back(o);
      bk.feat_id[j] = ids[k]; bk.feat_ref[j] = g; bk.id2slot[ids[k]] = j;
      bk.group_refs[g] += 1;
    }
  }
  n_rejected_ += rejected;
  ops.clear();
  for (int b = 0; b < B_; ++b) ops.insert(ops.end(), per_[b].begin(), per_[b].end());
  host_s_ += now_s() - t0;
  Check(xivo_hip_edit_batch(ctx_, F, (int)ops.size(), ops.empty() ? nullptr : ops.data()), "edit_batch");
  if (mask_out) std::memcpy(mask_out, mask_.data(), mask_.size());
}

void BatchEstimator::Poses(xivo_pose_in* out) {
  Check(xivo_hip_get_scene(ctx_, 0, B_, out, nullptr, nullptr), "get_scene");
}

}  // namespace hip
}  // namespace xivo

// ---- C entry points for the Python tests / scripts (ctypes) ------------------------------------------------------
extern "C" {

struct xivo_batch_cfg {   // flat mirror of xivo::hip::BatchConfig
  int n_groups, n_features;
  xivo_cam cam;
  double visual_meas_std, MH_thresh, MH_adjust_factor;
  int min_inliers, min_new_features, fix_group_block, disable_MH_gating;   // cfg use_MH_gating = false
  double initial_std_x, initial_std_y, initial_std_z, min_depth, max_depth;
  xivo_prop_opts prop;
  int use_1pt_RANSAC, use_invdepth;                  // cfg use_1pt_RANSAC, 1pt_RANSAC_thresh, 1pt_RANSAC_Chi2 (src/estimator.cpp:130-134)
  double ransac_thresh, ransac_Chi2;
};

int xivo_batch_create(const xivo_batch_cfg* c, int B, int device, const xivo_pose_in* poses0, const double* P0, void** out) {
  try {
    xivo::hip::BatchConfig cfg;
    cfg.n_groups = c->n_groups; cfg.n_features = c->n_features; cfg.cam = c->cam;
    cfg.visual_meas_std = c->visual_meas_std; cfg.MH_thresh = c->MH_thresh; cfg.MH_adjust_factor = c->MH_adjust_factor;
    cfg.min_inliers = c->min_inliers; cfg.min_new_features = c->min_new_features; cfg.fix_group_block = c->fix_group_block;
    cfg.use_MH_gating = c->disable_MH_gating ? 0 : 1;
    cfg.use_1pt_RANSAC = c->use_1pt_RANSAC; cfg.ransac_thresh = c->ransac_thresh; cfg.ransac_Chi2 = c->ransac_Chi2;
    cfg.use_invdepth = c->use_invdepth;
    cfg.initial_std_x = c->initial_std_x; cfg.initial_std_y = c->initial_std_y; cfg.initial_std_z = c->initial_std_z;
    cfg.min_depth = c->min_depth; cfg.max_depth = c->max_depth; cfg.prop = c->prop;
    *out = new xivo::hip::BatchEstimator(cfg, B, device, poses0, P0);
    return 0;
  } catch (const std::exception&) { return -1; }
}
void xivo_batch_destroy(void* h) { delete static_cast<xivo::hip::BatchEstimator*>(h); }
int xivo_batch_imu(void* h, double t, const double* gyro, const double* accel) {
  try { static_cast<xivo::hip::BatchEstimator*>(h)->InertialMeas(t, gyro, accel); return 0; } catch (const std::exception&) { return -1; }
}
int xivo_batch_visual(void* h, double t, const int* off, const long long* ids, const double* meas, unsigned char* mask_out) {
  try {
    static_cast<xivo::hip::BatchEstimator*>(h)->VisualMeasPointCloud(t, off, reinterpret_cast<const int64_t*>(ids), meas, mask_out);
    return 0;
  } catch (const std::exception&) { return -1; }
}
int xivo_batch_poses(void* h, xivo_pose_in* out) {
  try { static_cast<xivo::hip::BatchEstimator*>(h)->Poses(out); return 0; } catch (const std::exception&) { return -1; }
}
int xivo_batch_book(void* h, int b, long long* feat_id, int* feat_ref, int* group_refs) {
  auto* e = static_cast<xivo::hip::BatchEstimator*>(h);
  if (b < 0 || b >= e->B()) return -1;
  const auto& bk = e->book(b);
  for (size_t j = 0; j < bk.feat_id.size(); ++j) { feat_id[j] = bk.feat_id[j]; feat_ref[j] = bk.feat_ref[j]; }
  for (size_t g = 0; g < bk.group_refs.size(); ++g) group_refs[g] = bk.group_refs[g];
  return 0;
}
void xivo_batch_stats(void* h, long* n_updates, long* n_rejected, double* host_seconds) {
  auto* e = static_cast<xivo::hip::BatchEstimator*>(h);
  *n_updates = e->n_updates(); *n_rejected = e->n_rejected(); *host_seconds = e->host_seconds();
}
int xivo_batch_cfg_size(void) { return (int)sizeof(xivo_batch_cfg); }   // checked against the ctypes mirror (tests)
long xivo_batch_not_spd(void* h) { return static_cast<xivo::hip::BatchEstimator*>(h)->n_not_spd(); }
void* xivo_batch_ctx(void* h) { return static_cast<xivo::hip::BatchEstimator*>(h)->ctx(); }

}  // extern "C"
