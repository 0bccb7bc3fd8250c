// xivo::hip::BatchEstimator - the per-message surface of the reference's Estimator (InertialMeas /
// VisualMeasPointCloud, src/estimator.h:89-142) for B independent filters that live on one GPU context.
//
// The reference is one singleton filter per process driven message by message; a GPU wants thousands of filters per
// launch. This class keeps the reference's message semantics per filter - IMU bookkeeping of Estimator::Propagate
// (src/estimator.cpp:548-575), IMU before camera at equal stamps, per camera frame the order of Estimator::UpdateStep
// (src/manager.cpp:30-110): tracker-dropped features out, filter update, MH-rejected features out, new features in -
// and turns one camera frame of all filters into five C-ABI calls on the resident state:
//   xivo_hip_propagate, xivo_hip_edit_batch, xivo_hip_set_pixels, xivo_hip_filter_update (+ get_gate, absorb_error),
//   xivo_hip_edit_batch.
// The life cycle is the simplified one of xivo_amd/sequence.py (which it reproduces decision for decision: the tests
// run both on the same input): features enter with the depth that comes with the track (`InitWithSimDepths`,
// src/manager.cpp:588) anchored to a group created from the current pose; no gauge features, no reference-group
// switching, no sub-filter warm-up. The host side holds only the slot book-keeping (gsel_ / fsel_,
// src/estimator.h:496-503); every number of the filter stays on the device.
#pragma once
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "../../include/xivo_hip.h"

namespace xivo {
namespace hip {

struct BatchConfig {
  int n_groups = 15, n_features = 30;              // kMaxGroup, kMaxFeature (src/core.h:95-105)
  xivo_cam cam{};                                   // pinhole for point-cloud input (Feature::Initialize un-projects with it)
  double visual_meas_std = 1.0;
  double MH_thresh = 5.991, MH_adjust_factor = 1.1;
  int min_inliers = 5;
  int use_MH_gating = 1;                            // cfg use_MH_gating (src/estimator.cpp:364)
  int use_1pt_RANSAC = 0;                           // cfg use_1pt_RANSAC: OnePointRANSAC between gating and the update
  double ransac_thresh = 5.0, ransac_Chi2 = 5.89;   // 1pt_RANSAC_thresh / 1pt_RANSAC_Chi2 (src/estimator.cpp:132-134)
  double initial_std_x = 1.0, initial_std_y = 1.0, initial_std_z = 0.1;   // pixels, pixels, log-depth (estimator.cpp:349-353)
  double min_depth = 0.05, max_depth = 10.0;
  int min_new_features = 3;                         // open a new group only when this many feature slots are free
  int fix_group_block = 1;                          // XIVO_HIP_FLAG_FIX_GROUP_BLOCK (see xivo_amd/sequence.py)
  int use_invdepth = 0;                             // the reference's USE_INVDEPTH build: features are (X/Z, Y/Z, 1/Z); initial_std_z is then an inverse-depth std
  xivo_prop_opts prop{};                            // Qimu, Qmodel, gravity, integrator, step size
  int N() const { return 23 + 6 * n_groups + 3 * n_features; }
};

class BatchEstimator {
 public:
  // poses0: initial nominal state of every filter; P0: one N x N column-major covariance shared by all
  BatchEstimator(const BatchConfig& cfg, int B, int device, const xivo_pose_in* poses0, const double* P0);
  ~BatchEstimator();
  BatchEstimator(const BatchEstimator&) = delete;
  BatchEstimator& operator=(const BatchEstimator&) = delete;

  // Estimator::InertialMeas for all filters at time t [s]: gyro, accel are [B][3]. The first call only initialises
  // last_gyro_ / last_accel_ (nothing to integrate from yet).
  void InertialMeas(double t, const double* gyro, const double* accel);
  // Estimator::VisualMeasPointCloud for all filters at time t: filter b's tracks are ids[off[b] .. off[b + 1]) with rows
  // (x, y, depth) in xp_and_depths. Runs the whole frame on the device. mask_out (may be null): [B][n_features] inliers.
  void VisualMeasPointCloud(double t, const int* off, const int64_t* ids, const double* xp_and_depths,
                            unsigned char* mask_out);

  void Poses(xivo_pose_in* out);                    // gsb / Vsb / bg / ba ... of every filter (reads the resident state)
  int B() const { return B_; }
  const BatchConfig& cfg() const { return cfg_; }
  xivo_hip_ctx* ctx() { return ctx_; }
  long n_updates() const { return n_updates_; }
  long n_not_spd() const { return n_not_spd_; }   // updates skipped because S was not positive definite
  long n_rejected() const { return n_rejected_; }
  double host_seconds() const { return host_s_; }   // time spent in the host-side life cycle (not in C-ABI calls)

  struct Book {                                     // one filter's slots
    std::vector<int> group_refs;                    // -1 free, else number of in-state features anchored there
    std::vector<int64_t> feat_id;                   // -1 free
    std::vector<int> feat_ref;
    std::unordered_map<int64_t, int> id2slot;
  };
  const Book& book(int b) const { return books_[b]; }

 private:
  void Check(int rc, const char* what);
  void DropFeature(Book& bk, int j);
  void DiscardEmptyGroups(int b, std::vector<xivo_edit_op>& ops);

  BatchConfig cfg_;
  int B_;
  xivo_hip_ctx* ctx_ = nullptr;
  std::vector<Book> books_;
  // Estimator::Propagate's bookkeeping per filter
  bool have_imu_ = false;
  double t_ = 0.0;
  std::vector<double> last_gyro_, last_accel_, slope_gyro_, slope_accel_;   // [B][3]
  std::vector<std::vector<xivo_imu_in>> pending_;   // [message][B]
  long n_updates_ = 0, n_rejected_ = 0, n_not_spd_ = 0;
  std::vector<int> status_;
  double host_s_ = 0.0;
  std::vector<unsigned char> mask_;
  std::vector<double> xp_;
  std::vector<unsigned char> in_state_;   // per track of the current frame: is it an in-state feature
  std::vector<std::vector<xivo_edit_op>> per_;   // per filter: the edit ops of the current phase
  std::vector<int> slot_track_all_;       // [B][F] track index of each in-state feature in the current frame
};

}  // namespace hip
}  // namespace xivo
