// Device helpers shared by the gating kernels (ekf_kernels.hip, ell_kernels.hip):
// the numeric core of Estimator::MHGating, src/update.cpp:60-96.
#pragma once
#include <hip/hip_runtime.h>

namespace xivo_hip {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// d = res^T (S.llt().solve(res)) for 2x2 S given by its lower triangle
// (src/update.cpp:65-69; Eigen LLT reads the lower triangle)
__device__ __forceinline__ double mh_dist_2x2(double s00, double s10, double s11, double r0, double r1) {
  const double l00 = sqrt(s00);
  const double l10 = s10 / l00;
  const double l11 = sqrt(s11 - l10 * l10);
  const double y0 = r0 / l00;
  const double y1 = (r1 - l10 * y0) / l11;
  const double x1 = y1 / l11;
  const double x0 = (y0 - l10 * x1) / l00;
  return r0 * x0 + r1 * x1;
}

// threshold relaxation loop of src/update.cpp:73-96, run by one wave over the
// F distances in LDS. Returns the threshold that was in force when the loop
// exited (inlier <=> dist < thresh).
// `present` = number of real entries among the F (absent ones carry +inf and can never become inliers).
__device__ inline double relax_threshold(const double* sdist, int F, double thresh, double mult, int min_inliers,
                                  int lane, int present = -1) {
  if (present < 0) present = F;
  if (min_inliers <= 0) return -1.0;  // loop body never runs: no inliers (update.cpp:73)
  for (int it = 0; it < 4096; ++it) {
    int cnt = 0;
    for (int f0 = 0; f0 < F; f0 += 64) {
      const int f = f0 + lane;
      const bool in = (f < F) && (sdist[f] < thresh);
      cnt += __popcll(__ballot(in));
    }
    if (cnt >= min_inliers || cnt >= present) return thresh;
    thresh *= mult;
  }
  return thresh;
}


}  // namespace xivo_hip
