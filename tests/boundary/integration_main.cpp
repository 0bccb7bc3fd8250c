// Compiles the reference-side binding printed in INTEGRATION.md section 3 (the body is spliced in VERBATIM by
// tests/test_boundary_cpu.py as integration_body.inc) against the reference's own types: common/alias.h (Eigen 3.3.9 +
// Sophus from /root/reference/thirdparty), members named as in src/estimator.h:423-509. TEST INFRASTRUCTURE ONLY.
#include <cstdio>
#include "alias.h"
#include "glog/logging.h"   // oracle/ref/shim: LOG(FATAL)
#include "xivo_hip.h"
#define USE_HIP_UPDATE
extern "C" int stub_calls(void);
namespace xivo {
class Estimator {
 public:
  void UpdateJosephForm();
  MatX P_, H_, K_, S_, I_KH_;
  VecX inn_, diagR_, err_;
  xivo_hip_ctx* hip_{nullptr};
};
#include "integration_body.inc"
}  // namespace xivo

int main() {
  using namespace xivo;
  const int N = 29, M = 8;
  Estimator e;
  if (xivo_hip_create(&e.hip_, 0, N, M, 1, 0) != XIVO_HIP_OK) return 2;
  e.P_.setIdentity(N, N); e.H_.setRandom(M, N); e.inn_.setOnes(M); e.diagR_.setConstant(M, 2.25); e.err_.setZero(N);
  e.UpdateJosephForm();
  const bool ok = stub_calls() == 64 && e.err_(0) == 1.0 && e.err_(M) == 0.0 && e.P_.isIdentity();
  std::printf("calls=%d ok=%d\n", stub_calls(), (int)ok);
  xivo_hip_destroy(e.hip_);
  return ok ? 0 : 1;
}
