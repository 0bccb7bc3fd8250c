// One-kernel measurement update for the shapes a CU holds (fused_update.hip).
#pragma once
#include "ell.h"

namespace xivo_hip {

struct FusedArgs {
  double* P; long strideP; int ldp;          // covariance, updated in place (symmetric by contract)
  EllBuffers ell;                            // row-pair compressed H (nc <= 12 common, pw <= 9 private slots in use)
  double* inn; long strideInn;               // [Mp] padded with 0; rejected pairs are neutralised in place
  double* diagR; long strideR;               // [Mp] padded with 1
  double* err; long strideErr;               // out: dx = K inn
  double* PHT; long stridePHT; int ldpht;    // written ONLY for a filter whose S the Cholesky cannot factor (the fallback's input)
  int* status;                               // out: 0, or 1 + the first non-positive pivot
  int Np, Mp, batch;
  // MH gating (src/update.cpp:60-96) in front of the factorisation; gate = 0: none
  int gate, F; double R, thresh, mult; int min_inliers;
  unsigned char* mask; double* dist;         // [batch x F]
  double* H; long strideH; int ldh;          // dense copies of the stacked rows kept consistent with the gate (or null)
  double* HT; long strideHT; int ldht;
  int pw;                                    // widest private slot count of the batch (host copy of ell.pw): <= 6 takes the six-slot instantiation
  int jbp;                                   // (set by the launcher) column blocks per LDS phase of the product
  int tsc_off;                               // (set by the launcher) LDS offset (doubles) of the per-wave transpose scratch of the product
};
bool fused_update_supported(int Mp, int Np);
int launch_fused_update(const FusedArgs& g, hipStream_t stream);
void fused_update_label(int Mp, int Np, int pw, char* buf, size_t n);
int fused_tiles_selftest(int nwl, int* per_simd);

}  // namespace xivo_hip
