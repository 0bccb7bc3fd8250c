// Row-pair compressed measurement Jacobian ("ELL" form of H) and the kernels that use it.
//
// The stacked H of Estimator::FilterUpdate (src/update.cpp:129-138) has two rows per feature, and
// Feature::FillJacobianBlock (src/feature.cpp:658-684) writes only seven 2x3 blocks into each row
// pair: 21 of the N columns (18 with the overwrite of :675-676). The reference multiplies H as a
// dense Eigen matrix (src/estimator.cpp:1259-1280); every term it adds for a structurally zero entry
// is an exact 0 * x = 0, so skipping those terms changes nothing but the summation order.
//
// Layout per filter, per row pair p (rows 2p, 2p+1), ELL_W = ELL_CW + ELL_PW slots:
//   slots [0, ELL_CW)      "common" columns: slot t names the SAME state column in every pair of the
//                          filter (the sensor pose / extrinsics blocks every feature touches); a value
//                          may be 0; unused slots have idx 0 and value 0
//   slots [ELL_CW, ELL_W)  "private" columns of the pair in ascending order (group anchor + feature
//                          blocks); unused slots have idx 0 and value 0
//   val[slot] = (H[2p][idx], H[2p+1][idx])
// A filter whose H does not fit (some pair has more than ELL_PW private columns: dense H, OOS rows)
// gets over[filter] = 1 and is updated by the dense kernels instead.
#pragma once
#include "common.h"

namespace xivo_hip {

constexpr int ELL_CW = 16, ELL_PW = 12, ELL_W = ELL_CW + ELL_PW;
constexpr int ELL_PIW = 12;   // 16-bit private slot indices per pair in the LDS copy of the slab kernels (24 bytes: 8-byte aligned rows)

struct EllBuffers {
  int* idx;        // [batch][pairs_max][ELL_W]
  double* val;     // [batch][pairs_max][ELL_W][2]
  int* nc;         // [batch] number of common slots in use
  int* pw;         // [batch] largest number of private slots any pair uses
  int* over;       // [batch] 1: does not fit
  int pairs_max;   // Mpmax / 2
  __host__ __device__ long stride_idx() const { return (long)pairs_max * ELL_W; }
  __host__ __device__ long stride_val() const { return (long)pairs_max * ELL_W * 2; }
};

// Hand-over of a dense H [M x N, column-major, leading dimension ldh] resident in device memory: builds the
// row-pair compressed form (+ nc / pw / over per filter) in one pass over the matrix, one launch for the batch,
// and copies inn / diagR into the padded working vectors (rows [M, Mp_clear) neutral).
int launch_meas_compress(const double* H, long strideH, int ldh, const double* inn, long strideInn,
                         const double* diagR, long strideR, int M, int N, int Np, int Mp_clear, EllBuffers e,
                         double* inn_out, long strideInnOut, double* R_out, long strideROut, int batch, hipStream_t s,
                         int* host_flags = nullptr /* device alias of pinned host memory, [batch][3] = over, nc, pw: the kernel
                                                      mirrors the three per-filter flags there so that the host reads them
                                                      after ONE stream synchronisation, without device-to-host copies */);
// whether the per-workgroup LDS lists of launch_meas_compress fit the 160 KiB of a CU for these shapes; when they do not
// (N beyond ~2800 at the largest M) the hand-over marks every filter "does not fit" instead (launch_meas_vectors: inn /
// diagR padded, over = 1) and the caller unpacks the dense rows: the dense pipeline has no such limit
bool meas_compress_fits(int Mp_clear, int Np);
int launch_meas_vectors(const double* inn, long strideInn, const double* diagR, long strideR, int M, int Mp_clear, EllBuffers e,
                        double* inn_out, long strideInnOut, double* R_out, long strideROut, int batch, hipStream_t s);
// dense padded H / H^T of the filters that fit the compressed form (over = 0), rebuilt from it
// (Mrows >= 0: only rows [0, Mrows) are rebuilt - mixed stacking keeps dense OOS rows behind them; HT may be null)
int launch_ell_to_dense(EllBuffers e, double* H, long strideH, int ldh, double* HT, long strideHT, int ldht, int Mp,
                        int Np, int batch, hipStream_t s, int Mrows = -1);
// zero rows [row0, row0 + nrows) x columns [c0, c1) of every filter's dense H
int launch_zero_rows(double* H, long strideH, int ldh, int row0, int nrows, int c0, int c1, int batch, hipStream_t s);

// out[x + ldo * m] = sum_slots val[m][slot] * Src[x + lds * idx[m][slot]]  (+ epilogue), x in [0, X)
enum EllMode : int {
  ELL_HP = 0,   // Src = P (symmetric): out = P H^T [Np x Mp], out2 = H P [Mp x Np]      (estimator.cpp:1259)
  ELL_S = 1,    // Src = P H^T [Np x Mp] (tile form) or HP [Mp x Np] (gather form): out = S = (HP) H^T + diag(R); the tile form
                // writes the lower triangle + the 64 x 64 diagonal squares only (nothing reads the rest of the symmetric S)
                //                                                                 (estimator.cpp:1259-1263)
  ELL_G = 2,    // Src = T  [Np x Np] : out = T H^T + K diag(R)  [Np x Mp]               (re-associated :1280-1287)
  ELL_GF = 3,   // ELL_G with the result stored as float (slab form only; strideOut / ldo in float elements)
};
// Estimator::MHGating numeric core (src/update.cpp:60-96) on the ELL rows: S_f = H_f (P H_f^T) + R I2
// from the already formed P H^T, threshold relaxation, then neutralisation of the rejected pairs
// (ELL values, H / H^T / HP / P H^T rows = 0, inn = 0, diagR = 1).
struct GateEllArgs {
  EllBuffers ell;
  double* H; long strideH; int ldh;        // dense copies kept consistent for xivo_hip_get_H
  double* HT; long strideHT; int ldht;
  double* HP; double* PHT;                 // same shapes / strides as H / HT
  double* inn; long strideInn;
  double* diagR; long strideR;
  unsigned char* mask; double* dist;       // [batch x F]
  int F, Np, batch;
  double R, thresh, mult; int min_inliers;
  // from_S: S = H P H^T + diag(R) of ALL candidate rows is already formed (ell<S>); the 2x2 blocks S_f are read
  // off its diagonal and the rows / columns of the rejected pairs are then decoupled in place (0, unit diagonal)
  double* S; long strideS; int lds; int Mp; int from_S;
  // from_S, optional: the 2 x 2 diagonal blocks of S as the S kernel left them, [pair][row in block][column in block]
  // (32 contiguous bytes per feature instead of three cache lines 2 lds doubles apart: the gate's reads of S were 3.9 M
  // random 64-byte fetches per 16384 filters); same values bit for bit. null: read them off S.
  const double* Sdiag; long strideSdiag;
};
struct EllMulArgs {
  EllBuffers ell;
  const double* Src; long strideSrc; int ldsrc;
  int cols;     // number of source columns a slot index can name (Np)
  const double* SrcAlt; long strideSrcAlt; int ldsrcAlt;   // ELL_S: H P [Mp x Np] for the gather fallback (may be null if the tile form fits)
  double* out; long strideOut; int ldo;
  double* out2; long strideOut2; int ldo2;      // ELL_HP only
  double* diag_out; long strideDiag;            // ELL_S, tile form with 64-wide slabs only: compact copy of the 2 x 2 diagonal blocks (or null)
  int* diag_done;                               // host: set to 1 by the launcher when the launch writes diag_out
  const double* diagR; long strideR;            // ELL_S, ELL_G
  const double* K; long strideK; int ldk;       // ELL_G
  int X;        // extent of the contiguous index (Np for HP/G, Mp for S)
  int Mp;       // rows of H in use (multiple of 16)
  int slabs_per_wg; // set by the launcher (slab form): consecutive slabs one workgroup streams
  int persist_stride; // set by the launcher (slab form, prefetching instantiation): > 0 = a workgroup walks filters b, b + stride, ...
  int rb_per_wg; // set by the launcher: 16-row blocks one workgroup walks
  int nc_max;   // upper bound of nc over the filters of the launch (host mirror)
  int pw_max;   // upper bound of pw (0 = unknown -> ELL_PW)
  int batch;
  // ELL_S only: run the gate (src/update.cpp:60-96, gate.from_S semantics) in the tail of the S kernel when one workgroup
  // forms the whole S of its filter (big batches); *gate_done tells the caller whether that happened (else: launch_gate_ell)
  GateEllArgs gate;
  int gate_here;
  int* gate_done;
};
int launch_ell_mul(int mode, const EllMulArgs& a, hipStream_t s);
// name of the kernel instantiation launch_ell_mul runs for these arguments
void ell_kernel_label(int mode, const EllMulArgs& a, char* buf, size_t n);
// true when launch_ell_mul will run the slab-in-LDS form for these arguments
bool ell_uses_slab_form(const EllMulArgs& a);

int launch_gate_ell(const GateEllArgs& a, hipStream_t s);

}  // namespace xivo_hip
