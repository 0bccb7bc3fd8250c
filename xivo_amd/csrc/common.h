// Shared declarations for the MI355X (gfx950) EKF measurement-update kernels.
//
// All device matrices are column-major fp64 (the reference's Eigen layout,
// /root/reference/common/alias.h:11, CMakeLists.txt:42) with every dimension
// padded to a multiple of 16 (the v_mfma_f64_16x16x4_f64 tile) and zero filled
// in the pad, so no kernel needs element-level bounds checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace xivo_hip {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

static inline int round_up16(int x) { return (x + 15) & ~15; }

// ---------------------------------------------------------------------------
// Batched "NT" GEMM:  C[i + j*ldc] (+)= sum_k A[i + k*lda] * B[j + k*ldb]
// i.e. C = A * B^T with both operands stored with the OUTPUT index contiguous.
// Up to two K-segments are summed into one output tile (used for
// P+ = T*A^T + K*diag(R)*K^T, estimator.cpp:1280-1287, in a single pass).
// ---------------------------------------------------------------------------
struct GemmSeg {
  const double* A;      // [rows_out x K], leading dim lda
  const double* B;      // [cols_out x K], leading dim ldb
  const double* scale;  // optional per-k scale applied to B (nullptr = none)
  long strideA, strideB, strideScale;  // per-filter strides (elements)
  int lda, ldb, K;      // K multiple of 16
  int a_f32;            // 1: A is stored as float (same lda / stride in ELEMENTS) - fp32 products only
  int b_f32;            // 1: B likewise
};

enum GemmEpilogue : int {
  EPI_NONE = 0,
  EPI_ADD_DIAG = 1,   // C[i][i] += diag[i]      (S = HPH^T + R, estimator.cpp:1261-1263)
  EPI_SUB_IDENT = 2,  // C[i][i] -= 1            (KH - I,       estimator.cpp:1276-1279)
  EPI_SUB_MAT = 3,    // C = acc - Msub          (T = K(HP) - P,  estimator.cpp:1280)
  EPI_ADD_MAT = 4,    // C = acc + Msub          (Phi P Phi^T + Q)
  EPI_RSUB_MAT = 5,   // C = Msub - acc          (P+ = P - W^T W, the symmetric form)
};

struct GemmArgs {
  GemmSeg seg[2];
  int nseg;
  double* C;
  long strideC;
  int ldc;
  int Mp, Np;          // output rows / cols (multiples of 16)
  double* C2;          // optional second output = C^T (ldc2), e.g. PH^T next to HP
  long strideC2;
  int ldc2;
  int c2_rows;         // > 0: C2 receives only the transposed copy of rows [0, c2_rows) of C (the columns of C2 its reader needs)
  const double* diag;  // EPI_ADD_DIAG
  long strideDiag;
  const double* Msub;  // EPI_SUB_MAT / EPI_ADD_MAT operand, same shape as C
  long strideMsub;
  int ldmsub;
  const double* McolScale;  // optional per-column scale of Msub (K diag(R))
  long strideMcol;
  int epilogue;
  int lower_only;      // 1: skip tiles strictly above the diagonal and mirror-store
  int no_mirror;       // with lower_only: store the lower triangle only (the reader knows the matrix is symmetric)
  int batch;
  int small_tiles;     // lower_only: 64 x 64 tiles (few filters: more workgroups per output)
  int tiles_m, tiles_n;
  int fp32;            // 1: fp32 MFMA with fp32 accumulation for this product (operands/results stay fp64 in HBM)
  const int* skip_status;  // optional [batch]: filters with a non-zero entry are left untouched (S not positive definite:
                           // the final covariance product must not overwrite P with the garbage of a failed factorisation)
};

// launches on `stream`; returns hipError_t as int
int launch_gemm_nt_f64(const GemmArgs& args, hipStream_t stream);
// name of the kernel instantiation launch_gemm_nt_f64 runs for these arguments (as rocprofv3 prints it, no spaces)
void gemm_kernel_label(const GemmArgs& args, char* buf, size_t n);
// tile actually chosen for (Mp, Np) - exposed for tests / DESIGN.md
void gemm_pick_tile(int Mp, int Np, int lower_only, int* WM, int* WN);

// ---------------------------------------------------------------------------
// Batched Cholesky (one workgroup per filter) + register-resident TRSM.
// ---------------------------------------------------------------------------
struct CholArgs {
  double* S;        // [Mp x Mp] in: S (lower triangle read); out: L in lower, L^T in upper
  long strideS;
  int lds;
  int Mp;           // multiple of 16
  double* invD;     // [Mp/16][2][256]: inv(L_jj) col-major, then its transpose
  long strideInvD;
  int* status;      // per filter: 0 ok, else 1 + first non-positive pivot index
  int batch;
  int latency;      // the solve behind it takes the latency route (streamed kernel: reads the mirrored upper triangle)
};
// MH gating folded into the prologue of the factorisation (chol_f64.hip, GATE instantiations): what gate_ell_kernel takes,
// minus S (the factorisation applies the decoupling of the rejected pairs where it loads S)
struct CholGateArgs {
  const double* Sdiag; long strideSdiag;    // compact 2 x 2 diagonal blocks of S as ell<S> leaves them, [pair][row][col]
  double* inn; long strideInn;
  double* diagR; long strideR;
  double* ellval; long strideVal; int ell_w;   // compressed values of the row pairs (ell.h: [pairs][ELL_W][2])
  double* PHT; long stridePHT; int ldpht; int Np;
  unsigned char* mask; double* dist;        // [batch x F]
  int F;
  double R, thresh, mult; int min_inliers;
};
bool chol_gate_supported(int Mp, int batch);
int launch_chol_f64(const CholArgs& args, hipStream_t stream, const CholGateArgs* gate = nullptr);
void chol_kernel_label(int Mp, int batch, char* buf, size_t n);

struct TrsmArgs {
  const double* LU;    // from chol: L lower, L^T upper
  long strideLU;
  int ldlu;
  const double* invD;
  long strideInvD;
  const double* PHT;   // RHS transposed: (HP)^T = P H^T, [Np x Mp] col-major
  long stridePHT;
  int ldpht;
  double* K;           // out: gain [Np x Mp] col-major (estimator.cpp:1265-1266)
  long strideK;
  int ldk;
  const double* inn;   // [Mp]
  long strideInn;
  double* err;         // out: dx = K * inn [Np]  (estimator.cpp:1267)
  long strideErr;
  int Mp, Np;
  int batch;
  int latency;         // with Yout: the latency route (trsm_latency_route) - streamed kernel whatever the factor's size
  int stream8;         // with Yout, at most eight block rows: the streamed kernel on eight-wave workgroups of 128 columns (states
                       // wider than one sixteen-wave workgroup: no nearly empty second workgroup that copies the factor for two waves)
  int fwd_only;        // 1: stop after the forward substitution: K receives W^T = (L^-1 HP)^T and dx = W^T y with
  const double* y;     //    y = L^-1 inn [Mp] (launch_fwd_vec) - the symmetric form P+ = P - W^T W needs no more
  long strideY;
  // T = K (HP) - P formed on the gain while it is still in registers (estimator.cpp:1280 distributed over HP):
  // lower triangle authoritative + mirror, as the stand-alone product writes it. nullptr: not wanted.
  double* T;           // (fwd_only: the covariance itself, updated in place: P+ = P - W^T W)
  long strideT;
  int ldt;
  const int* skip_status;   // fwd_only / joseph: per filter, non-zero = leave P untouched
  int joseph;          // 2: T is the covariance itself and receives the whole Joseph update in place (whitened form, chol_trsm.hip)
  const double* Pm;    // the prior covariance [Np x Np]
  long stridePm;
  int ldpm;
  int t_jbp;           // (set by the launcher) column blocks per LDS phase
  // whitened outputs for a covariance update OUTSIDE the solve kernel (shapes one workgroup does not hold: N > 256 or
  // M > 176): K receives V^T = (W - D)^T instead of the gain, Yout receives Y^T = (W + D)^T, both [Np x Mp]; then
  // P+ = P - V^T Y as a tiled symmetric product (the Joseph expression for the computed gain, chol_trsm.hip TF == 4)
  double* Yout;
  long strideY2;
  int ldy2;
  // out_f32 (with Yout, streamed kernel only): the whitened outputs leave as FLOAT, both into the Yout buffer - Y^T at float
  // element 0, V^T at float element Np * Mp (leading dimension ldy2, in floats) - for a product on the fp32 MFMA
  // (XIVO_HIP_FLAG_FP32_WHITENED); K keeps the fp64 stash of W only
  int out_f32;
};
int launch_trsm_f64(const TrsmArgs& args, hipStream_t stream);
// P+ = G K^T - T with the rows of G in registers (one workgroup per filter; see chol_trsm.hip)
struct PnewRegArgs {
  const double* G;     // [Np x Mp] col-major
  long strideG;
  int ldg;
  const double* K;     // [Np x Mp] col-major
  long strideK;
  int ldk;
  const double* T;     // [Np x Np] symmetric
  long strideT;
  int ldt;
  double* P;           // out [Np x Np]
  long strideP;
  int ldp;
  const int* skip_status;   // per filter: non-zero = leave P untouched
  int Mp, Np, batch;
  int jbp;             // (set by the launcher)
};
bool pnew_reg_supported(int Mp, int Np);
int launch_pnew_reg_f64(const PnewRegArgs& args, hipStream_t stream);
void pnew_reg_kernel_label(int Mp, char* buf, size_t n);
// whether launch_trsm_f64 forms T itself for these shapes (whole factor in LDS, one column chunk per filter)
bool trsm_forms_T(int Mp, int Np);
// the whitened in-solve update on ten- / twelve-wave workgroups with W in registers (seven block rows, narrow state: solve_fused.hip)
bool trsm_narrow_supported(int Mp, int Np);
int launch_trsm_narrow(const TrsmArgs& args, hipStream_t stream);
void trsm_narrow_label(int Mp, int Np, char* buf, size_t n);
bool trsm_latency_route(int Mp, int batch);   // few filters: streamed solve on 128-column workgroups + tiled product
// y = L^-1 inn for every filter (one wave each): the forward substitution of the innovation vector
int launch_fwd_vec(const double* LU, long strideLU, int ldlu, const double* invD, long strideInvD, const double* inn, long strideInn,
                   double* y, long strideY, int Mp, int batch, hipStream_t stream);
void trsm_kernel_label(int Mp, char* buf, size_t n, int forms_T = 0, bool latency = false, bool stream8 = false);   // forms_T: 0 solve only, 1 + T = K(HP) - P, 2 + P - W^T W (symmetric form)

}  // namespace xivo_hip

namespace xivo_hip {
// Row-group partition of the lower triangle for the symmetric GEMM kernel.
struct SymGroups {
  int n;          // number of groups (<= 8)
  int r0[8], r1[8];  // block rows [r0, r1) of each group
};
// symmetric (lower triangle + mirror) product for outputs of at most 16 blocks (256)
int launch_gemm_sym_f64(const GemmArgs& args, hipStream_t stream);
bool gemm_sym_supported(int Mp);
}  // namespace xivo_hip
