// Device answer to an innovation covariance the un-pivoted Cholesky cannot factor.
//
// The reference solves  K^T = S.ldlt().solve(H P)  with Eigen's diagonally pivoted L D L^T
// (/root/reference/src/estimator.cpp:1266; Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked), which never
// fails: an S that is indefinite or negative definite (a corrupted R, a covariance that lost definiteness) still yields
// a gain and the Joseph update goes through. The batched pipelines here factor S = L L^T without pivoting and report
// such a filter through its status; this kernel then reproduces the reference for exactly those filters (a rare path:
// one workgroup per flagged filter, plain vector arithmetic, nothing tuned):
//   1  S = H (P H^T) + diag(R) rebuilt in full (the failed factorisation left garbage in its buffer)
//   2  pivoted L D L^T in place: at step k the largest remaining |diagonal| moves to position k (row / column swap,
//      transposition recorded), then the up-looking column update of Eigen's unblocked kernel
//   3  K^T = P^T L^-T D^-1 L^-1 P (H P), one thread per state column; D^-1 with Eigen's tolerance (1 / max double)
//   4  dx = K inn
//   5  the Joseph update as coded:  A = I - K H,  P+ = A P A^T + K R K^T  (estimator.cpp:1276-1287) - exact for ANY
//      gain, which is the point of that form when S is not what the filter assumed
// The status of a filter that went through here is cleared and ldlt_used[filt] set, so AbsorbError and the caller treat
// it as the reference would: updated. EXCEPT when the arithmetic is not finite (a NaN / Inf measurement, R or covariance:
// the Cholesky flags a NaN pivot like a negative one): Eigen would hand the NaNs on, here the filter keeps its prior
// covariance, its dx is zeroed, its status stays non-zero and ldlt_used reads 0 - the caller's only failure signal is kept.
#include "ell.h"
#include "ekf_kernels.h"

namespace xivo_hip {

namespace {

struct HRow {   // row-pair compressed or dense access to one filter's H
  const int* idx; const double* val; const double* Hd; int ldh; int N; bool dense; int dense_from;   // rows >= dense_from are dense
  const double* lead; int ldlead, lead_k;   // calibration columns of a compressed row (dense block over the leading lead_k state columns), or null
  // calls f(col, value) for every stored entry of row m (dense: every column, zeros included)
  template <class F>
  __device__ __forceinline__ void for_each(int m, F&& f) const {
    if (dense || m >= dense_from) {
      for (int n = 0; n < N; ++n) f(n, Hd[m + (long)n * ldh]);
    } else {
      const int p = m >> 1, h = m & 1;
      for (int t = 0; t < ELL_W; ++t) {
        const double v = val[((long)p * ELL_W + t) * 2 + h];
        if (v != 0.0) f(idx[(long)p * ELL_W + t], v);
      }
      for (int k = 0; lead && k < lead_k; ++k) {
        const double v = lead[m + (long)k * ldlead];
        if (v != 0.0) f(k, v);
      }
    }
  }
};

__global__ __launch_bounds__(1024) void ldlt_fallback_kernel(LdltFallbackArgs a) {
  const int filt = blockIdx.x;
  if (a.status[filt] == 0) {      // factored: nothing to do - but this launch is also what clears the "used" flag of the call
    if (threadIdx.x == 0) a.used[filt] = 0;   // (a separate memset in front of every update cost a dispatch: 10 us of a B = 1 step)
    return;
  }
  const int tid = threadIdx.x, nt = blockDim.x;
  const int N = a.N, M = a.M;
  double* S = a.S + (long)filt * a.strideS;
  const double* PHT = a.PHT + (long)filt * a.stridePHT;
  double* K = a.K + (long)filt * a.strideK;
  double* Am = a.A + (long)filt * a.strideA;
  double* T = a.T + (long)filt * a.strideT;
  double* P = a.P + (long)filt * a.strideP;
  const double* inn = a.inn + (long)filt * a.strideInn;
  const double* dR = a.diagR + (long)filt * a.strideR;
  const long ld = a.lds;
  HRow H;
  H.dense = a.use_dense || a.ell.over[filt] != 0;
  H.idx = a.ell.idx + (long)filt * a.ell.stride_idx(); H.val = a.ell.val + (long)filt * a.ell.stride_val();
  H.Hd = a.H + (long)filt * a.strideH; H.ldh = a.ldh; H.N = N;
  H.dense_from = a.mixed_row0 >= 0 ? a.mixed_row0 : (1 << 30);
  H.lead = a.lead ? a.lead + (long)filt * a.strideLead : nullptr; H.ldlead = a.ldlead; H.lead_k = a.lead_k;

  __shared__ int perm[512];       // transpositions (M <= 384)
  __shared__ double sred[1024];
  __shared__ int sidx[1024];
  __shared__ double sD[512];
  __shared__ int s_bad;           // a pivot or an entry of dx is not finite
  if (tid == 0) s_bad = 0;
  const double dmax = 1.7976931348623157e308;

  // ---- 1. S = H (P H^T) + diag(R), every entry
  for (long e = tid; e < (long)M * M; e += nt) {
    const int mr = (int)(e % M), mc = (int)(e / M);
    double acc = mr == mc ? dR[mr] : 0.0;
    H.for_each(mr, [&](int n, double v) { acc = fma(v, PHT[n + (long)mc * a.ldpht], acc); });
    S[mr + mc * ld] = acc;
  }
  __syncthreads();
  // symmetrise (H P H^T is symmetric up to rounding; Eigen reads the lower triangle only)
  for (long e = tid; e < (long)M * M; e += nt) {
    const int mr = (int)(e % M), mc = (int)(e / M);
    if (mr < mc) S[mr + mc * ld] = S[mc + mr * ld];
  }
  __syncthreads();

  // ---- 2. pivoted L D L^T (lower triangle authoritative; the upper one is kept as its mirror so that swaps are plain)
  for (int k = 0; k < M; ++k) {
    // largest |diagonal| among k..M-1 (first index wins ties, as Eigen's maxCoeff)
    double best = -1.0; int bi = k;
    for (int i = k + tid; i < M; i += nt) { const double v = fabs(S[i + i * ld]); if (v > best) { best = v; bi = i; } }
    sred[tid] = best; sidx[tid] = bi;
    __syncthreads();
    for (int s = nt >> 1; s > 0; s >>= 1) {
      if (tid < s) {
        const double v2 = sred[tid + s]; const int i2 = sidx[tid + s];
        if (v2 > sred[tid] || (v2 == sred[tid] && i2 < sidx[tid])) { sred[tid] = v2; sidx[tid] = i2; }
      }
      __syncthreads();
    }
    const int p = sidx[0];
    if (tid == 0) perm[k] = p;
    if (p != k) {   // symmetric swap of rows / columns k and p
      for (int j = tid; j < M; j += nt) { const double t0 = S[k + j * ld]; S[k + j * ld] = S[p + j * ld]; S[p + j * ld] = t0; }
      __syncthreads();
      for (int j = tid; j < M; j += nt) { const double t0 = S[j + k * ld]; S[j + k * ld] = S[j + p * ld]; S[j + p * ld] = t0; }
      __syncthreads();
    }
    // temp = D(0..k) .* A10^T ; A(k,k) -= A10 temp ; A21 -= A20 temp ; A21 /= A(k,k)
    for (int i = k + tid; i < M; i += nt) {
      double acc = S[i + k * ld];
      for (int j = 0; j < k; ++j) acc = fma(-S[i + j * ld], sD[j] * S[k + j * ld], acc);
      S[i + k * ld] = acc;
    }
    __syncthreads();
    const double dkk = S[k + k * ld];
    if (tid == 0) { sD[k] = dkk; if (!(fabs(dkk) <= dmax)) s_bad = 1; }
    const bool valid = fabs(dkk) > 0.0;
    for (int i = k + 1 + tid; i < M; i += nt) S[i + k * ld] = valid ? S[i + k * ld] / dkk : S[i + k * ld];
    __syncthreads();
  }

  // ---- 3. K^T = S^-1 (H P): thread n owns row n of K (the solution for state column n), in place
  const double tol = 1.0 / 1.7976931348623157e308;
  for (int n = tid; n < N; n += nt) {
    for (int m = 0; m < M; ++m) K[n + (long)m * a.ldk] = PHT[n + (long)m * a.ldpht];
    for (int k = 0; k < M; ++k) {   // P b
      const int p = perm[k];
      if (p != k) { const double t0 = K[n + (long)k * a.ldk]; K[n + (long)k * a.ldk] = K[n + (long)p * a.ldk]; K[n + (long)p * a.ldk] = t0; }
    }
    for (int i = 0; i < M; ++i) {   // L^-1
      double acc = K[n + (long)i * a.ldk];
      for (int j = 0; j < i; ++j) acc = fma(-S[i + j * ld], K[n + (long)j * a.ldk], acc);
      K[n + (long)i * a.ldk] = acc;
    }
    for (int i = 0; i < M; ++i) {   // D^-1 (pseudo-inverse with Eigen's tolerance)
      const double d = S[i + i * ld];
      K[n + (long)i * a.ldk] = fabs(d) > tol ? K[n + (long)i * a.ldk] / d : 0.0;
    }
    for (int i = M - 1; i >= 0; --i) {   // L^-T
      double acc = K[n + (long)i * a.ldk];
      for (int j = i + 1; j < M; ++j) acc = fma(-S[j + i * ld], K[n + (long)j * a.ldk], acc);
      K[n + (long)i * a.ldk] = acc;
    }
    for (int k = M - 1; k >= 0; --k) {   // P^T
      const int p = perm[k];
      if (p != k) { const double t0 = K[n + (long)k * a.ldk]; K[n + (long)k * a.ldk] = K[n + (long)p * a.ldk]; K[n + (long)p * a.ldk] = t0; }
    }
    // ---- 4. dx = K inn
    double dx = 0.0;
    for (int m = 0; m < M; ++m) dx = fma(K[n + (long)m * a.ldk], inn[m], dx);
    a.err[(long)filt * a.strideErr + n] = dx;
    if (!(fabs(dx) <= dmax)) s_bad = 1;   // (every writer stores the same value)
  }
  __syncthreads();
  if (s_bad) {   // not finite: prior covariance kept, nothing to absorb, status stays as the factorisation left it
    for (int n = tid; n < N; n += nt) a.err[(long)filt * a.strideErr + n] = 0.0;
    if (tid == 0) a.used[filt] = 0;
    return;
  }

  // ---- 5. A = I - K H (row n per thread: no write conflicts), T = A P, P+ = T A^T + K R K^T
  for (int n = tid; n < N; n += nt) {
    for (int c = 0; c < N; ++c) Am[n + (long)c * a.lda] = n == c ? 1.0 : 0.0;
    for (int m = 0; m < M; ++m) {
      const double knm = K[n + (long)m * a.ldk];
      H.for_each(m, [&](int c, double v) { Am[n + (long)c * a.lda] = fma(-knm, v, Am[n + (long)c * a.lda]); });
    }
  }
  __syncthreads();
  for (long e = tid; e < (long)N * N; e += nt) {
    const int i = (int)(e % N), j = (int)(e / N);
    double acc = 0.0;
    for (int c = 0; c < N; ++c) acc = fma(Am[i + (long)c * a.lda], P[c + (long)j * a.ldp], acc);
    T[i + (long)j * a.ldt] = acc;
  }
  __syncthreads();
  for (long e = tid; e < (long)N * N; e += nt) {
    const int i = (int)(e % N), j = (int)(e / N);
    if (i < j) continue;                       // lower triangle + mirror, as every other pipeline here writes P+
    double acc = 0.0;
    for (int c = 0; c < N; ++c) acc = fma(T[i + (long)c * a.ldt], Am[j + (long)c * a.lda], acc);
    for (int m = 0; m < M; ++m) acc = fma(K[i + (long)m * a.ldk] * dR[m], K[j + (long)m * a.ldk], acc);
    P[i + (long)j * a.ldp] = acc;
    P[j + (long)i * a.ldp] = acc;
  }
  if (tid == 0) { a.status[filt] = 0; a.used[filt] = 1; }
}

}  // namespace

int launch_ldlt_fallback(const LdltFallbackArgs& a, hipStream_t s) {
  if (a.batch <= 0) return 0;
  if (a.M > 512) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(ldlt_fallback_kernel, dim3(a.batch), dim3(1024), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace xivo_hip
