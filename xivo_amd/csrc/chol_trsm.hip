// Batched SPD factorisation and gain solve for the EKF measurement update.
//
// Replaces `K_.transpose() = S_.ldlt().solve(H_ * P_)` and `err_ = K_ * inn_`
// (/root/reference/src/estimator.cpp:1265-1267). The reference uses Eigen's
// pivoted LDL^T; S = HPH^T + R is SPD, so an un-pivoted blocked Cholesky is the
// same linear map up to rounding (tests assert 1e-8 on dx, 1e-6 on P).
//
// chol_f64_kernel : one 256-thread workgroup per filter, left-looking, 16x16
//   blocks (the v_mfma_f64_16x16x4_f64 tile). The diagonal block is factored in
//   registers by one wave (row per lane, v_readlane broadcasts); its explicit
//   inverse is kept so every panel / triangular-solve step is an MFMA.
//   Output: L in the lower triangle, L^T mirrored into the upper triangle (so
//   the backward solve also reads its A operand with the lane index contiguous).
// trsm_f64_kernel : each wave64 owns 16 right-hand-side columns and keeps the
//   whole Mp x 16 solution in accumulator registers. The f64 MFMA C/D layout
//   (row = (lane>>4) + 4*reg, col = lane&15) is exactly the B-operand layout of
//   the four k-slices of the next MFMA, so forward and backward substitution
//   chain with no data movement at all; dx = K*inn falls out as a lane reduce.
#include "common.h"

namespace xivo_hip {

namespace {

__device__ __forceinline__ double readlane_d(double v, int srclane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256) void chol_f64_kernel(CholArgs g) {
  const int filt = blockIdx.x;
  if (filt >= g.batch) return;
  double* __restrict__ S = g.S + (long)filt * g.strideS;
  double* invD = g.invD + (long)filt * g.strideInvD;
  const long ld = g.lds;
  const int nb = g.Mp / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;

  __shared__ double sP[4][16 * 17];
  __shared__ double sInv[2][256];
  __shared__ int sStatus;
  if (tid == 0) sStatus = 0;

  for (int j = 0; j < nb; ++j) {
    // ---- 1. partial sums of the diagonal block: sum_{k<j} L_jk L_jk^T, k split over waves
    {
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
      for (int k = wave; k < j; k += 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double a = S[(16 * j + li) + (long)(16 * k + 4 * s + lg) * ld];
          acc = mfma(a, a, acc);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) sP[wave][li * 17 + lg + 4 * r] = acc[r];
    }
    __syncthreads();

    // ---- 2. wave 0: factor the 16x16 diagonal block in registers, invert it
    if (wave == 0) {
      double x[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        double v = S[(16 * j + li) + (long)(16 * j + c) * ld];
        v -= sP[0][li * 17 + c] + sP[1][li * 17 + c] + sP[2][li * 17 + c] + sP[3][li * 17 + c];
        x[c] = v;
      }
      int bad = 0;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        double dcc = readlane_d(x[c], c);
        if (!(dcc > 0.0)) {
          if (!bad) bad = 1 + 16 * j + c;
          dcc = 1.0;
        }
        const double d = sqrt(dcc);
        const double rd = 1.0 / d;
        x[c] = (li == c) ? d : x[c] * rd;
#pragma unroll
        for (int q = c + 1; q < 16; ++q) {
          const double lqc = readlane_d(x[c], q);
          x[q] = fma(-x[c], lqc, x[q]);
        }
      }
      if (bad && lane == 0 && sStatus == 0) sStatus = bad;
      // inverse: lane jj solves L y = e_jj
      double y[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        double acc = (li == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) {
          const double lik = readlane_d(x[k], i);
          acc = fma(-lik, y[k], acc);
        }
        const double lii = readlane_d(x[i], i);
        y[i] = acc / lii;
      }
      if (lg == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          if (c <= li) {
            S[(16 * j + li) + (long)(16 * j + c) * ld] = x[c];
            S[(16 * j + c) + (long)(16 * j + li) * ld] = x[c];
          }
          sInv[0][c + li * 16] = y[c];   // inv(L)(c, li)
          sInv[1][li + c * 16] = y[c];   // inv(L)^T(li, c)
        }
      }
    }
    __syncthreads();
    // publish inverse blocks (coalesced)
    for (int e = tid; e < 512; e += 256) invD[(long)j * 512 + e] = (&sInv[0][0])[e];

    // ---- 3. panel: L_ij^T = inv(L_jj) * (S_ij^T - sum_k L_jk L_ik^T), i > j
    for (int i = j + 1 + wave; i < nb; i += 4) {
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
      for (int k = 0; k < j; ++k) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const long col = (long)(16 * k + 4 * s + lg) * ld;
          const double a = S[(16 * j + li) + col];
          const double b = S[(16 * i + li) + col];
          acc = mfma(a, b, acc);
        }
      }
      d4 rhs;
#pragma unroll
      for (int r = 0; r < 4; ++r) rhs[r] = S[(16 * i + li) + (long)(16 * j + lg + 4 * r) * ld] - acc[r];
      d4 out = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) out = mfma(sInv[0][li + (4 * s + lg) * 16], rhs[s], out);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        S[(16 * i + li) + (long)(16 * j + lg + 4 * r) * ld] = out[r];   // L(i-block, j-block)
        S[(16 * j + lg + 4 * r) + (long)(16 * i + li) * ld] = out[r];   // L^T mirrored to upper
      }
    }
    __syncthreads();
  }
  if (tid == 0) g.status[filt] = sStatus;
}

template <int NBM>
__global__ __launch_bounds__(256) void trsm_f64_kernel(TrsmArgs g) {
  const int chunks = (g.Np + 63) / 64;
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;
  const int filt = (slot / chunks) * 8 + xcd;
  const int chunk = slot % chunks;
  if (filt >= g.batch) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int c0 = chunk * 64 + wave * 16;
  if (c0 >= g.Np) return;
  const int nb = g.Mp / 16;

  const double* __restrict__ LU = g.LU + (long)filt * g.strideLU;
  const double* __restrict__ invD = g.invD + (long)filt * g.strideInvD;
  const double* __restrict__ HP = g.HP + (long)filt * g.strideHP;
  const long ld = g.ldlu;

  d4 X[NBM];
#pragma unroll
  for (int i = 0; i < NBM; ++i) {
    X[i] = d4{0.0, 0.0, 0.0, 0.0};
    if (i < nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) X[i][r] = HP[(16 * i + lg + 4 * r) + (long)(c0 + li) * g.ldhp];
    }
  }

  // forward: L Y = HP
#pragma unroll
  for (int k = 0; k < NBM; ++k) {
    if (k < nb) {
      d4 t = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) t = mfma(invD[(long)k * 512 + li + (4 * s + lg) * 16], X[k][s], t);
      X[k] = t;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = k + 1; i < NBM; ++i) {
          if (i < nb) {
            const double a = LU[(16 * i + li) + (long)(16 * k + 4 * s + lg) * ld];
            X[i] = mfma(-a, t[s], X[i]);
          }
        }
      }
    }
  }
  // backward: L^T K^T = Y
#pragma unroll
  for (int k = NBM - 1; k >= 0; --k) {
    if (k < nb) {
      d4 t = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) t = mfma(invD[(long)k * 512 + 256 + li + (4 * s + lg) * 16], X[k][s], t);
      X[k] = t;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < k; ++i) {
          const double a = LU[(16 * i + li) + (long)(16 * k + 4 * s + lg) * ld];
          X[i] = mfma(-a, t[s], X[i]);
        }
      }
    }
  }

  // K[(c0+li), m] = X[m-block][..];  dx[c0+li] = sum_m K * inn
  double* __restrict__ K = g.K + (long)filt * g.strideK;
  const double* __restrict__ inn = g.inn + (long)filt * g.strideInn;
  double part = 0.0;
#pragma unroll
  for (int i = 0; i < NBM; ++i) {
    if (i < nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * i + lg + 4 * r;
        K[(c0 + li) + (long)m * g.ldk] = X[i][r];
        part = fma(X[i][r], inn[m], part);
      }
    }
  }
  part += __shfl_xor(part, 16);
  part += __shfl_xor(part, 32);
  if (lg == 0) g.err[(long)filt * g.strideErr + c0 + li] = part;
}

template <int NBM>
int launch_trsm_t(const TrsmArgs& g, hipStream_t stream) {
  const int chunks = (g.Np + 63) / 64;
  const int grid = ((g.batch + 7) / 8) * 8 * chunks;
  hipLaunchKernelGGL((trsm_f64_kernel<NBM>), dim3(grid), dim3(256), 0, stream, g);
  return (int)hipGetLastError();
}

}  // namespace

int launch_chol_f64(const CholArgs& g, hipStream_t stream) {
  if (g.batch <= 0) return 0;
  hipLaunchKernelGGL(chol_f64_kernel, dim3(g.batch), dim3(256), 0, stream, g);
  return (int)hipGetLastError();
}

int launch_trsm_f64(const TrsmArgs& g, hipStream_t stream) {
  if (g.batch <= 0) return 0;
  const int nb = g.Mp / 16;
  if (nb <= 4) return launch_trsm_t<4>(g, stream);
  if (nb <= 8) return launch_trsm_t<8>(g, stream);
  if (nb <= 12) return launch_trsm_t<12>(g, stream);
  if (nb <= 16) return launch_trsm_t<16>(g, stream);
  if (nb <= 20) return launch_trsm_t<20>(g, stream);
  if (nb <= 24) return launch_trsm_t<24>(g, stream);
  return (int)hipErrorInvalidValue;
}

}  // namespace xivo_hip
