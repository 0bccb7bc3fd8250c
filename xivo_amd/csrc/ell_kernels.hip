// Kernels on the row-pair compressed H (see ell.h). HBM/L2-bound VALU work: every lane owns one
// value of the contiguous index (a state index n, or a measurement row for S) and streams whole
// columns of the source matrix, 512 B per wave load; no MFMA (a row pair has 21 non-zeros).
#include "ell.h"
#include "gate_device.h"

namespace xivo_hip {

namespace {

// ---------------------------------------------------------------- build from dense H^T
// One workgroup per filter. Phase A: per state column, in how many row pairs it is non-zero.
// Columns used by more than half of the non-empty pairs become the "common" slots (ascending, at most
// ELL_CW). Phase B: one wave per pair compacts the remaining non-zero columns in ascending order.
__global__ __launch_bounds__(256) void ell_build_kernel(const double* __restrict__ HTall, long strideHT, int ldht,
                                                        int Np, int Mp, EllBuffers e) {
  extern __shared__ int sh[];           // [Np] occupancy -> common slot + 1 ; then [pairs] non-empty flags
  const int filt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pairs = Mp / 2;
  int* occ = sh;
  int* nonempty = sh + Np;
  __shared__ int ccols[ELL_CW];
  __shared__ int s_nc, s_ne;
  const double* HT = HTall + (long)filt * strideHT;
  for (int p = tid; p < pairs; p += 256) nonempty[p] = 0;
  if (tid == 0) { e.over[filt] = 0; s_ne = 0; }
  __syncthreads();
  for (int n = tid; n < Np; n += 256) {
    int c = 0;
    for (int p = 0; p < pairs; ++p) {
      const bool nz = HT[n + (long)(2 * p) * ldht] != 0.0 || HT[n + (long)(2 * p + 1) * ldht] != 0.0;
      if (nz) { ++c; nonempty[p] = 1; }
    }
    occ[n] = c;
  }
  __syncthreads();
  for (int p = tid; p < pairs; p += 256)
    if (nonempty[p]) atomicAdd(&s_ne, 1);
  __syncthreads();
  if (tid == 0) {
    int nc = 0;
    const int ne = s_ne;
    for (int n = 0; n < Np; ++n) {
      const bool common = ne > 0 && 2 * occ[n] > ne && nc < ELL_CW;
      if (common) ccols[nc] = n;
      occ[n] = common ? ++nc : 0;        // slot + 1, 0 = private
    }
    s_nc = nc;
    e.nc[filt] = nc;
  }
  __syncthreads();
  const int nc = s_nc;
  int* idx = e.idx + (long)filt * e.stride_idx();
  double* val = e.val + (long)filt * e.stride_val();
  for (int p = wave; p < pairs; p += 4) {
    const double* h0 = HT + (long)(2 * p) * ldht;
    const double* h1 = h0 + ldht;
    int* pi = idx + (long)p * ELL_W;
    double* pv = val + (long)p * ELL_W * 2;
    if (lane < ELL_CW) {
      const int k = lane < nc ? ccols[lane] : 0;
      pi[lane] = k;
      pv[2 * lane] = lane < nc ? h0[k] : 0.0;
      pv[2 * lane + 1] = lane < nc ? h1[k] : 0.0;
    }
    int pos = 0;
    for (int n0 = 0; n0 < Np; n0 += 64) {
      const int n = n0 + lane;
      double a = 0.0, b = 0.0;
      bool nz = false;
      if (n < Np && occ[n] == 0) { a = h0[n]; b = h1[n]; nz = (a != 0.0) || (b != 0.0); }
      const unsigned long long m = __ballot(nz);
      const int my = pos + __popcll(m & ((1ull << lane) - 1ull));
      if (nz && my < ELL_PW) { pi[ELL_CW + my] = n; pv[2 * (ELL_CW + my)] = a; pv[2 * (ELL_CW + my) + 1] = b; }
      pos += __popcll(m);
    }
    if (pos > ELL_PW) { if (lane == 0) e.over[filt] = 1; }
    else if (lane >= pos && lane < ELL_PW) { pi[ELL_CW + lane] = -1; pv[2 * (ELL_CW + lane)] = 0.0; pv[2 * (ELL_CW + lane) + 1] = 0.0; }
  }
}

// ---------------------------------------------------------------- out = H_ell (x) Src
// Workgroup = (filter, block of 8 row pairs = 16 rows of H, chunk of 256 values of the contiguous
// index). The workgroups of one filter share an XCD so the source columns they re-read hit its L2.
// The common columns are loaded once per lane and reused by the 8 pairs.
template <int MODE>
__global__ __launch_bounds__(256) void ell_mul_kernel(EllMulArgs a) {
  const int xchunks = (a.X + 255) / 256;
  const int per = (a.Mp / 16) * xchunks;
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;
  const int filt = (slot / per) * 8 + xcd;
  if (filt >= a.batch) return;
  const int sub = slot % per;
  const int rb = sub / xchunks, xc = sub % xchunks;
  const int x = xc * 256 + threadIdx.x;
  const bool live = x < a.X;
  const int xs = live ? x : 0;
  const int* __restrict__ idx = a.ell.idx + (long)filt * a.ell.stride_idx() + (long)rb * 8 * ELL_W;
  const double* __restrict__ val = a.ell.val + (long)filt * a.ell.stride_val() + (long)rb * 8 * ELL_W * 2;
  const double* __restrict__ Src = a.Src + (long)filt * a.strideSrc + xs;
  const int nc = a.ell.nc[filt];

  double cm[ELL_CW];
#pragma unroll
  for (int t = 0; t < ELL_CW; ++t) cm[t] = t < nc ? Src[(long)idx[t] * a.ldsrc] : 0.0;

  double acc[8][2];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int* __restrict__ pi = idx + p * ELL_W;
    const double* __restrict__ pv = val + p * ELL_W * 2;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int t = 0; t < ELL_CW; ++t) {
      if (t < nc) { a0 = fma(pv[2 * t], cm[t], a0); a1 = fma(pv[2 * t + 1], cm[t], a1); }
    }
#pragma unroll
    for (int t = ELL_CW; t < ELL_W; ++t) {
      const int k = pi[t];
      if (k >= 0) {
        const double s = Src[(long)k * a.ldsrc];
        a0 = fma(pv[2 * t], s, a0); a1 = fma(pv[2 * t + 1], s, a1);
      }
    }
    acc[p][0] = a0; acc[p][1] = a1;
  }
  if (!live) return;
  const int m0 = rb * 16;
  double* __restrict__ out = a.out + (long)filt * a.strideOut + x;
  if (MODE == ELL_HP) {
    double* __restrict__ out2 = a.out2 + (long)filt * a.strideOut2 + m0 + (long)x * a.ldo2;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      out[(long)(m0 + 2 * p) * a.ldo] = acc[p][0];
      out[(long)(m0 + 2 * p + 1) * a.ldo] = acc[p][1];
      *reinterpret_cast<d2*>(out2 + 2 * p) = d2{acc[p][0], acc[p][1]};
    }
  } else if (MODE == ELL_S) {
    const double* __restrict__ dR = a.diagR + (long)filt * a.strideR;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + 2 * p + i;
        out[(long)m * a.ldo] = acc[p][i] + (x == m ? dR[m] : 0.0);
      }
    }
  } else {
    const double* __restrict__ dR = a.diagR + (long)filt * a.strideR;
    const double* __restrict__ K = a.K + (long)filt * a.strideK + x;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + 2 * p + i;
        out[(long)m * a.ldo] = fma(K[(long)m * a.ldk], dR[m], acc[p][i]);
      }
    }
  }
}

// ---------------------------------------------------------------- gating on ELL rows
__global__ __launch_bounds__(256) void gate_ell_kernel(GateEllArgs a) {
  const int filt = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  extern __shared__ double sdist[];  // F doubles + 1
  const int* __restrict__ idx = a.ell.idx + (long)filt * a.ell.stride_idx();
  double* val = a.ell.val + (long)filt * a.ell.stride_val();
  double* PHT = a.PHT + (long)filt * a.strideHT;
  double* inn = a.inn + (long)filt * a.strideInn;
  for (int f = tid; f < a.F; f += 256) {
    const int* pi = idx + (long)f * ELL_W;
    const double* pv = val + (long)f * ELL_W * 2;
    const double* c0 = PHT + (long)(2 * f) * a.ldht;      // P J0^T
    const double* c1 = c0 + a.ldht;                       // P J1^T
    double s00 = 0, s10 = 0, s11 = 0;
    for (int t = 0; t < ELL_W; ++t) {
      const int k = pi[t];
      if (k < 0) break;
      const double v0 = pv[2 * t], v1 = pv[2 * t + 1];
      const double p0 = c0[k], p1 = c1[k];
      s00 = fma(v0, p0, s00);
      s10 = fma(v1, p0, s10);
      s11 = fma(v1, p1, s11);
    }
    sdist[f] = mh_dist_2x2(s00 + a.R, s10, s11 + a.R, inn[2 * f], inn[2 * f + 1]);
  }
  __syncthreads();
  if (wave == 0) {
    const double th = relax_threshold(sdist, a.F, a.thresh, a.mult, a.min_inliers, lane);
    if (lane == 0) sdist[a.F] = th;
  }
  __syncthreads();
  const double th = sdist[a.F];
  for (int f = tid; f < a.F; f += 256) {
    const bool in = sdist[f] < th;
    a.mask[(long)filt * a.F + f] = in ? 1 : 0;
    a.dist[(long)filt * a.F + f] = sdist[f];
    if (!in) {
      inn[2 * f] = 0.0; inn[2 * f + 1] = 0.0;
      double* dr = a.diagR + (long)filt * a.strideR;
      dr[2 * f] = 1.0; dr[2 * f + 1] = 1.0;
      for (int t = 0; t < 2 * ELL_W; ++t) val[(long)f * ELL_W * 2 + t] = 0.0;
    }
  }
  double* H = a.H + (long)filt * a.strideH;
  double* HT = a.HT + (long)filt * a.strideHT;
  double* HP = a.HP + (long)filt * a.strideH;
  for (int f = 0; f < a.F; ++f) {
    if (sdist[f] < th) continue;
    for (int n = tid; n < a.Np; n += 256) {
      H[2 * f + (long)n * a.ldh] = 0.0;
      H[2 * f + 1 + (long)n * a.ldh] = 0.0;
      HP[2 * f + (long)n * a.ldh] = 0.0;
      HP[2 * f + 1 + (long)n * a.ldh] = 0.0;
      HT[n + (long)(2 * f) * a.ldht] = 0.0;
      HT[n + (long)(2 * f + 1) * a.ldht] = 0.0;
      PHT[n + (long)(2 * f) * a.ldht] = 0.0;
      PHT[n + (long)(2 * f + 1) * a.ldht] = 0.0;
    }
  }
}

}  // namespace

#define CHECK_LAUNCH() return (int)hipGetLastError()

int launch_ell_build(const double* HT, long strideHT, int ldht, int Np, int Mp, EllBuffers e, int batch,
                     hipStream_t s) {
  if (batch <= 0) return 0;
  const size_t lds = (size_t)(Np + Mp / 2) * sizeof(int);
  hipLaunchKernelGGL(ell_build_kernel, dim3(batch), dim3(256), lds, s, HT, strideHT, ldht, Np, Mp, e);
  CHECK_LAUNCH();
}

int launch_ell_mul(int mode, const EllMulArgs& a, hipStream_t s) {
  if (a.batch <= 0) return 0;
  const int xchunks = (a.X + 255) / 256;
  const int per = (a.Mp / 16) * xchunks;
  const int grid = ((a.batch + 7) / 8) * 8 * per;
  switch (mode) {
    case ELL_HP: hipLaunchKernelGGL((ell_mul_kernel<ELL_HP>), dim3(grid), dim3(256), 0, s, a); break;
    case ELL_S: hipLaunchKernelGGL((ell_mul_kernel<ELL_S>), dim3(grid), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((ell_mul_kernel<ELL_G>), dim3(grid), dim3(256), 0, s, a); break;
  }
  CHECK_LAUNCH();
}

int launch_gate_ell(const GateEllArgs& a, hipStream_t s) {
  if (a.batch <= 0) return 0;
  hipLaunchKernelGGL(gate_ell_kernel, dim3(a.batch), dim3(256), (size_t)(a.F + 1) * sizeof(double), s, a);
  CHECK_LAUNCH();
}

}  // namespace xivo_hip
