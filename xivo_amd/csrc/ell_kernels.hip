// Kernels on the row-pair compressed H (see ell.h). HBM/L2-bound VALU work: every lane owns one
// value of the contiguous index (a state index n, or a measurement row for S) and streams whole
// columns of the source matrix, 512 B per wave load; no MFMA (a row pair has 21 non-zeros).
#include "ell.h"
#include "gate_device.h"
#include <stdio.h>
#include <stdlib.h>

namespace xivo_hip {

namespace {

// ---------------------------------------------------------------- hand-over of a dense H
// Estimator::FilterUpdate hands UpdateJosephForm a dense H_ (M x N, column-major: src/update.cpp:129-138,
// src/estimator.h:496). One workgroup per filter turns it into the row-pair compressed form in a single
// pass over the matrix (each element leaves HBM once, 16-byte loads, lanes = row pairs so that a column
// of H is one contiguous run across the workgroup):
//   pass 1  thread p walks the columns n = 0..N-1 of its row pair (UNR loads in flight), appends the
//           non-zero (n, H[2p][n], H[2p+1][n]) to its list in LDS and the wave counts, per column, the
//           pairs that use it (ballot + popcount, one LDS row per wave - no atomics)
//   common  columns used by more than half of the non-empty pairs become the "common" slots (ascending,
//           at most ELL_CW) - one wave, ballot prefix sums
//   pass 2  thread p walks its list (ascending n): common columns go to their slot, the rest are packed
//           into the private slots; unused slots get (idx 0, value 0). Every thread writes one contiguous
//           112-byte index row and one 448-byte value row (merged into full lines by the L2).
// Rows [M, Mp_clear) are cleared (stale rows of a previous, larger M), inn / diagR copied with the
// neutral padding (0 / 1).
struct MeasCompressArgs {
  const double* H; long strideH; int ldh;
  const double* inn; long strideInn;
  const double* diagR; long strideR;
  int M, N, Np;
  int pairs_clear;                       // Mp_clear / 2
  EllBuffers ell;
  double* inn_out; long strideInnOut;
  double* R_out; long strideROut;
  int list_ld;                           // pairs rounded up to a multiple of 4 (LDS list row length)
  int* host_flags;                       // [batch][3] mirror of over / nc / pw in host-mapped memory (or null)
};

// UNR = 16-byte loads in flight per thread: 16 (32 measured within noise of 16 at 16384 filters; 48 for a single filter -
// fewer round trips in the column walk - measured SLOWER, 48.5 against 43.5 us: round 3)
template <bool ALIGNED, int UNR = 16>
__global__ __launch_bounds__(256) void meas_compress_kernel(MeasCompressArgs a) {
  extern __shared__ __attribute__((aligned(16))) double csh[];
  // LDS: lst_v[ELL_W][list_ld] d2 | lst_n[ELL_W][list_ld] int | occw[nw][Np] int | cslot[Np] int
  const int filt = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int pairs = (a.M + 1) / 2;
  d2* lst_v = reinterpret_cast<d2*>(csh);
  int* lst_n = reinterpret_cast<int*>(lst_v + ELL_W * a.list_ld);
  int* occw = lst_n + ELL_W * a.list_ld;
  int* cslot = occw + nw * a.Np;
  __shared__ int ccols[ELL_CW];
  __shared__ int s_nc, s_pw, s_over, s_ne[4];
  const double* __restrict__ H = a.H + (long)filt * a.strideH;
  if (tid == 0) { s_pw = 0; s_over = 0; }
  const bool mine = tid < pairs;
  const bool second = 2 * tid + 1 < a.M;          // M odd: the last pair has one row only
  const double* __restrict__ hp = H + (mine ? 2 * tid : 0);
  int cnt = 0;
  for (int n0 = 0; n0 < a.N; n0 += UNR) {
    d2 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int n = n0 + u < a.N ? n0 + u : a.N - 1;
      const double* q = hp + (long)n * a.ldh;
      if (ALIGNED) v[u] = *reinterpret_cast<const d2*>(q);
      else { v[u][0] = q[0]; v[u][1] = second ? q[1] : 0.0; }
    }
    int mycount = 0;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool in = mine && n0 + u < a.N;
      if (!second) v[u][1] = 0.0;
      const bool nz = in && (v[u][0] != 0.0 || v[u][1] != 0.0);
      const int c = __popcll(__ballot(nz));
      if (lane == u) mycount = c;
      if (nz) {
        if (cnt < ELL_W) { lst_v[cnt * a.list_ld + tid] = v[u]; lst_n[cnt * a.list_ld + tid] = n0 + u; }
        ++cnt;
      }
    }
    if (lane < UNR && n0 + lane < a.Np) occw[wave * a.Np + n0 + lane] = mycount;   // UNR <= 64
  }
  for (int n = a.N + (UNR - a.N % UNR) % UNR + tid; n < a.Np; n += blockDim.x)   // pad columns never visited above
    for (int w = 0; w < nw; ++w) occw[w * a.Np + n] = 0;
  {
    const int ne_w = __popcll(__ballot(cnt > 0));
    if (lane == 0) s_ne[wave] = ne_w;
  }
  __syncthreads();
  if (wave == 0) {
    int ne = 0;
    for (int w = 0; w < nw; ++w) ne += s_ne[w];
    int base = 0;
    for (int n0 = 0; n0 < a.Np; n0 += 64) {
      const int n = n0 + lane;
      int occ = 0;
      if (n < a.Np) for (int w = 0; w < nw; ++w) occ += occw[w * a.Np + n];
      const bool flag = n < a.N && ne > 0 && 2 * occ > ne;
      const unsigned long long m = __ballot(flag);
      const int rank = base + __popcll(m & ((1ull << lane) - 1ull));
      const bool common = flag && rank < ELL_CW;
      if (n < a.Np) cslot[n] = common ? rank + 1 : 0;
      if (common) ccols[rank] = n;
      base += __popcll(m);
    }
    if (lane == 0) { const int nc = base < ELL_CW ? base : ELL_CW; s_nc = nc; a.ell.nc[filt] = nc; }
  }
  __syncthreads();
  const int nc = s_nc;
  // pass 2 (round 5): the rows are assembled in LDS and leave as ONE flat, coalesced copy - a filter's compressed rows are
  // contiguous in memory (pairs x 28 index slots, pairs x 28 x 16 bytes of values). Every thread used to write its own
  // 112-byte index row and 448-byte value row slot by slot: 64 lanes x 16 bytes at a 448-byte stride per store instruction,
  // 0.4 of the kernel's 1.4 ms per 16384 filters (ablation without the stores). The staging reuses the lists' LDS, so every
  // thread first takes its list into registers (static indices: no scratch).
  const int walk = cnt < ELL_W ? cnt : ELL_W;
  d2 rv[ELL_W]; int rn[ELL_W];
#pragma unroll
  for (int k = 0; k < ELL_W; ++k) {
    const bool on = k < walk;          // (cnt > 0 only for tid < pairs <= list_ld)
    rn[k] = on ? lst_n[k * a.list_ld + tid] : 0;
    rv[k] = on ? lst_v[k * a.list_ld + tid] : d2{0.0, 0.0};
  }
  __syncthreads();
  const int stage_rows = a.pairs_clear < a.list_ld ? a.pairs_clear : a.list_ld;   // rows beyond the lists' LDS (stale rows of a larger M) are cleared in place
  if (tid < stage_rows) {
    int* oi = lst_n + tid * ELL_W;
    d2* ov = lst_v + tid * ELL_W;
#pragma unroll
    for (int t = 0; t < ELL_CW; ++t) { oi[t] = t < nc ? ccols[t] : 0; ov[t] = d2{0.0, 0.0}; }
#pragma unroll
    for (int t = 0; t < ELL_PW; ++t) { oi[ELL_CW + t] = 0; ov[ELL_CW + t] = d2{0.0, 0.0}; }
    int pos = 0;
#pragma unroll
    for (int k = 0; k < ELL_W; ++k) {
      if (k < walk) {
        const int cs = cslot[rn[k]];
        if (cs) ov[cs - 1] = rv[k];
        else {
          if (pos < ELL_PW) { oi[ELL_CW + pos] = rn[k]; ov[ELL_CW + pos] = rv[k]; }
          ++pos;
        }
      }
    }
    if (cnt > ELL_W) pos = ELL_PW + 1;            // list overflow: more than 28 non-zero columns cannot fit
    if (pos > ELL_PW) s_over = 1;
    atomicMax(&s_pw, pos);
  }
  __syncthreads();
  {
    int* gi = a.ell.idx + (long)filt * a.ell.stride_idx();
    d2* gv = reinterpret_cast<d2*>(a.ell.val + (long)filt * a.ell.stride_val());
    for (int e = tid; e < stage_rows * ELL_W; e += blockDim.x) { gi[e] = lst_n[e]; gv[e] = lst_v[e]; }
    for (int e = stage_rows * ELL_W + tid; e < a.pairs_clear * ELL_W; e += blockDim.x) {
      const int t = e % ELL_W;
      gi[e] = t < nc ? ccols[t] : 0; gv[e] = d2{0.0, 0.0};
    }
  }
  for (int m = tid; m < 2 * a.pairs_clear; m += blockDim.x) {
    a.inn_out[(long)filt * a.strideInnOut + m] = m < a.M ? a.inn[(long)filt * a.strideInn + m] : 0.0;
    a.R_out[(long)filt * a.strideROut + m] = m < a.M ? a.diagR[(long)filt * a.strideR + m] : 1.0;
  }
  __syncthreads();
  if (tid == 0) {
    a.ell.pw[filt] = s_pw; a.ell.over[filt] = s_over;
    if (a.host_flags) { a.host_flags[3 * filt] = s_over; a.host_flags[3 * filt + 1] = s_nc; a.host_flags[3 * filt + 2] = s_pw; }
  }
}

// Hand-over without compression (the LDS lists of meas_compress_kernel do not fit: very wide states): every filter is
// marked "does not fit" and takes the dense pipeline; inn / diagR get the same neutral padding.
__global__ __launch_bounds__(256) void meas_vectors_kernel(const double* __restrict__ inn, long strideInn, const double* __restrict__ diagR,
                                                          long strideR, int M, int Mp_clear, EllBuffers e, double* __restrict__ inn_out,
                                                          long strideInnOut, double* __restrict__ R_out, long strideROut) {
  const int filt = blockIdx.x;
  for (int m = threadIdx.x; m < Mp_clear; m += blockDim.x) {
    inn_out[(long)filt * strideInnOut + m] = m < M ? inn[(long)filt * strideInn + m] : 0.0;
    R_out[(long)filt * strideROut + m] = m < M ? diagR[(long)filt * strideR + m] : 1.0;
  }
  if (threadIdx.x == 0) { e.over[filt] = 1; e.nc[filt] = ELL_CW; e.pw[filt] = ELL_PW + 1; }
}

// Dense H / H^T (padded, both zero filled) of the filters whose rows fit the compressed form, rebuilt from it
// on demand (dense pipeline, xivo_hip_get_H); filters with over = 1 already hold their dense rows.
// rows >= Mrows keep what they hold (mixed stacking: the dense OOS rows behind the compressed in-state rows); HT may be null.
__global__ __launch_bounds__(256) void ell_to_dense_kernel(EllBuffers e, double* Hall, long strideH, int ldh,
                                                           double* HTall, long strideHT, int ldht, int Mp, int Np, int Mrows) {
  const int filt = blockIdx.x, tid = threadIdx.x;
  if (e.over[filt]) return;
  double* H = Hall + (long)filt * strideH;
  double* HT = HTall ? HTall + (long)filt * strideHT : nullptr;
  for (long i = tid; i < (long)Mp * Np; i += 256) {
    if ((i % Mp) < Mrows) H[(i % Mp) + (i / Mp) * ldh] = 0.0;
    if (HT && (i / Np) < Mrows) HT[(i % Np) + (i / Np) * ldht] = 0.0;
  }
  __syncthreads();
  const int* idx = e.idx + (long)filt * e.stride_idx();
  const double* val = e.val + (long)filt * e.stride_val();
  for (int i = tid; i < (Mrows / 2) * ELL_W; i += 256) {
    const int p = i / ELL_W, n = idx[i];
    const double v0 = val[2 * (long)i], v1 = val[2 * (long)i + 1];
    if (v0 != 0.0 || v1 != 0.0) {     // unused slots name column 0 with value 0
      H[2 * p + (long)n * ldh] = v0; H[2 * p + 1 + (long)n * ldh] = v1;
      if (HT) { HT[n + (long)(2 * p) * ldht] = v0; HT[n + (long)(2 * p + 1) * ldht] = v1; }
    }
  }
}

// rows [row0, row0 + nrows) of every filter's dense H, columns [c0, c1) (all of them: c0 = 0, c1 = Np): zero
__global__ __launch_bounds__(256) void zero_rows_kernel(double* Hall, long strideH, int ldh, int row0, int nrows, int c0, int c1) {
  double* H = Hall + (long)blockIdx.x * strideH;
  for (long i = threadIdx.x; i < (long)nrows * (c1 - c0); i += 256) H[row0 + (i % nrows) + (long)(c0 + i / nrows) * ldh] = 0.0;
}

// ---------------------------------------------------------------- out = H_ell (x) Src, gather form
// Fallback for sources too wide for the slab form below (more than ~528 columns). Workgroup = (filter, chunk of
// 256 values of the contiguous index, range of 16-row blocks of H); every lane streams the columns its pair
// names straight from L2 / Infinity Cache (a group block is re-fetched by every feature anchored to it: ~4x
// the size of the source per filter). Slot indices / values are wave-uniform scalar loads.
template <int MODE, int CWU>
__global__ __launch_bounds__(256) void ell_mul_kernel(EllMulArgs a) {
  const int xchunks = (a.X + 255) / 256;
  const int nrb = a.Mp / 16;
  const int rsplit = (nrb + a.rb_per_wg - 1) / a.rb_per_wg;
  const int per = rsplit * xchunks;
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;
  const int filt = (slot / per) * 8 + xcd;
  if (filt >= a.batch) return;
  const int sub = slot % per;
  const int rs = sub / xchunks, xc = sub % xchunks;
  const int x = xc * 256 + threadIdx.x;
  const bool live = x < a.X;
  const int xs = live ? x : 0;
  const double* __restrict__ Src = a.Src + (long)filt * a.strideSrc + xs;
  const int* __restrict__ idx0 = a.ell.idx + (long)filt * a.ell.stride_idx();
  const double* __restrict__ val0 = a.ell.val + (long)filt * a.ell.stride_val();

  // branch-free on purpose: unused slots name column 0 with value 0, so every load below is
  // unconditional and a pair's loads are all in flight together
  double cm[CWU];
#pragma unroll
  for (int t = 0; t < CWU; ++t) cm[t] = Src[(long)idx0[t] * a.ldsrc];

  const double* __restrict__ dR = a.diagR + (long)filt * a.strideR;
  const double* __restrict__ K = a.K + (long)filt * a.strideK + xs;
  double* __restrict__ out = a.out + (long)filt * a.strideOut + xs;
  double* __restrict__ out2 = a.out2 + (long)filt * a.strideOut2 + (long)xs * a.ldo2;
  const int p_end = 8 * min(nrb, (rs + 1) * a.rb_per_wg);
  // runtime loop over row pairs, two per trip: bounds the loads in flight (24 per lane) and the registers
#pragma unroll 2
  for (int p = 8 * rs * a.rb_per_wg; p < p_end; ++p) {
    const int* __restrict__ pi = idx0 + (long)p * ELL_W + ELL_CW;
    const double* __restrict__ pv = val0 + (long)p * ELL_W * 2;
    double sv[ELL_PW];
#pragma unroll
    for (int t = 0; t < ELL_PW; ++t) sv[t] = Src[(long)pi[t] * a.ldsrc];
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int t = 0; t < CWU; ++t) { a0 = fma(pv[2 * t], cm[t], a0); a1 = fma(pv[2 * t + 1], cm[t], a1); }
#pragma unroll
    for (int t = 0; t < ELL_PW; ++t) {
      a0 = fma(pv[2 * (ELL_CW + t)], sv[t], a0); a1 = fma(pv[2 * (ELL_CW + t) + 1], sv[t], a1);
    }
    if (!live) continue;
    const int m = 2 * p;
    if (MODE == ELL_HP) {
      out[(long)m * a.ldo] = a0;
      out[(long)(m + 1) * a.ldo] = a1;
      *reinterpret_cast<d2*>(out2 + m) = d2{a0, a1};
    } else if (MODE == ELL_S) {
      out[(long)m * a.ldo] = a0 + (x == m ? dR[m] : 0.0);
      out[(long)(m + 1) * a.ldo] = a1 + (x == m + 1 ? dR[m + 1] : 0.0);
    } else {
      out[(long)m * a.ldo] = fma(K[(long)m * a.ldk], dR[m], a0);
      out[(long)(m + 1) * a.ldo] = fma(K[(long)(m + 1) * a.ldk], dR[m + 1], a1);
    }
  }
}

// ---------------------------------------------------------------- gating on ELL rows
// (device function: the stand-alone kernel below, and the tail of the S kernel when one workgroup owns a whole filter)
__device__ __forceinline__ void gate_ell_body(const GateEllArgs& a, int filt, double* sdist /* LDS: F doubles + 1 */) {
  const int nt = blockDim.x, nwv = nt >> 6;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int* __restrict__ idx = a.ell.idx + (long)filt * a.ell.stride_idx();
  double* val = a.ell.val + (long)filt * a.ell.stride_val();
  double* PHT = a.PHT + (long)filt * a.strideHT;
  double* inn = a.inn + (long)filt * a.strideInn;
  double* Sm = a.from_S ? a.S + (long)filt * a.strideS : nullptr;
  if (a.from_S) {
    const double* dr0 = a.diagR + (long)filt * a.strideR;
    for (int f = tid; f < a.F; f += nt) {
      const double* sd = a.Sdiag ? a.Sdiag + (long)filt * a.strideSdiag + 4 * f : nullptr;   // [row in block][column in block]
      const double s00 = (sd ? sd[0] : Sm[2 * f + (long)(2 * f) * a.lds]) - dr0[2 * f] + a.R;
      const double s10 = sd ? sd[2] : Sm[2 * f + 1 + (long)(2 * f) * a.lds];
      const double s11 = (sd ? sd[3] : Sm[2 * f + 1 + (long)(2 * f + 1) * a.lds]) - dr0[2 * f + 1] + a.R;
      sdist[f] = mh_dist_2x2(s00, s10, s11, inn[2 * f], inn[2 * f + 1]);
    }
  }
  // one wave per feature, one lane per slot: the 2 x 28 gathers of a feature are one round trip
  for (int f = wave; f < (a.from_S ? 0 : a.F); f += nwv) {
    const double* c0 = PHT + (long)(2 * f) * a.ldht;      // P J0^T
    const double* c1 = c0 + a.ldht;                       // P J1^T
    double s00 = 0, s10 = 0, s11 = 0;
    if (lane < ELL_W) {
      const int k = idx[(long)f * ELL_W + lane];
      const double v0 = val[((long)f * ELL_W + lane) * 2], v1 = val[((long)f * ELL_W + lane) * 2 + 1];
      const double p0 = c0[k], p1 = c1[k];
      s00 = v0 * p0; s10 = v1 * p0; s11 = v1 * p1;
    }
    s00 = wave_sum(s00); s10 = wave_sum(s10); s11 = wave_sum(s11);
    if (lane == 0) sdist[f] = mh_dist_2x2(s00 + a.R, s10, s11 + a.R, inn[2 * f], inn[2 * f + 1]);
  }
  __syncthreads();
  if (wave == 0) {
    const double th = relax_threshold(sdist, a.F, a.thresh, a.mult, a.min_inliers, lane);
    if (lane == 0) sdist[a.F] = th;
  }
  __syncthreads();
  const double th = sdist[a.F];
  for (int f = tid; f < a.F; f += nt) {
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
  double* H = a.H ? a.H + (long)filt * a.strideH : nullptr;       // dense copies, when they are materialised
  double* HT = a.HT ? a.HT + (long)filt * a.strideHT : nullptr;
  double* HP = a.HP ? a.HP + (long)filt * a.strideH : nullptr;
  for (int f = 0; f < a.F; ++f) {
    if (sdist[f] < th) continue;
    for (int n = tid; n < a.Np; n += nt) {
      if (H) {
        H[2 * f + (long)n * a.ldh] = 0.0;
        H[2 * f + 1 + (long)n * a.ldh] = 0.0;
        HT[n + (long)(2 * f) * a.ldht] = 0.0;
        HT[n + (long)(2 * f + 1) * a.ldht] = 0.0;
      }
      PHT[n + (long)(2 * f) * a.ldht] = 0.0;
      PHT[n + (long)(2 * f + 1) * a.ldht] = 0.0;
      if (HP) {                                   // (strided: 2 Np scattered 8-byte stores per rejected feature)
        HP[2 * f + (long)n * a.ldh] = 0.0;
        HP[2 * f + 1 + (long)n * a.ldh] = 0.0;
      }
    }
    if (Sm) {
      for (int jx = tid; jx < a.Mp; jx += nt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = 2 * f + i;
          Sm[r + (long)jx * a.lds] = 0.0;
          Sm[jx + (long)r * a.lds] = 0.0;
        }
      }
    }
  }
  if (Sm) {
    __syncthreads();
    for (int f = tid; f < a.F; f += nt) {
      if (sdist[f] < th) continue;
      Sm[2 * f + (long)(2 * f) * a.lds] = 1.0;
      Sm[2 * f + 1 + (long)(2 * f + 1) * a.lds] = 1.0;
    }
  }
}

__global__ __launch_bounds__(1024) void gate_ell_kernel(GateEllArgs a) {
  extern __shared__ double sdist[];  // F doubles + 1; 256 threads for a big batch, 1024 when few filters must finish fast
  gate_ell_body(a, blockIdx.x, sdist);
}

// ---------------------------------------------------------------- slab-in-LDS variant
// The gather form above re-fetches a source column once per row pair that names it (a group block is
// shared by the features anchored to it, and the 12 pose columns by everybody): ~4x the size of the
// source per filter, served by L2 / Infinity Cache. Here a workgroup first parks an XC-wide slab of
// the source (all columns x XC values of the contiguous index) in LDS - each source element leaves HBM
// exactly once - together with the slot VALUES of all row pairs of the filter, and then its waves
// walk the pairs: lane = x gathers the pair's private columns from the slab (ds_read_b64), the
// coefficients arrive as wave-uniform ds_read_b128 broadcasts (one (row 2p, row 2p+1) pair each)
// and the slot indices as scalar loads. (Coefficients as scalar loads serialise on the ~100 SGPRs
// and miss the scalar cache - 38 KB per slab; measured 3x slower.)
//   XC = 64: lane = x, both rows of the pair per lane     XC = 32: half-waves take one row each
// LDS: slab[cols][XC] doubles, element (k, x) at k * XC + (x ^ (k & 15)) - the XOR keeps both the
// row-wise loads and the transposing load of ELL_S (lanes = k) conflict-free without a pad column;
// then ops[pairs][CWU + ELL_PW][2].
#ifndef XIVO_ELL_HP_PF
#define XIVO_ELL_HP_PF 0   // A/B: P H^T with 8 waves and the next slab prefetched into registers under the walk
#endif
#ifndef XIVO_ELL_ABL
#define XIVO_ELL_ABL 0   // timing-only ablations of ell_tile_kernel (scripts/build_variant.sh): 1 no pair walk, 2 no slab loads, 3 no output stores
#endif
typedef __attribute__((address_space(4))) const double ell_cdouble;
typedef __attribute__((address_space(4))) const int ell_cint;

// threads per workgroup: 16 waves (128 VGPRs each) walk the pairs fastest; the S instantiation (transposing
// prefetch, 40 doubles per thread) needs the 256-VGPR budget of 8 waves
// prefetch variant (next slab fetched into registers under the pair walk): the hot configuration only
constexpr bool ell_tile_pf(int mode, int cwu, int xc, int pwu) {
  // (tried for the P H^T and G instantiations too, at 8 waves / 256 VGPRs: 0.83 -> 0.92 and 0.88 -> 1.79 ms)
  return xc == 64 && cwu == 12 && (mode == ELL_S || (XIVO_ELL_HP_PF && mode == ELL_HP && pwu == 9)) && pwu >= 0;
}
// raw buffer access: one 32-bit per-lane offset + a scalar offset per access, no 64-bit address registers per store
typedef unsigned ell_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ell_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ double ell_buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void ell_buf_st(double v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ell_u2, v), r, voff, soff, 0);
}

// matrix-core walk (see the kernel): the hot instantiations
constexpr bool ell_tile_mf(int mode, int cwu, int xc, int pwu) {
  return (xc == 64 || xc == 32) && cwu % 4 == 0 && pwu % 3 == 0 && (mode == ELL_HP || mode == ELL_S);
}
constexpr int ell_tile_threads(int mode, int cwu, int xc, int pwu) {
  return (mode == ELL_S || ell_tile_pf(mode, cwu, xc, pwu)) ? 512 : 1024;
}

// TALL: the prefetch registers cover more than 256 source rows (cols up to 320); the short form saves the S instantiation 16 VGPRs
template <int MODE, int CWU, int XC, int PWU, bool TALL = true>
__global__ __launch_bounds__(ell_tile_threads(MODE, CWU, XC, PWU)) void ell_tile_kernel(EllMulArgs a) {
  constexpr int NSLOT = CWU + PWU;   // PWU: private slots actually walked (9 for XIVO's group + feature blocks)
  constexpr int NT = ell_tile_threads(MODE, CWU, XC, PWU), NW = NT / 64;
  // slab elements per thread held in registers while the previous slab is being consumed
  // covers every cols the LDS can hold (XC = 64: <= 272; XC = 32: <= 528)
  // (prefetching under the pair walk pays for ELL_S only: at 16 waves per CU the 128-VGPR budget of the other
  //  two modes cannot hold a slab share next to the walk without spilling - measured slower)
  constexpr bool PF = ell_tile_pf(MODE, CWU, XC, PWU);
  constexpr int UNR = (PF || (CWU == 12 && XC == 64)) ? 2 : 1;   // pairs in flight per wave (register budget)
  constexpr int RN = PF ? (MODE == ELL_S ? (TALL ? 40 : 32) : 34) : 8;
  extern __shared__ __attribute__((aligned(16))) double tile[];
  const int xchunks = (a.X + XC - 1) / XC;
  const int wgs = (xchunks + a.slabs_per_wg - 1) / a.slabs_per_wg;     // workgroups per filter
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;
  int filt = (slot / wgs) * 8 + xcd;     // (persistent form: the first filter of this workgroup; then filt += a.persist_stride)
  if (filt >= a.batch) return;
  const int s_begin = (slot % wgs) * a.slabs_per_wg;
  const int s_end = min(xchunks, s_begin + a.slabs_per_wg);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double* Src = a.Src + (long)filt * a.strideSrc;
  const int cols = a.cols;
  const int pairs = a.Mp / 2;
  double* ops = tile + (long)cols * XC;
  // the private slot indices of every pair, 16 bits each, ELL_PIW per pair (24 bytes: three 8-byte broadcast reads). As
  // scalar loads they cost the walk most of its time: one scalar-cache miss per pair (9 KB of indices per filter, the cache
  // is shared between CUs), and SMEM shares lgkmcnt with the LDS - every wait for an index drained the LDS reads of the
  // other pair in flight too (walk alone, 16384 filters: P H^T 1.91 ms, S 1.77 ms).
  unsigned short* pidx = reinterpret_cast<unsigned short*>(ops + (long)pairs * NSLOT * 2);
  auto stage_ops = [&](int f) {  // coefficients of every pair of filter f: coalesced 16-byte loads -> LDS
    const d2* __restrict__ gv = reinterpret_cast<const d2*>(a.ell.val + (long)f * a.ell.stride_val());
    for (int e = tid; e < pairs * NSLOT; e += NT) {
      const int p = e / NSLOT, t = e % NSLOT;
      *reinterpret_cast<d2*>(ops + 2 * e) = gv[p * ELL_W + (t < CWU ? t : ELL_CW + (t - CWU))];
    }
    const int* __restrict__ gi = a.ell.idx + (long)f * a.ell.stride_idx();
    for (int e = tid; e < pairs * ELL_PIW; e += NT) {
      const int p = e / ELL_PIW, t = e % ELL_PIW;
      pidx[e] = t < PWU ? (unsigned short)gi[p * ELL_W + ELL_CW + t] : (unsigned short)0;
    }
    if (tid < ELL_CW) pidx[pairs * ELL_PIW + tid] = (unsigned short)gi[tid];   // the common slots (the same in every pair)
  };
  stage_ops(filt);
  // Persistent form (big batches of the prefetching instantiation: a.persist_stride > 0, one workgroup per CU walks filters
  // filt, filt + stride, ...): the next filter's first slab is prefetched under the last walk of this one like any other
  // slab, and its coefficients are staged while that prefetch is still on its way - a workgroup that owns the CU's LDS has
  // nobody to hide the start-up of a filter (two dependent round trips before the first walk) behind. (Holding the next
  // coefficients in registers under the last walk as well costs 24 VGPRs the S instantiation does not have: 116 spills.)
  // The slab of step s+1 is fetched into registers while the pairs are walked over the slab of step s
  // (one workgroup per CU owns the LDS, so nothing else would hide the HBM latency of the next slab).
  //   ELL_S : source = P H^T [cols x Mp], column j contiguous over the state index: slab[k][jj] = PHT[k, x0 + jj];
  //           thread (wave, lane) fetches column jj = wave + NW q, rows k = lane + 64 u
  //   else  : source [X x cols], contiguous index first: slab[k][xx] = Src[x0 + xx, k];
  //           thread fetches xx = tid % XC, k = tid / XC + (NT / XC) u
  double r[RN];
  auto fetch = [&](const double* FSrc, int sidx) {
    const int x0 = sidx * XC;
    if (MODE == ELL_S) {
      // one buffer resource per filter, the column as a scalar offset, the row as one 32-bit per-lane offset: no 64-bit
      // address registers (hoisted out of the slab loop they spilled, and a scratch reload waits on vmcnt - behind every
      // prefetch load in flight)
      constexpr int KU = RN / (XC / NW);
      const __amdgpu_buffer_rsrc_t rs = ell_rsrc(FSrc);
      const unsigned vo = (unsigned)lane * 8u;
#pragma unroll
      for (int q = 0; q < XC / NW; ++q) {
        const int jj = wave + NW * q;
        const bool ok = x0 + jj < a.X;
        const unsigned so = (unsigned)((x0 + (ok ? jj : 0)) * a.ldsrc) * 8u;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const int k = lane + 64 * u;
          r[q * KU + u] = (XIVO_ELL_ABL != 2 && ok && k < cols) ? ell_buf_ld(rs, vo, so + (unsigned)(64 * u) * 8u) : 0.0;
        }
      }
    } else {
      // uniform 64-bit base (+ u * step, scalar) and one 32-bit per-lane byte offset: no per-load address VGPRs.
      // cols is a multiple of 16 and NT / XC divides 16, so "k < cols" is the same for every lane of a step
      const int xx = tid % XC, kq = tid / XC;
      const bool ok = x0 + xx < a.X;
      const unsigned voff = ((unsigned)(ok ? xx : 0) + (unsigned)kq * (unsigned)a.ldsrc) * 8u;
      const char* base = reinterpret_cast<const char*>(FSrc + x0);
      const size_t step = (size_t)(NT / XC) * (size_t)a.ldsrc * 8u;
      const int nu = cols / (NT / XC);
#pragma unroll
      for (int u = 0; u < RN; ++u) {
        double v = 0.0;
        if (XIVO_ELL_ABL != 2 && u < nu) v = *reinterpret_cast<const double*>(base + (size_t)u * step + voff);
        r[u] = ok ? v : 0.0;
      }
    }
  };
  auto park = [&]() {
    if (MODE == ELL_S) {
      constexpr int KU = RN / (XC / NW);
#pragma unroll
      for (int q = 0; q < XC / NW; ++q) {
        const int jj = wave + NW * q;
#pragma unroll
        for (int u = 0; u < KU; ++u) { const int k = lane + 64 * u; if (k < cols) tile[k * XC + (jj ^ (k & 15))] = r[q * KU + u]; }
      }
    } else {
      const int xx = tid % XC, kq = tid / XC;
#pragma unroll
      for (int u = 0; u < RN; ++u) { const int k = kq + (NT / XC) * u; if (k < cols) tile[k * XC + (xx ^ (k & 15))] = r[u]; }
    }
  };

  const int xx = lane % XC, half = lane / XC;
  auto slab = [&](int k) -> double { return tile[k * XC + (xx ^ (k & 15))]; };

  const int fstep = PF ? a.persist_stride : 0;
  if (PF) fetch(Src, s_begin);
  for (;;) {   // (one trip unless persistent)
  ell_cint* idx0 = (ell_cint*)(a.ell.idx + (long)filt * a.ell.stride_idx());
  ell_cdouble* dR = (ell_cdouble*)(a.diagR + (long)filt * a.strideR);
  const int nfilt = filt + fstep;
  const bool more = fstep > 0 && nfilt < a.batch;
  for (int sidx = s_begin; sidx < s_end; ++sidx) {
    if (PF) {
      park();
    } else if (MODE == ELL_S) {
      const int x0 = sidx * XC;
      for (int jj = wave; jj < XC; jj += NW) {
        const bool ok = x0 + jj < a.X;
        const double* __restrict__ col = Src + (long)(x0 + (ok ? jj : 0)) * a.ldsrc;
        for (int k0 = 0; k0 < cols; k0 += 64 * 8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) { const int k = k0 + lane + 64 * u; r[u] = (ok && k < cols) ? col[k] : 0.0; }
#pragma unroll
          for (int u = 0; u < 8; ++u) { const int k = k0 + lane + 64 * u; if (k < cols) tile[k * XC + (jj ^ (k & 15))] = r[u]; }
        }
      }
    } else {
      // straight copy, 8 loads in flight per thread
      const int x0 = sidx * XC;
      const int xq = tid % XC, kq = tid / XC;
      const bool ok = x0 + xq < a.X;
      const double* __restrict__ row = Src + x0 + (ok ? xq : 0);
      for (int k0 = 0; k0 < cols; k0 += (NT / XC) * 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + kq + (NT / XC) * u; r[u] = (XIVO_ELL_ABL != 2 && ok && k < cols) ? row[(long)k * a.ldsrc] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + kq + (NT / XC) * u; if (k < cols) tile[k * XC + (xq ^ (k & 15))] = r[u]; }
      }
    }
    // (prefetching form: LDS-only barrier - __syncthreads() would also wait for the acknowledgement of the output stores the
    //  walk before just issued, a round trip to memory per slab; the prefetched registers were waited for by park())
    if (PF) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
    if (PF) {
      if (sidx + 1 < s_end) fetch(Src, sidx + 1);
      else if (more) fetch(a.Src + (long)nfilt * a.strideSrc, s_begin);
    }

    if constexpr (ell_tile_mf(MODE, CWU, XC, PWU)) {
      // ---- walk, matrix-core form (the hot instantiations: 12 common + 9 private slots, 64-wide slabs) --------------------
      // The scalar-coefficient walk below is bound by the LDS: every coefficient pair reaches its 64 lanes as a 16-byte
      // broadcast read that costs the LDS return path the full 1 KB (8 clocks), 21 of them per row pair and wave - 1.9 of
      // the 2.8 ms of P H^T and 1.8 of the 2.3 ms of S per 16384 filters were the walk, not the slab traffic. Here a wave
      // takes 8 row pairs x 64 slab columns at a time:
      //  * the 12 common columns are a dense product (16 rows x 12) (12 x 64): three v_mfma_f64_16x16x4_f64 per 16-column
      //    block, the B operand = the slab rows the common slots name (12 registers per lane and slab, as before), the A
      //    operand = the coefficients, one 8-byte per-lane read per k-step;
      //  * MFMA row lg + 4 r of the tile stands for row (r & 1) of pair p0 + lg + 4 (r >> 1): both rows of a pair sit in
      //    the same lane, so the private part keeps one slab gather per (pair, slot, column block) for both rows, and its
      //    coefficient reads serve four pairs per instruction (lane groups lg read different pairs) instead of one.
      // LDS time per 8 pairs x 64 columns: 72 gathers + 18 coefficient reads + 3 operand reads, ~450 clocks against ~1630.
      constexpr int NC = XC / 16, KS = CWU / 4;   // 16-column blocks per slab, k-steps of the common product
      const int li = lane & 15, lg = lane >> 4;
      const unsigned short* cidx = pidx + pairs * ELL_PIW;
      int ck[KS];                                                        // slab rows the common slots 4 s + lg name
#pragma unroll
      for (int s = 0; s < KS; ++s) ck[s] = cidx[4 * s + lg];
      const int x0 = sidx * XC;
      const __amdgpu_buffer_rsrc_t rO = ell_rsrc(a.out + (long)filt * a.strideOut);
      const unsigned vO = (unsigned)(2 * lg * a.ldo + li) * 8u;          // row 2 lg, column li of a tile; the rest of the address is wave-uniform
      double dRc[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) dRc[c] = 0.0;
      if (MODE == ELL_S) {
#pragma unroll
        for (int c = 0; c < NC; ++c) if (x0 + 16 * c + li < a.X) dRc[c] = a.diagR[(long)filt * a.strideR + x0 + 16 * c + li];
      }
      // (S: the row-pair blocks to the right of this slab's diagonal square are skipped - lower triangle + diagonal blocks)
      const int rb_end = XIVO_ELL_ABL == 1 ? 0 : (MODE == ELL_S ? min(pairs / 8, NC * (sidx + 1)) : pairs / 8);
      const int pa = (li & 3) + 4 * (li >> 3), ra = (li >> 2) & 1;     // A operand: MFMA row li = pair pa, row ra of it
      for (int rb = wave; rb < rb_end; rb += NW) {
        const int p0 = 8 * rb;
        d4 acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
        // the private slots first (plain FMAs into the zeroed tile), the dense common part on top of them: the other way
        // round the scheduler hoists every LDS read of the private part above the MFMA chain it depends on - and spills
        if constexpr (NT == 512) {   // (the 8-wave instantiations have the registers for it: 256 per wave)
          // all slot indices of this lane's two pairs up front (packed 16-bit, 6 registers per pair): a step then waits for
          // ONE LDS round trip - its gathers - instead of two (index, then gather); the next step's coefficients are
          // requested before this step's FMAs
          uint2 iq[2][ELL_PIW / 4];
  #pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint2* pq = reinterpret_cast<const uint2*>(pidx + (p0 + lg + 4 * q) * ELL_PIW);
  #pragma unroll
            for (int u = 0; u < ELL_PIW / 4; ++u) iq[q][u] = pq[u];
          }
  #pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int p = p0 + lg + 4 * q;
            const d2* pv = reinterpret_cast<const d2*>(ops) + (long)p * NSLOT + CWU;
            static_assert(PWU % 3 == 0, "three slots per step");
            d2 vn[3];
  #pragma unroll
            for (int u = 0; u < 3; ++u) vn[u] = pv[u];
  #pragma unroll
            for (int t0 = 0; t0 < PWU; t0 += 3) {
              // three slots per step, their 12 gathers in flight together (the scheduler, left alone, serialises them in the
              // second half-tile - one LDS round trip per gather)
              d2 v[3]; double g[3][NC];
  #pragma unroll
              for (int u = 0; u < 3; ++u) {
                const int t = t0 + u;
                v[u] = vn[u];
                const unsigned w = (t & 2) ? iq[q][t >> 2].y : iq[q][t >> 2].x;
                const int k = (int)((t & 1) ? (w >> 16) : (w & 0xffffu));
                const double* trow = tile + k * XC;
                const int sw = k & 15;
  #pragma unroll
                for (int c = 0; c < NC; ++c) g[u][c] = trow[(16 * c + li) ^ sw];
              }
              if (t0 + 3 < PWU) {
  #pragma unroll
                for (int u = 0; u < 3; ++u) vn[u] = pv[t0 + 3 + u];
              }
              __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
              for (int u = 0; u < 3; ++u) {
  #pragma unroll
                for (int c = 0; c < NC; ++c) {
                  acc[c][2 * q] = fma(v[u][0], g[u][c], acc[c][2 * q]);
                  acc[c][2 * q + 1] = fma(v[u][1], g[u][c], acc[c][2 * q + 1]);
                }
              }
            }
          }
        } else {
  #pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int p = p0 + lg + 4 * q;
            const d2* pv = reinterpret_cast<const d2*>(ops) + (long)p * NSLOT + CWU;
            const unsigned short* pkp = pidx + p * ELL_PIW;
            static_assert(PWU % 3 == 0, "three slots per step");
  #pragma unroll 1
            for (int t0 = 0; t0 < PWU; t0 += 3) {
              // three slots per step, their 12 gathers in flight together (the scheduler, left alone, serialises them in the
              // second half-tile - one LDS round trip per gather)
              d2 v[3]; double g[3][NC];
  #pragma unroll
              for (int u = 0; u < 3; ++u) {
                v[u] = pv[t0 + u];
                const int k = (int)pkp[t0 + u];
                const double* trow = tile + k * XC;
                const int sw = k & 15;
  #pragma unroll
                for (int c = 0; c < NC; ++c) g[u][c] = trow[(16 * c + li) ^ sw];
              }
              __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
              for (int u = 0; u < 3; ++u) {
  #pragma unroll
                for (int c = 0; c < NC; ++c) {
                  acc[c][2 * q] = fma(v[u][0], g[u][c], acc[c][2 * q]);
                  acc[c][2 * q + 1] = fma(v[u][1], g[u][c], acc[c][2 * q + 1]);
                }
              }
            }
          }
        }
        {
          // (the B operand is re-read from the slab per tile, 12 reads: held across the private part its 24 registers made
          //  the scheduler serialise that part's gathers)
          double av[KS], cb[KS][NC];
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            av[s] = ops[((long)(p0 + pa) * NSLOT + 4 * s + lg) * 2 + ra];
#pragma unroll
            for (int c = 0; c < NC; ++c) cb[s][c] = tile[ck[s] * XC + ((16 * c + li) ^ (ck[s] & 15))];
          }
#pragma unroll
          for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int s = 0; s < KS; ++s) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], cb[s][c], acc[c], 0, 0, 0);
          }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int x = x0 + 16 * c + li;
          if (x >= a.X || (XIVO_ELL_ABL == 3 && acc[c][0] != 12345.678)) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int pp = p0 + lg + 4 * (r >> 1), i = r & 1, m = 2 * pp + i;
            const unsigned sO = (unsigned)((2 * p0 + 8 * (r >> 1) + i) * a.ldo + x0 + 16 * c) * 8u;
            if (MODE == ELL_HP) ell_buf_st(acc[c][r], rO, vO, sO);
            else {
              const double sv_ = acc[c][r] + (x == m ? dRc[c] : 0.0);
              ell_buf_st(sv_, rO, vO, sO);
              if (a.diag_out && (x >> 1) == pp) a.diag_out[(long)filt * a.strideDiag + 4 * pp + 2 * (x & 1) + i] = sv_;
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      continue;
    }
    const int x = sidx * XC + xx;
    const bool live = x < a.X;
    const double* __restrict__ K = a.K + (long)filt * a.strideK + (live ? x : 0);
    double* __restrict__ out = a.out + (long)filt * a.strideOut + (live ? x : 0);
    float* __restrict__ outf = reinterpret_cast<float*>(a.out) + (long)filt * a.strideOut + (live ? x : 0);
    double cm[CWU];
#pragma unroll
    for (int t = 0; t < CWU; ++t) cm[t] = slab(idx0[t]);
    // S is symmetric and every reader (gating, both Cholesky kernels) takes its lower triangle + the diagonal blocks: lane
    // = row x of S, pair p = columns 2p, 2p + 1 - the pairs to the right of this slab's 64 x 64 diagonal square are skipped
    const int p_end = XIVO_ELL_ABL == 1 ? 0 : (MODE == ELL_S ? min(pairs, (sidx * XC + XC) / 2) : pairs);
    // ELL_S: lane = row x of S adds R to its own diagonal element only - one vector load per slab instead of a scalar
    // load (and an lgkmcnt drain) per stored row
    double dRx = 0.0;
    if (MODE == ELL_S && live) dRx = a.diagR[(long)filt * a.strideR + x];
#pragma unroll UNR
    for (int p = wave; p < p_end; p += NW) {
      const d2* pv = reinterpret_cast<const d2*>(ops) + (long)p * NSLOT;
      unsigned pk[ELL_PIW];
      {
        const uint2* pq = reinterpret_cast<const uint2*>(pidx + p * ELL_PIW);   // wave-uniform address: broadcast reads
#pragma unroll
        for (int u = 0; u < ELL_PIW / 4; ++u) {
          const uint2 q = pq[u];
          pk[4 * u] = q.x & 0xffffu; pk[4 * u + 1] = q.x >> 16; pk[4 * u + 2] = q.y & 0xffffu; pk[4 * u + 3] = q.y >> 16;
        }
      }
      double sv[PWU];
#pragma unroll
      for (int t = 0; t < PWU; ++t) sv[t] = slab((int)pk[t]);
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int t = 0; t < CWU; ++t) { const d2 v = pv[t]; a0 = fma(v[0], cm[t], a0); a1 = fma(v[1], cm[t], a1); }
#pragma unroll
      for (int t = 0; t < PWU; ++t) { const d2 v = pv[CWU + t]; a0 = fma(v[0], sv[t], a0); a1 = fma(v[1], sv[t], a1); }
      if (!live || (XIVO_ELL_ABL == 3 && a0 != 12345.678)) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (XC == 32 && i != half) continue;      // XC = 32: each half-wave stores its own row
        const int m = 2 * p + i;
        const double acc = i ? a1 : a0;
        if (MODE == ELL_HP) out[(long)m * a.ldo] = acc;
        else if (MODE == ELL_S) {
          const double sv_ = acc + (x == m ? dRx : 0.0);
          out[(long)m * a.ldo] = sv_;
          if (XC == 64 && a.diag_out && (x >> 1) == p) a.diag_out[(long)filt * a.strideDiag + 4 * p + 2 * (x & 1) + i] = sv_;
        }
        else if (MODE == ELL_GF) outf[(long)m * a.ldo] = (float)fma(K[(long)m * a.ldk], dR[m], acc);
        else out[(long)m * a.ldo] = fma(K[(long)m * a.ldk], dR[m], acc);
      }
    }
    // the slab is overwritten by the next step. LDS-only barrier: __syncthreads() would also wait for the output stores
    // just issued and for the next slab's prefetch loads to land
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  // Estimator::MHGating on the S just formed (update.cpp:60-96), when this workgroup wrote all of it: the 2 x 2 blocks S_f
  // sit on its diagonal; saves a launch and the pass that would re-fetch them
  if (MODE == ELL_S && a.gate_here) {
    __syncthreads();                      // every store of S by this workgroup is acknowledged before anybody reads it back
    gate_ell_body(a.gate, filt, tile);
    if (more) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // its LDS scratch is the next slab's
  }
  if (!more) break;
  stage_ops(nfilt);                       // (visible to the walk behind the barrier that follows the next park)
  filt = nfilt;
  Src = a.Src + (long)filt * a.strideSrc;
  }
}

}  // namespace

#define CHECK_LAUNCH() return (int)hipGetLastError()

int launch_meas_compress(const double* H, long strideH, int ldh, const double* inn, long strideInn,
                         const double* diagR, long strideR, int M, int N, int Np, int Mp_clear, EllBuffers e,
                         double* inn_out, long strideInnOut, double* R_out, long strideROut, int batch, hipStream_t s,
                         int* host_flags) {
  if (batch <= 0) return 0;
  MeasCompressArgs a;
  a.host_flags = host_flags;
  a.H = H; a.strideH = strideH; a.ldh = ldh; a.inn = inn; a.strideInn = strideInn; a.diagR = diagR; a.strideR = strideR;
  a.M = M; a.N = N; a.Np = Np; a.pairs_clear = Mp_clear / 2; a.ell = e;
  a.inn_out = inn_out; a.strideInnOut = strideInnOut; a.R_out = R_out; a.strideROut = strideROut;
  const int nt = ((a.pairs_clear + 63) / 64) * 64;
  if (nt > 256) return (int)hipErrorInvalidValue;
  a.list_ld = (a.pairs_clear + 3) & ~3;   // only threads that own a pair ever touch the lists: 45 instead of 72 KB at 80 pairs, 3 workgroups per CU
  const size_t lds = (size_t)ELL_W * a.list_ld * (sizeof(d2) + sizeof(int)) + (size_t)(nt / 64 + 1) * Np * sizeof(int);
  // 16-byte loads need an even leading dimension / stride and an aligned base
  const bool aligned = (ldh % 2 == 0) && (strideH % 2 == 0) && ((reinterpret_cast<uintptr_t>(H) & 15u) == 0) && (M % 2 == 0);
  static bool attr_set = false;
  if (!attr_set) {   // dynamic + the kernel's 96 bytes of static LDS must fit the 160 KiB of a CU
    const int cap = 160 * 1024 - 256;
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&meas_compress_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&meas_compress_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e1 != hipSuccess || e2 != hipSuccess) return (int)(e1 != hipSuccess ? e1 : e2);
    attr_set = true;
  }
  if (lds > (size_t)(160 * 1024 - 256)) return (int)hipErrorInvalidValue;
  if (aligned) hipLaunchKernelGGL(meas_compress_kernel<true>, dim3(batch), dim3(nt), lds, s, a);
  else hipLaunchKernelGGL(meas_compress_kernel<false>, dim3(batch), dim3(nt), lds, s, a);
  CHECK_LAUNCH();
}

static size_t meas_compress_lds(int Mp_clear, int Np) {
  const int pairs_clear = Mp_clear / 2;
  const int nt = ((pairs_clear + 63) / 64) * 64;
  const int list_ld = (pairs_clear + 3) & ~3;
  return (size_t)ELL_W * list_ld * (sizeof(d2) + sizeof(int)) + (size_t)(nt / 64 + 1) * Np * sizeof(int);
}
bool meas_compress_fits(int Mp_clear, int Np) {
  return Mp_clear / 2 <= 256 && meas_compress_lds(Mp_clear, Np) <= (size_t)(160 * 1024 - 256);
}
int launch_meas_vectors(const double* inn, long strideInn, const double* diagR, long strideR, int M, int Mp_clear, EllBuffers e,
                        double* inn_out, long strideInnOut, double* R_out, long strideROut, int batch, hipStream_t s) {
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(meas_vectors_kernel, dim3(batch), dim3(256), 0, s, inn, strideInn, diagR, strideR, M, Mp_clear, e, inn_out,
                     strideInnOut, R_out, strideROut);
  CHECK_LAUNCH();
}

int launch_ell_to_dense(EllBuffers e, double* H, long strideH, int ldh, double* HT, long strideHT, int ldht, int Mp,
                        int Np, int batch, hipStream_t s, int Mrows) {
  if (batch <= 0) return 0;
  hipLaunchKernelGGL(ell_to_dense_kernel, dim3(batch), dim3(256), 0, s, e, H, strideH, ldh, HT, strideHT, ldht, Mp, Np,
                     Mrows < 0 ? Mp : Mrows);
  CHECK_LAUNCH();
}
int launch_zero_rows(double* H, long strideH, int ldh, int row0, int nrows, int c0, int c1, int batch, hipStream_t s) {
  if (batch <= 0 || nrows <= 0 || c1 <= c0) return 0;
  hipLaunchKernelGGL(zero_rows_kernel, dim3(batch), dim3(256), 0, s, H, strideH, ldh, row0, nrows, c0, c1);
  CHECK_LAUNCH();
}

template <int MODE, int CWU, int XC, int PWU, bool TALL = true>
static int launch_ell_tile_t(const EllMulArgs& a_in, size_t lds, hipStream_t s) {
  if constexpr (TALL && MODE == ELL_S && ell_tile_pf(MODE, CWU, XC, PWU)) {
    if (a_in.cols <= 256) return launch_ell_tile_t<MODE, CWU, XC, PWU, false>(a_in, lds, s);
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ell_tile_kernel<MODE, CWU, XC, PWU, TALL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  EllMulArgs a = a_in;
  const int xchunks = (a.X + XC - 1) / XC;
  // big batch: one workgroup streams all slabs of its filter (next slab prefetched under the pair walk,
  // coefficients staged once); small batch: one workgroup per slab (latency)
  // (P H^T without the prefetch too: the slot values are staged once per filter instead of once per slab,
  //  3.2 -> 2.9 ms per 16384 filters)
  a.slabs_per_wg = ((ell_tile_pf(MODE, CWU, XC, PWU) || MODE == ELL_HP) && a.batch >= 1024) ? xchunks : 1;
  const int wgs = (xchunks + a.slabs_per_wg - 1) / a.slabs_per_wg;
  int grid = ((a.batch + 7) / 8) * 8 * wgs;
  // persistent form of the prefetching instantiation: one workgroup per CU (they own the LDS one at a time anyway) walks
  // filters b, b + G, ...; G a multiple of 8 keeps a filter on the XCD its index names
  a.persist_stride = 0;
  if (ell_tile_pf(MODE, CWU, XC, PWU) && wgs == 1) {
    static int cus = 0;
    if (!cus) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); if (cus < 8) cus = 8; }
    const int G = cus / 8 * 8;
    if (grid > 2 * G) { a.persist_stride = G; grid = G; }
  }
  if (!(MODE == ELL_S && XC == 64)) a.diag_out = nullptr;
  if (a_in.diag_done) *a_in.diag_done = a.diag_out ? 1 : 0;
  if (!(MODE == ELL_S && wgs == 1)) a.gate_here = 0;      // the gate rides along only when one workgroup forms the whole S of its filter
  if (a_in.gate_done) *a_in.gate_done = a.gate_here;
  hipLaunchKernelGGL((ell_tile_kernel<MODE, CWU, XC, PWU, TALL>), dim3(grid), dim3(ell_tile_threads(MODE, CWU, XC, PWU)), lds, s, a);
  return (int)hipGetLastError();
}
// which form / instantiation the launcher picks: xc = 64 / 32 (slab form) or 0 (gather form)
static void ell_pick(const EllMulArgs& a, int* xc, int* cwu, int* pwu, size_t* lds) {
  *cwu = a.nc_max <= 12 ? 12 : ELL_CW;
  *pwu = (*cwu == 12 && a.pw_max > 0 && a.pw_max <= 9) ? 9 : ELL_PW;
  const size_t pidx = ((size_t)(a.Mp / 2) * ELL_PIW + ELL_CW) * sizeof(unsigned short);   // the private slot indices, 16 bits each, + the common ones
  const size_t ops = (size_t)(a.Mp / 2) * (*cwu + *pwu) * 2 * sizeof(double) + pidx;
  const size_t lds64 = (size_t)a.cols * 64 * sizeof(double) + ops, lds32 = (size_t)a.cols * 32 * sizeof(double) + ops;
  const size_t cap = 160 * 1024;
  *xc = 0; *lds = 0;
  if (lds64 <= cap) { *xc = 64; *lds = lds64; return; }
  if (lds32 <= cap) { *xc = 32; *lds = lds32; }
}

bool ell_uses_slab_form(const EllMulArgs& a) {
  int xc, cwu, pwu; size_t lds;
  ell_pick(a, &xc, &cwu, &pwu, &lds);
  return xc != 0;
}

void ell_kernel_label(int mode, const EllMulArgs& a, char* buf, size_t n) {
  int xc, cwu, pwu; size_t lds;
  ell_pick(a, &xc, &cwu, &pwu, &lds);
  if (xc) snprintf(buf, n, "ell_tile_kernel<%d,%d,%d,%d>", mode, cwu, xc, pwu);
  else snprintf(buf, n, "ell_mul_kernel<%d,%d>", mode, cwu);
}

template <int MODE>
static int launch_ell_tile_m(const EllMulArgs& a, hipStream_t s, bool* done) {
  int xc, cwu, pwu; size_t lds;
  ell_pick(a, &xc, &cwu, &pwu, &lds);
  *done = xc != 0;
  const bool n12 = cwu == 12;
  if (xc == 64 && pwu == 9) return launch_ell_tile_t<MODE, 12, 64, 9>(a, lds, s);
  if (xc == 64) return n12 ? launch_ell_tile_t<MODE, 12, 64, ELL_PW>(a, lds, s) : launch_ell_tile_t<MODE, ELL_CW, 64, ELL_PW>(a, lds, s);
  if (xc == 32 && pwu == 9) return launch_ell_tile_t<MODE, 12, 32, 9>(a, lds, s);
  if (xc == 32) return n12 ? launch_ell_tile_t<MODE, 12, 32, ELL_PW>(a, lds, s) : launch_ell_tile_t<MODE, ELL_CW, 32, ELL_PW>(a, lds, s);
  return 0;
}

int launch_ell_mul(int mode, const EllMulArgs& a_in, hipStream_t s) {
  if (a_in.batch <= 0) return 0;
  if (a_in.diag_done) *a_in.diag_done = 0;      // only the tile form with 64-wide slabs writes the compact diagonal blocks
  EllMulArgs a = a_in;
  {
    {
      bool done = false;
      const int rc = mode == ELL_HP ? launch_ell_tile_m<ELL_HP>(a, s, &done)
                     : (mode == ELL_S ? launch_ell_tile_m<ELL_S>(a, s, &done)
                     : (mode == ELL_GF ? launch_ell_tile_m<ELL_GF>(a, s, &done) : launch_ell_tile_m<ELL_G>(a, s, &done)));
      if (done) return rc;
    }
  }
  if (mode == ELL_S) {   // the gather form reads H P [Mp x Np]; only the tile form can use P H^T
    a.Src = a.SrcAlt; a.strideSrc = a.strideSrcAlt; a.ldsrc = a.ldsrcAlt;
  }
  const int xchunks = (a.X + 255) / 256;
  const int nrb = a.Mp / 16;
  // big batch: one workgroup per (filter, chunk) walks all row blocks; small batch: spread them
  a.rb_per_wg = (long)a.batch * xchunks >= 2048 ? nrb : ((long)a.batch * xchunks >= 512 ? (nrb + 1) / 2 : 1);
  const int rsplit = (nrb + a.rb_per_wg - 1) / a.rb_per_wg;
  const int grid = ((a.batch + 7) / 8) * 8 * rsplit * xchunks;
  // common slots actually in use, rounded to the instantiated widths
#define ELL_LAUNCH(MODE)                                                                                  \
  do {                                                                                                    \
    if (a.nc_max <= 12) hipLaunchKernelGGL((ell_mul_kernel<MODE, 12>), dim3(grid), dim3(256), 0, s, a);   \
    else hipLaunchKernelGGL((ell_mul_kernel<MODE, ELL_CW>), dim3(grid), dim3(256), 0, s, a);              \
  } while (0)
  switch (mode) {
    case ELL_HP: ELL_LAUNCH(ELL_HP); break;
    case ELL_S: ELL_LAUNCH(ELL_S); break;
    default: ELL_LAUNCH(ELL_G); break;
  }
#undef ELL_LAUNCH
  CHECK_LAUNCH();
}

int launch_gate_ell(const GateEllArgs& a, hipStream_t s) {
  if (a.batch <= 0) return 0;
  hipLaunchKernelGGL(gate_ell_kernel, dim3(a.batch), dim3(a.batch < 256 ? 1024 : 256), (size_t)(a.F + 1) * sizeof(double), s, a);
  CHECK_LAUNCH();
}

}  // namespace xivo_hip
