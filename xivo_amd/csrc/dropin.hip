// The two boundary kernels of the one-filter "plumbing" call (capi.hip: xivo_hip_update_joseph_host).
//
// The reference's Estimator::UpdateJosephForm (/root/reference/src/estimator.cpp:1257-1288) works on members that live
// in host memory: P_, H_, inn_, diagR_ in, err_ and P_ out. A drop-in keeps that contract, so every call crosses PCIe
// twice. Both crossings are done by a kernel that addresses host memory directly (the context's page-locked, device-mapped
// staging block) - no DMA descriptors, no extra launches, no host synchronisation before the last kernel has finished:
//   dropin_in_kernel   host P_ (N x N, ld = ldps; its lower triangle, mirrored) -> padded device P  [skipped when P is resident]
//                      host block of compressed rows      -> the filter's row-pair compressed H, inn, diagR, nc / pw / over
//   dropin_out_kernel  device P -> host P_ ; device err_ -> host ; factorisation status + fallback flag -> host
// The dense H_ itself never crosses: the host scans it once while staging it (it has to touch every byte anyway) and
// stages the row-pair compressed rows (ell.h) instead - 43 KB for 320 KB at N = 250 / M = 160.
#include "ekf_kernels.h"
#include "p_unpack_device.h"

namespace xivo_hip {

namespace {

__global__ __launch_bounds__(256) void dropin_in_kernel(DropinInArgs a) {
  __shared__ double tile[kPUnpackTile][kPUnpackTile + 1];
  const int nPb = a.Psrc ? p_unpack_pairs(a.Np) : 0;
  const int blk = blockIdx.x;
  if (blk < nPb) {   // lower triangle of the host matrix authoritative, read once (p_unpack_device.h): half of P_ crosses PCIe
    p_unpack_tile_pair(a.Psrc, a.ldps, a.N, a.P, a.ldp, a.Np, blk, tile);
    return;
  }
  // the staged block, segment by segment (every segment starts 16-byte aligned on both sides)
  const int t = (blk - nPb) * 256 + threadIdx.x, nt = (gridDim.x - nPb) * 256;
  const int n_idx = a.pairs_clear * ELL_W;           // ints
  const int n_val = a.pairs_clear * ELL_W * 2;       // doubles
  const char* src = reinterpret_cast<const char*>(a.block);
  const int* s_idx = reinterpret_cast<const int*>(src + a.off_idx);
  const double* s_val = reinterpret_cast<const double*>(src + a.off_val);
  const double* s_inn = reinterpret_cast<const double*>(src + a.off_inn);
  const double* s_R = reinterpret_cast<const double*>(src + a.off_R);
  const int* s_flags = reinterpret_cast<const int*>(src + a.off_flags);
  for (int i = t; i < n_idx / 4; i += nt) reinterpret_cast<int4*>(a.idx)[i] = reinterpret_cast<const int4*>(s_idx)[i];
  for (int i = t; i < n_val / 2; i += nt) reinterpret_cast<d2*>(a.val)[i] = reinterpret_cast<const d2*>(s_val)[i];
  for (int i = t; i < a.Mpmax; i += nt) { a.inn[i] = s_inn[i]; a.diagR[i] = s_R[i]; }
  if (t == 0) { *a.nc = s_flags[0]; *a.pw = s_flags[1]; *a.over = s_flags[2]; }
}

__global__ __launch_bounds__(256) void dropin_out_kernel(DropinOutArgs a) {
  const int nPb = a.Pdst ? (a.N * a.N + 255) / 256 : 0;
  const int blk = blockIdx.x;
  if (blk < nPb) {
    const int e = blk * 256 + threadIdx.x;
    if (e >= a.N * a.N) return;
    const int i = e % a.N, j = e / a.N;
    a.Pdst[i + (long)j * a.ldpd] = a.P[i + (long)j * a.ldp];
    return;
  }
  for (int i = threadIdx.x; i < a.N; i += 256) a.err_dst[i] = a.err[i];
  if (threadIdx.x == 0) { a.flags_dst[0] = *a.status; a.flags_dst[1] = *a.ldlt_used; }
}

}  // namespace

int launch_dropin_in(const DropinInArgs& a, hipStream_t s) {
  const int nPb = a.Psrc ? p_unpack_pairs(a.Np) : 0;
  hipLaunchKernelGGL(dropin_in_kernel, dim3(nPb + 4), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

int launch_dropin_out(const DropinOutArgs& a, hipStream_t s) {
  const int nPb = a.Pdst ? (a.N * a.N + 255) / 256 : 0;
  hipLaunchKernelGGL(dropin_out_kernel, dim3(nPb + 1), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace xivo_hip
