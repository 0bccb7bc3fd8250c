// Host-layout covariance (N x N column-major, leading dimension lds) -> padded device covariance (Np x Np, ldp, zero pad),
// LOWER TRIANGLE AUTHORITATIVE: element (i, j), i < j, is taken from (j, i). Every pipeline here treats P as symmetric
// (DESIGN.md 1b); the reference never re-symmetrises its P_ (src/estimator.cpp:1280-1287), so what a caller hands over
// can differ between the triangles by rounding - or by more after an inconsistent host edit. Reading one triangle makes
// the device state exactly symmetric from the first kernel on and the result independent of which consumer reads which
// half; it also halves what crosses PCIe when the source is host memory (xivo_hip_update_joseph_host).
// One call handles the pair of 32 x 32 tiles (I, J) / (J, I), I >= J, of one filter: 256 threads, tile through LDS so that
// both the source reads and the two destination writes run along columns (contiguous).
#pragma once
#include <hip/hip_runtime.h>

namespace xivo_hip {

constexpr int kPUnpackTile = 32;
__host__ __device__ inline int p_unpack_tiles(int Np) { return (Np + kPUnpackTile - 1) / kPUnpackTile; }
__host__ __device__ inline int p_unpack_pairs(int Np) { const int t = p_unpack_tiles(Np); return t * (t + 1) / 2; }

__device__ __forceinline__ void p_unpack_tile_pair(const double* __restrict__ src, int lds, int N, double* __restrict__ P, int ldp, int Np,
                                                   int pair, double (*tile)[kPUnpackTile + 1]) {
  int I = 0;
  while ((I + 1) * (I + 2) / 2 <= pair) ++I;          // pair -> (I, J), I >= J
  const int J = pair - I * (I + 1) / 2;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int i0 = I * kPUnpackTile, j0 = J * kPUnpackTile;
  for (int c = ty; c < kPUnpackTile; c += 8) {
    const int i = i0 + tx, j = j0 + c;
    double v = 0.0;
    if (i < N && j < N) v = src[i + (long)j * lds];
    tile[c][tx] = v;                                   // tile[column][row]
  }
  __syncthreads();
  for (int c = ty; c < kPUnpackTile; c += 8) {
    const int i = i0 + tx, j = j0 + c;
    if (i < Np && j < Np) {
      // diagonal tile: the lower triangle wins - element (i, j) with i < j is element (j, i) of the source
      const double v = (I == J && tx < c) ? tile[tx][c] : tile[c][tx];
      P[i + (long)j * ldp] = v;
    }
    if (I != J) {                                      // the mirrored tile (J, I): P(j0 + tx, i0 + c) = src(i0 + c, j0 + tx)
      const int r = j0 + tx, q = i0 + c;
      if (r < Np && q < Np) P[r + (long)q * ldp] = tile[tx][c];
    }
  }
}

}  // namespace xivo_hip
