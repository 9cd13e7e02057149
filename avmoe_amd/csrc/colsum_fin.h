// Column sums over per-block partial rows fused with a per-column epilogue (shared by fwd_kernels.hip / bwd_kernels.hip).
#pragma once
#include "kernels.h"

namespace avmoe {

// Column sums of TWO adjacent slots followed, in the same launch, by a per-column epilogue fin(col, sum0, sum1) (the finalize kernels
// that only need their own column).  Each slot is summed exactly as kk_colsum_f32 does it (same streams, same order): bit-identical.
template <int CW, int NTHR, class Fin>
__global__ void __launch_bounds__(NTHR) kk_colsum_fin(const float* in, long R, int ncol, long row_stride, long slot_in, Fin fin) {
  constexpr int NS = NTHR / CW;
  __shared__ double red[NS][CW];
  const int c = threadIdx.x % CW, k = threadIdx.x / CW;
  const int col = blockIdx.x * CW + c;
  float sums[2] = {0.f, 0.f};
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const float* p = in + (long)sl * slot_in + col;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (col < ncol) {
      long r = k;
      for (; r + 3L * NS < R; r += 4L * NS) {
        a0 += p[r * row_stride]; a1 += p[(r + NS) * row_stride]; a2 += p[(r + 2L * NS) * row_stride]; a3 += p[(r + 3L * NS) * row_stride];
      }
      for (; r < R; r += NS) a0 += p[r * row_stride];
    }
    red[k][c] = ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    __syncthreads();
    if (k == 0 && col < ncol) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < NS; ++w) s += red[w][c];
      sums[sl] = (float)(s * 1.f);
    }
    __syncthreads();
  }
  if (k == 0 && col < ncol) fin(col, sums[0], sums[1]);
}
template <class Fin>
static int launch_colsum_fin(const float* in, long R, int ncol, long row_stride, long slot_in, const Fin& fin, hipStream_t st) {
  if (R >= 128) hipLaunchKernelGGL((kk_colsum_fin<16, 1024, Fin>), dim3(cdiv(ncol, 16)), dim3(1024), 0, st, in, R, ncol, row_stride, slot_in, fin);
  else hipLaunchKernelGGL((kk_colsum_fin<64, 256, Fin>), dim3(cdiv(ncol, 64)), dim3(256), 0, st, in, R, ncol, row_stride, slot_in, fin);
  AVMOE_CHECK_LAUNCH("colsum_fin");
  return OK;
}

}  // namespace avmoe
