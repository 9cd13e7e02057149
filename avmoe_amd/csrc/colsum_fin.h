// Column sums over per-block partial rows fused with a per-column epilogue (shared by fwd_kernels.hip / bwd_kernels.hip).
#pragma once
#include "kernels.h"
#include "device_utils.h"

namespace avmoe {

// sum over the NTHR threads of a block in a fixed order (waves first, then the waves' sums in wave order); valid in every thread
template <int NTHR>
__device__ __forceinline__ float block_sum_fixed(float v, float* scratch /* NTHR / 64 floats */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NTHR / 64; ++w) t += scratch[w];
  return t;
}

// Column sums of TWO adjacent slots followed, in the same launch, by a per-column epilogue fin(col, sum0, sum1) (the finalize kernels
// that only need their own column).  Each slot is summed exactly as kk_colsum_f32 does it (same streams, same order): bit-identical.
// Blocks past the `ncolblk` column blocks run fin.extra<NTHR>(block - ncolblk, number of extra blocks): the finalize work that
// does not hang on a column (per-expert scalar sums ..), so that it needs no launch of its own.
template <int CW, int NTHR, class Fin>
__global__ void __launch_bounds__(NTHR) kk_colsum_fin(const float* in, long R, int ncol, long row_stride, long slot_in, Fin fin, int ncolblk) {
  constexpr int NS = NTHR / CW;
  __shared__ double red[NS][CW];
  if ((int)blockIdx.x >= ncolblk) { fin.template extra<NTHR>((int)blockIdx.x - ncolblk, (int)gridDim.x - ncolblk); return; }
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
struct NoExtra { template <int NTHR> __device__ void extra(int, int) const {} };
template <class Fin>
static int launch_colsum_fin(const float* in, long R, int ncol, long row_stride, long slot_in, const Fin& fin, hipStream_t st, int nextra = 0) {
  if (R >= 128) hipLaunchKernelGGL((kk_colsum_fin<16, 1024, Fin>), dim3(cdiv(ncol, 16) + nextra), dim3(1024), 0, st, in, R, ncol, row_stride, slot_in, fin, cdiv(ncol, 16));
  else hipLaunchKernelGGL((kk_colsum_fin<64, 256, Fin>), dim3(cdiv(ncol, 64) + nextra), dim3(256), 0, st, in, R, ncol, row_stride, slot_in, fin, cdiv(ncol, 64));
  AVMOE_CHECK_LAUNCH("colsum_fin");
  return OK;
}

}  // namespace avmoe
