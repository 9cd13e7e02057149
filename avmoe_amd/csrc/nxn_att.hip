// AVVP unimodal N x N block (mgn.py:132-139), the softmax of the token-token scores as ONE kernel per chunk of frames:
//
//     att[s] = softmax_rows(X[s] X[s]^T)          X: (frames, N, C) bf16,  att: (frames, N, Np) bf16,  lse: (frames, N) f32
//
// A block owns 128 query rows of one frame and keeps their MFMA fragments in registers; the key tiles (128 tokens x C, 24 - 48 KB)
// stream through the LDS -- twice: sweep 1 accumulates the running (max, sum exp) of every row, sweep 2 recomputes the scores and
// stores exp(score - lse) -- so the scores never exist outside the accumulators and X (0.8 - 0.9 MB per frame) is served by the L2.
// Products are computed transposed (A operand = key fragment, B operand = query fragment; the key rows of a 16-column tile are a
// permutation of the tile's columns) so that lane (r, q) ends up with 32 CONSECUTIVE columns of row r: 64 contiguous bytes per lane
// and slab, stored straight from registers.  With the row log-sum-exp given (the backward of a chunked site) only sweep 2 runs.
// Built for C = 96 and C = 192 (bf16, N a multiple of 128): the stage-0 / stage-1 sites, where N is large; everything else takes
// the engine's softmax epilogues (gemm.hip).
#include "kernels.h"
#include "common.h"
#include "prof.h"
#include <algorithm>

namespace avmoe {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ unsigned pack2(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}

template <int KS>          // C = 32 KS
__global__ void __launch_bounds__(256, 2) kk_nxn_att(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Kt, float* __restrict__ lse_g,
                                                      unsigned short* __restrict__ att, const unsigned short* __restrict__ att_in, int N, int Np, int have_lse) {
  // att_in != nullptr (have_lse set): the softmax BACKWARD instead -- keys = dxr, lse_g = the row dots, output = att_in * (X dxr^T - rowdot)
  constexpr int C = 32 * KS, RB = C * 2 + 16, CPR = C / 8, NLD = 128 * CPR / 256;          // LDS row pitch, 16-byte chunks per key row, loads per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const long f = blockIdx.y;
  const unsigned short* Xf = X + f * (long)N * C;
  const unsigned short* Kf = Kt + f * (long)N * C;          // the key rows (X itself for the softmax)
  const int i0 = blockIdx.x * 128 + 32 * wave;                     // this wave's 32 query rows
  bf16x8 qf[2][KS];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[tm][ks] = *(const bf16x8*)(Xf + (long)(i0 + 16 * tm + r) * C + ks * 32 + 8 * q);
  float mrow[2] = {-INFINITY, -INFINITY}, lrow[2] = {0.f, 0.f}, lse[2];
  if (have_lse) { lse[0] = lse_g[f * N + i0 + r]; lse[1] = lse_g[f * N + i0 + 16 + r]; }
  const int ntile = N / 128;
  u32x4 st[NLD];
  auto gload = [&](int jt) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + 256 * i;
      st[i] = *(const u32x4*)(Kf + (long)(jt * 128 + c / CPR) * C + (c % CPR) * 8);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + 256 * i;
      *(u32x4*)(smem + (c / CPR) * RB + (c % CPR) * 16) = st[i];
    }
  };
  for (int sweep = have_lse ? 1 : 0; sweep < 2; ++sweep) {
    gload(0);
    for (int jt = 0; jt < ntile; ++jt) {
      __syncthreads();                                             // the previous tile's fragments have been read
      lstore();
      __syncthreads();
      if (jt + 1 < ntile) gload(jt + 1);                           // in flight during this tile's products
      f32x4 acc[2][8];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[tm][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          // key row of fragment row r of tile t: column (r >> 2) * 32 + 4 t + (r & 3) of this key tile  ->  lane (r, q) owns columns 32 q + 4 t + e
          const bf16x8 kf = *(const bf16x8*)(smem + ((r >> 2) * 32 + 4 * t + (r & 3)) * RB + ks * 64 + q * 16);
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) acc[tm][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[tm][ks], acc[tm][t], 0, 0, 0);
        }
      if (sweep == 0) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          float mx = mrow[tm];
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, acc[tm][t][e]);
          float s = lrow[tm] * __expf(mrow[tm] - mx);
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += __expf(acc[tm][t][e] - mx);
          mrow[tm] = mx; lrow[tm] = s;
        }
      } else {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const long off = (f * N + i0 + 16 * tm + r) * (long)Np + jt * 128 + 32 * q;
          unsigned short* dst = att + off;
#pragma unroll
          for (int h = 0; h < 4; ++h) {                              // 8 columns = tiles 2 h and 2 h + 1
            const f32x4 a = acc[tm][2 * h], b = acc[tm][2 * h + 1];
            const float l = lse[tm];
            if (att_in) {
              const u32x4 w = *(const u32x4*)(att_in + off + 8 * h);
              auto lo = [](unsigned u) { return __builtin_bit_cast(float, u << 16); };
              auto hi = [](unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); };
              *(u32x4*)(dst + 8 * h) = u32x4{pack2(lo(w[0]) * (a[0] - l), hi(w[0]) * (a[1] - l)), pack2(lo(w[1]) * (a[2] - l), hi(w[1]) * (a[3] - l)),
                                             pack2(lo(w[2]) * (b[0] - l), hi(w[2]) * (b[1] - l)), pack2(lo(w[3]) * (b[2] - l), hi(w[3]) * (b[3] - l))};
            } else {
              *(u32x4*)(dst + 8 * h) = u32x4{pack2(__expf(a[0] - l), __expf(a[1] - l)), pack2(__expf(a[2] - l), __expf(a[3] - l)),
                                             pack2(__expf(b[0] - l), __expf(b[1] - l)), pack2(__expf(b[2] - l), __expf(b[3] - l))};
            }
          }
        }
      }
    }
    if (sweep == 0) {                                                // rows are spread over the four q lanes of each r
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        float m = mrow[tm], l = lrow[tm];
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
          const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
          const float mn = fmaxf(m, m2);
          l = l * __expf(m - mn) + l2 * __expf(m2 - mn); m = mn;
        }
        lse[tm] = m + __logf(l);
        if (q == 0) lse_g[f * N + i0 + 16 * tm + r] = lse[tm];
      }
    }
  }
}

}  // namespace

bool nxn_att_ok(int bf16, int N, int C, int Np) { return bf16 && (C == 96 || C == 192) && N % 128 == 0 && N >= 128 && Np % 8 == 0; }

static int nxn_launch(const char* name, const void* X, const void* K, int frames, int N, int C, int Np, float* lse, void* out, const void* att_in, int have_lse,
                      hipStream_t st) {
  if (!nxn_att_ok(1, N, C, Np)) { set_last_error("nxn_att: shape not served (C = 96 / 192, N a multiple of 128)"); return ERR_UNSUPPORTED; }
  if (frames <= 0) return OK;
  const double bytes = (double)frames * N * ((double)C * 2 * (att_in ? 2 : 1) + (double)Np * 2 * (att_in ? 2 : 1) + 4);
  const double flops = (have_lse ? 1.0 : 2.0) * 2.0 * frames * (double)N * N * C;
  ProfScope ps_(name, (long)frames * N, bytes, flops, st);
  const dim3 grid((unsigned)(N / 128), (unsigned)frames);
  if (C == 96) hipLaunchKernelGGL(kk_nxn_att<3>, grid, dim3(256), 128 * (96 * 2 + 16), st, (const unsigned short*)X, (const unsigned short*)K, lse,
                                  (unsigned short*)out, (const unsigned short*)att_in, N, Np, have_lse);
  else hipLaunchKernelGGL(kk_nxn_att<6>, grid, dim3(256), 128 * (192 * 2 + 16), st, (const unsigned short*)X, (const unsigned short*)K, lse,
                          (unsigned short*)out, (const unsigned short*)att_in, N, Np, have_lse);
  AVMOE_CHECK_LAUNCH("nxn_att");
  return OK;
}
int k_nxn_att(const void* X, int frames, int N, int C, int Np, float* lse, void* att, int have_lse, hipStream_t st) {
  return nxn_launch("k_nxn_att", X, X, frames, N, C, Np, lse, att, nullptr, have_lse, st);
}
// dS = att * (X dxr^T - rowdot): the softmax backward of the same block, the gradient of the scores never stored
int k_nxn_att_bwd(const void* X, const void* dxr, int frames, int N, int C, int Np, const float* rowdot, const void* att, void* dS, hipStream_t st) {
  return nxn_launch("k_nxn_att_bwd", X, dxr, frames, N, C, Np, (float*)rowdot, dS, att, 1, st);
}

}  // namespace avmoe
