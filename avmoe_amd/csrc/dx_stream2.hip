// dX[s] = dZx[s] Wt + [dL2 | dsx | 1][s] [T ; 1 ; dm1/N][s] + rs2x X[s]   (moe_backward.cpp, phase 5) in the persistent eight-wave form of
// dpost_pair.hip / tok_pair2.hip, for the tuned bf16 shape (384 channels per group, 128 bottleneck columns, <= 72 latent columns,
// dX overwritten); every other case keeps gemm_stream.hip's twelve-wave kernel.
//
// 64-token tiles of X (the row-scale operand), dZx and dL2x (per frame; the last tile of a frame ragged when 64 does not divide its tokens), and the tile's 64 row scales, go global -> LDS directly (two buffers, the
// next tile in flight during the arithmetic).  Wave w keeps the fragments of ITS three 16-channel tiles of Wt (four K steps) and of the
// frame's T[s] (three K steps, re-gathered when the block moves on to the next frame: contiguous tile ranges) in registers and computes
// the products transposed, so that lane (r, q) ends up with four consecutive channels of token r: the row-scale term is added from the X
// tile in the LDS and the result stored as 8 bytes per lane, 96 contiguous bytes per row and wave.
#include "gemm.h"
#include "common.h"
#include "prof.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>

#ifndef DX2_AUX
#define DX2_AUX 0      // cache policy of the direct loads (common.h::AVMOE_LDS_AUX)
#endif

namespace avmoe {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct DX2Args {
  const char* X; long ldx;             // bf16 [tokens][ldx], group g at column g * 384 (the operand of the row-scale term)
  const char* dZx; long ldz;           // bf16 [tokens][ldz], group g at column g * 128
  const char* dL2; long ldl;           // bf16 [tokens][ldl >= 72]: columns 0 .. K2 - 1 used
  const float* rs;                     // fp32 [tokens]
  const unsigned short* Wt; long ldw, sWg;      // bf16 [g][128][ldw]: row = bottleneck column, column = channel
  const unsigned short* Text; long ldt, sT1;    // bf16 [frame][K2][ldt], group g at column g * 384
  char* dX; long ldc;                  // bf16 [tokens][ldc], group g at column g * 384
  int N, tps, ntiles, K2;              // tokens per frame, 64-token tiles per frame (the last one ragged when 64 does not divide N), tiles in all, rows of T[s]
};

constexpr int BM = 64, NTHR = 512;
constexpr int RBX = 384 * 2 + 16, RBZ = 128 * 2 + 16, RBL = 72 * 2 + 16;
constexpr int OFFZ = BM * RBX, OFFL = OFFZ + BM * RBZ, OFFR = OFFL + BM * RBL, BUF = OFFR + 256;      // 76 pieces of 1 KB + the 64 row scales
constexpr int DX2_LDS = 2 * BUF;

__device__ __forceinline__ unsigned int f2bf(float x) { return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bflo(unsigned int u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bfhi(unsigned int u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

__global__ void __launch_bounds__(NTHR, 1) kk_dx_stream2(const DX2Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int g = blockIdx.y;
  const char* Xb = p.X + (long)g * 384 * 2;
  const char* Zb = p.dZx + (long)g * 128 * 2;
  const char* Lb = p.dL2;
  const long ldx = p.ldx, ldz = p.ldz, ldl = p.ldl;
  const int c0 = 48 * wave;                                 // this wave's channels c0 .. c0 + 47 of the group

  // eight consecutive contraction rows k0 .. k0 + 7 of column n of an MN-major matrix ([row][column]); rows >= kend read as zero
  auto frag_mn = [&](const unsigned short* base, long ld, int n, int k0, int kend) {
    u32x4 v = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned int h = (unsigned int)base[(long)min(k0 + j, kend - 1) * ld + n];      // (every load unconditional: a load under a condition is waited for one by one)
      h = (k0 + j < kend) ? h : 0u;
      v[j >> 1] |= (j & 1) ? (h << 16) : h;
    }
    return __builtin_bit_cast(bf16x8, v);
  };
  bf16x8 bw[3][4], bt[3][3];
  {
    const unsigned short* W = p.Wt + (long)g * p.sWg;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bw[ct][ks] = frag_mn(W, p.ldw, c0 + 16 * ct + r, 32 * ks + 8 * q, 128);
  }
  // the lane's eight columns 32 ks + 8 q .. of the dL2x rows beyond K2 are not data: masks for the three K steps of the second segment
  u32x4 lmask[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int valid = p.K2 - (32 * ks + 8 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) lmask[ks][e] = (2 * e + 1 < valid) ? 0xffffffffu : ((2 * e < valid) ? 0x0000ffffu : 0u);
  }

  auto gload = [&](int buf, int tile) {
    const int fs = tile / p.tps, fj = tile - fs * p.tps;
    const long m0 = (long)fs * p.N + (long)fj * BM;         // first token of the tile
    const int last = p.N - fj * BM - 1;                     // rows beyond the frame's last token re-read it (a ragged last tile; never stored)
    char* dst = smem + buf * BUF + 1024 * wave;
    auto src_x = [&](int j) { const int slot = 64 * j + lane, row = min(slot / 49, last), cc = min(slot % 49, 47); return Xb + ((m0 + row) * ldx + cc * 8) * 2; };
    auto src_z = [&](int j) { const int slot = 64 * j + lane, row = min(slot / 17, last), cc = min(slot % 17, 15); return Zb + ((m0 + row) * ldz + cc * 8) * 2; };
    auto src_l = [&](int j) { const int slot = 64 * j + lane, row = min(slot / 10, last), cc = min(slot % 10, 8); return Lb + ((m0 + row) * ldl + cc * 8) * 2; };
#pragma unroll
    for (int i = 0; i < 6; ++i) __builtin_amdgcn_global_load_lds((gptr_t)src_x(wave + 8 * i), (lptr_t)(dst + 8192 * i), 16, 0, DX2_AUX);
    if (wave == 0) __builtin_amdgcn_global_load_lds((gptr_t)src_x(48), (lptr_t)(dst + 8192 * 6), 16, 0, DX2_AUX);
    else __builtin_amdgcn_global_load_lds((gptr_t)src_z(wave - 1), (lptr_t)(dst + 8192 * 6), 16, 0, DX2_AUX);
    __builtin_amdgcn_global_load_lds((gptr_t)src_z(wave + 7), (lptr_t)(dst + 8192 * 7), 16, 0, DX2_AUX);
    if (wave < 2) __builtin_amdgcn_global_load_lds((gptr_t)src_z(wave + 15), (lptr_t)(dst + 8192 * 8), 16, 0, DX2_AUX);
    else __builtin_amdgcn_global_load_lds((gptr_t)src_l(wave - 2), (lptr_t)(dst + 8192 * 8), 16, 0, DX2_AUX);
    if (wave < 4) __builtin_amdgcn_global_load_lds((gptr_t)src_l(wave + 6), (lptr_t)(dst + 8192 * 9), 16, 0, DX2_AUX);
    else if (wave == 4) __builtin_amdgcn_global_load_lds((gptr_t)(p.rs + m0 + min(lane, last)), (lptr_t)(smem + buf * BUF + OFFR), 4, 0, DX2_AUX);      // the tile's 64 row scales
  };

  // contiguous tile ranges (few frame changes per block)
  int tile = (int)((long)p.ntiles * blockIdx.x / gridDim.x);
  const int t_end = (int)((long)p.ntiles * (blockIdx.x + 1) / gridDim.x);
  if (tile >= t_end) return;
  gload(0, tile);
  int cur_s = -1;
  __syncthreads();
  for (int it = 0; tile < t_end; ++it, ++tile) {
    const char* sX = smem + (it & 1) * BUF;
    const char* sZ = sX + OFFZ;
    const char* sL = sX + OFFL;
    const float* sR = (const float*)(sX + OFFR);
    const int s = tile / p.tps;
    if (s != cur_s) {                                      // this frame's T[s] (block-uniform, a few times per block; ordinary loads: the compiler waits for them here)
      cur_s = s;
      const unsigned short* T = p.Text + (long)s * p.sT1 + (long)g * 384;
#pragma unroll
      for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) bt[ct][ks] = frag_mn(T, p.ldt, c0 + 16 * ct + r, 32 * ks + 8 * q, p.K2);
    }
    if (tile + 1 < t_end) gload((it + 1) & 1, tile + 1);
    const int fj = tile - s * p.tps, valid = p.N - fj * BM;      // rows of this tile inside the frame (>= 64: all of them)
    const long m0 = (long)s * p.N + (long)fj * BM;
#pragma unroll
    for (int mp = 0; mp < 2; ++mp) {                       // two 16-token slabs at a time (independent accumulator chains)
      f32x4 acc[2][3];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) acc[h][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bf16x8 af = *(const bf16x8*)(sZ + (16 * (2 * mp + h) + r) * RBZ + ks * 64 + q * 16);
#pragma unroll
          for (int ct = 0; ct < 3; ++ct) acc[h][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[ct][ks], af, acc[h][ct], 0, 0, 0);
        }
#pragma unroll
      for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          u32x4 v = *(const u32x4*)(sL + (16 * (2 * mp + h) + r) * RBL + ks * 64 + q * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] &= lmask[ks][e];
          const bf16x8 af = __builtin_bit_cast(bf16x8, v);
#pragma unroll
          for (int ct = 0; ct < 3; ++ct) acc[h][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bt[ct][ks], af, acc[h][ct], 0, 0, 0);
        }
#pragma unroll
      for (int h = 0; h < 2; ++h) {                        // lane (r, q): token r of the slab, channels c0 + 16 ct + 4 q .. + 3
        const int row = 16 * (2 * mp + h) + r;
        if (row >= valid) continue;
        const float rs = sR[row];
        char* out = p.dX + ((m0 + row) * p.ldc + (long)g * 384 + c0 + 4 * q) * 2;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
          const u32x2 xv = *(const u32x2*)(sX + row * RBX + (c0 + 16 * ct + 4 * q) * 2);
          const f32x4 a = acc[h][ct];
          *(u32x2*)(out + 32 * ct) = u32x2{f2bf(a[0] + rs * bflo(xv[0])) | (f2bf(a[1] + rs * bfhi(xv[0])) << 16),
                                           f2bf(a[2] + rs * bflo(xv[1])) | (f2bf(a[3] + rs * bfhi(xv[1])) << 16)};
        }
      }
    }
    __syncthreads();                                      // (waits for the direct loads above: the next tile is in place)
  }
}

}  // namespace

// 0 = launched, 1 = shape not served (the caller runs the twelve-wave streaming GEMM), < 0 error
int k_dx_stream2(const void* X, long ldx, const void* dZx, long ldz, const void* dL2, long ldl, int K2, const float* rs, const void* Wt, long ldw, long sWg,
                 const void* Text, long ldt, long sT1, void* dX, long ldc, int S, int N, int G, int Cg, int K1, hipStream_t st) {
  if (Cg != 384 || K1 != 128 || K2 < 1 || K2 > 72 || ldl < 72 || N < BM || S < 1 || ldx % 8 || ldz % 8 || ldl % 8 || ldc % 4 || !rs ||
      ((uintptr_t)X % 16) || ((uintptr_t)dZx % 16) || ((uintptr_t)dL2 % 16) || ((uintptr_t)dX % 8) || ((uintptr_t)rs % 4) || (long)S * N < 2048)
    return 1;
  const int cus = cu_count();                             // (cached per device: common.cpp)
  if (cus <= 0) { set_last_error("dx_stream2: device query"); return ERR_LAUNCH; }
  DX2Args p;
  p.X = (const char*)X; p.ldx = ldx; p.dZx = (const char*)dZx; p.ldz = ldz; p.dL2 = (const char*)dL2; p.ldl = ldl; p.rs = rs;
  p.Wt = (const unsigned short*)Wt; p.ldw = ldw; p.sWg = sWg; p.Text = (const unsigned short*)Text; p.ldt = ldt; p.sT1 = sT1;
  p.dX = (char*)dX; p.ldc = ldc; p.N = N; p.tps = (N + BM - 1) / BM; p.ntiles = S * p.tps; p.K2 = K2;
  const int gx = std::min(std::max(1, cus / G), p.ntiles);
  static LdsAttrOnce attr;
  AVMOE_TRY(attr.ensure((const void*)kk_dx_stream2, DX2_LDS, "dx_stream2"));
  const double ntok = (double)S * N;
  const double bytes = ntok * G * (384.0 * 2 * 2 + 128.0 * 2) + ntok * (ldl * 2.0 + 4.0);
  ProfScope ps("k_dx_stream2", (long)ntok, bytes, 2.0 * ntok * G * 384.0 * (128 + K2), st);
  hipLaunchKernelGGL(kk_dx_stream2, dim3((unsigned)gx, (unsigned)G), dim3(NTHR), DX2_LDS, st, p);
  AVMOE_CHECK_LAUNCH("dx_stream2");
  return OK;
}

}  // namespace avmoe
