// Second-moment (Gram) matrices of the bottleneck activations z' over ALL tokens, for the shape of tile_fast.hip
// (bottleneck 64 in 2 groups, 4 experts, bf16 activations):
//
//   G[cb][j][l] = scale * sum_t w[e][t] * Zp[t][col(cb) + j] * Zp[t][col(cb) + l]      cb = group * E + e,  32 x 32 each
//
//   forward  (net_trans_v3.py:397-400 folded, algebra_ref.py "Szz"):  w = 1, scale = 1 / NT      -> BatchNorm-2 statistics
//   backward (algebra_ref.py "dG"):                                    w = dSoo (LayerNorm-post variance gradient per token)
//
// One streaming pass over Zp (each row read once, whole 512-byte rows), instead of the 8 batched 32 x 32 token
// contractions of the generic engine that each pull their 64-byte column slice out of every row.  Token tiles go
// global -> registers -> LDS (two buffers); a wave owns two (group, expert) pairs and reads its operand fragments with the
// transposing LDS read ds_read_b64_tr_b16 (tokens are the contraction index); the same fragment serves as A and as B
// operand of v_mfma_f32_16x16x32_bf16 (the weighted form scales the A copy).  Per-block partial sums are written to
// `part` and summed over the blocks by the deterministic column-sum kernel (no float atomics).
#include "kernels.h"
#include "device_utils.h"
#include "prof.h"
#include <algorithm>

namespace avmoe {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int GG = 2, GD = 32;                                  // groups, bottleneck per group
constexpr int GT = 64;                                          // tokens per tile

// GE experts (4: two (group, expert) pairs per wave; 2: one pair per wave)
// XF: the input is z (before BatchNorm-1); z' = act(sc z + sh), rounded to bf16, is formed on the way from global memory to LDS
// (bn1: the [mean | rstd | scale | shift] rows of the plan, relu_mask: bit e = expert e has the ReLU) -- the forward then needs
// neither the separate MID pass nor a stored copy of z'; with `colpart` the per-block column sums of z' are written as well
// (BatchNorm-2's first moments).
template <bool WEIGHTED, int GE, bool XF>
__global__ void __launch_bounds__(256) kg_gram64(const unsigned short* __restrict__ Zp, const float* __restrict__ w /* [E][NT] */,
                                                 float* __restrict__ part, long NT, int tiles_per_blk, const float* __restrict__ bn1,
                                                 unsigned relu_mask, float* __restrict__ colpart) {
  constexpr int GDZ = GE * GG * GD;                             // row width (256 / 128)
  constexpr int G_ROWB = GDZ * 2 + 16;                          // LDS bytes per token row
  constexpr int G_TILE = GT * G_ROWB;
  constexpr int PPW = GG * GE / 4;                              // (group, expert) pairs per wave
  constexpr int CPR = GDZ / 8;                                  // 16-byte chunks per row
  constexpr int NLD = GT * CPR / 256;                           // chunks per thread and tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_w = (float*)(smem + 2 * G_TILE);                     // [2 buffers][E][GT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const long tile0 = (long)blockIdx.x * tiles_per_blk;
  const long ntiles = (NT + GT - 1) / GT;
  const long tile1 = min(ntiles, tile0 + (long)tiles_per_blk);

  f32x4 acc[PPW][2][2];                                          // [pair of this wave][row tile][column tile]
#pragma unroll
  for (int p = 0; p < PPW; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[p][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[NLD];                                                 // 64 rows x CPR chunks of 16 B / 256 threads
  float rw = 0.f;
  // XF: this thread always serves the same 8 columns (256 % CPR == 0): their scale / shift / activation stay in registers
  float xsc[XF ? 8 : 1], xsh[XF ? 8 : 1], csum[XF ? 8 : 1];
  bool xrelu = false;
  long ld_t0 = 0;                                                // first token of the tile that sits in ra[]
  if constexpr (XF) {
    const int col0 = (tid % CPR) * 8;
    xrelu = (relu_mask >> ((col0 / GD) % GE)) & 1u;
#pragma unroll
    for (int j = 0; j < 8; ++j) { xsc[j] = bn1[2 * GDZ + col0 + j]; xsh[j] = bn1[3 * GDZ + col0 + j]; csum[j] = 0.f; }
  }
  auto gload = [&](long tile) {
    const long t0 = tile * GT;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + 256 * i, row = c / CPR, cc = c % CPR;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (t0 + row < NT) v = *(const u32x4*)(Zp + (t0 + row) * GDZ + cc * 8);
      ra[i] = v;
    }
    ld_t0 = t0;
    if constexpr (WEIGHTED) {
      const int e = tid >> 6, tl = tid & 63;                     // experts x 64 tokens
      rw = (e < GE && t0 + tl < NT) ? w[(long)e * NT + t0 + tl] : 0.f;
    }
  };
  auto lstore = [&](int buf) {
    char* s = smem + buf * G_TILE;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + 256 * i;
      u32x4 v = ra[i];
      if constexpr (XF) {                                        // z -> z' here, AFTER the MFMAs of the previous tile: the loads stay in flight behind them
        if (ld_t0 + c / CPR < NT) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float y0 = bf2f((unsigned short)(v[j] & 0xffffu)) * xsc[2 * j] + xsh[2 * j];
            float y1 = bf2f((unsigned short)(v[j] >> 16)) * xsc[2 * j + 1] + xsh[2 * j + 1];
            if (xrelu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
            const unsigned short b0 = f2bf(y0), b1 = f2bf(y1);
            csum[2 * j] += bf2f(b0); csum[2 * j + 1] += bf2f(b1);
            v[j] = (unsigned)b0 | ((unsigned)b1 << 16);
          }
        }
      }
      *(u32x4*)(s + (c / CPR) * G_ROWB + (c % CPR) * 16) = v;
    }
    if constexpr (WEIGHTED) { if (tid < GE * GT) s_w[buf * GE * GT + tid] = rw; }
  };
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

  if (tile0 < tile1) { gload(tile0); lstore(0); }
  __syncthreads();
  for (long tile = tile0; tile < tile1; ++tile) {
    const int buf = (int)((tile - tile0) & 1);
    if (tile + 1 < tile1) gload(tile + 1);
    const char* s = smem + buf * G_TILE;
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
      const int cb = PPW * wave + p, gi = cb / GE, e = cb % GE;
      const int col0 = gi * (GE * GD) + e * GD;
#pragma unroll
      for (int ks = 0; ks < GT / 32; ++ks) {
        bf16x8 f[2], fa[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {                        // fragment of columns col0 + 16 ct .., tokens 32 ks + 8 q .. + 7
          const char* ad = s + (ks * 32 + 8 * q + (r >> 2)) * G_ROWB + (col0 + 16 * ct + 4 * (r & 3)) * 2;
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad));
          const s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad + 4 * G_ROWB));
          const s16x8 v = {v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
          f[ct] = __builtin_bit_cast(bf16x8, v);
          fa[ct] = f[ct];
          if constexpr (WEIGHTED) {                              // element i of the fragment <-> token 32 ks + 8 q + i
            const float* wp = s_w + buf * GE * GT + e * GT + ks * 32 + 8 * q;
            s16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (short)f2bf(bf2f((unsigned short)v[i]) * wp[i]);
            fa[ct] = __builtin_bit_cast(bf16x8, o);
          }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) acc[p][it][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[it], f[jt], acc[p][it][jt], 0, 0, 0);
      }
    }
    if (tile + 1 < tile1) lstore(buf ^ 1);
    __syncthreads();
  }
  // C layout: lane (r, q) holds rows 4 q + x, column r of each 16 x 16 tile
  float* out = part + (long)blockIdx.x * (GG * GE * GD * GD);
#pragma unroll
  for (int p = 0; p < PPW; ++p) {
    const int cb = PPW * wave + p;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int x = 0; x < 4; ++x) out[(long)cb * GD * GD + (16 * it + 4 * q + x) * GD + 16 * jt + r] = acc[p][it][jt][x];
  }
  if constexpr (XF) {
    if (colpart) {                                               // column sums of z' of this block's tokens: 256 / CPR threads per column chunk
      __syncthreads();
      float* sc = (float*)smem;                                  // [256 / CPR][GDZ]
#pragma unroll
      for (int j = 0; j < 8; ++j) sc[(tid / CPR) * GDZ + (tid % CPR) * 8 + j] = csum[j];
      __syncthreads();
      for (int col = tid; col < GDZ; col += 256) {
        float v = 0.f;
        for (int k = 0; k < 256 / CPR; ++k) v += sc[k * GDZ + col];
        colpart[(long)blockIdx.x * GDZ + col] = v;
      }
    }
  }
}

}  // namespace

// G (g*E matrices of dgp x dgp, the layout of Szz / dGq) from Zp (NT, DZ) bf16; w = nullptr: unweighted.  Only for the
// register-resident shape (tile_fast_ok) in bf16.  With `bn1` the input is z and z' = act(BN1(z)) is formed on the fly (XF above);
// `mz` (DZ floats, needs `colpart`: GRAM_BLOCKS * DZ floats) then receives the column means of z'.
int k_gram64(const Plan& pl, const void* Zp, const float* w, float scale, float* part, float* out, hipStream_t st, const float* bn1,
             float* colpart, float* mz) {
  const Dims& d = pl.d;
  ProfScope ps_("k_gram64", (long)d.NT, (double)d.NT * (d.DZ * 2.0 + (w ? 4.0 * d.E : 0.0)), 2.0 * d.NT * (double)d.g * d.E * d.dgp * d.dgp, st);
  if (!tile_fast_shape(d) || !d.bf16 || (d.E != 4 && d.E != 2)) { set_last_error("gram64: shape not covered"); return ERR_UNSUPPORTED; }
  const long ntiles = cdiv((long)d.NT, GT);
  const int nblk = (int)std::min<long>(GRAM_BLOCKS, ntiles);
  const int tpb = (int)cdiv(ntiles, (long)nblk);
  const size_t sh = 2 * (size_t)GT * (d.DZ * 2 + 16) + 2 * (size_t)d.E * GT * sizeof(float);
  const void* kerns[8] = {(const void*)kg_gram64<false, 4, false>, (const void*)kg_gram64<true, 4, false>, (const void*)kg_gram64<false, 2, false>,
                          (const void*)kg_gram64<true, 2, false>, (const void*)kg_gram64<false, 4, true>, (const void*)kg_gram64<true, 4, true>,
                          (const void*)kg_gram64<false, 2, true>, (const void*)kg_gram64<true, 2, true>};
  static bool attr = false;
  if (!attr) {
    for (const void* k : kerns)
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) {
        set_last_error("gram64: LDS attribute"); return ERR_LAUNCH;
      }
    attr = true;
  }
  const int used = (int)cdiv(ntiles, (long)tpb);
  const unsigned short* zp = (const unsigned short*)Zp;
  unsigned relu_mask = 0;
  for (int e = 0; e < d.E; ++e) relu_mask |= d.relu_of_e[e] ? (1u << e) : 0u;
  if (mz && !colpart) { set_last_error("gram64: column means need the per-block workspace"); return ERR_BAD_ARG; }
  // Dims::excl (shared_gpu): 80 KB per block -- two blocks (the residency the kernel has anyway at 4 experts) fill the CU, no block of another
  // stream's kernel beside this matrix-pipe kernel (tile_fast.hip has the reason)
  const size_t shx = d.excl ? std::max(sh, (size_t)80 * 1024) : sh;
#define GRAM_LAUNCH(WT, GE_, XF_) hipLaunchKernelGGL((kg_gram64<WT, GE_, XF_>), dim3(used), dim3(256), shx, st, zp, w, part, (long)d.NT, tpb, bn1, relu_mask, mz ? colpart : nullptr)
  if (d.E == 4) {
    if (bn1) { if (w) GRAM_LAUNCH(true, 4, true); else GRAM_LAUNCH(false, 4, true); }
    else { if (w) GRAM_LAUNCH(true, 4, false); else GRAM_LAUNCH(false, 4, false); }
  } else {
    if (bn1) { if (w) GRAM_LAUNCH(true, 2, true); else GRAM_LAUNCH(false, 2, true); }
    else { if (w) GRAM_LAUNCH(true, 2, false); else GRAM_LAUNCH(false, 2, false); }
  }
#undef GRAM_LAUNCH
  AVMOE_CHECK_LAUNCH("gram64");
  const int ncol = d.g * d.E * d.dgp * d.dgp;
  if (mz) return k_colsum2_f32(colpart, used, d.DZ, d.DZ, mz, 1.f / (float)d.NT, part, used, ncol, ncol, out, scale, st);
  return k_colsum_f32(part, used, ncol, ncol, 1, 0, out, 0, scale, st);
}

}  // namespace avmoe
