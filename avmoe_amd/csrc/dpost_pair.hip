// dApost = dOut Bpost  AND  dBpost = dOut^T Apost  from ONE pass over dOut (net_trans_v3.py:430-434,485-486 backward).
//
// Both products read the site's largest backward tensor, dOut (tokens x C, bf16): as two kernels (the 9-wave streaming GEMM and the
// engine's split-K token contraction) it crossed the memory interface twice, 1536 B per token each time at the C = 768 sites.  Here a
// persistent block of EIGHT waves (two per SIMD: 256 registers each, one block per CU) streams 64-token tiles of dOut and Apost
// through the LDS (two buffers, the next tile's loads in flight during the arithmetic) and does both products on the tile it holds:
//
//   dApost (64 x 144 per tile, contraction over the group's 384 channels): wave w keeps the fragments of column tile w of Bpost in
//     registers for the whole kernel (B stationary, as in gemm_stream.hip); the ninth column tile -- the 3 E scalar columns -- sits in
//     the LDS (12 KB) and is done by wave mt for the 16-token slab mt, one slab per SIMD;
//   dBpost (384 x 144 per group, contraction over the tokens): 216 accumulator tiles, 27 per wave = 108 registers: wave w owns channel
//     tiles 3 w .. 3 w + 2 against all nine column tiles; both operands are read TRANSPOSED from the token-major tiles
//     (ds_read_tr16_b64: the contraction index is the token), 12 fragment reads for 27 matrix instructions per 32 tokens.
//
// The blocks' partial dBpost go to the split-K slab workspace and are summed in block order by kk_dpair_reduce (no float atomics).
// Tuned instance only: Cg = 384, E * dgp = 128 bottleneck columns + <= 16 scalar columns (KPp = 144); everything else keeps the two kernels.
#include "gemm.h"
#include "common.h"
#include "prof.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>

#ifndef DPAIR_AUX
#define DPAIR_AUX 0      // cache policy of the direct loads (common.h::AVMOE_LDS_AUX): the non-temporal hint measured neutral or worse here (its X is re-read by the next kernel of the chain)
#endif

namespace avmoe {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

struct DPairArgs {
  const char* dOut; long ldo;            // bf16 [tokens][ldo], group g at column g * 384
  const char* Bpost; long ldb, sBg;      // bf16 [g][384][ldb]: row = channel, column = bottleneck / scalar column
  const char* Apost; long lda;           // bf16 [tokens][lda], group g at column g * 144
  char* dAp; long ldc;                   // bf16 [tokens][ldc], group g at column g * 128 (the bottleneck columns)
  float* dApx; long ldx; int XW;         // fp32 [tokens][ldx], group g at column g * XW (the scalar columns)
  float* slabs;                          // [gridDim.x][g][384][KP] fp32
  int ntok, KP, NX, ntiles;              // NX: scalar columns stored (KP - 128 rounded up to 4 as the streaming GEMM did)
};

#ifndef DPAIR_DISSECT
#define DPAIR_DISSECT 0          // development builds: 1 = no dApost phase, 2 = no dBpost phase, 3 = neither (the tile stream alone)
#endif
constexpr int KS = 12, BM = 64, NTHR = 512, RB = KS * 64 + 16, STG = BM * RB, RBA = 144 * 2 + 16, STGA = BM * RBA;
constexpr int DPAIR_LDS = 2 * STG + 2 * STGA + KS * 1024;

__device__ __forceinline__ unsigned int f2bf(float x) { return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)x); }

__device__ __forceinline__ bf16x8 tr_frag(const char* ad, int rb) {      // [k = 8 rows from ad][16 columns] -> lane (r, q): column r, rows 0 .. 7 (ad already offset by lane)
  const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad));
  const s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad + 4 * rb));
  const s16x8 w = {v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
  return __builtin_bit_cast(bf16x8, w);
}

// The transposing LDS reads of the dBpost phase as inline assembly: for the intrinsic the compiler waits for EVERY outstanding direct-to-LDS
// load first (it cannot tell that they go to the other buffer), which left the next tile's loads overlapping the dApost phase only.  The waits
// for these reads are explicit (tr_wait*: the fragments pass through the wait so that their consumers stay behind it).
template <int OFF>
__device__ __forceinline__ void tr_issue(u32x2& d, unsigned addr) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory"); }
template <int TK, int C0>     // three column tiles C0 .. C0 + 2 of the Apost tile, K step TK
__device__ __forceinline__ void tr_issue_b3(u32x2 (&b)[3][2], unsigned lb) {
  tr_issue<TK * 32 * RBA + (C0 + 0) * 32>(b[0][0], lb); tr_issue<TK * 32 * RBA + (C0 + 0) * 32 + 4 * RBA>(b[0][1], lb);
  tr_issue<TK * 32 * RBA + (C0 + 1) * 32>(b[1][0], lb); tr_issue<TK * 32 * RBA + (C0 + 1) * 32 + 4 * RBA>(b[1][1], lb);
  tr_issue<TK * 32 * RBA + (C0 + 2) * 32>(b[2][0], lb); tr_issue<TK * 32 * RBA + (C0 + 2) * 32 + 4 * RBA>(b[2][1], lb);
}
template <int TK>
__device__ __forceinline__ void tr_issue_a3(u32x2 (&a)[3][2], unsigned la) {
  tr_issue<TK * 32 * RB + 0>(a[0][0], la); tr_issue<TK * 32 * RB + 0 + 4 * RB>(a[0][1], la);
  tr_issue<TK * 32 * RB + 32>(a[1][0], la); tr_issue<TK * 32 * RB + 32 + 4 * RB>(a[1][1], la);
  tr_issue<TK * 32 * RB + 64>(a[2][0], la); tr_issue<TK * 32 * RB + 64 + 4 * RB>(a[2][1], la);
}
__device__ __forceinline__ void tr_wait3(u32x2 (&x)[3][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]) :: "memory");
}
__device__ __forceinline__ void tr_wait6(u32x2 (&x)[3][2], u32x2 (&y)[3][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]),
               "+v"(y[0][0]), "+v"(y[0][1]), "+v"(y[1][0]), "+v"(y[1][1]), "+v"(y[2][0]), "+v"(y[2][1]) :: "memory");
}
__device__ __forceinline__ bf16x8 tr_pack(const u32x2 (&f)[2]) { return __builtin_bit_cast(bf16x8, u32x4{f[0][0], f[0][1], f[1][0], f[1][1]}); }

__global__ void __launch_bounds__(NTHR, 1) kk_dpair(const DPairArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sB8 = smem + 2 * STG + 2 * STGA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int g = blockIdx.y;
  const unsigned short* Bb = (const unsigned short*)p.Bpost + (long)g * p.sBg;
  auto frag_mn = [&](int n, int k0) {          // channels k0 .. k0 + 7 of column n of Bpost ([channel][column]); columns >= KP read as zero
    // (every load unconditional -- clamped column, masked afterwards: under `if (n < KP)` each fragment was its own branch and full wait,
    //  fourteen memory round trips in a row in front of the block's first tile)
    u32x4 v = {0u, 0u, 0u, 0u};
    const unsigned short* bp = Bb + (long)k0 * p.ldb + min(n, p.KP - 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned int h = (unsigned int)bp[(long)j * p.ldb];
      h = n < p.KP ? h : 0u;
      v[j >> 1] |= (j & 1) ? (h << 16) : h;
    }
    return v;
  };
  bf16x8 bfr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) bfr[ks] = __builtin_bit_cast(bf16x8, frag_mn(16 * wave + r, 32 * ks + 8 * q));
  for (int ks = wave; ks < KS; ks += 8) *(u32x4*)(sB8 + (ks * 64 + lane) * 16) = frag_mn(128 + r, 32 * ks + 8 * q);

  f32x4 accB[3][9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int c = 0; c < 9; ++c) accB[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const char* Ob = p.dOut + (long)g * 384 * 2;
  const char* Pb = p.Apost + (long)g * 144 * 2;
  // The tiles go global -> LDS directly (global_load_lds_dwordx4: no staging registers -- the kernel sits at its 256): a wave-instruction
  // fills 1 KB of the LDS image IN ORDER, so lane l of piece j loads whatever belongs at 16-byte slot 64 j + l of the padded image
  // (row = slot / 49, chunk = slot % 49; the pad chunk of a row re-reads its neighbour).  49 pieces for dOut, 19 for Apost, dealt to the waves.
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const long ldo = p.ldo, lda = p.lda;
  auto gload = [&](int buf, int tile) {
    const long m0 = (long)tile * BM;
    char* dst = smem + buf * (STG + STGA) + 1024 * wave;   // (one image per buffer: the dOut tile, then the Apost tile -- piece j at 1024 j)
    auto src_o = [&](int j) { const int slot = 64 * j + lane, row = slot / 49, cc = min(slot % 49, 47); return Ob + ((m0 + row) * ldo + cc * 8) * 2; };
    auto src_p = [&](int j) { const int slot = 64 * (j - 49) + lane, row = slot / 19, cc = min(slot % 19, 17); return Pb + ((m0 + row) * lda + cc * 8) * 2; };
    // pieces wave + 8 i: constant LDS offsets from one base (the compiler sees that they do not overlap); pieces 0 .. 48 are dOut's
    // (i < 6 for every wave, i = 6 for wave 0), the rest Apost's -- no per-piece select between the two sources (a select of loaded
    // strides made the compiler wait for every load in flight in front of the next piece)
#pragma unroll
    for (int i = 0; i < 6; ++i) __builtin_amdgcn_global_load_lds((gptr_t)src_o(wave + 8 * i), (lptr_t)(dst + 8192 * i), 16, 0, DPAIR_AUX);
    if (wave == 0) __builtin_amdgcn_global_load_lds((gptr_t)src_o(48), (lptr_t)(dst + 8192 * 6), 16, 0, DPAIR_AUX);
    else __builtin_amdgcn_global_load_lds((gptr_t)src_p(wave + 48), (lptr_t)(dst + 8192 * 6), 16, 0, DPAIR_AUX);
    __builtin_amdgcn_global_load_lds((gptr_t)src_p(wave + 56), (lptr_t)(dst + 8192 * 7), 16, 0, DPAIR_AUX);
    if (wave < 4) __builtin_amdgcn_global_load_lds((gptr_t)src_p(wave + 64), (lptr_t)(dst + 8192 * 8), 16, 0, DPAIR_AUX);
  };

  int tile = blockIdx.x;
  const int step = gridDim.x;
  if (tile < p.ntiles) gload(0, tile);
  __syncthreads();
  for (int it = 0; tile < p.ntiles; ++it, tile += step) {
    const char* sA = smem + (it & 1) * (STG + STGA);
    const char* sP = sA + STG;
    const int nxt = tile + step, m0 = tile * BM;
    if (nxt < p.ntiles) gload((it + 1) & 1, nxt);         // (the other buffer: its readers passed the barrier that ended the previous iteration)
    // ---- dApost: the tile's 64 rows against this wave's column tile (and the scalar tile for slab mt == wave) ----
#ifndef DPAIR_CH
#define DPAIR_CH 2               // 16-token slabs done together (independent accumulator chains; 4: the same time, 8 more registers)
#endif
#pragma unroll
    for (int mp = 0; mp < ((DPAIR_DISSECT & 1) ? 0 : 4 / DPAIR_CH); ++mp) {
      f32x4 acc[DPAIR_CH];
#pragma unroll
      for (int h = 0; h < DPAIR_CH; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int h = 0; h < DPAIR_CH; ++h) {
          const bf16x8 af = *(const bf16x8*)(sA + (16 * (DPAIR_CH * mp + h) + r) * RB + ks * 64 + q * 16);
          acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks], af, acc[h], 0, 0, 0);
        }
#pragma unroll
      for (int h = 0; h < DPAIR_CH; ++h) {                 // lane (r, q): token r of the slab, columns 4 q .. 4 q + 3 of the tile
        const long m = m0 + 16 * (DPAIR_CH * mp + h) + r;
        *(u32x2*)(p.dAp + (m * p.ldc + g * 128 + 16 * wave + 4 * q) * 2) = u32x2{f2bf(acc[h][0]) | (f2bf(acc[h][1]) << 16), f2bf(acc[h][2]) | (f2bf(acc[h][3]) << 16)};
      }
    }
    if ((DPAIR_DISSECT & 1) == 0 && wave < 4) {            // the scalar columns of slab `wave` (one slab per SIMD), two chains over the even / odd K steps
      f32x4 a8[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 af = *(const bf16x8*)(sA + (16 * wave + r) * RB + ks * 64 + q * 16);
        a8[ks & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(sB8 + (ks * 64 + lane) * 16), af, a8[ks & 1], 0, 0, 0);
      }
      if (4 * q < p.NX) *(f32x4*)(p.dApx + (long)(m0 + 16 * wave + r) * p.ldx + g * p.XW + 4 * q) = a8[0] + a8[1];
    }
    // ---- dBpost += dOut_tile^T Apost_tile : channel tiles 3 wave .. + 2 x nine column tiles, 32 tokens per step ----
    if constexpr ((DPAIR_DISSECT & 2) == 0) {
      const unsigned l0 = (unsigned)(size_t)(lptr_t)sA;
      const unsigned la = l0 + (8 * q + (r >> 2)) * RB + (3 * wave * 16 + 4 * (r & 3)) * 2;
      const unsigned lb = l0 + STG + (8 * q + (r >> 2)) * RBA + (4 * (r & 3)) * 2;
      u32x2 fa[3][2], fb0[3][2], fb1[3][2];
      auto mm = [&](int c0, const u32x2 (&fb)[3][2]) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int i = 0; i < 3; ++i) accB[i][c0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pack(fa[i]), tr_pack(fb[c]), accB[i][c0 + c], 0, 0, 0);
      };
      // (the reads of the next three column tiles are in flight during the nine matrix instructions of the current three)
      tr_issue_a3<0>(fa, la); tr_issue_b3<0, 0>(fb0, lb); tr_wait6(fa, fb0);
      tr_issue_b3<0, 3>(fb1, lb); mm(0, fb0); tr_wait3(fb1);
      tr_issue_b3<0, 6>(fb0, lb); mm(3, fb1); tr_wait3(fb0);
      mm(6, fb0);
      tr_issue_a3<1>(fa, la); tr_issue_b3<1, 0>(fb0, lb); tr_wait6(fa, fb0);
      tr_issue_b3<1, 3>(fb1, lb); mm(0, fb0); tr_wait3(fb1);
      tr_issue_b3<1, 6>(fb0, lb); mm(3, fb1); tr_wait3(fb0);
      mm(6, fb0);
    }
    __syncthreads();                                      // (waits for the direct loads above: the next tile is in place)
  }
  // lane (r, q): dBpost[channel 16 ct + 4 q + e][column 16 c + r]
  float* sl = p.slabs + ((long)blockIdx.x * gridDim.y + g) * 384 * (long)p.KP;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int col = 16 * c + r;
      if (col < p.KP) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sl[(long)(16 * (3 * wave + i) + 4 * q + e) * p.KP + col] = accB[i][c][e];
      }
    }
}

// dBp[g][ch][col] = sum over the blocks' slabs, in a fixed order: a wave covers 16 consecutive 4-element vectors, four lanes per vector
// (lane = part * 16 + vector: 256 contiguous bytes per part and slab; part p sums slabs p, p + 4, ...), fixed xor tree over the parts
__global__ void __launch_bounds__(256) kk_dpair_reduce(const float* __restrict__ slabs, int nslab, int G, int KP, int KPp, float* __restrict__ out) {
  const long per = (long)G * 384 * KP, nvec = per / 4;                    // (KP % 4 == 0: the launcher checks)
  const int lane = threadIdx.x & 63, part = lane >> 4;
  const long v = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (lane & 15);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (v < nvec) {
    const float* sl = slabs + v * 4;
    int b = part;
    for (; b + 12 < nslab; b += 16) {
      const f32x4 a0 = *(const f32x4*)(sl + (long)b * per), a1 = *(const f32x4*)(sl + (long)(b + 4) * per);
      const f32x4 a2 = *(const f32x4*)(sl + (long)(b + 8) * per), a3 = *(const f32x4*)(sl + (long)(b + 12) * per);
      s += (a0 + a1) + (a2 + a3);
    }
    for (; b < nslab; b += 4) s += *(const f32x4*)(sl + (long)b * per);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { s[e] += __shfl_xor(s[e], 16, 64); s[e] += __shfl_xor(s[e], 32, 64); }
  if (v < nvec && part == 0) {
    const long el = v * 4, row = el / KP; const int col = (int)(el - row * KP);
    *(f32x4*)(out + row * KPp + col) = s;
  }
}

}  // namespace

// 0 = launched, 1 = shape not served (the caller runs the two kernels), < 0 error
int k_dpost_pair(const void* dOut, long ldo, const void* Bpost, long ldb, long sBg, const void* Apost, long lda, void* dAp, long ldc, float* dApx, long ldx, int XW,
                 float* dBp, int ntok, int G, int Cg, int nmain, int KP, int KPp, float* slabs, size_t slab_cap, hipStream_t st) {
  if (Cg != 384 || nmain != 128 || KPp != 144 || KP <= 128 || KP > 144 || KP % 4 || ((uintptr_t)dBp % 16) || XW < 16 || ntok % BM || ldo % 8 || lda % 8 || !slabs ||
      ((uintptr_t)dOut % 16) || ((uintptr_t)Apost % 16) || ((uintptr_t)dAp % 8) || ((uintptr_t)dApx % 16) || ldx % 4 || (G * XW) % 4)
    return 1;
  // small sites: the two kernels (the 128-slab reduction alone costs 11 us; measured at 20 480 tokens: 44 us against 37); avmoe_test_hooks bit 2: test hook
  if (ntok < 32768 && !(ntok >= 4096 && (test_hook_mask() & HOOK_DPAIR_FORCE))) return 1;
  const int cus = cu_count();                             // (cached per device: common.cpp)
  if (cus <= 0) { set_last_error("dpost_pair: device query"); return ERR_LAUNCH; }
  const int ntiles = cdiv(ntok, BM);
  const size_t per = (size_t)G * 384 * KP;
  int gx = std::max(1, cus / G);
  gx = std::min<long>(std::min(gx, ntiles), (long)(slab_cap / per));
  if (gx < 1) return 1;
  DPairArgs p;
  p.dOut = (const char*)dOut; p.ldo = ldo; p.Bpost = (const char*)Bpost; p.ldb = ldb; p.sBg = sBg; p.Apost = (const char*)Apost; p.lda = lda;
  p.dAp = (char*)dAp; p.ldc = ldc; p.dApx = dApx; p.ldx = ldx; p.XW = XW; p.slabs = slabs; p.ntok = ntok; p.KP = KP;
  p.NX = (KP % 4 == 0 ? KP : KPp) - 128; p.ntiles = ntiles;
  static LdsAttrOnce attr;
  AVMOE_TRY(attr.ensure((const void*)kk_dpair, DPAIR_LDS, "dpost_pair"));
  {
    const double bytes = (double)ntok * G * (384.0 * 2 + 144.0 * 2 + 128.0 * 2 + 16.0 * 4) + 2.0 * gx * per * 4.0;
    ProfScope ps("k_dpost_pair", (long)ntok, bytes, 2.0 * 2.0 * ntok * (double)G * 384 * 144, st);
    hipLaunchKernelGGL(kk_dpair, dim3((unsigned)gx, (unsigned)G), dim3(NTHR), DPAIR_LDS, st, p);
    AVMOE_CHECK_LAUNCH("dpost_pair");
  }
  {
    ProfScope ps("k_dpair_reduce", (long)per, (double)per * 4.0 * (gx + 1), 0.0, st);
    hipLaunchKernelGGL(kk_dpair_reduce, dim3((unsigned)((per / 4 + 63) / 64)), dim3(256), 0, st, slabs, gx, G, KP, KPp, dBp);
    AVMOE_CHECK_LAUNCH("dpair_reduce");
  }
  return OK;
}

}  // namespace avmoe
