// The four per-frame products of the hop-1 (latent-token) chain against the OTHER modality's tokens Y (S, M, Cy) -- net_trans_v3.py:379-383
// with the remap of :469-471 folded in -- as streaming kernels (round 5; the tiled engine keeps every other shape):
//
//   forward    R[s]   = Q Y[s]^T              (64 x M)     contraction over the channels      k_hop1_yk
//              V[s]   = [Bm ; wbar][s] Y[s]   (65 x Cy)    contraction over the frame's tokens k_hop1_yt (per frame)
//   backward   dBm[s] = dV[s] Y[s]^T          (65 x M)     contraction over the channels      k_hop1_yk
//              dQ     = sum_s dR[s] Y[s]      (64 x Cy)    contraction over ALL tokens        k_hop1_yt (sum) + kk_hop1_sum
//
// The small operand (<= 80 latent rows) is what the engine's 64 x 64 / 128 x 128 tiles re-read from the L2 for every token tile of Y
// (as many bytes as Y itself) with two K stages in flight; here it is stationary (yk: MFMA fragments in registers, re-gathered at a
// frame change) or streams beside Y in its own small tile (yt), and Y goes global -> LDS directly in whole token rows (the recipe of
// dpost_pair.hip / tok_pair2.hip / dx_stream2.hip: persistent blocks, one per CU, no staging registers, the only full wait of the loop in
// front of its barrier).
#include "gemm.h"
#include "common.h"
#include "prof.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <type_traits>

namespace avmoe {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ unsigned int f2bf(float x) { return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)x); }

// ======================================================================================================================================
// yk:  C[s][row][tok] = sum_c A[(s)][row][c] Y[s][tok][c]       (rows <= 16 NRT, Cy = 32 KS channels, M a multiple of 4)
// 32-token tiles of Y (all channels) in THREE LDS buffers: a round trip to HBM under load (~3.5 us) is several times the arithmetic of a
// tile, so one tile in flight per CU (two buffers: 3.4 TB/s measured) leaves the kernel waiting for its own loads; with three, two tiles are
// in flight while one is multiplied.  That takes counted waits: the loop's barrier is a bare s_barrier behind `s_waitcnt vmcnt(n)` with
// n = this wave's direct loads of the NEXT tile + its stores of the two previous iterations (the in-order counter: everything older -- the
// tile about to be read -- has landed; __syncthreads would drain the queue).  Every wave issues exactly one store instruction per tile
// (lanes without a valid element store to a dump word) so that n is exact.
// The LDS image has no pad: 16-byte chunk c of token row t sits at chunk position c ^ (t & 15) (the lanes of a direct load fetch whatever
// belongs at their slot), which keeps the 16 rows of a fragment read on distinct banks and the tile a whole number of 1 KB pieces per wave.
// Wave (rt, th): row tile rt of A as KS stationary fragments, the 16-token half th of the tile; the product is computed transposed
// (tokens x rows), so that a lane ends up with four consecutive tokens of one row: one 8- / 16-byte store.
// ======================================================================================================================================
struct YKArgs {
  const char* Y; long ldy;                         // bf16 [S * M][ldy]
  const unsigned short* A; long lda, sA1;          // bf16 [(frame)][rows][lda] ; sA1 = 0: shared by the frames
  char* C; long ldc, sC1; int c_bf16;              // [frame][row][ldc] bf16 or fp32
  char* dump;                                      // >= 16 writable bytes nobody reads
  int M, tpf, ntiles, rows;                        // tokens per frame, 32-token tiles per frame (the last one ragged), tiles in all
};

constexpr int YK_BT = 32, YK_NBUF = 3;
// cache policy of the direct loads (common.h::AVMOE_LDS_AUX): Y is streamed non-temporally by all three kernels.  Alone on the GPU the
// hint moves time between them (yk 128 -> 109 us and the all-token sum 112 -> 98 us, but the per-frame kernel that re-reads Y behind yk
// 126 -> 155 us: it had been finding part of Y in the Infinity Cache); what counts is the two-stream step, where the other site's kernels
// keep what the Y streams no longer evict: 4.621 -> 4.570 ms (four interleaved repetitions of the default bench command per setting).
#ifndef YK_AUX
#define YK_AUX AVMOE_LDS_AUX
#endif
#ifndef YTS_AUX
#define YTS_AUX AVMOE_LDS_AUX
#endif
#ifndef YTF_AUX
#define YTF_AUX AVMOE_LDS_AUX
#endif
#ifndef HOP1_DISSECT
#define HOP1_DISSECT 0      // development builds (timing only): bit 0 = no matrix phase, bit 1 = no store, bit 2 = no direct loads, bit 3 = tiles dealt round-robin
#endif

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int KS, int NRT>
__global__ void __launch_bounds__(128 * NRT, 1) kk_hop1_yk(const YKArgs p) {
  constexpr int NW = 2 * NRT, CH = 4 * KS, RB = 16 * CH, NP = YK_BT * CH / 64, BUF = NP * 1024, NI = (NP + NW - 1) / NW, NLO = NP / NW;
  static_assert(CH % 16 == 0 && (YK_BT * CH) % 64 == 0, "whole swizzle groups, whole pieces");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int rt = wave % NRT, th = wave / NRT;
  const long ldy = p.ldy;
  const bool extra = NP % NW != 0 && wave < NP % NW;      // this wave issues NLO + 1 direct loads per tile (wave-uniform)

  auto gload = [&](int buf, int tile) {
    const int fs = tile / p.tpf, fj = tile - fs * p.tpf;
    const long m0 = (long)fs * p.M + (long)fj * YK_BT;
    const int last = min(p.M - fj * YK_BT, YK_BT) - 1;       // rows beyond the frame's last token re-read it (never stored)
    char* dst = smem + buf * BUF + 1024 * wave;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if ((i < NLO || extra) && !(HOP1_DISSECT & 4)) {       // (compile-time true except for the last round of pieces: wave-uniform)
        const int slot = 64 * (wave + NW * i) + lane, row = slot / CH, cc = (slot % CH) ^ (row & 15);
        __builtin_amdgcn_global_load_lds((gptr_t)(p.Y + ((m0 + min(row, last)) * ldy + cc * 8) * 2), (lptr_t)(dst + 1024 * NW * i), 16, 0, YK_AUX);
      }
    }
  };

#if HOP1_DISSECT & 8      // (timing experiment: tiles dealt round-robin to the blocks instead of contiguous ranges)
  const int tstep = gridDim.x;
  int tile = blockIdx.x;
  const int t_end = p.ntiles;
#else
  constexpr int tstep = 1;
  int tile = (int)((long)p.ntiles * blockIdx.x / gridDim.x);
  const int t_end = (int)((long)p.ntiles * (blockIdx.x + 1) / gridDim.x);
#endif
  if (tile >= t_end) return;
  bf16x8 af[KS];
  int cur_s = -1;
  gload(0, tile);
  if (tile + tstep < t_end) gload(1, tile + tstep);
  // fragment addresses: chunk 4 ks + q of row 16 th + r sits at chunk (4 ks + q) ^ r = 4 (ks ^ (r >> 2)) + (q ^ (r & 3))
  int yo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) yo[j] = (16 * th + r) * RB + 64 * (j ^ (r >> 2)) + 16 * (q ^ (r & 3));
  for (int it = 0; tile < t_end; ++it, tile += tstep) {
    // In-order counter, issue order per iteration i: [loads of tile i + 2] [the store of tile i].  Tile `tile` has landed once everything
    // but what was issued after its loads is complete: the loads of tile + 1, and the stores of the two previous iterations.
    if (tile + tstep < t_end) {
      if (it == 0) { if (extra) wait_vm<NLO + 1>(); else wait_vm<NLO>(); }
      else if (it == 1) { if (extra) wait_vm<NLO + 2>(); else wait_vm<NLO + 1>(); }
      else { if (extra) wait_vm<NLO + 3>(); else wait_vm<NLO + 2>(); }
    } else {
      wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* sY = smem + (it % YK_NBUF) * BUF;
    const int s = tile / p.tpf;
    const bool newA = s != cur_s && (p.sA1 != 0 || cur_s < 0);
    // the request for tile + 2 (its buffer was read in the previous iteration: every wave has passed this iteration's barrier since) goes out
    // BEFORE the arithmetic, so that two tiles are in flight while this one is multiplied and while the next wait runs -- except in front of a
    // frame change, whose ordinary loads of A would be queued behind it
    if (!newA && tile + 2 * tstep < t_end) gload((it + 2) % YK_NBUF, tile + 2 * tstep);
    if (newA) {          // this frame's A (ordinary loads, complete when the branch ends; they drain the tile in flight: once per frame)
      const unsigned short* A = p.A + (long)s * p.sA1 + (long)min(16 * rt + r, p.rows - 1) * p.lda + 8 * q;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) af[ks] = *(const bf16x8*)(A + 32 * ks);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(af[ks]));
      if (tile + 2 * tstep < t_end) gload((it + 2) % YK_NBUF, tile + 2 * tstep);
    }
    cur_s = s;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < ((HOP1_DISSECT & 1) ? 0 : KS); ks += 2) {
      const bf16x8 y0 = *(const bf16x8*)(sY + yo[ks & 3] + 256 * (ks >> 2)), y1 = *(const bf16x8*)(sY + yo[(ks + 1) & 3] + 256 * ((ks + 1) >> 2));
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y0, af[ks], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y1, af[ks + 1], acc1, 0, 0, 0);
    }
    {   // lane (r, q): row 16 rt + r, tokens 16 th + 4 q .. + 3 of the tile ; exactly ONE store instruction per wave and tile
      const int fj = tile - s * p.tpf, row = 16 * rt + r, t0 = fj * YK_BT + 16 * th + 4 * q;
      const f32x4 a = acc0 + acc1;
      const bool ok = row < p.rows && t0 < p.M && !(HOP1_DISSECT & 2);      // (M is a multiple of 4: a group of four tokens is inside the frame or outside)
      const long e = (long)s * p.sC1 + (long)row * p.ldc + t0;
      if (p.c_bf16) {
        char* dst = ok ? p.C + e * 2 : p.dump;
        *(u32x2*)dst = u32x2{f2bf(a[0]) | (f2bf(a[1]) << 16), f2bf(a[2]) | (f2bf(a[3]) << 16)};
      } else {
        char* dst = ok ? p.C + e * 4 : p.dump;
        *(f32x4*)dst = a;
      }
    }
  }
}

// ======================================================================================================================================
// yt:  C[row][ch] = sum_tok A[row][tok] Y[tok][ch]     -- per frame (PER_FRAME: A = [frame][row][tokens], K-major rows) or over all tokens
// (A = [token][rows], token-major; the blocks' partial sums go to a slab and kk_hop1_sum adds them in block order).  tok_pair2.hip's
// scheme -- every accumulator in registers: wave w owns channel tiles NCT w .. NCT w + NCT - 1 of this block's 128 NCT channels against
// all NRT row tiles; Y is read transposed (the contraction index is the token) -- on 32-token tiles (one K step) of Y and A in FOUR LDS
// buffers with counted waits (kk_hop1_yk): three tiles in flight while one is multiplied.
// ======================================================================================================================================
struct YTArgs {
  const char* Y; long ldy;                         // bf16 [tokens][ldy]; this block's channels from column 128 NCT blockIdx.y
  const char* A; long lda, sA1;                    // bf16: PER_FRAME [frame][rows][lda] ; else [token][lda]
  int rows;
  char* C; long ldc, sC1; int c_bf16;              // PER_FRAME: [frame][row][ldc] bf16 or fp32
  float* slab;                                     // sum: [gridDim.x][rows16][ldslab] fp32, this block's channels from column 128 NCT blockIdx.y
  long ldslab;
  int M, tpf;                                      // PER_FRAME: tokens per frame, 32-token tiles per frame (the last one ragged)
  int S;                                           // PER_FRAME: frames ; blocks take contiguous frame ranges
  long ntok; int ntiles;                           // sum: tokens / tiles in all (the last tile ragged)
};

template <int OFF>
__device__ __forceinline__ void tr_issue(u32x2& d, unsigned addr) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory"); }
template <int OFF, int ROWB>        // one 16-column fragment: token rows 8 q .. 8 q + 7, both halves
__device__ __forceinline__ void tr_frag2(u32x2 (&f)[2], unsigned base) { tr_issue<OFF>(f[0], base); tr_issue<OFF + 4 * ROWB>(f[1], base); }
__device__ __forceinline__ bf16x8 tr_pack(const u32x2 (&f)[2]) { return __builtin_bit_cast(bf16x8, u32x4{f[0][0], f[0][1], f[1][0], f[1][1]}); }
// the explicit wait for transposing reads issued above (every fragment passes through a volatile statement behind the wait, so that its
// consumers stay behind it)
template <int N>
__device__ __forceinline__ void tr_wait(u32x2 (&x)[N][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(x[i][0]), "+v"(x[i][1]) :: "memory");
}

constexpr int YT_BM = 32, YT_NBUF = 4;
template <int NCT, int NRT, bool PER_FRAME> struct YTGeom {
  static constexpr int CB = 128 * NCT, CHY = CB / 8 + 1, RBY = 16 * CHY, NPY = (YT_BM * CHY + 63) / 64;      // Y tile: 32 rows x CHY chunks (the last one a pad)
  static constexpr int RA = PER_FRAME ? 16 * NRT : YT_BM, CHA = (PER_FRAME ? YT_BM / 8 : 2 * NRT) + 1, RBA = 16 * CHA, NPA = (RA * CHA + 63) / 64;
  static constexpr int OFFA = NPY * 1024, BUF = OFFA + NPA * 1024, NP = NPY + NPA, NLO = NP / 8, NI = (NP + 7) / 8, LDS = YT_NBUF * BUF;
};

template <int NCT, int NRT, bool PER_FRAME>
__global__ void __launch_bounds__(512, 1) kk_hop1_yt(const YTArgs p) {
  using G = YTGeom<NCT, NRT, PER_FRAME>;
  constexpr int BM = YT_BM, CB = G::CB, CHY = G::CHY, RBY = G::RBY, NPY = G::NPY, CHA = G::CHA, RBA = G::RBA, OFFA = G::OFFA, BUF = G::BUF, NP = G::NP, NLO = G::NLO, NI = G::NI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const char* Yb = p.Y + (long)blockIdx.y * CB * 2;
  const long ldy = p.ldy, lda = p.lda;
  const bool extra = NP % 8 != 0 && wave < NP % 8;        // this wave issues NLO + 1 direct loads per tile (wave-uniform)

  f32x4 acc[NRT][NCT];
#pragma unroll
  for (int i = 0; i < NRT; ++i)
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // tile -> (first token row of Y, rows of the tile that are data, A source)
  auto gload = [&](int buf, long m0, int valid, const char* Asrc, int acol_max) {
    char* dst = smem + buf * BUF + 1024 * wave;
    const int last = min(valid, BM) - 1;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int j0 = 8 * i;                                  // pieces j0 .. j0 + 7 of this round: piece j0 + wave is this wave's
      auto ld_y = [&]() {
        const int slot = 64 * (j0 + wave) + lane, row = min(slot / CHY, last), cc = min(slot % CHY, CHY - 2);
        __builtin_amdgcn_global_load_lds((gptr_t)(Yb + ((m0 + row) * ldy + cc * 8) * 2), (lptr_t)(dst + 8192 * i), 16, 0, (PER_FRAME ? YTF_AUX : YTS_AUX));
      };
      auto ld_a = [&]() {
        const int slot = 64 * (j0 + wave - NPY) + lane;
        if constexpr (PER_FRAME) {                           // [row][tokens of the tile]: 4 chunks of 8 tokens + the pad chunk
          const int row = min(slot / CHA, p.rows - 1), cc = min(min(slot % CHA, CHA - 2), acol_max);
          __builtin_amdgcn_global_load_lds((gptr_t)(Asrc + ((long)row * lda + cc * 8) * 2), (lptr_t)(dst + 8192 * i), 16, 0, (PER_FRAME ? YTF_AUX : YTS_AUX));
        } else {                                             // [token][rows]
          const int row = min(slot / CHA, last), cc = min(slot % CHA, CHA - 2);
          __builtin_amdgcn_global_load_lds((gptr_t)(Asrc + ((long)row * lda + cc * 8) * 2), (lptr_t)(dst + 8192 * i), 16, 0, (PER_FRAME ? YTF_AUX : YTS_AUX));
        }
      };
      // (i is a constant after unrolling: only the round that holds the Y / A boundary and the last round keep a wave-uniform branch)
      if (j0 + 8 <= NPY) ld_y();
      else if (j0 >= NPY) { if (i < NLO || extra) ld_a(); }
      else { if (j0 + wave < NPY) ld_y(); else if (i < NLO || extra) ld_a(); }
    }
  };

  // the work of this block: PER_FRAME frames [f0, f1) ; sum: tiles [t0, t1) of the flat token list
  int t0 = 0, t1 = 0;
  if constexpr (PER_FRAME) {
    t0 = (int)((long)p.S * blockIdx.x / gridDim.x) * p.tpf; t1 = (int)((long)p.S * (blockIdx.x + 1) / gridDim.x) * p.tpf;
  } else {
    t0 = (int)((long)p.ntiles * blockIdx.x / gridDim.x); t1 = (int)((long)p.ntiles * (blockIdx.x + 1) / gridDim.x);
  }
  if (t0 >= t1) return;
  auto issue = [&](int buf, int tile) {
    if constexpr (PER_FRAME) {
      const int s = tile / p.tpf, fj = tile - s * p.tpf;
      gload(buf, (long)s * p.M + (long)fj * BM, p.M - fj * BM, p.A + ((long)s * p.sA1 + (long)fj * BM) * 2, (int)((lda - (long)fj * BM) / 8) - 1);
    } else {
      gload(buf, (long)tile * BM, (int)min((long)BM, p.ntok - (long)tile * BM), p.A + (long)tile * BM * lda * 2, 0);
    }
  };
  issue(0, t0);
  if (t0 + 1 < t1) issue(1, t0 + 1);
  if (t0 + 2 < t1) issue(2, t0 + 2);
  for (int it = 0, tile = t0; tile < t1; ++it, ++tile) {
    // this tile has landed once everything but the (up to two) tiles requested after it is complete; the stores of a frame's end are
    // issued behind a request and only ever make the count an over-estimate of what has to be waited for (in-order counter)
    {
      const int ahead = min(2, t1 - 1 - tile);
      if (ahead == 2) { if (extra) wait_vm<2 * NLO + 2>(); else wait_vm<2 * NLO>(); }
      else if (ahead == 1) { if (extra) wait_vm<NLO + 1>(); else wait_vm<NLO>(); }
      else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* sY = smem + (it % YT_NBUF) * BUF;
    if (tile + 3 < t1) issue((it + 3) % YT_NBUF, tile + 3);      // (its buffer was read in the previous iteration: every wave has passed this barrier since)
    int valid;                                            // tokens of this tile that are data
    if constexpr (PER_FRAME) { const int fj = tile % p.tpf; valid = p.M - fj * BM; }
    else valid = (int)min((long)BM, p.ntok - (long)tile * BM);
    {
      const unsigned l0 = (unsigned)(size_t)(lptr_t)sY;
      const unsigned ly = l0 + (8 * q + (r >> 2)) * RBY + (NCT * wave * 16 + 4 * (r & 3)) * 2;
      const unsigned la = l0 + OFFA + (8 * q + (r >> 2)) * RBA + (4 * (r & 3)) * 2;      // (token-major A: transposed reads, as Y)
      const char* pa = sY + OFFA + r * RBA + q * 16;                                   // (row-major A: plain 16-byte reads)
      u32x2 fb[NCT][2];
      u32x4 fa[NRT];                                         // row-major A: plain reads
      u32x2 ft[NRT][2];                                      // token-major A: transposing reads
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        if (c == 0) tr_frag2<0, RBY>(fb[0], ly);
        if (c == 1) tr_frag2<32, RBY>(fb[1 % NCT], ly);
        if (c == 2) tr_frag2<64, RBY>(fb[2 % NCT], ly);
      }
      if constexpr (PER_FRAME) {
#pragma unroll
        for (int i = 0; i < NRT; ++i) fa[i] = *(const u32x4*)(pa + 16 * i * RBA);
      } else {
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
          if (i == 0) tr_frag2<0, RBA>(ft[0], la);
          if (i == 1) tr_frag2<32, RBA>(ft[1 % NRT], la);
          if (i == 2) tr_frag2<64, RBA>(ft[2 % NRT], la);
          if (i == 3) tr_frag2<96, RBA>(ft[3 % NRT], la);
          if (i == 4) tr_frag2<128, RBA>(ft[4 % NRT], la);
        }
      }
      tr_wait<NCT>(fb);
      if constexpr (!PER_FRAME) tr_wait<NRT>(ft);
      // tokens 8 q + j of the tile beyond `valid` are not data (a ragged last tile): their A entries are zeroed
      const int nv = valid - 8 * q;
      u32x4 mk;
#pragma unroll
      for (int e = 0; e < 4; ++e) mk[e] = (2 * e + 1 < nv) ? 0xffffffffu : ((2 * e < nv) ? 0x0000ffffu : 0u);
#pragma unroll
      for (int i = 0; i < NRT; ++i) {
        const u32x4 av = PER_FRAME ? fa[i] : u32x4{ft[i][0][0], ft[i][0][1], ft[i][1][0], ft[i][1][1]};
        const bf16x8 a = __builtin_bit_cast(bf16x8, av & mk);
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, tr_pack(fb[c]), acc[i][c], 0, 0, 0);
      }
    }
    if constexpr (PER_FRAME) {
      if ((tile + 1) % p.tpf == 0) {                       // the frame ends: its rows (block-uniform)
        const int s = tile / p.tpf;
#pragma unroll
        for (int i = 0; i < NRT; ++i)
#pragma unroll
          for (int c = 0; c < NCT; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {                  // lane (r, q): row 16 i + 4 q + e, channel 16 (NCT wave + c) + r
              const int row = 16 * i + 4 * q + e;
              const long o = (long)s * p.sC1 + (long)row * p.ldc + (long)blockIdx.y * CB + 16 * (NCT * wave + c) + r;
              if (row < p.rows) {
                if (p.c_bf16) *(unsigned short*)(p.C + o * 2) = (unsigned short)f2bf(acc[i][c][e]);
                else *(float*)(p.C + o * 4) = acc[i][c][e];
              }
              acc[i][c][e] = 0.f;
            }
          }
      }
    }
  }
  if constexpr (!PER_FRAME) {
    float* sl = p.slab + (long)blockIdx.x * 16 * NRT * p.ldslab + (long)blockIdx.y * CB;
#pragma unroll
    for (int i = 0; i < NRT; ++i)
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) sl[(long)(16 * i + 4 * q + e) * p.ldslab + 16 * (NCT * wave + c) + r] = acc[i][c][e];
  }
}

// out[row][c] = sum over the blocks' slabs, in block order (four lanes per 4-element vector, as kk_dpair_reduce), stored bf16 or fp32
__global__ void __launch_bounds__(256) kk_hop1_sum(const float* __restrict__ slab, int nb, long per, int rows, int cols, long ldslab, char* __restrict__ out, long ldo, int o_bf16) {
  const int lane = threadIdx.x & 63, part = lane >> 4;
  const long v = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (lane & 15);      // 4-element vector of the rows x cols result
  const int vpr = cols / 4;
  const long nvec = (long)rows * vpr;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const int row = (int)(v / vpr), c4 = (int)(v % vpr) * 4;
  if (v < nvec) {
    const float* sl = slab + (long)row * ldslab + c4;
    int b = part;
    for (; b + 12 < nb; b += 16) {
      const f32x4 a0 = *(const f32x4*)(sl + (long)b * per), a1 = *(const f32x4*)(sl + (long)(b + 4) * per);
      const f32x4 a2 = *(const f32x4*)(sl + (long)(b + 8) * per), a3 = *(const f32x4*)(sl + (long)(b + 12) * per);
      s += (a0 + a1) + (a2 + a3);
    }
    for (; b < nb; b += 4) s += *(const f32x4*)(sl + (long)b * per);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { s[e] += __shfl_xor(s[e], 16, 64); s[e] += __shfl_xor(s[e], 32, 64); }
  if (v < nvec && part == 0) {
    const long o = (long)row * ldo + c4;
    if (o_bf16) *(u32x2*)(out + o * 2) = u32x2{f2bf(s[0]) | (f2bf(s[1]) << 16), f2bf(s[2]) | (f2bf(s[3]) << 16)};
    else *(f32x4*)(out + o * 4) = s;
  }
}

// Sites of fewer than 32 768 tokens of Y keep the tiled engine: a persistent block per CU then has two or three tiles (or a single frame)
// to amortise its prologue over and the engine's small tiles fill the chip better (measured at 20 480 tokens: yk 20.7 us against 12 - 15,
// per-frame yt 32.9 against ~18).  avmoe_test_hooks bit 4 (test hook): small sites as well.
bool hop1s_small(long ntok) { return ntok < 32768 && !(test_hook_mask() & HOOK_HOP1S_FORCE); }

}  // namespace

// 0 = launched, 1 = shape not served (the caller runs the tiled engine), < 0 error
int k_hop1_yk(const void* Y, long ldy, int S, int M, int Cy, const void* A, long lda, long sA1, int rows, void* C, long ldc, long sC1, int c_bf16, void* dump, hipStream_t st) {
  if (Cy != 768 || rows < 1 || rows > 80 || M < 4 || M % 4 || S < 1 || ldy % 8 || lda % 8 || sA1 % 8 || ldc % 4 || sC1 % 4 || !dump ||
      ((uintptr_t)Y % 16) || ((uintptr_t)A % 16) || ((uintptr_t)C % 16) || ((uintptr_t)dump % 16) || hop1s_small((long)S * M))
    return 1;
  const int cus = cu_count();
  if (cus <= 0) { set_last_error("hop1_yk: device query"); return ERR_LAUNCH; }
  YKArgs p;
  p.Y = (const char*)Y; p.ldy = ldy; p.A = (const unsigned short*)A; p.lda = lda; p.sA1 = sA1; p.C = (char*)C; p.ldc = ldc; p.sC1 = sC1; p.c_bf16 = c_bf16; p.dump = (char*)dump;
  p.M = M; p.tpf = (M + YK_BT - 1) / YK_BT; p.ntiles = S * p.tpf; p.rows = rows;
  const int gx = std::min(cus, p.ntiles);
  constexpr int KS = 24, LDS = YK_NBUF * (YK_BT * 4 * KS / 64) * 1024;
  const double bytes = (double)S * M * Cy * 2 + (double)(sA1 ? S : 1) * rows * Cy * 2 + (double)S * rows * M * (c_bf16 ? 2 : 4);
  ProfScope ps("k_hop1_yk", (long)S * M, bytes, 2.0 * S * M * Cy * rows, st);
  if (rows <= 64) {
    static LdsAttrOnce attr;
    AVMOE_TRY(attr.ensure((const void*)kk_hop1_yk<KS, 4>, LDS, "hop1_yk"));
    hipLaunchKernelGGL((kk_hop1_yk<KS, 4>), dim3((unsigned)gx), dim3(512), LDS, st, p);
  } else {
    static LdsAttrOnce attr;
    AVMOE_TRY(attr.ensure((const void*)kk_hop1_yk<KS, 5>, LDS, "hop1_yk"));
    hipLaunchKernelGGL((kk_hop1_yk<KS, 5>), dim3((unsigned)gx), dim3(640), LDS, st, p);
  }
  AVMOE_CHECK_LAUNCH("hop1_yk");
  return OK;
}

namespace {
template <int NCT, int NRT, bool PF>
int launch_yt(const YTArgs& p, int gx, int gy, hipStream_t st) {
  constexpr int LDS = YTGeom<NCT, NRT, PF>::LDS;
  static LdsAttrOnce attr;
  AVMOE_TRY(attr.ensure((const void*)kk_hop1_yt<NCT, NRT, PF>, LDS, "hop1_yt"));
  hipLaunchKernelGGL((kk_hop1_yt<NCT, NRT, PF>), dim3((unsigned)gx, (unsigned)gy), dim3(512), LDS, st, p);
  AVMOE_CHECK_LAUNCH("hop1_yt");
  return OK;
}
}  // namespace

// per frame: C[s] (rows x Cy) = A[s] (rows x M, K-major rows of lda elements) Y[s]
int k_hop1_yt_frames(const void* Y, long ldy, int S, int M, int Cy, const void* A, long lda, long sA1, int rows, void* C, long ldc, long sC1, int c_bf16, hipStream_t st) {
  if (Cy % 256 || rows < 17 || rows > 80 || M < 1 || S < 1 || ldy % 8 || lda % 8 || sA1 % 8 || lda < M ||
      ((uintptr_t)Y % 16) || ((uintptr_t)A % 16) || ((uintptr_t)C % 4) || hop1s_small((long)S * M))
    return 1;
  const int cus = cu_count();
  if (cus <= 0) { set_last_error("hop1_yt: device query"); return ERR_LAUNCH; }
  // channels per block: 384 or 256, whichever balances the frames better over one block per CU
  auto eff = [&](int cb) {
    if (Cy % cb) return 0.0;
    const int gy = Cy / cb, gx = std::max(1, std::min(S, cus / gy));
    return (double)S / ((double)((S + gx - 1) / gx) * gx) * std::min(1.0, (double)gx * gy / cus);
  };
  const int cb = eff(384) >= eff(256) ? 384 : 256;
  const int gy = Cy / cb, gx = std::max(1, std::min(S, cus / gy));
  YTArgs p{};
  p.Y = (const char*)Y; p.ldy = ldy; p.A = (const char*)A; p.lda = lda; p.sA1 = sA1; p.rows = rows; p.C = (char*)C; p.ldc = ldc; p.sC1 = sC1; p.c_bf16 = c_bf16;
  p.M = M; p.tpf = (M + YT_BM - 1) / YT_BM; p.S = S;
  const double bytes = (double)S * M * Cy * 2 + (double)S * rows * M * 2 * gy + (double)S * rows * Cy * (c_bf16 ? 2 : 4);
  ProfScope ps("k_hop1_yt_frames", (long)S * M, bytes, 2.0 * S * M * Cy * rows, st);
  if (rows <= 64) return cb == 384 ? launch_yt<3, 4, true>(p, gx, gy, st) : launch_yt<2, 4, true>(p, gx, gy, st);
  return cb == 384 ? launch_yt<3, 5, true>(p, gx, gy, st) : launch_yt<2, 5, true>(p, gx, gy, st);
}

// over all tokens: C (rows x Cy) = A^T Y with A = [token][lda] (columns 0 .. rows - 1), C bf16 or fp32 ; slabs: fp32 workspace
int k_hop1_yt_sum(const void* Y, long ldy, long ntok, int Cy, const void* A, long lda, int rows, void* C, long ldc, int c_bf16, float* slabs, size_t slab_cap, hipStream_t st) {
  if (Cy % 384 || rows < 17 || rows > 64 || lda < 64 || ntok < YT_BM || ldy % 8 || lda % 8 || ldc % 4 || !slabs ||
      ((uintptr_t)Y % 16) || ((uintptr_t)A % 16) || ((uintptr_t)C % 16) || ((uintptr_t)slabs % 16) || hop1s_small(ntok))
    return 1;
  const int cus = cu_count();
  if (cus <= 0) { set_last_error("hop1_yt: device query"); return ERR_LAUNCH; }
  const int gy = Cy / 384, ntiles = (int)((ntok + YT_BM - 1) / YT_BM), gx = std::max(1, std::min(ntiles, cus / gy));
  if ((size_t)gx * 64 * Cy > slab_cap) return 1;
  YTArgs p{};
  p.Y = (const char*)Y; p.ldy = ldy; p.A = (const char*)A; p.lda = lda; p.rows = rows; p.slab = slabs; p.ldslab = Cy; p.ntok = ntok; p.ntiles = ntiles;
  {
    const double bytes = (double)ntok * Cy * 2 + (double)ntok * 64 * 2 * gy + (double)gx * 64 * Cy * 4;
    ProfScope ps("k_hop1_yt_sum", ntok, bytes, 2.0 * ntok * Cy * rows, st);
    AVMOE_TRY((launch_yt<3, 4, false>(p, gx, gy, st)));
  }
  {
    const long nvec = (long)rows * Cy / 4;
    ProfScope ps("k_hop1_sum", (long)rows * Cy, (double)gx * rows * Cy * 4.0, 0.0, st);
    hipLaunchKernelGGL(kk_hop1_sum, dim3((unsigned)((nvec + 63) / 64)), dim3(256), 0, st, slabs, gx, (long)64 * Cy, rows, Cy, (long)Cy, (char*)C, ldc, c_bf16);
    AVMOE_CHECK_LAUNCH("hop1_sum");
  }
  return OK;
}

}  // namespace avmoe
