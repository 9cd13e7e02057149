// The gradient of a token tensor that is X of site A and Y of site B (the two adapter sites of one backbone layer, net_trans_v3.py:695-698),
// written ONCE (round 5):
//
//   dT[s] =  dZx[s] Wt + [dL2 | dsx | 1][s] [T ; 1 ; dm1/N][s] + rs2x X[s]          site A's dX   (moe_backward.cpp, phase 5)
//          + [Bm ; wbar]_B[s]^T dV_B[s] + dR_B[s]^T Q_B                             site B's dY   (moe_backward.cpp, phase 6)
//
// As two kernels (kk_dx_stream2 overwriting, the twelve-wave streaming GEMM adding behind an event) the tensor crossed the memory interface
// three times -- written, read, written: 1 GB more than necessary at the cfg-2 audio tokens -- and site B's dY operands are skinny (65 + 64
// values per token) beside the row it adds to.  Here they are two more K segments of the dX pass.
//
// Shape of the kernel: dx_stream2.hip's (eight waves, one block per CU and group, wave w = 48 channels of the group: every B-side matrix
// as stationary MFMA fragments -- Wt and Q_B for the whole kernel, T[s] and dV_B[s] re-gathered at a frame change --, token tiles by
// direct global -> LDS loads, products computed transposed so that a lane ends up with four consecutive channels of a token) on 32-token
// tiles in THREE LDS buffers with counted waits (hop1_stream.hip::kk_hop1_yk: two tiles in flight while one is multiplied; the in-order
// memory counter leaves the newest tile's loads and the last two iterations' stores pending).  Every wave issues exactly six store
// instructions per tile (rows beyond a ragged frame end go to a dump word) so that the counts are exact.
#include "gemm.h"
#include "common.h"
#include "prof.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>

#ifndef DX3_AUX
#define DX3_AUX 0      // cache policy of the direct loads (common.h::AVMOE_LDS_AUX): the non-temporal hint measured neutral or worse here (its X is re-read by the next kernel of the chain)
#endif

namespace avmoe {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct DX3Args {
  // site A (the tensor is its X)
  const char* X; long ldx;             // bf16 [tokens][ldx], group g at column g * 384 (the operand of the row-scale term)
  const char* dZx; long ldz;           // bf16 [tokens][ldz], group g at column g * 128
  const char* dL2; long ldl;           // bf16 [tokens][ldl >= 72]: columns 0 .. K2 - 1 used
  const float* rs;                     // fp32 [tokens]
  const unsigned short* Wt; long ldw, sWg;      // bf16 [g][128][ldw]: row = bottleneck column, column = channel
  const unsigned short* Text; long ldt, sT1;    // bf16 [frame][K2][ldt], group g at column g * 384
  // site B (the tensor is its Y)
  const char* Bm; long ldb, sB1;       // bf16 [frame][rows >= KB][ldb]: [Bm ; wbar], tokens along the row
  const char* dRT; long ldr;           // bf16 [tokens][ldr]: columns 0 .. KQ - 1 used (ldr >= 8 ncr)
  const unsigned short* dV; long ldv, sV1;      // bf16 [frame][KB][ldv], group g at column g * 384
  const unsigned short* Q; long ldq;            // bf16 [KQ][ldq], group g at column g * 384
  char* dX; long ldc;                  // bf16 [tokens][ldc], group g at column g * 384
  char* dump;                          // >= 256 writable bytes nobody reads
  int N, tps, ntiles, K2, KB, KQ, ncr; // tokens per frame, 32-token tiles per frame (the last one ragged), tiles in all, rows of T[s] / [Bm ; wbar] / Q, 16-byte chunks of a dRT row
};

#ifndef DX3_DISSECT
#define DX3_DISSECT 0      // development builds (timing only): bit 0 = no matrix phase, bit 1 = stores to the dump word
#endif
// Block shape, measured on MI355X (cfg-2 audio tokens, 327 680 x 768, every launch alone on the GPU; the two kernels it replaces: 270 + 241 us):
//   4 waves (one per SIMD: 512 registers each, the 72 stationary fragments fit), 32-token tiles x 3 buffers   344 - 352 us   <- built
//   4 waves, 16-token tiles x 5 (x 4) buffers                                                                 417 us (per-tile overhead: waits, barrier, addresses)
//   8 waves (256 registers: the frame-change gathers spill, the tile loop does not), 32 x 3 / 16 x 5           362 / 395 us
// Timing-only builds of the first (DX3_DISSECT): the tile stream alone 159 us (5.1 TB/s), + the stores 270 us (4.85 TB/s over the
// 1.31 GB moved: the memory-bound floor), + the matrix phase instead of the stores 258 us; everything 349 us.  A wave that has its SIMD to
// itself hides nothing: what the instruction stream of a tile costs beyond the matrix instructions is added to the step.  Three changes took
// 344 - 366 us to 310 - 317 us (same boxes; visual tokens 123 -> 108 - 112 us):
//   * the per-piece source addresses (a division by the padded row length, two clamps, a 64-bit multiply-add each: ~500 vector instructions
//     per tile, as many cycles as the matrix phase) -> lane offsets computed once, `scalar base + 32-bit lane offset` loads;
//   * read -> wait -> six products per K step (the pipe idles for one LDS latency per step) -> the next step's read issued in front of
//     the products, counted lgkmcnt waits (two steps ahead: no further gain);
//   * the stationary fragments that do not fit the 256 architectural registers were copied in and out of the accumulation registers around
//     every product (four v_accvgpr_read per matrix instruction) -> pinned there, read in place.
// Built and dropped: the slabs as one software pipeline across tiles (slab j's row-scale term, conversion and stores between the products
// of slab j + 1: a second accumulator set and the slab's X values held in registers) -- 538 registers, 24 - 48 of them spilled.
#ifndef DX3_NWV
#define DX3_NWV 4
#endif
#ifndef DX3_BM
#define DX3_BM 32
#define DX3_NBUF 3
#endif
constexpr int BM = DX3_BM, NSL = BM / 16, NWV = DX3_NWV, NTHR = 64 * NWV, NCT = 24 / NWV, NBUF = DX3_NBUF;      // four waves, one per SIMD: 6 channel tiles each (the 72 stationary fragments need the whole register file of a SIMD lane: 512 registers per wave)
constexpr int CHX = 49, CHZ = 17, CHL = 10, CHR = 9, CHB = BM / 8 + 1;              // 16-byte chunks per LDS row (the last one a pad)
constexpr int RBX = 16 * CHX, RBZ = 16 * CHZ, RBL = 16 * CHL, RBR = 16 * CHR, RBB = 16 * CHB;
constexpr int PX = (BM * CHX + 63) / 64, PZ = (BM * CHZ + 63) / 64, PL = (BM * CHL + 63) / 64, PR = (BM * CHR + 63) / 64, PB = (96 * CHB + 63) / 64;      // 1 KB pieces per sub-tile
constexpr int OFFZ = PX * 1024, OFFL = OFFZ + PZ * 1024, OFFR = OFFL + PL * 1024, OFFB = OFFR + PR * 1024, OFFS = OFFB + PB * 1024, BUF = OFFS + 256;
constexpr int DX3_LDS = NBUF * BUF;
static_assert(DX3_LDS <= 160 * 1024, "the buffers fit one CU's LDS");

__device__ __forceinline__ unsigned int f2bf(float x) { return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bflo(unsigned int u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bfhi(unsigned int u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// s_waitcnt vmcnt(n) for a run-time (wave-uniform) n: the immediate has to be a constant
__device__ __forceinline__ void wait_vm_n(int n) {
  switch (n) {
#define W1(k) case k: wait_vm<k>(); break;
#define W8(k) W1(k) W1(k + 1) W1(k + 2) W1(k + 3) W1(k + 4) W1(k + 5) W1(k + 6) W1(k + 7)
    W8(0) W8(8) W8(16) W8(24) W8(32) W8(40) W8(48) W1(56) W1(57) W1(58) W1(59) W1(60) W1(61) W1(62)
#undef W8
#undef W1
    default: wait_vm<63>(); break;
  }
}
// LDS reads issued and waited for by hand (the tile is read-only between two barriers: no memory clobber, the compiler's own reads may move)
template <int OFF>
__device__ __forceinline__ void tr_issue(u32x2& d, unsigned addr) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int OFF>
__device__ __forceinline__ void rd128(u32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
// wait until at most N LDS reads are pending; the registers named are those the reads before them filled (their users stay behind the wait)
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N> __device__ __forceinline__ void wait_lgkm2(u32x2& a, u32x2& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }

__global__ void __launch_bounds__(NTHR, 1) kk_dx_stream3(const DX3Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;      // (wave: a scalar -- the piece dealing below branches on it)
  const int g = blockIdx.y;
  const char* Xb = p.X + (long)g * 384 * 2;
  const char* Zb = p.dZx + (long)g * 128 * 2;
  const long ldx = p.ldx, ldz = p.ldz, ldl = p.ldl, ldr = p.ldr, ldb = p.ldb;
  const int c0 = 16 * NCT * wave;                           // this wave's channels c0 .. c0 + 16 NCT - 1 of the group

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
  bf16x8 bw[NCT][4], bt[NCT][3], bq[NCT][2], bv[NCT][3];
  {
    const unsigned short* W = p.Wt + (long)g * p.sWg;
    const unsigned short* Qg = p.Q + (long)g * 384;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bw[ct][ks] = frag_mn(W, p.ldw, c0 + 16 * ct + r, 32 * ks + 8 * q, 128);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bq[ct][ks] = frag_mn(Qg, p.ldq, c0 + 16 * ct + r, 32 * ks + 8 * q, p.KQ);
    }
    // Wt and Q_B (stationary for the whole kernel: 144 registers) pinned to the accumulation half of the register file, where the matrix
    // instructions read them directly.  Left alone the compiler keeps every fragment in the 256 architectural registers and copies the
    // overflow in and out of the other 256 around each product: four v_accvgpr_read per matrix instruction, 320 per tile (-5.5 %).
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+a"(bw[ct][ks]));
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+a"(bq[ct][ks]));
    }
  }
  constexpr int NPC = PX + PZ + PL + PR + PB, NS = NSL * NCT;            // pieces per tile ; stores per tile and wave
  const int nl = (NPC - 1 - wave) / NWV + 1 + (wave == NWV - 1 ? 1 : 0);       // direct loads per tile of this wave: pieces wave + NWV i, + the row scales for the last wave

  // One buffer image = [X 25 | dZx 9 | dL2 5 | dRT 5 | [Bm ; wbar] 8] pieces of 1 KB + the tile's 32 row scales; piece P = wave + NWV i.
  // Where a lane's 16 bytes of piece P come from, relative to the tile's first row of that tensor, does not depend on the tile: the
  // offsets are computed ONCE (32 bits each; one register per piece of this wave) and a tile costs five scalar base addresses + one
  // instruction per piece.  (Computed per tile -- a division by the padded row length, two clamps and a 64-bit multiply-add per piece -- the
  // addresses were ~500 vector instructions per tile on a wave that has its SIMD to itself: as long as the whole matrix phase.)
  constexpr int B1 = PX, B2 = B1 + PZ, B3 = B2 + PL, B4 = B3 + PR, B5 = B4 + PB, NPW = (B5 + NWV - 1) / NWV;
  // source of slot (64 j + ln) of a sub-tile: `last` = last row of the tile inside the frame, `bmax` = last 16-byte chunk inside a [Bm ; wbar] row
  auto off_x = [&](int j, int ln, int last) { const int slot = 64 * j + ln, row = min(slot / CHX, last), cc = min(slot % CHX, CHX - 2); return (unsigned)((row * ldx + cc * 8) * 2); };
  auto off_z = [&](int j, int ln, int last) { const int slot = 64 * j + ln, row = min(slot / CHZ, last), cc = min(slot % CHZ, CHZ - 2); return (unsigned)((row * ldz + cc * 8) * 2); };
  auto off_l = [&](int j, int ln, int last) { const int slot = 64 * j + ln, row = min(slot / CHL, last), cc = min(slot % CHL, CHL - 2); return (unsigned)((row * ldl + cc * 8) * 2); };
  auto off_r = [&](int j, int ln, int last) { const int slot = 64 * j + ln, row = min(slot / CHR, last), cc = min(slot % CHR, p.ncr - 1); return (unsigned)((row * ldr + cc * 8) * 2); };
  // [Bm ; wbar] of the frame: row l (those beyond KB - 1 re-read the last one: their B-side fragments are zero), the tile's 32 tokens as 4 chunks
  auto off_b = [&](int j, int ln, int bmax) { const int slot = 64 * j + ln, row = min(slot / CHB, p.KB - 1), cc = min(min(slot % CHB, CHB - 2), bmax); return (unsigned)((row * ldb + cc * 8) * 2); };
  auto off_p = [&](int P, int ln, int last, int bmax) {          // (P: wave-uniform)
    return P < B1 ? off_x(P, ln, last) : P < B2 ? off_z(P - B1, ln, last) : P < B3 ? off_l(P - B2, ln, last) : P < B4 ? off_r(P - B3, ln, last) : P < B5 ? off_b(P - B4, ln, bmax) : 0u;
  };
  unsigned voff[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    voff[i] = off_p(wave + NWV * i, lane, BM - 1, CHB - 2);
    asm volatile("" : "+v"(voff[i]));                       // (opaque: one register each for the whole kernel, not re-derived per tile)
  }
  const int nfr = p.ntiles / p.tps;
  auto gload = [&](int buf, int tile) {
    const int fs = tile / p.tps, fj = tile - fs * p.tps;
    const long m0 = (long)fs * p.N + (long)fj * BM;         // first token of the tile
    const int last = min(p.N - fj * BM, BM) - 1;            // last row of the tile inside the frame (a ragged last tile: the rows beyond are never stored)
    char* dst = smem + buf * BUF + 1024 * wave;
    // five tensors, five scalar bases
    const char* base[5] = {Xb + m0 * ldx * 2, Zb + m0 * ldz * 2, p.dL2 + m0 * ldl * 2, p.dRT + m0 * ldr * 2, p.Bm + ((long)fs * p.sB1 + (long)fj * BM) * 2};
    // A ragged tile's rows beyond the frame read the next frame's first rows ([Bm ; wbar]: the next row's first tokens) -- finite or not, they
    // only reach outputs that are not stored.  The last frame has no next one: its ragged tile takes the clamped addresses, computed on the spot.
    const bool clamp = last < BM - 1 && fs == nfr - 1;
    if (clamp) {                                            // (wave-uniform; once per launch, in one block)
      const int bmax = (int)((ldb - (long)fj * BM) / 8) - 1;
#pragma unroll 1
      for (int i = 0; i < NPW; ++i) {
        const int P = wave + NWV * i;
        if (P >= B5) break;
        const int t = P < B1 ? 0 : P < B2 ? 1 : P < B3 ? 2 : P < B4 ? 3 : 4;
        __builtin_amdgcn_global_load_lds((gptr_t)(base[t] + off_p(P, lane, last, bmax)), (lptr_t)(dst + 1024 * NWV * i), 16, 0, DX3_AUX);
      }
      if (wave == NWV - 1) __builtin_amdgcn_global_load_lds((gptr_t)(p.rs + m0 + min(lane, last)), (lptr_t)(smem + buf * BUF + OFFS), 4, 0, DX3_AUX);
      return;
    }
    // piece P = wave + NWV i (i is a constant after unrolling: only the rounds that hold a boundary between two sub-tiles keep a wave-uniform branch)
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int P = wave + NWV * i;
      char* d = dst + 1024 * NWV * i;
      int t;                                                // which tensor
      if (NWV * i + NWV <= B1) t = 0;
      else if (NWV * i >= B1 && NWV * i + NWV <= B2) t = 1;
      else if (NWV * i >= B2 && NWV * i + NWV <= B3) t = 2;
      else if (NWV * i >= B3 && NWV * i + NWV <= B4) t = 3;
      else if (NWV * i >= B4 && NWV * i + NWV <= B5) t = 4;
      else t = P < B1 ? 0 : P < B2 ? 1 : P < B3 ? 2 : P < B4 ? 3 : P < B5 ? 4 : -1;
      if (t < 0) continue;
      unsigned o = voff[i];
      asm volatile("" : "+v"(o));                           // (the zero-extension stays here, beside the scalar base: the load takes `scalar base + 32-bit lane offset` as it is)
      __builtin_amdgcn_global_load_lds((gptr_t)(base[t] + o), (lptr_t)d, 16, 0, DX3_AUX);
    }
    if (wave == NWV - 1) __builtin_amdgcn_global_load_lds((gptr_t)(p.rs + m0 + min(lane, BM - 1)), (lptr_t)(smem + buf * BUF + OFFS), 4, 0, DX3_AUX);       // the tile's row scales
  };

  // contiguous tile ranges (few frame changes per block)
  int tile = (int)((long)p.ntiles * blockIdx.x / gridDim.x);
  const int t_end = (int)((long)p.ntiles * (blockIdx.x + 1) / gridDim.x);
  if (tile >= t_end) return;
#pragma unroll
  for (int j = 0; j < NBUF - 1; ++j)
    if (tile + j < t_end) gload(j, tile + j);
  int cur_s = -1;
  for (int it = 0; tile < t_end; ++it, ++tile) {
    // In-order counter, issue order per iteration i: [loads of tile i + NBUF - 1] [the NS stores of tile i].  Tile `tile` has landed once
    // everything but what was issued after its loads is complete: the loads of the (up to NBUF - 2) tiles requested after it and the stores of
    // the last (up to NBUF - 1) iterations.
#ifndef DX3_ISSUE
#define DX3_ISSUE 0        // development: 1 = the next request between the two slabs of a tile instead of in front of them (NBUF 3, two slabs)
#endif
#if DX3_ISSUE == 1
    wait_vm_n(min(1, t_end - 1 - tile) * nl + (it == 0 ? 0 : (it == 1 ? NS : NS + NS / 2)));
#else
    wait_vm_n(min(NBUF - 2, t_end - 1 - tile) * nl + min(it, NBUF - 1) * NS);
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* sX = smem + (it % NBUF) * BUF;
    const char* sZ = sX + OFFZ;
    const char* sL = sX + OFFL;
    const char* sR = sX + OFFR;
    const float* sS = (const float*)(sX + OFFS);
    const int s = tile / p.tps;
    if (s != cur_s) {          // this frame's T[s] and dV_B[s] (block-uniform, a few times per block; ordinary loads, complete when the branch ends)
      cur_s = s;
      const unsigned short* T = p.Text + (long)s * p.sT1 + (long)g * 384;
      const unsigned short* V = p.dV + (long)s * p.sV1 + (long)g * 384;
      // (two batches of 72 two-byte loads: all 144 at once do not fit the registers the stationary fragments leave)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) bt[ct][ks] = frag_mn(T, p.ldt, c0 + 16 * ct + r, 32 * ks + 8 * q, p.K2);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) asm volatile("" : "+v"(bt[ct][ks]));
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) bv[ct][ks] = frag_mn(V, p.ldv, c0 + 16 * ct + r, 32 * ks + 8 * q, p.KB);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) asm volatile("" : "+v"(bv[ct][ks]));
    }
    // the request for tile + NBUF - 1 (its buffer was read in the previous iteration: every wave has passed this iteration's barrier since)
    // before the arithmetic: NBUF - 1 tiles are in flight while this one is multiplied
#if DX3_ISSUE != 1
    if (tile + NBUF - 1 < t_end) gload((it + NBUF - 1) % NBUF, tile + NBUF - 1);
#endif
    const int fj = tile - s * p.tps, valid = p.N - fj * BM;      // rows of this tile inside the frame (>= BM: all of them)
    const long m0 = (long)s * p.N + (long)fj * BM;
    // one 16-token slab at a time (three independent accumulator chains; two slabs at once do not fit beside the 36 stationary fragments)
#pragma unroll
    for (int h = 0; h < NSL; ++h) {
#if DX3_ISSUE == 1
      if (h == 1 && tile + NBUF - 1 < t_end) gload((it + NBUF - 1) % NBUF, tile + NBUF - 1);
#endif
      f32x4 acc[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      // The slab's A-side fragments, one K step of 32 each, in the order they are multiplied:
      //    0 ..  2   [Bm ; wbar]_B^T (the tile is [l][token]: transposing reads, two per step; lane (r, q): token r, rows 8 q .. 8 q + 7 of the step)   x dV_B
      //    3 ..  6   dZx                                              x Wt
      //    7 ..  9   [dL2 | dsx | 1] (columns beyond K2 masked)       x [T ; 1 ; dm1/N]
      //   10 .. 11   dR_B^T                                           x Q_B  (its fragments are zero beyond KQ rows)
      // software-pipelined by hand: step k + 1's read is issued BEFORE step k's six products and waited for with a counted lgkmcnt -- this
      // wave has its SIMD to itself, nobody else covers the 100+ cycles of an LDS read (as the compiler schedules it, read -> wait -> six
      // products, the matrix pipe idles for one read latency per step: as long as the products themselves).
      if (!(DX3_DISSECT & 1)) {
        const unsigned aB = (unsigned)(size_t)(lptr_t)(sX + OFFB) + (8 * q + (r >> 2)) * RBB + (4 * (r & 3)) * 2 + 32 * h;
        const unsigned aZ = (unsigned)(size_t)(lptr_t)(sZ + (16 * h + r) * RBZ + q * 16);
        const unsigned aL = (unsigned)(size_t)(lptr_t)(sL + (16 * h + r) * RBL + q * 16);
        const unsigned aR = (unsigned)(size_t)(lptr_t)(sR + (16 * h + r) * RBR + q * 16);
        u32x2 pa, pb, qa, qb;
        u32x4 u, w;
#define DX3_MM(frag, ks, af_)                                                                                                     \
  {                                                                                                                               \
    const bf16x8 af__ = __builtin_bit_cast(bf16x8, af_);                                                                          \
    _Pragma("unroll") for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag[ct][ks], af__, acc[ct], 0, 0, 0); \
  }
#define DX3_MASK(v, ks)                                                                                                           \
  {                                                                                                                               \
    const int nvk = p.K2 - (32 * (ks) + 8 * q);                                                                                   \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] &= (2 * e + 1 < nvk) ? 0xffffffffu : ((2 * e < nvk) ? 0x0000ffffu : 0u);   \
  }
        tr_issue<0 * 32 * RBB>(pa, aB); tr_issue<0 * 32 * RBB + 4 * RBB>(pb, aB);
        tr_issue<1 * 32 * RBB>(qa, aB); tr_issue<1 * 32 * RBB + 4 * RBB>(qb, aB);
        wait_lgkm2<2>(pa, pb);
        DX3_MM(bv, 0, (u32x4{pa[0], pa[1], pb[0], pb[1]}));
        tr_issue<2 * 32 * RBB>(pa, aB); tr_issue<2 * 32 * RBB + 4 * RBB>(pb, aB);
        wait_lgkm2<2>(qa, qb);
        DX3_MM(bv, 1, (u32x4{qa[0], qa[1], qb[0], qb[1]}));
        rd128<0>(u, aZ);
        wait_lgkm2<1>(pa, pb);
        DX3_MM(bv, 2, (u32x4{pa[0], pa[1], pb[0], pb[1]}));
        rd128<64>(w, aZ);  wait_lgkm<1>(u); DX3_MM(bw, 0, u);
        rd128<128>(u, aZ); wait_lgkm<1>(w); DX3_MM(bw, 1, w);
        rd128<192>(w, aZ); wait_lgkm<1>(u); DX3_MM(bw, 2, u);
        rd128<0>(u, aL);   wait_lgkm<1>(w); DX3_MM(bw, 3, w);
        rd128<64>(w, aL);  wait_lgkm<1>(u); DX3_MASK(u, 0); DX3_MM(bt, 0, u);
        rd128<128>(u, aL); wait_lgkm<1>(w); DX3_MASK(w, 1); DX3_MM(bt, 1, w);
        rd128<0>(w, aR);   wait_lgkm<1>(u); DX3_MASK(u, 2); DX3_MM(bt, 2, u);
        rd128<64>(u, aR);  wait_lgkm<1>(w); DX3_MM(bq, 0, w);
        wait_lgkm<0>(u);   DX3_MM(bq, 1, u);
#undef DX3_MM
#undef DX3_MASK
      }
      {                                                    // lane (r, q): token r of the slab, channels c0 + 16 ct + 4 q .. + 3 ; three stores per slab, always
        const int row = 16 * h + r;
        const bool ok = row < valid && !(DX3_DISSECT & 2);
        const float rs = sS[row];
        char* out = ok ? p.dX + ((m0 + row) * p.ldc + (long)g * 384 + c0 + 4 * q) * 2 : p.dump;      // (the dump: 32 (NCT - 1) + 8 bytes)
        constexpr int step = 32;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const u32x2 xv = *(const u32x2*)(sX + row * RBX + (c0 + 16 * ct + 4 * q) * 2);
          const f32x4 a = acc[ct];
          const bf16x2 lo = __builtin_convertvector(f32x2{a[0] + rs * bflo(xv[0]), a[1] + rs * bfhi(xv[0])}, bf16x2);      // (one v_cvt_pk_bf16_f32 per pair)
          const bf16x2 hi = __builtin_convertvector(f32x2{a[2] + rs * bflo(xv[1]), a[3] + rs * bfhi(xv[1])}, bf16x2);
          *(u32x2*)(out + step * ct) = u32x2{__builtin_bit_cast(unsigned int, lo), __builtin_bit_cast(unsigned int, hi)};
        }
      }
    }
  }
}

}  // namespace

// 0 = launched, 1 = shape not served (the caller runs the dX product and site B's dY product separately), < 0 error
int k_dx_stream3(const void* X, long ldx, const void* dZx, long ldz, const void* dL2, long ldl, int K2, const float* rs, const void* Wt, long ldw, long sWg,
                 const void* Text, long ldt, long sT1, const void* Bm, long ldb, long sB1, int KB, const void* dRT, long ldr, const void* dV, long ldv, long sV1,
                 const void* Q, long ldq, int KQ, void* dX, long ldc, void* dump, int S, int N, int G, int Cg, int K1, hipStream_t st) {
  if (Cg != 384 || K1 != 128 || K2 < 1 || K2 > 72 || ldl < 72 || KB < 1 || KB > 96 || KQ < 1 || KQ > 64 || ldr < 8 || ldb < N || N < 16 || S < 1 ||      // (N < 16: a ragged tile of a non-final frame would read 32 - 2 N rows past its successor)
     
      ldx % 8 || ldz % 8 || ldl % 8 || ldr % 8 || ldb % 8 || sB1 % 8 || ldc % 4 || !rs || !dump ||
      ((uintptr_t)X % 16) || ((uintptr_t)dZx % 16) || ((uintptr_t)dL2 % 16) || ((uintptr_t)dRT % 16) || ((uintptr_t)Bm % 16) || ((uintptr_t)dX % 8) ||
      ((uintptr_t)rs % 4) || ((uintptr_t)dump % 16) || (long)S * N < 2048)
    return 1;
  const int cus = cu_count();
  if (cus <= 0) { set_last_error("dx_stream3: device query"); return ERR_LAUNCH; }
  DX3Args p;
  p.X = (const char*)X; p.ldx = ldx; p.dZx = (const char*)dZx; p.ldz = ldz; p.dL2 = (const char*)dL2; p.ldl = ldl; p.rs = rs;
  p.Wt = (const unsigned short*)Wt; p.ldw = ldw; p.sWg = sWg; p.Text = (const unsigned short*)Text; p.ldt = ldt; p.sT1 = sT1;
  p.Bm = (const char*)Bm; p.ldb = ldb; p.sB1 = sB1; p.dRT = (const char*)dRT; p.ldr = ldr; p.dV = (const unsigned short*)dV; p.ldv = ldv; p.sV1 = sV1;
  p.Q = (const unsigned short*)Q; p.ldq = ldq;
  p.dX = (char*)dX; p.ldc = ldc; p.dump = (char*)dump; p.N = N; p.tps = (N + BM - 1) / BM; p.ntiles = S * p.tps; p.K2 = K2; p.KB = KB; p.KQ = KQ;
  p.ncr = (int)std::min<long>(8, ldr / 8);
  const int gx = std::min(std::max(1, cus / G), p.ntiles);
  static LdsAttrOnce attr;
  AVMOE_TRY(attr.ensure((const void*)kk_dx_stream3, DX3_LDS, "dx_stream3"));
  const double ntok = (double)S * N;
  // UNIQUE algorithmic bytes: X, dX (384 channels x 2 bytes each per group), dZx (128 per group), one dL2 row + its row scale, site B's operands
  // (dR^T: 64 values, [Bm ; wbar]: KB values per token) ONCE -- each channel group's block fetches them again (PMC: 4 236 B per token at cfg-2)
  const double bytes = ntok * G * (384.0 * 2 * 2 + 128.0 * 2) + ntok * (ldl * 2.0 + 4.0) + ntok * (64.0 + KB) * 2.0;
  ProfScope ps("k_dx_stream3", (long)ntok, bytes, 2.0 * ntok * G * 384.0 * (128 + K2 + KB + KQ), st);
  hipLaunchKernelGGL(kk_dx_stream3, dim3((unsigned)gx, (unsigned)G), dim3(NTHR), DX3_LDS, st, p);
  AVMOE_CHECK_LAUNCH("dx_stream3");
  return OK;
}

}  // namespace avmoe
