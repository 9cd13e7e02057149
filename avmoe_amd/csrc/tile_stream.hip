// Bottleneck-space kernels of the tuned shape (bottleneck 64 in 2 groups, 32 latent tokens: tile_fast.hip), STREAMING form (round 6).
//
// Same arithmetic, lane layout and per-block partial sums as tile_fast.hip (a wave owns 16 tokens of one expert; lane (r, q) holds the
// entries 16 c + 4 q + x of token r), different memory side.  tile_fast.hip's blocks (one per 112 - 256 tokens of a frame, three or four
// per CU) request a tile's rows into registers, compute, store -- one or two tiles of loads in flight per wave for part of the time, and
// 1280 + blocks whose prologues and tails nothing hides.  Here:
//   * ONE persistent block per CU (8 waves = tile slots x experts) walks a contiguous range of tile_fast's blocks ("virtual blocks": the
//     per-block partial sums keep their layout, so every finishing kernel is unchanged), its per-site constants filled once;
//   * every wave has a PRIVATE ring of KFS_P + 1 tile slots in the LDS and requests its own rows KFS_P tiles ahead with
//     global_load_lds (per-lane source address = exactly the segments it loads into registers today, lane-linear image = exactly the
//     register layout, read back with one ds_read_b128 per 16 bytes): no staging registers, no cross-wave hand-over, NO barrier in the
//     tile loop; the 4-byte streams (row statistics, the scalar columns of dApost) ride in the same ring as dword loads;
//   * waits are counted (s_waitcnt vmcnt(loads of the newer tiles + the stores issued since)): a tile's stores are never waited for
//     by the next tile's loads, and ~74 KB are in flight per CU all the time (scripts/lds_stream_probe.hip: 32 KB suffice for the
//     tile stream alone; the stores are what needs the depth);
//   * no ordinary global load inside the loop (hipcc answers one with s_waitcnt vmcnt(0), which would drain the ring): per-frame
//     scalars are put into the LDS by the prologue.
// scripts/hbm_probe.hip measured the memory side of these passes (64-byte segments, 2 reads : 1 write) at 5.6 TB/s with one block per
// CU against 4.3 - 4.5 with 1024 - 4096 blocks; the register-resident kernels reach 3.3 - 4.2.
#include "kernels.h"
#include "device_utils.h"
#include "prof.h"
#include <algorithm>

#ifndef KFS_AUX
#define KFS_AUX 0              // cache policy of the ring's loads (common.h: 2 = non-temporal)
#endif
#ifndef KFS_P
#define KFS_P 2                // tiles requested ahead per wave
#endif
#ifndef KFS_DISSECT
#define KFS_DISSECT 0          // development builds (timing only): bit 0 = no arithmetic between unpack and pack, 1 = no stores, 2 = no ring loads, 4 = raw copy (no unpack / pack either)
#endif
#ifndef KFS_EXACT
#define KFS_EXACT 1            // 1: the waits count the stores issued since (never wait for a store); 0: loads only (conservative)
#endif

namespace avmoe {

#include "tile_fast_dev.h"

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N < 63 ? N : 63) : "memory"); }
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void glds16(const void* g, char* l) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, KFS_AUX); }
__device__ __forceinline__ void glds4(const void* g, char* l) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 4, 0, KFS_AUX); }

// waves of a streaming block: tile slots x experts, 8 waves (6 for three experts)
template <int E> struct WS {
  static constexpr int NS = (E == 3) ? 2 : 8 / E;
  static constexpr int NW = E * NS, NTHR = 64 * NW;
};

// a wave's walk over ITS tiles of the block's virtual blocks vb0 .. vb1 - 1 (tile_fast's blocks: frame s = vb / bps, tokens
// [b per, min(N, b per + per)) with b = vb % bps): tile slot ts takes the tiles ts, ts + NS, ... of every virtual block
struct TileIt {
  int vb, s, b, n0;        // virtual block, its frame and index inside the frame (kept incrementally: no divisions in the loop), first token of the tile
};
template <int NS>
__device__ __forceinline__ void it_settle(TileIt& it, int vb1, int bps, int per, int N, int ts) {
  while (it.vb < vb1) {
    if (it.n0 < min(N, it.b * per + per)) return;
    ++it.vb;
    if (++it.b == bps) { it.b = 0; ++it.s; }
    it.n0 = it.b * per + 16 * ts;
  }
}
template <int NS>
__device__ __forceinline__ TileIt it_first(int vb0, int vb1, int bps, int per, int N, int ts) {
  TileIt it{vb0, vb0 / bps, vb0 % bps, (vb0 % bps) * per + 16 * ts};
  it_settle<NS>(it, vb1, bps, per, N, ts);
  return it;
}
template <int NS>
__device__ __forceinline__ void it_next(TileIt& it, int vb1, int bps, int per, int N, int ts) {
  it.n0 += 16 * NS;
  it_settle<NS>(it, vb1, bps, per, N, ts);
}

// ---- LDS accesses of the tile loops as inline assembly ---------------------------------------------------------------------------
// hipcc answers EVERY LDS access it can see that follows a direct global -> LDS load with s_waitcnt vmcnt(0) (it cannot tell the ring from
// the rest of the LDS), which would drain the ring once per tile: after the prologue no C++-level LDS access exists in these kernels.
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(lptr_t)p; }
template <int OFF> __device__ __forceinline__ void lds_rd16(f32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory"); }
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// ... and the registers such reads fill pass through an (empty) asm statement behind the wait: volatile asm statements keep their order, so no
// use of them is scheduled between the read and the wait
__device__ __forceinline__ void lds_use(f32x4& a) { asm volatile("" : "+v"(a)); }
template <typename... R> __device__ __forceinline__ void lds_use(f32x4& a, R&... rest) { lds_use(a); lds_use(rest...); }
__device__ __forceinline__ float lds_rd1(unsigned addr) { float v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); return v; }
__device__ __forceinline__ void lds_wr16(unsigned addr, const float4& v) {
  const f32x4 x = {v.x, v.y, v.z, v.w};
  asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(x) : "memory");
}
__device__ __forceinline__ void lds_wr1(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ uint4 as_u4(const f32x4& v) { return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])); }

// wait until this wave's tile `it` has landed: `newer` tiles were requested after it, `stored` tiles' stores were issued since its request
// (in-order counter: everything but those may still be in flight; whatever else was issued in between -- a flush's stores, the next frame's
// constants -- only makes the wait stricter).  NL / NST: ring loads / stores per wave and tile.
template <int NL, int NST>
__device__ __forceinline__ void wait_tile(int newer, int stored) {
  if (!KFS_EXACT) stored = 0;
#define KFS_W(M_, K_) if (newer == M_ && stored == K_) { wait_vm<M_ * NL + K_ * NST>(); return; }
  KFS_W(0, 1) KFS_W(0, 2) KFS_W(0, 3)
  KFS_W(1, 0) KFS_W(1, 1) KFS_W(1, 2) KFS_W(1, 3)
  KFS_W(2, 0) KFS_W(2, 1) KFS_W(2, 2) KFS_W(2, 3)
#undef KFS_W
  wait_vm<0>();
}

// The per-virtual-block partial sums of a wave -- NC column accumulators of 64 entries and up to 4 scalars -- go through the LDS once:
// [wave][NC * 64 + 4] floats, one barrier, then the waves of tile slot 0 add the tile slots of their expert in slot order.
template <int NC> constexpr int fold_stride() { return NC * FDD + 4; }
template <int NC>
__device__ __forceinline__ void fold_put_cols(unsigned s_x, int k, float4 (&acc)[4], int wave, int lane) {
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float4 v;
#pragma unroll
    for (int x = 0; x < 4; ++x) at(v, x) = rsum16(at(acc[c], x));
    if (r == 0) lds_wr16(s_x + 4 * (wave * fold_stride<NC>() + k * FDD + 16 * c + 4 * q), v);
  }
}
template <int NC>
__device__ __forceinline__ void fold_put_scalar(unsigned s_x, int k, float v, int wave, int lane) {
  if (lane == 0) lds_wr1(s_x + 4 * (wave * fold_stride<NC>() + NC * FDD + k), v);
}
__device__ __forceinline__ void lds_rd1_issue(float& v, unsigned addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory"); }
__device__ __forceinline__ void lds_use(float& a) { asm volatile("" : "+v"(a)); }
// the NV values idx[0 .. NV) of expert e summed over the tile slots in slot order: every read issued, one wait   (idx: k * 64 + dd of a column,
// NC * 64 + k of a scalar)
template <int E, int NS, int NC, int NV>
__device__ __forceinline__ void fold_get(unsigned s_x, const int (&idx)[NV], int e, float (&out)[NV]) {
  float v[NV][NS];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int u = 0; u < NS; ++u) lds_rd1_issue(v[i][u], s_x + 4 * ((u * E + e) * fold_stride<NC>() + idx[i]));
  lds_wait();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < NS; ++u) { lds_use(v[i][u]); a += v[i][u]; }
    out[i] = a;
  }
}

// =====================================================================================================
// POST_SMALL backward (bf16, split dApost, Gram mode)      tile_fast.hip::kf_post_small_bwd<__bf16, E, true> with dSooT
//   + the weighted Gram products dG[i][e] = sum_t dSoo[t] z'[t] z'[t]^T of gram.hip::kg_gram64<true, ., true> in the same pass: the wave
//   that computes a token's dSoo holds that token's z' -- the separate pass over Z (and dSooT, which nobody else reads) is gone.
//   z' goes registers -> a wave-private LDS tile (bf16, token-major) -> ds_read_b64_tr_b16 (tokens become the contraction index) ->
//   v_mfma_f32_16x16x16_bf16 (16 tokens per tile); same operand roundings as gram.hip (z' and dSoo z' in bf16, fp32 accumulation).
// =====================================================================================================
struct SPostBArgs { P16 gate; int relu_of_e[MAX_E]; FastDims t; int ln_post, use_gate, bps, nvb, nfr; const float* dApx; int dapw; float* gpart; };

constexpr int PSB_NL = 6, PSB_NST = 2;                      // ring loads / stores per wave and tile
constexpr int PSB_TILE = 4 * 1024 + 2 * 256;                // bytes of a wave's tile slot: Z (2 x 1 KB), dApost (2 x 1 KB), 2 x 64 scalars
constexpr int PSB_ZPITCH = 2 * FDD + 16;                    // bytes of a token row of the wave's z' tile (bf16)
constexpr int PSB_GRAM = 16 * PSB_ZPITCH + 64;              // ... + the tile's 16 dSoo
template <int E> constexpr int psb_fold_floats() { return 2 * WS<E>::NW * fold_stride<2>(); }        // two buffers: one barrier per virtual block
template <int E> constexpr int psb_fixed_floats(int nfr) { return psb_fold_floats<E>() + E * 4 * FDD + ((nfr * E + 3) & ~3); }
template <int E> constexpr size_t psb_lds(int nfr) { return (size_t)psb_fixed_floats<E>(nfr) * 4 + (size_t)WS<E>::NW * (PSB_GRAM + (KFS_P + 1) * PSB_TILE); }

// one transposed mat-vec step of mmT_split with the matrix operand split into bf16 planes ONCE (registers, whole kernel)
__device__ __forceinline__ f32x4 mm_presplit(const kf_bf16x8& ah, const kf_bf16x8& al, const float4& p0, const float4& p1) {
  kf_bf16x8 ph, pl;
  kf_split8(p0, p1, ph, pl);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, ph, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, pl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ph, acc, 0, 0, 0);
  return acc;
}
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
template <int OFF> __device__ __forceinline__ void lds_rd_tr(u32x2& d, unsigned addr) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory"); }
__device__ __forceinline__ void lds_wr8(unsigned addr, const u32x2& v) { asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_use(u32x2& a) { asm volatile("" : "+v"(a)); }

template <int E>
__global__ void __launch_bounds__(WS<E>::NTHR, 2) kfs_post_small_bwd(SPostBArgs a, const unsigned short* __restrict__ Z, const float* __restrict__ bn1, const float* __restrict__ Gq,
                                                                  const float* __restrict__ uvh, const float* __restrict__ probs, const float* __restrict__ rpmup,
                                                                  const unsigned short* __restrict__ dAp16, unsigned short* __restrict__ dzp, float* __restrict__ colpart,
                                                                  float* __restrict__ blkscal) {
  constexpr int DZ = E * FDD, NS = WS<E>::NS, NW = WS<E>::NW, NTHR = WS<E>::NTHR, D = KFS_P + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_fold = (float*)smem;                     // [2][NW][2 * 64 + 4]  the partial sums of a virtual block, two buffers
  float* s_uv = s_fold + psb_fold_floats<E>();      // [E][us (64) | vh (64) | sc (64) | sh (64)]: LayerNorm-post row constants, BatchNorm-1 scale / shift
  float* s_qv = s_uv + E * 4 * FDD;                 // [nfr][E]             probs * gate of the block's frames
  const FastDims& t = a.t;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int e = wave % E, ts = wave / E;
  char* gram = smem + (size_t)psb_fixed_floats<E>(a.nfr) * 4 + (size_t)wave * PSB_GRAM;
  char* ring = smem + (size_t)psb_fixed_floats<E>(a.nfr) * 4 + (size_t)NW * PSB_GRAM + (size_t)wave * D * PSB_TILE;
  const unsigned ring_a = lds_off(ring), fold_a = lds_off(s_fold), qv_a = lds_off(s_qv), uv_a = lds_off(s_uv + e * 4 * FDD), gram_a = lds_off(gram);
  const int vb0 = (int)((long)a.nvb * blockIdx.x / gridDim.x), vb1 = (int)((long)a.nvb * (blockIdx.x + 1) / gridDim.x);
  const int bps = a.bps, per = t.per, N = t.N;
  const int s_first = vb0 / bps;

  // ---- the few LDS constants (C++-level stores: before the ring starts) ----
  for (int i = threadIdx.x; i < a.nfr * E; i += NTHR) {
    const int ee = i % E, s = min(s_first + i / E, t.S - 1);
    s_qv[i] = probs[(long)s * E + ee] * (a.use_gate ? a.gate.p[ee][0] : 1.f);
  }
  for (int i = threadIdx.x; i < E * FDD; i += NTHR) {
    const int ee = i >> 6, dd = i & 63, col = (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31);
    float* u = s_uv + ee * 4 * FDD;
    u[dd] = uvh[col]; u[FDD + dd] = uvh[DZ + col]; u[2 * FDD + dd] = bn1[2 * DZ + col]; u[3 * FDD + dd] = bn1[3 * DZ + col];
  }
  __syncthreads();                 // (no C++-level LDS access from here on)

  // this wave's rows of tile (vb, n0) -> ring slot `slot`; the eight 4-byte values of a token as 2 x 4 consecutive floats (lane -> token
  // lane >> 2, value lane & 3: [dApx g0 (3) | rp] and [dApx g1 (3) | mup])
  auto request = [&](int slot, const TileIt& it) {
    if (KFS_DISSECT & 4) return;
    const long f0 = (long)it.s * N;
    const long tk = f0 + min(it.n0 + r, N - 1);      // (rows past the frame's last token re-read it; masked below)
    char* dst = ring + slot * PSB_TILE;
    const char* zs = (const char*)(Z + tk * DZ + e * FDG + seg_off8(q));
    glds16(zs, dst); glds16(zs + E * FDG * 2, dst + 1024);
    const char* ds = (const char*)(dAp16 + tk * 2 * a.dapw + e * FDG + seg_off8(q));
    glds16(ds, dst + 2048); glds16(ds + a.dapw * 2, dst + 3072);
    const long t2 = f0 + min(it.n0 + (lane >> 2), N - 1);
    const int j = lane & 3;
    const float* pa = j < 3 ? a.dApx + (t2 * 2) * 16 + 3 * e + j : rpmup + (long)e * t.NT + t2;
    const float* pb = j < 3 ? a.dApx + (t2 * 2 + 1) * 16 + 3 * e + j : rpmup + (long)t.NT * E + (long)e * t.NT + t2;
    glds4(pa, dst + 4096); glds4(pb, dst + 4352);
  };
  TileIt pf = it_first<NS>(vb0, vb1, bps, per, N, ts);
  int nreq = 0;
#pragma unroll
  for (int k = 0; k < KFS_P; ++k)
    if (pf.vb < vb1) { request(nreq % D, pf); ++nreq; it_next<NS>(pf, vb1, bps, per, N, ts); }

  // ---- per-expert constants, registers for the whole kernel (the register-resident kernels re-read them from the LDS per tile); their
  // loads fly beside the first requests (hipcc waits for both together: once per block) ----
  kf_bf16x8 gh[2][2], gl[2][2];                     // Gq^T rows 16 ct + r of group gi, entries 4 q .. + 3 and 16 + 4 q .. + 3, as bf16 planes
#pragma unroll
  for (int gi = 0; gi < 2; ++gi)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const float* g = Gq + (long)(gi * E + e) * FDG * FDG + 16 * ct + r;           // Gq[k][c]: the transposed operand reads a column
      float4 m0, m1;
#pragma unroll
      for (int x = 0; x < 4; ++x) { at(m0, x) = g[(4 * q + x) * FDG]; at(m1, x) = g[(16 + 4 * q + x) * FDG]; }
      kf_split8(m0, m1, gh[gi][ct], gl[gi][ct]);
    }
  f32x4 gacc[2][3];                                 // weighted Gram of this wave's tokens: [group][16 x 16 tiles (0,0), (0,1), (1,1)] -- symmetric: (1,0) = (0,1)^T
#pragma unroll
  for (int i = 0; i < 6; ++i) (&gacc[0][0])[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool relu = a.relu_of_e[e];
  const float fC = (float)t.C;
  int it = 0;                       // tiles this wave has computed
  int s = s_first, vbb = vb0 - s_first * bps;      // frame / index inside the frame of the virtual block
  // Partial sums.  The finishing kernels need: the column sums and dSo / dSoo totals over ALL blocks' rows, dq per FRAME (sum over the frame's
  // rows).  So the column accumulators and sdSo / sdSoo run over the block's whole range and are folded ONCE, into the rows of its last
  // virtual block; sdq is folded when the frame ends (or the range does), into that virtual block's row; every other row gets zeros.  One
  // barrier per frame instead of one per virtual block, and the fold's ~600 VALU instructions once per block.
  float sdq = 0.f, sdSo = 0.f, sdSoo = 0.f;
  float4 cs0[4], cs1[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { cs0[c] = zero4(); cs1[c] = zero4(); }
  int nfold = 0;
  for (int vb = vb0; vb < vb1; ++vb, s += (vbb + 1 == bps), vbb = (vbb + 1 == bps) ? 0 : vbb + 1) {
    const int n_beg = vbb * per, n_end = min(N, n_beg + per);
    const float qv = lds_rd1(qv_a + 4 * ((s - s_first) * E + e));
    int qsel = 0;      // the lane (of the four that hold a token) that takes this tile's scalar terms: four times as many partial sums
    for (int n0 = n_beg + 16 * ts; n0 < n_end; n0 += 16 * NS, qsel = (qsel + 1) & 3, ++it) {
      // In-order counter; issue order per tile i: [wait for tile i] [requests of tile i + P] [stores of tile i]
      wait_tile<PSB_NL, PSB_NST>(min(KFS_P - 1, nreq - it - 1), min(it, KFS_P));
      if (pf.vb < vb1) { request(nreq % D, pf); ++nreq; it_next<NS>(pf, vb1, bps, per, N, ts); }
      const bool ok = n0 + r < N;
      const long tok = (long)s * N + n0 + r;
      const unsigned sl = ring_a + (it % D) * PSB_TILE;
      f32x4 rz0, rz1, rd0, rd1, ra, rb;
      lds_rd16<0>(rz0, sl + lane * 16); lds_rd16<1024>(rz1, sl + lane * 16); lds_rd16<2048>(rd0, sl + lane * 16); lds_rd16<3072>(rd1, sl + lane * 16);
      lds_rd16<4096>(ra, sl + r * 16); lds_rd16<4352>(rb, sl + r * 16);
      f32x4 sc[4], sh[4];                            // (row constants of this expert: from the LDS per tile, like us / vh below -- the registers are the Gram's)
      lds_rd16<512>(sc[0], uv_a + 16 * q); lds_rd16<576>(sc[1], uv_a + 16 * q); lds_rd16<640>(sc[2], uv_a + 16 * q); lds_rd16<704>(sc[3], uv_a + 16 * q);
      lds_rd16<768>(sh[0], uv_a + 16 * q); lds_rd16<832>(sh[1], uv_a + 16 * q); lds_rd16<896>(sh[2], uv_a + 16 * q); lds_rd16<960>(sh[3], uv_a + 16 * q);
      lds_wait();
      lds_use(rz0, rz1, rd0, rd1, ra, rb);
      lds_use(sc[0], sc[1], sc[2], sc[3]); lds_use(sh[0], sh[1], sh[2], sh[3]);
#if KFS_DISSECT & 16
      if (ok && !(KFS_DISSECT & 2)) {
        unsigned short* o = dzp + tok * DZ + e * FDG + seg_off8(q);
        *(uint4*)o = as_u4(rz0 + rd0); *(uint4*)(o + E * FDG) = as_u4(rz1 + rd1);
      }
      continue;
#endif
      float4 zp[4], zraw[4], dzo[4];
      unpack_seg(as_u4(rz0), zraw[0], zraw[1]); unpack_seg(as_u4(rz1), zraw[2], zraw[3]);
      const float da1 = ra[0] + rb[0], da2 = ra[1] + rb[1], da3 = ra[2] + rb[2];
      const float rp = ok ? ra[3] : 1.f, mup = ok ? rb[3] : 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float y = at(zraw[c], x) * sc[c][x] + sh[c][x];
          at(zp[c], x) = relu ? fmaxf(y, 0.f) : y;
        }
      float zz = 0.f;
      {
        float4 d[4];                                 // (dApost stays packed in between: registers)
        unpack_seg(as_u4(rd0), d[0], d[1]); unpack_seg(as_u4(rd1), d[2], d[3]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int x = 0; x < 4; ++x) zz += at(d[c], x) * at(zp[c], x);
      }
      zz = qsum4(zz);
      float dSo = 0.f, dSoo = 0.f;
      if (ok) {
        const float dq = rp * zz + rp * da1 - rp * mup * da2 + da3;
        if (a.ln_post) {
          const float drp = qv * zz + qv * da1 - qv * mup * da2;
          float dmup = -qv * rp * da2;
          const float dvarp = drp * (-0.5f) * rp * rp * rp;
          dSoo = dvarp / fC;                         // (divisions as in tile_fast.hip: the two forms agree bit for bit per token)
          dmup -= 2.f * mup * dvarp;
          dSo = dmup / fC;
        }
        if (q == qsel) { sdq += dq; sdSo += dSo; sdSoo += dSoo; }
      }
      const float k1 = qv * rp;
      float4 d[4];
      asm volatile("" : "+v"(rd0), "+v"(rd1));       // (keeps the second unpack from being merged with the first)
      unpack_seg(as_u4(rd0), d[0], d[1]); unpack_seg(as_u4(rd1), d[2], d[3]);
      if (a.ln_post) {
        // ---- weighted Gram: z' (bf16, as gram.hip rounds it) token-major into this wave's LDS tile, the tile's dSoo beside it ----
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const u32x2 pk = {(unsigned)f2bf(zp[c].x) | ((unsigned)f2bf(zp[c].y) << 16), (unsigned)f2bf(zp[c].z) | ((unsigned)f2bf(zp[c].w) << 16)};
          lds_wr8(gram_a + r * PSB_ZPITCH + (16 * c + 4 * q) * 2, pk);
        }
        if (q == 0) lds_wr1(gram_a + 16 * PSB_ZPITCH + 4 * r, dSoo);
        f32x4 us[4], vh[4];                          // (this expert's us / vh rows: read here, used by the last loop of the tile)
        lds_rd16<0>(us[0], uv_a + 16 * q); lds_rd16<64>(us[1], uv_a + 16 * q); lds_rd16<128>(us[2], uv_a + 16 * q); lds_rd16<192>(us[3], uv_a + 16 * q);
        lds_rd16<256>(vh[0], uv_a + 16 * q); lds_rd16<320>(vh[1], uv_a + 16 * q); lds_rd16<384>(vh[2], uv_a + 16 * q); lds_rd16<448>(vh[3], uv_a + 16 * q);
        // fragments: lane (r, q) gets [token 4 q + j][column 16 ct + r], j = 0 .. 3 (the lanes of a q group read a 4 x 16 block, transposed)
        u32x2 f[4];
        f32x4 w4;
        const unsigned fa0 = gram_a + (4 * q + (r >> 2)) * PSB_ZPITCH + 8 * (r & 3);
        lds_wait();                                  // (the tile's writes above: same wave, in order -- the wait covers the uv reads too)
        lds_rd_tr<0>(f[0], fa0); lds_rd_tr<32>(f[1], fa0); lds_rd_tr<64>(f[2], fa0); lds_rd_tr<96>(f[3], fa0);
        lds_rd16<16 * PSB_ZPITCH>(w4, gram_a + 16 * q);
        lds_wait();
        lds_use(f[0]); lds_use(f[1]); lds_use(f[2]); lds_use(f[3]); lds_use(w4);
        lds_use(us[0], us[1], us[2], us[3]); lds_use(vh[0], vh[1], vh[2], vh[3]);
        s16x4 fb[4], fw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fb[i] = __builtin_bit_cast(s16x4, f[i]);
          const float v0 = __uint_as_float(f[i][0] << 16) * w4[0], v1 = __uint_as_float(f[i][0] & 0xffff0000u) * w4[1];
          const float v2 = __uint_as_float(f[i][1] << 16) * w4[2], v3 = __uint_as_float(f[i][1] & 0xffff0000u) * w4[3];
          const u32x2 pw = {(unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16), (unsigned)f2bf(v2) | ((unsigned)f2bf(v3) << 16)};
          fw[i] = __builtin_bit_cast(s16x4, pw);
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          gacc[gi][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fw[2 * gi], fb[2 * gi], gacc[gi][0], 0, 0, 0);
          gacc[gi][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fw[2 * gi], fb[2 * gi + 1], gacc[gi][1], 0, 0, 0);
          gacc[gi][2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(fw[2 * gi + 1], fb[2 * gi + 1], gacc[gi][2], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int gi = c >> 1, ct = c & 1;
          const f32x4 w = mm_presplit(gh[gi][ct], gl[gi][ct], zp[2 * gi], zp[2 * gi + 1]);
          float4 o;
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float zv = at(zp[c], x);
            at(o, x) = k1 * at(d[c], x) + dSo * us[c][x] + dSoo * (2.f * w[x] + 2.f * vh[c][x]);
            at(cs0[c], x) += dSo * zv; at(cs1[c], x) += dSoo * zv;
          }
          dzo[c] = o;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) dzo[c] = make_float4(k1 * d[c].x, k1 * d[c].y, k1 * d[c].z, k1 * d[c].w);
      }
      // exactly PSB_NST store instructions per tile (lane r = 0 of a tile in range is valid: none of them is skipped)
      if (ok && !(KFS_DISSECT & 2)) st_row<__bf16, E>((__bf16*)dzp + tok * DZ, e, q, dzo);
    }
    const bool final = vb + 1 == vb1, fend = final || vbb + 1 == bps;
    const int dd = lane, col = (dd >> 5) * (E * FDG) + e * FDG + (dd & 31);
    if (fend) {        // two LDS buffers, one barrier per fold: buffer k & 1 is rewritten two folds later, behind the next barrier
      const unsigned fb_a = fold_a + 4 * (nfold & 1) * NW * fold_stride<2>();
      ++nfold;
      if (final) {
        fold_put_cols<2>(fb_a, 0, cs0, wave, lane); fold_put_cols<2>(fb_a, 1, cs1, wave, lane);
        fold_put_scalar<2>(fb_a, 1, wave_sum(sdSo), wave, lane); fold_put_scalar<2>(fb_a, 2, wave_sum(sdSoo), wave, lane);
      }
      fold_put_scalar<2>(fb_a, 0, wave_sum(sdq), wave, lane);
      sdq = 0.f;
      lds_barrier();
      if (ts == 0) {
        const int idx[3] = {dd, FDD + dd, 2 * FDD + min(lane, 2)};
        float v[3];
        fold_get<E, NS, 2, 3>(fb_a, idx, e, v);
        colpart[((long)vb * 4 + 0) * (E * FDD) + col] = final ? v[0] : 0.f;
        colpart[((long)vb * 4 + 1) * (E * FDD) + col] = final ? v[1] : 0.f;
        if (lane < 3) blkscal[((long)vb * E + e) * 4 + lane] = (final || lane == 0) ? v[2] : 0.f;
      }
    } else if (ts == 0) {
      colpart[((long)vb * 4 + 0) * (E * FDD) + col] = 0.f;
      colpart[((long)vb * 4 + 1) * (E * FDD) + col] = 0.f;
      if (lane < 3) blkscal[((long)vb * E + e) * 4 + lane] = 0.f;
    }
  }
  // the Gram partial sums of this wave (all its tiles): slab blockIdx * NS + ts, matrices [group * E + e][32][32] (gram.hip's layout: lane
  // (r, q) holds rows 4 q + x, column r of each 16 x 16 tile); summed over the slabs by the deterministic column-sum kernel
  if (a.ln_post) {
    float* out = a.gpart + ((long)blockIdx.x * NS + ts) * (2 * E * FDG * FDG);
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      float* o = out + (long)(gi * E + e) * FDG * FDG;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        o[(4 * q + x) * FDG + r] = gacc[gi][0][x];
        o[(4 * q + x) * FDG + 16 + r] = gacc[gi][1][x];
        o[(16 + r) * FDG + 4 * q + x] = gacc[gi][1][x];          // the mirror tile
        o[(16 + 4 * q + x) * FDG + 16 + r] = gacc[gi][2][x];
      }
    }
  }
}

template <int E>
int launch_psb(const SPostBArgs& a, int gx, const Plan& pl, char* saved, char* scratch, hipStream_t st) {
  const size_t lds = psb_lds<E>(a.nfr);
  if (lds > 160 * 1024) { set_last_error("post_small_bwd (streaming): LDS budget"); return ERR_UNSUPPORTED; }      // (psb_geom said otherwise)
  static LdsAttrOnce attr;
  AVMOE_TRY(attr.ensure((const void*)kfs_post_small_bwd<E>, 160 * 1024, "post_small_bwd (streaming)"));
  // shared_gpu (include/avmoe.h): the whole LDS of the CU, so that no block of another stream's kernel fits beside this one
  hipLaunchKernelGGL((kfs_post_small_bwd<E>), dim3((unsigned)gx), dim3(WS<E>::NTHR), pl.d.excl ? (size_t)160 * 1024 : lds, st, a, (const unsigned short*)(saved + pl.o_Z),
                     (const float*)(saved + pl.o_bn1), (const float*)(saved + pl.o_Gq), (const float*)(saved + pl.o_uvh), (const float*)(saved + pl.o_probs),
                     (const float*)(saved + pl.o_rpmup), (const unsigned short*)(scratch + pl.o_dAp), (unsigned short*)(scratch + pl.o_dzp), (float*)(scratch + pl.o_colpart),
                     (float*)(scratch + pl.o_blkscal));
  AVMOE_CHECK_LAUNCH("post_small_bwd (streaming)");
  if (a.ln_post) {                 // dGq = the sum of the slabs
    const int ncol = 2 * E * FDG * FDG;
    return k_colsum_f32(a.gpart, (long)gx * WS<E>::NS, ncol, ncol, 1, 0, (float*)(scratch + pl.o_dGq), 0, 1.f, st);
  }
  return OK;
}

// =====================================================================================================
// MID backward (bf16)      tile_fast.hip::kf_mid_bwd<__bf16, E>: BN2-moment terms + BatchNorm-1 / ReLU mask, dz' rewritten in place
// =====================================================================================================
struct SMidBArgs { int relu_of_e[MAX_E]; FastDims t; int moments, bps, nvb; };

constexpr int MDB_NL = 4, MDB_NST = 2;                      // ring loads / stores per wave and tile
constexpr int MDB_TILE = 4 * 1024;                          // Z (2 x 1 KB), dz' (2 x 1 KB)
template <int E> constexpr int mdb_fixed_floats() { return 2 * WS<E>::NW * fold_stride<2>() + E * 5 * FDD; }
template <int E> constexpr size_t mdb_lds() { return (size_t)mdb_fixed_floats<E>() * 4 + (size_t)WS<E>::NW * (KFS_P + 1) * MDB_TILE; }

template <int E>
__global__ void __launch_bounds__(WS<E>::NTHR, 2) kfs_mid_bwd(SMidBArgs a, const unsigned short* __restrict__ Z, const float* __restrict__ bn1, const float* __restrict__ dsm,
                                                           const float* __restrict__ sdSzz, unsigned short* __restrict__ dzp, float* __restrict__ colpart) {
  constexpr int DZ = E * FDD, NS = WS<E>::NS, NW = WS<E>::NW, NTHR = WS<E>::NTHR, D = KFS_P + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_fold = (float*)smem;                             // [2][NW][2 * 64 + 4]
  float* s_bn = s_fold + 2 * NW * fold_stride<2>();         // [E][mean | rstd | sc | sh | dm]
  const FastDims& t = a.t;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int e = wave % E, ts = wave / E;
  char* ring = smem + (size_t)mdb_fixed_floats<E>() * 4 + (size_t)wave * D * MDB_TILE;
  const unsigned ring_a = lds_off(ring), fold_a = lds_off(s_fold), bn_a = lds_off(s_bn + e * 5 * FDD);
  const int vb0 = (int)((long)a.nvb * blockIdx.x / gridDim.x), vb1 = (int)((long)a.nvb * (blockIdx.x + 1) / gridDim.x);
  const int bps = a.bps, per = t.per, N = t.N;
  for (int i = threadIdx.x; i < E * FDD; i += NTHR) {
    const int ee = i >> 6, dd = i & 63, col = (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31);
    float* b = s_bn + ee * 5 * FDD;
    b[dd] = bn1[col]; b[FDD + dd] = bn1[DZ + col]; b[2 * FDD + dd] = bn1[2 * DZ + col];
    b[3 * FDD + dd] = bn1[3 * DZ + col]; b[4 * FDD + dd] = a.moments ? dsm[2 * DZ + col] : 0.f;
  }
  __syncthreads();                 // (no C++-level LDS access from here on)
  auto request = [&](int slot, const TileIt& it) {
    const long tk = (long)it.s * N + min(it.n0 + r, N - 1);
    char* dst = ring + slot * MDB_TILE;
    const char* zs = (const char*)(Z + tk * DZ + e * FDG + seg_off8(q));
    glds16(zs, dst); glds16(zs + E * FDG * 2, dst + 1024);
    const char* ds = (const char*)(dzp + tk * DZ + e * FDG + seg_off8(q));
    glds16(ds, dst + 2048); glds16(ds + E * FDG * 2, dst + 3072);
  };
  TileIt pf = it_first<NS>(vb0, vb1, bps, per, N, ts);
  int nreq = 0;
#pragma unroll
  for (int k = 0; k < KFS_P; ++k)
    if (pf.vb < vb1) { request(nreq % D, pf); ++nreq; it_next<NS>(pf, vb1, bps, per, N, ts); }
  kf_bf16x8 gh[2][2], gl[2][2];                     // sdSzz^T rows 16 ct + r of group gi as bf16 planes (mm_presplit)
#pragma unroll
  for (int gi = 0; gi < 2; ++gi)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const float* g = sdSzz + (long)(gi * E + e) * FDG * FDG + 16 * ct + r;
      float4 m0 = zero4(), m1 = zero4();
      if (a.moments) {
#pragma unroll
        for (int x = 0; x < 4; ++x) { at(m0, x) = g[(4 * q + x) * FDG]; at(m1, x) = g[(16 + 4 * q + x) * FDG]; }
      }
      kf_split8(m0, m1, gh[gi][ct], gl[gi][ct]);
    }
  const bool relu = a.relu_of_e[e];
  int it = 0;
  int s = vb0 / bps, vbb = vb0 - s * bps;
  float4 cs0[4], cs1[4];            // (column sums over the block's whole range: folded once, into the rows of its last virtual block -- see kfs_post_small_bwd)
#pragma unroll
  for (int c = 0; c < 4; ++c) { cs0[c] = zero4(); cs1[c] = zero4(); }
  for (int vb = vb0; vb < vb1; ++vb, s += (vbb + 1 == bps), vbb = (vbb + 1 == bps) ? 0 : vbb + 1) {
    const int n_beg = vbb * per, n_end = min(N, n_beg + per);
    for (int n0 = n_beg + 16 * ts; n0 < n_end; n0 += 16 * NS, ++it) {
      wait_tile<MDB_NL, MDB_NST>(min(KFS_P - 1, nreq - it - 1), min(it, KFS_P));
      if (pf.vb < vb1) { request(nreq % D, pf); ++nreq; it_next<NS>(pf, vb1, bps, per, N, ts); }
      const bool ok = n0 + r < N;
      const long tok = (long)s * N + n0 + r;
      const unsigned sl = ring_a + (it % D) * MDB_TILE;
      f32x4 rz0, rz1, rd0, rd1, sc[4], sh[4];
      lds_rd16<0>(rz0, sl + lane * 16); lds_rd16<1024>(rz1, sl + lane * 16); lds_rd16<2048>(rd0, sl + lane * 16); lds_rd16<3072>(rd1, sl + lane * 16);
      lds_rd16<512>(sc[0], bn_a + 16 * q); lds_rd16<576>(sc[1], bn_a + 16 * q); lds_rd16<640>(sc[2], bn_a + 16 * q); lds_rd16<704>(sc[3], bn_a + 16 * q);
      lds_rd16<768>(sh[0], bn_a + 16 * q); lds_rd16<832>(sh[1], bn_a + 16 * q); lds_rd16<896>(sh[2], bn_a + 16 * q); lds_rd16<960>(sh[3], bn_a + 16 * q);
      lds_wait();
      lds_use(rz0, rz1, rd0, rd1);
      lds_use(sc[0], sc[1], sc[2], sc[3]); lds_use(sh[0], sh[1], sh[2], sh[3]);
      float4 z[4], dz[4], zp[4], dyo[4];
      unpack_seg(as_u4(rz0), z[0], z[1]); unpack_seg(as_u4(rz1), z[2], z[3]);
      unpack_seg(as_u4(rd0), dz[0], dz[1]); unpack_seg(as_u4(rd1), dz[2], dz[3]);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float y = at(z[c], x) * sc[c][x] + sh[c][x];
          at(zp[c], x) = relu ? fmaxf(y, 0.f) : y;
        }
      f32x4 mean[4], rstd[4], dm[4];
      lds_rd16<0>(mean[0], bn_a + 16 * q); lds_rd16<64>(mean[1], bn_a + 16 * q); lds_rd16<128>(mean[2], bn_a + 16 * q); lds_rd16<192>(mean[3], bn_a + 16 * q);
      lds_rd16<256>(rstd[0], bn_a + 16 * q); lds_rd16<320>(rstd[1], bn_a + 16 * q); lds_rd16<384>(rstd[2], bn_a + 16 * q); lds_rd16<448>(rstd[3], bn_a + 16 * q);
      lds_rd16<1024>(dm[0], bn_a + 16 * q); lds_rd16<1088>(dm[1], bn_a + 16 * q); lds_rd16<1152>(dm[2], bn_a + 16 * q); lds_rd16<1216>(dm[3], bn_a + 16 * q);
      lds_wait();
      lds_use(mean[0], mean[1], mean[2], mean[3]); lds_use(rstd[0], rstd[1], rstd[2], rstd[3]); lds_use(dm[0], dm[1], dm[2], dm[3]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int gi = c >> 1, ct = c & 1;
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        if (a.moments) w = mm_presplit(gh[gi][ct], gl[gi][ct], zp[2 * gi], zp[2 * gi + 1]);
        float4 dy;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float zv = at(z[c], x);
          const float zh = (zv - mean[c][x]) * rstd[c][x];
          const float d = at(dz[c], x) + dm[c][x] + w[x];
          const float v = (!ok || (relu && at(zp[c], x) <= 0.f)) ? 0.f : rndT<__bf16>(d);     // as stored: the BN1 sums see the same numbers
          at(dy, x) = v;
          at(cs0[c], x) += v; at(cs1[c], x) += v * zh;
        }
        dyo[c] = dy;
      }
      if (ok) st_row<__bf16, E>((__bf16*)dzp + tok * DZ, e, q, dyo);
    }
    const int dd = lane, col = (dd >> 5) * (E * FDG) + e * FDG + (dd & 31);
    if (vb + 1 == vb1) {
      fold_put_cols<2>(fold_a, 0, cs0, wave, lane); fold_put_cols<2>(fold_a, 1, cs1, wave, lane);
      lds_barrier();
      if (ts == 0) {
        const int idx[2] = {dd, FDD + dd};
        float v[2];
        fold_get<E, NS, 2, 2>(fold_a, idx, e, v);
        colpart[((long)vb * 4 + 2) * (E * FDD) + col] = v[0];
        colpart[((long)vb * 4 + 3) * (E * FDD) + col] = v[1];
      }
    } else if (ts == 0) {
      colpart[((long)vb * 4 + 2) * (E * FDD) + col] = 0.f;
      colpart[((long)vb * 4 + 3) * (E * FDD) + col] = 0.f;
    }
  }
}

template <int E>
int launch_mdb(const SMidBArgs& a, int gx, const Plan& pl, char* saved, char* scratch, hipStream_t st) {
  static LdsAttrOnce attr;
  AVMOE_TRY(attr.ensure((const void*)kfs_mid_bwd<E>, 160 * 1024, "mid_bwd (streaming)"));
  hipLaunchKernelGGL((kfs_mid_bwd<E>), dim3((unsigned)gx), dim3(WS<E>::NTHR), pl.d.excl ? (size_t)160 * 1024 : mdb_lds<E>(), st, a, (const unsigned short*)(saved + pl.o_Z),
                     (const float*)(saved + pl.o_bn1), (const float*)(scratch + pl.o_dsm), (const float*)(scratch + pl.o_sdSzz), (unsigned short*)(scratch + pl.o_dzp),
                     (float*)(scratch + pl.o_colpart));
  AVMOE_CHECK_LAUNCH("mid_bwd (streaming)");
  return OK;
}

// =====================================================================================================
// POST_SMALL forward (bf16)      tile_fast.hip::kf_post_small<__bf16, E>     (net_trans_v3.py:430-434,485-486)
//   z' = act(BN1(z)), LayerNorm-post statistics from the d x d quadratic form, Apost rows [q rp z' | q rp, -q rp mup, q per expert].
//   The 3 E scalar columns of a virtual block's rows are written after its tiles, from rp / mup kept in the LDS (tile_fast.hip re-reads
//   them from global memory): one thread per (token, group) row writes its 6 E bytes as a run.
// =====================================================================================================
struct SPostArgs { P16 gate; int relu_of_e[MAX_E]; FastDims t; int ln_post, use_gate, bps, nvb, nfr; float ln_eps; };

constexpr int PSF_NL = 2, PSF_NST = 4;                      // ring loads / stores per wave and tile
constexpr int PSF_TILE = 2 * 1024;                          // Z (2 x 1 KB)
template <int E> constexpr int psf_fixed_floats(int nfr, int per) { return E * 4 * FDD + ((nfr * E + 3) & ~3) + 2 * E * per; }
template <int E> constexpr size_t psf_lds(int nfr, int per) { return (size_t)psf_fixed_floats<E>(nfr, per) * 4 + (size_t)WS<E>::NW * (KFS_P + 1) * PSF_TILE; }

template <int E>
__global__ void __launch_bounds__(WS<E>::NTHR, 2) kfs_post_small(SPostArgs a, const unsigned short* __restrict__ Z, const float* __restrict__ bn1, const float* __restrict__ Gq,
                                                              const float* __restrict__ uvh, const float* __restrict__ probs, unsigned short* __restrict__ Apost,
                                                              float* __restrict__ rpmup) {
  constexpr int DZ = E * FDD, NS = WS<E>::NS, NTHR = WS<E>::NTHR, D = KFS_P + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_uv = (float*)smem;                       // [E][us | vh | sc | sh]
  float* s_qv = s_uv + E * 4 * FDD;                 // [nfr][E]   probs * gate of the block's frames
  float* s_rm = s_qv + ((a.nfr * E + 3) & ~3);      // [per][E][rp, mup]  of the virtual block's tokens
  const FastDims& t = a.t;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int e = wave % E, ts = wave / E;
  char* ring = smem + (size_t)psf_fixed_floats<E>(a.nfr, t.per) * 4 + (size_t)wave * D * PSF_TILE;
  const unsigned ring_a = lds_off(ring), qv_a = lds_off(s_qv), uv_a = lds_off(s_uv + e * 4 * FDD), rm_a = lds_off(s_rm);
  const int vb0 = (int)((long)a.nvb * blockIdx.x / gridDim.x), vb1 = (int)((long)a.nvb * (blockIdx.x + 1) / gridDim.x);
  const int bps = a.bps, per = t.per, N = t.N;
  const int s_first = vb0 / bps;
  for (int i = threadIdx.x; i < a.nfr * E; i += NTHR) {
    const int ee = i % E, s = min(s_first + i / E, t.S - 1);
    s_qv[i] = probs[(long)s * E + ee] * (a.use_gate ? a.gate.p[ee][0] : 1.f);
  }
  for (int i = threadIdx.x; i < E * FDD; i += NTHR) {
    const int ee = i >> 6, dd = i & 63, col = (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31);
    float* u = s_uv + ee * 4 * FDD;
    u[dd] = uvh[col]; u[FDD + dd] = uvh[DZ + col]; u[2 * FDD + dd] = bn1[2 * DZ + col]; u[3 * FDD + dd] = bn1[3 * DZ + col];
  }
  float H1 = 0.f, H2 = 0.f;
  for (int gi = 0; gi < 2; ++gi) { H1 += uvh[2 * DZ + gi * E + e]; H2 += uvh[2 * DZ + 2 * E + gi * E + e]; }
  __syncthreads();                 // (no C++-level LDS access from here on)
  auto request = [&](int slot, const TileIt& it) {
    const long tk = (long)it.s * N + min(it.n0 + r, N - 1);
    char* dst = ring + slot * PSF_TILE;
    const char* zs = (const char*)(Z + tk * DZ + e * FDG + seg_off8(q));
    glds16(zs, dst); glds16(zs + E * FDG * 2, dst + 1024);
  };
  TileIt pf = it_first<NS>(vb0, vb1, bps, per, N, ts);
  int nreq = 0;
#pragma unroll
  for (int k = 0; k < KFS_P; ++k)
    if (pf.vb < vb1) { request(nreq % D, pf); ++nreq; it_next<NS>(pf, vb1, bps, per, N, ts); }
  kf_bf16x8 gh[2][2], gl[2][2];                     // Gq^T rows as bf16 planes (mm_presplit)
#pragma unroll
  for (int gi = 0; gi < 2; ++gi)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const float* g = Gq + (long)(gi * E + e) * FDG * FDG + 16 * ct + r;
      float4 m0, m1;
#pragma unroll
      for (int x = 0; x < 4; ++x) { at(m0, x) = g[(4 * q + x) * FDG]; at(m1, x) = g[(16 + 4 * q + x) * FDG]; }
      kf_split8(m0, m1, gh[gi][ct], gl[gi][ct]);
    }
  const bool relu = a.relu_of_e[e];
  const float fC = (float)t.C;
  int it = 0;
  int s = s_first, vbb = vb0 - s_first * bps;
  for (int vb = vb0; vb < vb1; ++vb, s += (vbb + 1 == bps), vbb = (vbb + 1 == bps) ? 0 : vbb + 1) {
    const int n_beg = vbb * per, n_end = min(N, n_beg + per);
    const float qv = lds_rd1(qv_a + 4 * ((s - s_first) * E + e));
    for (int n0 = n_beg + 16 * ts; n0 < n_end; n0 += 16 * NS, ++it) {
      wait_tile<PSF_NL, PSF_NST>(min(KFS_P - 1, nreq - it - 1), min(it, KFS_P));
      if (pf.vb < vb1) { request(nreq % D, pf); ++nreq; it_next<NS>(pf, vb1, bps, per, N, ts); }
      const bool ok = n0 + r < N;
      const long tok = (long)s * N + n0 + r;
      const unsigned sl = ring_a + (it % D) * PSF_TILE;
      f32x4 rz0, rz1, sc[4], sh[4];
      lds_rd16<0>(rz0, sl + lane * 16); lds_rd16<1024>(rz1, sl + lane * 16);
      lds_rd16<512>(sc[0], uv_a + 16 * q); lds_rd16<576>(sc[1], uv_a + 16 * q); lds_rd16<640>(sc[2], uv_a + 16 * q); lds_rd16<704>(sc[3], uv_a + 16 * q);
      lds_rd16<768>(sh[0], uv_a + 16 * q); lds_rd16<832>(sh[1], uv_a + 16 * q); lds_rd16<896>(sh[2], uv_a + 16 * q); lds_rd16<960>(sh[3], uv_a + 16 * q);
      lds_wait();
      lds_use(rz0, rz1); lds_use(sc[0], sc[1], sc[2], sc[3]); lds_use(sh[0], sh[1], sh[2], sh[3]);
      float4 zraw[4], zp[4];
      unpack_seg(as_u4(rz0), zraw[0], zraw[1]); unpack_seg(as_u4(rz1), zraw[2], zraw[3]);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float y = at(zraw[c], x) * sc[c][x] + sh[c][x];
          at(zp[c], x) = relu ? fmaxf(y, 0.f) : y;
        }
      float rp = 1.f, mup = 0.f;
      if (a.ln_post) {
        f32x4 us[4], vh[4];
        lds_rd16<0>(us[0], uv_a + 16 * q); lds_rd16<64>(us[1], uv_a + 16 * q); lds_rd16<128>(us[2], uv_a + 16 * q); lds_rd16<192>(us[3], uv_a + 16 * q);
        lds_rd16<256>(vh[0], uv_a + 16 * q); lds_rd16<320>(vh[1], uv_a + 16 * q); lds_rd16<384>(vh[2], uv_a + 16 * q); lds_rd16<448>(vh[3], uv_a + 16 * q);
        lds_wait();
        lds_use(us[0], us[1], us[2], us[3]); lds_use(vh[0], vh[1], vh[2], vh[3]);
        float so = 0.f, soo = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int gi = c >> 1, ct = c & 1;
          const f32x4 w = mm_presplit(gh[gi][ct], gl[gi][ct], zp[2 * gi], zp[2 * gi + 1]);
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float zv = at(zp[c], x);
            so += zv * us[c][x];
            soo += zv * (w[x] + 2.f * vh[c][x]);
          }
        }
        const float So = qsum4(so) + H1, Soo = qsum4(soo) + H2;
        mup = So / fC;
        rp = rsqrtf(fmaxf(Soo / fC - mup * mup, 0.f) + a.ln_eps);
      }
      if (q == 2) {                                  // (this virtual block's rp / mup, for its scalar-column pass below)
        const unsigned o = rm_a + 4 * (((n0 - n_beg) + r) * E + e) * 2;
        lds_wr1(o, rp); lds_wr1(o + 4, mup);
      }
      if (ok) {
        const float scl = qv * rp;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          const float4 v0 = make_float4(scl * zp[2 * gi].x, scl * zp[2 * gi].y, scl * zp[2 * gi].z, scl * zp[2 * gi].w);
          const float4 v1 = make_float4(scl * zp[2 * gi + 1].x, scl * zp[2 * gi + 1].y, scl * zp[2 * gi + 1].z, scl * zp[2 * gi + 1].w);
          st_seg<__bf16>((__bf16*)Apost + (tok * 2 + gi) * t.KPp + e * FDG, v0, v1, q);
        }
        if (q == 2) { rpmup[(long)e * t.NT + tok] = rp; rpmup[(long)t.NT * E + (long)e * t.NT + tok] = mup; }
      }
    }
    // the 3 E scalar columns [q rp, -q rp mup, q] of the virtual block's (token, group) rows, for all experts at once
    lds_barrier();
    {
      float qe[E];
#pragma unroll
      for (int ee = 0; ee < E; ++ee) qe[ee] = lds_rd1(qv_a + 4 * ((s - s_first) * E + ee));
      for (int idx = threadIdx.x; idx < 2 * (n_end - n_beg); idx += NTHR) {
        const int tl = idx >> 1;
        const long tok = (long)s * N + n_beg + tl;
        float v[3 * E], rm[2 * E];
#pragma unroll
        for (int ee = 0; ee < 2 * E; ++ee) lds_rd1_issue(rm[ee], rm_a + 4 * (tl * E * 2 + ee));
        lds_wait();
#pragma unroll
        for (int ee = 0; ee < E; ++ee) {
          lds_use(rm[2 * ee]); lds_use(rm[2 * ee + 1]);
          v[3 * ee] = qe[ee] * rm[2 * ee]; v[3 * ee + 1] = -qe[ee] * rm[2 * ee] * rm[2 * ee + 1]; v[3 * ee + 2] = qe[ee];
        }
        unsigned* d32 = (unsigned*)(Apost + (tok * 2 + (idx & 1)) * t.KPp + E * FDG);      // (E * FDG even, KPp a multiple of 4: 4-byte aligned)
#pragma unroll
        for (int j = 0; j + 1 < 3 * E; j += 2) d32[j >> 1] = (unsigned)f2bf(v[j]) | ((unsigned)f2bf(v[j + 1]) << 16);
        if constexpr ((3 * E) & 1) Apost[(tok * 2 + (idx & 1)) * t.KPp + E * FDG + 3 * E - 1] = f2bf(v[3 * E - 1]);
      }
    }
    lds_barrier();                 // (s_rm is rewritten by the next virtual block's tiles)
  }
}

template <int E>
int launch_psf(const SPostArgs& a, int gx, const Plan& pl, char* saved, hipStream_t st) {
  static LdsAttrOnce attr;
  AVMOE_TRY(attr.ensure((const void*)kfs_post_small<E>, 160 * 1024, "post_small (streaming)"));
  hipLaunchKernelGGL((kfs_post_small<E>), dim3((unsigned)gx), dim3(WS<E>::NTHR), pl.d.excl ? (size_t)160 * 1024 : psf_lds<E>(a.nfr, a.t.per), st, a,
                     (const unsigned short*)(saved + pl.o_Z), (const float*)(saved + pl.o_bn1), (const float*)(saved + pl.o_Gq), (const float*)(saved + pl.o_uvh),
                     (const float*)(saved + pl.o_probs), (unsigned short*)(saved + pl.o_Apost), (float*)(saved + pl.o_rpmup));
  AVMOE_CHECK_LAUNCH("post_small (streaming)");
  return OK;
}

// =====================================================================================================
// PRE_SMALL forward (bf16, fused hop-2 logits)      tile_fast.hip::kf_pre_small<__bf16, E, false>     (net_trans_v3.py:385-395)
//   hop-2 softmax over the latent tokens, the folded LayerNorm-before, z in place, BatchNorm-1 column sums.
//   One tile ahead (the cross-modal waves' tiles are 6.4 KB: Z, two planes of logits, row sums); the per-FRAME constants of the cross-modal
//   experts (TT^T, TW^T, Tsum: 14 KB per latent slot) sit in two LDS buffers -- the next frame's are requested by direct dword loads whose
//   per-lane SOURCE addresses do the transposition and the padding, at the start of the current frame, and published by the barrier of the
//   frame's last virtual block (in-order counter: a wave that has waited for a tile requested after them has them).
// =====================================================================================================
struct SPreArgs { P16 glat; int lat_of_e[MAX_E]; FastDims t; int ln_before, bps, nvb; float ln_eps; const float* L2g; float* L2w; };

constexpr int PRS_P = 1, PRS_D = PRS_P + 1;
constexpr int PRS_LATF = 55 * 64;                           // dwords of a latent slot's constants: TT^T 32 x 36 | TW^T 64 x 36 | Tsum 32 | pad (whole 64-dword pieces)
constexpr int PRS_TILE_X = 2 * 1024 + 4 * 1024 + 256, PRS_TILE_U = 2 * 1024 + 256;      // ring bytes per tile: cross-modal / unimodal wave
constexpr int PRS_NL_X = 7, PRS_NST_X = 7, PRS_NL_U = 3, PRS_NST_U = 4;
template <int E> constexpr int prs_fixed_floats(int El) { return 2 * WS<E>::NW * fold_stride<2>() + E * 2 * FDD + 2 * El * PRS_LATF; }
template <int E> constexpr size_t prs_lds(int El) { return (size_t)prs_fixed_floats<E>(El) * 4 + (size_t)WS<E>::NS * PRS_D * ((size_t)El * PRS_TILE_X + (size_t)(E - El) * PRS_TILE_U); }

// a transposed mat-vec step with the matrix rows in the LDS (fp32, leading dimension LD32): A-operand rows (col0 + r), entries 4 q .. and 16 + 4 q ..
__device__ __forceinline__ void mm_lds_issue(f32x4& a0, f32x4& a1, unsigned mt_a, int col0, int r, int q) {
  const unsigned ad = mt_a + 4 * ((col0 + r) * LD32 + 4 * q);
  lds_rd16<0>(a0, ad); lds_rd16<64>(a1, ad);
}
__device__ __forceinline__ f32x4 mm_lds_finish(const f32x4& a0, const f32x4& a1, const float4& p0, const float4& p1) {
  kf_bf16x8 ah, al;
  kf_split8(make_float4(a0[0], a0[1], a0[2], a0[3]), make_float4(a1[0], a1[1], a1[2], a1[3]), ah, al);
  return mm_presplit(ah, al, p0, p1);
}

template <int E>
__global__ void __launch_bounds__(WS<E>::NTHR, 2) kfs_pre_small(SPreArgs a, unsigned short* __restrict__ Z, const float* __restrict__ sxs, const float* __restrict__ TT,
                                                             const float* __restrict__ TW, const float* __restrict__ Tsum, const float* __restrict__ wsum,
                                                             const float* __restrict__ dconst, unsigned short* __restrict__ aout, float* __restrict__ rmu,
                                                             float* __restrict__ colpart) {
  constexpr int DZ = E * FDD, NS = WS<E>::NS, NW = WS<E>::NW, NTHR = WS<E>::NTHR, D = PRS_D;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FastDims& t = a.t;
  const int El = t.El;
  float* s_fold = (float*)smem;                             // [2][NW][2 * 64 + 4]
  float* s_ce = s_fold + 2 * NW * fold_stride<2>();         // [E][wsum | dconst]
  float* s_lat = s_ce + E * 2 * FDD;                        // [2 buffers][El][PRS_LATF]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int e = wave % E, ts = wave / E;
  const int l = a.lat_of_e[e];
  const bool lat = l >= 0;
  // this wave's ring: the cross-modal waves' tiles are larger (tile slots ts, experts in order: cross-modal experts first or not -- a prefix sum)
  int roff = ts * (El * PRS_TILE_X + (E - El) * PRS_TILE_U);
  for (int ee = 0; ee < e; ++ee) roff += a.lat_of_e[ee] >= 0 ? PRS_TILE_X : PRS_TILE_U;
  const int tile_b = lat ? PRS_TILE_X : PRS_TILE_U;
  char* ring = smem + (size_t)prs_fixed_floats<E>(El) * 4 + (size_t)roff * D;
  const unsigned ring_a = lds_off(ring), fold_a = lds_off(s_fold), ce_a = lds_off(s_ce + e * 2 * FDD), lat_a = lds_off(s_lat);
  const int vb0 = (int)((long)a.nvb * blockIdx.x / gridDim.x), vb1 = (int)((long)a.nvb * (blockIdx.x + 1) / gridDim.x);
  const int bps = a.bps, per = t.per, N = t.N;
  const int s_first = vb0 / bps, s_last = (vb1 - 1) / bps;
  const float gv = lat ? a.glat.p[e][0] : 0.f;
  const float fC = (float)t.C, invC = 1.f / (float)t.C;

  for (int i = threadIdx.x; i < E * FDD; i += NTHR) {
    const int ee = i >> 6, dd = i & 63, col = (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31);
    s_ce[ee * 2 * FDD + dd] = wsum[col]; s_ce[ee * 2 * FDD + FDD + dd] = dconst[col];
  }
  __syncthreads();                 // (no C++-level LDS access from here on)

  // frame s's constants of every latent slot -> buffer s & 1: 55 dword pieces per slot dealt to the waves; LDS dword p of a slot <- the source
  // element that belongs there (transposition and padding by address)
  auto request_lat = [&](int s) {
    for (int ll = 0; ll < El; ++ll) {
      int ee = 0;
      for (int x = 0; x < E; ++x) if (a.lat_of_e[x] == ll) ee = x;
      const float* tt = TT + ((long)s * El + ll) * FK * FK;
      const float* tw = TW + ((long)s * t.KLT + (long)ll * FK) * DZ;
      const float* tsu = Tsum + (long)s * t.KLT + (long)ll * FK;
      char* dst = (char*)(s_lat + ((s & 1) * El + ll) * PRS_LATF);
      for (int j = wave; j < PRS_LATF / 64; j += NW) {
        const int p = 64 * j + lane;
        const float* src = tt;
        if (p < FK * LD32) { const int c = p / LD32, k = p - c * LD32; if (k < FK) src = tt + k * FK + c; }
        else if (p < FK * LD32 + FDD * LD32) { const int pp = p - FK * LD32, dd = pp / LD32, k = pp - dd * LD32; if (k < FK) src = tw + (long)k * DZ + (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31); }
        else if (p < FK * LD32 + FDD * LD32 + FK) src = tsu + (p - FK * LD32 - FDD * LD32);
        glds4(src, dst + 256 * j);
      }
    }
  };
  auto request = [&](int slot, const TileIt& it) {
    const long f0 = (long)it.s * N;
    const long tk = f0 + min(it.n0 + r, N - 1);
    char* dst = ring + slot * tile_b;
    const char* zs = (const char*)(Z + tk * DZ + e * FDG + seg_off8(q));
    glds16(zs, dst); glds16(zs + E * FDG * 2, dst + 1024);
    const long t2 = f0 + min(it.n0 + (lane >> 2), N - 1);
    glds4(sxs + ((lane & 1) ? (long)t.NT : 0L) + t2, dst + 2048);          // [token][Sx, Sxx, Sx, Sxx]
    if (lat) {
      const char* lg = (const char*)(a.L2g + tk * t.KL + (long)l * FK + 4 * q);
      glds16(lg, dst + 2304); glds16(lg + 64, dst + 3328);
      glds16(lg + (long)t.NT * t.KL * 4, dst + 4352); glds16(lg + (long)t.NT * t.KL * 4 + 64, dst + 5376);
    }
  };
  if (El > 0) request_lat(s_first);
  TileIt pf = it_first<NS>(vb0, vb1, bps, per, N, ts);
  int nreq = 0;
  if (pf.vb < vb1) { request(0, pf); ++nreq; it_next<NS>(pf, vb1, bps, per, N, ts); }
  wait_vm<0>();
  lds_barrier();                   // the first frame's constants (and nothing of the ring is read before its own wait)
  wait_vm<0>();

  int it = 0;
  int s = s_first, vbb = vb0 - s_first * bps;
  int since = 2;                   // tiles this wave has waited for since its last request_lat (>= 2: those loads have landed)
  bool fresh = true;               // the next tile is the wave's first of a frame whose successor's constants are not requested yet
  float4 cs0[4], cs1[4];            // (column sums over the block's whole range: folded once, into the rows of its last virtual block -- see kfs_post_small_bwd)
#pragma unroll
  for (int c = 0; c < 4; ++c) { cs0[c] = zero4(); cs1[c] = zero4(); }
  for (int vb = vb0; vb < vb1; ++vb) {
    const int n_beg = vbb * per, n_end = min(N, n_beg + per);
    const unsigned lt_a = lat_a + 4 * (((s & 1) * El + (lat ? l : 0)) * PRS_LATF);      // TT^T | TW^T | Tsum of this wave's slot, this frame
    if (fresh && El > 0 && s < s_last) { request_lat(s + 1); since = 0; }       // (every wave, whether it has a tile in this virtual block or not)
    fresh = false;
    for (int n0 = n_beg + 16 * ts; n0 < n_end; n0 += 16 * NS, ++it) {
      if (lat) wait_tile<PRS_NL_X, PRS_NST_X>(min(PRS_P - 1, nreq - it - 1), min(it, PRS_P));
      else wait_tile<PRS_NL_U, PRS_NST_U>(min(PRS_P - 1, nreq - it - 1), min(it, PRS_P));
      ++since;
      if (pf.vb < vb1) { request(nreq % D, pf); ++nreq; it_next<NS>(pf, vb1, bps, per, N, ts); }
      const bool ok = n0 + r < N;
      const long tok = (long)s * N + n0 + r;
      const unsigned sl = ring_a + (it % D) * tile_b;
      f32x4 rz0, rz1, rs, ws[4], dc[4];
      lds_rd16<0>(rz0, sl + lane * 16); lds_rd16<1024>(rz1, sl + lane * 16); lds_rd16<2048>(rs, sl + r * 16);
      lds_rd16<0>(ws[0], ce_a + 16 * q); lds_rd16<64>(ws[1], ce_a + 16 * q); lds_rd16<128>(ws[2], ce_a + 16 * q); lds_rd16<192>(ws[3], ce_a + 16 * q);
      lds_rd16<256>(dc[0], ce_a + 16 * q); lds_rd16<320>(dc[1], ce_a + 16 * q); lds_rd16<384>(dc[2], ce_a + 16 * q); lds_rd16<448>(dc[3], ce_a + 16 * q);
      float Sx, Sxx;
      float4 z[4], zo[4], lg[2], av[2] = {zero4(), zero4()};
      f32x4 pw[4];                                   // (a TW) for the four 16-entry chunks
#pragma unroll
      for (int c = 0; c < 4; ++c) pw[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (lat) {
        f32x4 a0, a1, b0, b1, tb0, tb1;
        lds_rd16<2304>(a0, sl + lane * 16); lds_rd16<3328>(a1, sl + lane * 16); lds_rd16<4352>(b0, sl + lane * 16); lds_rd16<5376>(b1, sl + lane * 16);
        lds_rd16<0>(tb0, lt_a + 4 * (FK * LD32 + FDD * LD32) + 16 * q); lds_rd16<64>(tb1, lt_a + 4 * (FK * LD32 + FDD * LD32) + 16 * q);
        lds_wait();
        lds_use(rz0, rz1, rs, a0, a1, b0, b1, tb0, tb1);
        lds_use(ws[0], ws[1], ws[2], ws[3]); lds_use(dc[0], dc[1], dc[2], dc[3]);
        Sx = ok ? rs[0] : 0.f; Sxx = ok ? rs[1] : 1.f;
        lg[0] = make_float4(a0[0] + b0[0], a0[1] + b0[1], a0[2] + b0[2], a0[3] + b0[3]);
        lg[1] = make_float4(a1[0] + b1[0], a1[1] + b1[1], a1[2] + b1[2], a1[3] + b1[3]);
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int x = 0; x < 4; ++x) mx = fmaxf(mx, at(lg[j], x));
        mx = qmax4(mx);
        float sum = 0.f;
        float4 ex[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int x = 0; x < 4; ++x) { at(ex[j], x) = __expf(at(lg[j], x) - mx); sum += at(ex[j], x); }
        sum = qsum4(sum);
        const float inv = ok ? 1.f / sum : 0.f;
        float u1 = 0.f, u2 = 0.f;
        const f32x4 tbv[2] = {tb0, tb1};
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float v = rndT<__bf16>(at(ex[j], x) * inv);
            at(av[j], x) = v;
            u1 += v * (tbv[j][x] / fC); u2 += v * at(lg[j], x);
          }
        u1 = qsum4(u1); u2 = qsum4(u2);
        // mat-vecs against this frame's TT and TW (matrix rows from the LDS, all reads of a step issued together)
        f32x4 m[12];
        mm_lds_issue(m[0], m[1], lt_a, 0, r, q); mm_lds_issue(m[2], m[3], lt_a, 16, r, q);
#pragma unroll
        for (int c = 0; c < 4; ++c) mm_lds_issue(m[4 + 2 * c], m[5 + 2 * c], lt_a + 4 * FK * LD32, 16 * c, r, q);
        lds_wait();
#pragma unroll
        for (int i = 0; i < 12; ++i) lds_use(m[i]);
        float u3 = 0.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const f32x4 w = mm_lds_finish(m[2 * ct], m[2 * ct + 1], av[0], av[1]);
#pragma unroll
          for (int x = 0; x < 4; ++x) u3 += w[x] * at(av[ct], x);
        }
        u3 = qsum4(u3);
#pragma unroll
        for (int c = 0; c < 4; ++c) pw[c] = mm_lds_finish(m[4 + 2 * c], m[5 + 2 * c], av[0], av[1]);
        Sx += gv * fC * u1;
        Sxx += 2.f * gv * u2 + gv * gv * u3;
      } else {
        lds_wait();
        lds_use(rz0, rz1, rs);
        lds_use(ws[0], ws[1], ws[2], ws[3]); lds_use(dc[0], dc[1], dc[2], dc[3]);
        Sx = ok ? rs[0] : 0.f; Sxx = ok ? rs[1] : 1.f;
        lg[0] = lg[1] = zero4();
      }
      unpack_seg(as_u4(rz0), z[0], z[1]); unpack_seg(as_u4(rz1), z[2], z[3]);
      float mu = 0.f, rr = 1.f;
      if (a.ln_before) {
        mu = Sx / fC;
        rr = rsqrtf(fmaxf(Sxx / fC - mu * mu, 0.f) + a.ln_eps);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float4 o;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float zr = at(z[c], x) + gv * pw[c][x];
          const float zv = rndT<__bf16>(a.ln_before ? rr * (zr - mu * ws[c][x]) + dc[c][x] : zr);   // as stored: BN1 statistics of the stored z
          at(o, x) = zv;
          if (ok) { at(cs0[c], x) += zv; at(cs1[c], x) += zv * zv; }
        }
        zo[c] = o;
      }
      // PRS_NST_X / PRS_NST_U store instructions per tile, exactly (lane r = 0 of a tile in range is valid)
      if (ok) {
        if (lat) {
          st_seg<__bf16>((__bf16*)aout + (long)l * t.aL + tok * FK, av[0], av[1], q);
          float* lw = a.L2w + tok * t.KL + (long)l * FK + 4 * q;         // the logits themselves (the backward's hop-2 block reads them)
          *(float4*)lw = lg[0]; *(float4*)(lw + 16) = lg[1];
        }
        st_row<__bf16, E>((__bf16*)Z + tok * DZ, e, q, zo);
        if (q == 0) { rmu[(long)e * t.NT + tok] = rr; rmu[(long)t.NT * E + (long)e * t.NT + tok] = mu; }
      }
    }
    const bool frame_ends = vbb + 1 == bps, final = vb + 1 == vb1;
    const int dd = lane, col = (dd >> 5) * (E * FDG) + e * FDG + (dd & 31);
    if (final) { fold_put_cols<2>(fold_a, 0, cs0, wave, lane); fold_put_cols<2>(fold_a, 1, cs1, wave, lane); }
    if (frame_ends && !final && El > 0) {
      if (since < 2) wait_vm<0>();                    // the next frame's constants this wave requested: landed before the barrier publishes them
      lds_barrier();
    }
    if (final) {
      lds_barrier();
      if (ts == 0) {
        const int idx[2] = {dd, FDD + dd};
        float v[2];
        fold_get<E, NS, 2, 2>(fold_a, idx, e, v);
        colpart[((long)vb * 4 + 0) * (E * FDD) + col] = v[0];
        colpart[((long)vb * 4 + 1) * (E * FDD) + col] = v[1];
      }
    } else if (ts == 0) {
      colpart[((long)vb * 4 + 0) * (E * FDD) + col] = 0.f;
      colpart[((long)vb * 4 + 1) * (E * FDD) + col] = 0.f;
    }
    if (frame_ends) { vbb = 0; ++s; fresh = true; } else ++vbb;
  }
}

template <int E>
int launch_prs(const SPreArgs& a, int gx, const Plan& pl, char* saved, char* scratch, hipStream_t st) {
  static LdsAttrOnce attr;
  AVMOE_TRY(attr.ensure((const void*)kfs_pre_small<E>, 160 * 1024, "pre_small (streaming)"));
  hipLaunchKernelGGL((kfs_pre_small<E>), dim3((unsigned)gx), dim3(WS<E>::NTHR), pl.d.excl ? (size_t)160 * 1024 : prs_lds<E>(a.t.El), st, a, (unsigned short*)(saved + pl.o_Z),
                     (const float*)(saved + pl.o_sx), (const float*)(saved + pl.o_TT), (const float*)(saved + pl.o_TW), (const float*)(saved + pl.o_Tsum),
                     (const float*)(saved + pl.o_wsum), (const float*)(saved + pl.o_dconst), (unsigned short*)(saved + pl.o_a), (float*)(saved + pl.o_rmu),
                     (float*)(scratch + pl.o_colpart));
  AVMOE_CHECK_LAUNCH("pre_small (streaming)");
  return OK;
}

// =====================================================================================================
// PRE_SMALL backward + the hop-2 block of the cross-modal experts, ONE pass (bf16)
//   tile_fast.hip::kf_pre_small_bwd<__bf16, E, false> followed by kf_pre_lat_bwd<__bf16, E>: BatchNorm-1 input gradient, folded-LayerNorm sums,
//   dzraw for every expert; for the cross-modal experts the softmax backward over the latent tokens and the mat-vecs against TW / TT -- from
//   dzraw IN REGISTERS (fp32: the gate_av gradient's <dzraw, a TW> no longer sees a bf16-rounded dzraw) instead of read back from dZx, the
//   expert's LayerNorm sums in registers instead of through dslat.
//   tile_fast.hip split the two because the hop-2 block's registers set the occupancy of the whole sweep; here every wave has 256 registers
//   anyway, the constants sit in the LDS, and the waves are dealt so that every SIMD gets one cross-modal and one unimodal expert.
//   Per virtual block: the tiles (no barrier), then a short pass that adds the experts' LayerNorm sums per token (dL2x's statistics columns,
//   rs2x) from an LDS table -- tile_fast.hip exchanged them per TILE behind a barrier.  One buffer of per-frame constants (TT^T, TW^T, Tsum),
//   refilled between two barriers at a frame change; TW (latent-major) is read from the TW^T image with strided 4-byte reads.
// =====================================================================================================
struct SPreBArgs { int lat_of_e[MAX_E]; P16 glat; FastDims t; int ln_before, use_bn, bn_train, bps, nvb;
                   const float* L2; const float* TT; const float* TW; const float* Tsum; const unsigned short* ain; unsigned short* aw; unsigned short* ag; float* dtbp; };

constexpr int PRB_TILE_U = 4 * 1024 + 256, PRB_TILE_X = PRB_TILE_U + 3 * 1024;      // ring bytes per tile: unimodal / cross-modal wave
constexpr int PRB_NL_U = 5, PRB_NST_U = 2, PRB_NL_X = 8, PRB_NST_X = 5;
constexpr int PRB_FOLD = 2 * FDD + FK + 4;                   // floats a wave folds: two column accumulators, the 32 dtb sums, the gate partial
template <int E> constexpr int prb_fixed_floats(int El, int per) { return WS<E>::NW * PRB_FOLD + E * 7 * FDD + per * E * 2 + El * PRS_LATF; }
template <int E> constexpr size_t prb_lds(int El, int per) { return (size_t)prb_fixed_floats<E>(El, per) * 4 + (size_t)WS<E>::NS * PRS_D * ((size_t)El * PRB_TILE_X + (size_t)(E - El) * PRB_TILE_U); }

template <int E>
__global__ void __launch_bounds__(WS<E>::NTHR, 2) kfs_pre_bwd(SPreBArgs a, const unsigned short* __restrict__ Z, const float* __restrict__ wsum, const float* __restrict__ dconst,
                                                           const float* __restrict__ rmu, const float* __restrict__ bn1, const float* __restrict__ dsm,
                                                           const unsigned short* __restrict__ dy_in, unsigned short* __restrict__ dZx, unsigned short* __restrict__ dL2x,
                                                           float* __restrict__ rs2x, float* __restrict__ colpart, float* __restrict__ blkscal) {
  constexpr int DZ = E * FDD, NS = WS<E>::NS, NW = WS<E>::NW, NTHR = WS<E>::NTHR, D = PRS_D;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FastDims& t = a.t;
  const int El = t.El;
  float* s_fold = (float*)smem;                             // [NW][PRB_FOLD]
  float* s_bne = s_fold + NW * PRB_FOLD;                    // [E][mean | rstd | sc | mdy | mdyz | wsum | dconst]
  float* s_ds = s_bne + E * 7 * FDD;                        // [per][E][dSx, dSxx]  of the virtual block's tokens
  float* s_lat = s_ds + t.per * E * 2;                      // [El][PRS_LATF]       this frame's TT^T | TW^T | Tsum
  const int hw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  // waves dealt so that a SIMD (hardware wave & 3) gets different experts in its two tile slots: the cross-modal experts' tiles cost three
  // times the unimodal ones'
  const int ts = hw / E, e = (E % 2 == 0) ? (hw + ts * (E / 2)) % E : hw % E;
  const int wave = ts * E + e;                              // logical index (ring, fold)
  const int l = a.lat_of_e[e];
  const bool lat = l >= 0;
  int roff = ts * (El * PRB_TILE_X + (E - El) * PRB_TILE_U);
  for (int ee = 0; ee < e; ++ee) roff += a.lat_of_e[ee] >= 0 ? PRB_TILE_X : PRB_TILE_U;
  const int tile_b = lat ? PRB_TILE_X : PRB_TILE_U;
  char* ring = smem + (size_t)prb_fixed_floats<E>(El, t.per) * 4 + (size_t)roff * D;
  const unsigned ring_a = lds_off(ring), fold_a = lds_off(s_fold), bn_a = lds_off(s_bne + e * 7 * FDD), ds_a = lds_off(s_ds);
  const unsigned lt_a = lds_off(s_lat + (lat ? l : 0) * PRS_LATF);
  const int vb0 = (int)((long)a.nvb * blockIdx.x / gridDim.x), vb1 = (int)((long)a.nvb * (blockIdx.x + 1) / gridDim.x);
  const int bps = a.bps, per = t.per, N = t.N;
  const int s_first = vb0 / bps;
  const float gv = lat ? a.glat.p[e][0] : 0.f;
  const float fC = (float)t.C;

  for (int i = threadIdx.x; i < E * FDD; i += NTHR) {
    const int ee = i >> 6, dd = i & 63, col = (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31);
    float* b = s_bne + ee * 7 * FDD;
    b[dd] = bn1[col]; b[FDD + dd] = bn1[DZ + col]; b[2 * FDD + dd] = bn1[2 * DZ + col];
    b[3 * FDD + dd] = a.bn_train ? dsm[3 * DZ + col] : 0.f; b[4 * FDD + dd] = a.bn_train ? dsm[4 * DZ + col] : 0.f;
    b[5 * FDD + dd] = wsum[col]; b[6 * FDD + dd] = dconst[col];
  }
  __syncthreads();                 // (no C++-level LDS access from here on)

  auto request_lat = [&](int s) {  // frame s's constants of every latent slot: 55 dword pieces per slot dealt to the waves (kfs_pre_small has the address map)
    for (int ll = 0; ll < El; ++ll) {
      int ee = 0;
      for (int x = 0; x < E; ++x) if (a.lat_of_e[x] == ll) ee = x;
      const float* tt = a.TT + ((long)s * El + ll) * FK * FK;
      const float* tw = a.TW + ((long)s * t.KLT + (long)ll * FK) * DZ;
      const float* tsu = a.Tsum + (long)s * t.KLT + (long)ll * FK;
      char* dst = (char*)(s_lat + ll * PRS_LATF);
      for (int j = hw; j < PRS_LATF / 64; j += NW) {
        const int p = 64 * j + lane;
        const float* src = tt;
        if (p < FK * LD32) { const int c = p / LD32, k = p - c * LD32; if (k < FK) src = tt + k * FK + c; }
        else if (p < FK * LD32 + FDD * LD32) { const int pp = p - FK * LD32, dd = pp / LD32, k = pp - dd * LD32; if (k < FK) src = tw + (long)k * DZ + (dd >> 5) * (E * FDG) + ee * FDG + (dd & 31); }
        else if (p < FK * LD32 + FDD * LD32 + FK) src = tsu + (p - FK * LD32 - FDD * LD32);
        glds4(src, dst + 256 * j);
      }
    }
  };
  auto request = [&](int slot, const TileIt& it) {
    const long f0 = (long)it.s * N;
    const long tk = f0 + min(it.n0 + r, N - 1);
    char* dst = ring + slot * tile_b;
    const char* zs = (const char*)(Z + tk * DZ + e * FDG + seg_off8(q));
    glds16(zs, dst); glds16(zs + E * FDG * 2, dst + 1024);
    const char* ys = (const char*)(dy_in + tk * DZ + e * FDG + seg_off8(q));
    glds16(ys, dst + 2048); glds16(ys + E * FDG * 2, dst + 3072);
    const long t2 = f0 + min(it.n0 + (lane >> 2), N - 1);
    glds4(rmu + ((lane & 1) ? (long)t.NT * E : 0L) + (long)e * t.NT + t2, dst + 4096);          // [token][rr, mu, rr, mu]
    if (lat) {
      glds16(a.ain + (long)l * t.aL + tk * FK + seg_off8(q), dst + 4352);
      const char* lg = (const char*)(a.L2 + tk * t.KL + (long)l * FK + 4 * q);
      glds16(lg, dst + 5376); glds16(lg + 64, dst + 6400);
    }
  };
  if (El > 0) request_lat(s_first);
  TileIt pf = it_first<NS>(vb0, vb1, bps, per, N, ts);
  int nreq = 0;
  if (pf.vb < vb1) { request(0, pf); ++nreq; it_next<NS>(pf, vb1, bps, per, N, ts); }
  wait_vm<0>();
  lds_barrier();                   // the first frame's constants

  int it = 0;
  int s = s_first, vbb = vb0 - s_first * bps;
  float sdg = 0.f;
  float4 cs0[4], cs1[4], ck[2];     // column sums over the block's range (folded once), dtb sums per frame
#pragma unroll
  for (int c = 0; c < 4; ++c) { cs0[c] = zero4(); cs1[c] = zero4(); }
  ck[0] = zero4(); ck[1] = zero4();
  for (int vb = vb0; vb < vb1; ++vb) {
    const int n_beg = vbb * per, n_end = min(N, n_beg + per);
    int qsel = 0;
    for (int n0 = n_beg + 16 * ts; n0 < n_end; n0 += 16 * NS, qsel = (qsel + 1) & 3, ++it) {
      if (lat) wait_tile<PRB_NL_X, PRB_NST_X>(0, min(it, PRS_P));
      else wait_tile<PRB_NL_U, PRB_NST_U>(0, min(it, PRS_P));
      if (pf.vb < vb1) { request(nreq % D, pf); ++nreq; it_next<NS>(pf, vb1, bps, per, N, ts); }
      const bool ok = n0 + r < N;
      const long tok = (long)s * N + n0 + r;
      const unsigned sl = ring_a + (it % D) * tile_b;
      f32x4 rz0, rz1, ry0, ry1, rm;
      lds_rd16<0>(rz0, sl + lane * 16); lds_rd16<1024>(rz1, sl + lane * 16); lds_rd16<2048>(ry0, sl + lane * 16); lds_rd16<3072>(ry1, sl + lane * 16);
      lds_rd16<4096>(rm, sl + r * 16);
      f32x4 mean[4], rstd[4], scv[4], mdy[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) { }
      lds_rd16<0>(mean[0], bn_a + 16 * q); lds_rd16<64>(mean[1], bn_a + 16 * q); lds_rd16<128>(mean[2], bn_a + 16 * q); lds_rd16<192>(mean[3], bn_a + 16 * q);
      lds_rd16<256>(rstd[0], bn_a + 16 * q); lds_rd16<320>(rstd[1], bn_a + 16 * q); lds_rd16<384>(rstd[2], bn_a + 16 * q); lds_rd16<448>(rstd[3], bn_a + 16 * q);
      lds_rd16<512>(scv[0], bn_a + 16 * q); lds_rd16<576>(scv[1], bn_a + 16 * q); lds_rd16<640>(scv[2], bn_a + 16 * q); lds_rd16<704>(scv[3], bn_a + 16 * q);
      lds_rd16<768>(mdy[0], bn_a + 16 * q); lds_rd16<832>(mdy[1], bn_a + 16 * q); lds_rd16<896>(mdy[2], bn_a + 16 * q); lds_rd16<960>(mdy[3], bn_a + 16 * q);
      lds_wait();
      lds_use(rz0, rz1, ry0, ry1, rm);
      lds_use(mean[0], mean[1], mean[2], mean[3]); lds_use(rstd[0], rstd[1], rstd[2], rstd[3]);
      lds_use(scv[0], scv[1], scv[2], scv[3]); lds_use(mdy[0], mdy[1], mdy[2], mdy[3]);
      const float rr = ok && a.ln_before ? rm[0] : 1.f, mu = ok && a.ln_before ? rm[1] : 0.f;
      const float irr = 1.f / rr;
      float4 zrow[4], dzr[4];
      unpack_seg(as_u4(rz0), zrow[0], zrow[1]); unpack_seg(as_u4(rz1), zrow[2], zrow[3]);
      unpack_seg(as_u4(ry0), dzr[0], dzr[1]); unpack_seg(as_u4(ry1), dzr[2], dzr[3]);          // (dy on the way in, dzraw on the way out)
      // ---- BN1 input gradient, first half (needs mdyz) ----
      {
        f32x4 mdyz[4];
        lds_rd16<1024>(mdyz[0], bn_a + 16 * q); lds_rd16<1088>(mdyz[1], bn_a + 16 * q); lds_rd16<1152>(mdyz[2], bn_a + 16 * q); lds_rd16<1216>(mdyz[3], bn_a + 16 * q);
        lds_wait();
        lds_use(mdyz[0], mdyz[1], mdyz[2], mdyz[3]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            float dz = at(dzr[c], x);
            if (a.use_bn) {
              if (a.bn_train) dz = scv[c][x] * (dz - mdy[c][x] - (at(zrow[c], x) - mean[c][x]) * rstd[c][x] * mdyz[c][x]);
              else dz = scv[c][x] * dz;
            }
            at(dzr[c], x) = ok ? dz : 0.f;
          }
      }
      // ---- folded-LayerNorm sums, dzraw ----
      float s_dr = 0.f, s_dmu = 0.f;
      if (a.ln_before) {
        f32x4 ws[4], dc[4];
        lds_rd16<1280>(ws[0], bn_a + 16 * q); lds_rd16<1344>(ws[1], bn_a + 16 * q); lds_rd16<1408>(ws[2], bn_a + 16 * q); lds_rd16<1472>(ws[3], bn_a + 16 * q);
        lds_rd16<1536>(dc[0], bn_a + 16 * q); lds_rd16<1600>(dc[1], bn_a + 16 * q); lds_rd16<1664>(dc[2], bn_a + 16 * q); lds_rd16<1728>(dc[3], bn_a + 16 * q);
        lds_wait();
        lds_use(ws[0], ws[1], ws[2], ws[3]); lds_use(dc[0], dc[1], dc[2], dc[3]);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float dz = at(dzr[c], x);
            const float zc = (at(zrow[c], x) - dc[c][x]) * irr;
            at(cs0[c], x) += dz; at(cs1[c], x) += -rr * mu * dz;
            s_dr += dz * zc; s_dmu += dz * ws[c][x];
            at(dzr[c], x) = rr * dz;
          }
      }
      float dSx = 0.f, dSxx = 0.f;
      if (a.ln_before) {
        const float sdr = qsum4(s_dr), sdm = qsum4(s_dmu);
        float dmu = -rr * sdm;
        const float dvar = sdr * (-0.5f) * rr * rr * rr;
        dSxx = dvar / fC;
        dmu -= 2.f * mu * dvar;
        dSx = dmu / fC;
      }
      if (!ok) { dSx = 0.f; dSxx = 0.f; }
      if (q == 0) {                                  // this expert's sums of the tile's tokens: the per-token pass after the virtual block adds the experts
        const unsigned o = ds_a + 4 * ((((n0 - n_beg) + r) * E + e) * 2);
        lds_wr1(o, dSx); lds_wr1(o + 4, dSxx);
      }
      if (lat) {
        // ---- the hop-2 block: softmax backward over the latent tokens, mat-vecs against TW / TT (tile_fast.hip::kf_pre_lat_bwd) ----
        f32x4 ra, l0, l1, tb0, tb1;
        lds_rd16<4352>(ra, sl + lane * 16); lds_rd16<5376>(l0, sl + lane * 16); lds_rd16<6400>(l1, sl + lane * 16);
        lds_rd16<0>(tb0, lt_a + 4 * (FK * LD32 + FDD * LD32) + 16 * q); lds_rd16<64>(tb1, lt_a + 4 * (FK * LD32 + FDD * LD32) + 16 * q);
        f32x4 m[12];
        mm_lds_issue(m[0], m[1], lt_a, 0, r, q); mm_lds_issue(m[2], m[3], lt_a, 16, r, q);                 // TT^T rows
#pragma unroll
        for (int c = 0; c < 4; ++c) mm_lds_issue(m[4 + 2 * c], m[5 + 2 * c], lt_a + 4 * FK * LD32, 16 * c, r, q);      // TW^T rows
        lds_wait();
        lds_use(ra, l0, l1, tb0, tb1);
#pragma unroll
        for (int i = 0; i < 12; ++i) lds_use(m[i]);
        float4 av[2], lg[2], tb[2];
        unpack_seg(as_u4(ra), av[0], av[1]);
        lg[0] = make_float4(l0[0], l0[1], l0[2], l0[3]); lg[1] = make_float4(l1[0], l1[1], l1[2], l1[3]);
        tb[0] = make_float4(tb0[0] / fC, tb0[1] / fC, tb0[2] / fC, tb0[3] / fC); tb[1] = make_float4(tb1[0] / fC, tb1[1] / fC, tb1[2] / fC, tb1[3] / fC);
        float u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int x = 0; x < 4; ++x) { u1 += at(av[j], x) * at(tb[j], x); u2 += at(av[j], x) * at(lg[j], x); }
        u1 = qsum4(u1); u2 = qsum4(u2);
        float dgr = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {                                 // dzraw . (a TW)
          const f32x4 p = mm_lds_finish(m[4 + 2 * c], m[5 + 2 * c], av[0], av[1]);
#pragma unroll
          for (int x = 0; x < 4; ++x) dgr += p[x] * at(dzr[c], x);
        }
        dgr = qsum4(dgr);
        const float du1 = dSx * gv * fC, du2 = 2.f * gv * dSxx, du3 = gv * gv * dSxx;
        float u3 = 0.f, sada = 0.f;
        float4 da[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const f32x4 ta = mm_lds_finish(m[2 * ct], m[2 * ct + 1], av[0], av[1]);
          // TW dzraw: rows k = 16 ct + r of TW (latent-major) from the TW^T image: entry (k, dd) at [dd][k]
          float tw_[16];
          const unsigned tw_a = lt_a + 4 * (FK * LD32 + 16 * ct + r);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int x = 0; x < 4; ++x) lds_rd1_issue(tw_[4 * j + x], tw_a + 4 * ((16 * j + 4 * q + x) * LD32));
          lds_wait();
#pragma unroll
          for (int i = 0; i < 16; ++i) lds_use(tw_[i]);
          kf_bf16x8 ah0, al0, ah1, al1;
          kf_split8(make_float4(tw_[0], tw_[1], tw_[2], tw_[3]), make_float4(tw_[4], tw_[5], tw_[6], tw_[7]), ah0, al0);
          kf_split8(make_float4(tw_[8], tw_[9], tw_[10], tw_[11]), make_float4(tw_[12], tw_[13], tw_[14], tw_[15]), ah1, al1);
          f32x4 twd;
          {                                                            // tile_fast.hip::mmT_split<4>: chunk pairs (0, 1) and (2, 3) chained
            kf_bf16x8 ph, pl;
            kf_split8(dzr[0], dzr[1], ph, pl);
            twd = f32x4{0.f, 0.f, 0.f, 0.f};
            twd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al0, ph, twd, 0, 0, 0);
            twd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, pl, twd, 0, 0, 0);
            twd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, ph, twd, 0, 0, 0);
            kf_split8(dzr[2], dzr[3], ph, pl);
            twd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al1, ph, twd, 0, 0, 0);
            twd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, pl, twd, 0, 0, 0);
            twd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, ph, twd, 0, 0, 0);
          }
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float ac = at(av[ct], x);
            u3 += ta[x] * ac;
            float d = 0.f;
            if (ok) {
              d = gv * twd[x] + du1 * at(tb[ct], x) + du2 * at(lg[ct], x) + 2.f * du3 * ta[x];
              sada += ac * d;
              at(ck[ct], x) += du1 * ac;
            }
            at(da[ct], x) = d;
          }
        }
        u3 = qsum4(u3); sada = qsum4(sada);
        if (ok && q == qsel) sdg += dSx * fC * u1 + dSxx * (2.f * u2 + 2.f * gv * u3) + dgr;
        if (ok) {
          float4 v0[2], v1[2], v2[2];
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              const float ac = at(av[j], x);
              at(v0[j], x) = du2 * ac + ac * (at(da[j], x) - sada); at(v1[j], x) = du3 * ac; at(v2[j], x) = gv * ac;
            }
          const long so = tok * t.KLp + (long)l * FK, sa = (long)l * t.aL + tok * FK;      // rows of dL2x ; planes of aw / ag
          st_seg<__bf16>((__bf16*)dL2x + so, v0[0], v0[1], q); st_seg<__bf16>((__bf16*)a.aw + sa, v1[0], v1[1], q); st_seg<__bf16>((__bf16*)a.ag + sa, v2[0], v2[1], q);
        }
      }
      // PRB_NST_X / PRB_NST_U store instructions per tile, exactly
      if (ok) st_row<__bf16, E>((__bf16*)dZx + tok * DZ, e, q, dzr);
    }
    // ---- end of the virtual block: the per-token sums over the experts; the folds ----
    const bool frame_ends = vbb + 1 == bps, final = vb + 1 == vb1, fend = frame_ends || final;
    if (lat && fend) {               // this frame's dtb sums of the wave: [2 x 64 | 32 dtb | gate]
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float4 v;
#pragma unroll
        for (int x = 0; x < 4; ++x) at(v, x) = rsum16(at(ck[j], x));
        if (r == 0) lds_wr16(fold_a + 4 * (wave * PRB_FOLD + 2 * FDD + 16 * j + 4 * q), v);
        ck[j] = zero4();
      }
    }
    if (final) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float4 v;
#pragma unroll
          for (int x = 0; x < 4; ++x) at(v, x) = rsum16(at(k ? cs1[c] : cs0[c], x));
          if (r == 0) lds_wr16(fold_a + 4 * (wave * PRB_FOLD + k * FDD + 16 * c + 4 * q), v);
        }
      const float g = wave_sum(sdg);
      if (lane == 0) lds_wr1(fold_a + 4 * (wave * PRB_FOLD + 2 * FDD + FK), g);
    }
    lds_barrier();
    {                                // dL2x[tok][KL] = sum_e dSx, [KL + 1] = 1 ; rs2x = 2 sum_e dSxx   (expert order)
      for (int tl = threadIdx.x; tl < n_end - n_beg; tl += NTHR) {
        float v[2 * E];
#pragma unroll
        for (int i = 0; i < 2 * E; ++i) lds_rd1_issue(v[i], ds_a + 4 * (tl * E * 2 + i));
        lds_wait();
        float accx = 0.f, accxx = 0.f;
#pragma unroll
        for (int ee = 0; ee < E; ++ee) { lds_use(v[2 * ee]); lds_use(v[2 * ee + 1]); accx += v[2 * ee]; accxx += v[2 * ee + 1]; }
        const long tok = (long)s * N + n_beg + tl;
        *(unsigned*)(dL2x + tok * t.KLp + t.KL) = (unsigned)f2bf(accx) | ((unsigned)f2bf(1.f) << 16);      // (KL even: 4-byte aligned)
        rs2x[tok] = 2.f * accxx;
      }
    }
    if (ts == 0) {
      const int dd = lane, col = (dd >> 5) * (E * FDG) + e * FDG + (dd & 31);
      float c0 = 0.f, c1 = 0.f, dtb = 0.f, g = 0.f;
      if (final || (fend && lat)) {
        float w[4][NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          const unsigned b = fold_a + 4 * ((u * E + e) * PRB_FOLD);
          lds_rd1_issue(w[0][u], b + 4 * dd); lds_rd1_issue(w[1][u], b + 4 * (FDD + dd));
          lds_rd1_issue(w[2][u], b + 4 * (2 * FDD + (dd & 31))); lds_rd1_issue(w[3][u], b + 4 * (2 * FDD + FK));
        }
        lds_wait();
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          lds_use(w[0][u]); lds_use(w[1][u]); lds_use(w[2][u]); lds_use(w[3][u]);
          c0 += w[0][u]; c1 += w[1][u]; dtb += w[2][u]; g += w[3][u];
        }
      }
      colpart[((long)vb * 4 + 0) * (E * FDD) + col] = final ? c0 : 0.f;
      colpart[((long)vb * 4 + 1) * (E * FDD) + col] = final ? c1 : 0.f;
      if (lane == 0) blkscal[((long)vb * E + e) * 4 + 3] = (final && lat) ? g : 0.f;
      if (lat && lane < FK) a.dtbp[(long)vb * t.KL + (long)l * FK + lane] = fend ? dtb : 0.f;
    }
    if (frame_ends && !final && El > 0) { request_lat(s + 1); wait_vm<0>(); }       // (every wave is past the barrier above: nobody reads frame s's constants any more)
    lds_barrier();
    if (frame_ends) { vbb = 0; ++s; } else ++vbb;
  }
}

template <int E>
int launch_prb(const SPreBArgs& a, int gx, const Plan& pl, char* saved, char* scratch, hipStream_t st) {
  static LdsAttrOnce attr;
  AVMOE_TRY(attr.ensure((const void*)kfs_pre_bwd<E>, 160 * 1024, "pre_small_bwd (streaming)"));
  hipLaunchKernelGGL((kfs_pre_bwd<E>), dim3((unsigned)gx), dim3(WS<E>::NTHR), pl.d.excl ? (size_t)160 * 1024 : prb_lds<E>(a.t.El, a.t.per), st, a,
                     (const unsigned short*)(saved + pl.o_Z), (const float*)(saved + pl.o_wsum), (const float*)(saved + pl.o_dconst), (const float*)(saved + pl.o_rmu),
                     (const float*)(saved + pl.o_bn1), (const float*)(scratch + pl.o_dsm), (const unsigned short*)(scratch + pl.o_dzp), (unsigned short*)(scratch + pl.o_Zw),
                     (unsigned short*)(scratch + pl.o_dL2x), (float*)(scratch + pl.o_rs2x), (float*)(scratch + pl.o_colpart), (float*)(scratch + pl.o_blkscal));
  AVMOE_CHECK_LAUNCH("pre_small_bwd (streaming)");
  return OK;
}

FastDims make_fd_s(const Dims& d, int per) {
  FastDims t;
  t.S = d.S; t.N = d.N; t.C = d.C; t.El = d.El; t.KL = d.KL; t.KLT = d.KLT; t.KLp = d.KLp; t.KPp = d.KPp; t.NT = d.NT; t.per = per; t.aL = d.aL;
  return t;
}

}  // namespace

// Sites the streaming form serves: the tuned shape in bf16 from 2048 tokens on.  (Round 6 first drew the line at 32 768 tokens -- a persistent
// block amortises its prologue over its range -- but measured at the reference's batch of 2 clips, 20 480 / 3920 tokens, every pass is as
// fast or faster in this form too: mid_bwd 28.6 -> 25.0 us, post_small 19.8 -> 17.0, pre_small_bwd + pre_lat_bwd 19.7 + 19.7 -> 33.8,
// post_small_bwd + Gram 33.6 + 16.0 -> 39.1; 1.615 -> 1.567 ms of kernel time per pair-step and four launches less.)
bool tile_stream_ok(const Dims& d) {
  const unsigned hooks = test_hook_mask();            // (include/avmoe.h: avmoe_test_hooks -- small test shapes through these kernels / the A/B against tile_fast.hip)
  return tile_fast_ok(d) && d.bf16 && d.zsz == 2 && (d.NT >= 2048 || (hooks & HOOK_KFS_FORCE)) && !(hooks & HOOK_KFS_OFF);
}

static bool psf_geom(const Dims& d, int* gx, int* nfr, int* per) {
  const int cus = cu_count();
  if (cus <= 0) return false;
  const int bps = d.nblk_tok / d.S;
  *gx = std::min(cus, d.nblk_tok);
  *nfr = cdiv(cdiv(d.nblk_tok, *gx) + 1, bps) + 1;
  *per = (int)round_up(cdiv(d.N, bps), 16);
  const size_t lds = d.E == 4 ? psf_lds<4>(*nfr, *per) : d.E == 2 ? psf_lds<2>(*nfr, *per) : psf_lds<3>(*nfr, *per);
  return lds <= 160 * 1024;
}
bool kfs_serves_post_small(const Dims& d) { int a, b, c; return tile_stream_ok(d) && psf_geom(d, &a, &b, &c); }
// 0 = launched, 1 = not served (the caller runs tile_fast.hip's kernel), < 0 error
int kfs_post_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st) {
  const Dims& d = pl.d;
  if (!kfs_serves_post_small(d)) return 1;
  int gx, nfr, per;
  psf_geom(d, &gx, &nfr, &per);
  SPostArgs a;
  for (int e = 0; e < MAX_E; ++e) { a.gate.p[e] = prm.e[e].gate; a.relu_of_e[e] = d.relu_of_e[e]; }
  a.t = make_fd_s(d, per); a.ln_post = d.ln_post; a.use_gate = d.use_gate && !d.gate_w; a.ln_eps = d.ln_eps;
  a.bps = d.nblk_tok / d.S; a.nvb = d.nblk_tok; a.nfr = nfr;
  if (d.E == 4) return launch_psf<4>(a, gx, pl, saved, st);
  if (d.E == 2) return launch_psf<2>(a, gx, pl, saved, st);
  return launch_psf<3>(a, gx, pl, saved, st);
}

// (cross-modal experts need the fused hop-2 logits of the down projection's pass: Dims::fuse_l2; no x + g xr experts)
bool kfs_serves_pre_small(const Dims& d) {
  if (!tile_stream_ok(d) || d.nxn || (d.El > 0 && !d.fuse_l2) || cu_count() <= 0) return false;
  const size_t lds = d.E == 4 ? prs_lds<4>(d.El) : d.E == 2 ? prs_lds<2>(d.El) : prs_lds<3>(d.El);
  return lds <= 160 * 1024;
}
// 0 = launched, 1 = not served (the caller runs tile_fast.hip's kernel), < 0 error
int kfs_pre_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st) {
  const Dims& d = pl.d;
  if (!kfs_serves_pre_small(d)) return 1;
  const int bps = d.nblk_tok / d.S;
  SPreArgs a;
  for (int e = 0; e < MAX_E; ++e) { a.glat.p[e] = prm.e[e].gate_lat; a.lat_of_e[e] = d.lat_of_e[e]; }
  a.t = make_fd_s(d, (int)round_up(cdiv(d.N, bps), 16)); a.ln_before = d.ln_before; a.ln_eps = d.ln_eps; a.bps = bps; a.nvb = d.nblk_tok;
  a.L2g = d.El > 0 ? (const float*)(scratch + pl.o_L2g) : nullptr; a.L2w = d.El > 0 ? (float*)(saved + pl.o_L2) : nullptr;
  const int gx = std::min(cu_count(), a.nvb);
  if (d.E == 4) return launch_prs<4>(a, gx, pl, saved, scratch, st);
  if (d.E == 2) return launch_prs<2>(a, gx, pl, saved, scratch, st);
  return launch_prs<3>(a, gx, pl, saved, scratch, st);
}

// pre_small_bwd + the cross-modal experts' hop-2 block in one pass
bool kfs_serves_pre_bwd(const Dims& d) {
  if (!tile_stream_ok(d) || d.nxn || cu_count() <= 0) return false;
  const int per = (int)round_up(cdiv(d.N, d.nblk_tok / d.S), 16);
  const size_t lds = d.E == 4 ? prb_lds<4>(d.El, per) : d.E == 2 ? prb_lds<2>(d.El, per) : prb_lds<3>(d.El, per);
  return lds <= 160 * 1024;
}
// 0 = launched (both kf_pre_small_bwd's and kf_pre_lat_bwd's work), 1 = not served, < 0 error
int kfs_pre_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st) {
  const Dims& d = pl.d;
  if (!kfs_serves_pre_bwd(d)) return 1;
  const int bps = d.nblk_tok / d.S;
  SPreBArgs a;
  for (int e = 0; e < MAX_E; ++e) { a.glat.p[e] = prm.e[e].gate_lat; a.lat_of_e[e] = d.lat_of_e[e]; }
  a.t = make_fd_s(d, (int)round_up(cdiv(d.N, bps), 16)); a.ln_before = d.ln_before; a.use_bn = d.use_bn; a.bn_train = d.use_bn && d.training;
  a.bps = bps; a.nvb = d.nblk_tok;
  a.L2 = (const float*)(saved + pl.o_L2); a.TT = (const float*)(saved + pl.o_TT); a.TW = (const float*)(saved + pl.o_TW); a.Tsum = (const float*)(saved + pl.o_Tsum);
  a.ain = (const unsigned short*)(saved + pl.o_a); a.aw = (unsigned short*)(scratch + pl.o_aw); a.ag = (unsigned short*)(scratch + pl.o_ag);
  a.dtbp = (float*)(scratch + pl.o_dtbp);
  const int gx = std::min(cu_count(), a.nvb);
  if (d.E == 4) return launch_prb<4>(a, gx, pl, saved, scratch, st);
  if (d.E == 2) return launch_prb<2>(a, gx, pl, saved, scratch, st);
  return launch_prb<3>(a, gx, pl, saved, scratch, st);
}

bool kfs_serves_mid_bwd(const Dims& d) { return tile_stream_ok(d) && cu_count() > 0; }
// 0 = launched, 1 = not served (the caller runs tile_fast.hip's kernel), < 0 error
int kfs_mid_bwd(const Plan& pl, char* saved, char* scratch, hipStream_t st) {
  const Dims& d = pl.d;
  if (!kfs_serves_mid_bwd(d)) return 1;
  const int bps = d.nblk_tok / d.S;
  SMidBArgs a;
  for (int e = 0; e < MAX_E; ++e) a.relu_of_e[e] = d.relu_of_e[e];
  a.t = make_fd_s(d, (int)round_up(cdiv(d.N, bps), 16)); a.moments = d.use_bn && d.training; a.bps = bps; a.nvb = d.nblk_tok;
  const int gx = std::min(cu_count(), a.nvb);
  if (d.E == 4) return launch_mdb<4>(a, gx, pl, saved, scratch, st);
  if (d.E == 2) return launch_mdb<2>(a, gx, pl, saved, scratch, st);
  return launch_mdb<3>(a, gx, pl, saved, scratch, st);
}

// launch geometry of the streaming kernels: persistent blocks, frames a block's range of virtual blocks can touch
struct SGeom { int bps, nvb, gx, nfr; };
static bool psb_geom(const Dims& d, SGeom* g) {
  const int cus = cu_count();
  if (cus <= 0) return false;
  const int ns = d.E == 3 ? 2 : 8 / d.E;
  g->bps = d.nblk_tok / d.S; g->nvb = d.nblk_tok;
  g->gx = std::min(std::min(cus, g->nvb), GRAM_SLABS / ns);
  g->nfr = cdiv(cdiv(g->nvb, g->gx) + 1, g->bps) + 1;
  const size_t lds = d.E == 4 ? psb_lds<4>(g->nfr) : d.E == 2 ? psb_lds<2>(g->nfr) : psb_lds<3>(g->nfr);
  return lds <= 160 * 1024;
}
// (with LayerNorm-post it also leaves dGq, the weighted Gram products: the caller skips gram.hip's pass)
bool kfs_serves_post_small_bwd(const Dims& d, int dap16) {
  SGeom g;
  return tile_stream_ok(d) && dap16 && d.gram64 && psb_geom(d, &g);
}

// 0 = launched, 1 = not served (the caller runs tile_fast.hip's kernel), < 0 error
int kfs_post_small_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st, int dap16) {
  const Dims& d = pl.d;
  if (!kfs_serves_post_small_bwd(d, dap16)) return 1;
  SGeom gm;
  if (!psb_geom(d, &gm)) { set_last_error("post_small_bwd: device query"); return ERR_LAUNCH; }
  const int bps = gm.bps, gx = gm.gx;
  SPostBArgs a;
  for (int e = 0; e < MAX_E; ++e) { a.gate.p[e] = prm.e[e].gate; a.relu_of_e[e] = d.relu_of_e[e]; }
  a.t = make_fd_s(d, (int)round_up(cdiv(d.N, bps), 16)); a.ln_post = d.ln_post; a.use_gate = d.use_gate && !d.gate_w;
  a.bps = bps; a.nvb = gm.nvb; a.nfr = gm.nfr;
  a.dApx = (const float*)(scratch + pl.o_dApx); a.dapw = d.E * d.dgp; a.gpart = (float*)(scratch + pl.o_gpartT);
  if (d.E == 4) return launch_psb<4>(a, gx, pl, saved, scratch, st);
  if (d.E == 2) return launch_psb<2>(a, gx, pl, saved, scratch, st);
  return launch_psb<3>(a, gx, pl, saved, scratch, st);
}

}  // namespace avmoe
