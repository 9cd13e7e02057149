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
// (in-order counter: everything but those may still be in flight).  NL / NST: ring loads / stores per wave and tile.
template <int NL, int NST>
__device__ __forceinline__ void wait_tile(int newer, int stored) {
  static_assert(KFS_P >= 1 && KFS_P <= 3, "tiles ahead");
  if (!KFS_EXACT) stored = 0;
  if (newer <= 0) { wait_vm<0>(); return; }          // (the last tiles of the block: no count to rely on behind a flush)
#define KFS_W(M_, K_) if (newer == M_ && stored == K_) { wait_vm<M_ * NL + K_ * NST>(); return; }
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
  for (int vb = vb0; vb < vb1; ++vb, s += (vbb + 1 == bps), vbb = (vbb + 1 == bps) ? 0 : vbb + 1) {
    const int n_beg = vbb * per, n_end = min(N, n_beg + per);
    const float qv = lds_rd1(qv_a + 4 * ((s - s_first) * E + e));
    float sdq = 0.f, sdSo = 0.f, sdSoo = 0.f;
    float4 cs0[4], cs1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { cs0[c] = zero4(); cs1[c] = zero4(); }
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
    // the virtual block's partial sums: column sums of dSo z' and dSoo z' (slots 0, 1 of colpart), three scalars per expert.  Two LDS
    // buffers, ONE barrier per virtual block: buffer k & 1 is rewritten two virtual blocks later, behind the next barrier, which the
    // reading waves pass only after they have read it.
    const unsigned fb_a = fold_a + 4 * ((vb - vb0) & 1) * NW * fold_stride<2>();
    fold_put_cols<2>(fb_a, 0, cs0, wave, lane); fold_put_cols<2>(fb_a, 1, cs1, wave, lane);
    fold_put_scalar<2>(fb_a, 0, wave_sum(sdq), wave, lane); fold_put_scalar<2>(fb_a, 1, wave_sum(sdSo), wave, lane); fold_put_scalar<2>(fb_a, 2, wave_sum(sdSoo), wave, lane);
    lds_barrier();
    if (ts == 0) {
      const int dd = lane, col = (dd >> 5) * (E * FDG) + e * FDG + (dd & 31);
      const int idx[3] = {dd, FDD + dd, 2 * FDD + min(lane, 2)};
      float v[3];
      fold_get<E, NS, 2, 3>(fb_a, idx, e, v);
      colpart[((long)vb * 4 + 0) * (E * FDD) + col] = v[0];
      colpart[((long)vb * 4 + 1) * (E * FDD) + col] = v[1];
      if (lane < 3) blkscal[((long)vb * E + e) * 4 + lane] = v[2];
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

FastDims make_fd_s(const Dims& d, int per) {
  FastDims t;
  t.S = d.S; t.N = d.N; t.C = d.C; t.El = d.El; t.KL = d.KL; t.KLT = d.KLT; t.KLp = d.KLp; t.KPp = d.KPp; t.NT = d.NT; t.per = per; t.aL = d.aL;
  return t;
}

}  // namespace

// Sites the streaming form serves: the tuned shape in bf16, enough virtual blocks to give every CU a few (a persistent block
// amortises its prologue over them); smaller sites keep tile_fast.hip's grid.
bool tile_stream_ok(const Dims& d) {
  const unsigned hooks = test_hook_mask();            // (include/avmoe.h: avmoe_test_hooks -- small test shapes through these kernels / the A/B against tile_fast.hip)
  return tile_fast_ok(d) && d.bf16 && d.zsz == 2 && (d.NT >= 32768 || (hooks & HOOK_KFS_FORCE)) && !(hooks & HOOK_KFS_OFF);
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
