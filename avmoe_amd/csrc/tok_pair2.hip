// The backward's two token contractions against X -- dWt = dZx^T X (over all tokens) and dT[s] = dL2[s]^T X[s] (per frame) -- as ONE
// streaming pass with every accumulator in registers (round 4; the tiled form is gemm.hip::gemm_tokpair_kernel, which stays for every
// other shape).  Built on dpost_pair.hip's recipe: a persistent block of eight waves per CU (two per SIMD, 256 registers) streams 32-token
// tiles of X (384 channels of the group), dZx (128 columns) and dL2 (64 latent rows) through the LDS by direct global -> LDS loads and
// contracts over the tokens with BOTH operands read transposed (inline-assembly ds_read_b64_tr_b16: the compiler would drain the loads in
// flight in front of the intrinsic).  Wave w owns channel tiles 3 w .. 3 w + 2 against all twelve row tiles (8 of dWt, 4 of dT): 36
// accumulator tiles = 144 registers, 15 fragment reads for 36 matrix instructions per 32 tokens.
//
// A block takes the contiguous tile range [T b / nb, T (b + 1) / nb) of the T = frames x tiles-per-frame tiles (nb <= frames: a frame spans
// at most two blocks); dT of a frame is flushed when the frame ends -- straight to dT[s] for the part that begins with the frame's first
// token, to a scratch slab for a block's leading frame part, which kk_tp2_finish adds (fixed order).  The blocks' partial dWt go to the slab
// workspace and are summed there in block order as well (no float atomics).
//
// Round 5 (second form): four LDS buffers of 32 tokens with counted waits instead of two of 64 behind __syncthreads, lane offsets of the
// direct loads computed once, counted lgkmcnt waits (six fragment reads stay in flight behind the products), frames of ANY length (the ragged
// last tile of a frame has its dZx / dL2 fragments masked) and tile ranges instead of whole half frames -- so that the 196-token visual
// frames of cfg-2 are served (54 + 25 us against the tiled form's 69 + 20); at the 1024-token audio frames the two forms measure the same
// (172 - 185 us on the same boxes: the pass is bound by its three-tensor tile stream -- 180 us with no fragment read and no product at all,
// X alone 143 us, dZx + dL2 alone 75 us: 256- and 144-byte row pieces -- and by ~30 us of slab traffic at its end).
#include "gemm.h"
#include "common.h"
#include "prof.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>

#ifndef TP2_AUX
#define TP2_AUX 0      // cache policy of the direct loads (common.h::AVMOE_LDS_AUX): the non-temporal hint measured neutral or worse here (its X is re-read by the next kernel of the chain)
#endif

namespace avmoe {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct TP2Args {
  const char* X; long ldx;          // bf16 [tokens][ldx], group g at column g * 384
  const char* dZx; long ldz;        // bf16 [tokens][ldz], group g at column g * 128
  const char* dL2; long ldl;        // bf16 [tokens][ldl >= 72]: latent rows 0 .. KL - 1 (<= 64), shared by the groups
  float* dT; long ldt;              // fp32 [frame][KL][ldt], group g at column g * 384
  float* slabW; float* slabT;       // [block][g][128][384] partial dWt ; [block][g][64][384] leading frame parts
  char* dump;                       // >= 16 writable bytes nobody reads
  int N, tpf, ntiles, KL;           // tokens per frame, 32-token tiles per frame (the last one ragged), tiles in all (frames x tpf)
};

// Round 5: 32-token tiles (one K step) in FOUR LDS buffers with counted waits (hop1_stream.hip::kk_hop1_yk's recipe: three tiles in flight
// while one is multiplied, the loop's barrier a bare s_barrier behind `s_waitcnt vmcnt(n)`; the first form had 64-token tiles in two buffers
// behind __syncthreads: one tile in flight), the lane offsets of the direct loads computed once (dx_stream3.hip), frames of any length (the
// ragged last tile of a frame: its rows beyond the frame read whatever follows and the dZx / dL2 fragments are zeroed there), and the
// blocks' tile ranges no longer whole half frames (a block takes tiles [T b / nb, T (b + 1) / nb): a frame spans at most two blocks as long
// as a block has at least one frame's tiles).
#ifndef TP2_DISSECT
#define TP2_DISSECT 0      // development builds (timing only): bit 0 = no matrix instructions, bit 1 = no fragment reads either, bit 2 = no direct loads
#endif
constexpr int BM = 32, NTHR = 512, NBUF = 4;
constexpr int CHX = 49, CHZ = 17, CHL = 10;                                    // 16-byte chunks per LDS row (the last one a pad)
constexpr int RBX = 16 * CHX, RBZ = 16 * CHZ, RBL = 16 * CHL;                  // LDS row pitches
constexpr int PX = (BM * CHX + 63) / 64, PZ = (BM * CHZ + 63) / 64, PL = (BM * CHL + 63) / 64;      // 1 KB pieces per sub-tile: 25 / 9 / 5
constexpr int OFFZ = PX * 1024, OFFL = OFFZ + PZ * 1024, BUF = OFFL + PL * 1024;
constexpr int TP2_LDS = NBUF * BUF;
constexpr int NFL = 48;                                                        // stores of a dT flush per wave (4 row tiles x 3 channel tiles x 4 rows)
static_assert(TP2_LDS <= 160 * 1024, "the buffers fit one CU's LDS");

template <int OFF>
__device__ __forceinline__ void tr_issue(u32x2& d, unsigned addr) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory"); }
template <int OFF, int ROWB>        // one 16-column fragment: token rows 8 q .. 8 q + 7 (+ the lane's row of four), both halves
__device__ __forceinline__ void tr_frag2(u32x2 (&f)[2], unsigned base) { tr_issue<OFF>(f[0], base); tr_issue<OFF + 4 * ROWB>(f[1], base); }
__device__ __forceinline__ void tr_wait3(u32x2 (&x)[3][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]) :: "memory");
}
__device__ __forceinline__ void tr_wait6(u32x2 (&x)[3][2], u32x2 (&y)[3][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]),
               "+v"(y[0][0]), "+v"(y[0][1]), "+v"(y[1][0]), "+v"(y[1][1]), "+v"(y[2][0]), "+v"(y[2][1]) :: "memory");
}
template <int N> __device__ __forceinline__ void tr_wait3n(u32x2 (&x)[3][2]) {      // ... until at most N reads are pending
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void tr_wait6n(u32x2 (&x)[3][2], u32x2 (&y)[3][2]) {
  asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]),
               "+v"(y[0][0]), "+v"(y[0][1]), "+v"(y[1][0]), "+v"(y[1][1]), "+v"(y[2][0]), "+v"(y[2][1]) : "n"(N) : "memory");
}
__device__ __forceinline__ bf16x8 tr_pack(const u32x2 (&f)[2]) { return __builtin_bit_cast(bf16x8, u32x4{f[0][0], f[0][1], f[1][0], f[1][1]}); }
__device__ __forceinline__ bf16x8 tr_pack_masked(const u32x2 (&f)[2], const u32x4& mk) { return __builtin_bit_cast(bf16x8, u32x4{f[0][0], f[0][1], f[1][0], f[1][1]} & mk); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// s_waitcnt vmcnt(n) for a run-time (wave-uniform) n: the immediate has to be a constant
__device__ __forceinline__ void wait_vm_n(int n) {
  switch (n) {
#define W1(k) case k: wait_vm<k>(); break;
#define W8(k) W1(k) W1(k + 1) W1(k + 2) W1(k + 3) W1(k + 4) W1(k + 5) W1(k + 6) W1(k + 7)
    W8(0) W8(8) W8(16) W8(24) W8(32) W8(40) W8(48) W1(56) W1(57) W1(58) W1(59) W1(60) W1(61) W1(62)
#undef W8
#undef W1
    default: wait_vm<63>(); break;        // (the counter's ceiling: waits for more than necessary, never for less)
  }
}

// the A fragments (row tiles of the result) of the 32-token step, three at a time: 0 - 7 = dZx column tiles, 8 - 11 = dL2 column tiles
template <int G3>
__device__ __forceinline__ void issue_a3(u32x2 (&f)[3][2], unsigned lz, unsigned ll) {
  if constexpr (G3 < 2) {
    tr_frag2<(3 * G3 + 0) * 32, RBZ>(f[0], lz); tr_frag2<(3 * G3 + 1) * 32, RBZ>(f[1], lz); tr_frag2<(3 * G3 + 2) * 32, RBZ>(f[2], lz);
  } else if constexpr (G3 == 2) {
    tr_frag2<6 * 32, RBZ>(f[0], lz); tr_frag2<7 * 32, RBZ>(f[1], lz); tr_frag2<0, RBL>(f[2], ll);
  } else {
    tr_frag2<1 * 32, RBL>(f[0], ll); tr_frag2<2 * 32, RBL>(f[1], ll); tr_frag2<3 * 32, RBL>(f[2], ll);
  }
}
__device__ __forceinline__ void issue_b3(u32x2 (&f)[3][2], unsigned lx) {
  tr_frag2<0, RBX>(f[0], lx); tr_frag2<32, RBX>(f[1], lx); tr_frag2<64, RBX>(f[2], lx);
}

__global__ void __launch_bounds__(NTHR, 1) kk_tok_pair2(const TP2Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  const int g = blockIdx.y;
  const char* Xb = p.X + (long)g * 384 * 2;
  const char* Zb = p.dZx + (long)g * 128 * 2;
  const char* Lb = p.dL2;
  const long ldx = p.ldx, ldz = p.ldz, ldl = p.ldl;

  f32x4 acc[12][3];                  // row tiles 0 - 7: dWt (kept for the whole block), 8 - 11: dT (per frame) ; x this wave's three channel tiles
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // One buffer image = [X 25 | dZx 9 | dL2 5] pieces of 1 KB; piece P = wave + 8 i (waves 0 .. 6: five pieces, wave 7: four).  Where a lane's
  // 16 bytes of piece P come from, relative to the tile's first row of that tensor, does not depend on the tile: computed once.
  constexpr int B1 = PX, B2 = B1 + PZ, B3 = B2 + PL, NPW = (B3 + 7) / 8;
  auto off_p = [&](int P, int ln, int last) {               // (P: wave-uniform) ; rows beyond `last` re-read it
    if (P < B1) { const int slot = 64 * P + ln, row = min(slot / CHX, last), cc = min(slot % CHX, CHX - 2); return (unsigned)((row * ldx + cc * 8) * 2); }
    if (P < B2) { const int slot = 64 * (P - B1) + ln, row = min(slot / CHZ, last), cc = min(slot % CHZ, CHZ - 2); return (unsigned)((row * ldz + cc * 8) * 2); }
    if (P < B3) { const int slot = 64 * (P - B2) + ln, row = min(slot / CHL, last), cc = min(slot % CHL, CHL - 2); return (unsigned)((row * ldl + cc * 8) * 2); }
    return 0u;
  };
  unsigned voff[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    voff[i] = off_p(wave + 8 * i, lane, BM - 1);
    asm volatile("" : "+v"(voff[i]));                       // (opaque: one register each for the whole kernel, not re-derived per tile)
  }
  int nl = 0;                                               // direct loads per tile of this wave: (B3 - 1 - wave) / 8 + 1
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int P = wave + 8 * i, t = P < B1 ? 0 : P < B2 ? 1 : P < B3 ? 2 : -1;
    if (!(t < 0 || (TP2_DISSECT & 4) || ((TP2_DISSECT & 8) && t != 0) || ((TP2_DISSECT & 16) && t == 0))) ++nl;
  }
  const int nfr = p.ntiles / p.tpf;
  auto gload = [&](int buf, int tile) {
    const int fs = tile / p.tpf, fj = tile - fs * p.tpf;
    const long m0 = (long)fs * p.N + (long)fj * BM;         // first token of the tile
    const int last = min(p.N - fj * BM, BM) - 1;            // last row of the tile inside the frame
    char* dst = smem + buf * BUF + 1024 * wave;
    const char* base[3] = {Xb + m0 * ldx * 2, Zb + m0 * ldz * 2, Lb + m0 * ldl * 2};
    // A ragged tile's rows beyond the frame read the next frame's first rows (their dZx / dL2 entries are masked out of the products); the
    // last frame has no next one: its ragged tile takes clamped addresses, computed on the spot (once per launch, in one block).
    if (last < BM - 1 && fs == nfr - 1) {
#pragma unroll 1
      for (int i = 0; i < NPW; ++i) {
        const int P = wave + 8 * i;
        if (P >= B3) break;
        __builtin_amdgcn_global_load_lds((gptr_t)(base[P < B1 ? 0 : P < B2 ? 1 : 2] + off_p(P, lane, last)), (lptr_t)(dst + 8192 * i), 16, 0, TP2_AUX);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int P = wave + 8 * i;
      int t;                                                // which tensor (a constant after unrolling except in the rounds that hold a boundary)
      if (8 * i + 8 <= B1) t = 0;
      else if (8 * i >= B1 && 8 * i + 8 <= B2) t = 1;
      else if (8 * i >= B2 && 8 * i + 8 <= B3) t = 2;
      else t = P < B1 ? 0 : P < B2 ? 1 : P < B3 ? 2 : -1;
      if (t < 0 || (TP2_DISSECT & 4) || ((TP2_DISSECT & 8) && t != 0) || ((TP2_DISSECT & 16) && t == 0)) continue;      // (bits 3 / 4: the X tiles only / the dZx and dL2 tiles only)
      unsigned o = voff[i];
      asm volatile("" : "+v"(o));                           // (the zero-extension stays here, beside the scalar base: `scalar base + 32-bit lane offset` loads)
      __builtin_amdgcn_global_load_lds((gptr_t)(base[t] + o), (lptr_t)(dst + 8192 * i), 16, 0, TP2_AUX);
    }
  };

  const int t0 = (int)((long)p.ntiles * blockIdx.x / gridDim.x), t1 = (int)((long)p.ntiles * (blockIdx.x + 1) / gridDim.x);
  if (t0 < t1) {
#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j)
      if (t0 + j < t1) gload(j, t0 + j);
  }
  int part0 = t0;                    // first tile of the frame part being accumulated in acc[8 .. 11]
  int fl1 = 0, fl2 = 0, fl3 = 0;     // flush stores issued one / two / three iterations ago
  for (int it = 0, tile = t0; tile < t1; ++it, ++tile) {
    // In-order counter, issue order per iteration i: [loads of tile i + 3] [the stores of a dT flush, if the frame part ended].  Tile `tile` has
    // landed once everything but what was issued after its loads is complete: the loads of the (up to two) tiles requested after it and the
    // flush stores of the last three iterations.
    wait_vm_n(min(NBUF - 2, t1 - 1 - tile) * nl + fl1 + fl2 + fl3);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* sX = smem + (it % NBUF) * BUF;
    if (tile + NBUF - 1 < t1) gload((it + NBUF - 1) % NBUF, tile + NBUF - 1);      // (its buffer was read in the previous iteration: every wave has passed this barrier since)
    const int fj = tile % p.tpf, valid = p.N - fj * BM;     // rows of this tile inside the frame (>= BM: all of them)
    {
      const unsigned l0 = (unsigned)(size_t)(lptr_t)sX;
      const unsigned lx = l0 + (8 * q + (r >> 2)) * RBX + (3 * wave * 16 + 4 * (r & 3)) * 2;
      const unsigned lz = l0 + OFFZ + (8 * q + (r >> 2)) * RBZ + (4 * (r & 3)) * 2;
      const unsigned ll = l0 + OFFL + (8 * q + (r >> 2)) * RBL + (4 * (r & 3)) * 2;
      // tokens 8 q + j of the tile beyond `valid` are not data (the ragged last tile of a frame): their dZx / dL2 entries are zeroed
      const int nv = valid - 8 * q;
      u32x4 mk;
#pragma unroll
      for (int e = 0; e < 4; ++e) mk[e] = (2 * e + 1 < nv) ? 0xffffffffu : ((2 * e < nv) ? 0x0000ffffu : 0u);
      u32x2 fb[3][2], fa0[3][2], fa1[3][2];
      auto mm = [&](int i0, const u32x2 (&fa)[3][2]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const bf16x8 a = tr_pack_masked(fa[i], mk);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            if (TP2_DISSECT & 1) acc[i0 + i][c][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, a)[0] ^ fb[c][0][0]);
            else acc[i0 + i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, tr_pack(fb[c]), acc[i0 + i][c], 0, 0, 0);
          }
        }
      };
      // (the reads of the next three row tiles are in flight during the nine matrix instructions of the current three)
      if (!(TP2_DISSECT & 2)) {
        // six reads stay in flight behind the nine matrix instructions of the current three row tiles (counted waits: LDS answers in order)
        u32x2 fa2[3][2];
        issue_b3(fb, lx); issue_a3<0>(fa0, lz, ll); issue_a3<1>(fa1, lz, ll); tr_wait6n<6>(fb, fa0);
        mm(0, fa0); issue_a3<2>(fa2, lz, ll); tr_wait3n<6>(fa1);
        mm(3, fa1); issue_a3<3>(fa0, lz, ll); tr_wait3n<6>(fa2);
        mm(6, fa2); tr_wait3n<0>(fa0);
        mm(9, fa0);
      }
    }
    int fl = 0;
    if (tile + 1 == t1 || (tile + 1) % p.tpf == 0) {       // the frame (or the block's range) ends: flush dT of this part (block-uniform)
      const int s = tile / p.tpf;
      const bool first = part0 % p.tpf == 0;               // begins with the frame's first token: the frame's own rows ; else: a leading frame part
      float* dst = first ? p.dT + (long)s * p.KL * p.ldt + g * 384 : p.slabT + ((long)blockIdx.x * gridDim.y + g) * 64 * 384;
      const long ld = first ? p.ldt : 384;
      // exactly NFL store instructions per wave (rows beyond KL go to the dump word): the counted waits above rely on it
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int l = 16 * i + 4 * q + e;
            float* o = l < p.KL ? dst + l * ld + 16 * (3 * wave + c) + r : (float*)p.dump;
            asm volatile("global_store_dword %0, %1, off" :: "v"(o), "v"(acc[8 + i][c][e]) : "memory");
            acc[8 + i][c][e] = 0.f;
          }
        }
      part0 = tile + 1;
      fl = NFL;
    }
    fl3 = fl2; fl2 = fl1; fl1 = fl;
  }
  // lane (r, q): dWt[row 16 i + 4 q + e][channel 16 ct + r]
  float* sl = p.slabW + ((long)blockIdx.x * gridDim.y + g) * 128 * 384;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) sl[(16 * i + 4 * q + e) * 384 + 16 * (3 * wave + c) + r] = acc[i][c][e];
}

// blocks [0, nA): dWt = the blocks' partial sums in block order (four lanes per 4-element vector, as kk_dpair_reduce) ; blocks [nA, ..):
// dT[s] += the leading half frame of every block whose range begins inside frame s
__global__ void __launch_bounds__(256) kk_tp2_finish(const float* __restrict__ slabW, const float* __restrict__ slabT, int nb, int G, int nA,
                                                     float* __restrict__ dWt, float* __restrict__ dT, long ldt, int KL, int tpf, int ntiles) {
  if ((int)blockIdx.x < nA) {
    const long per = (long)G * 128 * 384, nvec = per / 4;
    const int lane = threadIdx.x & 63, part = lane >> 4;
    const long v = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (lane & 15);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (v < nvec) {
      const float* sl = slabW + v * 4;
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
    if (v < nvec && part == 0) *(f32x4*)(dWt + v * 4) = s;
    return;
  }
  // one block per (leading-half-frame block, group): 64 x 384 floats
  const int idx = blockIdx.x - nA, b = idx / G, g = idx % G;
  const long t0 = (long)ntiles * b / nb;                    // (kk_tok_pair2's tile ranges)
  if (t0 % tpf == 0) return;                               // this block's range begins with a frame
  const long s = t0 / tpf;
  const float* src = slabT + ((long)b * G + g) * 64 * 384;
  float* dst = dT + s * KL * ldt + g * 384;
  for (int i = threadIdx.x; i < KL * 96; i += blockDim.x) {
    const int l = i / 96, c4 = (i % 96) * 4;
    const f32x4 a = *(const f32x4*)(src + l * 384 + c4);
    f32x4* d = (f32x4*)(dst + l * ldt + c4);
    *d = *d + a;
  }
}

}  // namespace

// 0 = launched, 1 = shape not served (the caller runs gemm_tokpair), < 0 error
int k_tok_pair2(const void* X, long ldx, const void* dZx, long ldz, const void* dL2, long ldl, int S, int N, int G, int Cg, int M1, int KL,
                float* dWt, float* dT, float* slabs, size_t slab_cap, hipStream_t st) {
  if (Cg != 384 || M1 != 128 || KL < 1 || KL > 64 || ldl < 72 || N < 16 || S < 1 || ldx % 8 || ldz % 8 || ldl % 8 || !slabs ||
      ((uintptr_t)X % 16) || ((uintptr_t)dZx % 16) || ((uintptr_t)dL2 % 16) || ((uintptr_t)dWt % 16) || ((uintptr_t)dT % 16) || (G * 384) % 4)
    return 1;
  const bool force = (test_hook_mask() & HOOK_TOKPAIR2_FORCE) != 0;          // test hook (avmoe_test_hooks: tests / bench.py's parity leg switch it inside one process): small sites as well
  if (!force && (long)S * N < 32768) return 1;                              // small sites: the tiled form fills the chip better
  const int cus = cu_count();                             // (cached per device: common.cpp)
  if (cus <= 0) { set_last_error("tok_pair2: device query"); return ERR_LAUNCH; }
  const int tpf = (N + BM - 1) / BM;
  // one block per CU and group over contiguous tile ranges; a frame may span two blocks but no more (its leading part goes to a slab that
  // kk_tp2_finish adds): a block takes at least one frame's tiles
  const int nb = std::min(std::max(1, cus / G), S);
  if (!force && nb * G * 4 < cus * 3) return 1;             // ... unless that leaves a quarter of the chip idle
  const size_t need0 = (size_t)nb * G * (128 + 64) * 384, need = need0 + 64;
  if (need > slab_cap) return 1;
  TP2Args p;
  p.X = (const char*)X; p.ldx = ldx; p.dZx = (const char*)dZx; p.ldz = ldz; p.dL2 = (const char*)dL2; p.ldl = ldl;
  p.dT = dT; p.ldt = (long)G * 384; p.slabW = slabs; p.slabT = slabs + (size_t)nb * G * 128 * 384; p.dump = (char*)(slabs + need0);
  p.N = N; p.tpf = tpf; p.ntiles = S * tpf; p.KL = KL;
  static LdsAttrOnce attr;
  AVMOE_TRY(attr.ensure((const void*)kk_tok_pair2, TP2_LDS, "tok_pair2"));
  const double ntok = (double)S * N;
  {
    const double bytes = ntok * G * (384.0 + 128.0) * 2 + ntok * ldl * 2 + (double)S * KL * G * 384 * 4 + (double)need * 4;
    ProfScope ps("k_tok_pair2", (long)ntok, bytes, 2.0 * ntok * G * 384.0 * (128 + KL), st);
    hipLaunchKernelGGL(kk_tok_pair2, dim3((unsigned)nb, (unsigned)G), dim3(NTHR), TP2_LDS, st, p);
    AVMOE_CHECK_LAUNCH("tok_pair2");
  }
  {
    const long nvec = (long)G * 128 * 384 / 4;
    constexpr int FT = 256;                               // (two waves per block, 768 blocks at cfg-2 instead of a round and a half of four-wave ones: 25 -> 35 us)
    const int nA = (int)((nvec + 16 * (FT / 64) - 1) / (16 * (FT / 64)));
    ProfScope ps("k_tp2_finish", (long)G * 128 * 384, (double)need * 4.0, 0.0, st);
    hipLaunchKernelGGL(kk_tp2_finish, dim3((unsigned)(nA + nb * G)), dim3(FT), 0, st, p.slabW, p.slabT, nb, G, nA, dWt, dT, p.ldt, KL, tpf, p.ntiles);
    AVMOE_CHECK_LAUNCH("tp2_finish");
  }
  return OK;
}

}  // namespace avmoe
