// The backward's two token contractions against X -- dWt = dZx^T X (over all tokens) and dT[s] = dL2[s]^T X[s] (per frame) -- as ONE
// streaming pass with every accumulator in registers (round 4; the tiled form is gemm.hip::gemm_tokpair_kernel, which stays for every
// other shape).  Built on dpost_pair.hip's recipe: a persistent block of eight waves per CU (two per SIMD, 256 registers) streams 64-token
// tiles of X (384 channels of the group), dZx (128 columns) and dL2 (64 latent rows) through the LDS by direct global -> LDS loads (two
// buffers) and contracts over the tokens with BOTH operands read transposed (inline-assembly ds_read_b64_tr_b16: the compiler would drain
// the loads in flight in front of the intrinsic).  Wave w owns channel tiles 3 w .. 3 w + 2 against all twelve row tiles (8 of dWt, 4 of
// dT): 36 accumulator tiles = 144 registers, 15 fragment reads for 36 matrix instructions per 32 tokens.
//
// A block's token range is a whole number of HALF frames (so that 2 S units divide evenly over one block per CU: 640 units = 128 blocks
// x 5 at cfg-2); dT of a frame is flushed when the frame ends -- straight to dT[s] for the part that begins with the frame's first token,
// to a scratch slab for a block's leading half frame, which kk_tp2_finish adds (a frame spans at most two blocks: fixed order).  The blocks'
// partial dWt go to the slab workspace and are summed there in block order as well (no float atomics).
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
  float* slabW; float* slabT;       // [block][g][128][384] partial dWt ; [block][g][64][384] leading half frames
  int tpf, tpb, KL;                 // tiles per frame, tiles per block
};

constexpr int BM = 64, NTHR = 512;
constexpr int RBX = 384 * 2 + 16, RBZ = 128 * 2 + 16, RBL = 72 * 2 + 16;      // LDS row pitches: 49 / 17 / 10 chunks of 16 bytes
constexpr int OFFZ = BM * RBX, OFFL = OFFZ + BM * RBZ, BUF = OFFL + BM * RBL;   // 50176 + 17408 + 10240 = 77824 = 76 pieces of 1 KB
constexpr int TP2_LDS = 2 * BUF;

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
__device__ __forceinline__ bf16x8 tr_pack(const u32x2 (&f)[2]) { return __builtin_bit_cast(bf16x8, u32x4{f[0][0], f[0][1], f[1][0], f[1][1]}); }

// the A fragments (row tiles of the result) of one 32-token step, three at a time: 0 - 7 = dZx column tiles, 8 - 11 = dL2 column tiles
template <int TK, int G3>
__device__ __forceinline__ void issue_a3(u32x2 (&f)[3][2], unsigned lz, unsigned ll) {
  if constexpr (G3 < 2) {
    tr_frag2<TK * 32 * RBZ + (3 * G3 + 0) * 32, RBZ>(f[0], lz); tr_frag2<TK * 32 * RBZ + (3 * G3 + 1) * 32, RBZ>(f[1], lz); tr_frag2<TK * 32 * RBZ + (3 * G3 + 2) * 32, RBZ>(f[2], lz);
  } else if constexpr (G3 == 2) {
    tr_frag2<TK * 32 * RBZ + 6 * 32, RBZ>(f[0], lz); tr_frag2<TK * 32 * RBZ + 7 * 32, RBZ>(f[1], lz); tr_frag2<TK * 32 * RBL + 0, RBL>(f[2], ll);
  } else {
    tr_frag2<TK * 32 * RBL + 1 * 32, RBL>(f[0], ll); tr_frag2<TK * 32 * RBL + 2 * 32, RBL>(f[1], ll); tr_frag2<TK * 32 * RBL + 3 * 32, RBL>(f[2], ll);
  }
}
template <int TK>
__device__ __forceinline__ void issue_b3(u32x2 (&f)[3][2], unsigned lx) {
  tr_frag2<TK * 32 * RBX + 0, RBX>(f[0], lx); tr_frag2<TK * 32 * RBX + 32, RBX>(f[1], lx); tr_frag2<TK * 32 * RBX + 64, RBX>(f[2], lx);
}

__global__ void __launch_bounds__(NTHR, 1) kk_tok_pair2(const TP2Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
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

  // pieces wave + 8 i of the buffer image [X tile: 49 pieces | dZx tile: 17 | dL2 tile: 10], see dpost_pair.hip
  auto gload = [&](int buf, int tile) {
    const long m0 = (long)tile * BM;
    char* dst = smem + buf * BUF + 1024 * wave;
    auto src_x = [&](int j) { const int slot = 64 * j + lane, row = slot / 49, cc = min(slot % 49, 47); return Xb + ((m0 + row) * ldx + cc * 8) * 2; };
    auto src_z = [&](int j) { const int slot = 64 * j + lane, row = slot / 17, cc = min(slot % 17, 15); return Zb + ((m0 + row) * ldz + cc * 8) * 2; };
    auto src_l = [&](int j) { const int slot = 64 * j + lane, row = slot / 10, cc = min(slot % 10, 8); return Lb + ((m0 + row) * ldl + cc * 8) * 2; };
#pragma unroll
    for (int i = 0; i < 6; ++i) __builtin_amdgcn_global_load_lds((gptr_t)src_x(wave + 8 * i), (lptr_t)(dst + 8192 * i), 16, 0, TP2_AUX);
    if (wave == 0) __builtin_amdgcn_global_load_lds((gptr_t)src_x(48), (lptr_t)(dst + 8192 * 6), 16, 0, TP2_AUX);
    else __builtin_amdgcn_global_load_lds((gptr_t)src_z(wave - 1), (lptr_t)(dst + 8192 * 6), 16, 0, TP2_AUX);
    __builtin_amdgcn_global_load_lds((gptr_t)src_z(wave + 7), (lptr_t)(dst + 8192 * 7), 16, 0, TP2_AUX);
    if (wave < 2) __builtin_amdgcn_global_load_lds((gptr_t)src_z(wave + 15), (lptr_t)(dst + 8192 * 8), 16, 0, TP2_AUX);
    else __builtin_amdgcn_global_load_lds((gptr_t)src_l(wave - 2), (lptr_t)(dst + 8192 * 8), 16, 0, TP2_AUX);
    if (wave < 4) __builtin_amdgcn_global_load_lds((gptr_t)src_l(wave + 6), (lptr_t)(dst + 8192 * 9), 16, 0, TP2_AUX);
  };

  const int t0 = blockIdx.x * p.tpb, t1 = t0 + p.tpb;
  gload(0, t0);
  __syncthreads();
  int part0 = t0;                    // first tile of the frame part being accumulated in acc[8 .. 11]
  for (int it = 0, tile = t0; tile < t1; ++it, ++tile) {
    const char* sX = smem + (it & 1) * BUF;
    if (tile + 1 < t1) gload((it + 1) & 1, tile + 1);      // (the other buffer: its readers passed the barrier that ended the previous iteration)
    {
      const unsigned l0 = (unsigned)(size_t)(lptr_t)sX;
      const unsigned lx = l0 + (8 * q + (r >> 2)) * RBX + (3 * wave * 16 + 4 * (r & 3)) * 2;
      const unsigned lz = l0 + OFFZ + (8 * q + (r >> 2)) * RBZ + (4 * (r & 3)) * 2;
      const unsigned ll = l0 + OFFL + (8 * q + (r >> 2)) * RBL + (4 * (r & 3)) * 2;
      u32x2 fb[3][2], fa0[3][2], fa1[3][2];
      auto mm = [&](int i0, const u32x2 (&fa)[3][2]) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[i0 + i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pack(fa[i]), tr_pack(fb[c]), acc[i0 + i][c], 0, 0, 0);
      };
      // (the reads of the next three row tiles are in flight during the nine matrix instructions of the current three)
      issue_b3<0>(fb, lx); issue_a3<0, 0>(fa0, lz, ll); tr_wait6(fb, fa0);
      issue_a3<0, 1>(fa1, lz, ll); mm(0, fa0); tr_wait3(fa1);
      issue_a3<0, 2>(fa0, lz, ll); mm(3, fa1); tr_wait3(fa0);
      issue_a3<0, 3>(fa1, lz, ll); mm(6, fa0); tr_wait3(fa1);
      mm(9, fa1);
      issue_b3<1>(fb, lx); issue_a3<1, 0>(fa0, lz, ll); tr_wait6(fb, fa0);
      issue_a3<1, 1>(fa1, lz, ll); mm(0, fa0); tr_wait3(fa1);
      issue_a3<1, 2>(fa0, lz, ll); mm(3, fa1); tr_wait3(fa0);
      issue_a3<1, 3>(fa1, lz, ll); mm(6, fa0); tr_wait3(fa1);
      mm(9, fa1);
    }
    if (tile + 1 == t1 || (tile + 1) % p.tpf == 0) {       // the frame (or the block's range) ends: flush dT of this part (block-uniform)
      const int s = tile / p.tpf;
      const bool first = part0 % p.tpf == 0;               // begins with the frame's first token: the frame's own rows ; else: a leading half frame
      float* dst = first ? p.dT + (long)s * p.KL * p.ldt + g * 384 : p.slabT + ((long)blockIdx.x * gridDim.y + g) * 64 * 384;
      const long ld = first ? p.ldt : 384;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int l = 16 * i + 4 * q + e;
            if (l < p.KL) dst[l * ld + 16 * (3 * wave + c) + r] = acc[8 + i][c][e];
            acc[8 + i][c][e] = 0.f;
          }
        }
      part0 = tile + 1;
    }
    __syncthreads();                                      // (waits for the direct loads above: the next tile is in place)
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
                                                     float* __restrict__ dWt, float* __restrict__ dT, long ldt, int KL, int tpf, int tpb) {
  if ((int)blockIdx.x < nA) {
    const long per = (long)G * 128 * 384, nvec = per / 4;
    const int lane = threadIdx.x & 63, part = lane >> 4;
    const long v = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (lane & 15);
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
  const long t0 = (long)b * tpb;
  if (t0 % tpf == 0) return;                               // this block's range begins with a frame
  const long s = t0 / tpf;
  const float* src = slabT + ((long)b * G + g) * 64 * 384;
  float* dst = dT + s * KL * ldt + g * 384;
  for (int i = threadIdx.x; i < KL * 96; i += 256) {
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
  if (Cg != 384 || M1 != 128 || KL < 1 || KL > 64 || ldl < 72 || N % BM || S < 1 || ldx % 8 || ldz % 8 || ldl % 8 || !slabs ||
      ((uintptr_t)X % 16) || ((uintptr_t)dZx % 16) || ((uintptr_t)dL2 % 16) || ((uintptr_t)dWt % 16) || ((uintptr_t)dT % 16) || (G * 384) % 4)
    return 1;
  const bool force = getenv("AVMOE_TOKPAIR2_FORCE") != nullptr;             // test hook (read by the product build too, per call: tests / bench.py's parity leg switch it inside one process): small sites as well -- every frame then spans two blocks
  if (!force && (long)S * N < 65536) return 1;                              // small sites: the tiled form fills the chip better
  const int cus = cu_count();                             // (cached per device: common.cpp)
  if (cus <= 0) { set_last_error("tok_pair2: device query"); return ERR_LAUNCH; }
  const int tpf = N / BM;
  const int ut = (tpf % 2 == 0) ? tpf / 2 : tpf;            // tiles per unit: half a frame (a whole one when its tile count is odd)
  const long U = (long)S * (tpf / ut);
  int nb = std::max(1, cus / G);
  while (nb > 1 && U % nb) --nb;                          // one block per CU where the units divide evenly; else the next smaller count that does
  if (!force && nb * G * 4 < cus * 3) return 1;             // ... unless that leaves a quarter of the chip idle
  const size_t need = (size_t)nb * G * (128 + 64) * 384;
  if (need > slab_cap) return 1;
  TP2Args p;
  p.X = (const char*)X; p.ldx = ldx; p.dZx = (const char*)dZx; p.ldz = ldz; p.dL2 = (const char*)dL2; p.ldl = ldl;
  p.dT = dT; p.ldt = (long)G * 384; p.slabW = slabs; p.slabT = slabs + (size_t)nb * G * 128 * 384;
  p.tpf = tpf; p.tpb = (int)(U / nb) * ut; p.KL = KL;
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
    const int nA = (int)((nvec + 63) / 64);
    ProfScope ps("k_tp2_finish", (long)G * 128 * 384, (double)need * 4.0, 0.0, st);
    hipLaunchKernelGGL(kk_tp2_finish, dim3((unsigned)(nA + nb * G)), dim3(256), 0, st, p.slabW, p.slabT, nb, G, nA, dWt, dT, p.ldt, KL, tpf, p.tpb);
    AVMOE_CHECK_LAUNCH("tp2_finish");
  }
  return OK;
}

}  // namespace avmoe
