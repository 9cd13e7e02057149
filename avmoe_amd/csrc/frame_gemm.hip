// Per-frame products of a few rows against ONE shared matrix over a SHORT inner dimension (round 5):
//
//     C[s] = alpha A[s] B^T      A[s]: M <= 64 rows x K <= 256 (K-major, one block of rows per frame s),  B: N x K (K-major, shared),  C[s]: M x N
//
// -- the hop-1 chain's middle: L1[s] = [R | qr | qb][s] [Wc | bc | 1]^T (K = M_y + 2 audio tokens, N = N_x = 1024 image
// tokens) and its backward twins, 64 latent rows per frame against the remap matrix.  On the tiled engine they are 5120 independent 64 x 64
// tiles of four K steps that fetch 16 KB for 0.5 MFLOP: 335 MB through the L2 -> LDS path for an 85 MB result, 45 us whatever the output type,
// the prefetch depth or the dispatch (scripts/gemm_hop1_micro.py, profiles/r05_gemm_hop1_pmc.txt; a first persistent 64 x 64 version of this
// file with counted waits and no index arithmetic in the loop took the same 45 us -- the loads are the bound, not the instructions around them).
//
// Here B is STATIONARY IN REGISTERS: a block of four waves owns 128 columns of B -- wave w the fragments of columns 32 w .. + 31 over the whole
// K, 8 registers per 32 K entries, loaded and tail-masked once -- and streams the frames of its range through two LDS stages by direct
// global -> LDS loads (a whole frame, 64 rows x K, per stage; rows 16 (4 NKS + 1) bytes apart so that the sixteen rows a fragment read
// touches fall on sixteen different bank groups), one barrier and one counted vmcnt wait per frame.  A wave multiplies all 64 rows by its
// 32 columns, transposed, so that a lane stores four consecutive columns of a row straight from the accumulators: per frame 4 NKS LDS
// reads, 8 NKS matrix instructions and eight stores per wave, and A is fetched N / 128 times instead of N / 64 times with nothing for B.
#include "gemm.h"
#include "common.h"
#include "prof.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>

namespace avmoe {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct FGArgs {
  const char* A; long lda, sA1;        // bf16 [frame][rows >= M][lda]
  const char* B; long ldb;             // bf16 [N][ldb]
  char* C; long ldc, sC1; int c_bf16;  // [frame][M][ldc] bf16 or fp32
  int M, N, K, S, nct, fpb;            // column tiles of 128, frames per block
  float alpha;
};

#ifndef FG_DISSECT
#define FG_DISSECT 0      // development (scripts/variant_lib.sh): 1 no stores, 2 no frame loads, 4 no matrix instructions -- wrong results, timing only
#endif
__device__ __attribute__((aligned(16))) char fg_dump[64 * 16];      // where the lanes outside C store (nobody reads it): every wave issues exactly eight stores per frame

__device__ __forceinline__ unsigned int fg_f2bf(float x) { return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)x); }

template <int NKS> struct FGShape {
  static constexpr int RC = 4 * NKS + 1;             // 16-byte pieces per LDS row (one of padding)
  static constexpr int RS = RC * 16;                 // row stride, bytes
  static constexpr int NP = RC;                      // 1 KB pieces per stage (64 rows x RC pieces / 64 lanes)
  static constexpr int NL = (NP + 3) / 4;            // loads per wave per frame (the last ones may repeat piece NP - 1: same data to the same place)
  static constexpr int STAGE = NP * 1024;
  static constexpr int LDS = 2 * STAGE;
};

// TR: C[s] is stored transposed, [n][m] with m contiguous (ldc between columns n) -- the product runs untransposed and a lane holds four consecutive rows
template <int NKS, bool TR>
__global__ void __launch_bounds__(256, 2) kk_frame_gemm(const FGArgs p) {
  using SH = FGShape<NKS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  // the blocks of one frame range (they read the same A) are neighbours on ONE XCD: block b runs on XCD b % 8
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int ct = idx % p.nct, fb = xcd + 8 * (idx / p.nct);
  const int s0 = fb * p.fpb, s1 = min(p.S, s0 + p.fpb);
  if (s0 >= s1) return;
  const int n0 = ct * 128 + 32 * wave;
  char* const dump = fg_dump + lane * 16;
  const int cmax = (p.K + 7) / 8 - 1;                // last 16-byte piece of a row that starts inside its K entries
  // lane offsets of the frame loads, once: LDS slot (piece, lane) -> (row, piece of the row); pieces beyond the row's data re-read its last one
  unsigned voff[SH::NL];
#pragma unroll
  for (int i = 0; i < SH::NL; ++i) {
    const int piece = min(wave + 4 * i, SH::NP - 1), slot = piece * 64 + lane, row = slot / SH::RC, c = slot - row * SH::RC;
    voff[i] = (unsigned)((long)min(row, p.M - 1) * p.lda * 2 + min(c, cmax) * 16);
  }
  auto gload = [&](int stage, int s) {
    const char* ab = p.A + (long)s * p.sA1 * 2;
    char* d = smem + stage * SH::STAGE;
#pragma unroll
    for (int i = 0; i < SH::NL; ++i) {
      unsigned o = voff[i];
      asm volatile("" : "+v"(o));
      const int piece = min(wave + 4 * i, SH::NP - 1);       // (wave-uniform)
      __builtin_amdgcn_global_load_lds((gptr_t)(ab + o), (lptr_t)(d + 1024 * piece), 16, 0, 0);
    }
  };
  gload(0, s0);
  // B fragments of this wave's 32 columns over the whole K, tail masked (A's K padding may hold anything finite or not: it is masked too, below)
  u32x4 mk;                                                    // this lane's eight K entries of the LAST K step that are data
  {
    const int nv = p.K - (32 * (NKS - 1) + 8 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) mk[e] = (2 * e + 1 < nv) ? 0xffffffffu : ((2 * e < nv) ? 0x0000ffffu : 0u);
  }
  bf16x8 bfr[NKS][2];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int n = min(n0 + 16 * c + r, p.N - 1);
      u32x4 v = *(const u32x4*)(p.B + (long)n * p.ldb * 2 + min(4 * ks + q, cmax) * 16);
      if (ks == NKS - 1) v &= mk;
      bfr[ks][c] = __builtin_bit_cast(bf16x8, v);
    }
  const int fbase = r * SH::RS + q * 16;
  for (int s = s0; s < s1; ++s) {
    const int stage = (s - s0) & 1;
    // in-order counter: [NL loads of frame s] [8 stores of frame s - 1]   (issued in the previous iteration, in this order)
    if (s == s0 || FG_DISSECT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + 1 < s1 && !(FG_DISSECT & 2)) gload(stage ^ 1, s + 1);
    const char* sS = smem + stage * SH::STAGE + fbase;
    f32x4 acc[2][4];                   // [column tile][row tile]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      u32x4 av[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) av[t] = *(const u32x4*)(sS + 16 * t * SH::RS + 64 * ks);
      if (ks == NKS - 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) av[t] &= mk;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          if (FG_DISSECT & 4) acc[c][t][0] += __builtin_bit_cast(f32x4, av[t])[c];
          else acc[c][t] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[t]), bfr[ks][c], acc[c][t], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][c], __builtin_bit_cast(bf16x8, av[t]), acc[c][t], 0, 0, 0);
    }
    // lane (r, q) holds C[16 t + r][n0 + 16 c + 4 q .. + 3] (TR: C[16 t + 4 q .. + 3][n0 + 16 c + r]): exactly eight stores per wave (the loop's wait counts them)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int m = TR ? 16 * t + 4 * q : 16 * t + r, n = TR ? n0 + 16 * c + r : n0 + 16 * c + 4 * q;
        const f32x4 v = acc[c][t] * p.alpha;
        if ((FG_DISSECT & 1) && v[0] != 1234.5f) continue;
        const int left = TR ? p.M - m : p.N - n;             // elements of the lane's four that exist
        const bool in = TR ? n < p.N : m < p.M;
        const bool ok = in && left >= 4;                     // (a ragged last group is stored element by element below: more stores only make the wait stricter)
        const long e = (long)s * p.sC1 + (TR ? (long)n * p.ldc + m : (long)m * p.ldc + n);
        if (p.c_bf16) {
          char* dst = ok ? p.C + e * 2 : dump;
          *(u32x2*)dst = u32x2{fg_f2bf(v[0]) | (fg_f2bf(v[1]) << 16), fg_f2bf(v[2]) | (fg_f2bf(v[3]) << 16)};
          if (!ok && in && left > 0) {
#pragma unroll
            for (int x = 0; x < 3; ++x) if (x < left) ((unsigned short*)p.C)[e + x] = (unsigned short)fg_f2bf(v[x]);
          }
        } else {
          char* dst = ok ? p.C + e * 4 : dump;
          *(f32x4*)dst = v;
          if (!ok && in && left > 0) {
#pragma unroll
            for (int x = 0; x < 3; ++x) if (x < left) ((float*)p.C)[e + x] = v[x];
          }
        }
      }
  }
}

template <int NKS, bool TR>
int fg_launch2(const FGArgs& p, int nblocks, hipStream_t st) {
  static LdsAttrOnce attr;
  AVMOE_TRY(attr.ensure((const void*)kk_frame_gemm<NKS, TR>, FGShape<NKS>::LDS, "frame_gemm"));
  hipLaunchKernelGGL((kk_frame_gemm<NKS, TR>), dim3((unsigned)nblocks), dim3(256), FGShape<NKS>::LDS, st, p);
  return OK;
}
template <int NKS>
int fg_launch(const FGArgs& p, bool tr, int nblocks, hipStream_t st) { return tr ? fg_launch2<NKS, true>(p, nblocks, st) : fg_launch2<NKS, false>(p, nblocks, st); }

// one launch over S frames of p.M <= 64 rows
int fg_run(FGArgs p, bool tr, int cus, hipStream_t st) {
  p.nct = (p.N + 127) / 128;
  // two blocks per CU: frames per block so that the grid is about 2 x CUs
  const int want = std::max(1, 2 * cus / p.nct);            // frame ranges
  p.fpb = std::max(1, (p.S + want - 1) / want);
  const int nfb = (int)round_up((p.S + p.fpb - 1) / p.fpb, 8);      // (whole rounds of the eight XCDs; blocks beyond the last frame return at once)
  const int nblocks = p.nct * nfb;
  int rc;
  switch ((p.K + 31) / 32) {
    case 1: rc = fg_launch<1>(p, tr, nblocks, st); break;
    case 2: rc = fg_launch<2>(p, tr, nblocks, st); break;
    case 3: rc = fg_launch<3>(p, tr, nblocks, st); break;
    case 4: rc = fg_launch<4>(p, tr, nblocks, st); break;
    case 5: rc = fg_launch<5>(p, tr, nblocks, st); break;
    case 6: rc = fg_launch<6>(p, tr, nblocks, st); break;
    case 7: rc = fg_launch<7>(p, tr, nblocks, st); break;
    default: rc = fg_launch<8>(p, tr, nblocks, st); break;
  }
  if (rc != OK) return rc;
  AVMOE_CHECK_LAUNCH("frame_gemm");
  return OK;
}

// ---- LONG inner dimension, few columns (N <= 224, K > 256): [Bm | ab][s] = A1[s] [Wc | bc] (N = M_y + 1, K = N_x) and its twins ----------------
// Nothing is stationary here (B is N x K = 400 KB): a block of eight waves owns TWO frames (128 rows) and ALL columns, and runs the K chunks
// of 64 as one software pipeline -- [A 128 x 64 | B 224 x 64] by direct global -> LDS loads into three stages (44 KB each), counted vmcnt
// waits, lane offsets computed once.  B is fetched once per frame PAIR: 112 MB through the L2 -> LDS path per product instead of the
// 335 MB of 64 x 64 tiles.  Wave (rg, cg) multiplies rows 32 rg .. + 31 by the column tiles 7 cg .. 7 cg + 6.  (The same pipeline with the chunks
// travelling global -> registers -> LDS, two register sets in flight and two LDS stages, measured the same 21 - 29 us per launch under the
// profiler's events, whose floor for an empty launch is 8 us: the direct loads are not what bounds it.  160 blocks of 8 waves: one per CU.)
struct FLArgs {
  const char* A; long lda, sA1;        // bf16 [frame][rows][lda]
  const char* B; long ldb;             // bf16 [N][ldb]
  char* C; long ldc, sC1; int c_bf16;  // [frame][M][ldc] (TR: [frame][N][ldc]) bf16 or fp32
  int M, N, K, S;                      // rows per frame that exist (<= 64), frames
  long rows;                           // tall form: frames are groups of 64 consecutive rows of ONE matrix of `rows` rows (else a huge number)
  float alpha;
};
constexpr int FL_KC = 64, FL_A = 128 * 128, FL_B = 224 * 128, FL_STAGE = FL_A + FL_B, FL_NBUF = 3, FL_LDS = FL_NBUF * FL_STAGE, FL_NL = 6;

template <bool TR>
__global__ void __launch_bounds__(512, 1) kk_frame_gemm_long(const FLArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  const int rg = wave & 3, cg = wave >> 2;
  const int f0 = 2 * blockIdx.x;                             // frames f0, f0 + 1
  const int cmax8 = (p.K + 7) / 8 - 1;                       // last 16-byte piece of a row that starts inside its K entries
  // rows of frame f that exist
  auto mrows = [&](int f) { return f < p.S ? (int)max(0L, min((long)p.M, p.rows - 64L * f)) : 0; };
  const int m0 = mrows(f0), m1 = mrows(f0 + 1);
  // stage image: rows of 128 bytes = 8 pieces of 16, piece c of row t at position c ^ (t & 7); A: 16 KB = 16 pieces of 1 KB (2 per wave), B: 28 (wave w:
  // w, w + 8, w + 16, min(w + 24, 27)).  Rows that do not exist re-read an existing one (never stored).
  long offl[FL_NL]; int cc[FL_NL], dst[FL_NL];
#pragma unroll
  for (int i = 0; i < FL_NL; ++i) {
    if (i < 2) {
      const int piece = 2 * wave + i, row = piece * 8 + (lane >> 3), f = row >> 6, m = row & 63;
      const int mm = f == 0 ? min(m, max(m0, 1) - 1) : min(m, max(m1, 1) - 1);
      const int ff = (f == 1 && m1 > 0) ? f0 + 1 : f0;
      offl[i] = ((long)ff * p.sA1 + (long)(ff == f0 + f ? mm : 0) * p.lda) * 2;
      cc[i] = (lane & 7) ^ (row & 7); dst[i] = 1024 * piece;
    } else {
      const int piece = min(wave + 8 * (i - 2), 27), row = piece * 8 + (lane >> 3);
      offl[i] = (long)min(row, p.N - 1) * p.ldb * 2;
      cc[i] = (lane & 7) ^ (row & 7); dst[i] = FL_A + 1024 * piece;
    }
  }
  const int nkc = (p.K + FL_KC - 1) / FL_KC;
  auto gload = [&](int buf, int kc) {
    char* d = smem + buf * FL_STAGE;
    const int k0 = kc * FL_KC;
#pragma unroll
    for (int i = 0; i < FL_NL; ++i) {
      const char* base = (i < 2 ? p.A : p.B) + offl[i];
      const int c = min(k0 / 8 + cc[i], cmax8);              // (pieces beyond the row's data re-read its last one: masked in the fragments)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + c * 16), (lptr_t)(d + dst[i]), 16, 0, 0);
    }
  };
#pragma unroll
  for (int j = 0; j < FL_NBUF - 1; ++j)
    if (j < nkc) gload(j, j);
  f32x4 acc[7][2];                   // [column tile][row tile]
#pragma unroll
  for (int c = 0; c < 7; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment addresses inside a stage: row t, K step ks, lane quarter q -> piece (4 ks + q) ^ (t & 7); rows 16 apart share t & 7
  const int ra = 32 * rg + r, rb = 112 * cg + r;
  const int fa0 = ra * 128 + ((q ^ (ra & 7)) * 16), fa1 = ra * 128 + (((4 + q) ^ (ra & 7)) * 16);
  const int fb0 = FL_A + rb * 128 + ((q ^ (rb & 7)) * 16), fb1 = FL_A + rb * 128 + (((4 + q) ^ (rb & 7)) * 16);
  for (int kc = 0; kc < nkc; ++kc) {
    // in-order counter: [6 loads of chunk kc] [6 loads of chunk kc + 1]
    if (kc + 1 < nkc) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kc + FL_NBUF - 1 < nkc) gload((kc + FL_NBUF - 1) % FL_NBUF, kc + FL_NBUF - 1);
    const char* sS = smem + (kc % FL_NBUF) * FL_STAGE;
    const int k0 = kc * FL_KC;
    const bool tail = k0 + FL_KC > p.K;                      // (block-uniform) the chunk holds columns beyond K: masked in the fragments
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 av[2], bv[7];
#pragma unroll
      for (int t = 0; t < 2; ++t) av[t] = *(const u32x4*)(sS + (ks ? fa1 : fa0) + t * 16 * 128);
#pragma unroll
      for (int c = 0; c < 7; ++c) bv[c] = *(const u32x4*)(sS + (ks ? fb1 : fb0) + c * 16 * 128);
      if (tail) {
        const int nv = p.K - (k0 + 32 * ks + 8 * q);         // this lane's eight K entries that are data
        u32x4 mk;
#pragma unroll
        for (int e = 0; e < 4; ++e) mk[e] = (2 * e + 1 < nv) ? 0xffffffffu : ((2 * e < nv) ? 0x0000ffffu : 0u);
#pragma unroll
        for (int t = 0; t < 2; ++t) av[t] &= mk;
#pragma unroll
        for (int c = 0; c < 7; ++c) bv[c] &= mk;
      }
#pragma unroll
      for (int c = 0; c < 7; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[c][t] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[t]), __builtin_bit_cast(bf16x8, bv[c]), acc[c][t], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bv[c]), __builtin_bit_cast(bf16x8, av[t]), acc[c][t], 0, 0, 0);
    }
  }
  // lane (r, q) holds C[32 rg + 16 t + r][112 cg + 16 c + 4 q .. + 3]   (TR: C[32 rg + 16 t + 4 q .. + 3][112 cg + 16 c + r])
  const int f = f0 + (rg >> 1), mf = (rg >> 1) ? m1 : m0;
#pragma unroll
  for (int c = 0; c < 7; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int m = 32 * (rg & 1) + 16 * t + (TR ? 4 * q : r), n = 112 * cg + 16 * c + (TR ? r : 4 * q);
      const f32x4 v = acc[c][t] * p.alpha;
      const int left = TR ? mf - m : p.N - n;
      const bool in = TR ? n < p.N : m < mf;
      if (!in || left <= 0) continue;
      const long e = (long)f * p.sC1 + (TR ? (long)n * p.ldc + m : (long)m * p.ldc + n);
      if (left >= 4) {
        if (p.c_bf16) *(u32x2*)(p.C + e * 2) = u32x2{fg_f2bf(v[0]) | (fg_f2bf(v[1]) << 16), fg_f2bf(v[2]) | (fg_f2bf(v[3]) << 16)};
        else *(f32x4*)(p.C + e * 4) = v;
      } else {
#pragma unroll
        for (int x = 0; x < 3; ++x)
          if (x < left) { if (p.c_bf16) ((unsigned short*)p.C)[e + x] = (unsigned short)fg_f2bf(v[x]); else ((float*)p.C)[e + x] = v[x]; }
      }
    }
}

template <bool TR>
int fl_launch(const FLArgs& p, hipStream_t st) {
  static LdsAttrOnce attr;
  AVMOE_TRY(attr.ensure((const void*)kk_frame_gemm_long<TR>, FL_LDS, "frame_gemm_long"));
  hipLaunchKernelGGL((kk_frame_gemm_long<TR>), dim3((unsigned)((p.S + 1) / 2)), dim3(512), FL_LDS, st, p);
  AVMOE_CHECK_LAUNCH("frame_gemm_long");
  return OK;
}

}  // namespace

// 0 = launched, 1 = shape not served (the caller runs the tiled engine), < 0 error
int launch_gemm_frames(const GemmArgs& a, hipStream_t st) {
  static const bool off = dev_env("AVMOE_NO_FRAME_GEMM") != nullptr;
  const bool tr = a.sCj != 1;                                 // C[s] stored [n][m]
  const long ldc = tr ? a.sCj : a.sCi;
  if (off || a.dtype != GEMM_BF16 || a.epi != GEMM_EPI_NONE || a.nb2 != 1 || a.nb3 > 1 || a.a_layout != K_MAJOR || a.b_layout != K_MAJOR || a.sB1 != 0 ||
      a.M < 1 || a.nb1 < 32 || a.ksplit > 1 || a.A2 || a.D || a.row_scale || a.accumulate || (tr && a.sCi != 1) || a.Cx || a.st_rows || a.N < 64 || a.K < 1 || (a.K > 256 && a.N > 224) ||
      a.lda % 8 || a.ldb % 8 || a.sA1 % 8 || ((uintptr_t)a.A % 16) || ((uintptr_t)a.B % 16) || ldc % 4 || a.sC1 % 4 || ((uintptr_t)a.C % 16) ||
      64 * a.lda * 2 >= (1L << 31))
    return 1;
  // rows are read in whole 16-byte pieces that START inside the row's K entries: lda / ldb >= round_up(K, 8)
  if (a.lda < ((a.K + 7) & ~7) || a.ldb < ((a.K + 7) & ~7)) return 1;
  // frames whose row blocks follow one another without a gap, in A and in C, are ONE tall matrix: cut into groups of 64 rows whatever M is
  // (the 65 rows per frame of dA1[s] = [dBm | dab][s] [Wc | bc]^T, moe_backward.cpp) -- the rows left over take a second, one-group launch
  const bool tall = !tr && a.M != 64 && a.sA1 == (long)a.M * a.lda && a.sC1 == (long)a.M * a.sCi;
  if (a.M > 64 && !tall) return 1;
  const int cus = cu_count();
  if (cus <= 0) { set_last_error("frame_gemm: device query"); return ERR_LAUNCH; }
  FGArgs p;
  p.A = (const char*)a.A; p.lda = a.lda; p.sA1 = a.sA1; p.B = (const char*)a.B; p.ldb = a.ldb;
  p.C = (char*)a.C; p.ldc = ldc; p.sC1 = a.sC1; p.c_bf16 = a.out_dtype == GEMM_BF16;
  p.M = a.M; p.N = a.N; p.K = a.K; p.S = a.nb1; p.alpha = a.alpha; p.nct = p.fpb = 0;
  const double nb = (double)a.nb1;
  const double bytes = nb * a.M * (double)a.K * 2 + (double)a.N * a.K * 2 + nb * a.M * (double)a.N * (p.c_bf16 ? 2.0 : 4.0);
  static const bool shapes = getenv("AVMOE_PROF_SHAPES") != nullptr;
  const char* pname = "gemm_frames";
  if (shapes && prof_enabled()) { char* nm = (char*)malloc(96); snprintf(nm, 96, "gemm_frames M%d N%d K%d b%d%s", a.M, a.N, a.K, a.nb1, tr ? " T" : ""); pname = nm; }
  ProfScope ps(pname, bytes, 2.0 * nb * a.M * (double)a.N * a.K, st);
  const long rows = (long)a.nb1 * a.M, full = rows / 64, rest = rows % 64;
  if (a.K > 256) {                                           // the long form: frame pairs x all columns
    FLArgs l;
    l.A = p.A; l.lda = a.lda; l.B = p.B; l.ldb = a.ldb; l.C = p.C; l.ldc = ldc; l.c_bf16 = p.c_bf16; l.N = a.N; l.K = a.K; l.alpha = a.alpha;
    if (tall) { l.M = 64; l.S = (int)(full + (rest > 0)); l.sA1 = 64 * a.lda; l.sC1 = 64 * ldc; l.rows = rows; }
    else { l.M = a.M; l.S = a.nb1; l.sA1 = a.sA1; l.sC1 = a.sC1; l.rows = 1L << 40; }
    return tr ? fl_launch<true>(l, st) : fl_launch<false>(l, st);
  }
  if (!tall) return fg_run(p, tr, cus, st);
  const long csz = p.c_bf16 ? 2 : 4;
  p.M = 64; p.S = (int)full; p.sA1 = 64 * a.lda; p.sC1 = 64 * ldc;
  if (full > 0) AVMOE_TRY(fg_run(p, false, cus, st));
  if (rest > 0) {
    p.A += full * 64 * a.lda * 2; p.C += full * 64 * ldc * csz; p.M = (int)rest; p.S = 1;
    AVMOE_TRY(fg_run(p, false, cus, st));
  }
  return OK;
}

}  // namespace avmoe
