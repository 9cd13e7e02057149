// Streaming form of the engine GEMM for the products that run over ALL tokens against a small shared matrix:
//
//   C[b2][m][n] = alpha * sum_k A[b2][m][k] B[b2][n][k]  (+ row_scale[m] * D[b2][m][n])      M ~ 10^5..10^6,  N, K <= 384
//
// (the grouped down projection, the fused output GEMM and dApost = dOut Bpost of the adapter site).  The tiled engine in
// gemm.hip runs these at a fraction of HBM speed because a block's load, MFMA and store phases do not overlap and B is
// re-staged through LDS for every tile.  Here B never moves again after the prologue:
//
//   * persistent blocks; wave w keeps the MFMA fragments of ITS TPW 16-column tiles of B in registers for the whole kernel
//     (B stationary), so the LDS only carries the streamed A rows;
//   * A row tiles (BM rows x K) go global -> registers -> LDS, two buffers, the loads of tile i+1 in flight during the
//     MFMAs and stores of tile i: one barrier per tile;
//   * products are computed transposed (A operand = B fragment, B operand = A fragment): lane (r, q) ends up with
//     C[row r][4 q .. 4 q + 3] of each tile -- 8 / 16 contiguous bytes per lane, stored straight from registers.
#include "gemm.h"
#include "common.h"
#include "prof.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>

namespace avmoe {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct StreamArgs {
  const char* A; const char* B; const char* A2; const char* B2; char* C; const char* D; const float* rs;
  int Mper, nsamp, N, K, K2, tps, ntiles, contig;          // tps: row tiles per sample ; ntiles = nsamp * tps
  long lda, ldb, lda2, ldb2, ldc, ldd;
  long sA1, sA2, sB2, s2A1, s2A2, s2B1, s2B2, sC1, sC2, sD1, sD2, sRS1, sRS2;   // batch strides (elements): 1 = sample, 2 = group
  float alpha; int b_mn, out_bf16;
  float* Cx; int nsplit; long ldcx, sCx2;          // fp32 side output for the columns >= nsplit (GemmArgs::Cx)
  float* st_rows; float* st_cols; long st_ntot;    // statistics of A (GemmArgs::st_rows / st_cols)
  const char* B3; float* C3; int N3; long ldb3, s3B1, s3B2, ldc3, s3C1, s3C2;      // extra columns against a per-sample matrix (GemmArgs::B3)
};

__device__ __forceinline__ float bfbits(unsigned int h) { return __builtin_bit_cast(float, h << 16); }
__device__ __forceinline__ unsigned int f2bfbits(float x) { return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ u32x4 mask_tail8(u32x4 v, int valid) {     // keep the first `valid` (< 8) bf16 of a 16-byte chunk
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (2 * e >= valid) v[e] = 0u;
    else if (2 * e + 1 >= valid) v[e] &= 0xFFFFu;
  }
  return v;
}

// A lane's run of 4 TPW bf16 output-row elements (columns nl ..): 16-byte loads where the run is 16-byte aligned (TPW even)
template <int TPW>
__device__ __forceinline__ void ld_run(const char* ptr, int nl, int N, u32x2 (&v)[TPW]) {
  if constexpr (TPW % 2 == 0) {
#pragma unroll
    for (int t = 0; t < TPW; t += 2) {
      if (nl + 4 * t + 4 < N) { const u32x4 w = *(const u32x4*)(ptr + 8 * t); v[t] = u32x2{w[0], w[1]}; v[t + 1] = u32x2{w[2], w[3]}; }
      else if (nl + 4 * t < N) v[t] = *(const u32x2*)(ptr + 8 * t);
    }
  } else {
#pragma unroll
    for (int t = 0; t < TPW; ++t)
      if (nl + 4 * t < N) v[t] = *(const u32x2*)(ptr + 8 * t);
  }
}

// One 8-element fragment of an MN-major matrix ([k][n], leading dim ld): elements k0 .. k0+7 of column n
__device__ __forceinline__ u32x4 frag_mn(const unsigned short* base, long ld, int n, int k0, int K) {
  u32x4 v = {0u, 0u, 0u, 0u};
  const unsigned short* bp = base + (long)k0 * ld + n;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned int h = (k0 + j < K) ? (unsigned int)bp[(long)j * ld] : 0u;
    v[j >> 1] |= (j & 1) ? (h << 16) : h;
  }
  return v;
}

// KS / KS2: 32-wide K steps of the first / second segment (KS2 = 0: none).  The second segment's B2 is per sample
// (MN-major) and is re-read into registers whenever the block moves on to the next sample.
// A2MN: the second segment's A2 is MN-major ([k2][m], leading dim lda2): it is read along m and transposed into the
// K-major LDS rows with 2-byte stores (its K2 is small), so the MFMA loop is the same.
// ACC: C += (bf16 C): the old values of a 16-row slab are requested one slab ahead of the MFMAs that need them.
// PF2: TWO row tiles of loads in flight per block (two register stages in front of the two LDS buffers): a block's bytes in flight
// -- what bounds an HBM stream once the arithmetic is hidden -- double for NLD more registers per stage.
// slabs of a row tile whose stores are issued after the next tile has been moved into the LDS (see the loop body)
// accumulating variants whose old C values are all requested BEFORE the next tile (like the D operand): where the registers allow
constexpr bool stream_acc_ahead(int KS, int KS2, int TPW, int NW, int MT) { return NW >= 9 && TPW * (KS + KS2) * 4 + MT * TPW * 8 + 60 <= 168; }
constexpr int stream_hold(int KS, int KS2, int TPW, int NW, int MT, bool ACC) {
  if (ACC && !stream_acc_ahead(KS, KS2, TPW, NW, MT)) return 0;
  const int est = TPW * (KS + KS2) * 4 + MT * TPW * 6 + 80;          // B fragments + held accumulators + D operands + the rest
  if (NW >= 9) return est <= 168 ? MT : 1;                          // 9 / 12 waves: three per SIMD
  if (NW == 4 && KS == 12) return MT;                               // the down projection: two 4-wave blocks per CU, 256 registers
  return 1;                                                         // several small blocks per CU: residency first
}
#ifndef STREAM_DISSECT
#define STREAM_DISSECT 0         // development builds: 1 = no stores, 2 = no MFMAs, 3 = no row-tile loads, 4 = no per-sample B2 reload (scripts/stream_dissect.sh)
#endif
// waves per SIMD promised to the compiler (second launch bound): what the intended residency needs, at most 3 (168 registers)
constexpr int stream_minw(int NW, int per_cu) { return (NW * per_cu + 3) / 4 < 3 ? (NW * per_cu + 3) / 4 : 3; }
#ifndef SC_DAP_MINW
#define SC_DAP_MINW 1        // 9-wave blocks (dApost): 5 here = two resident blocks per CU (<= 96 VGPRs, 4 spills): measured no gain
#endif
// STATS: the row tile's statistics as a side product (GemmArgs::st_rows / st_cols), all of them on the matrix pipe from fragments of
// the tile that sits in the LDS anyway:  sum_k A[m][k] = (ones . A^T)[.][m] ,  sum_k A[m][k]^2 = diag(A A^T)  (wave (2 mt + j) % NW
// does statistic j of slab mt with the fragments it reads for its products) ,  column sums = (A^T-fragments . ones) with the
// fragments read transposed (ds_read_tr16_b64: tokens along the contraction), 16-channel tile ct by wave ct % NW.
// X3: 16-column tiles per wave of the EXTRA product against the per-sample matrix B3 (GemmArgs::B3; statistics variants only): their
// fragments are re-read whenever the block moves on to the next sample (contiguous tile ranges: once or twice per block).
template <int KS, int KS2, int TPW, int NW, int BM, bool A2MN, bool ACC, bool PF2, int MINW, bool STATS, int X3 = 0>
__global__ void __launch_bounds__(NW * 64, MINW) gemm_stream_kernel(const StreamArgs p) {
  static_assert(!STATS || (KS2 == 0 && !A2MN && !ACC && BM % 32 == 0), "statistics: plain K-major single-segment products only");
  static_assert(X3 == 0 || STATS, "extra per-sample columns: statistics variants only");
  constexpr int NT = NW * 64, MT = BM / 16, KSA = KS + KS2;
  constexpr int HOLD = stream_hold(KS, KS2, TPW, NW, BM / 16, ACC);
  constexpr int CPR1 = A2MN ? KS * 4 : KSA * 4;           // 16-byte chunks per row that come from K-major sources
  constexpr int TOT2 = A2MN ? KS2 * 32 * (BM / 8) : 0, NLD2 = A2MN ? (TOT2 + NT - 1) / NT : 1;
  constexpr int RB = KSA * 64 + 16;            // LDS bytes per A row (odd multiple of 16: conflict-free 16-byte fragment reads)
  constexpr int STG = BM * RB;
  constexpr int CPR = CPR1, TOT = BM * CPR, NLD = (TOT + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int g = blockIdx.y;
  const int osz = p.out_bf16 ? 2 : 4;
  const char* Bb = p.B + (long)g * p.sB2 * 2;

  // ---- B fragments of this wave's column tiles: registers for the whole kernel ----
  // row i = r of tile t  <->  column nrow(t) ; lane (r, q) then owns the output run  nl .. nl + 4 TPW  (see the stores)
  const int nw0 = wave * TPW * 16;
  bf16x8 bfr[TPW][KS];
  // (K-major B: every fragment load UNCONDITIONAL -- clamped row / K offset, what lies outside zeroed afterwards -- so that all TPW x KS of them
  //  are in flight at once; under per-fragment conditions each load was its own branch and `s_waitcnt vmcnt(0)`: a memory round trip per
  //  fragment, twelve in a row in front of the down projection's first tile and again at every frame change)
  if (!STATS && !p.b_mn) {        // (not the statistics variants: they sit at the register cap and measured slower with the twelve raw chunks live, 227 -> 239 us)
    u32x4 raw[TPW][KS];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int n = nw0 + (r >> 2) * (4 * TPW) + 4 * t + (r & 3);
      const char* rowp = Bb + (long)min(n, p.N - 1) * p.ldb * 2;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { const int k0 = 32 * ks + 8 * q; raw[t][ks] = *(const u32x4*)(rowp + (k0 < p.K ? k0 : 0) * 2); }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int n = nw0 + (r >> 2) * (4 * TPW) + 4 * t + (r & 3);
      const bool live = n < p.N && !(X3 > 0 && nw0 >= p.N);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k0 = 32 * ks + 8 * q;
        u32x4 v = raw[t][ks];
        if (k0 + 8 > p.K) v = mask_tail8(v, p.K - k0 > 0 ? p.K - k0 : 0);
        if (!live || k0 >= p.K) v = u32x4{0u, 0u, 0u, 0u};
        bfr[t][ks] = __builtin_bit_cast(bf16x8, v);
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int n = nw0 + (r >> 2) * (4 * TPW) + 4 * t + (r & 3);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k0 = 32 * ks + 8 * q;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (n < p.N && k0 < p.K && !(X3 > 0 && nw0 >= p.N)) {
          if (!p.b_mn) {
            v = *(const u32x4*)(Bb + ((long)n * p.ldb + k0) * 2);
            if (k0 + 8 > p.K) v = mask_tail8(v, p.K - k0);
          } else {
            v = frag_mn((const unsigned short*)Bb, p.ldb, n, k0, p.K);
          }
        }
        bfr[t][ks] = __builtin_bit_cast(bf16x8, v);
      }
    }
  }
  bf16x8 bfr2[TPW][KS2 > 0 ? KS2 : 1];
  int cur_s = -1;
  // X3: the waves whose column tiles lie beyond the N columns of B own tiles of the EXTRA product against the per-sample B3 instead
  // (same fragments, same loop; their fragments are re-read when the block moves on to the next sample and their results go, in fp32,
  // to C3).  Wave-uniform.
  const bool xw = X3 > 0 && nw0 >= p.N;
  const bf16x8 ones8 = __builtin_bit_cast(bf16x8, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});      // 8 x bf16 1.0

  // Everything pending at this point (the B fragments) is waited for HERE, once: a wait the compiler has to place itself ends up
  // in front of the first MFMA of the loop body (the fragments are loaded under conditions), where -- the memory counter being
  // in-order -- it also waits for the NEXT tile's loads every iteration and the stream stops overlapping with the arithmetic.
  __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)

  // The loads of a row tile are UNCONDITIONAL (clamped addresses; rows / columns outside the matrix are zeroed when the tile is
  // moved into the LDS): the number of loads in flight behind any other memory operation is then a compile-time constant and the
  // waits in the loop are exact counts, not "everything".
  u32x4 rs0[NLD], rs0b[NLD2], rs1[PF2 ? NLD : 1], rs1b[PF2 ? NLD2 : 1];
  auto gload = [&](int tile, u32x4 (&ra)[NLD], u32x4 (&ra2)[NLD2]) {
    const int s = tile / p.tps, m0 = (tile - s * p.tps) * BM;
    const char* A1 = p.A + ((long)s * p.sA1 + (long)g * p.sA2) * 2;
    const char* A2 = KS2 > 0 ? p.A2 + ((long)s * p.s2A1 + (long)g * p.s2A2) * 2 : A1;
    if constexpr (A2MN) {
#pragma unroll
      for (int i = 0; i < NLD2; ++i) {
        const int c = min(tid + i * NT, TOT2 - 1);
        const int k2 = c / (BM / 8), m = m0 + (c % (BM / 8)) * 8;
        ra2[i] = *(const u32x4*)(A2 + ((long)(k2 < p.K2 ? k2 : 0) * p.lda2 + (m < p.Mper ? m : 0)) * 2);
      }
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = min(tid + i * NT, TOT - 1);
      const int row = c / CPR, cc = c % CPR, gm = min(m0 + row, p.Mper - 1);
      const bool seg2 = KS2 > 0 && !A2MN && cc >= KS * 4;
      const int k = (seg2 ? cc - KS * 4 : cc) * 8;
      const char* src = seg2 ? A2 + ((long)gm * p.lda2 + (k < p.K2 ? k : 0)) * 2 : A1 + ((long)gm * p.lda + (k < p.K ? k : 0)) * 2;
#if STREAM_DISSECT == 3
      ra[i] = u32x4{(unsigned)c, 0u, 0u, 0u}; (void)src;
#else
      ra[i] = *(const u32x4*)src;
#endif
    }
  };
  auto lstore = [&](int buf, int tile, const u32x4 (&ra)[NLD], const u32x4 (&ra2)[NLD2]) {
    const int s = tile / p.tps, m0 = (tile - s * p.tps) * BM;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * NT;
      if (c < TOT) {
        const int row = c / CPR, cc = c % CPR;
        const bool seg2 = KS2 > 0 && !A2MN && cc >= KS * 4;
        const int k = (seg2 ? cc - KS * 4 : cc) * 8, kend = seg2 ? p.K2 : p.K;
        u32x4 v = ra[i];
        if (m0 + row >= p.Mper || k >= kend) v = u32x4{0u, 0u, 0u, 0u};
        else if (k + 8 > kend) v = mask_tail8(v, kend - k);
        *(u32x4*)(smem + buf * STG + row * RB + cc * 16) = v;
      }
    }
    if constexpr (A2MN) {
#pragma unroll
      for (int i = 0; i < NLD2; ++i) {
        const int c = tid + i * NT;
        if (c < TOT2) {
          const int k2 = c / (BM / 8), mr = (c % (BM / 8)) * 8;
          u32x4 v = ra2[i];
          if (k2 >= p.K2 || m0 + mr >= p.Mper) v = u32x4{0u, 0u, 0u, 0u};
          else if (m0 + mr + 8 > p.Mper) v = mask_tail8(v, p.Mper - m0 - mr);
          unsigned short* dst = (unsigned short*)(smem + buf * STG + mr * RB + (KS * 32 + k2) * 2);
#pragma unroll
          for (int j = 0; j < 8; ++j) dst[j * (RB / 2)] = (unsigned short)((v[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu);
        }
      }
    }
  };

  // tiles of this block: interleaved over the blocks, or one contiguous range (few sample changes when B2 is per sample)
  int tile, t_end, step;
  if (p.contig) {
    tile = (int)((long)p.ntiles * blockIdx.x / gridDim.x); t_end = (int)((long)p.ntiles * (blockIdx.x + 1) / gridDim.x); step = 1;
  } else {
    tile = blockIdx.x; t_end = p.ntiles; step = gridDim.x;
  }
  if (tile >= t_end) return;
  gload(tile, rs0, rs0b); lstore(0, tile, rs0, rs0b);
  __syncthreads();
  // One row tile.  Order of the memory operations (the counter they share is in-order):
  //   [B2 of a new sample] [epilogue operands of THIS tile: D, old C] [loads of tile ld_tile -> register stage ldA]
  //   per 16-row slab: MFMAs (the loads above are in flight), epilogue arithmetic (waits for D only: an exact count), and -- for
  //   all but the last HOLD slabs -- its stores
  //   stage stA (the NEXT tile, requested one or two tiles ago) -> the other LDS buffer    stores of the last HOLD slabs    barrier
  // -- consuming the stage waits for everything issued before it; the stores of the last slabs would be the freshest of those, so
  // they are issued AFTER it (HOLD: as many slabs as the register budget of the configuration holds; accumulating variants: none).
  auto body = [&](int it, int tile, u32x4 (&ldA)[NLD], u32x4 (&ldB)[NLD2], int ld_tile, const u32x4 (&stA)[NLD], const u32x4 (&stB)[NLD2]) {
    const int nxt = tile + step;
    const char* sA = smem + (it & 1) * STG;
    const int s = tile / p.tps, m0 = (tile - s * p.tps) * BM;
#if STREAM_DISSECT == 4          // dev: no per-sample reload of the second segment's B fragments
    if (KS2 > 0 && s != cur_s && p.K2 < 0) {
#else
    if (KS2 > 0 && s != cur_s) {
#endif
      cur_s = s;
      const unsigned short* B2 = (const unsigned short*)p.B2 + (long)s * p.s2B1 + (long)g * p.s2B2;
      {
        // The sample's B2 ([k2][n], MN-major) goes through the LDS, 32 rows (one K step) at a time: coalesced 16-byte row loads (all of
        // them requested up front), stored with the columns in fragment order (a wave's tile t = 16 consecutive LDS columns), read
        // back transposed (ds_read_tr16_b64) -- instead of 8 two-byte gathers per fragment register, a round trip each.
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
        constexpr int NCOLS = TPW * NW * 16, CPRB = NCOLS / 8, RB2 = NCOLS * 2 + 16;
        char* sB2 = smem + 2 * STG;
        constexpr bool AHEAD = KS2 * TPW <= 6;             // all K steps requested up front where the registers allow, else one at a time
        u32x4 st[AHEAD ? (KS2 > 0 ? KS2 : 1) : 1][TPW];
        auto b2load = [&](int ks, u32x4 (&dst)[TPW]) {
#pragma unroll
          for (int i = 0; i < TPW; ++i) {
            const int c = tid + i * NT, row = 32 * ks + c / CPRB, col = (c % CPRB) * 8;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row < p.K2 && col < p.N) {
              v = *(const u32x4*)(B2 + (long)row * p.ldb2 + col);
              if (col + 8 > p.N) v = mask_tail8(v, p.N - col);
            }
            dst[i] = v;
          }
        };
        if constexpr (AHEAD) {
#pragma unroll
          for (int ks = 0; ks < KS2; ++ks) b2load(ks, st[ks]);
        }
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
          if constexpr (!AHEAD) b2load(ks, st[0]);
          const u32x4 (&cur)[TPW] = st[AHEAD ? ks : 0];
#pragma unroll
          for (int i = 0; i < TPW; ++i) {
            const int c = tid + i * NT, row = c / CPRB, col = (c % CPRB) * 8;
            const int w = col / (16 * TPW), loc = col % (16 * TPW);          // 8 columns = two groups of 4 (same wave, consecutive tiles or row groups)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int l4 = loc + 4 * h, hi = l4 / (4 * TPW), tt = (l4 % (4 * TPW)) / 4;
              *(u32x2*)(sB2 + row * RB2 + (w * 16 * TPW + tt * 16 + hi * 4) * 2) = u32x2{cur[i][2 * h], cur[i][2 * h + 1]};
            }
          }
          __syncthreads();
#pragma unroll
          for (int t = 0; t < TPW; ++t) {
            const char* ad = sB2 + (8 * q + (r >> 2)) * RB2 + (wave * 16 * TPW + t * 16 + 4 * (r & 3)) * 2;
            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad));
            const s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad + 4 * RB2));
            const s16x8 wv = {v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
            bfr2[t][ks] = __builtin_bit_cast(bf16x8, wv);
          }
          __syncthreads();
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);        // (once per sample: keeps the conditional loads above out of the counts below)
    }
    if constexpr (X3 > 0) {
      if (xw && s != cur_s) {                // this sample's B3 fragments: row n - N of [N3][K], eight k per lane
        cur_s = s;
        const char* B3b = p.B3 + ((long)s * p.s3B1 + (long)g * p.s3B2) * 2;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const int n = nw0 - p.N + (r >> 2) * (4 * TPW) + 4 * t + (r & 3);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int k0 = 32 * ks + 8 * q;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (n < p.N3 && k0 < p.K) {
              v = *(const u32x4*)(B3b + ((long)n * p.ldb3 + k0) * 2);
              if (k0 + 8 > p.K) v = mask_tail8(v, p.K - k0);
            }
            bfr[t][ks] = __builtin_bit_cast(bf16x8, v);
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);      // (keeps the conditional loads above out of the counts below; every wave, every tile: nothing else is in flight here)
    }
    char* Cb = p.C + ((long)s * p.sC1 + (long)g * p.sC2) * osz;
    const char* Db = p.D ? p.D + ((long)s * p.sD1 + (long)g * p.sD2) * 2 : nullptr;
    const float* rsb = p.rs ? p.rs + (long)s * p.sRS1 + (long)g * p.sRS2 : nullptr;
    // lane (r, q) owns C[m0 + 16 mt + r][nl .. nl + 4 TPW): tile t supplies elements 4 t .. 4 t + 3 of that run
    const int nl = nw0 + 4 * TPW * q;
    constexpr bool ACCA = ACC && stream_acc_ahead(KS, KS2, TPW, NW, MT);      // old C of every slab requested up front
    u32x2 dv[MT][TPW], cvn[ACC ? TPW : 1], cva[ACCA ? MT : 1][TPW];
    float rsv[MT];
    auto ldc = [&](int mt) {
#pragma unroll
      for (int t = 0; t < (ACC ? TPW : 1); ++t) cvn[t] = u32x2{0u, 0u};
      if constexpr (ACC) {
        const int m = m0 + 16 * mt + r;
        if (m < p.Mper) ld_run<TPW>(Cb + ((long)m * p.ldc + nl) * 2, nl, p.N, cvn);
      }
    };
    if constexpr (ACCA) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + 16 * mt + r;
#pragma unroll
        for (int t = 0; t < TPW; ++t) cva[mt][t] = u32x2{0u, 0u};
        if (m < p.Mper) ld_run<TPW>(Cb + ((long)m * p.ldc + nl) * 2, nl, p.N, cva[mt]);
      }
    } else if constexpr (ACC) ldc(0);
    if (Db) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + 16 * mt + r;
        rsv[mt] = (m < p.Mper) ? rsb[m] : 0.f;
#pragma unroll
        for (int t = 0; t < TPW; ++t) dv[mt][t] = u32x2{0u, 0u};
        if (m < p.Mper) ld_run<TPW>(Db + ((long)m * p.ldd + nl) * 2, nl, p.N, dv[mt]);
      }
    }
    gload(ld_tile < t_end ? ld_tile : tile, ldA, ldB);          // (past the end: this tile again -- the count stays the same)
    asm volatile("" ::: "memory");          // the requests stay HERE, in front of the MFMAs (free to move, the compiler sinks them to their use)
    auto store_slab = [&](int mt, const f32x4 (&acc)[TPW]) {
      const int m = m0 + 16 * mt + r;
#if STREAM_DISSECT == 1          // dev: no stores (kept alive by a condition that never holds)
      if (m < p.Mper && acc[0][0] == 1.2345e-30f) {
#else
      if (m < p.Mper) {
#endif
        char* cp = Cb + ((long)m * p.ldc + nl) * osz;
        if (X3 > 0 && xw) {                                  // the extra product: fp32 rows of C3, columns nl - N ..
          float* c3 = p.C3 + (long)s * p.s3C1 + (long)g * p.s3C2 + (long)m * p.ldc3 + (nl - p.N);
#pragma unroll
          for (int t = 0; t < TPW; ++t)
            if (nl - p.N + 4 * t < p.N3) *(f32x4*)(c3 + 4 * t) = acc[t];
        } else if (p.Cx && nl >= p.nsplit) {                        // this lane's run belongs to the fp32 side output
          float* xp = p.Cx + (long)g * p.sCx2 + (long)m * p.ldcx + (nl - p.nsplit);
#pragma unroll
          for (int t = 0; t < TPW; ++t)
            if (nl + 4 * t < p.N) *(f32x4*)(xp + 4 * t) = acc[t];
        } else if (p.out_bf16) {
          auto pk = [&](int t, int e) { return f2bfbits(acc[t][e]) | (f2bfbits(acc[t][e + 1]) << 16); };
          if constexpr (TPW % 2 == 0) {                      // runs of 8 elements: 16-byte stores (N % 4 == 0)
#pragma unroll
            for (int t = 0; t < TPW; t += 2) {
              if (nl + 4 * t + 4 < p.N) *(u32x4*)(cp + 8 * t) = u32x4{pk(t, 0), pk(t, 2), pk(t + 1, 0), pk(t + 1, 2)};
              else if (nl + 4 * t < p.N) *(u32x2*)(cp + 8 * t) = u32x2{pk(t, 0), pk(t, 2)};
            }
          } else {
#pragma unroll
            for (int t = 0; t < TPW; ++t)
              if (nl + 4 * t < p.N) *(u32x2*)(cp + 8 * t) = u32x2{pk(t, 0), pk(t, 2)};
          }
        } else {
#pragma unroll
          for (int t = 0; t < TPW; ++t)
            if (nl + 4 * t < p.N) *(f32x4*)(cp + 16 * t) = acc[t];
        }
      }
    };
    f32x4 held[HOLD > 0 ? HOLD : 1][TPW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 acc[TPW];
      u32x2 cvc[ACC ? TPW : 1];
#pragma unroll
      for (int t = 0; t < (ACC ? TPW : 1); ++t) cvc[t] = ACCA ? cva[ACCA ? mt : 0][t] : cvn[t];
      if (ACC && !ACCA && mt + 1 < MT) ldc(mt + 1);
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 st_s = {0.f, 0.f, 0.f, 0.f}, st_q = {0.f, 0.f, 0.f, 0.f};

#if STREAM_DISSECT == 5          // dev: statistics variant without the row sums
      const bool do_s = false, do_q = false;
#else
      const bool do_s = STATS && (2 * mt) % NW == wave, do_q = STATS && (2 * mt + 1) % NW == wave;
#endif
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 af = *(const bf16x8*)(sA + (16 * mt + r) * RB + ks * 64 + q * 16);
        if constexpr (STATS) {
          if (do_s) st_s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, af, st_s, 0, 0, 0);
          if (do_q) st_q = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, af, st_q, 0, 0, 0);
        }
#pragma unroll
#if STREAM_DISSECT == 2          // dev: LDS reads without the MFMAs
        for (int t = 0; t < TPW; ++t) acc[t][0] += __builtin_bit_cast(f32x4, af)[t & 3];
#else
        for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[t][ks], af, acc[t], 0, 0, 0);
#endif
      }
      if constexpr (KS2 > 0) {
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
          const bf16x8 af = *(const bf16x8*)(sA + (16 * mt + r) * RB + (KS + ks) * 64 + q * 16);
#pragma unroll
          for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr2[t][ks], af, acc[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] *= p.alpha;
        if (Db) {
          acc[t][0] += rsv[mt] * bfbits(dv[mt][t][0] & 0xFFFFu); acc[t][1] += rsv[mt] * bfbits(dv[mt][t][0] >> 16);
          acc[t][2] += rsv[mt] * bfbits(dv[mt][t][1] & 0xFFFFu); acc[t][3] += rsv[mt] * bfbits(dv[mt][t][1] >> 16);
        }
        if constexpr (ACC) {
          acc[t][0] += bfbits(cvc[t][0] & 0xFFFFu); acc[t][1] += bfbits(cvc[t][0] >> 16);
          acc[t][2] += bfbits(cvc[t][1] & 0xFFFFu); acc[t][3] += bfbits(cvc[t][1] >> 16);
        }
      }
      if constexpr (STATS) {
        const int m = m0 + 16 * mt + r;
        if (m < p.Mper) {
          float* sr = p.st_rows + (long)(2 * g) * p.st_ntot + (long)s * p.Mper + m;
          if (do_s && q == 0) sr[0] = st_s[0];
          if (do_q && q == (r >> 2)) sr[p.st_ntot] = st_q[r & 3];
        }
      }
      if (mt < MT - HOLD) store_slab(mt, acc);
      else {
#pragma unroll
        for (int t = 0; t < TPW; ++t) held[HOLD > 0 ? mt - (MT - HOLD) : 0][t] = acc[t];
      }
    }
#if STREAM_DISSECT == 6          // dev: statistics variant without the column sums
    if constexpr (false) {
#else
    if constexpr (STATS) {
#endif
      typedef __attribute__((ext_vector_type(4))) short s16x4;
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
      float* cdst = p.st_cols + ((long)tile * gridDim.y + g) * p.K;
#pragma unroll
      for (int cj = 0; cj < (2 * KS + NW - 1) / NW; ++cj) {
        const int ct = wave + cj * NW;                     // 16-channel tile of this wave
        if (ct < 2 * KS) {
          f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tk = 0; tk < BM / 32; ++tk) {
            const char* ad = sA + (tk * 32 + 8 * q + (r >> 2)) * RB + (ct * 16 + 4 * (r & 3)) * 2;
            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad));
            const s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad + 4 * RB));
            const s16x8 w = {v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), ones8, c, 0, 0, 0);
          }
          const int ch = ct * 16 + 4 * q;                  // lane (r, q): channels ch .. ch + 3 (the same on every r)
          if (r == 0 && ch < p.K) {
            if (ch + 4 <= p.K) *(f32x4*)(cdst + ch) = c;
            else for (int e = 0; e < 4 && ch + e < p.K; ++e) cdst[ch + e] = c[e];
          }
        }
      }
    }
    if (nxt < t_end) lstore((it + 1) & 1, nxt, stA, stB);
#pragma unroll
    for (int h = 0; h < HOLD; ++h) store_slab(MT - HOLD + h, held[h]);
    __syncthreads();
  };
  if constexpr (!PF2) {
    for (int it = 0; tile < t_end; ++it, tile += step) body(it, tile, rs0, rs0b, tile + step, rs0, rs0b);
  } else {
    gload(tile + step < t_end ? tile + step : tile, rs0, rs0b);          // stage 0: the next tile, already in flight
    for (int it = 0; tile < t_end;) {
      body(it, tile, rs1, rs1b, tile + 2 * step, rs0, rs0b); ++it; tile += step;
      if (tile >= t_end) break;
      body(it, tile, rs0, rs0b, tile + 2 * step, rs1, rs1b); ++it; tile += step;
    }
  }
}

template <int KS, int KS2, int TPW, int NW, int BM, bool A2MN, bool ACC, bool PF2, int MINW, bool STATS = false, int X3 = 0>
int launch_inst2(const StreamArgs& s, int nb2, int per_cu, hipStream_t st) {
  constexpr int LDS = 2 * BM * ((KS + KS2) * 64 + 16) + (KS2 > 0 ? 32 * (TPW * NW * 32 + 16) : 0);      // two A stages + one K step of B2
  auto kern = gemm_stream_kernel<KS, KS2, TPW, NW, BM, A2MN, ACC, PF2, MINW, STATS, X3>;
  static LdsAttrOnce attr;
  AVMOE_TRY(attr.ensure((const void*)kern, LDS, "gemm_stream"));
  static int cus = 0;
  if (!cus) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { set_last_error("gemm_stream: device query"); return ERR_LAUNCH; }
    cus = prop.multiProcessorCount;
  }
  int gx = std::max(1, cus * per_cu / nb2);
  gx = std::min(gx, s.ntiles);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)nb2), dim3(NW * 64), LDS, st, s);
  AVMOE_CHECK_LAUNCH("gemm_stream");
  return OK;
}
template <int KS, int KS2, int TPW, int NW, int BM, bool A2MN, bool ACC, int MINW = (NW == 9 ? SC_DAP_MINW : 1), bool STATS = false, int X3 = 0>
int launch_inst(const StreamArgs& s, int nb2, int per_cu, hipStream_t st) {
  if constexpr (STATS) return launch_inst2<KS, KS2, TPW, NW, BM, A2MN, ACC, false, MINW, true, X3>(s, nb2, per_cu, st);
  // PF2 (two row tiles of loads in flight per block) is a development switch, AVMOE_STREAM_PF2=1: measured on MI355X it changes
  // no configuration by more than +-3 % (same-box A/B of the cfg-2 step: 6.51 vs 6.50 ms) -- these kernels are not short of
  // bytes in flight; the 9-wave dApost configuration is held back by residency (105 VGPRs x 9 waves: one block per CU).
  // (Not built for the MN-major second segment: those kernels sit at their register limit already and would spill.)
  static const bool pf2 = dev_env("AVMOE_STREAM_PF2") != nullptr;
  if constexpr (A2MN) return launch_inst2<KS, KS2, TPW, NW, BM, A2MN, ACC, false, MINW>(s, nb2, per_cu, st);
  else {
    if (pf2) return launch_inst2<KS, KS2, TPW, NW, BM, A2MN, ACC, true, MINW>(s, nb2, per_cu, st);
    return launch_inst2<KS, KS2, TPW, NW, BM, A2MN, ACC, false, MINW>(s, nb2, per_cu, st);
  }
}

}  // namespace

// Returns OK when the product was launched, 1 when this shape is not one the streaming kernel is built for (the caller
// then uses the tiled engine), negative on error.
int launch_gemm_stream(const GemmArgs& a_in, hipStream_t st) {
  GemmArgs a = a_in;
  bool a2mn = false;
  if (a.A2 && a.a_layout == MN_MAJOR && a.b_layout == MN_MAJOR && a.s2B1 == 0 && a.sB1 != 0) {
    // [MN-major A, per-sample B] + [K-major A2, shared B2]  (dY = Bm^T dV + dR^T Q): the shared-B segment becomes the
    // stationary first segment, the MN-major one the per-sample second segment
    std::swap(a.A, a.A2); std::swap(a.B, a.B2); std::swap(a.K, a.K2); std::swap(a.lda, a.lda2); std::swap(a.ldb, a.ldb2);
    std::swap(a.sA1, a.s2A1); std::swap(a.sA2, a.s2A2); std::swap(a.sB1, a.s2B1); std::swap(a.sB2, a.s2B2);
    a.a_layout = K_MAJOR; a.b_layout = MN_MAJOR;
    a2mn = true;
  }
  static const bool dy_whole = dev_env("AVMOE_STREAM_DY_WHOLE") != nullptr;      // dev switch
  if (a2mn && a.nb2 == 1 && a.N > 384 && a.N % 32 == 0 && !a.D && !a.Cx && !dy_whole) {
    // dY over more than 24 column tiles: the two halves of the columns as two "groups" (A shared, B / B2 / C offset by half the
    // columns) -- two tiles per wave instead of four: the 12-wave configuration then has registers for its fragments and for the
    // B2 staging (the four-tile one spills), at the price of reading the skinny A twice
    a.nb2 = 2; a.N /= 2;
    a.sA2 = 0; a.s2A2 = 0; a.sB2 = a.b_layout == MN_MAJOR ? (long)a.N : (long)a.N * a.ldb; a.s2B2 = a.N; a.sC2 = a.N;
  }
  if (a.dtype != GEMM_BF16 || a.a_layout != K_MAJOR || a.ksplit > 1 || (a.accumulate && a.out_dtype != GEMM_BF16) || a.sCj != 1 || (long)a.M * a.nb1 < 256 ||
      a.K > 384 || a.N > 768 || (a.nb1 > 1 && (a.sB1 != 0 || a.M < 64)))
    return 1;
  if (a.A2 && (a.K2 > 96 || a.s2A1 == 0)) return 1;
  if (a.Cx && (a.nb1 != 1 || a.accumulate || a.nsplit % 32 || ((uintptr_t)a.Cx % 16) || a.ldcx % 4 || a.sCx2 % 4)) return 1;
  const int osz = a.out_dtype == GEMM_BF16 ? 2 : 4;
  if (((uintptr_t)a.C % 16) || (a.sCi * osz) % 16 || (a.sC1 * osz) % 16 || (a.sC2 * osz) % 16 || (a.N % 4)) return 1;
  if (a.D && (a.row_scale == nullptr || ((uintptr_t)a.D % 16) || (a.sDi * 2) % 16 || (a.sD1 * 2) % 16 || (a.sD2 * 2) % 16)) return 1;
  if (a.b_layout == K_MAJOR && (((uintptr_t)a.B % 16) || (a.ldb * 2) % 16 || (a.sB2 * 2) % 16)) return 1;
  StreamArgs s;
  s.A = (const char*)a.A; s.B = (const char*)a.B; s.A2 = (const char*)a.A2; s.B2 = (const char*)a.B2; s.C = (char*)a.C;
  s.D = (const char*)a.D; s.rs = a.row_scale;
  s.Mper = a.M; s.nsamp = a.nb1; s.N = a.N; s.K = a.K; s.K2 = a.A2 ? a.K2 : 0;
  s.lda = a.lda; s.ldb = a.ldb; s.lda2 = a.lda2; s.ldb2 = a.ldb2; s.ldc = a.sCi; s.ldd = a.sDi;
  s.sA1 = a.sA1; s.sA2 = a.sA2; s.sB2 = a.sB2; s.s2A1 = a.s2A1; s.s2A2 = a.s2A2; s.s2B1 = a.s2B1; s.s2B2 = a.s2B2;
  s.sC1 = a.sC1; s.sC2 = a.sC2; s.sD1 = a.sD1; s.sD2 = a.sD2; s.sRS1 = a.sRS1; s.sRS2 = a.sRS2;
  s.alpha = a.alpha; s.b_mn = a.b_layout == MN_MAJOR; s.out_bf16 = a.out_dtype == GEMM_BF16;
  if (a.A2 && (((uintptr_t)a.B2 % 16) || a.ldb2 % 8 || a.s2B1 % 8 || a.s2B2 % 8)) return 1;      // B2 rows are read as 16-byte vectors
  s.contig = a.A2 != nullptr;
  static const char* contig_env = dev_env("AVMOE_STREAM_CONTIG");          // dev: force the tile-to-block assignment
  if (contig_env) s.contig = atoi(contig_env);
  s.Cx = a.Cx; s.nsplit = a.nsplit; s.ldcx = a.ldcx; s.sCx2 = a.sCx2;
  const int ks = cdiv(a.K, 32), ks2 = a.A2 ? cdiv(a.K2, 32) : 0, tiles = cdiv(a.N, 16);
  const double nb = (double)a.nb1 * a.nb2;
  const double abytes = (nb * a.M * (double)(a.K + s.K2) + (double)a.nb2 * a.N * (double)a.K + nb * a.N * (double)s.K2) * 2.0 +
                        nb * a.M * (double)a.N * osz * (a.accumulate ? 2.0 : 1.0) + (a.D ? nb * a.M * (double)a.N * 2.0 : 0.0);
  const double flops = 2.0 * nb * a.M * (double)a.N * (a.K + s.K2);
#ifndef SC_OUT_BM
#define SC_OUT_BM 64
#endif
#ifndef SC_OUT_PC
#define SC_OUT_PC 1
#endif
#ifndef SC_DOWN_PC
#define SC_DOWN_PC 2
#endif
#ifndef SC_DAP_PC
#define SC_DAP_PC 1
#endif
#ifndef SC_DAP_BM
#define SC_DAP_BM 64
#endif
#ifndef SC_DX_PC
#define SC_DX_PC 1
#endif
#ifndef SC_ON_BM
#define SC_ON_BM 32
#endif
#ifndef SC_ON_PC
#define SC_ON_PC 6
#endif
#ifndef SC_DXN_PC
#define SC_DXN_PC 4
#endif
#ifndef SC_X64_TPW
#define SC_X64_TPW 1         // the down projection + extra columns: column tiles per wave (12 / TPW waves), rows per tile, blocks per CU
#define SC_X64_BM 64
#define SC_X64_PC 1
#endif
#ifndef SC_DXN_BM
#define SC_DXN_BM 32
#endif
#ifndef SC_DY_PC
#define SC_DY_PC 1
#endif
// (K steps of 32 of segment 1 / 2, column tiles per wave, waves, rows per tile, blocks per CU) -- picked by a sweep on MI355X
// (scripts/stream_sweep.py): many waves per block and ONE block per CU win for the write-heavy shapes
  s.st_rows = a.st_rows; s.st_cols = a.st_cols; s.st_ntot = a.st_ntot;
  s.B3 = (const char*)a.B3; s.C3 = a.C3; s.N3 = a.N3; s.ldb3 = a.ldb3; s.s3B1 = a.s3B1; s.s3B2 = a.s3B2; s.ldc3 = a.ldc3; s.s3C1 = a.s3C1; s.s3C2 = a.s3C2;
  if (a.B3 && !a.st_rows) return 1;
  if (a.st_rows) {                  // the product + the statistics of A: the three configurations that serve down projections
    if (a.A2 || a2mn || a.accumulate || a.D || a.Cx || !a.st_cols || !a.st_tiles || a.out_dtype != GEMM_BF16) return 1;
    if (a.B3) {                     // + the extra columns against the per-sample B3: the k384_n128 configuration, one more tile per wave
      if (!a.C3 || a.N3 < 1 || a.N3 > 64 || a.N3 % 4 || a.nb1 < 2 || ((uintptr_t)a.B3 % 16) || (a.ldb3 * 2) % 16 || (a.s3B1 * 2) % 16 || (a.s3B2 * 2) % 16 ||
          ((uintptr_t)a.C3 % 16) || a.ldc3 % 4 || a.s3C1 % 4 || a.s3C2 % 4 || tiles > 8 || ks > 12 || a.N3 % 16)
        return 1;
      s.contig = 1;                 // (contiguous tile ranges: a block changes sample once or twice)
      if (a.N != 128) return 1;     // (eight stationary tiles + four per-sample ones: twelve waves, one tile each, one 64-row block per CU)
      s.tps = cdiv(a.M, SC_X64_BM); s.ntiles = s.tps * a.nb1; *a.st_tiles = s.tps;
      const double b3 = nb * a.N3 * (double)a.K * 2.0 + nb * a.M * (double)a.N3 * 4.0;
      ProfScope ps("gemm_stream_k384_n128+stats+x64", (long)a.M * a.nb1, abytes + b3, flops + 2.0 * nb * a.M * (double)a.N3 * a.K, st);
      return launch_inst<12, 0, SC_X64_TPW, 12 / SC_X64_TPW, SC_X64_BM, false, false, stream_minw(12 / SC_X64_TPW, SC_X64_PC), true, 1>(s, a.nb2, SC_X64_PC, st);
    }
#define STATS_CASE(COND, KS_, TPW_, NW_, BM_, PERCU_, NAME)                                                   \
    if ((COND) && ks <= KS_ && tiles <= TPW_ * NW_) {                                                         \
      s.tps = cdiv(a.M, BM_); s.ntiles = s.tps * a.nb1; *a.st_tiles = s.tps;                                  \
      ProfScope ps(NAME, (long)a.M * a.nb1, abytes, flops, st);                                               \
      return launch_inst<KS_, 0, TPW_, NW_, BM_, false, false, stream_minw(NW_, PERCU_), true>(s, a.nb2, PERCU_, st);                \
    }
    STATS_CASE(ks <= 2, 2, 2, 6, 128, 4, "gemm_stream_k64_n192+stats")
    STATS_CASE(ks <= 5, 5, 2, 12, 64, 1, "gemm_stream_k160_n384+stats")
    STATS_CASE(true, 12, 2, 4, 32, SC_DOWN_PC, "gemm_stream_k384_n128+stats")
#undef STATS_CASE
    return 1;
  }
#ifdef STREAM_SWEEP
  // development build: AVMOE_STREAM_CFG="KS,KS2,TPW,NW,BM,blocks per CU" picks one of the configurations below for every shape it fits
  // (scripts/stream_sweep.py); waves per SIMD of the launch bound = what that residency needs
  if (const char* e = dev_env("AVMOE_STREAM_CFG")) {
    int c[6] = {0, 0, 0, 0, 0, 0};
    sscanf(e, "%d,%d,%d,%d,%d,%d", &c[0], &c[1], &c[2], &c[3], &c[4], &c[5]);
#define SW(KS_, KS2_, TPW_, NW_, BM_, PC_)                                                                                     \
    if (!a2mn && !a.accumulate && c[0] == KS_ && c[1] == KS2_ && c[2] == TPW_ && c[3] == NW_ && c[4] == BM_ && c[5] == PC_ &&   \
        ks <= KS_ && ks2 <= KS2_ && (ks2 > 0) == (KS2_ > 0) && tiles <= TPW_ * NW_) {                                           \
      s.tps = cdiv(a.M, BM_); s.ntiles = s.tps * a.nb1;                                                                         \
      ProfScope ps("gemm_stream_sweep", (long)a.M * a.nb1, abytes, flops, st);                                                  \
      return launch_inst<KS_, KS2_, TPW_, NW_, BM_, false, false, (NW_ * PC_ + 3) / 4>(s, a.nb2, PC_, st);                      \
    }
    SW(12, 0, 1, 9, 32, 1) SW(12, 0, 1, 9, 64, 1) SW(12, 0, 2, 5, 32, 1) SW(12, 0, 2, 5, 32, 2) SW(12, 0, 2, 5, 64, 2) SW(12, 0, 3, 3, 32, 2)
    SW(12, 0, 3, 3, 32, 3) SW(12, 0, 3, 3, 64, 2) SW(12, 0, 2, 4, 32, 2) SW(12, 0, 2, 4, 64, 2) SW(12, 0, 2, 4, 32, 3) SW(12, 0, 1, 8, 32, 1)
    SW(12, 0, 1, 8, 32, 2)
    SW(5, 0, 2, 12, 64, 1) SW(5, 0, 4, 6, 64, 2) SW(5, 0, 4, 6, 32, 2) SW(5, 0, 3, 8, 64, 1) SW(5, 0, 3, 8, 32, 2) SW(5, 0, 6, 4, 32, 2)
    SW(5, 0, 6, 4, 32, 3) SW(5, 0, 2, 12, 32, 1) SW(5, 0, 4, 6, 32, 3)
    SW(4, 3, 2, 12, 64, 1) SW(4, 3, 4, 6, 32, 2) SW(4, 3, 3, 8, 32, 1) SW(4, 3, 3, 8, 32, 2) SW(4, 3, 2, 12, 32, 1) SW(4, 3, 4, 6, 64, 2)
#undef SW
    return 1;
  }
#endif
#define STREAM_CASE(COND, KS_, KS2_, TPW_, NW_, BM_, PERCU_, A2MN_, ACCOK_, NAME)   \
  if ((COND) && a2mn == A2MN_ && ks <= KS_ && ks2 <= KS2_ && (ks2 > 0) == (KS2_ > 0) && tiles <= TPW_ * NW_) {   \
    s.tps = cdiv(a.M, BM_); s.ntiles = s.tps * a.nb1;                               \
    ProfScope ps(NAME, (long)a.M * a.nb1, abytes, flops, st);                       \
    if constexpr (ACCOK_) {                                                         \
      if (a.accumulate) return launch_inst<KS_, KS2_, TPW_, NW_, BM_, A2MN_, true, stream_minw(NW_, PERCU_)>(s, a.nb2, PERCU_, st);   \
    } else if (a.accumulate) return 1;                                              \
    return launch_inst<KS_, KS2_, TPW_, NW_, BM_, A2MN_, false, stream_minw(NW_, PERCU_)>(s, a.nb2, PERCU_, st);   \
  }
  // narrow channel groups (C / g <= 64: the first stages of Swin / HTS-AT): long row tiles, few waves -- the wide configurations
  // below leave most of their waves without a column tile there
  STREAM_CASE(ks <= 2 && ks2 == 0, 2, 0, 2, 6, 128, 4, false, false, "gemm_stream_k64_n192")    // down projection / dApost from <= 64 channels per group
  STREAM_CASE(tiles <= 4, 5, 0, 1, 4, SC_ON_BM, SC_ON_PC, false, true, "gemm_stream_k160_n64")               // output GEMM into <= 64 channels per group
  STREAM_CASE(tiles <= 4 && a.M % SC_DXN_BM == 0, 4, 3, 1, 4, SC_DXN_BM, SC_DXN_PC, false, true, "gemm_stream_k128+96_n64")   // dX of such a site
  STREAM_CASE(tiles <= 12 && a.M % 64 == 0, 2, 3, 1, 12, 64, 1, true, true, "gemm_stream_k64+96mn_n192")   // dY into <= 192 channels
  STREAM_CASE(true, 5, 0, 2, 12, SC_OUT_BM, SC_OUT_PC, false, true, "gemm_stream_k160_n384")    // output GEMM: K = 4*32 + 12, N = 384 per group (+= for accumulate_out)
  STREAM_CASE(true, 12, 0, 2, 4, 32, SC_DOWN_PC, false, false, "gemm_stream_k384_n128")   // grouped down projection
  STREAM_CASE(true, 12, 0, 1, 9, SC_DAP_BM, SC_DAP_PC, false, false, "gemm_stream_k384_n144")   // dApost = dOut Bpost (N = 140)
  STREAM_CASE(a.M % 64 == 0, 4, 3, 2, 12, 64, SC_DX_PC, false, true, "gemm_stream_k128+96_n384")   // dX = dZx Wt + [dL2|dsx|1][T;1;dm1/N] + 2 dSxx X
  STREAM_CASE(true, 4, 3, 2, 12, 32, 1, false, true, "gemm_stream_k128+96_n384r")      // ... ragged frames
  STREAM_CASE(a.M % 64 == 0 && tiles <= 24, 2, 3, 2, 12, 64, SC_DY_PC, true, true, "gemm_stream_k64+96mn_n384")   // dY = dR^T Q + [Bm ; wbar]^T dV, half the columns per block
  STREAM_CASE(tiles <= 24, 2, 3, 2, 12, 32, 1, true, true, "gemm_stream_k64+96mn_n384r")      // ... ragged frames
  STREAM_CASE(a.M % 64 == 0, 2, 3, 4, 12, 64, SC_DY_PC, true, true, "gemm_stream_k64+96mn_n768")   // ... all 768 columns per block
  STREAM_CASE(true, 2, 3, 4, 12, 32, 1, true, true, "gemm_stream_k64+96mn_n768r")      // ... ragged frames
#undef STREAM_CASE
  return 1;
}

bool gemm_stream_stats_ok(int M, int nb1, int N, int K, long lda, long ldc) {
  const int ks = cdiv(K, 32), tiles = cdiv(N, 16);
  if (M < 64 || (long)M * nb1 < 256 || K > 384 || K % 8 || lda % 8 || ldc % 8 || N % 8) return false;       // (the checks of launch_gemm_stream for such a product)
  return (ks <= 2 && tiles <= 12) || (ks <= 5 && tiles <= 24) || tiles <= 8;
}

}  // namespace avmoe
