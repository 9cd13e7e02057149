// Generic strided / batched MFMA GEMM engine for gfx950 -- see gemm.h for the contract.
//
// Block = 256 threads = 4 waves (2 x 2), block tile BM x BN, K staged 128 bytes per row per step
// through two LDS buffers (global -> VGPR prefetch of tile t+1 is issued before the MFMAs of tile t
// and written to the other buffer after them: one barrier per K tile).
//
// MFMA shapes: bf16 -> v_mfma_f32_16x16x32_bf16 ; f32 -> v_mfma_f32_16x16x4_f32 (exact fp32).
//   A operand: lane l supplies A[i = l&15][k-slot of quarter q = l>>4]
//   B operand: lane l supplies B[j = l&15][same k-slot]
//   C/D      : lane l holds C[i = 4q + reg][j = l&15]
// k-slot mapping (identical for both operands, any order is fine for a sum):
//   bf16: step ks, element e (0..7) -> k = 32 ks + 8 q + e
//   f32 : chunk kc, element e (0..3) -> k = 16 kc + 4 q + e   (one 16x16x4 MFMA per e)
// MN_MAJOR operands sit in LDS as [k][i]; bf16 fragments come out of it with the gfx950 transposing
// read ds_read_b64_tr_b16 (two reads of 4 k-rows x 16 columns per 16-lane group), f32 with ds_read_b32.
//
// The epilogue goes through LDS so that global stores (and the optional D read) are whole rows:
// 16 B per lane along the contiguous index of C.
#include "gemm.h"
#include "common.h"
#include "prof.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace avmoe {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct DevArgs {
  const char* A; const char* B; char* C; const char* D; const float* rs; float* slabs;
  int M, N, K, nb2, nb3, ksplit, kper, nbatch, tiles_n;
  long sA3, sB3, sC3;
  long lda, ldb, sA1, sA2, sB1, sB2, sCi, sCj, sC1, sC2, sRS1, sRS2, sDi, sD1, sD2;
  float alpha; int accumulate, out_bf16, vec_c, vec_d;
  int fold_rps, fold_valid;      // batch folded into M: rows per sample / stored rows per sample (0 = no fold)
  const char* A2; const char* B2; int K2; long lda2, ldb2, s2A1, s2A2, s2B1, s2B2;
  int epi; float* row_part; const float* row_lse;          // softmax epilogues (GemmArgs::epi)
  const char* A3s; const char* B3s; const char* A4s; const char* B4s; int K3s, K4s;      // third / fourth K segment (GemmArgs::A3s ..)
  long lda3s, ldb3s, s3sA1, s3sB1, s3sB2, lda4s, ldb4s, s4sA1, s4sB1, s4sB2;
};

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) {
  return __builtin_bit_cast(float, ((unsigned int)h) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float x) {
  return __builtin_bit_cast(unsigned short, (__bf16)x);
}

// zero the elements >= valid of a 16-byte chunk (K tail of a K_MAJOR operand)
template <typename T>
__device__ __forceinline__ u32x4 mask_tail(u32x4 v, int valid) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) if (e >= valid) v[e] = 0u;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (2 * e >= valid) v[e] = 0u;
      else if (2 * e + 1 >= valid) v[e] &= 0xFFFFu;
    }
  }
  return v;
}

// bytes of K per operand row and stage: 128; 256 for the 64 x 64 tile with both operands MN-major (the token contractions: their K
// loop is bound by one memory round trip per step, and half the steps measured 140 -> 117 us on dQ = sum_s dR[s] Y[s]; for K-major
// operands -- 256-byte row pieces -- the same change measured 3 - 60 % SLOWER)
// Element-type tag of the fp32 instances that run on the bf16 matrix pipe in three-plane form (GemmArgs::split3; gemm_segment has the arithmetic)
struct f32s3 { float v; };
// ... and in TWO-plane form (GemmArgs::split3 == 2): x ~ p0 + p1 (16 mantissa bits), the three plane products of order <= 1 -- 2^-16 relative per
// product (1.5e-5), half the matrix-pipe work, the split done once per element on the way into the LDS.  Offered through avmoe_gemm
// (fp32_planes = 2); the site calls keep three planes (moe_run.h: AVMOE_BWD_PLANES -- measured -18 % of the fp32 step, but structurally cancelling
// gradients leave the 1e-3 bar)
struct f32s2 { float v; };
template <typename T> struct is_split : std::integral_constant<bool, std::is_same<T, f32s3>::value || std::is_same<T, f32s2>::value> {};
constexpr int stage_kbytes(int BM, int BN, bool AMN, bool BMN) { return (BM <= 64 && BN <= 64 && AMN && BMN) ? 256 : 128; }

// One K segment of the block's tile: 2-stage LDS pipeline over [kbeg, kend), accumulating into acc.
// Ends on a barrier, so a following segment (or the epilogue) may reuse the LDS buffer.
// (NTHR threads = NTHR / 64 waves arranged WGM x WGN over the block tile; the defaults are the 2 x 2 waves of gemm_kernel)
template <typename T, int BM, int BN, bool AMN, bool BMN, int TM, int TN, int NTHR = 256, int WGN = 2>
__device__ __forceinline__ void gemm_segment(char* smem, const char* Ab, const char* Bb, long lda, long ldb, int M, int N, int m0,
                                             int n0, int kbeg, int kend, f32x4 (&acc)[TM][TN]) {
  constexpr int ESZ = sizeof(T);
  constexpr int KBY = stage_kbytes(BM, BN, AMN, BMN), BK = KBY / ESZ, CPK = KBY / 16;      // K bytes / elements / 16-byte chunks per row and stage
  constexpr int EPC = 16 / ESZ;      // elements per 16-byte chunk
  constexpr int WGM = NTHR / 64 / WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  static_assert(WM == 16 * TM && WN == 16 * TN, "wave tile");
  constexpr int A_ROWB = AMN ? BM * ESZ + 16 : KBY + 16;
  constexpr int A_BYTES = (AMN ? BK : BM) * A_ROWB;
  constexpr int B_ROWB = BMN ? BN * ESZ + 16 : KBY + 16;
  constexpr int B_BYTES = (BMN ? BK : BN) * B_ROWB;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int A_CPR = BM * ESZ / 16, B_CPR = BN * ESZ / 16;   // chunks per LDS row (MN_MAJOR)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int nkt = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  struct { int M, N; long lda, ldb; } p{M, N, lda, ldb};

  constexpr int NLA = (AMN ? BK * A_CPR : BM * CPK) / NTHR, NLB = (BMN ? BK * B_CPR : BN * CPK) / NTHR;
  u32x4 ra[NLA], rb[NLB];
  auto gload = [&](int kt) {
    const int k0 = kbeg + kt * BK;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int c = tid + NTHR * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      if constexpr (!AMN) {
        const int row = c / CPK, cc = c % CPK;
        const int gr = min(m0 + row, p.M - 1);
        const int k = k0 + cc * EPC;
        if (k < kend) {
          v = *(const u32x4*)(Ab + ((long)gr * p.lda + k) * ESZ);
          if (k + EPC > kend) v = mask_tail<T>(v, kend - k);
        }
      } else {
        const int krow = c / A_CPR, cc = c % A_CPR;
        const int k = k0 + krow, i0 = m0 + cc * EPC;
        if (k < kend && i0 < p.M) v = *(const u32x4*)(Ab + ((long)k * p.lda + i0) * ESZ);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int c = tid + NTHR * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      if constexpr (!BMN) {
        const int row = c / CPK, cc = c % CPK;
        const int gr = min(n0 + row, p.N - 1);
        const int k = k0 + cc * EPC;
        if (k < kend) {
          v = *(const u32x4*)(Bb + ((long)gr * p.ldb + k) * ESZ);
          if (k + EPC > kend) v = mask_tail<T>(v, kend - k);
        }
      } else {
        const int krow = c / B_CPR, cc = c % B_CPR;
        const int k = k0 + krow, j0 = n0 + cc * EPC;
        if (k < kend && j0 < p.N) v = *(const u32x4*)(Bb + ((long)k * p.ldb + j0) * ESZ);
      }
      rb[i] = v;
    }
  };

  auto lstore = [&](int buf) {
    char* sA = smem + buf * STAGE;
    char* sB = sA + A_BYTES;
    if constexpr (std::is_same<T, f32s2>::value) {
      // two-plane form: the split happens ONCE per element, here -- a row of the stage keeps its bytes, first half the hi plane (bf16), second
      // half the lo plane; compute() then reads ready bf16 fragments (16-byte / transposing reads, as the bf16 instances do) and issues three
      // matrix instructions per block: no conversion arithmetic in the K loop, and none repeated by the waves that share an operand
      auto split4 = [](const u32x4& v, u32x2& hi, u32x2& lo) {
        unsigned short h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x = __uint_as_float(v[j]);
          const __bf16 hb = (__bf16)x;
          h[j] = __builtin_bit_cast(unsigned short, hb);
          l[j] = __builtin_bit_cast(unsigned short, (__bf16)(x - (float)hb));
        }
        hi = u32x2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
        lo = u32x2{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
      };
#pragma unroll
      for (int i = 0; i < NLA; ++i) {
        const int c = tid + NTHR * i;
        u32x2 hi, lo;
        split4(ra[i], hi, lo);
        char* d = AMN ? sA + (c / A_CPR) * A_ROWB + (c % A_CPR) * 8 : sA + (c / CPK) * A_ROWB + (c % CPK) * 8;
        *(u32x2*)d = hi; *(u32x2*)(d + (AMN ? BM * 2 : KBY / 2)) = lo;
      }
#pragma unroll
      for (int i = 0; i < NLB; ++i) {
        const int c = tid + NTHR * i;
        u32x2 hi, lo;
        split4(rb[i], hi, lo);
        char* d = BMN ? sB + (c / B_CPR) * B_ROWB + (c % B_CPR) * 8 : sB + (c / CPK) * B_ROWB + (c % CPK) * 8;
        *(u32x2*)d = hi; *(u32x2*)(d + (BMN ? BN * 2 : KBY / 2)) = lo;
      }
    } else {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int c = tid + NTHR * i;
      if constexpr (!AMN) *(u32x4*)(sA + (c / CPK) * A_ROWB + (c % CPK) * 16) = ra[i];
      else *(u32x4*)(sA + (c / A_CPR) * A_ROWB + (c % A_CPR) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int c = tid + NTHR * i;
      if constexpr (!BMN) *(u32x4*)(sB + (c / CPK) * B_ROWB + (c % CPK) * 16) = rb[i];
      else *(u32x4*)(sB + (c / B_CPR) * B_ROWB + (c % B_CPR) * 16) = rb[i];
    }
    }
  };

  auto compute = [&](int buf) {
    const char* sA = smem + buf * STAGE;
    const char* sB = sA + A_BYTES;
    if constexpr (ESZ == 2) {
      typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 af[TM], bfr[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          if constexpr (!AMN) {
            af[tm] = *(const bf16x8*)(sA + (wm0 + 16 * tm + r) * A_ROWB + ks * 64 + q * 16);
          } else {
            const char* ad = sA + (ks * 32 + 8 * q + (r >> 2)) * A_ROWB + (wm0 + 16 * tm + 4 * (r & 3)) * 2;
            s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad));
            s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad + 4 * A_ROWB));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            s16x8 w = {v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
            af[tm] = __builtin_bit_cast(bf16x8, w);
          }
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          if constexpr (!BMN) {
            bfr[tn] = *(const bf16x8*)(sB + (wn0 + 16 * tn + r) * B_ROWB + ks * 64 + q * 16);
          } else {
            const char* ad = sB + (ks * 32 + 8 * q + (r >> 2)) * B_ROWB + (wn0 + 16 * tn + 4 * (r & 3)) * 2;
            s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad));
            s16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad + 4 * B_ROWB));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            s16x8 w = {v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
            bfr[tn] = __builtin_bit_cast(bf16x8, w);
          }
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tm], bfr[tn], acc[tm][tn], 0, 0, 0);
      }
    } else if constexpr (std::is_same<T, f32s2>::value) {
      // two planes, split at lstore: bf16 fragments of the hi / lo images (plane offset: half a K-major row / BM (BN) bf16 of an MN-major row)
      typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
      typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 ah[TM], al[TM];
        auto frag = [&](const char* base, int rowb, bool mn, int col0, int plane_off, bf16x8& h, bf16x8& l) {
          if (!mn) {
            const char* ad = base + (col0 + r) * rowb + ks * 64 + q * 16;
            h = *(const bf16x8*)ad; l = *(const bf16x8*)(ad + plane_off);
          } else {
            const char* ad = base + (ks * 32 + 8 * q + (r >> 2)) * rowb + (col0 + 4 * (r & 3)) * 2;
            const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad)), h2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad + 4 * rowb));
            const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad + plane_off)), l2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ad + plane_off + 4 * rowb));
            h = __builtin_bit_cast(bf16x8, s16x8{h1[0], h1[1], h1[2], h1[3], h2[0], h2[1], h2[2], h2[3]});
            l = __builtin_bit_cast(bf16x8, s16x8{l1[0], l1[1], l1[2], l1[3], l2[0], l2[1], l2[2], l2[3]});
          }
        };
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) frag(sA, A_ROWB, AMN, wm0 + 16 * tm, AMN ? BM * 2 : KBY / 2, ah[tm], al[tm]);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          bf16x8 bh, bl;
          frag(sB, B_ROWB, BMN, wn0 + 16 * tn, BMN ? BN * 2 : KBY / 2, bh, bl);
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            f32x4 c = acc[tm][tn];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[tm], bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bl, c, 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tm], bh, c, 0, 0, 0);
          }
        }
      }
    } else if constexpr (is_split<T>::value) {
      constexpr bool TWO = std::is_same<T, f32s2>::value;      // (kept: the in-register two-plane form; f32s2 takes the branch above)
      // fp32 operands on the bf16 matrix pipe WITHOUT giving up fp32 products: every value as three bf16 planes (8 + 8 + 8 mantissa bits:
      // x = p0 + p1 + p2 exactly up to 2^-24 |x|), the six plane products of order <= 2 as v_mfma_f32_16x16x32_bf16, fp32 accumulation:
      // 5.8e-9 relative per product against 2 - 4e-7 for the fp32 rounding of the sum itself (measured) at
      // 6 x 16 instead of 8 x 32 matrix-pipe cycles per 16 x 16 x 32 block.  The planes are formed in registers from the fp32 LDS tile.
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 af[TM][3];
        auto frag = [&](const char* base, int rowb, bool mn, int col0, bf16x8 (&p)[3]) {
          float x[8];
          if (!mn) {
            const f32x4 v0 = *(const f32x4*)(base + (col0 + r) * rowb + ks * 128 + q * 32);
            const f32x4 v1 = *(const f32x4*)(base + (col0 + r) * rowb + ks * 128 + q * 32 + 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) { x[j] = v0[j]; x[4 + j] = v1[j]; }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = *(const float*)(base + (32 * ks + 8 * q + j) * rowb + (col0 + r) * 4);
          }
          float rr[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) { const __bf16 h = (__bf16)x[j]; p[0][j] = h; rr[j] = x[j] - (float)h; }
#pragma unroll
          for (int j = 0; j < 8; ++j) { const __bf16 h = (__bf16)rr[j]; p[1][j] = h; rr[j] -= (float)h; }
          if constexpr (!TWO) {
#pragma unroll
            for (int j = 0; j < 8; ++j) p[2][j] = (__bf16)rr[j];
          }
        };
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) frag(sA, A_ROWB, AMN, wm0 + 16 * tm, af[tm]);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          bf16x8 bf[3];
          frag(sB, B_ROWB, BMN, wn0 + 16 * tn, bf);
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            f32x4 c = acc[tm][tn];
            if constexpr (!TWO) {
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tm][2], bf[0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tm][1], bf[1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tm][0], bf[2], c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tm][1], bf[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tm][0], bf[1], c, 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tm][0], bf[0], c, 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll
      for (int kc = 0; kc < BK / 16; ++kc) {
        f32x4 af[TM], bfr[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          if constexpr (!AMN) {
            af[tm] = *(const f32x4*)(sA + (wm0 + 16 * tm + r) * A_ROWB + kc * 64 + q * 16);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              af[tm][e] = *(const float*)(sA + (16 * kc + 4 * q + e) * A_ROWB + (wm0 + 16 * tm + r) * 4);
          }
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          if constexpr (!BMN) {
            bfr[tn] = *(const f32x4*)(sB + (wn0 + 16 * tn + r) * B_ROWB + kc * 64 + q * 16);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              bfr[tn][e] = *(const float*)(sB + (16 * kc + 4 * q + e) * B_ROWB + (wn0 + 16 * tn + r) * 4);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[tm][e], bfr[tn][e], acc[tm][tn], 0, 0, 0);
      }
    }
  };

  if (nkt > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) gload(kt + 1);
    compute(kt & 1);
    if (kt + 1 < nkt) lstore((kt + 1) & 1);
    __syncthreads();
  }

}

// The K loop of a tile for two K-major bf16 operands (round 5, frame_gemm.hip's recipe; every tile size): chunks of BK K entries, [A BM rows | B BN rows]
// of 2 BK bytes, go global -> LDS directly (1 KB pieces dealt to the waves, lane offsets once; the 16-byte pieces of a row XOR-swizzled by the row
// index: fragment reads without bank conflicts and no pad), NST stages with counted vmcnt waits, one barrier per chunk; nothing is staged in
// registers, nothing is written to the LDS by the waves, no condition inside the loop -- rows beyond M / N re-read the last one (never stored),
// pieces beyond a row's K entries its last piece, and the chunk that holds kend masks its fragments.  Same accumulator layout as gemm_segment: the
// epilogue is shared.
#ifndef GEMM_BIG_DIRECT
#define GEMM_BIG_DIRECT 1      // development: 0 = the register-staged loop (gemm_segment) for an A/B
#endif
#ifndef GEMM_TILE_DIRECT
#define GEMM_TILE_DIRECT 1     // the same for the four-wave tiles (128 / 64 / 32)
#endif
#ifndef GEMM_BIG_BK
#define GEMM_BIG_BK 64         // 64: two stages of 64 KB; 32: four stages of 32 KB (three chunks in flight)
#endif
template <int N> __device__ __forceinline__ void gemm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
template <int BM, int BN, int NTHR, int WGN, int TM, int TN, int BK, int NST>
__device__ __forceinline__ void gemm_segment_direct(char* smem, const char* Ab, const char* Bb, long lda, long ldb, int M, int N, int K, int m0, int n0,
                                                    int kbeg, int kend, f32x4 (&acc)[TM][TN]) {
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr int ROWB = 2 * BK, PPR = ROWB / 16, RPP = 64 / PPR;      // bytes / 16-byte pieces per row; rows per 1 KB piece
  constexpr int NW = NTHR / 64, HA = BM * ROWB, HALF = HA, STAGE = (BM + BN) * ROWB, KS = BK / 32;      // K steps per chunk
  constexpr int NLA = HA / 1024 / NW, NLB = BN * ROWB / 1024 / NW;      // 1 KB pieces per wave, of A and of B
  static_assert(NLA >= 1 && NLB >= 1 && NLA * NW * 1024 == HA && NLB * NW * 1024 == BN * ROWB, "whole pieces per wave");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  const int wm0 = (wave / WGN) * (16 * TM), wn0 = (wave % WGN) * (16 * TN);
  const int nkt = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  const int cmax8 = (K + 7) / 8 - 1;                       // last 16-byte piece of a row that starts inside its K entries
  auto swz = [](int row) { return PPR == 8 ? (row & 7) : ((row >> 2) & 3); };
  // wave w loads pieces NLA w .. NLA w + NLA - 1 of A and NLB w .. of B
  long offa[NLA], offb[NLB]; int ca[NLA], cb[NLB];
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    const int row = (NLA * wave + i) * RPP + lane / PPR;
    ca[i] = (lane % PPR) ^ swz(row);
    offa[i] = (long)min(m0 + row, M - 1) * lda * 2;
  }
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    const int row = (NLB * wave + i) * RPP + lane / PPR;
    cb[i] = (lane % PPR) ^ swz(row);
    offb[i] = (long)min(n0 + row, N - 1) * ldb * 2;
  }
  auto gload = [&](int kt) {
    char* d = smem + (kt % NST) * STAGE;
    const int k8 = (kbeg + kt * BK) / 8;
#pragma unroll
    for (int i = 0; i < NLA; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(Ab + offa[i] + min(k8 + ca[i], cmax8) * 16), (lptr_t)(d + 1024 * (NLA * wave + i)), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NLB; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(Bb + offb[i] + min(k8 + cb[i], cmax8) * 16), (lptr_t)(d + HALF + 1024 * (NLB * wave + i)), 16, 0, 0);
  };
  int fa[KS], fb[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    fa[ks] = (wm0 + r) * ROWB + (((4 * ks + q) ^ swz(r)) * 16);
    fb[ks] = HALF + (wn0 + r) * ROWB + (((4 * ks + q) ^ swz(r)) * 16);
  }
#pragma unroll
  for (int j = 0; j < NST - 1; ++j)
    if (j < nkt) gload(j);
  for (int kt = 0; kt < nkt; ++kt) {
    // in-order counter: the NLA + NLB loads of each chunk requested after chunk kt may stay in flight
    const int newer = min(NST - 2, nkt - 1 - kt);
    if (newer <= 0) gemm_wait_vm<0>();
    else if (newer == 1) gemm_wait_vm<NLA + NLB>();
    else gemm_wait_vm<2 * (NLA + NLB)>();
    __builtin_amdgcn_s_barrier();                            // everybody's pieces of chunk kt have landed, and everybody is done reading the stage requested next
    asm volatile("" ::: "memory");
    if (kt + NST - 1 < nkt) gload(kt + NST - 1);
    const char* sS = smem + (kt % NST) * STAGE;
    const int k0 = kbeg + kt * BK;
    const bool tail = k0 + BK > kend;                        // (block-uniform)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 af[TM], bfr[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) af[tm] = *(const u32x4*)(sS + fa[ks] + tm * 16 * ROWB);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bfr[tn] = *(const u32x4*)(sS + fb[ks] + tn * 16 * ROWB);
      if (tail) {
        const int nv = kend - (k0 + 32 * ks + 8 * q);        // this lane's eight K entries that belong to the segment
        u32x4 mk;
#pragma unroll
        for (int e = 0; e < 4; ++e) mk[e] = (2 * e + 1 < nv) ? 0xffffffffu : ((2 * e < nv) ? 0x0000ffffu : 0u);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) af[tm] &= mk;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bfr[tn] &= mk;
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[tm]), __builtin_bit_cast(bf16x8, bfr[tn]), acc[tm][tn], 0, 0, 0);
    }
  }
  __syncthreads();                                           // the epilogue reuses the buffer
}

template <typename T, int BM, int BN, bool AMN, bool BMN, bool SEG2, bool SEG4 = false>
__global__ void __launch_bounds__(256, (is_split<T>::value ? 2 : 1)) gemm_kernel(const DevArgs p) {
  constexpr int ESZ = sizeof(T);
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  // XCD-aware tile order: the dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own L2),
  // so workgroup w is given the logical tile  start(w % 8) + w / 8  -- the N-tiles of one M-tile (which share their A rows)
  // and neighbouring M-tiles then run on the same XCD at about the same time.
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const unsigned total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned xcd = lin & 7u, idx = lin >> 3, base = total >> 3, rem = total & 7u;
    const unsigned logical = xcd * base + min(xcd, rem) + idx;
    bx = (int)(logical % gridDim.x); by = (int)(logical / gridDim.x);
  }
  const int m0 = (bx / p.tiles_n) * BM, n0 = (bx % p.tiles_n) * BN;
  const int split = by % p.ksplit, b = by / p.ksplit;
  const int b13 = b / p.nb2, b2 = b % p.nb2;
  const int b1 = b13 / p.nb3, b3 = b13 % p.nb3;
  const char* Ab = p.A + ((long)b1 * p.sA1 + (long)b2 * p.sA2 + (long)b3 * p.sA3) * ESZ;
  const char* Bb = p.B + ((long)b1 * p.sB1 + (long)b2 * p.sB2 + (long)b3 * p.sB3) * ESZ;
  const int kbeg = split * p.kper;
  const int kend = min(p.K, kbeg + p.kper);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (std::is_same<T, __bf16>::value && !AMN && !BMN && GEMM_TILE_DIRECT) gemm_segment_direct<BM, BN, 256, 2, TM, TN, 64, 2>(smem, Ab, Bb, p.lda, p.ldb, p.M, p.N, p.K, m0, n0, kbeg, kend, acc);
  else gemm_segment<T, BM, BN, AMN, BMN, TM, TN>(smem, Ab, Bb, p.lda, p.ldb, p.M, p.N, m0, n0, kbeg, kend, acc);
  if constexpr (SEG2) {     // second K segment: A2 K-major, B2 MN-major, own batch strides (C += A2 . B2^T)
    const char* A2 = p.A2 + ((long)b1 * p.s2A1 + (long)b2 * p.s2A2) * ESZ;
    const char* B2 = p.B2 + ((long)b1 * p.s2B1 + (long)b2 * p.s2B2) * ESZ;
    gemm_segment<T, BM, BN, false, true, TM, TN>(smem, A2, B2, p.lda2, p.ldb2, p.M, p.N, m0, n0, 0, p.K2, acc);
  }
  if constexpr (SEG4) {     // third (A MN-major) and fourth (A K-major) K segments, B MN-major: the other site's dY folded into this dX product
    const char* A3 = p.A3s + (long)b1 * p.s3sA1 * ESZ;
    const char* B3 = p.B3s + ((long)b1 * p.s3sB1 + (long)b2 * p.s3sB2) * ESZ;
    gemm_segment<T, BM, BN, true, true, TM, TN>(smem, A3, B3, p.lda3s, p.ldb3s, p.M, p.N, m0, n0, 0, p.K3s, acc);
    if (p.K4s > 0) {
      const char* A4 = p.A4s + (long)b1 * p.s4sA1 * ESZ;
      const char* B4 = p.B4s + ((long)b1 * p.s4sB1 + (long)b2 * p.s4sB2) * ESZ;
      gemm_segment<T, BM, BN, false, true, TM, TN>(smem, A4, B4, p.lda4s, p.ldb4s, p.M, p.N, m0, n0, 0, p.K4s, acc);
    }
  }

  // ---- epilogue through LDS ----
  constexpr int CLD = BN + 4;
  float* Cs = (float*)smem;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        Cs[(wm0 + 16 * tm + 4 * q + e) * CLD + wn0 + 16 * tn + r] = acc[tm][tn][e];
  __syncthreads();

  if (p.ksplit > 1) {
    float* dst = p.slabs + ((long)split * p.nbatch + b) * (long)p.M * p.N;
    constexpr int TPR = BN / 4, RPP = 256 / TPR;      // (BN = 160: 40 threads per row, 6 rows per pass, 16 threads idle, a partial last pass)
#pragma unroll 1
    for (int pass = 0; pass < (BM + RPP - 1) / RPP; ++pass) {
      const int i = pass * RPP + tid / TPR, j = (tid % TPR) * 4;
      const int gi = m0 + i, gj = n0 + j;
      if (gi < p.M && gj < p.N && tid < RPP * TPR && i < BM) {
        const f32x4 v = *(const f32x4*)&Cs[i * CLD + j];
        float* d = dst + (long)gi * p.N + gj;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (gj + e < p.N) d[e] = v[e];
      }
    }
    return;
  }

  if (p.epi == GEMM_EPI_ROWSTATS) {          // (max, sum exp) of this tile's part of every row; two threads per row (BM = BN = 128)
    constexpr int TPR = 256 / BM, CPT = BN / TPR;
    const int i = tid / TPR, h = tid % TPR, gi = m0 + i;
    float mx = -INFINITY;
    for (int j = h * CPT; j < (h + 1) * CPT; ++j)
      if (n0 + j < p.N) mx = fmaxf(mx, p.alpha * Cs[i * CLD + j]);
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = h * CPT; j < (h + 1) * CPT; ++j)
      if (n0 + j < p.N) sum += __expf(p.alpha * Cs[i * CLD + j] - mx);
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) sum += __shfl_xor(sum, o, 64);
    if (h == 0 && gi < p.M) {
      float* rp = p.row_part + (((long)b * p.M + gi) * p.tiles_n + n0 / BN) * 2;
      rp[0] = mx; rp[1] = sum;
    }
    return;
  }
  char* Cb = p.C + ((long)b1 * p.sC1 + (long)b2 * p.sC2 + (long)b3 * p.sC3) * (p.out_bf16 ? 2 : 4);
  const char* Db = p.D ? p.D + ((long)b1 * p.sD1 + (long)b2 * p.sD2) * ESZ : nullptr;
  const float* rsb = p.rs ? p.rs + (long)b1 * p.sRS1 + (long)b2 * p.sRS2 : nullptr;

  auto load_out = [&](const char* ptr) -> float {
    return p.out_bf16 ? bf16_bits_to_f32(*(const unsigned short*)ptr) : *(const float*)ptr;
  };
  auto store_out = [&](char* ptr, float v) {
    if (p.out_bf16) *(unsigned short*)ptr = f32_to_bf16_bits(v);
    else *(float*)ptr = v;
  };
  auto load_d = [&](long off) -> float {
    if constexpr (ESZ == 2) return bf16_bits_to_f32(*(const unsigned short*)(Db + off * 2));
    else return *(const float*)(Db + off * 4);
  };
  const int osz = p.out_bf16 ? 2 : 4;

  if (p.sCj == 1) {
    constexpr int TPR = BN / 4, RPP = 256 / TPR;
#pragma unroll 1
    for (int pass = 0; pass < (BM + RPP - 1) / RPP; ++pass) {
      const int i = pass * RPP + tid / TPR, j = (tid % TPR) * 4;
      const int gi = m0 + i, gj = n0 + j;
      if (gi >= p.M || gj >= p.N || tid >= RPP * TPR || i >= BM) continue;
      long crow = (long)gi * p.sCi;
      if (p.fold_rps) {                                   // row gi of the folded GEMM = row ri of sample sidx; gap rows are not stored
        const int sidx = gi / p.fold_rps, ri = gi - sidx * p.fold_rps;
        if (ri >= p.fold_valid) continue;
        crow = (long)sidx * p.sC1 + (long)ri * p.sCi;
      }
      f32x4 v = *(const f32x4*)&Cs[i * CLD + j];
      const float rsv = rsb ? rsb[gi] : 0.f;
      char* cp = Cb + (crow + gj) * osz;
      const bool full = (gj + 3 < p.N);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
      if (p.epi == GEMM_EPI_EXP) {
        const float lse = p.row_lse[(long)b * p.M + gi];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __expf(v[e] - lse);
      } else if (p.epi == GEMM_EPI_MULSUB) {
        const float sub = p.row_lse[(long)b * p.M + gi];
        if (full && p.vec_d) {
          if constexpr (ESZ == 2) {
            const u32x2 dv = *(const u32x2*)(Db + ((long)gi * p.sDi + gj) * 2);
            v[0] = bf16_bits_to_f32(dv[0] & 0xFFFFu) * (v[0] - sub); v[1] = bf16_bits_to_f32(dv[0] >> 16) * (v[1] - sub);
            v[2] = bf16_bits_to_f32(dv[1] & 0xFFFFu) * (v[2] - sub); v[3] = bf16_bits_to_f32(dv[1] >> 16) * (v[3] - sub);
          } else {
            const f32x4 dv = *(const f32x4*)(Db + ((long)gi * p.sDi + gj) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = dv[e] * (v[e] - sub);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (gj + e < p.N) ? load_d((long)gi * p.sDi + gj + e) * (v[e] - sub) : 0.f;
        }
      }
      if (Db && p.epi != GEMM_EPI_MULSUB) {
        if (full && p.vec_d) {
          if constexpr (ESZ == 2) {
            const u32x2 dv = *(const u32x2*)(Db + ((long)gi * p.sDi + gj) * 2);
            v[0] += rsv * bf16_bits_to_f32(dv[0] & 0xFFFFu);
            v[1] += rsv * bf16_bits_to_f32(dv[0] >> 16);
            v[2] += rsv * bf16_bits_to_f32(dv[1] & 0xFFFFu);
            v[3] += rsv * bf16_bits_to_f32(dv[1] >> 16);
          } else {
            const f32x4 dv = *(const f32x4*)(Db + ((long)gi * p.sDi + gj) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rsv * dv[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (gj + e < p.N) v[e] += rsv * load_d((long)gi * p.sDi + gj + e);
        }
      }
      if (full && p.vec_c) {
        if (p.out_bf16) {
          if (p.accumulate) {
            const u32x2 o = *(const u32x2*)cp;
            v[0] += bf16_bits_to_f32(o[0] & 0xFFFFu); v[1] += bf16_bits_to_f32(o[0] >> 16);
            v[2] += bf16_bits_to_f32(o[1] & 0xFFFFu); v[3] += bf16_bits_to_f32(o[1] >> 16);
          }
          u32x2 o;
          o[0] = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
          o[1] = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
          *(u32x2*)cp = o;
        } else {
          if (p.accumulate) {
            const f32x4 o = *(const f32x4*)cp;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += o[e];
          }
          *(f32x4*)cp = v;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (gj + e < p.N) {
            float x = v[e];
            if (p.accumulate) x += load_out(cp + e * osz);
            store_out(cp + e * osz, x);
          }
        }
      }
    }
  } else {   // sCi == 1 : C stored transposed (i contiguous)
    constexpr int TPC = BM / 4, CPP = 256 / TPC;
#pragma unroll 1
    for (int pass = 0; pass < BN / CPP; ++pass) {
      const int j = pass * CPP + tid / TPC, i = (tid % TPC) * 4;
      const int gi = m0 + i, gj = n0 + j;
      if (gi >= p.M || gj >= p.N) continue;
      char* cp = Cb + ((long)gj * p.sCj + gi) * osz;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (gi + e < p.M) {
          char* ce = cp + e * osz;
          if (p.fold_rps) {
            const int sidx = (gi + e) / p.fold_rps, ri = (gi + e) - sidx * p.fold_rps;
            if (ri >= p.fold_valid) continue;
            ce = Cb + ((long)sidx * p.sC1 + (long)gj * p.sCj + ri) * osz;
          }
          float x = p.alpha * Cs[(i + e) * CLD + j];
          if (Db) x += (rsb ? rsb[gi + e] : 0.f) * load_d((long)(gi + e) * p.sDi + gj);
          if (p.accumulate) x += load_out(ce);
          store_out(ce, x);
        }
      }
    }
  }
}

// Large plain products (both extents in the thousands: the token remap's logits / weight gradients at the stage-0 sites of the real
// backbones, N x M = 4096 x 2304 ..): 256 x 256 block tile, 8 waves (2 x 4, a wave owns 128 x 64 = 32 MFMA tiles: 12 LDS fragment
// reads feed 32 MFMAs per K step, against 8 for 16 on the 128 x 128 tile, whose LDS traffic bounds it at ~0.55 PFLOP/s).  bf16
// operands, fp32 accumulation, the same two-stage K pipeline (gemm_segment); plain epilogue only (alpha, accumulate, split-K slabs),
// staged through LDS one half of the rows at a time.
template <bool AMN, bool BMN>
__global__ void __launch_bounds__(512) gemm_big_kernel(const DevArgs p) {
  constexpr int BM = 256, BN = 256, NTHR = 512, WGN = 4, WM = 128, WN = 64, TM = 8, TN = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wrow = wave / WGN, wn0 = (wave % WGN) * WN;
  int bx = blockIdx.x, by = blockIdx.y;
  {   // XCD-aware tile order (see gemm_kernel)
    const unsigned total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned xcd = lin & 7u, idx = lin >> 3, base = total >> 3, rem = total & 7u;
    const unsigned logical = xcd * base + min(xcd, rem) + idx;
    bx = (int)(logical % gridDim.x); by = (int)(logical / gridDim.x);
  }
  const int m0 = (bx / p.tiles_n) * BM, n0 = (bx % p.tiles_n) * BN;
  const int split = by % p.ksplit, b = by / p.ksplit;
  const int b13 = b / p.nb2, b2 = b % p.nb2;
  const int b1 = b13 / p.nb3, b3 = b13 % p.nb3;
  const char* Ab = p.A + ((long)b1 * p.sA1 + (long)b2 * p.sA2 + (long)b3 * p.sA3) * 2;
  const char* Bb = p.B + ((long)b1 * p.sB1 + (long)b2 * p.sB2 + (long)b3 * p.sB3) * 2;
  const int kbeg = split * p.kper, kend = min(p.K, kbeg + p.kper);
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (!AMN && !BMN && GEMM_BIG_DIRECT) gemm_segment_direct<256, 256, 512, 4, 8, 4, GEMM_BIG_BK, (GEMM_BIG_BK == 64 ? 2 : 4)>(smem, Ab, Bb, p.lda, p.ldb, p.M, p.N, p.K, m0, n0, kbeg, kend, acc);
  else gemm_segment<__bf16, BM, BN, AMN, BMN, TM, TN, NTHR, WGN>(smem, Ab, Bb, p.lda, p.ldb, p.M, p.N, m0, n0, kbeg, kend, acc);

  constexpr int CLD = BN + 4;
  float* Cs = (float*)smem;
  const int osz = p.out_bf16 ? 2 : 4;
  char* Cb = p.C + ((long)b1 * p.sC1 + (long)b2 * p.sC2 + (long)b3 * p.sC3) * osz;
  float* slab = p.ksplit > 1 ? p.slabs + ((long)split * p.nbatch + b) * (long)p.M * p.N : nullptr;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {                      // rows [128 h, 128 h + 128) of the tile: the waves of wave-row h hand over
    if (wrow == h) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int e = 0; e < 4; ++e) Cs[(16 * tm + 4 * q + e) * CLD + wn0 + 16 * tn + r] = acc[tm][tn][e];
    }
    __syncthreads();
    if (p.sCj == 1 || slab) {
      constexpr int TPR = BN / 4, RPP = NTHR / TPR;
#pragma unroll 1
      for (int pass = 0; pass < WM / RPP; ++pass) {
        const int i = pass * RPP + tid / TPR, j = (tid % TPR) * 4;
        const int gi = m0 + 128 * h + i, gj = n0 + j;
        if (gi >= p.M || gj >= p.N) continue;
        long crow = (long)gi * p.sCi;
        if (p.fold_rps) {                                 // row gi of the folded GEMM = row ri of sample sidx; gap rows are not stored
          const int sidx = gi / p.fold_rps, ri = gi - sidx * p.fold_rps;
          if (ri >= p.fold_valid) continue;
          crow = (long)sidx * p.sC1 + (long)ri * p.sCi;
        }
        f32x4 v = *(const f32x4*)&Cs[i * CLD + j];
        if (slab) {
          float* d = slab + (long)gi * p.N + gj;
#pragma unroll
          for (int e = 0; e < 4; ++e) if (gj + e < p.N) d[e] = v[e];
          continue;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
        char* cp = Cb + (crow + gj) * osz;
        if (gj + 3 < p.N && p.vec_c) {
          if (p.out_bf16) {
            if (p.accumulate) {
              const u32x2 o = *(const u32x2*)cp;
              v[0] += bf16_bits_to_f32(o[0] & 0xFFFFu); v[1] += bf16_bits_to_f32(o[0] >> 16);
              v[2] += bf16_bits_to_f32(o[1] & 0xFFFFu); v[3] += bf16_bits_to_f32(o[1] >> 16);
            }
            u32x2 o;
            o[0] = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
            o[1] = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
            *(u32x2*)cp = o;
          } else {
            if (p.accumulate) {
              const f32x4 o = *(const f32x4*)cp;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += o[e];
            }
            *(f32x4*)cp = v;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (gj + e < p.N) {
              float x = v[e];
              char* ce = cp + e * osz;
              if (p.accumulate) x += p.out_bf16 ? bf16_bits_to_f32(*(const unsigned short*)ce) : *(const float*)ce;
              if (p.out_bf16) *(unsigned short*)ce = f32_to_bf16_bits(x); else *(float*)ce = x;
            }
          }
        }
      }
    } else {   // sCi == 1 : C stored transposed (i contiguous)
      constexpr int TPC = WM / 4, CPP = NTHR / TPC;
#pragma unroll 1
      for (int pass = 0; pass < BN / CPP; ++pass) {
        const int j = pass * CPP + tid / TPC, i = (tid % TPC) * 4;
        const int gi = m0 + 128 * h + i, gj = n0 + j;
        if (gi >= p.M || gj >= p.N) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (gi + e < p.M) {
            char* ce = Cb + ((long)gj * p.sCj + gi + e) * osz;
            if (p.fold_rps) {
              const int sidx = (gi + e) / p.fold_rps, ri = (gi + e) - sidx * p.fold_rps;
              if (ri >= p.fold_valid) continue;
              ce = Cb + ((long)sidx * p.sC1 + (long)gj * p.sCj + ri) * osz;
            }
            float x = p.alpha * Cs[(i + e) * CLD + j];
            if (p.accumulate) x += p.out_bf16 ? bf16_bits_to_f32(*(const unsigned short*)ce) : *(const float*)ce;
            if (p.out_bf16) *(unsigned short*)ce = f32_to_bf16_bits(x); else *(float*)ce = x;
          }
        }
      }
    }
    __syncthreads();
  }
}

// split-K second pass: C = alpha * sum_s slab[s] (+ row_scale * D) (+ C)
// A wave covers 64 / P consecutive output vectors; the slabs of one vector are shared by P lanes (lane = part * (64 / P) + vector,
// part p sums slabs p, p + P, ...) and the P partial sums are combined by a fixed xor-shuffle tree -- a deterministic order.
// With one thread per vector (P = 1) a long split of a small matrix keeps a hundred workgroups busy with 64 dependent loads each.
template <typename T>
__global__ void __launch_bounds__(256) gemm_splitk_reduce(const DevArgs p, int lgP) {
  const unsigned N = (unsigned)p.N, per = (unsigned)p.M * N, total = per * (unsigned)p.nbatch;
  const unsigned VEC = (N % 4u == 0u) ? 4u : 1u;
  const unsigned nvec = total / VEC;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned P = 1u << lgP, V = 64u >> lgP;           // parts per vector, vectors per wave
  const unsigned part = lane / V, vl = lane % V;
  const unsigned nwv = (nvec + V - 1) / V;                 // wave-sized groups of vectors
  for (unsigned w = blockIdx.x * 4u + wave; w < nwv; w += gridDim.x * 4u) {
    const unsigned v = w * V + vl;
    const bool live = v < nvec;
    const unsigned idx = live ? v * VEC : 0u;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const float* sl = p.slabs + idx;
    if (live) {
      if (VEC == 4u) {
        int sp = (int)part;
        for (; sp + 3 * (int)P < p.ksplit; sp += 4 * (int)P) {
          const f32x4 a0 = *(const f32x4*)(sl + (size_t)sp * total), a1 = *(const f32x4*)(sl + (size_t)(sp + P) * total);
          const f32x4 a2 = *(const f32x4*)(sl + (size_t)(sp + 2 * P) * total), a3 = *(const f32x4*)(sl + (size_t)(sp + 3 * P) * total);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[e] += (a0[e] + a1[e]) + (a2[e] + a3[e]);
        }
        for (; sp < p.ksplit; sp += (int)P) {
          const f32x4 a0 = *(const f32x4*)(sl + (size_t)sp * total);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[e] += a0[e];
        }
      } else {
        for (int sp = (int)part; sp < p.ksplit; sp += (int)P) s[0] += sl[(size_t)sp * total];
      }
    }
    for (unsigned off = V; off < 64u; off <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += __shfl_xor(s[e], (int)off, 64);
    }
    if (!live || part != 0u) continue;
    const unsigned b = idx / per, rem = idx - b * per;
    const unsigned i = rem / N, j = rem - i * N;
    const int b13 = (int)b / p.nb2, b2 = (int)b % p.nb2;
    const int b1 = b13 / p.nb3, b3 = b13 % p.nb3;
    const float rsv = (p.D && p.rs) ? p.rs[(long)b1 * p.sRS1 + (long)b2 * p.sRS2 + i] : 0.f;
    for (unsigned e = 0; e < VEC; ++e) {
      float x = s[e] * p.alpha;
      if (p.D) {
        const long off = (long)b1 * p.sD1 + (long)b2 * p.sD2 + (long)i * p.sDi + (j + e);
        float dv;
        if constexpr (sizeof(T) == 2) dv = bf16_bits_to_f32(((const unsigned short*)p.D)[off]);
        else dv = ((const float*)p.D)[off];
        x += rsv * dv;
      }
      const long coff = (long)b1 * p.sC1 + (long)b2 * p.sC2 + (long)b3 * p.sC3 + (long)i * p.sCi + (long)(j + e) * p.sCj;
      if (p.out_bf16) {
        unsigned short* c = (unsigned short*)p.C + coff;
        if (p.accumulate) x += bf16_bits_to_f32(*c);
        *c = f32_to_bf16_bits(x);
      } else {
        float* c = (float*)p.C + coff;
        if (p.accumulate) x += *c;
        *c = x;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Two token contractions against the same tensor in one pass (TokPairArgs, gemm.h): block (channel tile, frame chunk) runs, frame by
// frame, the K loop of  A1^T X  into an accumulator it keeps for the whole chunk and the K loop of  A2^T X  into one it stores per
// frame; the second loop finds the frame's X tiles in the L2 the first one just filled.
// ---------------------------------------------------------------------------------------------
struct TokPairDev {
  const char* A1; const char* A2; const char* X; float* C2; float* slabs;
  long lda1, lda2, ldx, sA1g;
  int M1, M2, S, N, g, Cg, tiles_n, F, TK;
};
template <int BM2>
__global__ void __launch_bounds__(256, 2) gemm_tokpair_kernel(const TokPairDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  int bx = blockIdx.x, by = blockIdx.y;
  {   // XCD-aware order (see gemm_kernel): the channel tiles of one frame chunk, which share their A tiles, on one XCD
    const unsigned total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned xcd = lin & 7u, idx = lin >> 3, base = total >> 3, rem = total & 7u;
    const unsigned logical = xcd * base + min(xcd, rem) + idx;
    bx = (int)(logical % gridDim.x); by = (int)(logical / gridDim.x);
  }
  const int gi = bx / p.tiles_n, n0 = (bx % p.tiles_n) * 128;
  const int s0 = by * p.F, s1 = min(p.S, s0 + p.F);
  constexpr int TM2 = BM2 / 32;
  f32x4 accW[4][4], accT[TM2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) accW[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wn0 = (wave & 1) * 64;
  for (int s = s0; s < s1; ++s) {
    const long t0 = (long)s * p.N;
    const char* Xb = p.X + (t0 * p.ldx + (long)gi * p.Cg) * 2;
#pragma unroll
    for (int i = 0; i < TM2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) accT[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // both K loops over the same TK tokens before moving on: the second one finds its X tiles (TK x 128 channels) still in the L2
#pragma unroll 1
    for (int k0 = 0; k0 < p.N; k0 += p.TK) {
      const int k1 = min(p.N, k0 + p.TK);
      gemm_segment<__bf16, 128, 128, true, true, 4, 4>(smem, p.A1 + (t0 * p.lda1 + (long)gi * p.sA1g) * 2, Xb, p.lda1, p.ldx, p.M1, p.Cg, 0, n0, k0, k1, accW);
      gemm_segment<__bf16, BM2, 128, true, true, TM2, 4>(smem, p.A2 + t0 * p.lda2 * 2, Xb, p.lda2, p.ldx, p.M2, p.Cg, 0, n0, k0, k1, accT);
    }
    float* c2 = p.C2 + (long)s * p.M2 * ((long)p.g * p.Cg) + (long)gi * p.Cg;
    const int wm2 = (wave >> 1) * (BM2 / 2);
#pragma unroll
    for (int tm = 0; tm < TM2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const int j = n0 + wn0 + 16 * tn + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = wm2 + 16 * tm + 4 * q + e;
          if (i < p.M2 && j < p.Cg) c2[(long)i * ((long)p.g * p.Cg) + j] = accT[tm][tn][e];
        }
      }
  }
  float* c1 = p.slabs + ((long)by * p.g + gi) * p.M1 * (long)p.Cg;
  const int wm1 = (wave >> 1) * 64;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int j = n0 + wn0 + 16 * tn + r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = wm1 + 16 * tm + 4 * q + e;
        if (i < p.M1 && j < p.Cg) c1[(long)i * p.Cg + j] = accW[tm][tn][e];
      }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, bool AMN, bool BMN, bool SEG2 = false, bool SEG4 = false>
static int launch_inst(const DevArgs& d, int batch_z, hipStream_t stream) {
  constexpr int ESZ = sizeof(T);
  constexpr int KBY = stage_kbytes(BM, BN, AMN, BMN), BK = KBY / ESZ, BK2 = stage_kbytes(BM, BN, false, true) / ESZ;
  constexpr int A_BYTES = (AMN ? BK : BM) * (AMN ? BM * ESZ + 16 : KBY + 16);
  constexpr int B_BYTES = (BMN ? BK : BN) * (BMN ? BN * ESZ + 16 : KBY + 16);
  constexpr int A2_BYTES = BM * (BK2 * ESZ + 16), B2_BYTES = BK2 * (BN * ESZ + 16);
  constexpr int KBY3 = stage_kbytes(BM, BN, true, true), BK3 = KBY3 / ESZ;      // third segment: both operands MN-major
  constexpr int S1 = 2 * (A_BYTES + B_BYTES), S2 = SEG2 ? 2 * (A2_BYTES + B2_BYTES) : 0, S3 = SEG4 ? 2 * BK3 * ((BM + BN) * ESZ + 32) : 0;
  constexpr int STAGES = (S1 > S2 ? S1 : S2) > S3 ? (S1 > S2 ? S1 : S2) : S3;
  constexpr int EPI = BM * (BN + 4) * 4;
  constexpr int LDS = STAGES > EPI ? STAGES : EPI;
  static LdsAttrOnce attr;
  auto kern = gemm_kernel<T, BM, BN, AMN, BMN, SEG2, SEG4>;
  AVMOE_TRY(attr.ensure((const void*)kern, LDS, "gemm"));
  const int tiles_m = cdiv(d.M, BM);
  dim3 grid((unsigned)(tiles_m * d.tiles_n), (unsigned)batch_z, 1);
  static const char* const names[2][2] = {{"gemm_KK", "gemm_KM"}, {"gemm_MK", "gemm_MM"}};
  static char name[64];
  if (!name[0]) {
    snprintf(name, sizeof(name), "%s%s_%s_%d", names[AMN][BMN], SEG4 ? "+KM+MM+KM" : (SEG2 ? "+KM" : ""), sizeof(T) == 2 ? "bf16" : (std::is_same<T, f32s3>::value ? "f32s3" : (std::is_same<T, f32s2>::value ? "f32s2" : "f32")), BM);
    if (BN != BM) snprintf(name + strlen(name), sizeof(name) - strlen(name), "x%d", BN);
  }
  static const bool shapes = getenv("AVMOE_PROF_SHAPES") != nullptr;
  const char* pname = name;
  if (shapes && prof_enabled()) {            // debug only: one family per distinct call shape (leaks the small strings)
    char* nm = (char*)malloc(96);
    snprintf(nm, 96, "%s M%d N%d K%d b%d ks%d", name, d.M, d.N, d.K, d.nbatch, d.ksplit);
    pname = nm;
  }
  const double nb = (double)d.nbatch;
  const double esz = sizeof(T), osz = d.ksplit > 1 ? 4.0 : (d.out_bf16 ? 2.0 : 4.0);
  // algorithmic bytes: every operand element once (broadcast operands counted once), C written once (+ read if accumulating)
  const double abytes = ((d.sA1 == 0 && d.sA2 == 0 ? 1.0 : nb) * d.M * (double)d.K + (d.sB1 == 0 && d.sB2 == 0 ? 1.0 : nb) * d.N * (double)d.K +
                         (SEG2 ? nb * (d.M + d.N) * (double)d.K2 : 0.0) + (SEG4 ? nb * (d.M + d.N) * (double)(d.K3s + d.K4s) : 0.0)) * esz +
                        nb * d.M * (double)d.N * osz * (d.accumulate ? 2.0 : 1.0) + (d.D ? nb * d.M * (double)d.N * esz : 0.0);
  ProfScope ps(pname, abytes, 2.0 * nb * d.M * (double)d.N * (d.K + (SEG2 ? d.K2 : 0) + (SEG4 ? d.K3s + d.K4s : 0)), stream);
  hipLaunchKernelGGL(kern, grid, dim3(256), LDS, stream, d);
  AVMOE_CHECK_LAUNCH("gemm_kernel");
  return OK;
}

#ifndef GEMM_MIN_BLOCKS
#define GEMM_MIN_BLOCKS 160   // fewer blocks than this: the next smaller tile
#endif
#ifndef GEMM_BIG_MIN
#define GEMM_BIG_MIN 768       // smallest extent (both M and N) that takes the 256 x 256 tile
#endif
template <bool AMN, bool BMN>
static int launch_big_inst(const DevArgs& d, int batch_z, hipStream_t stream) {
  constexpr int A_BYTES = AMN ? 64 * (256 * 2 + 16) : 256 * (128 + 16), B_BYTES = BMN ? 64 * (256 * 2 + 16) : 256 * (128 + 16);
  constexpr int STAGES = 2 * (A_BYTES + B_BYTES), EPI = 128 * (256 + 4) * 4;
  constexpr int LDS = STAGES > EPI ? STAGES : EPI;
  static LdsAttrOnce attr;
  auto kern = gemm_big_kernel<AMN, BMN>;
  AVMOE_TRY(attr.ensure((const void*)kern, LDS, "gemm (256 x 256 tile)"));
  dim3 grid((unsigned)(cdiv(d.M, 256) * d.tiles_n), (unsigned)batch_z, 1);
  static const char* const names[2][2] = {{"gemm_KK", "gemm_KM"}, {"gemm_MK", "gemm_MM"}};
  static char name[64];
  if (!name[0]) snprintf(name, sizeof(name), "%s_bf16_256", names[AMN][BMN]);
  static const bool shapes = getenv("AVMOE_PROF_SHAPES") != nullptr;
  const char* pname = name;
  if (shapes && prof_enabled()) {
    char* nm = (char*)malloc(96);
    snprintf(nm, 96, "%s M%d N%d K%d b%d ks%d", name, d.M, d.N, d.K, d.nbatch, d.ksplit);
    pname = nm;
  }
  const double nb = (double)d.nbatch, osz = d.ksplit > 1 ? 4.0 : (d.out_bf16 ? 2.0 : 4.0);
  const double abytes = ((d.sA1 == 0 && d.sA2 == 0 ? 1.0 : nb) * d.M * (double)d.K + (d.sB1 == 0 && d.sB2 == 0 ? 1.0 : nb) * d.N * (double)d.K) * 2.0 +
                        nb * d.M * (double)d.N * osz * (d.accumulate ? 2.0 : 1.0);
  ProfScope ps(pname, abytes, 2.0 * nb * d.M * (double)d.N * d.K, stream);
  hipLaunchKernelGGL(kern, grid, dim3(512), LDS, stream, d);
  AVMOE_CHECK_LAUNCH("gemm_big_kernel");
  return OK;
}

template <typename T, int BM, int BN>
static int launch_layout(const GemmArgs& a, const DevArgs& d, int bz, hipStream_t s) {
  const bool amn = a.a_layout == MN_MAJOR, bmn = a.b_layout == MN_MAJOR;
  if (a.A3s) {
    if constexpr (BM == BN && !std::is_same<T, float>::value) return launch_inst<T, BM, BN, false, true, true, true>(d, bz, s);      // (validated: K-major A, MN-major B, a second segment)
    else { set_last_error("gemm: third / fourth K segments are built for the square tiles of the bf16 and the plane forms"); return ERR_UNSUPPORTED; }
  }
  if (a.A2) {
    if (!amn && bmn) return launch_inst<T, BM, BN, false, true, true>(d, bz, s);
    if (amn && bmn) return launch_inst<T, BM, BN, true, true, true>(d, bz, s);
    set_last_error("gemm: second K segment is built for B MN-major first segments only");
    return ERR_UNSUPPORTED;
  }
  if (!amn && !bmn) return launch_inst<T, BM, BN, false, false>(d, bz, s);
  if (!amn && bmn) return launch_inst<T, BM, BN, false, true>(d, bz, s);
  if (amn && !bmn) return launch_inst<T, BM, BN, true, false>(d, bz, s);
  return launch_inst<T, BM, BN, true, true>(d, bz, s);
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

size_t gemm_slab_bytes(const GemmArgs& a) {
  if (a.ksplit <= 1) return 0;
  return (size_t)a.ksplit * a.nb1 * a.nb2 * a.nb3 * (size_t)a.M * a.N * sizeof(float);
}

int launch_gemm(const GemmArgs& a, hipStream_t stream) {
  if (!a.A || !a.B || !a.C) { set_last_error("gemm: null operand"); return ERR_BAD_ARG; }
  if (a.M <= 0 || a.N <= 0 || a.K < 0 || a.nb1 <= 0 || a.nb2 <= 0) {
    set_last_error("gemm: bad extents M=%d N=%d K=%d batch=%dx%d", a.M, a.N, a.K, a.nb1, a.nb2);
    return ERR_BAD_ARG;
  }
  if (a.dtype != GEMM_F32 && a.dtype != GEMM_BF16) { set_last_error("gemm: dtype"); return ERR_UNSUPPORTED; }
  if (!(a.sCj == 1 || a.sCi == 1)) { set_last_error("gemm: C needs a unit stride"); return ERR_UNSUPPORTED; }
  const int esz = a.dtype == GEMM_BF16 ? 2 : 4;
  auto mult16 = [&](long elems) { return (elems * esz) % 16 == 0; };
  if (!aligned16(a.A) || !aligned16(a.B) || !mult16(a.lda) || !mult16(a.ldb) || !mult16(a.sA1) ||
      !mult16(a.sA2) || !mult16(a.sB1) || !mult16(a.sB2)) {
    set_last_error("gemm: operand alignment contract violated (A=%p B=%p lda=%ld ldb=%ld)", a.A, a.B, a.lda, a.ldb);
    return ERR_ALIGNMENT;
  }
  if (a.ksplit > 1 && !a.slabs) { set_last_error("gemm: split-K without slab workspace"); return ERR_WORKSPACE; }
  if (a.epi != GEMM_EPI_MULSUB && (a.row_scale != nullptr) != (a.D != nullptr)) { set_last_error("gemm: row_scale and D go together"); return ERR_BAD_ARG; }

  DevArgs d;
  d.A = (const char*)a.A; d.B = (const char*)a.B; d.C = (char*)a.C; d.D = (const char*)a.D;
  d.rs = a.row_scale; d.slabs = a.slabs;
  d.M = a.M; d.N = a.N; d.K = a.K; d.nb2 = a.nb2; d.nb3 = a.nb3 > 1 ? a.nb3 : 1; d.ksplit = a.ksplit > 1 ? a.ksplit : 1;
  d.nbatch = a.nb1 * a.nb2 * d.nb3;
  d.sA3 = a.sA3; d.sB3 = a.sB3; d.sC3 = a.sC3;
  if (d.nb3 > 1 && (a.D || a.A2 || a.epi != GEMM_EPI_NONE || a.Cx || a.st_rows || !mult16(a.sA3) || !mult16(a.sB3))) {
    set_last_error("gemm: the third batch level serves plain products only (16-byte aligned strides)"); return ERR_BAD_ARG;
  }
  d.lda = a.lda; d.ldb = a.ldb; d.sA1 = a.sA1; d.sA2 = a.sA2; d.sB1 = a.sB1; d.sB2 = a.sB2;
  d.sCi = a.sCi; d.sCj = a.sCj; d.sC1 = a.sC1; d.sC2 = a.sC2;
  d.sRS1 = a.sRS1; d.sRS2 = a.sRS2; d.sDi = a.sDi; d.sD1 = a.sD1; d.sD2 = a.sD2;
  d.alpha = a.alpha; d.accumulate = a.accumulate; d.out_bf16 = a.out_dtype == GEMM_BF16;
  d.A2 = (const char*)a.A2; d.B2 = (const char*)a.B2; d.K2 = a.K2; d.lda2 = a.lda2; d.ldb2 = a.ldb2;
  d.s2A1 = a.s2A1; d.s2A2 = a.s2A2; d.s2B1 = a.s2B1; d.s2B2 = a.s2B2;
  d.A3s = (const char*)a.A3s; d.B3s = (const char*)a.B3s; d.A4s = (const char*)a.A4s; d.B4s = (const char*)a.B4s; d.K3s = a.K3s; d.K4s = a.K4s;
  d.lda3s = a.lda3s; d.ldb3s = a.ldb3s; d.s3sA1 = a.s3sA1; d.s3sB1 = a.s3sB1; d.s3sB2 = a.s3sB2;
  d.lda4s = a.lda4s; d.ldb4s = a.ldb4s; d.s4sA1 = a.s4sA1; d.s4sB1 = a.s4sB1; d.s4sB2 = a.s4sB2;
  if (a.A3s) {
    const bool ok = a.A2 && a.B3s && a.K3s > 0 && a.a_layout == K_MAJOR && a.b_layout == MN_MAJOR && a.ksplit <= 1 && a.epi == GEMM_EPI_NONE && d.nb3 == 1 &&
                    (a.dtype == GEMM_BF16 || a.split3) && aligned16(a.A3s) && aligned16(a.B3s) && mult16(a.lda3s) && mult16(a.ldb3s) && mult16(a.s3sA1) &&
                    mult16(a.s3sB1) && mult16(a.s3sB2) &&
                    (a.K4s == 0 || (a.A4s && a.B4s && aligned16(a.A4s) && aligned16(a.B4s) && mult16(a.lda4s) && mult16(a.ldb4s) && mult16(a.s4sA1) && mult16(a.s4sB1) && mult16(a.s4sB2)));
    if (!ok) { set_last_error("gemm: third / fourth K segments need a K-major A / MN-major B product with a second segment, no split-K, aligned operands"); return ERR_BAD_ARG; }
  }
  d.epi = a.epi; d.row_part = a.row_part; d.row_lse = a.row_lse;
  if (a.epi != GEMM_EPI_NONE) {
    if (a.ksplit > 1 || a.sCj != 1 || (a.D != nullptr) != (a.epi == GEMM_EPI_MULSUB) || a.row_scale || a.accumulate || a.A2 || a.Cx || a.st_rows ||
        (a.tile != 0 && a.tile != 128) || (a.epi == GEMM_EPI_ROWSTATS && !a.row_part) || (a.epi != GEMM_EPI_ROWSTATS && !a.row_lse)) {
      set_last_error("gemm: softmax epilogue needs a plain row-major single-pass product on the 128 x 128 tile");
      return ERR_BAD_ARG;
    }
  }
  if ((a.A2 != nullptr) != (a.B2 != nullptr) || (a.A2 && (a.ksplit > 1 || !aligned16(a.A2) || !aligned16(a.B2) || !mult16(a.lda2) || !mult16(a.ldb2) ||
                                                  !mult16(a.s2A1) || !mult16(a.s2A2) || !mult16(a.s2B1) || !mult16(a.s2B2)))) {
    set_last_error("gemm: bad second K segment (pair of pointers, no split-K, 16-byte aligned strides)");
    return ERR_BAD_ARG;
  }
  {
    static const bool nostream = dev_env("AVMOE_GEMM_NOSTREAM") != nullptr;     // dev switch: A/B against the tiled engine
    if (!nostream && a.epi == GEMM_EPI_NONE && a.nb3 <= 1 && !a.A3s) {
      const int s = launch_gemm_stream(a, stream);
      if (s <= 0) return s;
      const int f = launch_gemm_frames(a, stream);           // a few rows per frame against one shared matrix (frame_gemm.hip)
      if (f <= 0) return f;
    }
  }
  if (a.Cx) { set_last_error("gemm: a split fp32 side output (Cx) is a feature of the streaming kernel only"); return ERR_UNSUPPORTED; }
  // Batch fold: per-sample row blocks of A (K-major, regularly spaced) against ONE shared B are the rows of a single tall
  // GEMM -- used when the per-sample M would leave a quarter or more of its tile rows empty (65 rows on a 128-row tile, 8
  // latent rows on a 64-row tile).  Rows of the gaps between the samples' blocks are computed and not stored.
  d.fold_rps = d.fold_valid = 0;
  {
    static const bool nofold = dev_env("AVMOE_GEMM_NOFOLD") != nullptr;      // dev switch
    if (!nofold && a.epi == GEMM_EPI_NONE && a.nb1 > 1 && a.nb2 == 1 && d.nb3 == 1 && a.a_layout == K_MAJOR && a.sB1 == 0 && d.ksplit == 1 && !a.A2 && !a.D && a.lda > 0 &&
        a.sA1 > 0 && a.sA1 % a.lda == 0) {
      const long rps = a.sA1 / a.lda, rows = (long)(a.nb1 - 1) * rps + a.M;
      const int t0 = a.tile ? a.tile : ((a.M > 64 && a.N > 64) ? 128 : 64);      // tile of the unfolded launch
      const bool wasteful = round_up(a.M, t0) * 4 >= (long)a.M * 5;             // >= 25 % of its tile rows are padding (measured:
                                                                                  // folding full tiles of the same size gains nothing)
      // ... or the samples' blocks are too short for the 256 x 256 tile while the tall product is not (cfg-5: 384 latent rows per frame
      // against the 4096 x 3138 remap matrix, 40 frames)
      const bool for_big = a.dtype == GEMM_BF16 && a.tile == 0 && a.M < 1024 && rows >= 2048 && a.N >= GEMM_BIG_MIN && a.K >= 256;
      if ((wasteful || for_big) && rps >= a.M && (rps - a.M) * 8 <= a.M && rows < (1L << 30)) {
        d.fold_rps = (int)rps; d.fold_valid = a.M; d.M = (int)rows; d.nbatch = 1; d.sA1 = 0;
      }
    }
  }
  const int osz = d.out_bf16 ? 2 : 4;
  const int vecb = 4 * osz;    // bytes of a 4-element output vector
  d.vec_c = (a.sCj == 1) && (((uintptr_t)a.C) % vecb == 0) && ((a.sCi * osz) % vecb == 0) &&
            ((a.sC1 * osz) % vecb == 0) && ((a.sC2 * osz) % vecb == 0);
  d.vec_d = a.D && (((uintptr_t)a.D) % (4 * esz) == 0) && ((a.sDi * esz) % (4 * esz) == 0) &&
            ((a.sD1 * esz) % (4 * esz) == 0) && ((a.sD2 * esz) % (4 * esz) == 0);
  const int bk = 256 / esz;          // (the longest K step of any instance)
  d.kper = d.ksplit > 1 ? (int)round_up(cdiv(a.K, d.ksplit), bk) : (a.K > 0 ? (int)round_up(a.K, bk) : bk);

  int tile = a.tile;
  if (a.epi != GEMM_EPI_NONE) tile = 128;
  else if (tile == 0 || d.fold_rps) tile = (d.M > 64 && a.N > 64) ? 128 : ((d.M <= 32 && a.N <= 32) ? 32 : 64);
  const int bz = d.nbatch * d.ksplit;
  if (bz > 65535) { set_last_error("gemm: batch*ksplit=%d exceeds grid.y", bz); return ERR_UNSUPPORTED; }
  // Small products are bound by the latency of their K loop, not by throughput: a tile size that leaves most of the 256 CUs without
  // a block is replaced by the next smaller one (same K order per output element: bit-identical results).  Measured at cfg-1 (fp32,
  // 20 frames): 1300 x 384 x 512 on 33 blocks of 128 x 128 takes 52 us.
  if (a.tile == 0 && a.epi == GEMM_EPI_NONE) {
    auto nblk = [&](int t) { return (long)cdiv(d.M, t) * cdiv(a.N, t) * bz; };
    if (tile == 128 && nblk(128) < GEMM_MIN_BLOCKS) tile = 64;
    if (tile == 64 && nblk(64) < GEMM_MIN_BLOCKS) tile = 32;
  }
  int st;
  // the 256 x 256 tile for large plain bf16 products (enough tiles of it to fill the chip)
  const bool big = a.dtype == GEMM_BF16 && a.tile == 0 && tile == 128 && a.epi == GEMM_EPI_NONE && !a.A2 && !a.D && d.M >= GEMM_BIG_MIN && a.N >= GEMM_BIG_MIN &&
                   a.K >= 256 && (long)cdiv(d.M, 256) * cdiv(a.N, 256) * bz >= 160;
  if (big) {
    d.tiles_n = cdiv(a.N, 256);
    const bool amn = a.a_layout == MN_MAJOR, bmn = a.b_layout == MN_MAJOR;
    st = amn ? (bmn ? launch_big_inst<true, true>(d, bz, stream) : launch_big_inst<true, false>(d, bz, stream))
             : (bmn ? launch_big_inst<false, true>(d, bz, stream) : launch_big_inst<false, false>(d, bz, stream));
  } else if (tile == 128 && a.tile == 0 && a.dtype == GEMM_F32 && a.split3 == 1 && a.epi == GEMM_EPI_NONE && !a.A2 && a.b_layout == MN_MAJOR &&
             round_up(a.N, 160) * 23 <= round_up(a.N, 128) * 20) {
    // 128 x 160 tile (round 6): fp32 three-plane products -- matrix-pipe-bound, six plane products per K step -- whose N leaves the last
    // 128-column tile mostly empty: the 128 + 3 E columns of dApost / dBpost (N = 140: 160 instead of 256 columns computed; fp32 cfg-2,
    // audio tokens: dApost 884 -> 761 us, dBpost 876 -> 630).  Same K order per output element: the same bits as every other tile.
    d.tiles_n = cdiv(a.N, 160);
    st = a.a_layout == MN_MAJOR ? launch_inst<f32s3, 128, 160, true, true>(d, bz, stream) : launch_inst<f32s3, 128, 160, false, true>(d, bz, stream);
  } else if (tile == 128 && a.tile == 0 && a.dtype == GEMM_F32 && a.split3 == 1 && a.epi == GEMM_EPI_NONE && !a.A2 && a.a_layout == K_MAJOR && !d.fold_rps &&
             d.ksplit == 1 && round_up(d.M, 96) * 23 <= round_up(d.M, 128) * 20) {
    // 96 x 128 tile: the same for M -- the 65 rows ([latent tokens ; wbar]) of the per-frame hop-1 products against Y fill half of a 128-row tile
    d.tiles_n = cdiv(a.N, 128);
    st = a.b_layout == MN_MAJOR ? launch_inst<f32s3, 96, 128, false, true>(d, bz, stream) : launch_inst<f32s3, 96, 128, false, false>(d, bz, stream);
  } else if (tile == 128) {
    d.tiles_n = cdiv(a.N, 128);
    st = a.dtype == GEMM_BF16 ? launch_layout<__bf16, 128, 128>(a, d, bz, stream)
                              : (a.split3 == 2 ? launch_layout<f32s2, 128, 128>(a, d, bz, stream) : a.split3 ? launch_layout<f32s3, 128, 128>(a, d, bz, stream) : launch_layout<float, 128, 128>(a, d, bz, stream));
  } else if (tile == 64) {
    d.tiles_n = cdiv(a.N, 64);
    st = a.dtype == GEMM_BF16 ? launch_layout<__bf16, 64, 64>(a, d, bz, stream)
                              : (a.split3 == 2 ? launch_layout<f32s2, 64, 64>(a, d, bz, stream) : a.split3 ? launch_layout<f32s3, 64, 64>(a, d, bz, stream) : launch_layout<float, 64, 64>(a, d, bz, stream));
  } else if (tile == 32) {                                   // small per-batch problems (K x K latent matrices, S x S frame attention)
    d.tiles_n = cdiv(a.N, 32);
    st = a.dtype == GEMM_BF16 ? launch_layout<__bf16, 32, 32>(a, d, bz, stream)
                              : (a.split3 == 2 ? launch_layout<f32s2, 32, 32>(a, d, bz, stream) : a.split3 ? launch_layout<f32s3, 32, 32>(a, d, bz, stream) : launch_layout<float, 32, 32>(a, d, bz, stream));      // (every tile: the result must not depend on the tile the block count picks)
  } else {
    set_last_error("gemm: tile %d not built", tile);
    return ERR_UNSUPPORTED;
  }
  if (st != OK) return st;
  if (d.ksplit > 1 && !a.keep_slabs) {
    const long total = (long)d.nbatch * a.M * a.N;
    if (total >= (1L << 31)) { set_last_error("gemm: split-K result too large"); return ERR_UNSUPPORTED; }
    const long nvec = (a.N % 4 == 0) ? total / 4 : total;
    int lgP = 0;                                             // lanes per output vector: enough threads for ~2 waves per SIMD
    while (lgP < 4 && (2 << lgP) <= d.ksplit && (nvec << lgP) < 131072) ++lgP;
    const long nwv = (nvec + (64 >> lgP) - 1) / (64 >> lgP);
    const int blocks = (int)std::min<long>((nwv + 3) / 4, 4096);
    ProfScope ps("gemm_splitk_reduce", total, (double)total * 4.0 * (d.ksplit + 1), 0.0, stream);      // (tagged with the result's element count)
    if (a.dtype == GEMM_BF16) hipLaunchKernelGGL(gemm_splitk_reduce<__bf16>, dim3(blocks), dim3(256), 0, stream, d, lgP);
    else hipLaunchKernelGGL(gemm_splitk_reduce<float>, dim3(blocks), dim3(256), 0, stream, d, lgP);
    AVMOE_CHECK_LAUNCH("gemm_splitk_reduce");
  }
  return OK;
}

__global__ void __launch_bounds__(256) kk_row_lse(const float* __restrict__ part, long rows, int tiles, float* __restrict__ lse) {
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
    const float* p = part + r * tiles * 2;
    float mx = -INFINITY;
    for (int t = 0; t < tiles; ++t) mx = fmaxf(mx, p[2 * t]);
    float sum = 0.f;
    for (int t = 0; t < tiles; ++t) sum += p[2 * t + 1] * __expf(p[2 * t] - mx);
    lse[r] = mx + __logf(sum);
  }
}
int gemm_row_lse(const float* row_part, long rows, int tiles, float* lse, hipStream_t stream) {
  if (rows <= 0) return OK;
  ProfScope ps("gemm_row_lse", 0.0, 0.0, stream);
  hipLaunchKernelGGL(kk_row_lse, dim3((unsigned)std::min<long>(cdiv(rows, 256), 4096)), dim3(256), 0, stream, row_part, rows, tiles, lse);
  AVMOE_CHECK_LAUNCH("gemm_row_lse");
  return OK;
}

int launch_gemm_tokpair(const TokPairArgs& a, hipStream_t stream) {
  static const bool off = dev_env("AVMOE_NO_TOKPAIR") != nullptr;      // dev switch: the two engine GEMMs instead
  if (off || a.M1 <= 0 || a.M1 > 128 || a.M2 <= 0 || a.M2 > 64 ||          // (the 128-row second accumulator spills: not served)
       a.M1 % 8 || a.M2 % 8 || a.Cg % 8 || a.lda1 % 8 || a.lda2 % 8 || a.ldx % 8 ||
      a.sA1g % 8 || ((uintptr_t)a.A1 % 16) || ((uintptr_t)a.A2 % 16) || ((uintptr_t)a.X % 16) || a.N < 64 || !a.slabs)
    return 1;
  const int tiles_n = cdiv(a.Cg, 128), nbx = a.g * tiles_n;
  const size_t per_slab = (size_t)a.g * a.M1 * a.Cg;
  int nchunks = std::min<long>(a.S, std::max<long>(1, cdiv(512, nbx)));
  nchunks = (int)std::min<size_t>(nchunks, a.slab_cap / per_slab);
  if (nchunks < 1) return 1;
  const int F = cdiv(a.S, nchunks);
  nchunks = cdiv(a.S, F);
  if (nchunks > 65535) return 1;
  TokPairDev p;
  p.A1 = (const char*)a.A1; p.A2 = (const char*)a.A2; p.X = (const char*)a.X; p.C2 = a.C2; p.slabs = a.slabs;
  p.lda1 = a.lda1; p.lda2 = a.lda2; p.ldx = a.ldx; p.sA1g = a.sA1g;
  p.M1 = a.M1; p.M2 = a.M2; p.S = a.S; p.N = a.N; p.g = a.g; p.Cg = a.Cg; p.tiles_n = tiles_n; p.F = F;
  {
    static const int tk = [] { const char* e = dev_env("AVMOE_TOKPAIR_TK"); return e && *e ? atoi(e) : 256; }();      // dev switch (multiple of 64)
    p.TK = tk > 0 ? (tk + 63) / 64 * 64 : a.N;
  }
  constexpr int LDS = 2 * 2 * 64 * (128 * 2 + 16);          // two stages of the (128, 128) MN-major / MN-major segment
  const double bytes = ((double)a.S * a.N) * ((double)a.g * a.Cg + (double)a.g * a.M1 + a.M2) * 2.0 + (double)a.S * a.M2 * a.g * a.Cg * 4.0;
  const double flops = 2.0 * (double)a.S * a.N * (double)a.g * a.Cg * ((double)a.M1 + a.M2);
  {
    ProfScope ps("gemm_tokpair", (long)a.S * a.N, bytes, flops, stream);
    static bool attr64 = false;
    if (!attr64) { if (hipFuncSetAttribute((const void*)gemm_tokpair_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return ERR_LAUNCH; attr64 = true; }
    hipLaunchKernelGGL(gemm_tokpair_kernel<64>, dim3(nbx, nchunks), dim3(256), LDS, stream, p);
    AVMOE_CHECK_LAUNCH("gemm_tokpair");
  }
  // C1 = the chunks' partial sums, in chunk order (the split-K reduce of the engine)
  DevArgs d{};
  d.C = (char*)a.C1; d.slabs = a.slabs; d.M = a.M1; d.N = a.Cg; d.nb2 = a.g; d.nb3 = 1; d.sA3 = d.sB3 = d.sC3 = 0; d.nbatch = a.g; d.ksplit = nchunks;
  d.sCi = a.Cg; d.sCj = 1; d.sC1 = 0; d.sC2 = (long)a.M1 * a.Cg; d.alpha = 1.f; d.accumulate = 0; d.out_bf16 = 0;
  const long total = (long)a.g * a.M1 * a.Cg, nvec = (a.Cg % 4 == 0) ? total / 4 : total;
  int lgP = 0;
  while (lgP < 4 && (2 << lgP) <= nchunks && (nvec << lgP) < 131072) ++lgP;
  const long nwv = (nvec + (64 >> lgP) - 1) / (64 >> lgP);
  ProfScope ps("gemm_splitk_reduce", total, (double)total * 4.0 * (nchunks + 1), 0.0, stream);
  hipLaunchKernelGGL(gemm_splitk_reduce<__bf16>, dim3((int)std::min<long>((nwv + 3) / 4, 4096)), dim3(256), 0, stream, d, lgP);
  AVMOE_CHECK_LAUNCH("gemm_splitk_reduce");
  return OK;
}

}  // namespace avmoe
