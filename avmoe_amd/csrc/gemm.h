// Generic strided / batched MFMA GEMM engine for gfx950 (MI355X).
//
// Every wide contraction of the adapter path -- remap, latent-token attention hops, grouped
// down/up projections, their input- and weight-gradient products -- is expressed as one call of
// this engine (see DESIGN.md "bottleneck-space algebra"), so the full-width token tensors are only
// ever touched by MFMA tiles.
//
//   C[b][i][j] (+)= alpha * sum_k A[b][i][k] * B[b][j][k]   (+ row_scale[b][i] * D[b][i][j])
//
// Operand layouts (per operand):
//   K_MAJOR  : k is the contiguous index  (element (i,k) at  i*ld + k)
//   MN_MAJOR : i is the contiguous index  (element (i,k) at  k*ld + i)   -- read through LDS with
//              ds_read_b64_tr_b16 (bf16) / strided ds_read_b32 (f32)
// C is addressed with explicit (sCi, sCj) strides; one of them must be 1.
//
// Contract for callers (checked in gemm.hip::validate):
//   * base pointers 16-byte aligned, ld * sizeof(T) and batch strides * sizeof(T) multiples of 16
//   * K_MAJOR operands: every row readable (finite) up to roundup(K, 16/sizeof(T)) elements; at
//     least one of the two operands is ZERO in that padding
//   * MN_MAJOR operands: every k-row readable up to roundup(M or N, 16/sizeof(T)) elements
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace avmoe {

enum GemmDtype { GEMM_F32 = 0, GEMM_BF16 = 1 };
enum GemmLayout { K_MAJOR = 0, MN_MAJOR = 1 };

struct GemmArgs {
  const void* A = nullptr;
  const void* B = nullptr;
  void* C = nullptr;
  int M = 0, N = 0, K = 0;
  int nb1 = 1, nb2 = 1;                 // batch = nb1 * nb2 ; b = b1 * nb2 + b2
  int nb3 = 1;                          // optional third batch level (tiled engine only, no D term): batch = nb1 * nb3 * nb2,
  long sA3 = 0, sB3 = 0, sC3 = 0;       // b = (b1 * nb3 + b3) * nb2 + b2 ; strides in elements
  int dtype = GEMM_F32;                 // A, B (and D) element type
  int out_dtype = GEMM_F32;             // C element type
  int a_layout = K_MAJOR, b_layout = K_MAJOR;
  long lda = 0, ldb = 0;
  long sA1 = 0, sA2 = 0, sB1 = 0, sB2 = 0;   // batch strides, elements
  long sCi = 0, sCj = 1, sC1 = 0, sC2 = 0;   // C strides, elements
  float alpha = 1.f;
  int accumulate = 0;                   // C += ...
  // optional epilogue term  row_scale[b][i] * D[b][i][j]   (D: dtype `dtype`, j contiguous)
  const float* row_scale = nullptr;
  long sRS1 = 0, sRS2 = 0;
  const void* D = nullptr;
  long sDi = 0, sD1 = 0, sD2 = 0;
  // split-K: partial sums go to fp32 slabs [ksplit][batch][M][N] in `slabs`, then a reduce pass
  int ksplit = 1;
  float* slabs = nullptr;
  int keep_slabs = 0;                   // split-K: leave the partial sums in `slabs` (no reduce pass): the caller's consumer adds them
  // fp32 operands only: the product on the bf16 matrix pipe in THREE-PLANE form -- every value as three bf16 planes (x = p0 + p1 + p2 up to
  // 2^-24 |x|), the six plane products of order <= 2, fp32 accumulation: 5.8e-9 relative per product (the fp32 rounding of the sum itself is
  // 2 - 4e-7) at 6 x 16 instead of 8 x 32 matrix-pipe cycles per 16 x 16 x 32 block.  The site's BACKWARD sets it; the forward keeps
  // v_mfma_f32_16x16x4_f32 (same values to the last bit as rounds 1 - 3: which ReLU units sit on which side of zero does not move).
  // Non-finite operands: an Inf splits into (Inf, NaN, NaN) -- the residual planes are Inf - Inf -- so a product that exact fp32 arithmetic
  // would return as Inf comes out NaN; finite inputs only (as everywhere on this path: workspaces are NaN-poisoned in the tests).
  // 2 = TWO planes (split once at the LDS store), the three plane products of order <= 1: 2^-16 relative per product, half the matrix-pipe work --
  // available through avmoe_gemm (fp32_planes = 2); the site calls do not use it (moe_run.h: AVMOE_BWD_PLANES has the measurement and the reason).
  int split3 = 0;
  int tile = 0;                         // 0 = auto, 64 or 128 = force block tile
  // optional second K segment, accumulated into the same tile before the epilogue:
  //   C += alpha * A2[b][i][k2] * B2[b][j][k2]   with A2 K_MAJOR (lda2), B2 MN_MAJOR (ldb2), own batch strides.
  // Fuses e.g. dX = dZx Wt + [dL2|dsx|1] [T;1;dm1/N] so the (NT, C) result is written once.
  const void* A2 = nullptr;
  const void* B2 = nullptr;
  int K2 = 0;
  long lda2 = 0, ldb2 = 0, s2A1 = 0, s2A2 = 0, s2B1 = 0, s2B2 = 0;
  // optional THIRD and FOURTH K segments (tiled engine, A K_MAJOR / B MN_MAJOR first segment with a second segment; both or none):
  //   C += alpha * ( A3[b][i][k3] B3s[b][j][k3]  +  A4[b][i][k4] B4s[b][j][k4] )      A3 MN_MAJOR, A4 K_MAJOR, B3s / B4s MN_MAJOR
  // -- the OTHER adapter site's dY = [Bm ; wbar]^T dV + dR^T Q folded into this site's dX product (moe_backward_dx_dy): the token gradient is
  // written once instead of written by one site and read back + added by the other.  K4 may be 0 (no cross-modal chain).
  const void* A3s = nullptr;
  const void* B3s = nullptr;
  const void* A4s = nullptr;
  const void* B4s = nullptr;
  int K3s = 0, K4s = 0;
  long lda3s = 0, ldb3s = 0, s3sA1 = 0, s3sB1 = 0, s3sB2 = 0;
  long lda4s = 0, ldb4s = 0, s4sA1 = 0, s4sB1 = 0, s4sB2 = 0;
  // optional split output (streaming kernel only; launch_gemm_stream returns 1 when it cannot honour it): columns >= nsplit
  // (a multiple of 32) go, in fp32, to Cx[b2][i][j - nsplit] (row stride ldcx, group stride sCx2) instead of C -- lets the
  // wide part of a product be stored in bf16 while a few columns that feed long fp32 sums keep full precision.
  float* Cx = nullptr;
  int nsplit = 0;
  long ldcx = 0, sCx2 = 0;
  // optional statistics of A as a side product (streaming kernel only, K_MAJOR bf16 A, no second segment; launch_gemm_stream returns
  // 1 when it cannot honour them): per row the sum and the sum of squares over THIS group's K columns,
  //   st_rows[(2 b2) * st_ntot + b1 * M + i] = sum_k A ,  st_rows[(2 b2 + 1) * st_ntot + b1 * M + i] = sum_k A^2 ,
  // and per row tile the column sums  st_cols[(b1 * tiles + t) * (nb2 * K) + b2 * K + k] = sum_{i in tile t} A[b1][i][k]
  // (tiles = row tiles per sample, returned through st_tiles) -- the LayerNorm sums and the router's token means of X without a
  // separate pass over X.
  // optional softmax epilogues of the tiled engine (128 x 128 tile, no split-K, row-major C):
  //   epi = GEMM_EPI_ROWSTATS : nothing is stored to C; per row and column tile the running softmax statistics of alpha * (A B^T)
  //         go to row_part[((b * M + i) * tiles + t) * 2 + {0, 1}] = (max, sum of exp(v - max)), tiles = ceil(N / 128)
  //   epi = GEMM_EPI_EXP      : C = exp(alpha * (A B^T) - row_lse[b * M + i])
  //   epi = GEMM_EPI_MULSUB   : C = D * (alpha * (A B^T) - row_lse[b * M + i])      (D without row_scale: the softmax backward
  //         dS = att * (d att - rowdot) with d att = the product, never stored)
  // -- a row softmax without the scores ever leaving the chip (gemm_row_lse combines the parts).
  int epi = 0;
  float* row_part = nullptr;
  const float* row_lse = nullptr;
  float* st_rows = nullptr;
  float* st_cols = nullptr;
  long st_ntot = 0;
  int* st_tiles = nullptr;
  // optional EXTRA output columns against a per-sample matrix, from the same pass over A (streaming kernel with statistics only;
  // launch_gemm_stream returns 1 when it cannot honour them):
  //   C3[b1][b2][i][j] = sum_k A[b1][b2][i][k] * B3[b1][b2][j][k]      j < N3 <= 64, B3 K_MAJOR (ldb3), fp32 result (row stride ldc3)
  // -- the hop-2 logits of a site, X[s] T[s]^T, as per-group partial sums out of the down projection's pass over X (the caller adds
  // the groups): X is read once instead of twice.
  const void* B3 = nullptr;
  int N3 = 0;
  long ldb3 = 0, s3B1 = 0, s3B2 = 0;
  float* C3 = nullptr;
  long ldc3 = 0, s3C1 = 0, s3C2 = 0;
};

// Returns 0 on success, negative avmoe status otherwise (message through set_last_error).
int launch_gemm(const GemmArgs& args, hipStream_t stream);

// Streaming (B-stationary, persistent) kernel for token-streaming shapes; 0 = launched, 1 = shape not covered, < 0 error.
// launch_gemm tries it first.
enum { GEMM_EPI_NONE = 0, GEMM_EPI_ROWSTATS = 1, GEMM_EPI_EXP = 2, GEMM_EPI_MULSUB = 3 };
// lse[r] = log sum_j exp(v[r][j]) from the (max, sum) parts of GEMM_EPI_ROWSTATS (rows = batch * M, tiles = ceil(N / 128))
int gemm_row_lse(const float* row_part, long rows, int tiles, float* lse, hipStream_t stream);
int launch_gemm_stream(const GemmArgs& args, hipStream_t stream);
// Per-frame products of <= 64 rows against ONE shared K-major matrix (frame_gemm.hip); 0 = launched, 1 = shape not covered, < 0 error.
int launch_gemm_frames(const GemmArgs& args, hipStream_t stream);

// Two token contractions against the same (S, N, g * Cg) bf16 tensor X in ONE pass over it:
//   C1[gi][i][j]  = sum over ALL tokens t   A1[t][gi * sA1g + i] * X[t][gi * Cg + j]      i < M1 <= 128   (split over frame chunks: slabs + reduce)
//   C2[s][i][c]   = sum over the tokens of frame s   A2[t][i] * X[t][c]                   i < M2 <= 128   (one result per frame)
// (the backward's dWt = dZx^T X and dT[s] = dL2[s]^T X[s]).  Both results fp32.  Returns 1 when the shape is not served.
struct TokPairArgs {
  const void* A1 = nullptr; long lda1 = 0; int M1 = 0; long sA1g = 0;
  const void* A2 = nullptr; long lda2 = 0; int M2 = 0;
  const void* X = nullptr; long ldx = 0;
  int S = 0, N = 0, g = 1, Cg = 0;
  float* C1 = nullptr; float* C2 = nullptr;
  float* slabs = nullptr; size_t slab_cap = 0;          // floats
};
int launch_gemm_tokpair(const TokPairArgs& args, hipStream_t stream);
// Whether the streaming kernel serves a per-sample bf16 product  (M rows per sample, nb1 samples) x (K per group, lda) -> N columns per group
// WITH the statistics of A (GemmArgs::st_rows / st_cols): what a plan asks before it drops the separate statistics pass.
bool gemm_stream_stats_ok(int M, int nb1, int N, int K, long lda, long ldc);

// Bytes of fp32 slab workspace a split-K launch of `args` needs (0 when ksplit <= 1).
size_t gemm_slab_bytes(const GemmArgs& args);

}  // namespace avmoe
