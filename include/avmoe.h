/* avmoe.h -- C ABI of the MI355X-native AVMoE adapter hot path (libavmoe_hip.so).
 *
 * The reference (yingchengy/AVMOE) has no FFI: its boundary for this path is the Python nn.Module API
 * of MoEAdapter / ExpertAdapter (AVMOE/AVE/nets/net_trans_v3.py:296-487 and the four task copies).
 * This header is what a binding for that path binds instead: plain pointers, sizes, strides and a
 * hipStream_t (passed as void*), int status returns, no exceptions, no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless said otherwise; the library never allocates or frees
 *     caller memory; scratch comes from a caller-provided workspace
 *   - all entry points are asynchronous on `stream`, re-entrant across streams, and keep no global
 *     mutable state besides a thread-local error string
 *   - return 0 on success, negative on error (avmoe_last_error() has the message)
 */
#ifndef AVMOE_H_
#define AVMOE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVMOE_ABI_VERSION 1

enum { AVMOE_OK = 0, AVMOE_ERR_BAD_ARG = -1, AVMOE_ERR_UNSUPPORTED = -2, AVMOE_ERR_ALIGNMENT = -3,
       AVMOE_ERR_WORKSPACE = -4, AVMOE_ERR_LAUNCH = -5 };
enum { AVMOE_F32 = 0, AVMOE_BF16 = 1 };
enum { AVMOE_K_MAJOR = 0, AVMOE_MN_MAJOR = 1 };

int avmoe_abi_version(void);
const char* avmoe_last_error(void);      /* thread-local, valid until the next failing call */

/* ---- sub-op: the strided / batched MFMA GEMM every wide contraction of the path runs on --------
 * C[b][i][j] (+)= alpha * sum_k A[b][i][k] * B[b][j][k]  (+ row_scale[b][i] * D[b][i][j])
 * Replaces the reference's conv2d(1x1) / linear / bmm calls (net_trans_v3.py:380-388,395,401,469-470).
 * Layout / alignment contract: avmoe_amd/csrc/gemm.h. */
typedef struct avmoe_gemm_desc {
  int32_t M, N, K;
  int32_t nb1, nb2;                 /* batch = nb1 * nb2 */
  int32_t dtype, out_dtype;         /* AVMOE_F32 | AVMOE_BF16 */
  int32_t a_layout, b_layout;       /* AVMOE_K_MAJOR | AVMOE_MN_MAJOR */
  int32_t accumulate, ksplit, tile;
  float alpha;
  int64_t lda, ldb, sA1, sA2, sB1, sB2;
  int64_t sCi, sCj, sC1, sC2;
  int64_t sRS1, sRS2, sDi, sD1, sD2;
} avmoe_gemm_desc;

size_t avmoe_gemm_workspace_bytes(const avmoe_gemm_desc* desc);
int avmoe_gemm(const avmoe_gemm_desc* desc, const void* A, const void* B, void* C,
               const float* row_scale, const void* D, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AVMOE_H_ */
