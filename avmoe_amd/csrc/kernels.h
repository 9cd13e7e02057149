// Launch wrappers of the bottleneck-space kernels (everything that is not a GEMM).
// Naming follows oracle/algebra_ref.py; buffer layouts are documented in moe_plan.h.
#pragma once
#include "moe_plan.h"
#include <hip/hip_runtime.h>

namespace avmoe {

struct P16 { const float* p[MAX_E]; };
struct W16 { float* p[MAX_E]; };
struct N16 { int64_t* p[MAX_E]; };       // num_batches_tracked counters (avmoe_expert_ptrs::bn1_nbt / bn2_nbt), NULL = none

// All wrappers return 0 or a negative status; `bf16` selects the operand type T.
// ---- forward: weight preparation -------------------------------------------------------------
// remap operands, folded expert weights, stacked latent tokens, the constant rows of Text: one launch
int k_prep_all(const Plan& pl, char* saved, const avmoe_moe_ptrs& prm, hipStream_t st);
// merged groups (Dims::mg): block-diagonal dense copies of the grouped down_sampler / up_sampler weights, and the way back for
// their gradients (only the diagonal blocks are parameters)
int k_merge_expand(const Plan& pl, char* saved, const avmoe_moe_ptrs& prm, hipStream_t st);
int k_merge_gather(const Plan& pl, char* scratch, const avmoe_moe_ptrs& grads, hipStream_t st);
avmoe_moe_ptrs merged_params(const Plan& pl, const avmoe_moe_ptrs& prm, char* saved);
avmoe_moe_ptrs merged_grads(const Plan& pl, const avmoe_moe_ptrs& grads, char* scratch);
// ---- forward: token statistics ---------------------------------------------------------------
int k_rowstats(int bf16, const void* X, long rows, int C, float* out_sum_sq /* [2][rows] */, hipStream_t st);
int k_xstats(const Plan& pl, const void* X, char* saved, char* scratch, hipStream_t st);   // row sums + column means of X
int k_sum_parts(const float* parts, int nparts, long n, float* out, hipStream_t st);       // out[i] = sum_p parts[p * n + i]
// ... and, in the same launch, rin[s][c] = scale * sum over the `tiles` column partials xpart[s][tile][c] (fused X statistics)
int k_xstats_fin(const float* parts, int nparts, long n, float* out, const float* xpart, int tiles, int C, int S, float* rin, long rin_ld,
                 float scale, hipStream_t st);
int k_colmean(int bf16, const void* X, int S, int N, int C, float* out, long out_ld, hipStream_t st);
// ---- forward: hop 1 ---------------------------------------------------------------------------
// qr / qb (after Q = T0 Wf) into qrqb and the extension columns of Rext ; the ybar rows of Rext / BmX
int k_qrqb_fill(const Plan& pl, char* saved, const float* fc_b, hipStream_t st);
int k_softmax_rows(int bf16_out, const float* in, long rows, int n, int ld_in, void* out, int ld_out, int grp, int valid,
                   int slot, int kvalid, hipStream_t st);
int k_finish_T(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, int src /*0 y, 1 x*/,
               hipStream_t st);
// ---- forward: router --------------------------------------------------------------------------
int k_onehot_noise(float* noise, int S, int E, int hot, float value, hipStream_t st);      // sub-ops of the C ABI (fwd_kernels.hip)
int k_add_bias(bool bf16, void* Z, long rows, int cols, int period, const float* rowb, const float* colb, hipStream_t st);
int k_router(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const float* noise, float* probs_out,
             int64_t* idx_out, float* lb_out, hipStream_t st);
// ---- forward: per token -----------------------------------------------------------------------
int k_pre_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);
int k_bn1_finalize(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);
int k_mid(const Plan& pl, char* saved, char* scratch, hipStream_t st);
int k_post_prep(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);
int k_post_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);

// ---- backward ---------------------------------------------------------------------------------
int k_post_small_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm,
                     const avmoe_moe_ptrs& grads, hipStream_t st, int dap16 = 0);   // dap16: dApost = [T columns | fp32 dApx]
int k_post_prep_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm,
                    const avmoe_moe_ptrs& grads, hipStream_t st);
int k_mid_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads,
              hipStream_t st);
int k_pre_small_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm,
                    const avmoe_moe_ptrs& grads, hipStream_t st);
int k_post_small_bwd_finalize(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm,
                              const avmoe_moe_ptrs& grads, hipStream_t st);
int k_mid_bwd_finalize(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads,
                       hipStream_t st);
int k_pre_small_bwd_finalize(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm,
                             const avmoe_moe_ptrs& grads, hipStream_t st);
int k_router_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads,
                 const float* lb_grad, hipStream_t st);
int k_softmax_rows_bwd(int bf16, const void* a, const float* da, long rows, int n, int ld, void* out_dl, void* out_t, int grp,
                       int ldT, hipStream_t st);
int k_finish_dT(const Plan& pl, char* saved, char* scratch, hipStream_t st);
int k_prep_dBm(const Plan& pl, char* scratch, hipStream_t st);
int k_dqrqb(const Plan& pl, char* scratch, const float* conv_b, hipStream_t st);
int k_hop1_finalize(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads,
                    hipStream_t st);
int k_down_bwd(const Plan& pl, char* scratch, const avmoe_moe_ptrs& prm, const avmoe_moe_ptrs& grads, hipStream_t st);

int k_xrstats(const Plan& pl, const void* X, char* saved, int slot, hipStream_t st);
int k_nxn_axpy(const Plan& pl, const void* X, char* saved, char* scratch, void* dX, int slot, int replaces, hipStream_t st);
// att = softmax_rows(X X^T) of `frames` frames in one kernel (nxn_att.hip); have_lse: the row log-sum-exp is given (only the exp sweep runs)
bool nxn_att_ok(int bf16, int N, int C, int Np);
int k_nxn_att(const void* X, int frames, int N, int C, int Np, float* lse, void* att, int have_lse, hipStream_t st);
int k_nxn_att_bwd(const void* X, const void* dxr, int frames, int N, int C, int Np, const float* rowdot, const void* att, void* dS, hipStream_t st);
// the same backward without the softmax in memory (nxn_att.hip, round 4): att re-formed from the kept row log-sum-exp in the accumulators
int k_nxn_y(const void* X, const void* dxr, int frames, int N, int C, int Np, const float* lse, float* rowdot, void* dX, hipStream_t st);   // rowdot = X . (att dxr) ; dX += att dxr
int k_nxn_xr(const void* X, int frames, int N, int C, int Np, const float* lse, void* xr, hipStream_t st);      // forward: xr = att^T X (lse from k_nxn_att with att == nullptr)
int k_nxn_dx(int key, const void* X, const void* dxr, int frames, int N, int C, int Np, const float* lse, const float* rowdot, void* dX, hipStream_t st);   // dX += dS X (key = 0) / dS^T X (key = 1), dS in the accumulators
// dApost = dOut Bpost and dBpost = dOut^T Apost from one pass over dOut (dpost_pair.hip); 1 = shape not served
int k_dpost_pair(const void* dOut, long ldo, const void* Bpost, long ldb, long sBg, const void* Apost, long lda, void* dAp, long ldc, float* dApx, long ldx, int XW,
                 float* dBp, int ntok, int G, int Cg, int nmain, int KP, int KPp, float* slabs, size_t slab_cap, hipStream_t st);
// dWt = dZx^T X and dT[s] = dL2[s]^T X[s] as one streaming pass (tok_pair2.hip); 1 = shape not served
int k_tok_pair2(const void* X, long ldx, const void* dZx, long ldz, const void* dL2, long ldl, int S, int N, int G, int Cg, int M1, int KL,
                float* dWt, float* dT, float* slabs, size_t slab_cap, hipStream_t st);
// dX = dZx Wt + dL2x T[s] + rs X in the eight-wave direct-load form (dx_stream2.hip); 1 = shape not served
int k_dx_stream2(const void* X, long ldx, const void* dZx, long ldz, const void* dL2, long ldl, int K2, const float* rs, const void* Wt, long ldw, long sWg,
                 const void* Text, long ldt, long sT1, void* dX, long ldc, int S, int N, int G, int Cg, int K1, hipStream_t st);
// the hop-1 chain's per-frame products against Y as streaming kernels (hop1_stream.hip; bf16, tuned widths); 1 = shape not served
int k_hop1_yk(const void* Y, long ldy, int S, int M, int Cy, const void* A, long lda, long sA1, int rows, void* C, long ldc, long sC1, int c_bf16, void* dump, hipStream_t st);   // C[s] = A[(s)] Y[s]^T ; dump: 16 writable bytes
int k_hop1_yt_frames(const void* Y, long ldy, int S, int M, int Cy, const void* A, long lda, long sA1, int rows, void* C, long ldc, long sC1, int c_bf16, hipStream_t st);   // C[s] = A[s] Y[s]
int k_hop1_yt_sum(const void* Y, long ldy, long ntok, int Cy, const void* A, long lda, int rows, void* C, long ldc, int c_bf16, float* slabs, size_t slab_cap, hipStream_t st);   // C = A^T Y over all tokens
// site A's dX product + site B's dY product into one tensor, written once (dx_stream3.hip); 1 = shape not served
int k_dx_stream3(const void* X, long ldx, const void* dZx, long ldz, const void* dL2, long ldl, int K2, const float* rs, const void* Wt, long ldw, long sWg,
                 const void* Text, long ldt, long sT1, const void* Bm, long ldb, long sB1, int KB, const void* dRT, long ldr, const void* dV, long ldv, long sV1,
                 const void* Q, long ldq, int KQ, void* dX, long ldc, void* dump, int S, int N, int G, int Cg, int K1, hipStream_t st);
int k_nxn_rowdot(int bf16, const void* X, const float* y, long rows, int C, void* dX, float* rowdot, hipStream_t st);   // dX += y ; rowdot = sum_c X y
// register-resident variants for bottleneck 64 / 2 groups / 32 latent tokens / 4 experts (tile_fast.hip)
bool tile_fast_ok(const Dims& d);
bool tile_fast_shape(const Dims& d);      // the shape alone (the Gram-fused mode of gram.hip serves it on either kernel family)
int kf_pre_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);
int kf_post_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);
int kf_post_small_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st, int dap16 = 0);
int kf_mid(const Plan& pl, char* saved, char* scratch, hipStream_t st);
int kf_mid_bwd(const Plan& pl, char* saved, char* scratch, hipStream_t st);
int kf_pre_small_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);
int kf_pre_lat_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);
// the same kernels in streaming form (tile_stream.hip: one persistent block per CU, wave-private LDS rings filled by direct loads);
// 0 = launched, 1 = not served (run the kf_* kernel), < 0 error
bool tile_stream_ok(const Dims& d);
bool kfs_serves_post_small_bwd(const Dims& d, int dap16);
int kfs_post_small_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st, int dap16);
bool kfs_serves_pre_small(const Dims& d);
int kfs_pre_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);
bool kfs_serves_post_small(const Dims& d);
int kfs_post_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);
bool kfs_serves_pre_bwd(const Dims& d);      // pre_small_bwd + pre_lat_bwd in one pass
int kfs_pre_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);
bool kfs_serves_mid_bwd(const Dims& d);
int kfs_mid_bwd(const Plan& pl, char* saved, char* scratch, hipStream_t st);
// register-resident kernels generalised over groups (1 / 2 / 4), per-group bottleneck (16 .. 96) and latent slots (16 / 32 / 96): tile_gen.hip
bool tile_gen_ok(const Dims& d);
int kg_pre_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);
int kg_post_small(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);
int kg_post_small_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st, int dap16 = 0);
int kg_mid(const Plan& pl, char* saved, char* scratch, hipStream_t st);
int kg_mid_bwd(const Plan& pl, char* saved, char* scratch, hipStream_t st);
int kg_pre_small_bwd(const Plan& pl, char* saved, char* scratch, const avmoe_moe_ptrs& prm, hipStream_t st);
// streaming Gram of z' for the register-resident shape in bf16 (gram.hip): out[g*E][dgp][dgp] = scale * sum_t w[e][t] z' z'^T
int k_gram64(const Plan& pl, const void* Zp, const float* w, float scale, float* part, float* out, hipStream_t st, const float* bn1 = nullptr,
             float* colpart = nullptr, float* mz = nullptr);   // bn1: the input is z, z' = act(BN1(z)) on the fly; mz: column means of z' as well
// generic helpers
int k_colsum_f32(const float* in, long R, int ncol, long row_stride, int nslot, long slot_in, float* out, long slot_out,
                 float scale, hipStream_t st);
// two independent column sums (one slot each) in one launch when they are of the same row-count class; bit-identical to two launches
int k_colsum2_f32(const float* in0, long R0, int ncol0, long rs0, float* out0, float scale0,
                  const float* in1, long R1, int ncol1, long rs1, float* out1, float scale1, hipStream_t st);
int k_reduce_colpart(const Plan& pl, char* scratch, int slot0, int nslots, hipStream_t st);   // colpart -> colsum
int k_fill_f32(float* p, long n, float v, hipStream_t st);
int k_cast(int bf16_out, const float* src, long rows, int cols, long ld_src, void* dst, long ld_dst, hipStream_t st);

}  // namespace avmoe
