/* avmoe_host.h -- the adapter-site ABI of avmoe.h evaluated on the HOST (CPU, fp32, host pointers).
 *
 * "a host (CPU, C++) implementation of the same ABI is the restatement used for CPU timing and no-GPU CI".  Same
 * descriptor (avmoe_moe_desc), same parameter structs (avmoe_moe_ptrs: one pointer per reference state_dict entry), same token-major
 * tensors and the same results as avmoe_moe_forward / avmoe_moe_backward -- but the reference's formulation evaluated directly
 * (net_trans_v3.py:377-487 op by op, hand-written reverse pass), no workspaces, no streams.  Library: avmoe_amd/lib/libavmoe_host.so
 * (avmoe_amd/csrc/host_moe.cpp, built with g++ by avmoe_amd.build.build_host()).  TEST / CI infrastructure: the product path
 * (libavmoe_hip.so) never loads it and has no CPU fallback.
 *
 * Served: AVE / AVQA / AVVP (N x N block) / AVS (self_attention_version "v2" and, since round 6, "v1": MultiheadAttention across the frames with
 * the caller's dropout multipliers, avmoe_expert_ptrs::sa_keep), training and eval BatchNorm, every flag, logit noise, load-balancing loss.
 * AVMOE_ERR_UNSUPPORTED: dtype != AVMOE_F32.                                                                                             */
#ifndef AVMOE_HOST_H
#define AVMOE_HOST_H
#include "avmoe.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* avmoe_host_last_error(void);
size_t avmoe_host_moe_saved_bytes(const avmoe_moe_desc* desc);       /* the backward recomputes the forward: `saved` may be NULL */

/* as avmoe_moe_forward: out (S, N, C), probs (S, E) or NULL, idx (S) int64 or NULL, lb (1 float) or NULL; in training mode the
 * BatchNorm running statistics / num_batches_tracked in `params` advance.                                                              */
int avmoe_host_moe_forward(const avmoe_moe_desc* desc, const float* X, const float* Y, const avmoe_moe_ptrs* params, const float* noise,
                           float* out, float* probs, int64_t* idx, float* lb, void* saved);

/* as avmoe_moe_backward: gradients of <out, dOut> + (*lb_grad) * lb; every non-NULL pointer in `grads` is overwritten, dX / dY are
 * overwritten.  `noise`: the same logit noise the forward saw (the forward is recomputed; running statistics do not advance again).
 * lb_grad: HOST pointer or NULL.                                                                                                       */
int avmoe_host_moe_backward(const avmoe_moe_desc* desc, const float* X, const float* Y, const avmoe_moe_ptrs* params, const float* noise,
                            const float* dOut, const float* lb_grad, void* saved, float* dX, float* dY, const avmoe_moe_ptrs* grads);

#ifdef __cplusplus
}
#endif
#endif
