// Internal entry points behind avmoe_moe_forward / avmoe_moe_backward.
#pragma once
#include "moe_plan.h"
#include "gemm.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>

// fp32 sites: the FORWARD's engine products on the bf16 matrix pipe in three-plane form too (gemm.h: GemmArgs::split3: 5.8e-9 relative per product;
// the backward's have been since round 4).  Round 4 kept v_mfma_f32_16x16x4_f32 here because ONE ReLU unit of the 32-site block-loop fixture sat
// within rounding of zero and changed side; with this round's arithmetic every fp32 test (412, block loop included) passes either way, and the
// three-plane form is worth 5 % of the fp32 step (cfg-2: 15.61 -> 14.84 ms, 2050 -> 2156 clip-pairs/s; the router's products stay exact fp32).
// -DAVMOE_FWD_SPLIT3=0: the exact-fp32 matrix pipe, bit-identical to rounds 1 - 4.
#ifndef AVMOE_FWD_SPLIT3
#define AVMOE_FWD_SPLIT3 1
#endif

// fp32 sites, BACKWARD: GemmArgs::split3 of its engine products.  1 = three planes (rounds 4 - 6).  2 = TWO planes split once on the way into
// the LDS (gemm.hip: f32s2, round 6) was built and measured: the fp32 cfg-2 step 15.8 -> 13.0 ms of kernel time (dBpost 867 -> 425 us, dWt 566 -> 314,
// the output-bound dX / dY products unchanged), but gradients that cancel structurally (bn1.bias of the unimodal experts: 5.7e-6 against a largest
// gradient of ~0.1) come out at 1.2e-4 -- past the 1e-3 bar of the fixture tests (tests/test_adapters_gpu.py).  fp32 is THE parity mode: not used.
#ifndef AVMOE_BWD_PLANES
#define AVMOE_BWD_PLANES 1
#endif

// fp32 sites, BACKWARD, round 6 (second half): the two-plane form for the products NOTHING DOWNSTREAM FORMS A CANCELLING SUM FROM -- a bit mask over
// classes of call sites in moe_backward.cpp:  1 dBpost = dOut^T Apost ;  2 dWt = dZx^T X ;  4 dT[s] = dL2[s]^T X[s] ;  8 dWf, dWcK (four products) ;
// 16 the token gradients dX / dY.  Measured on MI355X (fp32 cfg-2 step, moe_backward.cpp rebuilt per mask, the fp32 fixture / cfg-2 / block-loop
// tests under each):  0: 14.60 ms ; 1: FAILS (dBpost feeds the BatchNorm-2 sums of post_prep_bwd: it is not a leaf) ; 10: 14.05 ; 16: 14.22 ;
// 26: 13.88 (2306 clip-pairs/s) ; 30: 13.98 ; 27: 13.23, fails.  26 it is: all 198 mid-size / fuzz / training-loop cases pass under 30 as well.
#ifndef AVMOE_LEAF2
#define AVMOE_LEAF2 26
#endif

// moe_backward_dx_dy on the tiled engine (the other site's dY as a third and fourth K segment of this site's dX product): bit 0 = fp32 sites,
// bit 1 = bf16 sites whose shape the streaming form (dx_stream3.hip) does not serve.
#ifndef AVMOE_DXDY_GEN
#define AVMOE_DXDY_GEN 1
#endif

namespace avmoe {

int choose_ksplit(const GemmArgs& g, size_t slab_floats_cap);
// helper streams (side.h) pay for their two event hand-overs only when the forked GEMM is long: sites of at least 2^25 token elements
inline bool side_worth(const Dims& d) { return (long)d.NT * d.C >= side_min_elements(); }

// AVS "v1" unimodal expert: xr slot = MultiheadAttention_e(X) - X across the frames, and its backward (mha_frames.hip)
int mha_frames_forward(const Plan& pl, const void* X, const avmoe_expert_ptrs& ep, int slot, char* saved, char* scratch, hipStream_t st);
int mha_frames_backward(const Plan& pl, const void* X, const avmoe_expert_ptrs& ep, const avmoe_expert_ptrs& eg, int slot, const void* dxr,
                        char* saved, char* scratch, float* slabs, size_t slab_cap, void* dX, hipStream_t st);

int moe_forward(const Plan& pl, const void* X, const void* Y, const avmoe_moe_ptrs& prm, const float* noise, void* out,
                float* probs_out, int64_t* idx_out, float* lb_out, char* saved, char* scratch, hipStream_t st);

int moe_backward(const Plan& pl, const void* X, const void* Y, const avmoe_moe_ptrs& prm, const void* dOut, const float* lb_grad,
                 char* saved, char* scratch, void* dX, void* dY, const avmoe_moe_ptrs& grads, hipStream_t st, int parts = 0);

// site A's dX + site B's dY into one token gradient, written once (moe_backward.cpp / dx_stream3.hip); 1 = shapes not served
int moe_backward_dx_dy(const Plan& pa, const void* X, char* sva, char* sca, const Plan& pb, char* svb, char* scb, void* dX, bool launch, hipStream_t st);

// sub-ops of the C ABI (moe_forward.cpp): one expert's output alone ; the remap materialised
int expert_forward(const Plan& pl, const void* X, const void* Y, const avmoe_moe_ptrs& prm, int e, void* out, char* saved, char* scratch,
                   hipStream_t st);
int remap_forward(const Plan& pl, const void* Y, const avmoe_moe_ptrs& prm, void* Yt, void* Yf, char* saved, char* scratch, hipStream_t st);

}  // namespace avmoe
