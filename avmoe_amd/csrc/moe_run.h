// Internal entry points behind avmoe_moe_forward / avmoe_moe_backward.
#pragma once
#include "moe_plan.h"
#include "gemm.h"
#include "kernels.h"
#include <algorithm>

namespace avmoe {

int choose_ksplit(const GemmArgs& g, size_t slab_floats_cap);

int moe_forward(const Plan& pl, const void* X, const void* Y, const avmoe_moe_ptrs& prm, const float* noise, void* out,
                float* probs_out, int64_t* idx_out, float* lb_out, char* saved, char* scratch, hipStream_t st);

int moe_backward(const Plan& pl, const void* X, const void* Y, const avmoe_moe_ptrs& prm, const void* dOut, const float* lb_grad,
                 char* saved, char* scratch, void* dX, void* dY, const avmoe_moe_ptrs& grads, hipStream_t st);

}  // namespace avmoe
