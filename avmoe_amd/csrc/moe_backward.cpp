// Backward orchestration (filled in below).
#include "moe_run.h"

namespace avmoe {

int moe_backward(const Plan& pl, const void* X, const void* Y, const avmoe_moe_ptrs& prm, const void* dOut, float lb_weight,
                 char* saved, char* scratch, void* dX, void* dY, const avmoe_moe_ptrs& grads, hipStream_t st) {
  set_last_error("moe_backward: not built yet");
  return ERR_UNSUPPORTED;
}

}  // namespace avmoe
