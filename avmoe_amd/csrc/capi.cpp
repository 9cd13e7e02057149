// C ABI (include/avmoe.h) -> internal C++ entry points.  No exceptions cross this boundary.
#include "../../include/avmoe.h"
#include "common.h"
#include "gemm.h"
#include "moe_run.h"
#include "prof.h"

using namespace avmoe;

static GemmArgs to_args(const avmoe_gemm_desc* d) {
  GemmArgs a;
  a.M = d->M; a.N = d->N; a.K = d->K; a.nb1 = d->nb1; a.nb2 = d->nb2;
  a.dtype = d->dtype; a.out_dtype = d->out_dtype; a.a_layout = d->a_layout; a.b_layout = d->b_layout;
  a.accumulate = d->accumulate; a.ksplit = d->ksplit; a.tile = d->tile; a.split3 = d->dtype == AVMOE_F32 ? (d->fp32_planes == 2 ? 2 : (d->fp32_planes != 0 ? 1 : 0)) : 0; a.alpha = d->alpha;
  a.lda = d->lda; a.ldb = d->ldb; a.sA1 = d->sA1; a.sA2 = d->sA2; a.sB1 = d->sB1; a.sB2 = d->sB2;
  a.sCi = d->sCi; a.sCj = d->sCj; a.sC1 = d->sC1; a.sC2 = d->sC2;
  a.sRS1 = d->sRS1; a.sRS2 = d->sRS2; a.sDi = d->sDi; a.sD1 = d->sD1; a.sD2 = d->sD2;
  return a;
}

extern "C" {

int avmoe_abi_version(void) { return AVMOE_ABI_VERSION; }
const char* avmoe_last_error(void) { return last_error(); }

size_t avmoe_gemm_workspace_bytes(const avmoe_gemm_desc* desc) {
  if (!desc) return 0;
  return gemm_slab_bytes(to_args(desc));
}

int avmoe_gemm(const avmoe_gemm_desc* desc, const void* A, const void* B, void* C, const float* row_scale,
               const void* D, void* workspace, void* stream) {
  if (!desc) { set_last_error("avmoe_gemm: null desc"); return ERR_BAD_ARG; }
  GemmArgs a = to_args(desc);
  a.A = A; a.B = B; a.C = C; a.row_scale = row_scale; a.D = D; a.slabs = (float*)workspace;
  return launch_gemm(a, (hipStream_t)stream);
}


size_t avmoe_moe_saved_bytes(const avmoe_moe_desc* desc) {
  Plan pl;
  return make_plan(desc, &pl) == OK ? pl.saved_bytes : 0;
}
size_t avmoe_moe_scratch_bytes(const avmoe_moe_desc* desc) {
  Plan pl;
  return make_plan(desc, &pl) == OK ? pl.scratch_bytes : 0;
}

int avmoe_moe_forward(const avmoe_moe_desc* desc, const void* X, const void* Y, const avmoe_moe_ptrs* params,
                      const float* noise, void* out, float* probs, int64_t* idx, float* lb, void* saved, void* scratch,
                      void* stream) {
  Plan pl;
  AVMOE_TRY(make_plan(desc, &pl));
  if (!X || !Y || !params || !out || !saved || !scratch) { set_last_error("avmoe_moe_forward: null pointer"); return ERR_BAD_ARG; }
  return moe_forward(pl, X, Y, *params, noise, out, probs, idx, lb, (char*)saved, (char*)scratch, (hipStream_t)stream);
}

int avmoe_moe_backward(const avmoe_moe_desc* desc, const void* X, const void* Y, const avmoe_moe_ptrs* params,
                       const void* dOut, const float* lb_grad, void* saved, void* scratch, void* dX, void* dY,
                       const avmoe_moe_ptrs* grads, void* stream) {
  Plan pl;
  AVMOE_TRY(make_plan(desc, &pl));
  if (!X || !Y || !params || !dOut || !saved || !scratch || !dX || !dY || !grads) {
    set_last_error("avmoe_moe_backward: null pointer"); return ERR_BAD_ARG;
  }
  return moe_backward(pl, X, Y, *params, dOut, lb_grad, (char*)saved, (char*)scratch, dX, dY, *grads, (hipStream_t)stream);
}

int avmoe_moe_backward_part(const avmoe_moe_desc* desc, const void* X, const void* Y, const avmoe_moe_ptrs* params,
                            const void* dOut, const float* lb_grad, void* saved, void* scratch, void* dX, void* dY,
                            const avmoe_moe_ptrs* grads, int32_t parts, void* stream) {
  Plan pl;
  AVMOE_TRY(make_plan(desc, &pl));
  if (!X || !Y || !params || !dOut || !saved || !scratch || !dX || !dY || !grads || parts < 0 || parts > 127) {
    set_last_error("avmoe_moe_backward_part: null pointer or parts not in 0..127"); return ERR_BAD_ARG;
  }
  if ((parts & 4) && (parts & 24)) {        // 4 IS sections 8 + 16: asking for both would run the hop-1 chain twice into the same accumulators
    set_last_error("avmoe_moe_backward_part: parts %d combines section 4 with its halves 8 / 16", parts); return ERR_BAD_ARG;
  }
  return moe_backward(pl, X, Y, *params, dOut, lb_grad, (char*)saved, (char*)scratch, dX, dY, *grads, (hipStream_t)stream, parts);
}

int avmoe_moe_backward_dx_dy(const avmoe_moe_desc* desc_a, const void* X_a, void* saved_a, void* scratch_a,
                             const avmoe_moe_desc* desc_b, void* saved_b, void* scratch_b, void* dT, void* stream) {
  Plan pa, pb;
  AVMOE_TRY(make_plan(desc_a, &pa));
  AVMOE_TRY(make_plan(desc_b, &pb));
  if (dT && (!X_a || !saved_a || !scratch_a || !saved_b || !scratch_b)) { set_last_error("avmoe_moe_backward_dx_dy: null pointer"); return ERR_BAD_ARG; }
  return moe_backward_dx_dy(pa, X_a, (char*)saved_a, (char*)scratch_a, pb, (char*)saved_b, (char*)scratch_b, dT, dT != nullptr, (hipStream_t)stream);
}

int avmoe_router_forward(const avmoe_moe_desc* desc, const float* rin, const avmoe_moe_ptrs* params, const float* noise,
                         float* probs, int64_t* idx, float* lb, void* saved, void* scratch, void* stream) {
  Plan pl;
  AVMOE_TRY(make_plan(desc, &pl));
  if (!rin || !params || !saved || !scratch) { set_last_error("avmoe_router_forward: null pointer"); return ERR_BAD_ARG; }
  const size_t bytes = (size_t)pl.d.S * 2 * pl.d.C * sizeof(float);
  if (hipMemcpyAsync((char*)saved + pl.o_rin, rin, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
    set_last_error("avmoe_router_forward: copy of rin failed"); return ERR_LAUNCH;
  }
  return k_router(pl, (char*)saved, (char*)scratch, *params, noise, probs, idx, lb, (hipStream_t)stream);
}

int avmoe_expert_forward_cross(const avmoe_moe_desc* desc, const void* X, const void* Y, const avmoe_moe_ptrs* params, int32_t j,
                               void* out, void* saved, void* scratch, void* stream) {
  Plan pl;
  AVMOE_TRY(make_plan(desc, &pl));
  if (!X || !Y || !params || !out || !saved || !scratch) { set_last_error("avmoe_expert_forward_cross: null pointer"); return ERR_BAD_ARG; }
  if (j < 0 || j >= desc->E_m) { set_last_error("avmoe_expert_forward_cross: expert %d of %d cross-modal experts", j, desc->E_m); return ERR_BAD_ARG; }
  return expert_forward(pl, X, Y, *params, j, out, (char*)saved, (char*)scratch, (hipStream_t)stream);
}

int avmoe_expert_forward_uni(const avmoe_moe_desc* desc, const void* X, const void* Y, const avmoe_moe_ptrs* params, int32_t j,
                             void* out, void* saved, void* scratch, void* stream) {
  Plan pl;
  AVMOE_TRY(make_plan(desc, &pl));
  if (!X || !Y || !params || !out || !saved || !scratch) { set_last_error("avmoe_expert_forward_uni: null pointer"); return ERR_BAD_ARG; }
  if (j < 0 || j >= desc->E_s) { set_last_error("avmoe_expert_forward_uni: expert %d of %d unimodal experts", j, desc->E_s); return ERR_BAD_ARG; }
  return expert_forward(pl, X, Y, *params, desc->E_m + j, out, (char*)saved, (char*)scratch, (hipStream_t)stream);
}

int avmoe_remap_forward(const avmoe_moe_desc* desc, const void* Y, const avmoe_moe_ptrs* params, void* Yt, void* Yf, void* saved,
                        void* scratch, void* stream) {
  Plan pl;
  AVMOE_TRY(make_plan(desc, &pl));
  if (!Y || !params || !Yt || !Yf || !saved || !scratch) { set_last_error("avmoe_remap_forward: null pointer"); return ERR_BAD_ARG; }
  return remap_forward(pl, Y, *params, Yt, Yf, (char*)saved, (char*)scratch, (hipStream_t)stream);
}

int avmoe_moe_buffer_info(const avmoe_moe_desc* desc, int32_t index, const char** name, int32_t* region, size_t* offset,
                          size_t* bytes) {
  Plan pl;
  AVMOE_TRY(make_plan(desc, &pl));
  if (index < 0 || index >= pl.nbuf) { set_last_error("buffer index %d out of range", index); return ERR_BAD_ARG; }
  if (name) *name = pl.info[index].name;
  if (region) *region = pl.info[index].region;
  if (offset) *offset = pl.info[index].offset;
  if (bytes) *bytes = pl.info[index].bytes;
  return OK;
}


uint32_t avmoe_test_hooks(uint32_t force_mask, int32_t nxn_chunk) {
  const unsigned prev = test_hook_mask();
  set_test_hooks(force_mask, nxn_chunk);
  return prev;
}

void avmoe_prof_enable(int on) { prof_enable(on != 0); }
void avmoe_prof_reset(void) { prof_reset(); }
size_t avmoe_prof_report(char* buf, size_t cap) { return prof_report(buf, cap); }

}  // extern "C"
