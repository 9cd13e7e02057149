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
 *     mutable state besides a thread-local error string, the optional profiler (avmoe_prof_*) and the
 *     test hooks (avmoe_test_hooks); the environment is read ONCE per process, never per call
 *   - return 0 on success, negative on error (avmoe_last_error() has the message)
 */
#ifndef AVMOE_H_
#define AVMOE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVMOE_ABI_VERSION 11

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
  int32_t fp32_planes;              /* ABI 9, fp32 operands: 0 = products on v_mfma_f32_16x16x4_f32, 1 = on the bf16 matrix pipe with every value as three
                                       bf16 planes (six plane products of order <= 2: 5.8e-9 relative per product, fp32 accumulation) -- what the site's
                                       site calls use; 2 (ABI 11) = two planes, the three plane products of order <= 1: 1.5e-5 relative per product at half the
                                       matrix-pipe work (not used by the site calls: csrc/moe_run.h); avmoe_amd/csrc/gemm.h::GemmArgs::split3 */
  float alpha;
  int64_t lda, ldb, sA1, sA2, sB1, sB2;
  int64_t sCi, sCj, sC1, sC2;
  int64_t sRS1, sRS2, sDi, sD1, sD2;
} avmoe_gemm_desc;

size_t avmoe_gemm_workspace_bytes(const avmoe_gemm_desc* desc);
int avmoe_gemm(const avmoe_gemm_desc* desc, const void* A, const void* B, void* C,
               const float* row_scale, const void* D, void* workspace, void* stream);


/* ---- the adapter site: MoEAdapter.forward / backward -------------------------------------------
 * Replaces  MoEAdapter.forward(x, vis_token[, is_training])  (AVE net_trans_v3.py:468-487,
 * AVQA net_avst_v2.py:381-399, AVVP mgn.py:185-217, AVS PVT_AVSModel_v2.py:283-312) and its autograd.
 *
 * Tensors are TOKEN-MAJOR and contiguous: X (S, N, C), Y (S, M, Cy), out / dOut / dX (S, N, C),
 * dY (S, M, Cy), element type desc.dtype (AVMOE_F32 | AVMOE_BF16).  [The reference hands the module
 * (S, C, N, 1) permuted VIEWS of exactly this memory -- net_trans_v3.py:695.]
 * Parameters / buffers / their gradients are fp32, one pointer per reference state_dict entry.      */
#define AVMOE_MAX_EXPERTS 16
enum { AVMOE_VARIANT_AVE = 0, AVMOE_VARIANT_AVVP = 1, AVMOE_VARIANT_AVS = 2 };   /* AVQA == AVE math */
enum { AVMOE_SELF_ATTN_NONE = 0, AVMOE_SELF_ATTN_LATENT_V2 = 1, AVMOE_SELF_ATTN_NXN = 2,
       AVMOE_SELF_ATTN_MHA_V1 = 3 };   /* AVS self_attention_version "v1": nn.MultiheadAttention(C, 4) across the FRAMES (ABI 3) */

typedef struct avmoe_moe_desc {
  int32_t S, N, C;          /* this modality: frames, tokens, channels  (C = input_dim = linear_out) */
  int32_t M, Cy;            /* other modality: tokens (conv_dim_in), channels (linear_in)            */
  int32_t E_m, E_s;         /* opt.num_multimodal_experts, opt.num_singlemodal_experts               */
  int32_t d, groups, K;     /* bottleneck = C // reduction_factor, opt.num_conv_group, num_tk         */
  int32_t use_bn, use_gate, ln_before, ln_post;
  int32_t variant, self_attn, lb_loss;
  int32_t dtype;            /* activations */
  int32_t training;         /* 1: BatchNorm batch statistics + running-stat update ; 0: running stats */
  float bn_eps, ln_eps, bn_momentum;
  /* backward only: add dX / dY to what the output buffers already hold instead of overwriting them -- lets a caller whose
   * token tensor feeds several sites (the audio tokens are X of the audio site and Y of the visual site,
   * net_trans_v3.py:695-698) collect the gradient in one buffer without a separate accumulation pass (ABI 2) */
  int32_t accumulate_dx, accumulate_dy;
  /* forward only (ABI 3): out += adapter(X, Y) instead of out = ... -- the caller's residual stream (x + attention(x), then
   * "+ adapter residual", net_trans_v3.py:706-709) takes the adapter's contribution inside the output GEMM's epilogue */
  int32_t accumulate_out;
  /* ABI 8: 1 = kernels of OTHER streams may be on the GPU while this call runs (a caller that overlaps two sites on two streams:
   * AdapterPair's two-stream mode; a backbone GEMM on another stream).  0 = the call is alone on the GPU (its own helper stream
   * never runs beside the kernels below).
   * Why: on MI355X / ROCm 7.2 a workgroup that shares a COMPUTE UNIT with a workgroup of another kernel that keeps the matrix pipe busy
   * (v_mfma_f32_32x32x16_bf16 chains, a GEMM tile's 16 independent v_mfma_f32_16x16x32_bf16 chains) occasionally gets wrong results in
   * its own upper lanes: v_mfma_f32_16x16x4_f32 sums (lanes 48 - 51) and per-block column sums.  scripts/mfma_probe.hip shows it with two
   * stand-alone kernels; every mode and mitigation of it, run on an MI355X, is kept in profiles/r05_mfma_probe.txt (+ _part2): 0
   * mismatches alone or beside LDS-only / sparse-MFMA aggressors, 10^5 - 10^6 per run beside a dense-MFMA aggressor whatever the
   * victim's LDS read width (ds_read_b128 or two ds_read_b64) and whatever wait states follow its MFMAs, and 0 again once the
   * victim's blocks take a whole CU's LDS each (no other block fits beside them; requests with which 2 or 4 of them fill a CU do not
   * suffice: a block can land beside foreign blocks that are already there).
   * Round 6 (profiles/r06_mfma_probe.txt): the same victim with its mat-vecs in the split-bf16 form of the tuned kernels (two bf16 planes,
   * v_mfma_f32_16x16x32_bf16) is corrupted just the same beside an aggressor that issues v_mfma_f32_32x32x16_bf16 (7 x 10^7 mismatches in 200
   * repetitions; 4 x 10^5 .. 1.5 x 10^6 beside sixteen independent chains of 16x16x32; 0 alone, 0 beside an aggressor of its own instruction
   * shape) -- no form of these kernels is immune by construction, only placement protects them.
   * What the flag does: every bottleneck-space kernel that uses the matrix pipe keeps other kernels' blocks off its compute units --
   *   - the generalised family (csrc/tile_gen.inc) and the any-shape fallback (csrc/tile_kernels.hip) launch with 150 KB of dynamic LDS, one
   *     block per CU; the backward does not fork its dBpost product beside them;
   *   - the tuned instance at LARGE bf16 sites (csrc/tile_stream.hip: pre_small, post_small, post_small_bwd, mid_bwd -- one persistent block
   *     per CU by design) asks for the whole 160 KB;
   *   - the tuned instance elsewhere (csrc/tile_fast.hip: every small site; pre_small_bwd / pre_lat_bwd of large sites) asks for 150 KB per
   *     block where the grid has at most one block per CU (nothing is lost) and for 80 KB where it is larger (two blocks fill a CU; a foreign
   *     block can only land beside the first or the last block of a CU); the streaming Gram kernel (csrc/gram.hip) asks for 80 KB likewise.
   * The streaming GEMM kernels and the tiled engine are not touched: their products have not moved in any two-stream repetition (they issue
   * long independent chains; what goes wrong in the probe is a short dependent chain whose result the VALU consumes at once).
   * bench.py reports `roofline.two_stream_bit_equal` (three steps on two streams against the same schedule on one stream, every gradient bit for
   * bit) on every default run; tests/test_two_stream_repeat_gpu.py is the longer guard.
   * Cost: residency of those kernels where the grid exceeds the chip (BASELINE config 4: 80 -> 90 ms per step).  The flag changes their
   * block shape, i.e. the summation order of the per-block BatchNorm column sums: results agree to fp32 rounding of those sums, and
   * repeat bit for bit for a given flag.
   * The Python wrappers set it in AdapterPair's two-stream mode and, by default (warned once), for every call issued on a stream other than
   * the device's default stream (avmoe_amd.adapters.set_shared_gpu). */
  int32_t shared_gpu;
} avmoe_moe_desc;

typedef struct avmoe_expert_ptrs {        /* <list>.{j}.*  ; unused entries NULL                     */
  float *gate, *my_tokens, *gate_lat;     /* gate ; my_tokens ; gate_av | gate_self                   */
  float *down_w, *up_w;                   /* down_sampler.weight (d, C/g) ; up_sampler.weight (C, d/g)*/
  float *bn1_w, *bn1_b, *bn2_w, *bn2_b;
  float *lnb_w, *lnb_b, *lnp_w, *lnp_b;   /* ln_before.* ; ln_post.*                                  */
  float *bn1_rm, *bn1_rv, *bn2_rm, *bn2_rv; /* running_mean / running_var (updated in place in training) */
  /* ABI 3 -- AVS unimodal expert with self_attention_version "v1" (PVT_AVSModel_v2.py:141-142,210-214):
   * self_attention.{in_proj_weight (3C, C), in_proj_bias (3C), out_proj.weight (C, C), out_proj.bias (C)} and, in `params`
   * only, sa_keep: the dropout multiplier of the attention weights, (N * 4, S, S) f32 holding 0 or 1 / (1 - p), or NULL for
   * no dropout (eval).  The caller draws it (the reference uses the global RNG) and keeps it alive until the backward.  */
  float *sa_in_w, *sa_in_b, *sa_out_w, *sa_out_b, *sa_keep;
  /* ABI 6 -- bn1.num_batches_tracked / bn2.num_batches_tracked (one int64 each), or NULL: a training-mode forward with BatchNorm
   * adds 1 to each inside its own kernels (torch.nn.BatchNorm2d semantics), so the caller needs no extra launch for the counters. */
  int64_t *bn1_nbt, *bn2_nbt;
} avmoe_expert_ptrs;

typedef struct avmoe_moe_ptrs {
  float *conv_w, *conv_b, *fc_w, *fc_b;   /* conv_adapter.{weight (N, M), bias} ; fc.{weight (C, Cy), bias} */
  float *r0_w, *r0_b, *r2_w, *r2_b, *r4_w, *r4_b;   /* router.{0,2,4}.*                               */
  avmoe_expert_ptrs e[AVMOE_MAX_EXPERTS]; /* multimodal experts first, then singlemodal               */
} avmoe_moe_ptrs;

size_t avmoe_moe_saved_bytes(const avmoe_moe_desc* desc);     /* 0 + error string on a bad descriptor */
size_t avmoe_moe_scratch_bytes(const avmoe_moe_desc* desc);

/* out (S,N,C) ; probs (S,E) f32 ; idx (S) int64 = first-max argmax of probs ; lb: 1 float (0 if !lb_loss).
 * noise: optional (S,E) f32 already scaled by 0.01 (AVS logit noise), or NULL.                       */
int avmoe_moe_forward(const avmoe_moe_desc* desc, const void* X, const void* Y, const avmoe_moe_ptrs* params,
                      const float* noise, void* out, float* probs, int64_t* idx, float* lb,
                      void* saved, void* scratch, void* stream);

/* Gradients of  <out, dOut> + (*lb_grad) * lb .  Every pointer in `grads` that is non-NULL is OVERWRITTEN
 * with the gradient of the matching parameter; dX / dY are overwritten.  `saved` must be the buffer the
 * matching forward filled.  lb_grad: DEVICE pointer to the upstream gradient of the load-balancing loss
 * (one float; read on the stream, so no host sync), or NULL for 0.                                    */
int avmoe_moe_backward(const avmoe_moe_desc* desc, const void* X, const void* Y, const avmoe_moe_ptrs* params,
                       const void* dOut, const float* lb_grad, void* saved, void* scratch,
                       void* dX, void* dY, const avmoe_moe_ptrs* grads, void* stream);

/* ABI 4 -- the same backward in stream-ordered sections.  `parts` is a bit mask: 1 = the GEMMs against dOut and the bottleneck /
 * weight space (touches neither dX nor dY), 2 = the GEMMs against X (every writer of dX), 4 = the chain back to Y and the remap
 * parameters (every writer of dY); 0 or 7 = all of it (== avmoe_moe_backward).  ABI 5: section 4 in two steps -- 8 = the chain
 * without the GEMM(s) that write dY (touches neither dX nor dY), 16 = those GEMMs alone.  Sections must be run in this order with the
 * same arguments; between calls the caller may record / wait events on the stream but must leave `saved` and `scratch` alone.
 * Purpose: a token tensor that feeds two sites (the audio tokens are X of the audio site and Y of the visual site,
 * net_trans_v3.py:695-698) collects both gradients in ONE buffer while the two sites run on two streams -- the smaller site
 * runs through and overwrites (its dY / dX), an event orders the larger site's sections 2 and 4 behind it, and that site adds
 * its dX / dY in the GEMM epilogues (accumulate_dx / accumulate_dy); or, cross-wise: each site overwrites its own tokens' gradient
 * (sections 1, 2, 8), then adds its dY to the other tensor (16) once the other site's section 2 is done.
 * Not available (AVMOE_ERR_UNSUPPORTED) for sites with latent self attention (AVS v2), whose last section writes dX too.
 * ABI 10: section 2 in two steps (plain sites: no N x N / frame / latent self attention) -- 32 = the GEMMs against X without the dX
 * product (touches neither dX nor dY), 64 = the dX product alone (after 1; independent of 32 and 8).                              */
int avmoe_moe_backward_part(const avmoe_moe_desc* desc, const void* X, const void* Y, const avmoe_moe_ptrs* params,
                            const void* dOut, const float* lb_grad, void* saved, void* scratch,
                            void* dX, void* dY, const avmoe_moe_ptrs* grads, int32_t parts, void* stream);

/* ABI 10 -- the gradient of a token tensor T that is X of site A and Y of site B (the two adapter sites of one backbone layer:
 * the audio tokens are X of the audio site and Y of the visual site, net_trans_v3.py:695-698), written ONCE:
 *     dT = dX_A + dY_B
 * with site B's dY product ([Bm ; wbar]^T dV + dR^T Q) as two more contraction segments of site A's dX pass, instead of one kernel that
 * overwrites dT and a second one that reads it back and adds (1 GB of HBM traffic less at BASELINE config 2's audio tokens).
 * Call after sections 1 + 32 + 8 of BOTH sites (their `saved` / `scratch` as those calls left them; B's workspaces are only read), in
 * place of section 64 of A and section 16 of B; the stream must be ordered behind both sites' section 8.  dT is overwritten.
 * Returns 0 = launched, 1 = these shapes are not served (nothing launched: run section 64 of A, then section 16 of B with
 * accumulate_dy), < 0 = error.  dT == NULL: nothing is launched, the return value only says whether the shapes are served.
 * Served: bf16 pairs of the tuned shape (one streaming kernel, csrc/dx_stream3.hip) and -- round 6 -- fp32 pairs of ANY shape without the AVVP N x N /
 * frame-attention / latent-self-attention variants on site A (the tiled engine with site B's product as a third and fourth K segment). */
int avmoe_moe_backward_dx_dy(const avmoe_moe_desc* desc_a, const void* X_a, void* saved_a, void* scratch_a,
                             const avmoe_moe_desc* desc_b, void* saved_b, void* scratch_b, void* dT, void* stream);

/* Sub-op (tests / partial adoption): the router alone -- Sequential(Linear(2C,128), ReLU, Linear(128,32), ReLU,
 * Linear(32,E)) + optional logit noise + softmax + first-max argmax  (net_trans_v3.py:460-466,477-479).
 * rin (S, 2C) f32 = [mean over tokens of x | mean over tokens of the remapped other modality]; same workspaces as the
 * site calls (rin is copied into `saved`; a following avmoe_moe_backward on it is not meaningful).  probs (S,E) f32,
 * idx (S) int64, lb (1 float, written only when desc.lb_loss) may each be NULL.                                    */
int avmoe_router_forward(const avmoe_moe_desc* desc, const float* rin, const avmoe_moe_ptrs* params, const float* noise,
                         float* probs, int64_t* idx, float* lb, void* saved, void* scratch, void* stream);

/* ABI 7 -- sub-ops (tests / partial adoption), same workspaces as the site calls; a following avmoe_moe_backward on them is not
 * meaningful.  The product path never materialises either; these exist so that a maintainer can compare the
 * library with the reference one module at a time.
 * avmoe_expert_forward_cross / _uni: what ExpertAdapter.forward returns for multimodal_experts[j] / singlemodal_experts[j]
 * (gate * LN_post(BN2(up(act(BN1(down(LN_before(x'))))))), net_trans_v3.py:377-435 ; mgn.py:132-139 and PVT_AVSModel_v2.py:210-227
 * for the unimodal variants) into out (S, N, C): the site forward with the router pushed to an exact one-hot on that expert.  In
 * training mode only THAT expert's BatchNorm running statistics and counters advance (the others' updates are discarded); a
 * non-finite value in another expert's output propagates (0 * Inf), as in the reference's mixture.
 * avmoe_remap_forward: the remapped other modality, materialised -- Yt = conv_adapter(Y) (S, N, Cy) and Yf = fc(Yt) (S, N, C), the
 * `vis_token` every expert and the router read (net_trans_v3.py:469-471); both in desc.dtype, both written.                      */
int avmoe_expert_forward_cross(const avmoe_moe_desc* desc, const void* X, const void* Y, const avmoe_moe_ptrs* params, int32_t j,
                               void* out, void* saved, void* scratch, void* stream);
int avmoe_expert_forward_uni(const avmoe_moe_desc* desc, const void* X, const void* Y, const avmoe_moe_ptrs* params, int32_t j,
                             void* out, void* saved, void* scratch, void* stream);
int avmoe_remap_forward(const avmoe_moe_desc* desc, const void* Y, const avmoe_moe_ptrs* params, void* Yt, void* Yf, void* saved,
                        void* scratch, void* stream);

/* Workspace introspection for tests: buffer `index` -> name / region (0 saved, 1 scratch) / offset / bytes.
 * Returns 0, or AVMOE_ERR_BAD_ARG when index is past the last buffer.                                 */
int avmoe_moe_buffer_info(const avmoe_moe_desc* desc, int32_t index, const char** name, int32_t* region,
                          size_t* offset, size_t* bytes);


/* ---- either side of the path on the training loop ---------------------------------------
 * avmoe_adam_step: one torch.optim.Adam step (no amsgrad; L2 weight_decay added to the gradient) over a FLAT fp32
 * parameter bucket -- replaces optimizer.step() over the adapter parameters (AVE/main_trans_v3.py:322).  `step` is the
 * 1-based step count (bias correction), grad_scale multiplies the gradient first (1/accum_itr, 1/world ...).
 * avmoe_expert_histogram: counts[e] += #{s : idx[s] == e} on the device (int64 counts, exact) -- replaces the host loop
 * over idx.tolist() that fills the expert-activation tables (AVE/main_trans_v3.py:183-207).                          */
int avmoe_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream);
int avmoe_expert_histogram(const int64_t* idx, int64_t S, int32_t E, int64_t* counts, void* stream);
/* ABI 4 -- dst1 += src1 and dst2 += src2 (n1 / n2 elements of `dtype`, 16-byte aligned) in ONE launch: how the two token tensors of a
 * site pair (net_trans_v3.py:695-698) collect their second gradient when the two sites' backward passes ran on two streams.   */
int avmoe_add2(void* dst1, const void* src1, int64_t n1, void* dst2, const void* src2, int64_t n2, int32_t dtype, void* stream);
/* Extension (no reference counterpart; BASELINE config 3 "router top-k=2"): idx (S, k) int64 = the k most probable experts of
 * every frame from probs (S, E) f32, most probable first, equal probabilities in expert order -- column 0 is the forward's
 * first-max argmax.  The mixture itself stays dense (net_trans_v3.py:482-486).                                          */
int avmoe_router_topk(const float* probs, int64_t S, int32_t E, int32_t k, int64_t* idx, void* stream);

/* ---- test hooks (ABI 11; process-wide; tests and bench.py's parity leg only) ---------------------------
 * The streaming kernels (csrc/dpost_pair.hip, tok_pair2.hip, hop1_stream.hip) serve sites from 32 768 tokens on; below that the
 * tiled engine is faster.  force_mask lifts those thresholds so that a test can run the benchmarked kernels on shapes its oracle
 * finishes in seconds: bit 1 = tok_pair2, 2 = dpost_pair (from 4096 tokens), 4 = the hop-1 products against Y, 8 = the streaming form of the
 * bottleneck-space kernels (csrc/tile_stream.hip); bit 16 switches that form OFF (the A/B against csrc/tile_fast.hip).  nxn_chunk > 0:
 * frames per chunk of the AVVP N x N block (0 = the library's own choice).  Returns the previous force_mask.  The initial values
 * come from the environment variables AVMOE_TOKPAIR2_FORCE / AVMOE_DPAIR_FORCE / AVMOE_HOP1S_FORCE / AVMOE_KFS_FORCE / AVMOE_KFS_OFF / AVMOE_NXN_CHUNK, read once
 * when the library is first asked -- no kernel choice depends on the environment at call time.                                  */
enum { AVMOE_HOOK_TOKPAIR2_FORCE = 1, AVMOE_HOOK_DPAIR_FORCE = 2, AVMOE_HOOK_HOP1S_FORCE = 4, AVMOE_HOOK_KFS_FORCE = 8, AVMOE_HOOK_KFS_OFF = 16 };
uint32_t avmoe_test_hooks(uint32_t force_mask, int32_t nxn_chunk);

/* ---- optional per-launch timing (HIP events on the launch stream; off by default; process-wide) --------
 * avmoe_prof_report writes a JSON array of {"name","calls","total_ms","alg_bytes","flops"} per kernel family
 * into buf (NUL-terminated, truncated to cap) and returns the full length.  Used by bench.py for the roofline. */
void avmoe_prof_enable(int on);
void avmoe_prof_reset(void);
size_t avmoe_prof_report(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* AVMOE_H_ */
