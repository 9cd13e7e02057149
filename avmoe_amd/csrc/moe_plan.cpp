#include <cstdlib>
#include <cstdio>
#include "moe_plan.h"
#include "gemm.h"
#include "kernels.h"
#include <algorithm>
#include <cstring>

namespace avmoe {

bool tile_fast_ok(const Dims& d);   // tile_fast.hip
bool tile_fast_shape(const Dims& d);
bool tile_gen_ok(const Dims& d);    // tile_gen.hip

size_t slab_floats(const Dims& d) {
  // worst split-K user: weight-gradient contractions over all tokens.  Sized generously:
  // ksplit_max * (largest small output), see moe_forward/backward for the actual launches.
  const size_t out1 = (size_t)d.g * d.E * d.dgp * d.dgp;              // Szz / dGq
  const size_t out2 = (size_t)d.C * d.KPp;                            // dBpost
  const size_t out3 = (size_t)d.g * d.E * d.dgp * d.Cg;               // dWt
  const size_t out4 = (size_t)d.N * d.Mk + (size_t)d.C * d.Cy;        // dWc, dWf
  const size_t out5 = (size_t)(d.KL ? d.KL : 1) * (d.Cy > d.C ? d.Cy : d.C);
  const size_t out6 = d.mha ? (size_t)3 * d.C * d.C : 0;              // d in_proj_weight
  size_t m = std::max(std::max(std::max(out1, out2), std::max(std::max(out3, out4), out5)), out6);
  size_t need = m * 96 + 1024;
  // dpost_pair.hip (bf16, 384 channels per group, 128 + <= 16 columns): one partial dBpost per block, one block per CU
  if (d.bf16 && d.Cg == 384 && d.E * d.dgp == 128 && d.KPp == 144) need = std::max(need, (size_t)352 * d.Cg * d.KPp + 1024);      // (tok_pair2.hip: 256 x (128 + 64) x 384 floats)
  return need;
}

int make_plan(const avmoe_moe_desc* q, Plan* pl) {
  if (!q || !pl) { set_last_error("moe: null descriptor"); return ERR_BAD_ARG; }
  std::memset(pl, 0, sizeof(Plan));
  Dims& d = pl->d;
  d.S = q->S; d.N = q->N; d.C = q->C; d.M = q->M; d.Cy = q->Cy;
  d.E_m = q->E_m; d.E_s = q->E_s; d.E = q->E_m + q->E_s;
  d.g = q->groups; d.d = q->d; d.K = q->K;
  d.use_bn = q->use_bn; d.use_gate = q->use_gate; d.gate_w = q->use_gate && !dev_env("AVMOE_GATE_TOKEN"); d.ln_before = q->ln_before; d.ln_post = q->ln_post;
  d.variant = q->variant; d.self_attn = q->self_attn; d.lb_loss = q->lb_loss; d.training = q->training;
  d.bf16 = q->dtype == AVMOE_BF16;
  d.bn_eps = q->bn_eps; d.ln_eps = q->ln_eps; d.bn_momentum = q->bn_momentum;
  d.acc_dx = q->accumulate_dx != 0; d.acc_dy = q->accumulate_dy != 0; d.acc_out = q->accumulate_out != 0;
  d.excl = q->shared_gpu != 0;
  if (q->dtype != AVMOE_F32 && q->dtype != AVMOE_BF16) { set_last_error("moe: dtype %d", q->dtype); return ERR_BAD_ARG; }
  if (d.S <= 0 || d.N <= 0 || d.C <= 0 || d.M <= 0 || d.Cy <= 0) {
    set_last_error("moe: non-positive extent S=%d N=%d C=%d M=%d Cy=%d", d.S, d.N, d.C, d.M, d.Cy);
    return ERR_BAD_ARG;
  }
  if (d.E <= 0 || d.E > MAX_E || d.E_m < 0 || d.E_s < 0) {
    set_last_error("moe: expert count %d+%d outside 1..%d", d.E_m, d.E_s, MAX_E);
    return ERR_BAD_ARG;
  }
  if (d.g <= 0 || d.d <= 0 || d.d % d.g || d.C % d.g) {
    set_last_error("moe: bottleneck %d / channels %d not divisible by groups %d", d.d, d.C, d.g);
    return ERR_BAD_ARG;
  }
  if ((d.C / d.g) % 8 || d.Cy % 8) {
    set_last_error("moe: C/groups (%d) and Cy (%d) must be multiples of 8 (16-byte rows)", d.C / d.g, d.Cy);
    return ERR_UNSUPPORTED;
  }
  const bool v2 = d.self_attn == AVMOE_SELF_ATTN_LATENT_V2;
  if ((d.E_m > 0 || v2) && (d.K <= 0 || d.K > 128)) {
    set_last_error("moe: num_tk K=%d outside 1..128", d.K);
    return ERR_UNSUPPORTED;
  }
  d.esz = d.bf16 ? 2 : 4;
  d.NT = d.S * d.N;
  d.El = 0;
  const bool v1 = d.self_attn == AVMOE_SELF_ATTN_MHA_V1;
  if (v1) {
    d.H = 4; d.dh = d.C / d.H; d.Sp = (int)round_up(d.S, 8);           // PVT_AVSModel_v2.py:138
    if (d.C % d.H || d.dh % 8) { set_last_error("moe: self attention v1 needs C / 4 heads to be a multiple of 8 (C=%d)", d.C); return ERR_UNSUPPORTED; }
  }
  for (int e = 0; e < d.E; ++e) {
    const bool multimodal = e < d.E_m;
    d.relu_of_e[e] = multimodal;
    d.nxn_of_e[e] = (!multimodal && (d.variant == AVMOE_VARIANT_AVVP || d.self_attn == AVMOE_SELF_ATTN_NXN || v1)) ? 1 : 0;
    d.xr_of_e[e] = -1;
    if (d.nxn_of_e[e]) {
      d.nxn = 1;
      if (v1) { d.mha = 1; d.xr_of_e[e] = d.nxr++; }
      else { d.xr_of_e[e] = 0; d.nxr = 1; }
    }
    d.lat_of_e[e] = -1;
    if (multimodal || v2) {
      d.lat_of_e[e] = d.El;
      d.e_of_lat[d.El] = e;
      d.src_of_lat[d.El] = multimodal ? 0 : 1;
      d.El++;
    }
  }
  d.Ey = d.E_m;
  d.Ex = d.El - d.Ey;
  if (d.El == 0) d.K = d.K > 0 ? d.K : 1;
  // ---- bottleneck layout ------------------------------------------------------------------------------------------------
  static const bool no_gen = dev_env("AVMOE_NO_GEN") != nullptr;          // development: without the generalised register-resident kernels
  const int g_site = d.g;
  // Merged groups.  A grouped 1x1 convolution is a dense one with a block-diagonal weight.  When the per-group bottleneck is tiny
  // (AVQA: 4 groups, bottleneck 12 -> 3 per group) padding every GROUP to the 16-entry granule of the register-resident kernels
  // multiplies the bottleneck-space traffic (4 x 16 = 64 entries for 12 real ones); the whole bottleneck padded once (16) is 4 x
  // smaller.  Such sites run as ONE group on block-diagonal copies of down_sampler / up_sampler (zeros outside the blocks are
  // exact, so every statistic and gradient is unchanged; gradients of the zero entries are discarded: moe_forward / moe_backward).
  // Only where the ungrouped products still fit the streaming GEMMs (C <= 384), and not for the shapes tile_fast.hip serves.
  bool try_merge = false;
  {
    const int merged = (int)round_up(d.d, 16), dg0 = d.d / g_site, grouped = g_site * (int)round_up(dg0, 16), ncg = merged / 16;
    const bool fast_shape = g_site == 2 && dg0 > 16 && dg0 <= 32 && d.K == 32 && d.E >= 2 && d.E <= 4;
    try_merge = !no_gen && !dev_env("AVMOE_NO_MERGE") && g_site > 1 && d.C <= 384 && merged < grouped && (ncg <= 4 || ncg == 6) && !fast_shape;
  }
  for (int attempt = try_merge ? 0 : 1; attempt < 2; ++attempt) {
    const bool merge = attempt == 0;
    d.mg = merge ? g_site : 0; d.mdg = merge ? d.d / g_site : 0; d.g = merge ? 1 : g_site;
    d.dg = d.d / d.g;
    d.dgp = (int)round_up(d.dg, 8);
    // A per-group bottleneck below 32 is padded to 32 when that makes the site the register-resident shape (2 groups, 32 latent
    // tokens, 2 - 4 experts: tile_fast.hip): bf16 Z rows of 64 E entries instead of fp32 rows of 4 * g * E * dgp bytes, and kernels
    // that run 1.3-3x faster than the generic ones (HTS-AT / Swin-B sites at r = 8: bottleneck 48: -29 %, 32: -15 %, 24: -19 %,
    // 12 / 16: -4 % of the site step).  Padding columns are zero weights, as for every other padded width.
    // (Per-group bottlenecks up to 16 stay at 16 entries and run on the generalised kernels of tile_gen.hip instead -- half the Z-space bytes.)
    if (d.g == 2 && d.dg < 32 && (d.dg > 16 || no_gen) && d.K == 32 && d.E >= 2 && d.E <= 4 && !dev_env("AVMOE_NO_PAD32")) d.dgp = 32;
    d.Cg = d.C / d.g;
    d.DD = d.g * d.dgp;
    d.DZ = d.E * d.DD;
    d.Kp = (int)round_up(d.K, 8);
    // Generalised register-resident kernels (tile_gen.hip) for every other shape they are built for: bottleneck entries per group
    // padded to a multiple of 16, latent-token slots to 16 / 32 / 96 (zero weights / masked slots, as for every padded width).
    d.gen = 0;
    static const bool no_fast = dev_env("AVMOE_NO_FAST") != nullptr;      // development: the generalised kernels at the tuned shape too (A/B)
    if (!no_gen && (no_fast || !(d.g == 2 && d.dgp == 32 && d.K == 32 && d.E >= 2 && d.E <= 4))) {
      Dims t = d;
      t.dgp = (int)round_up(d.dg, 16);
      if (d.El > 0) t.Kp = d.K <= 16 ? 16 : (d.K <= 32 ? 32 : (d.K <= 96 ? 96 : (int)round_up(d.K, 16)));
      if (tile_gen_ok(t) && t.g * t.dgp <= 256) { d.dgp = t.dgp; d.Kp = t.Kp; d.DD = d.g * d.dgp; d.DZ = d.E * d.DD; d.gen = 1; }
    }
    if (!merge || d.gen) break;          // a merged site must land on the generalised kernels; otherwise plan it grouped
  }
  if (d.DD > 256) { set_last_error("moe: padded bottleneck %d > 256", d.DD); return ERR_UNSUPPORTED; }
  d.KL = d.El * d.Kp;          // latent rows per sample, each slot padded to Kp rows (pad rows are zero)
  d.KLT = d.KL + 2;
  d.KLp = (int)round_up(d.KL + 2, 8);
  d.aL = (long)d.NT * d.Kp;
  d.Kcy = d.Ey * d.Kp;
  d.Kcyb = d.Kcy + 1;
  d.Kcx = d.Ex * d.Kp;
  d.Kcyp = (int)round_up(d.Kcy > 0 ? d.Kcy : 1, 8);
  d.Kcxp = (int)round_up(d.Kcx > 0 ? d.Kcx : 1, 8);
  d.KP = d.E * d.dgp + 3 * d.E;
  d.KPp = (int)round_up(d.KP, 8);
  d.XW = (int)round_up(d.KPp - d.E * d.dgp, 16);
  // AVVP N x N block: softmax / its gradient are (frames, N, N) tensors (the scores and d att are fused into GEMM epilogues).  While all frames fit in a fraction of the
  // 256 MiB Infinity Cache budget they are formed once and kept for the backward; beyond that the block runs a few frames at a
  // time through one workspace of at most 2 GiB (forward AND backward, which then recomputes scores + softmax; a cache-sized 96 MiB
  // workspace -- one 4096-token frame per chunk, split-K in every product -- ran 27 % slower at cfg-3): no (S, N, N) tensor in
  // HBM -- at AVVP stage 0 (N = 4096, 640 frames) that would be 21 GB (bf16) + 43 GB (fp32 scores) per site.
  d.nxc = d.S;
  d.nflash = d.nxn && !d.mha && nxn_att_ok(d.bf16, d.N, d.C, (int)round_up(d.N, 8)) && !dev_env("AVMOE_NXN_OLD_BWD");
  if (d.nflash) {
    if (const int ch = test_hook_nxn_chunk()) d.nxc = std::max(1, std::min(d.S, ch));     // tests (avmoe_test_hooks): the frame loop of the strip kernels
  } else if (d.nxn && !d.mha) {
    const size_t per_frame = (size_t)d.N * round_up(d.N, 8) * (2 * (size_t)d.esz);      // att + dSc (the scores / d att themselves never leave the chip)
    const size_t keep_all = (size_t)256 << 20;
    size_t budget = (size_t)2048 << 20;          // chunk workspace (scratch, reused by every site)
    if (const char* ev = dev_env("AVMOE_NXN_BUDGET_MB")) budget = (size_t)std::max(1, atoi(ev)) << 20;      // dev: sweep
    if ((size_t)d.S * per_frame > keep_all) d.nxc = (int)std::max<size_t>(1, std::min<size_t>((size_t)d.S, budget / per_frame));
    if (const int ch = test_hook_nxn_chunk()) d.nxc = std::max(1, std::min(d.S, ch));     // tests (avmoe_test_hooks): force the chunked path on small shapes
  }
  d.Mk = (int)round_up(d.M + 2, 8);
  d.Mb = (int)round_up(d.M + 1, 8);
  d.Np = (int)round_up(d.N, 8);
  // per-token kernels: blocks per sample so that the grid has a few waves per SIMD
  // (A/B at cfg-2 with the wave-per-expert kernels: 256 tokens per block at N = 1024, two blocks for the 196 tokens of the visual side)
  int bps = std::max(1, std::min(cdiv(d.N, 64), std::max(cdiv(d.N, d.N >= 512 ? 256 : 112), cdiv(256, d.S))));
  if (const char* ev = dev_env("AVMOE_BPS")) {     // development: "<bps for N >= 512>,<bps for N < 512>"
    int a = 0, b = 0;
    if (sscanf(ev, "%d,%d", &a, &b) == 2) bps = std::max(1, d.N >= 512 ? a : b);
  }
  d.nblk_tok = bps * d.S;
  d.zsz = (tile_fast_ok(d) || d.gen) ? d.esz : 4;     // Z / dz' in the activation type on the register-resident paths
  d.gram64 = tile_fast_shape(d) && d.bf16 && (d.E == 4 || d.E == 2) && !dev_env("AVMOE_NO_GRAM64");      // gram.hip is built for 2 and 4 experts
  d.xchunks = std::max(1, std::min(cdiv(d.N, 32), cdiv(4096, d.S)));
  d.fuse_xs = d.bf16 && d.zsz == 2 && gemm_stream_stats_ok(d.N, d.S, d.E * d.dgp, d.Cg, d.C, d.DZ) && !dev_env("AVMOE_NO_FUSE_XSTATS");
  // the hop-2 logits out of the same pass over X: the tuned shape (tile_fast.hip's pre_small adds the per-group partial sums), no latent
  // self attention (its latent tokens come from X itself), <= 64 latent rows per frame, the k384_n128 streaming configuration
  d.fuse_l2 = d.fuse_xs && tile_fast_ok(d) && !d.gen && d.El > 0 && d.Ex == 0 && d.KL <= 64 && d.KL % 16 == 0 && d.S >= 2 && d.E * d.dgp == 128 && d.Cg <= 384 &&
              d.Cg > 160 && !dev_env("AVMOE_NO_FUSE_L2");

  size_t off[2] = {0, 0};
  int n = 0;
#define X(name, region, eb, cnt)                                                   \
  {                                                                                \
    const size_t bytes = (size_t)(eb) * (size_t)(cnt);                             \
    pl->o_##name = off[region];                                                    \
    pl->info[n++] = BufInfo{#name, region, off[region], bytes};                    \
    off[region] += (size_t)round_up((long)bytes, 256);                             \
  }
  AVMOE_BUFFERS(X)
#undef X
  pl->nbuf = n;
  pl->saved_bytes = off[0] + 256;
  pl->scratch_bytes = off[1] + 256;
  return OK;
}

}  // namespace avmoe
