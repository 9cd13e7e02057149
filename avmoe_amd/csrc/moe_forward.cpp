// Forward orchestration of one MoEAdapter site: a fixed sequence of engine GEMMs and bottleneck-space
// kernels on the caller's stream (no allocation, no host sync: capturable in a hipGraph).
// Stage names follow oracle/algebra_ref.py::AlgebraRef.forward.
#include <cstdlib>
#include "moe_run.h"
#include "side.h"

namespace avmoe {

int choose_ksplit(const GemmArgs& g, size_t slab_floats_cap) {
  const int tile = (g.tile ? g.tile : ((g.M > 64 && g.N > 64) ? 128 : ((g.M <= 32 && g.N <= 32) ? 32 : 64)));
  const long nb = (long)g.nb1 * g.nb2 * (g.nb3 > 0 ? g.nb3 : 1);          // (every batch level has its own slabs: gemm_slab_bytes)
  const long tiles = (long)cdiv(g.M, tile) * cdiv(g.N, tile) * nb;
  const int bk = g.dtype == GEMM_BF16 ? 64 : 32;
  static const long target = dev_env("AVMOE_KS_TARGET") ? atol(dev_env("AVMOE_KS_TARGET")) : 512;     // workgroups wanted (dev override)
  long ks = std::max<long>(1, target / std::max<long>(tiles, 1));
  ks = std::min<long>(ks, std::max<long>(1, g.K / (4 * bk)));
  ks = std::min<long>(ks, 64);
  const size_t per = (size_t)nb * g.M * g.N;
  while (ks > 1 && per * ks > slab_floats_cap) --ks;
  return (int)ks;
}

// The "v1" experts run through the x + gate * xr code of the AVVP ones with gate 1: a device constant stands in for gate_av.
avmoe_moe_ptrs with_unit_gates(const Plan& pl, const avmoe_moe_ptrs& prm, char* sv) {
  avmoe_moe_ptrs p = prm;
  if (pl.d.mha)
    for (int e = 0; e < pl.d.E; ++e)
      if (pl.d.nxn_of_e[e]) p.e[e].gate_lat = (float*)(sv + pl.o_scal) + 1;
  return p;
}

int moe_forward(const Plan& pl, const void* X, const void* Y, const avmoe_moe_ptrs& prm_in, const float* noise, void* out,
                float* probs_out, int64_t* idx_out, float* lb_out, char* sv, char* sc, hipStream_t st) {
  const Dims& d = pl.d;
  avmoe_moe_ptrs prm = with_unit_gates(pl, prm_in, sv);
  if (d.mg) {                                              // merged groups: run on block-diagonal dense copies of the grouped weights
    AVMOE_TRY(k_merge_expand(pl, sv, prm_in, st));
    prm = merged_params(pl, prm, sv);
  }
  const int dt = d.bf16 ? GEMM_BF16 : GEMM_F32;
  float* slabs = (float*)(sc + pl.o_slabs);
  const size_t slab_cap = slab_floats(d);
  auto base = [&]() { GemmArgs g; g.dtype = dt; g.out_dtype = GEMM_F32; g.slabs = slabs; g.split3 = AVMOE_FWD_SPLIT3; return g; };      // (split3: fp32 sites only, gemm.h)
  const bool hop1s = d.bf16 && !dev_env("AVMOE_NO_HOP1S");    // the per-frame products against Y as streaming kernels (hop1_stream.hip)

  // ---- weights-derived operands --------------------------------------------------------------
  AVMOE_TRY(k_prep_all(pl, sv, prm, st));                  // (also scal[1] = 1: the unit gate of with_unit_gates)
  // ---- token statistics of X: row sums (LayerNorm), column means (router) ---------------------
  if (!d.fuse_xs) AVMOE_TRY(k_xstats(pl, X, sv, sc, st));          // (fused: they come out of the down projection below)
  // Zx = X Wt^T per frame, with the row sums / column sums of X as side products (fused statistics).  It depends on X and the
  // weights only, the hop-1 chain below on Y and the weights only: the two run on two streams and meet at the router.
  auto fused_down = [&](hipStream_t xs) -> int {
    GemmArgs g = base();
    int tiles = 0;
    g.A = X; g.B = sv + pl.o_Wt; g.C = sv + pl.o_Z; g.out_dtype = GEMM_BF16;
    g.M = d.N; g.N = d.E * d.dgp; g.K = d.Cg; g.lda = d.C; g.ldb = d.Cg; g.nb1 = d.S; g.nb2 = d.g;
    g.sA1 = (long)d.N * d.C; g.sA2 = d.Cg; g.sB2 = (long)d.E * d.dgp * d.Cg; g.sCi = d.DZ; g.sC1 = (long)d.N * d.DZ; g.sC2 = (long)d.E * d.dgp;
    g.st_rows = (float*)(sc + pl.o_sxp); g.st_cols = (float*)(sc + pl.o_xpart); g.st_ntot = d.NT; g.st_tiles = &tiles;
    if (d.fuse_l2) {                                       // + L2g[gi][s] = X[s][:, group gi] T[s][:, group gi]^T : the hop-2 logits, per group
      g.B3 = sv + pl.o_Text; g.N3 = d.KL; g.ldb3 = d.C; g.s3B1 = (long)d.KLT * d.C; g.s3B2 = d.Cg;
      g.C3 = (float*)(sc + pl.o_L2g); g.ldc3 = d.KL; g.s3C1 = (long)d.N * d.KL; g.s3C2 = (long)d.NT * d.KL;
    }
    const int rc = launch_gemm_stream(g, xs);
    if (rc != OK) {
      if (rc == 1) set_last_error("moe_forward: the streaming down projection with statistics does not serve this shape (plan / kernel mismatch)");
      return rc == 1 ? ERR_UNSUPPORTED : rc;
    }
    if ((2L * d.NT) % 4 == 0)      // the two finishing sums (row sums over the groups, column means over the tiles) in one launch
      return k_xstats_fin((const float*)(sc + pl.o_sxp), d.g, 2L * d.NT, (float*)(sv + pl.o_sx), (const float*)(sc + pl.o_xpart), tiles, d.C, d.S,
                          (float*)(sv + pl.o_rin), 2L * d.C, 1.f / (float)d.N, xs);
    AVMOE_TRY(k_sum_parts((const float*)(sc + pl.o_sxp), d.g, 2L * d.NT, (float*)(sv + pl.o_sx), xs));
    return k_colsum_f32((const float*)(sc + pl.o_xpart), tiles, d.C, d.C, d.S, (long)tiles * d.C, (float*)(sv + pl.o_rin), 2L * d.C,
                        1.f / (float)d.N, xs);
  };
  // (with the logits fused in, the pass needs the latent tokens T: it runs behind the hop-1 chain, on the caller's stream)
  Side* side = (d.fuse_xs && !d.fuse_l2 && (side_mask() & 1) && side_worth(d)) ? side_acquire(st) : nullptr;
  SideScope fk(side, st);
  if (side) {
    AVMOE_TRY(fk.fork());
    AVMOE_TRY(fused_down(side->s));
  }

  // ---- hop 1, cross-modal experts: latent tokens read the (never materialised) remapped Y -----
  if (d.Kcy > 0) {
    GemmArgs g = base();                                   // Q = T0 Wf
    g.A = sv + pl.o_T0T; g.B = sv + pl.o_WfT; g.C = sv + pl.o_Qx;
    g.M = d.Kcy; g.N = d.Cy; g.K = d.C; g.lda = d.C; g.b_layout = MN_MAJOR; g.ldb = d.Cy;
    g.sCi = d.Cy; g.out_dtype = dt;
    AVMOE_TRY(launch_gemm(g, st));
  }
  AVMOE_TRY(k_qrqb_fill(pl, sv, prm.fc_b, st));
  if (d.Kcy > 0) {
    {                                                      // R[s] = Q Y[s]^T
      int rc = 1;                                          // (hop1_stream.hip: Q stationary, Y streamed in whole token rows; 1 = shape not served)
      if (hop1s) rc = k_hop1_yk(Y, d.Cy, d.S, d.M, d.Cy, sv + pl.o_Qx, d.Cy, 0, d.Kcy, sv + pl.o_Rext, d.Mk, (long)d.Kcyb * d.Mk, 1, slabs, st);
      if (rc < 0) return rc;
      GemmArgs g = base();
      g.A = sv + pl.o_Qx; g.B = Y; g.C = sv + pl.o_Rext;
      g.M = d.Kcy; g.N = d.M; g.K = d.Cy; g.lda = d.Cy; g.ldb = d.Cy; g.nb1 = d.S; g.sB1 = (long)d.M * d.Cy;
      g.sCi = d.Mk; g.sC1 = (long)d.Kcyb * d.Mk; g.out_dtype = dt;
      if (rc != OK) AVMOE_TRY(launch_gemm(g, st));
    }
    {                                                      // L1[s] = [R | qr | qb] [Wc | bc | 1]^T
      GemmArgs g = base();
      g.A = sv + pl.o_Rext; g.B = sv + pl.o_WcK; g.C = sc + pl.o_L1;
      g.M = d.Kcy; g.N = d.N; g.K = d.M + 2; g.lda = d.Mk; g.ldb = d.Mk; g.nb1 = d.S; g.sA1 = (long)d.Kcyb * d.Mk;
      g.sCi = d.Np; g.sC1 = (long)d.Kcyb * d.Np;
      AVMOE_TRY(launch_gemm(g, st));
    }
    AVMOE_TRY(k_softmax_rows(d.bf16, (const float*)(sc + pl.o_L1), (long)d.S * d.Kcyb, d.N, d.Np, sv + pl.o_A1y, d.Np,
                             d.Kcyb, d.Kcy, d.Kp, d.K, st));
    {                                                      // [Bm | ab][s] = A1[s] [Wc | bc]
      GemmArgs g = base();
      g.A = sv + pl.o_A1y; g.B = sv + pl.o_WcT; g.C = sv + pl.o_BmX;
      g.M = d.Kcy; g.N = d.M + 1; g.K = d.N; g.lda = d.Np; g.ldb = d.Np; g.nb1 = d.S; g.sA1 = (long)d.Kcyb * d.Np;
      g.sCi = d.Mb; g.sC1 = (long)d.Kcyb * d.Mb; g.out_dtype = dt;
      AVMOE_TRY(launch_gemm(g, st));
    }
  }
  {                                                        // V[s] = [Bm ; wbar][s] Y[s]   (token contraction)
    int rc = 1;
    if (hop1s) rc = k_hop1_yt_frames(Y, d.Cy, d.S, d.M, d.Cy, sv + pl.o_BmX, d.Mb, (long)d.Kcyb * d.Mb, d.Kcyb, sv + pl.o_V, d.Cy, (long)d.Kcyb * d.Cy, 1, st);
    if (rc < 0) return rc;
    GemmArgs g = base();
    g.A = sv + pl.o_BmX; g.B = Y; g.C = sv + pl.o_V;
    g.M = d.Kcyb; g.N = d.Cy; g.K = d.M; g.lda = d.Mb; g.b_layout = MN_MAJOR; g.ldb = d.Cy; g.nb1 = d.S;
    g.sA1 = (long)d.Kcyb * d.Mb; g.sB1 = (long)d.M * d.Cy; g.sCi = d.Cy; g.sC1 = (long)d.Kcyb * d.Cy; g.out_dtype = dt;
    if (rc != OK) AVMOE_TRY(launch_gemm(g, st));
  }
  {                                                        // TV = V Wf^T
    GemmArgs g = base();
    g.A = sv + pl.o_V; g.B = sv + pl.o_WfT; g.C = sc + pl.o_TV;
    g.M = d.S * d.Kcyb; g.N = d.C; g.K = d.Cy; g.lda = d.Cy; g.ldb = d.Cy; g.sCi = d.C;
    AVMOE_TRY(launch_gemm(g, st));
  }
  AVMOE_TRY(k_finish_T(pl, sv, sc, prm, 0, st));

  // ---- hop 1, latent self attention on X (AVS v2) ----------------------------------------------
  if (d.Kcx > 0) {
    const char* T0x = sv + pl.o_T0T + (size_t)d.Kcy * d.C * d.esz;
    {
      GemmArgs g = base();
      g.A = T0x; g.B = X; g.C = sc + pl.o_L1;
      g.M = d.Kcx; g.N = d.N; g.K = d.C; g.lda = d.C; g.ldb = d.C; g.nb1 = d.S; g.sB1 = (long)d.N * d.C;
      g.sCi = d.Np; g.sC1 = (long)d.Kcx * d.Np;
      AVMOE_TRY(launch_gemm(g, st));
    }
    AVMOE_TRY(k_softmax_rows(d.bf16, (const float*)(sc + pl.o_L1), (long)d.S * d.Kcx, d.N, d.Np, sv + pl.o_A1x, d.Np,
                             d.Kcx, d.Kcx, d.Kp, d.K, st));
    {
      GemmArgs g = base();
      g.A = sv + pl.o_A1x; g.B = X; g.C = sc + pl.o_TV;
      g.M = d.Kcx; g.N = d.C; g.K = d.N; g.lda = d.Np; g.b_layout = MN_MAJOR; g.ldb = d.C; g.nb1 = d.S;
      g.sA1 = (long)d.Kcx * d.Np; g.sB1 = (long)d.N * d.C; g.sCi = d.C; g.sC1 = (long)d.Kcx * d.C;
      AVMOE_TRY(launch_gemm(g, st));
    }
    AVMOE_TRY(k_finish_T(pl, sv, sc, prm, 1, st));
  }

  // ---- per-sample K-space matrices of the latent tokens ------------------------------------------
  // (row sums of Text: k_finish_T writes them with the rows, k_prep_all those of the constant rows)
  if (d.El > 0) {                                          // TT[s][l] = T T^T
    GemmArgs g = base();
    g.A = sv + pl.o_Text; g.B = sv + pl.o_Text; g.C = sv + pl.o_TT;
    g.M = d.K; g.N = d.K; g.K = d.C; g.lda = d.C; g.ldb = d.C; g.nb1 = d.S; g.nb2 = d.El;
    g.sA1 = g.sB1 = (long)d.KLT * d.C; g.sA2 = g.sB2 = (long)d.Kp * d.C;
    g.sCi = d.K; g.sC1 = (long)d.El * d.K * d.K; g.sC2 = (long)d.K * d.K;
    AVMOE_TRY(launch_gemm(g, st));
  }
  auto down_gemm = [&](const void* rows, long nrows, void* dst, int out_dt = GEMM_F32) {   // rows (nrows, C) -> (nrows, DZ) through Wt
    GemmArgs g = base();
    g.A = rows; g.B = sv + pl.o_Wt; g.C = dst; g.out_dtype = out_dt;
    g.M = (int)nrows; g.N = d.E * d.dgp; g.K = d.Cg; g.lda = d.C; g.ldb = d.Cg; g.nb2 = d.g;
    g.sA2 = d.Cg; g.sB2 = (long)d.E * d.dgp * d.Cg; g.sCi = d.DZ; g.sC2 = (long)d.E * d.dgp;
    return launch_gemm(g, st);
  };
  AVMOE_TRY(down_gemm(sv + pl.o_Text, (long)d.S * d.KLT, sv + pl.o_TW));   // TW (all latent rows x all experts)
  // ---- the X-side GEMMs ------------------------------------------------------------------------
  if (d.fuse_xs) {
    if (side) AVMOE_TRY(fk.join());
    else AVMOE_TRY(fused_down(st));
  } else {
    AVMOE_TRY(down_gemm(X, d.NT, sv + pl.o_Z, d.zsz == 2 ? GEMM_BF16 : GEMM_F32));                              // Zx = X Wt^T
  }
  // ---- router (its input: the token means of X -- after the statistics above) ----------------------
  AVMOE_TRY(k_router(pl, sv, sc, prm, noise, probs_out, idx_out, lb_out, st));
  if (d.KL > 0 && !d.fuse_l2) {                            // L2[s] = X[s] T[s]^T   (fused: per-group partial sums came out of the down projection; pre_small adds them into L2)
    GemmArgs g = base();
    g.A = X; g.B = sv + pl.o_Text; g.C = sv + pl.o_L2;
    g.M = d.N; g.N = d.KL; g.K = d.C; g.lda = d.C; g.ldb = d.C; g.nb1 = d.S;
    g.sA1 = (long)d.N * d.C; g.sB1 = (long)d.KLT * d.C; g.sCi = d.KL; g.sC1 = (long)d.N * d.KL;
    AVMOE_TRY(launch_gemm(g, st));
  }
  // ---- AVVP unimodal N x N block (mgn.py:132-139): xr = softmax_rows(X X^T)^T X, shared by the unimodal experts;
  //      their input x + gate_av * xr enters the LN-folded projection through ZR = xr Wt^T and three row sums ----
  if (d.mha) {                                             // AVS "v1": one xr = MHA_e(X) - X per unimodal expert, through that expert's Wt rows
    for (int e = 0; e < d.E; ++e) {
      if (!d.nxn_of_e[e]) continue;
      const int slot = d.xr_of_e[e];
      AVMOE_TRY(mha_frames_forward(pl, X, prm.e[e], slot, sv, sc, st));
      AVMOE_TRY(k_xrstats(pl, X, sv, slot, st));
      GemmArgs g = base();                                 // ZR[:, expert e] = xr Wt_e^T
      g.A = sv + pl.o_xr + (size_t)slot * d.NT * d.C * d.esz; g.B = sv + pl.o_Wt + (size_t)e * d.dgp * d.Cg * d.esz;
      g.C = sv + pl.o_ZR + (size_t)e * d.dgp * 4;
      g.M = d.NT; g.N = d.dgp; g.K = d.Cg; g.lda = d.C; g.ldb = d.Cg; g.nb2 = d.g;
      g.sA2 = d.Cg; g.sB2 = (long)d.E * d.dgp * d.Cg; g.sCi = d.DZ; g.sC2 = (long)d.E * d.dgp;
      AVMOE_TRY(launch_gemm(g, st));
    }
  } else if (d.nxn) {
    for (int s0 = 0; s0 < d.S; s0 += d.nxc) {              // d.nxc frames at a time through one workspace (all of them when they fit: moe_plan.cpp)
      const int ns = std::min(d.nxc, d.S - s0);
      const char* Xc = (const char*)X + (size_t)s0 * d.N * d.C * d.esz;
      // att[s] = softmax_rows(X[s] X[s]^T) without the scores leaving the chip: the product runs twice with softmax epilogues -- (max, sum exp)
      // per row and column tile, then exp(score - lse) straight to att; the row log-sum-exp is kept for the backward
      float* lse = (float*)(sv + pl.o_nlse) + (size_t)s0 * d.N;
      if (d.nflash) {     // the large-N sites: row statistics, then xr = att^T X with att re-formed in the accumulators (nxn_att.hip)
        AVMOE_TRY(k_nxn_att(Xc, ns, d.N, d.C, d.Np, lse, nullptr, 0, st));
        AVMOE_TRY(k_nxn_xr(Xc, ns, d.N, d.C, d.Np, lse, sv + pl.o_xr + (size_t)s0 * d.N * d.C * d.esz, st));
        continue;
      }
      if (nxn_att_ok(d.bf16, d.N, d.C, d.Np)) {                 // ... as one kernel for the large-N sites (nxn_att.hip)
        AVMOE_TRY(k_nxn_att(Xc, ns, d.N, d.C, d.Np, lse, sv + pl.o_att, 0, st));
      } else
      for (int pass = 0; pass < 2; ++pass) {
        GemmArgs g = base();
        g.A = Xc; g.B = Xc; g.C = sv + pl.o_att;
        g.M = d.N; g.N = d.N; g.K = d.C; g.lda = d.C; g.ldb = d.C; g.nb1 = ns; g.sA1 = g.sB1 = (long)d.N * d.C;
        g.sCi = d.Np; g.sC1 = (long)d.N * d.Np; g.out_dtype = dt;
        if (pass == 0) { g.epi = GEMM_EPI_ROWSTATS; g.row_part = (float*)(sc + pl.o_npart); }
        else { g.epi = GEMM_EPI_EXP; g.row_lse = lse; }
        AVMOE_TRY(launch_gemm(g, st));
        if (pass == 0) AVMOE_TRY(gemm_row_lse((const float*)(sc + pl.o_npart), (long)ns * d.N, cdiv(d.N, 128), lse, st));
      }
      {                                                    // xr[s] = att[s]^T X[s]
        GemmArgs g = base();
        g.A = sv + pl.o_att; g.B = Xc; g.C = sv + pl.o_xr + (size_t)s0 * d.N * d.C * d.esz;
        g.M = d.N; g.N = d.C; g.K = d.N; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.Np; g.ldb = d.C; g.nb1 = ns;
        g.sA1 = (long)d.N * d.Np; g.sB1 = (long)d.N * d.C; g.sCi = d.C; g.sC1 = (long)d.N * d.C; g.out_dtype = dt;
        g.ksplit = choose_ksplit(g, slab_cap);             // few frames x few column tiles: split the token contraction to fill the chip
        AVMOE_TRY(launch_gemm(g, st));
      }
    }
    AVMOE_TRY(k_xrstats(pl, X, sv, 0, st));
    AVMOE_TRY(down_gemm(sv + pl.o_xr, d.NT, sv + pl.o_ZR));
  }
  // ---- bottleneck space --------------------------------------------------------------------------
  AVMOE_TRY(k_pre_small(pl, sv, sc, prm, st));
  AVMOE_TRY(k_bn1_finalize(pl, sv, sc, prm, st));
  if (!d.gram64) AVMOE_TRY(k_mid(pl, sv, sc, st));
  if (d.use_bn && d.training && d.gram64) {
    // No MID pass and no stored z' on this path: ONE streaming pass over z forms z' = act(BN1(z)) on the way into LDS and leaves
    // both BatchNorm-2 moments -- Szz[i][e] = z'^T z' / NT and the column means mz (every later kernel recomputes z' from z anyway)
    AVMOE_TRY(k_gram64(pl, sv + pl.o_Z, nullptr, 1.f / (float)d.NT, (float*)(sc + pl.o_gpartT), (float*)(sv + pl.o_Szz), st,
                       (const float*)(sv + pl.o_bn1), (float*)(sc + pl.o_gcolT), (float*)(sv + pl.o_mz)));
  } else if (d.use_bn && d.training) {                     // ... as 8 batched token contractions of the engine
    GemmArgs g = base();
    g.A = sc + pl.o_Zp; g.B = sc + pl.o_Zp; g.C = sv + pl.o_Szz;
    g.M = d.dgp; g.N = d.dgp; g.K = d.NT; g.a_layout = g.b_layout = MN_MAJOR; g.lda = g.ldb = d.DZ;
    g.nb2 = d.g * d.E; g.sA2 = g.sB2 = d.dgp; g.sCi = d.dgp; g.sC2 = (long)d.dgp * d.dgp;
    g.alpha = 1.f / (float)d.NT;
    g.ksplit = choose_ksplit(g, slab_cap);
    AVMOE_TRY(launch_gemm(g, st));
  }
  AVMOE_TRY(k_post_prep(pl, sv, sc, prm, st));
  AVMOE_TRY(k_post_small(pl, sv, sc, prm, st));
  {                                                        // out = Apost Bpost^T  (mixture, gates, BN2, LN-post folded in)
    GemmArgs g = base();
    g.A = sv + pl.o_Apost; g.B = sv + pl.o_Bpost; g.C = out;
    g.M = d.NT; g.N = d.Cg; g.K = d.KP; g.lda = (long)d.g * d.KPp; g.ldb = d.KPp; g.nb2 = d.g;
    g.sA2 = d.KPp; g.sB2 = (long)d.Cg * d.KPp; g.sCi = d.C; g.sC2 = d.Cg; g.out_dtype = dt;
    g.accumulate = d.acc_out;
    AVMOE_TRY(launch_gemm(g, st));
  }
  return OK;
}

// ---- sub-ops of the C ABI (for tests / partial adoption; the product path never calls them) -------------------------
// What ExpertAdapter.forward of expert e returns -- gate * LN_post(BN2(up(act(BN1(down(LN_before(x'))))))), net_trans_v3.py:377-435 --
// as the site forward with the router pushed to an exact one-hot on that expert: a logit offset of 3e4 makes softmax return 1.0 for it
// and 0.0 for the others (exp underflows to zero), so out = 1 * out_e + 0 * out_others.  Same kernels and workspaces as
// avmoe_moe_forward.  A NON-FINITE value in another expert's output still propagates (0 * Inf = NaN), exactly as it does in the
// reference's own mixture (net_trans_v3.py:485-486) -- compare modules on finite parameters.
int expert_forward(const Plan& pl, const void* X, const void* Y, const avmoe_moe_ptrs& prm, int e, void* out, char* sv, char* sc, hipStream_t st) {
  const Dims& d = pl.d;
  if (e < 0 || e >= d.E) { set_last_error("expert_forward: expert %d of %d", e, d.E); return ERR_BAD_ARG; }
  float* noise = (float*)(sc + pl.o_dp);                   // (S, E) floats of a backward-only buffer
  AVMOE_TRY(k_onehot_noise(noise, d.S, d.E, e, 3.0e4f, st));
  // Training mode: only the SELECTED expert's BatchNorm running statistics and counters advance (what ExpertAdapter.forward of that
  // one module does).  The other experts still run -- their outputs enter the mixture with weight exactly 0 -- but their
  // running_mean / running_var / num_batches_tracked updates go to a write-only dump in a backward-only scratch buffer (training
  // mode never READS running statistics), so comparing a site module by module does not advance every module E times.
  avmoe_moe_ptrs q = prm;
  if (d.training && d.use_bn) {
    float* dump = (float*)(sc + pl.o_dWf);               // C x Cy floats >= max(C, d) ; 256-byte aligned
    for (int o = 0; o < d.E; ++o) {
      if (o == e) continue;
      q.e[o].bn1_rm = q.e[o].bn1_rv = q.e[o].bn2_rm = q.e[o].bn2_rv = dump;
      q.e[o].bn1_nbt = q.e[o].bn2_nbt = nullptr;
    }
  }
  return moe_forward(pl, X, Y, q, noise, out, nullptr, nullptr, nullptr, sv, sc, st);
}

// The remap MATERIALISED (the product path folds it away): Yt = conv_adapter(Y) (S, N, Cy), then
// Yf = fc(Yt) (S, N, C) -- `vis_token` of net_trans_v3.py:469-471, token-major.  Two engine GEMMs on the T-typed weight copies of
// k_prep_all + the two biases.
int remap_forward(const Plan& pl, const void* Y, const avmoe_moe_ptrs& prm, void* Yt, void* Yf, char* sv, char* sc, hipStream_t st) {
  const Dims& d = pl.d;
  if (!prm.fc_b) { set_last_error("remap_forward: fc.bias missing"); return ERR_BAD_ARG; }
  const int dt = d.bf16 ? GEMM_BF16 : GEMM_F32;
  AVMOE_TRY(k_prep_all(pl, sv, prm, st));
  {                                                        // Yt[s] = Wc Y[s]
    GemmArgs g; g.dtype = dt; g.out_dtype = dt; g.slabs = (float*)(sc + pl.o_slabs);
    g.A = sv + pl.o_WcK; g.B = Y; g.C = Yt;
    g.M = d.N; g.N = d.Cy; g.K = d.M; g.lda = d.Mk; g.b_layout = MN_MAJOR; g.ldb = d.Cy; g.nb1 = d.S; g.sB1 = (long)d.M * d.Cy;
    g.sCi = d.Cy; g.sC1 = (long)d.N * d.Cy;
    AVMOE_TRY(launch_gemm(g, st));
  }
  AVMOE_TRY(k_add_bias(d.bf16, Yt, (long)d.S * d.N, d.Cy, d.N, prm.conv_b, nullptr, st));
  {                                                        // Yf = Yt Wf^T
    GemmArgs g; g.dtype = dt; g.out_dtype = dt; g.slabs = (float*)(sc + pl.o_slabs);
    g.A = Yt; g.B = sv + pl.o_WfT; g.C = Yf;
    g.M = d.S * d.N; g.N = d.C; g.K = d.Cy; g.lda = d.Cy; g.ldb = d.Cy; g.sCi = d.C;
    AVMOE_TRY(launch_gemm(g, st));
  }
  return k_add_bias(d.bf16, Yf, (long)d.S * d.N, d.C, 1, nullptr, prm.fc_b, st);
}

}  // namespace avmoe
