// Backward orchestration of one MoEAdapter site.  Mirrors oracle/algebra_ref.py::AlgebraRef.backward:
// phase 1 the two GEMMs against dOut, phases 2-4 bottleneck / weight space, phase 5 GEMMs against X,
// phase 6 the hop-1 (latent token) chain back to Y and the remap parameters.  Stream-ordered, no
// allocation, no host sync.
#include <cstdlib>
#include "moe_run.h"
#include "side.h"

namespace avmoe {

#define MEMSET0(ptr, bytes)                                                                 \
  do {                                                                                      \
    hipError_t e__ = hipMemsetAsync((ptr), 0, (bytes), st);                                 \
    if (e__ != hipSuccess) { set_last_error("memset: %s", hipGetErrorString(e__)); return ERR_LAUNCH; } \
  } while (0)

avmoe_moe_ptrs with_unit_gates(const Plan& pl, const avmoe_moe_ptrs& prm, char* sv);   // moe_forward.cpp

// parts: bit mask of the sections to run, 0 = 7 = the whole backward:
//   1  phases 1-4: the GEMMs against dOut and the bottleneck / weight space (touches neither dX nor dY)
//   2  phase 5: the GEMMs against X -- every writer of dX ;  in two steps (plain sites only): 32 = phase 5 without the dX product
//      (touches neither dX nor dY), 64 = the dX product alone -- or moe_backward_dx_dy below in its place
//   4  phase 6: the hop-1 chain back to Y and the remap parameters -- every writer of dY;  in two steps: 8 = phase 6 without the
//      GEMM(s) that write dY (touches neither dX nor dY), 16 = those GEMMs alone (after 8)
// Sections are stream-ordered through `scratch`: a caller may put event records / waits between them (AdapterPair orders the
// two sites' accumulations into the shared token gradients this way) but nothing that touches the workspaces.
// the operands of  dX[s] = dZx[s] Wt + [dL2 | dsx | 1][s] [T ; 1 ; dm1/N][s] + 2 dSxx X[s]  (phase 5; moe_backward_dx_dy adds the other site's dY to it)
static void fill_dx_args(GemmArgs& g, const Plan& pl, const void* X, const char* sv, const char* sc, void* dX) {
  const Dims& d = pl.d;
  g.A = sc + pl.o_Zw; g.B = sv + pl.o_Wt; g.C = dX;
  g.M = d.N; g.N = d.Cg; g.K = d.E * d.dgp; g.lda = d.DZ; g.b_layout = MN_MAJOR; g.ldb = d.Cg; g.nb1 = d.S; g.nb2 = d.g;
  g.sA1 = (long)d.N * d.DZ; g.sA2 = (long)d.E * d.dgp; g.sB2 = (long)d.E * d.dgp * d.Cg;
  g.sCi = d.C; g.sC1 = (long)d.N * d.C; g.sC2 = d.Cg; g.out_dtype = d.bf16 ? GEMM_BF16 : GEMM_F32;
  g.row_scale = (const float*)(sc + pl.o_rs2x); g.sRS1 = d.N; g.D = X; g.sDi = d.C; g.sD1 = (long)d.N * d.C; g.sD2 = d.Cg;
  g.A2 = sc + pl.o_dL2x; g.B2 = sv + pl.o_Text; g.K2 = d.KLT; g.lda2 = d.KLp; g.ldb2 = d.C;
  g.s2A1 = (long)d.N * d.KLp; g.s2B1 = (long)d.KLT * d.C; g.s2B2 = d.Cg;
}

int moe_backward(const Plan& pl, const void* X, const void* Y, const avmoe_moe_ptrs& prm_in, const void* dOut, const float* lb_grad,
                 char* sv, char* sc, void* dX, void* dY, const avmoe_moe_ptrs& grads_in, hipStream_t st, int parts) {
  const Dims& d = pl.d;
  if (parts == 0) parts = 7;
  if ((parts & (32 | 64)) && (d.Kcx > 0 || d.mha || d.nxn)) { set_last_error("split backward: sections 32 / 64 serve plain sites only (this one goes on accumulating into dX)"); return ERR_UNSUPPORTED; }
  if ((parts & 2) && (parts & (32 | 64))) { set_last_error("split backward: section 2 IS sections 32 + 64"); return ERR_BAD_ARG; }
  if ((parts & 7) != 7 && d.Kcx > 0) { set_last_error("split backward: sites with latent self attention write dX in the last section"); return ERR_UNSUPPORTED; }
  avmoe_moe_ptrs prm = with_unit_gates(pl, prm_in, sv);
  avmoe_moe_ptrs grads = grads_in;
  if (d.mg) { prm = merged_params(pl, prm, sv); grads = merged_grads(pl, grads_in, sc); }     // the forward left the dense copies in `saved`
  const int dt = d.bf16 ? GEMM_BF16 : GEMM_F32;
  float* slabs = (float*)(sc + pl.o_slabs);
  const size_t slab_cap = slab_floats(d);
  auto base = [&]() { GemmArgs g; g.dtype = dt; g.out_dtype = GEMM_F32; g.slabs = slabs; g.split3 = AVMOE_BWD_PLANES; return g; };      // (split3: fp32 sites only, gemm.h; moe_run.h: two planes in the backward)
  auto run_on = [&](GemmArgs& g, bool split, hipStream_t on) {
    if (split) g.ksplit = choose_ksplit(g, slab_cap);
    return launch_gemm(g, on);
  };
  auto run = [&](GemmArgs& g, bool split) { return run_on(g, split, st); };
  // fp32 sites: products whose result is a parameter gradient or a token gradient (nothing downstream forms a cancelling sum from it) in the
  // TWO-plane form (gemm.hip: f32s2; moe_run.h: AVMOE_LEAF2 has the classes and the measurement).  bf16 sites ignore split3.
  static const int leaf2 = dev_env("AVMOE_LEAF2") ? atoi(dev_env("AVMOE_LEAF2")) : AVMOE_LEAF2;
  auto leaf = [&](GemmArgs& g, int cls) { if (leaf2 & cls) g.split3 = 2; };
  const bool hop1s = d.bf16 && !dev_env("AVMOE_NO_HOP1S");    // the per-frame products against Y as streaming kernels (hop1_stream.hip)
  // Independent branches run on a helper stream (side.h) and are joined before their first consumer and before the section ends:
  //   section 1: dBpost = dOut^T Apost (+ its split-K reduce; the only user of the slabs until the join) beside dApost -> post_small_bwd -> Gram
  //   section 2: the dX GEMM (nothing in this call reads dX) beside the dWt / dT chain
  Side* side = side_worth(d) ? side_acquire(st) : nullptr;
  SideScope fk1(side, st), fk2(side, st);                  // joined on every way out (error returns included)
  const size_t esz = d.esz;
  // development builds: AVMOE_BWD_STOP=n returns after the n-th step of section 1 (1 the two GEMMs against dOut, 2 post_small_bwd, 3 the
  // Gram products, 4 post_prep_bwd, 5 mid_bwd, 6 router_bwd) -- workspaces as that step left them (tests/dev/race_buffers.py)
  static const int bwd_stop = dev_env("AVMOE_BWD_STOP") ? atoi(dev_env("AVMOE_BWD_STOP")) : 0;
#define BWD_STOP(n) do { if (bwd_stop == (n)) return OK; } while (0)
  if (parts & 1) {   // =============================== section 1: phases 1 - 4 ===============================
  // the accumulators that start from zero (dtbp, dTW, dWcK, dqp, dRT) are adjacent in the plan: one memset instead of five
  MEMSET0(sc + pl.o_dtbp, (pl.o_dRT - pl.o_dtbp) + (size_t)d.S * d.M * d.Kcyp * esz);

  // ---- phase 1: dApost = dOut Bpost ; dBpost = dOut^T Apost -------------------------------------
  // (the engine's Gram path splits K itself: it needs the slabs; sites on the generalised kernels keep post_small_bwd off the GPU while
  // a GEMM of the helper stream runs -- tile_gen.inc::gen_lds_request has the reason)
  const bool fork1 = side && (side_mask() & 2) && (d.gram64 || !d.ln_post) && !d.gen;
  int dap16 = 0;     // dApost stored as [E x 32 bottleneck columns in T | 3 E scalar columns in fp32 (dApx)]: the register-resident bf16 path
  bool pair1 = false;                                      // both products from one pass over dOut (dpost_pair.hip: the tuned bf16 shape)
  if (d.zsz == 2 && d.bf16 && !dev_env("AVMOE_NO_DPAIR")) {
    const int rc = k_dpost_pair(dOut, d.C, sv + pl.o_Bpost, d.KPp, (long)d.Cg * d.KPp, sv + pl.o_Apost, (long)d.g * d.KPp, sc + pl.o_dAp, (long)d.g * d.E * d.dgp,
                                (float*)(sc + pl.o_dApx), (long)d.g * d.XW, d.XW, (float*)(sc + pl.o_dBp), d.NT, d.g, d.Cg, d.E * d.dgp, d.KP, d.KPp,
                                slabs, slab_cap, st);
    if (rc < 0) return rc;
    pair1 = rc == OK;
    if (pair1) dap16 = 1;
  }
  if (!pair1) {
    GemmArgs g = base();
    g.A = dOut; g.B = sv + pl.o_Apost; g.C = sc + pl.o_dBp;
    g.M = d.Cg; g.N = d.KP; g.K = d.NT; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.C; g.ldb = (long)d.g * d.KPp; g.nb2 = d.g;
    g.sA2 = d.Cg; g.sB2 = d.KPp; g.sCi = d.KPp; g.sC2 = (long)d.Cg * d.KPp;
    leaf(g, 1);
    if (fork1) AVMOE_TRY(fk1.fork());
    AVMOE_TRY(run_on(g, true, fork1 ? side->s : st));
  }
  if (!pair1) {
    GemmArgs g = base();
    g.A = dOut; g.B = sv + pl.o_Bpost; g.C = sc + pl.o_dAp;
    // N = KP, or the padded KPp when KP is not a multiple of 4 (2 or 3 experts): Bpost's padding columns are zero, the streaming
    // kernel wants whole 4-column vectors, and nobody reads the padding of dAp
    g.M = d.NT; g.N = (d.KP % 4 == 0) ? d.KP : d.KPp; g.K = d.Cg; g.lda = d.C; g.b_layout = MN_MAJOR; g.ldb = d.KPp; g.nb2 = d.g;
    g.sA2 = d.Cg; g.sB2 = (long)d.Cg * d.KPp; g.sCi = (long)d.g * d.KPp; g.sC2 = d.KPp;
    static const bool dap_f32 = dev_env("AVMOE_DAP_F32") != nullptr;      // dev switch
    if (d.zsz == 2 && !dap_f32) {          // bf16 main columns + fp32 scalar columns: 640 instead of 1152 bytes per token, written and re-read
      GemmArgs h = g;
      h.out_dtype = GEMM_BF16; h.Cx = (float*)(sc + pl.o_dApx); h.nsplit = d.E * d.dgp; h.ldcx = (long)d.g * d.XW; h.sCx2 = d.XW;
      h.sCi = (long)d.g * h.nsplit; h.sC2 = h.nsplit;      // the T columns as rows of their own (E * dgp wide: whole 128-byte lines), not inside KPp-wide rows
      const int r = launch_gemm_stream(h, st);
      if (r < 0) return r;
      dap16 = r == 0;
    }
    if (!dap16) AVMOE_TRY(run(g, false));
  }
  BWD_STOP(1);
  // ---- phase 2: bottleneck space (LayerNorm-post statistics), then weight space ------------------
  AVMOE_TRY(k_post_small_bwd(pl, sv, sc, prm, grads, st, dap16));
  BWD_STOP(2);
  if (d.ln_post && d.gram64 && tile_fast_ok(d) && kfs_serves_post_small_bwd(d, dap16)) {
    // (dGq came out of post_small_bwd's own pass: tile_stream.hip)
  } else if (d.ln_post && d.gram64) {                      // dG[i][e] = sum_t dSoo z' z'^T : one streaming pass over z (z' formed on the fly)
    AVMOE_TRY(k_gram64(pl, sv + pl.o_Z, (const float*)(sc + pl.o_dSooT), 1.f, (float*)(sc + pl.o_gpartT), (float*)(sc + pl.o_dGq), st,
                       (const float*)(sv + pl.o_bn1)));
  } else if (d.ln_post) {                                  // ... as batched engine GEMMs on Zw = dSoo z'
    GemmArgs g = base();
    g.A = sc + pl.o_Zw; g.B = sc + pl.o_Zp; g.C = sc + pl.o_dGq;
    g.M = d.dgp; g.N = d.dgp; g.K = d.NT; g.a_layout = g.b_layout = MN_MAJOR; g.lda = g.ldb = d.DZ;
    g.nb2 = d.g * d.E; g.sA2 = g.sB2 = d.dgp; g.sCi = d.dgp; g.sC2 = (long)d.dgp * d.dgp;
    AVMOE_TRY(run(g, true));
  }
  if (fork1 && !pair1) AVMOE_TRY(fk1.join());
  BWD_STOP(3);
  AVMOE_TRY(k_post_prep_bwd(pl, sv, sc, prm, grads, st));
  BWD_STOP(4);
  // ---- phase 3: ReLU / BN1 ; router --------------------------------------------------------------
  AVMOE_TRY(k_mid_bwd(pl, sv, sc, prm, grads, st));
  BWD_STOP(5);
  AVMOE_TRY(k_router_bwd(pl, sv, sc, prm, grads, lb_grad, st));
  BWD_STOP(6);
  // ---- phase 4: folded LayerNorm / hop-2 softmax ---------------------------------------------------
  if (d.nxn) MEMSET0(sc + pl.o_dZR, (size_t)d.NT * d.DZ * esz);
  AVMOE_TRY(k_pre_small_bwd(pl, sv, sc, prm, grads, st));
  }
  if (parts & (2 | 32 | 64)) {   // =============================== section 2: phase 5 ====================================
  const char* dZx = sc + pl.o_Zw;
  const bool do_dx = (parts & (2 | 64)) != 0, do_rest = (parts & (2 | 32)) != 0;
  const bool fork2 = side && (side_mask() & 4) && !d.mha && !d.nxn && (parts & 2);             // (those variants go on accumulating into dX below)

  // ---- phase 5: GEMMs against X --------------------------------------------------------------------
  if (do_dx) {   // dX[s] = dZx[s] Wt + [dL2 | dsx | 1][s] [T ; 1 ; dm1/N][s] + 2 dSxx X[s]   -- one pass: two K segments + row-scale epilogue
    GemmArgs g = base();
    fill_dx_args(g, pl, X, sv, sc, dX);
    g.accumulate = d.acc_dx;
    leaf(g, 16);
    if (fork2) AVMOE_TRY(fk2.fork());
    int dx2 = 1;                                           // the eight-wave direct-load form (dx_stream2.hip: tuned bf16 shape, dX overwritten); 1 = not served
    if (d.bf16 && !d.acc_dx && !dev_env("AVMOE_NO_DX2")) {
      dx2 = k_dx_stream2(X, d.C, dZx, d.DZ, sc + pl.o_dL2x, d.KLp, d.KLT, (const float*)(sc + pl.o_rs2x), sv + pl.o_Wt, d.Cg, (long)d.E * d.dgp * d.Cg,
                         sv + pl.o_Text, d.C, (long)d.KLT * d.C, dX, d.C, d.S, d.N, d.g, d.Cg, d.E * d.dgp, fork2 ? side->s : st);
      if (dx2 < 0) return dx2;
    }
    if (dx2 != OK) AVMOE_TRY(run_on(g, false, fork2 ? side->s : st));
  }
  if (do_rest) {
  if (d.mha) {   // ---- AVS "v1": per expert back through ZR = xr Wt_e^T, the row sums and xr = MHA_e(X) - X -------------------
    for (int e = 0; e < d.E; ++e) {
      if (!d.nxn_of_e[e]) continue;
      const int slot = d.xr_of_e[e];
      {                                                    // dxr = dZR[:, expert e] Wt_e + (2 dSxx) xr
        GemmArgs g = base();
        g.A = sc + pl.o_dZR + (size_t)e * d.dgp * esz; g.B = sv + pl.o_Wt + (size_t)e * d.dgp * d.Cg * esz; g.C = sc + pl.o_dxr;
        g.M = d.NT; g.N = d.Cg; g.K = d.dgp; g.lda = d.DZ; g.b_layout = MN_MAJOR; g.ldb = d.Cg; g.nb2 = d.g;
        g.sA2 = (long)d.E * d.dgp; g.sB2 = (long)d.E * d.dgp * d.Cg; g.sCi = d.C; g.sC2 = d.Cg; g.out_dtype = dt;
        g.row_scale = (const float*)(sc + pl.o_dsr) + (size_t)slot * 3 * d.NT + d.NT;
        g.D = sv + pl.o_xr + (size_t)slot * d.NT * d.C * esz; g.sDi = d.C; g.sD2 = d.Cg;
        AVMOE_TRY(run(g, false));
      }
      AVMOE_TRY(k_nxn_axpy(pl, X, sv, sc, dX, slot, 1, st));   // dxr += dsr2 X + dsr0 ; dX += dsr2 xr - dxr (input replaced)
      AVMOE_TRY(mha_frames_backward(pl, X, prm.e[e], grads.e[e], slot, sc + pl.o_dxr, sv, sc, slabs, slab_cap, dX, st));
    }
  } else if (d.nxn) {   // ---- AVVP N x N block: back through ZR = xr Wt^T, the three row sums and xr = att^T X --------------------
    {                                                      // dxr = dZR Wt + (2 g^2 dSxx) xr
      GemmArgs g = base();
      g.A = sc + pl.o_dZR; g.B = sv + pl.o_Wt; g.C = sc + pl.o_dxr;
      g.M = d.NT; g.N = d.Cg; g.K = d.E * d.dgp; g.lda = d.DZ; g.b_layout = MN_MAJOR; g.ldb = d.Cg; g.nb2 = d.g;
      g.sA2 = (long)d.E * d.dgp; g.sB2 = (long)d.E * d.dgp * d.Cg; g.sCi = d.C; g.sC2 = d.Cg; g.out_dtype = dt;
      g.row_scale = (const float*)(sc + pl.o_dsr) + d.NT; g.D = sv + pl.o_xr; g.sDi = d.C; g.sD2 = d.Cg;
      AVMOE_TRY(run(g, false));
    }
    AVMOE_TRY(k_nxn_axpy(pl, X, sv, sc, dX, 0, 0, st));      // dxr += dsr2 X + dsr0 ; dX += dsr2 xr
    for (int s0 = 0; s0 < d.S; s0 += d.nxc) {              // d.nxc frames at a time (all of them when the forward kept att: moe_plan.cpp)
      const int ns = std::min(d.nxc, d.S - s0);
      const size_t xo = (size_t)s0 * d.N * d.C * esz;
      const char* Xc = (const char*)X + xo;
      const char* dxc = sc + pl.o_dxr + xo;
      char* dXc = (char*)dX + xo;
      if (d.nflash) {
        // the strip kernels re-form att from the kept row log-sum-exp in their accumulators: neither the softmax nor y = att dxr exists in memory
        const float* lse = (const float*)(sv + pl.o_nlse) + (size_t)s0 * d.N;
        AVMOE_TRY(k_nxn_y(Xc, dxc, ns, d.N, d.C, d.Np, lse, (float*)(sc + pl.o_nrd), dXc, st));           // rowdot = X . (att dxr) ; dX += att dxr
        // dX += dS X + dS^T X, dS = att * (X dxr^T - rowdot) formed in the accumulators of both kernels
        AVMOE_TRY(k_nxn_dx(0, Xc, dxc, ns, d.N, d.C, d.Np, lse, (const float*)(sc + pl.o_nrd), dXc, st));
        AVMOE_TRY(k_nxn_dx(1, Xc, dxc, ns, d.N, d.C, d.Np, lse, (const float*)(sc + pl.o_nrd), dXc, st));
        continue;
      } else {
        if (d.nxc < d.S && nxn_att_ok(d.bf16, d.N, d.C, d.Np)) {
          AVMOE_TRY(k_nxn_att(Xc, ns, d.N, d.C, d.Np, (float*)(sv + pl.o_nlse) + (size_t)s0 * d.N, sv + pl.o_att, 1, st));
        } else if (d.nxc < d.S) {                            // the forward ran in chunks too: att of these frames again, from the kept row log-sum-exp
          GemmArgs g = base();
          g.A = Xc; g.B = Xc; g.C = sv + pl.o_att;
          g.M = d.N; g.N = d.N; g.K = d.C; g.lda = d.C; g.ldb = d.C; g.nb1 = ns; g.sA1 = g.sB1 = (long)d.N * d.C;
          g.sCi = d.Np; g.sC1 = (long)d.N * d.Np; g.out_dtype = dt;
          g.epi = GEMM_EPI_EXP; g.row_lse = (const float*)(sv + pl.o_nlse) + (size_t)s0 * d.N;
          g.split3 = AVMOE_FWD_SPLIT3;                       // (the forward's arithmetic: the chunked site must re-form the very att the unchunked one kept)
          AVMOE_TRY(run(g, false));
        }
        {                                                    // y[s] = att[s] dxr[s]   (fp32): the direct term of dX and the softmax's row term
          GemmArgs g = base();
          g.A = sv + pl.o_att; g.B = dxc; g.C = sc + pl.o_nyt;
          g.M = d.N; g.N = d.C; g.K = d.N; g.lda = d.Np; g.b_layout = MN_MAJOR; g.ldb = d.C; g.nb1 = ns;
          g.sA1 = (long)d.N * d.Np; g.sB1 = (long)d.N * d.C; g.sCi = d.C; g.sC1 = (long)d.N * d.C;
          AVMOE_TRY(run(g, true));
        }
        // dX[s] += y[s] ; rowdot_i = X_i . y_i  ( = sum_j att_ij d att_ij, with d att_ij = X_i . dxr_j )
        AVMOE_TRY(k_nxn_rowdot(d.bf16, Xc, (const float*)(sc + pl.o_nyt), (long)ns * d.N, d.C, dXc, (float*)(sc + pl.o_nrd), st));
        if (nxn_att_ok(d.bf16, d.N, d.C, d.Np)) {            // dSc[s] = att[s] * (X[s] dxr[s]^T - rowdot): d att never leaves the chip
          AVMOE_TRY(k_nxn_att_bwd(Xc, dxc, ns, d.N, d.C, d.Np, (const float*)(sc + pl.o_nrd), sv + pl.o_att, sc + pl.o_dSc, st));
        } else {
          GemmArgs g = base();
          g.A = Xc; g.B = dxc; g.C = sc + pl.o_dSc;
          g.M = d.N; g.N = d.N; g.K = d.C; g.lda = d.C; g.ldb = d.C; g.nb1 = ns; g.sA1 = g.sB1 = (long)d.N * d.C;
          g.sCi = d.Np; g.sC1 = (long)d.N * d.Np; g.out_dtype = dt;
          g.epi = GEMM_EPI_MULSUB; g.row_lse = (const float*)(sc + pl.o_nrd); g.D = sv + pl.o_att; g.sDi = d.Np; g.sD1 = (long)d.N * d.Np;
          AVMOE_TRY(run(g, false));
        }
      }
      for (int tr = 0; tr < 2; ++tr) {                     // dX[s] += dSc[s] X[s]  and  dSc[s]^T X[s]
        GemmArgs g = base();
        g.A = sc + pl.o_dSc; g.B = Xc; g.C = dXc;
        g.M = d.N; g.N = d.C; g.K = d.N; g.a_layout = tr ? MN_MAJOR : K_MAJOR; g.lda = d.Np; g.b_layout = MN_MAJOR; g.ldb = d.C;
        g.nb1 = ns; g.sA1 = (long)d.N * d.Np; g.sB1 = (long)d.N * d.C; g.sCi = d.C; g.sC1 = (long)d.N * d.C; g.out_dtype = dt;
        g.accumulate = 1;
        AVMOE_TRY(run(g, true));
      }
    }
  }
  if (d.El > 0) {                                          // dTW[s][slot l] = gate * a^T dzraw  (own expert's columns), all slots in one launch
    const int e0 = d.e_of_lat[0];                          // (latent experts are consecutive: the multimodal ones come first, AVS v2 makes all of them latent)
    GemmArgs g = base();
    g.A = sc + pl.o_ag; g.B = dZx + (size_t)e0 * d.dgp * esz;
    g.C = sc + pl.o_dTW + (size_t)e0 * d.dgp * esz;
    g.M = d.K; g.N = d.dgp; g.K = d.N; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.Kp; g.ldb = d.DZ; g.nb1 = d.S; g.nb2 = d.g; g.nb3 = d.El;
    g.sA1 = (long)d.N * d.Kp; g.sB1 = (long)d.N * d.DZ; g.sB2 = (long)d.E * d.dgp;
    g.sA3 = d.aL; g.sB3 = d.dgp; g.sC3 = (long)d.Kp * d.DZ + d.dgp;      // (ag: per-slot planes [slot][token][Kp])
    g.sCi = d.DZ; g.sC1 = (long)d.KLT * d.DZ; g.sC2 = (long)d.E * d.dgp; g.out_dtype = dt;
    AVMOE_TRY(run(g, false));
  }
  bool pair_done = false;                                  // dWt = dZx^T X and dT[s] = dL2[s]^T X[s] in ONE pass over X (bf16 sites with latent tokens)
  if (d.bf16 && d.El > 0 && !dev_env("AVMOE_NO_TOKPAIR2")) {      // ... as one streaming pass with every accumulator in registers (tok_pair2.hip: the tuned shape)
    const int rc = k_tok_pair2(X, d.C, dZx, d.DZ, sc + pl.o_dL2x, d.KLp, d.S, d.N, d.g, d.Cg, d.E * d.dgp, d.KL, (float*)(sc + pl.o_dWt), (float*)(sc + pl.o_dT),
                               slabs, slab_cap, st);
    if (rc < 0) return rc;
    pair_done = rc == OK;
  }
  if (d.bf16 && d.El > 0 && !pair_done) {
    TokPairArgs t;
    t.A1 = dZx; t.lda1 = d.DZ; t.M1 = d.E * d.dgp; t.sA1g = (long)d.E * d.dgp;
    t.A2 = sc + pl.o_dL2x; t.lda2 = d.KLp; t.M2 = d.KL;
    t.X = X; t.ldx = d.C; t.S = d.S; t.N = d.N; t.g = d.g; t.Cg = d.Cg;
    t.C1 = (float*)(sc + pl.o_dWt); t.C2 = (float*)(sc + pl.o_dT); t.slabs = slabs; t.slab_cap = slab_cap;
    const int rc = launch_gemm_tokpair(t, st);
    if (rc < 0) return rc;
    pair_done = rc == OK;
  }
  {                                                        // dWt = dZx^T X  + dTW^T Text
    GemmArgs g = base();
    g.A = dZx; g.B = X; g.C = sc + pl.o_dWt;
    g.M = d.E * d.dgp; g.N = d.Cg; g.K = d.NT; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.DZ; g.ldb = d.C; g.nb2 = d.g;
    g.sA2 = (long)d.E * d.dgp; g.sB2 = d.Cg; g.sCi = d.Cg; g.sC2 = (long)d.E * d.dgp * d.Cg;
    leaf(g, 2);
    if (!pair_done) AVMOE_TRY(run(g, true));
    if (d.El > 0) {
      GemmArgs h = g;
      h.A = sc + pl.o_dTW; h.B = sv + pl.o_Text; h.K = d.S * d.KLT; h.accumulate = 1; h.ksplit = 1;
      AVMOE_TRY(run(h, true));
    }
    if (d.mha) {                                           // + dZR[:, expert e]^T xr_e
      for (int e = 0; e < d.E; ++e) {
        if (!d.nxn_of_e[e]) continue;
        GemmArgs h = g;
        h.A = sc + pl.o_dZR + (size_t)e * d.dgp * esz; h.B = sv + pl.o_xr + (size_t)d.xr_of_e[e] * d.NT * d.C * esz;
        h.C = sc + pl.o_dWt + (size_t)e * d.dgp * d.Cg * 4; h.M = d.dgp; h.accumulate = 1; h.ksplit = 1;
        AVMOE_TRY(run(h, true));
      }
    } else if (d.nxn) {                                    // + dZR^T xr
      GemmArgs h = g;
      h.A = sc + pl.o_dZR; h.B = sv + pl.o_xr; h.accumulate = 1; h.ksplit = 1;
      AVMOE_TRY(run(h, true));
    }
  }
  if (d.El > 0) {
    if (!pair_done) {                                      // dT[s] = dL2[s]^T X[s]
      GemmArgs g = base();
      g.A = sc + pl.o_dL2x; g.B = X; g.C = sc + pl.o_dT;
      g.M = d.KL; g.N = d.C; g.K = d.N; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.KLp; g.ldb = d.C; g.nb1 = d.S;
      g.sA1 = (long)d.N * d.KLp; g.sB1 = (long)d.N * d.C; g.sCi = d.C; g.sC1 = (long)d.KL * d.C;
      leaf(g, 4);
      AVMOE_TRY(run(g, false));
    }
    {                                                      // dT += dTW Wt
      GemmArgs g = base();
      g.A = sc + pl.o_dTW; g.B = sv + pl.o_Wt; g.C = sc + pl.o_dT;
      g.M = d.KL; g.N = d.Cg; g.K = d.E * d.dgp; g.lda = d.DZ; g.b_layout = MN_MAJOR; g.ldb = d.Cg; g.nb1 = d.S; g.nb2 = d.g;
      g.sA1 = (long)d.KLT * d.DZ; g.sA2 = (long)d.E * d.dgp; g.sB2 = (long)d.E * d.dgp * d.Cg;
      g.sCi = d.C; g.sC1 = (long)d.KL * d.C; g.sC2 = d.Cg; g.accumulate = 1;
      AVMOE_TRY(run(g, false));
    }
    {                                                      // dTT[s][l] = sum_t du3 a a^T   (symmetric)
      GemmArgs g = base();
      g.A = sc + pl.o_aw; g.B = sv + pl.o_a; g.C = sc + pl.o_dTT;
      g.M = d.K; g.N = d.K; g.K = d.N; g.a_layout = g.b_layout = MN_MAJOR; g.lda = g.ldb = d.Kp; g.nb1 = d.S; g.nb2 = d.El;
      g.sA1 = g.sB1 = (long)d.N * d.Kp; g.sA2 = g.sB2 = d.aL;      // (aw, a: per-slot planes)
      g.sCi = d.Kp; g.sC1 = (long)d.El * d.K * d.Kp; g.sC2 = (long)d.K * d.Kp; g.out_dtype = dt;
      AVMOE_TRY(run(g, false));
    }
    {                                                      // dT += 2 dTT T
      GemmArgs g = base();
      g.A = sc + pl.o_dTT; g.B = sv + pl.o_Text; g.C = sc + pl.o_dT;
      g.M = d.K; g.N = d.C; g.K = d.K; g.lda = d.Kp; g.b_layout = MN_MAJOR; g.ldb = d.C; g.nb1 = d.S; g.nb2 = d.El;
      g.sA1 = (long)d.El * d.K * d.Kp; g.sA2 = (long)d.K * d.Kp; g.sB1 = (long)d.KLT * d.C; g.sB2 = (long)d.Kp * d.C;
      g.sCi = d.C; g.sC1 = (long)d.KL * d.C; g.sC2 = (long)d.Kp * d.C; g.alpha = 2.f; g.accumulate = 1;
      AVMOE_TRY(run(g, false));
    }
  }
  AVMOE_TRY(k_finish_dT(pl, sv, sc, st));                  // + dtbar / C ; dTy (T, with the dm2 row) ; dTx ; dT0 ; drw, dbf
  AVMOE_TRY(k_down_bwd(pl, sc, prm, grads, st));
  if (d.mg) AVMOE_TRY(k_merge_gather(pl, sc, grads_in, st));   // diagonal blocks of the dense weight gradients -> the caller's grouped ones
  }
  if (fork2) AVMOE_TRY(fk2.join());
  }
  // ======================= section 3: phase 6 (4 = all of it; 8 = everything but the writers of dY; 16 = the writers of dY) ==========
  const bool do6a = (parts & (4 | 8)) != 0, do6b = (parts & (4 | 16)) != 0;
  if (!do6a && !do6b) return OK;

  if (do6a) {
  // ---- phase 6a: cross-modal hop-1 chain back to Y and the remap parameters -------------------------
  {                                                        // dV = dTy Wf   (row Kcy: d ybar)
    GemmArgs g = base();
    g.A = sc + pl.o_dTy; g.B = sv + pl.o_WfT; g.C = sc + pl.o_dV;
    g.M = d.S * d.Kcyb; g.N = d.Cy; g.K = d.C; g.lda = d.C; g.b_layout = MN_MAJOR; g.ldb = d.Cy; g.sCi = d.Cy; g.out_dtype = dt;
    AVMOE_TRY(run(g, false));
  }
  {                                                        // dWf = dTy^T V
    GemmArgs g = base();
    g.A = sc + pl.o_dTy; g.B = sv + pl.o_V; g.C = sc + pl.o_dWf;
    g.M = d.C; g.N = d.Cy; g.K = d.S * d.Kcyb; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.C; g.ldb = d.Cy; g.sCi = d.Cy;
    leaf(g, 8);
    AVMOE_TRY(run(g, true));
  }
  {                                                        // dBm[s] = dV[s] Y[s]^T
    int rc = 1;                                            // (hop1_stream.hip: dV[s] stationary per frame; 1 = shape not served)
    if (hop1s) rc = k_hop1_yk(Y, d.Cy, d.S, d.M, d.Cy, sc + pl.o_dV, d.Cy, (long)d.Kcyb * d.Cy, d.Kcyb, sc + pl.o_dBm, d.Mb, (long)d.Kcyb * d.Mb, 0, slabs, st);
    if (rc < 0) return rc;
    GemmArgs g = base();
    g.A = sc + pl.o_dV; g.B = Y; g.C = sc + pl.o_dBm;
    g.M = d.Kcyb; g.N = d.M; g.K = d.Cy; g.lda = d.Cy; g.ldb = d.Cy; g.nb1 = d.S;
    g.sA1 = (long)d.Kcyb * d.Cy; g.sB1 = (long)d.M * d.Cy; g.sCi = d.Mb; g.sC1 = (long)d.Kcyb * d.Mb;
    if (rc != OK) AVMOE_TRY(run(g, false));
  }
  AVMOE_TRY(k_prep_dBm(pl, sc, st));
  }
  if (d.Kcy > 0) {
    if (do6a) {
    {                                                      // dA1[s] = [dBm | dab][s] [Wc | bc]^T
      GemmArgs g = base();
      g.A = sc + pl.o_dBmT; g.B = sv + pl.o_WcK; g.C = sc + pl.o_L1;
      g.M = d.Kcyb; g.N = d.N; g.K = d.M + 1; g.lda = d.Mb; g.ldb = d.Mk; g.nb1 = d.S;
      g.sA1 = (long)d.Kcyb * d.Mb; g.sCi = d.Np; g.sC1 = (long)d.Kcyb * d.Np;
      AVMOE_TRY(run(g, false));
    }
    {                                                      // dWcK[:, :M+1] += A1^T [dBm | dab]
      GemmArgs g = base();
      g.A = sv + pl.o_A1y; g.B = sc + pl.o_dBmT; g.C = sc + pl.o_dWcK;
      g.M = d.N; g.N = d.M + 1; g.K = d.S * d.Kcyb; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.Np; g.ldb = d.Mb;
      g.sCi = d.Mk; g.accumulate = 1;
      leaf(g, 8);
      AVMOE_TRY(run(g, true));
    }
    AVMOE_TRY(k_softmax_rows_bwd(d.bf16, sv + pl.o_A1y, (const float*)(sc + pl.o_L1), (long)d.S * d.Kcyb, d.N, d.Np,
                                 sc + pl.o_dL1, nullptr, 1, 1, st));
    {                                                      // dR[s]^T = (dL1[s] Wc)^T   stored [m][kc]
      GemmArgs g = base();
      g.A = sc + pl.o_dL1; g.B = sv + pl.o_WcT; g.C = sc + pl.o_dRT;
      g.M = d.Kcy; g.N = d.M; g.K = d.N; g.lda = d.Np; g.ldb = d.Np; g.nb1 = d.S; g.sA1 = (long)d.Kcyb * d.Np;
      g.sCi = 1; g.sCj = d.Kcyp; g.sC1 = (long)d.M * d.Kcyp; g.out_dtype = dt;
      AVMOE_TRY(run(g, false));
    }
    {                                                      // dWcK[:, :M+2] += dL1^T [R | qr | qb]
      GemmArgs g = base();
      g.A = sc + pl.o_dL1; g.B = sv + pl.o_Rext; g.C = sc + pl.o_dWcK;
      g.M = d.N; g.N = d.M + 2; g.K = d.S * d.Kcyb; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.Np; g.ldb = d.Mk;
      g.sCi = d.Mk; g.accumulate = 1;
      leaf(g, 8);
      AVMOE_TRY(run(g, true));
    }
    AVMOE_TRY(k_dqrqb(pl, sc, prm.conv_b, st));
    {                                                      // dQ = sum_s dR[s] Y[s]
      int rc = 1;                                          // (hop1_stream.hip: one pass over Y with every accumulator in registers + a slab sum)
      if (hop1s) rc = k_hop1_yt_sum(Y, d.Cy, (long)d.S * d.M, d.Cy, sc + pl.o_dRT, d.Kcyp, d.Kcy, sc + pl.o_dQT, d.Cy, 1, slabs, slab_cap, st);
      if (rc < 0) return rc;
      GemmArgs g = base();
      g.A = sc + pl.o_dRT; g.B = Y; g.C = sc + pl.o_dQT; g.out_dtype = dt;      // (straight in the operand dtype of its two consumers)
      g.M = d.Kcy; g.N = d.Cy; g.K = d.S * d.M; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.Kcyp; g.ldb = d.Cy; g.sCi = d.Cy;
      if (rc != OK) AVMOE_TRY(run(g, true));
    }
    }
    if (do6b)
    {   // dY[s] = [Bm ; wbar][s]^T dV[s] + dR[s]^T Q   -- one pass over dY (two K segments)
      GemmArgs g = base();
      g.A = sv + pl.o_BmX; g.B = sc + pl.o_dV; g.C = dY;
      g.M = d.M; g.N = d.Cy; g.K = d.Kcyb; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.Mb; g.ldb = d.Cy; g.nb1 = d.S;
      g.sA1 = (long)d.Kcyb * d.Mb; g.sB1 = (long)d.Kcyb * d.Cy; g.sCi = d.Cy; g.sC1 = (long)d.M * d.Cy; g.out_dtype = dt;
      g.A2 = sc + pl.o_dRT; g.B2 = sv + pl.o_Qx; g.K2 = d.Kcy; g.lda2 = d.Kcyp; g.ldb2 = d.Cy; g.s2A1 = (long)d.M * d.Kcyp;
      g.accumulate = d.acc_dy;
      leaf(g, 16);
      AVMOE_TRY(run(g, false));
    }
    if (do6a) {
    {                                                      // dT0[y slots] += dQ Wf^T
      GemmArgs g = base();
      g.A = sc + pl.o_dQT; g.B = sv + pl.o_WfT; g.C = sc + pl.o_dT0;
      g.M = d.Kcy; g.N = d.C; g.K = d.Cy; g.lda = d.Cy; g.ldb = d.Cy; g.sCi = d.C; g.accumulate = 1;
      AVMOE_TRY(run(g, false));
    }
    {                                                      // dWf += T0^T dQ
      GemmArgs g = base();
      g.A = sv + pl.o_T0T; g.B = sc + pl.o_dQT; g.C = sc + pl.o_dWf;
      g.M = d.C; g.N = d.Cy; g.K = d.Kcy; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.C; g.ldb = d.Cy; g.sCi = d.Cy;
      g.accumulate = 1;
      leaf(g, 8);
      AVMOE_TRY(run(g, false));
    }
    }
  }
  if (d.Kcy == 0 && do6b) {                                // no cross-modal expert: dY[s] = wbar (x) d ybar[s]
    GemmArgs g = base();
    g.A = sv + pl.o_BmX; g.B = sc + pl.o_dV; g.C = dY;
    g.M = d.M; g.N = d.Cy; g.K = d.Kcyb; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.Mb; g.ldb = d.Cy; g.nb1 = d.S;
    g.sA1 = (long)d.Kcyb * d.Mb; g.sB1 = (long)d.Kcyb * d.Cy; g.sCi = d.Cy; g.sC1 = (long)d.M * d.Cy; g.out_dtype = dt;
    g.accumulate = d.acc_dy;
    AVMOE_TRY(run(g, false));
  }
  // ---- phase 6b: latent self attention on X (AVS v2) ----------------------------------------------
  if (d.Kcx > 0) {
    const char* T0x = sv + pl.o_T0T + (size_t)d.Kcy * d.C * esz;
    {                                                      // dA1x[s] = dTx[s] X[s]^T
      GemmArgs g = base();
      g.A = sc + pl.o_dTx; g.B = X; g.C = sc + pl.o_L1;
      g.M = d.Kcx; g.N = d.N; g.K = d.C; g.lda = d.C; g.ldb = d.C; g.nb1 = d.S;
      g.sA1 = (long)d.Kcx * d.C; g.sB1 = (long)d.N * d.C; g.sCi = d.Np; g.sC1 = (long)d.Kcx * d.Np;
      AVMOE_TRY(run(g, false));
    }
    {                                                      // dX[s] += A1x[s]^T dTx[s]
      GemmArgs g = base();
      g.A = sv + pl.o_A1x; g.B = sc + pl.o_dTx; g.C = dX;
      g.M = d.N; g.N = d.C; g.K = d.Kcx; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.Np; g.ldb = d.C; g.nb1 = d.S;
      g.sA1 = (long)d.Kcx * d.Np; g.sB1 = (long)d.Kcx * d.C; g.sCi = d.C; g.sC1 = (long)d.N * d.C; g.out_dtype = dt; g.accumulate = 1;
      AVMOE_TRY(run(g, false));
    }
    MEMSET0(sc + pl.o_dL1xT, (size_t)d.NT * d.Kcxp * esz);
    AVMOE_TRY(k_softmax_rows_bwd(d.bf16, sv + pl.o_A1x, (const float*)(sc + pl.o_L1), (long)d.S * d.Kcx, d.N, d.Np,
                                 sc + pl.o_dL1, sc + pl.o_dL1xT, d.Kcx, d.Kcxp, st));
    {                                                      // dT0[x slots] += sum_{s,n} dL1x^T X
      GemmArgs g = base();
      g.A = sc + pl.o_dL1xT; g.B = X; g.C = sc + pl.o_dT0 + (size_t)d.Kcy * d.C * 4;
      g.M = d.Kcx; g.N = d.C; g.K = d.NT; g.a_layout = g.b_layout = MN_MAJOR; g.lda = d.Kcxp; g.ldb = d.C; g.sCi = d.C;
      g.accumulate = 1;
      AVMOE_TRY(run(g, true));
    }
    {                                                      // dX += dL1x^T T0x
      GemmArgs g = base();
      g.A = sc + pl.o_dL1xT; g.B = T0x; g.C = dX;
      g.M = d.NT; g.N = d.C; g.K = d.Kcx; g.lda = d.Kcxp; g.b_layout = MN_MAJOR; g.ldb = d.C; g.sCi = d.C; g.out_dtype = dt;
      g.accumulate = 1;
      AVMOE_TRY(run(g, false));
    }
  }
  if (do6a) AVMOE_TRY(k_hop1_finalize(pl, sv, sc, prm, grads, st));
  return OK;
}

// The gradient of a token tensor that is X of site A and Y of site B (the two sites of one backbone layer), written once: site A's dX
// product with site B's dY product as two more K segments (dx_stream3.hip).  After sections 1 + 32 + 8 of BOTH sites; replaces section 64 of
// A and section 16 of B.  launch = false: only says whether the shapes are served.  0 = served (launched), 1 = not served, < 0 error.
int moe_backward_dx_dy(const Plan& pa, const void* X, char* sva, char* sca, const Plan& pb, char* svb, char* scb, void* dX, bool launch, hipStream_t st) {
  const Dims& a = pa.d;
  const Dims& b = pb.d;
  if (a.bf16 != b.bf16 || a.mha || a.nxn || a.Kcx > 0 || b.M != a.N || b.Cy != a.C || b.S != a.S || b.Kcx > 0) return 1;
  if (!a.bf16 || a.mg || b.mg || a.Cg != 384 || a.E * a.dgp != 128 || a.KLT > 72 || a.KLp < 72 || b.Kcy < 1 || b.Kcy > 64 || b.Kcyb > 96 || (long)a.S * a.N < 2048) {
    // Any other shape, round 6: the same sum on the tiled engine -- site A's dX product with site B's  dY[s] = [Bm ; wbar][s]^T dV[s] + dR[s]^T Q
    // as a third and a fourth K segment (gemm.h: A3s ..): the token gradient is written once instead of written by site A and read back + added by
    // site B.  fp32 sites only (AVMOE_DXDY_GEN = 1; 3 = bf16 sites of the generalised shapes too: measured, moe_run.h).
    if (!(AVMOE_DXDY_GEN & (a.bf16 ? 2 : 1))) return 1;
    if (!launch) return OK;
    GemmArgs g; g.dtype = a.bf16 ? GEMM_BF16 : GEMM_F32; g.split3 = (AVMOE_LEAF2 & 16) ? 2 : AVMOE_BWD_PLANES;
    fill_dx_args(g, pa, X, sva, sca, dX);
    g.A3s = svb + pb.o_BmX; g.B3s = scb + pb.o_dV; g.K3s = b.Kcyb; g.lda3s = b.Mb; g.ldb3s = b.Cy;
    g.s3sA1 = (long)b.Kcyb * b.Mb; g.s3sB1 = (long)b.Kcyb * b.Cy; g.s3sB2 = a.Cg;
    if (b.Kcy > 0) {
      g.A4s = scb + pb.o_dRT; g.B4s = svb + pb.o_Qx; g.K4s = b.Kcy; g.lda4s = b.Kcyp; g.ldb4s = b.Cy; g.s4sA1 = (long)b.M * b.Kcyp; g.s4sB1 = 0; g.s4sB2 = a.Cg;
    }
    return launch_gemm(g, st);
  }
  if (!launch) return OK;
  return k_dx_stream3(X, a.C, sca + pa.o_Zw, a.DZ, sca + pa.o_dL2x, a.KLp, a.KLT, (const float*)(sca + pa.o_rs2x), sva + pa.o_Wt, a.Cg, (long)a.E * a.dgp * a.Cg,
                      sva + pa.o_Text, a.C, (long)a.KLT * a.C, svb + pb.o_BmX, b.Mb, (long)b.Kcyb * b.Mb, b.Kcyb, scb + pb.o_dRT, b.Kcyp,
                      scb + pb.o_dV, b.Cy, (long)b.Kcyb * b.Cy, svb + pb.o_Qx, b.Cy, b.Kcy, dX, a.C, sca + pa.o_slabs, a.S, a.N, a.g, a.Cg, a.E * a.dgp, st);
}

}  // namespace avmoe
