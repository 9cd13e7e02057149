// Dimensions and workspace layout of one MoEAdapter site.
//
// `saved` persists from forward to backward (the Python facade keeps it in the autograd context);
// `scratch` is transient per call.  Both are caller-allocated (C ABI: the library never allocates).
// Every buffer has a name so tests can pull any intermediate out of the workspace and compare it with
// the same-named tensor of oracle/algebra_ref.py (avmoe_moe_buffer_info).
#pragma once
#include "../../include/avmoe.h"
#include "common.h"
#include <stddef.h>
#include <stdlib.h>

namespace avmoe {

constexpr int MAX_E = AVMOE_MAX_EXPERTS;
constexpr int GRAM_BLOCKS = 512;   // blocks (= partial sums) of the streaming Gram kernel
constexpr int GRAM_SLABS = 1024;   // slabs of the Gram partial-sum workspace (tile_stream.hip: blocks x tile slots)

struct Dims {
  // raw
  int S, N, C, M, Cy, E, E_m, E_s, g, d, K;
  int use_bn, use_gate, ln_before, ln_post, variant, self_attn, lb_loss, training, bf16;
  int gate_w;     // the experts' output gates live in WEIGHT space: Bpost_e is scaled by gate_e, the token-space kernels run with gate 1, and
                  // dgate_e = <dBpost_e, Bpost_e / gate_e> comes out of post_prep_bwd in fp32 (weight_kernels.hip) -- no token-space rounding in it
  int acc_dx, acc_dy;   // backward: accumulate into dX / dY
  int acc_out;          // forward: accumulate into out
  float bn_eps, ln_eps, bn_momentum;
  // derived
  int esz;        // bytes of an activation / operand element (T)
  int zsz;        // bytes of a Z / dz' element: T on the register-resident path (tile_fast.hip), else fp32
  int mg, mdg;    // merged groups: the site's real group count / per-group bottleneck when it is run as ONE group (0: not merged)
  int gen;        // generalised register-resident kernels (tile_gen.hip): per-group bottleneck padded to 16 n, latent slots to 16 / 32 / 96
  int gram64;     // register-resident shape in bf16: z' is kept forward -> backward and the d x d Grams come from gram.hip
  int NT;         // S * N tokens
  int dg, dgp;    // bottleneck per group, padded to 8
  int Cg;         // channels per group
  int DD;         // g * dgp : padded bottleneck width of one expert
  int DZ;         // E * DD : row width of Z / Zx   (layout [group][expert][dgp])
  int El, Ey, Ex; // latent experts: total / source Y (cross-modal) / source X (AVS v2)
  int KL;         // El * Kp latent rows per sample (latent experts in expert order, slots padded to Kp rows)
  int KLT;        // KL + 2 : rows of the extended token matrix Text[s] (ones row, dm1/N row)
  int KLp;        // row width of dL2ext  (>= KL + 2, multiple of 8); the logits L2 are (NT, KL) rows: 256 bytes at two slots of 32
  long aL;        // a / aw / ag are per-latent planes [latent slot][token][Kp] (whole-line stores, their GEMMs run per slot): plane stride NT * Kp
  int Kcy, Kcyb;  // Ey * Kp ; Kcy + 1 (extra wbar / ybar row)
  int Kcx;        // Ex * Kp
  int Kp, Kcyp, Kcxp;  // K, Kcy, Kcx padded to 8 (row strides of dTT / dRT / dL1xT)
  int KP, KPp;    // post GEMM depth per group: E*dgp + 3E, padded to 8
  int XW;         // width (floats) of a (token, group) row of dApx: the 3 E scalar columns of dApost, padded to 16
  int Mk;         // width of Rext / WcK : M + 2 (qr, qb / bc, 1 columns), padded to 8
  int Mb;         // width of Bm_ext: M + 1 (ab column), padded to 8
  int Np;         // N padded to 8
  int lat_of_e[MAX_E];   // latent slot of expert e or -1
  int e_of_lat[MAX_E];
  int src_of_lat[MAX_E]; // 0 = Y (cross-modal), 1 = X (v2)
  int relu_of_e[MAX_E];
  int nxn_of_e[MAX_E];   // AVVP unimodal expert: input is X + gate_av * (softmax(X X^T)^T X)  (mgn.py:132-139)
  int nxn;               // any such expert
  // The unimodal input x + gate * xr of those experts: AVVP shares one xr (slot 0); the AVS "v1" experts (input REPLACED by
  // MultiheadAttention(x) across the frames, PVT_AVSModel_v2.py:210-214) run through the same code with
  // xr = MHA_e(x) - x, gate 1 and one xr slot per expert.
  int nxc;               // frames per chunk of the N x N block (== S: scores / softmax kept for the backward; < S: recomputed there)
  int nflash;            // N x N block through the strip kernels (nxn_att.hip): neither the softmax nor its gradient is ever in memory -- no chunks, no (frames, N, N) workspace
  int nxr;               // xr slots
  int xr_of_e[MAX_E];    // slot of expert e or -1
  int mha;               // "v1" experts present
  int H, dh, Sp;         // heads, head width, frames padded to 8
  int nblk_tok;   // blocks used by the per-token kernels (column-partial slabs are sized by it)
  int xchunks;    // row chunks per frame of the fused X statistics pass
  int fuse_xs;    // the statistics of X (LayerNorm row sums, router column means) come out of the down projection's streaming GEMM: no separate pass
  int excl;       // other kernels may share the GPU with this call (avmoe_moe_desc::shared_gpu): the generalised bottleneck-space kernels take a CU's LDS to themselves
  int fuse_l2;    // ... and so do the hop-2 logits X[s] T[s]^T, as per-group partial sums (L2g) that pre_small adds up: X is read ONCE by the forward's X-side products
};

// name, region (0 saved / 1 scratch), element bytes expr (4 or d.esz), element count expr
#define AVMOE_BUFFERS(X)                                                                      \
  /* ---- weights-derived operands (rebuilt every forward) ---- */                             \
  X(WcK, 0, d.esz, (size_t)d.N * d.Mk)              /* [n][m | bc | 1]                */       \
  X(WcT, 0, d.esz, (size_t)(d.M + 1) * d.Np)        /* [m | bc-row][n]                */       \
  X(WfT, 0, d.esz, (size_t)d.C * d.Cy)              /* fc.weight in T                 */       \
  X(rw, 0, 4, (size_t)d.C)                          /* Wf 1                           */       \
  X(wbar, 0, 4, (size_t)d.Mb)                       /* mean_n Wc                      */       \
  X(scal, 0, 4, 64)                                 /* [0] = mean(bc)                 */       \
  X(T0T, 0, d.esz, (size_t)(d.KL ? d.KL : 1) * d.C) /* stacked my_tokens in T         */       \
  X(Wt, 0, d.esz, (size_t)d.g * d.E * d.dgp * d.Cg) /* [i][e][jp][c] = Wd*gamma_b     */       \
  X(wsum, 0, 4, (size_t)d.DZ)                                                                   \
  X(dconst, 0, 4, (size_t)d.DZ)                                                                 \
  X(mWd, 0, 4, d.mg ? (size_t)d.E * d.d * d.C : 1)  /* merged groups: block-diagonal down_sampler (d, C) per expert */ \
  X(mWu, 0, 4, d.mg ? (size_t)d.E * d.C * d.d : 1)  /* ... up_sampler (C, d) per expert                             */ \
  /* ---- hop 1 (cross-modal experts, source = remapped Y); Kcyb rows per sample ---- */       \
  X(Qx, 0, d.esz, (size_t)(d.Kcy ? d.Kcy : 1) * d.Cy)  /* T0 Wf                        */      \
  X(qrqb, 0, 4, (size_t)2 * (d.Kcy ? d.Kcy : 1))                                               \
  X(Rext, 0, d.esz, (size_t)d.S * d.Kcyb * d.Mk)    /* [R | qr | qb] ; last row 0     */       \
  X(A1y, 0, d.esz, (size_t)d.S * d.Kcyb * d.Np)     /* softmax_n ; last row 0         */       \
  X(BmX, 0, d.esz, (size_t)d.S * d.Kcyb * d.Mb)     /* [A1 Wc | ab] ; last row wbar   */       \
  X(V, 0, d.esz, (size_t)d.S * d.Kcyb * d.Cy)                                                  \
  X(A1x, 0, d.esz, (size_t)d.S * (d.Kcx ? d.Kcx : 1) * d.Np)                                   \
  X(Text, 0, d.esz, (size_t)d.S * d.KLT * d.C)      /* T[s] rows + ones row + dm1 row */       \
  X(Tsum, 0, 4, (size_t)2 * d.S * d.KLT)            /* row sum / sumsq of Text        */       \
  X(TT, 0, 4, (size_t)d.S * (d.El ? d.El : 1) * d.K * d.K)                                     \
  X(TW, 0, 4, (size_t)d.S * d.KLT * d.DZ)           /* Text rows through Wt           */       \
  /* ---- router ---- */                                                                       \
  X(rin, 0, 4, (size_t)d.S * 2 * d.C)                                                           \
  X(rh1, 0, 4, (size_t)d.S * 128)                                                               \
  X(rh2, 0, 4, (size_t)d.S * 32)                                                                \
  X(probs, 0, 4, (size_t)d.S * d.E)                                                             \
  /* ---- per token ---- */                                                                    \
  X(sx, 0, 4, (size_t)2 * d.NT)                     /* row sum / sumsq of X           */       \
  X(Z, 0, d.zsz, (size_t)d.NT * d.DZ)                   /* Zx then z (in place)           */       \
  X(L2, 0, 4, (size_t)d.NT * (d.KL ? d.KL : 8))                                                             \
  X(a, 0, d.esz, (size_t)(d.El ? d.El : 1) * d.NT * (d.Kp ? d.Kp : 8))                                                     \
  X(rmu, 0, 4, (size_t)2 * d.NT * d.E)              /* [r | mu][expert][token]        */       \
  X(rpmup, 0, 4, (size_t)2 * d.NT * d.E)            /* [rp | mup][expert][token]      */       \
  X(bn1, 0, 4, (size_t)4 * d.DZ)                    /* mean, rstd, scale, shift       */       \
  X(mz, 0, 4, (size_t)d.DZ)                                                                     \
  X(Szz, 0, 4, (size_t)d.g * d.E * d.dgp * d.dgp)                                               \
  X(bn2, 0, 4, (size_t)4 * d.E * d.C)               /* mo, rs2, k2, h2   [e][c]       */       \
  X(Bpost, 0, d.esz, (size_t)d.C * d.KPp)                                                       \
  X(Gq, 0, 4, (size_t)d.g * d.E * d.dgp * d.dgp)    /* Wh^T Wh per (i,e)              */       \
  X(uvh, 0, 4, (size_t)2 * d.DZ + 2 * d.g * d.E)    /* usum, vh, H1[i][e], H2[i][e]   */       \
  X(Apost, 0, d.esz, (size_t)d.NT * d.g * d.KPp)                                                \
  /* ---- AVVP N x N block (only sized when present) ---- */                                     \
  X(att, 0, d.esz, (d.nxn && !d.mha && !d.nflash) ? (size_t)d.nxc * d.N * d.Np : 1)   /* softmax_rows(X X^T), nxc frames */  \
  X(nlse, 0, 4, (d.nxn && !d.mha) ? (size_t)d.NT : 1)                      /* row log-sum-exp of X X^T (the backward re-forms att from it) */  \
  X(xr, 0, d.esz, d.nxn ? (size_t)d.nxr * d.NT * d.C : 1)       /* att^T X  |  MHA_e(X) - X per slot */  \
  X(sxr, 0, 4, d.nxn ? (size_t)d.nxr * 3 * d.NT : 1)            /* sum xr, sum xr^2, x . xr  per slot */  \
  X(ZR, 0, 4, d.nxn ? (size_t)d.NT * d.DZ : 1)                  /* xr through Wt                  */  \
  X(nyt, 1, 4, (d.nxn && !d.mha && !d.nflash) ? (size_t)d.nxc * d.N * d.C : 1)            /* y = att dxr of a chunk (fp32) */  \
  X(nrd, 1, 4, (d.nxn && !d.mha) ? (size_t)d.nxc * d.N : 1)                  /* its row dots with X */  \
  X(npart, 1, 4, (d.nxn && !d.mha && !d.nflash) ? (size_t)d.nxc * d.N * ((d.N + 127) / 128) * 2 : 1)   /* per column tile (max, sum exp) of the score rows */  \
  X(dZR, 1, d.esz, d.nxn ? (size_t)d.NT * d.DZ : 1)                                                    \
  X(dsr, 1, 4, d.nxn ? (size_t)d.nxr * 3 * d.NT : 1)                                                   \
  X(dxr, 1, d.esz, d.nxn ? (size_t)d.NT * d.C : 1)                                                     \
  X(dSc, 1, d.esz, (d.nxn && !d.mha && !d.nflash) ? (size_t)d.nxc * d.N * d.Np : 1)                                     \
  /* ---- AVS "v1" MultiheadAttention across the frames, per slot (only sized when present) ---- */ \
  X(mWin, 0, d.esz, d.mha ? (size_t)d.nxr * 3 * d.C * d.C : 1)      /* in_proj_weight in T              */  \
  X(mWout, 0, d.esz, d.mha ? (size_t)d.nxr * d.C * d.C : 1)         /* out_proj.weight in T             */  \
  X(mQKV, 0, d.esz, d.mha ? (size_t)d.nxr * d.NT * 3 * d.C : 1)     /* [s][n][q | k | v]                */  \
  X(mP, 0, d.esz, d.mha ? (size_t)d.nxr * d.N * d.H * d.S * d.Sp : 1)   /* softmax  [n][h][s][s']         */  \
  X(mPd, 0, d.esz, d.mha ? (size_t)d.nxr * d.N * d.H * d.S * d.Sp : 1)  /* after dropout                  */  \
  X(mO, 0, d.esz, d.mha ? (size_t)d.nxr * d.NT * d.C : 1)           /* heads concatenated, before out_proj */ \
  X(mSc, 1, 4, d.mha ? (size_t)d.N * d.H * d.S * d.Sp : 1)          /* scores ; d Pd in the backward    */  \
  X(mdO, 1, d.esz, d.mha ? (size_t)d.NT * d.C : 1)                                                         \
  X(mdS, 1, d.esz, d.mha ? (size_t)d.N * d.H * d.S * d.Sp : 1)                                             \
  X(mdQKV, 1, d.esz, d.mha ? (size_t)d.NT * 3 * d.C : 1)                                                   \
  X(mdW, 1, 4, d.mha ? (size_t)3 * d.C * d.C + 4 * d.C : 1)         /* sink for parameter gradients nobody asked for */ \
  X(mpart, 1, 4, d.mha ? (size_t)256 * 3 * d.C : 1)                 /* row-chunk partial sums of the bias gradients  */ \
  /* ---- transient ---- */                                                                    \
  X(L1, 1, 4, (size_t)d.S * (d.Kcyb > d.Kcx ? d.Kcyb : d.Kcx) * d.Np)    /* L1 ; dA1 in bwd */  \
  X(TV, 1, 4, (size_t)d.S * (d.Kcyb > d.Kcx ? d.Kcyb : d.Kcx) * d.C)                           \
  X(Zp, 1, d.esz, (size_t)d.NT * d.DZ)              /* z'                              */       \
  X(gcolT, 1, 4, d.gram64 ? (size_t)GRAM_BLOCKS * d.DZ : 1)   /* per-block column sums of z' (Gram kernel)   */    \
  X(dSooT, 1, 4, d.gram64 ? (size_t)d.E * d.NT : 1)      /* dSoo per (expert, token)            */    \
  X(gpartT, 1, 4, d.gram64 ? (size_t)GRAM_SLABS * d.g * d.E * d.dgp * d.dgp : 1)  /* Gram partials */ \
  X(Zw, 1, d.esz, (size_t)d.NT * d.DZ)              /* dSoo z' ; later dZx            */       \
  X(colpart, 1, 4, (size_t)d.nblk_tok * 4 * d.DZ)   /* per-block column partial sums  */       \
  X(colsum, 1, 4, (size_t)4 * d.DZ)                 /* colpart summed over blocks     */       \
  X(xpart, 1, 4, (size_t)d.S * (d.fuse_xs ? (d.N + 31) / 32 : d.xchunks) * d.C)     /* column partials of X           */       \
  X(sxp, 1, 4, d.fuse_xs ? (size_t)2 * d.g * d.NT : 1)   /* per-group row sums of X (fused statistics) */   \
  X(L2g, 1, 4, d.fuse_l2 ? (size_t)d.g * d.NT * d.KL : 1)   /* per-group partial hop-2 logits (fused into the down projection) */   \
  X(gpart, 1, 4, (size_t)8 * d.g * d.E * (d.dgp * d.dgp + 2 * d.dgp + 2))  /* Gram partials */ \
  X(rowpart, 1, 4, (size_t)512 * (d.C > d.Cy ? d.C : d.Cy) * 2)  /* chunked row reductions */   \
  X(slabs, 1, 4, slab_floats(d))                    /* split-K partials               */       \
  /* ---- backward only ---- */                                                                \
  /* zero-initialised accumulators: ONE memset at the start of the backward covers [dtbp, dRT] -- keep them adjacent */ \
  X(dtbp, 1, 4, (size_t)d.nblk_tok * (d.KL ? d.KL : 1))                                         \
  X(dTW, 1, d.esz, (size_t)d.S * d.KLT * d.DZ)                                                  \
  X(dWcK, 1, 4, (size_t)d.N * d.Mk)                                                             \
  X(dqp, 1, 4, (size_t)2 * d.S * d.Kcyb + 2 * d.Kcyb)                                           \
  X(dvec, 1, 4, (size_t)2 * d.C + d.Mb + 64)        /* drw, dbf, dwbar, dbcbar (the padding of dwbar stays zero) */ \
  X(dRT, 1, d.esz, (size_t)d.S * d.M * d.Kcyp)                                                  \
  X(dAp, 1, 4, (size_t)d.NT * d.g * d.KPp)          /* f32 ; T (+ dApx) on the register-resident bf16 path */ \
  X(dApx, 1, 4, d.zsz == 2 ? (size_t)d.NT * d.g * d.XW : 1)   /* the 3 E scalar columns of dApost per (token, group), fp32 */ \
  X(dBp, 1, 4, (size_t)d.C * d.KPp)                                                             \
  X(dzp, 1, d.zsz, (size_t)d.NT * d.DZ)                 /* dz' -> dy (in place)           */       \
  X(blkscal, 1, 4, (size_t)d.nblk_tok * d.E * 4)    /* per-block scalar partials      */       \
  X(dGq, 1, 4, (size_t)d.g * d.E * d.dgp * d.dgp)                                               \
  X(sdSzz, 1, 4, (size_t)d.g * d.E * d.dgp * d.dgp) /* 2 dSzz / NT                    */       \
  X(dsm, 1, 4, (size_t)8 * d.DZ + 8 * d.E)          /* dusum,dvh,dmz/NT,mdy,mdyz,ddconst,dwsum ; dH1,dH2 */ \
  X(dmodv, 1, 4, (size_t)3 * d.E * d.C)             /* dmo, dv2, dgate per (e, c)     */       \
  X(dp, 1, 4, (size_t)d.S * d.E)                                                                \
  X(rbw, 1, 4, (size_t)d.S * (d.E + 32 + 128 + 2 * d.C))   /* dlog, dh2r, dh1, drin   */       \
  X(dsxs, 1, 4, (size_t)2 * d.NT)                                                               \
  X(rs2x, 1, 4, (size_t)d.NT)                       /* 2 * sum_e dSxx                 */       \
  X(dslat, 1, 4, (size_t)2 * (d.El ? d.El : 1) * d.NT)   /* dSx, dSxx per (latent expert, token): pre_small_bwd -> pre_lat_bwd */ \
  X(dL2x, 1, d.esz, (size_t)d.NT * d.KLp)           /* [dL2 | dsx | 1]                */       \
  X(aw, 1, d.esz, (size_t)(d.El ? d.El : 1) * d.NT * (d.Kp ? d.Kp : 8))   /* du3 * a  (planes)       */       \
  X(ag, 1, d.esz, (size_t)(d.El ? d.El : 1) * d.NT * (d.Kp ? d.Kp : 8))   /* gate_lat * a  (planes)  */       \
  X(dtbar, 1, 4, (size_t)d.S * (d.KL ? d.KL : 1))                                               \
  X(dTT, 1, d.esz, (size_t)d.S * (d.El ? d.El : 1) * d.K * d.Kp)                                \
  X(dWt, 1, 4, (size_t)d.g * d.E * d.dgp * d.Cg)                                                \
  X(dT, 1, 4, (size_t)d.S * (d.KL ? d.KL : 1) * d.C)                                            \
  X(dTy, 1, d.esz, (size_t)d.S * d.Kcyb * d.C)      /* y-slots of dT + dm2 row, in T  */       \
  X(dTx, 1, d.esz, (size_t)d.S * (d.Kcx ? d.Kcx : 1) * d.C)                                     \
  X(dT0, 1, 4, (size_t)(d.KL ? d.KL : 1) * d.C)                                                 \
  X(dabx, 1, 4, (size_t)d.S * d.Kcyb)                                                           \
  X(dV, 1, d.esz, (size_t)d.S * d.Kcyb * d.Cy)                                                  \
  X(dBm, 1, 4, (size_t)d.S * d.Kcyb * d.Mb)                                                     \
  X(dBmT, 1, d.esz, (size_t)d.S * d.Kcyb * d.Mb)                                                \
  X(dL1, 1, d.esz, (size_t)d.S * (d.Kcyb > d.Kcx ? d.Kcyb : d.Kcx) * d.Np)                      \
  X(dL1xT, 1, d.esz, (size_t)d.NT * d.Kcxp)                                                     \
  X(dQ, 1, 4, (size_t)(d.Kcy ? d.Kcy : 1) * d.Cy)                                               \
  X(dQT, 1, d.esz, (size_t)(d.Kcy ? d.Kcy : 1) * d.Cy)                                          \
  X(dWf, 1, 4, (size_t)d.C * d.Cy)                                                              \
  X(gWd, 1, 4, d.mg ? (size_t)d.E * d.d * d.C : 1)  /* merged groups: gradients of the block-diagonal copies */ \
  X(gWu, 1, 4, d.mg ? (size_t)d.E * d.C * d.d : 1)

size_t slab_floats(const Dims& d);
// smallest site (token elements) that forks a helper stream inside its calls (side.h; AVMOE_SIDE_MIN)
inline long side_min_elements() {
  static const long thr = [] { const char* e = getenv("AVMOE_SIDE_MIN"); return e && *e ? atol(e) : (1L << 25); }();
  return thr;
}

struct BufInfo { const char* name; int region; size_t offset, bytes; };

struct Plan {
  Dims d;
#define X(name, region, eb, cnt) size_t o_##name;
  AVMOE_BUFFERS(X)
#undef X
  size_t saved_bytes, scratch_bytes;
  int nbuf;
  BufInfo info[160];
};

// Validates the descriptor and fills the plan.  Returns 0 or a negative status.
int make_plan(const avmoe_moe_desc* desc, Plan* plan);

}  // namespace avmoe
