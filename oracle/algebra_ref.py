"""Bottleneck-space restatement of the adapter forward AND hand-derived backward.  TEST INFRASTRUCTURE.

The product (avmoe_amd/csrc) does not evaluate the reference's op sequence.  It uses an algebraically
equivalent factorisation in which the full-width token tensors X:(S,N,C), Y:(S,M,Cy), dOut are only
ever touched by skinny GEMMs, and every normalisation / softmax / gate lives in the K- or d-dimensional
"bottleneck space" (DESIGN.md section 3).  This file is that factorisation written with plain torch ops
and NO autograd: every backward formula below is the one the HIP path implements, stage by stage, with
the same intermediate names.  tests/test_algebra_ref.py checks it against oracle/avmoe_oracle.py
(autograd of the direct restatement, itself pinned on the reference's vectors), so an algebra slip is
caught on the CPU before any kernel exists.

Facts used (all exact in real arithmetic):
  remap never materialised   Yf = Wc Y Wf^T + bc (x) rw + 1 (x) bf,   rw = Wf 1
      hop-1 logits  L1 = (T0 Wf) Y^T Wc^T + (T0 rw)(x)bc + (T0 bf)(x)1
      hop-1 update  T  = T0 + ((A1 Wc) Y) Wf^T + (A1 bc)(x)rw + 1(x)bf        (A1 rows sum to 1)
      router mean   mean_n Yf = (wbar^T Y) Wf^T + mean(bc) rw + bf,  wbar = mean_n Wc
  LayerNorm folded into the down projection
      z = r (W~ x' - mu W~1) + Wd beta,   W~ = Wd diag(gamma)
      x' = x + g T^T a   =>  W~ x' = W~ x + g (T W~^T)^T a ;  sum x' = sum x + g C a.tbar ;
                               sum x'^2 = sum x^2 + 2 g a.L2 + g^2 a^T (T T^T) a
  BatchNorm-2 statistics of o = Wu z' from the first / second moments of z' (d-space)
  LayerNorm-post statistics of ob = W^u z' + h2 from a d x d quadratic form
  mixture + gates + both norms' affine folded into ONE output GEMM  out = Apost Bpost^T
"""
from __future__ import annotations

import math
import torch
import torch.nn.functional as F

from .avmoe_oracle import AdapterConfig


def _softmax_bwd(a, da):
    return a * (da - (a * da).sum(-1, keepdim=True))


def _sym(M):
    return M + M.transpose(-1, -2)


class _Expert:
    """Static description of one expert (order: multimodal first -- net_trans_v3.py:482)."""

    def __init__(self, cfg: AdapterConfig, j: int):
        self.j = j
        self.pre = cfg.expert_prefixes()[j]
        self.multimodal = j < cfg.E_m
        self.relu = self.multimodal                       # net_trans_v3.py:400 vs :416-422
        self.nxn = (not self.multimodal) and cfg.variant == "avvp"
        self.mha = (not self.multimodal) and cfg.self_attn == "v1"     # input replaced by MultiheadAttention across the frames
        if self.multimodal:
            self.latent, self.gname = "y", "gate_av"
        elif cfg.self_attn == "v2":
            self.latent, self.gname = "x", "gate_self"
        else:
            self.latent, self.gname = None, ("gate_av" if self.nxn else None)


class AlgebraRef:
    def __init__(self, cfg: AdapterConfig, P, B):
        self.cfg, self.P, self.B = cfg, P, B
        self.experts = [_Expert(cfg, j) for j in range(cfg.E)]

    # ------------------------------------------------------------------------------------------
    def forward(self, X, Y, training=True, noise=None, mha_keep=None):
        cfg, P, B = self.cfg, self.P, self.B
        C, N, Cy, M, g, d, K = cfg.Cx, cfg.Nx, cfg.Cy, cfg.Ny, cfg.groups, cfg.d, cfg.K
        dg, Cg = d // g, C // g
        S = X.shape[0]
        ntot = S * N
        sv = dict(X=X, Y=Y, training=training, S=S)
        Wc = P["conv_adapter.weight"][:, :, 0, 0]
        bc, Wf, bf = P["conv_adapter.bias"], P["fc.weight"], P["fc.bias"]
        rw, wbar, bcbar = Wf.sum(1), Wc.mean(0), bc.mean()
        sv.update(Wc=Wc, bc=bc, Wf=Wf, bf=bf, rw=rw, wbar=wbar, bcbar=bcbar)

        # ---- router (net_trans_v3.py:472-479) on means that never touch Yf ----
        ybar = torch.einsum("m,smy->sy", wbar, Y)
        m2 = ybar @ Wf.t() + bcbar * rw + bf
        m1 = X.mean(1)
        rin = torch.cat([m1, m2], -1)
        a1p = rin @ P["router.0.weight"].t() + P["router.0.bias"]
        h1 = F.relu(a1p)
        a2p = h1 @ P["router.2.weight"].t() + P["router.2.bias"]
        h2r = F.relu(a2p)
        logits = h2r @ P["router.4.weight"].t() + P["router.4.bias"]
        if noise is not None:
            logits = logits + noise
        p = F.softmax(logits, -1)
        idx = torch.argmax(p, -1)
        sv.update(ybar=ybar, rin=rin, a1p=a1p, h1=h1, a2p=a2p, h2r=h2r, p=p)

        out = torch.zeros_like(X)
        new_buffers = {}
        sv["E"] = []
        for ex in self.experts:
            pre = ex.pre
            e = dict()
            sv["E"].append(e)
            Xe = X
            # ---- AVVP unimodal N x N block (mgn.py:132-139): materialised input of this expert ----
            if ex.nxn:
                att = F.softmax(X @ X.transpose(1, 2), -1)
                xr = att.transpose(1, 2) @ X
                Xe = X + P[f"{pre}.gate_av"] * xr
                e.update(att=att, xr=xr)
            # ---- AVS "v1" (PVT_AVSModel_v2.py:210-214): xr = MHA(X) - X over the FRAMES, then as above with gate 1
            #      (csrc/mha_frames.hip: the same products, per (token, head) on strided views) ----
            if ex.mha:
                H = cfg.mha_heads
                dh = C // H
                Win, bin_ = P[f"{pre}.self_attention.in_proj_weight"], P[f"{pre}.self_attention.in_proj_bias"]
                Wout, bout = P[f"{pre}.self_attention.out_proj.weight"], P[f"{pre}.self_attention.out_proj.bias"]
                qkv = (X @ Win.t() + bin_).reshape(S, N, 3, H, dh)                 # [s][n][q|k|v][h][j]
                q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]                 # (S, N, H, dh)
                Pm = F.softmax(torch.einsum("snhj,tnhj->nhst", q, k) / math.sqrt(dh), -1)      # (N, H, S, S')
                keep = None if (mha_keep is None or not training) else mha_keep[pre].reshape(N, H, S, S).to(Pm.dtype)
                Pd = Pm if keep is None else Pm * keep
                Oh = torch.einsum("nhst,tnhj->snhj", Pd, v).reshape(S, N, C)
                xr = Oh @ Wout.t() + bout - X
                Xe = X + xr
                e.update(mha=dict(q=q, k=k, v=v, P=Pm, Pd=Pd, keep=keep, O=Oh, Win=Win, Wout=Wout), xr=xr)
            e["Xe"] = Xe
            # ---- hop 1: latent tokens summarise the (never materialised) remapped other modality ----
            if ex.latent == "y":
                T0 = P[f"{pre}.my_tokens"]
                Q, qr, qb = T0 @ Wf, T0 @ rw, T0 @ bf
                R = torch.einsum("ky,smy->skm", Q, Y)
                L1 = torch.einsum("skm,nm->skn", R, Wc) + qr[None, :, None] * bc[None, None, :] + qb[None, :, None]
                A1 = F.softmax(L1, -1)
                Bm = torch.einsum("skn,nm->skm", A1, Wc)
                ab = torch.einsum("skn,n->sk", A1, bc)
                V = torch.einsum("skm,smy->sky", Bm, Y)
                T = T0[None] + torch.einsum("sky,cy->skc", V, Wf) + ab[..., None] * rw + bf
                e.update(T0=T0, Q=Q, qr=qr, qb=qb, R=R, A1=A1, Bm=Bm, ab=ab, V=V, T=T)
            elif ex.latent == "x":
                T0 = P[f"{pre}.my_tokens"]
                A1 = F.softmax(torch.einsum("kc,snc->skn", T0, X), -1)
                T = T0[None] + torch.einsum("skn,snc->skc", A1, X)
                e.update(T0=T0, A1=A1, T=T)
            # ---- folded down projection weights ----
            Wd_g = P[f"{pre}.down_sampler.weight"][:, :, 0, 0].reshape(g, dg, Cg)
            if cfg.ln_before:
                gb = P[f"{pre}.ln_before.weight"].reshape(g, Cg)
                bb = P[f"{pre}.ln_before.bias"].reshape(g, Cg)
                Wt = Wd_g * gb[:, None, :]
                dconst = torch.einsum("ijc,ic->ij", Wd_g, bb)
                wsum = Wt.sum(-1)
            else:
                Wt, dconst, wsum = Wd_g, None, None
            Xg = Xe.reshape(S, N, g, Cg)
            Zx = torch.einsum("snic,ijc->snij", Xg, Wt)
            sx, sxx = Xe.sum(-1), (Xe * Xe).sum(-1)
            e.update(Wd_g=Wd_g, Wt=Wt, dconst=dconst, wsum=wsum)
            if ex.latent:
                gv = P[f"{pre}.{ex.gname}"]
                L2 = torch.einsum("snc,skc->snk", Xe, T)
                a = F.softmax(L2, -1)
                tbar = T.mean(-1)
                TT = torch.einsum("skc,slc->skl", T, T)
                TW = torch.einsum("skic,ijc->skij", T.reshape(S, K, g, Cg), Wt)
                u1 = torch.einsum("snk,sk->sn", a, tbar)
                u2 = (a * L2).sum(-1)
                u3 = torch.einsum("snk,skl,snl->sn", a, TT, a)
                Sx = sx + gv * C * u1
                Sxx = sxx + 2 * gv * u2 + gv * gv * u3
                zraw = Zx + gv * torch.einsum("snk,skij->snij", a, TW)
                e.update(gv=gv, L2=L2, a=a, tbar=tbar, TT=TT, TW=TW, u1=u1, u2=u2, u3=u3)
            else:
                Sx, Sxx, zraw = sx, sxx, Zx
            if cfg.ln_before:
                mu = Sx / C
                var = Sxx / C - mu * mu
                r = torch.rsqrt(var + cfg.ln_eps)
                z = r[..., None, None] * (zraw - mu[..., None, None] * wsum) + dconst
                e.update(mu=mu, r=r)
            else:
                z = zraw
            e["zraw"] = zraw
            e["z"] = z
            # ---- BN1 (+ReLU for cross-modal experts) ----
            if cfg.use_bn:
                g1 = P[f"{pre}.bn1.weight"].reshape(g, dg)
                b1 = P[f"{pre}.bn1.bias"].reshape(g, dg)
                if training:
                    mean1 = z.mean((0, 1))
                    var1 = z.var((0, 1), unbiased=False)
                    new_buffers[f"{pre}.bn1.running_mean"] = (1 - cfg.bn_momentum) * B[f"{pre}.bn1.running_mean"] + \
                        cfg.bn_momentum * mean1.reshape(-1)
                    new_buffers[f"{pre}.bn1.running_var"] = (1 - cfg.bn_momentum) * B[f"{pre}.bn1.running_var"] + \
                        cfg.bn_momentum * var1.reshape(-1) * (ntot / max(ntot - 1, 1))
                else:
                    mean1 = B[f"{pre}.bn1.running_mean"].reshape(g, dg)
                    var1 = B[f"{pre}.bn1.running_var"].reshape(g, dg)
                r1 = torch.rsqrt(var1 + cfg.bn_eps)
                zh = (z - mean1) * r1
                yb = zh * g1 + b1
                e.update(g1=g1, r1=r1, zh=zh)
            else:
                yb = z
            zp = F.relu(yb) if ex.relu else yb
            e.update(yb=yb, zp=zp)
            # ---- BN2 statistics of o = Wu z' from d-space moments ----
            Wu_g = P[f"{pre}.up_sampler.weight"][:, :, 0, 0].reshape(g, Cg, dg)
            if cfg.use_bn:
                g2 = P[f"{pre}.bn2.weight"].reshape(g, Cg)
                b2 = P[f"{pre}.bn2.bias"].reshape(g, Cg)
                if training:
                    mz = zp.mean((0, 1))
                    Szz = torch.einsum("snij,snil->ijl", zp, zp) / ntot
                    mo = torch.einsum("icj,ij->ic", Wu_g, mz)
                    Eo2 = torch.einsum("icj,ijl,icl->ic", Wu_g, Szz, Wu_g)
                    v2 = Eo2 - mo * mo
                    new_buffers[f"{pre}.bn2.running_mean"] = (1 - cfg.bn_momentum) * B[f"{pre}.bn2.running_mean"] + \
                        cfg.bn_momentum * mo.reshape(-1)
                    new_buffers[f"{pre}.bn2.running_var"] = (1 - cfg.bn_momentum) * B[f"{pre}.bn2.running_var"] + \
                        cfg.bn_momentum * v2.reshape(-1) * (ntot / max(ntot - 1, 1))
                    e.update(mz=mz, Szz=Szz)
                else:
                    mo = B[f"{pre}.bn2.running_mean"].reshape(g, Cg)
                    v2 = B[f"{pre}.bn2.running_var"].reshape(g, Cg)
                rs2 = torch.rsqrt(v2 + cfg.bn_eps)
                k2 = g2 * rs2
                h2 = b2 - mo * k2
                e.update(g2=g2, mo=mo, rs2=rs2)
            else:
                k2 = torch.ones(g, Cg, dtype=X.dtype)
                h2 = torch.zeros(g, Cg, dtype=X.dtype)
            Wh = Wu_g * k2[..., None]
            e.update(Wu_g=Wu_g, k2=k2, h2=h2, Wh=Wh)
            # ---- LN-post statistics of ob = Wh z' + h2 from a d-space quadratic form ----
            if cfg.ln_post:
                gp = P[f"{pre}.ln_post.weight"].reshape(g, Cg)
                bp = P[f"{pre}.ln_post.bias"].reshape(g, Cg)
                usum = Wh.sum(1)
                G = torch.einsum("icj,icl->ijl", Wh, Wh)
                vh = torch.einsum("icj,ic->ij", Wh, h2)
                H1, H2 = h2.sum(), (h2 * h2).sum()
                So = torch.einsum("snij,ij->sn", zp, usum) + H1
                Soo = torch.einsum("snij,ijl,snil->sn", zp, G, zp) + 2 * torch.einsum("snij,ij->sn", zp, vh) + H2
                mup = So / C
                varp = Soo / C - mup * mup
                rp = torch.rsqrt(varp + cfg.ln_eps)
                e.update(usum=usum, G=G, vh=vh)
            else:
                gp = torch.ones(g, Cg, dtype=X.dtype)
                bp = torch.zeros(g, Cg, dtype=X.dtype)
                rp = torch.ones(S, N, dtype=X.dtype)
                mup = torch.zeros(S, N, dtype=X.dtype)
            gate = P[f"{pre}.gate"] if cfg.use_gate else torch.ones(1, dtype=X.dtype)
            # The expert's gate lives in WEIGHT space (round 6, csrc/weight_kernels.hip: Dims::gate_w): the token-space rows carry the router
            # probability only, the expert's columns of Bpost carry gate_e.  Same product; dgate_e = <dBpost_e, Bpost_e / gate_e> then needs no
            # token-space intermediate (a zero-initialised gate, net_trans_v3.py:309, leaves every token-space gradient of the expert exactly 0).
            q = p[:, ex.j][:, None].expand(S, N)
            # ---- the ONE output GEMM  out += Apost Bpost^T ----
            Az = (q * rp)[..., None, None] * zp
            c1, c2, c3 = q * rp, -q * rp * mup, q
            Bmain0 = Wh * gp[..., None]                                         # ungated
            Bh0, Bg0, Bb0 = (gp * h2).reshape(-1), gp.reshape(-1), bp.reshape(-1)
            Bmain, Bh, Bg, Bb = gate * Bmain0, gate * Bh0, gate * Bg0, gate * Bb0
            out = out + torch.einsum("snij,icj->snic", Az, Bmain).reshape(S, N, C) + \
                c1[..., None] * Bh + c2[..., None] * Bg + c3[..., None] * Bb
            e.update(gp=gp, rp=rp, mup=mup, gate=gate, q=q, Az=Az, c1=c1, c2=c2, c3=c3, Bmain=Bmain,
                     Bh=Bh, Bg=Bg, Bb=Bb, Bmain0=Bmain0, Bh0=Bh0, Bg0=Bg0, Bb0=Bb0)
        if cfg.use_bn and training:
            for ex in self.experts:
                for bn in ("bn1", "bn2"):
                    new_buffers[f"{ex.pre}.{bn}.num_batches_tracked"] = B[f"{ex.pre}.{bn}.num_batches_tracked"] + 1
        lb = -(torch.log(p.mean(0))).sum() if cfg.lb_loss else torch.zeros((), dtype=X.dtype)
        self.sv = sv
        return dict(out=out, probs=p, idx=idx, lb=lb, new_buffers=new_buffers if (cfg.use_bn and training) else None)

    # ------------------------------------------------------------------------------------------
    def backward(self, dout, lb_weight=0.0):
        cfg, P, sv = self.cfg, self.P, self.sv
        C, N, Cy, M, g, d, K = cfg.Cx, cfg.Nx, cfg.Cy, cfg.Ny, cfg.groups, cfg.d, cfg.K
        dg, Cg = d // g, C // g
        X, Y, S, training = sv["X"], sv["Y"], sv["S"], sv["training"]
        ntot = S * N
        Wc, bc, Wf, bf, rw, wbar, bcbar = (sv[k] for k in ("Wc", "bc", "Wf", "bf", "rw", "wbar", "bcbar"))
        p = sv["p"]
        G_ = {k: torch.zeros_like(v) for k, v in P.items()}
        dX, dY = torch.zeros_like(X), torch.zeros_like(Y)
        dWc, dbc, dWf, dbf = torch.zeros_like(Wc), torch.zeros_like(bc), torch.zeros_like(Wf), torch.zeros_like(bf)
        drw = torch.zeros_like(rw)
        dp = torch.zeros_like(p)
        doutg = dout.reshape(S, N, g, Cg)

        for ex, e in zip(self.experts, sv["E"]):
            pre = ex.pre
            zp, q, rp, mup = e["zp"], e["q"], e["rp"], e["mup"]
            Xe = e["Xe"]
            # ---- phase 1: the two GEMMs against dOut ----
            dAz = torch.einsum("snic,icj->snij", doutg, e["Bmain"])            # dOut . Bpost
            da1, da2, da3 = dout @ e["Bh"], dout @ e["Bg"], dout @ e["Bb"]
            dBmain = torch.einsum("snic,snij->icj", doutg, e["Az"])            # dOut^T . Apost
            dBh = torch.einsum("snc,sn->c", dout, e["c1"]).reshape(g, Cg)
            dBg = torch.einsum("snc,sn->c", dout, e["c2"]).reshape(g, Cg)
            dBb = torch.einsum("snc,sn->c", dout, e["c3"]).reshape(g, Cg)
            if cfg.use_gate:          # gradients of the GATED weights so far: the gate's own gradient, then the chain to the ungated ones
                G_[f"{pre}.gate"] += (dBmain * e["Bmain0"]).sum() + (dBh.reshape(-1) * e["Bh0"]).sum() + \
                    (dBg.reshape(-1) * e["Bg0"]).sum() + (dBb.reshape(-1) * e["Bb0"]).sum()
                dBmain, dBh, dBg, dBb = e["gate"] * dBmain, e["gate"] * dBh, e["gate"] * dBg, e["gate"] * dBb
            # ---- POST_SMALL backward (per token, d-space) ----
            zz = (dAz * zp).sum((-1, -2))
            dq = rp * zz + rp * da1 - rp * mup * da2 + da3
            dzp = (q * rp)[..., None, None] * dAz
            Wh, h2, k2, Wu_g, gp = e["Wh"], e["h2"], e["k2"], e["Wu_g"], e["gp"]
            if cfg.ln_post:
                drp = q * zz + q * da1 - q * mup * da2
                dmup = -q * rp * da2
                dvarp = drp * (-0.5) * rp ** 3
                dSoo = dvarp / C
                dmup = dmup - 2 * mup * dvarp
                dSo = dmup / C
                Gm, usum, vh = e["G"], e["usum"], e["vh"]
                dzp = dzp + dSo[..., None, None] * usum + \
                    dSoo[..., None, None] * (torch.einsum("ijl,snil->snij", _sym(Gm), zp) + 2 * vh)
                dusum = torch.einsum("sn,snij->ij", dSo, zp)
                dH1 = dSo.sum()
                dG = torch.einsum("sn,snij,snil->ijl", dSoo, zp, zp)
                dvh = 2 * torch.einsum("sn,snij->ij", dSoo, zp)
                dH2 = dSoo.sum()
            dp[:, ex.j] = dq.sum(1)                                             # (dAz already carries the gate)
            # ---- phase 2: weight space (C x d sized) ----
            if cfg.ln_post:
                dWh = gp[..., None] * dBmain + torch.einsum("ijl,icl->icj", _sym(dG), Wh) + dusum[:, None, :] + \
                    dvh[:, None, :] * h2[..., None]
                dh2 = gp * dBh + torch.einsum("ij,icj->ic", dvh, Wh) + dH1 + 2 * h2 * dH2
                G_[f"{pre}.ln_post.weight"] += ((dBmain * Wh).sum(-1) + dBh * h2 + dBg).reshape(-1)
                G_[f"{pre}.ln_post.bias"] += dBb.reshape(-1)
            else:
                dWh, dh2 = dBmain, dBh
            dWu_g = dWh * k2[..., None]
            dk2 = (dWh * Wu_g).sum(-1)
            if cfg.use_bn:
                mo, rs2, g2 = e["mo"], e["rs2"], e["g2"]
                G_[f"{pre}.bn2.bias"] += dh2.reshape(-1)
                dmo = -k2 * dh2
                dk2 = dk2 - mo * dh2
                G_[f"{pre}.bn2.weight"] += (dk2 * rs2).reshape(-1)
                dv2 = dk2 * g2 * (-0.5) * rs2 ** 3
                if training:
                    mz, Szz = e["mz"], e["Szz"]
                    dmo = dmo - 2 * mo * dv2
                    dWu_g = dWu_g + dmo[..., None] * mz[:, None, :] + \
                        dv2[..., None] * torch.einsum("ijl,icl->icj", _sym(Szz), Wu_g)
                    dmz = torch.einsum("ic,icj->ij", dmo, Wu_g)
                    dSzz = torch.einsum("ic,icj,icl->ijl", dv2, Wu_g, Wu_g)
                    dzp = dzp + dmz / ntot + torch.einsum("ijl,snil->snij", _sym(dSzz), zp) / ntot
            G_[f"{pre}.up_sampler.weight"] += dWu_g.reshape(C, dg, 1, 1)
            # ---- phase 3: ReLU, BN1 ----
            dy = dzp * (e["yb"] > 0).to(dzp.dtype) if ex.relu else dzp
            if cfg.use_bn:
                zh, g1, r1 = e["zh"], e["g1"], e["r1"]
                G_[f"{pre}.bn1.weight"] += (dy * zh).sum((0, 1)).reshape(-1)
                G_[f"{pre}.bn1.bias"] += dy.sum((0, 1)).reshape(-1)
                if training:
                    dz = g1 * r1 * (dy - dy.mean((0, 1)) - zh * (dy * zh).mean((0, 1)))
                else:
                    dz = g1 * r1 * dy
            else:
                dz = dy
            # ---- phase 4: PRE_SMALL backward (per token, K/d-space) ----
            Wt, Wd_g = e["Wt"], e["Wd_g"]
            if cfg.ln_before:
                mu, r, wsum, zraw = e["mu"], e["r"], e["wsum"], e["zraw"]
                ddconst = dz.sum((0, 1))
                dzraw = r[..., None, None] * dz
                dr = (dz * (zraw - mu[..., None, None] * wsum)).sum((-1, -2))
                dmu = -r * (dz * wsum).sum((-1, -2))
                dwsum = -((r * mu)[..., None, None] * dz).sum((0, 1))
                dvar = dr * (-0.5) * r ** 3
                dSxx = dvar / C
                dmu = dmu - 2 * mu * dvar
                dSx = dmu / C
            else:
                dzraw = dz
                dSx = torch.zeros(S, N, dtype=X.dtype)
                dSxx = torch.zeros(S, N, dtype=X.dtype)
            dZx, dsx, dsxx = dzraw, dSx, dSxx
            dWt = torch.zeros_like(Wt)
            dXe = torch.zeros_like(X)
            if ex.latent:
                gv, a, L2, T, TW, TT, tbar = e["gv"], e["a"], e["L2"], e["T"], e["TW"], e["TT"], e["tbar"]
                u1, u2, u3 = e["u1"], e["u2"], e["u3"]
                aTW = torch.einsum("snk,skij->snij", a, TW)
                G_[f"{pre}.{ex.gname}"] += (dSx * C * u1).sum() + (dSxx * (2 * u2 + 2 * gv * u3)).sum() + \
                    (dzraw * aTW).sum()
                du1, du2, du3 = dSx * gv * C, 2 * gv * dSxx, gv * gv * dSxx
                da = gv * torch.einsum("skij,snij->snk", TW, dzraw) + du1[..., None] * tbar[:, None, :] + \
                    du2[..., None] * L2 + du3[..., None] * torch.einsum("skl,snl->snk", _sym(TT), a)
                dTW = gv * torch.einsum("snk,snij->skij", a, dzraw)
                dtbar = torch.einsum("sn,snk->sk", du1, a)
                dTT = torch.einsum("sn,snk,snl->skl", du3, a, a)
                dL2 = du2[..., None] * a + _softmax_bwd(a, da)
                # ---- phase 5 (latent part): per-sample GEMMs against X ----
                dXe = dXe + torch.einsum("snk,skc->snc", dL2, T)
                dT = torch.einsum("snk,snc->skc", dL2, Xe)
                dT = dT + torch.einsum("skij,ijc->skic", dTW, Wt).reshape(S, K, C)
                dWt = dWt + torch.einsum("skij,skic->ijc", dTW, T.reshape(S, K, g, Cg))
                dT = dT + torch.einsum("skl,slc->skc", _sym(dTT), T) + dtbar[..., None] / C
            # ---- phase 5: shared-weight GEMMs against X ----
            dXe = dXe + torch.einsum("snij,ijc->snic", dZx, Wt).reshape(S, N, C) + dsx[..., None] + \
                2 * dsxx[..., None] * Xe
            dWt = dWt + torch.einsum("snij,snic->ijc", dZx, Xe.reshape(S, N, g, Cg))
            if cfg.ln_before:
                gb = P[f"{pre}.ln_before.weight"].reshape(g, Cg)
                bb = P[f"{pre}.ln_before.bias"].reshape(g, Cg)
                dWt = dWt + dwsum[..., None]
                dWd_g = dWt * gb[:, None, :] + ddconst[..., None] * bb[:, None, :]
                G_[f"{pre}.ln_before.weight"] += (dWt * Wd_g).sum(1).reshape(-1)
                G_[f"{pre}.ln_before.bias"] += torch.einsum("ij,ijc->ic", ddconst, Wd_g).reshape(-1)
            else:
                dWd_g = dWt
            G_[f"{pre}.down_sampler.weight"] += dWd_g.reshape(d, Cg, 1, 1)
            # ---- N x N block backward (AVVP) ----
            if ex.nxn:
                att, xr = e["att"], e["xr"]
                gav = P[f"{pre}.gate_av"]
                G_[f"{pre}.gate_av"] += (dXe * xr).sum()
                dxr = gav * dXe
                datt = torch.einsum("snc,smc->snm", X, dxr)            # att[n, m]: d/d att[n,m] = x_n . dxr_m
                dX = dX + dXe + att @ dxr
                dSc = _softmax_bwd(att, datt)
                dX = dX + dSc @ X + dSc.transpose(1, 2) @ X
            elif ex.mha:               # xr = MHA(X) - X replaced the input: X only acts through MHA (the direct route cancels)
                m = e["mha"]
                H = cfg.mha_heads
                dh = C // H
                S_, N_ = X.shape[0], X.shape[1]
                dxr = dXe
                G_[f"{pre}.self_attention.out_proj.bias"] += dxr.sum((0, 1))
                G_[f"{pre}.self_attention.out_proj.weight"] += torch.einsum("sno,sni->oi", dxr, m["O"])
                dO = (dxr @ m["Wout"]).reshape(S_, N_, H, dh)
                dPd = torch.einsum("snhj,tnhj->nhst", dO, m["v"])
                dv = torch.einsum("nhst,snhj->tnhj", m["Pd"], dO)
                dP = dPd if m["keep"] is None else dPd * m["keep"]
                dS = _softmax_bwd(m["P"], dP) / math.sqrt(dh)
                dq = torch.einsum("nhst,tnhj->snhj", dS, m["k"])
                dk = torch.einsum("nhst,snhj->tnhj", dS, m["q"])
                dqkv = torch.stack([dq, dk, dv], dim=2).reshape(S_, N_, 3 * C)
                G_[f"{pre}.self_attention.in_proj_bias"] += dqkv.sum((0, 1))
                G_[f"{pre}.self_attention.in_proj_weight"] += torch.einsum("snj,snc->jc", dqkv, X)
                dX = dX + dqkv @ m["Win"]
            else:
                dX = dX + dXe
            # ---- phase 6: hop-1 backward ----
            if ex.latent == "x":
                T0, A1 = e["T0"], e["A1"]
                dT0 = dT.sum(0)
                dA1 = torch.einsum("skc,snc->skn", dT, X)
                dX = dX + torch.einsum("skn,skc->snc", A1, dT)
                dL1 = _softmax_bwd(A1, dA1)
                dT0 = dT0 + torch.einsum("skn,snc->kc", dL1, X)
                dX = dX + torch.einsum("skn,kc->snc", dL1, T0)
                G_[f"{pre}.my_tokens"] += dT0
            elif ex.latent == "y":
                T0, Q, qr, qb, R, A1, Bm, ab, V = (e[k] for k in ("T0", "Q", "qr", "qb", "R", "A1", "Bm", "ab", "V"))
                dT0 = dT.sum(0)
                dV = torch.einsum("skc,cy->sky", dT, Wf)
                dWf += torch.einsum("skc,sky->cy", dT, V)
                dab = torch.einsum("skc,c->sk", dT, rw)
                drw += torch.einsum("skc,sk->c", dT, ab)
                dbf += dT.sum((0, 1))
                dBm = torch.einsum("sky,smy->skm", dV, Y)
                dY = dY + torch.einsum("skm,sky->smy", Bm, dV)
                dA1 = torch.einsum("skm,nm->skn", dBm, Wc) + dab[..., None] * bc
                dWc += torch.einsum("skn,skm->nm", A1, dBm)
                dbc += torch.einsum("skn,sk->n", A1, dab)
                dL1 = _softmax_bwd(A1, dA1)
                dR = torch.einsum("skn,nm->skm", dL1, Wc)
                dWc += torch.einsum("skn,skm->nm", dL1, R)
                dqr = torch.einsum("skn,n->k", dL1, bc)
                dbc += torch.einsum("skn,k->n", dL1, qr)
                dqb = dL1.sum((0, 2))
                dQ = torch.einsum("skm,smy->ky", dR, Y)
                dY = dY + torch.einsum("skm,ky->smy", dR, Q)
                dT0 = dT0 + dQ @ Wf.t() + dqr[:, None] * rw + dqb[:, None] * bf
                dWf += T0.t() @ dQ
                drw += T0.t() @ dqr
                dbf += T0.t() @ dqb
                G_[f"{pre}.my_tokens"] += dT0

        # ---- router backward (mixture weights, LB loss, MLP, the two means) ----
        if cfg.lb_loss and lb_weight != 0.0:
            dp = dp + lb_weight * (-1.0 / (S * p.mean(0)))[None, :]
        dlog = _softmax_bwd(p, dp)
        W1, W2, W3 = P["router.0.weight"], P["router.2.weight"], P["router.4.weight"]
        G_["router.4.weight"] += dlog.t() @ sv["h2r"]
        G_["router.4.bias"] += dlog.sum(0)
        dh2r = (dlog @ W3) * (sv["a2p"] > 0).to(X.dtype)
        G_["router.2.weight"] += dh2r.t() @ sv["h1"]
        G_["router.2.bias"] += dh2r.sum(0)
        dh1 = (dh2r @ W2) * (sv["a1p"] > 0).to(X.dtype)
        G_["router.0.weight"] += dh1.t() @ sv["rin"]
        G_["router.0.bias"] += dh1.sum(0)
        drin = dh1 @ W1
        dm1, dm2 = drin[:, :C], drin[:, C:]
        dX = dX + dm1[:, None, :] / N
        dybar = dm2 @ Wf
        dWf += dm2.t() @ sv["ybar"]
        drw += bcbar * dm2.sum(0)
        dbf += dm2.sum(0)
        dbcbar = (dm2 @ rw).sum()
        dY = dY + wbar[None, :, None] * dybar[:, None, :]
        dwbar = torch.einsum("smy,sy->m", Y, dybar)
        dWc += dwbar[None, :] / N
        dbc += dbcbar / N
        dWf += drw[:, None]
        G_["conv_adapter.weight"] += dWc[:, :, None, None]
        G_["conv_adapter.bias"] += dbc
        G_["fc.weight"] += dWf
        G_["fc.bias"] += dbf
        G_["X"], G_["Y"] = dX, dY
        return G_
