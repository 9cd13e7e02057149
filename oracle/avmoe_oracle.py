"""CPU oracle for the AVMoE adapter hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, token-major restatement (plain eager PyTorch, CPU, autograd for
the backward) of what the reference's two hot-path modules compute:

    ExpertAdapter   /root/reference/AVMOE/AVE/nets/net_trans_v3.py:296-435
    MoEAdapter      /root/reference/AVMOE/AVE/nets/net_trans_v3.py:438-487
    (AVQA twin      AVQA/net_grd_avst/net_avst_v2.py:215-399  -- identical arithmetic)
    (AVVP variant   AVVP/nets/mgn.py:39-224                   -- N x N unimodal attention, LB loss)
    (AVS variant    AVS/avs_scripts/avs_s4/model/PVT_AVSModel_v2.py:90-318 -- logit noise,
                    probs + LB loss returned, optional self attention of the unimodal experts:
                    "v2" latent tokens on X, "v1" nn.MultiheadAttention across the frames)

It is NOT the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it, and only as the checker / the timed CPU baseline.  The product path lives in
avmoe_amd/ and fails loudly when the HIP library is missing.

Parity pinning: oracle/gen_golden.py imports the real reference modules (in the build
container only, where /root/reference exists), runs them on seeded inputs and writes
tests/golden/*.npz; tests/test_oracle_golden.py checks this restatement against every one of
those vectors (outputs, indices, probabilities, LB loss, input grads and every parameter grad).

Layout convention: the reference API passes x as (S, C, N, 1) -- a permuted view of
token-major (S, N, C) memory (net_trans_v3.py:695).  Everything here is token-major:
X:(S,Nx,Cx), Y:(S,Ny,Cy), out:(S,Nx,Cx).  Parameters are addressed by the reference's own
state_dict key names so fixtures and checkpoints map 1:1.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, asdict
from typing import Dict, Optional

import torch
import torch.nn.functional as F


@dataclass
class AdapterConfig:
    """Static description of one MoEAdapter site (mirrors the ctor args + `opt` flags)."""
    Cx: int                      # input_dim == output_dim == linear_out
    Nx: int                      # conv_dim_out  (this modality's token count)
    Cy: int                      # linear_in     (other modality's channels)
    Ny: int                      # conv_dim_in   (other modality's token count)
    E_m: int = 2                 # opt.num_multimodal_experts
    E_s: int = 2                 # opt.num_singlemodal_experts
    reduction: int = 8           # reduction_factor / opt.Adapter_downsample
    groups: int = 2              # opt.num_conv_group
    K: int = 32                  # num_tk / opt.num_tokens
    use_bn: bool = True
    use_gate: bool = True
    ln_before: bool = True       # opt.is_before_layernorm
    ln_post: bool = True         # opt.is_post_layernorm
    variant: str = "ave"         # ave | avqa | avvp | avs
    self_attn: str = "none"      # none | v2 (AVS latent self attention) | v1 (AVS MultiheadAttention across frames) |
                                 # nxn (AVVP, implied by variant)
    mha_heads: int = 4           # PVT_AVSModel_v2.py:138
    mha_dropout: float = 0.2     # PVT_AVSModel_v2.py:139 (on the attention weights, training only)
    lb_loss: bool = False        # opt.use_load_balacing_loss
    bn_eps: float = 1e-5
    ln_eps: float = 1e-5
    bn_momentum: float = 0.1

    @property
    def d(self) -> int:
        return self.Cx // self.reduction

    @property
    def E(self) -> int:
        return self.E_m + self.E_s

    def expert_prefixes(self):
        """Expert order is all multimodal experts, then singlemodal (net_trans_v3.py:482)."""
        return [f"multimodal_experts.{j}" for j in range(self.E_m)] + \
               [f"singlemodal_experts.{j}" for j in range(self.E_s)]

    def uni_has_attn(self) -> bool:
        return self.variant == "avvp" or self.self_attn in ("v1", "v2", "nxn")

    def to_dict(self):
        return asdict(self)


# ----------------------------------------------------------------------------------------------
# parameter construction (reference default inits: net_trans_v3.py:300-374, 439-466)
# ----------------------------------------------------------------------------------------------
def param_shapes(cfg: AdapterConfig) -> Dict[str, tuple]:
    """state_dict key -> shape, exactly as the reference modules register them."""
    g, d, C = cfg.groups, cfg.d, cfg.Cx
    sh = {
        "conv_adapter.weight": (cfg.Nx, cfg.Ny, 1, 1), "conv_adapter.bias": (cfg.Nx,),
        "fc.weight": (C, cfg.Cy), "fc.bias": (C,),
        "router.0.weight": (128, 2 * C), "router.0.bias": (128,),
        "router.2.weight": (32, 128), "router.2.bias": (32,),
        "router.4.weight": (cfg.E, 32), "router.4.bias": (cfg.E,),
    }
    for j, pre in enumerate(cfg.expert_prefixes()):
        multimodal = j < cfg.E_m
        if cfg.use_gate:
            sh[f"{pre}.gate"] = (1,)
        if multimodal:
            sh[f"{pre}.my_tokens"] = (cfg.K, C)
            sh[f"{pre}.gate_av"] = (1,)
        else:
            if cfg.variant == "avvp":
                sh[f"{pre}.gate_av"] = (1,)          # mgn.py:83
            elif cfg.self_attn == "v2":
                sh[f"{pre}.my_tokens"] = (cfg.K, C)  # PVT_AVSModel_v2.py:144-145
                sh[f"{pre}.gate_self"] = (1,)
            elif cfg.self_attn == "v1":              # nn.MultiheadAttention(C, 4, dropout=.2), PVT_AVSModel_v2.py:141-142
                sh[f"{pre}.self_attention.in_proj_weight"] = (3 * C, C)
                sh[f"{pre}.self_attention.in_proj_bias"] = (3 * C,)
                sh[f"{pre}.self_attention.out_proj.weight"] = (C, C)
                sh[f"{pre}.self_attention.out_proj.bias"] = (C,)
        sh[f"{pre}.down_sampler.weight"] = (d, C // g, 1, 1)
        sh[f"{pre}.up_sampler.weight"] = (C, d // g, 1, 1)
        if cfg.use_bn:
            sh[f"{pre}.bn1.weight"] = (d,); sh[f"{pre}.bn1.bias"] = (d,)
            sh[f"{pre}.bn2.weight"] = (C,); sh[f"{pre}.bn2.bias"] = (C,)
        if cfg.ln_before:
            sh[f"{pre}.ln_before.weight"] = (C,); sh[f"{pre}.ln_before.bias"] = (C,)
        if cfg.ln_post:
            sh[f"{pre}.ln_post.weight"] = (C,); sh[f"{pre}.ln_post.bias"] = (C,)
    return sh


def buffer_shapes(cfg: AdapterConfig) -> Dict[str, tuple]:
    sh = {}
    if cfg.use_bn:
        for pre in cfg.expert_prefixes():
            sh[f"{pre}.bn1.running_mean"] = (cfg.d,); sh[f"{pre}.bn1.running_var"] = (cfg.d,)
            sh[f"{pre}.bn1.num_batches_tracked"] = ()
            sh[f"{pre}.bn2.running_mean"] = (cfg.Cx,); sh[f"{pre}.bn2.running_var"] = (cfg.Cx,)
            sh[f"{pre}.bn2.num_batches_tracked"] = ()
    return sh


def init_params(cfg: AdapterConfig, seed: int = 0, dtype=torch.float32, randomize: bool = True):
    """Parameters + buffers with torch-default-like inits.  With randomize=True the gates and the
    norm affines are drawn away from their (degenerate) defaults, because zero gates make the
    whole adapter output exactly 0 (net_trans_v3.py:309,317)."""
    gen = torch.Generator().manual_seed(seed)
    P, B = {}, {}

    def uni(shape, bound):
        return (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1).mul(bound).to(dtype)

    for k, shp in param_shapes(cfg).items():
        leaf = k.split(".")[-1]
        mod = k.split(".")[-2] if "." in k else ""
        if leaf in ("gate", "gate_av", "gate_self"):
            v = torch.rand(shp, generator=gen, dtype=torch.float64).mul(0.6).add(0.3).to(dtype) \
                if randomize else torch.zeros(shp, dtype=dtype)
        elif leaf == "my_tokens":
            v = torch.rand(shp, generator=gen, dtype=torch.float64).to(dtype)      # torch.rand init
        elif leaf in ("in_proj_weight", "in_proj_bias"):
            v = uni(shp, 1.0 / math.sqrt(cfg.Cx)) if (randomize or leaf == "in_proj_weight") else torch.zeros(shp, dtype=dtype)
        elif mod in ("bn1", "bn2", "ln_before", "ln_post"):
            if randomize:
                v = torch.rand(shp, generator=gen, dtype=torch.float64).add(0.5).to(dtype) \
                    if leaf == "weight" else uni(shp, 0.5)
            else:
                v = torch.ones(shp, dtype=dtype) if leaf == "weight" else torch.zeros(shp, dtype=dtype)
        elif leaf == "weight":        # conv / linear: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
            fan_in = int(math.prod(shp[1:]))
            v = uni(shp, 1.0 / math.sqrt(fan_in))
        else:                         # conv / linear bias
            wshape = param_shapes(cfg)[k[: -len("bias")] + "weight"]
            fan_in = int(math.prod(wshape[1:]))
            v = uni(shp, 1.0 / math.sqrt(fan_in))
        P[k] = v
    for k, shp in buffer_shapes(cfg).items():
        leaf = k.split(".")[-1]
        if leaf == "running_mean":
            B[k] = uni(shp, 0.2) if randomize else torch.zeros(shp, dtype=dtype)
        elif leaf == "running_var":
            B[k] = torch.rand(shp, generator=gen, dtype=torch.float64).add(0.5).to(dtype) \
                if randomize else torch.ones(shp, dtype=dtype)
        else:
            B[k] = torch.zeros(shp, dtype=torch.int64)
    return P, B


# ----------------------------------------------------------------------------------------------
# arithmetic
# ----------------------------------------------------------------------------------------------
def grouped_linear(Z, W4, g):
    """1x1 grouped conv on token-major data (net_trans_v3.py:325-329,395,401).
    Z:(S,N,Cin)  W4:(Cout, Cin/g, 1, 1).  Output channels of group i read input chunk i."""
    W = W4[:, :, 0, 0]
    cout, cin_g = W.shape
    outs = []
    for i in range(g):
        Zi = Z[..., i * cin_g:(i + 1) * cin_g]
        Wi = W[i * (cout // g):(i + 1) * (cout // g)]
        outs.append(Zi @ Wi.t())
    return torch.cat(outs, dim=-1)


def batch_norm_tokens(Z, weight, bias, rmean, rvar, training, eps, momentum, new_buffers, key):
    """BatchNorm2d over (S,N) per channel (net_trans_v3.py:397-403).  Train: biased batch var for
    the normalisation, unbiased var into running stats."""
    if training:
        n = Z.shape[0] * Z.shape[1]
        mu = Z.mean(dim=(0, 1))
        var = Z.var(dim=(0, 1), unbiased=False)
        if new_buffers is not None:
            with torch.no_grad():
                new_buffers[f"{key}.running_mean"] = (1 - momentum) * rmean + momentum * mu
                new_buffers[f"{key}.running_var"] = (1 - momentum) * rvar + momentum * var * (n / max(n - 1, 1))
    else:
        mu, var = rmean, rvar
    return (Z - mu) * torch.rsqrt(var + eps) * weight + bias


def latent_attention(X, src, tokens):
    """Two-hop latent-token attention (net_trans_v3.py:379-388): unscaled, single head.
    X:(S,N,C) queries/receivers, src:(S,N,C) the token set summarised by `tokens` (K,C)."""
    T0 = tokens.unsqueeze(0).expand(X.shape[0], -1, -1)              # :379
    A1 = F.softmax(T0 @ src.transpose(1, 2), dim=-1)                 # :380-381  (S,K,N)
    T = T0 + A1 @ src                                                # :382-383  (S,K,C)
    A2 = F.softmax(X @ T.transpose(1, 2), dim=-1)                    # :385-387  (S,N,K)
    return A2 @ T                                                    # :388      (S,N,C)


def frames_mha(X, Win, bin_, Wout, bout, heads, keep=None):
    """nn.MultiheadAttention(C, heads, dropout) called with batch_first=False on (S, N, C) (PVT_AVSModel_v2.py:212-214): the
    SEQUENCE axis is the frames S, the batch axis the tokens N.  torch.nn.functional.multi_head_attention_forward with
    need_weights=True: q scaled by 1/sqrt(C/heads), softmax over the key frames, dropout on the attention weights, out_proj.
    keep: (N * heads, S, S) multiplier (0 or 1/(1-p)) standing for that dropout, or None (eval / p = 0)."""
    S, N, C = X.shape
    dh = C // heads
    q, k, v = (X @ Win.t() + bin_).split(C, dim=-1)                               # (S, N, C) each
    hd = lambda t: t.reshape(S, N * heads, dh).transpose(0, 1)                    # (N * heads, S, dh)
    att = F.softmax((hd(q) * (1.0 / math.sqrt(dh))) @ hd(k).transpose(1, 2), dim=-1)   # (N * heads, S, S)
    if keep is not None:
        att = att * keep
    o = (att @ hd(v)).transpose(0, 1).reshape(S, N, C)
    return o @ Wout.t() + bout


def expert_forward(P, B, pre, X, Yf, cfg: AdapterConfig, multimodal, training, new_buffers, mha_keep=None,
                   relu_masks=None, record=None):
    """ExpertAdapter.forward, token-major (net_trans_v3.py:376-435).
    relu_masks / record (checker-side instruments, both None in the reference's arithmetic): `record[pre]` receives the ReLU
    pre-activations (S, N, d) of a cross-modal expert; `relu_masks[pre]` (bool, same shape) REPLACES `relu(z)` by `z * mask` --
    the same function wherever the mask equals `z > 0`, used to compare gradients with an implementation whose units within
    rounding of zero fell on the other side of the kink."""
    if multimodal:
        X = X + P[f"{pre}.gate_av"] * latent_attention(X, Yf, P[f"{pre}.my_tokens"])   # :390
    elif cfg.variant == "avvp":
        att = F.softmax(X @ X.transpose(1, 2), dim=-1)               # mgn.py:134-136 (S,N,N)
        X = X + P[f"{pre}.gate_av"] * (att.transpose(1, 2) @ X)      # mgn.py:137-139  x_cn @ att
    elif cfg.self_attn == "v2":
        X = X + P[f"{pre}.gate_self"] * latent_attention(X, X, P[f"{pre}.my_tokens"])  # S4 :215-227
    elif cfg.self_attn == "v1":                                      # S4 :210-214 -- REPLACES x
        X = frames_mha(X, P[f"{pre}.self_attention.in_proj_weight"], P[f"{pre}.self_attention.in_proj_bias"],
                       P[f"{pre}.self_attention.out_proj.weight"], P[f"{pre}.self_attention.out_proj.bias"], cfg.mha_heads,
                       None if mha_keep is None else mha_keep.get(pre))
    elif cfg.self_attn not in ("none",):
        raise NotImplementedError(f"self_attn={cfg.self_attn}")
    if cfg.ln_before:
        X = F.layer_norm(X, (cfg.Cx,), P[f"{pre}.ln_before.weight"], P[f"{pre}.ln_before.bias"], cfg.ln_eps)
    Z = grouped_linear(X, P[f"{pre}.down_sampler.weight"], cfg.groups)             # :395
    if cfg.use_bn:
        Z = batch_norm_tokens(Z, P[f"{pre}.bn1.weight"], P[f"{pre}.bn1.bias"],
                              B[f"{pre}.bn1.running_mean"], B[f"{pre}.bn1.running_var"],
                              training, cfg.bn_eps, cfg.bn_momentum, new_buffers, f"{pre}.bn1")
    if multimodal:
        if record is not None:
            record[pre] = Z.detach()
        if relu_masks is not None and pre in relu_masks:
            Z = Z * relu_masks[pre].to(Z.dtype)
        else:
            Z = F.relu(Z)                                            # :400  (cross-modal only)
    O = grouped_linear(Z, P[f"{pre}.up_sampler.weight"], cfg.groups)               # :401
    if cfg.use_bn:
        O = batch_norm_tokens(O, P[f"{pre}.bn2.weight"], P[f"{pre}.bn2.bias"],
                              B[f"{pre}.bn2.running_mean"], B[f"{pre}.bn2.running_var"],
                              training, cfg.bn_eps, cfg.bn_momentum, new_buffers, f"{pre}.bn2")
    if cfg.ln_post:
        O = F.layer_norm(O, (cfg.Cx,), P[f"{pre}.ln_post.weight"], P[f"{pre}.ln_post.bias"], cfg.ln_eps)
    if cfg.use_gate:
        O = P[f"{pre}.gate"] * O                                     # :433-434
    return O


def load_balancing_loss(probs):
    """Reference quirk (avs_s4/model/PVT_AVSModel_v2.py:314-318): probs is (S,1,E), so the mean over
    dim 0 is (1,E), `uniform` = 1/size(0) = 1.0 and kl_div(..., 'batchmean') = -sum_e log(mean_s p)."""
    pbar = probs.mean(dim=0)
    return -(torch.log(pbar)).sum()


def moe_forward(P, B, X, Y, cfg: AdapterConfig, training=True, noise=None, update_buffers=True, mha_keep=None,
                relu_masks=None, record=None):
    """MoEAdapter.forward (net_trans_v3.py:468-487) on token-major X:(S,Nx,Cx), Y:(S,Ny,Cy).

    Returns dict(out (S,Nx,Cx), probs (S,E), idx (S,) int64, lb (0-d), Yf, new_buffers).
    `noise` (S,E) is the AVS logit noise already scaled by 0.01 (PVT_AVSModel_v2.py:294-296);
    pass None for no noise.  `mha_keep`: {expert prefix: (N * heads, S, S) dropout multiplier} for self_attn == "v1" in training
    mode (the reference draws it from the global RNG); None = no dropout."""
    Wc = P["conv_adapter.weight"][:, :, 0, 0]
    Yt = torch.einsum("nm,smc->snc", Wc, Y) + P["conv_adapter.bias"][None, :, None]    # :469
    Yf = Yt @ P["fc.weight"].t() + P["fc.bias"]                                          # :470
    rin = torch.cat([X.mean(dim=1), Yf.mean(dim=1)], dim=-1)                             # :472-476
    h = F.relu(rin @ P["router.0.weight"].t() + P["router.0.bias"])
    h = F.relu(h @ P["router.2.weight"].t() + P["router.2.bias"])
    logits = h @ P["router.4.weight"].t() + P["router.4.bias"]                           # :477
    if noise is not None:
        logits = logits + noise
    probs = F.softmax(logits, dim=-1)                                                    # :478
    idx = torch.argmax(probs, dim=-1)                                                    # :479
    new_buffers = {} if (training and update_buffers and cfg.use_bn) else None
    out = torch.zeros_like(X)
    for j, pre in enumerate(cfg.expert_prefixes()):                                      # :482
        o = expert_forward(P, B, pre, X, Yf, cfg, j < cfg.E_m, training, new_buffers, mha_keep, relu_masks, record)
        out = out + probs[:, j].reshape(-1, 1, 1) * o                                    # :485-486
    lb = load_balancing_loss(probs) if cfg.lb_loss else torch.zeros((), dtype=X.dtype)
    if new_buffers is not None:
        for pre in cfg.expert_prefixes():
            for bn in ("bn1", "bn2"):
                new_buffers[f"{pre}.{bn}.num_batches_tracked"] = B[f"{pre}.{bn}.num_batches_tracked"] + 1
    return dict(out=out, probs=probs, idx=idx, lb=lb, Yf=Yf, new_buffers=new_buffers)


def moe_forward_backward(P, B, X, Y, cfg: AdapterConfig, grad_out, training=True, noise=None,
                         lb_weight: float = 0.0, mha_keep=None, relu_masks=None, record=None):
    """Forward + autograd backward.  Loss = <out, grad_out> + lb_weight * lb.
    Returns (fwd dict, grads dict with 'X', 'Y' and one entry per parameter key)."""
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    Xg = X.detach().clone().requires_grad_(True)
    Yg = Y.detach().clone().requires_grad_(True)
    fwd = moe_forward(Pg, B, Xg, Yg, cfg, training=training, noise=noise, mha_keep=mha_keep, relu_masks=relu_masks, record=record)
    loss = (fwd["out"] * grad_out).sum()
    if cfg.lb_loss and lb_weight != 0.0:
        loss = loss + lb_weight * fwd["lb"]
    loss.backward()
    grads = {"X": Xg.grad, "Y": Yg.grad}
    for k, v in Pg.items():
        grads[k] = v.grad if v.grad is not None else torch.zeros_like(v)
    fwd = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in fwd.items()}
    return fwd, grads


def moe_grads_over_draws(P, B, X, Y, cfg: AdapterConfig, grad_outs, keys, training=True, noise=None, lb_weight: float = 0.0,
                         mha_keep=None, relu_masks=None):
    """The gradients of the parameters `keys` for SEVERAL upstream gradients `grad_outs` of one forward (inputs, parameters and ReLU
    mask fixed): [{key: grad} per draw].  One forward, one pruned autograd pass per draw -- what the checkers use to estimate the
    error of single-sum gradients (scalar gates) as a ratio of RMS values over draws instead of a ratio of two random sums."""
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    fwd = moe_forward(Pg, B, X, Y, cfg, training=training, noise=noise, update_buffers=False, mha_keep=mha_keep, relu_masks=relu_masks)
    res = []
    for G in grad_outs:
        loss = (fwd["out"] * G).sum()
        if cfg.lb_loss and lb_weight != 0.0:
            loss = loss + lb_weight * fwd["lb"]
        gs = torch.autograd.grad(loss, [Pg[k] for k in keys], retain_graph=True, allow_unused=True)
        res.append({k: (g.detach() if g is not None else torch.zeros_like(Pg[k])) for k, g in zip(keys, gs)})
    return res


# ----------------------------------------------------------------------------------------------
# work model (SURVEY 8d / BASELINE.md 4) -- used by bench.py for the roofline line
# ----------------------------------------------------------------------------------------------
def reference_flops_forward(cfg: AdapterConfig, S: int) -> float:
    """Algorithmic FLOPs of one MoEAdapter forward in the reference's formulation (MAC = 2)."""
    C, d, g, K = cfg.Cx, cfg.d, cfg.groups, cfg.K
    f = 2.0 * S * cfg.Nx * cfg.Ny * cfg.Cy + 2.0 * S * cfg.Nx * cfg.Cy * C
    f += 2.0 * S * (2 * C * 128 + 128 * 32 + 32 * cfg.E)
    f += cfg.E_m * 8.0 * S * K * C * cfg.Nx
    f += cfg.E * 4.0 * S * cfg.Nx * C * d / g
    f += 2.0 * S * cfg.E * C * cfg.Nx
    if cfg.variant == "avvp":
        f += cfg.E_s * 4.0 * S * cfg.Nx * cfg.Nx * C
    elif cfg.self_attn == "v2":
        f += cfg.E_s * 8.0 * S * K * C * cfg.Nx
    return f
