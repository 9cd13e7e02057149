"""Seeded random configurations (variants, expert counts, bottlenecks, groups, latent tokens, flags, ragged token counts) through
the C ABI against the oracle in fp32 (1e-3, indices exact): catches interactions the hand-picked cases miss -- padded
bottlenecks, the register-resident path with 2 / 3 / 4 experts, the batch fold, the 32-tile, the attention variants."""
import os
import random

import pytest
import torch

from oracle import avmoe_oracle as O
from tests.golden_util import grad_errors

pytestmark = pytest.mark.gpu


def random_case(seed):
    rng = random.Random(seed)
    variant = rng.choice(["ave", "ave", "avqa", "avvp", "avs", "avs"])
    g = rng.choice([1, 2, 2, 2, 4])
    E_m, E_s = rng.choice([(1, 1), (2, 2), (1, 2), (2, 1), (0, 2), (1, 0), (3, 1), (2, 0)])
    if variant == "avvp" and E_s == 0:
        E_s = 1
    K = rng.choice([32, 32, 32, 2, 8, 13, 20])
    d_g = rng.choice([4, 6, 8, 12, 16, 24, 32, 32, 40])                 # bottleneck per group
    r = rng.choice([2, 3, 4])
    Cx = 8 * g * ((d_g * r + 7) // 8)                                  # C / g a multiple of 8, bottleneck d = Cx // reduction
    red = max(1, Cx // (d_g * g))
    while (Cx // red) % g or Cx // red == 0:
        red -= 1
    self_attn = "none"
    if variant == "avs":
        self_attn = rng.choice(["none", "v2", "v1"])
    elif variant == "ave" and rng.random() < 0.2:
        self_attn = "v1"
    if self_attn == "v1" and ((Cx // 4) % 8 or E_s == 0):
        self_attn = "none"
    big = bool(os.environ.get("AVMOE_FUZZ_BIG"))          # one-off sweeps: token counts that span several blocks / streaming tiles
    cfg = O.AdapterConfig(Cx=Cx, Nx=rng.choice([300, 333, 512, 640, 777, 1024] if big else [5, 17, 40, 64, 97, 130]),
                          Cy=8 * rng.randint(2, 12), Ny=rng.choice([64, 150, 196, 300] if big else [3, 20, 50, 77]),
                          E_m=E_m, E_s=E_s, reduction=red, groups=g, K=K, variant=variant, self_attn=self_attn,
                          use_bn=rng.random() < 0.85, use_gate=rng.random() < 0.85, ln_before=rng.random() < 0.7,
                          ln_post=rng.random() < 0.8, lb_loss=variant in ("avvp", "avs") and rng.random() < 0.6)
    return cfg, rng.choice([2, 4, 7] if big else [1, 2, 3, 5]), rng.random() < 0.8


SEEDS = range(int(os.environ.get("AVMOE_FUZZ_FROM", "0")), int(os.environ.get("AVMOE_FUZZ_TO", "48")))      # widen for a one-off sweep


def relu_margin(P, B, X, Y, cfg, training, noise, keep):
    """Smallest |pre-activation| in front of the cross-modal experts' ReLU in the oracle's forward: a value within rounding of 0
    makes the mask -- and with it the gradients -- depend on the summation order (seen once in ~1000 random configurations)."""
    mins, orig = [1.0], O.F.relu

    def rec(x, inplace=False):
        if x.dim() == 3 and x.shape[-1] == cfg.d:
            mins.append(float(x.abs().min()))
        return orig(x)
    O.F.relu = rec
    try:
        O.moe_forward(P, B, X, Y, cfg, training=training, noise=noise, mha_keep=keep, update_buffers=False)
    finally:
        O.F.relu = orig
    return min(mins)


@pytest.mark.parametrize("seed", SEEDS)
def test_random_configuration_matches_oracle(seed):
    from tests.moe_gpu_util import MoeRun
    cfg, S, training = random_case(seed)
    P, B = O.init_params(cfg, seed=seed)
    g = torch.Generator().manual_seed(1000 + seed)
    X = 0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    Y = 0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)
    G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    noise = 0.01 * torch.randn(S, cfg.E, generator=g) if (cfg.variant == "avs" and seed % 2) else None
    keep = None
    if cfg.self_attn == "v1" and training:
        keep = {pre: (torch.rand(cfg.Nx * cfg.mha_heads, S, S, generator=g) >= cfg.mha_dropout).float() / (1.0 - cfg.mha_dropout)
                for pre in cfg.expert_prefixes()[cfg.E_m:]}
    lbw = 0.01 if cfg.lb_loss else 0.0
    fwd, grads = O.moe_forward_backward(P, B, X, Y, cfg, G, training=training, noise=noise, lb_weight=lbw, mha_keep=keep)
    top2 = torch.topk(fwd["probs"], min(2, cfg.E), dim=-1).values
    if cfg.E > 1 and float((top2[:, 0] - top2[:, -1]).min()) < 1e-4:
        pytest.skip("router margin below the fp32 noise of two different summation orders")
    if relu_margin(P, B, X, Y, cfg, training, noise, keep) < 1e-5:
        pytest.skip("a ReLU pre-activation within fp32 rounding of zero: the gradient is not defined to 1e-3 there")
    run = MoeRun(cfg, P, B, X, Y, bf16=False, training=training, noise=noise, mha_keep=keep).forward()
    assert torch.equal(run.idx.cpu(), fwd["idx"]), cfg
    scale = float(fwd["out"].abs().max())
    assert float((run.out.float().cpu() - fwd["out"]).abs().max()) < 1e-3 * max(scale, 1e-6), cfg
    got = run.backward(G, lb_weight=lbw)
    errs = grad_errors(got, {f"grad.{k}": v for k, v in grads.items()})
    gmax = max(s for _, s in errs.values())
    # (absolute floor 1e-4: gradients that are analytically zero -- a bias in front of a BatchNorm -- are sums of O(1) terms
    #  cancelling to fp32 rounding on both sides)
    bad = {k: (e, s) for k, (e, s) in errs.items() if e > 1e-3 * max(s, 1e-3 * gmax) and e > 1e-4}
    if training and cfg.use_bn and not cfg.ln_before:      # a channel shift in front of a train-mode BatchNorm: its gradient is exactly 0,
        bad = {k: v for k, v in bad.items() if not k.endswith("self_attention.out_proj.bias")}      # both sides hold cancellation noise
    assert not bad, (cfg, bad)
    assert run.guards_intact(), ("a kernel wrote past its workspace", cfg)


@pytest.mark.parametrize("seed", SEEDS)
def test_random_configuration_bf16_close_to_oracle(seed):
    """The same configurations with bf16 activations against the fp32 oracle on the bf16-rounded inputs: outputs within 5e-2,
    gradients norm-wise 25 % (token tensors) / 30 % (parameters, analytically small ones held to a fraction of the largest
    parameter-gradient norm).  A smoke bar for the bf16 code paths of every variant -- a wrong kernel is off by 100 % -- not a
    precision claim: these problems are tiny (5 .. 130 tokens, BatchNorm over a handful of rows), where bf16 rounding alone moves
    the two-hop attention gradients by 10 - 20 %; the precision bars are the mid-size and fixture tests."""
    from tests.moe_gpu_util import MoeRun
    cfg, S, training = random_case(seed)
    P, B = O.init_params(cfg, seed=seed)
    g = torch.Generator().manual_seed(1000 + seed)
    X = (0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g)).bfloat16().float()
    Y = (0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)).bfloat16().float()
    G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g).bfloat16().float()
    noise = 0.01 * torch.randn(S, cfg.E, generator=g) if (cfg.variant == "avs" and seed % 2) else None
    keep = None
    if cfg.self_attn == "v1" and training:
        keep = {pre: (torch.rand(cfg.Nx * cfg.mha_heads, S, S, generator=g) >= cfg.mha_dropout).float() / (1.0 - cfg.mha_dropout)
                for pre in cfg.expert_prefixes()[cfg.E_m:]}
    lbw = 0.01 if cfg.lb_loss else 0.0
    fwd, grads = O.moe_forward_backward(P, B, X, Y, cfg, G, training=training, noise=noise, lb_weight=lbw, mha_keep=keep)
    top2 = torch.topk(fwd["probs"], min(2, cfg.E), dim=-1).values
    if cfg.E > 1 and float((top2[:, 0] - top2[:, -1]).min()) < 2e-2:
        pytest.skip("router margin within bf16 noise")
    if training and cfg.use_bn and S * cfg.Nx < 64:
        pytest.skip("BatchNorm over fewer than 64 rows: bf16 storage of the bottleneck activations is not comparable at 25 %")
    run = MoeRun(cfg, P, B, X, Y, bf16=True, training=training, noise=noise, mha_keep=keep).forward()
    assert torch.equal(run.idx.cpu(), fwd["idx"]), cfg
    assert float((run.out.float().cpu() - fwd["out"]).norm()) <= 5e-2 * float(fwd["out"].norm()) + 1e-6, cfg
    got = run.backward(G, lb_weight=lbw)
    refn = {k: float(v.norm()) for k, v in grads.items()}
    gmax = max(v for k, v in refn.items() if k not in ("X", "Y"))
    bad = {}
    for k, v in got.items():
        err = float((v.float().cpu() - grads[k]).norm())
        tol = 0.25 if k in ("X", "Y") else 0.30
        if err > tol * max(refn[k], 0.25 * gmax if k not in ("X", "Y") else refn[k]) + 1e-7:
            bad[k] = (err, refn[k])
    assert not bad, (cfg, bad)
    assert run.guards_intact(), ("a kernel wrote past its workspace", cfg)
