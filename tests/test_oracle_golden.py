"""Pins oracle/avmoe_oracle.py against vectors captured from the real reference modules
(tests/golden/*.npz, written by oracle/gen_golden.py).  CPU only."""
import pytest
import torch

from oracle import avmoe_oracle as O
from tests.golden_util import golden_names, load_golden, split_params, assert_grads_close, mha_keep_of

NAMES = golden_names()


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference_vectors(name):
    meta, cfg, t = load_golden(name)
    P, B = split_params(t)
    training = bool(meta["module_train"])
    noise = t.get("noise")
    fwd, grads = O.moe_forward_backward(P, B, t["X"], t["Y"], cfg, t["grad_out"], training=training,
                                        noise=noise, lb_weight=meta["lb_weight"], mha_keep=mha_keep_of(t))
    assert torch.equal(fwd["idx"], t["idx"]), "router argmax must be bit-exact"
    assert _rel(fwd["out"], t["out"]) < 2e-5
    assert _rel(fwd["probs"], t["probs"]) < 1e-5
    if cfg.lb_loss:
        assert abs(float(fwd["lb"]) - float(t["lb"])) < 1e-4 * max(1.0, abs(float(t["lb"])))
    assert_grads_close(grads, t, rtol=2e-4)
    if training and cfg.use_bn:
        for k, v in fwd["new_buffers"].items():
            ref = t[f"newbuffer.{k}"]
            assert torch.allclose(v.to(ref.dtype), ref, rtol=1e-5, atol=1e-6), k


def test_fixture_set_covers_every_variant():
    whiches = {load_golden(n)[0]["which"] for n in NAMES}
    assert {"ave", "avqa", "avvp", "avs", "avs_ms3"} <= whiches


def test_zero_gates_give_exact_zero():
    """freshly constructed adapters (gates = 0) output exactly 0."""
    cfg = O.AdapterConfig(Cx=32, Nx=10, Cy=16, Ny=12, reduction=4, groups=2, K=4)
    P, B = O.init_params(cfg, seed=3, randomize=False)
    X = torch.randn(3, cfg.Nx, cfg.Cx)
    Y = torch.randn(3, cfg.Ny, cfg.Cy)
    out = O.moe_forward(P, B, X, Y, cfg)["out"]
    assert float(out.abs().max()) == 0.0


def test_single_expert_probs_are_one_and_lb_of_uniform():
    cfg = O.AdapterConfig(Cx=32, Nx=10, Cy=16, Ny=12, reduction=4, groups=2, K=4, E_m=1, E_s=0)
    P, B = O.init_params(cfg, seed=4)
    r = O.moe_forward(P, B, torch.randn(2, 10, 32), torch.randn(2, 12, 16), cfg)
    assert torch.all(r["probs"] == 1.0)
    E = 4
    lb = O.load_balancing_loss(torch.full((5, E), 1.0 / E))
    assert abs(float(lb) - E * torch.log(torch.tensor(float(E))).item()) < 1e-5


def test_relu_mask_instruments_do_not_change_the_arithmetic():
    """The checker-side instruments of the oracle (record the ReLU pre-activations / replace relu(z) by z * mask) reproduce the plain
    run exactly when the mask is the oracle's own `z > 0`, and a flipped unit changes only what passes through it."""
    cfg = O.AdapterConfig(Cx=64, Nx=24, Cy=48, Ny=20, reduction=4, groups=2, K=8)
    P, B = O.init_params(cfg, seed=3)
    g = torch.Generator().manual_seed(5)
    X, Y = 0.3 * torch.randn(3, cfg.Nx, cfg.Cx, generator=g), 0.3 * torch.randn(3, cfg.Ny, cfg.Cy, generator=g)
    G = torch.randn(3, cfg.Nx, cfg.Cx, generator=g)
    rec = {}
    f0, g0 = O.moe_forward_backward(P, B, X, Y, cfg, G, record=rec)
    assert set(rec) == {"multimodal_experts.0", "multimodal_experts.1"} and rec["multimodal_experts.0"].shape == (3, cfg.Nx, cfg.d)
    masks = {k: z > 0 for k, z in rec.items()}
    f1, g1 = O.moe_forward_backward(P, B, X, Y, cfg, G, relu_masks=masks)
    assert torch.equal(f0["out"], f1["out"])
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    z = rec["multimodal_experts.0"]
    s, n, j = [int(i) for i in (z.abs() == z.abs().min()).nonzero()[0]]      # the unit closest to the kink
    masks["multimodal_experts.0"][s, n, j] ^= True
    _f2, g2 = O.moe_forward_backward(P, B, X, Y, cfg, G, relu_masks=masks)
    assert not torch.equal(g0["multimodal_experts.0.bn1.bias"], g2["multimodal_experts.0.bn1.bias"])
    assert torch.equal(g0["multimodal_experts.1.bn1.bias"], g2["multimodal_experts.1.bn1.bias"]) or \
        float((g0["multimodal_experts.1.bn1.bias"] - g2["multimodal_experts.1.bn1.bias"]).abs().max()) < 1e-3
