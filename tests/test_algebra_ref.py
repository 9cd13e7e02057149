"""The bottleneck-space factorisation (oracle/algebra_ref.py: the exact arithmetic the HIP path runs,
forward and hand-derived backward) against the direct oracle's autograd, in fp64, for every fixture
configuration (parameters / inputs taken from the golden files) plus the golden vectors themselves."""
import pytest
import torch

from oracle import avmoe_oracle as O
from oracle.algebra_ref import AlgebraRef
from tests.golden_util import golden_names, load_golden, split_params, assert_grads_close, mha_keep_of

NAMES = golden_names()


@pytest.mark.parametrize("name", NAMES)
def test_factorisation_equals_direct_restatement_fp64(name):
    meta, cfg, t = load_golden(name, dtype=torch.float64)
    P, B = split_params(t)
    training = bool(meta["module_train"])
    noise = t.get("noise")
    fwd, grads = O.moe_forward_backward(P, B, t["X"], t["Y"], cfg, t["grad_out"], training=training,
                                        noise=noise, lb_weight=meta["lb_weight"], mha_keep=mha_keep_of(t))
    A = AlgebraRef(cfg, P, B)
    r = A.forward(t["X"], t["Y"], training=training, noise=noise, mha_keep=mha_keep_of(t))
    assert torch.equal(r["idx"], fwd["idx"])
    assert float((r["out"] - fwd["out"]).abs().max()) < 1e-10 * max(1.0, float(fwd["out"].abs().max()))
    assert float((r["probs"] - fwd["probs"]).abs().max()) < 1e-12
    assert abs(float(r["lb"]) - float(fwd["lb"])) < 1e-10
    if training and cfg.use_bn:
        for k, v in fwd["new_buffers"].items():
            assert torch.allclose(r["new_buffers"][k].double(), v.double(), rtol=1e-9, atol=1e-11), k
    g = A.backward(t["grad_out"], lb_weight=meta["lb_weight"])
    ref = {f"grad.{k}": v for k, v in grads.items()}
    assert_grads_close({k: g[k] for k in grads}, ref, rtol=1e-8, floor_frac=1e-6)


@pytest.mark.parametrize("name", ["ave_train", "avvp_train", "avs_v2_train", "avqa_train"])
def test_factorisation_fp32_against_reference_vectors(name):
    meta, cfg, t = load_golden(name)
    P, B = split_params(t)
    A = AlgebraRef(cfg, P, B)
    r = A.forward(t["X"], t["Y"], training=bool(meta["module_train"]), noise=t.get("noise"))
    assert torch.equal(r["idx"], t["idx"])
    assert float((r["out"] - t["out"]).abs().max()) < 1e-4 * float(t["out"].abs().max())
    g = A.backward(t["grad_out"], lb_weight=meta["lb_weight"])
    assert_grads_close({k[5:]: g[k[5:]] for k in t if k.startswith("grad.")}, t, rtol=1e-3)
