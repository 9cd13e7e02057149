"""GPU parity of avmoe_moe_backward (HIP path behind the C ABI): gradients wrt both token tensors and every
parameter against the vectors captured from the reference (fp32, 1e-3) and at bf16 tolerance."""
import pytest
import torch

from tests.golden_util import golden_names, load_golden, split_params, grad_errors, mha_keep_of

pytestmark = pytest.mark.gpu

SUPPORTED = golden_names()          # every task variant, incl. the AVVP N x N unimodal block


def _report(errs, rtol, floor_frac=1e-3):
    gmax = max(s for _, s in errs.values())
    return {k: (e, s) for k, (e, s) in errs.items() if not (e <= rtol * max(s, floor_frac * gmax))}


@pytest.mark.parametrize("name", SUPPORTED)
def test_backward_fp32_matches_reference_vectors(name, capsys):
    from tests.moe_gpu_util import MoeRun
    meta, cfg, t = load_golden(name)
    P, B = split_params(t)
    run = MoeRun(cfg, P, B, t["X"], t["Y"], bf16=False, training=bool(meta["module_train"]), noise=t.get("noise"), mha_keep=mha_keep_of(t)).forward()
    g = run.backward(t["grad_out"], lb_weight=meta["lb_weight"])
    errs = grad_errors(g, t)
    bad = _report(errs, 1e-3)
    if bad:
        with capsys.disabled():
            print(f"\n[{name}] gradient errors (err, scale):")
            for k, (e, s) in errs.items():
                print(f"   {k:48s} {e:10.3e} {s:10.3e}{'  <<<<' if k in bad else ''}")
    assert not bad, bad
    assert not any(torch.isnan(v).any() for v in g.values()), "a gradient was left unwritten"


@pytest.mark.parametrize("name", ["ave_train", "ave_wide_train", "avs_v2_train"])
def test_backward_bf16_close_to_reference_vectors(name):
    """bf16 activations / operands, fp32 accumulation and fp32 bottleneck space.  Gradients are compared
    norm-wise with the fp32 reference vectors: 6 % for the token tensors (dY runs through the whole
    bf16 hop-1 chain), 12 % for parameters (the router
    sees differences of nearly equal per-expert sums, which amplifies bf16 noise); analytically-zero
    gradients (e.g. a bias in front of a BatchNorm) are held to 3 % of the largest gradient norm."""
    from tests.moe_gpu_util import MoeRun
    meta, cfg, t = load_golden(name)
    P, B = split_params(t)
    run = MoeRun(cfg, P, B, t["X"], t["Y"], bf16=True, training=bool(meta["module_train"]), noise=t.get("noise"), mha_keep=mha_keep_of(t)).forward()
    g = run.backward(t["grad_out"], lb_weight=meta["lb_weight"])
    refn = {k: float(t[f"grad.{k}"].norm()) for k in g}
    gmax = max(v for k, v in refn.items() if k not in ("X", "Y"))
    bad = {}
    for k, v in g.items():
        err = float((v - t[f"grad.{k}"]).norm())
        tol = 0.06 if k in ("X", "Y") else 0.12
        if err > tol * max(refn[k], 0.25 * gmax if k not in ("X", "Y") else refn[k]):
            bad[k] = (err, refn[k])
    assert not bad, bad
