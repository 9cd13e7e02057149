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
    """bf16 activations / operands / bottleneck-space tensors, fp32 accumulation, against the fp32 vectors captured from the reference:
    every gradient norm-wise within max(1 %, 2 x the error of the reference formulation itself under bf16 autocast) of the oracle on
    the bf16-rounded inputs (tests/golden_util.py::bf16_budget_violations), and within 1.5 x that bar of the reference's own vectors
    (which saw the unrounded inputs)."""
    from oracle import avmoe_oracle as O
    from tests.golden_util import bf16_budget_violations
    from tests.moe_gpu_util import MoeRun
    meta, cfg, t = load_golden(name)
    P, B = split_params(t)
    training = bool(meta["module_train"])
    run = MoeRun(cfg, P, B, t["X"], t["Y"], bf16=True, training=training, noise=t.get("noise"), mha_keep=mha_keep_of(t)).forward()
    g = run.backward(t["grad_out"], lb_weight=meta["lb_weight"])
    Xb, Yb, Gb = t["X"].bfloat16().float(), t["Y"].bfloat16().float(), t["grad_out"].bfloat16().float()
    _, grads = O.moe_forward_backward(P, B, Xb, Yb, cfg, Gb, training=training, noise=t.get("noise"), lb_weight=meta["lb_weight"],
                                      mha_keep=mha_keep_of(t))
    bad = bf16_budget_violations(O, cfg, P, B, Xb, Yb, Gb, g, grads, training=training, lb_weight=meta["lb_weight"], noise=t.get("noise"),
                                 mha_keep=mha_keep_of(t))
    assert not bad, bad
    ref = {k: t[f"grad.{k}"] for k in g}                                         # the reference's vectors (unrounded inputs)
    bad = bf16_budget_violations(O, cfg, P, B, Xb, Yb, Gb, g, ref, training=training, lb_weight=meta["lb_weight"], noise=t.get("noise"),
                                 mha_keep=mha_keep_of(t), factor=3.0, floor=1.5e-2, zero_abs=2e-4)
    assert not bad, bad

@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("sections", [(1, 2, 4), (3, 8, 16), (1, 2, 8, 16)])
def test_backward_sections_equal_the_whole(bf16, sections):
    """avmoe_moe_backward_part with parts = 1, 2, 4 in turn (ABI 4), or with the last section in two steps -- 8 = without the GEMMs
    that write dY, 16 = those GEMMs (ABI 5) -- == avmoe_moe_backward, bit for bit; a site with latent self attention (its last
    section writes dX too) refuses the split."""
    import ctypes as C
    from avmoe_amd import _capi as capi
    from avmoe_amd import _capi_moe as cm
    from tests.moe_gpu_util import MoeRun
    meta, cfg, t = load_golden("ave_wide_train")
    P, B = split_params(t)
    whole = MoeRun(cfg, P, B, t["X"], t["Y"], bf16=bf16, training=True).forward()
    g0 = whole.backward(t["grad_out"])
    run = MoeRun(cfg, P, B, t["X"], t["Y"], bf16=bf16, training=True).forward()
    run.dOut = t["grad_out"].to(run.dev, run.tdt).contiguous()
    run.dX, run.dY = torch.empty_like(run.X), torch.empty_like(run.Y)
    run.grads = {k: torch.full_like(v, float("nan")) for k, v in run.params.items()}
    gptrs = cm.make_ptrs(run.grads, cfg.E_m, cfg.E_s)
    lbw = torch.zeros(1, device=run.dev)
    for parts in sections:
        st = run.L.avmoe_moe_backward_part(C.byref(run.desc), run.X.data_ptr(), run.Y.data_ptr(), C.byref(run.ptrs), run.dOut.data_ptr(),
                                           lbw.data_ptr(), run.saved.data_ptr(), run.scratch.data_ptr(), run.dX.data_ptr(), run.dY.data_ptr(),
                                           C.byref(gptrs), parts, torch.cuda.current_stream().cuda_stream)
        capi.check(st, "avmoe_moe_backward_part")
    torch.cuda.synchronize()
    assert torch.equal(run.dX.float().cpu(), g0["X"]) and torch.equal(run.dY.float().cpu(), g0["Y"])
    for k, v in run.grads.items():
        assert torch.equal(v.cpu(), g0[k]), k
    meta, cfg, t = load_golden("avs_v2_train")
    P, B = split_params(t)
    r2 = MoeRun(cfg, P, B, t["X"], t["Y"], bf16=bf16, training=True).forward()
    r2.dOut = t["grad_out"].to(r2.dev, r2.tdt).contiguous()
    r2.dX, r2.dY = torch.empty_like(r2.X), torch.empty_like(r2.Y)
    gr = {k: torch.zeros_like(v) for k, v in r2.params.items()}
    st = r2.L.avmoe_moe_backward_part(C.byref(r2.desc), r2.X.data_ptr(), r2.Y.data_ptr(), C.byref(r2.ptrs), r2.dOut.data_ptr(), lbw.data_ptr(),
                                      r2.saved.data_ptr(), r2.scratch.data_ptr(), r2.dX.data_ptr(), r2.dY.data_ptr(),
                                      C.byref(cm.make_ptrs(gr, cfg.E_m, cfg.E_s)), 1, torch.cuda.current_stream().cuda_stream)
    assert st == -2                                       # AVMOE_ERR_UNSUPPORTED
