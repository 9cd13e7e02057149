"""GPU parity of avmoe_moe_forward (the HIP path behind the C ABI) against the golden vectors captured
from the reference and, stage by stage, against oracle/algebra_ref.py."""
import pytest
import torch

from oracle.algebra_ref import AlgebraRef
from tests.golden_util import golden_names, load_golden, split_params, mha_keep_of

pytestmark = pytest.mark.gpu

SUPPORTED = golden_names()          # every task variant, incl. the AVVP N x N unimodal block


@pytest.mark.parametrize("name", SUPPORTED)
def test_forward_fp32_matches_reference_vectors(name, capsys):
    from tests.moe_gpu_util import MoeRun, compare_forward_intermediates
    meta, cfg, t = load_golden(name)
    P, B = split_params(t)
    training = bool(meta["module_train"])
    run = MoeRun(cfg, P, B, t["X"], t["Y"], bf16=False, training=training, noise=t.get("noise"), mha_keep=mha_keep_of(t)).forward()
    out = run.out.float().cpu()
    err = float((out - t["out"]).abs().max() / t["out"].abs().max())
    ok = err < 1e-3 and torch.equal(run.idx.cpu(), t["idx"])
    if not ok:
        A = AlgebraRef(cfg, P, B)
        A.forward(t["X"], t["Y"], training=training, noise=t.get("noise"), mha_keep=mha_keep_of(t))
        with capsys.disabled():
            print(f"\n[{name}] out rel err {err:.3e}")
            compare_forward_intermediates(run, A)
    assert torch.equal(run.idx.cpu(), t["idx"]), "router argmax must be bit-exact"
    assert err < 1e-3, err                      # north_star: within 1e-3 rel fp32
    assert float((run.probs.cpu() - t["probs"]).abs().max()) < 1e-5
    if cfg.lb_loss:
        assert abs(float(run.lb.cpu()) - float(t["lb"])) < 1e-4 * max(1.0, abs(float(t["lb"])))
    if training and cfg.use_bn:
        for k, v in run.buffers.items():
            assert torch.allclose(v.cpu(), t[f"newbuffer.{k}"], rtol=2e-4, atol=2e-5), k


@pytest.mark.parametrize("name", ["ave_train", "ave_wide_train", "avs_v2_train"])
def test_forward_bf16_close_to_reference_vectors(name):
    """bf16 I/O + fp32 accumulate against the fp32 reference output (which saw the unrounded inputs): within 1.5e-2, max-abs relative
    (measured 4e-3 .. 9e-3 on these vectors; the bf16 bar of the parity tests proper is 1e-2 against the oracle on rounded inputs)."""
    from tests.moe_gpu_util import MoeRun
    meta, cfg, t = load_golden(name)
    P, B = split_params(t)
    run = MoeRun(cfg, P, B, t["X"], t["Y"], bf16=True, training=bool(meta["module_train"]), noise=t.get("noise"), mha_keep=mha_keep_of(t)).forward()
    out = run.out.float().cpu()
    err = float((out - t["out"]).abs().max() / t["out"].abs().max())
    print(f"[{name}] bf16 forward max-abs relative error {err:.3e}")
    assert err < 1.5e-2, err
    assert torch.equal(run.idx.cpu(), t["idx"])


@pytest.mark.parametrize("name", ["ave_train", "avs_train_noise", "avvp_train"])
def test_router_subop_matches_reference_vectors(name):
    """avmoe_router_forward (the router alone through the C ABI) on the means the full forward computed: probabilities within
    1e-5 of the reference vectors, first-max argmax bit-exact, LB loss where the variant has one."""
    import ctypes as C
    from avmoe_amd import _capi as capi
    from tests.moe_gpu_util import MoeRun
    meta, cfg, t = load_golden(name)
    P, B = split_params(t)
    noise = t.get("noise")
    run = MoeRun(cfg, P, B, t["X"], t["Y"], bf16=False, training=bool(meta["module_train"]), noise=noise, mha_keep=mha_keep_of(t)).forward()
    rin = run.buf("rin").to(run.dev).contiguous()                    # (S, 2C): [mean_n x | mean_n remap(y)]
    probs = torch.empty_like(run.probs)
    idx = torch.empty_like(run.idx)
    lb = torch.zeros(1, device=run.dev)
    st = run.L.avmoe_router_forward(C.byref(run.desc), rin.data_ptr(), C.byref(run.ptrs),
                                    run.noise.data_ptr() if run.noise is not None else None, probs.data_ptr(), idx.data_ptr(),
                                    lb.data_ptr(), run.saved.data_ptr(), run.scratch.data_ptr(), torch.cuda.current_stream().cuda_stream)
    capi.check(st, "avmoe_router_forward")
    torch.cuda.synchronize()
    assert torch.allclose(probs.cpu(), t["probs"], atol=1e-5)
    assert torch.equal(idx.cpu(), t["idx"].reshape(-1))
    if cfg.lb_loss:
        assert abs(float(lb) - float(t["lb"])) < 1e-4 * max(1.0, abs(float(t["lb"])))
