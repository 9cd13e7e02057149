"""GPU parity of the C ABI's sub-ops (include/avmoe.h, ABI 7) -- one module of the reference at a time:
avmoe_expert_forward_cross / _uni against ExpertAdapter.forward (net_trans_v3.py:377-435; mgn.py:132-139 and
PVT_AVSModel_v2.py:210-227 for the unimodal variants) and avmoe_remap_forward against `vis_token` of net_trans_v3.py:469-471,
on the inputs / parameters of the reference-generated fixtures, the expected values from oracle/avmoe_oracle.py (pinned on those
fixtures: tests/test_oracle_golden.py)."""
import ctypes as C

import pytest
import torch

from oracle import avmoe_oracle as oracle
from tests.golden_util import load_golden, split_params, mha_keep_of

pytestmark = pytest.mark.gpu

CASES = ["ave_train", "ave_eval", "ave_nobn", "ave_noln_nogate", "ave_swap_train", "avqa_train", "avvp_train", "avvp_eval",
         "avs_v2_train", "avs_v1_eval", "avs_k87_train"]


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("name", CASES)
def test_expert_subops_match_expert_forward(name):
    """Every expert of the site alone: out_e within 1e-3 (max-abs relative, fp32) of the oracle's ExpertAdapter.forward on the
    oracle's own remapped tokens; the mixture of the sub-op outputs with the site's probabilities gives the site's output."""
    from avmoe_amd import _capi as capi
    from tests.moe_gpu_util import MoeRun
    meta, cfg, t = load_golden(name)
    P, B = split_params(t)
    training = bool(meta["module_train"])
    keep = mha_keep_of(t)
    X, Y = t["X"], t["Y"]
    with torch.no_grad():
        Wc = P["conv_adapter.weight"][:, :, 0, 0]
        Yf = (torch.einsum("nm,smc->snc", Wc, Y) + P["conv_adapter.bias"][None, :, None]) @ P["fc.weight"].t() + P["fc.bias"]
    mix = torch.zeros_like(t["out"])
    for e, pre in enumerate(cfg.expert_prefixes()):
        with torch.no_grad():
            ref = oracle.expert_forward(P, B, pre, X, Yf, cfg, e < cfg.E_m, training, None, keep)
        run = MoeRun(cfg, P, B, X, Y, bf16=False, training=training, mha_keep=keep)       # fresh buffers: the call advances them
        out = torch.full_like(run.X, float("nan"))
        fn, j = (run.L.avmoe_expert_forward_cross, e) if e < cfg.E_m else (run.L.avmoe_expert_forward_uni, e - cfg.E_m)
        st = fn(C.byref(run.desc), run.X.data_ptr(), run.Y.data_ptr(), C.byref(run.ptrs), j, out.data_ptr(), run.saved.data_ptr(),
                run.scratch.data_ptr(), _stream())
        capi.check(st, "avmoe_expert_forward")
        torch.cuda.synchronize()
        assert run.guards_intact()
        got = out.cpu()
        scale = float(ref.abs().max())
        assert scale > 0
        err = float((got - ref).abs().max()) / scale
        assert err < 1e-3, (pre, err)
        mix += t["probs"][:, e].reshape(-1, 1, 1) * got
    assert float((mix - t["out"]).abs().max() / t["out"].abs().max()) < 1e-3


def test_expert_subop_advances_only_its_own_running_statistics():
    """Training mode: the sub-op is ExpertAdapter.forward of ONE module -- only that expert's BatchNorm running statistics move
    (to what the site forward gives them); the other experts' buffers stay bit for bit what they were (round-3 advisor finding:
    a module-by-module comparison advanced every module E times)."""
    from avmoe_amd import _capi as capi
    from tests.moe_gpu_util import MoeRun
    meta, cfg, t = load_golden("ave_train")
    P, B = split_params(t)
    full = MoeRun(cfg, P, B, t["X"], t["Y"], bf16=False, training=True).forward()         # the site forward: every expert advances
    for e, pre in enumerate(cfg.expert_prefixes()):
        run = MoeRun(cfg, P, B, t["X"], t["Y"], bf16=False, training=True)
        before = {k: v.clone() for k, v in run.buffers.items()}
        out = torch.empty_like(run.X)
        fn, j = (run.L.avmoe_expert_forward_cross, e) if e < cfg.E_m else (run.L.avmoe_expert_forward_uni, e - cfg.E_m)
        capi.check(fn(C.byref(run.desc), run.X.data_ptr(), run.Y.data_ptr(), C.byref(run.ptrs), j, out.data_ptr(), run.saved.data_ptr(),
                      run.scratch.data_ptr(), _stream()), "avmoe_expert_forward")
        torch.cuda.synchronize()
        assert run.guards_intact()
        moved = 0
        for k, v in run.buffers.items():
            if k.startswith(pre + "."):
                assert torch.allclose(v, full.buffers[k], rtol=1e-5, atol=1e-7), k      # what the site forward leaves for this expert
                moved += int(not torch.equal(v, before[k]))
            else:
                assert torch.equal(v, before[k]), (pre, k)
        assert moved >= 4                                                                   # bn1 / bn2 running_mean and running_var


def test_expert_subop_rejects_bad_index():
    from avmoe_amd import _capi as capi
    from tests.moe_gpu_util import MoeRun
    meta, cfg, t = load_golden("ave_train")
    P, B = split_params(t)
    run = MoeRun(cfg, P, B, t["X"], t["Y"], bf16=False, training=False)
    out = torch.empty_like(run.X)
    for fn, j in ((run.L.avmoe_expert_forward_cross, cfg.E_m), (run.L.avmoe_expert_forward_uni, -1)):
        st = fn(C.byref(run.desc), run.X.data_ptr(), run.Y.data_ptr(), C.byref(run.ptrs), j, out.data_ptr(), run.saved.data_ptr(),
                run.scratch.data_ptr(), _stream())
        assert st == -1                                       # AVMOE_ERR_BAD_ARG
        assert b"expert" in run.L.avmoe_last_error()


@pytest.mark.parametrize("name,bf16", [("ave_train", False), ("ave_swap_train", False), ("avvp_train", False), ("avs_k87_train", False),
                                       ("ave_wide_train", True)])
def test_remap_subop_matches_vis_token(name, bf16):
    """conv_adapter then fc, materialised: Yt and Yf within 1e-3 (fp32; bf16: 1.5e-2 on rounded inputs) of the reference's arithmetic."""
    from avmoe_amd import _capi as capi
    from tests.moe_gpu_util import MoeRun
    meta, cfg, t = load_golden(name)
    P, B = split_params(t)
    run = MoeRun(cfg, P, B, t["X"], t["Y"], bf16=bf16, training=False)
    S, N, C_, Cy = run.X.shape[0], run.X.shape[1], run.X.shape[2], run.Y.shape[2]
    Yt = torch.full((S, N, Cy), float("nan"), device=run.dev, dtype=run.tdt)
    Yf = torch.full((S, N, C_), float("nan"), device=run.dev, dtype=run.tdt)
    st = run.L.avmoe_remap_forward(C.byref(run.desc), run.Y.data_ptr(), C.byref(run.ptrs), Yt.data_ptr(), Yf.data_ptr(),
                                   run.saved.data_ptr(), run.scratch.data_ptr(), _stream())
    capi.check(st, "avmoe_remap_forward")
    torch.cuda.synchronize()
    assert run.guards_intact()
    with torch.no_grad():
        Yin = run.Y.float().cpu()                                                         # (the rounded inputs on the bf16 path)
        Wc = P["conv_adapter.weight"][:, :, 0, 0]
        rt = torch.einsum("nm,smc->snc", Wc, Yin) + P["conv_adapter.bias"][None, :, None]  # net_trans_v3.py:469
        rf = rt @ P["fc.weight"].t() + P["fc.bias"]                                         # :470
    tol = 1.5e-2 if bf16 else 1e-3
    assert float((Yt.float().cpu() - rt).abs().max() / rt.abs().max()) < tol
    assert float((Yf.float().cpu() - rf).abs().max() / rf.abs().max()) < tol
    if not bf16:                                                                            # the oracle's own Yf, for good measure
        with torch.no_grad():
            fwd = oracle.moe_forward(P, B, t["X"], t["Y"], cfg, training=False, update_buffers=False, mha_keep=mha_keep_of(t))
        assert float((Yf.cpu() - fwd["Yf"]).abs().max() / fwd["Yf"].abs().max()) < 1e-3
