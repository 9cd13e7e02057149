"""The host (CPU, C++) implementation of the ABI -- include/avmoe_host.h, avmoe_amd/csrc/host_moe.cpp: SURVEY 8(b)'s "restatement used for
no-GPU CI" -- pinned on the vectors captured from the real reference modules (tests/golden/*.npz), like the Python oracle: outputs,
probabilities, bit-exact argmax, load-balancing loss, gradients wrt both inputs and every parameter, updated BatchNorm buffers.
CPU only; all 21 fixtures (round 6: the frame-attention "v1" experts too).  (Checker-side code: the product library has no CPU path and never loads this one.)"""
import ctypes as C

import pytest
import torch

from avmoe_amd import _capi_moe as cm
from tests.golden_util import golden_names, load_golden, split_params, assert_grads_close, mha_keep_of
from tests.moe_gpu_util import make_desc

SERVED = list(golden_names())          # every fixture (round 6: the frame-attention "v1" experts too)


@pytest.fixture(scope="module")
def host():
    from avmoe_amd import build as b
    try:
        L = C.CDLL(b.build_host(verbose=False))
    except Exception as e:             # no g++ / libgomp on this box: the checker library is test infrastructure, not the product
        pytest.skip(f"libavmoe_host.so cannot be built here: {e}")
    L.avmoe_host_last_error.restype = C.c_char_p
    for f in (L.avmoe_host_moe_forward, L.avmoe_host_moe_backward):
        f.restype = C.c_int
    L.avmoe_host_moe_forward.argtypes = [C.POINTER(cm.MoeDesc), C.c_void_p, C.c_void_p, C.POINTER(cm.MoePtrs), C.c_void_p] + [C.c_void_p] * 5
    L.avmoe_host_moe_backward.argtypes = [C.POINTER(cm.MoeDesc), C.c_void_p, C.c_void_p, C.POINTER(cm.MoePtrs), C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(cm.MoePtrs)]
    return L


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def test_the_fixture_selection_covers_ave_avqa_avs():
    assert {"ave_train", "ave_eval", "ave_nobn", "ave_noln_nogate", "avqa_train", "avs_train_noise", "avs_k87_train", "avvp_train", "avvp_eval",
            "avs_v2_train", "avs_v1_train", "avs_v1_eval"} <= set(SERVED) and len(SERVED) == 21


@pytest.mark.parametrize("name", SERVED)
def test_host_implementation_matches_reference_vectors(host, name):
    meta, cfg, t = load_golden(name)
    P, B = split_params(t)
    training = bool(meta["module_train"])
    S = t["X"].shape[0]
    desc = make_desc(cfg, S, False, training)
    params = {k: v.clone().contiguous() for k, v in P.items()}
    bufs = {k: v.clone().contiguous() for k, v in B.items()}
    keep = {f"{pre}.{cm.SA_KEEP}": v.to(torch.float32).contiguous() for pre, v in (mha_keep_of(t) or {}).items()}      # "v1": the recorded dropout multipliers
    ptrs = cm.make_ptrs({**params, **bufs, **keep}, cfg.E_m, cfg.E_s)
    X, Y, G = t["X"].contiguous(), t["Y"].contiguous(), t["grad_out"].contiguous()
    noise = t["noise"].contiguous() if "noise" in t else None
    nz = noise.data_ptr() if noise is not None else None
    out = torch.empty_like(X)
    probs = torch.empty(S, cfg.E)
    idx = torch.empty(S, dtype=torch.int64)
    lb = torch.zeros(1)
    st = host.avmoe_host_moe_forward(C.byref(desc), X.data_ptr(), Y.data_ptr(), C.byref(ptrs), nz, out.data_ptr(), probs.data_ptr(), idx.data_ptr(),
                                     lb.data_ptr(), None)
    assert st == 0, host.avmoe_host_last_error()
    assert torch.equal(idx, t["idx"].reshape(-1)), "router argmax must be bit-exact"
    assert _rel(out, t["out"]) < 2e-5
    assert _rel(probs, t["probs"].reshape(S, -1)) < 1e-5
    if cfg.lb_loss:
        assert abs(float(lb) - float(t["lb"])) < 1e-4 * max(1.0, abs(float(t["lb"])))
    if training and cfg.use_bn:      # running statistics and counters advanced exactly once
        for k, v in bufs.items():
            ref = t[f"newbuffer.{k}"]
            assert torch.allclose(v.to(ref.dtype), ref, rtol=1e-5, atol=1e-6), k
    grads = {k: torch.full_like(v, float("nan")) for k, v in params.items()}
    gptrs = cm.make_ptrs(grads, cfg.E_m, cfg.E_s)
    dX, dY = torch.empty_like(X), torch.empty_like(Y)
    lbg = torch.tensor([float(meta["lb_weight"])])
    before = {k: v.clone() for k, v in bufs.items()}
    st = host.avmoe_host_moe_backward(C.byref(desc), X.data_ptr(), Y.data_ptr(), C.byref(ptrs), nz, G.data_ptr(), lbg.data_ptr() if cfg.lb_loss else None,
                                      None, dX.data_ptr(), dY.data_ptr(), C.byref(gptrs))
    assert st == 0, host.avmoe_host_last_error()
    assert all(torch.equal(v, before[k]) for k, v in bufs.items()), "the backward's recomputation must not advance the running statistics again"
    assert_grads_close({**grads, "X": dX, "Y": dY}, t, rtol=2e-4)


def test_host_refuses_what_it_does_not_serve(host):
    meta, cfg, t = load_golden("ave_train")
    desc = make_desc(cfg, t["X"].shape[0], True, True)          # bf16 activations: the host library is fp32 only
    ptrs = cm.MoePtrs()
    st = host.avmoe_host_moe_forward(C.byref(desc), t["X"].data_ptr(), t["Y"].data_ptr(), C.byref(ptrs), None, torch.empty_like(t["X"]).data_ptr(), None, None, None, None)
    assert st == -2 and b"fp32" in host.avmoe_host_last_error()
