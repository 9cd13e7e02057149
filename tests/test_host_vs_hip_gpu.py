"""HIP path (fp32) against the host (CPU, C++) implementation of the same ABI -- include/avmoe_host.h: an independent statement of the
reference's arithmetic that involves neither torch's autograd nor the re-factorised algebra -- on seeded mid-size inputs."""
import ctypes as C

import pytest
import torch

from avmoe_amd import _capi_moe as cm
from oracle import avmoe_oracle as O

pytestmark = pytest.mark.gpu

CASES = {
    "ave_fast": (dict(Cx=128, Nx=150, Cy=64, Ny=50, reduction=2, groups=2, K=32, variant="ave"), 4),
    "avqa_merged": (dict(Cx=96, Nx=200, Cy=192, Ny=120, reduction=8, groups=4, K=2, variant="avqa", E_m=1, E_s=2), 3),
    "avvp_nxn": (dict(Cx=96, Nx=128, Cy=64, Ny=70, reduction=4, groups=2, K=8, variant="avvp", lb_loss=True), 3),
    "avs_v2_k20": (dict(Cx=96, Nx=77, Cy=128, Ny=60, reduction=3, groups=2, K=20, variant="avs", self_attn="v2", lb_loss=True), 4),
}


@pytest.mark.parametrize("name", list(CASES))
def test_hip_fp32_matches_the_host_implementation(name):
    from avmoe_amd import build as b
    from tests.moe_gpu_util import MoeRun, make_desc
    try:
        H = C.CDLL(b.build_host(verbose=False))
    except Exception as e:             # no g++ / libgomp on this box: the checker library is test infrastructure, not the product
        pytest.skip(f"libavmoe_host.so cannot be built here: {e}")
    H.avmoe_host_last_error.restype = C.c_char_p
    H.avmoe_host_moe_forward.argtypes = [C.POINTER(cm.MoeDesc), C.c_void_p, C.c_void_p, C.POINTER(cm.MoePtrs), C.c_void_p] + [C.c_void_p] * 5
    H.avmoe_host_moe_backward.argtypes = [C.POINTER(cm.MoeDesc), C.c_void_p, C.c_void_p, C.POINTER(cm.MoePtrs), C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(cm.MoePtrs)]
    kw, S = CASES[name]
    cfg = O.AdapterConfig(**kw)
    P, B = O.init_params(cfg, seed=11)
    g = torch.Generator().manual_seed(12)
    X = 0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    Y = 0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)
    G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    lbw = 0.01 if cfg.lb_loss else 0.0
    # host
    params = {k: v.clone().contiguous() for k, v in P.items()}
    bufs = {k: v.clone().contiguous() for k, v in B.items()}
    ptrs = cm.make_ptrs({**params, **bufs}, cfg.E_m, cfg.E_s)
    desc = make_desc(cfg, S, False, True)
    out_h, probs_h, idx_h, lb_h = torch.empty_like(X), torch.empty(S, cfg.E), torch.empty(S, dtype=torch.int64), torch.zeros(1)
    assert H.avmoe_host_moe_forward(C.byref(desc), X.data_ptr(), Y.data_ptr(), C.byref(ptrs), None, out_h.data_ptr(), probs_h.data_ptr(), idx_h.data_ptr(),
                                    lb_h.data_ptr(), None) == 0, H.avmoe_host_last_error()
    grads_h = {k: torch.zeros_like(v) for k, v in params.items()}
    gp = cm.make_ptrs(grads_h, cfg.E_m, cfg.E_s)
    dX, dY, lbg = torch.empty_like(X), torch.empty_like(Y), torch.tensor([lbw])
    assert H.avmoe_host_moe_backward(C.byref(desc), X.data_ptr(), Y.data_ptr(), C.byref(ptrs), None, G.data_ptr(), lbg.data_ptr() if cfg.lb_loss else None, None,
                                     dX.data_ptr(), dY.data_ptr(), C.byref(gp)) == 0, H.avmoe_host_last_error()
    # HIP
    run = MoeRun(cfg, P, B, X, Y, bf16=False, training=True).forward()
    assert torch.equal(run.idx.cpu(), idx_h)
    assert float((run.out.float().cpu() - out_h).abs().max() / out_h.abs().max()) < 1e-3
    assert float((run.probs.cpu() - probs_h).abs().max()) < 1e-5
    got = run.backward(G, lb_weight=lbw)
    ref = {**grads_h, "X": dX, "Y": dY}
    gmax = max(float(v.abs().max()) for v in ref.values())
    bad = {k: float((got[k].float().cpu() - v).abs().max()) for k, v in ref.items()
           if float((got[k].float().cpu() - v).abs().max()) > 1e-3 * max(float(v.abs().max()), 1e-3 * gmax)}
    assert not bad, bad
    assert run.guards_intact()
