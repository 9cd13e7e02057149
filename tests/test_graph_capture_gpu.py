"""The C-ABI calls allocate nothing and never synchronise the host, so a forward + backward on caller-owned workspaces can be
captured into a HIP graph (include/avmoe.h): the replay is bit-identical to the eager call."""
import ctypes as C

import pytest
import torch

from oracle import avmoe_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bf16", [False, True])
def test_forward_backward_capture_and_replay(bf16):
    from tests.moe_gpu_util import MoeRun
    from avmoe_amd import _capi as capi
    cfg = O.AdapterConfig(Cx=128, Nx=150, Cy=64, Ny=50, reduction=2, groups=2, K=32)      # register-resident shape
    S = 4
    P, B = O.init_params(cfg, seed=1)
    g = torch.Generator().manual_seed(0)
    X = 0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    Y = 0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)
    G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    run = MoeRun(cfg, P, B, X, Y, bf16=bf16, training=False)       # eval: no running-statistics update between the two runs
    run.forward()
    ref = run.backward(G)
    ref_out = run.out.clone()
    L = run.L

    def calls(stream):
        capi.check(L.avmoe_moe_forward(C.byref(run.desc), run.X.data_ptr(), run.Y.data_ptr(), C.byref(run.ptrs), None,
                                       run.out.data_ptr(), run.probs.data_ptr(), run.idx.data_ptr(), run.lb.data_ptr(),
                                       run.saved.data_ptr(), run.scratch.data_ptr(), stream), "forward")
        capi.check(L.avmoe_moe_backward(C.byref(run.desc), run.X.data_ptr(), run.Y.data_ptr(), C.byref(run.ptrs),
                                        run.dOut.data_ptr(), None, run.saved.data_ptr(), run.scratch.data_ptr(),
                                        run.dX.data_ptr(), run.dY.data_ptr(), C.byref(run.gptrs), stream), "backward")

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        calls(s.cuda_stream)                                       # warm-up on the capture stream
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        calls(torch.cuda.current_stream().cuda_stream)
    for t in (run.out, run.dX, run.dY, *run.grads.values()):
        t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(run.out, ref_out)
    assert torch.equal(run.dX.float().cpu(), ref["X"]) and torch.equal(run.dY.float().cpu(), ref["Y"])
    for k, v in run.grads.items():
        assert torch.equal(v.cpu(), ref[k]), k
