"""Development: capture avmoe_moe_forward + avmoe_moe_backward (C ABI, one stream, caller-owned workspaces) into a HIP graph and
replay it.  python scripts/graph_capture_cabi.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import avmoe_oracle as O
from tests.moe_gpu_util import MoeRun
from avmoe_amd import _capi as capi

cfg = O.AdapterConfig(Cx=768, Nx=256, Cy=768, Ny=196, reduction=12, groups=2, K=32)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P, B = O.init_params(cfg, seed=1)
g = torch.Generator().manual_seed(0)
X = 0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g); Y = 0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)
G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
run = MoeRun(cfg, P, B, X, Y, bf16=True, training=True)
run.forward(); ref_g = run.backward(G); ref_out = run.out.clone()
L = run.L

def calls(stream):
    st = L.avmoe_moe_forward(C.byref(run.desc), run.X.data_ptr(), run.Y.data_ptr(), C.byref(run.ptrs), None, run.out.data_ptr(),
                             run.probs.data_ptr(), run.idx.data_ptr(), run.lb.data_ptr(), run.saved.data_ptr(), run.scratch.data_ptr(), stream)
    capi.check(st, "fwd")
    st = L.avmoe_moe_backward(C.byref(run.desc), run.X.data_ptr(), run.Y.data_ptr(), C.byref(run.ptrs), run.dOut.data_ptr(), None,
                              run.saved.data_ptr(), run.scratch.data_ptr(), run.dX.data_ptr(), run.dY.data_ptr(), C.byref(run.gptrs), stream)
    capi.check(st, "bwd")

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

s = torch.cuda.Stream()
with torch.cuda.stream(s):
    calls(s.cuda_stream)
torch.cuda.synchronize()
t_eager = timeit(lambda: calls(torch.cuda.current_stream().cuda_stream))
# BatchNorm running statistics move with every call: restore, then compare one replay with one eager call
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=s):
    calls(torch.cuda.current_stream().cuda_stream)
run.dX.zero_(); run.out.zero_()
graph.replay(); torch.cuda.synchronize()
e1 = float((run.out.float() - ref_out.float()).abs().max()); e2 = float((run.dX.float().cpu() - ref_g["X"]).abs().max())
t_graph = timeit(graph.replay)
print(f"S={S}: eager {t_eager:.3f} ms, graph replay {t_graph:.3f} ms, |out diff| {e1:.2e}, |dX diff| {e2:.2e}")
