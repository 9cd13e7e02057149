"""dev: does launching two sites' kernels from TWO host threads (one HIP stream each) raise the launch rate?  Host time of 200 forward calls of two
small sites, issued from one thread alternately and from two threads concurrently (ctypes releases the GIL inside the call).
    python tests/dev/two_thread_launch.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import avmoe_oracle as O
from tests.moe_gpu_util import MoeRun
import ctypes as C
from avmoe_amd import _capi as capi

cfg = O.AdapterConfig(Cx=768, Nx=196, Cy=768, Ny=196, reduction=12, groups=2, K=32, E_m=2, E_s=2)
P, B = O.init_params(cfg, seed=1)
g = torch.Generator().manual_seed(0)
X = 0.3 * torch.randn(20, cfg.Nx, cfg.Cx, generator=g); Y = 0.3 * torch.randn(20, cfg.Ny, cfg.Cy, generator=g)
runs = [MoeRun(cfg, P, B, X, Y, bf16=True, training=True) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
L = capi.lib()
def fwd(r, s):
    st = L.avmoe_moe_forward(C.byref(r.desc), r.X.data_ptr(), r.Y.data_ptr(), C.byref(r.ptrs), None, r.out.data_ptr(), r.probs.data_ptr(), r.idx.data_ptr(),
                             r.lb.data_ptr(), r.saved.data_ptr(), r.scratch.data_ptr(), s.cuda_stream)
    assert st == 0
for r, s in zip(runs, streams): fwd(r, s)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    fwd(runs[0], streams[0]); fwd(runs[1], streams[1])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"one thread : host {1e6 * (t1 - t0) / n:7.1f} us per pair of calls (wall incl. GPU {1e6 * (t2 - t0) / n:7.1f})")
def worker(i):
    torch.cuda.set_device(0)
    for _ in range(n): fwd(runs[i], streams[i])
ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"two threads: host {1e6 * (t1 - t0) / n:7.1f} us per pair of calls (wall incl. GPU {1e6 * (t2 - t0) / n:7.1f})")
