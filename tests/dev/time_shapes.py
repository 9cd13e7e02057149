"""Development: fwd + bwd time of ONE MoEAdapter site (autograd facade, bf16, S = 320 frames) on the register-resident shape and
on neighbouring shapes that take the generic kt_* kernels.  python tests/dev/time_shapes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import avmoe_oracle as O
from tests.test_adapters_api import build_module

dev = torch.device("cuda:0")
def run(name, S=320, **kw):
    cfg = O.AdapterConfig(**kw)
    m = build_module("ave" if cfg.variant == "ave" else cfg.variant, cfg).to(dev).train()
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith(("gate", "gate_av")): p.fill_(0.3)
    g = torch.Generator().manual_seed(0)
    X = (0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g)).to(dev, torch.bfloat16).requires_grad_(True)
    Y = (0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)).to(dev, torch.bfloat16).requires_grad_(True)
    G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g).to(dev, torch.bfloat16).permute(0, 2, 1).unsqueeze(-1)
    def step():
        for p in m.parameters(): p.grad = None
        X.grad = Y.grad = None
        out = m(X.permute(0, 2, 1).unsqueeze(-1), Y.permute(0, 2, 1).unsqueeze(-1))[0]
        out.backward(G)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    gb = 5.0 * S * cfg.Nx * cfg.Cx * 2 / 1e9
    print(f"{name:44s} {dt:8.3f} ms   ({gb / dt * 1e3:6.0f} GB/s of the 5-pass ideal over X)")

base = dict(Cx=768, Nx=1024, Cy=768, Ny=196, groups=2, K=32, variant="ave")
run("cfg-2 audio site  d=64 (register-resident)", reduction=12, **base)
run("same, d=96 (r=8)        generic kernels", reduction=8, **base)
run("same, d=32 (r=24)       generic kernels", reduction=24, **base)
run("same, K=16              generic kernels", reduction=12, **{**base, "K": 16})
run("same, 1+1 experts       generic kernels", reduction=12, E_m=1, E_s=1, **base)
run("same, 4 groups          generic kernels", reduction=12, **{**base, "groups": 4})
