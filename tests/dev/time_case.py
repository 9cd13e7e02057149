"""Development: fwd + bwd time of one site through the C ABI for a few AVS configurations (v1 / v2 / none).
python scripts/time_case.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import avmoe_oracle as O
from tests.moe_gpu_util import MoeRun
from avmoe_amd import _capi as capi

def run_case(name, cfgd, S, bf16=True):
    cfg = O.AdapterConfig(**cfgd)
    P, B = O.init_params(cfg, seed=1)
    g = torch.Generator().manual_seed(0)
    X = 0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g); Y = 0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)
    G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    keep = None
    if cfg.self_attn == "v1":
        keep = {pre: (torch.rand(cfg.Nx * 4, S, S, generator=g) >= 0.2).float() / 0.8 for pre in cfg.expert_prefixes()[cfg.E_m:]}
    r = MoeRun(cfg, P, B, X, Y, bf16=bf16, training=True, mha_keep=keep)
    for _ in range(3): r.forward(); r.backward(G)
    L = capi.lib(); L.avmoe_prof_reset(); L.avmoe_prof_enable(1)
    n = 5
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r.forward(); r.backward(G)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    L.avmoe_prof_enable(0)
    rep = sorted(capi.prof_report(), key=lambda q: -q["total_ms"])
    print(f"== {name}: {dt:.3f} ms fwd+bwd (with per-kernel events)")
    for q in rep[:8]: print(f"     {q['name'][:50]:50s} x{q['calls']//n:3d} {q['total_ms']/n:8.3f} ms")

base = dict(Cx=64, Nx=3136, Cy=96, Ny=1024, reduction=8, groups=2, K=32, variant="avs", E_m=1, E_s=1, lb_loss=True)
run_case("S4 stage 0, no self attention", dict(base), 10)
run_case("S4 stage 0, v2", dict(base, self_attn="v2"), 10)
run_case("S4 stage 0, v1", dict(base, self_attn="v1"), 10)
b2 = dict(Cx=320, Nx=196, Cy=384, Ny=256, reduction=8, groups=2, K=32, variant="avs", E_m=1, E_s=1, lb_loss=True)
run_case("S4 stage 2, v2", dict(b2, self_attn="v2"), 10)
run_case("S4 stage 2, v1", dict(b2, self_attn="v1"), 10)
