"""Development: the cfg-4 stage-2 audio-side shape (AVQA: C = 384, 4 groups, bottleneck 48, 2 latent tokens, 1 + 2 experts) vs the
oracle in fp32, with variations -- bench.py's parity leg found 12 % on d X at S = 20.   python tests/dev/bisect_cfg4.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import avmoe_oracle as O
from tests.moe_gpu_util import MoeRun
from tests.golden_util import grad_errors

base = dict(Cx=384, Nx=256, Cy=768, Ny=144, E_m=1, E_s=2, reduction=8, groups=4, K=2, variant="avqa")
def run(tag, S=20, seed=0, show=0, **kw):
    cfg = O.AdapterConfig(**{**base, **kw})
    P, B = O.init_params(cfg, seed=seed)
    g = torch.Generator().manual_seed(1234)
    X = 0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g); Y = 0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)
    G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    fwd, grads = O.moe_forward_backward(P, B, X, Y, cfg, G, training=True)
    r = MoeRun(cfg, P, B, X, Y, bf16=False, training=True).forward()
    fe = float((r.out.float().cpu() - fwd["out"]).abs().max() / fwd["out"].abs().max())
    got = r.backward(G)
    errs = grad_errors(got, {f"grad.{k}": v for k, v in grads.items()})
    gmax = max(s for _, s in errs.values())
    rel = {k: e / max(s, 1e-3 * gmax) for k, (e, s) in errs.items()}
    worst = max(rel.items(), key=lambda kv: kv[1])
    print(f"{tag:34s} d={cfg.d:3d} fwd {fe:.1e}  worst grad {worst[0][:40]:40s} {worst[1]:.2e}   #>1e-3: {sum(v > 1e-3 for v in rel.values())}", flush=True)
    if show:
        for k, v in sorted(rel.items(), key=lambda kv: -kv[1])[:show]:
            print(f"      {k:46s} {v:.3e}")

run("as in bench (S=20, seed 0)", show=8)
run("S=2", S=2)
run("S=8", S=8)
run("seed 4", seed=4)
run("K=8", K=8)
run("K=32", K=32)
run("groups 2", groups=2)
run("E 2+2", E_m=2, E_s=2)
run("variant ave", variant="ave")
run("Nx=128", Nx=128)
run("Ny=64", Ny=64)
run("Cy=384", Cy=384)
run("r=4 (d=96)", reduction=4)
run("Cx=768 (d=96, dg=24)", Cx=768)
