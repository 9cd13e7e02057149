"""Development: gradient error of the HIP path vs the oracle for variations of one configuration (fp32).
python tests/dev/bisect_cfg.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import avmoe_oracle as O
from tests.moe_gpu_util import MoeRun
from tests.golden_util import grad_errors

base = dict(Cx=240, Nx=17, Cy=48, Ny=20, E_m=1, E_s=2, reduction=3, groups=2, K=32, use_bn=True, use_gate=True, ln_before=False,
            ln_post=True, variant="avs", self_attn="v2")
def run(tag, S=2, seed=722, **kw):
    cfg = O.AdapterConfig(**{**base, **kw})
    P, B = O.init_params(cfg, seed=seed)
    g = torch.Generator().manual_seed(1000 + seed)
    X = 0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g); Y = 0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)
    G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    fwd, grads = O.moe_forward_backward(P, B, X, Y, cfg, G, training=True)
    r = MoeRun(cfg, P, B, X, Y, bf16=False, training=True).forward()
    fe = float((r.out.float().cpu() - fwd["out"]).abs().max() / fwd["out"].abs().max())
    got = r.backward(G)
    errs = grad_errors(got, {f"grad.{k}": v for k, v in grads.items()})
    gmax = max(s for _, s in errs.values())
    worst = max(errs.items(), key=lambda kv: kv[1][0] / max(kv[1][1], 1e-3 * gmax))
    print(f"{tag:38s} d={cfg.d:3d} fwd {fe:.1e}  worst grad {worst[0][:40]:40s} {worst[1][0] / max(worst[1][1], 1e-3 * gmax):.2e}")

run("as found")
run("ln_before", ln_before=True)
run("no self attention", self_attn="none")
run("variant ave", variant="ave", self_attn="none")
run("K=8", K=8)
run("d=60 (r=4)", reduction=4)
run("d=120 (r=2)", reduction=2)
run("Cx=192 d=64", Cx=192)
run("E 2+2", E_m=2, E_s=2)
run("E 1+1", E_m=1, E_s=1)
run("Nx=40", Nx=40)
run("S=4", S=4)
run("no bn", use_bn=False)
run("no ln_post", ln_post=False)
run("other seed", seed=5)
