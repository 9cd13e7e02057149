"""dev: where does the bf16 router-gradient error of a no-BatchNorm register-resident site come from?  HIP bf16 vs HIP fp32 vs oracle."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import avmoe_oracle as O          # noqa: E402
from tests.moe_gpu_util import MoeRun         # noqa: E402

def rn(a, b): return float((a - b).norm() / b.norm().clamp_min(1e-30))

for name, kw in (("fast_nobn", dict(use_bn=False)), ("fast_bn", dict()), ("fast_nobn_noln", dict(use_bn=False, ln_post=False))):
    cfg = O.AdapterConfig(Cx=128, Nx=97, Cy=64, Ny=40, reduction=2, groups=2, K=32, variant="ave", **kw)
    S = 4
    P, B = O.init_params(cfg, seed=21)
    g = torch.Generator().manual_seed(77)
    X = 0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g); Y = 0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g); G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    Xb, Yb, Gb = X.bfloat16().float(), Y.bfloat16().float(), G.bfloat16().float()
    fwd, grads = O.moe_forward_backward(P, B, Xb, Yb, cfg, Gb, training=True)
    r32 = MoeRun(cfg, P, B, Xb, Yb, bf16=False, training=True).forward(); g32 = r32.backward(Gb)
    dp32 = r32.buf("dp", shape=(S, cfg.E))
    r16 = MoeRun(cfg, P, B, X, Y, bf16=True, training=True).forward(); g16 = r16.backward(G)
    dp16 = r16.buf("dp", shape=(S, cfg.E))
    print(f"== {name}: gram64-mode bf16")
    print("  out bf16 vs oracle", rn(r16.out.float().cpu(), fwd["out"]), " f32 vs oracle", rn(r32.out.float().cpu(), fwd["out"]))
    print("  dp bf16 vs f32", rn(dp16, dp32), "\n  dp32", dp32[0], "\n  dp16", dp16[0])
    for k in ("router.0.weight", "router.4.bias", "X", "Y", "multimodal_experts.0.gate", "singlemodal_experts.0.gate", "multimodal_experts.0.ln_post.weight",
              "multimodal_experts.0.up_sampler.weight", "singlemodal_experts.0.up_sampler.weight"):
        if k in grads:
            print(f"  {k:44s} bf16 {rn(g16[k], grads[k]):.3e}   f32 {rn(g32[k], grads[k]):.3e}")
