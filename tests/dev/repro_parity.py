"""Development: bench.py's parity leg for one configuration, per site, through BOTH harnesses (module API / MoeRun on the C ABI).
python tests/dev/repro_parity.py cfg4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from oracle import avmoe_oracle as O
from tests.moe_gpu_util import MoeRun

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
c = dict(bench.CONFIGS[name], name=name)
dev = torch.device("cuda:0")
_, (work, lbw) = bench.cpu_baseline(c, budget_s=0.0)
for w in work:
    for tag, cfg, P, B, X, Y, G, ref in (("audio", w["ca"], w["Pa"], w["Ba"], w["fa"], w["fv"], w["ga"], w["ra"]),
                                         ("visual", w["cv"], w["Pv"], w["Bv"], w["fv"], w["fa"], w["gv"], w["rv"])):
        fwd, grads = ref
        gmax = max(float(v.abs().max()) for v in grads.values())
        run = MoeRun(cfg, P, B, X, Y, bf16=False, training=True).forward()
        got = run.backward(G, lb_weight=lbw)
        rel = {k: float((got[k] - v).abs().max()) / max(float(v.abs().max()), 1e-3 * gmax) for k, v in grads.items()}
        k1 = max(rel, key=rel.get)
        m = bench.new_site(c, cfg.Cx, cfg.Nx, cfg.Cy, cfg.Ny)
        m.load_state_dict({**P, **B}); m.to(dev).train()
        Xd, Yd = X.to(dev).requires_grad_(True), Y.to(dev).requires_grad_(True)
        r = m(Xd.permute(0, 2, 1).unsqueeze(-1), Yd.permute(0, 2, 1).unsqueeze(-1))
        out = r[0].squeeze(-1).permute(0, 2, 1)
        loss = (out * G.to(dev)).sum()
        lb = r[-1] if c["variant"] in ("avvp", "avs") else None
        if torch.is_tensor(lb) and lbw:
            loss = loss + lbw * lb
        loss.backward(); torch.cuda.synchronize()
        g2 = {k: p.grad.cpu() for k, p in m.named_parameters()}; g2["X"], g2["Y"] = Xd.grad.cpu(), Yd.grad.cpu()
        rel2 = {k: float((g2[k] - v).abs().max()) / max(float(v.abs().max()), 1e-3 * gmax) for k, v in grads.items()}
        k2 = max(rel2, key=rel2.get)
        print(f"{tag:6s} C={cfg.Cx:4d} N={cfg.Nx:4d}  C-ABI worst {rel[k1]:.2e} ({k1[:36]})   module worst {rel2[k2]:.2e} ({k2[:36]})"
              f"   out {float((out.detach().cpu() - fwd['out']).abs().max() / fwd['out'].abs().max()):.1e}", flush=True)
