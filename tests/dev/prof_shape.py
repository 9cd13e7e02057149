"""Development: kernel families of one site step at a given shape.  python tests/dev/prof_shape.py reduction=8 K=32 ..."""
import os, sys
os.environ["AVMOE_PROF_SHAPES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import avmoe_oracle as O
from tests.test_adapters_api import build_module
from avmoe_amd import _capi as capi
kw = dict(Cx=768, Nx=1024, Cy=768, Ny=196, groups=2, K=32, variant="ave", reduction=12)
S = 320
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "S": S = int(v)
    else: kw[k] = int(v) if v.lstrip("-").isdigit() else v
cfg = O.AdapterConfig(**kw)
dev = torch.device("cuda:0")
m = build_module(cfg.variant, cfg).to(dev).train()
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
    m(X.permute(0, 2, 1).unsqueeze(-1), Y.permute(0, 2, 1).unsqueeze(-1))[0].backward(G)
for _ in range(3): step()
torch.cuda.synchronize()
L = capi.lib(); L.avmoe_prof_reset(); L.avmoe_prof_enable(1)
n = 3
for _ in range(n): step()
torch.cuda.synchronize(); L.avmoe_prof_enable(0)
rep = sorted(capi.prof_report(), key=lambda r: -r["total_ms"])
print(f"total {sum(r['total_ms'] for r in rep) / n:.3f} ms/step   {kw}")
for r in rep[:28]:
    ms = r["total_ms"] / r["calls"]
    gbs = r["alg_bytes"] / r["calls"] / ms / 1e6 if r["alg_bytes"] else 0
    print(f"{r['name'][:58]:58s} x{r['calls']//n:3d} {ms*1e3:8.1f} us  {r['total_ms']/n:7.3f} ms/step  {gbs:7.0f} GB/s")
