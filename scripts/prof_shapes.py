"""Per-call-shape GEMM timing of one bench step (debug): AVMOE_PROF_SHAPES=1 python scripts/prof_shapes.py"""
import os, sys, json
os.environ["AVMOE_PROF_SHAPES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from avmoe_amd import _capi as capi
c = dict(bench.CONFIGS["cfg2"])
C_a, N_a, C_v, N_v, _ = c["pairs"][0]
c.update(C=C_a, N_a=N_a, N_v=N_v)
dev = torch.device("cuda:0")
audio, visual = bench.build_pair(c, (C_a, N_a, C_v, N_v), dev)
if len(sys.argv) > 1 and sys.argv[1].isdigit(): c["B"] = int(sys.argv[1])
S = c["B"] * c["T"]
g = torch.Generator().manual_seed(0)
tdt = torch.float32 if "f32" in sys.argv else torch.bfloat16
f_a = (0.3 * torch.randn(S, c["N_a"], c["C"], generator=g)).to(dev, tdt).requires_grad_(True)
f_v = (0.3 * torch.randn(S, c["N_v"], c["C"], generator=g)).to(dev, tdt).requires_grad_(True)
g_a = torch.randn(S, c["N_a"], c["C"], generator=g).to(dev, tdt).permute(0, 2, 1).unsqueeze(-1)
g_v = torch.randn(S, c["N_v"], c["C"], generator=g).to(dev, tdt).permute(0, 2, 1).unsqueeze(-1)
def step():
    xa, xv = f_a.permute(0, 2, 1).unsqueeze(-1), f_v.permute(0, 2, 1).unsqueeze(-1)
    oa, _ = audio(xa, xv); ov, _ = visual(xv, xa)
    torch.autograd.backward([oa, ov], [g_a, g_v])
for _ in range(3): step()
torch.cuda.synchronize()
L = capi.lib(); L.avmoe_prof_reset(); L.avmoe_prof_enable(1)
n = 3
for _ in range(n): step()
torch.cuda.synchronize(); L.avmoe_prof_enable(0)
rep = sorted(capi.prof_report(), key=lambda r: -r["total_ms"])
tot = sum(r["total_ms"] for r in rep) / n
print(f"total {tot:.3f} ms/step")
for r in rep[:90]:
    ms = r["total_ms"] / r["calls"]
    gbs = r["alg_bytes"] / r["calls"] / ms / 1e6 if r["alg_bytes"] else 0
    tf = r["flops"] / r["calls"] / ms / 1e9 if r["flops"] else 0
    print(f"{r['name'][:58]:58s} x{r['calls']//n:3d} {ms*1e3:8.1f} us  {r['total_ms']/n:7.3f} ms/step  {gbs:7.0f} GB/s {tf:7.1f} TF")
