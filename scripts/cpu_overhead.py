"""dev: host time per bench step (enqueue only) vs the GPU-synchronised step time"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from avmoe_amd.adapters import AdapterPair
from avmoe_amd.dp import AdapterGradReducer

c = dict(bench.CONFIGS["cfg2"])
Ca, Na, Cv, Nv, _ = c["pairs"][0]
c.update(C=Ca, N_a=Na, N_v=Nv)
dev = torch.device("cuda:0")
audio, visual = bench.build_pair(c, (Ca, Na, Cv, Nv), dev)
red = AdapterGradReducer(list(audio.parameters()) + list(visual.parameters()), sites=[audio, visual])
S = c["B"] * c["T"]
g = torch.Generator().manual_seed(0)
fa = (0.3 * torch.randn(S, c["N_a"], c["C"], generator=g)).to(dev, torch.bfloat16).requires_grad_(True)
fv = (0.3 * torch.randn(S, c["N_v"], c["C"], generator=g)).to(dev, torch.bfloat16).requires_grad_(True)
ga = torch.randn(S, c["N_a"], c["C"], generator=g).to(dev, torch.bfloat16).permute(0, 2, 1).unsqueeze(-1)
gv = torch.randn(S, c["N_v"], c["C"], generator=g).to(dev, torch.bfloat16).permute(0, 2, 1).unsqueeze(-1)
for conc in (True, False):
    pair = AdapterPair(audio, visual, concurrent=conc)
    def step():
        red.begin(True)
        oa, _, ov, _ = pair(fa.permute(0, 2, 1).unsqueeze(-1), fv.permute(0, 2, 1).unsqueeze(-1))
        torch.autograd.backward([oa, ov], [ga, gv])
        red.finish(); fa.grad = None; fv.grad = None; red.zero_grad()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"concurrent={conc}: host enqueue {1e3 * (t1 - t0) / 20:.2f} ms/step, with GPU drain {1e3 * (t2 - t0) / 20:.2f} ms/step")
