"""dev experiment: forward of the two cfg-2 sites sequentially vs on two HIP streams"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from avmoe_amd import adapters as A

c = dict(bench.CFG2)
dev = torch.device("cuda:0")
audio, visual = bench.build_site(c, dev)
S = c["B"] * c["T"]
g = torch.Generator().manual_seed(0)
fa = (0.3 * torch.randn(S, c["N_a"], c["C"], generator=g)).to(dev, torch.bfloat16)
fv = (0.3 * torch.randn(S, c["N_v"], c["C"], generator=g)).to(dev, torch.bfloat16)
Pa, Pb = audio._param_tensors(), visual._param_tensors()
s2 = torch.cuda.Stream()

def seq():
    with torch.no_grad():
        A._site_forward(audio, fa, fv, None, tuple(Pa), tuple(Pa.values()))
        A._site_forward(visual, fv, fa, None, tuple(Pb), tuple(Pb.values()))

def par():
    with torch.no_grad():
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            A._site_forward(visual, fv, fa, None, tuple(Pb), tuple(Pb.values()))
        A._site_forward(audio, fa, fv, None, tuple(Pa), tuple(Pa.values()))
        torch.cuda.current_stream().wait_stream(s2)

for name, fn in (("sequential", seq), ("two streams", par), ("sequential", seq), ("two streams", par)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{name:12s} forward both sites: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
