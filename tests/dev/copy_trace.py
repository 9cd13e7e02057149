"""dev: which Python lines issue device copies (aten::copy_ -> __amd_rocclr_copyBuffer) in one bench step:   python tests/dev/copy_trace.py [cfg]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
c = dict(bench.CONFIGS[name], name=name)
dev = torch.device("cuda", 0)
wl = bench.Workload(c, torch.bfloat16 if c["dtype"] == "bf16" else torch.float32, dev, 0, 1, "concurrent")
for _ in range(2):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    wl.step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::clone", "aten::contiguous"):
        st = [s for s in (ev.stack or []) if "site-packages" not in s and "python3" not in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (n, where), k in cnt.most_common(25):
    print(f"{k:6d}  {n:14s} {where}")
