"""dev: does the host run ahead of the GPU?  Enqueue time of each step (no synchronisation inside) against the GPU time per step:
python tests/dev/host_ahead.py [steps] [cfg2|cfg1|...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
name = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
c = dict(bench.CONFIGS[name], name=name)
dev = torch.device("cuda:0")
wl = bench.Workload(c, torch.bfloat16 if c["dtype"] == "bf16" else torch.float32, dev, 0, 1, "concurrent")
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
enq = []
t_all = time.perf_counter()
for _ in range(n):
    t0 = time.perf_counter()
    wl.step()
    enq.append(time.perf_counter() - t0)
t_enq = time.perf_counter() - t_all
torch.cuda.synchronize()
t_tot = time.perf_counter() - t_all
print("enqueue ms per step:", " ".join(f"{1e3 * e:.2f}" for e in enq))
print(f"all enqueued after {1e3 * t_enq:.1f} ms, GPU done after {1e3 * t_tot:.1f} ms  ({1e3 * t_tot / n:.2f} ms per step)")
