"""dev: where the HOST time of a launch-bound step goes -- cProfile over AdapterPair fwd + bwd steps.
python tests/dev/host_profile.py [B of cfg-2, default 2]   |   python tests/dev/host_profile.py cfg1"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

arg = sys.argv[1] if len(sys.argv) > 1 else "2"
if arg.startswith("cfg"):
    c = dict(bench.CONFIGS[arg], name=arg); B = c["B"]; n1, n2 = 10, 20
else:
    B = int(arg); c = dict(bench.CONFIGS["cfg2"], name="cfg2"); c["B"] = B; n1, n2 = 100, 200
dev = torch.device("cuda:0")
wl = bench.Workload(c, torch.bfloat16 if c["dtype"] == "bf16" else torch.float32, dev, 0, 1, "concurrent")
for _ in range(5): wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n1): wl.step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{c['name']} B={B}: host enqueue {1e3 * (t1 - t0) / n1:.3f} ms/step, with drain {1e3 * (t2 - t0) / n1:.3f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(n2): wl.step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
