"""dev: where the HOST time of a launch-bound step (cfg-2 at B = 2) goes -- cProfile over 200 steps of AdapterPair fwd + bwd."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
c = dict(bench.CONFIGS["cfg2"], name="cfg2"); c["B"] = B
dev = torch.device("cuda:0")
wl = bench.Workload(c, torch.bfloat16, dev, 0, 1, "concurrent")
for _ in range(10): wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): wl.step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3 * (t1 - t0) / 100:.3f} ms/step, with drain {1e3 * (t2 - t0) / 100:.3f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): wl.step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
