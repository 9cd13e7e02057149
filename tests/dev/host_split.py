"""dev: host time of a launch-bound step split into the C ABI calls and the Python around them (the autograd worker thread that
runs the backward is invisible to cProfile): python tests/dev/host_split.py [cfg1|cfg2b2]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from avmoe_amd import adapters as A, _capi as capi
name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
c = dict(bench.CONFIGS["cfg2" if name == "cfg2b2" else name], name=name)
if name == "cfg2b2": c["B"] = 2
dev = torch.device("cuda:0")
wl = bench.Workload(c, torch.bfloat16 if c["dtype"] == "bf16" else torch.float32, dev, 0, 1, "concurrent")
for _ in range(3): wl.step()
torch.cuda.synchronize()
acc = collections.defaultdict(float); cnt = collections.Counter()
def timed(key, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try: return fn(*a, **k)
        finally: acc[key] += time.perf_counter() - t0; cnt[key] += 1
    return w
L = capi.lib()
class LW:      # the library with its two hot entry points timed
    def __init__(s, L): s._L = L; s.avmoe_moe_forward = timed("C forward", L.avmoe_moe_forward); s.avmoe_moe_backward_part = timed("C backward", L.avmoe_moe_backward_part)
    def __getattr__(s, k): return getattr(s._L, k)
lw = LW(L)
capi.lib = lambda: lw
A._site_forward = timed("py _site_forward (incl. C)", A._site_forward)
A._SiteBackward.__init__ = timed("py _SiteBackward.__init__", A._SiteBackward.__init__)
A._SiteBackward.run = timed("py _SiteBackward.run (incl. C)", A._SiteBackward.run)
A._SiteBackward.finish = timed("py _SiteBackward.finish", A._SiteBackward.finish)
A._PairFunction.backward = staticmethod(timed("py pair backward (all)", A._PairFunction.backward))
# the torch calls of the facade (events, stream contexts, allocations, record_stream): where the Python share of a backward goes
if os.environ.get("HOST_SPLIT_TORCH", "1") != "0":
    torch.cuda.Event.record = timed("torch Event.record", torch.cuda.Event.record)
    torch.cuda.Stream.wait_event = timed("torch Stream.wait_event", torch.cuda.Stream.wait_event)
    torch.Tensor.record_stream = timed("torch Tensor.record_stream", torch.Tensor.record_stream)
    torch.cuda.StreamContext.__enter__ = timed("torch stream ctx enter", torch.cuda.StreamContext.__enter__)
    torch.cuda.StreamContext.__exit__ = timed("torch stream ctx exit", torch.cuda.StreamContext.__exit__)
    for fn in ("empty", "empty_like", "zeros", "zeros_like", "full"):
        setattr(torch, fn, timed("torch." + fn, getattr(torch, fn)))
    torch.cuda.current_stream = timed("torch.cuda.current_stream", torch.cuda.current_stream)
    A._PairFunction.forward = staticmethod(timed("py pair forward (all)", A._PairFunction.forward))
    wl.step = timed("bench step (all)", wl.step)
n = 10
t0 = time.perf_counter()
for _ in range(n): wl.step()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"{name}: host {1e3 * (t1 - t0) / n:.2f} ms per step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:34s} {1e3 * v / n:8.2f} ms/step  x{cnt[k] // n:4d}  {1e6 * v / cnt[k]:7.1f} us each")
