"""dev: memory-operation skeleton of the innermost loops of the kernels in an AMDGPU .s file -- shows waits that sit behind a loop's stores
(the in-order memory counter makes such a wait pay for the store's acknowledgement):  python scripts/loop_waits.py file.s [name filter]"""
import re, sys, subprocess
txt = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
kern, depth_of = None, {}
events = {}
cur_hdr = None
for ln in txt:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        kern = m.group(1); cur_hdr = None; continue
    if kern is None or flt not in kern:
        continue
    m = re.search(r"Loop Header: Depth=(\d+)", ln)
    if m and ln.strip().startswith(";"):
        pass
    m = re.match(r"^\.(LBB\d+_\d+):", ln)
    if m:
        cur_lbl = m.group(1); cur_hdr = None
        continue
    m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", ln)
    if m:
        cur_hdr = (m.group(1), int(m.group(2))); continue
    m = re.search(r"=>\s*This Inner Loop Header: Depth=(\d+)", ln)
    if m:
        cur_hdr = ("L" + cur_lbl[1:], int(m.group(1))); continue
    if cur_hdr is None:
        continue
    t = ln.strip()
    ev = None
    if t.startswith("s_waitcnt") and "vmcnt" in t: ev = "W" + re.search(r"vmcnt\((\d+)\)", t).group(1)
    elif t.startswith("global_load") or t.startswith("buffer_load"): ev = "L"
    elif t.startswith("scratch_load"): ev = "sl"
    elif t.startswith("global_store") or t.startswith("buffer_store"): ev = "S"
    elif t.startswith("scratch_store"): ev = "ss"
    elif t.startswith("v_mfma"): ev = "m"
    elif t.startswith("s_barrier"): ev = "B"
    if ev:
        events.setdefault((kern, cur_hdr), []).append(ev)
for (k, h), ev in events.items():
    if "L" not in ev and "S" not in ev:
        continue
    out = []
    for e in ev:
        if out and out[-1][0] == e: out[-1][1] += 1
        else: out.append([e, 1])
    name = re.sub(r"^_ZN5avmoe(12_GLOBAL__N_1)?\d+", "", k)
    print(f"{name[:60]:60s} {h[0]} d{h[1]}: " + " ".join(e if n == 1 else f"{e}x{n}" for e, n in out))
