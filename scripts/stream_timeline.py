"""dev: per-stream timeline of the LAST bench step in a rocprofv3 --kernel-trace CSV (two-stream mode: which stream is the critical path,
where a stream waits).  python scripts/stream_timeline.py <kernel_trace.csv> <kernels per step on the main stream's first kernel name>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "") + "/" + r.get("Stream_Id", "")) for r in rows), key=lambda x: x[0])
# a step begins with kk_prep_all of the forward; two per step (two sites): find the last two occurrences that start a step
idx = [i for i, k in enumerate(ks) if "kk_prep_all" in k[2]]
start = idx[-2] if len(idx) >= 2 else 0
prev = idx[-4] if len(idx) >= 4 else 0
step = ks[prev:start]                     # the last COMPLETE step
t0 = step[0][0]
print(f"step: {len(step)} kernels, wall {(max(k[1] for k in step) - t0) / 1e3:.1f} us, sum of durations {sum(k[1] - k[0] for k in step) / 1e3:.1f} us")
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"avmoe::|\(anonymous namespace\)::", "", n); n = re.sub(r"_ZN5avmoe\d*(_GLOBAL__N_1)?\d*", "", n)
    return n[:44]
streams = {}
for k in step: streams.setdefault(k[3], []).append(k)
for q, lst in streams.items():
    busy = sum(k[1] - k[0] for k in lst)
    print(f"\n== stream {q}: {len(lst)} kernels, busy {busy / 1e3:.1f} us, first at +{(lst[0][0] - t0) / 1e3:.1f}, last ends +{(lst[-1][1] - t0) / 1e3:.1f}")
    pe = lst[0][0]
    for k in lst:
        gap = (k[0] - pe) / 1e3
        if gap > 15 or (k[1] - k[0]) > 60e3:
            print(f"   +{(k[0] - t0) / 1e3:8.1f} us  gap {gap:7.1f}  dur {(k[1] - k[0]) / 1e3:7.1f}  {short(k[2])}")
        pe = max(pe, k[1])
