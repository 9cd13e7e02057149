"""dev: GPU idle time and overlap from a rocprofv3 --kernel-trace CSV:  python scripts/trace_gaps.py <kernel_trace.csv> [steps]
Prints, over the last `steps` bench steps' worth of kernels: wall time, union of kernel intervals (GPU busy), sum of kernel durations,
the largest idle gaps and the kernels that precede / follow them."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows), key=lambda x: x[0])
n = len(ks)
ks = ks[n // 2:]                      # second half of the run: steady state
t0, t1 = ks[0][0], max(k[1] for k in ks)
busy, cur_s, cur_e, gaps = 0, ks[0][0], ks[0][1], []
last = ks[0]
for k in ks[1:]:
    if k[0] > cur_e:
        gaps.append((k[0] - cur_e, last[2][:60], k[2][:60]))
        busy += cur_e - cur_s; cur_s, cur_e = k[0], k[1]
    else:
        cur_e = max(cur_e, k[1])
    if k[1] >= cur_e: last = k
busy += cur_e - cur_s
tot = sum(k[1] - k[0] for k in ks)
print(f"kernels {len(ks)}  wall {(t1 - t0) / 1e6:.2f} ms  busy(union) {busy / 1e6:.2f} ms  idle {((t1 - t0) - busy) / 1e6:.2f} ms ({100 * (1 - busy / (t1 - t0)):.1f} %)  sum of durations {tot / 1e6:.2f} ms  overlap factor {tot / busy:.2f}")
gaps.sort(reverse=True)
print("largest idle gaps (us): after -> before")
agg = {}
for g, a, b in gaps:
    agg.setdefault((a, b), [0, 0]); agg[(a, b)][0] += g; agg[(a, b)][1] += 1
for (a, b), (g, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:25]:
    print(f"  {g / 1e3:9.1f} us total  x{c:4d}  {a}  ->  {b}")
