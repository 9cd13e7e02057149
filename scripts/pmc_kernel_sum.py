"""Sums rocprofv3 --pmc counters per kernel over one or more counter_collection.csv files -- one file per pass of counters, SQ_WAVES in every
pass so that "per wave" divides by the waves of the SAME pass (dev tool):
   python scripts/pmc_kernel_sum.py <substring of the kernel name> file.csv [file.csv ...]"""
import csv
import sys
from collections import defaultdict

pat, files = sys.argv[1], sys.argv[2:]
tot = defaultdict(lambda: defaultdict(float))          # kernel -> (file, counter) -> sum
disp = defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat not in k:
            continue
        k = k.replace("void ", "").replace("avmoe::(anonymous namespace)::", "").replace("avmoe::", "").split("(")[0]
        tot[k][(f, r["Counter_Name"])] += float(r["Counter_Value"])
        disp[(k, f)].add(r["Dispatch_Id"])
for k in sorted(tot):
    print(f"{k}: {max(len(v) for (kk, _), v in disp.items() if kk == k)} launches per pass")
    for (f, c) in sorted(tot[k], key=lambda fc: fc[1]):
        if c == "SQ_WAVES" and f != files[0]:
            continue
        v, n, waves = tot[k][(f, c)], len(disp[(k, f)]), tot[k].get((f, "SQ_WAVES"), 0.0)
        print(f"  {c:28s} {v:16.0f}   per launch {v / n:14.0f}" + (f"   per wave {v / waves:10.1f}" if waves else ""))
