"""dev: the ordered kernel sequence of the LAST bench step from a rocprofv3 --kernel-trace CSV (name, duration, start-to-start gap):
python scripts/step_sequence.py <kernel_trace.csv> <steps in the run (warm-up + timed)>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2])
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
per = len(ks) // nsteps
last = ks[-per:]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("avmoe::", "")
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:100]
print(f"{len(ks)} kernels in the run, {per} per step; wall of the last step {(last[-1][1] - last[0][0]) / 1e3:.1f} us, sum of durations {sum(k[1] - k[0] for k in last) / 1e3:.1f} us")
prev = None
for i, (s, e, n) in enumerate(last):
    print(f"{i:4d} {(e - s) / 1e3:8.1f} us  gap {((s - prev) / 1e3 if prev else 0):7.1f}  {short(n)}")
    prev = s
