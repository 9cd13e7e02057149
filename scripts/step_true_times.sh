#!/bin/bash
# development: true (rocprofv3 kernel-trace) durations of the kernels of ONE cfg-2 step, every launch alone (--pair same, helper streams off),
# grouped by kernel name: launches, total us, share.   scripts/step_true_times.sh tag [AVMOE_LIB]
R=$PWD; T=$1; O=$R/gpurun_out/r6/st_$T; mkdir -p $O
[ -n "$2" ] && export AVMOE_LIB=$2
cd /tmp && export TMPDIR=/tmp
AVMOE_NO_SIDE=1 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --pair same --steps 4 --warmup 2 --reps 1 --no-cpu-baseline --no-f32 --no-roofline --no-other-configs > $O/log.txt 2>&1
cd $R
python3 - $O/*/*kernel_trace.csv <<'PY'
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda x: x[0])
per = len(ks) // 6
last = ks[-per:]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("avmoe::", "")
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n)
    n = re.sub(r"^_ZN5avmoe\d*", "", n)
    return n[:70]
tot = sum(e - s for s, e, _ in last)
print(f"{per} kernels per step; wall of the last step {(last[-1][1] - last[0][0]) / 1e3:.1f} us, sum of durations {tot / 1e3:.1f} us")
g = collections.OrderedDict()
for s, e, n in last:
    k = short(n); c = g.setdefault(k, [0, 0]); c[0] += 1; c[1] += e - s
acc = 0
for k, (c, d) in sorted(g.items(), key=lambda kv: -kv[1][1]):
    acc += d
    print(f"{d / 1e3:8.1f} us  x{c:<3d} {100 * d / tot:5.1f}%  cum {100 * acc / tot:5.1f}%  {k}")
PY
