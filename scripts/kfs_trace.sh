#!/bin/bash
# development: true kernel durations (rocprofv3 kernel trace, no event brackets) of the bottleneck-space kernels for one library
#   scripts/kfs_trace.sh tag [AVMOE_LIB path]
R=$PWD; T=$1; O=$R/gpurun_out/r6/tr_$T; mkdir -p $O
[ -n "$2" ] && export AVMOE_LIB=$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --pair same --steps 6 --warmup 2 --reps 1 --no-cpu-baseline --no-f32 --no-other-configs --no-roofline > $O/log.txt 2>&1
cd $R
python3 - $O/*/*kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("kfs_", "kf_", "gram64", "colsum")):
        print("%9.1f us avg  x%-4s %9.1f min %9.1f max  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, n[:110]))
PY
