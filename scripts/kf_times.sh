#!/bin/bash
# dev: per-kernel average durations of the register-resident kernels (cfg-2, --pair serial), from a rocprofv3 kernel trace
R=$PWD; O=$R/gpurun_out/kft; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --no-cpu-baseline --no-f32 --reps 1 --no-roofline --pair serial --steps 10 --warmup 3 > $O/log 2>&1
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/kft/t/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'kf_' in n or 'kg_' in n:
        print(f"{float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}  x{r['Calls']}  {n[:70]}")
PY
grep -h '"metric"' $O/log | tail -1 | cut -c1-200
