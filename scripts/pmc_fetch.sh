#!/bin/bash
# dev: HBM-side read / write bytes per kernel of one bench step (two PMC passes)   usage: scripts/pmc_fetch.sh <outdir-name>
R=$PWD; O=$R/gpurun_out/$1; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O.f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O.w.log 2>&1
python3 $R/scripts/pmc_summary.py $O/f/*/*counter_collection.csv $O/w/*/*counter_collection.csv 2 | grep -E "kernel|kf_|stream|xstats" | cut -c1-150
