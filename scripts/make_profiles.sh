#!/bin/bash
# Produces the round's judged artefacts on the GPU box (into gpurun_out/prof_final; copy them to profiles/ afterwards):
#   bench line (default flags), rocprofv3 kernel-trace stats of the same command, PMC read / write traffic per kernel.
R=$PWD; O=$R/gpurun_out/prof_final; mkdir -p $O
python3 bench.py > $O/bench_line.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline > $O/kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kts -- python3 $R/bench.py --no-cpu-baseline --pair serial > $O/kts.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --pair serial > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --pair serial > $O/w.log 2>&1
cd $R
cp $O/kt/*/*kernel_stats.csv $O/kernel_stats_default.csv
cp $O/kts/*/*kernel_stats.csv $O/kernel_stats_serial.csv
grep -h "^{\"metric\"" $O/kts.log | tail -1 > $O/bench_line_serial.json
python3 scripts/pmc_traffic_json.py $O/f/*/*counter_collection.csv $O/w/*/*counter_collection.csv > $O/pmc_traffic.json
python3 scripts/pmc_summary.py $O/f/*/*counter_collection.csv $O/w/*/*counter_collection.csv 3 > $O/pmc_traffic.txt
tail -1 $O/kt.log | cut -c1-300
