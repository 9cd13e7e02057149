#!/bin/bash
# Produces the round's judged artefacts on the GPU box (into gpurun_out/prof_final; copy them to profiles/<round>_* afterwards):
#   PMC read / write traffic per kernel and per step (first: bench.py's `traffic` field reads the round's pmc_traffic.json), matrix-pipe
#   utilisation counters, rocprofv3 kernel-trace stats of the default command (+ --pair serial, + every launch alone), then the bench line.
#   usage: scripts/make_profiles.sh [round tag, default r06]
R=$PWD; O=$R/gpurun_out/prof_final; T=${1:-r06}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-f32 --reps 1"
# counters: their own runs, kernel trace only (3 steps: 1 warm-up + 2 timed; the profiling pass of bench.py is off)
S="--steps 2 --warmup 1 --no-roofline --pair same"      # the timed region's kernel variants (dX overwrites, dY accumulates), one stream
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- $B $S > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- $B $S > $O/w.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/m -- $B $S > $O/m.log 2>&1
cd $R
python3 scripts/pmc_traffic_json.py $O/f/*/*counter_collection.csv $O/w/*/*counter_collection.csv 3 > $O/pmc_traffic.json
python3 scripts/pmc_summary.py $O/f/*/*counter_collection.csv $O/w/*/*counter_collection.csv 3 > $O/pmc_traffic.txt
python3 scripts/pmc_mfma.py $O/m/*/*counter_collection.csv > $O/pmc_mfma.txt 2>> $O/m.log
cp $O/pmc_traffic.json $R/profiles/${T}_pmc_traffic.json          # (on the box: what the bench lines below report as `traffic`)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kts -- $B --pair serial > $O/kts.log 2>&1
# every launch alone on the GPU, in the variants of the timed region (--pair same: the two-stream schedule on one stream; no helper streams
# inside a site): what bench.py's profiling pass times with HIP events
AVMOE_NO_SIDE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kta -- $B --pair same > $O/kta.log 2>&1
cd $R
cp $O/kta/*/*kernel_stats.csv $O/kernel_stats_alone.csv
cp $O/kt/*/*kernel_stats.csv $O/kernel_stats_default.csv
cp $O/kts/*/*kernel_stats.csv $O/kernel_stats_serial.csv
grep -h "^{\"metric\"" $O/kts.log | tail -1 > $O/bench_line_serial.json
python3 bench.py > $O/bench_line.json 2> $O/bench.err
python3 bench.py --batch 2 --no-cpu-baseline --no-f32 > $O/bench_line_b2.json 2>> $O/bench.err
tail -3 $O/pmc_traffic.txt; head -12 $O/pmc_mfma.txt
tail -1 $O/bench_line.json | cut -c1-300
