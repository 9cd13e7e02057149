#!/bin/bash
# One bench line per BASELINE configuration (cfg-2 is the default `python bench.py`): written to gpurun_out/<dir>/bench_<cfg>.json
#   scripts/bench_all.sh [outdir]        (cfg-3 is tried at its own B = 64 first, then at B = 16 if that does not fit)
O=${1:-gpurun_out/bench_all}; mkdir -p $O
for c in cfg1 cfg4 cfg5; do
  timeout 900 python bench.py --config $c --steps 10 --warmup 3 > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"; tail -c 600 $O/bench_$c.json | head -c 300; echo
done
timeout 1200 python bench.py --config cfg3 --steps 5 --warmup 2 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; rc=$?; echo "cfg3 B=64 rc=$rc"
if [ $rc -ne 0 ]; then
  tail -3 $O/bench_cfg3.err
  timeout 900 python bench.py --config cfg3 --batch 16 --steps 5 --warmup 2 > $O/bench_cfg3.json 2> $O/bench_cfg3_b16.err; echo "cfg3 B=16 rc=$?"
fi
python - $O <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_cfg*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["unit"], d["ms_per_step"], "ms", d["config"].get("clips_per_gpu"), d.get("parity", {}).get("verdict"), d.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
