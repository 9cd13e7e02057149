#!/bin/bash
# development: A/B of environment settings on the cfg-2 step at B = 2 clips (the reference's batch: launch-bound), interleaved repeats
#   scripts/ab_b2.sh "VAR=a" "VAR=b" ...      ("-" = no setting)
for rep in 1 2 3; do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then pre=""; else pre="$cfg"; fi
    v=$(env $pre python bench.py --batch 2 --steps 60 --warmup 10 --reps 3 --no-cpu-baseline --no-roofline --no-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['repeat_ms_per_step'])")
    echo "rep $rep  [$cfg]  $v ms"
  done
done
