#!/bin/bash
# development: A/B of environment settings on the default bench command (interleaved repeats, to see past box noise)
#   scripts/ab_bench.sh "VAR=a" "VAR=b" ...      ("-" = no setting)
for rep in 1 2 3; do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then pre=""; else pre="$cfg"; fi
    v=$(env $pre python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "rep $rep  [$cfg]  $v ms"
  done
done
