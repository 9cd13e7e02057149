#!/bin/bash
# dev: ms per step of the other configurations under the in-tree library and a variant      scripts/cfg_ab.sh <variant lib> "cfg4 cfg5 cfg3"
for c in $2; do
  for v in base var base var; do
    if [ $v = base ]; then unset AVMOE_LIB; else export AVMOE_LIB=$1; fi
    python bench.py --config $c --steps 6 --warmup 2 --reps 2 --no-cpu-baseline --no-f32 --no-other-configs --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v ms_per_step', d['ms_per_step'])"
  done
done
