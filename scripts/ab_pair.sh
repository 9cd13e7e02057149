#!/bin/bash
# development: how the two sites of a pair share the token gradients (two buffers + add / ordered accumulate variants); AVMOE_SIDE_PRIORITY=-1: high-priority side stream
for rep in 1 2 3; do for m in concurrent twobuf hybrid ordered; do v=$(python bench.py --pair $m --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-f32 --reps 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"); echo "rep $rep [$m] $v ms"; done; done
