#!/bin/bash
# dev: A/B of the helper streams inside the forward / backward (AVMOE_NO_SIDE=1 off; AVMOE_SIDE_MASK bits 1 fwd, 2 bwd section 1, 4 bwd section 2)
B="python3 bench.py --no-cpu-baseline --no-f32 --no-roofline --reps 3"
P='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], j["repeat_ms_per_step"])'
for m in 0 1 2 4 3 5 6 7 0 7; do
  AVMOE_SIDE_MASK=$m $B 2>/dev/null | python3 -c "$P" "mask$m B32"
done
for m in 0 7; do
  AVMOE_SIDE_MIN=0 AVMOE_SIDE_MASK=$m $B --batch 8 2>/dev/null | python3 -c "$P" "mask$m B8"
done
