#!/bin/bash
# dev: the fp32 fixture / parity tests and the fp32 cfg-2 step with moe_backward.cpp built under -DAVMOE_LEAF2=mask (moe_run.h; scripts/variant_lib.sh leaf<mask> moe_backward.cpp -DAVMOE_LEAF2=<mask>)
#   scripts/leaf2_try.sh "0 1 10 16 27"
mkdir -p gpurun_out/r6
for m in $1; do
  echo "== AVMOE_LEAF2=$m"
  if [ $m = 0 ]; then unset AVMOE_LIB; else export AVMOE_LIB=$PWD/avmoe_amd/lib/variants/libleaf$m.so; fi
  timeout 900 python -m pytest tests/test_adapters_gpu.py tests/test_moe_backward_gpu.py tests/test_cfg2_shape_gpu.py tests/test_blocks_gpu.py -q -x -k "not bf16" 2>&1 | tail -1
  python bench.py --dtype f32 --steps 10 --warmup 3 --reps 2 --no-cpu-baseline --no-f32 --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 ms_per_step', d['ms_per_step'], d['value'])"
done
