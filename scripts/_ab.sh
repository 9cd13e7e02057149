mkdir -p gpurun_out/r5
python -m pytest tests/test_moe_vs_oracle_midsize_gpu.py tests/test_moe_backward_gpu.py tests/test_moe_forward_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -1
H=$PWD/avmoe_amd/lib/variants/libhead.so
for c in cfg5 cfg4 cfg1; do
for t in new old; do
  if [ $t = new ]; then L=""; else L="AVMOE_LIB=$H"; fi
  v=$(env $L python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('repeat_ms_per_step'))")
  echo "$c [$t] $v"
done; done
