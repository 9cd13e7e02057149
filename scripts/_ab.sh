python -m pytest tests/test_moe_backward_gpu.py tests/test_moe_vs_oracle_midsize_gpu.py -x -q 2>&1 | tail -1
H=$PWD/avmoe_amd/lib/variants/libhead.so
for rep in 1 2; do
for cfg in - AVMOE_LIB=$H; do
  if [ "$cfg" = "-" ]; then pre=""; else pre="$cfg"; fi
  v=$(env $pre python bench.py --config cfg1 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "cfg1 rep $rep [$cfg] $v"
done; done
