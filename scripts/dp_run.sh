mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests -m gpu -x -q -k "cfg2 or fast" 2>&1 | tail -3
AVMOE_FAMILIES_OUT=gpurun_out/r4/fam_dp0.json python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32 --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"], d[\"repeat_ms_per_step\"])"
for v in $DPV; do
AVMOE_LIB=$PWD/avmoe_amd/lib/variants/libdp$v.so AVMOE_FAMILIES_OUT=gpurun_out/r4/fam_dp$v.json python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-other-configs > /dev/null 2>&1
done
