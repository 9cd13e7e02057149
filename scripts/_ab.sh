python -m pytest tests/test_cfg2_shape_gpu.py tests/test_moe_forward_gpu.py tests/test_moe_backward_gpu.py tests/test_moe_vs_oracle_midsize_gpu.py -x -q 2>&1 | tail -1
H=$PWD/avmoe_amd/lib/variants/libhead.so
for rep in 1 2; do
bash scripts/fam_one.sh new$rep "zzz" > /dev/null 2>&1
AVMOE_LIB=$H bash scripts/fam_one.sh old$rep "zzz" > /dev/null 2>&1
done
