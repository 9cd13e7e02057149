# dev (-DAVMOE_DEV build): the tuned register-resident kernels (tile_fast.hip) against the generalised instance (tile_gen.inc) at the
# cfg-2 shape: parity tests under AVMOE_NO_FAST=1, then the per-family times of one bench step each, --pair serial, and the step times
O=gpurun_out/fast_vs_gen; mkdir -p $O
AVMOE_NO_FAST=1 python -m pytest tests/test_moe_vs_oracle_midsize_gpu.py tests/test_cfg2_shape_gpu.py tests/test_moe_backward_gpu.py -q -m gpu -k "fast or ship or cfg2 or golden or fixture or part" 2>&1 | tail -3
for v in fast gen; do
  if [ $v = gen ]; then export AVMOE_NO_FAST=1; else unset AVMOE_NO_FAST; fi
  AVMOE_FAMILIES_OUT=$O/fam_$v.json python bench.py --no-cpu-baseline --no-f32 --reps 1 --pair serial > $O/serial_$v.json 2>$O/err_$v
  python bench.py --no-cpu-baseline --no-f32 --no-roofline --reps 3 > $O/conc_$v.json 2>>$O/err_$v
  python - $O $v <<'PY'
import json, sys
O, v = sys.argv[1:]
rep = json.load(open(f"{O}/fam_{v}.json"))
tot = 0.0
for r in sorted(rep, key=lambda r: r["name"]):
    if r["name"].startswith(("k_pre_small", "k_post_small", "k_mid", "k_gram64")):
        tot += r["total_ms"] / 3
        print(f"  {v:4s} {r['name']:32s} {r['total_ms'] / r['calls'] * 1e3:8.1f} us")
s = json.loads(open(f"{O}/serial_{v}.json").read().strip().splitlines()[-1]); c = json.loads(open(f"{O}/conc_{v}.json").read().strip().splitlines()[-1])
print(f"{v}: bottleneck-space kernels {tot:.3f} ms/step ; step serial {s['ms_per_step']} ms, two-stream {c['repeat_ms_per_step']}")
PY
done
