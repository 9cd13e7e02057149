#!/bin/bash
# development: per-family launch durations (HIP events, --pair serial) of the working-tree library and of variant libraries, side by side
#   scripts/fam_ab.sh outdir [libA.so libB.so ...]
O=$1; shift; mkdir -p $O
AVMOE_FAMILIES_OUT=$O/fam_work.json python bench.py --pair serial --steps 10 --warmup 3 --reps 1 --no-cpu-baseline --no-f32 > $O/bench_work.json 2>/dev/null
for L in "$@"; do
  n=$(basename $L .so)
  AVMOE_LIB=$L AVMOE_FAMILIES_OUT=$O/fam_$n.json python bench.py --pair serial --steps 10 --warmup 3 --reps 1 --no-cpu-baseline --no-f32 > $O/bench_$n.json 2>/dev/null
done
python - $O "$@" <<'PY'
import json, sys, os
O = sys.argv[1]; names = ["work"] + [os.path.basename(l)[:-3] for l in sys.argv[2:]]
fam = {n: {r["name"]: r for r in json.load(open(f"{O}/fam_{n}.json"))} for n in names}
keys = sorted(fam["work"], key=lambda k: -fam["work"][k]["total_ms"])
print(f"{'family':70s} " + " ".join(f"{n[:12]:>12s}" for n in names))
tot = {n: 0.0 for n in names}
for k in keys:
    row = []
    for n in names:
        r = fam[n].get(k)
        v = r["total_ms"] / r["calls"] * 1e3 if r else float("nan")
        row.append(v)
        if r: tot[n] += r["total_ms"]
    if fam["work"][k]["total_ms"] / sum(x["total_ms"] for x in fam["work"].values()) > 0.004:
        print(f"{k[:70]:70s} " + " ".join(f"{v:12.1f}" for v in row))
print("total ms (profiled steps): " + " ".join(f"{n}={tot[n]:.2f}" for n in names))
PY
