#!/bin/bash
# development: per-family launch durations (HIP events, every launch alone on the GPU) of ONE library for the cfg-2 step
#   scripts/fam_one.sh tag [grep pattern] [extra bench args]      AVMOE_LIB=... selects the library
O=${FAM_OUT:-gpurun_out/r6}; mkdir -p $O; T=$1; PAT=${2:-.}; shift; shift
AVMOE_PROF_SHAPES=1 AVMOE_FAMILIES_OUT=$O/fam_$T.json python bench.py --pair same --steps 10 --warmup 3 --reps 1 --no-cpu-baseline --no-f32 --no-other-configs "$@" > $O/bench_$T.json 2>$O/bench_$T.err
python - $O/fam_$T.json "$PAT" <<'PY'
import json, sys, re
r = json.load(open(sys.argv[1])); pat = re.compile(sys.argv[2])
steps = 3
print("GPU ms/step %.3f, launches %d" % (sum(x["total_ms"] for x in r) / steps, sum(x["calls"] for x in r) // steps))
for x in sorted(r, key=lambda x: -x["total_ms"]):
    if pat.search(x["name"]):
        print("%8.1f us x%-3d %7.3f ms/step  %s" % (x["total_ms"] / x["calls"] * 1e3, x["calls"] // steps, x["total_ms"] / steps, x["name"]))
PY
python -c "import json,sys; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
