#!/bin/bash
# dev: matrix-pipe / VALU / LDS counters of the fp32 cfg-2 step per kernel (which unit bounds the three-plane engine products)
R=$PWD; O=$R/gpurun_out/r6/pmc_f32; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"; S="--dtype f32 --pair same --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-f32 --no-roofline --no-other-configs"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/m -- $B $S > $O/m.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/v -- $B $S > $O/v.log 2>&1
cd $R
python3 scripts/pmc_mfma.py $O/m/*/*counter_collection.csv > $O/pmc_mfma_f32.txt 2>> $O/m.log
head -16 $O/pmc_mfma_f32.txt
python3 - $O/v/*/*counter_collection.csv <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("avmoe::", "").replace("(anonymous namespace)::", "")[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
names = sorted({c for v in agg.values() for c in v})
print("kernel".ljust(72), " ".join(c[-18:].rjust(18) for c in names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_ACTIVE_INST_VALU", 0))[:14]:
    print(k.ljust(72), " ".join(f"{v.get(c, 0):18.3e}" for c in names))
PY
