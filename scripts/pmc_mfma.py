"""Matrix-pipe utilisation per kernel from a rocprofv3 --pmc pass: python scripts/pmc_mfma.py <counter_collection.csv>
SQ_VALU_MFMA_BUSY_CYCLES = cycles in which a SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs (checked: the dX streaming
GEMM issues 6.88 M v_mfma_f32_16x16x32_bf16 of 16 cycles each = 110.1 M, the counter reads 110.1 M); GRBM_GUI_ACTIVE = busy cycles
summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS section).  Matrix-pipe utilisation of a launch =
    SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).
The path is HBM-bound by construction: the number documents how far the matrix pipe is from being the limiter.
SQ_WAIT_ANY / SQ_WAVE_CYCLES = share of wave lifetime spent parked on s_waitcnt / barriers (memory latency)."""
import csv
import re
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        name = row["Kernel_Name"]
        m = re.search(r"(k[fg]_\w+?)I[DfL]|(gemm_stream_kernel<[^>]*>)|(k[kwg]_\w+?)I[DfL]|(gemm_kernelI\w+?)EEv|avmoe::(\w+)", name)
        if m:
            name = next(g for g in m.groups() if g)
        name = re.sub(r"\(.*", "", re.sub(r"^void ", "", name.replace("(anonymous namespace)::", "")))
        tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_BUSY_CYCLES":
            cnt[name] += 1
rows = []
for n, c in tot.items():
    busy, mfma, wave, wait = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_WAVE_CYCLES", 0.0), c.get("SQ_WAIT_ANY", 0.0)
    rows.append((busy, n, cnt[n], mfma / (128.0 * busy) if busy else 0.0, wait / wave if wave else 0.0))
rows.sort(reverse=True)
allb = sum(r[0] for r in rows) or 1.0
allm = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for c in tot.values())
print(f"{'kernel':78s} {'launches':>8s} {'share of GPU time':>18s} {'matrix-pipe util':>17s} {'WAIT_ANY/WAVE_CYCLES':>21s}")
for busy, n, k, mf, wt in rows[:40]:
    print(f"{n[:78]:78s} {k:8d} {busy / allb:18.3f} {mf:17.4f} {wt:21.3f}")
print(f"{'ALL KERNELS':78s} {'':8s} {1.0:18.3f} {allm / (128.0 * allb):17.4f}")
