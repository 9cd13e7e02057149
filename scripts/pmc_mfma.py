"""Matrix-pipe utilisation per kernel from a rocprofv3 --pmc pass: python scripts/pmc_mfma.py <counter_collection.csv>
SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is busy (MI355X_MICROARCH.md: = 32 x #MFMA for the 32x32x16
bf16 form), SQ_BUSY_CYCLES the cycles an SQ has work; the ratio (x the 4 SIMDs an SQ serves, where the counter is per SQ) is the
share of the launch in which matrix cores are issuing.  The path is HBM-bound by construction: the number
documents how far the matrix pipe is from being the limiter."""
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
        name = re.sub(r"\(.*", "", re.sub(r"^void ", "", name))
        tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_BUSY_CYCLES":
            cnt[name] += 1
rows = []
for n, c in tot.items():
    busy, mfma, wave, wait = c.get("SQ_BUSY_CYCLES", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_WAVE_CYCLES", 0.0), c.get("SQ_WAIT_ANY", 0.0)
    rows.append((busy, n, cnt[n], mfma / busy if busy else 0.0, wait / wave if wave else 0.0))
rows.sort(reverse=True)
allb = sum(r[0] for r in rows) or 1.0
allm = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for c in tot.values())
print(f"{'kernel':78s} {'launches':>8s} {'share of SQ_BUSY':>17s} {'MFMA_BUSY/SQ_BUSY':>18s} {'WAIT_ANY/WAVE_CYCLES':>21s}")
for busy, n, k, mf, wt in rows[:40]:
    print(f"{n[:78]:78s} {k:8d} {busy / allb:17.3f} {mf:18.4f} {wt:21.3f}")
print(f"{'ALL KERNELS':78s} {'':8s} {1.0:17.3f} {allm / allb:18.4f}")
