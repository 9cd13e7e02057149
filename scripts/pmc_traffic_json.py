"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs of one bench run -> per-launch HBM bytes per profiler family
(the names bench.py's roofline uses).   python scripts/pmc_traffic_json.py <fetch_csv> <write_csv> > profiles/r01_pmc_traffic.json
FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 B: MI355X_MICROARCH.md, HBM section); both counters are in KB."""
import csv
import json
import re
import sys
from collections import defaultdict

LAY = {("0", "0"): "KK", ("0", "1"): "KM", ("1", "0"): "MK", ("1", "1"): "MM"}


def family(sym: str):
    m = re.search(r"gemm_kernelI(DF16b|f)Li(\d+)ELi\d+ELb([01])ELb([01])ELb([01])E", sym)
    if m:
        return f"gemm_{LAY[(m.group(3), m.group(4))]}{'+KM' if m.group(5) == '1' else ''}_{'bf16' if m.group(1) == 'DF16b' else 'f32'}_{m.group(2)}"
    m = re.search(r"gemm_kernel<(float|__bf16), (\d+), \d+, (true|false), (true|false), (true|false)>", sym)
    if m:
        b = {"true": "1", "false": "0"}
        return f"gemm_{LAY[(b[m.group(3)], b[m.group(4)])]}{'+KM' if m.group(5) == 'true' else ''}_{'bf16' if m.group(1) == '__bf16' else 'f32'}_{m.group(2)}"
    m = re.search(r"kf_(\w+?)I(?:DF16b|f)Li", sym) or re.search(r"kf_(\w+?)<", sym)
    if m:
        return "k_" + m.group(1)
    if "gemm_splitk_reduce" in sym:
        return "gemm_splitk_reduce"
    if "kk_xstats" in sym:
        return "k_xstats"
    m = re.search(r"gemm_stream_kernel<([^>]*)>", sym)
    if m:
        return "gemm_stream<" + m.group(1).replace(" ", "") + ">"
    return None


def load(path, scale):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            f = family(row["Kernel_Name"])
            if f:
                tot[f] += float(row["Counter_Value"]) * 1024.0 * scale
                cnt[f] += 1
    return tot, cnt


def main():
    rd, rc = load(sys.argv[1], 2.0)
    wr, wc = load(sys.argv[2], 1.0)
    out = {}
    for f in sorted(set(rd) | set(wr)):
        n = max(rc.get(f, 0), wc.get(f, 0), 1)
        out[f] = {"launches_profiled": n, "read_bytes_per_launch": round(rd.get(f, 0.0) / max(rc.get(f, 1), 1)),
                  "write_bytes_per_launch": round(wr.get(f, 0.0) / max(wc.get(f, 1), 1))}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
