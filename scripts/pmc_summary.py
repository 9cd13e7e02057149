"""Aggregates rocprofv3 --pmc counter_collection.csv files per kernel: python scripts/pmc_summary.py <fetch_csv> <write_csv> [steps]
FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md 'HBM'); units of both counters are KB."""
import csv
import re
import sys
from collections import defaultdict


def load(path):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            m = re.search(r"(kf_\w+?)I[DfL]|(gemm_stream_kernel<[^>]*>)|(k[kw]_\w+?)I[DfL]|(gemm_kernelI\w+?)EEv|avmoe::(\w+)", name)
            if m:
                name = next(g for g in m.groups() if g)
            name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
            name = re.sub(r"^void ", "", name)
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return tot, cnt


def main():
    f, fc = load(sys.argv[1])
    w, wc = load(sys.argv[2])
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    names = sorted(set(f) | set(w), key=lambda n: -(2 * f.get(n, 0) + w.get(n, 0)))
    print(f"{'kernel':90s} {'calls':>6s} {'read MB/step':>13s} {'write MB/step':>14s}")
    tr = tw = 0.0
    for n in names[:60]:
        r = 2.0 * f.get(n, 0.0) / 1024.0 / steps
        wr = w.get(n, 0.0) / 1024.0 / steps
        tr += r; tw += wr
        print(f"{n[:90]:90s} {fc.get(n, wc.get(n, 0)):6d} {r:13.1f} {wr:14.1f}")
    print(f"{'TOTAL (listed)':90s} {'':6s} {tr:13.1f} {tw:14.1f}")


if __name__ == "__main__":
    main()
