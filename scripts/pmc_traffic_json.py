"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs of one bench run -> per-launch HBM bytes per profiler family
(the names bench.py's roofline uses).   python scripts/pmc_traffic_json.py <fetch_csv> <write_csv> > profiles/r01_pmc_traffic.json
Both counters are in KB.  FETCH_SIZE is doubled for every kernel: gfx950 tallies a 128-byte read request at 64 B
(MI355X_MICROARCH.md, HBM section).  Calibrated per round on k_mid_bwd, whose only reads are Z and dz' at a known byte count:
round 3 (wave-per-expert kernels: the experts' 64-byte segments of a row are requested together, as whole 128-byte lines)
2 x 168.6 MB = 337 MB against 335.5 MB known; in rounds 1 - 2 the expert-outer kernels read one 64-byte segment per request and
their FETCH_SIZE was NOT doubled (HALF_LINE_READERS was the six k_* families)."""

HALF_LINE_READERS = ()
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
    m = re.search(r"kfs_(\w+?)ILi", sym) or re.search(r"kfs_(\w+?)<", sym)      # streaming form (tile_stream.hip)
    if m:
        return "k_" + m.group(1) + " (stream)"
    for sym_part, fam in (("kk_dx_stream3", "k_dx_stream3"), ("kk_hop1_yk", "k_hop1_yk"), ("kk_hop1_sum", "k_hop1_sum"),
                          ("kk_frame_gemm_long", "gemm_frames"), ("kk_frame_gemm", "gemm_frames"), ("kg_gram64", "k_gram64")):
        if sym_part in sym:
            return fam
    m = re.search(r"kk_hop1_yt(?:ILi\d+ELi\d+ELb([01])E|<\d+, \d+, (true|false)>)", sym)
    if m:
        return "k_hop1_yt_frames" if (m.group(1) == "1" or m.group(2) == "true") else "k_hop1_yt_sum"
    if "gemm_splitk_reduce" in sym:
        return "gemm_splitk_reduce"
    if "kk_xstats" in sym:
        return "k_xstats"
    if "kk_tp2_finish" in sym:
        return "k_tp2_finish"
    if "kk_tok_pair2" in sym:
        return "k_tok_pair2"
    if "kk_dx_stream2" in sym:
        return "k_dx_stream2"
    if "kk_dpair_reduce" in sym:
        return "k_dpair_reduce"
    if "kk_dpair" in sym:
        return "k_dpost_pair"
    m = re.search(r"gemm_stream_kernel<([^>]*)>", sym)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        return STREAM.get(tuple(a[:6]), "gemm_stream<" + ",".join(a) + ">")
    return None


STREAM = {("5", "0", "2", "12", "64", "false"): "gemm_stream_k160_n384", ("12", "0", "2", "4", "32", "false"): "gemm_stream_k384_n128",
          ("12", "0", "1", "9", "32", "false"): "gemm_stream_k384_n144", ("4", "3", "2", "12", "64", "false"): "gemm_stream_k128+96_n384",
          ("4", "3", "2", "12", "32", "false"): "gemm_stream_k128+96_n384r", ("2", "3", "4", "12", "64", "true"): "gemm_stream_k64+96mn_n768",
          ("2", "3", "4", "12", "32", "true"): "gemm_stream_k64+96mn_n768r",
          ("12", "0", "1", "9", "64", "false"): "gemm_stream_k384_n144", ("12", "0", "1", "12", "64", "false"): "gemm_stream_k384_n128+stats+x64",
          ("2", "3", "2", "12", "64", "true"): "gemm_stream_k64+96mn_n384", ("2", "3", "2", "12", "32", "true"): "gemm_stream_k64+96mn_n384r"}


def load(path, scale):
    """per family: mean and LARGEST per-launch bytes (the largest launch is the audio-side site of cfg-2)"""
    vals = defaultdict(list)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            f = family(row["Kernel_Name"])
            if f:
                vals[f].append(float(row["Counter_Value"]) * 1024.0 * (1.0 if (scale == 2.0 and f in HALF_LINE_READERS) else scale))
    return vals


def total_bytes(path, scale):
    """every kernel of the run, family or not (whole-line readers doubled as in load())"""
    tot = 0.0
    with open(path) as fh:
        for row in csv.DictReader(fh):
            f = family(row["Kernel_Name"])
            tot += float(row["Counter_Value"]) * 1024.0 * (1.0 if (scale == 2.0 and f in HALF_LINE_READERS) else scale)
    return tot


def main():
    """argv: <fetch_csv> <write_csv> [steps profiled]"""
    rd = load(sys.argv[1], 2.0)
    wr = load(sys.argv[2], 1.0)
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    out = {}
    import os
    stamp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "avmoe_amd", "lib", "libavmoe_hip.stamp")
    if os.path.isfile(stamp):          # which build of the library the counters belong to (bench.py reports `traffic` only for that build)
        with open(stamp) as fh:
            out["__lib_stamp__"] = fh.read().strip()
    if steps > 0:       # HBM-side bytes of ONE bench step over every kernel (bench.py: roofline.traffic)
        out["__total_bytes_per_step__"] = round((total_bytes(sys.argv[1], 2.0) + total_bytes(sys.argv[2], 1.0)) / steps)
        out["__read_bytes_per_step__"] = round(total_bytes(sys.argv[1], 2.0) / steps)
        out["__write_bytes_per_step__"] = round(total_bytes(sys.argv[2], 1.0) / steps)
    for f in sorted(set(rd) | set(wr)):
        if f.startswith("__"):
            continue
        r, w = rd.get(f, [0.0]), wr.get(f, [0.0])
        out[f] = {"launches_profiled": max(len(r), len(w)), "read_bytes_per_launch": round(sum(r) / len(r)),
                  "write_bytes_per_launch": round(sum(w) / len(w)), "read_bytes_largest_launch": round(max(r)),
                  "write_bytes_largest_launch": round(max(w))}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
