"""dev: sweep of streaming-kernel configurations (needs the SWEEP table compiled into gemm_stream.hip)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gemm_micro as gm
from avmoe_amd import _capi as capi


def time_one(L, entry, n):
    d, A, B, Cm, rs, D, nbytes = entry
    ws = torch.empty(16, device="cuda:0", dtype=torch.uint8)

    def call():
        capi.check(L.avmoe_gemm(C.byref(d), A.data_ptr(), B.data_ptr(), Cm.data_ptr(), rs.data_ptr() if rs is not None else None,
                                D.data_ptr() if D is not None else None, ws.data_ptr(), None), n)
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 125.0


def main():
    dev = torch.device("cuda:0")
    L = capi.lib()
    S = gm.shapes(dev)
    k5 = [(5, t, w, bm) for (t, w) in ((3, 8), (4, 6), (2, 12), (6, 4)) for bm in (32, 64, 128)]
    k12 = [(12, t, w, bm) for (t, w) in ((1, 8), (1, 9), (2, 4), (2, 5)) for bm in (16, 32, 64)] + [(12, 3, 3, 32), (12, 3, 3, 64)]
    for names, cfgs in ((["out"], k5), (["dApost", "down"], k12)):
        for cfg in cfgs:
            os.environ["AVMOE_STREAM_CFG"] = ",".join(map(str, cfg))
            row = []
            for pc in (1, 2, 3, 4):
                os.environ["AVMOE_STREAM_PERCU"] = str(pc)
                vals = []
                for n in names:
                    try:
                        vals.append(f"{time_one(L, S[n], n):6.0f}")
                    except Exception as e:
                        vals.append("   n/a")
                row.append("/".join(vals))
            print(f"{'+'.join(names):12s} cfg {cfg}:  percu1..4 = " + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
