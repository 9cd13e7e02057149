"""dev: the streaming-GEMM shapes of the cfg-2 audio-side site (bf16 outputs, as the step runs them), HIP-event timed --
run once per library variant (AVMOE_LIB=...):   python scripts/stream_dissect.py [tag]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gemm_micro as gm
from gemm_micro import NT, Cc, G, KP, KPp, DZ, EDG, Cg, desc
from avmoe_amd import _capi as capi


def main():
    dev = torch.device("cuda:0")
    L = capi.lib()
    bf = torch.bfloat16
    X = torch.randn(NT, Cc, device=dev, dtype=bf)
    Ap = torch.randn(NT, G, KPp, device=dev, dtype=bf)
    Bp = torch.randn(G, Cg, KPp, device=dev, dtype=bf)
    out = torch.empty(NT, Cc, device=dev, dtype=bf)
    dAp = torch.empty(NT, G, KPp, device=dev, dtype=bf)
    Wt = torch.randn(G, EDG, Cg, device=dev, dtype=bf)
    Z = torch.empty(NT, DZ, device=dev, dtype=bf)
    dZx = torch.randn(NT, DZ, device=dev, dtype=bf)
    rs = torch.randn(NT, device=dev)
    S = {
        "out": (desc(M=NT, N=Cg, K=KP, nb2=G, lda=G * KPp, sA2=KPp, ldb=KPp, sB2=Cg * KPp, sCi=Cc, sC2=Cg, out_dtype=capi.BF16),
                Ap, Bp, out, None, None, NT * (G * KPp * 2 + Cc * 2)),
        "dApost": (desc(M=NT, N=KP, K=Cg, nb2=G, lda=Cc, sA2=Cg, b_layout=1, ldb=KPp, sB2=Cg * KPp, sCi=G * KPp, sC2=KPp, out_dtype=capi.BF16),
                   X, Bp, dAp, None, None, NT * (Cc * 2 + G * KPp * 2)),
        "down": (desc(M=NT, N=EDG, K=Cg, nb2=G, lda=Cc, sA2=Cg, ldb=Cg, sB2=EDG * Cg, sCi=DZ, sC2=EDG, out_dtype=capi.BF16),
                 X, Wt, Z, None, None, NT * (Cc * 2 + DZ * 2)),
        "dX1": (desc(M=1024, N=Cg, K=EDG, nb1=320, nb2=G, lda=DZ, sA1=1024 * DZ, sA2=EDG, b_layout=1, ldb=Cg, sB2=EDG * Cg,
                     sCi=Cc, sC1=1024 * Cc, sC2=Cg, out_dtype=capi.BF16, sRS1=1024, sDi=Cc, sD1=1024 * Cc, sD2=Cg),
                dZx, Wt, out, rs, X, NT * (DZ * 2 + Cc * 2 + Cc * 2)),
    }
    tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("AVMOE_LIB", "work"))
    ws = torch.empty(16, device=dev, dtype=torch.uint8)
    cfgs = [c for c in os.environ.get("SWEEP_CFGS", "").split() if c] or [None]          # STREAM_SWEEP builds: "KS,KS2,TPW,NW,BM,PC ..."
    for cfg in cfgs:
        if cfg:
            os.environ["AVMOE_STREAM_CFG"] = cfg
        run(L, S, ws, f"{tag} {cfg or ''}")


def run(L, S, ws, tag):
    row = []
    for n, ent in S.items():
        d, A, B, Cm, r_, D, nbytes = ent[:7]
        def call():
            capi.check(L.avmoe_gemm(C.byref(d), A.data_ptr(), B.data_ptr(), Cm.data_ptr(), r_.data_ptr() if r_ is not None else None,
                                    D.data_ptr() if D is not None else None, ws.data_ptr(), None), n)
        try:
            call()
        except Exception:
            row.append(f"{n}     n/a")
            continue
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        row.append(f"{n} {us:7.1f} us {nbytes / us / 1e3:6.0f} GB/s")
    print(f"{tag:24s} " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
