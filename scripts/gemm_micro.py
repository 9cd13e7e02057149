"""Micro-benchmark of the engine GEMM shapes that stream over all tokens of the cfg-2 audio site (dev tool).
usage: python scripts/gemm_micro.py [name ...]      (HIP-event timed, prints us / GB/s per shape)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from avmoe_amd import _capi as capi

NT, Cc, G, KP, KPp, DZ, EDG = 327680, 768, 2, 140, 144, 256, 128
Cg = Cc // G


def desc(**kw):
    d = capi.GemmDesc()
    d.nb1 = d.nb2 = 1
    d.dtype, d.out_dtype = capi.BF16, capi.F32
    d.alpha, d.ksplit, d.sCj = 1.0, 1, 1
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def shapes(dev):
    bf = torch.bfloat16
    X = torch.randn(NT, Cc, device=dev, dtype=bf)
    Ap = torch.randn(NT, G, KPp, device=dev, dtype=bf)
    Bp = torch.randn(G, Cg, KPp, device=dev, dtype=bf)
    out = torch.empty(NT, Cc, device=dev, dtype=bf)
    dAp = torch.empty(NT, G, KPp, device=dev, dtype=torch.float32)
    Wt = torch.randn(G, EDG, Cg, device=dev, dtype=bf)
    Z = torch.empty(NT, DZ, device=dev, dtype=torch.float32)
    dZx = torch.randn(NT, DZ, device=dev, dtype=bf)
    rs = torch.randn(NT, device=dev)
    S = {}
    # out = Apost Bpost^T
    S["out"] = (desc(M=NT, N=Cg, K=KP, nb2=G, lda=G * KPp, sA2=KPp, ldb=KPp, sB2=Cg * KPp, sCi=Cc, sC2=Cg, out_dtype=capi.BF16),
                Ap, Bp, out, None, None, NT * (G * KPp * 2 + Cc * 2))
    # dApost = dOut Bpost
    S["dApost"] = (desc(M=NT, N=KP, K=Cg, nb2=G, lda=Cc, sA2=Cg, b_layout=1, ldb=KPp, sB2=Cg * KPp, sCi=G * KPp, sC2=KPp),
                   X, Bp, dAp, None, None, NT * (Cc * 2 + G * KPp * 4))
    # down: Z = X Wt^T  (Wt K-major [g][E*dgp][Cg])
    S["down"] = (desc(M=NT, N=EDG, K=Cg, nb2=G, lda=Cc, sA2=Cg, ldb=Cg, sB2=EDG * Cg, sCi=DZ, sC2=EDG),
                 X, Wt, Z, None, None, NT * (Cc * 2 + DZ * 4))
    # dX-like: per sample dX = dZx Wt + rs * X   (no second segment here)
    S["dX1"] = (desc(M=1024, N=Cg, K=EDG, nb1=320, nb2=G, lda=DZ, sA1=1024 * DZ, sA2=EDG, b_layout=1, ldb=Cg, sB2=EDG * Cg,
                     sCi=Cc, sC1=1024 * Cc, sC2=Cg, out_dtype=capi.BF16, sRS1=1024, sDi=Cc, sD1=1024 * Cc, sD2=Cg),
                dZx, Wt, out, rs, X, NT * (DZ * 2 + Cc * 2 + Cc * 2))
    # dBpost = dOut^T Apost  (split-K)
    dBp = torch.empty(G, Cg, KPp, device=dev, dtype=torch.float32)
    S["dBpost"] = (desc(M=Cg, N=KP, K=NT, nb2=G, a_layout=1, b_layout=1, lda=Cc, sA2=Cg, ldb=G * KPp, sB2=KPp, sCi=KPp,
                        sC2=Cg * KPp, ksplit=64),
                   X, Ap, dBp, None, None, NT * (Cc * 2 + G * KPp * 2))
    return S


def main():
    dev = torch.device("cuda:0")
    L = capi.lib()
    S = shapes(dev)
    names = sys.argv[1:] or list(S)
    for tile in (0, 64):
        for n in names:
            d, A, B, Cm, rs, D, nbytes = S[n]
            d.tile = tile
            ws_bytes = L.avmoe_gemm_workspace_bytes(C.byref(d))
            ws = torch.empty(max(ws_bytes, 16), device=dev, dtype=torch.uint8)

            def call():
                capi.check(L.avmoe_gemm(C.byref(d), A.data_ptr(), B.data_ptr(), Cm.data_ptr(), rs.data_ptr() if rs is not None else None,
                                        D.data_ptr() if D is not None else None, ws.data_ptr(), None), n)
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100.0
            print(f"{n:8s} tile {tile:3d}: {us:8.1f} us   {nbytes / us / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
