"""Micro-benchmark of the hop-1 chain's per-frame products on the tiled engine (dev tool): L1[s] = [R | qr | qb] [Wc | bc | 1]^T at the cfg-2
visual-side site (64 x 1024 per frame, K = 198) with the shape varied one parameter at a time -- where do its 45 us go?
usage: python scripts/gemm_hop1_micro.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from avmoe_amd import _capi as capi


def run(L, dev, tag, S=320, M=64, rows=65, N=1024, K=198, lda=200, out_bf16=False, tile=0, fold=False):
    bf = torch.bfloat16
    A = torch.randn(S * rows, lda, device=dev, dtype=bf)
    B = torch.randn(N, lda, device=dev, dtype=bf)
    Cm = torch.empty(S * rows, N, device=dev, dtype=bf if out_bf16 else torch.float32)
    d = capi.GemmDesc()
    d.nb1 = d.nb2 = 1
    d.dtype, d.out_dtype = capi.BF16, (capi.BF16 if out_bf16 else capi.F32)
    d.alpha, d.ksplit, d.sCj = 1.0, 1, 1
    d.K, d.N, d.lda, d.ldb, d.sCi, d.tile = K, N, lda, lda, N, tile
    if fold:
        d.M = S * rows
    else:
        d.M, d.nb1, d.sA1, d.sC1 = M, S, rows * lda, rows * N
    ws = torch.empty(max(L.avmoe_gemm_workspace_bytes(C.byref(d)), 16), device=dev, dtype=torch.uint8)

    def call():
        capi.check(L.avmoe_gemm(C.byref(d), A.data_ptr(), B.data_ptr(), Cm.data_ptr(), None, None, ws.data_ptr(), None), tag)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50.0
    fl = 2.0 * S * (rows if fold else M) * N * K
    print(f"{tag:44s} {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  out {Cm.numel() * Cm.element_size() / 1e6:5.0f} MB", flush=True)


def main():
    dev = torch.device("cuda:0")
    L = capi.lib()
    run(L, dev, "as in the step (b320, 64-tile, fp32 out)")
    run(L, dev, "bf16 output", out_bf16=True)
    run(L, dev, "32-tile", tile=32)
    run(L, dev, "128-tile", tile=128)
    run(L, dev, "K = 192", K=192)
    run(L, dev, "K = 128", K=128)
    run(L, dev, "K = 64", K=64)
    run(L, dev, "K = 398 (lda 400)", K=398, lda=400)
    run(L, dev, "N = 512", N=512)
    run(L, dev, "N = 256", N=256)
    run(L, dev, "160 frames", S=160)
    run(L, dev, "folded, tile chosen by the engine", fold=True)
    run(L, dev, "folded, 64-tile", fold=True, tile=64)
    run(L, dev, "folded, 128-tile", fold=True, tile=128)
    run(L, dev, "folded, 128-tile, bf16 out", fold=True, tile=128, out_bf16=True)


if __name__ == "__main__":
    main()
