"""Micro-benchmark of the hop-1 chain's eight per-frame products at the cfg-2 shapes (dev tool): what `launch_gemm` runs for them --
frame_gemm.hip's short-K / long-K kernels (AVMOE_NO_FRAME_GEMM=1 in a development build: the tiled engine) -- every launch in a queue of 20.
usage: python scripts/frame_gemm_micro.py        (under rocprofv3 --pmc: scripts/pmc_kernel_sum.py summarises the counters per kernel)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from avmoe_amd import _capi as capi


def run(L, dev, tag, S=320, M=64, rows=65, N=1024, K=198, out_bf16=False, tr=False):
    bf = torch.bfloat16
    lda = -(-K // 8) * 8
    A = torch.randn(S * rows, lda, device=dev, dtype=bf)
    B = torch.randn(N, lda, device=dev, dtype=bf)
    ldc = -(-(M if tr else N) // 8) * 8
    Cm = torch.empty(S * (N if tr else rows), ldc, device=dev, dtype=bf if out_bf16 else torch.float32)
    d = capi.GemmDesc()
    d.nb1, d.nb2 = S, 1
    d.dtype, d.out_dtype = capi.BF16, (capi.BF16 if out_bf16 else capi.F32)
    d.alpha, d.ksplit = 1.0, 1
    d.M, d.K, d.N, d.lda, d.ldb, d.sA1 = M, K, N, lda, lda, rows * lda
    if tr:
        d.sCi, d.sCj, d.sC1 = 1, ldc, N * ldc
    else:
        d.sCi, d.sCj, d.sC1 = ldc, 1, rows * ldc
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
    fl = 2.0 * S * M * N * K
    print(f"{tag:58s} {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  out {S * M * N * Cm.element_size() / 1e6:5.0f} MB", flush=True)


def main():
    dev = torch.device("cuda:0")
    L = capi.lib()
    # audio-side site: X = 1024 audio tokens per frame, Y = 196 visual tokens
    run(L, dev, "a: L1 = [R|qr|qb] [Wc|bc|1]^T      64 x 1024, K 198, fp32", N=1024, K=198)
    run(L, dev, "a: dA1 = [dBm|dab] [Wc|bc]^T        65 x 1024, K 197, fp32 (tall)", M=65, N=1024, K=197)
    run(L, dev, "a: [Bm|ab] = A1 [Wc|bc]             64 x 197, K 1024, bf16", N=197, K=1024, out_bf16=True)
    run(L, dev, "a: dR^T = (dL1 Wc)^T                64 x 196, K 1024, bf16 [n][m]", N=196, K=1024, out_bf16=True, tr=True)
    # visual-side site: X = 196 visual tokens, Y = 1024 audio tokens
    run(L, dev, "v: L1                               64 x 196, K 1026, fp32", N=196, K=1026)
    run(L, dev, "v: dA1                              65 x 196, K 1025, fp32 (tall)", M=65, N=196, K=1025)
    run(L, dev, "v: [Bm|ab]                          64 x 1025, K 196, bf16", N=1025, K=196, out_bf16=True)
    run(L, dev, "v: dR^T                             64 x 1024, K 196, bf16 [n][m]", N=1024, K=196, out_bf16=True, tr=True)


if __name__ == "__main__":
    main()
