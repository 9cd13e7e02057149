"""dev: the cfg-2 remap products (20800 x 768 x 768, K-major / MN-major B) on the 256 x 256, 128 x 128 and 64 x 64 tiles of the engine:
python tests/dev/remap_tile.py   (HIP-event timed; us and TFLOP/s per tile choice)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avmoe_amd import _capi as capi

L = capi.lib()
dev = torch.device("cuda:0")
bf = torch.bfloat16


def run(M, N, K, a_mn, b_mn, tile, ksplit=1, reps=20):
    A = torch.randn((K, M) if a_mn else (M, K), device=dev, dtype=bf)
    B = torch.randn((K, N) if b_mn else (N, K), device=dev, dtype=bf)
    Cm = torch.empty(M, N, device=dev, dtype=bf)
    d = capi.GemmDesc()
    d.M, d.N, d.K, d.nb1, d.nb2 = M, N, K, 1, 1
    d.dtype, d.out_dtype = capi.BF16, capi.BF16
    d.a_layout, d.b_layout = int(a_mn), int(b_mn)
    d.accumulate, d.ksplit, d.tile, d.alpha = 0, ksplit, tile, 1.0
    d.lda, d.ldb = (M if a_mn else K), (N if b_mn else K)
    d.sCi, d.sCj = N, 1
    nbytes = L.avmoe_gemm_workspace_bytes(C.byref(d))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        capi.check(L.avmoe_gemm(C.byref(d), A.data_ptr(), B.data_ptr(), Cm.data_ptr(), None, None, ws.data_ptr(), st), "gemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.avmoe_gemm(C.byref(d), A.data_ptr(), B.data_ptr(), Cm.data_ptr(), None, None, ws.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    return us, 2.0 * M * N * K / us / 1e6


for (M, N, K, a_mn, b_mn, ks) in ((20800, 768, 768, False, False, 1), (20800, 768, 768, False, True, 1), (768, 768, 20800, True, True, 14),
                                  (20800, 1024, 200, False, False, 1), (41600, 4096, 2304, False, False, 1)):
    row = []
    for tile in (0, 128, 64):
        us, tf = run(M, N, K, a_mn, b_mn, tile, ksplit=ks)
        row.append(f"tile {tile or 'auto'}: {us:7.1f} us {tf:6.1f} TF")
    print(f"M{M} N{N} K{K} {'M' if a_mn else 'K'}{'M' if b_mn else 'K'} ks{ks}:  " + "   ".join(row))
