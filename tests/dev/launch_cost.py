"""dev: host cost of one call through the C ABI: avmoe_add2 (argument checks + one hipLaunchKernelGGL) against avmoe_gemm on a tiny fp32 / bf16 product
(validation, tile choice, the streaming / per-frame kernels asked first, one launch).      python tests/dev/launch_cost.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avmoe_amd import _capi as capi
L = capi.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
n = 4000

def timed(fn):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n

a, b, c, d_ = (torch.zeros(256, device=dev) for _ in range(4))
if hasattr(L, "avmoe_add2"):
    L.avmoe_add2.restype = C.c_int
    f = lambda: L.avmoe_add2(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_long(256), C.c_void_p(c.data_ptr()), C.c_void_p(d_.data_ptr()), C.c_long(256), C.c_int(0), C.c_void_p(st))
    try:
        print("avmoe_add2         host %.2f us per call (wall %.2f)" % timed(f))
    except Exception as e:
        print("avmoe_add2: ", e)
for dtype, tdt in ((capi.F32, torch.float32), (capi.BF16, torch.bfloat16)):
    for planes in ((0, 1) if dtype == capi.F32 else (0,)):
        M, N, K = 64, 64, 64
        A = torch.randn(M, K, device=dev).to(tdt); B = torch.randn(N, K, device=dev).to(tdt); Cc = torch.zeros(M, N, device=dev)
        d = capi.GemmDesc()
        d.M, d.N, d.K, d.nb1, d.nb2 = M, N, K, 1, 1
        d.dtype, d.out_dtype = dtype, capi.F32
        d.a_layout, d.b_layout = 0, 0
        d.accumulate, d.ksplit, d.tile, d.alpha = 0, 1, 0, 1.0
        d.fp32_planes = planes
        d.lda, d.ldb = K, K
        d.sCi, d.sCj = N, 1
        f = lambda: L.avmoe_gemm(C.byref(d), A.data_ptr(), B.data_ptr(), Cc.data_ptr(), None, None, None, st)
        print("avmoe_gemm dtype %d planes %d  host %.2f us per call (wall %.2f)" % ((dtype, planes) + timed(f)))
x = torch.zeros(1024, device=dev)
print("torch x.add_(1)    host %.2f us per call (wall %.2f)" % timed(lambda: x.add_(1)))
