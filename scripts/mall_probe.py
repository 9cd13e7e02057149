"""dev: does a buffer WRITTEN by one kernel come back from the 256 MiB Infinity Cache when the next kernel reads it?
Times `V = W + 1` right after `W = U * 2` (stock elementwise kernels, bf16) for growing buffer sizes; a read served on-die
shows up as a bandwidth step below ~256 MiB of (written + streamed) bytes.   python scripts/mall_probe.py"""
import torch

dev = torch.device("cuda:0")
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)          # 1 GiB flush buffer


def run(mb, flush, reps=10):
    n = mb * (1 << 20) // 2
    U = torch.randn(n, device=dev).bfloat16()
    W = torch.empty_like(U)
    V = torch.empty_like(U)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tw, tr = [], []
    for _ in range(reps):
        big.fill_(1)                                                # evict everything
        if not flush:
            pass
        e0.record()
        torch.mul(U, 2, out=W)                                      # reads U (HBM), writes W
        e1.record()
        if flush:
            big.fill_(2)
            e1.record()
        torch.add(W, 1, out=V)                                      # reads W (cache?), writes V
        e2.record()
        torch.cuda.synchronize()
        tw.append(e0.elapsed_time(e1))
        tr.append(e1.elapsed_time(e2))
    tw.sort(); tr.sort()
    return tw[len(tw) // 2], tr[len(tr) // 2]


print(f"{'MiB':>6} {'producer ms':>12} {'consumer ms (W hot)':>20} {'GB/s':>8} {'consumer ms (flushed)':>22} {'GB/s':>8}")
for mb in (8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512):
    a, b = run(mb, False)
    _, c = run(mb, True)
    by = 2 * mb * (1 << 20)
    print(f"{mb:6d} {a:12.4f} {b:20.4f} {by / b / 1e6:8.0f} {c:22.4f} {by / c / 1e6:8.0f}")
