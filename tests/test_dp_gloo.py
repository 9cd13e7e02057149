"""Multi-process data-parallel path on CPU (gloo, world size 2): the adapter-gradient reducer averages every
parameter gradient of a real adapter site across ranks, skips communication on accumulation micro-steps and
leaves `param.grad` as views of its flat buckets.  (The adapter arithmetic itself needs a GPU; gradients here
come from a synthetic per-rank loss over the site's real parameter set.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from avmoe_amd.dp import AdapterGradReducer
        from tests.golden_util import load_golden
        from tests.test_adapters_api import build_module
        _, cfg, _ = load_golden("ave_train")
        torch.manual_seed(0)
        m = build_module("ave", cfg)                       # identical parameters on every rank
        params = [p for p in m.parameters()]
        red = AdapterGradReducer(params, bucket_mb=0.05)   # several small buckets
        assert len(red.buckets) > 3

        def coeffs(step):
            g = torch.Generator().manual_seed(1000 * step + rank)
            return [torch.randn(p.shape, generator=g) for p in params]

        def backward(step):
            loss = sum((p * c).sum() for p, c in zip(params, coeffs(step)))
            loss.backward()

        # micro-step 1: accumulate only (no communication)
        red.begin(sync=False); backward(1); red.finish()
        local1 = [p.grad.clone() for p in params]
        # micro-step 2: accumulate + all-reduce
        red.begin(sync=True); backward(2); red.finish()
        ok = True
        for i, p in enumerate(params):
            expect = torch.zeros_like(p)
            for r in range(world):
                for step in (1, 2):
                    g = torch.Generator().manual_seed(1000 * step + r)
                    cs = [torch.randn(q.shape, generator=g) for q in params]
                    expect += cs[i]
            expect /= world
            ok &= torch.allclose(p.grad, expect, atol=1e-5)
            ok &= any(p.grad.data_ptr() >= b.flat.data_ptr() and
                      p.grad.data_ptr() < b.flat.data_ptr() + b.flat.numel() * 4 for b in red.buckets)
        ok &= all(torch.allclose(a, c[0]) for a, c in zip(local1, zip(coeffs(1))))
        red.zero_grad()
        ok &= all(float(p.grad.abs().max()) == 0.0 for p in params)
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_adapter_grad_reducer_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
