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


def _worker_sites(rank, world, port, out):
    """Shared buckets: three adapter sites, two of them in ONE bucket; the sites' backward is emulated through the gradient-sink
    protocol of avmoe_amd/adapters.py::_SiteBackward (write the slice at sink.offsets, done()); `average="optimizer"` leaves the
    1 / world to the optimizer's gradient scale."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from avmoe_amd.dp import AdapterGradReducer
        from tests.golden_util import load_golden
        from tests.test_adapters_api import build_module
        _, cfg, _ = load_golden("ave_train")
        torch.manual_seed(0)
        sites = [build_module("ave", cfg) for _ in range(3)]            # execution order
        head = torch.nn.Linear(8, 3)                                      # a plain (non-site) parameter set beside them
        site_bytes = sites[0].grad_layout()[2] * 4
        ok = True
        for average in ("auto", "optimizer"):
            red = AdapterGradReducer([p for m in sites for p in m.parameters()] + list(head.parameters()),
                                     bucket_mb=2.2 * site_bytes / (1 << 20), sites=sites, average=average)
            site_buckets = [b for b in red.buckets if b.sinks]
            ok &= [len(b.sinks) for b in site_buckets] == [2, 1]          # reverse execution order: (site 2, site 1), (site 0)
            ok &= site_buckets[0].sinks[0] is sites[2]._grad_sink and site_buckets[0].sinks[1] is sites[1]._grad_sink
            ok &= all(m._grad_sink.flat.data_ptr() == m._grad_sink.bucket.flat.data_ptr() + 4 * m._grad_sink.base for m in sites)

            def site_backward(i, step):
                sink = sites[i]._grad_sink
                g = torch.Generator().manual_seed(97 * step + 13 * i + rank)
                buf = torch.randn(sink.total, generator=g)
                if sink.fresh:
                    sink.flat.copy_(buf)
                else:
                    sink.flat.add_(buf)
                sink.done()
                return buf

            def run(step, sync):
                red.begin(sync=sync)
                for m in sites:
                    m._grad_sink.calls += 1                               # the forward of every site
                (head(torch.ones(2, 8) * (rank + 1)).sum() * step).backward()
                launched = []
                for i in (2, 1, 0):                                       # backward order
                    site_backward(i, step)
                    launched.append([b.work is not None for b in site_buckets])
                red.finish()
                return launched

            l1 = run(1, False)                                            # accumulation micro-step: nothing is sent
            ok &= all(not any(x) for x in l1)
            l2 = run(2, True)
            if world > 1:
                ok &= l2 == [[False, False], [True, False], [True, True]]    # a bucket goes out when its LAST site has reported
            for i, m in enumerate(sites):
                expect = torch.zeros(m._grad_sink.total)
                for r in range(world):
                    for step in (1, 2):
                        g = torch.Generator().manual_seed(97 * step + 13 * i + r)
                        expect += torch.randn(m._grad_sink.total, generator=g)
                expect *= (1.0 / world) if average == "auto" else 1.0    # "optimizer": the sum; FlatAdam scales by red.grad_scale
                ok &= torch.allclose(m._grad_sink.flat, expect, atol=1e-5)
                for (k, p), o in zip(m.named_parameters(), m._grad_sink.offsets):
                    ok &= p.grad.data_ptr() == m._grad_sink.flat.data_ptr() + 4 * o
            ok &= red.grad_scale == (1.0 if average == "auto" else 1.0 / world)
            w_expect = sum(3.0 * 2 * (r + 1) for r in range(world)) * (1.0 / world if average == "auto" else 1.0)
            ok &= torch.allclose(head.weight.grad, torch.full_like(head.weight, w_expect), atol=1e-5)
            red.zero_grad()
            ok &= all(float(b.flat.abs().max()) == 0.0 for b in red.buckets) and all(m._grad_sink.fresh for m in sites)
            for m in sites:
                del m._grad_sink
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_site_sinks_share_buckets_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_sites, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


@pytest.mark.timeout(180)
def test_adapter_grad_reducer_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def _local_reducer(n_sites=3, **kw):
    """A reducer over real adapter sites in ONE process: `world` is forced to 2 and `_launch` recorded instead of run, so the
    bookkeeping that decides WHEN a bucket goes out can be checked without a process group."""
    sys.path.insert(0, ROOT)
    from avmoe_amd.dp import AdapterGradReducer
    from tests.golden_util import load_golden
    from tests.test_adapters_api import build_module
    _, cfg, _ = load_golden("ave_train")
    torch.manual_seed(0)
    sites = [build_module("ave", cfg) for _ in range(n_sites)]
    site_bytes = sites[0].grad_layout()[2] * 4
    red = AdapterGradReducer([p for m in sites for p in m.parameters()], bucket_mb=2.2 * site_bytes / (1 << 20), sites=sites, **kw)
    launched = []

    class _Work:
        def wait(self):
            pass

    def fake_launch(b):
        launched.append(b)
        b.work = _Work()
    red.world, red._launch = 2, fake_launch
    return red, sites, launched


def _sink_backward(site, value):
    sink = site._grad_sink
    buf = torch.full((sink.total,), float(value))
    if sink.fresh:
        sink.flat.copy_(buf)
    else:
        sink.flat.add_(buf)
    sink.done()


def test_sink_reporting_twice_does_not_release_the_bucket_early():
    """A site whose forward + backward run twice in sequence inside one step reports twice; the bucket it shares with another
    site must still wait for THAT site (round-3 advisor finding: `pending` was decremented per report)."""
    red, sites, launched = _local_reducer()
    b01 = sites[2]._grad_sink.bucket
    assert sites[1]._grad_sink.bucket is b01 and sites[0]._grad_sink.bucket is not b01
    red.begin(sync=True)
    for _ in range(2):                                    # site 2: forward, backward, forward, backward
        sites[2]._grad_sink.calls += 1
        _sink_backward(sites[2], 1.0)
    assert launched == []                                 # site 1 has not written its slice yet
    sites[1]._grad_sink.calls += 1
    _sink_backward(sites[1], 3.0)
    assert launched == [b01]
    assert float(sites[2]._grad_sink.flat[0]) == 2.0      # the second backward was accumulated, not lost
    # a site with two forward calls pending holds its bucket until the second backward
    red.begin(sync=True); red.zero_grad(); launched.clear()
    sites[0]._grad_sink.calls += 2
    _sink_backward(sites[0], 1.0)
    assert launched == []
    _sink_backward(sites[0], 1.0)
    assert launched == [sites[0]._grad_sink.bucket]


def test_gradient_after_the_collective_went_out_fails_loudly():
    red, sites, launched = _local_reducer()
    red.begin(sync=True)
    sites[0]._grad_sink.calls += 1
    _sink_backward(sites[0], 1.0)                         # its bucket (one site) goes out
    assert launched == [sites[0]._grad_sink.bucket]
    sites[0]._grad_sink.calls += 1
    with pytest.raises(RuntimeError, match="after its bucket's all-reduce"):
        _sink_backward(sites[0], 1.0)


def test_lazy_zero_grad_skips_the_fill_and_finish_zeroes_idle_sites():
    red, sites, launched = _local_reducer()
    red.begin(sync=True)
    for m in sites:
        m._grad_sink.calls += 1
    for m in reversed(sites):
        _sink_backward(m, 5.0)
    red.finish()
    red.zero_grad(lazy=True)
    assert all(m._grad_sink.fresh and m._grad_sink.stale for m in sites)
    assert float(sites[0]._grad_sink.flat[0]) == 5.0      # no fill kernel ran
    red.begin(sync=True)                                  # next step: only site 2 and site 1 get a backward
    for m in sites[1:]:
        m._grad_sink.calls += 1
    _sink_backward(sites[2], 7.0); _sink_backward(sites[1], 7.0)
    red.finish()
    assert float(sites[2]._grad_sink.flat[0]) == 7.0 and float(sites[1]._grad_sink.flat.abs().max()) == 7.0
    assert float(sites[0]._grad_sink.flat.abs().max()) == 0.0      # idle site: zeros went out, not the previous step's gradient
    red.zero_grad()                                       # the eager form still zeroes everything
    assert all(float(b.flat.abs().max()) == 0.0 for b in red.buckets)


def test_second_reducer_detaches_the_first():
    sys.path.insert(0, ROOT)
    from avmoe_amd.dp import AdapterGradReducer
    lin = torch.nn.Linear(4, 4)
    r1 = AdapterGradReducer(lin.parameters())
    assert len(r1._hooks) == 2
    r2 = AdapterGradReducer(lin.parameters())
    assert r1._hooks == [] and len(r2._hooks) == 2
    r2.begin(sync=False)
    lin(torch.ones(1, 4)).sum().backward()                # only r2's hooks fire: r1's buckets see nothing
    assert all(b.pending == b.hooked for b in r1.buckets) and all(b.pending == 0 for b in r2.buckets)
    assert lin.weight.grad.data_ptr() == r2.buckets[0].flat.data_ptr() or lin.bias.grad.data_ptr() == r2.buckets[0].flat.data_ptr()
    red, sites, _ = _local_reducer(2)
    red.close()
    assert all(not hasattr(m, "_grad_sink") for m in sites)
