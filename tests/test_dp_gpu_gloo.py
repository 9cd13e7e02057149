"""World-size-2 data-parallel step on ONE MI355X (two processes share cuda:0, gloo carries the exchange): the real HIP
adapter path with AdapterPair on two streams, the gradient sink (backward writes parameter gradients straight into the
reducer's flat buckets) and the bucket all-reduce launched from inside the backward.  After finish(), every rank must hold
the average of the per-rank gradients -- checked against single-process runs of both ranks' inputs.  (RCCL itself needs
one GPU per rank; the reducer logic, ordering and stream handling are identical.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(dev):
    sys.path.insert(0, ROOT)
    from oracle import avmoe_oracle as O
    from tests.test_adapters_api import build_module
    ca = O.AdapterConfig(Cx=128, Nx=96, Cy=64, Ny=48, reduction=2, groups=2, K=32)      # register-resident shape
    cb = O.AdapterConfig(Cx=64, Nx=48, Cy=128, Ny=96, reduction=4, groups=2, K=8)
    torch.manual_seed(0)                                                                # identical parameters on every rank
    sa, sb = build_module("ave", ca).to(dev).train(), build_module("ave", cb).to(dev).train()
    with torch.no_grad():
        for m in (sa, sb):
            for k, p in m.named_parameters():
                if k.endswith(("gate", "gate_av")):
                    p.fill_(0.4)
    return ca, cb, sa, sb


def _inputs(ca, cb, rank, dev, dtype):
    g = torch.Generator().manual_seed(100 + rank)
    fa = (0.5 * torch.randn(3, ca.Cx, ca.Nx, 1, generator=g)).to(dev, dtype)
    fv = (0.5 * torch.randn(3, cb.Cx, cb.Nx, 1, generator=g)).to(dev, dtype)
    ga = torch.randn(3, ca.Cx, ca.Nx, 1, generator=g).to(dev, dtype)
    gv = torch.randn(3, cb.Cx, cb.Nx, 1, generator=g).to(dev, dtype)
    return fa, fv, ga, gv


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        ca, cb, sa, sb = _build(dev)
        from avmoe_amd.adapters import AdapterPair
        from avmoe_amd.dp import AdapterGradReducer
        dtype = torch.bfloat16
        # reference: the gradients of BOTH ranks' inputs computed locally without any reducer, averaged
        expect = None
        bufs = [{k: b.clone() for k, b in m.named_buffers()} for m in (sa, sb)]
        for r in range(world):
            for m, bb in zip((sa, sb), bufs):
                m.zero_grad(); m.load_state_dict({**m.state_dict(), **bb})
            fa, fv, ga, gv = _inputs(ca, cb, r, dev, dtype)
            oa, _ = sa(fa, fv); ov, _ = sb(fv, fa)
            torch.autograd.backward([oa, ov], [ga, gv])
            gr = [p.grad.clone() for m in (sa, sb) for p in m.parameters()]
            expect = gr if expect is None else [a + b for a, b in zip(expect, gr)]
        expect = [e / world for e in expect]
        for m, bb in zip((sa, sb), bufs):
            m.zero_grad(set_to_none=True); m.load_state_dict({**m.state_dict(), **bb})
        # the data-parallel step of this rank
        params = list(sa.parameters()) + list(sb.parameters())
        red = AdapterGradReducer(params, sites=[sa, sb])
        pair = AdapterPair(sa, sb, concurrent=True)
        fa, fv, ga, gv = _inputs(ca, cb, rank, dev, dtype)
        red.begin(sync=True)
        oa, _, ov, _ = pair(fa, fv)
        torch.autograd.backward([oa, ov], [ga, gv])
        red.finish()
        torch.cuda.synchronize()
        ok = len(red.sinks) == 2
        gmax = max(float(e.abs().max()) for e in expect)
        for p, e in zip(params, expect):
            ok &= float((p.grad - e).abs().max()) <= 1e-5 * max(float(e.abs().max()), 1e-3 * gmax)
            ok &= any(b.flat.data_ptr() <= p.grad.data_ptr() < b.flat.data_ptr() + b.flat.numel() * 4 for b in red.buckets)
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_step_world2_on_one_gpu():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_bench_two_ranks_finish_and_print_one_line():
    """The driver's multi-GPU invocation of bench.py (torch.distributed.run, --gpus N): every rank must reach the end -- rank 0's
    roofline pass may not enter a collective the other ranks never join -- and rank 0 prints exactly one JSON line.  Two ranks on
    one GPU with gloo (AVMOE_BENCH_BACKEND, development switch); RCCL needs one GPU per rank."""
    import json
    import subprocess
    env = dict(os.environ, AVMOE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--reps", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["roofline"] is not None and d["cpu_baseline"] is None
    assert d["config"]["grad_allreduce_bytes"] > 0


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the shape of the driver's N = 1 command with a larger N): bench.py starts
    torch.distributed.run itself, as a child, before touching the GPU, and relays rank 0's single JSON line with n_gpus and the
    all-reduce message size filled in (gloo stands in for RCCL: two ranks share the one GPU of the test box)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(AVMOE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--reps", "1", "--no-f32"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["grad_allreduce_bytes"] > 0
    assert d["roofline"]["achieved"] > 0 and d["roofline"]["mfma_frac"] > 0 and isinstance(d["roofline"]["dominant_kernel"], str)
    assert d["roofline"]["dominant_us"] > 0 and 0 <= d["roofline"]["dominant_frac"] < 1      # (B = 2: the dominant launch may be a glue kernel without algorithmic bytes)
    # the exchange on the line (scalars + the `rccl` object): ranks, bytes, messages, the exposed part of the all-reduce
    assert d["rccl_ranks"] == 2 and d["grad_allreduce_bytes"] == d["config"]["grad_allreduce_bytes"] and d["allreduce_buckets"] >= 1
    assert d["exposed_allreduce_ms"] is not None and d["exposed_allreduce_ms"] >= 0.0 and d["rccl"]["backend"] == "gloo"
