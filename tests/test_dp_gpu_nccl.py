"""World-size-2 data-parallel step over RCCL (`torch.distributed` backend "nccl" on ROCm), one process per GPU -- the collective
that replaces the reference's `nn.DataParallel` (AVVP/main.py:421, AVQA/net_grd_avst/main_avst_v2.py:321,
AVS/avs_scripts/avs_s4/train_v2.py:140).  Needs two visible GPUs: skipped on the one-GPU boxes the round's tests run on, and runs
unchanged on an 8-GPU node.  One AdapterPair step per rank with gradient sinks; after finish() every rank must hold the MEAN of the
two single-rank gradients, with the collective itself averaging (`average="auto"`: ncclAvg, probed at construction) and with the
1 / world left to the optimizer (`average="optimizer"`: plain sum + FlatAdam's grad_scale).  Also exercised: the side-stream event
ordering in front of the bucket's launch (the two sites of the pair finish on two streams) and `exposed_ms`."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_dp_gpu_gloo import _build, _free_port, _inputs

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (>= 2 GPUs)")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, average, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        sys.path.insert(0, ROOT)
        ca, cb, sa, sb = _build(dev)
        from avmoe_amd.adapters import AdapterPair
        from avmoe_amd.dp import AdapterGradReducer
        dtype = torch.bfloat16
        # what every rank must end up with: both ranks' gradients computed locally (no reducer), averaged
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
        params = list(sa.parameters()) + list(sb.parameters())
        red = AdapterGradReducer(params, sites=[sa, sb], average=average)
        ok = len(red.sinks) == 2
        ok &= red._avg_op == (average == "auto")                           # RCCL reduces with ncclAvg; the probe must have passed
        ok &= red.grad_scale == (1.0 if average == "auto" else 1.0 / world)
        pair = AdapterPair(sa, sb, concurrent=True)
        fa, fv, ga, gv = _inputs(ca, cb, rank, dev, dtype)
        red.time_exposed = True
        for step in range(2):                                              # second step: lazily "zeroed" buckets, re-recorded events
            red.begin(sync=True)
            oa, _, ov, _ = pair(fa, fv)
            torch.autograd.backward([oa, ov], [ga, gv])
            red.finish()
            torch.cuda.synchronize()
            gmax = max(float(e.abs().max()) for e in expect)
            for p, e in zip(params, expect):
                got = p.grad * red.grad_scale                              # "optimizer": the sum, scaled where FlatAdam would scale it
                ok &= float((got - e).abs().max()) <= 1e-5 * max(float(e.abs().max()), 1e-3 * gmax)
                ok &= any(b.flat.data_ptr() <= p.grad.data_ptr() < b.flat.data_ptr() + b.flat.numel() * 4 for b in red.buckets)
            red.zero_grad(lazy=True)
            for m, bb in zip((sa, sb), bufs):
                m.load_state_dict({**m.state_dict(), **bb})
        ex = red.exposed_ms()
        ok &= len(ex) == 2 and all(t >= 0.0 for t in ex)
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("average", ["auto", "optimizer"])
def test_data_parallel_step_world2_rccl(average):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), average, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


@pytest.mark.timeout(900)
def test_bench_two_ranks_over_rccl():
    """The driver's multi-GPU command on two real GPUs: rank 0's line carries the exchange (ranks, bytes, messages, exposed time)."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("AVMOE_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--reps", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["rccl"]["backend"] == "nccl" and d["grad_allreduce_bytes"] > 0
    assert d["exposed_allreduce_ms"] is not None and d["value"] > 0
