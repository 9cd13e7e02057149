#!/usr/bin/env python3
"""bench.py -- clip-pairs/sec of the AVMoE adapter hot path (fwd+bwd) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|f32] [--no-cpu-baseline] [--no-roofline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "cfg-2"): synthetic token tensors f_a:(S=B*T, N_a=1024, C=768),
f_v:(S, N_v=196, C=768) with B=32 clips x T=10 frames PER GPU, one adapter site = the audio-side MoEAdapter
(x=f_a, vis_token=f_v) + the visual-side MoEAdapter (x=f_v, vis_token=f_a)  [AVE net_trans_v3.py:695-698],
4 experts (2 cross-modal + 2 unimodal), bottleneck 64 (reduction 12), 2 conv groups, 32 latent tokens,
BatchNorm (training mode) + both LayerNorms on, gates = 0.5, bf16 activations with fp32 accumulation and
fp32 parameters.  One step = forward of both adapters + backward to both token tensors and every adapter /
router parameter (+ the RCCL all-reduce of those parameter gradients when N > 1).  Inputs are resident in
HBM before the timed region.  value = clips processed by all ranks / max-over-ranks time.

The JSON line also carries
  roofline      the dominant kernel family (largest share of GPU time in a HIP-event profiling pass over the
                same step): algorithmic bytes per launch / average launch duration vs the 8 TB/s HBM peak
  cpu_baseline  oracle/avmoe_oracle.py (eager PyTorch, fp32) timed on this box's host cores on a bounded
                sample of the same workload (same shapes, B=2 clips), rank 0 at N=1 only
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
from types import SimpleNamespace as NS

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG2 = dict(B=32, T=10, N_v=196, N_a=1024, C=768, E_m=2, E_s=2, reduction=12, groups=2, K=32)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_PEAK_TF = {"bf16": 2500.0, "f32": 157.3}


def make_opt(c):
    return NS(num_conv_group=c["groups"], is_before_layernorm=1, is_post_layernorm=1, is_self_attention=0,
              self_attention_version="v1", num_multimodal_experts=c["E_m"], num_singlemodal_experts=c["E_s"],
              use_load_balacing_loss=0)


def build_site(c, device):
    """The two MoEAdapters of one site, reference default init (seed 0), then gates <- 0.5."""
    from avmoe_amd.adapters import MoEAdapter
    torch.manual_seed(0)
    opt = make_opt(c)
    mk = lambda Nx, Ny: MoEAdapter(input_dim=c["C"], output_dim=c["C"], adapter_kind="bottleneck", dim_list=None,
                                   layer_idx=0, reduction_factor=c["reduction"], opt=opt, use_bn=True, use_gate=True,
                                   num_tk=c["K"], conv_dim_in=Ny, conv_dim_out=Nx, linear_in=c["C"], linear_out=c["C"])
    audio, visual = mk(c["N_a"], c["N_v"]), mk(c["N_v"], c["N_a"])
    for m in (audio, visual):
        with torch.no_grad():
            for k, p in m.named_parameters():
                if k.endswith(("gate", "gate_av")):
                    p.fill_(0.5)
        m.to(device).train()
    return audio, visual


def algorithmic_bytes_per_clip_pair(c, esz):
    """ideal fusion reads f_a,f_v (fwd) + writes 2 residuals + reads 2 upstream grads + re-reads
    f_a,f_v (bwd) + writes 2 input grads = 5 passes over both token tensors."""
    return 5.0 * c["T"] * (c["N_a"] + c["N_v"]) * c["C"] * esz


def reference_flops_forward(Cx, Nx, Cy, Ny, S, E_m, E_s, d, g, K):
    """Algorithmic FLOPs of one MoEAdapter forward in the reference's formulation (multiply-add = 2): token
    remap + fc, router, the four latent-attention products per cross-modal expert, grouped down + up, mixture."""
    E = E_m + E_s
    f = 2.0 * S * Nx * Ny * Cy + 2.0 * S * Nx * Cy * Cx
    f += 2.0 * S * (2 * Cx * 128 + 128 * 32 + 32 * E)
    f += E_m * 8.0 * S * K * Cx * Nx
    f += E * 4.0 * S * Nx * Cx * d / g
    f += 2.0 * S * E * Cx * Nx
    return f


def reference_flops_per_clip_pair(c):
    kw = dict(S=c["T"], E_m=c["E_m"], E_s=c["E_s"], d=c["C"] // c["reduction"], g=c["groups"], K=c["K"])
    return 3.0 * (reference_flops_forward(c["C"], c["N_a"], c["C"], c["N_v"], **kw) +
                  reference_flops_forward(c["C"], c["N_v"], c["C"], c["N_a"], **kw))


def cpu_baseline(c, budget_s=15.0):
    """Eager-PyTorch fp32 oracle on the host cores, same shapes at B=2 (S=20): 1 warm-up, then timed steps until ~budget_s of
    CPU work (3 .. 8 of them); the median is reported."""
    from oracle import avmoe_oracle as O
    # eager PyTorch on many tiny bmm/softmax ops gets SLOWER past a few dozen threads (measured on the 256-thread
    # GPU host: 132 s/step with 256 threads); use at most 32 and report the number actually used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    Bc = 2
    S = Bc * c["T"]
    mk = lambda Nx, Ny: O.AdapterConfig(Cx=c["C"], Nx=Nx, Cy=c["C"], Ny=Ny, E_m=c["E_m"], E_s=c["E_s"],
                                        reduction=c["reduction"], groups=c["groups"], K=c["K"])
    ca, cv = mk(c["N_a"], c["N_v"]), mk(c["N_v"], c["N_a"])
    Pa, Ba = O.init_params(ca, seed=0)
    Pv, Bv = O.init_params(cv, seed=1)
    g = torch.Generator().manual_seed(1234)
    fa = 0.3 * torch.randn(S, c["N_a"], c["C"], generator=g)
    fv = 0.3 * torch.randn(S, c["N_v"], c["C"], generator=g)
    ga, gv = torch.randn(fa.shape, generator=g), torch.randn(fv.shape, generator=g)
    times = []
    t_start = time.time()
    for it in range(9):
        t0 = time.time()
        O.moe_forward_backward(Pa, Ba, fa, fv, ca, ga, training=True)
        O.moe_forward_backward(Pv, Bv, fv, fa, cv, gv, training=True)
        dt = time.time() - t0
        if it > 0:
            times.append(dt)
        if time.time() - t_start > budget_s and len(times) >= 3:
            break
    med = statistics.median(times)
    return dict(value=Bc / med, unit="clip-pairs/s", cores=cores, kind="port",
                sample=f"oracle/avmoe_oracle.py eager PyTorch fp32, cfg-2 shapes at B={Bc} clips (S={S} frames), "
                       f"{len(times)} timed fwd+bwd steps after 1 warm-up, median {med:.2f} s/step, "
                       f"{torch.get_num_threads()} threads")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=CFG2["B"], help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--pair", default="concurrent", choices=["concurrent", "serial", "off"],
                    help="how the two sites of the layer are run: AdapterPair on two streams / on one stream / two separate calls")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the adapter path has no CPU fallback")
    if os.environ.get("AVMOE_BENCH_BACKEND", "nccl") != "nccl":      # development: several ranks share the GPUs that exist
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("AVMOE_BENCH_BACKEND", "nccl")      # "gloo": development only (several ranks on one GPU)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from avmoe_amd import _capi as capi
    from avmoe_amd.dp import AdapterGradReducer
    os.environ.setdefault("AVMOE_PROF_SHAPES", "1")     # profiler families per kernel and launch shape (read at first launch)
    capi.lib()
    c = dict(CFG2, B=args.batch)
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    esz = 2 if args.dtype == "bf16" else 4
    S = c["B"] * c["T"]
    audio, visual = build_site(c, device)
    params = list(audio.parameters()) + list(visual.parameters())
    reducer = AdapterGradReducer(params, bucket_mb=64.0, sites=[audio, visual])
    from avmoe_amd.adapters import AdapterPair
    pair = AdapterPair(audio, visual, concurrent=(args.pair == "concurrent"))

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    f_a = (0.3 * torch.randn(S, c["N_a"], c["C"], generator=g)).to(device, tdt).requires_grad_(True)
    f_v = (0.3 * torch.randn(S, c["N_v"], c["C"], generator=g)).to(device, tdt).requires_grad_(True)
    g_a = torch.randn(S, c["N_a"], c["C"], generator=g).to(device, tdt)
    g_v = torch.randn(S, c["N_v"], c["C"], generator=g).to(device, tdt)
    ga4, gv4 = g_a.permute(0, 2, 1).unsqueeze(-1), g_v.permute(0, 2, 1).unsqueeze(-1)

    def step(sync=True):
        reducer.begin(sync=sync)
        xa, xv = f_a.permute(0, 2, 1).unsqueeze(-1), f_v.permute(0, 2, 1).unsqueeze(-1)   # the reference's (S,C,N,1) views
        if args.pair == "off":
            out_a, _ = audio(xa, xv)                   # net_trans_v3.py:695
            out_v, _ = visual(xv, xa)                  # net_trans_v3.py:697
        else:
            out_a, _, out_v, _ = pair(xa, xv)          # the same two calls as one autograd node (AdapterPair)
        torch.autograd.backward([out_a, out_v], [ga4, gv4])
        reducer.finish()
        f_a.grad = None
        f_v.grad = None
        reducer.zero_grad()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = 1e3 * dt / args.steps
    value = c["B"] * world / (dt / args.steps)

    roofline = None
    if not args.no_roofline and rank == 0:
        # Profiling pass (HIP events around every launch, one family per kernel AND launch shape).  The two sites are run
        # back to back here (AdapterPair(concurrent=False)) so that every kernel is timed with the GPU to itself; the timed
        # region above overlaps them on two streams, which stretches each kernel's own duration.
        L = capi.lib()
        # Rank 0 only: the steps of this pass must not enter a collective (sync=False = an accumulation micro-step).
        pair_timed, pair = pair, AdapterPair(audio, visual, concurrent=False)
        for _ in range(2):
            step(sync=False)
        L.avmoe_prof_reset()
        L.avmoe_prof_enable(1)
        nprof = 3
        for _ in range(nprof):
            step(sync=False)
        torch.cuda.synchronize()
        L.avmoe_prof_enable(0)
        pair = pair_timed
        rep = capi.prof_report()
        L.avmoe_prof_reset()
        tot_ms = sum(r["total_ms"] for r in rep)
        dom = max(rep, key=lambda r: r["total_ms"])
        avg_ms = dom["total_ms"] / dom["calls"]
        gbs = dom["alg_bytes"] / dom["calls"] / (avg_ms * 1e-3) / 1e9
        tfs = dom["flops"] / dom["calls"] / (avg_ms * 1e-3) / 1e12
        roofline = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                        traffic=None, kernel=dom["name"], launches_per_step=dom["calls"] // nprof,
                        measured="sites serialised (kernel alone on the GPU); the timed region overlaps the two sites",
                        avg_launch_us=round(avg_ms * 1e3, 2), share_of_gpu_time=round(dom["total_ms"] / tot_ms, 3),
                        kernel_tflops=round(tfs, 1), kernel_mfma_frac=round(tfs / MFMA_PEAK_TF[args.dtype], 4),
                        path_algorithmic_gbs=round(algorithmic_bytes_per_clip_pair(c, esz) * value / world / 1e9, 1),
                        path_reference_tflops=round(reference_flops_per_clip_pair(c) * value / world / 1e12, 1),
                        families=sorted([dict(name=r["name"], calls=r["calls"] // nprof, ms_per_step=round(r["total_ms"] / nprof, 4),
                                              gbs=round(r["alg_bytes"] / max(r["total_ms"], 1e-9) / 1e6, 1))
                                         for r in rep], key=lambda r: -r["ms_per_step"])[:14])
        tj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
        if os.path.isfile(tj):       # HBM bytes per launch of the dominant family, from the committed rocprofv3 --pmc passes
            with open(tj) as fh:
                t = json.load(fh).get(dom["name"].split(" NT")[0].split(" M")[0])
            if t:      # the dominant launch is the audio-side (largest) one of its family
                roofline["traffic"] = t["read_bytes_largest_launch"] + t["write_bytes_largest_launch"]

    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline(c)

    if rank == 0:
        line = {
            "metric": "clip-pairs/sec (adapter fwd+bwd, AVE-shape synthetic)", "value": round(value, 2),
            "unit": "clip-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "cfg-2: one AVMoE adapter site (audio-side + visual-side MoEAdapter), fwd+bwd incl. "
                                   "input and parameter grads", "clips_per_gpu": c["B"], "frames_per_clip": c["T"],
                       "N_a": c["N_a"], "N_v": c["N_v"], "C": c["C"], "experts": "2 cross-modal + 2 unimodal",
                       "bottleneck": c["C"] // c["reduction"], "latent_tokens": c["K"], "groups": c["groups"],
                       "parallelism": f"dp{world}", "grad_allreduce_bytes": reducer.message_bytes() if world > 1 else 0},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                       # the other ranks wait for rank 0's profiling pass: clean teardown of the communicator
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
