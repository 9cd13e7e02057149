#!/usr/bin/env python3
"""Run-to-run reproducibility of an AdapterPair step at the bench's parity shapes (B = 2 clips), with the caching allocator's
free memory poisoned (0xFF bytes = NaN in bf16 / fp32) before every run.  The path has no float atomics and a fixed reduction
order, so every output and gradient must repeat BIT FOR BIT; anything that moves is a race or a read of uninitialised memory.

    python tests/dev/race_hunt.py [--config cfg3] [--runs 30] [--dtype bf16] [--modes concurrent,serial] [--shapes 0,1,2,3]

Prints, per (shape, mode): the runs that differ from run 0, which tensors and how many elements moved (and whether NaN showed up).
(Round 4: written for the advisor's finding that cfg-3's bf16 forward gave 2.45e-2 on one box and 6.58e-3 on every other.)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from oracle import avmoe_oracle as O  # noqa: E402
from avmoe_amd.adapters import AdapterPair, release_workspaces  # noqa: E402


def poison(dev, gib=3.0):
    release_workspaces()
    torch.cuda.empty_cache()
    t = torch.empty(int(gib * (1 << 30)), dtype=torch.uint8, device=dev)
    t.fill_(0xFF)
    torch.cuda.synchronize()
    del t                                   # stays in the allocator's cache: the next workspaces are carved out of it


def run_once(c, w, dev, tdt, mode, S):
    ca, cv = w["ca"], w["cv"]
    ma, mv = bench.new_site(c, ca.Cx, ca.Nx, ca.Cy, ca.Ny), bench.new_site(c, cv.Cx, cv.Nx, cv.Cy, cv.Ny)
    ma.load_state_dict({**w["Pa"], **w["Ba"]}); mv.load_state_dict({**w["Pv"], **w["Bv"]})
    for m in (ma, mv):
        m.to(dev).train()
    fa, fv = w["fa"].to(dev, tdt).requires_grad_(True), w["fv"].to(dev, tdt).requires_grad_(True)
    xa, xv = fa.permute(0, 2, 1).unsqueeze(-1), fv.permute(0, 2, 1).unsqueeze(-1)
    pair = AdapterPair(ma, mv, concurrent=(mode == "concurrent"))
    lbs = []
    if c["variant"] == "avs":
        out_a, _ia, _p, lb_a, out_v, _iv, _q, lb_v = pair(xa, xv, is_training=False)
        lbs = [lb_a, lb_v]
    elif c["variant"] == "avvp":
        out_a, lb_a, out_v, lb_v = pair(xa, xv)
        lbs = [lb_a, lb_v]
    else:
        out_a, _ia, out_v, _iv = pair(xa, xv)
    ota, otv = out_a.squeeze(-1).permute(0, 2, 1), out_v.squeeze(-1).permute(0, 2, 1)
    loss = (ota.float() * w["ga"].to(dev)).sum() + (otv.float() * w["gv"].to(dev)).sum()
    for lb in lbs:
        if torch.is_tensor(lb):
            loss = loss + 0.01 * lb
    loss.backward()
    torch.cuda.synchronize()
    res = {"out_a": ota.detach().float().cpu(), "out_v": otv.detach().float().cpu(), "d_fa": fa.grad.float().cpu(), "d_fv": fv.grad.float().cpu()}
    for tag, m in (("a", ma), ("v", mv)):
        for k, p in m.named_parameters():
            res[f"{tag}.{k}"] = p.grad.float().cpu()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--runs", type=int, default=30)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--modes", default="concurrent,serial")
    ap.add_argument("--shapes", default=None)
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--no-poison", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    c = dict(bench.CONFIGS[a.config], name=a.config)
    tdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    S = a.clips * c["T"]
    g = torch.Generator().manual_seed(1234)
    work = []
    for i, (Ca, Na, Cv, Nv, cnt) in enumerate(c["pairs"]):
        ca, cv = bench._oracle_cfgs(c, (Ca, Na, Cv, Nv))
        Pa, Ba = O.init_params(ca, seed=2 * i)
        Pv, Bv = O.init_params(cv, seed=2 * i + 1)
        fa = 0.3 * torch.randn(S, Na, Ca, generator=g)
        fv = 0.3 * torch.randn(S, Nv, Cv, generator=g)
        ga, gv = torch.randn(fa.shape, generator=g), torch.randn(fv.shape, generator=g)
        if a.dtype == "bf16":
            ga, gv = ga.bfloat16().float(), gv.bfloat16().float()
        work.append(dict(ca=ca, cv=cv, Pa=Pa, Ba=Ba, Pv=Pv, Bv=Bv, fa=fa, fv=fv, ga=ga, gv=gv))
    sel = [int(x) for x in a.shapes.split(",")] if a.shapes else range(len(work))
    bad_total = 0
    for si in sel:
        w = work[si]
        tag = f"shape{si} C_a={w['ca'].Cx} N_a={w['ca'].Nx} C_v={w['cv'].Cx} N_v={w['cv'].Nx}"
        firsts = {}
        for mode in a.modes.split(","):
            ref, nbad = None, 0
            for it in range(a.runs):
                if not a.no_poison:
                    poison(dev)
                r = run_once(c, w, dev, tdt, mode, S)
                nan = [k for k, v in r.items() if not torch.isfinite(v).all()]
                if nan:
                    print(f"{tag} {mode} run {it}: NON-FINITE in {nan[:6]}", flush=True)
                if ref is None:
                    ref = r
                    continue
                moved = []
                for k, v in r.items():
                    if not torch.equal(v, ref[k]):
                        dlt = (v - ref[k]).abs()
                        moved.append((k, int((dlt > 0).sum()), float(dlt.max() / (ref[k].abs().max() + 1e-30))))
                if moved:
                    nbad += 1
                    print(f"{tag} {mode} run {it}: {len(moved)} tensors moved: " + "; ".join(f"{k} n={n} rel={e:.2e}" for k, n, e in moved[:8]), flush=True)
            firsts[mode] = ref
            bad_total += nbad
            print(f"{tag} {mode}: {nbad} of {a.runs - 1} runs differ from run 0", flush=True)
        if len(firsts) == 2:                 # the forward must not depend on the stream mode at all
            m0, m1 = list(firsts)
            for k in ("out_a", "out_v"):
                if not torch.equal(firsts[m0][k], firsts[m1][k]):
                    d = (firsts[m0][k] - firsts[m1][k]).abs()
                    print(f"{tag}: {k} differs between {m0} and {m1}: n={int((d > 0).sum())} rel={float(d.max() / firsts[m0][k].abs().max()):.2e}", flush=True)
    print("RACE_HUNT", "CLEAN" if bad_total == 0 else f"{bad_total} differing runs")


if __name__ == "__main__":
    main()
