#!/usr/bin/env python3
"""dev: are the single-scalar gate gradients (`gate`, `gate_av`, `gate_self`: one number per expert) of the bf16 path systematically off, or is
a 10 - 30 % relative error on ONE draw of the upstream gradient the tail of a ratio of two random sums?

A scalar gate's gradient is  <G, y_e>  (G the upstream gradient, y_e the expert's ungated contribution): for a random G both the value and its
rounding error are zero-mean sums over the same ~1e6 terms, so value and error are (nearly independent) normals and the per-draw RELATIVE error
is a ratio of normals -- heavy-tailed: a handful of 10 - 50 x outliers among ~100 scalars is what chance alone produces.  The estimate that
does not have that tail is the ratio of RMS values over several draws of G with everything else fixed:

    eps(tensor) = sqrt(mean_k err_k^2) / sqrt(mean_k ref_k^2) ,   err_k = hip_k - oracle_k   (same inputs, the HIP path's ReLU mask)

printed for the HIP bf16 path and for the reference formulation itself under torch.autocast(bfloat16) (own mask), next to the largest
single-draw relative error of each.       python tests/dev/gate_grad_seeds.py --config cfg4 --shapes 1 --draws 12"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from oracle import avmoe_oracle as O  # noqa: E402
from avmoe_amd.adapters import AdapterPair  # noqa: E402
from avmoe_amd import debug as dbg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg4")
    ap.add_argument("--shapes", default=None)
    ap.add_argument("--draws", type=int, default=12)
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--all", action="store_true", help="every parameter tensor, not only the one-element ones")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    c = dict(bench.CONFIGS[a.config], name=a.config)
    lbw = 0.01 if c["variant"] in ("avvp", "avs") else 0.0
    S = a.clips * c["T"]
    g = torch.Generator().manual_seed(1234)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    sel = [int(x) for x in a.shapes.split(",")] if a.shapes else None
    for i, (Ca, Na, Cv, Nv, cnt) in enumerate(c["pairs"]):
        ca, cv = bench._oracle_cfgs(c, (Ca, Na, Cv, Nv))
        Pa, Ba = O.init_params(ca, seed=2 * i)
        Pv, Bv = O.init_params(cv, seed=2 * i + 1)
        fa = (0.3 * torch.randn(S, Na, Ca, generator=g)).bfloat16().float()
        fv = (0.3 * torch.randn(S, Nv, Cv, generator=g)).bfloat16().float()
        _ga, _gv = torch.randn(fa.shape, generator=g), torch.randn(fv.shape, generator=g)      # (keeps the generator in step with bf16_errors.py)
        if sel is not None and i not in sel:
            continue
        acc = {}                # tensor -> [sum err_hip^2, sum err_eager^2, sum ref^2, max rel hip, max rel eager, numel]
        gd = torch.Generator().manual_seed(99 + i)
        for k in range(a.draws):
            Ga, Gv = torch.randn(fa.shape, generator=gd).bfloat16().float(), torch.randn(fv.shape, generator=gd).bfloat16().float()
            ma, mv = bench.new_site(c, ca.Cx, ca.Nx, ca.Cy, ca.Ny), bench.new_site(c, cv.Cx, cv.Nx, cv.Cy, cv.Ny)
            ma.load_state_dict({**Pa, **Ba}); mv.load_state_dict({**Pv, **Bv})
            for m in (ma, mv):
                m.to(dev).train()
                dbg.keep_saved(m)
            xa_, xv_ = fa.to(dev, torch.bfloat16).requires_grad_(True), fv.to(dev, torch.bfloat16).requires_grad_(True)
            xa, xv = xa_.permute(0, 2, 1).unsqueeze(-1), xv_.permute(0, 2, 1).unsqueeze(-1)
            pair = AdapterPair(ma, mv, concurrent=True)
            lbs = []
            if c["variant"] == "avs":
                out_a, _ia, _p, lb_a, out_v, _iv, _q, lb_v = pair(xa, xv, is_training=False); lbs = [lb_a, lb_v]
            elif c["variant"] == "avvp":
                out_a, lb_a, out_v, lb_v = pair(xa, xv); lbs = [lb_a, lb_v]
            else:
                out_a, _ia, out_v, _iv = pair(xa, xv)
            ota, otv = out_a.squeeze(-1).permute(0, 2, 1), out_v.squeeze(-1).permute(0, 2, 1)
            loss = (ota.float() * Ga.to(dev)).sum() + (otv.float() * Gv.to(dev)).sum()
            for lb in lbs:
                if torch.is_tensor(lb) and lbw:
                    loss = loss + lbw * lb
            loss.backward()
            torch.cuda.synchronize()
            mka, mkv = dbg.relu_masks(ma), dbg.relu_masks(mv)
            ra = O.moe_forward_backward(Pa, Ba, fa, fv, ca, Ga, training=True, lb_weight=lbw, relu_masks=mka)[1]
            rv = O.moe_forward_backward(Pv, Bv, fv, fa, cv, Gv, training=True, lb_weight=lbw, relu_masks=mkv)[1]

            def run_eager(P, B, X, Y, cfg, G):
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    return O.moe_forward_backward({k_: v.to(dev) for k_, v in P.items()}, {k_: v.to(dev) for k_, v in B.items()}, X.to(dev), Y.to(dev), cfg,
                                                  G.to(dev), training=True, lb_weight=lbw)[1]
            ea, ev = run_eager(Pa, Ba, fa, fv, ca, Ga), run_eager(Pv, Bv, fv, fa, cv, Gv)
            for pre, m, r, e in (("a.", ma, ra, ea), ("v.", mv, rv, ev)):
                pg = dict(m.named_parameters())
                for name, ref in r.items():
                    if name in ("X", "Y") or (ref.numel() != 1 and not a.all):
                        continue
                    hip = pg[name].grad.float().cpu()
                    eh, ee, rr = float((hip - ref).norm()), float((e[name].float().cpu() - ref).norm()), float(ref.norm())
                    s = acc.setdefault(pre + name, [0.0, 0.0, 0.0, 0.0, 0.0, ref.numel()])
                    s[0] += eh * eh; s[1] += ee * ee; s[2] += rr * rr
                    s[3] = max(s[3], eh / max(rr, 1e-30)); s[4] = max(s[4], ee / max(rr, 1e-30))
        print(f"== {a.config} shape {i}: C_a={Ca} N_a={Na} C_v={Cv} N_v={Nv}  bf16, {a.draws} draws of the upstream gradient (inputs and parameters fixed)", flush=True)
        print(f"   {'tensor':44s} {'eps hip':>9s} {'eps eager':>9s} {'hip/eager':>9s}   {'worst draw hip':>14s} {'worst draw eager':>16s}", flush=True)
        rows = sorted(acc.items(), key=lambda kv: -(kv[1][0] / max(kv[1][2], 1e-60)))
        worst = 0.0
        for name, (sh, se, sr, mh, me, n) in rows:
            eh, ee = (sh / max(sr, 1e-60)) ** 0.5, (se / max(sr, 1e-60)) ** 0.5
            worst = max(worst, eh)
            print(f"   {name:44s} {eh:9.2e} {ee:9.2e} {eh / max(ee, 1e-30):9.2f}   {mh:14.2e} {me:16.2e}", flush=True)
        print(f"   -- largest eps (hip) {worst:.3e} over {len(rows)} tensors", flush=True)


if __name__ == "__main__":
    main()
