#!/usr/bin/env python3
"""dev: per-tensor bf16 gradient errors of an AdapterPair step at a configuration's site shapes (B = 2 clips), against the oracle
evaluated on the bf16-rounded inputs WITH THE HIP PATH'S ReLU mask (the `*_same_mask` numbers of bench.py's parity leg, but every
tensor, not the worst one).    python tests/dev/bf16_errors.py --config cfg4 [--shapes 0,2] [--top 12] [--f32]

Columns: tensor, norm-wise error (floor 1e-3 of the largest gradient norm), its norm relative to the largest, max-abs error / max."""
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
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--shapes", default=None)
    ap.add_argument("--top", type=int, default=12)
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--pair", default="concurrent")
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--eager", action="store_true", help="adds the error of the reference formulation itself under torch.autocast(bfloat16) (the oracle run eagerly on the GPU) and the ratio")
    ap.add_argument("--exact-weights", action="store_true", help="experiment: every parameter rounded to a bf16-representable value first (what is left is NOT the rounding of the raw weights)")
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
        if a.exact_weights:
            Pa = {k: (v.bfloat16().float() if v.is_floating_point() else v) for k, v in Pa.items()}
            Pv = {k: (v.bfloat16().float() if v.is_floating_point() else v) for k, v in Pv.items()}
        fa = 0.3 * torch.randn(S, Na, Ca, generator=g)
        fv = 0.3 * torch.randn(S, Nv, Cv, generator=g)
        Ga, Gv = torch.randn(fa.shape, generator=g), torch.randn(fv.shape, generator=g)
        if sel is not None and i not in sel:
            continue
        bf16 = not a.f32
        if bf16:
            fa, fv, Ga, Gv = (t.bfloat16().float() for t in (fa, fv, Ga, Gv))
        ma, mv = bench.new_site(c, ca.Cx, ca.Nx, ca.Cy, ca.Ny), bench.new_site(c, cv.Cx, cv.Nx, cv.Cy, cv.Ny)
        ma.load_state_dict({**Pa, **Ba}); mv.load_state_dict({**Pv, **Bv})
        for m in (ma, mv):
            m.to(dev).train()
            dbg.keep_saved(m)
        tdt = torch.bfloat16 if bf16 else torch.float32
        xa_, xv_ = fa.to(dev, tdt).requires_grad_(True), fv.to(dev, tdt).requires_grad_(True)
        xa, xv = xa_.permute(0, 2, 1).unsqueeze(-1), xv_.permute(0, 2, 1).unsqueeze(-1)
        pair = AdapterPair(ma, mv, concurrent=(a.pair != "serial"))
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
        ra = O.moe_forward_backward(Pa, Ba, fa, fv, ca, Ga, training=True, lb_weight=lbw, relu_masks=mka)
        rv = O.moe_forward_backward(Pv, Bv, fv, fa, cv, Gv, training=True, lb_weight=lbw, relu_masks=mkv)
        eo = max(float((ota.detach().float().cpu() - ra[0]["out"]).abs().max() / ra[0]["out"].abs().max()),
                 float((otv.detach().float().cpu() - rv[0]["out"]).abs().max() / rv[0]["out"].abs().max()))
        items = [("a." + k, dict(ma.named_parameters())[k].grad.float().cpu(), v) for k, v in ra[1].items() if k not in ("X", "Y")]
        items += [("v." + k, dict(mv.named_parameters())[k].grad.float().cpu(), v) for k, v in rv[1].items() if k not in ("X", "Y")]
        items += [("tok.f_a", xa_.grad.float().cpu(), ra[1]["X"] + rv[1]["Y"]), ("tok.f_v", xv_.grad.float().cpu(), rv[1]["X"] + ra[1]["Y"])]
        nmax = max(float(v.norm()) for _k, _g, v in items)
        eager = {}
        if a.eager and bf16:          # the reference formulation itself in bf16: the oracle eagerly on the GPU under autocast, same inputs, own mask
            def run_eager(P, B, X, Y, cfg, G):
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    return O.moe_forward_backward({k: v.to(dev) for k, v in P.items()}, {k: v.to(dev) for k, v in B.items()}, X.to(dev), Y.to(dev), cfg, G.to(dev),
                                                  training=True, lb_weight=lbw)[1]
            ea, ev = run_eager(Pa, Ba, fa, fv, ca, Ga), run_eager(Pv, Bv, fv, fa, cv, Gv)
            eager = {"a." + k: v.float().cpu() for k, v in ea.items() if k not in ("X", "Y")}
            eager.update({"v." + k: v.float().cpu() for k, v in ev.items() if k not in ("X", "Y")})
            eager["tok.f_a"] = (ea["X"] + ev["Y"]).float().cpu(); eager["tok.f_v"] = (ev["X"] + ea["Y"]).float().cpu()
        rows = []
        for k, gg, v in items:
            den = max(float(v.norm()), 1e-3 * nmax)
            e = float((gg - v).norm()) / den
            ee = float((eager[k] - v).norm()) / den if k in eager else float("nan")
            rows.append((e, k, float(v.norm()) / nmax, float((gg - v).abs().max() / v.abs().max().clamp_min(1e-30)), v.numel(), ee))
        rows.sort(reverse=True)
        print(f"== {a.config} shape {i}: C_a={Ca} N_a={Na} C_v={Cv} N_v={Nv}  {'bf16' if bf16 else 'f32'}{'  exact-weights' if a.exact_weights else ''}  out_rel {eo:.3e}", flush=True)
        for e, k, rn, em, n, ee in rows[:a.top]:
            extra = f"   eager {ee:.3e}  hip/eager {e / ee:5.2f}" if ee == ee and ee > 0 else ""
            print(f"   {k:52s} err {e:.3e}   norm/max {rn:.2e}   maxabs {em:.2e}   n={n}{extra}", flush=True)
        if eager:
            above = [(k, e, ee) for e, k, _rn, _em, _n, ee in rows if e > 0.05 and e > ee]
            print(f"   -- tensors above 5 % AND above the eager-autocast error: {len(above)} of {len(rows)}: " + ", ".join(f"{k} {e:.3f} ({ee:.3f})" for k, e, ee in above), flush=True)


if __name__ == "__main__":
    main()
