#!/usr/bin/env python3
"""dev: which named workspace buffer of a site moves first when an AdapterPair forward (optionally + backward) is repeated on two
streams?  Workspaces are pre-filled with 0xFF (torch.empty of uint8 is patched), every named buffer of both sites (`saved` and the
per-stream `scratch`) is snapshotted after each run and compared with run 0, in plan order (= roughly pipeline order).

    python tests/dev/race_buffers.py --config cfg3 --shape 0 --runs 8 [--mode concurrent] [--backward] [--bg matmul]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from oracle import avmoe_oracle as O  # noqa: E402
import avmoe_amd.adapters as A  # noqa: E402
from avmoe_amd import _capi as capi, _capi_moe as cm, debug as dbg  # noqa: E402

_empty = torch.empty


def _poisoned_empty(*a, **k):
    t = _empty(*a, **k)
    if t.dtype == torch.uint8 and t.is_cuda:
        t.fill_(0xFF)
    return t


def make_gemm_bg(spec, dev):
    """'M,N,K,nb,a_mn,b_mn[,f32]' -> a callable that enqueues that avmoe_gemm once on the current stream"""
    import ctypes as C
    f = spec.split(",")
    M, N, K, nb, a_mn, b_mn = (int(x) for x in f[:6])
    f32 = len(f) > 6 and f[6] == "f32"
    L = capi.lib()
    tdt = torch.float32 if f32 else torch.bfloat16
    A = torch.randn((nb, K, M) if a_mn else (nb, M, K), device=dev).to(tdt)
    B = torch.randn((nb, K, N) if b_mn else (nb, N, K), device=dev).to(tdt)
    Cm = torch.zeros(nb, M, N, device=dev, dtype=tdt)
    d = capi.GemmDesc()
    d.M, d.N, d.K, d.nb1, d.nb2 = M, N, K, nb, 1
    d.dtype = d.out_dtype = capi.F32 if f32 else capi.BF16
    d.a_layout, d.b_layout, d.accumulate, d.ksplit, d.tile, d.alpha = a_mn, b_mn, 0, 1, 0, 1.0
    d.lda, d.ldb = (M if a_mn else K), (N if b_mn else K)
    d.sA1, d.sA2, d.sB1, d.sB2 = M * K, M * K, N * K, N * K
    d.sCi, d.sCj, d.sC1, d.sC2 = N, 1, M * N, M * N
    nbytes = L.avmoe_gemm_workspace_bytes(C.byref(d))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)

    def run():
        capi.check(L.avmoe_gemm(C.byref(d), A.data_ptr(), B.data_ptr(), Cm.data_ptr(), None, None, ws.data_ptr(), torch.cuda.current_stream().cuda_stream), "avmoe_gemm")
    run.keep = (A, B, Cm, ws, d)
    return run


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--shape", type=int, default=0)
    ap.add_argument("--runs", type=int, default=8)
    ap.add_argument("--mode", default="concurrent")
    ap.add_argument("--backward", action="store_true")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="ab", help="a / b / ab: run only site A's or B's forward through the module API (on main / side stream)")
    ap.add_argument("--bg", default=None, help="'matmul': an unrelated torch matmul loop on the other stream instead of the other site")
    ap.add_argument("--manual", action="store_true", help="the two module forwards on two streams by hand (no AdapterPair): allows --va / --vb")
    ap.add_argument("--va", default=None, help="variant of site A (ave / avvp / avqa / avs), default: the configuration's")
    ap.add_argument("--vb", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    c = dict(bench.CONFIGS[a.config], name=a.config)
    tdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    S = 2 * c["T"]
    g = torch.Generator().manual_seed(1234)
    for i, (Ca, Na, Cv, Nv, cnt) in enumerate(c["pairs"]):
        ca, cv = bench._oracle_cfgs(c, (Ca, Na, Cv, Nv))
        Pa, Ba = O.init_params(ca, seed=2 * i)
        Pv, Bv = O.init_params(cv, seed=2 * i + 1)
        fa = 0.3 * torch.randn(S, Na, Ca, generator=g)
        fv = 0.3 * torch.randn(S, Nv, Cv, generator=g)
        Ga, Gv = torch.randn(fa.shape, generator=g), torch.randn(fv.shape, generator=g)
        if i == a.shape:
            break
    A.torch.empty = _poisoned_empty
    L = capi.lib()
    ref = None
    bgA = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16) if a.bg else None
    bg_gemm = make_gemm_bg(a.bg[5:], dev) if (a.bg or "").startswith("gemm:") else None
    bg_reps = int(os.environ.get("BG_REPS", "60"))

    def background():
        for _ in range(bg_reps if bg_gemm else 40):
            if bg_gemm:
                bg_gemm()
            else:
                bgA @ bgA
    side = A.side_stream(dev)
    for it in range(a.runs):
        A.release_workspaces()
        c_a, c_b = dict(c, variant=a.va or c["variant"]), dict(c, variant=a.vb or c["variant"])
        ma, mv = bench.new_site(c_a, ca.Cx, ca.Nx, ca.Cy, ca.Ny), bench.new_site(c_b, cv.Cx, cv.Nx, cv.Cy, cv.Ny)
        ma.load_state_dict({**Pa, **Ba}, strict=False); mv.load_state_dict({**Pv, **Bv}, strict=False)
        for m in (ma, mv):
            m.to(dev).train()
            dbg.keep_saved(m)
        xa_, xv_ = fa.to(dev, tdt).requires_grad_(a.backward), fv.to(dev, tdt).requires_grad_(a.backward)
        xa, xv = xa_.permute(0, 2, 1).unsqueeze(-1), xv_.permute(0, 2, 1).unsqueeze(-1)
        torch.cuda.synchronize()
        outs = {}
        if a.manual:
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                rb = mv(xv, xa)
            ra = ma(xa, xv)
            torch.cuda.synchronize()
            outs["out_a"], outs["out_v"] = ra[0].detach().float().cpu(), rb[0].detach().float().cpu()
        elif a.bg:            # one site on its stream, unrelated work on the other
            main = torch.cuda.current_stream(dev)
            if a.only == "b":
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    r = mv(xv, xa)
                background()
            else:
                side.wait_stream(main)
                if a.backward:                 # forward alone, then the backward beside the background work
                    r = ma(xa, xv)
                    torch.cuda.synchronize()
                    with torch.cuda.stream(side):
                        background()
                    r[0].backward(Ga.to(dev).to(r[0].dtype).permute(0, 2, 1).unsqueeze(-1))
                else:
                    r = ma(xa, xv)
                    with torch.cuda.stream(side):
                        background()
            torch.cuda.synchronize()
            outs["out"] = r[0].detach().float().cpu()
        else:
            pair = A.AdapterPair(ma, mv, concurrent=(a.mode == "concurrent"))
            r = pair(xa, xv)
            if c["variant"] == "avvp":
                out_a, _la, out_v, _lv = r
            elif c["variant"] == "avs":
                out_a, out_v = r[0], r[4]
            else:
                out_a, _ia, out_v, _iv = r
            if a.backward:
                loss = (out_a.squeeze(-1).permute(0, 2, 1).float() * Ga.to(dev)).sum() + (out_v.squeeze(-1).permute(0, 2, 1).float() * Gv.to(dev)).sum()
                loss.backward()
            torch.cuda.synchronize()
            outs["out_a"], outs["out_v"] = out_a.detach().float().cpu(), out_v.detach().float().cpu()
        snap = {}
        for tag, m in (("a", ma), ("v", mv)):
            st = m.__dict__.get("_last_saved")
            if st is None:
                continue
            desc, saved = st
            table = cm.buffer_table(L, desc)
            # the scratch of the stream the site ran on
            scr = None
            for key, buf in A._SCRATCH.items():
                if (tag == "v") == (key[1] == side.cuda_stream) and key[2] == 0:
                    scr = buf
            for name, region, off, nb in table:
                src = saved if region == 0 else scr
                if src is None or off + nb > src.numel():
                    continue
                snap[f"{tag}.{'sv' if region == 0 else 'sc'}.{name}"] = src[off:off + nb].cpu()
        snap.update({k: v.view(torch.uint8).reshape(-1) for k, v in outs.items()})
        if ref is None:
            ref = snap
            continue
        moved = [(k, int((v != ref[k]).sum()), v.numel()) for k, v in snap.items() if k in ref and v.numel() == ref[k].numel() and not torch.equal(v, ref[k])]
        print(f"run {it}: {len(moved)} buffers moved: " + "; ".join(f"{k} {n}/{t}" for k, n, t in moved[:40]), flush=True)
        for key, dt_ in (("v.sv.rmu", torch.float32), ("v.sv.Z", tdt), ("v.sv.sx", torch.float32), ("a.sv.rmu", torch.float32), ("a.sv.Z", tdt),
                         ("a.sc.colpart", torch.float32), ("a.sc.dsm", torch.float32), ("v.sc.colpart", torch.float32)):
            if key in snap and key in ref and not torch.equal(snap[key], ref[key]):
                x, y = snap[key].view(dt_).float(), ref[key].view(dt_).float()
                idx = (x != y).nonzero().reshape(-1)
                NT_ = S * (cv.Nx if key[0] == "v" else ca.Nx)
                print(f"   {key}: {idx.numel()} elements; nan now {int(torch.isnan(x).sum())} ref {int(torch.isnan(y).sum())}; first idx {idx[:24].tolist()}  (NT={NT_}, row width {x.numel() // NT_ if 'Z' in key else '-'})")
                print("      now", [round(float(v), 5) for v in x[idx[:8]]], " ref", [round(float(v), 5) for v in y[idx[:8]]])


if __name__ == "__main__":
    main()
