"""Helpers to load the golden fixtures written by oracle/gen_golden.py (data only)."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name, dtype=torch.float32):
    """-> (meta, cfg, tensors) ; tensors maps the npz keys to torch tensors."""
    from oracle.avmoe_oracle import AdapterConfig
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    cfg = AdapterConfig(**meta["cfg"])
    t = {}
    for k in z.files:
        if k == "meta":
            continue
        a = z[k]
        v = torch.from_numpy(np.array(a))
        if v.is_floating_point():
            v = v.to(dtype)
        t[k] = v
    return meta, cfg, t


def split_params(t):
    P = {k[len("param."):]: v for k, v in t.items() if k.startswith("param.")}
    B = {k[len("buffer."):]: v for k, v in t.items() if k.startswith("buffer.")}
    return P, B


def mha_keep_of(t):
    """{expert prefix: dropout multiplier} recorded from the reference's "v1" MultiheadAttention draw, or None."""
    d = {k[len("mha_keep."):]: v for k, v in t.items() if k.startswith("mha_keep.")}
    return d or None


def grad_errors(grads, t, keys=None):
    """Per-key error of `grads` against the fixture's `grad.<key>` entries.

    Returns {key: (max_abs_err, ref_scale)}.  Structurally-zero gradients exist (e.g. ln_before.bias
    in front of a train-mode BatchNorm: the shift is removed again), so callers compare the error to
    max(ref_scale, floor) where floor is a fraction of the largest gradient in the whole set."""
    out = {}
    for k, g in grads.items():
        if keys is not None and k not in keys:
            continue
        ref = t[f"grad.{k}"]
        out[k] = (float((g.detach().cpu().double() - ref.double()).abs().max()), float(ref.abs().max()))
    return out


def assert_grads_close(grads, t, rtol, floor_frac=1e-3, keys=None):
    errs = grad_errors(grads, t, keys)
    gmax = max(s for _, s in errs.values())
    bad = {k: (e, s) for k, (e, s) in errs.items() if e > rtol * max(s, floor_frac * gmax)}
    assert not bad, f"gradient mismatch (err, scale): {bad}"


def bf16_budget_violations(O, cfg, P, B, Xb, Yb, Gb, got, grads, training=True, lb_weight=0.0, noise=None, mha_keep=None, factor=2.0,
                           floor=1e-2, zero_norm=1e-6, zero_abs=1e-4):
    """Per-tensor bar for a bf16 run of the HIP path, anchored on what bf16 does to THIS computation: the oracle evaluated eagerly on
    the GPU under torch.autocast(bfloat16) (fp32 parameters, matmuls in bf16, softmax / norms in fp32 -- the reference's own
    formulation in bf16) against the fp32 oracle `grads`, both on the bf16-rounded inputs Xb / Yb / Gb.
        relnorm(hip) <= max(floor, factor * relnorm(eager bf16))             for every gradient tensor;
        structurally zero gradients (|g| < zero_norm of the largest gradient norm: a bias in front of a train-mode BatchNorm) are held
        in absolute terms to max(zero_abs of the largest norm, factor * the eager-bf16 error).
    Returns {key: (hip error, eager error)} of the tensors above their bar (empty = pass)."""
    import torch
    dev = torch.device("cuda:0")
    kd = None if mha_keep is None else {k: v.to(dev) for k, v in mha_keep.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, ge = O.moe_forward_backward({k: v.to(dev) for k, v in P.items()}, {k: v.to(dev) for k, v in B.items()}, Xb.to(dev), Yb.to(dev), cfg,
                                       Gb.to(dev), training=training, lb_weight=lb_weight, noise=None if noise is None else noise.to(dev),
                                       mha_keep=kd)
    # the tiny tensors (the scalar gates gate / gate_av / gate_self and the E-element bias of the router's last layer: each number a sum
    # over every token or frame of terms of both signs) are judged together, as ONE vector with a 3 % floor: a single one of them can
    # cancel to a small fraction of its terms, where "relative error" only measures that cancellation (eager bf16: 0.1 % .. 20 % on
    # individual gates at the same inputs)
    import torch as _t
    sk = [k for k, v in grads.items() if v.numel() <= 16]
    gv = {k: v for k, v in grads.items() if v.numel() > 16}
    hv = {k: got[k].float().cpu() for k in gv}
    ev = {k: ge[k].float().cpu() for k in gv}
    if sk:
        gv["<tiny tensors>"] = _t.cat([grads[k].reshape(-1) for k in sk])
        hv["<tiny tensors>"] = _t.cat([got[k].float().cpu().reshape(-1) for k in sk])
        ev["<tiny tensors>"] = _t.cat([ge[k].float().cpu().reshape(-1) for k in sk])
    nmax = max(float(v.norm()) for v in grads.values())
    bad = {}
    for k, v in gv.items():
        err = float((hv[k] - v).norm())
        err_e = float((ev[k] - v).norm())
        if k == "<tiny tensors>":
            rel, rel_e = err / float(v.norm()), err_e / float(v.norm())
            if rel > max(3e-2, factor * rel_e):
                bad[k] = (rel, rel_e)
            continue
        if float(v.norm()) < zero_norm * nmax:
            if err > max(zero_abs * nmax, factor * err_e):
                bad[k] = ("structurally zero", err / nmax, err_e / nmax)
            continue
        rel, rel_e = err / float(v.norm()), err_e / float(v.norm())
        if rel > max(floor, factor * rel_e):
            bad[k] = (rel, rel_e)
    return bad
