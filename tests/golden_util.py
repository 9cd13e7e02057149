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
