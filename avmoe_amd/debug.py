"""Checker-side views into the `saved` workspace of an adapter site (development / parity tooling).

Nothing here is on the forward / backward path: `keep_saved(module)` only makes the facade remember the workspace of the
module's LAST forward call, and `relu_masks(module)` reads -- through `avmoe_moe_buffer_info`, the same named-buffer table the
GPU tests use -- which ReLU units of the cross-modal experts (net_trans_v3.py:400) the HIP path switched on in that call.
bench.py's parity leg and the tests hand those masks to the oracle so that gradients are compared on the SAME side of the
kink for the few units whose pre-activation lies within rounding of zero.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import _capi as capi
from . import _capi_moe as cm


def keep_saved(module, on: bool = True):
    """Make `module` (a MoEAdapter) remember (descriptor, saved workspace) of its last forward call."""
    module.__dict__["_keep_saved"] = bool(on)
    if not on:
        module.__dict__.pop("_last_saved", None)
    return module


def _layout(desc):
    table = {n: (r, o, b) for (n, r, o, b) in cm.buffer_table(capi.lib(), desc)}
    E = desc.E_m + desc.E_s
    NT = desc.S * desc.N
    DZ = table["wsum"][2] // 4
    g = 1 if table["mWd"][2] > 4 else desc.groups          # merged groups: the site runs as ONE group on block-diagonal weights
    dgp = DZ // (E * g)
    zsz = table["Z"][2] // (NT * DZ)
    return table, E, NT, DZ, g, dgp, zsz


def relu_masks(module) -> Dict[str, torch.Tensor]:
    """{expert prefix: bool (S, N, d)} -- True where the HIP path's ReLU let the unit through (BN1(z) > 0), for every
    cross-modal expert of the module's last forward call (needs keep_saved(module) before that call)."""
    st = module.__dict__.get("_last_saved")
    if st is None:
        raise capi.AvmoeError("relu_masks: call keep_saved(module) before the forward")
    return relu_masks_of(*st)


def relu_masks_of(desc, saved) -> Dict[str, torch.Tensor]:
    """The same from a descriptor and the `saved` workspace (uint8 tensor) a forward call through the C ABI filled."""
    table, E, NT, DZ, g, dgp, zsz = _layout(desc)
    torch.cuda.synchronize(saved.device)

    def buf(name, dtype):
        _r, off, nb = table[name]
        return saved[off:off + nb].view(dtype)

    Z = buf("Z", torch.bfloat16 if zsz == 2 else torch.float32)[:NT * DZ].float().reshape(desc.S, desc.N, g, E, dgp)
    bn1 = buf("bn1", torch.float32)[:4 * DZ].reshape(4, g, E, dgp)
    # fp64: the exact sign of z * scale + shift -- what the kernels' fused multiply-add rounds (sign-preservingly) to fp32
    y = Z.double() * bn1[2].double() + bn1[3].double()
    dg = desc.d // g
    out = {}
    for j, pre in enumerate(cm.expert_prefixes(desc.E_m, desc.E_s)):
        if j < desc.E_m:
            out[pre] = (y[:, :, :, j, :dg] > 0).reshape(desc.S, desc.N, desc.d).cpu()
    return out
