"""avmoe_amd.blocks.DualBackboneLoop against a vector recorded from the REFERENCE's own block loop
(AVE/nets/net_trans_v3.py:639-727 `forward_swin`, driven by oracle/gen_golden_loop.py on the stand-in backbones of
tests/loop_fakes.py with the reference's MoEAdapter at all 32 adapter sites: Swin (2,2,18,2) x HTS-AT (2,2,6,2), num_skip 2).

  * CPU (not gpu): the loop's schedule with the pinned CPU oracle standing in for the sites -- which block pairs get adapters,
    the 18-vs-6 alignment, skipped stages, where the residuals are added, downsampling, the adapter_index_dict
  * GPU: the same loop with the HIP sites (AdapterPair on two streams, residual adds fused into the output GEMMs) -- final
    streams, expert indices, input gradients within 1e-3, every parameter gradient of every site within 3e-3 (fp32 kernels vs the fp64 recording)
"""
import json
import os

import numpy as np
import pytest
import torch
from torch import nn

from oracle import avmoe_oracle as O
from tests import loop_fakes as LF

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_loop", "blockloop_ave.npz")
LISTS = ("audio_moe_adapter_blocks_p1", "vis_moe_adapter_blocks_p1", "audio_moe_adapter_blocks_p2", "vis_moe_adapter_blocks_p2")


def load():
    z = np.load(PATH)
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, {k: torch.from_numpy(np.array(z[k])) for k in z.files if k != "meta"}


def site_cfg(list_name, nv, na):
    if list_name.startswith("audio"):
        return O.AdapterConfig(Cx=LF.CA, Nx=na, Cy=LF.CV, Ny=nv, E_m=LF.E_M, E_s=LF.E_S, reduction=LF.REDUCTION, groups=LF.GROUPS, K=LF.K_TOK)
    return O.AdapterConfig(Cx=LF.CV, Nx=nv, Cy=LF.CA, Ny=na, E_m=LF.E_M, E_s=LF.E_S, reduction=LF.REDUCTION, groups=LF.GROUPS, K=LF.K_TOK)


class OracleSite(nn.Module):
    """The pinned CPU oracle behind the reference's site call signature (test infrastructure: schedule check on the CPU)."""

    def __init__(self, cfg, state):
        super().__init__()
        self.cfg = cfg
        self.P = nn.ParameterDict({k.replace(".", "__"): nn.Parameter(v.clone()) for k, v in state.items() if v.is_floating_point() and "running" not in k})
        self.B = {k: v.clone() for k, v in state.items() if "running" in k or not v.is_floating_point()}

    def forward(self, x, y):
        X, Y = x.squeeze(-1).permute(0, 2, 1), y.squeeze(-1).permute(0, 2, 1)
        P = {k.replace("__", "."): v for k, v in self.P.items()}
        f = O.moe_forward(P, self.B, X, Y, self.cfg, training=True, update_buffers=False)
        return f["out"].permute(0, 2, 1).unsqueeze(-1), f["idx"].unsqueeze(-1)


def _states(t, name, i):
    pre = f"state.{name}.{i}."
    return {k[len(pre):]: v for k, v in t.items() if k.startswith(pre)}


def _check(meta, t, fin_v, fin_a, rec, gv, ga, grad_of, tol=1e-3, ptol=None, floor=1e-3):
    ptol = tol if ptol is None else ptol
    assert rec == meta["index_dict"]
    for got, key in ((fin_v, "out.f_v"), (fin_a, "out.f_a"), (gv, "grad.f_v"), (ga, "grad.f_a")):
        ref = t[key]
        assert float((got.detach().float().cpu() - ref).abs().max()) <= tol * float(ref.abs().max()), key
    keys = [k for k in t if k.startswith("grad.") and k not in ("grad.f_v", "grad.f_a")]
    gmax = max(float(t[k].abs().max()) for k in keys)
    bad = {}
    for k in keys:
        got = grad_of(k[len("grad."):]).detach().float().cpu()
        err, sc = float((got - t[k]).abs().max()), float(t[k].abs().max())
        if err > ptol * max(sc, floor * gmax):
            bad[k] = (err, sc)
    assert not bad, bad
    assert len(keys) == 1088


def test_loop_schedule_matches_reference_loop_cpu():
    from avmoe_amd.blocks import DualBackboneLoop
    meta, t = load()
    shapes = LF.site_shapes()
    # fp64 like the recording: the schedule (and the oracle) then agree with the reference's loop to 1e-11; in fp32 the same
    # chain of 32 sites moves the most cancellation-prone gradients (router.4.bias, gates) by up to 1.8e-3
    dbl = lambda st: {k: (v.double() if v.is_floating_point() else v) for k, v in st.items()}
    lists = {name: nn.ModuleList([OracleSite(site_cfg(name, nv, na), dbl(_states(t, name, i))) for i, (nv, na) in enumerate(shapes)])
             for name in LISTS}
    vs, as_, _ = LF.make_stages()
    f_v, f_a = t["f_v"].double().requires_grad_(True), t["f_a"].double().requires_grad_(True)
    loop = DualBackboneLoop(lists[LISTS[0]], lists[LISTS[1]], lists[LISTS[2]], lists[LISTS[3]], num_skip=LF.NUM_SKIP)
    fin_v, fin_a, rec = loop(vs, as_, f_v, f_a)
    torch.autograd.backward([fin_v, fin_a], [t["G_v"].double(), t["G_a"].double()])

    def grad_of(key):                                   # "<list>.<i>.<param key>"
        name, i, pk = key.split(".", 2)
        return lists[name][int(i)].P[pk.replace(".", "__")].grad
    _check(meta, t, fin_v, fin_a, rec.to_dict(), f_v.grad, f_a.grad, grad_of, tol=1e-6)     # the fixture is stored in fp32


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [True, False])
def test_loop_with_hip_sites_matches_reference_loop(fuse):
    from avmoe_amd.blocks import DualBackboneLoop
    from tests.test_adapters_api import build_module
    dev = torch.device("cuda:0")
    meta, t = load()
    shapes = LF.site_shapes()
    lists = {}
    for name in LISTS:
        mods = []
        for i, (nv, na) in enumerate(shapes):
            m = build_module("ave", site_cfg(name, nv, na))
            m.load_state_dict(_states(t, name, i), strict=True)
            mods.append(m.to(dev).train())
        lists[name] = nn.ModuleList(mods)
    vs, as_, _ = LF.make_stages()
    f_v, f_a = t["f_v"].to(dev).requires_grad_(True), t["f_a"].to(dev).requires_grad_(True)
    loop = DualBackboneLoop(lists[LISTS[0]], lists[LISTS[1]], lists[LISTS[2]], lists[LISTS[3]], num_skip=LF.NUM_SKIP, fuse_residual=fuse)
    fin_v, fin_a, rec = loop(vs, as_, f_v, f_a)
    torch.autograd.backward([fin_v, fin_a], [t["G_v"].to(dev), t["G_a"].to(dev)])
    torch.cuda.synchronize()

    def grad_of(key):
        name, i, pk = key.split(".", 2)
        return dict(lists[name][int(i)].named_parameters())[pk].grad
    # fp32 kernels against the fp64 recording, 32 sites deep: streams and input gradients within 1e-3; parameter gradients within
    # 3e-3 of max(own scale, 1e-2 of the largest gradient).  The tensors that need the floor are the router's last-layer bias and
    # the scalar gates of late sites: sums over every token of terms of both signs whose fp32 rounding is set by the size of the
    # terms, not of the sum (eager fp32 PyTorch over the same chain: 1.2e-3 on that scale)
    _check(meta, t, fin_v, fin_a, rec.to_dict(), f_v.grad, f_a.grad, grad_of, tol=1e-3, ptol=3e-3, floor=1e-2)
    # train-mode BatchNorm running statistics after the pass
    for k, ref in t.items():
        if k.startswith("newbuffer.") and ref.is_floating_point():
            name, i, bk = k[len("newbuffer."):].split(".", 2)
            got = dict(lists[name][int(i)].named_buffers())[bk].float().cpu()
            assert float((got - ref).abs().max()) <= 1e-3 * max(float(ref.abs().max()), 1e-3), k
