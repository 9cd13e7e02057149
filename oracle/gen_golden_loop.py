#!/usr/bin/env python3
"""Golden vector of the reference's OWN dual-backbone block loop.  TEST INFRASTRUCTURE ONLY; runs only where /root/reference
exists (the build container).

The loop lives inside `MMIL_Net.forward_swin` (AVE/nets/net_trans_v3.py:639-759).  The real model cannot be built here (timm,
HTS-AT and their checkpoints are absent), but the loop itself only touches a small surface of the two backbones.  So: a BARE
instance of the reference class (`object.__new__(MMIL_Net)` + `nn.Module.__init__`) gets stand-in backbones
(tests/loop_fakes.py: parameter-free blocks, a Swin (2, 2, 18, 2) vs HTS-AT (2, 2, 6, 2) stage layout so that the
reference's 18-entry alignment list and `num_skip = 2` are exercised) and four lists of the reference's own MoEAdapter
modules; then the reference's `forward_swin` runs as written, forward + backward.  Recorded (data only):

    inputs f_v, f_a, upstream grads ; every adapter parameter / buffer ; the loop's final f_v, f_a ; the
    adapter_index_dict the reference returns ; grads wrt f_v, f_a and every adapter parameter.
The reference modules are evaluated in fp64 (`.double()`, stored as fp32): through 32 sites in a row the reference's own fp32
rounding reaches 2.5e-3 on the scalar gate gradients, which would make a 1e-3 bar against an fp32 recording meaningless.

    python oracle/gen_golden_loop.py        ->  tests/golden_loop/blockloop_ave.npz
"""
from __future__ import annotations

import json
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import gen_golden as GG                                   # noqa: E402  (stubs, BatchNorm workaround, import recipe)
from oracle.avmoe_oracle import AdapterConfig, init_params            # noqa: E402
from tests import loop_fakes as LF                                    # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden_loop")
LISTS = ("audio_moe_adapter_blocks_p1", "vis_moe_adapter_blocks_p1", "audio_moe_adapter_blocks_p2", "vis_moe_adapter_blocks_p2")


def site_cfg(list_name, nv, na):
    if list_name.startswith("audio"):
        return AdapterConfig(Cx=LF.CA, Nx=na, Cy=LF.CV, Ny=nv, E_m=LF.E_M, E_s=LF.E_S, reduction=LF.REDUCTION, groups=LF.GROUPS, K=LF.K_TOK)
    return AdapterConfig(Cx=LF.CV, Nx=nv, Cy=LF.CA, Ny=na, E_m=LF.E_M, E_s=LF.E_S, reduction=LF.REDUCTION, groups=LF.GROUPS, K=LF.K_TOK)


def main():
    GG._install_stubs()
    GG._install_bn_workaround()
    mod = GG._import_variant("ave")
    net = object.__new__(mod.MMIL_Net)
    nn.Module.__init__(net)
    shapes = LF.site_shapes()
    params = {}
    for li, name in enumerate(LISTS):
        sites = []
        for i, (nv, na) in enumerate(shapes):
            cfg = site_cfg(name, nv, na)
            m = GG._build_reference("ave", cfg)
            P, B = init_params(cfg, seed=5000 + 100 * li + i, randomize=True)
            m.load_state_dict({**P, **B}, strict=True)
            m.double()            # the 32-site-deep chain is evaluated in fp64: the stored vector is the reference's arithmetic, not its fp32 noise
            for k, v in {**P, **B}.items():
                params[f"{name}.{i}.{k}"] = v
            sites.append(m)
        setattr(net, name, nn.ModuleList(sites))
    vs, as_, rec_a = LF.make_stages()
    rec_v = LF.Record()
    f_v0, f_a0, G_v, G_a = LF.inputs()
    f_v = f_v0.double().requires_grad_(True)
    f_a = f_a0.double().requires_grad_(True)
    ident = lambda x: x
    net.swin = NS(patch_embed=lambda vis: f_v, layers=vs, norm=rec_v)
    net.htsat = NS(spectrogram_extractor=lambda a: torch.zeros(a.shape[0], 1, 4, 4), logmel_extractor=ident, bn0=ident, training=False,
                   freq_ratio=10 ** 6, spec_size=1, reshape_wav2img=ident, patch_embed=lambda a: f_a, ape=False, pos_drop=ident, layers=as_)
    net.opt = NS(num_skip=LF.NUM_SKIP, is_audio_adapter_p1=1, is_audio_adapter_p2=1, is_cmbs=1, is_temporal_att=1)
    net.temporal_attn = lambda v, a: (v, a, None)
    net.CMBS = lambda v, a: (None, None, None)
    margins = []
    for name in LISTS:
        for m in getattr(net, name):
            m.router.register_forward_hook(lambda _m, _i, o: margins.append(
                float((lambda p: (p[..., 0] - p[..., -1]).min())(torch.softmax(o.detach(), -1).reshape(-1, o.shape[-1]).sort(-1, descending=True).values))))
    net.train()
    assert LF.S == 10, "forward_swin hard-codes 10 frames per clip in its head (net_trans_v3.py:737)"
    audio = [torch.zeros(1, LF.S, 8)]
    vis = torch.zeros(1, LF.S, 1, 1, 1)
    _, _, _, index_dict = net.forward_swin(audio, vis, None)
    fin_v, fin_a = rec_v.seen, rec_a.seen
    assert min(margins) > 1e-4, f"router margin too small: {min(margins)}"
    ((fin_v * G_v.double()).sum() + (fin_a * G_a.double()).sum()).backward()

    arrays = {"f_v": f_v0.numpy(), "f_a": f_a0.numpy(), "G_v": G_v.numpy(), "G_a": G_a.numpy(),
              "out.f_v": fin_v.detach().float().numpy(), "out.f_a": fin_a.detach().float().numpy(),
              "grad.f_v": f_v.grad.float().numpy(), "grad.f_a": f_a.grad.float().numpy()}
    for k, v in params.items():
        arrays[f"state.{k}"] = v.numpy()
    ng = 0
    for name in LISTS:
        for k, p in getattr(net, name).named_parameters():
            arrays[f"grad.{name}.{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).float().numpy()
            ng += 1
        for k, b in getattr(net, name).named_buffers():
            arrays[f"newbuffer.{name}.{k}"] = (b.detach().float() if b.is_floating_point() else b.detach()).numpy()
    meta = dict(index_dict=index_dict, router_margin_min=min(margins), sites_per_list=len(shapes), torch=torch.__version__,
                reference="AVE/nets/net_trans_v3.py:639-759 forward_swin on tests/loop_fakes.py stand-ins")
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "blockloop_ave.npz")
    np.savez_compressed(path, **arrays)
    print(f"blockloop_ave: {os.path.getsize(path) / 1024:.1f} KiB, {4 * len(shapes)} sites, {ng} parameter gradients, "
          f"router margin {min(margins):.2e}, |f_v| {float(fin_v.abs().max()):.3f} |f_a| {float(fin_a.abs().max()):.3f}")


if __name__ == "__main__":
    main()
