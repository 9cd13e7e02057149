"""dev: the largest parameter-gradient errors of the 32-site block loop against the fp64 recording (tests/test_blockloop_golden.py), as
err / max(own scale, 1e-2 of the largest gradient):   python tests/dev/blockloop_errors.py [top]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch import nn
from tests import test_blockloop_golden as T
from tests.test_adapters_api import build_module
from avmoe_amd.blocks import DualBackboneLoop
LF = T.LF
dev = torch.device("cuda:0")
meta, t = T.load()
lists = {}
for name in T.LISTS:
    mods = []
    for i, (nv, na) in enumerate(LF.site_shapes()):
        m = build_module("ave", T.site_cfg(name, nv, na))
        m.load_state_dict(T._states(t, name, i), strict=True)
        mods.append(m.to(dev).train())
    lists[name] = nn.ModuleList(mods)
vs, as_, _ = LF.make_stages()
f_v, f_a = t["f_v"].to(dev).requires_grad_(True), t["f_a"].to(dev).requires_grad_(True)
loop = DualBackboneLoop(*[lists[n] for n in T.LISTS], num_skip=LF.NUM_SKIP, fuse_residual=True)
fin_v, fin_a, rec = loop(vs, as_, f_v, f_a)
torch.autograd.backward([fin_v, fin_a], [t["G_v"].to(dev), t["G_a"].to(dev)])
torch.cuda.synchronize()
keys = [k for k in t if k.startswith("grad.") and k not in ("grad.f_v", "grad.f_a")]
gmax = max(float(t[k].abs().max()) for k in keys)
rows = []
for k in keys:
    name, i, pk = k[len("grad."):].split(".", 2)
    got = dict(lists[name][int(i)].named_parameters())[pk].grad.detach().float().cpu()
    err, sc = float((got - t[k]).abs().max()), float(t[k].abs().max())
    rows.append((err / max(sc, 1e-2 * gmax), err, sc, k))
rows.sort(reverse=True)
for r in rows[: int(sys.argv[1]) if len(sys.argv) > 1 else 8]:
    print(f"{r[0]:.2e}  err {r[1]:.3e}  scale {r[2]:.3e}  {r[3]}")
if os.environ.get("BLOCKLOOP_DUMP"):      # every parameter gradient + the site configurations, for a diff between two builds / settings
    out = {}
    for k in keys:
        name, i, pk = k[len("grad."):].split(".", 2)
        out[k] = dict(lists[name][int(i)].named_parameters())[pk].grad.detach().float().cpu()
    out["fin_v"], out["fin_a"] = fin_v.detach().cpu(), fin_a.detach().cpu()
    torch.save(out, os.environ["BLOCKLOOP_DUMP"])
    for name in T.LISTS:
        for i, (nv, na) in enumerate(LF.site_shapes()):
            c = T.site_cfg(name, nv, na)
            print("site", name, i, "C", c.Cx, "N", c.Nx, "Cy", c.Cy, "M", c.Ny, "E", c.E_m, c.E_s, "r", c.reduction, "g", c.groups, "K", c.K)
