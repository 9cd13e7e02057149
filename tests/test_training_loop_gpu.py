"""Everything either side of the path together on one GPU: frozen stand-in backbones, the reference's parameter selection,
DualBackboneLoop with fused residuals, gradient sink + AdapterGradReducer, FlatAdam, expert-activation counters.  A few
optimizer steps on a fixed batch must drive a regression loss down -- an end-to-end check that gradients, buckets and the
optimizer are wired to the same parameters."""
import pytest
import torch
from torch import nn

from oracle import avmoe_oracle as O
from tests.test_adapters_gpu import build_module
from tests.test_blocks import Stage
from tests.test_blocks_gpu import AudBlock, VisBlock

pytestmark = pytest.mark.gpu


class TinyModel(nn.Module):
    def __init__(self, Cv, Nv, Ca, Na):
        super().__init__()
        self.swin = nn.Module(); self.htsat = nn.Module()
        self.swin.blocks = nn.ModuleList([VisBlock(Cv), VisBlock(Cv)])
        self.htsat.blocks = nn.ModuleList([AudBlock(Ca), AudBlock(Ca)])
        mk = lambda cx, nx, cy, ny: build_module("ave", O.AdapterConfig(Cx=cx, Nx=nx, Cy=cy, Ny=ny, reduction=4, groups=2, K=8))
        self.audio_moe_adapter_blocks_p1 = nn.ModuleList([mk(Ca, Na, Cv, Nv) for _ in range(2)])
        self.vis_moe_adapter_blocks_p1 = nn.ModuleList([mk(Cv, Nv, Ca, Na) for _ in range(2)])
        self.audio_moe_adapter_blocks_p2 = nn.ModuleList([mk(Ca, Na, Cv, Nv) for _ in range(2)])
        self.vis_moe_adapter_blocks_p2 = nn.ModuleList([mk(Cv, Nv, Ca, Na) for _ in range(2)])
        self.mlp_class = nn.Linear(Cv + Ca, 3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_adapters_train_end_to_end(dtype):
    from avmoe_amd.adapters import MoEAdapter
    from avmoe_amd.blocks import DualBackboneLoop
    from avmoe_amd.dp import AdapterGradReducer
    from avmoe_amd.train import FlatAdam, select_trainable, ExpertActivationCounter
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    S, Cv, Nv, Ca, Na = 6, 64, 100, 48, 160
    model = TinyModel(Cv, Nv, Ca, Na).to(dev)
    with torch.no_grad():                                   # the reference starts from zero gates (adapter output 0): open them a little
        for k, p in model.named_parameters():
            if k.endswith(("gate", "gate_av")):
                p.fill_(0.1)
    groups = select_trainable(model, lr=5e-3, lr_mlp=5e-3)
    trainable = [g_["params"] for g_ in groups if g_["params"].requires_grad]
    assert all(not p.requires_grad for p in model.swin.parameters()) and all(not p.requires_grad for p in model.htsat.parameters())
    sites = [m for m in model.modules() if isinstance(m, MoEAdapter)]
    red = AdapterGradReducer(trainable, sites=sites)
    opt = FlatAdam(red, lr=5e-3)
    model.swin.to(dtype); model.htsat.to(dtype)               # frozen backbones in the activation dtype
    loop = DualBackboneLoop(model.audio_moe_adapter_blocks_p1, model.vis_moe_adapter_blocks_p1,
                            model.audio_moe_adapter_blocks_p2, model.vis_moe_adapter_blocks_p2)
    stages_v = [Stage(model.swin.blocks, nn.Identity())]
    stages_a = [Stage(model.htsat.blocks, None)]
    g = torch.Generator().manual_seed(3)
    f_v0 = (0.5 * torch.randn(S, Nv, Cv, generator=g)).to(dev, dtype)
    f_a0 = (0.5 * torch.randn(S, Na, Ca, generator=g)).to(dev, dtype)
    target = torch.randn(S, 3, generator=g).to(dev)
    counter = ExpertActivationCounter(["audio_p1", "video_p1"], num_layers=2, num_experts=4, device=dev)
    losses = []
    for it in range(12):
        red.begin(sync=True)
        f_v, f_a, rec = loop(stages_v, stages_a, f_v0, f_a0)
        feat = torch.cat([f_v.float().mean(1), f_a.float().mean(1)], -1)
        loss = ((model.mlp_class(feat) - target) ** 2).mean()
        loss.backward()
        red.finish()
        opt.step()
        red.zero_grad()
        losses.append(float(loss.detach()))
        for layer in range(2):
            counter.update("audio_p1", layer, rec.entries["audio"]["p1"][layer])
            counter.update("video_p1", layer, rec.entries["video"]["p1"][layer])
    assert losses[-1] < 0.6 * losses[0], losses
    assert all(torch.isfinite(p).all() for p in trainable)
    c = counter.numpy()
    assert int(c["audio_p1"].sum()) == 12 * 2 * S and int(c["video_p1"].sum()) == 12 * 2 * S
