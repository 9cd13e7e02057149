"""Host logic of avmoe_amd.train (CPU): the reference's parameter-selection rule and adapter-only checkpoints."""
import os
import tempfile
from types import SimpleNamespace as NS

import torch
from torch import nn

from avmoe_amd import train as T
from oracle import avmoe_oracle as O
from tests.test_adapters_api import build_module


class _Toy(nn.Module):
    """named like the reference's MMIL_Net members (net_trans_v3.py:520-660)"""

    def __init__(self):
        super().__init__()
        cfg = O.AdapterConfig(Cx=32, Nx=10, Cy=24, Ny=6, reduction=4, groups=2, K=4)
        self.ViT = nn.ModuleDict({"blk": nn.Linear(4, 4), "norm": nn.LayerNorm(4)})
        self.htsat = nn.Linear(4, 4)
        self.audio_moe_adapter_blocks_p1 = nn.ModuleList([build_module("ave", cfg)])
        self.mlp_class = nn.Linear(4, 2)
        self.other = nn.Linear(3, 3)


def test_select_trainable_follows_the_reference_rule():
    m = _Toy()
    groups = T.select_trainable(m, lr=1e-3, lr_mlp=5e-4, is_vit_ln=True)
    by = {g["name"]: g for g in groups}
    assert len(groups) == len(list(m.named_parameters()))
    assert by["ViT.norm.weight"]["params"].requires_grad and not by["ViT.blk.weight"]["params"].requires_grad
    assert not by["htsat.weight"]["params"].requires_grad and not by["other.weight"]["params"].requires_grad
    assert all(g["params"].requires_grad for n, g in by.items() if "adapter_blocks" in n)
    assert by["mlp_class.weight"]["lr"] == 5e-4 and by["mlp_class.weight"]["params"].requires_grad
    assert by["audio_moe_adapter_blocks_p1.0.fc.weight"]["lr"] == 1e-3
    T.select_trainable(m, lr=1e-3, lr_mlp=5e-4, is_vit_ln=False)
    assert not by["ViT.norm.weight"]["params"].requires_grad


def test_adapter_only_checkpoint_roundtrip_and_prefix_strip():
    torch.manual_seed(0)
    a, b = _Toy(), _Toy()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "adapters.pt")
        T.save_adapters(a, path)
        sd = torch.load(path)
        assert sd and all("adapter" in k for k in sd)
        assert "audio_moe_adapter_blocks_p1.0.multimodal_experts.0.bn1.num_batches_tracked" in sd
        res = T.load_adapters(b, path)
        assert not [k for k in res.missing_keys if "adapter" in k] and not res.unexpected_keys
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        if "adapter" in k:
            assert torch.equal(x, y), k
    full = a.state_dict()                                  # a released full checkpoint loads with strict=False too
    T.load_adapters(_Toy(), full)
    del sd["audio_moe_adapter_blocks_p1.0.fc.weight"]
    try:
        T.load_adapters(_Toy(), sd)
        raise AssertionError("a checkpoint without an adapter entry must be reported")
    except KeyError:
        pass
    assert T.strip_prefix({"sed_model.layers.0.w": 1}) == {"layers.0.w": 1}


def test_plain_bucket_offsets_are_aligned_and_flat_adam_ranges_cover_the_bucket():
    """dp._Bucket pads every parameter to 64 elements (FlatAdam re-points param.data to the bucket offsets and the
    GEMM engine needs 16-byte aligned operands); FlatAdam's per-learning-rate ranges tile the whole bucket in order."""
    from avmoe_amd.dp import _Bucket
    ps = [nn.Parameter(torch.zeros(n)) for n in (1, 2, 130, 64, 7)]
    b = _Bucket()
    for p in ps:
        b.add_param(p)
    b.materialize(torch.device("cpu"))
    assert b.offsets == [0, 64, 128, 320, 384] and b.flat.numel() == 448
    for p, o in zip(ps, b.offsets):
        assert p.grad.data_ptr() == b.flat.data_ptr() + 4 * o and p.grad.data_ptr() % 16 == 0
