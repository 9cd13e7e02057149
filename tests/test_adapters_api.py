"""Host-side mirror of the reference module API (CPU): constructor signatures, state_dict keys / shapes for
every task variant (checked against the fixtures captured from the reference), and the no-fallback rule."""
from types import SimpleNamespace as NS

import pytest
import torch

from avmoe_amd import adapters, _capi
from tests.golden_util import load_golden, split_params


def _opt(cfg):
    return NS(num_conv_group=cfg.groups, is_before_layernorm=int(cfg.ln_before), is_post_layernorm=int(cfg.ln_post),
              is_self_attention=int(cfg.self_attn in ("v1", "v2")),
              self_attention_version=cfg.self_attn if cfg.self_attn in ("v1", "v2") else "v2",
              num_multimodal_experts=cfg.E_m, num_singlemodal_experts=cfg.E_s,
              use_load_balacing_loss=int(cfg.lb_loss), Adapter_downsample=cfg.reduction, is_bn=int(cfg.use_bn),
              is_gate=int(cfg.use_gate), num_tokens=cfg.K)


def build_module(which, cfg):
    """Construct the facade exactly the way the reference's task models construct their MoEAdapter."""
    common = dict(input_dim=cfg.Cx, output_dim=cfg.Cx, adapter_kind="bottleneck", dim_list=None, layer_idx=0,
                  opt=_opt(cfg), conv_dim_in=cfg.Ny, conv_dim_out=cfg.Nx, linear_in=cfg.Cy, linear_out=cfg.Cx)
    if which == "ave":
        return adapters.MoEAdapter(reduction_factor=cfg.reduction, use_bn=cfg.use_bn, use_gate=cfg.use_gate,
                                   num_tk=cfg.K, **common)
    if which in ("avs", "avs_ms3"):
        return adapters.MoEAdapterAVS(reduction_factor=cfg.reduction, use_bn=cfg.use_bn, use_gate=cfg.use_gate,
                                      num_tk=cfg.K, **common)
    if which == "avqa":
        return adapters.MoEAdapterAVQA(reduction_factor=cfg.reduction, use_bn=cfg.use_bn, use_gate=cfg.use_gate, **common)
    return adapters.MoEAdapterAVVP(**common)


@pytest.mark.parametrize("name", ["ave_train", "ave_nobn", "ave_noln_nogate", "avqa_train", "avvp_train",
                                  "avs_train_noise", "avs_v2_train", "avs_ms3_eval", "avs_k87_train", "avs_v1_train",
                                  "avs_v1_eval", "avvp_fast_train", "avs_v1_fast_train", "ave_ship_train"])
def test_state_dict_matches_reference_checkpoint_layout(name):
    meta, cfg, t = load_golden(name)
    P, B = split_params(t)
    m = build_module(meta["which"], cfg)
    sd = m.state_dict()
    assert set(sd.keys()) == set(P.keys()) | set(B.keys())
    for k, v in {**P, **B}.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    m.load_state_dict({**P, **B}, strict=True)            # released checkpoints drop in


def test_fresh_module_has_reference_default_init():
    _, cfg, _ = load_golden("ave_train")
    m = build_module("ave", cfg)
    e = m.multimodal_experts[0]
    assert float(e.gate) == 0.0 and float(e.gate_av) == 0.0                  # net_trans_v3.py:309,317
    assert 0.0 <= float(e.my_tokens.min()) and float(e.my_tokens.max()) < 1.0  # torch.rand, :315
    assert e.down_sampler.weight.shape == (cfg.d, cfg.Cx // cfg.groups, 1, 1) and e.down_sampler.bias is None


def test_no_cpu_fallback_and_unbuilt_variants_fail_loudly():
    _, cfg, t = load_golden("ave_train")
    m = build_module("ave", cfg)
    x = t["X"].permute(0, 2, 1).unsqueeze(-1)
    y = t["Y"].permute(0, 2, 1).unsqueeze(-1)
    with pytest.raises(_capi.AvmoeError):
        m(x, y)
    with pytest.raises(_capi.AvmoeError):
        m.multimodal_experts[0](x, y)
    with pytest.raises(NotImplementedError):
        adapters.ExpertAdapter(96, 96, "basic", opt=_opt(cfg))
    o = _opt(cfg)
    o.is_self_attention, o.self_attention_version = 1, "v1"       # AVS default version: nn.MultiheadAttention parameters
    e = adapters.ExpertAdapter(96, 96, "bottleneck", 8, o, is_multimodal=False, variant="avs")
    assert e.self_attention.in_proj_weight.shape == (288, 96) and e.self_attention.out_proj.weight.shape == (96, 96)


def test_library_exports_every_declared_symbol():
    import ctypes
    L = _capi.lib()
    for sym in _capi.exported_symbols():
        assert hasattr(L, sym), sym
    assert L.avmoe_abi_version() == 11


def test_descriptor_validation_without_a_gpu():
    """Host logic of the C ABI: workspace planning and descriptor errors need no device."""
    import ctypes as C
    from avmoe_amd import _capi_moe as cm
    from tests.moe_gpu_util import make_desc
    _, cfg, _ = load_golden("ave_train")
    L = _capi.lib()
    d = make_desc(cfg, 6, False, True)
    assert L.avmoe_moe_saved_bytes(C.byref(d)) > 0 and L.avmoe_moe_scratch_bytes(C.byref(d)) > 0
    names = [n for (n, _, _, _) in cm.buffer_table(L, d)]
    assert {"Z", "Apost", "Bpost", "Text", "dTy"} <= set(names) and len(names) == len(set(names))
    d.C = 100                                        # C / groups not a multiple of 8
    assert L.avmoe_moe_saved_bytes(C.byref(d)) == 0 and b"multiples of 8" in L.avmoe_last_error()
    d = make_desc(cfg, 6, False, True)
    d.E_m, d.E_s = 0, 0
    assert L.avmoe_moe_saved_bytes(C.byref(d)) == 0 and b"expert count" in L.avmoe_last_error()


def test_site_cache_follows_replaced_parameters():
    """The per-site bookkeeping caches NAMES and owners, never Parameter objects: a Parameter replaced after the first call
    (`load_state_dict(assign=True)`, attribute assignment) is the one the next call hands to the kernels; deepcopy / pickle do
    not carry the cache."""
    import copy
    import pickle
    _, cfg, _ = load_golden("ave_train")
    m = build_module("ave", cfg)
    first = m._param_tensors()
    assert list(first) == [k for k, _ in m.named_parameters()]
    new_gate = torch.nn.Parameter(torch.full((1,), 0.25))
    m.multimodal_experts[0].gate = new_gate                                  # attribute assignment
    assert m._param_tensors()["multimodal_experts.0.gate"] is new_gate
    sd = {k: v.clone() + 1.0 for k, v in m.state_dict().items() if v.is_floating_point()}
    m.load_state_dict(sd, strict=False, assign=True)                         # every Parameter object replaced
    now = m._param_tensors()
    for k, p in m.named_parameters():
        assert now[k] is p
        assert now[k] is not first[k]
    m.multimodal_experts[0].gate = None                                      # a parameter removed: registry rebuilt, not a stale object
    assert "multimodal_experts.0.gate" not in m._param_tensors()
    m2 = copy.deepcopy(m)
    assert "_avmoe_cache" not in m2.__dict__
    assert all(a is b for a, b in zip(m2._param_tensors().values(), m2.parameters()))
    m3 = pickle.loads(pickle.dumps(m))
    assert "_avmoe_cache" not in m3.__dict__ and list(m3._param_tensors()) == [k for k, _ in m3.named_parameters()]
