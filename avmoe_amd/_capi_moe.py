"""ctypes declarations + thin helpers for the MoE-adapter section of include/avmoe.h."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

MAX_EXPERTS = 16
VARIANT = {"ave": 0, "avqa": 0, "avvp": 1, "avs": 2}
SELF_ATTN = {"none": 0, "v2": 1, "nxn": 2, "v1": 3}


class MoeDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("S", "N", "C", "M", "Cy", "E_m", "E_s", "d", "groups", "K", "use_bn",
                                         "use_gate", "ln_before", "ln_post", "variant", "self_attn", "lb_loss",
                                         "dtype", "training")] + \
               [(n, C.c_float) for n in ("bn_eps", "ln_eps", "bn_momentum")] + \
               [(n, C.c_int32) for n in ("accumulate_dx", "accumulate_dy", "accumulate_out", "shared_gpu")]


_EXPERT_FIELDS = ("gate", "my_tokens", "gate_lat", "down_w", "up_w", "bn1_w", "bn1_b", "bn2_w", "bn2_b",
                  "lnb_w", "lnb_b", "lnp_w", "lnp_b", "bn1_rm", "bn1_rv", "bn2_rm", "bn2_rv",
                  "sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b", "sa_keep", "bn1_nbt", "bn2_nbt")
_TOP_FIELDS = ("conv_w", "conv_b", "fc_w", "fc_b", "r0_w", "r0_b", "r2_w", "r2_b", "r4_w", "r4_b")


class ExpertPtrs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _EXPERT_FIELDS]


class MoePtrs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _TOP_FIELDS] + [("e", ExpertPtrs * MAX_EXPERTS)]


SA_KEEP = "self_attention.__keep__"

# state_dict leaf (relative to the expert prefix) -> ExpertPtrs field
EXPERT_KEY_TO_FIELD = {
    "gate": "gate", "my_tokens": "my_tokens", "gate_av": "gate_lat", "gate_self": "gate_lat",
    "down_sampler.weight": "down_w", "up_sampler.weight": "up_w",
    "bn1.weight": "bn1_w", "bn1.bias": "bn1_b", "bn2.weight": "bn2_w", "bn2.bias": "bn2_b",
    "ln_before.weight": "lnb_w", "ln_before.bias": "lnb_b", "ln_post.weight": "lnp_w", "ln_post.bias": "lnp_b",
    "bn1.running_mean": "bn1_rm", "bn1.running_var": "bn1_rv", "bn2.running_mean": "bn2_rm", "bn2.running_var": "bn2_rv",
    "self_attention.in_proj_weight": "sa_in_w", "self_attention.in_proj_bias": "sa_in_b",
    "self_attention.out_proj.weight": "sa_out_w", "self_attention.out_proj.bias": "sa_out_b",
    SA_KEEP: "sa_keep",            # not a state_dict entry: the dropout multiplier of one call (include/avmoe.h)
    "bn1.num_batches_tracked": "bn1_nbt", "bn2.num_batches_tracked": "bn2_nbt",      # int64 counters, bumped inside the forward (ABI 6)
}
INT64_FIELDS = ("bn1_nbt", "bn2_nbt")
TOP_KEY_TO_FIELD = {
    "conv_adapter.weight": "conv_w", "conv_adapter.bias": "conv_b", "fc.weight": "fc_w", "fc.bias": "fc_b",
    "router.0.weight": "r0_w", "router.0.bias": "r0_b", "router.2.weight": "r2_w", "router.2.bias": "r2_b",
    "router.4.weight": "r4_w", "router.4.bias": "r4_b",
}


def declare(L):
    L.avmoe_adam_step.restype = C.c_int
    L.avmoe_adam_step.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_float] * 5 + [C.c_int64, C.c_float, C.c_void_p]
    L.avmoe_add2.restype = C.c_int
    L.avmoe_add2.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    L.avmoe_router_topk.restype = C.c_int
    L.avmoe_router_topk.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.avmoe_expert_histogram.restype = C.c_int
    L.avmoe_expert_histogram.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    L.avmoe_router_forward.restype = C.c_int
    L.avmoe_router_forward.argtypes = [C.POINTER(MoeDesc), C.c_void_p, C.POINTER(MoePtrs)] + [C.c_void_p] * 7
    for fn in (L.avmoe_expert_forward_cross, L.avmoe_expert_forward_uni):      # sub-ops (ABI 7): one expert's output alone
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(MoeDesc), C.c_void_p, C.c_void_p, C.POINTER(MoePtrs), C.c_int32] + [C.c_void_p] * 4
    L.avmoe_remap_forward.restype = C.c_int                                    # the remap materialised
    L.avmoe_remap_forward.argtypes = [C.POINTER(MoeDesc), C.c_void_p, C.POINTER(MoePtrs)] + [C.c_void_p] * 5
    L.avmoe_moe_saved_bytes.restype = C.c_size_t
    L.avmoe_moe_saved_bytes.argtypes = [C.POINTER(MoeDesc)]
    L.avmoe_moe_scratch_bytes.restype = C.c_size_t
    L.avmoe_moe_scratch_bytes.argtypes = [C.POINTER(MoeDesc)]
    L.avmoe_moe_forward.restype = C.c_int
    L.avmoe_moe_forward.argtypes = [C.POINTER(MoeDesc), C.c_void_p, C.c_void_p, C.POINTER(MoePtrs), C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.avmoe_moe_backward.restype = C.c_int
    L.avmoe_moe_backward.argtypes = [C.POINTER(MoeDesc), C.c_void_p, C.c_void_p, C.POINTER(MoePtrs), C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(MoePtrs),
                                     C.c_void_p]
    L.avmoe_moe_backward_part.restype = C.c_int
    L.avmoe_moe_backward_part.argtypes = [C.POINTER(MoeDesc), C.c_void_p, C.c_void_p, C.POINTER(MoePtrs), C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(MoePtrs),
                                          C.c_int32, C.c_void_p]
    if hasattr(L, "avmoe_moe_backward_dx_dy"):       # (ABI 10; a development A/B may load an older library through AVMOE_LIB, with AVMOE_NO_FUSED_DX=1)
        L.avmoe_moe_backward_dx_dy.restype = C.c_int
        L.avmoe_moe_backward_dx_dy.argtypes = [C.POINTER(MoeDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(MoeDesc), C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p]
    L.avmoe_moe_buffer_info.restype = C.c_int
    L.avmoe_moe_buffer_info.argtypes = [C.POINTER(MoeDesc), C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]


def expert_prefixes(E_m: int, E_s: int) -> List[str]:
    return [f"multimodal_experts.{j}" for j in range(E_m)] + [f"singlemodal_experts.{j}" for j in range(E_s)]


def make_ptrs(tensors: Dict[str, "object"], E_m: int, E_s: int) -> MoePtrs:
    """Fill a MoePtrs from {state_dict key: CUDA fp32 tensor}.  Missing keys stay NULL.  The caller keeps
    the tensors alive for the duration of the call."""
    import torch
    P = MoePtrs()
    for k, f in TOP_KEY_TO_FIELD.items():
        t = tensors.get(k)
        if t is not None:
            assert t.dtype == torch.float32 and t.is_contiguous(), k
            setattr(P, f, t.data_ptr())
    for j, pre in enumerate(expert_prefixes(E_m, E_s)):
        for leaf, f in EXPERT_KEY_TO_FIELD.items():
            t = tensors.get(f"{pre}.{leaf}")
            if t is not None:
                assert t.dtype == (torch.int64 if f in INT64_FIELDS else torch.float32) and t.is_contiguous(), (pre, leaf)
                setattr(P.e[j], f, t.data_ptr())
    return P


class PtrFiller:
    """Fills MoePtrs structs for a FIXED list of state_dict keys: the key -> (expert, field) resolution is done once, a call then
    only writes one data_ptr per tensor (the per-call cost of the facade matters at small batches: 48 sites per AVE step)."""

    def __init__(self, keys, E_m: int, E_s: int):
        pre = expert_prefixes(E_m, E_s)
        self.slots = []                                   # (expert index or -1, field name) per key; None for keys the ABI does not take
        for k in keys:
            slot = None
            if k in TOP_KEY_TO_FIELD:
                slot = (-1, TOP_KEY_TO_FIELD[k])
            else:
                for j, pr in enumerate(pre):
                    if k.startswith(pr + "."):
                        f = EXPERT_KEY_TO_FIELD.get(k[len(pr) + 1:])
                        if f is not None:
                            slot = (j, f)
                        break
            self.slots.append(slot)

    def fill(self, P: MoePtrs, tensors, base_ptr: int = 0, offsets=None):
        """tensors: in key order (None entries are skipped; the caller vouches for contiguous fp32); with `offsets` (elements of
        fp32) the pointers are base_ptr + 4 * offset."""
        ex = {}                                           # P.e[j] builds a new ctypes view on every access: fetch each once
        for i, slot in enumerate(self.slots):
            if slot is None:
                continue
            if offsets is not None:
                ptr = base_ptr + 4 * offsets[i]
            else:
                t = tensors[i]
                if t is None:
                    continue
                ptr = t.data_ptr()
            j = slot[0]
            if j < 0:
                setattr(P, slot[1], ptr)
            else:
                e = ex.get(j)
                if e is None:
                    e = ex[j] = P.e[j]
                setattr(e, slot[1], ptr)
        return P


def buffer_table(L, desc: MoeDesc):
    """[(name, region, offset, bytes)] of the workspace layout for `desc`."""
    out = []
    i = 0
    while True:
        name, region, off, nb = C.c_char_p(), C.c_int32(), C.c_size_t(), C.c_size_t()
        st = L.avmoe_moe_buffer_info(C.byref(desc), i, C.byref(name), C.byref(region), C.byref(off), C.byref(nb))
        if st != 0:
            break
        out.append((name.value.decode(), region.value, off.value, nb.value))
        i += 1
    return out
