"""The steps either side of the adapter path on the reference's training loop, MI355X-side:

  * `select_trainable`      -- the parameter-freezing / LR-group rule of AVE/main_trans_v3.py:264-315
  * `FlatAdam`              -- torch.optim.Adam semantics as ONE HIP kernel per flat gradient bucket of an
                               `AdapterGradReducer` (parameters are re-pointed to views of a flat buffer with the bucket's
                               layout), StepLR-style decay; replaces optimizer.step() at main_trans_v3.py:136-138,322-323
  * `ExpertActivationCounter` -- per-layer expert-activation tables accumulated on the device (no idx.tolist() sync per
                               site), main_trans_v3.py:155-226
  * `save_adapters` / `load_adapters` / `strip_prefix` -- adapter-only checkpoints with the reference's state_dict keys
                               (strict=False load of released checkpoints, main_trans_v3.py:254; HTS-AT key strip,
                               net_trans_v3.py:560-563)

Everything here is host logic around three tiny C-ABI entry points (avmoe_adam_step, avmoe_expert_histogram); nothing
falls back to the CPU for GPU tensors."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional

import torch

from . import _capi as capi


# ---------------------------------------------------------------------------------------------------------------------
def select_trainable(model: torch.nn.Module, lr: float, lr_mlp: float, is_vit_ln: bool = False,
                     trainable_substrings=("adapter_blocks", "CMBS", "mlp_class", "temporal_attn"),
                     frozen_substrings=("htsat",), backbone_substrings=("ViT", "swin")) -> List[dict]:
    """Sets requires_grad exactly like the reference launcher and returns its Adam parameter groups: every parameter gets
    its own group, lr_mlp for names containing 'mlp_class', lr otherwise (AVE/main_trans_v3.py:264-315)."""
    groups = []
    for name, p in model.named_parameters():
        p.requires_grad = False
        if any(s in name for s in backbone_substrings):
            p.requires_grad = bool(is_vit_ln) and "norm" in name
        elif any(s in name for s in frozen_substrings):
            p.requires_grad = False
        elif any(s in name for s in trainable_substrings):
            p.requires_grad = True
        groups.append({"params": p, "lr": lr_mlp if "mlp_class" in name else lr, "name": name})
    return groups


# ---------------------------------------------------------------------------------------------------------------------
class FlatAdam:
    """Adam over the flat fp32 buckets of an AdapterGradReducer: per bucket ONE kernel updates parameters, exp_avg and
    exp_avg_sq in place (avmoe_adam_step).  Parameters of a bucket are moved into one flat buffer with the bucket's layout
    (`param.data` become views of it), so values, state_dict keys and autograd are unchanged.

        red = AdapterGradReducer(params, sites=sites)
        opt = FlatAdam(red, lr=args.lr, step_size=args.decay_epoch, gamma=args.decay)
        ...  red.begin(sync); loss.backward(); red.finish()
        if sync: opt.step(); red.zero_grad()
        opt.epoch_end()                      # StepLR.step()
    """

    def __init__(self, reducer, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 step_size: Optional[int] = None, gamma: float = 0.1, grad_scale: float = 1.0, param_groups=None):
        """param_groups: the list `select_trainable` returns ({"params": p, "lr": ...} per parameter, as handed to
        torch.optim.Adam at AVE/main_trans_v3.py:313-322): each parameter is stepped with ITS group's learning rate (the
        reference's `lr_mlp` for the classifier head vs `lr` for the adapters); parameters not listed use `lr`.  Inside a
        bucket, neighbouring parameters with the same rate share one kernel launch (an adapter site is one range)."""
        self.reducer, self.lr0, self.betas, self.eps, self.wd = reducer, lr, betas, eps, weight_decay
        self.step_size, self.gamma, self.grad_scale = step_size, gamma, grad_scale
        self.t, self.epoch = 0, 0
        self.state = []
        lr_of = {}
        for grp in (param_groups or []):
            ps = grp["params"]
            for p in ([ps] if isinstance(ps, torch.Tensor) else ps):
                lr_of[id(p)] = float(grp.get("lr", lr))
        for b in reducer.buckets:
            flat_g = b.flat
            if not flat_g.is_cuda:
                raise capi.AvmoeError("FlatAdam updates GPU buckets (no CPU fallback)")
            flat_p = torch.zeros_like(flat_g)
            spans = []                                   # (offset, end, lr) per parameter, in bucket order
            for p in b.params:                           # parameter offsets = offsets of their .grad views in the bucket
                off = (p.grad.data_ptr() - flat_g.data_ptr()) // 4
                view = flat_p[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                if view.data_ptr() % 16:
                    raise capi.AvmoeError("FlatAdam: a re-pointed parameter is not 16-byte aligned (the GEMM engine needs "
                                          "aligned operands); build the reducer with aligned buckets (avmoe_amd.dp)")
                spans.append((off, off + p.numel(), lr_of.get(id(p), float(lr))))
            spans.sort()
            ranges = []                                  # merged [begin, end, lr0]: alignment padding rides with its left neighbour
            for i, (o, e, r) in enumerate(spans):
                end = spans[i + 1][0] if i + 1 < len(spans) else flat_g.numel()
                if ranges and ranges[-1][2] == r:
                    ranges[-1][1] = end
                else:
                    ranges.append([o if ranges else 0, end, r])
            self.state.append(dict(p=flat_p, g=flat_g, m=torch.zeros_like(flat_g), v=torch.zeros_like(flat_g), ranges=ranges))

    @property
    def decay(self) -> float:
        """StepLR factor of the current epoch (AVE/main_trans_v3.py:323)."""
        return self.gamma ** (self.epoch // self.step_size) if self.step_size else 1.0

    @property
    def lr(self) -> float:
        return self.lr0 * self.decay

    def step(self):
        L = capi.lib()
        self.t += 1
        # 1 / world of a sum-reducing AdapterGradReducer(average="optimizer") rides in the kernel's gradient scale: no division pass
        scale = self.grad_scale * float(getattr(self.reducer, "grad_scale", 1.0))
        for s in self.state:
            for (o, e, r) in s["ranges"]:
                st = L.avmoe_adam_step(s["p"].data_ptr() + 4 * o, s["g"].data_ptr() + 4 * o, s["m"].data_ptr() + 4 * o,
                                       s["v"].data_ptr() + 4 * o, C.c_int64(e - o), C.c_float(r * self.decay),
                                       C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                       C.c_float(self.wd), C.c_int64(self.t), C.c_float(scale),
                                       torch.cuda.current_stream(s["p"].device).cuda_stream)
                capi.check(st, "avmoe_adam_step")

    def epoch_end(self):
        self.epoch += 1


# ---------------------------------------------------------------------------------------------------------------------
class ExpertActivationCounter:
    """counts[table][layer][expert] accumulated on the device from the `idx` tensors the sites return; `.numpy()` at the end
    of the evaluation is the only host sync (the reference calls idx.squeeze().tolist() after every site)."""

    def __init__(self, tables: Iterable[str], num_layers: int, num_experts: int, device):
        self.names = list(tables)
        self.L, self.E = num_layers, num_experts
        self.counts = torch.zeros(len(self.names), num_layers, num_experts, dtype=torch.int64, device=device)

    def update(self, table: str, layer: int, idx: torch.Tensor):
        if not idx.is_cuda or idx.dtype != torch.int64:
            raise capi.AvmoeError("expert indices must be an int64 GPU tensor (as returned by MoEAdapter.forward)")
        idx = idx.reshape(-1).contiguous()
        row = self.counts[self.names.index(table), layer]
        st = capi.lib().avmoe_expert_histogram(idx.data_ptr(), C.c_int64(idx.numel()), C.c_int32(self.E), row.data_ptr(),
                                               torch.cuda.current_stream(idx.device).cuda_stream)
        capi.check(st, "avmoe_expert_histogram")

    def numpy(self) -> Dict[str, "object"]:
        c = self.counts.cpu().numpy()
        return {n: c[i] for i, n in enumerate(self.names)}


def topk_experts(probs: torch.Tensor, k: int) -> torch.Tensor:
    """(S, k) int64: the k most probable experts per frame, most probable first, ties in expert order (column 0 == the `idx`
    the sites return).  Extension for statistics (BASELINE config 3); the mixture stays dense as in the reference."""
    if not probs.is_cuda:
        raise capi.AvmoeError("topk_experts runs on the GPU (no CPU fallback)")
    p = probs.reshape(-1, probs.shape[-1]).to(torch.float32).contiguous()
    out = torch.empty(p.shape[0], k, dtype=torch.int64, device=p.device)
    st = capi.lib().avmoe_router_topk(p.data_ptr(), C.c_int64(p.shape[0]), C.c_int32(p.shape[1]), C.c_int32(k), out.data_ptr(),
                                      torch.cuda.current_stream(p.device).cuda_stream)
    capi.check(st, "avmoe_router_topk")
    return out


# ---------------------------------------------------------------------------------------------------------------------
def adapter_state_dict(model: torch.nn.Module, substrings=("adapter",)) -> Dict[str, torch.Tensor]:
    """The adapter / router entries of model.state_dict() (keys containing 'adapter': `*_adapter_blocks_p{1,2}.*`)."""
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if any(s in k for s in substrings)}


def save_adapters(model: torch.nn.Module, path: str, substrings=("adapter",)):
    torch.save(adapter_state_dict(model, substrings), path)


def load_adapters(model: torch.nn.Module, path_or_state, strict_adapters: bool = True):
    """Loads a full released checkpoint or an adapter-only file with strict=False (the reference's own call,
    AVE/main_trans_v3.py:254) and reports what did not match.  strict_adapters: every adapter key of the MODEL must have been
    found (raises otherwise) -- backbone / head keys may be missing."""
    sd = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, str) else path_or_state
    res = model.load_state_dict(sd, strict=False)
    if strict_adapters:
        missing = [k for k in res.missing_keys if "adapter" in k]
        if missing:
            raise KeyError(f"checkpoint lacks adapter entries: {missing[:5]}{' ...' if len(missing) > 5 else ''}")
    return res


def strip_prefix(state_dict: Dict[str, torch.Tensor], n: int = 10) -> Dict[str, torch.Tensor]:
    """k[n:] for every key -- how the reference loads the HTS-AT checkpoint whose keys carry a 10-character module prefix
    (net_trans_v3.py:560-563)."""
    return {k[n:]: v for k, v in state_dict.items()}
