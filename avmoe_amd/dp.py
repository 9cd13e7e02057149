"""Data-parallel training of the adapter sites: one process per GPU, batches sharded over clips, and ONE
exchange per optimizer step -- an all-reduce (sum, then / world) of adapter + router gradients only.  The
backbones are frozen, so nothing else is communicated.  `torch.distributed` backend "nccl" is RCCL on ROCm
(xGMI inside a node); "gloo" is used by the CPU tests.

The reference has no distributed training: its multi-GPU mode is `nn.DataParallel` (AVVP/main.py:421,
AVQA/net_grd_avst/main_avst_v2.py:321, AVS/avs_scripts/avs_s4/train_v2.py:140), which reduces gradients to
GPU 0 every step and computes BatchNorm statistics per replica.  Per-rank BatchNorm statistics are therefore
reference semantics and are kept; only the gradient reduction is re-designed:

  * gradients live in a few flat fp32 buckets (`param.grad` are views into them), so the exchange is a
    handful of large all-reduces instead of one small one per tensor -- xGMI is point-to-point, large
    messages are what keeps the links busy;
  * buckets are filled in reverse registration order (the order the backward produces gradients) and each
    bucket's all-reduce is launched asynchronously as soon as its last gradient has been accumulated, so
    communication overlaps the rest of the backward;
  * on gradient-accumulation micro-steps (`sync=False`) nothing is sent (reference accum_itr semantics,
    AVE/main_trans_v3.py:136-138).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


ALIGN = 64        # elements: every parameter's slice of a flat bucket starts 256-byte aligned -- FlatAdam re-points param.data to
                  # the same offsets and the GEMM engine wants 16-byte aligned operands (the rule of MoEAdapter.grad_layout)


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter], device, dtype):
        self.params = params
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += -(-p.numel() // ALIGN) * ALIGN
        self.flat = torch.zeros(off, device=device, dtype=dtype)
        for p, o in zip(params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view_as(p)
        self.pending = len(params)
        self.work = None


class _SiteSink:
    """Hands a MoEAdapter site its slice of gradient memory: the site's backward writes all its parameter gradients there
    (layout = site.grad_layout()) and calls done()."""

    def __init__(self, site, bucket, reducer):
        self.names, self.offsets, self.total = site.grad_layout()
        self.bucket, self.reducer, self.fresh = bucket, reducer, True
        self.flat = bucket.flat
        self.calls = 0                                   # forward calls of the site still waiting for their backward

    def matches(self, names, tensors) -> bool:
        return tuple(names) == self.names and all(v.device == self.flat.device for v in tensors.values())

    def done(self):
        self.fresh = False
        self.calls -= 1
        if self.calls <= 0:
            self.reducer._bucket_filled(self.bucket)


class _SiteBucket:
    def __init__(self, site):
        names, offs, total = site.grad_layout()
        ps = dict(site.named_parameters())
        self.params = [ps[k] for k in names]
        self.offsets = list(offs)
        self.flat = torch.zeros(total, device=self.params[0].device, dtype=torch.float32)
        for p, o in zip(self.params, offs):
            p.grad = self.flat[o:o + p.numel()].view_as(p)
        self.pending = 1
        self.work = None


class AdapterGradReducer:
    """Bucketed, overlapped gradient all-reduce for the trainable parameters of adapter sites.

        red = AdapterGradReducer(model.parameters(), bucket_mb=32)
        for micro, batch in enumerate(loader):
            red.begin(sync=(micro + 1) % accum == 0)     # arm the hooks for this backward
            loss(model(batch)).backward()
            red.finish()                                   # wait for the buckets (no-op when sync=False)
            if sync: opt.step(); red.zero_grad()
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 32.0,
                 process_group: Optional[dist.ProcessGroup] = None, sites=None):
        """`sites`: MoEAdapter modules (GPU) whose backward should write its parameter gradients straight into a bucket
        of this reducer (one bucket per site, no per-parameter accumulation kernels); their parameters may also be
        listed in `params`, all other parameters are bucketed by size."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.buckets: List[_Bucket] = []
        self.sinks: List[_SiteSink] = []
        owned = set()
        for site in (sites or []):
            sp = list(site.parameters())
            if not sp or not all(p.requires_grad and p.dtype == torch.float32 for p in sp):
                continue                                  # partly frozen site: plain autograd accumulation
            b = _SiteBucket(site)
            self.buckets.append(b)
            sink = _SiteSink(site, b, self)
            site._grad_sink = sink
            self.sinks.append(sink)
            owned.update(id(p) for p in sp)
        ps = [p for p in params if p.requires_grad and id(p) not in owned]
        if not ps and not self.buckets:
            raise ValueError("no trainable parameters")
        cap = int(bucket_mb * (1 << 20))
        cur, cur_bytes = [], 0
        for p in reversed(ps):                       # backward produces gradients roughly in reverse order
            if cur and cur_bytes + p.numel() * 4 > cap:
                self.buckets.append(_Bucket(cur, p.device, torch.float32))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += p.numel() * 4
        if cur:
            self.buckets.append(_Bucket(cur, ps[0].device, torch.float32))
        self._sync = True
        self._owner = {}
        for b in self.buckets:
            if isinstance(b, _SiteBucket):
                continue
            for p in b.params:
                self._owner[p] = b
                p.register_post_accumulate_grad_hook(self._hook)

    def _bucket_filled(self, b):
        b.pending = 0
        if self._sync and self.world > 1:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _hook(self, p):
        b = self._owner[p]
        # autograd may have replaced .grad (first accumulation into a None grad): keep the bucket view authoritative
        if p.grad.data_ptr() < b.flat.data_ptr() or p.grad.data_ptr() >= b.flat.data_ptr() + b.flat.numel() * 4:
            for q, off in zip(b.params, b.offsets):
                if q is p:
                    b.flat[off:off + p.numel()].view_as(p).copy_(p.grad)
                    p.grad = b.flat[off:off + p.numel()].view_as(p)
                    break
        b.pending -= 1
        if b.pending == 0 and self._sync and self.world > 1:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def begin(self, sync: bool = True):
        self._sync = sync
        for b in self.buckets:
            b.pending = 1 if isinstance(b, _SiteBucket) else len(b.params)
            b.work = None
        for s in self.sinks:
            s.calls = 0

    def finish(self):
        if not self._sync or self.world == 1:
            return
        for b in self.buckets:
            if b.work is None:                       # a parameter received no gradient this step: reduce anyway
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            b.work.wait()
            b.flat.div_(self.world)

    def zero_grad(self):
        for b in self.buckets:
            b.flat.zero_()
        for s in self.sinks:
            s.fresh = True                               # the next backward of the site overwrites instead of adding

    def message_bytes(self) -> int:
        return sum(b.flat.numel() * 4 for b in self.buckets)
