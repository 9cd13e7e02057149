"""Data-parallel training of the adapter sites: one process per GPU, batches sharded over clips, and ONE
exchange per optimizer step -- an all-reduce (sum, then / world) of adapter + router gradients only.  The
backbones are frozen, so nothing else is communicated.  `torch.distributed` backend "nccl" is RCCL on ROCm
(xGMI inside a node); "gloo" is used by the CPU tests.

The reference has no distributed training: its multi-GPU mode is `nn.DataParallel` (AVVP/main.py:421,
AVQA/net_grd_avst/main_avst_v2.py:321, AVS/avs_scripts/avs_s4/train_v2.py:140), which reduces gradients to
GPU 0 every step and computes BatchNorm statistics per replica.  Per-rank BatchNorm statistics are therefore
reference semantics and are kept; only the gradient reduction is re-designed:

  * gradients live in a few LARGE flat fp32 buckets (`param.grad` are views into them; default 64 MB): the adapter sites
    handed in as `sites=` take consecutive SLICES of shared buckets, in reverse execution order, and their backward writes
    every parameter gradient straight into its slice (no per-parameter accumulation kernels) -- a Swin-L model's 48 sites
    (345-475 MB of fp32 gradients) travel as ~7 messages, not 48: xGMI is point-to-point, large messages are what
    keeps the seven links of a GPU busy;
  * a bucket's all-reduce is launched asynchronously as soon as the LAST site / parameter of the bucket has reported (behind
    the events of every stream that wrote into it), so communication overlaps the rest of the backward;
  * the mean over the ranks costs no kernel: RCCL reduces with ncclAvg (`ReduceOp.AVG`); with `average="optimizer"` the
    collective is a plain sum and `FlatAdam` folds 1 / world into `avmoe_adam_step`'s `grad_scale` (gloo -- CPU tests and
    development only -- has no AVG: sum, then one division per bucket);
  * on gradient-accumulation micro-steps (`sync=False`) nothing is sent (reference accum_itr semantics,
    AVE/main_trans_v3.py:136-138).
"""
from __future__ import annotations

import weakref
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


ALIGN = 64        # elements: every parameter's slice of a flat bucket starts 256-byte aligned -- FlatAdam re-points param.data to
                  # the same offsets and the GEMM engine wants 16-byte aligned operands (the rule of MoEAdapter.grad_layout)


class _Bucket:
    """One flat fp32 gradient buffer = one all-reduce message: slices of whole adapter sites (written by the sites' backward
    through a _SiteSink) and / or individual parameters (accumulated by autograd, counted by a hook)."""

    def __init__(self):
        self.params: List[torch.nn.Parameter] = []
        self.offsets: List[int] = []
        self.size = 0
        self.sinks: List["_SiteSink"] = []
        self.hooked = 0                                  # parameters that report through the autograd hook
        self.flat = None
        self.pending = 0
        self.work = None
        self.seen = set()                                # hooked parameters that have reported since arm()

    def add_param(self, p):
        self.params.append(p); self.offsets.append(self.size)
        self.size += -(-p.numel() // ALIGN) * ALIGN
        self.hooked += 1

    def add_site(self, site):
        """-> (base offset of the site's slice, its length) ; the slice has the layout of site.grad_layout()"""
        names, offs, total = site.grad_layout(ALIGN)
        ps = dict(site.named_parameters())
        base = self.size
        for k, o in zip(names, offs):
            self.params.append(ps[k]); self.offsets.append(base + o)
        self.size += total
        return base, total

    def materialize(self, device):
        self.flat = torch.zeros(self.size, device=device, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view_as(p)

    def arm(self):
        self.pending = len(self.sinks) + self.hooked
        self.work = None
        self.seen.clear()


class _SiteSink:
    """Hands a MoEAdapter site its slice of a (shared) bucket: the site's backward writes all its parameter gradients there
    (layout = site.grad_layout()) and calls done()."""

    def __init__(self, site, bucket, base, total, reducer):
        self.names, self.offsets, self.total = site.grad_layout(ALIGN)
        assert self.total == total
        self.bucket, self.reducer, self.fresh = bucket, reducer, True
        self.site = site
        self.base = base
        self.flat = None                                 # the slice (set once the bucket's memory exists)
        self.calls = 0                                   # forward calls of the site still waiting for their backward
        self.reported = False                            # counted against the bucket's `pending` since the last begin()
        self.stale = False                               # the slice still holds the PREVIOUS optimizer step's gradient (lazy zero_grad)
        self.event = None                                # recorded on the stream the site's last backward ran on
        self._ev = None

    def bind(self):
        self.flat = self.bucket.flat[self.base:self.base + self.total]

    def matches(self, names, tensors) -> bool:
        return tuple(names) == self.names and all(v.device == self.flat.device for v in tensors.values())

    def done(self):
        self.fresh = False
        self.stale = False
        self.calls -= 1
        if self.calls <= 0:
            if self.flat.is_cuda:                        # the bucket's collective must wait for THIS stream's writes, whichever
                if self._ev is None:                     # stream context launches it (the two sites of an AdapterPair finish on
                    self._ev = torch.cuda.Event()        # two different streams).  One event per sink, re-recorded every step.
                self.event = self._ev
                self.event.record(torch.cuda.current_stream(self.flat.device))
            self.reducer._sink_reported(self)


class AdapterGradReducer:
    """Bucketed, overlapped gradient all-reduce for the trainable parameters of adapter sites.

        red = AdapterGradReducer(model.parameters(), bucket_mb=64, sites=adapter_sites)
        for micro, batch in enumerate(loader):
            red.begin(sync=(micro + 1) % accum == 0)     # arm the buckets for this backward
            loss(model(batch)).backward()
            red.finish()                                   # wait for the buckets (no-op when sync=False)
            if sync: opt.step(); red.zero_grad()
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 64.0,
                 process_group: Optional[dist.ProcessGroup] = None, sites=None, average: str = "auto"):
        """`sites`: MoEAdapter modules, in EXECUTION order, whose backward should write its parameter gradients straight into
        this reducer's buckets (a gradient sink per site; consecutive sites share buckets of up to `bucket_mb`, packed in reverse
        execution order = the order the backward finishes them); their parameters may also be listed in `params`, all other
        parameters are bucketed by size.
        average: "auto" -- the collective itself averages where the backend can (RCCL / NCCL: ReduceOp.AVG), else sum + one
        division per bucket; "optimizer" -- plain sum, `grad_scale` = 1 / world is left to the optimizer (FlatAdam picks it up)."""
        if average not in ("auto", "optimizer"):
            raise ValueError("average must be 'auto' or 'optimizer'")
        params = list(params)
        # a reducer built earlier over the same sites / parameters lets go of them first (its hooks would keep firing otherwise)
        for owner in list(sites or []) + params:
            ref = getattr(owner, "_avmoe_reducer_ref", None)
            old = ref() if ref is not None else None
            if old is not None and old is not self:
                old.close()
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self._avg_op = average == "auto" and backend == "nccl" and self.world > 1      # RCCL: ncclAvg, no extra kernel
        if self._avg_op:
            # probe once (every rank takes the same branch: the probe is itself a collective): a communicator that refuses AVG
            # falls back to sum + one division per bucket instead of failing in the first training step
            try:
                dev = next((p.device for p in params if p.is_cuda), None)
                probe = torch.ones(1, device=dev if dev is not None else "cuda")
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=process_group)
                self._avg_op = bool(abs(float(probe.item()) - 1.0) < 1e-6)
            except Exception:
                self._avg_op = False
        self._divide = average == "auto" and not self._avg_op and self.world > 1
        self.grad_scale = 1.0 / self.world if average == "optimizer" else 1.0
        self.buckets: List[_Bucket] = []
        self.sinks: List[_SiteSink] = []
        cap = max(1, int(bucket_mb * (1 << 20)) // 4)     # elements
        owned = set()
        device = None
        cur = None
        for site in reversed(list(sites or [])):         # the backward finishes the sites in reverse execution order
            sp = list(site.parameters())
            if not sp or not all(p.requires_grad and p.dtype == torch.float32 for p in sp):
                continue                                  # partly frozen site: plain autograd accumulation
            total = site.grad_layout(ALIGN)[2]
            if cur is None or (cur.size and cur.size + total > cap):
                cur = _Bucket()
                self.buckets.append(cur)
            base, total = cur.add_site(site)
            sink = _SiteSink(site, cur, base, total, self)
            cur.sinks.append(sink)
            site._grad_sink = sink
            site._avmoe_reducer_ref = weakref.ref(self)
            self.sinks.append(sink)
            owned.update(id(p) for p in sp)
            device = sp[0].device
        ps = [p for p in params if p.requires_grad and id(p) not in owned]
        if not ps and not self.buckets:
            raise ValueError("no trainable parameters")
        cur = None
        for p in reversed(ps):                           # backward produces gradients roughly in reverse order
            if cur is None or (cur.size and cur.size + p.numel() > cap):
                cur = _Bucket()
                self.buckets.append(cur)
            cur.add_param(p)
            device = p.device
        self._owner = {}
        self._hooks = []
        for b in self.buckets:
            b.materialize(b.params[0].device if b.params else device)
            for s in b.sinks:
                s.bind()
            if b.hooked:                                 # a bucket holds either site slices (their sinks report) or plain parameters
                for p in b.params:
                    self._owner[p] = b
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._hook))
                    p._avmoe_reducer_ref = weakref.ref(self)
        self._sync = True
        self.time_exposed = False                        # measurement aid: event-time what finish() waits for (exposed_ms)
        self._exposed = []
        self.begin(True)

    # ---- launching ------------------------------------------------------------------------------------------------
    def _launch(self, b):
        if b.flat.is_cuda:
            cur = torch.cuda.current_stream(b.flat.device)
            for s in b.sinks:
                if s.event is not None:
                    cur.wait_event(s.event)              # every stream that wrote a slice of this bucket
        op = dist.ReduceOp.AVG if self._avg_op else dist.ReduceOp.SUM
        b.work = dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)

    def _late(self, b):
        # a gradient arriving after the bucket's collective went out would be written under the in-flight all-reduce and
        # never be reduced: fail loudly (every backward of a sync step but the last belongs in begin(sync=False) micro-steps)
        if b.work is not None:
            raise RuntimeError("AdapterGradReducer: a gradient was produced after its bucket's all-reduce had been launched "
                               "(a second forward + backward of the same site inside one begin(sync=True) step: the first backward "
                               "completed the bucket) -- run all but the last forward + backward of a step under begin(sync=False)")

    def _maybe_launch(self, b):
        # a bucket goes out when every sink / hooked parameter has reported ONCE since begin() and no site of it still owes a
        # backward.  A site that runs forward, forward, backward, backward inside one sync step is covered (`calls` counts the forwards
        # still owed); forward, backward, forward, backward is NOT once this site was the last of its bucket to report: the first
        # backward sends the bucket out and the second one raises in _late() -- run all but the last pass of a step under
        # begin(sync=False) (tests/test_dp_gloo.py::test_gradient_after_the_collective_went_out_fails_loudly)
        if b.pending <= 0 and self._sync and self.world > 1 and b.work is None and all(s.calls <= 0 for s in b.sinks):
            self._launch(b)

    def _sink_reported(self, s):
        b = s.bucket
        self._late(b)
        if not s.reported:
            s.reported = True
            b.pending -= 1
        self._maybe_launch(b)

    def _hook(self, p):
        b = self._owner[p]
        # autograd may have replaced .grad (first accumulation into a None grad): keep the bucket view authoritative
        if p.grad.data_ptr() < b.flat.data_ptr() or p.grad.data_ptr() >= b.flat.data_ptr() + b.flat.numel() * 4:
            for q, off in zip(b.params, b.offsets):
                if q is p:
                    b.flat[off:off + p.numel()].view_as(p).copy_(p.grad)
                    p.grad = b.flat[off:off + p.numel()].view_as(p)
                    break
        self._late(b)
        if id(p) not in b.seen:
            b.seen.add(id(p))
            b.pending -= 1
        self._maybe_launch(b)

    def begin(self, sync: bool = True):
        self._sync = sync
        for b in self.buckets:
            b.arm()
        for s in self.sinks:
            s.calls = 0
            s.reported = False
            s.event = None

    def finish(self):
        if self._sync:
            for s in self.sinks:                     # lazy zero_grad: a site that received no gradient this step contributes zeros,
                if s.stale:                          # not what the previous step left in its slice
                    s.flat.zero_()
                    s.stale = False
        if not self._sync or self.world == 1:
            return
        timed = self.time_exposed and self.buckets[0].flat.is_cuda
        if timed:                                    # what the compute stream still has to wait for once the backward is enqueued
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for b in self.buckets:
            if b.work is None:                       # a site / parameter received no gradient this step: reduce anyway
                self._launch(b)
            b.work.wait()
            if self._divide:
                b.flat.div_(self.world)
        if timed:
            e1.record()
            self._exposed.append((e0, e1))

    def exposed_ms(self) -> List[float]:
        """per finish() call since `time_exposed` was switched on: milliseconds the caller's stream spent between the end of the
        enqueued backward and the last bucket's all-reduce (the part of the exchange the backward did not hide).  Synchronises."""
        out = []
        for e0, e1 in self._exposed:
            e1.synchronize()
            out.append(e0.elapsed_time(e1))
        self._exposed = []
        return out

    def zero_grad(self, lazy: bool = False):
        """lazy=False: every bucket is filled with zeros (one fill kernel per bucket).  lazy=True: buckets made of site slices
        only are NOT filled -- the next backward of a site OVERWRITES its whole slice (`fresh`; the alignment padding between the
        parameters is never written by anybody and stays zero), so the fill is a launch and a pass for nothing; until then
        `param.grad` of those sites still shows the previous step's values, and a site that gets no backward in the next sync
        step is zeroed by finish() before the collective goes out.  Needs finish() to be called every step (as documented)."""
        for b in self.buckets:
            if not (lazy and b.sinks and not b.hooked):
                b.flat.zero_()
        for s in self.sinks:
            s.fresh = True                               # the next backward of the site overwrites instead of adding
            s.stale = bool(lazy and not s.bucket.hooked)

    def close(self):
        """Detach from the parameters: removes the autograd hooks and the sites' gradient sinks (a second reducer built over
        the same parameters would otherwise keep firing this one's hooks and launch collectives on its stale buckets)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for s in self.sinks:
            site = getattr(s, "site", None)
            if site is not None and getattr(site, "_grad_sink", None) is s:
                del site._grad_sink
        self.sinks = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def message_bytes(self) -> int:
        return sum(b.flat.numel() * 4 for b in self.buckets)

    def messages(self) -> List[int]:
        """bytes of every all-reduce message of one optimizer step"""
        return [b.flat.numel() * 4 for b in self.buckets]
