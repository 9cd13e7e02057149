"""Host-side mirror of the reference's adapter modules, running on the HIP path.

`MoEAdapter` / `ExpertAdapter` keep the reference's constructor arguments, `forward` signatures, return
tuples and `state_dict` key names for each of its five task copies, so the task models (AVE / AVQA /
AVVP / AVS) can import them instead of their own classes and load released checkpoints unchanged:

    AVE   AVMOE/AVE/nets/net_trans_v3.py:296-487           -> avmoe_amd.adapters.MoEAdapter (= ave)
    AVQA  AVMOE/AVQA/net_grd_avst/net_avst_v2.py:215-399   -> avmoe_amd.adapters.MoEAdapterAVQA
    AVVP  AVMOE/AVVP/nets/mgn.py:39-224                    -> avmoe_amd.adapters.MoEAdapterAVVP
    AVS   AVMOE/AVS/avs_scripts/avs_{s4,ms3}/model/PVT_AVSModel_v2.py:90-318 -> avmoe_amd.adapters.MoEAdapterAVS

The submodules (`conv_adapter`, `fc`, `router`, `multimodal_experts.{j}.bn1`, ...) are real torch modules
used ONLY as parameter containers (same names, shapes and default initialisation as the reference); their
own `forward` is never called.  All arithmetic happens in avmoe_amd/lib/libavmoe_hip.so through
`AdapterFunction` (a `torch.autograd.Function` over the C ABI).  There is no eager / CPU fallback: a
missing library or a non-GPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import os
import torch
import torch.nn as nn

from . import _capi as capi
from . import _capi_moe as cm

_SCRATCH: Dict[tuple, torch.Tensor] = {}
_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def side_stream(dev: torch.device) -> "torch.cuda.Stream":
    """THE side stream of a device: every AdapterPair shares it (pairs run one after another, so one second stream -- and one
    second scratch workspace -- is all the concurrency there is to have)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


def release_workspaces():
    """Drops the cached transient workspaces (one per device and stream) -- e.g. after the largest site shape of a run has
    changed.  Synchronises first: kernels still in flight may be using them."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    _SCRATCH.clear()


def _scratch(dev: torch.device, nbytes: int, slot: int = 0, stream: int = None) -> torch.Tensor:
    """Transient workspace per (device, stream), grown on demand and reused (its users are ordered by that stream).
    slot: a second workspace on the same stream (AdapterPair.same_stream interleaves the sections of two sites on ONE stream).
    stream: the handle of the stream the caller is on, when it has it at hand (a torch.cuda.current_stream() lookup costs ~3 us; at the
    reference's batch of 2 clips a pair-step made 20 of them)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(),
           stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream, slot)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _SCRATCH[key] = buf
    return buf


_FUSED_DX = os.environ.get("AVMOE_NO_FUSED_DX") is None      # development A/B: the token gradients as two kernels each (dX overwrites, the other site's dY adds)
_warned_side_stream = False
_SHARED_GPU = None          # process-wide override of avmoe_moe_desc.shared_gpu: None = decide per call (below), True / False = always / never


def set_shared_gpu(mode):
    """avmoe_moe_desc.shared_gpu for every site call of this process: True = always ask for the protected launches (another stream's
    kernels may be on the GPU: a second adapter site, a backbone GEMM), False = never (the caller asserts that nothing else runs
    on the GPU while adapter calls do), None (default) = per call: protected whenever the call is issued on a stream other than the
    device's default stream, and always in AdapterPair's two-stream mode.  A module attribute `shared_gpu` (True / False) overrides
    this for one site.  include/avmoe.h says which kernel families the flag protects and what it costs."""
    global _SHARED_GPU
    if mode not in (None, True, False):
        raise ValueError("set_shared_gpu: None, True or False")
    _SHARED_GPU = mode


def _shared_gpu_of(module, dev, asked):
    """The shared_gpu flag of one call: AdapterPair's explicit request, else the site's attribute, else the process-wide mode,
    else 'is this call on a side stream?' (a caller that left the default stream did so to overlap something)."""
    if asked:
        return True
    own = module.__dict__.get("shared_gpu", None)
    if own is not None:
        return bool(own)
    if _SHARED_GPU is not None:
        return _SHARED_GPU
    side = torch.cuda.current_stream(dev).cuda_stream != torch.cuda.default_stream(dev).cuda_stream
    global _warned_side_stream
    if side and not _warned_side_stream:      # said once: the protected launches change the BatchNorm summation order of some kernel families and cost residency
        _warned_side_stream = True
        import warnings
        warnings.warn("avmoe_amd: adapter call on a non-default stream -> avmoe_moe_desc.shared_gpu = 1 (CU-exclusive launches of the generalised "
                      "bottleneck-space kernels, include/avmoe.h); avmoe_amd.adapters.set_shared_gpu(False) if nothing else runs on the GPU", stacklevel=3)
    return side


def _site_forward(module, X, Y, noise, names, params, add_to=None, shared_gpu=None, stream=None):
    """One avmoe_moe_forward call.  Returns (out, probs, idx, lb, state) with state = what the backward needs.
    add_to: a contiguous tensor like X that receives `+= adapter(X, Y)` in place (avmoe_moe_desc.accumulate_out) and is
    returned as `out`.  shared_gpu: True = another stream's kernels may run beside this call (AdapterPair); None = _shared_gpu_of.
    stream: the handle (int) of the CURRENT stream when the caller already has it (AdapterPair), else looked up."""
    if not (X.is_cuda and Y.is_cuda):
        raise capi.AvmoeError("avmoe_amd runs on MI355X only: tensors must live on a GPU (no CPU fallback)")
    if stream is None:
        stream = torch.cuda.current_stream(X.device).cuda_stream
    if X.dtype != Y.dtype or X.dtype not in (torch.float32, torch.bfloat16):
        raise capi.AvmoeError(f"activations must both be float32 or bfloat16, got {X.dtype} / {Y.dtype}")
    L = capi.lib()
    X = X.contiguous()
    Y = Y.contiguous()
    S, N, Cc = X.shape
    desc = module._desc(S, N, Y.shape[1], X.dtype == torch.bfloat16)
    shared_gpu = _shared_gpu_of(module, X.device, shared_gpu)
    desc.shared_gpu = int(shared_gpu)                     # (include/avmoe.h: kernels of another stream may be on the GPU during this call and its backward)
    keep = module._attention_keep(S, N, X.device)
    ptrs = module._fill_ptrs(params, keep, names=names, device=X.device)
    wkey = (S, N, Y.shape[1], X.dtype, module.training, bool(shared_gpu))
    sizes = module.__dict__.setdefault("_ws_sizes", {}).get(wkey)      # workspace sizes of this call shape (two plan evaluations otherwise)
    if sizes is None:
        sizes = (L.avmoe_moe_saved_bytes(C.byref(desc)), L.avmoe_moe_scratch_bytes(C.byref(desc)))
        if sizes[0] == 0:
            raise capi.AvmoeError(L.avmoe_last_error().decode())
        module.__dict__["_ws_sizes"][wkey] = sizes
    saved = torch.empty(sizes[0], dtype=torch.uint8, device=X.device)
    scratch = _scratch(X.device, sizes[1], stream=stream)
    if add_to is not None:
        if add_to.shape != X.shape or add_to.dtype != X.dtype or not add_to.is_contiguous() or add_to.device != X.device:
            raise capi.AvmoeError("add_to must be a contiguous tensor with the shape, dtype and device of the tokens")
        out, desc.accumulate_out = add_to, 1
    else:
        out = torch.empty_like(X)
    E = module.num_multimodal_experts + module.num_singlemodal_experts
    probs = torch.empty(S, E, device=X.device, dtype=torch.float32)
    idx = torch.empty(S, device=X.device, dtype=torch.int64)
    lb = torch.empty((), device=X.device, dtype=torch.float32)      # always written by the router's kernels (0 without the LB loss)
    if noise is not None:
        noise = noise.to(torch.float32).contiguous()
    st = L.avmoe_moe_forward(C.byref(desc), X.data_ptr(), Y.data_ptr(), C.byref(ptrs),
                             noise.data_ptr() if noise is not None else None, out.data_ptr(), probs.data_ptr(),
                             idx.data_ptr(), lb.data_ptr(), saved.data_ptr(), scratch.data_ptr(), stream)
    desc.accumulate_out = 0
    capi.check(st, "avmoe_moe_forward")
    if module.__dict__.get("_keep_saved"):                # avmoe_amd.debug.keep_saved: checker-side view of the last call's workspace
        module.__dict__["_last_saved"] = (desc, saved)
    return out, probs, idx, lb, ((desc, keep, ptrs), saved, X, Y)


class _SiteBackward:
    """One site's backward through the C ABI, writing (or, with acc_*, adding) the token gradients into dX / dY.
    run(0) = the whole of it; run(mask) = the sections of avmoe_moe_backward_part (1: touches neither dX nor dY, 2: every
    writer of dX, 4: every writer of dY), in this order.  finish() returns the parameter gradients in `names` order (None where not needed,
    or for all of them when a gradient sink took them)."""

    def __init__(self, module, state, names, params, needs, d_out, d_lb, dX, dY, acc_dx=False, acc_dy=False, scratch_slot=0, stream=None):
        self.L = capi.lib()
        (desc, keep, fwd_ptrs), self.saved, self.X, self.Y = state
        self.desc, self.names, self.module = desc, names, module
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.X.device).cuda_stream      # (the stream this object is built AND run on)
        tensors = dict(zip(names, params))
        # the forward's pointer struct, as long as it was built from these very parameters (autograd's saved-tensor version checks -- or, with an
        # anchor parameter, the pair's own stamp -- refuse a backward on anything else); "v1" sites rebuild (their struct carries dropout draws)
        self.ptrs = fwd_ptrs if (fwd_ptrs is not None and not keep) else module._fill_ptrs(params, keep)
        # Parameter gradients.  With a gradient sink attached (avmoe_amd.dp.AdapterGradReducer(sites=...)) the kernels
        # write straight into the reducer's flat bucket -- `param.grad` are views of it -- and autograd gets None: no
        # per-parameter accumulation kernels.  Otherwise: fresh tensors, accumulated by autograd as usual.
        sink = self.sink = getattr(module, "_grad_sink", None)
        self.use_sink = sink is not None and all(needs) and sink.matches(names, tensors)
        filler = module._site_cache()["fill_p"]
        if self.use_sink:
            # zeros, not empty: on accumulation micro-steps the alignment padding is added to the bucket as well
            self.flat = sink.flat if sink.fresh else torch.zeros_like(sink.flat)
            self.grads = None                            # the kernels write at flat + offset: no per-parameter views needed
            hit = module.__dict__.get("_gptrs_cache")     # the struct of the last call, as long as the sink's memory is the same
            if hit is not None and hit[0] == (self.flat.data_ptr(), id(sink)):
                self.gptrs = hit[1]
            else:
                self.gptrs = filler.fill(cm.MoePtrs(), None, base_ptr=self.flat.data_ptr(), offsets=sink.offsets)
                module.__dict__["_gptrs_cache"] = ((self.flat.data_ptr(), id(sink)), self.gptrs)
        else:
            self.gptrs = cm.MoePtrs()
            self.grads = {k: (torch.empty_like(v) if needs[i] else None) for i, (k, v) in enumerate(tensors.items())}
            filler.fill(self.gptrs, [self.grads[k] for k in names])
        self.d_out = d_out.to(self.X.dtype).contiguous()
        self.lbg = d_lb.to(torch.float32).reshape(1).contiguous() if (d_lb is not None and desc.lb_loss) else None
        wkey = (desc.S, desc.N, desc.M, self.X.dtype, bool(desc.training), bool(desc.shared_gpu))
        sizes = module.__dict__.get("_ws_sizes", {}).get(wkey)
        self.scratch = _scratch(self.X.device, sizes[1] if sizes else self.L.avmoe_moe_scratch_bytes(C.byref(desc)), scratch_slot, stream=self.stream)
        self.dX, self.dY, self.acc = dX, dY, (int(acc_dx), int(acc_dy))

    def run(self, parts=0):
        d = self.desc
        d.accumulate_dx, d.accumulate_dy = self.acc
        st = self.L.avmoe_moe_backward_part(C.byref(d), self.X.data_ptr(), self.Y.data_ptr(), C.byref(self.ptrs), self.d_out.data_ptr(),
                                            self.lbg.data_ptr() if self.lbg is not None else None, self.saved.data_ptr(),
                                            self.scratch.data_ptr(), self.dX.data_ptr(), self.dY.data_ptr(), C.byref(self.gptrs),
                                            int(parts), self.stream)
        d.accumulate_dx = d.accumulate_dy = 0
        capi.check(st, "avmoe_moe_backward")
        return self

    def fused_ok(self, other):
        """Does the library serve `token gradient = this site's dX + the other site's dY` as ONE kernel (avmoe_moe_backward_dx_dy)?
        (asked once per pair of call shapes: the answer is a function of the two descriptors)"""
        da, db = self.desc, other.desc
        key = (id(other.module), da.S, da.N, da.M, da.dtype, da.training, db.S, db.N, db.M, db.dtype, db.training)
        cache = self.module.__dict__.setdefault("_fused_ok", {})
        ok = cache.get(key)
        if ok is None:
            # (a library older than ABI 10 loaded through AVMOE_LIB lacks the entry point: the two-kernel hand-over then)
            ok = cache[key] = hasattr(self.L, "avmoe_moe_backward_dx_dy") and \
                self.L.avmoe_moe_backward_dx_dy(C.byref(da), None, None, None, C.byref(db), None, None, None, None) == 0
        return ok

    def run_fused(self, other):
        """dX (this site) + dY (the other site) written once into self.dX: after sections 1 + 32 + 8 of both sites, in place of this
        site's section 64 and the other site's section 16; the current stream must be ordered behind the other site's section 8."""
        st = self.L.avmoe_moe_backward_dx_dy(C.byref(self.desc), self.X.data_ptr(), self.saved.data_ptr(), self.scratch.data_ptr(),
                                             C.byref(other.desc), other.saved.data_ptr(), other.scratch.data_ptr(), self.dX.data_ptr(), self.stream)
        capi.check(st, "avmoe_moe_backward_dx_dy")
        return self

    def finish_needs_stream(self):
        """does finish() issue a torch op (which runs on the CURRENT stream)?"""
        return self.sink is not None and ((self.use_sink and not self.sink.fresh) or (not self.use_sink and self.sink.stale))

    def finish(self):
        sink = self.sink
        if self.use_sink:
            if not sink.fresh:
                sink.flat.add_(self.flat)                # accumulation micro-step: one fused add for the whole site
            sink.done()
            return (None,) * len(self.names)
        if sink is not None:
            # a sink is attached but this backward did not go through it (a parameter frozen after the reducer was built, a layout
            # mismatch): autograd ADDS the tensors returned below into param.grad -- views of the sink's slice.  After a lazy zero_grad
            # the slice still holds the previous step's values: zero it first, and take it out of the `stale` state so that the
            # reducer's finish() does not wipe what autograd is about to add
            if sink.stale:
                sink.flat.zero_()
            sink.stale = False
            sink.fresh = False
            sink.calls -= 1
        return tuple(self.grads[k] for k in self.names)


def _site_backward(module, state, names, params, needs, d_out, d_lb, dX, dY, acc_dx=False, acc_dy=False):
    """The whole backward of one site in one call (see _SiteBackward)."""
    return _SiteBackward(module, state, names, params, needs, d_out, d_lb, dX, dY, acc_dx, acc_dy).run(0).finish()


class AdapterFunction(torch.autograd.Function):
    """out, probs, lb = f(X, Y, noise, *params) on token-major X:(S,N,C), Y:(S,M,Cy)."""

    @staticmethod
    def forward(ctx, module, X, Y, noise, names, *params):
        out, probs, idx, lb, state = _site_forward(module, X, Y, noise, names, params)
        ctx.set_materialize_grads(False)                 # no zero tensors (= fill kernels) for the gradients of probs / idx / lb nobody sent
        ctx.module, ctx.names, ctx.state = module, names, state[:2]
        sink = getattr(module, "_grad_sink", None)
        if sink is not None and any(ctx.needs_input_grad[5:]):
            sink.calls += 1                              # the site's bucket is complete after as many backward calls
        ctx.save_for_backward(state[2], state[3], *params)
        ctx.mark_non_differentiable(probs, idx)
        return out, probs, idx, lb

    @staticmethod
    def backward(ctx, d_out, _d_probs, _d_idx, d_lb):
        X, Y, *params = ctx.saved_tensors
        if d_out is None:                                # only the load-balancing loss was differentiated
            d_out = torch.zeros_like(X)
        dX, dY = torch.empty_like(X), torch.empty_like(Y)
        pg = _site_backward(ctx.module, (*ctx.state, X, Y), ctx.names, params, ctx.needs_input_grad[5:], d_out, d_lb, dX, dY)
        return (None, dX, dY, None, None) + pg


class _PairFunction(torch.autograd.Function):
    """Both adapter sites of one backbone layer (net_trans_v3.py:695-698): site A adapts the tokens Xa with Xb as the other
    modality, site B adapts Xb with Xa.  Each token tensor therefore receives two gradients (as X of one site and as Y of
    the other); the second one is ADDED inside the GEMM epilogues (avmoe_moe_desc.accumulate_*), so no separate
    accumulation pass over the token gradients runs."""

    @staticmethod
    def forward(ctx, site_a, site_b, side, Xa, Xb, base_a, base_b, names_a, names_b, noises, *params):
        """side: the second HIP stream (two-stream mode) or None (both sites on the caller's stream).
        base_a / base_b: None, or the residual streams that take `+= adapter output` in place (returned as the outputs).
        noises: (noise_a, noise_b) -- the AVS logit noise of the two sites, (S, E) already scaled, or None each.
        params: every parameter of the two sites -- or, with gradient sinks on both (`lean`, AdapterPair.forward), ONE anchor
        parameter: the sinks take the gradients, autograd only has to know that the outputs depend on something trainable."""
        na = len(names_a)
        ctx.lean = len(params) == 1 and na > 1
        if ctx.lean:
            pa, pb = tuple(site_a._param_tensors().values()), tuple(site_b._param_tensors().values())
            # autograd's saved-tensor checks see the anchor only: remember which memory and which version every other parameter
            # had, the backward re-fetches them by name and refuses to run on anything else
            ctx.pstamp = tuple((t.data_ptr(), t._version) for t in pa + pb)
        else:
            pa, pb = params[:na], params[na:]
        Xa, Xb = Xa.contiguous(), Xb.contiguous()
        main = torch.cuda.current_stream(Xa.device)
        side, ctx.events = (side[0], side[1:]) if isinstance(side, tuple) else (side, None)
        # fork / join through the pair's OWN events, made once and re-recorded every step (Stream.wait_stream makes and destroys an
        # event per call, and destroying an event that is still pending blocks the host until the GPU reaches it: the host then never
        # runs ahead of the GPU across a step boundary -- measured: 167 us of GPU idle time per cfg-2 step)
        def fork(ev):
            ev.record(main); side.wait_event(ev)
        def join(ev):
            ev.record(side); main.wait_event(ev)
        if side is not None:                           # site B on the side stream, concurrently with site A
            ctx_ev = ctx.events if ctx.events else tuple(torch.cuda.Event() for _ in range(6))
            fork(ctx_ev[2])
            two = side.cuda_stream != main.cuda_stream      # (same_stream: the schedule of the two-stream mode on one stream -- nothing overlaps)
            with torch.cuda.stream(side):
                out_b, pr_b, idx_b, lb_b, st_b = _site_forward(site_b, Xb, Xa, noises[1], names_b, pb, add_to=base_b, shared_gpu=two, stream=side.cuda_stream)
            out_a, pr_a, idx_a, lb_a, st_a = _site_forward(site_a, Xa, Xb, noises[0], names_a, pa, add_to=base_a, shared_gpu=two, stream=main.cuda_stream)
            join(ctx_ev[3])
            for t_ in (out_b, idx_b, pr_b, lb_b, st_b[1]):
                t_.record_stream(main)
        else:
            out_a, pr_a, idx_a, lb_a, st_a = _site_forward(site_a, Xa, Xb, noises[0], names_a, pa, add_to=base_a, stream=main.cuda_stream)
            out_b, pr_b, idx_b, lb_b, st_b = _site_forward(site_b, st_a[3], st_a[2], noises[1], names_b, pb, add_to=base_b, stream=main.cuda_stream)
        dirty = [t for t in (base_a, base_b) if t is not None]
        if dirty:
            ctx.mark_dirty(*dirty)
        ctx.has_base = (base_a is not None, base_b is not None)
        ctx.set_materialize_grads(False)                 # no zero tensors (= fill kernels) for the gradients of the index outputs
        ctx.side = side
        ctx.sites, ctx.names, ctx.states = (site_a, site_b), (names_a, names_b), (st_a[:2], st_b[:2])
        needs_ab = ((True,), (True,)) if ctx.lean else (ctx.needs_input_grad[10:10 + na], ctx.needs_input_grad[10 + na:])
        for site, needs in ((site_a, needs_ab[0]), (site_b, needs_ab[1])):
            sink = getattr(site, "_grad_sink", None)
            if sink is not None and any(needs) and (not ctx.lean or ctx.needs_input_grad[10]):
                sink.calls += 1
        ctx.save_for_backward(st_a[2], st_a[3], *params)
        ctx.mark_non_differentiable(idx_a, idx_b, pr_a, pr_b)
        return out_a, out_b, idx_a, idx_b, pr_a, pr_b, lb_a, lb_b

    @staticmethod
    def backward(ctx, d_a, d_b, _ia, _ib, _pa, _pb, d_lba, d_lbb):
        Xa, Xb, *params = ctx.saved_tensors
        d_a = torch.zeros_like(Xa) if d_a is None else d_a
        d_b = torch.zeros_like(Xb) if d_b is None else d_b
        (site_a, site_b), (names_a, names_b) = ctx.sites, ctx.names
        na = len(names_a)
        gXa, gXb = torch.empty_like(Xa), torch.empty_like(Xb)
        if ctx.lean:                                       # the sinks take every parameter gradient; autograd gets None for the anchor
            if getattr(site_a, "_grad_sink", None) is None or getattr(site_b, "_grad_sink", None) is None:
                raise capi.AvmoeError("AdapterPair: the gradient sink of a site was detached between its forward and its backward")
            params = tuple(site_a._param_tensors().values()) + tuple(site_b._param_tensors().values())
            if tuple((t.data_ptr(), t._version) for t in params) != ctx.pstamp:
                raise capi.AvmoeError("AdapterPair: a parameter of the pair was replaced or modified in place between the forward and "
                                      "its backward (optimizer step / load_state_dict between micro-steps, or a retained graph)")
            needs_a, needs_b = (True,) * na, (True,) * len(names_b)
        else:
            needs_a, needs_b = ctx.needs_input_grad[10:10 + na], ctx.needs_input_grad[10 + na:]
        gba = d_a if ctx.has_base[0] else None            # out = base + adapter(...): the residual stream passes the gradient on
        gbb = d_b if ctx.has_base[1] else None
        if ctx.side is not None:
            # Two streams, cross-ordered hand-over: ONE buffer per token tensor, no add kernel.  Each site OVERWRITES its own token
            # gradient with its dX (section 2) and ADDS its dY to the other tensor's buffer (section 16) once the other site's dX is
            # there -- both sites run sections 1, 2 and 8 (everything but the GEMM that writes dY) without waiting for each other; only
            # the two dY GEMMs are ordered behind the events.  (Measured on MI355X at cfg-2, round 2: 5.79 ms per step against 5.97 for
            # two buffers + a fused add and 6.04 / 6.06 for the variants that serialise one site's tail behind the other -- dropped.)
            side, main = ctx.side, torch.cuda.current_stream(Xa.device)
            ev_a, ev_b, ev_fork, ev_join, ev_a6, ev_b6 = ctx.events if ctx.events else tuple(torch.cuda.Event() for _ in range(6))
            ev_fork.record(main); side.wait_event(ev_fork)
            slot_b = 1 if side.cuda_stream == main.cuda_stream else 0      # (same_stream: the two sites' sections interleave on ONE stream)
            with torch.cuda.stream(side):
                cbk = _SiteBackward(site_b, (*ctx.states[1], Xb, Xa), names_b, params[na:], needs_b, d_b, d_lbb, gXb, gXa, acc_dx=False, acc_dy=True,
                                    scratch_slot=slot_b, stream=side.cuda_stream)
            cak = _SiteBackward(site_a, (*ctx.states[0], Xa, Xb), names_a, params[:na], needs_a, d_a, d_lba, gXa, gXb, acc_dx=False, acc_dy=True,
                                stream=main.cuda_stream)
            # Round 5: where the library serves it (avmoe_moe_backward_dx_dy: the tuned bf16 shapes), a token gradient is written ONCE -- the
            # dX product of its own site with the other site's dY product folded in as two more contraction segments -- instead of
            # overwritten by one site and read back + added by the other.  That product moves to the END of its site's backward (it
            # needs the other site's hop-1 chain, section 8); decided per tensor.
            fuse_a = _FUSED_DX and cak.fused_ok(cbk)     # gXa = A's dX + B's dY
            fuse_b = _FUSED_DX and cbk.fused_ok(cak)     # gXb = B's dX + A's dY
            # (the backward objects carry their stream's handle: their C calls need no stream context around them)
            cbk.run(1 | 32 if fuse_b else 3)
            if not fuse_b:
                ev_b.record(side)                        # gXb holds site B's dX
            cbk.run(8)                                   # the hop-1 chain up to (not including) the GEMM that writes dY
            ev_b6.record(side)
            cak.run(1 | 32 if fuse_a else 3)
            if not fuse_a:
                ev_a.record(main)                        # gXa holds site A's dX
            cak.run(8)
            ev_a6.record(main)
            if fuse_b:
                side.wait_event(ev_a6)
                cbk.run_fused(cak)                       # gXb complete
            if not fuse_a:
                side.wait_event(ev_a)                    # gXa holds site A's dX
                cbk.run(16)                              # += site B's dY
            if cbk.finish_needs_stream():
                with torch.cuda.stream(side):            # (an accumulation micro-step adds its bucket with a torch op: on the site's stream)
                    pgb = cbk.finish()
            else:
                pgb = cbk.finish()
            if fuse_a:
                main.wait_event(ev_b6)
                cak.run_fused(cbk)                       # gXa complete
            if not fuse_b:
                main.wait_event(ev_b)                    # gXb holds site B's dX
                cak.run(16)
            pga = cak.finish()
            ev_join.record(side); main.wait_event(ev_join)
            for t_ in tuple(g_ for g_ in pgb if g_ is not None) + (cbk.d_out,):
                t_.record_stream(main)
            gXa.record_stream(side); gXb.record_stream(side)
            return (None, None, None, gXa, gXb, gba, gbb, None, None, None) + ((None,) if ctx.lean else pga + pgb)
        # One stream: both sites add into both token gradients; the second one to run re-reads them in its GEMM epilogues.  The
        # larger tensor is re-read more cheaply by the dX kernel (fewer stationary fragments per wave), so the site whose X is the
        # larger tensor runs second
        first_b = Xa.numel() >= Xb.numel()
        def run_a(acc):
            return _site_backward(site_a, (*ctx.states[0], Xa, Xb), names_a, params[:na], needs_a, d_a,
                                  d_lba, gXa, gXb, acc_dx=acc, acc_dy=acc)             # dX -> gXa, dY -> gXb
        def run_b(acc):
            return _site_backward(site_b, (*ctx.states[1], Xb, Xa), names_b, params[na:], needs_b, d_b,
                                  d_lbb, gXb, gXa, acc_dx=acc, acc_dy=acc)             # dX -> gXb, dY -> gXa
        if first_b:
            pgb = run_b(False); pga = run_a(True)
        else:
            pga = run_a(False); pgb = run_b(True)
        return (None, None, None, gXa, gXb, gba, gbb, None, None, None) + ((None,) if ctx.lean else pga + pgb)


class ExpertAdapter(nn.Module):
    """Parameter container with the reference's ExpertAdapter state (net_trans_v3.py:296-374).  It only
    runs as part of a MoEAdapter (the experts of one site are evaluated together on the GPU)."""

    def __init__(self, input_dim, output_dim, adapter_kind, reduction_factor=16, opt=None, use_bn=True,
                 use_gate=True, num_tk=87, is_multimodal=True, variant="ave"):
        super().__init__()
        if adapter_kind != "bottleneck":
            # "basic" exists in the reference (net_trans_v3.py:365-371) but is never instantiated
            raise NotImplementedError(f"adapter_kind={adapter_kind!r}")
        self.adapter_kind, self.use_bn, self.is_multimodal, self.opt, self.num_tk = \
            adapter_kind, bool(use_bn), is_multimodal, opt, num_tk
        g = opt.num_conv_group
        self.gate = nn.Parameter(torch.zeros(1)) if use_gate else None
        self.down_sample_size = input_dim // reduction_factor
        if is_multimodal:
            self.my_tokens = nn.Parameter(torch.rand((num_tk, input_dim)))
            self.gate_av = nn.Parameter(torch.zeros(1))
        elif variant == "avvp":
            self.gate_av = nn.Parameter(torch.zeros(1))                       # mgn.py:83
        elif variant == "avs" and getattr(opt, "is_self_attention", 0) and \
                getattr(opt, "self_attention_version", "v1") == "v2":         # PVT_AVSModel_v2.py:143-145
            self.my_tokens = nn.Parameter(torch.rand((num_tk, input_dim)))
            self.gate_self = nn.Parameter(torch.zeros(1))
        elif getattr(opt, "is_self_attention", 0):
            # AVS self_attention_version "v1" (its default, PVT_AVSModel_v2.py:141-142) and the AVE / AVQA is_self_attention
            # switch (net_trans_v3.py:342-343): the unimodal expert's input is replaced by MultiheadAttention(x, x, x) over the
            # FRAMES.  Parameter container only (in_proj_weight, in_proj_bias, out_proj.*); csrc/mha_frames.hip does the work.
            self.num_head, self.head_dropout = 4, 0.2
            self.self_attention = nn.MultiheadAttention(input_dim, num_heads=self.num_head, dropout=self.head_dropout)
        self.down_sampler = nn.Conv2d(input_dim, self.down_sample_size, 1, groups=g, bias=False)
        self.up_sampler = nn.Conv2d(self.down_sample_size, output_dim, 1, groups=g, bias=False)
        if use_bn:
            self.bn1 = nn.BatchNorm2d(self.down_sample_size)
            self.bn2 = nn.BatchNorm2d(output_dim)
        if opt.is_before_layernorm:
            self.ln_before = nn.LayerNorm(output_dim)
        if opt.is_post_layernorm:
            self.ln_post = nn.LayerNorm(output_dim)

    def forward(self, x, vis_token=None):
        raise capi.AvmoeError("ExpertAdapter runs inside MoEAdapter.forward on the HIP path; it has no standalone forward")


class MoEAdapter(nn.Module):
    """AVE signature (net_trans_v3.py:438-487).  forward(x, vis_token) -> (out (S,C,N,1), expert_indices (S,1))."""

    variant = "ave"

    def __init__(self, input_dim, output_dim, adapter_kind, dim_list, layer_idx, reduction_factor=16, opt=None,
                 use_bn=True, use_gate=True, num_tk=87, conv_dim_in=0, conv_dim_out=0, linear_in=0, linear_out=0):
        super().__init__()
        if input_dim != output_dim or linear_out != input_dim:
            raise ValueError("the adapter sites have input_dim == output_dim == linear_out (net_trans_v3.py:599-637)")
        self.opt, self.use_bn, self.use_gate, self.num_tk = opt, bool(use_bn), bool(use_gate), num_tk
        self.input_dim, self.reduction_factor = input_dim, reduction_factor
        self.conv_adapter = nn.Conv2d(conv_dim_in, conv_dim_out, kernel_size=1)
        self.fc = nn.Linear(linear_in, linear_out)
        self.num_multimodal_experts = opt.num_multimodal_experts
        self.num_singlemodal_experts = opt.num_singlemodal_experts
        mk = lambda mm: ExpertAdapter(input_dim, output_dim, adapter_kind, reduction_factor, opt, use_bn, use_gate,
                                      num_tk, is_multimodal=mm, variant=self.variant)
        self.multimodal_experts = nn.ModuleList([mk(True) for _ in range(self.num_multimodal_experts)])
        self.singlemodal_experts = nn.ModuleList([mk(False) for _ in range(self.num_singlemodal_experts)])
        E = self.num_multimodal_experts + self.num_singlemodal_experts
        self.router = nn.Sequential(nn.Linear(input_dim + linear_out, 128), nn.ReLU(), nn.Linear(128, 32), nn.ReLU(),
                                    nn.Linear(32, E))

    # ---- plumbing ---------------------------------------------------------------------------------
    def _self_attn(self):
        if self.variant == "avvp":
            return "nxn"
        if self.variant == "avs" and getattr(self.opt, "is_self_attention", 0) and \
                getattr(self.opt, "self_attention_version", "v1") == "v2":
            return "v2"
        if getattr(self.opt, "is_self_attention", 0) and self.num_singlemodal_experts > 0:
            return "v1"
        return "none"

    def _attention_keep(self, S, N, device):
        """{key: (N * heads, S, S) f32} dropout multipliers (0 or 1 / (1 - p)) of the "v1" experts' attention weights for one
        training-mode call -- the draw nn.MultiheadAttention makes with the global RNG (one per unimodal expert, in expert
        order); {} in eval mode.  `self.attention_keep` (same keys) overrides the draw: tests replay a recorded one."""
        if self._self_attn() != "v1" or not self.training:
            return {}
        forced = getattr(self, "attention_keep", None)
        out = {}
        for j, ex in enumerate(self.singlemodal_experts):
            key = f"singlemodal_experts.{j}.{cm.SA_KEEP}"
            p = ex.self_attention.dropout
            if forced is not None:
                out[key] = forced[key].to(device=device, dtype=torch.float32).contiguous()
            elif p > 0.0:
                out[key] = (torch.rand(N * ex.self_attention.num_heads, S, S, device=device) >= p).float().mul_(1.0 / (1.0 - p))
        return out

    def _desc(self, S, N, M, bf16):
        d = cm.MoeDesc()
        d.S, d.N, d.C, d.M, d.Cy = S, N, self.input_dim, M, self.fc.in_features
        if N != self.conv_adapter.out_channels or M != self.conv_adapter.in_channels:
            raise capi.AvmoeError(f"token counts ({N}, {M}) do not match conv_adapter "
                                  f"({self.conv_adapter.out_channels}, {self.conv_adapter.in_channels})")
        d.E_m, d.E_s = self.num_multimodal_experts, self.num_singlemodal_experts
        d.d, d.groups, d.K = self.input_dim // self.reduction_factor, self.opt.num_conv_group, self.num_tk
        d.use_bn, d.use_gate = int(self.use_bn), int(self.use_gate)
        d.ln_before, d.ln_post = int(bool(self.opt.is_before_layernorm)), int(bool(self.opt.is_post_layernorm))
        d.variant, d.self_attn = cm.VARIANT[self.variant], cm.SELF_ATTN[self._self_attn()]
        d.lb_loss = int(self.variant in ("avvp", "avs") and getattr(self.opt, "use_load_balacing_loss", 0) == 1)
        d.dtype, d.training = (capi.BF16 if bf16 else capi.F32), int(self.training)
        bn = self.multimodal_experts[0].bn1 if (self.use_bn and self.num_multimodal_experts) else \
            (self.singlemodal_experts[0].bn1 if self.use_bn else None)
        d.bn_eps = bn.eps if bn is not None else 1e-5
        d.bn_momentum = bn.momentum if bn is not None else 0.1
        experts = list(self.multimodal_experts) + list(self.singlemodal_experts)
        d.ln_eps = experts[0].ln_before.eps if self.opt.is_before_layernorm else 1e-5
        return d

    def _site_cache(self):
        """Per-module bookkeeping that does not change from call to call (walking named_parameters / named_buffers and resolving
        state_dict keys costs ~0.25 ms per call otherwise -- at the reference's batch of 2 clips that is most of a site's step):
        parameter NAMES and where each parameter / float buffer lives (owner module, attribute), the key -> ABI-field resolution,
        the BatchNorm counters.  The tensors themselves are fetched from their owners' `_parameters` / `_buffers` on every call, so
        a Parameter that was REPLACED (`load_state_dict(assign=True)`, `m.gate = nn.Parameter(..)`, swap-on-convert `.to()`) is
        picked up; `refresh()` drops the cache (call after ADDING / REMOVING parameters or submodules)."""
        c = self.__dict__.get("_avmoe_cache")
        if c is None:
            owners, names = [], []                                     # (owner module, attribute) per parameter, named_parameters order
            seen = set()
            for mod_name, mod in self.named_modules():
                for attr, p in mod._parameters.items():
                    if p is not None and id(p) not in seen:
                        seen.add(id(p))
                        names.append((mod_name + "." if mod_name else "") + attr)
                        owners.append((mod, attr))
            if tuple(names) != tuple(k for k, _ in self.named_parameters()):
                raise capi.AvmoeError("MoEAdapter: parameter registry does not match named_parameters() (parametrized module?)")
            bufs, nbts = [], []                                        # (key, owner module, attribute): float buffers / int64 counters
            for mod_name, mod in self.named_modules():
                for attr, b in mod._buffers.items():
                    if b is not None and b.is_floating_point():
                        bufs.append(((mod_name + "." if mod_name else "") + attr, mod, attr))
                    elif b is not None and attr == "num_batches_tracked" and isinstance(mod, nn.BatchNorm2d):
                        nbts.append(((mod_name + "." if mod_name else "") + attr, mod, attr))
            E_m, E_s = self.num_multimodal_experts, self.num_singlemodal_experts
            c = dict(names=tuple(names), owners=owners, bufs=bufs, nbts=nbts, fill_p=cm.PtrFiller(names, E_m, E_s),
                     fill_b=cm.PtrFiller([k for k, _, _ in bufs], E_m, E_s), fill_n=cm.PtrFiller([k for k, _, _ in nbts], E_m, E_s))
            self.__dict__["_avmoe_cache"] = c
        return c

    def refresh(self):
        for k in ("_avmoe_cache", "_ptrs_cache", "_gptrs_cache", "_ws_sizes"):
            self.__dict__.pop(k, None)

    def __getstate__(self):
        st = self.__dict__.copy()                                      # per-process bookkeeping does not travel (deepcopy / pickle)
        for k in ("_avmoe_cache", "_last_saved", "_ptrs_cache", "_gptrs_cache", "_ws_sizes"):
            st.pop(k, None)
        return st

    def _param_tensors(self):
        """{state_dict key: the Parameter CURRENTLY registered under it} (fetched from the owning modules on every call)."""
        c = self._site_cache()
        out = {k: m._parameters.get(a) for k, (m, a) in zip(c["names"], c["owners"])}
        if any(v is None for v in out.values()):                       # a parameter was removed / set to None: rebuild the registry once
            self.refresh()
            c = self._site_cache()
            out = {k: m._parameters[a] for k, (m, a) in zip(c["names"], c["owners"])}
        return out

    def _fill_ptrs(self, params, keep=None, names=None, device=None):
        """MoePtrs of this site's parameters (in _site_cache order), float buffers, BatchNorm counters and -- for the "v1" experts --
        dropout draws.  The filled struct is kept and handed out again while every pointer is the one it was built from (a call
        costs ~60 data_ptr() reads instead of ~60 validations + ctypes stores: at the reference's batch of 2 clips the host side of
        a site step is what bounds it); `names` / `device`: validate dtype / layout / placement when the struct is (re)built."""
        c = self._site_cache()
        bufs = [m._buffers[a] for _, m, a in c["bufs"]]
        bump = bool(self.training and self.use_bn and c["nbts"])
        cnts = [m._buffers[a] for _, m, a in c["nbts"]] if bump else []
        key = (bump, tuple(t.data_ptr() for t in params), tuple(t.data_ptr() for t in bufs), tuple(t.data_ptr() for t in cnts))
        hit = self.__dict__.get("_ptrs_cache")
        if not keep and hit is not None and hit[0] == key:
            return hit[1]
        if names is not None:
            for k, v in zip(names, params):
                if v.dtype != torch.float32 or not v.is_contiguous() or v.device != device:
                    raise capi.AvmoeError(f"parameter {k} must be a contiguous float32 tensor on {device}")
        P = cm.MoePtrs()
        c["fill_p"].fill(P, params)
        for b in bufs:
            if b.dtype != torch.float32 or not b.is_contiguous():
                raise capi.AvmoeError("BatchNorm running statistics must be contiguous float32 tensors")
        c["fill_b"].fill(P, bufs)
        if bump:                                                      # num_batches_tracked += 1 happens inside the forward's kernels (ABI 6)
            for t in cnts:
                if t.dtype != torch.int64 or (bufs and t.device != bufs[0].device):
                    raise capi.AvmoeError("BatchNorm num_batches_tracked must be int64 tensors on the module's device")
            c["fill_n"].fill(P, cnts)
        if keep:
            P2 = cm.make_ptrs(keep, self.num_multimodal_experts, self.num_singlemodal_experts)
            for j in range(self.num_multimodal_experts + self.num_singlemodal_experts):
                if P2.e[j].sa_keep:
                    P.e[j].sa_keep = P2.e[j].sa_keep
        elif names is not None:
            self.__dict__["_ptrs_cache"] = (key, P)                   # (only structs built WITH validation are kept)
        return P

    def grad_layout(self, align: int = 64):
        """(names, offsets, total) of this site's parameter gradients inside one flat fp32 buffer (offsets in elements,
        aligned) -- the layout a gradient sink must use (avmoe_amd.dp)."""
        names, offs, off = [], [], 0
        for k, v in self.named_parameters():
            names.append(k); offs.append(off)
            off += -(-v.numel() // align) * align
        return tuple(names), tuple(offs), off

    def _buffer_tensors(self):
        return {k: v for k, v in self.named_buffers() if v.is_floating_point()}

    def _run(self, x, vis_token, noise=None):
        # the reference hands (S, C, N, 1) permuted views of token-major memory (net_trans_v3.py:695-698)
        X = x.squeeze(-1).permute(0, 2, 1)
        Y = vis_token.squeeze(-1).permute(0, 2, 1)
        P = self._param_tensors()
        names = tuple(P.keys())
        out, probs, idx, lb = AdapterFunction.apply(self, X, Y, noise, names, *P.values())
        return out.permute(0, 2, 1).unsqueeze(-1), probs, idx, lb

    def forward(self, x, vis_token=None):
        out, _probs, idx, _lb = self._run(x, vis_token)
        return out, idx.unsqueeze(-1)                                         # net_trans_v3.py:487


class MoEAdapterAVQA(MoEAdapter):
    """AVQA signature: no num_tk argument, K comes from opt.num_tokens (net_avst_v2.py:233,350-351)."""

    variant = "avqa"

    def __init__(self, input_dim, output_dim, adapter_kind, dim_list, layer_idx, reduction_factor=16, opt=None,
                 use_bn=True, use_gate=True, conv_dim_in=0, conv_dim_out=0, linear_in=0, linear_out=0):
        super().__init__(input_dim, output_dim, adapter_kind, dim_list, layer_idx, reduction_factor, opt, use_bn,
                         use_gate, opt.num_tokens, conv_dim_in, conv_dim_out, linear_in, linear_out)


class MoEAdapterAVVP(MoEAdapter):
    """AVVP signature (mgn.py:161-217): r / bn / gate / K come from `opt`; forward -> (out, load_balancing_loss)."""

    variant = "avvp"

    def __init__(self, input_dim, output_dim, adapter_kind, dim_list, layer_idx, opt=None, conv_dim_in=0,
                 conv_dim_out=0, linear_in=0, linear_out=0):
        super().__init__(input_dim, output_dim, adapter_kind, dim_list, layer_idx, opt.Adapter_downsample, opt,
                         bool(opt.is_bn), bool(opt.is_gate), opt.num_tokens, conv_dim_in, conv_dim_out, linear_in,
                         linear_out)

    def forward(self, x, vis_token=None):
        out, _probs, _idx, lb = self._run(x, vis_token)
        return out, (lb if self.opt.use_load_balacing_loss == 1 else 0.)       # mgn.py:212-217


class MoEAdapterAVS(MoEAdapter):
    """AVS signature (PVT_AVSModel_v2.py:253-312): forward(x, vis_token, is_training=True) ->
    (out, expert_indices (S,1), gating_probs (S,1,E), load_balancing_loss)."""

    variant = "avs"

    def forward(self, x, vis_token=None, is_training=True):
        noise = None
        if is_training:                                                        # PVT_AVSModel_v2.py:294-296
            E = self.num_multimodal_experts + self.num_singlemodal_experts
            noise = torch.randn(x.shape[0], E, device=x.device, dtype=torch.float32) * 0.01
        out, probs, idx, lb = self._run(x, vis_token, noise)
        return out, idx.unsqueeze(-1), probs.unsqueeze(1), (lb if self.opt.use_load_balacing_loss == 1 else 0.)


def _storage_range(t: torch.Tensor):
    lo = t.untyped_storage().data_ptr()
    return lo, lo + t.untyped_storage().nbytes()


def _safe_inplace(base: torch.Tensor, others) -> bool:
    """May `base` be overwritten in place?  It has to own its storage (not a view of something else somebody may read)
    and that storage must not overlap any of `others`."""
    if base._base is not None or base.untyped_storage().nbytes() != base.numel() * base.element_size():
        return False
    lo, hi = _storage_range(base)
    for o in others:
        olo, ohi = _storage_range(o)
        if lo < ohi and olo < hi:
            return False
    return True


class AdapterPair(nn.Module):
    """The two adapter sites of one backbone layer run as one autograd node:

        pair = AdapterPair(audio_site, visual_site)            # two MoEAdapter* modules of one variant (shared, not copied)
        out_a, idx_a, out_v, idx_v = pair(f_a, f_v)            # AVE / AVQA: == audio_site(f_a, f_v) + visual_site(f_v, f_a)
        out_a, lb_a, out_v, lb_v = pair(f_a, f_v)              # AVVP  (mgn.py:212-217)
        out_a, idx_a, probs_a, lb_a, out_v, idx_v, probs_v, lb_v = pair(f_a, f_v, is_training=True)      # AVS (PVT_AVSModel_v2.py:294-312)

    Numerically identical to calling the two sites one after the other; in the backward the gradient each token tensor
    receives from its second use is added inside the GEMM epilogues instead of by a separate accumulation kernel."""

    def __init__(self, site_a: MoEAdapter, site_b: MoEAdapter, concurrent: bool = True):
        """concurrent=True runs the two sites on two HIP streams (their kernels overlap; each token tensor collects its two gradients
        in ONE buffer: a site overwrites its own tokens' gradient with its dX and, behind an event, adds its dY to the other tensor in
        the GEMM epilogue -- nobody waits before the last section); False runs them back to back on the caller's stream.  Sites with
        latent self attention on their own tokens (AVS "v2") write dX in their last section too, so their backward cannot be split:
        such a pair always runs back to back."""
        super().__init__()
        if site_a.variant != site_b.variant:
            raise ValueError(f"AdapterPair: the two sites have different signatures ({site_a.variant} / {site_b.variant})")
        splittable = all(m._self_attn() != "v2" for m in (site_a, site_b))
        self.concurrent, self._side, self._events = bool(concurrent) and splittable, None, None
        # same_stream (measurement aid, bench.py's per-launch profiling pass): the two-stream SCHEDULE -- every kernel in the variant
        # the concurrent mode launches (dX overwrites, dY adds behind the other site's dX) -- issued on the caller's stream alone,
        # so that an event bracket around a launch measures that launch and not the other stream's kernels
        self.same_stream = False
        self.variant = site_a.variant
        self.site_a, self.site_b = site_a, site_b

    def forward(self, x_a, x_b, add_to=(None, None), is_training=True):
        """add_to = (base_a, base_b): token-major (S, N, C) residual streams (or None) that receive `+= adapter output` IN PLACE
        inside the output GEMM and are returned (in the (S, C, N, 1) view) instead of the bare adapter outputs -- the
        `f = f + f_res` of net_trans_v3.py:709,712 without a separate pass.  Only hand in tensors nobody else needs at their
        old value (e.g. the fresh result of `x + attention(x)`).
        is_training: the AVS sites' flag (logit noise, drawn for site A first -- the order of two separate calls)."""
        Xa = x_a.squeeze(-1).permute(0, 2, 1)
        Xb = x_b.squeeze(-1).permute(0, 2, 1)
        Pa, Pb = self.site_a._param_tensors(), self.site_b._param_tensors()
        if self.concurrent and self._side is None:
            self._side = side_stream(x_a.device)
            self._events = tuple(torch.cuda.Event() for _ in range(6))      # backward hand-over (2), fork, join, hop-1 chains done (2)
        for base, X in zip(add_to, (Xa, Xb)):
            if base is not None and not _safe_inplace(base, (Xa, Xb)):
                raise capi.AvmoeError("add_to must own its storage (no view) and must not overlap the token tensors")
        noises = (None, None)
        if self.variant == "avs" and is_training:                             # PVT_AVSModel_v2.py:294-296
            noises = tuple(torch.randn(x.shape[0], m.num_multimodal_experts + m.num_singlemodal_experts, device=x.device,
                                       dtype=torch.float32) * 0.01 for m, x in ((self.site_a, x_a), (self.site_b, x_b)))
        side = ((torch.cuda.current_stream(x_a.device) if self.same_stream else self._side,) + self._events) if self.concurrent else None
        # With a gradient sink on both sites (avmoe_amd.dp.AdapterGradReducer(sites=...)) and every parameter trainable, the backward
        # writes the parameter gradients into the reducer's buckets and autograd gets None for them anyway: ONE anchor parameter
        # then stands in for the ~120 of the pair (unwrapping, saving and returning a gradient slot for each of them is a third of
        # the host time of a step at the reference's batch of 2 clips -- tests/dev/host_split.py).
        lean = torch.is_grad_enabled() and all(
            getattr(m, "_grad_sink", None) is not None and m._grad_sink.flat is not None and all(p.requires_grad for p in P.values())
            and m._grad_sink.matches(tuple(P.keys()), P) for m, P in ((self.site_a, Pa), (self.site_b, Pb)))
        plist = (next(iter(Pa.values())),) if lean and len(Pa) > 1 else (*Pa.values(), *Pb.values())
        out_a, out_b, idx_a, idx_b, pr_a, pr_b, lb_a, lb_b = _PairFunction.apply(
            self.site_a, self.site_b, side, Xa, Xb, add_to[0], add_to[1], tuple(Pa.keys()), tuple(Pb.keys()), noises, *plist)
        out_a, out_b = out_a.permute(0, 2, 1).unsqueeze(-1), out_b.permute(0, 2, 1).unsqueeze(-1)
        if self.variant in ("ave", "avqa"):
            return out_a, idx_a.unsqueeze(-1), out_b, idx_b.unsqueeze(-1)
        use_lb = (self.site_a.opt.use_load_balacing_loss == 1, self.site_b.opt.use_load_balacing_loss == 1)
        lb_a, lb_b = (lb_a if use_lb[0] else 0.), (lb_b if use_lb[1] else 0.)
        if self.variant == "avvp":
            return out_a, lb_a, out_b, lb_b
        return (out_a, idx_a.unsqueeze(-1), pr_a.unsqueeze(1), lb_a, out_b, idx_b.unsqueeze(-1), pr_b.unsqueeze(1), lb_b)
