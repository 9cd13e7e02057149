"""The reference-shaped nn.Module API end to end on the GPU: same constructor call, (S,C,N,1) permuted views in,
reference return tuples out, autograd through the HIP path, in-place BatchNorm buffer updates."""
import pytest
import torch

from oracle import avmoe_oracle as O
from tests.golden_util import load_golden, split_params, assert_grads_close, mha_keep_of
from tests.test_adapters_api import build_module

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["ave_train", "ave_eval", "avqa_train", "avvp_train", "avs_train_nonoise", "avs_v2_train", "ave_noln_nogate",
                                  "avs_v1_train", "avs_v1_eval", "avvp_fast_train", "avs_v1_fast_train", "ave_ship_train"])
def test_module_forward_backward_matches_reference_vectors(name):
    meta, cfg, t = load_golden(name)
    P, B = split_params(t)
    dev = torch.device("cuda:0")
    m = build_module(meta["which"], cfg).to(dev)
    m.load_state_dict({**P, **B}, strict=True)
    m.train(bool(meta["module_train"]))
    if mha_keep_of(t) is not None:          # "v1": replay the dropout draw recorded from the reference's MultiheadAttention
        from avmoe_amd._capi_moe import SA_KEEP
        m.attention_keep = {f"{pre}.{SA_KEEP}": v for pre, v in mha_keep_of(t).items()}
    X = t["X"].to(dev).requires_grad_(True)
    Y = t["Y"].to(dev).requires_grad_(True)
    xin, yin = X.permute(0, 2, 1).unsqueeze(-1), Y.permute(0, 2, 1).unsqueeze(-1)
    if meta["which"].startswith("avs"):
        out, idx, probs, lb = m(xin, yin, is_training=False)
        assert probs.shape == (X.shape[0], 1, cfg.E)
        assert torch.allclose(probs.reshape(-1, cfg.E).cpu(), t["probs"], atol=1e-5)
    elif meta["which"] == "avvp":
        out, lb = m(xin, yin)
        idx = None
        assert abs(float(lb) - float(t["lb"])) < 1e-4 * max(1.0, abs(float(t["lb"])))
    else:
        out, idx = m(xin, yin)
        lb = 0.0
    assert out.shape == xin.shape
    if idx is not None:
        assert idx.shape == (X.shape[0], 1) and idx.dtype == torch.int64
        assert torch.equal(idx.reshape(-1).cpu(), t["idx"])
    out_tm = out.squeeze(-1).permute(0, 2, 1)
    assert out_tm.is_contiguous()                      # the caller's residual add needs no copy
    assert float((out_tm.cpu() - t["out"]).abs().max() / t["out"].abs().max()) < 1e-3
    loss = (out_tm * t["grad_out"].to(dev)).sum()
    if torch.is_tensor(lb) and meta["lb_weight"]:
        loss = loss + meta["lb_weight"] * lb
    loss.backward()
    grads = {"X": X.grad.cpu(), "Y": Y.grad.cpu()}
    for k, v in m.named_parameters():
        assert v.grad is not None, k
        grads[k] = v.grad.cpu()
    assert_grads_close(grads, t, rtol=1e-3)
    if meta["module_train"] and cfg.use_bn:
        for k, v in m.named_buffers():
            ref = t[f"newbuffer.{k}"]
            assert torch.allclose(v.cpu().to(ref.dtype), ref, rtol=2e-4, atol=2e-5), k


def test_frozen_parameters_get_no_gradient_and_bf16_runs():
    meta, cfg, t = load_golden("ave_train")
    P, B = split_params(t)
    dev = torch.device("cuda:0")
    m = build_module("ave", cfg).to(dev)
    m.load_state_dict({**P, **B})
    for k, v in m.named_parameters():
        v.requires_grad_("router" in k)
    X = t["X"].to(dev, torch.bfloat16)
    Y = t["Y"].to(dev, torch.bfloat16)
    out, idx = m(X.permute(0, 2, 1).unsqueeze(-1), Y.permute(0, 2, 1).unsqueeze(-1))
    assert out.dtype == torch.bfloat16
    out.float().sum().backward()
    for k, v in m.named_parameters():
        assert (v.grad is not None) == ("router" in k), k


def test_gradient_sink_matches_autograd_accumulation():
    """AdapterGradReducer(sites=[...]): the backward writes parameter gradients straight into the reducer's flat bucket
    (param.grad are views of it) -- same numbers as plain autograd accumulation, also over two accumulation micro-steps,
    and num_batches_tracked advances once per training forward for every BatchNorm."""
    import copy
    from avmoe_amd.dp import AdapterGradReducer
    dev = torch.device("cuda:0")
    cfg = O.AdapterConfig(Cx=64, Nx=50, Cy=48, Ny=20, reduction=4, groups=2, K=6)
    ref = build_module("ave", cfg).to(dev).train()
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if k.endswith(("gate", "gate_av")):
                p.fill_(0.4)
    fused = copy.deepcopy(ref)
    red = AdapterGradReducer(list(fused.parameters()), sites=[fused])
    g = torch.Generator().manual_seed(5)
    for micro in range(2):
        X = torch.randn(4, cfg.Cx, cfg.Nx, 1, generator=g).to(dev).requires_grad_(True)
        Y = torch.randn(4, cfg.Cy, cfg.Ny, 1, generator=g).to(dev)
        G = torch.randn(4, cfg.Cx, cfg.Nx, 1, generator=g).to(dev)
        Xf = X.detach().clone().requires_grad_(True)
        ref(X, Y)[0].backward(G)
        red.begin(sync=(micro == 1))
        fused(Xf, Y)[0].backward(G)
        red.finish()
        assert torch.equal(X.grad, Xf.grad)
    for (k, p), (_, q) in zip(ref.named_parameters(), fused.named_parameters()):
        assert q.grad.data_ptr() >= red.buckets[0].flat.data_ptr()
        scale = float(p.grad.abs().max()) + 1e-6
        assert float((p.grad - q.grad).abs().max()) <= 1e-5 * scale, k
    for (k, b), (_, c) in zip(ref.named_buffers(), fused.named_buffers()):
        assert torch.equal(b, c), k
        if k.endswith("num_batches_tracked"):
            assert int(b) == 2
    red.zero_grad()
    assert all(float(q.grad.abs().max()) == 0.0 for q in fused.parameters())


def test_lazy_zero_grad_with_a_site_that_falls_off_its_sink():
    """a parameter frozen AFTER the reducer was built makes the site's backward take the autograd path although a sink is
    attached; after zero_grad(lazy=True) its slice still holds the previous step's gradients -- they must not be added to, and
    finish() must not wipe the new ones."""
    import copy
    from avmoe_amd.dp import AdapterGradReducer
    dev = torch.device("cuda:0")
    cfg = O.AdapterConfig(Cx=64, Nx=50, Cy=48, Ny=20, reduction=4, groups=2, K=6)
    ref = build_module("ave", cfg).to(dev).train()
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if k.endswith(("gate", "gate_av")):
                p.fill_(0.4)
    fused = copy.deepcopy(ref)
    red = AdapterGradReducer(list(fused.parameters()), sites=[fused])
    g = torch.Generator().manual_seed(6)
    def batch():
        return (torch.randn(4, cfg.Cx, cfg.Nx, 1, generator=g).to(dev), torch.randn(4, cfg.Cy, cfg.Ny, 1, generator=g).to(dev),
                torch.randn(4, cfg.Cx, cfg.Nx, 1, generator=g).to(dev))
    X, Y, G = batch()
    red.begin(sync=True); fused(X, Y)[0].backward(G); red.finish()          # step 1 through the sink: the slice holds its gradients
    red.zero_grad(lazy=True)
    frozen = "fc.bias"
    dict(fused.named_parameters())[frozen].requires_grad_(False)            # ... now the site falls off the sink
    dict(ref.named_parameters())[frozen].requires_grad_(False)
    X, Y, G = batch()
    ref(X, Y)[0].backward(G)
    red.begin(sync=True); fused(X, Y)[0].backward(G); red.finish()
    for (k, p), (_, q) in zip(ref.named_parameters(), fused.named_parameters()):
        if k == frozen:
            continue
        scale = float(p.grad.abs().max()) + 1e-6
        assert float((p.grad - q.grad).abs().max()) <= 1e-5 * scale, k


@pytest.mark.parametrize("concurrent", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_adapter_pair_equals_two_sites(dtype, concurrent):
    """AdapterPair(audio_site, visual_site) == the two MoEAdapter calls of net_trans_v3.py:695-698; the token gradients
    (each tensor is X of one site and Y of the other) are accumulated inside the GEMM epilogues."""
    from avmoe_amd.adapters import AdapterPair
    dev = torch.device("cuda:0")
    ca = O.AdapterConfig(Cx=64, Nx=272, Cy=48, Ny=80, reduction=4, groups=2, K=8)
    cb = O.AdapterConfig(Cx=48, Nx=80, Cy=64, Ny=272, reduction=4, groups=2, K=8)
    torch.manual_seed(3)
    sa, sb = build_module("ave", ca).to(dev).train(), build_module("ave", cb).to(dev).train()
    with torch.no_grad():
        for m in (sa, sb):
            for k, p in m.named_parameters():
                if k.endswith(("gate", "gate_av")):
                    p.fill_(0.3)
    g = torch.Generator().manual_seed(9)
    fa = (0.5 * torch.randn(4, ca.Cx, ca.Nx, 1, generator=g)).to(dev, dtype)
    fv = (0.5 * torch.randn(4, cb.Cx, cb.Nx, 1, generator=g)).to(dev, dtype)
    ga = torch.randn(4, ca.Cx, ca.Nx, 1, generator=g).to(dev, dtype)
    gv = torch.randn(4, cb.Cx, cb.Nx, 1, generator=g).to(dev, dtype)
    bufs = [{k: b.clone() for k, b in m.named_buffers()} for m in (sa, sb)]

    def run(paired):
        for m, bb in zip((sa, sb), bufs):
            m.zero_grad()
            m.load_state_dict({**m.state_dict(), **bb})
        xa, xv = fa.clone().requires_grad_(True), fv.clone().requires_grad_(True)
        if paired:
            # concurrent: two streams, ONE gradient buffer per token tensor (sections of avmoe_moe_backward_part + events)
            oa, ia, ov, iv = AdapterPair(sa, sb, concurrent=concurrent)(xa, xv)
        else:
            (oa, ia), (ov, iv) = sa(xa, xv), sb(xv, xa)
        torch.autograd.backward([oa, ov], [ga, gv])
        return oa.detach(), ov.detach(), ia, iv, xa.grad, xv.grad, [p.grad.clone() for m in (sa, sb) for p in m.parameters()]

    ref, got = run(False), run(True)
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]) and torch.equal(ref[2], got[2]) and torch.equal(ref[3], got[3])
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    for r_, g_ in ((ref[4], got[4]), (ref[5], got[5])):
        assert float((r_.float() - g_.float()).abs().max()) <= tol * float(r_.float().abs().max())
    for r_, g_ in zip(ref[6], got[6]):
        assert torch.equal(r_, g_)


def test_flat_adam_matches_torch_adam_and_histogram_matches_bincount():
    """avmoe_amd.train.FlatAdam (one HIP kernel per flat bucket) against torch.optim.Adam over three steps of the same
    gradients, with weight decay and StepLR-style decay; ExpertActivationCounter against torch.bincount."""
    import copy
    from avmoe_amd.dp import AdapterGradReducer
    from avmoe_amd.train import FlatAdam, ExpertActivationCounter
    dev = torch.device("cuda:0")
    cfg = O.AdapterConfig(Cx=64, Nx=50, Cy=48, Ny=20, reduction=4, groups=2, K=6)
    ref = build_module("ave", cfg).to(dev).train()
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if k.endswith(("gate", "gate_av")):
                p.fill_(0.4)
    fused = copy.deepcopy(ref)
    topt = torch.optim.Adam(ref.parameters(), lr=3e-3, weight_decay=1e-2)
    sched = torch.optim.lr_scheduler.StepLR(topt, step_size=2, gamma=0.5)
    red = AdapterGradReducer(list(fused.parameters()), sites=[fused])
    fopt = FlatAdam(red, lr=3e-3, weight_decay=1e-2, step_size=2, gamma=0.5)
    g = torch.Generator().manual_seed(2)
    for epoch in range(3):
        X = torch.randn(4, cfg.Cx, cfg.Nx, 1, generator=g).to(dev)
        Y = torch.randn(4, cfg.Cy, cfg.Ny, 1, generator=g).to(dev)
        G = torch.randn(4, cfg.Cx, cfg.Nx, 1, generator=g).to(dev)
        topt.zero_grad()
        ref(X, Y)[0].backward(G)
        red.begin(sync=True)
        fused(X, Y)[0].backward(G)                      # fills the flat bucket through the gradient sink
        red.finish()
        with torch.no_grad():                           # both optimizers see bit-identical gradients: the comparison is
            for p, q in zip(ref.parameters(), fused.parameters()):      # about the update arithmetic only
                q.grad.copy_(p.grad)
        topt.step(); sched.step()
        fopt.step(); fopt.epoch_end(); red.zero_grad()
        with torch.no_grad():
            for (k, p), (_, q) in zip(ref.named_parameters(), fused.named_parameters()):
                # |update| <= lr; where sqrt(v) ~ eps the two fp32 evaluation orders differ by a small fraction of it
                assert float((p - q).abs().max()) <= 1e-6 * float(p.abs().max()) + 2e-3 * 3e-3, (epoch, k)
                q.copy_(p)
    cnt = ExpertActivationCounter(["audio_p1", "video_p1"], num_layers=3, num_experts=4, device=dev)
    idx = torch.randint(0, 4, (37, 1), generator=g).to(dev)
    cnt.update("video_p1", 2, idx); cnt.update("video_p1", 2, idx); cnt.update("audio_p1", 0, idx[:5])
    tabs = cnt.numpy()
    assert (tabs["video_p1"][2] == 2 * torch.bincount(idx.reshape(-1).cpu(), minlength=4).numpy()).all()
    assert tabs["audio_p1"][0].sum() == 5 and tabs["audio_p1"][1:].sum() == 0 and tabs["video_p1"][:2].sum() == 0


def test_router_topk_matches_stable_sort_and_argmax():
    """avmoe_router_topk (extension, BASELINE config 3 "top-k=2"): the k most probable experts per frame, ties in expert order;
    column 0 is the site's own first-max argmax -- on real AVVP probabilities and on rows with exact ties."""
    from avmoe_amd.train import topk_experts
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    probs = torch.softmax(torch.randn(1000, 4, generator=g), -1)
    probs[::7] = 0.25                                   # exact four-way ties
    probs[1::7, 1] = probs[1::7, 0]                     # two-way ties
    for k in (1, 2, 4):
        got = topk_experts(probs.to(dev), k).cpu()
        ref = torch.sort(probs, dim=-1, descending=True, stable=True).indices[:, :k]
        assert torch.equal(got, ref)
    assert torch.equal(topk_experts(probs.to(dev), 1).cpu().reshape(-1), torch.argmax(probs, -1))
    meta, cfg, t = load_golden("avvp_train")            # the module's probabilities -> top-2
    P, B = split_params(t)
    from tests.moe_gpu_util import MoeRun
    run = MoeRun(cfg, P, B, t["X"], t["Y"], training=True).forward()
    top2 = topk_experts(run.probs, 2).cpu()
    assert torch.equal(top2[:, 0], t["idx"]) and torch.equal(top2, torch.sort(t["probs"], dim=-1, descending=True, stable=True).indices[:, :2])


@pytest.mark.parametrize("concurrent", [False, True])
def test_adapter_pair_with_frame_attention_experts(concurrent):
    """AdapterPair over sites whose unimodal experts use MultiheadAttention across the frames (is_self_attention, "v1"): the
    dropout draw of each site travels with its forward state to the backward, on either stream."""
    from avmoe_amd.adapters import AdapterPair
    from avmoe_amd._capi_moe import SA_KEEP
    dev = torch.device("cuda:0")
    S = 5
    ca = O.AdapterConfig(Cx=64, Nx=40, Cy=96, Ny=24, reduction=4, groups=2, K=8, self_attn="v1")
    cb = O.AdapterConfig(Cx=96, Nx=24, Cy=64, Ny=40, reduction=4, groups=2, K=8, self_attn="v1")
    torch.manual_seed(3)
    sa, sb = build_module("ave", ca).to(dev).train(), build_module("ave", cb).to(dev).train()
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for m, c in ((sa, ca), (sb, cb)):
            for k, p in m.named_parameters():
                if k.endswith(("gate", "gate_av")):
                    p.fill_(0.3)
            m.attention_keep = {f"singlemodal_experts.{j}.{SA_KEEP}": (torch.rand(c.Nx * 4, S, S, generator=g) >= 0.2).float() / 0.8
                                for j in range(c.E_s)}
    fa = (0.5 * torch.randn(S, ca.Cx, ca.Nx, 1, generator=g)).to(dev)
    fv = (0.5 * torch.randn(S, cb.Cx, cb.Nx, 1, generator=g)).to(dev)
    ga, gv = torch.randn(S, ca.Cx, ca.Nx, 1, generator=g).to(dev), torch.randn(S, cb.Cx, cb.Nx, 1, generator=g).to(dev)
    bufs = [{k: b.clone() for k, b in m.named_buffers()} for m in (sa, sb)]

    def run(paired):
        for m, bb in zip((sa, sb), bufs):
            m.zero_grad()
            m.load_state_dict({**m.state_dict(), **bb})
        xa, xv = fa.clone().requires_grad_(True), fv.clone().requires_grad_(True)
        if paired:
            oa, ia, ov, iv = AdapterPair(sa, sb, concurrent=concurrent)(xa, xv)
        else:
            (oa, ia), (ov, iv) = sa(xa, xv), sb(xv, xa)
        torch.autograd.backward([oa, ov], [ga, gv])
        return oa.detach(), ov.detach(), xa.grad, xv.grad, [p.grad.clone() for m in (sa, sb) for p in m.parameters()]

    ref, got = run(False), run(True)
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    for r_, g_ in ((ref[2], got[2]), (ref[3], got[3])):
        assert float((r_ - g_).abs().max()) <= 1e-5 * float(r_.abs().max())
    for r_, g_ in zip(ref[4], got[4]):
        assert torch.equal(r_, g_)
    assert any("self_attention.in_proj_weight" in k for k, _ in sa.named_parameters())


@pytest.mark.parametrize("frozen_inputs", [False, True])
def test_adapter_pair_with_gradient_sinks_is_lean_and_equal(frozen_inputs):
    """With AdapterGradReducer(sites=...) on both sites the pair hands autograd ONE anchor parameter instead of every parameter (the
    sinks take the gradients): same outputs, same parameter gradients (now views of the reducer's bucket) and token gradients as
    the plain pair -- also when the token tensors come from a frozen backbone (nothing but the anchor requires grad)."""
    import copy
    from avmoe_amd.adapters import AdapterPair, _PairFunction
    from avmoe_amd.dp import AdapterGradReducer
    dev = torch.device("cuda:0")
    ca = O.AdapterConfig(Cx=64, Nx=150, Cy=48, Ny=80, reduction=4, groups=2, K=8)
    cb = O.AdapterConfig(Cx=48, Nx=80, Cy=64, Ny=150, reduction=4, groups=2, K=8)
    torch.manual_seed(3)
    sa, sb = build_module("ave", ca).to(dev).train(), build_module("ave", cb).to(dev).train()
    with torch.no_grad():
        for m in (sa, sb):
            for k, p in m.named_parameters():
                if k.endswith(("gate", "gate_av")):
                    p.fill_(0.3)
    ra, rb = copy.deepcopy(sa), copy.deepcopy(sb)                       # the plain pair (no sinks)
    red = AdapterGradReducer([p for m in (sa, sb) for p in m.parameters()], sites=[sa, sb])
    g = torch.Generator().manual_seed(9)
    fa = (0.5 * torch.randn(4, ca.Cx, ca.Nx, 1, generator=g)).to(dev)
    fv = (0.5 * torch.randn(4, cb.Cx, cb.Nx, 1, generator=g)).to(dev)
    ga, gv = torch.randn(4, ca.Cx, ca.Nx, 1, generator=g).to(dev), torch.randn(4, cb.Cx, cb.Nx, 1, generator=g).to(dev)

    def run(pair, begin=None):
        xa, xv = fa.clone().requires_grad_(not frozen_inputs), fv.clone().requires_grad_(not frozen_inputs)
        seen = []
        orig = _PairFunction.apply
        if begin is not None:
            begin()
        oa, _, ov, _ = pair(xa, xv)
        torch.autograd.backward([oa, ov], [ga, gv])
        return oa.detach(), ov.detach(), xa.grad, xv.grad

    ref = run(AdapterPair(ra, rb))
    lean_pair = AdapterPair(sa, sb)
    n_inputs = []
    orig_apply = _PairFunction.apply
    _PairFunction.apply = staticmethod(lambda *a: (n_inputs.append(len(a)), orig_apply(*a))[1])
    try:
        got = run(lean_pair, begin=lambda: red.begin(sync=True))
        red.finish()
    finally:
        _PairFunction.apply = orig_apply
    assert n_inputs == [11]                                              # ten fixed arguments + one anchor parameter
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    if not frozen_inputs:
        assert torch.equal(ref[2], got[2]) and torch.equal(ref[3], got[3])
    else:
        assert got[2] is None and got[3] is None
    for (k, p), (_, q) in zip(list(ra.named_parameters()) + list(rb.named_parameters()), list(sa.named_parameters()) + list(sb.named_parameters())):
        assert q.grad.data_ptr() >= red.buckets[0].flat.data_ptr()
        assert torch.equal(p.grad, q.grad), k


@pytest.mark.parametrize("concurrent", [False, True])
@pytest.mark.parametrize("which", ["avvp", "avs", "avs_v2"])
def test_adapter_pair_avvp_avs_signatures(which, concurrent):
    """AdapterPair over the AVVP (out, lb) and AVS (out, idx, probs, lb) signatures == two module calls, including the gradient
    of the load-balancing losses and the AVS logit noise (drawn in the order of the two calls).  "avs_v2": latent self attention
    on the site's own tokens -- the pair then runs back to back whatever `concurrent` says."""
    from avmoe_amd.adapters import AdapterPair
    dev = torch.device("cuda:0")
    variant = "avvp" if which == "avvp" else "avs"
    kw = dict(reduction=4, groups=2, K=8, variant=variant, lb_loss=True, self_attn=("v2" if which == "avs_v2" else "none"))
    ca = O.AdapterConfig(Cx=64, Nx=72, Cy=48, Ny=40, **kw)
    cb = O.AdapterConfig(Cx=48, Nx=40, Cy=64, Ny=72, **kw)
    torch.manual_seed(5)
    sa, sb = build_module(variant, ca).to(dev).train(), build_module(variant, cb).to(dev).train()
    with torch.no_grad():
        for m in (sa, sb):
            for k, p in m.named_parameters():
                if k.endswith(("gate", "gate_av")):
                    p.fill_(0.3)
    g = torch.Generator().manual_seed(9)
    S = 6
    fa, fv = (0.5 * torch.randn(S, ca.Cx, ca.Nx, 1, generator=g)).to(dev), (0.5 * torch.randn(S, cb.Cx, cb.Nx, 1, generator=g)).to(dev)
    ga, gv = torch.randn(S, ca.Cx, ca.Nx, 1, generator=g).to(dev), torch.randn(S, cb.Cx, cb.Nx, 1, generator=g).to(dev)
    bufs = [{k: b.clone() for k, b in m.named_buffers()} for m in (sa, sb)]
    pair = AdapterPair(sa, sb, concurrent=concurrent)
    assert pair.concurrent == (concurrent and which != "avs_v2")

    def run(paired):
        for m, bb in zip((sa, sb), bufs):
            m.zero_grad()
            m.load_state_dict({**m.state_dict(), **bb})
        torch.manual_seed(77)                          # the AVS noise of both runs
        xa, xv = fa.clone().requires_grad_(True), fv.clone().requires_grad_(True)
        extra = ()
        if variant == "avvp":
            oa, la, ov, lv = pair(xa, xv) if paired else (*sa(xa, xv), *sb(xv, xa))
        else:
            r = pair(xa, xv, is_training=True) if paired else (*sa(xa, xv, is_training=True), *sb(xv, xa, is_training=True))
            oa, ia, pa, la, ov, iv, pv, lv = r
            assert pa.shape == (S, 1, ca.E) and ia.shape == (S, 1)
            extra = (ia, pa, iv, pv)
        torch.autograd.backward([oa, ov, 0.7 * la + 1.3 * lv], [ga, gv, None])
        return (oa.detach(), ov.detach(), la.detach(), lv.detach(), *extra), (xa.grad, xv.grad), [p.grad.clone() for m in (sa, sb) for p in m.parameters()]

    ref, got = run(False), run(True)
    for r_, g_ in zip(ref[0], got[0]):
        assert torch.equal(r_, g_)
    for r_, g_ in zip(ref[1], got[1]):
        assert float((r_ - g_).abs().max()) <= 1e-5 * float(r_.abs().max())
    for r_, g_ in zip(ref[2], got[2]):
        assert torch.equal(r_, g_)
    with pytest.raises(ValueError):
        AdapterPair(sa, build_module("ave", O.AdapterConfig(Cx=48, Nx=40, Cy=64, Ny=72, reduction=4, groups=2, K=8)).to(dev))


def test_flat_adam_two_lr_groups_and_plain_buckets():
    """(1) the reference's two Adam learning-rate groups (AVE/main_trans_v3.py:313-322: lr_mlp for 'mlp_class', lr for the
    adapters; train.sh ships a factor of 100 between them) through select_trainable -> FlatAdam(param_groups=...) against
    torch.optim.Adam with the same groups; (2) the reducer WITHOUT sites= on a 1 + 1 expert site (2-element router.4.bias and
    1-element gates in front of GEMM operands): parameters re-pointed by FlatAdam stay 16-byte aligned and the next
    forward / backward through the HIP path works (unaligned router.0.weight)."""
    import copy
    from avmoe_amd.dp import AdapterGradReducer
    from avmoe_amd.train import FlatAdam, select_trainable
    dev = torch.device("cuda:0")
    cfg = O.AdapterConfig(Cx=64, Nx=50, Cy=48, Ny=20, reduction=4, groups=2, K=6, E_m=1, E_s=1)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.audio_moe_adapter_blocks_p1 = torch.nn.ModuleList([build_module("ave", cfg)])
            self.mlp_class = torch.nn.Linear(cfg.Cx, 5)

        def forward(self, X, Y):
            out, _ = self.audio_moe_adapter_blocks_p1[0](X, Y)
            return self.mlp_class(out.squeeze(-1).mean(-1))
    torch.manual_seed(11)
    ref = Net().to(dev).train()
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if k.endswith(("gate", "gate_av")):
                p.fill_(0.4)
    fused = copy.deepcopy(ref)
    lr, lr_mlp = 5e-3, 5e-5
    tg = select_trainable(ref, lr=lr, lr_mlp=lr_mlp)
    topt = torch.optim.Adam([{"params": [g_["params"]], "lr": g_["lr"]} for g_ in tg if g_["params"].requires_grad])
    fg = select_trainable(fused, lr=lr, lr_mlp=lr_mlp)
    red = AdapterGradReducer([p for p in fused.parameters() if p.requires_grad])           # no sites=: plain size-capped buckets
    fopt = FlatAdam(red, lr=lr, param_groups=fg)
    for p in fused.parameters():
        assert p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0
    start = {k: p.detach().clone() for k, p in fused.named_parameters()}
    g = torch.Generator().manual_seed(2)
    for it in range(3):
        X = torch.randn(4, cfg.Cx, cfg.Nx, 1, generator=g).to(dev)
        Y = torch.randn(4, cfg.Cy, cfg.Ny, 1, generator=g).to(dev)
        G = torch.randn(4, 5, generator=g).to(dev)
        topt.zero_grad()
        ref(X, Y).backward(G)
        red.begin(sync=True)
        fused(X, Y).backward(G)                           # the HIP path reads the re-pointed (flat-buffer) parameters
        red.finish()
        with torch.no_grad():
            for p, q in zip(ref.parameters(), fused.parameters()):
                assert float((p.grad - q.grad).abs().max()) <= 1e-4 * float(p.grad.abs().max()) + 1e-7
                q.grad.copy_(p.grad)
        topt.step()
        fopt.step(); red.zero_grad()
        with torch.no_grad():
            for (k, p), (_, q) in zip(ref.named_parameters(), fused.named_parameters()):
                step_lr = lr_mlp if "mlp_class" in k else lr
                assert float((p - q).abs().max()) <= 1e-6 * float(p.abs().max()) + 2e-3 * step_lr, (it, k)
                q.copy_(p)
    moved = {k: float((p.detach() - start[k]).abs().max()) for k, p in fused.named_parameters()}
    assert max(v for k, v in moved.items() if "mlp_class" in k) <= 3.5 * lr_mlp          # |Adam update| <= ~lr per step
    assert max(v for k, v in moved.items() if "mlp_class" not in k) > 10 * lr_mlp
