"""Round-6 kernels against the kernels they replace (bf16, the tuned bottleneck shape: C = 768, bottleneck 64 in 2 groups, 32 latent tokens).

csrc/tile_stream.hip runs the bottleneck-space passes of csrc/tile_fast.hip with another memory side (one persistent block per CU, wave-private
LDS rings filled by direct global -> LDS loads, counted waits) and the SAME per-token arithmetic: per-token outputs must agree bit for bit
or to the last bf16 bit, sums over tokens to fp32 summation order.  A mis-counted wait (a tile read before it landed), a wrong ring slot or a
ragged tile that reads the wrong rows shows up as an O(1) error here where the bf16-vs-oracle budget (1e-2) could hide it.  The hooks
(include/avmoe.h: avmoe_test_hooks) lift the 32 768-token threshold / switch the streaming form off.
"""
import pytest
import torch

from oracle import avmoe_oracle as O

pytestmark = pytest.mark.gpu

# (frames, tokens of X, tokens of Y, cross-modal experts, unimodal experts): whole and ragged 16-token tiles, a frame shorter than one
# tile, more virtual blocks than CUs (persistent blocks that walk several frames) and fewer, 2 / 3 / 4 experts
SHAPES = {
    "whole_tiles": (12, 256, 64, 2, 2),
    "ragged_196": (20, 196, 64, 2, 2),
    "short_frame_20": (40, 20, 32, 2, 2),
    "many_frames": (300, 48, 32, 2, 2),
    "long_frames": (3, 2304, 64, 2, 2),
    "two_experts": (16, 180, 64, 1, 1),
    "three_experts": (16, 180, 64, 1, 2),
    "one_frame": (1, 1024, 64, 2, 2),
}


def _cfg(N, M, E_m, E_s):
    return O.AdapterConfig(Cx=768, Nx=N, Cy=768, Ny=M, reduction=12, groups=2, K=32, E_m=E_m, E_s=E_s, variant="ave")


def _run_site(cfg, S, seed, want=()):
    from tests.moe_gpu_util import MoeRun
    from avmoe_amd import _capi
    P, B = O.init_params(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    X = 0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    Y = 0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)
    G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    L = _capi.lib()
    L.avmoe_prof_reset(); L.avmoe_prof_enable(1)
    try:
        run = MoeRun(cfg, P, B, X, Y, bf16=True, training=True).forward()
        got = run.backward(G)
        torch.cuda.synchronize()
        ran = [f["name"] for f in _capi.prof_report()]
    finally:
        L.avmoe_prof_enable(0); L.avmoe_prof_reset()
    assert run.guards_intact(), "a kernel wrote past its workspace"
    bufs = {n: run.buf(n).clone() for n in want}
    return run.out.float().cpu(), run.idx.cpu(), {k: v.float().cpu() for k, v in got.items()}, bufs, ran


@pytest.mark.parametrize("shape", list(SHAPES))
def test_streaming_bottleneck_kernels_equal_the_register_resident_ones(shape):
    from avmoe_amd import _capi
    S, N, M, E_m, E_s = SHAPES[shape]
    cfg = _cfg(N, M, E_m, E_s)
    want = ("dGq", "dzp")       # the weighted Gram products (fused into post_small_bwd's pass by the streaming form) ; dz' as mid_bwd leaves it
    with _capi.test_hooks(_capi.HOOK_KFS_OFF):
        out_f, idx_f, g_f, b_f, ran_f = _run_site(cfg, S, 11, want)
    with _capi.test_hooks(_capi.HOOK_KFS_FORCE):
        out_s, idx_s, g_s, b_s, ran_s = _run_site(cfg, S, 11, want)
    # same bf16 operands (z' and dSoo z'), fp32 accumulation in another order; the streaming form computes one off-diagonal 16 x 16 tile per
    # group and mirrors it, gram.hip computes both -- sum_t bf16(w z_a) z_b against sum_t bf16(w z_b) z_a: one bf16 rounding of every term
    # apart (measured 2.1e-3 of the largest entry, dSoo has both signs); a wrong fragment or a wrong mirror index would be O(1)
    dg = float((b_s["dGq"] - b_f["dGq"]).abs().max()) / max(float(b_f["dGq"].abs().max()), 1e-30)
    assert dg <= 6e-3, ("dGq", dg)
    assert not any(n.endswith("(stream)") for n in ran_f), ran_f
    if E_m + E_s != 3:                                                  # (three experts: no Gram-fused mode, the streaming form steps aside)
        assert any(n.endswith("(stream)") for n in ran_s), ran_s      # (the hook did switch the kernels)
    assert torch.equal(idx_f, idx_s)
    assert torch.isfinite(out_s).all() and all(torch.isfinite(v).all() for v in g_s.values())
    # same per-token arithmetic; the BatchNorm / LayerNorm statistics are sums over tokens in another order (fp32), so a per-token value
    # can move by a last bf16 bit
    assert float((out_s - out_f).abs().max()) <= 2e-2 * float(out_f.abs().max())
    assert float((out_s - out_f).norm() / out_f.norm()) <= 2e-3
    gmax = max(float(v.norm()) for v in g_f.values())
    for k, v in g_f.items():
        err = float((g_s[k] - v).norm()) / max(float(v.norm()), 1e-3 * gmax)
        # (two bf16 evaluations: the bar of tests/test_round5_kernels_gpu.py; measured <= 4.3e-3 on the major tensors)
        tol = 1e-2 if float(v.norm()) >= 1e-2 * gmax else 6e-2
        if k.endswith(("gate_av", "gate_self")):     # the fused backward pass forms <dzraw, a TW> from dzraw in fp32 registers, tile_fast.hip from the bf16 copy it reads back
            tol = 6e-2
        assert err <= tol, (k, err)


def test_streaming_kernels_repeat_bit_for_bit():
    """No atomics, fixed tile-to-wave assignment: two runs of the same step agree exactly."""
    from avmoe_amd import _capi
    S, N, M, E_m, E_s = SHAPES["ragged_196"]
    cfg = _cfg(N, M, E_m, E_s)
    with _capi.test_hooks(_capi.HOOK_KFS_FORCE):
        out_a, _, g_a, _, _ = _run_site(cfg, S, 5)
        out_b, _, g_b, _, _ = _run_site(cfg, S, 5)
    assert torch.equal(out_a, out_b)
    for k in g_a:
        assert torch.equal(g_a[k], g_b[k]), k


@pytest.mark.parametrize("stream", [False, True])
def test_split_bf16_matvecs_with_large_mean_small_variance_activations(stream):
    """The mat-vecs of the bf16 instantiations run on the bf16 matrix pipe with both operands as TWO bf16 planes, lo . lo dropped
    (csrc/tile_fast.hip::mmT_split, csrc/tile_stream.hip::mm_presplit).  Worst case for them: bottleneck activations z' with a
    large mean and a small variance (BatchNorm-1 with bias 4 and weight 0.05 in front of the ReLU: z' ~ 4 +- 0.05), so that the LayerNorm-post
    variance  Soo / C - mup^2 = (z'^T G z' + 2 z' . vh + H2) / C - ((z' . us + H1) / C)^2  is a small difference of large numbers.
    ISOLATED from everything else bf16 does to such activations (the stored z and Apost, the Gram kernel's operands: the oracle comparison
    below only bounds those against eager autocast): rp, the 1 / sigma the kernel keeps in fp32, is recomputed in fp64 from the kernel's OWN
    buffers (Z as stored, the BatchNorm-1 rows, Gq, us / vh / H) -- what is left is the mat-vec's arithmetic alone."""
    from tests.moe_gpu_util import MoeRun
    from avmoe_amd import _capi
    cfg = _cfg(256, 64, 2, 2)
    P, B = O.init_params(cfg, seed=17)
    for k in list(P):
        if k.endswith("bn1.bias"):
            P[k] = torch.full_like(P[k], 4.0)
        if k.endswith("bn1.weight"):
            P[k] = torch.full_like(P[k], 0.05)
    g = torch.Generator().manual_seed(171)
    S, E, dg, gr = 12, cfg.E, 32, 2
    X = (0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g)).bfloat16().float()
    Y = (0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)).bfloat16().float()
    G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g).bfloat16().float()
    fwd, grads = O.moe_forward_backward(P, B, X, Y, cfg, G, training=True)
    with _capi.test_hooks(_capi.HOOK_KFS_FORCE if stream else _capi.HOOK_KFS_OFF):
        run = MoeRun(cfg, P, B, X, Y, bf16=True, training=True).forward()
        rp = run.buf("rpmup", shape=(2, E, S, cfg.Nx))[0].double()
        Z = run.buf("Z", torch.bfloat16, (S, cfg.Nx, gr, E, dg)).double()
        bn1 = run.buf("bn1", shape=(4, gr, E, dg)).double()
        Gq = run.buf("Gq", shape=(gr, E, dg, dg)).double()
        uvh = run.buf("uvh").double()
        got = run.backward(G)
    DZ = gr * E * dg
    us, vh = uvh[:DZ].reshape(gr, E, dg), uvh[DZ:2 * DZ].reshape(gr, E, dg)
    H1, H2 = uvh[2 * DZ:2 * DZ + gr * E].reshape(gr, E).sum(0), uvh[2 * DZ + gr * E:2 * DZ + 2 * gr * E].reshape(gr, E).sum(0)
    relu = torch.tensor([1.0 if e < cfg.E_m else 0.0 for e in range(E)], dtype=torch.float64)      # (cross-modal experts have the ReLU: net_trans_v3.py:398)
    zp = Z * bn1[2] + bn1[3]
    zp = torch.where(relu.view(1, 1, 1, E, 1) > 0, zp.clamp_min(0.0), zp)
    So = torch.einsum("snged,ged->sne", zp, us) + H1
    t1, t2 = torch.einsum("snged,gedf,sngef->sne", zp, Gq, zp), 2.0 * torch.einsum("snged,ged->sne", zp, vh)
    Soo = t1 + t2 + H2
    mup = So / cfg.Cx
    var = (Soo / cfg.Cx - mup * mup).clamp_min(0.0)
    rp_ref = torch.rsqrt(var + cfg.ln_eps).permute(2, 0, 1)
    # Soo = || Wh z' + h2 ||^2 with Wh z' ~ -h2 (BatchNorm-2 removes the large mean): the three terms cancel to Soo / amp of their size
    amp = float(((t1.abs() + t2.abs() + H2.abs()) / Soo.abs()).max())
    err = float(((rp - rp_ref).abs() / rp_ref).max())
    assert amp > 50.0, amp                                    # (the stress case is one)
    # 2^-17 per product (two bf16 planes, lo . lo dropped), amplified: rp = Soo^-1/2 takes half of Soo's relative error.  Measured round 6:
    # rp off by 0.9 % at this setting -- in a regime where bf16 storage of z' itself (spacing 0.031 at 4.0) has long destroyed the variance
    assert err <= 2e-5 * amp + 1e-4, (err, amp)
    assert torch.equal(run.idx.cpu(), fwd["idx"])
    out = run.out.float().cpu()
    assert torch.isfinite(out).all() and all(torch.isfinite(v).all() for v in got.values())
    # against the oracle: bounded by what bf16 storage does to THESE activations in the reference formulation itself (bf16 spacing at 4.0 is
    # 0.031 against a spread of 0.05): the oracle under autocast on the same inputs
    dev = torch.device("cuda:0")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        fe, _ = O.moe_forward_backward({k: v.to(dev) for k, v in P.items()}, {k: v.to(dev) for k, v in B.items()}, X.to(dev), Y.to(dev), cfg, G.to(dev), training=True)
    e_hip = float((out - fwd["out"]).norm() / fwd["out"].norm())
    e_eager = float((fe["out"].float().cpu() - fwd["out"]).norm() / fwd["out"].norm())
    assert e_hip <= max(1e-2, 2.0 * e_eager), (e_hip, e_eager)


@pytest.mark.parametrize("site", ["audio", "visual"])
def test_scalar_gate_gradients_over_draws_of_the_upstream_gradient(site, avmoe_hooks):
    """The one-number gradients (`gate`, `gate_av` of every expert; the benchmarked cfg-2 site shapes at B = 2 clips, bf16, the streaming kernels
    forced) judged WITHOUT the ratio-of-two-random-sums tail: <G, y> for a random G has value and rounding error both zero-mean over the
    same ~1e7 terms, so the relative error of ONE draw is heavy-tailed (10 - 50 x outliers on a few of ~100 scalars are chance; the reference
    formulation under autocast shows the same: profiles/r06_gate_grad_draws.txt).  Over 6 draws of G with the forward fixed,

        eps = rms_k(hip_k - oracle_k) / rms_k(oracle_k)       (oracle: fp32 on the bf16-rounded inputs, the HIP path's ReLU mask)

    Bars: eps <= 1.5 % for the output gates (formed in fp32 in weight space since round 6; measured 0.2 - 0.8 %, eager autocast 0.3 - 0.9 %),
    <= 4 % for gate_av (measured 0.5 - 1.2 %; eager autocast 4 - 9 %), and never above 1.5 x the eager-autocast eps of the same tensor + 0.5 %."""
    from tests.moe_gpu_util import MoeRun
    from avmoe_amd import debug as dbg
    avmoe_hooks(1 | 2 | 4 | 8)
    dims = {"audio": dict(Cx=768, Nx=1024, Cy=768, Ny=196), "visual": dict(Cx=768, Nx=196, Cy=768, Ny=1024)}[site]
    cfg = O.AdapterConfig(**dims, reduction=12, groups=2, K=32, E_m=2, E_s=2)
    P, B = O.init_params(cfg, seed=5)
    g = torch.Generator().manual_seed(99)
    S, D = 20, 6
    X = (0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g)).bfloat16().float()
    Y = (0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)).bfloat16().float()
    Gs = [torch.randn(S, cfg.Nx, cfg.Cx, generator=g).bfloat16().float() for _ in range(D)]
    keys = [k for k, v in P.items() if v.numel() == 1]
    assert len(keys) >= 6          # four output gates + the cross-modal experts' gate_av
    run = MoeRun(cfg, P, B, X, Y, bf16=True, training=True).forward()
    masks = dbg.relu_masks_of(run.desc, run.saved)
    refs = O.moe_grads_over_draws(P, B, X, Y, cfg, Gs, keys, training=True, relu_masks=masks)
    dev = torch.device("cuda:0")
    Pd, Bd = {k: v.to(dev) for k, v in P.items()}, {k: v.to(dev) for k, v in B.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16):          # the reference formulation itself in bf16 (own mask)
        eag = O.moe_grads_over_draws(Pd, Bd, X.to(dev), Y.to(dev), cfg, [G.to(dev) for G in Gs], keys, training=True)
    acc = {k: [0.0, 0.0, 0.0] for k in keys}
    for G, r, e in zip(Gs, refs, eag):
        got = run.backward(G)
        for k in keys:
            acc[k][0] += float((got[k] - r[k]).pow(2).sum()); acc[k][1] += float((e[k].float().cpu() - r[k]).pow(2).sum()); acc[k][2] += float(r[k].pow(2).sum())
    assert run.guards_intact()
    bad = {}
    for k, (sh, se, sr) in acc.items():
        eh, ee = (sh / sr) ** 0.5, (se / sr) ** 0.5
        bar = 4e-2 if k.endswith("gate_av") else 1.5e-2
        if eh > bar or eh > 1.5 * ee + 5e-3:
            bad[k] = (eh, ee)
    assert not bad, f"(eps hip, eps eager autocast): {bad}"


@pytest.mark.parametrize("reductions,em_b", [((6, 3), 2), ((8, 8), 2), ((6, 3), 0), ((6, 3), -1)])      # bottleneck 16 per group ; 12 / 6 per group (merged-group weights, moe_plan.h: mg) ; site B without a cross-modal expert (its dY is the wbar row alone: no fourth segment)
def test_fp32_site_pair_writes_each_token_gradient_once(reductions, em_b):
    """Round 6: fp32 site pairs of ANY shape get `token gradient = this site's dX + the other site's dY` from ONE engine product (the other site's
    [Bm ; wbar]^T dV + dR^T Q as a third and fourth K segment: avmoe_moe_backward_dx_dy on the tiled engine) -- it runs (profiler family), and
    the gradients equal those of the two sites called one after the other (overwrite, then read back + add) to summation-order noise."""
    from avmoe_amd.adapters import AdapterPair
    from avmoe_amd import _capi
    from tests.test_adapters_gpu import build_module
    dev = torch.device("cuda:0")
    em_a = 0 if em_b < 0 else 2          # (-1: site A is the one without a cross-modal expert -- its dX product has no latent-token segment to speak of)
    em_b = 2 if em_b < 0 else em_b
    ca = O.AdapterConfig(Cx=192, Nx=300, Cy=96, Ny=77, reduction=reductions[0], groups=2, K=12, E_m=em_a, E_s=2 if em_a else 3)
    cb = O.AdapterConfig(Cx=96, Nx=77, Cy=192, Ny=300, reduction=reductions[1], groups=2, K=12, E_m=em_b, E_s=2 if em_b else 3)
    torch.manual_seed(4)
    sa, sb = build_module("ave", ca).to(dev).train(), build_module("ave", cb).to(dev).train()
    with torch.no_grad():
        for m in (sa, sb):
            for k, p in m.named_parameters():
                if k.endswith(("gate", "gate_av")):
                    p.fill_(0.4)
    g = torch.Generator().manual_seed(10)
    S = 6
    fa, fv = (0.5 * torch.randn(S, ca.Cx, ca.Nx, 1, generator=g)).to(dev), (0.5 * torch.randn(S, cb.Cx, cb.Nx, 1, generator=g)).to(dev)
    ga, gv = torch.randn(S, ca.Cx, ca.Nx, 1, generator=g).to(dev), torch.randn(S, cb.Cx, cb.Nx, 1, generator=g).to(dev)
    bufs = [{k: b.clone() for k, b in m.named_buffers()} for m in (sa, sb)]

    def run(paired):
        for m, bb in zip((sa, sb), bufs):
            m.zero_grad()
            m.load_state_dict({**m.state_dict(), **bb})
        xa, xv = fa.clone().requires_grad_(True), fv.clone().requires_grad_(True)
        if paired:
            oa, _ia, ov, _iv = AdapterPair(sa, sb, concurrent=True)(xa, xv)
        else:
            (oa, _ia), (ov, _iv) = sa(xa, xv), sb(xv, xa)
        torch.autograd.backward([oa, ov], [ga, gv])
        torch.cuda.synchronize()
        return xa.grad, xv.grad, [p.grad.clone() for m in (sa, sb) for p in m.parameters()]

    ref = run(False)
    L = _capi.lib()
    L.avmoe_prof_reset(); L.avmoe_prof_enable(1)
    try:
        got = run(True)
        ran = [f["name"] for f in _capi.prof_report()]
    finally:
        L.avmoe_prof_enable(0); L.avmoe_prof_reset()
    assert sum("+KM+MM+KM" in n for n in ran) >= 1, ran
    for r_, g_ in ((ref[0], got[0]), (ref[1], got[1])):
        assert float((r_ - g_).abs().max()) <= 2e-6 * float(r_.abs().max())
    for r_, g_ in zip(ref[2], got[2]):
        assert torch.equal(r_, g_)
