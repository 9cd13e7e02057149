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
