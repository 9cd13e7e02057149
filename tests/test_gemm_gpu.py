"""GPU parity of the generic MFMA GEMM engine (avmoe_gemm) against torch.matmul in fp64 on the same
inputs: all four operand-layout combinations, both dtypes, both tiles, ragged M/N/K, batching with
broadcast operands, transposed C, accumulate, the row_scale*D epilogue and split-K."""
import ctypes as C
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_gemm(M, N, K, dtype, a_mn, b_mn, nb1=1, nb2=1, tile=0, c_transposed=False, out_bf16=False,
              accumulate=False, epilogue=False, ksplit=1, bcast_a=False, seed=0, exact_ints=False, bcast_b=False, planes=0, wide_range=False):
    from avmoe_amd import _capi as capi
    L = capi.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seed)
    tdt = torch.bfloat16 if dtype == capi.BF16 else torch.float32
    epc = 8 if dtype == capi.BF16 else 4
    nb = nb1 * nb2

    def rnd(*shape):
        if exact_ints:
            return torch.randint(-3, 4, shape, generator=g).double()
        x = torch.randn(*shape, generator=g, dtype=torch.float64)
        if wide_range:                      # magnitudes over 2^-20 .. 2^20: every plane of the three-plane form carries bits somewhere
            x = x * torch.exp2(torch.randint(-20, 21, shape, generator=g).double())
        return x

    Kp, Mp, Np = -(-K // epc) * epc, -(-M // epc) * epc, -(-N // epc) * epc
    A = rnd(1 if bcast_a else nb, M, K)
    Bm = rnd(1 if bcast_b else nb, N, K)
    # device storage with padded leading dims; padding holds finite garbage (7.0) except A's K padding
    if a_mn:
        Ad = torch.full((A.shape[0], K, Mp), 7.0, dtype=tdt)
        Ad[:, :, :M] = A.transpose(1, 2).to(tdt)
        lda, sA = Mp, K * Mp
    else:
        Ad = torch.zeros((A.shape[0], M, Kp), dtype=tdt)
        Ad[:, :, :K] = A.to(tdt)
        lda, sA = Kp, M * Kp
    if b_mn:
        Bd = torch.full((Bm.shape[0], K, Np), 7.0, dtype=tdt)
        Bd[:, :, :N] = Bm.transpose(1, 2).to(tdt)
        ldb, sB = Np, K * Np
    else:
        Bd = torch.full((Bm.shape[0], N, Kp), 7.0, dtype=tdt)      # garbage in the K padding: the engine masks it
        Bd[:, :, :K] = Bm.to(tdt)
        ldb, sB = Kp, N * Kp
    Ad, Bd = Ad.to(dev), Bd.to(dev)
    Ar = (Ad[:, :, :M].transpose(1, 2) if a_mn else Ad[:, :, :K]).double().cpu()
    Br = (Bd[:, :, :N].transpose(1, 2) if b_mn else Bd[:, :, :K]).double().cpu()
    ref = 0.5 * torch.matmul(Ar.expand(nb, M, K), Br.expand(nb, N, K).transpose(1, 2))

    odt = torch.bfloat16 if out_bf16 else torch.float32
    Cd = torch.zeros((nb, N, M) if c_transposed else (nb, M, N), dtype=odt)
    if accumulate:
        C0 = torch.randn(Cd.shape, generator=g).to(odt)
        Cd.copy_(C0)
        ref = ref + (C0.double().transpose(1, 2) if c_transposed else C0.double())
    Cd = Cd.to(dev)
    rs = Dd = None
    if epilogue:
        rs = torch.randn(nb, M, generator=g, dtype=torch.float32)
        Dm = torch.randn(nb, M, N, generator=g).to(tdt)
        ref = ref + rs.double()[:, :, None] * Dm.double()
        rs, Dd = rs.to(dev), Dm.to(dev)

    d = capi.GemmDesc()
    d.M, d.N, d.K, d.nb1, d.nb2 = M, N, K, nb1, nb2
    d.dtype, d.out_dtype = dtype, (capi.BF16 if out_bf16 else capi.F32)
    d.a_layout, d.b_layout = int(a_mn), int(b_mn)
    d.accumulate, d.ksplit, d.tile, d.alpha = int(accumulate), ksplit, tile, 0.5
    d.fp32_planes = planes
    d.lda, d.ldb = lda, ldb
    d.sA1, d.sA2 = (0, 0) if bcast_a else (sA * nb2, sA)
    d.sB1, d.sB2 = (0, 0) if bcast_b else (sB * nb2, sB)
    if c_transposed:
        d.sCi, d.sCj = 1, M
    else:
        d.sCi, d.sCj = N, 1
    d.sC1, d.sC2 = M * N * nb2, M * N
    d.sRS1, d.sRS2, d.sDi, d.sD1, d.sD2 = M * nb2, M, N, M * N * nb2, M * N
    ws = None
    nbytes = L.avmoe_gemm_workspace_bytes(C.byref(d))
    if nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = L.avmoe_gemm(C.byref(d), Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(),
                      rs.data_ptr() if rs is not None else None, Dd.data_ptr() if Dd is not None else None,
                      ws.data_ptr() if ws is not None else None, torch.cuda.current_stream().cuda_stream)
    capi.check(st, "avmoe_gemm")
    torch.cuda.synchronize()
    got = Cd.double().cpu()
    if c_transposed:
        got = got.transpose(1, 2)
    return got, ref


def _tol(dtype):
    return 2e-5 if dtype == 0 else 1e-4     # bf16 inputs are exact in the fp64 reference; fp32 accumulate


LAYOUTS = list(itertools.product([False, True], [False, True]))


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("a_mn,b_mn", LAYOUTS)
def test_exact_integer_tiles_asymmetric(dtype, a_mn, b_mn):
    """Small-integer operands: every product and partial sum is exact, so any fragment-layout slip
    (row/col swap, k permutation mismatch between operands) shows up as a hard mismatch."""
    for tile in (32, 64, 128):
        got, ref = _run_gemm(80, 48, 96, dtype, a_mn, b_mn, tile=tile, exact_ints=True, seed=tile)
        assert torch.equal(got, ref), f"tile {tile}: max err {(got - ref).abs().max()}"


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("a_mn,b_mn", LAYOUTS)
@pytest.mark.parametrize("shape", [(200, 136, 196), (64, 24, 40), (333, 70, 1030), (16, 16, 8)])
def test_ragged_shapes(dtype, a_mn, b_mn, shape):
    M, N, K = shape
    got, ref = _run_gemm(M, N, K, dtype, a_mn, b_mn, seed=M + N + K)
    err = (got - ref).abs().max() / (ref.abs().max() + 1e-9)
    assert err < _tol(dtype), float(err)


@pytest.mark.parametrize("dtype", [0, 1])
def test_batched_broadcast_and_two_level_batch(dtype):
    got, ref = _run_gemm(72, 40, 64, dtype, False, True, nb1=3, nb2=2, bcast_a=True, seed=5)
    assert (got - ref).abs().max() / ref.abs().max() < _tol(dtype)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("tile", [32, 64, 128])
def test_transposed_c_accumulate_epilogue(dtype, tile):
    got, ref = _run_gemm(150, 90, 72, dtype, False, False, nb1=2, tile=tile, c_transposed=True,
                         accumulate=True, seed=7)
    assert (got - ref).abs().max() / ref.abs().max() < _tol(dtype)
    got, ref = _run_gemm(150, 92, 72, dtype, True, False, nb1=2, tile=tile, accumulate=True, epilogue=True, seed=8)
    assert (got - ref).abs().max() / ref.abs().max() < _tol(dtype)
    got, ref = _run_gemm(150, 92, 72, dtype, False, True, tile=tile, out_bf16=True, accumulate=True,
                         epilogue=True, seed=9)
    assert (got - ref).abs().max() / ref.abs().max() < 2e-2


@pytest.mark.parametrize("dtype", [0, 1])
def test_split_k(dtype):
    got, ref = _run_gemm(96, 40, 5000, dtype, True, True, nb1=2, ksplit=7, epilogue=True, accumulate=True, seed=11)
    assert (got - ref).abs().max() / ref.abs().max() < 5 * _tol(dtype)
    got, ref = _run_gemm(70, 130, 700, dtype, False, False, ksplit=3, c_transposed=True, seed=12)
    assert (got - ref).abs().max() / ref.abs().max() < 5 * _tol(dtype)


@pytest.mark.parametrize("a_mn,b_mn", LAYOUTS)
@pytest.mark.parametrize("tile", [32, 64, 128])
def test_fp32_three_plane_form_is_an_fp32_product(a_mn, b_mn, tile):
    """ABI 9 `fp32_planes` (what a site's backward uses for its fp32 products): three bf16 planes per value, six plane products --
    as close to the fp64 product as the fp32 matrix pipe is (both are bounded by the fp32 rounding of the accumulated sum), on
    every tile and layout, for values whose bits spread over all three planes, with exact integers exact, and the same numbers
    whatever tile runs them."""
    got_i, ref_i = _run_gemm(80, 48, 96, 0, a_mn, b_mn, tile=tile, exact_ints=True, seed=tile, planes=1)
    assert torch.equal(got_i, ref_i)
    for shape, kw in (((200, 136, 196), {}), ((333, 70, 1030), {}), ((96, 40, 5000), dict(ksplit=7, nb1=2)), ((150, 92, 72), dict(accumulate=True, epilogue=True))):
        M, N, K = shape
        got3, ref = _run_gemm(M, N, K, 0, a_mn, b_mn, tile=tile, seed=M + K, planes=1, wide_range=True, **kw)
        got1, _ = _run_gemm(M, N, K, 0, a_mn, b_mn, tile=tile, seed=M + K, planes=0, wide_range=True, **kw)
        n = float(ref.norm())
        e3, e1 = float((got3 - ref).norm()) / n, float((got1 - ref).norm()) / n
        assert e3 < 1e-6 and e3 < 2.0 * e1 + 1e-8, (shape, e3, e1)
    g128, _ = _run_gemm(200, 136, 196, 0, a_mn, b_mn, tile=128, seed=3, planes=1)
    gt, _ = _run_gemm(200, 136, 196, 0, a_mn, b_mn, tile=tile, seed=3, planes=1)
    assert torch.equal(g128, gt)


@pytest.mark.parametrize("a_mn,b_mn", LAYOUTS)
@pytest.mark.parametrize("tile", [32, 64, 128])
def test_fp32_two_plane_form(a_mn, b_mn, tile):
    """ABI 11 `fp32_planes = 2` (csrc/gemm.hip: f32s2 -- two bf16 planes per value, split once on the way into the LDS, three plane products):
    2^-16 relative per product, exact on integers that fit two planes, every tile and layout, ragged shapes, split K, epilogue; the same numbers
    whatever tile runs them.  (Offered through avmoe_gemm; the site calls keep three planes: csrc/moe_run.h.)"""
    got_i, ref_i = _run_gemm(80, 48, 96, 0, a_mn, b_mn, tile=tile, exact_ints=True, seed=tile, planes=2)
    assert torch.equal(got_i, ref_i)
    for shape, kw in (((200, 136, 196), {}), ((333, 70, 1030), {}), ((96, 40, 5000), dict(ksplit=7, nb1=2)), ((150, 92, 72), dict(accumulate=True, epilogue=True))):
        M, N, K = shape
        got2, ref = _run_gemm(M, N, K, 0, a_mn, b_mn, tile=tile, seed=M + K, planes=2, **kw)
        e2 = float((got2 - ref).norm()) / float(ref.norm())
        assert e2 < 3e-5, (shape, e2)
    g128, _ = _run_gemm(200, 136, 196, 0, a_mn, b_mn, tile=128, seed=3, planes=2)
    gt, _ = _run_gemm(200, 136, 196, 0, a_mn, b_mn, tile=tile, seed=3, planes=2)
    assert torch.equal(g128, gt)


def test_alignment_contract_is_enforced():
    from avmoe_amd import _capi as capi
    L = capi.lib()
    d = capi.GemmDesc()
    d.M = d.N = d.K = 16
    d.nb1 = d.nb2 = 1
    d.lda = d.ldb = 18           # 72-byte rows: not a multiple of 16
    d.sCi, d.sCj, d.alpha = 16, 1, 1.0
    x = torch.zeros(1024, device="cuda:0")
    st = L.avmoe_gemm(C.byref(d), x.data_ptr(), x.data_ptr(), x.data_ptr(), None, None, None, None)
    assert st == -3 and b"alignment" in L.avmoe_last_error()


def test_gemm_throughput_report(capsys):
    """Not a pass/fail perf gate -- prints achieved TFLOP/s for the path's main GEMM shapes (cfg-2)."""
    from avmoe_amd import _capi as capi
    L = capi.lib()
    dev = torch.device("cuda:0")
    rows = []
    for (name, M, N, K, nb, a_mn, b_mn) in [
        ("X@[W|T] tokens x C -> 320", 327680, 320, 768, 1, False, False),
        ("post: tokens x 160 -> C", 327680, 768, 160, 1, False, False),
        ("token contraction (per sample)", 64, 768, 1024, 320, True, True),
        ("square 4096", 4096, 4096, 4096, 1, False, False),
    ]:
        for dtype in (capi.BF16, capi.F32):
            tdt = torch.bfloat16 if dtype == capi.BF16 else torch.float32
            A = torch.randn((nb, K, M) if a_mn else (nb, M, K), device=dev).to(tdt)
            B = torch.randn((nb, K, N) if b_mn else (nb, N, K), device=dev).to(tdt)
            Cc = torch.empty((nb, M, N), device=dev, dtype=tdt)
            d = capi.GemmDesc()
            d.M, d.N, d.K, d.nb1, d.nb2 = M, N, K, nb, 1
            d.dtype = d.out_dtype = dtype
            d.a_layout, d.b_layout = int(a_mn), int(b_mn)
            d.ksplit, d.alpha = 1, 1.0
            d.lda, d.ldb = (M if a_mn else K), (N if b_mn else K)
            d.sA1, d.sB1, d.sC1 = M * K, N * K, M * N
            d.sCi, d.sCj = N, 1
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(2):
                capi.check(L.avmoe_gemm(C.byref(d), A.data_ptr(), B.data_ptr(), Cc.data_ptr(), None, None, None, st))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                capi.check(L.avmoe_gemm(C.byref(d), A.data_ptr(), B.data_ptr(), Cc.data_ptr(), None, None, None, st))
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            fl = 2.0 * M * N * K * nb
            by = (A.numel() + B.numel() + Cc.numel()) * A.element_size()
            rows.append(f"{name:34s} {'bf16' if dtype else 'f32 '} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s  "
                        f"{by / ms / 1e6:8.1f} GB/s")
    with capsys.disabled():
        print("\n[gemm throughput]\n" + "\n".join(rows))


@pytest.mark.parametrize("case", [
    dict(M=8200, N=384, K=140, b_mn=False, nb2=2, out_bf16=True),                   # output GEMM class (K tail 12, bf16 C)
    dict(M=9001, N=128, K=384, b_mn=False, nb2=2),                                  # grouped down projection class
    dict(M=8197, N=140, K=384, b_mn=True, nb2=2),                                   # dApost class (B MN-major, N tail)
    dict(M=8300, N=100, K=70, b_mn=True, nb2=1, epilogue=True, out_bf16=True),      # row-scale * D epilogue
    dict(M=8193, N=12, K=8, b_mn=False, nb2=3, epilogue=True),                      # tiny N / K
])
def test_streaming_kernel_shapes(case):
    """M >= 256 with N, K <= 384 and K-major bf16 A is served by the B-stationary streaming kernel (gemm_stream.hip);
    the ragged M / N / K tails, both B layouts, both output types and the epilogue are checked against fp64."""
    case = dict(case)
    out_bf16 = case.get("out_bf16", False)
    got, ref = _run_gemm(case.pop("M"), case.pop("N"), case.pop("K"), 1, False, case.pop("b_mn"), seed=13, **case)
    err = (got - ref).abs().max() / ref.abs().max()
    assert err < (2e-2 if out_bf16 else 1e-4), float(err)
    ints, ref = _run_gemm(8200, 144, 96, 1, False, True, nb2=2, exact_ints=True, seed=3)
    assert torch.equal(ints, ref)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("M,rps,c_transposed", [(64, 65, False), (65, 65, False), (64, 65, True), (32, 33, True), (48, 48, False), (40, 41, False), (8, 9, False)])
def test_batch_fold_shared_b(dtype, M, rps, c_transposed):
    """Per-sample row blocks of A against ONE shared B run as a single tall GEMM (gemm.hip "batch fold", the latent x
    token-remap products of moe_forward.cpp / moe_backward.cpp): the rows between the blocks hold NaN and must neither leak
    into the result nor be stored -- C's gap rows keep their sentinel."""
    from avmoe_amd import _capi as capi
    L = capi.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11 + M + rps)
    tdt = torch.bfloat16 if dtype == capi.BF16 else torch.float32
    nb, N, K = 7, 197, 100
    Kp = 104
    A = torch.full((nb, rps, Kp), float("nan"), dtype=tdt)
    A[:, :M, :] = 0
    A[:, :M, :K] = torch.randn(nb, M, K, generator=g).to(tdt)
    Bm = torch.full((N, Kp), 7.0, dtype=tdt)
    Bm[:, :K] = torch.randn(N, K, generator=g).to(tdt)
    ref = 0.5 * torch.matmul(A[:, :M, :K].double(), Bm[:, :K].double().t())            # (nb, M, N)
    Nld, Mld = 200, 72
    Cd = torch.full((nb, Nld, Mld) if c_transposed else (nb, rps, Nld), -5.0, dtype=torch.float32)
    Ad, Bd, Cd = A.to(dev), Bm.to(dev), Cd.to(dev)
    d = capi.GemmDesc()
    d.M, d.N, d.K, d.nb1, d.nb2 = M, N, K, nb, 1
    d.dtype, d.out_dtype, d.a_layout, d.b_layout = dtype, capi.F32, 0, 0
    d.accumulate, d.ksplit, d.tile, d.alpha = 0, 1, 0, 0.5
    d.lda, d.ldb, d.sA1, d.sA2, d.sB1, d.sB2 = Kp, Kp, rps * Kp, 0, 0, 0
    if c_transposed:
        d.sCi, d.sCj, d.sC1 = 1, Mld, Nld * Mld
    else:
        d.sCi, d.sCj, d.sC1 = Nld, 1, rps * Nld
    st = L.avmoe_gemm(C.byref(d), Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), None, None, None, torch.cuda.current_stream().cuda_stream)
    capi.check(st, "avmoe_gemm")
    torch.cuda.synchronize()
    got = Cd.double().cpu()
    if c_transposed:
        val, rest = got[:, :N, :M].transpose(1, 2), torch.cat([got[:, N:, :].reshape(-1), got[:, :N, M:].reshape(-1)])
    else:
        val, rest = got[:, :M, :N], torch.cat([got[:, M:, :].reshape(-1), got[:, :M, N:].reshape(-1)])
    assert torch.isfinite(val).all()
    assert float((val - ref).abs().max()) <= _tol(dtype) * float(ref.abs().max())
    assert bool((rest == -5.0).all())


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("a_mn,b_mn", LAYOUTS)
def test_small_batched_problems_take_the_32_tile(dtype, a_mn, b_mn):
    """M, N <= 32 per batch entry (K x K latent matrices, S x S frame attention of the "v1" experts): the 32 x 32 tile, all
    layouts, two-level batch, ragged extents, split-K."""
    for (M, N, K, ks) in ((10, 10, 16, 1), (32, 32, 200, 1), (10, 16, 10, 1), (24, 31, 1000, 4)):
        got, ref = _run_gemm(M, N, K, dtype, a_mn, b_mn, nb1=3, nb2=4, ksplit=ks, seed=M + N)
        assert float((got - ref).abs().max()) <= _tol(dtype) * float(ref.abs().max()), (M, N, K, ks)


@pytest.mark.parametrize("case", [
    dict(M=8200, N=48, K=140, b_mn=False, nb2=2, out_bf16=True),                    # output GEMM into 48 channels per group
    dict(M=8200, N=64, K=70, b_mn=False, nb2=2, out_bf16=True, accumulate=True),    # ... accumulating (residual stream)
    dict(M=9001, N=128, K=48, b_mn=False, nb2=2),                                   # down projection from 48 channels per group
    dict(M=8197, N=140, K=64, b_mn=True, nb2=2),                                    # dApost from 64 channels per group
    dict(M=8193, N=12, K=8, b_mn=True, nb2=3),                                      # tiny everything
])
def test_streaming_kernel_narrow_channel_groups(case):
    """The narrow-group configurations of the streaming kernel (C / g <= 64: first Swin / HTS-AT stages) against fp64."""
    case = dict(case)
    out_bf16 = case.get("out_bf16", False)
    got, ref = _run_gemm(case.pop("M"), case.pop("N"), case.pop("K"), 1, False, case.pop("b_mn"), seed=17, **case)
    err = (got - ref).abs().max() / ref.abs().max()
    assert err < (2e-2 if out_bf16 else 1e-4), float(err)


@pytest.mark.parametrize("a_mn,b_mn", LAYOUTS)
def test_big_tile_exact_integers(a_mn, b_mn):
    """The 256 x 256 tile (large plain bf16 products; tile = 0 picks it): small-integer operands make every product and partial sum
    exact, so a fragment-layout slip in the 2 x 4 wave grid or the two-half epilogue is a hard mismatch.  Ragged M / N / K."""
    got, ref = _run_gemm(2100, 2700, 300, 1, a_mn, b_mn, exact_ints=True, seed=5)
    assert torch.equal(got, ref), f"max err {(got - ref).abs().max()}"


@pytest.mark.parametrize("a_mn,b_mn", LAYOUTS)
@pytest.mark.parametrize("kw", [dict(), dict(out_bf16=True), dict(accumulate=True), dict(c_transposed=True), dict(accumulate=True, out_bf16=True),
                                dict(ksplit=3), dict(nb1=2, nb2=2), dict(c_transposed=True, accumulate=True)])
def test_big_tile_variants(a_mn, b_mn, kw):
    M, N, K = (1100, 1300, 520) if kw.get("nb1") else (2300, 2052, 1030)
    got, ref = _run_gemm(M, N, K, 1, a_mn, b_mn, seed=11, **kw)
    err = (got - ref).abs().max() / (ref.abs().max() + 1e-9)
    assert err < (1e-2 if kw.get("out_bf16") else 1e-4), float(err)


def test_big_tile_throughput():
    """(informational) the large-product tile against the 128 x 128 one on a stage-0 remap shape: 4096 x 3136 x 15400"""
    import time
    from avmoe_amd import _capi as capi
    L = capi.lib()
    dev = torch.device("cuda:0")
    M, N, K = 4096, 3136, 15400
    A = torch.randn(M, K, device=dev).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    Cd = torch.empty(M, N, device=dev)
    for tile in (128, 0):
        d = capi.GemmDesc()
        d.M, d.N, d.K, d.nb1, d.nb2 = M, N, K, 1, 1
        d.dtype, d.out_dtype, d.a_layout, d.b_layout = capi.BF16, capi.F32, 0, 0
        d.accumulate, d.ksplit, d.tile, d.alpha = 0, 1, tile, 1.0
        d.lda, d.ldb, d.sCi, d.sCj = K, K, N, 1
        run = lambda: capi.check(L.avmoe_gemm(C.byref(d), A.data_ptr(), B.data_ptr(), Cd.data_ptr(), None, None, None,
                                              torch.cuda.current_stream().cuda_stream), "gemm")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"\n[gemm {M}x{N}x{K} bf16] tile {'256 (auto)' if tile == 0 else tile}: {dt * 1e3:.3f} ms  {2.0 * M * N * K / dt / 1e12:.0f} TFLOP/s")


@pytest.mark.parametrize("kw", [dict(), dict(out_bf16=True), dict(c_transposed=True)])
def test_batch_fold_onto_the_big_tile(kw):
    """Per-sample row blocks too short for the 256 x 256 tile (384 latent rows per frame) against ONE shared B fold into a tall
    product that takes it (cfg-5: the remap logits of a stage-0 site)."""
    got, ref = _run_gemm(384, 1100, 520, 1, False, False, nb1=6, bcast_b=True, seed=3, **kw)
    err = (got - ref).abs().max() / (ref.abs().max() + 1e-9)
    assert err < (1e-2 if kw.get("out_bf16") else 1e-4), float(err)
    got, ref = _run_gemm(384, 1100, 520, 1, False, False, nb1=6, bcast_b=True, exact_ints=True, seed=4)
    assert torch.equal(got, ref)


FRAME_SHAPES = [  # M, N, K, frames, rows per frame, bf16 output, C stored [n][m]     (K <= 256: B in registers; longer K: few columns only)
    (64, 1024, 196, 40, 65, False, False), (64, 1025, 196, 33, 65, True, False), (64, 1024, 198, 32, 64, False, False), (64, 1024, 197, 48, 65, False, False),
    (64, 200, 256, 40, 65, True, False), (17, 77, 225, 36, 20, False, False), (33, 130, 70, 35, 40, False, False), (1, 64, 32, 32, 1, True, False),
    (48, 321, 135, 700, 50, False, False), (64, 801, 40, 64, 66, False, False),
    (65, 1024, 197, 40, 65, False, False), (65, 130, 100, 33, 65, True, False), (50, 200, 60, 41, 50, False, False),      # frames without gaps: one tall matrix in groups of 64 rows
    (64, 1024, 196, 40, 65, True, True), (64, 1024, 196, 36, 65, False, True), (62, 131, 250, 33, 70, True, True), (5, 64, 33, 32, 8, False, True),
    # K > 256 and N <= 224: the long form (frame pairs x all columns, K chunks streamed)
    (64, 196, 1024, 48, 65, False, False), (64, 197, 1024, 41, 65, True, False), (64, 196, 1026, 36, 65, False, False), (65, 196, 1025, 33, 65, False, False),
    (64, 196, 1024, 40, 65, True, True), (30, 100, 300, 33, 31, False, True), (64, 224, 257, 34, 64, True, False), (17, 64, 700, 35, 17, False, False),
    (64, 130, 264, 33, 66, False, False),
]


@pytest.mark.parametrize("shape", FRAME_SHAPES)
def test_frame_products_against_one_shared_matrix(shape):
    """frame_gemm.hip (the hop-1 chain's middle: L1[s] = [R | qr | qb][s] [Wc | bc | 1]^T and its twins, moe_forward.cpp / moe_backward.cpp):
    a few rows per frame against ONE shared K-major matrix over a short inner dimension.  Rows between the frames' blocks and the K padding
    of A hold NaN, B's K padding garbage: none of it may leak; C outside the (frames, M, N) blocks keeps its sentinel; the kernel family must
    be the one that ran."""
    from avmoe_amd import _capi as capi
    L = capi.lib()
    dev = torch.device("cuda:0")
    M, N, K, nb, rps, out_bf16, tr = shape
    g = torch.Generator().manual_seed(5 + M + N + K)
    Kp = -(-K // 8) * 8
    A = torch.full((nb, rps, Kp), float("nan"), dtype=torch.bfloat16)
    A[:, :M, :K] = torch.randn(nb, M, K, generator=g).to(torch.bfloat16)
    Bm = torch.full((N, Kp), 7.0, dtype=torch.bfloat16)
    Bm[:, :K] = torch.randn(N, K, generator=g).to(torch.bfloat16)
    ref = 0.5 * torch.matmul(A[:, :M, :K].double(), Bm[:, :K].double().t())
    Nld, Mld = -(-N // 8) * 8 + 8, -(-M // 8) * 8 + 8
    odt = torch.bfloat16 if out_bf16 else torch.float32
    Cd = torch.full((nb, N + 3, Mld) if tr else (nb, rps, Nld), -5.0, dtype=odt)
    Ad, Bd, Cd = A.to(dev), Bm.to(dev), Cd.to(dev)
    d = capi.GemmDesc()
    d.M, d.N, d.K, d.nb1, d.nb2 = M, N, K, nb, 1
    d.dtype, d.out_dtype, d.a_layout, d.b_layout = capi.BF16, (capi.BF16 if out_bf16 else capi.F32), 0, 0
    d.accumulate, d.ksplit, d.tile, d.alpha = 0, 1, 0, 0.5
    d.lda, d.ldb, d.sA1, d.sA2, d.sB1, d.sB2 = Kp, Kp, rps * Kp, 0, 0, 0
    if tr:
        d.sCi, d.sCj, d.sC1 = 1, Mld, (N + 3) * Mld
    else:
        d.sCi, d.sCj, d.sC1 = Nld, 1, rps * Nld
    L.avmoe_prof_reset(); L.avmoe_prof_enable(1)
    try:
        st = L.avmoe_gemm(C.byref(d), Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), None, None, None, torch.cuda.current_stream().cuda_stream)
        capi.check(st, "avmoe_gemm")
        torch.cuda.synchronize()
        ran = [f["name"] for f in capi.prof_report()]
    finally:
        L.avmoe_prof_enable(0); L.avmoe_prof_reset()
    assert any(n.startswith("gemm_frames") for n in ran), ran
    got = Cd.double().cpu()
    if tr:
        val, rest = got[:, :N, :M].transpose(1, 2), torch.cat([got[:, N:, :].reshape(-1), got[:, :N, M:].reshape(-1)])
    else:
        val, rest = got[:, :M, :N], torch.cat([got[:, M:, :].reshape(-1), got[:, :M, N:].reshape(-1)])
    assert torch.isfinite(val).all()
    tol = (6e-3 if out_bf16 else 1e-4) * float(ref.abs().max())
    assert float((val - ref).abs().max()) <= tol
    assert bool((rest == -5.0).all())


@pytest.mark.parametrize("a_mn", [False, True])
@pytest.mark.parametrize("case", ["plain", "accumulate_epilogue", "ksplit", "ragged_rows"])
def test_fp32_three_plane_wide_tile_equals_the_square_tiles(a_mn, case):
    """Round 6: the 128 x 160 tile the engine picks by itself for fp32 three-plane products with an MN-major B whose N leaves the last 128-column
    tile mostly empty (the 128 + 3 E = 140 columns of dApost / dBpost) -- the same bits as the 64 x 64 tile on the same operands (same K order per
    output element), every row and column of every tile written exactly once (accumulate + epilogue: a row stored twice would show), ragged
    last row tile, split K."""
    kw = dict(plain={}, accumulate_epilogue=dict(accumulate=True, epilogue=True), ksplit=dict(ksplit=5), ragged_rows={})[case]
    M = 5200 + (37 if case == "ragged_rows" else 0)
    N, K = 140, 96 if case != "ksplit" else 1500
    from avmoe_amd import _capi
    L = _capi.lib()
    L.avmoe_prof_reset(); L.avmoe_prof_enable(1)
    try:
        got, ref = _run_gemm(M, N, K, 0, a_mn, True, nb2=2, tile=0, seed=11, planes=1, **kw)      # (nb2 and M: >= 160 of the 128-row tiles, or the engine steps down to 64 x 64)
        ran = [f["name"] for f in _capi.prof_report()]
    finally:
        L.avmoe_prof_enable(0); L.avmoe_prof_reset()
    assert any("128x160" in n for n in ran), ran
    g64, _ = _run_gemm(M, N, K, 0, a_mn, True, nb2=2, tile=64, seed=11, planes=1, **kw)
    assert torch.equal(got, g64)
    assert float((got - ref).norm() / ref.norm()) < 1e-6


@pytest.mark.parametrize("b_mn", [False, True])
def test_fp32_three_plane_96_row_tile_equals_the_square_tiles(b_mn):
    """Round 6: the 96 x 128 tile the engine picks for fp32 three-plane products with a K-major A whose M leaves a 128-row tile half empty (the
    65 rows of the per-frame hop-1 products): the bits of the 64 x 64 tile, every row written once (accumulate + epilogue)."""
    from avmoe_amd import _capi
    L = _capi.lib()
    for kw in ({}, dict(accumulate=True, epilogue=True)):
        L.avmoe_prof_reset(); L.avmoe_prof_enable(1)
        try:
            got, ref = _run_gemm(65, 256, 200, 0, False, b_mn, nb1=100, tile=0, seed=5, planes=1, **kw)
            ran = [f["name"] for f in _capi.prof_report()]
        finally:
            L.avmoe_prof_enable(0); L.avmoe_prof_reset()
        assert any("_96x128" in n for n in ran), ran
        g64, _ = _run_gemm(65, 256, 200, 0, False, b_mn, nb1=100, tile=64, seed=5, planes=1, **kw)
        assert torch.equal(got, g64)
        assert float((got - ref).norm() / ref.norm()) < 1e-6
