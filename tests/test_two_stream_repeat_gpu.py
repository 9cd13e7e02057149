"""Two-stream mode must not change a single bit: AdapterPair(concurrent=True) repeated on identical inputs, and against the same pair
on one stream (forward).  Round 4 found v_mfma_f32_16x16x4_f32 returning wrong sums in the bottleneck-space kernels of bf16 sites
whenever a bf16 GEMM of the engine ran on the same compute units from the other stream (one 16-token tile in ~10^3, off by ~1 %; about
every second step at the Swin-L / HTS-AT stage-0 shapes; never on one stream): the mat-vecs of the bf16 instantiations now run on the
bf16 matrix pipe in split form (csrc/tile_gen.inc::mmT_split).  This is the guard: the stage-0 site shape of BASELINE configs 3 / 4
(C 96 / 192, 4096 / 2304 tokens, 20 frames), forward + backward, workspaces poisoned."""
import pytest
import torch

from oracle import avmoe_oracle as O

pytestmark = pytest.mark.gpu


def _pair(dev, concurrent, Pa, Ba, Pv, Bv, ca, cv):
    from tests.test_adapters_api import build_module
    from avmoe_amd.adapters import AdapterPair
    ma, mv = build_module("ave", ca).to(dev).train(), build_module("ave", cv).to(dev).train()
    ma.load_state_dict({**Pa, **Ba}); mv.load_state_dict({**Pv, **Bv})
    return ma, mv, AdapterPair(ma, mv, concurrent=concurrent)


SHAPES = {
    # the stage-0 shape of BASELINE configs 3 / 4 (generalised kernels, merged groups): moved every second step before the fix
    "stage0_swinl_htsat": (dict(reduction=8, groups=2, K=32, E_m=2, E_s=2), (96, 4096), (192, 2304), 20),
    # stage 2 / 3 shapes at 8 clips (generalised kernels, small grids): 4 - 9 of 9 steps moved before the CU-exclusive launches
    "stage2_swinl_htsat_8clips": (dict(reduction=8, groups=2, K=32, E_m=2, E_s=2), (384, 256), (768, 144), 80),
    "stage3_swinl_htsat_8clips": (dict(reduction=8, groups=2, K=32, E_m=2, E_s=2), (768, 64), (1536, 36), 80),
    # the benchmarked shape (tuned register-resident kernels): has never moved -- kept as a guard
    "cfg2": (dict(reduction=12, groups=2, K=32, E_m=2, E_s=2), (768, 1024), (768, 196), 20),
    # fp32, the mode BASELINE config 1 is benchmarked in (two streams, Swin-B x HTS-AT): stage 0 and stage 2
    "f32_stage0_swinb_htsat": (dict(reduction=8, groups=2, K=32, E_m=2, E_s=2), (96, 4096), (128, 2304), 20, torch.float32),
    "f32_stage2_swinb_htsat_8clips": (dict(reduction=8, groups=2, K=32, E_m=2, E_s=2), (384, 256), (512, 144), 80, torch.float32),
    # three groups: no register-resident instance -- the any-shape kernels of csrc/tile_kernels.hip (fp32 matrix pipe with LDS operands),
    # which take the CU-exclusive launch since round 5
    "anyshape_3groups": (dict(reduction=8, groups=3, K=32, E_m=2, E_s=2), (96, 4096), (192, 2304), 20),
    "anyshape_3groups_f32_8clips": (dict(reduction=8, groups=3, K=32, E_m=2, E_s=2), (384, 256), (768, 144), 80, torch.float32),
}


@pytest.mark.timeout(900)
@pytest.mark.parametrize("shape", list(SHAPES))
def test_two_stream_pair_repeats_bit_for_bit(shape):
    from avmoe_amd.adapters import release_workspaces
    dev = torch.device("cuda:0")
    kw, (Ca, Na), (Cv, Nv), S, *rest = SHAPES[shape]
    dt = rest[0] if rest else torch.bfloat16
    ca = O.AdapterConfig(Cx=Ca, Nx=Na, Cy=Cv, Ny=Nv, **kw)
    cv = O.AdapterConfig(Cx=Cv, Nx=Nv, Cy=Ca, Ny=Na, **kw)
    Pa, Ba = O.init_params(ca, seed=0)
    Pv, Bv = O.init_params(cv, seed=1)
    g = torch.Generator().manual_seed(1234)
    fa, fv = 0.3 * torch.randn(S, ca.Nx, ca.Cx, generator=g), 0.3 * torch.randn(S, cv.Nx, cv.Cx, generator=g)
    ga, gv = torch.randn(fa.shape, generator=g).to(dt), torch.randn(fv.shape, generator=g).to(dt)

    def run(concurrent):
        release_workspaces()
        torch.cuda.empty_cache()
        junk = torch.full((1 << 30,), 0xFF, dtype=torch.uint8, device=dev)      # the next workspaces come out of NaN-patterned memory
        torch.cuda.synchronize()
        del junk
        ma, mv, pair = _pair(dev, concurrent, Pa, Ba, Pv, Bv, ca, cv)
        xa = fa.to(dev, dt).requires_grad_(True)
        xv = fv.to(dev, dt).requires_grad_(True)
        out_a, _ia, out_v, _iv = pair(xa.permute(0, 2, 1).unsqueeze(-1), xv.permute(0, 2, 1).unsqueeze(-1))
        torch.autograd.backward([out_a, out_v], [ga.to(dev).permute(0, 2, 1).unsqueeze(-1), gv.to(dev).permute(0, 2, 1).unsqueeze(-1)])
        torch.cuda.synchronize()
        res = {"out_a": out_a.detach().float().cpu(), "out_v": out_v.detach().float().cpu(), "d_fa": xa.grad.float().cpu(), "d_fv": xv.grad.float().cpu()}
        for tag, m in (("a", ma), ("v", mv)):
            res.update({f"{tag}.{k}": p.grad.float().cpu() for k, p in m.named_parameters()})
        return res

    ref = run(True)
    assert all(torch.isfinite(v).all() for v in ref.values())
    for it in range(10):         # (a longer guard than five repetitions)
        got = run(True)
        moved = [k for k, v in got.items() if not torch.equal(v, ref[k])]
        assert not moved, f"two-stream run {it + 1} differs from run 0 in {moved[:8]}"
    # one stream: the same arithmetic up to the summation order of the BatchNorm column sums (with other kernels on the GPU the
    # bottleneck-space kernels run one eight-wave block per CU instead of four-wave blocks: avmoe_moe_desc::shared_gpu)
    one = run(False)
    for k in ("out_a", "out_v"):
        assert float((one[k] - ref[k]).abs().max()) <= 1e-2 * float(ref[k].abs().max()), k
    two_of_one = run(False)
    assert all(torch.equal(v, one[k]) for k, v in two_of_one.items())
