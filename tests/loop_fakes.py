"""Stand-in frozen backbones for the block-loop fixture (tests/golden/blockloop_ave.npz).

oracle/gen_golden_loop.py hangs these on a bare instance of the reference's `MMIL_Net` and runs the reference's OWN
`forward_swin` loop (AVE/nets/net_trans_v3.py:639-727) over them with the reference's MoEAdapter modules at every adapter
site; tests/test_blockloop_golden_gpu.py runs avmoe_amd.blocks.DualBackboneLoop with the HIP sites over the SAME stand-ins.
Every piece is a distinct parameter-free map (so a wrong order of operations changes the result), cheap on the CPU and
deterministic.  Nothing here is reference code: it is the surface the loop touches (timm Swin-V2 block: `_attn, norm1, norm2,
mlp, drop_path1, drop_path2`; HTS-AT block: `blk(x) -> (x, attn)`; stage: `.blocks`, `.downsample`)."""
import torch
from torch import nn

# stage layout: Swin depths (2, 2, 18, 2) vs HTS-AT (2, 2, 6, 2) is what the reference's 18-entry alignment list assumes
# (net_trans_v3.py:675-680); num_skip = 2 leaves adapters in stages 0 and 2 (:687)
DEPTH_V, DEPTH_A = (2, 2, 18, 2), (2, 2, 6, 2)
NUM_SKIP = 2
S, CV, CA, NV0, NA0 = 10, 32, 16, 48, 80         # frames, channels (kept through the stages), stage-0 token counts
REDUCTION, GROUPS, K_TOK, E_M, E_S = 2, 2, 8, 1, 1


class Scale(nn.Module):
    def __init__(self, k):
        super().__init__()
        self.k = k

    def forward(self, x):
        return x * self.k


class VisBlock(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.norm1, self.norm2, self.mlp = Scale(1.0 + 0.01 * n), Scale(1.0 - 0.01 * n), Scale(0.03 + 0.001 * n)
        self.drop_path1, self.drop_path2 = nn.Identity(), nn.Identity()
        self.n = n

    def _attn(self, x):
        return 0.03 * x.roll(1, dims=1) - 0.01 * self.n


class AudBlock(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.n = n

    def forward(self, x):
        return 0.9 * x + 0.02 * self.n + 0.1 * x.flip(1), None


class Halve(nn.Module):
    """Stand-in patch merging: half the tokens, same width."""

    def __init__(self, k):
        super().__init__()
        self.k = k

    def forward(self, x):
        s, n, c = x.shape
        return self.k * x.reshape(s, n // 2, 2 * c)[..., :c] + 0.05 * x.reshape(s, n // 2, 2 * c)[..., c:]


class Record(nn.Module):
    """Identity that keeps what passed through (the fixture generator reads the loop's final streams from it)."""

    def __init__(self):
        super().__init__()
        self.seen = None

    def forward(self, x):
        self.seen = x
        return x


class Stage(nn.Module):
    def __init__(self, blocks, downsample):
        super().__init__()
        self.blocks = nn.ModuleList(blocks)
        self.downsample = downsample


def make_stages():
    """(visual stages, audio stages, final audio recorder)."""
    vs, as_ = [], []
    n = 0
    for li, (dv, da) in enumerate(zip(DEPTH_V, DEPTH_A)):
        vb = [VisBlock(n + i) for i in range(dv)]
        ab = [AudBlock(n + 100 + i) for i in range(da)]
        n += dv
        last = li == len(DEPTH_V) - 1
        vs.append(Stage(vb, nn.Identity() if last else Halve(0.8 + 0.05 * li)))
        as_.append(Stage(ab, Record() if last else Halve(0.9 - 0.05 * li)))
    return vs, as_, as_[-1].downsample


def site_shapes():
    """(Nv, Na) of every adapted block pair, in loop order (stages 0 and 2)."""
    out = []
    nv, na = NV0, NA0
    for li, da in enumerate(DEPTH_A):
        if not (NUM_SKIP > 1 and (li + 1) % NUM_SKIP == 0):
            out += [(nv, na)] * da
        nv, na = nv // 2, na // 2
    return out


def inputs():
    g = torch.Generator().manual_seed(4242)
    f_v = 0.5 * torch.randn(S, NV0, CV, generator=g)
    f_a = 0.5 * torch.randn(S, NA0, CA, generator=g)
    nv, na = NV0 // 8, NA0 // 8
    G_v = torch.randn(S, nv, CV, generator=g)
    G_a = torch.randn(S, na, CA, generator=g)
    return f_v, f_a, G_v, G_a
