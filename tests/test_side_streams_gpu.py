"""The helper streams inside a forward / backward call (csrc/side.{h,cpp}: the fused-statistics down projection beside the hop-1
chain, dBpost beside dApost -> post_small_bwd, the dX GEMM beside the dWt / dT chain) only re-order independent work: the outputs and
every gradient must be BIT-identical with the forks forced on (AVMOE_SIDE_MIN=0) and switched off (AVMOE_NO_SIDE=1).  The switches are
read once per process, so each setting runs in its own interpreter and reports a digest of its results."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import hashlib, sys
sys.path.insert(0, {root!r})
import torch
from oracle import avmoe_oracle as O
from tests.moe_gpu_util import MoeRun
bf16 = sys.argv[1] == "bf16"
cfg = O.AdapterConfig(Cx=768, Nx=256, Cy=768, Ny=196, reduction=12, groups=2, K=32)      # the cfg-2 site shape, fewer tokens
P, B = O.init_params(cfg, seed=3)
g = torch.Generator().manual_seed(11)
S = 6
X = 0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g); Y = 0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)
G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
h = hashlib.sha256()
for rep in range(3):                      # repeated: a race between the branches would not hit the same bits every time
    run = MoeRun(cfg, P, B, X, Y, bf16=bf16, training=True).forward()
    got = run.backward(G)
    torch.cuda.synchronize()
    h.update(run.out.float().cpu().numpy().tobytes())
    for k in sorted(got):
        h.update(got[k].float().cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
"""


def _digest(env_extra, mode):
    env = dict(os.environ)
    env.pop("AVMOE_NO_SIDE", None); env.pop("AVMOE_SIDE_MIN", None)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT), mode], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("DIGEST ")]
    assert lines, out.stdout[-2000:]
    return lines[-1].split()[1]


@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_forks_do_not_change_a_bit(mode):
    forked = _digest({"AVMOE_SIDE_MIN": "0"}, mode)
    plain = _digest({"AVMOE_NO_SIDE": "1"}, mode)
    assert forked == plain
