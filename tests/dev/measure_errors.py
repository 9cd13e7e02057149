"""dev: measured parity errors of the HIP path at the BENCHMARKED shape (cfg-2), per tensor -- the numbers the bars in
tests/test_cfg2_shape_gpu.py are set from.   python tests/dev/measure_errors.py [--full]

  1. cfg-2 at B = 2 clips (S = 20), both sites: HIP fp32 vs oracle (max-abs relative), HIP bf16 vs oracle on bf16-rounded inputs
     (max-abs relative for out, norm-wise relative for every gradient)
  2. --full: S = 320, HIP bf16 vs HIP fp32 on the same bf16-rounded inputs (norm-wise relative, out and every gradient)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import avmoe_oracle as O          # noqa: E402
from tests.moe_gpu_util import MoeRun         # noqa: E402

SITES = {"audio": dict(Cx=768, Nx=1024, Cy=768, Ny=196), "visual": dict(Cx=768, Nx=196, Cy=768, Ny=1024)}


def data(cfg, S, seed):
    g = torch.Generator().manual_seed(seed)
    X = 0.3 * torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    Y = 0.3 * torch.randn(S, cfg.Ny, cfg.Cy, generator=g)
    G = torch.randn(S, cfg.Nx, cfg.Cx, generator=g)
    return X, Y, G


def relmax(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def relnorm(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    full = "--full" in sys.argv
    for site, shp in SITES.items():
        cfg = O.AdapterConfig(**shp, reduction=12, groups=2, K=32, E_m=2, E_s=2)
        P, B = O.init_params(cfg, seed=5)
        S = 20
        X, Y, G = data(cfg, S, 99)
        fwd, grads = O.moe_forward_backward(P, B, X, Y, cfg, G, training=True)
        run = MoeRun(cfg, P, B, X, Y, bf16=False, training=True).forward()
        got = run.backward(G)
        print(f"== {site} site, S={S}: HIP fp32 vs oracle fp32 ==")
        print(f"   idx equal {torch.equal(run.idx.cpu(), fwd['idx'])}   out relmax {relmax(run.out.float().cpu(), fwd['out']):.3e}"
              f"   probs max abs {float((run.probs.cpu() - fwd['probs']).abs().max()):.3e}")
        gmax = max(float(v.abs().max()) for v in grads.values())
        for k, v in got.items():
            print(f"   {k:46s} max-abs err/scale {float((v - grads[k]).abs().max()) / max(float(grads[k].abs().max()), 1e-3 * gmax):.3e}"
                  f"   scale {float(grads[k].abs().max()):.3e}")
        Xb, Yb, Gb = X.bfloat16().float(), Y.bfloat16().float(), G.bfloat16().float()
        fwd, grads = O.moe_forward_backward(P, B, Xb, Yb, cfg, Gb, training=True)
        run = MoeRun(cfg, P, B, X, Y, bf16=True, training=True).forward()
        got = run.backward(G)
        print(f"== {site} site, S={S}: HIP bf16 vs oracle on bf16-rounded inputs ==")
        print(f"   idx equal {torch.equal(run.idx.cpu(), fwd['idx'])}   out relmax {relmax(run.out.float().cpu(), fwd['out']):.3e}"
              f"   out relnorm {relnorm(run.out.float().cpu(), fwd['out']):.3e}")
        gn = max(float(v.norm()) for k, v in grads.items() if k not in ("X", "Y"))
        for k, v in got.items():
            print(f"   {k:46s} relnorm {relnorm(v.float(), grads[k]):.3e}   relmax {relmax(v.float(), grads[k]):.3e}"
                  f"   norm/gmax {float(grads[k].norm()) / gn:.3e}")
        # the reference's OWN arithmetic under bf16 autocast (eager PyTorch on the GPU; fp32 parameters, matmuls in bf16,
        # softmax / norms in fp32) against the same fp32 oracle: the error a bf16 run of the reference formulation carries
        dev = torch.device("cuda:0")
        Pd = {k: v.to(dev) for k, v in P.items()}
        Bd = {k: v.to(dev) for k, v in B.items()}
        with torch.autocast("cuda", dtype=torch.bfloat16):
            fe, ge = O.moe_forward_backward(Pd, Bd, Xb.to(dev), Yb.to(dev), cfg, Gb.to(dev), training=True)
        print(f"== {site} site, S={S}: EAGER bf16-autocast oracle (GPU) vs oracle fp32 ==")
        print(f"   idx equal {torch.equal(fe['idx'].cpu(), fwd['idx'])}   out relmax {relmax(fe['out'].float().cpu(), fwd['out']):.3e}"
              f"   out relnorm {relnorm(fe['out'].float().cpu(), fwd['out']):.3e}")
        for k in got:
            print(f"   {k:46s} relnorm {relnorm(ge[k].float().cpu(), grads[k]):.3e}   hip {relnorm(got[k].float(), grads[k]):.3e}")
        del run, Pd, Bd, fe, ge
        torch.cuda.empty_cache()
        if full:
            S = 320
            X, Y, G = data(cfg, S, 101)
            Xb, Yb, Gb = X.bfloat16().float(), Y.bfloat16().float(), G.bfloat16().float()
            r32 = MoeRun(cfg, P, B, Xb, Yb, bf16=False, training=True).forward()
            g32 = r32.backward(Gb)
            o32, i32 = r32.out.float().cpu(), r32.idx.cpu()
            del r32
            torch.cuda.empty_cache()
            r16 = MoeRun(cfg, P, B, Xb, Yb, bf16=True, training=True).forward()
            g16 = r16.backward(Gb)
            print(f"== {site} site, S={S}: HIP bf16 vs HIP fp32 ==")
            print(f"   idx equal {torch.equal(r16.idx.cpu(), i32)}   out relmax {relmax(r16.out.float().cpu(), o32):.3e}"
                  f"   out relnorm {relnorm(r16.out.float().cpu(), o32):.3e}")
            gn = max(float(v.norm()) for k, v in g32.items() if k not in ("X", "Y"))
            for k, v in g16.items():
                print(f"   {k:46s} relnorm {relnorm(v.float(), g32[k]):.3e}   relmax {relmax(v.float(), g32[k]):.3e}"
                      f"   norm/gmax {float(g32[k].norm()) / gn:.3e}")
            del r16
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
