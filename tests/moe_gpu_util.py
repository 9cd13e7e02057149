"""GPU-side harness shared by the parity tests: runs avmoe_moe_forward / backward through the C ABI on
fixture or synthetic data and exposes every workspace buffer by name."""
import ctypes as C

import numpy as np
import torch

from avmoe_amd import _capi as capi
from avmoe_amd import _capi_moe as cm


def make_desc(cfg, S, bf16, training):
    d = cm.MoeDesc()
    d.S, d.N, d.C, d.M, d.Cy = S, cfg.Nx, cfg.Cx, cfg.Ny, cfg.Cy
    d.E_m, d.E_s, d.d, d.groups, d.K = cfg.E_m, cfg.E_s, cfg.d, cfg.groups, cfg.K
    d.use_bn, d.use_gate, d.ln_before, d.ln_post = int(cfg.use_bn), int(cfg.use_gate), int(cfg.ln_before), int(cfg.ln_post)
    d.variant, d.self_attn, d.lb_loss = cm.VARIANT[cfg.variant], cm.SELF_ATTN[cfg.self_attn], int(cfg.lb_loss)
    d.dtype, d.training = (capi.BF16 if bf16 else capi.F32), int(training)
    d.bn_eps, d.ln_eps, d.bn_momentum = cfg.bn_eps, cfg.ln_eps, cfg.bn_momentum
    return d


class MoeRun:
    """One forward (+ optional backward) of the HIP path on cuda:0."""

    def __init__(self, cfg, P, B, X, Y, bf16=False, training=True, noise=None, mha_keep=None):
        self.L = capi.lib()
        self.cfg, self.bf16, self.training = cfg, bf16, training
        dev = torch.device("cuda:0")
        self.dev = dev
        S = X.shape[0]
        self.S = S
        self.desc = make_desc(cfg, S, bf16, training)
        tdt = torch.bfloat16 if bf16 else torch.float32
        self.tdt = tdt
        self.params = {k: v.detach().to(dev, torch.float32).contiguous() for k, v in P.items()}
        self.buffers = {k: v.detach().to(dev, torch.float32).contiguous() for k, v in B.items()
                        if v.is_floating_point()}
        self.X = X.to(dev, tdt).contiguous()
        self.Y = Y.to(dev, tdt).contiguous()
        self.noise = noise.to(dev, torch.float32).contiguous() if noise is not None else None
        nsaved = self.L.avmoe_moe_saved_bytes(C.byref(self.desc))
        nscratch = self.L.avmoe_moe_scratch_bytes(C.byref(self.desc))
        if nsaved == 0:
            raise capi.AvmoeError(self.L.avmoe_last_error().decode())
        # workspaces exactly as large as the library asks for, followed by a guard band the kernels must never touch
        self.GUARD = 4096
        self._saved_all = torch.full((nsaved + self.GUARD,), 0xAB, dtype=torch.uint8, device=dev)
        self._scratch_all = torch.full((nscratch + self.GUARD,), 0xAB, dtype=torch.uint8, device=dev)
        # the workspaces arrive UNINITIALISED in production (torch.empty in the facade): poison them (0xFF.. = NaN in fp32 and bf16)
        # so that a kernel that relies on zeroed padding shows up as NaN here; AVMOE_TEST_ZERO_WS=1 restores zeroed workspaces
        import os
        poison = 0 if os.environ.get("AVMOE_TEST_ZERO_WS") else 0xFF
        self._saved_all[:nsaved] = poison
        self._scratch_all[:nscratch] = poison
        self.saved, self.scratch = self._saved_all[:nsaved], self._scratch_all[:nscratch]
        self.table = {n: (r, o, b) for (n, r, o, b) in cm.buffer_table(self.L, self.desc)}
        self.out = torch.empty_like(self.X)
        self.probs = torch.empty(S, cfg.E, device=dev, dtype=torch.float32)
        self.idx = torch.empty(S, device=dev, dtype=torch.int64)
        self.lb = torch.zeros(1, device=dev, dtype=torch.float32)
        # "v1" experts: dropout multipliers of the attention weights, {expert prefix: (N * heads, S, S)} (include/avmoe.h sa_keep)
        self.keep = {f"{pre}.{cm.SA_KEEP}": v.detach().to(dev, torch.float32).contiguous() for pre, v in (mha_keep or {}).items()}
        self.ptrs = cm.make_ptrs({**self.params, **self.buffers, **self.keep}, cfg.E_m, cfg.E_s)

    def forward(self):
        st = self.L.avmoe_moe_forward(C.byref(self.desc), self.X.data_ptr(), self.Y.data_ptr(), C.byref(self.ptrs),
                                      self.noise.data_ptr() if self.noise is not None else None,
                                      self.out.data_ptr(), self.probs.data_ptr(), self.idx.data_ptr(), self.lb.data_ptr(),
                                      self.saved.data_ptr(), self.scratch.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
        capi.check(st, "avmoe_moe_forward")
        torch.cuda.synchronize()
        return self

    def backward(self, dout, lb_weight=0.0):
        dev = self.dev
        self.dOut = dout.to(dev, self.tdt).contiguous()
        self.dX = torch.empty_like(self.X)
        self.dY = torch.empty_like(self.Y)
        self.grads = {k: torch.full_like(v, float("nan")) for k, v in self.params.items()}
        self.gptrs = cm.make_ptrs(self.grads, self.cfg.E_m, self.cfg.E_s)
        self.lbw = torch.full((1,), float(lb_weight), device=dev, dtype=torch.float32)
        st = self.L.avmoe_moe_backward(C.byref(self.desc), self.X.data_ptr(), self.Y.data_ptr(), C.byref(self.ptrs),
                                       self.dOut.data_ptr(), self.lbw.data_ptr(), self.saved.data_ptr(),
                                       self.scratch.data_ptr(), self.dX.data_ptr(), self.dY.data_ptr(),
                                       C.byref(self.gptrs), torch.cuda.current_stream().cuda_stream)
        capi.check(st, "avmoe_moe_backward")
        torch.cuda.synchronize()
        g = {k: v.cpu() for k, v in self.grads.items()}
        g["X"], g["Y"] = self.dX.float().cpu(), self.dY.float().cpu()
        return g

    def guards_intact(self):
        """True when no kernel wrote past the end of the saved / scratch workspace."""
        return bool((self._saved_all[-self.GUARD:] == 0xAB).all()) and bool((self._scratch_all[-self.GUARD:] == 0xAB).all())

    def buf(self, name, dtype=None, shape=None):
        """Workspace buffer `name` as a CPU tensor (dtype: torch dtype of the elements)."""
        region, off, nbytes = self.table[name]
        raw = (self.saved if region == 0 else self.scratch)[off:off + nbytes]
        dt = dtype or torch.float32
        t = raw.view(dt)
        if shape is not None:
            n = int(np.prod(shape))
            t = t[:n].reshape(shape)
        return t.float().cpu()


def compare_forward_intermediates(run: MoeRun, A, verbose=True):
    """Per-stage max abs error of the HIP forward against oracle/algebra_ref.py's same-named tensors.
    Returns {name: (err, scale)}."""
    cfg, S = run.cfg, run.S
    d = run.desc
    T = run.tdt
    g, dg = cfg.groups, cfg.d // cfg.groups
    E, K, N, Cc = cfg.E, cfg.K, cfg.Nx, cfg.Cx
    # paddings as the library planned them (they depend on which kernel family serves the shape): read them off the buffer sizes
    dgp = run.table["wsum"][2] // 4 // (E * g)          # wsum: DZ = E * g * dgp floats
    sv = A.sv
    lat = [j for j, ex in enumerate(A.experts) if ex.latent]
    El = len(lat)
    esz = 2 if run.bf16 else 4
    KLT_ = run.table["Text"][2] // esz // (S * Cc)       # Text: S * KLT * C elements, KLT = El * Kp + 2
    Kp = (KLT_ - 2) // El if El else -(-K // 8) * 8       # latent slots are padded to Kp rows (padding rows are zero)
    KL = El * Kp
    KLT, KLp = KL + 2, -(-(KL + 2) // 8) * 8
    DZ = E * g * dgp
    KP = E * dgp + 3 * E
    KPp = -(-KP // 8) * 8
    res = {}

    def rec(name, got, ref):
        ref = ref.float()
        res[name] = (float((got - ref).abs().max()), float(ref.abs().max()))

    rec("rin", run.buf("rin", shape=(S, 2 * Cc)), sv["rin"])
    rec("probs", run.probs.cpu(), sv["p"])
    Text = run.buf("Text", T, (S, KLT, Cc))
    L2 = run.buf("L2", shape=(S, N, max(KL, 8)))
    a = run.buf("a", T, (max(El, 1), S, N, Kp))           # per-latent-slot planes [slot][token][Kp]
    Z = run.buf("Z", shape=(S, N, g, E, dgp))
    TW = run.buf("TW", shape=(S, KLT, g, E, dgp))
    TT = run.buf("TT", shape=(S, max(El, 1), K, K))
    rmu = run.buf("rmu", shape=(2, E, S, N))              # [r | mu][expert][token]
    rpm = run.buf("rpmup", shape=(2, E, S, N))
    bn1 = run.buf("bn1", shape=(4, g, E, dgp))
    mz = run.buf("mz", shape=(g, E, dgp))
    Szz = run.buf("Szz", shape=(g, E, dgp, dgp))
    bn2 = run.buf("bn2", shape=(4, E, Cc))
    Gq = run.buf("Gq", shape=(g, E, dgp, dgp))
    Ap = run.buf("Apost", T, (S, N, g, KPp))
    for j, (ex, e) in enumerate(zip(A.experts, sv["E"])):
        tag = f"e{j}."
        if ex.latent:
            l = lat.index(j)
            rec(tag + "T", Text[:, l * Kp:l * Kp + K], e["T"])
            rec(tag + "TT", TT[:, l], e["TT"])
            rec(tag + "TW", TW[:, l * Kp:l * Kp + K, :, j, :dg], e["TW"])
            rec(tag + "L2", L2[:, :, l * Kp:l * Kp + K], e["L2"])
            rec(tag + "a", a[l, :, :, :K], e["a"])
        rec(tag + "z", Z[:, :, :, j, :dg], e["z"])
        if cfg.ln_before:
            rec(tag + "r", rmu[0, j], e["r"])
            rec(tag + "mu", rmu[1, j], e["mu"])
        if cfg.use_bn:
            rec(tag + "bn1.rstd", bn1[1, :, j, :dg], e["r1"].expand(g, dg))
            if run.training:
                rec(tag + "mz", mz[:, j, :dg], e["mz"])
                rec(tag + "Szz", Szz[:, j, :dg, :dg], e["Szz"])
            rec(tag + "mo", bn2[0, j].reshape(g, -1), e["mo"])
        rec(tag + "k2", bn2[2, j].reshape(g, -1), e["k2"])
        rec(tag + "h2", bn2[3, j].reshape(g, -1), e["h2"])
        if cfg.ln_post:
            rec(tag + "G", Gq[:, j, :dg, :dg], e["G"])
            rec(tag + "rp", rpm[0, j], e["rp"])
            rec(tag + "mup", rpm[1, j], e["mup"])
        rec(tag + "Az", Ap[:, :, :, j * dgp:j * dgp + dg], e["Az"])
        rec(tag + "c1", Ap[:, :, 0, E * dgp + 3 * j + 0], e["c1"])
        rec(tag + "c2", Ap[:, :, 0, E * dgp + 3 * j + 1], e["c2"])
    if verbose:
        for k, (err, sc) in res.items():
            flag = "  <<<<" if err > 1e-3 * max(sc, 1e-3) else ""
            print(f"   {k:14s} err {err:10.3e}  scale {sc:10.3e}{flag}")
    return res
