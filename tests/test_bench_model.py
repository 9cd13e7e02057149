"""bench.py's work model on the CPU (no GPU, no kernels): the per-clip-pair byte / FLOP figures are SURVEY 8(d)'s, every
configuration BASELINE.json names is defined, and a bare `--gpus N` builds the launcher command the driver would use."""
import sys
import types

import pytest

import bench


def test_cfg2_work_model_matches_survey_8d():
    c = bench.CONFIGS["cfg2"]
    assert bench.algorithmic_bytes_per_clip_pair(c, 2) == pytest.approx(93.7e6, rel=2e-3)          # 93.7 MB per clip-pair
    assert bench.reference_flops_per_clip_pair(c) == pytest.approx(90.70e9, rel=2e-3)              # 90.70 GFLOP fwd+bwd
    assert bench.algorithmic_bytes_per_clip_pair(c, 2) * c["B"] == pytest.approx(3.00e9, rel=2e-3)


def test_every_baseline_configuration_is_defined():
    assert set(bench.CONFIGS) == {"cfg1", "cfg2", "cfg3", "cfg4", "cfg5"}
    c1 = bench.CONFIGS["cfg1"]
    assert sum(p[4] for p in c1["pairs"]) == 24 and c1["dtype"] == "f32" and c1["B"] == 2            # 12 block pairs x {p1, p2}
    assert (96, 4096, 128, 2304, 4) in c1["pairs"] and (768, 64, 1024, 36, 4) in c1["pairs"]
    c4 = bench.CONFIGS["cfg4"]
    assert (c4["E_m"], c4["E_s"], c4["K"], c4["groups"]) == (1, 2, 2, 4)
    assert bench.CONFIGS["cfg5"]["K"] == 87 and bench.CONFIGS["cfg3"]["variant"] == "avvp"


def test_self_launch_builds_the_torchrun_command(monkeypatch):
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0)
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(types.SimpleNamespace(gpus=4), ["--gpus", "4", "--steps", "3"])
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_parity_verdict_of_the_bf16_gradient_views():
    """bench.parity_verdict: `ok` is about the hard bars (fp32 1e-3, bf16 output, indices); `ok_bf16_grads` is `no tensor above 5 % AND above the
    eager-autocast error` (vectors on the one draw, single sums over several draws of the upstream gradient) where that list was measured, and falls
    back to the single-draw criterion of rounds 2 - 5 -- which stays on the line under its own name -- where it was not."""
    import bench
    base = dict(idx_equal=True, out_rel_f32=2e-6, grad_rel_f32=1e-4, out_rel_bf16=5e-3, grad_relnorm_bf16_same_mask=0.08, grad_relnorm_bf16_major=0.01,
                grad_relnorm_bf16_tiny_joint=0.01, grad_relnorm_bf16_rest=0.01, grad_abs_bf16_structural_zero=1e-4)
    r = bench.parity_verdict(dict(base, bf16_tensors_above_5pct_and_eager=[]))
    assert r["ok"] and r["ok_bf16_grads"] is True and r["ok_bf16_grads_single_draw"] is False and r["ok_bf16_grads_as_tested"] is True
    r = bench.parity_verdict(dict(base, bf16_tensors_above_5pct_and_eager=["router.0.weight (v) 0.064 (eager 0.059)"]))
    assert r["ok"] and r["ok_bf16_grads"] is False
    r = bench.parity_verdict(dict(base))                                  # the multi-draw view not measured: the single-draw criterion decides
    assert r["ok_bf16_grads"] is False
    r = bench.parity_verdict(dict(base, grad_relnorm_bf16_same_mask=0.03))
    assert r["ok_bf16_grads"] is True and r["ok_bf16_grads_single_draw"] is True
    r = bench.parity_verdict(dict(base, grad_rel_f32=2e-3, bf16_tensors_above_5pct_and_eager=[]))      # a hard bar
    assert not r["ok"] and "grad_rel_f32" in r["failed"]
    r = bench.parity_verdict(dict(idx_equal=True, out_rel_f32=2e-6, grad_rel_f32=1e-4))                # an fp32 configuration: no bf16 leg
    assert r["ok"] and r["ok_bf16_grads"] is None
