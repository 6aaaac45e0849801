"""bench.py's multi-rank contract on CPU (gloo): `--gpus N` started as one process spawns N ranks as a child torchrun, the ranks
rendezvous, time K steps between barriers, take the max over ranks and rank 0 prints ONE JSON line with n_gpus = rccl_ranks = N."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"], {"MASTER_PORT": "29547"})
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["ms_per_step"] >= 3.5            # the max over ranks: rank 1 sleeps 4 ms per step, rank 0 only 2


def test_single_rank_does_not_spawn():
    out = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--dry-run"])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], cwd=ROOT, env=env, capture_output=True, text=True)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_parity_record_and_telemetry_summary_without_a_gpu():
    """The two round-6 additions to the bench line: `parity` quotes the committed yardstick (engine and HF-bf16, both against fp32) next to
    north_star's literal tolerance; ChipTelemetry.summary turns {shader cycles, 100-MHz ticks} samples into MHz per phase (here fed by hand)."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    rec = bench.parity_record()
    assert rec["north_star_literal_tolerance"] == 1e-3 and rec["engine_vs_fp32"]["tokens"] == rec["hf_bf16_vs_fp32"]["tokens"] >= 200
    for k in ("rms", "max", "mean_of_per_batch_max"):
        assert 0 < rec["engine_vs_fp32"][k] <= rec["hf_bf16_vs_fp32"][k]                 # the committed measurement satisfies the criterion in force
    assert all(0 < r <= 1.0 for r in rec["hidden_state_taps_engine_over_hf_bf16_rel_l2"].values())
    t = bench.ChipTelemetry.__new__(bench.ChipTelemetry)                                # no device: fill the fields summary() reads
    t.period, t.files, t.errors, t.pci = 0.25, {"power1_input": "x"}, 0, "0000:00:00.0"
    t.labels = ["gen", "gen", "update_actor", "update_actor", "ref"]
    t.host = [{"power1_input": 9.0e8}, {"power1_input": 9.0e8}, {"power1_input": 1.35e9}, {"power1_input": 1.33e9}, {"power1_input": 1.3e9}]
    ticks = torch.full((5, 8, 1), 2000, dtype=torch.int64)
    cyc = torch.tensor([44000, 44400, 35400, 35600, 34000], dtype=torch.int64)[:, None, None].expand(5, 8, 1)
    t.buf = torch.cat([cyc, ticks], dim=2).contiguous()

    class _S:
        def synchronize(self): pass
    t.stream = _S()
    s = t.summary()
    assert abs(s["by_phase"]["gen"]["clock_mhz_mean"] - 2210.0) < 1e-6 and abs(s["by_phase"]["update_actor"]["clock_mhz_mean"] - 1775.0) < 1e-6
    assert abs(s["gemm_phases"]["clock_mhz_mean"] - (1770 + 1780 + 1700) / 3) < 1e-6 and s["gemm_phases"]["samples"] == 3
    assert abs(s["gemm_phases"]["socket_power_w_mean"] - (1350 + 1330 + 1300) / 3) < 1e-6


def test_vision_plan_cache_returns_the_same_plan_as_a_fresh_computation():
    import numpy as np
    sys.path.insert(0, ROOT)
    from spatialthinker_amd import indexing as ix
    g = np.array([[1, 8, 12], [1, 4, 4], [2, 6, 6]])
    kw = dict(merge=2, window=56, patch=14, head_dim=80)
    ix._VISION_PLANS.clear()
    a = ix.plan_vision(g, **kw)
    assert ix.plan_vision(g.copy(), **kw) is a and len(ix._VISION_PLANS) == 1              # keyed by the grid values
    fresh = ix._plan_vision(g, **kw)
    for f in ("patch_gather", "merged_inverse", "cu_window", "cu_image", "cos", "sin"):
        assert np.array_equal(getattr(a, f), getattr(fresh, f)), f
    assert ix.plan_vision(g[:2], **kw) is not a and len(ix._VISION_PLANS) == 2
