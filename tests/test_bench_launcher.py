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
