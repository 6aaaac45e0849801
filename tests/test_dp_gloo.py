"""World-size-2 CPU (gloo) tests of the data-parallel path: the bucketed gradient all-reduce + averaging that replaces FSDP's
reduce-scatter, rank sharding of the rollout batch, and cross-rank metric gathering."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spatialthinker_amd.actor import ActorHyper, PolicyEngine
    from verl.single_controller.worker_group import SPMDWorkerGroup
    eng = PolicyEngine.__new__(PolicyEngine)
    n = 1_000_003                                             # not a multiple of the bucket: exercises the tail bucket
    g = torch.Generator().manual_seed(100 + rank)
    eng.store = types.SimpleNamespace(grad=torch.randn(n, generator=g))
    eng.h = ActorHyper(allreduce_bucket_mb=1)                 # 262144-float buckets -> 4 buckets
    eng.world, eng.pg = world, None
    eng.all_reduce_grads()
    torch.save(eng.store.grad, os.path.join(out_dir, f"g{rank}.pt"))
    wg = SPMDWorkerGroup(types.SimpleNamespace())
    gathered = wg.gather_objects({"rank": rank, "loss": 0.5 * rank})
    torch.save(gathered, os.path.join(out_dir, f"m{rank}.pt"))
    dist.destroy_process_group()


def test_bucketed_grad_allreduce_averages_over_ranks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = sum(torch.randn(1_000_003, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    for r in range(world):
        got = torch.load(tmp_path / f"g{r}.pt")
        torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-7)
        assert torch.load(tmp_path / f"m{r}.pt") == [{"rank": 0, "loss": 0.0}, {"rank": 1, "loss": 0.5}]


def test_trainer_rank_sharding_matches_dataproto_chunk():
    """RayPPOTrainer._shard(rank) must equal DataProto.chunk(world)[rank] (Dispatch.DP_COMPUTE_PROTO, decorator.py:106-108)."""
    from verl.protocol import DataProto
    from verl.trainer.ray_trainer import RayPPOTrainer
    full = {"input_ids": torch.arange(24).view(8, 3), "s": np.array([f"s{i}" for i in range(8)], dtype=object)}
    chunks = DataProto.from_single_dict(full).chunk(4)
    for rank in range(4):
        t = RayPPOTrainer.__new__(RayPPOTrainer)
        t.rank, t.local_prompts = rank, 2
        sh = t._shard(full)
        assert torch.equal(sh["input_ids"], chunks[rank].batch["input_ids"]) and sh["s"].tolist() == chunks[rank].non_tensor_batch["s"].tolist()
