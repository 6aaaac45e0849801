"""World-size-2 and -4 CPU (gloo) tests of the data-parallel path: the bucketed gradient all-reduce + averaging that replaces FSDP's
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


@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_grad_allreduce_averages_over_ranks(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = sum(torch.randn(1_000_003, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    for r in range(world):
        got = torch.load(tmp_path / f"g{r}.pt")
        torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-7)
        assert torch.load(tmp_path / f"m{r}.pt") == [{"rank": q, "loss": 0.5 * q} for q in range(world)]


def _overlap_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spatialthinker_amd.actor import GradReducer
    n = 700_001
    grad = torch.randn(n, generator=torch.Generator().manual_seed(200 + rank))
    red = GradReducer(grad, world, None, bucket_elems=100_000)
    # slices announced early, in "backward order" (tail first, then layers from the back), with gaps left for finish()
    red.ready(650_000, n)
    for lo in (500_000, 350_000, 200_000):
        red.ready(lo, lo + 150_000)
    with pytest.raises(AssertionError):
        red.ready(300_000, 360_000)                           # overlaps an announced slice
    assert red.early_elems == 50_001 + 3 * 150_000
    red.finish()                                              # [0, 200000) goes out here
    assert red.sent == [] and red.works == []
    # a second optimizer step on the same reducer, nothing announced: everything goes out in finish()
    grad2 = grad.clone()
    red.finish()
    torch.testing.assert_close(grad, grad2)                   # mean over identical... (each rank holds the same averaged buffer)
    torch.save(grad, os.path.join(out_dir, f"o{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_early_slices_plus_remainder_equal_one_full_allreduce(tmp_path, world):
    """GradReducer: slices sent while backward is still running + the remainder sent by finish() = the plain averaged all-reduce."""
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = sum(torch.randn(700_001, generator=torch.Generator().manual_seed(200 + r)) for r in range(world)) / world
    for r in range(world):
        torch.testing.assert_close(torch.load(tmp_path / f"o{r}.pt"), want, rtol=1e-6, atol=1e-7)
