"""World-size-2, -4 and -8 CPU (gloo) tests of the data-parallel path (atol 1e-6: eight fp32 summands in another order differ by an ulp of the
largest partial sum where the total cancels to ~0): the bucketed gradient all-reduce + averaging that replaces FSDP's
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


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bucketed_grad_allreduce_averages_over_ranks(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = sum(torch.randn(1_000_003, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
    for r in range(world):
        got = torch.load(tmp_path / f"g{r}.pt")
        torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
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


@pytest.mark.parametrize("world", [2, 4, 8])
def test_early_slices_plus_remainder_equal_one_full_allreduce(tmp_path, world):
    """GradReducer: slices sent while backward is still running + the remainder sent by finish() = the plain averaged all-reduce."""
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = sum(torch.randn(700_001, generator=torch.Generator().manual_seed(200 + r)) for r in range(world)) / world
    for r in range(world):
        torch.testing.assert_close(torch.load(tmp_path / f"o{r}.pt"), want, rtol=1e-6, atol=1e-6)


def _modes_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spatialthinker_amd.actor import GradReducer
    n = 300_007                                                  # not a multiple of world or of the bucket: padded shards, tail bucket
    base = torch.randn(n, generator=torch.Generator().manual_seed(300 + rank))
    out = {}
    for mode, payload in (("allreduce", "fp32"), ("reduce_scatter", "fp32"), ("reduce_scatter", "bf16"), ("allreduce", "bf16")):
        grad = base.clone()
        red = GradReducer(grad, world, None, bucket_elems=70_000, mode=mode, payload=payload)
        red.ready(250_000, n)                                    # early slices in backward order, remainder in finish()
        red.ready(100_000, 250_000)
        red.finish()
        out[f"{mode}/{payload}"] = grad
    torch.save(out, os.path.join(out_dir, f"x{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_direct_reduce_scatter_all_gather_equals_allreduce(tmp_path, world):
    """SURVEY §5.8's exchange (all-to-all of shards -> fixed-order fp32 shard sums -> all-gather) against the plain all-reduce: the
    same averaged gradient on every rank (bit-identical ACROSS ranks by construction; equal to the all-reduce up to fp32 summation
    order, exactly equal at world 2), and the bf16-payload variant within bf16 rounding of it."""
    mp.spawn(_modes_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"x{r}.pt") for r in range(world)]
    want = sum(torch.randn(300_007, generator=torch.Generator().manual_seed(300 + r)) for r in range(world)) / world
    for r in range(world):
        torch.testing.assert_close(res[r]["allreduce/fp32"], want, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(res[r]["reduce_scatter/fp32"], want, rtol=1e-6, atol=1e-6)
        assert torch.equal(res[r]["reduce_scatter/fp32"], res[0]["reduce_scatter/fp32"])          # every rank holds the same bits
        assert torch.equal(res[r]["reduce_scatter/bf16"], res[0]["reduce_scatter/bf16"])
        for k in ("reduce_scatter/bf16", "allreduce/bf16"):
            assert float((res[r][k] - want).abs().max()) < 2 ** -7 * float(want.abs().max()) * 2
    if world == 2:
        assert torch.equal(res[0]["reduce_scatter/fp32"], res[0]["allreduce/fp32"])


def _mean_over_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spatialthinker_amd.actor import GradReducer
    n, sp = 100_003, 2                                           # 100003 = 8 x 12500 + 3: the last shard of every bucket is padded
    out = {}
    for mode, payload in (("allreduce", "fp32"), ("reduce_scatter", "fp32"), ("reduce_scatter", "bf16"), ("allreduce", "bf16")):
        grad = torch.randn(n, generator=torch.Generator().manual_seed(400 + rank))
        red = GradReducer(grad, world, None, bucket_elems=33_000, mode=mode, payload=payload, mean_over=world // sp)
        red.ready(66_000, n)                                     # [66000, 100003): one full bucket + a 1003-element tail bucket (< world shards of 126)
        red.finish()
        out[f"{mode}/{payload}"] = grad
    torch.save(out, os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_world_8_with_sp_2_sums_over_all_ranks_and_divides_by_the_dp_replicas(tmp_path):
    """Ulysses sp = 2 on 8 ranks: the sp ranks of a group hold PARTIAL sums of the same rows, so the exchange sums over all 8 ranks and
    divides by world / sp = 4 (reference: Gather's grad_scaler, verl/utils/ulysses.py:227-235 + FSDP's mean over the dp ranks) — in both
    exchange modes and both payloads, with shard boundaries that do not divide the buckets (tail bucket smaller than a shard row)."""
    world = 8
    mp.spawn(_mean_over_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"s{r}.pt") for r in range(world)]
    want = sum(torch.randn(100_003, generator=torch.Generator().manual_seed(400 + r)) for r in range(world)) / 4
    for r in range(world):
        for k in ("allreduce/fp32", "reduce_scatter/fp32"):
            torch.testing.assert_close(res[r][k], want, rtol=1e-6, atol=1e-6)
        assert torch.equal(res[r]["reduce_scatter/fp32"], res[0]["reduce_scatter/fp32"])
        assert torch.equal(res[r]["reduce_scatter/bf16"], res[0]["reduce_scatter/bf16"])
        for k in ("reduce_scatter/bf16", "allreduce/bf16"):
            assert float((res[r][k] - want).abs().max()) < 2 ** -7 * float(want.abs().max()) * 4


def _announce_worker(rank, world, port, out_dir):
    """update_policy on two ranks with a stub engine: which gradient slices are announced, in which order, on which pass."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spatialthinker_amd.actor import ActorHyper, PolicyEngine
    L, per, head = 3, 1000, 500                                  # flat layout: [vit+embed 700 | L layers x 1000 | final norm + head 500]
    n = 700 + L * per + head
    eng = PolicyEngine.__new__(PolicyEngine)
    eng.h = ActorHyper(micro_batch_size_per_device_for_update=2, global_batch_size_per_device=8, allreduce_bucket_mb=1,
                       grad_exchange=os.environ.get("TEST_EXCHANGE", "allreduce"))
    eng.store = types.SimpleNamespace(grad=torch.zeros(n), device="cpu")
    eng.world, eng.pg, eng.sync_grads, eng.overlap_allreduce, eng._reducer = world, None, True, True, None
    eng.fuse_micro_batches, eng.sched_steps, eng.opt_steps = 2, 0, 0
    eng.share_prompts, eng.tokens_per_pass_grad, eng.last_plan = False, 10 ** 9, {}
    log, passes = [], [0]

    def fwd_bwd(b, loss_in, temperature, **kw):
        on_final = kw.get("on_final")
        passes[0] += 1
        g = eng.store.grad
        g += float(rank + 1)                                     # this pass's contribution everywhere
        if on_final is not None:
            lo_head = 700 + L * per
            on_final(lo_head, n); log.append((passes[0], lo_head, n))
            for i in reversed(range(L)):
                on_final(700 + i * per, 700 + (i + 1) * per); log.append((passes[0], 700 + i * per, 700 + (i + 1) * per))
        return None, torch.zeros(2, 8)

    eng.model = types.SimpleNamespace(forward_backward=fwd_bwd)
    eng._stage = lambda data, sl: None
    steps = []

    def opt_step():
        eng.all_reduce_grads()
        steps.append(eng.store.grad.clone())
        eng.store.grad.zero_()
        return 1.0
    eng.optimizer_step = opt_step
    eng.current_lr = lambda: 0.0
    N, R = 16, 4
    data = dict(input_ids=torch.zeros(N, 8, dtype=torch.long), attention_mask=torch.ones(N, 8, dtype=torch.long), responses=torch.zeros(N, R, dtype=torch.long),
                old_log_probs=torch.zeros(N, R), advantages=torch.zeros(N, R), ref_log_probs=torch.zeros(N, R), position_ids=torch.zeros(N, 8, dtype=torch.long))
    eng.update_policy(data, 1.0)
    torch.save({"log": log, "steps": steps, "passes": passes[0]}, os.path.join(out_dir, f"a{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["allreduce", "reduce_scatter"])
def test_update_policy_announces_layer_slices_only_on_the_last_pass_of_each_optimizer_step(tmp_path, exchange):
    """16 rows, mini-batch 8, micro 2, two micro-batches fused per pass: 2 optimizer steps x 2 passes.  Only the LAST pass of an
    optimizer step announces slices (head + final norm first, then the layers from the back: the backward order); both ranks
    announce the same sequence; what the optimizer sees is the rank-averaged sum of both passes everywhere — announced or not."""
    os.environ["TEST_EXCHANGE"] = exchange
    try:
        mp.spawn(_announce_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    finally:
        del os.environ["TEST_EXCHANGE"]
    a = [torch.load(tmp_path / f"a{r}.pt") for r in range(2)]
    assert a[0]["log"] == a[1]["log"] and a[0]["passes"] == 4
    assert [p for p, _, _ in a[0]["log"]] == [2] * 4 + [4] * 4                       # passes 2 and 4 close the optimizer steps
    assert [(lo, hi) for _, lo, hi in a[0]["log"][:4]] == [(3700, 4200), (2700, 3700), (1700, 2700), (700, 1700)]
    for r in range(2):
        assert len(a[r]["steps"]) == 2
        for g in a[r]["steps"]:
            torch.testing.assert_close(g, torch.full_like(g, 2 * (1 + 2) / 2.0))     # two passes x (1 + 2) summed over ranks / world


def _pool_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spatialthinker_amd.actor import GradReducer
    n = 300_007
    res = {}
    for mode, payload in (("reduce_scatter", "fp32"), ("reduce_scatter", "bf16"), ("allreduce", "bf16"), ("allreduce", "fp32")):
        grad = torch.randn(n, generator=torch.Generator().manual_seed(300 + rank))
        red = GradReducer(grad, world, None, bucket_elems=100_000, mode=mode, payload=payload)
        allocs = []
        for step in range(3):                                 # three optimizer steps on one reducer: slices early + the remainder
            grad.copy_(torch.randn(n, generator=torch.Generator().manual_seed(300 + rank + 10 * step)))
            red.ready(200_000, n)
            red.ready(100_000, 200_000)
            red.finish()
            allocs.append((red.pool.allocations, red.pool.allocated_bytes))
        st = red.stats()
        res[(mode, payload)] = dict(allocs=allocs, stats=st, grad=grad.clone(), free=sum(len(v) for v in red.pool.free.values()))
    torch.save(res, os.path.join(out_dir, f"p{rank}.pt"))
    dist.destroy_process_group()


def test_staging_buffers_are_allocated_once_and_exchange_timing_is_reported(tmp_path):
    """Round-4 hardening of the N > 1 path: the staging buffers of the direct / bf16 exchanges come from a pool that stops growing after
    the first optimizer step (no allocator traffic per step), every buffer is back in the pool after finish(), and stats() reports one
    exchange per finish() with exposed <= total time."""
    world = 2
    mp.spawn(_pool_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(tmp_path / f"p{r}.pt", weights_only=False)
        for (mode, payload), v in res.items():
            a = v["allocs"]
            assert a[1] == a[0] and a[2] == a[0], (mode, payload, a)                 # steps 2 and 3 allocate nothing
            if mode == "allreduce" and payload == "fp32":
                assert a[0] == (0, 0)                                                # in-place all-reduce: no staging at all
            else:
                assert a[0][0] > 0 and v["free"] == a[0][0]                          # every staging buffer went back to the pool
            st = v["stats"]
            assert st["exchanges"] == 3 and 0.0 <= st["allreduce_exposed_s"] <= st["allreduce_s"] + 1e-9
            assert abs(st["early_fraction"] / 3 - 200_007 / 300_007) < 1e-6
            want = sum(torch.randn(300_007, generator=torch.Generator().manual_seed(300 + q + 20)) for q in range(world)) / world
            tol = dict(rtol=1e-6, atol=1e-6) if payload == "fp32" else dict(rtol=2e-2, atol=2e-2)
            torch.testing.assert_close(v["grad"], want, **tol)
