"""Ulysses sequence parallelism on two gloo ranks (CPU): the all-to-all pair around attention (verl/utils/ulysses.py; reference
verl/utils/ulysses.py:63-298, call sites flash_attention_utils.py:98-106,146-148 and dp_actor.py:107-133).  Ground truth is the same
computation on one process."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _attention(q, k, v, cu):
    """Causal GQA attention over packed sequences, fp64: q (T, hq, D), k / v (T, hkv, D), cu = sequence boundaries."""
    T, hq, D = q.shape
    g = hq // k.shape[1]
    out = torch.zeros_like(q)
    for a, b in zip(cu[:-1], cu[1:]):
        for h in range(hq):
            s = q[a:b, h] @ k[a:b, h // g].T / D ** 0.5
            s = s.masked_fill(torch.triu(torch.ones(b - a, b - a, dtype=torch.bool), 1), float("-inf"))
            out[a:b, h] = torch.softmax(s, -1) @ v[a:b, h // g]
    return out


def _inputs():
    g = torch.Generator().manual_seed(3)
    T, H, hq, hkv, D = 37, 24, 4, 2, 6                     # 37 tokens: not a multiple of 2 -> padding path
    x = torch.randn(T, H, generator=g, dtype=torch.float64)
    wq, wk, wv = (torch.randn(H, n * D, generator=g, dtype=torch.float64) * 0.3 for n in (hq, hkv, hkv))
    wo = torch.randn(hq * D, H, generator=g, dtype=torch.float64) * 0.3
    cu = [0, 11, 30, 37]
    return x, wq, wk, wv, wo, cu, (hq, hkv, D)


def _layer_single(x, wq, wk, wv, wo, cu, dims):
    hq, hkv, D = dims
    T = x.shape[0]
    a = _attention((x @ wq).view(T, hq, D), (x @ wk).view(T, hkv, D), (x @ wv).view(T, hkv, D), cu)
    return a.reshape(T, hq * D) @ wo


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from verl.utils import ulysses as U
    from verl.protocol import DataProto
    from verl.workers.sharding_manager import FSDPUlyssesShardingManager
    res = {}
    sp_group = dist.new_group(list(range(world)))
    assert U.get_ulysses_sequence_parallel_world_size() == 1 and U.gather_seq_scatter_heads(torch.ones(2, 2), 0, 1) is not None   # no group: identity
    U.set_ulysses_sequence_parallel_group(sp_group)
    assert U.get_ulysses_sequence_parallel_world_size() == world and U.get_ulysses_sequence_parallel_rank() == rank
    x, wq, wk, wv, wo, cu, (hq, hkv, D) = _inputs()
    T = x.shape[0]
    ws = [w.clone().requires_grad_(True) for w in (wq, wk, wv, wo)]
    # ---- the token stream padded and sliced as dp_actor.py:107-113 does with the ids
    ids = torch.arange(1, T + 1)[None]
    pos = torch.cat([torch.arange(b - a) for a, b in zip(cu[:-1], cu[1:])])[None]
    ids_l, pos_p, pad = U.ulysses_pad_and_slice_inputs(ids, pos, sp_size=world)
    Tp = T + pad
    assert pad == (-T) % world and ids_l.shape == (1, Tp // world) and pos_p.shape == (1, Tp)
    assert torch.equal(pos_p[0, T:], torch.arange(pad))
    xl = U.slice_input_tensor(x, dim=0, padding=True).clone().requires_grad_(True)           # this rank's rows (zero rows as padding)
    assert xl.shape[0] == Tp // world
    # ---- local projections, all-to-all, attention on ALL tokens for heads / sp heads, all-to-all back, local output projection
    q = (xl @ ws[0]).view(-1, hq, D); k = (xl @ ws[1]).view(-1, hkv, D); v = (xl @ ws[2]).view(-1, hkv, D)
    qf = U.gather_seq_scatter_heads(q, seq_dim=0, head_dim=1, unpadded_dim_size=T)
    kf = U.gather_seq_scatter_heads(k, seq_dim=0, head_dim=1, unpadded_dim_size=T)
    vf = U.gather_seq_scatter_heads(v, seq_dim=0, head_dim=1, unpadded_dim_size=T)
    assert qf.shape == (T, hq // world, D) and kf.shape == (T, hkv // world, D)
    af = _attention(qf, kf, vf, cu)
    al = U.gather_heads_scatter_seq(af, head_dim=1, seq_dim=0)                                # pads the sequence again
    assert al.shape == (Tp // world, hq, D)
    yl = al.reshape(-1, hq * D) @ ws[3]
    # ---- the per-slice outputs gathered and unpadded (dp_actor.py:131-133), a loss on the full stream, backward through everything
    y = U.gather_outputs_and_unpad(yl, gather_dim=0, unpad_dim=0, padding_size=pad, grad_scaler=False)
    assert y.shape == (T, x.shape[1])
    tgt = torch.cos(torch.arange(y.numel(), dtype=torch.float64)).view_as(y)
    ((y * tgt).sum()).backward()
    res["y"] = y.detach()
    res["dx_local"] = xl.grad.detach()
    for w in ws:                                             # weight gradients: partial sums over this rank's tokens -> sum over the group
        dist.all_reduce(w.grad, group=sp_group)
    res["dw"] = [w.grad.detach() for w in ws]
    # ---- Gather's grad_scaler (the data-parallel average also runs over the sp ranks)
    z = torch.full((3, 2), float(rank + 1), dtype=torch.float64, requires_grad=True)
    zz = U.gather_outputs_and_unpad(z, gather_dim=0, grad_scaler=True)
    zz.sum().backward()
    res["z_grad"], res["zz"] = z.grad.clone(), zz.detach()
    # ---- async all-to-all form
    wait = U.all_to_all_tensor(torch.arange(8.0).view(4, 2) + 10 * rank, scatter_dim=1, gather_dim=0, async_op=True)
    res["a2a"] = wait()
    # ---- sharding manager: rows of the group gathered on the way in, this rank's chunk on the way out
    class _Dim:
        def get_group(self): return sp_group
        def size(self): return world
        def get_local_rank(self): return rank
    U.set_ulysses_sequence_parallel_group(None)
    mgr = FSDPUlyssesShardingManager({"sp": _Dim()})
    dp = DataProto.from_dict({"a": torch.arange(3)[:, None] + 100 * rank}, non_tensors={"s": np.array([f"r{rank}_{i}" for i in range(3)], dtype=object)})
    with mgr:
        assert U.get_ulysses_sequence_parallel_group() is sp_group
        full = mgr.preprocess_data(dp)
        res["full_a"], res["full_s"] = full.batch["a"].clone(), list(full.non_tensor_batch["s"])
        back = mgr.postprocess_data(full)
        res["back_a"], res["back_s"] = back.batch["a"].clone(), list(back.non_tensor_batch["s"])
    assert U.get_ulysses_sequence_parallel_group() is None
    from verl.protocol import TensorBatch, allgather_dict_tensors
    ag = allgather_dict_tensors({"b": torch.full((2, 2), float(rank)), "a": torch.arange(2) + 10 * rank}, size=world, group=sp_group)
    agb = allgather_dict_tensors(TensorBatch({"a": torch.arange(2) + 10 * rank}), size=world, group=sp_group)
    res["ag_a"], res["ag_b"], res["agb"] = ag["a"], ag["b"], (len(agb), agb["a"].clone())
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier(); dist.destroy_process_group()


def test_ulysses_layer_equals_the_single_process_layer_forward_and_backward(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, 29591, str(tmp_path)), nprocs=world)
    r = [torch.load(tmp_path / f"r{k}.pt", weights_only=False) for k in range(world)]
    x, wq, wk, wv, wo, cu, dims = _inputs()
    xs = x.clone().requires_grad_(True)
    ws = [w.clone().requires_grad_(True) for w in (wq, wk, wv, wo)]
    y = _layer_single(xs, *ws, cu, dims)
    tgt = torch.cos(torch.arange(y.numel(), dtype=torch.float64)).view_as(y)
    (y * tgt).sum().backward()
    T = x.shape[0]
    per = (T + (-T) % world) // world
    for k in range(world):
        torch.testing.assert_close(r[k]["y"], y.detach(), rtol=1e-10, atol=1e-12)              # every rank holds the full output
        want_dx = torch.zeros(per, x.shape[1], dtype=torch.float64)
        rows = xs.grad[k * per:min(T, (k + 1) * per)]
        want_dx[:rows.shape[0]] = rows
        torch.testing.assert_close(r[k]["dx_local"], want_dx, rtol=1e-10, atol=1e-12)           # gradient of this rank's slice, 0 on the pad rows
        for got, w in zip(r[k]["dw"], ws):
            torch.testing.assert_close(got, w.grad, rtol=1e-10, atol=1e-12)
        assert torch.equal(r[k]["z_grad"], torch.full((3, 2), float(world), dtype=torch.float64))          # slice of ones x sp
        assert torch.equal(r[k]["zz"], torch.cat([torch.full((3, 2), float(j + 1), dtype=torch.float64) for j in range(world)]))
        # all-to-all: piece k of every rank's columns, rank-major rows
        want = torch.cat([(torch.arange(8.0).view(4, 2) + 10 * j)[:, k:k + 1] for j in range(world)], 0)
        assert torch.equal(r[k]["a2a"], want)
        assert r[k]["full_a"].flatten().tolist() == [0, 1, 2, 100, 101, 102] and r[k]["full_s"] == [f"r{j}_{i}" for j in range(world) for i in range(3)]
        assert r[k]["back_a"].flatten().tolist() == [100 * k + i for i in range(3)] and r[k]["back_s"] == [f"r{k}_{i}" for i in range(3)]
        assert r[k]["ag_a"].tolist() == [0, 1, 10, 11] and r[k]["ag_b"].shape == (4, 2) and r[k]["agb"][0] == 4 and r[k]["agb"][1].tolist() == [0, 1, 10, 11]


def _golden_worker(rank, world, port, out_dir):
    """the calls of tests/golden/make_ulysses_golden.py:worker, on this repo's module"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_ulysses_golden import inputs
    from verl.utils import ulysses as U
    U.set_ulysses_sequence_parallel_group(dist.group.WORLD)
    i = inputs(rank)
    res = {}
    ids, pos, pad = U.ulysses_pad_and_slice_inputs(i["ids"], i["pos"], sp_size=world)
    res.update(pad_ids=ids, pad_pos=pos, pad_size=torch.tensor(pad))
    res["slice_pad"] = U.slice_input_tensor(i["pos"].double(), dim=1, padding=True)
    res["gather_seq"] = U.gather_seq_scatter_heads(i["x_seq"], seq_dim=1, head_dim=2)
    res["gather_seq_unpad"] = U.gather_seq_scatter_heads(i["x_seq"], seq_dim=1, head_dim=2, unpadded_dim_size=37)
    res["gather_heads"] = U.gather_heads_scatter_seq(i["x_head"], head_dim=2, seq_dim=1)
    for scaler in (True, False):
        y = i["y"].clone().requires_grad_(True)
        full = U.gather_outputs_and_unpad(y, gather_dim=0, unpad_dim=0, padding_size=1, grad_scaler=scaler)
        (full * i["gy"]).sum().backward()
        res[f"gather_out_{int(scaler)}"] = full.detach()
        res[f"gather_out_grad_{int(scaler)}"] = y.grad
    xs = i["x_seq"].clone().requires_grad_(True)
    o = U.gather_seq_scatter_heads(xs, seq_dim=1, head_dim=2)
    (o * torch.arange(o.numel(), dtype=torch.float64).view_as(o)).sum().backward()
    res["gather_seq_grad"] = xs.grad
    U.set_ulysses_sequence_parallel_group(None)
    np.savez(os.path.join(out_dir, f"g{rank}.npz"), **{k: v.numpy() for k, v in res.items()})
    dist.barrier(); dist.destroy_process_group()


def test_ulysses_utilities_equal_the_reference_modules_outputs(tmp_path):
    """tests/golden/ulysses.npz = the reference's verl/utils/ulysses.py run on two gloo ranks in the build container
    (tests/golden/make_ulysses_golden.py): pad-and-slice of the token stream, both all-to-all directions (with and without the unpadded
    length), the gather of the per-slice outputs with and without the gradient scaler, and the backward of each — equal bit for bit."""
    world = 2
    mp.spawn(_golden_worker, args=(world, 29593, str(tmp_path)), nprocs=world)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ulysses.npz"))
    n = 0
    for r in range(world):
        got = np.load(tmp_path / f"g{r}.npz")
        keys = [k[len(f"r{r}_"):] for k in gold.files if k.startswith(f"r{r}_")]
        assert sorted(keys) == sorted(got.files)
        for k in keys:
            want = gold[f"r{r}_{k}"]
            assert got[k].shape == want.shape and got[k].dtype == want.dtype and np.array_equal(got[k], want), (r, k)
            n += 1
    assert n == 24


class _StubEngine:
    """Stands in for PolicyEngine on a box without a GPU: records how many rows it was handed and returns a function of the row ids."""

    def __init__(self):
        self.rows_seen = []
        self.last_prompt_cache_hit = False

    def compute_log_prob(self, d, temperature, *a, **k):
        ids = d["responses"]
        self.rows_seen.append(int(ids.shape[0]))
        return ids.float() * 0.5

    def update_policy(self, d, temperature):
        self.rows_seen.append(int(d["responses"].shape[0]))
        assert d["old_log_probs"].shape == d["responses"].shape and d["ref_log_probs"].shape == d["responses"].shape
        return {"actor/pg_loss": float(d["responses"].float().mean())}


def _sp_worker_calls(rank, world, port, out_dir):
    """compute_log_probs -> union -> compute_ref_log_probs -> union -> update_actor through FSDPWorker with sp = 2, the way
    RayPPOTrainer.fit drives it (ray_trainer.py:620-680) — on the trainer's OWN batch object (SPMDWorkerGroup hands no copy)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from verl.protocol import DataProto
    from verl.trainer.config import load_config
    from verl.workers import fsdp_workers as FW
    cfg = load_config(["worker.actor.ulysses_sequence_parallel_size=2", "worker.actor.global_batch_size=4", "worker.rollout.n=1",
                       "worker.actor.micro_batch_size_per_device_for_update=2", "worker.actor.micro_batch_size_per_device_for_experience=2",
                       "worker.actor.model.model_path=random:tiny"])
    cfg.deep_post_init()
    w = FW.FSDPWorker(cfg.worker, "actor_rollout_ref")
    assert w.sp_size == 2 and w.sp_group is not None
    w.actor, w.ref_policy = _StubEngine(), _StubEngine()

    class _Flops:
        def estimate_flops(self, n, dt): return 1.0, 1.0
    w.flops_counter = _Flops()
    FW.torch.cuda.reset_peak_memory_stats = lambda *a, **k: None            # no GPU on this box: the worker's memory bookkeeping is not under test
    FW.torch.cuda.synchronize = lambda *a, **k: None
    FW.torch.cuda.max_memory_allocated = lambda *a, **k: 0
    FW.torch.cuda.max_memory_reserved = lambda *a, **k: 0
    N, R = 4, 3
    resp = (torch.arange(N)[:, None] * 10 + torch.arange(R)[None] + 1000 * rank)
    batch = DataProto.from_dict({"responses": resp, "response_mask": torch.ones(N, R, dtype=torch.int64)},
                                non_tensors={"uid": np.array([f"r{rank}_{i}" for i in range(N)], dtype=object)})
    batch.meta_info["global_token_num"] = [R] * N
    old = w.compute_log_probs(batch)
    assert len(batch) == N and len(old) == N, (len(batch), len(old))                     # the caller's batch keeps ITS rows
    assert torch.equal(batch.batch["responses"], resp) and list(batch.non_tensor_batch["uid"]) == [f"r{rank}_{i}" for i in range(N)]
    batch = batch.union(old)                                                              # ray_trainer.py:636 (raised before the fix: 8 rows vs 4)
    ref = w.compute_ref_log_probs(batch)
    assert len(batch) == N and len(ref) == N
    batch = batch.union(ref)
    batch.batch["advantages"] = torch.zeros(N, R)
    m = w.update_actor(batch)
    assert len(batch) == N
    res = dict(old=old.batch["old_log_probs"].clone(), ref=ref.batch["ref_log_probs"].clone(), rows_actor=w.actor.rows_seen,
               rows_ref=w.ref_policy.rows_seen, pg=float(m.non_tensor_batch["actor/pg_loss"][0]), resp=resp)
    torch.save(res, os.path.join(out_dir, f"w{rank}.pt"))
    dist.barrier(); dist.destroy_process_group()


def test_worker_methods_under_sp2_never_change_the_callers_batch(tmp_path):
    world = 2
    mp.spawn(_sp_worker_calls, args=(world, 29595, str(tmp_path)), nprocs=world)
    r = [torch.load(tmp_path / f"w{k}.pt", weights_only=False) for k in range(world)]
    both = torch.cat([r[0]["resp"], r[1]["resp"]]).float()
    for k in range(world):
        assert r[k]["rows_actor"] == [8, 8] and r[k]["rows_ref"] == [8]                   # the engines saw the whole sp group's rows ...
        assert torch.equal(r[k]["old"], r[k]["resp"].float() * 0.5) and torch.equal(r[k]["ref"], r[k]["resp"].float() * 0.5)   # ... each rank got ITS rows back
        assert abs(r[k]["pg"] - float(both.mean())) < 1e-3
