"""Ulysses sequence parallelism in the engine (reference verl/utils/ulysses.py:63-298, flash_attention_utils.py:98-106,
146-148, dp_actor.py:107-133): two ranks share the one GPU of the box and exchange over gloo (RCCL refuses two ranks on one device) —
every packed pass is cut into two row slices, the all-to-all pair trades rows for heads around the HIP attention kernels, and the
result must be what ONE rank computes on the whole pass: log-probs, and the SUM of the two ranks' gradients."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

import tiny  # noqa: E402

SP_CFG = dict(tiny.TINY, hidden_size=512, intermediate_size=768, num_heads=4, num_kv_heads=2)      # head_dim 128, heads divisible by 2


def _batch_and_inputs(eng):
    batch = tiny.make_batch()
    R = batch["R"]
    from oracle import positions as P
    ids, mask = batch["input_ids"], batch["attention_mask"]
    Pn = batch["P"]
    pos = np.zeros((2, 3, ids.shape[1]), dtype=np.int64)
    for i in range(2):
        pp = P.mrope_position_ids(ids[i, :Pn], batch["image_grid_thw"][i:i + 1], mask[i, :Pn], image_token_id=tiny.TINY["image_token_id"],
                                  vision_start_token_id=tiny.TINY["vision_start_token_id"])
        pp[:, mask[i, :Pn] == 0] = 0
        pos[i, :, :Pn] = pp
        pos[i, :, Pn:] = pp[:, -1:] + np.arange(1, R + 1)
    b = eng.stage(ids, mask, pos, R, batch["pixel_values"], batch["image_grid_thw"])
    rs = np.random.RandomState(3)
    rmask = mask[:, -R:]
    dv = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dt)
    old = (rs.standard_normal(rmask.shape) * 0.1 - 6.5).astype(np.float32)
    adv = rs.standard_normal((2, 1)).astype(np.float32).repeat(R, 1) * rmask
    li = dict(old_log_probs=dv(old), ref_log_probs=dv(old), advantages=dv(adv), response_mask=dv(rmask, torch.int64))
    return b, li


def _run(eng, group):
    eng.set_sequence_parallel(group)
    b, li = _batch_and_inputs(eng)
    lp = eng.log_probs(b, 1.0).clone()
    eng.p.grad.zero_()
    lp2, metrics = eng.forward_backward(b, li, 1.0, clip_low=0.2, clip_high=0.3, clip_dual=3.0, kl_kind="low_var_kl", kl_coef=1e-2, grad_accum=1.0)
    torch.cuda.synchronize()
    return lp, lp2.clone(), metrics.clone(), eng.p.grad.clone()


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spatialthinker_amd import model as mdl
    cfg = mdl.VLConfig(**SP_CFG)
    store = mdl.ParamStore(cfg, trainable=True)
    store.init_random(5)
    eng = mdl.Qwen25VL(cfg, store)
    group = dist.new_group(list(range(world)))
    lp, lp2, metrics, grad = _run(eng, group)
    g_host = grad.cpu()
    dist.all_reduce(g_host, group=group)                          # the sp ranks' gradients ADD UP (partial sums over their token slices)
    res = {"lp": lp.cpu(), "lp2": lp2.cpu(), "metrics": metrics.cpu(), "grad_sum": g_host, "grad_own_norm": float(grad.float().norm())}
    if rank == 0:                                                 # ground truth in the same process: the same engine without the group
        lp_1, lp2_1, metrics_1, grad_1 = _run(eng, None)
        res.update(lp_1=lp_1.cpu(), lp2_1=lp2_1.cpu(), metrics_1=metrics_1.cpu(), grad_1=grad_1.cpu(), layout={k: (int(o), int(np.prod(s)))
                   for (k, s), o in zip(store.layout.items(), [store.offsets[k] for k in store.layout])})
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier(); dist.destroy_process_group()


def test_two_sequence_parallel_ranks_reproduce_the_single_rank_pass(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, 29593, str(tmp_path)), nprocs=world)
    r = [torch.load(tmp_path / f"r{k}.pt", weights_only=False) for k in range(world)]
    one = r[0]
    for k in range(world):
        assert torch.equal(r[k]["lp"], r[0]["lp"]) and torch.equal(r[k]["metrics"], r[0]["metrics"])      # every rank holds the full result
        d = float((r[k]["lp"] - one["lp_1"]).abs().max())
        assert d < 0.02, d                                        # same arithmetic on other GEMM row counts: bf16 noise at most
        assert float((r[k]["lp2"] - one["lp2_1"]).abs().max()) < 0.02
        torch.testing.assert_close(r[k]["metrics"], one["metrics_1"], rtol=2e-2, atol=2e-3)
    g2, g1 = r[0]["grad_sum"].double(), one["grad_1"].double()
    assert r[0]["grad_own_norm"] < 0.999 * float(g2.norm()) and r[1]["grad_own_norm"] > 0     # each rank really holds only a part
    worst = 0.0
    for name, (off, n) in one["layout"].items():
        a, b_ = g2[off:off + n], g1[off:off + n]
        if float(b_.norm()) < 1e-12:
            assert float(a.norm()) < 1e-6, name
            continue
        worst = max(worst, float((a - b_).norm() / b_.norm()))
    assert worst < 0.03, worst                                    # per-tensor relative L2 of the summed gradient vs the single-rank gradient
