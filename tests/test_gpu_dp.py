"""Data-parallel gradient exchange on the GPU: the slices update_policy hands to the all-reduce while the last
backward pass is still running must really be final, and the overlapped exchange must leave exactly the weights the plain
(after-backward) exchange leaves.  A 1-GPU box can only build a 1-rank RCCL group: it exercises the asynchronous collectives,
their stream ordering and the bookkeeping; the arithmetic over ranks is covered by the 2-rank gloo tests (test_dp_gloo.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

import tiny  # noqa: E402


def _data(z, rs):
    """4 rows (the two tiny samples twice, second copy with other responses) in update_policy's input layout."""
    batch = tiny.make_batch()
    R = batch["R"]
    ids = np.concatenate([batch["input_ids"], batch["input_ids"]], 0).copy()
    mask = np.concatenate([batch["attention_mask"], batch["attention_mask"]], 0).copy()
    pos = np.concatenate([z["position_ids"], z["position_ids"]], 0)
    S = ids.shape[1]
    ids[2:, S - R:] = rs.randint(3, 900, (2, R))
    mask[2, S - 3:] = 0
    n0 = int(batch["patch_counts"][0])
    px = [torch.from_numpy(batch["pixel_values"][:n0]), torch.from_numpy(batch["pixel_values"][n0:])]
    gr = [batch["image_grid_thw"][0:1], batch["image_grid_thw"][1:2]]
    mm = np.array([{"pixel_values": px[i % 2], "image_grid_thw": gr[i % 2]} for i in range(4)], dtype=object)
    rmask = mask[:, -R:]
    old = (rs.standard_normal((4, R)) * 0.1 - 6.0).astype(np.float32)
    adv = rs.standard_normal((4, 1)).astype(np.float32).repeat(R, 1) * rmask
    t = torch.from_numpy
    return dict(input_ids=t(ids), attention_mask=t(mask), position_ids=t(pos), responses=t(ids[:, -R:].copy()), multi_modal_inputs=mm,
                old_log_probs=t(old), ref_log_probs=t(old.copy()), advantages=t(adv))


@pytest.fixture(scope="module")
def setup(golden_dir):
    from spatialthinker_amd import model as mdl
    z = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    cfg = mdl.VLConfig(**tiny.TINY)
    params = {k: torch.from_numpy(v) for k, v in tiny.make_params().items()}
    return z, cfg, params


def _engine(cfg, params, **hyper):
    from spatialthinker_amd import model as mdl
    from spatialthinker_amd.actor import ActorHyper, PolicyEngine
    store = mdl.ParamStore(cfg, trainable=True)
    store.load_hf_state_dict(params)
    store.refresh_transposes()
    return PolicyEngine(cfg, store, ActorHyper(micro_batch_size_per_device_for_update=2, global_batch_size_per_device=4, lr=1e-3,
                                               lr_warmup_steps=0, **hyper))


def test_slices_announced_during_backward_are_final_and_disjoint(setup):
    z, cfg, params = setup
    eng = _engine(cfg, params)
    st = eng.store
    data = _data(z, np.random.RandomState(5))
    R = data["responses"].shape[1]
    b = eng._stage(data, slice(0, 4))
    dv = lambda k, dt=torch.float32: data[k].to("cuda", dt)
    li = dict(old_log_probs=dv("old_log_probs"), ref_log_probs=dv("ref_log_probs"), advantages=dv("advantages"),
              response_mask=data["attention_mask"][:, -R:].to("cuda", torch.int64))
    seen = []

    def on_final(lo, hi):
        torch.cuda.synchronize()
        seen.append((lo, hi, st.grad[lo:hi].clone()))

    st.grad.zero_()
    eng.model.forward_backward(b, li, 1.0, clip_low=0.2, clip_high=0.3, clip_dual=3.0, kl_kind="low_var_kl", kl_coef=1e-2,
                               grad_accum=1.0, loss_rows=2, on_final=on_final)
    torch.cuda.synchronize()
    assert len(seen) == cfg.num_layers + 1
    spans = sorted((lo, hi) for lo, hi, _ in seen)
    assert all(a[1] <= b_[0] for a, b_ in zip(spans, spans[1:])), spans               # disjoint
    assert spans[0][0] == st.offsets["l.0.in_norm"] and spans[-1][1] == st.numel       # every LM layer + final norm + head
    assert all(a[1] == b_[0] for a, b_ in zip(spans, spans[1:]))                       # contiguous: nothing in between is skipped
    for lo, hi, snap in seen:
        assert float(snap.abs().max()) > 0
        assert torch.equal(snap, st.grad[lo:hi]), (lo, hi)                             # nothing wrote there afterwards
    assert float(st.grad[:spans[0][0]].abs().max()) > 0                                # ViT + embedding: left for finish()
    st.grad.zero_()


@pytest.mark.parametrize("exchange", ["allreduce", "reduce_scatter"])
def test_overlapped_exchange_leaves_the_same_weights_as_the_plain_one(setup, exchange):
    """exchange = reduce_scatter: SURVEY §5.8's direct schedule (all-to-all of shards, shard sums and all-gathers on a side stream)
    through RCCL — on a 1-rank group the collectives are copies, so the weights must be bit-identical to the all-reduce's; the
    stream ordering (side stream waits for the gradient kernels, the optimizer waits for the side stream) is what is under test."""
    z, cfg, params = setup
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        out = []
        for overlap in (False, True):
            eng = _engine(cfg, params, allreduce_bucket_mb=1, grad_exchange=exchange)
            eng.fuse_micro_batches = 1                       # two passes per optimizer step: only the second may announce slices
            eng.sync_grads, eng.overlap_allreduce = True, overlap
            eng.sched_steps = 1                              # past the lr = 0 first call
            early = []
            if overlap:
                red = eng.grad_reducer()
                ready0 = red.ready
                red.ready = lambda lo, hi: (early.append((lo, hi, eng.store.grad.abs().sum().item())), ready0(lo, hi))[1]
            data = _data(z, np.random.RandomState(5))
            m = eng.update_policy(data, 1.0)
            torch.cuda.synchronize()
            out.append((eng.store.flat.clone(), m["actor/grad_norm"]))
            if overlap:
                assert len(early) == cfg.num_layers + 1      # one optimizer step, announcements from its LAST pass only
                assert red.sent == [] and red.works == []
        assert out[0][1] == out[1][1]
        assert torch.equal(out[0][0], out[1][0])
        assert not torch.equal(out[0][0], _engine(cfg, params).store.flat)            # the step did move the weights
    finally:
        dist.destroy_process_group()
