"""Training options of FSDPWorker._build_model_optimizer / DataParallelPPOActor._optimizer_step on the GPU engine:
optim.strategy=adamw (torch.optim.AdamW(fused=True) semantics), freeze_vision_tower, the non-finite gradient-norm skip, and
save -> load -> next-step-bit-identical resume through the worker's checkpoint files."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tiny  # noqa: E402
from test_gpu_dp import _data, _engine  # noqa: E402


@pytest.fixture(scope="module")
def setup(golden_dir):
    from spatialthinker_amd import model as mdl
    z = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    return z, mdl.VLConfig(**tiny.TINY), {k: torch.from_numpy(v) for k, v in tiny.make_params().items()}


def _mismatch(got: torch.Tensor, want: torch.Tensor):
    """(fraction of bit-identical elements, number of elements further apart than one bf16 step of the larger magnitude — or, where a
    difference of nearly equal terms lands next to zero, than 2^-20 of the tensor's scale)"""
    g, w = got.float(), want.float()
    tol = torch.maximum(torch.maximum(g.abs(), w.abs()) * 2.0 ** -7, w.abs().max() * 2.0 ** -20)
    return float((got.view(torch.int16) == want.view(torch.int16)).float().mean()), int(((g - w).abs() > tol).sum())


def test_plain_adamw_matches_torch_fused_adamw():
    """st_adamw_step vs torch.optim.AdamW(fused=True) on bf16 parameters (the optimizer the reference builds for
    optim.strategy=adamw, fsdp_workers.py:284-291), 4 steps incl. weight decay and a clip coefficient.  Tolerance: parameters,
    exp_avg and exp_avg_sq within one bf16 step after every step and >= 99.9 % of the elements bit-identical (the kernel follows
    torch's fp32/double operation order; the residue is its lerp/fma contraction; measured: step 1 bit-identical everywhere)."""
    from spatialthinker_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    n = 1 << 20
    p0 = (torch.randn(n, device="cuda", generator=g) * 0.05).bfloat16()
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p_ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, fused=True)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    scale = torch.tensor([0.37], device="cuda")
    for t in range(1, 5):
        grad = torch.randn(n, device="cuda", generator=g) * (0.02 * t)
        p_ref.grad = (grad * scale).bfloat16()
        opt.step()
        ops.adamw_step_(p, grad, m, v, t=t, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=scale)
        st = opt.state[p_ref]
        for name, got, want in (("p", p, p_ref.data), ("m", m, st["exp_avg"]), ("v", v, st["exp_avg_sq"])):
            exact, far = _mismatch(got, want)
            print(f"step {t} {name}: bit-identical {exact:.5f}, beyond one bf16 step: {far}")
            assert far == 0 and exact >= 0.999, (t, name, far, exact)


def test_fp32_master_adamw_matches_torch_fused_adamw_on_fp32_parameters(measured):
    """st_adamw_master_step vs torch.optim.AdamW(fused=True) on FP32 parameters — the reference's default actor (torch_dtype unset:
    fp32 shards under MixedPrecision(param_dtype=bf16), fsdp_workers.py:186-189, 284-291).  Five steps with weight decay and a clip
    coefficient: exp_avg / exp_avg_sq bit-identical to torch's, the parameters bit-identical for 97-99 % of the elements and one fp32
    ulp apart for the rest (the final division's rounding), and the bf16 working copy exactly the rounding of the master."""
    from spatialthinker_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    n = (1 << 20) + 3                                             # not a multiple of 4: the scalar tail
    p0 = torch.randn(n, device="cuda", generator=g) * 0.05
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p_ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, fused=True)
    master, pw = p0.clone(), p0.bfloat16()
    m, v = torch.zeros_like(p0), torch.zeros_like(p0)
    scale = torch.tensor([0.37], device="cuda")
    worst = 0.0
    for t in range(1, 6):
        grad = torch.randn(n, device="cuda", generator=g) * (0.02 * t)
        p_ref.grad = grad * scale
        opt.step()
        ops.adamw_master_step_(master, pw, grad, m, v, t=t, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=scale)
        st = opt.state[p_ref]
        for name, got, want in (("p", master, p_ref.data), ("m", m, st["exp_avg"]), ("v", v, st["exp_avg_sq"])):
            rel = float(((got - want).abs() / (want.abs() + 1e-12 * float(want.abs().max()) + 1e-30)).max())
            frac = float((got == want).float().mean())
            print(f"step {t} {name}: bit-identical {frac:.5f}, max relative difference {rel:.3g}")
            worst = max(worst, float((got - want).abs().max() / want.abs().max()))
            if name != "p":
                assert torch.equal(got, want), (t, name)              # exp_avg / exp_avg_sq: bit-identical (measured over 5 steps)
            else:                                                     # parameters: 98.8 % -> 96.9 % bit-identical over 5 steps, the rest one ulp
                assert frac >= 0.95 and float((got - want).abs().max()) <= 4 * 2.0 ** -24 * float(want.abs().max()), (t, frac)
        assert torch.equal(pw, master.bfloat16())
    measured("adamw_master_vs_torch_fused_max_abs_over_scale", worst)


def test_fp32_master_mode_runs_through_update_policy(setup):
    """ParamStore.enable_fp32_master(): update_policy updates the master, the bf16 working copy follows it exactly, and the master keeps
    the bits below the bf16 grid (what it is for)."""
    z, cfg, params = setup
    eng = _engine(cfg, params, optim_strategy="adamw")
    eng.store.enable_fp32_master()
    eng.sched_steps = 1
    m0 = eng.store.master.clone()
    eng.update_policy(_data(z, np.random.RandomState(5)), 1.0)
    st = eng.store
    assert st.m.dtype == torch.float32 and st.v.dtype == torch.float32 and st.c is None
    assert torch.equal(st.flat, st.master.bfloat16())
    assert float((st.master != m0).float().mean()) > 0.5                        # the update reached the master ...
    assert float((st.master != st.flat.float()).float().mean()) > 0.5          # ... which keeps the bits below the bf16 grid


def test_strategy_adamw_runs_through_update_policy(setup):
    z, cfg, params = setup
    eng = _engine(cfg, params, optim_strategy="adamw")
    eng.sched_steps = 1
    before = eng.store.flat.clone()
    eng.update_policy(_data(z, np.random.RandomState(5)), 1.0)
    assert not torch.equal(before, eng.store.flat) and float(eng.store.c.abs().max()) == 0.0     # no Kahan buffer in this mode
    with pytest.raises(NotImplementedError):
        e2 = _engine(cfg, params, optim_strategy="sgd")
        e2.sched_steps = 1
        e2.update_policy(_data(z, np.random.RandomState(5)), 1.0)


def test_freeze_vision_tower_leaves_the_vit_untouched(setup):
    """fsdp_workers.py:226-232: model.visual.requires_grad_(False) — no gradient, no decay, no optimizer state for the tower, while
    the language model trains exactly as it would receive the same gradients."""
    z, cfg, params = setup
    data = _data(z, np.random.RandomState(5))
    a, b = _engine(cfg, params, freeze_vision_tower=True), _engine(cfg, params)
    for e in (a, b):
        e.sched_steps = 1
    ma, mb = a.update_policy(data, 1.0), b.update_policy(data, 1.0)
    lo = a.store.offsets["embed"]
    fresh = _engine(cfg, params).store
    assert torch.equal(a.store.flat[:lo], fresh.flat[:lo])                                          # ViT weights bit-unchanged
    assert float(a.store.m[:lo].abs().max()) == 0 and float(a.store.v[:lo].abs().max()) == 0 and float(a.store.c[:lo].abs().max()) == 0
    assert not torch.equal(b.store.flat[:lo], fresh.flat[:lo])                                      # the unfrozen run does move it
    assert not torch.equal(a.store.flat[lo:], fresh.flat[lo:])
    assert ma["actor/grad_norm"][0] < mb["actor/grad_norm"][0]                                      # the ViT's share of the norm is gone
    assert ma["actor/pg_loss"] == mb["actor/pg_loss"]                                               # same forward


def test_non_finite_grad_norm_skips_the_update(setup, capsys):
    """dp_actor.py:161-166: a non-finite global norm -> message, zero_grad, NO optimizer step (weights, states, step count keep)."""
    z, cfg, params = setup
    eng = _engine(cfg, params)
    eng.sched_steps = 1
    st = eng.store
    w0, m0 = st.flat.clone(), st.m.clone()
    st.grad.normal_()
    st.grad[12345] = float("nan")
    norm = eng.optimizer_step()
    assert not np.isfinite(norm) and "Gradient norm is not finite. Skip update." in capsys.readouterr().out
    assert torch.equal(st.flat, w0) and torch.equal(st.m, m0) and eng.opt_steps == 0 and float(st.grad.abs().max()) == 0.0
    st.grad.normal_()
    st.grad[7] = float("inf")
    assert not np.isfinite(eng.optimizer_step()) and eng.opt_steps == 0
    st.grad.normal_()
    assert np.isfinite(eng.optimizer_step()) and eng.opt_steps == 1 and not torch.equal(st.flat, w0)


def test_worker_checkpoint_resume_is_bit_identical(tmp_path):
    """FSDPWorker.save_checkpoint / load_checkpoint (fsdp_workers.py:399-421 + fsdp_checkpoint_manager.py): a worker restored from
    the files continues exactly like the one that wrote them — weights, AdamW states, step counters, scheduler position and the
    rollout seed stream (same generated tokens, same updated weights)."""
    from verl.protocol import DataProto
    from verl.trainer.config import load_config
    from verl.utils.dataset import SyntheticSTVQADataset, collate_fn
    from verl.workers.fsdp_workers import FSDPWorker

    def make():
        cfg = load_config(["data.rollout_batch_size=2", "data.max_prompt_length=64", "data.max_response_length=12",
                           "worker.actor.model.model_path=random:tiny", "worker.actor.global_batch_size=2", "worker.actor.fsdp.torch_dtype=bf16",
                           "worker.actor.optim.strategy=adamw_bf16", "worker.actor.optim.lr=1.0e-3", "worker.rollout.n=2",
                           "worker.actor.micro_batch_size_per_device_for_update=2", "worker.actor.micro_batch_size_per_device_for_experience=4"])
        cfg.deep_post_init()
        w = FSDPWorker(cfg.worker, "actor_rollout_ref")
        w.init_model()
        return w

    def step(w, seed):
        ds = SyntheticSTVQADataset(w.model_config, w.tokenizer, size=8, max_prompt_length=64, seed=seed, grid=(1, 8, 8), text_tokens=(8, 12))
        b = DataProto.from_single_dict(collate_fn([ds[0], ds[1]]))
        gen = b.pop(batch_keys=["input_ids", "attention_mask", "position_ids"], non_tensor_batch_keys=["raw_prompt_ids", "multi_modal_data", "multi_modal_inputs"])
        out = w.generate_sequences(gen)
        b = b.repeat(2, interleave=True).union(out)
        b.meta_info["global_token_num"] = b.batch["attention_mask"].sum(-1).tolist()
        b = b.union(w.compute_log_probs(b)).union(w.compute_ref_log_probs(b))
        rs = np.random.RandomState(seed)
        b.batch["advantages"] = torch.from_numpy(rs.standard_normal((4, 1)).astype(np.float32)).repeat(1, 12) * b.batch["response_mask"]
        w.update_actor(b)
        return b.batch["responses"].clone()

    a = make()
    step(a, 1); step(a, 2)                                   # the first update runs at lr = 0 (scheduler quirk), the second moves the weights
    a.save_checkpoint(str(tmp_path / "actor"))
    hf = tmp_path / "actor" / "huggingface"
    for f in ("config.json", "generation_config.json", "model.safetensors"):
        assert (hf / f).exists(), f
    resp_a = step(a, 3)
    b = make()
    assert torch.equal(b.actor.store.flat, make().actor.store.flat) and not torch.equal(b.actor.store.flat, a.actor.store.flat)
    b.load_checkpoint(str(tmp_path / "actor"))
    assert b.actor.opt_steps == 2 and b.actor.sched_steps == 2 and b._gen_calls == 2
    resp_b = step(b, 3)
    assert torch.equal(resp_a, resp_b)                                                               # same rollout seed stream
    for name in ("flat", "m", "v", "c"):
        assert torch.equal(getattr(a.actor.store, name), getattr(b.actor.store, name)), name
    # the saved directory is a loadable model on its own
    from spatialthinker_amd.pretrained import load_model
    _, st2, _ = load_model(str(hf), trainable=False)
    assert st2.flat.shape == a.actor.store.flat.shape


@pytest.mark.gpu
def test_reference_actor_class_runs_the_engine_behind_the_reference_signatures():
    """verl.workers.actor.DataParallelPPOActor(config, actor_module) over the worker's engines: compute_log_prob(DataProto) equals what
    FSDPWorker.compute_log_probs / compute_ref_log_probs return for the same batch (bit for bit: the same packed passes), and
    update_policy(DataProto) returns the reference's metric keys and moves the weights."""
    from verl.protocol import DataProto
    from verl.trainer.config import load_config
    from verl.utils.dataset import SyntheticSTVQADataset, collate_fn
    from verl.workers.actor import DataParallelPPOActor
    from verl.workers.fsdp_workers import FSDPWorker
    cfg = load_config(["data.rollout_batch_size=2", "data.max_prompt_length=64", "data.max_response_length=12",
                       "worker.actor.model.model_path=random:tiny", "worker.actor.global_batch_size=2", "worker.actor.fsdp.torch_dtype=bf16",
                       "worker.actor.optim.strategy=adamw_bf16", "worker.actor.optim.lr=1.0e-3", "worker.rollout.n=2",
                       "worker.actor.micro_batch_size_per_device_for_update=2", "worker.actor.micro_batch_size_per_device_for_experience=4"])
    cfg.deep_post_init()
    w = FSDPWorker(cfg.worker, "actor_rollout_ref")
    w.init_model()
    ds = SyntheticSTVQADataset(w.model_config, w.tokenizer, size=8, max_prompt_length=64, seed=5, grid=(1, 8, 8), text_tokens=(8, 12))
    b = DataProto.from_single_dict(collate_fn([ds[0], ds[1]]))
    gen = b.pop(batch_keys=["input_ids", "attention_mask", "position_ids"], non_tensor_batch_keys=["raw_prompt_ids", "multi_modal_data", "multi_modal_inputs"])
    b = b.repeat(2, interleave=True).union(w.generate_sequences(gen))
    b.meta_info["global_token_num"] = b.batch["attention_mask"].sum(-1).tolist()
    old, ref = w.compute_log_probs(b), w.compute_ref_log_probs(b)
    b.meta_info["temperature"] = cfg.worker.rollout.temperature
    actor = DataParallelPPOActor(cfg.worker.actor, w.actor)
    ref_actor = DataParallelPPOActor(cfg.worker.ref, w.ref_policy)
    assert torch.equal(actor.compute_log_prob(b), old.batch["old_log_probs"])
    assert torch.equal(ref_actor.compute_log_prob(b), ref.batch["ref_log_probs"])
    b = b.union(old).union(ref)
    b.batch["advantages"] = torch.linspace(-1, 1, 4)[:, None].repeat(1, 12) * b.batch["response_mask"]
    w.actor.sched_steps = 1                                   # past the scheduler's lr = 0 first step
    before = w.actor.store.flat.clone()
    metrics = actor.update_policy(b)
    assert {"actor/pg_loss", "actor/pg_clipfrac_higher", "actor/pg_clipfrac_lower", "actor/ppo_kl", "actor/grad_norm"} <= set(metrics), sorted(metrics)
    assert not torch.equal(w.actor.store.flat, before)
