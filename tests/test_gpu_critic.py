"""The critic (/root/reference/verl/workers/critic/dp_critic.py, adv_estimator = gae): value head and value loss kernels
against torch, the engine's values / gradients on the tiny model against the fp32 oracle restatement of dp_critic._forward_micro_batch
(oracle.qwen25vl.response_values) with autograd through verl.trainer.core_algos.compute_value_loss (golden-pinned to the reference's),
and update_critic as a loop.  The reference cannot construct this model (transformers has no token-classification class for
qwen2_5_vl): the head follows HF's *ForTokenClassification pattern — score = Linear(H, 1) on the final-norm hidden state."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tiny  # noqa: E402
from oracle import qwen25vl as Q  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    from spatialthinker_amd import ops as o
    return o


def _bf(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16()


def test_value_head_kernels_vs_torch(ops):
    rs = np.random.RandomState(0)
    T, H = 77, 256
    hn, w, b = _bf(rs.standard_normal((T, H))), _bf(rs.standard_normal(H) * 0.1), _bf([0.25] + [0] * 7)
    v = ops.value_head_fwd(hn.cuda(), w.cuda(), b.cuda()).cpu()
    want = (hn.float() @ w.float() + 0.25).bfloat16().float()              # nn.Linear in bf16: one rounding of the fp32 sum
    assert float((v - want).abs().max()) <= 2 ** -7 * float(want.abs().max())
    assert torch.equal(v, v.bfloat16().float())                            # the head's output is a bf16 value
    dv = torch.from_numpy(rs.standard_normal(T).astype(np.float32))
    dw, db = torch.full((H,), 3.0, device="cuda"), torch.full((8,), 5.0, device="cuda")
    dhn = ops.value_head_bwd(hn.cuda(), w.cuda(), dv.cuda(), dw, db).cpu()
    assert float((dhn.float() - (dv[:, None] * w.float()[None, :]).bfloat16().float()).abs().max()) == 0.0
    assert float((dw.cpu() - 3.0 - dv @ hn.float()).abs().max()) <= 1e-4 * float((dv @ hn.float()).abs().max())
    assert abs(float(db[0]) - 5.0 - float(dv.sum())) <= 1e-4 and float((db[1:] - 5.0).abs().max()) == 0.0


@pytest.mark.parametrize("clip", [0.5, 0.05])
def test_value_loss_kernel_vs_reference_formula_autograd(ops, clip):
    from verl.trainer import core_algos
    rs = np.random.RandomState(int(clip * 100))
    B, R = 6, 11
    vp = torch.from_numpy(rs.standard_normal((B, R)).astype(np.float32)).requires_grad_(True)
    ret = torch.from_numpy(rs.standard_normal((B, R)).astype(np.float32))
    val = vp.detach() + torch.from_numpy((rs.standard_normal((B, R)) * 0.1).astype(np.float32))
    val[0, :3] = vp.detach()[0, :3] + clip                                 # exactly on the clamp bound
    mask = torch.from_numpy((rs.uniform(size=(B, R)) < 0.7).astype(np.int64))
    accum = 4.0
    loss, frac = core_algos.compute_value_loss(vp, ret, val, mask, clip)
    (loss / accum).backward()
    g, met = ops.value_loss(vp.detach().reshape(-1).cuda(), ret.reshape(-1).cuda(), val.reshape(-1).cuda(), mask.reshape(-1).cuda(),
                            cliprange_value=clip, grad_accum=accum)
    met = met.cpu()
    assert abs(float(met[0]) - float(loss.detach())) <= 1e-6 * max(1.0, abs(float(loss.detach()))) and abs(float(met[1]) - float(frac)) <= 1e-6
    assert abs(float(met[2]) - float((vp.detach() * mask).sum() / (mask.sum() + 1e-8))) <= 1e-6 and float(met[3]) == float(mask.sum())
    assert float((g.cpu().view(B, R) - vp.grad).abs().max()) <= 1e-7


@pytest.fixture(scope="module")
def env(golden_dir):
    import dataclasses
    from spatialthinker_amd import model as mdl
    z = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    cfg = dataclasses.replace(mdl.VLConfig(**tiny.TINY), value_head=True)
    params = dict(tiny.make_params())
    rs = np.random.RandomState(5)
    params["score.weight"] = _bf(rs.standard_normal((1, cfg.hidden_size)) * 0.05).float().numpy()
    params["score.bias"] = np.asarray([0.125], dtype=np.float32)
    params.pop("lm_head.weight", None)
    store = mdl.ParamStore(cfg, trainable=True)
    store.load_hf_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return z, cfg, params, store, mdl.Qwen25VL(cfg, store), tiny.make_batch()


def test_critic_layout_roundtrip(env):
    z, cfg, params, store, eng, batch = env
    back = store.export_hf()
    assert "lm_head.weight" not in back
    for k, v in params.items():
        assert torch.equal(back[k].float().cpu().reshape(-1), torch.from_numpy(v).reshape(-1)), k


def test_values_and_gradients_vs_fp32_oracle(env, measured):
    """values within a bf16 evaluation's noise of the fp32 oracle; gradients of the clipped value loss vs autograd through the oracle."""
    from verl.trainer import core_algos
    z, cfg, params, store, eng, batch = env
    R = batch["R"]
    b = eng.stage(batch["input_ids"], batch["attention_mask"], z["position_ids"], R, batch["pixel_values"], batch["image_grid_thw"])
    v = eng.values(b).cpu()
    p32 = {k: torch.from_numpy(v_).clone().requires_grad_(k.endswith(("score.weight", "score.bias", "q_proj.weight", "gate_proj.weight", "norm.weight")))
           for k, v_ in params.items()}
    ocfg = Q.VLConfig(**tiny.TINY)
    vo = Q.response_values(p32, ocfg, torch.from_numpy(batch["input_ids"]), torch.from_numpy(batch["attention_mask"]), torch.from_numpy(z["position_ids"]), R,
                           torch.from_numpy(batch["pixel_values"]), batch["image_grid_thw"])
    am = torch.from_numpy(batch["attention_mask"])[:, -R - 1:-1]
    m = am.bool()
    scale = float(vo.detach()[m].abs().max())
    err = float((v[m] - vo.detach()[m]).abs().max())
    print(f"values: max |dv| vs fp32 oracle {err:.5f} at scale {scale:.3f}")
    measured("critic_values_max_abs_err_over_scale", err / scale)
    assert err <= 0.017 * scale                                             # measured 0.0113 (bf16 backbone: the log-prob test's noise level)
    rs = np.random.RandomState(3)
    old = (vo.detach() + torch.from_numpy((rs.standard_normal(vo.shape) * 0.3).astype(np.float32)))
    ret = torch.from_numpy(rs.standard_normal(vo.shape).astype(np.float32))
    loss, _ = core_algos.compute_value_loss(vo, ret, old, am, 0.2)
    loss.backward()
    store.grad.zero_()
    dv = lambda a, dt=torch.float32: torch.as_tensor(a).to("cuda", dt)
    vp, met = eng.value_forward_backward(b, dict(values=dv(old), returns=dv(ret), action_mask=dv(am, torch.int64)), cliprange_value=0.2, grad_accum=1.0)
    assert abs(float(met[0]) - float(loss)) <= 0.02 * abs(float(loss)) + 1e-3
    grads = store.export_hf(store.g)
    worst = 0.0
    for n_, t in p32.items():
        if t.grad is None:
            continue
        got = grads[n_].float().cpu().reshape(t.grad.shape)
        rel = float(np.linalg.norm(got.numpy() - t.grad.numpy()) / max(np.linalg.norm(t.grad.numpy()), 1e-12))
        print(f"  grad {n_}: relative L2 error {rel:.4f}")
        worst = max(worst, rel)
    measured("critic_grad_rel_l2_worst", worst)
    assert worst <= 0.024                                                   # measured 0.9-1.6 % over ViT, LM, norm and head tensors


def test_update_critic_loop_fits_fixed_returns(env):
    """CriticEngine.compute_values / update_critic (dp_critic.py:140-225): mini x micro loop, clip, AdamW; on a fixed batch the value loss
    must fall and the values move towards the returns."""
    import dataclasses
    from spatialthinker_amd import model as mdl
    from spatialthinker_amd.actor import ActorHyper, CriticEngine
    z, cfg, params, _, _, batch = env
    store = mdl.ParamStore(cfg, trainable=True)
    store.load_hf_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    store.refresh_transposes()
    N = batch["input_ids"].shape[0]
    eng = CriticEngine(cfg, store, ActorHyper(micro_batch_size_per_device_for_update=max(1, N // 2), global_batch_size_per_device=N, lr=5e-5, max_grad_norm=1.0,
                                              cliprange_value=10.0, weight_decay=0.0))
    eng.sched_steps = 1
    R = batch["R"]
    counts, off, mm = batch["patch_counts"], 0, []
    for i, cnt in enumerate(counts):                                          # one image per sample, as the dataset hands them over
        mm.append({"pixel_values": torch.from_numpy(batch["pixel_values"][off:off + cnt]), "image_grid_thw": torch.from_numpy(batch["image_grid_thw"][i:i + 1])})
        off += cnt
    mm_arr = np.empty(len(mm), dtype=object)
    mm_arr[:] = mm
    text = dict(input_ids=torch.from_numpy(batch["input_ids"]), attention_mask=torch.from_numpy(batch["attention_mask"]),
                position_ids=torch.from_numpy(z["position_ids"]), responses=torch.from_numpy(batch["input_ids"][:, -R:]), multi_modal_inputs=mm_arr)
    v0 = eng.compute_values(text).cpu()
    am = CriticEngine.action_mask(batch["attention_mask"], R)
    assert torch.equal(v0 * (1 - am), torch.zeros_like(v0))                 # masked positions are zero (dp_critic.py:174)
    returns = (v0 + 1.0) * am                                               # a target one unit above the current values
    losses = []
    for _ in range(16):
        text["values"], text["returns"] = eng.compute_values(text).cpu(), returns
        met = eng.update_critic(text)
        losses.append(float(np.mean(met["critic/vf_loss"])))
        assert set(met) >= {"critic/vf_loss", "critic/vf_clipfrac", "critic/vpred_mean", "critic/grad_norm", "critic/lr"}
    v1 = eng.compute_values(text).cpu()
    print("vf_loss", [round(x, 4) for x in losses], "mean value", float(v0[am.bool()].mean()), "->", float(v1[am.bool()].mean()))
    assert min(losses[-3:]) < 0.5 * losses[0] and float((v1 - v0)[am.bool()].mean()) > 0.3
