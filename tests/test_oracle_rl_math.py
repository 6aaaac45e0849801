"""Pins oracle/rl_math.py and oracle/positions.py against the golden vectors produced from the
reference's own functions (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import positions as P
from oracle import rl_math as M


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "rl_math.npz"), allow_pickle=True)


@pytest.mark.parametrize("G", [4, 8, 16])
def test_grpo_advantage(g, G):
    adv, ret = M.grpo_outcome_advantage(g[f"grpo{G}_rewards"], g[f"grpo{G}_mask"], g[f"grpo{G}_uid"].tolist())
    np.testing.assert_array_equal(adv, g[f"grpo{G}_adv"])          # bit-exact fp32
    np.testing.assert_array_equal(ret, adv)
    # zero-variance group: the reference emits fp32 rounding noise / 1e-6, not exactly 0
    uid = g[f"grpo{G}_uid"]
    assert np.all(np.abs(adv[uid == 0]) < 0.2)


def test_policy_loss_and_grad(g):
    out = M.policy_loss(g["pl_old"], g["pl_new"], g["pl_adv"], g["pl_mask"], 0.2, 0.3, 3.0)
    np.testing.assert_allclose(np.array(out), g["pl_out"], rtol=2e-6, atol=1e-7)
    assert g["pl_out"][1] > 0 and g["pl_out"][2] > 0               # both clip branches exercised
    grad = M.policy_loss_grad(g["pl_old"], g["pl_new"], g["pl_adv"], g["pl_mask"], 0.2, 0.3, 3.0)
    np.testing.assert_allclose(grad, g["pl_grad"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("kind", ["kl", "abs", "mse", "low_var_kl", "chi2"])
def test_kl(g, kind):
    np.testing.assert_allclose(M.kl_penalty(g["pl_new"], g["pl_ref"], kind), g[f"kl_{kind}"], rtol=1e-6, atol=1e-6)  # exp(d)-d-1 cancels: 1-ulp exp differences
    m = g["pl_mask"].astype(np.float32)
    grad = M.kl_penalty_grad(g["pl_new"], g["pl_ref"], kind) * m / (m.sum() + 1e-8)
    np.testing.assert_allclose(grad, g[f"kl_{kind}_mean_grad"], rtol=1e-5, atol=2e-9)


def test_micro_batch_loss(g):
    met, grad = M.actor_micro_batch_loss(g["pl_new"], g["pl_old"], g["pl_ref"], g["pl_adv"], g["pl_mask"],
                                         kl_kind="low_var_kl", kl_coef=1e-2, grad_accum=4)
    np.testing.assert_allclose([met["pg_loss"], met["kl_loss"], met["entropy_loss"]], g["mb_metrics"], rtol=3e-6)
    np.testing.assert_allclose(grad, g["mb_grad"], rtol=1e-5, atol=1e-9)


def test_response_mask(g):
    np.testing.assert_array_equal(M.response_mask(g["rm_ids"], 3), g["rm_single"])
    np.testing.assert_array_equal(M.response_mask(g["rm_ids"], [3, 7]), g["rm_multi"])


def test_log_probs(g):
    import torch

    z = torch.from_numpy(g["lp512_logits_bf16_bits"]).view(torch.bfloat16).float().numpy()
    np.testing.assert_allclose(M.log_probs_from_logits(z, g["lp512_labels"]), g["lp512_logp"], rtol=0, atol=2e-6)
    V = 152064
    rs = np.random.RandomState(V)
    zz = M.bf16_round((rs.standard_normal((5, V)) * 3).astype(np.float32))
    lab = rs.randint(0, V, size=5)
    np.testing.assert_array_equal(lab, g["lp152064_labels"])
    np.testing.assert_allclose(M.log_probs_from_logits(zz, lab), g["lp152064_logp"], rtol=0, atol=2e-5)  # reference sums 152064 exps in fp32; oracle in fp64


def test_adamw_kahan_bit_exact(golden_dir):
    a = np.load(os.path.join(golden_dir, "adamw.npz"))
    opt = M.AdamWKahanBF16(lr=1e-6, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, scalar_mode="cpu")
    p = a["p0"]
    sched_steps = 0
    for k in range(6):
        lr = M.constant_schedule_lr(1e-6, 0, sched_steps)
        assert lr == a["lrs"][k]
        p = opt.step(p, a[f"g{k}"], lr=lr)
        for name, mine in (("p", p), ("m", opt.m), ("v", opt.v), ("c", opt.c)):
            np.testing.assert_array_equal(mine, a[f"{name}{k + 1}"], err_msg=f"{name} step {k + 1}")
        if k % 2 == 1:
            sched_steps += 1
    assert a["lrs"][0] == 0.0 and a["lrs"][2] == 1e-6          # first update_actor call trains at lr 0
    assert np.any(a["p6"] != a["p0"])


def test_rope_index(golden_dir):
    z = np.load(os.path.join(golden_dir, "positions.npz"))
    for i in range(4):
        thw = z[f"rope{i}_thw"]
        pos = P.mrope_position_ids(z[f"rope{i}_ids"], thw if len(thw) else None, z[f"rope{i}_mask"],
                                   image_token_id=990, vision_start_token_id=991)
        np.testing.assert_array_equal(pos, z[f"rope{i}_pos"])


def test_vision_indices(golden_dir):
    z = np.load(os.path.join(golden_dir, "positions.npz"))
    for i in range(4):
        thw = z[f"vwin{i}_thw"]
        for win in (56, 112):
            idx, cu = P.vision_window_index(thw, window_size=win)
            np.testing.assert_array_equal(idx, z[f"vwin{i}_{win}_idx"])
            np.testing.assert_array_equal(cu, z[f"vwin{i}_{win}_cu"])
        np.testing.assert_array_equal(P.vision_position_ids(thw), z[f"vwin{i}_pos"])


def test_balanced_partitions(golden_dir):
    for case in json.load(open(os.path.join(golden_dir, "balance.json"))):
        assert P.balanced_partitions(case["lens"], case["k"]) == case["parts"]


def test_position_continuation_and_packing():
    p = np.array([[[0, 0, 0, 1, 2], [0, 0, 0, 1, 5], [0, 0, 0, 1, 7]]])
    out = P.continue_position_ids(p, 3)
    assert out.shape == (1, 3, 8)
    np.testing.assert_array_equal(out[0, :, 5:], [[3, 4, 5], [6, 7, 8], [8, 9, 10]])
    np.testing.assert_array_equal(P.packed_cu_seqlens_from_positions(np.array([0, 1, 2, 0, 1, 0, 1, 2, 3])), [0, 3, 5, 9])
