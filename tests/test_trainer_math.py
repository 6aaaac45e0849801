"""Host-side RL math and bookkeeping of the trainer / workers against fixtures produced by the reference's own functions
(tests/golden/make_golden.py gen_rl_extra -> rl_extra.npz; rl_math.npz for the response mask): KL penalty + controllers
(SURVEY a19), the non-GRPO estimators and value loss (f-4), FlopsCounter (a28), response mask (a4, bit-exact) and the rollout
post-processing (a2, fixture 12)."""
import os
from types import SimpleNamespace

import numpy as np
import torch

from verl.protocol import DataProto
from verl.trainer import core_algos
from verl.trainer.ray_trainer import apply_kl_penalty
from verl.utils import torch_functional as VF
from verl.utils.flops_counter import FlopsCounter
from verl.workers.rollout import assemble_rollout_batch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
Z = np.load(os.path.join(GOLD, "rl_extra.npz"))
T = lambda k: torch.from_numpy(Z[k])


def test_response_mask_product_bit_exact():
    z = np.load(os.path.join(GOLD, "rl_math.npz"))
    ids = torch.from_numpy(z["rm_ids"])
    assert np.array_equal(VF.get_response_mask(ids, 3).numpy(), z["rm_single"])
    assert np.array_equal(VF.get_response_mask(ids, [3, 7]).numpy(), z["rm_multi"])
    assert VF.get_response_mask(ids, [3, 7], dtype=torch.int32).dtype == torch.int32


def test_pad_2d_list_and_rollout_postprocess_bit_exact():
    n, pad, eos = int(Z["ro_n"][0]), int(Z["ro_pad"][0]), Z["ro_eos"].tolist()
    lens, flat = Z["ro_completions"], Z["ro_completion_tokens"]
    off = np.concatenate([[0], np.cumsum(lens)])
    comp = [flat[off[i]:off[i + 1]].tolist() for i in range(len(lens))]
    R = Z["ro_responses"].shape[1]
    resp = VF.pad_2d_list_to_length(comp, pad, max_length=R)
    assert np.array_equal(resp.numpy(), Z["ro_responses"])
    out = assemble_rollout_batch(T("ro_ids"), T("ro_mask"), T("ro_pos"), resp, n, eos)
    for k in ("prompts", "responses", "input_ids", "attention_mask", "response_mask", "position_ids"):
        assert out[k].dtype == torch.int64 and np.array_equal(out[k].numpy(), Z["ro_" + k]), k
    # text-only prompts carry (b, P) position ids
    out2 = assemble_rollout_batch(T("ro_ids"), T("ro_mask"), T("ro_pos")[:, 0], resp, n, eos)
    assert np.array_equal(out2["position_ids"].numpy(), Z["ro_position_ids"][:, 0])


def test_apply_kl_penalty_and_adaptive_controller_step():
    for kind in ("kl", "abs", "mse", "low_var_kl", "chi2"):
        ctrl = core_algos.AdaptiveKLController(init_kl_coef=0.05, target_kl=0.02, horizon=100.0)
        data = DataProto.from_dict({"token_level_scores": T("klp_scores"), "response_mask": T("klp_mask"), "old_log_probs": T("klp_old"),
                                    "ref_log_probs": T("klp_ref")})
        data, met = apply_kl_penalty(data, ctrl, kl_penalty=kind)
        np.testing.assert_allclose(data.batch["token_level_rewards"].numpy(), Z[f"klp_{kind}_rewards"], rtol=0, atol=0)
        assert met["critic/kl"] == Z[f"klp_{kind}_stats"][0] and met["critic/kl_coef"] == 0.05
        assert ctrl.kl_coef == Z[f"klp_{kind}_stats"][1]
    # without a reference policy the penalty is zero and the rewards are the scores
    data = DataProto.from_dict({"token_level_scores": T("klp_scores"), "response_mask": T("klp_mask"), "old_log_probs": T("klp_old")})
    data, met = apply_kl_penalty(data, core_algos.FixedKLController(0.3), "kl")
    assert torch.equal(data.batch["token_level_rewards"], T("klp_scores")) and met["critic/kl"] == 0.0


def test_kl_controllers_trajectory():
    ctrl = core_algos.AdaptiveKLController(init_kl_coef=0.01, target_kl=0.05, horizon=64.0)
    got = []
    for i, k in enumerate(Z["klc_kls"].tolist()):
        ctrl.update(current_kl=k, n_steps=8 + i)
        got.append(ctrl.kl_coef)
    assert got == Z["klc_traj"].tolist()
    fixed = core_algos.get_kl_controller(SimpleNamespace(kl_type="fixed", kl_coef=0.3, kl_horizon=0, kl_target=0))
    fixed.update(current_kl=5.0, n_steps=10)
    assert fixed.kl_coef == Z["klc_fixed"][0]
    assert isinstance(core_algos.get_kl_controller(SimpleNamespace(kl_type="adaptive", kl_coef=0.3, kl_horizon=10, kl_target=0.1)),
                      core_algos.AdaptiveKLController)


def test_other_estimators_and_value_loss():
    mask, uid = T("est_mask"), Z["est_uid"].astype(object)
    adv, ret = core_algos.compute_gae_advantage_return(T("est_dense_rew"), T("est_values"), mask, 0.99, 0.95)
    np.testing.assert_allclose(adv.numpy(), Z["gae_adv"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ret.numpy(), Z["gae_ret"], rtol=1e-6, atol=1e-6)
    adv, ret = core_algos.compute_rloo_outcome_advantage(T("est_rew"), mask, uid)
    np.testing.assert_allclose(adv.numpy(), Z["rloo_adv"], rtol=1e-6, atol=1e-7)
    assert adv is ret
    adv, ret = core_algos.compute_reinforce_plus_plus_outcome_advantage(T("est_dense_rew"), mask, 0.97)
    np.testing.assert_allclose(ret.numpy(), Z["rpp_ret"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(adv.numpy(), Z["rpp_adv"], rtol=2e-6, atol=2e-6)
    adv, _ = core_algos.compute_remax_outcome_advantage(T("est_rew"), T("est_base"), mask)
    np.testing.assert_allclose(adv.numpy(), Z["remax_adv"], rtol=0, atol=0)
    vl, vc = core_algos.compute_value_loss(T("vl_vpred"), T("rpp_ret"), T("est_values"), mask, 0.3)
    np.testing.assert_allclose([vl.item(), vc.item()], Z["vl_out"], rtol=1e-6)
    np.testing.assert_allclose(core_algos.masked_whiten(T("est_dense_rew"), mask).numpy(), Z["whiten"], rtol=2e-6, atol=2e-6)


def test_masked_whiten_uses_global_statistics_when_reducing_across_ranks():
    """Two half-batches whitened with an all_reduce that adds the OTHER half's partial sums = whitening the whole batch."""
    v, m = T("est_dense_rew"), T("est_mask")
    want = core_algos.masked_whiten(v, m)
    halves = [(v[:6], m[:6]), (v[6:], m[6:])]
    for me, other in ((0, 1), (1, 0)):
        calls = {"i": 0}

        def all_reduce(t, me=me, other=other, calls=calls):
            vo, mo = halves[other]
            i = calls["i"]; calls["i"] += 1
            if i == 0:
                return t + mo.sum().float()
            if i == 1:
                return t + (vo * mo).sum()
            n = m.sum().float()
            mean = (v * m).sum() / (n + 1e-8)
            return t + (((vo - mean) ** 2) * mo).sum()
        got = core_algos.masked_whiten(halves[me][0], halves[me][1], all_reduce=all_reduce)
        np.testing.assert_allclose(got.numpy(), want[me * 6:(me + 1) * 6].numpy(), rtol=1e-5, atol=1e-6)


def test_flops_counter_matches_reference_formula():
    from spatialthinker_amd.model import VLConfig
    seqlens, dt = Z["flops_seqlens"].tolist(), float(Z["flops_dt"][0])
    for cfg, want in zip((VLConfig.qwen2_5_vl_7b(), VLConfig.qwen2_5_vl_3b()), Z["flops_achieved"]):
        est, promised = FlopsCounter(cfg).estimate_flops(seqlens, dt)
        np.testing.assert_allclose(est, want, rtol=1e-12)
        assert promised == 2500.0


def test_small_helpers_equal_the_reference_functions():
    """small_helpers.npz = the reference's VF.masked_var / masked_whiten / pad_sequence_to_length and core_algos.compute_rewards on seeded
    inputs (tests/golden/make_golden.py small): bit for bit (same torch ops in the same order)."""
    import os
    import numpy as np
    import torch
    from verl.trainer import core_algos as C
    from verl.utils import torch_functional as VF
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "small_helpers.npz"))
    t = {k: torch.from_numpy(z[k]) for k in z.files}
    eq = lambda got, key: torch.equal(got, t[key])
    assert eq(VF.masked_var(t["v"], t["m"]), "var_unbiased") and eq(VF.masked_var(t["v"], t["m"], unbiased=False), "var_biased")
    assert eq(VF.masked_var(t["v"], t["one"]), "var_one")                                   # one selected entry: no Bessel correction (and a warning)
    assert eq(VF.masked_whiten(t["v"], t["m"]), "whiten") and eq(VF.masked_whiten(t["v"], t["m"], eps=1e-3), "whiten_eps")
    assert eq(VF.pad_sequence_to_length(t["ids"], 9, 77), "pad_right") and eq(VF.pad_sequence_to_length(t["ids"], 9, 77, left_pad=True), "pad_left")
    assert eq(VF.pad_sequence_to_length(t["ids"], 6, 77), "pad_noop") and eq(VF.pad_sequence_to_length(t["ids"], 4, 77), "pad_shorter")
    assert eq(C.compute_rewards(t["sc"], t["lp"], t["rp"], 0.037), "rewards")
    assert issubclass(C.FixedKLController, C.KLController) and issubclass(C.AdaptiveKLController, C.KLController)


def test_micro_batch_rearrangement_and_greedy_partition_equal_the_reference():
    """balance_micro.json = the reference's rearrange_micro_batches / greedy_partition / get_reverse_idx / ceildiv (make_golden.py micro)."""
    import json
    import os
    import torch
    from verl.protocol import TensorBatch
    from verl.utils.seqlen_balancing import ceildiv, get_reverse_idx, greedy_partition, rearrange_micro_batches
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "balance_micro.json")))
    for c in g["micro"]:
        lens = torch.tensor(c["lens"])
        mask = (torch.arange(c["width"])[None, :] < lens[:, None]).long()
        batch = TensorBatch({"attention_mask": mask, "row": torch.arange(len(lens))})
        micro, idx = rearrange_micro_batches(batch, c["max_token_len"])
        assert [list(p) for p in idx] == c["idx"]
        for m, p in zip(micro, idx):
            assert isinstance(m, TensorBatch) and m["row"].tolist() == list(p)      # (the balanced split does not promise tokens <= max_token_len per micro-batch)
        as_dict, _ = rearrange_micro_batches({"attention_mask": mask, "row": torch.arange(len(lens))}, c["max_token_len"])
        assert [m["row"].tolist() for m in as_dict] == c["idx"]
    for c in g["greedy"]:
        assert greedy_partition(c["lens"], c["k"], c["equal_size"]) == c["parts"]
    assert get_reverse_idx(g["perm"]) == g["reverse"] and all(ceildiv(a, b) == r for a, b, r in g["ceildiv"])
