"""End-to-end parity of the HIP engine (bf16, spatialthinker_amd.model) on the tiny Qwen2.5-VL config:
  * log-probs vs the fp32 oracle AND vs the HF-derived golden fixture (tests/golden/model_tiny.npz);
    criterion: the engine's error against fp32 is NO LARGER (factor 1.0) than a bf16 evaluation's
    own error — measured here by running HF transformers itself in bf16 on the GPU next to it.  Round 6: the yardstick is POOLED over
    16 random batches (tools/parity_probe.py: rms, mean of the per-batch maxima, overall maximum of |dlogp| over ~290 response tokens).
    The earlier form compared the maximum over the 18 response tokens of ONE batch with a 1.5x allowance; that statistic is one draw
    of a heavy-tailed quantity — over 16 batches the engine / HF-bf16 ratio of it ranges 0.43 .. 1.61 (median 0.91) while every pooled
    statistic has the engine BELOW HF-bf16 (rms 0.0080 vs 0.0091, per-tap hidden-state errors 0.88-0.99x; profiles/r06_parity_probe.txt);
  * gradients of the GRPO micro-batch loss vs autograd through the oracle, next to HF-bf16's own autograd gradients of the same scalar."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tiny  # noqa: E402
from oracle import qwen25vl as Q  # noqa: E402
from oracle import rl_math as M  # noqa: E402


@pytest.fixture(scope="module")
def env(golden_dir):
    from spatialthinker_amd import model as mdl
    z = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    cfg = mdl.VLConfig(**tiny.TINY)
    params = tiny.make_params()
    store = mdl.ParamStore(cfg, trainable=True)
    store.load_hf_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    eng = mdl.Qwen25VL(cfg, store)
    batch = tiny.make_batch()
    return z, cfg, params, store, eng, batch


def _probe():
    spec = importlib.util.spec_from_file_location("parity_probe", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "parity_probe.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def yardstick():
    """pooled |dlogp| statistics of the engine and of HF-bf16 against HF-fp32 over 16 random tiny batches, measured live on this GPU"""
    res = _probe().run(16, verbose=False)
    e, h = res["logp"]["engine"], res["logp"]["hf_bf16"]
    print(f"pooled over {e['tokens']} response tokens: engine rms {e['rms']:.5f} / mean per-batch max {e['mean_of_per_batch_max']:.4f} / max {e['max']:.4f}; "
          f"HF-bf16 rms {h['rms']:.5f} / {h['mean_of_per_batch_max']:.4f} / {h['max']:.4f}")
    return res


def _stage(eng, z, batch):
    return eng.stage(batch["input_ids"], batch["attention_mask"], z["position_ids"], batch["R"], batch["pixel_values"], batch["image_grid_thw"])


def test_hf_roundtrip_of_param_layout(env):
    z, cfg, params, store, eng, batch = env
    back = store.export_hf()
    for k, v in params.items():
        assert torch.equal(back[k].float().cpu(), torch.from_numpy(v)), k


def test_engine_error_is_no_larger_than_hf_bf16s_own_pooled_over_16_batches(yardstick):
    """SURVEY 8c (iii) at factor 1.0: every pooled statistic of the engine's |dlogp| against fp32 is at most HF-bf16's own, and so is the
    relative L2 error of every hidden-state tap (image features merged in, after each LM layer, final norm) — the per-block view that
    would name the op if one of them added error."""
    e, h = yardstick["logp"]["engine"], yardstick["logp"]["hf_bf16"]
    assert e["rms"] <= h["rms"] and e["mean_abs"] <= h["mean_abs"], (e, h)                        # measured 0.0080 vs 0.0091, 0.0066 vs 0.0072
    assert e["mean_of_per_batch_max"] <= h["mean_of_per_batch_max"] and e["max"] <= h["max"], (e, h)   # 0.0171 vs 0.0194, 0.0214 vs 0.0296
    for t in yardstick["taps"]:
        assert t["engine_rel_l2"] <= 1.0 * t["hf_bf16_rel_l2"], t                                 # 0.99 / 0.98 / 0.94 / 0.88 x


def test_log_probs_vs_fp32_oracle_and_hf_golden(env, yardstick):
    z, cfg, params, store, eng, batch = env
    b = _stage(eng, z, batch)
    lp = eng.log_probs(b, temperature=1.0).cpu().numpy()
    mask = batch["attention_mask"][:, -batch["R"]:].astype(bool)
    err_golden = np.abs(lp[mask] - z["logp"][mask]).max()
    # HF itself in bf16 on this GPU: the size of a bf16 evaluation's own error on THIS batch (one draw) and pooled (the bound)
    err_hf = _hf_bf16_error(z, cfg, params, batch)
    yard_max = yardstick["logp"]["hf_bf16"]["max"]
    print(f"engine max|dlogp| vs HF-fp32 golden: {err_golden:.4f}; HF-bf16 vs HF-fp32 on this batch: {err_hf:.4f}, pooled maximum: {yard_max:.4f}")
    assert err_golden <= yard_max                  # the committed HF-fp32 fixture: within what HF-bf16 itself shows over 16 batches (0.0158 vs 0.0296)
    assert np.all(lp[~mask] == 0)
    # temperature is applied to the logits (dp_actor.py:126)
    lp2 = eng.log_probs(b, temperature=0.5).cpu().numpy()
    orc = Q.response_log_probs({k: torch.from_numpy(v) for k, v in params.items()}, Q.VLConfig(**tiny.TINY),
                               torch.from_numpy(batch["input_ids"]), torch.from_numpy(batch["attention_mask"]),
                               torch.from_numpy(z["position_ids"]), batch["R"], 0.5, torch.from_numpy(batch["pixel_values"]),
                               batch["image_grid_thw"]).numpy()
    err_t = np.abs(lp2[mask] - orc[mask]).max()
    print(f"temperature 0.5: max|dlogp| vs fp32 oracle {err_t:.4f}")
    assert err_t <= 2 * yard_max                    # logits are doubled before the softmax: 2x the temperature-1 yardstick


def _hf_bf16_error(z, cfg, params, batch):
    try:
        from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    except Exception:                                                    # pragma: no cover
        return 2e-2
    c = tiny.TINY
    hc = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"], num_hidden_layers=c["num_layers"],
                         num_attention_heads=c["num_heads"], num_key_value_heads=c["num_kv_heads"], vocab_size=c["vocab_size"],
                         rms_norm_eps=c["rms_eps"], rope_parameters=dict(rope_type="default", rope_theta=c["rope_theta"], mrope_section=c["mrope_section"]),
                         tie_word_embeddings=False, max_position_embeddings=4096, bos_token_id=None, eos_token_id=tiny.EOS_ID, pad_token_id=tiny.PAD_ID),
        vision_config=dict(depth=c["v_depth"], hidden_size=c["v_hidden"], num_heads=c["v_heads"], intermediate_size=c["v_intermediate"],
                           out_hidden_size=c["hidden_size"], patch_size=c["v_patch"], spatial_merge_size=c["v_merge"],
                           temporal_patch_size=c["v_temporal_patch"], window_size=c["v_window"], fullatt_block_indexes=c["v_fullatt"],
                           in_channels=c["v_in_channels"]),
        image_token_id=c["image_token_id"], video_token_id=1009, vision_start_token_id=c["vision_start_token_id"],
        vision_end_token_id=tiny.VISION_END, tie_word_embeddings=False, bos_token_id=None, eos_token_id=tiny.EOS_ID, pad_token_id=tiny.PAD_ID)
    hc._attn_implementation = "sdpa"
    m = Qwen2_5_VLForConditionalGeneration(hc)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    m = m.to(torch.bfloat16).cuda().eval()
    ids, mask, P, R = batch["input_ids"], batch["attention_mask"], batch["P"], batch["R"]
    errs, off = [], 0
    for bidx in range(ids.shape[0]):
        sel = mask[bidx] == 1
        n = int(batch["patch_counts"][bidx])
        with torch.no_grad():
            o = m(input_ids=torch.from_numpy(ids[bidx][sel])[None].cuda(), attention_mask=None,
                  position_ids=torch.from_numpy(z["position_ids"][bidx][:, sel])[:, None, :].cuda(),
                  pixel_values=torch.from_numpy(batch["pixel_values"][off:off + n]).cuda(),
                  image_grid_thw=torch.from_numpy(batch["image_grid_thw"][bidx:bidx + 1]).cuda(), use_cache=False)
        off += n
        lg = o.logits[0].float().cpu()
        labels = torch.roll(torch.from_numpy(ids[bidx][sel]), -1)
        lp = torch.log_softmax(lg, -1).gather(-1, labels[:, None])[:, 0].numpy()
        full = np.zeros(P + R, dtype=np.float32); full[sel] = lp
        rm = mask[bidx, -R:].astype(bool)
        errs.append(np.abs(full[-R - 1:-1][rm] - z["logp"][bidx][rm]).max())
    return float(max(errs))


def test_grpo_micro_batch_gradients_vs_oracle_autograd(env, yardstick):
    z, cfg, params, store, eng, batch = env
    rs = np.random.RandomState(21)
    B, R = batch["input_ids"].shape[0], batch["R"]
    rmask = batch["attention_mask"][:, -R:]
    p32 = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params.items()}
    ocfg = Q.VLConfig(**tiny.TINY)
    lp = Q.response_log_probs(p32, ocfg, torch.from_numpy(batch["input_ids"]), torch.from_numpy(batch["attention_mask"]),
                              torch.from_numpy(z["position_ids"]), R, 1.0, torch.from_numpy(batch["pixel_values"]), batch["image_grid_thw"])
    old = (lp.detach().numpy() + 0.3 * rs.standard_normal((B, R))).astype(np.float32)
    ref = (lp.detach().numpy() + 0.2 * rs.standard_normal((B, R))).astype(np.float32)
    adv = rs.standard_normal((B, 1)).astype(np.float32).repeat(R, 1) * rmask
    met, g = M.actor_micro_batch_loss(lp.detach().numpy(), old, ref, adv, rmask, kl_kind="low_var_kl", kl_coef=1e-2, grad_accum=2)
    lp.backward(torch.from_numpy(g))
    store.grad.zero_()
    b = _stage(eng, z, batch)
    dv = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dt)
    lp_eng, metrics = eng.forward_backward(b, dict(old_log_probs=dv(old), ref_log_probs=dv(ref), advantages=dv(adv),
                                                   response_mask=dv(rmask, torch.int64)), 1.0,
                                           clip_low=0.2, clip_high=0.3, clip_dual=3.0, kl_kind="low_var_kl", kl_coef=1e-2, grad_accum=2.0)
    mask = rmask.astype(bool)
    assert np.abs(lp_eng.cpu().numpy()[mask] - lp.detach().numpy()[mask]).max() <= yardstick["logp"]["hf_bf16"]["max"]      # measured 0.0158 vs 0.0296
    grads = store.export_hf(store.g)
    # the yardstick for gradients (round 6): HF transformers in bf16 differentiating the SAME scalar sum(logp * g) with its own autograd on
    # this GPU, against the same fp32 truth — per tensor, side by side.  (HF keeps bf16 gradients; the engine accumulates fp32 and rounds
    # the operands of each product once.)
    hf_rel = _probe().grad_yardstick(params, batch, z["position_ids"], g, {k: t.grad.numpy() for k, t in p32.items()})
    worst, table = [], []
    for k, t in p32.items():
        want = t.grad.numpy()
        got = grads[k].float().cpu().numpy()
        rel = np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-12)
        worst.append((rel, k))
        if k in hf_rel and not k.endswith("k_proj.bias"):     # a key bias shifts every score of a query alike: its true gradient is ZERO up to
            table.append((rel, hf_rel[k], k))                 # rounding, so a relative error measures nothing (engine 0.030, HF-bf16 0.022: both noise)
    worst.sort(reverse=True)
    print("worst gradient relative errors:", worst[:6])
    ratios = np.array([e / max(h, 1e-12) for e, h, _ in table])
    e_all, h_all = np.array([e for e, _, _ in table]), np.array([h for _, h, _ in table])
    print(f"engine vs HF-bf16 gradient error over {len(table)} tensors: worst {e_all.max():.4f} vs {h_all.max():.4f}, median {np.median(e_all):.4f} vs "
          f"{np.median(h_all):.4f}; engine / HF-bf16 ratio: median {np.median(ratios):.2f}, max {ratios.max():.2f} "
          f"({table[int(ratios.argmax())][2]}), tensors where the engine is worse: {int((ratios > 1).sum())}")
    assert len(table) == len(p32) - cfg.num_layers
    assert e_all.max() <= h_all.max() and np.median(e_all) <= np.median(h_all), (e_all.max(), h_all.max())
    assert (ratios > 1).mean() <= GRAD_WORSE_FRACTION, sorted(((e / h, k) for e, h, k in table), reverse=True)[:6]
    # padded parameter regions must receive exactly zero gradient
    vi, vip = cfg.v_intermediate, cfg.v_inter_pad
    assert float(store.g["v.0.gu_w"][vi:vip].abs().max()) == 0 and float(store.g["v.0.down_w"][:, vi:].abs().max()) == 0
    assert float(store.g["v.patch_embed"][:, cfg.patch_k:].abs().max()) == 0


def test_shared_image_runs_the_vision_tower_once(env):
    """Rollouts of one prompt share their image: staging with `image_map` runs the ViT once and sums the feature gradients;
    log-probs are bit-identical to the per-sample path and the gradients agree up to bf16 summation order."""
    z, cfg, params, store, eng, batch = env
    R = batch["R"]
    n0 = int(batch["patch_counts"][0])
    ids = np.repeat(batch["input_ids"][:1], 3, 0); mask = np.repeat(batch["attention_mask"][:1], 3, 0)
    pos = np.repeat(z["position_ids"][:1], 3, 0)
    px0, g0 = batch["pixel_values"][:n0], batch["image_grid_thw"][:1]
    rs = np.random.RandomState(3)
    rmask = mask[:, -R:]
    old = rs.standard_normal((3, R)).astype(np.float32) * 0.1 - 5.0
    adv = rs.standard_normal((3, 1)).astype(np.float32).repeat(R, 1) * rmask
    dv = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dt)
    loss_in = dict(old_log_probs=dv(old), ref_log_probs=dv(old), advantages=dv(adv), response_mask=dv(rmask, torch.int64))
    kw = dict(clip_low=0.2, clip_high=0.3, clip_dual=3.0, kl_kind="low_var_kl", kl_coef=1e-2, grad_accum=1.0)
    out = []
    for shared in (False, True):
        if shared:
            b = eng.stage(ids, mask, pos, R, px0, g0, image_map=[0, 0, 0])
        else:
            b = eng.stage(ids, mask, pos, R, np.concatenate([px0] * 3, 0), np.concatenate([g0] * 3, 0))
        store.grad.zero_()
        lp, _ = eng.forward_backward(b, loss_in, 1.0, **kw)
        out.append((lp.clone(), store.grad.clone()))
    assert torch.equal(out[0][0], out[1][0])
    ga, gb = out[0][1], out[1][1]
    rel = float((ga - gb).norm() / ga.norm())
    assert rel < 5e-3, rel
    store.grad.zero_()


def test_shared_prompt_packing_matches_per_sequence_packing(env):
    """Rollouts of one prompt packed as [prompt][resp_1]..[resp_k] (prompt + image computed once, shared-prefix attention)
    vs the reference's per-sequence packing: same log-probs and gradients up to bf16 reduction order."""
    z, cfg, params, store, eng, batch = env
    R, k = batch["R"], 3
    n0 = int(batch["patch_counts"][0])
    rs = np.random.RandomState(11)
    ids = np.repeat(batch["input_ids"][:1], k, 0).copy(); mask = np.repeat(batch["attention_mask"][:1], k, 0).copy()
    pos = np.repeat(z["position_ids"][:1], k, 0)
    S = ids.shape[1]
    for r in range(1, k):                                    # different responses (and lengths) behind the same prompt
        ids[r, S - R:] = rs.randint(3, 900, R)
        mask[r, S - R + rs.randint(2, R):] = 0
    px0, g0 = batch["pixel_values"][:n0], batch["image_grid_thw"][:1]
    rmask = mask[:, -R:]
    old = rs.standard_normal((k, R)).astype(np.float32) * 0.1 - 6.0
    adv = rs.standard_normal((k, 1)).astype(np.float32).repeat(R, 1) * rmask
    dv = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dt)
    loss_in = dict(old_log_probs=dv(old), ref_log_probs=dv(old), advantages=dv(adv), response_mask=dv(rmask, torch.int64))
    kw = dict(clip_low=0.2, clip_high=0.3, clip_dual=3.0, kl_kind="low_var_kl", kl_coef=1e-2, grad_accum=1.0)
    res = []
    for shared in (False, True):
        if shared:
            b = eng.stage(ids, mask, pos, R, px0, g0, groups=[0] * k)
            assert b.pk.T < 0.6 * int(mask.sum())            # the prompt is stored once
        else:
            b = eng.stage(ids, mask, pos, R, np.concatenate([px0] * k, 0), np.concatenate([g0] * k, 0))
        store.grad.zero_()
        lp, _ = eng.forward_backward(b, loss_in, 1.0, **kw)
        lp_ng = eng.log_probs(b, 1.0)
        assert torch.equal(lp, lp_ng)
        res.append((lp.clone(), store.grad.clone()))
    m = torch.from_numpy(rmask.astype(bool)).cuda()
    assert float((res[0][0] - res[1][0])[m].abs().max()) < 3e-2
    ga, gb = res[0][1], res[1][1]
    rel = float((ga - gb).norm() / ga.norm())
    assert rel < 2e-2, rel
    store.grad.zero_()


def test_fused_micro_batches_keep_per_micro_batch_loss_normalisation(env):
    """Two reference micro-batches in ONE pass (loss_rows = micro-batch size) == two separate passes: same metrics per
    micro-batch, same accumulated gradient up to bf16 reduction order."""
    z, cfg, params, store, eng, batch = env
    R = batch["R"]
    rs = np.random.RandomState(5)
    ids = np.concatenate([batch["input_ids"], batch["input_ids"]], 0).copy()
    mask = np.concatenate([batch["attention_mask"], batch["attention_mask"]], 0).copy()
    pos = np.concatenate([z["position_ids"], z["position_ids"]], 0)
    S = ids.shape[1]
    ids[2:, S - R:] = rs.randint(3, 900, (2, R))
    mask[2, S - 3:] = 0
    px = np.concatenate([batch["pixel_values"], batch["pixel_values"]], 0)
    gr = np.concatenate([batch["image_grid_thw"], batch["image_grid_thw"]], 0)
    n_patch = batch["patch_counts"]
    rmask = mask[:, -R:]
    old = rs.standard_normal((4, R)).astype(np.float32) * 0.1 - 6.0
    adv = rs.standard_normal((4, 1)).astype(np.float32).repeat(R, 1) * rmask
    dv = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dt)
    kw = dict(clip_low=0.2, clip_high=0.3, clip_dual=3.0, kl_kind="low_var_kl", kl_coef=1e-2, grad_accum=2.0)
    li = lambda sl: dict(old_log_probs=dv(old[sl]), ref_log_probs=dv(old[sl]), advantages=dv(adv[sl]), response_mask=dv(rmask[sl], torch.int64))
    store.grad.zero_()
    mets = []
    off = int(n_patch[0] + n_patch[1])
    for sl, ps in ((slice(0, 2), slice(0, off)), (slice(2, 4), slice(off, 2 * off))):
        b = eng.stage(ids[sl], mask[sl], pos[sl], R, px[ps], gr[sl])
        _, m = eng.forward_backward(b, li(sl), 1.0, **kw)
        mets.append(m)
    g_sep = store.grad.clone()
    store.grad.zero_()
    b = eng.stage(ids, mask, pos, R, px, gr)
    _, m2 = eng.forward_backward(b, li(slice(0, 4)), 1.0, loss_rows=2, **kw)
    g_fused = store.grad.clone()
    assert m2.shape == (2, 8)
    assert torch.allclose(torch.stack(mets), m2, rtol=2e-2, atol=2e-3)
    rel = float((g_sep - g_fused).norm() / g_sep.norm())
    assert rel < 2e-2, rel
    store.grad.zero_()


def test_recompute_light_activations_is_bit_identical(env):
    z, cfg, params, store, eng, batch = env
    R = batch["R"]
    rs = np.random.RandomState(2)
    rmask = batch["attention_mask"][:, -R:]
    B = rmask.shape[0]
    old = rs.standard_normal((B, R)).astype(np.float32) * 0.1 - 6.0
    adv = rs.standard_normal((B, 1)).astype(np.float32).repeat(R, 1) * rmask
    dv = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dt)
    li = dict(old_log_probs=dv(old), ref_log_probs=dv(old), advantages=dv(adv), response_mask=dv(rmask, torch.int64))
    kw = dict(clip_low=0.2, clip_high=0.3, clip_dual=3.0, kl_kind="low_var_kl", kl_coef=1e-2, grad_accum=1.0)
    grads = []
    for flag in (False, True):
        eng.recompute_light = flag
        store.grad.zero_()
        eng.forward_backward(_stage(eng, z, batch), li, 1.0, **kw)
        grads.append(store.grad.clone())
    eng.recompute_light = False
    store.grad.zero_()
    assert torch.equal(grads[0], grads[1])


def test_vision_tower_taps_vs_hf_golden(env):
    """Direct check of the ViT path (patch embed, window reorder, 2-D RoPE, windowed + full attention, SwiGLU MLP, merger, inverse
    reorder) against the HF-fp32 taps of model_tiny.npz: `vit_last*` = output of the last block (window order), `image_embeds*` =
    merged features in image order.  The ViT only contributes a fifth of the LM tokens, so log-prob tolerances alone would dilute
    an error here.  Bound: bf16 evaluation error, normalised by the tap's RMS."""
    z, cfg, params, store, eng, batch = env
    off = 0
    for bi, n in enumerate(batch["patch_counts"]):
        n = int(n)
        sel = batch["attention_mask"][bi] == 1
        b = eng.stage(batch["input_ids"][bi:bi + 1], batch["attention_mask"][bi:bi + 1], z["position_ids"][bi:bi + 1], batch["R"],
                      batch["pixel_values"][off:off + n], batch["image_grid_thw"][bi:bi + 1])
        off += n
        saved = []
        img = eng._vit_forward(b, saved).float().cpu().numpy()
        last = saved[-1][0][:n].float().cpu().numpy()
        for name, got, want in (("image_embeds", img[:n // 4], z[f"image_embeds{bi}"]), ("vit_last", last, z[f"vit_last{bi}"])):
            rms = float(np.sqrt((want ** 2).mean()))
            err = float(np.abs(got - want).max()) / rms
            rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
            print(f"ViT tap {name}{bi}: max|err|/rms = {err:.4f}, relative L2 = {rel:.5f}")
            assert got.shape == want.shape
            assert rel < VIT_TAP_REL_L2 and err < VIT_TAP_MAX_OVER_RMS, (name, bi, rel, err)


GRAD_WORSE_FRACTION = 0.35        # share of tensors whose engine gradient error may exceed HF-bf16's own on this ONE batch: measured 16 of 67 (ratio median 0.89, max 1.11) — a tensor-by-tensor comparison of two noise realisations; the worst and the median decide
VIT_TAP_REL_L2, VIT_TAP_MAX_OVER_RMS = 9.1e-3, 5.0e-2        # 1.3x the measured 0.0070 / 0.0386
