"""Pins oracle/qwen25vl.py (fp32 restatement of the Qwen2.5-VL forward on PACKED input) against
HF transformers outputs on the tiny config (fixtures: tests/golden/model_tiny.npz, produced
per-sample and un-padded by HF itself).  Tolerance 1e-5 abs on fp32 logits / log-probs
(SURVEY.md §8c parity metric (i))."""
import os

import numpy as np
import pytest
import torch

import tiny
from oracle import positions as P
from oracle import qwen25vl as Q


@pytest.fixture(scope="module")
def setup(golden_dir):
    z = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    cfg = Q.VLConfig(**tiny.TINY)
    params = {k: torch.from_numpy(v) for k, v in tiny.make_params().items()}
    return z, cfg, params, tiny.make_batch()


def test_position_ids_match_reference_pipeline(setup):
    z, cfg, params, batch = setup
    Pn, R = batch["P"], batch["R"]
    for b in range(batch["input_ids"].shape[0]):
        pp = P.mrope_position_ids(batch["input_ids"][b, :Pn], batch["image_grid_thw"][b:b + 1], batch["attention_mask"][b, :Pn],
                                  image_token_id=cfg.image_token_id, vision_start_token_id=cfg.vision_start_token_id)
        pp[:, batch["attention_mask"][b, :Pn] == 0] = 0
        np.testing.assert_array_equal(P.continue_position_ids(pp, R), z["position_ids"][b])


def test_vision_tower(setup):
    z, cfg, params, batch = setup
    off = 0
    for b, n in enumerate(batch["patch_counts"]):
        taps = {}
        out = Q.vision_tower(params, cfg, torch.from_numpy(batch["pixel_values"][off:off + n]), batch["image_grid_thw"][b:b + 1], taps)
        off += n
        np.testing.assert_allclose(out.numpy(), z[f"image_embeds{b}"], atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(taps[f"vit_block{cfg.v_depth - 1}"].numpy(), z[f"vit_last{b}"], atol=2e-5, rtol=1e-5)


def test_packed_logprobs_and_logits(setup):
    z, cfg, params, batch = setup
    lp = Q.response_log_probs(params, cfg, torch.from_numpy(batch["input_ids"]), torch.from_numpy(batch["attention_mask"]),
                              torch.from_numpy(z["position_ids"]), batch["R"], 1.0,
                              torch.from_numpy(batch["pixel_values"]), batch["image_grid_thw"])
    mask = batch["attention_mask"][:, -batch["R"]:].astype(bool)
    np.testing.assert_allclose(lp.numpy()[mask], z["logp"][mask], atol=1e-5, rtol=0)
    # logits of the last 3 valid tokens of each sequence + per-layer hidden of the last token
    valid = torch.from_numpy(batch["attention_mask"]).bool()
    idx = valid.reshape(-1).nonzero()[:, 0]
    ids = torch.from_numpy(batch["input_ids"]).reshape(-1)[idx]
    pos = torch.from_numpy(z["position_ids"]).permute(1, 0, 2).reshape(3, -1)[:, idx]
    lens = valid.sum(-1)
    cu = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)])
    taps = {}
    logits = Q.forward_logits(params, cfg, ids, pos, cu, torch.from_numpy(batch["pixel_values"]), batch["image_grid_thw"], taps)
    for b in range(2):
        e = int(cu[b + 1])
        np.testing.assert_allclose(logits[e - 3:e].numpy(), z["logits_last3"][b], atol=1e-5, rtol=1e-5)
        for layer in range(cfg.num_layers - 1):      # HF's last tuple entry is post-final-norm
            np.testing.assert_allclose(taps[f"lm_layer{layer}"][e - 1].numpy(), z["hidden_last_token"][b][layer + 1], atol=1e-5, rtol=1e-5)
    # temperature and row-selection are row-wise identities
    rows = torch.tensor([3, 10, int(cu[1]) + 2])
    sub = Q.forward_logits(params, cfg, ids, pos, cu, torch.from_numpy(batch["pixel_values"]), batch["image_grid_thw"], rows=rows)
    np.testing.assert_allclose(sub.numpy(), logits[rows].numpy(), atol=5e-6)


def test_kv_cache_greedy_decode_matches_hf_generate(setup, golden_dir):
    """oracle.qwen25vl.generate_greedy (prefill + lm_layer_decode loop over a KV cache: the decode loop the CPU baseline times) vs
    transformers' generate(do_sample=False) on the tiny model — token for token (tests/golden/generate_tiny.npz)."""
    import os
    z, cfg, params, batch = setup
    gold = np.load(os.path.join(golden_dir, "generate_tiny.npz"))
    Pn, off = batch["P"], 0
    for b in range(2):
        sel = batch["attention_mask"][b, :Pn] == 1
        n = int(batch["patch_counts"][b])
        out = Q.generate_greedy(params, cfg, torch.from_numpy(batch["input_ids"][b, :Pn][sel]), torch.from_numpy(z["position_ids"][b][:, :Pn][:, sel]),
                                10, torch.from_numpy(batch["pixel_values"][off:off + n]), batch["image_grid_thw"][b:b + 1])
        off += n
        assert out.tolist() == gold[f"greedy{b}"].tolist()
