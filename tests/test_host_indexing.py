"""CPU tests of the product's host-side index logic against the reference-derived golden vectors."""
import os

import numpy as np

import tiny
from spatialthinker_amd import indexing as I


def test_rope_index_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "positions.npz"))
    for i in range(4):
        thw = z[f"rope{i}_thw"]
        pos = I.get_rope_index(z[f"rope{i}_ids"], thw if len(thw) else None, z[f"rope{i}_mask"], image_token_id=990,
                               vision_start_token_id=991)
        np.testing.assert_array_equal(pos, z[f"rope{i}_pos"])


def test_rope_index_with_video_blocks_golden(golden_dir):
    """Video grids (reference :88-101, :117-118; outside the SpatialThinker data, closed in round 4 for completeness): temporal index
    = trunc(frame * second_per_grid_t * 2), mixed image + video blocks in sequence order, with and without second_per_grid_ts — the
    product function, its by-name wrapper and the oracle restatement against the reference function's own output."""
    import types
    import torch
    from oracle import positions as OP
    from verl.models.transformers.qwen2_vl import get_rope_index as by_name
    z = np.load(os.path.join(golden_dir, "positions.npz"))
    proc = types.SimpleNamespace(tokenizer=types.SimpleNamespace(convert_tokens_to_ids={"<|image_pad|>": 990, "<|video_pad|>": 989, "<|vision_start|>": 991}.get),
                                 image_processor=types.SimpleNamespace(merge_size=2))
    for i in range(3):
        img, vid, secs = z[f"ropev{i}_img"], z[f"ropev{i}_vid"], z[f"ropev{i}_secs"]
        kw = dict(image_token_id=990, vision_start_token_id=991, video_grid_thw=vid, second_per_grid_ts=secs if len(secs) else None, video_token_id=989)
        pos = I.get_rope_index(z[f"ropev{i}_ids"], img if len(img) else None, z[f"ropev{i}_mask"], **kw)
        np.testing.assert_array_equal(pos, z[f"ropev{i}_pos"])
        orc = OP.mrope_position_ids_with_video(z[f"ropev{i}_ids"], img, vid, secs if len(secs) else None, z[f"ropev{i}_mask"], image_token_id=990,
                                               video_token_id=989, vision_start_token_id=991)
        np.testing.assert_array_equal(orc, z[f"ropev{i}_pos"])
        got = by_name(proc, torch.from_numpy(z[f"ropev{i}_ids"]), image_grid_thw=torch.from_numpy(img) if len(img) else None,
                      video_grid_thw=torch.from_numpy(vid), second_per_grid_ts=torch.from_numpy(secs) if len(secs) else None,
                      attention_mask=torch.from_numpy(z[f"ropev{i}_mask"]))
        np.testing.assert_array_equal(got.numpy(), z[f"ropev{i}_pos"])


def test_vision_indices_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "positions.npz"))
    for i in range(4):
        thw = z[f"vwin{i}_thw"]
        for win in (56, 112):
            idx, cu = I.vision_window_index(thw, 2, win, 14)
            np.testing.assert_array_equal(idx, z[f"vwin{i}_{win}_idx"])
            np.testing.assert_array_equal(cu, z[f"vwin{i}_{win}_cu"])
        np.testing.assert_array_equal(I.vision_position_ids(thw, 2), z[f"vwin{i}_pos"])
        plan = I.plan_vision(thw, merge=2, window=112, patch=14, head_dim=80)
        assert plan.cos.shape == (plan.n_patches, 40)
        np.testing.assert_array_equal(np.sort(plan.patch_gather), np.arange(plan.n_patches))
        np.testing.assert_array_equal(z[f"vwin{i}_112_idx"][plan.merged_inverse], np.arange(len(plan.merged_inverse)))


def test_pack_batch_layout(golden_dir):
    z = np.load(os.path.join(golden_dir, "model_tiny.npz"))
    b = tiny.make_batch()
    pk = I.pack_batch(b["input_ids"], b["attention_mask"], z["position_ids"], b["R"], image_token_id=tiny.TINY["image_token_id"])
    lens = b["attention_mask"].sum(1)
    assert pk.T == lens.sum() and pk.T_pad % 128 == 0 and pk.max_seqlen == lens.max()
    np.testing.assert_array_equal(pk.cu_seqlens, np.concatenate([[0], np.cumsum(lens)]))
    # every valid response token gets exactly one logit row, the row just before it in the packed stream
    assert len(pk.logit_rows) == b["attention_mask"][:, -b["R"]:].sum()
    np.testing.assert_array_equal(pk.ids[pk.logit_rows + 1], pk.labels)
    flat_resp = b["responses"].reshape(-1)
    np.testing.assert_array_equal(flat_resp[pk.out_index], pk.labels)
    assert (pk.ids[pk.image_rows] == tiny.TINY["image_token_id"]).all() and len(pk.image_rows) == 16 + 12
    assert (pk.embed_ids[pk.image_rows] == -1).all() and (pk.embed_ids[pk.T:] == -1).all()
    assert (pk.pos[:, pk.cu_seqlens[:-1]] == 0).all()       # every packed sequence starts at position 0


def _random_rollout_batch(rs, n_prompts=3, G=4, Pc=24, R=9):
    B, S = n_prompts * G, Pc + R
    ids = rs.randint(20, 500, (B, S)).astype(np.int64)
    mask = np.ones((B, S), dtype=np.int64)
    for p in range(n_prompts):
        lead = rs.randint(0, 6)                                  # left padding of the prompt
        rows = slice(p * G, (p + 1) * G)
        ids[rows, :Pc] = ids[p * G, :Pc]
        mask[rows, :lead] = 0
        ids[rows, Pc - 5:Pc - 2] = 7                              # three image placeholders inside the prompt
    for r in range(B):
        L = rs.randint(0, R + 1)                                  # response length, 0 allowed
        mask[r, Pc + L:] = 0
    pos = np.maximum(np.cumsum(mask, 1) - 1, 0)
    return ids, mask, pos, Pc, R, G


def test_grouped_packing_reconstructs_every_sequence_and_its_logit_rows():
    """pack_batch(groups=...) stores the prompt of a rollout group once; following the segment arrays back must give exactly the
    tokens, positions and (logit row -> label) pairs of the per-sequence packing."""
    from spatialthinker_amd import indexing as ix
    rs = np.random.RandomState(0)
    for trial in range(5):
        ids, mask, pos, Pc, R, G = _random_rollout_batch(rs)
        B = ids.shape[0]
        a = ix.pack_batch(ids, mask, pos, R, image_token_id=7)
        g = ix.pack_batch(ids, mask, pos, R, image_token_id=7, groups=[r // G for r in range(B)])
        assert g.T < a.T and np.array_equal(a.labels, g.labels) and np.array_equal(a.out_index, g.out_index)
        # the token predicted-from at every logit row is the same token in both layouts
        assert np.array_equal(a.ids[a.logit_rows], g.ids[g.logit_rows])
        assert np.array_equal(a.pos[:, a.logit_rows], g.pos[:, g.logit_rows])
        # segments tile the packed stream; prefixes are prompt segments; dependents end where the group ends
        cover = np.zeros(g.T, dtype=int)
        for b_, e_, pb_, pe_, de_ in zip(g.seg_b, g.seg_e, g.pre_b, g.pre_e, g.dep_e):
            cover[b_:e_] += 1
            assert de_ >= e_ and (pb_ == pe_ or (pe_ <= b_ and de_ == e_))
        assert np.all(cover == 1)
        # every row's full token sequence = its group's prompt rows + its own response rows
        resp_segs = [(b_, e_, pb_, pe_) for b_, e_, pb_, pe_ in zip(g.seg_b, g.seg_e, g.pre_b, g.pre_e) if pe_ > pb_]
        it = iter(resp_segs)
        for r in range(B):
            want = ids[r][mask[r] == 1]
            n_resp = int(mask[r, Pc:].sum())
            n_prompt = int(mask[r, :Pc].sum())
            if n_resp == 0:
                continue
            b_, e_, pb_, pe_ = next(it)
            got = np.concatenate([g.ids[pb_:pe_], g.ids[b_:e_]])
            assert pe_ - pb_ == n_prompt and np.array_equal(got, want), (trial, r)
        # image rows: once per group
        assert len(g.image_rows) == 3 * (B // G) and len(a.image_rows) == 3 * B
        # duplicates only at the last prompt row of a group
        if g.logit_dup is not None:
            assert (g.logit_dup >= 0).sum() == len(g.logit_rows)


def test_pack_responses_layout():
    from spatialthinker_amd import indexing as ix
    rs = np.random.RandomState(3)
    ids, mask, pos, Pc, R, G = _random_rollout_batch(rs)
    B = ids.shape[0]
    n_prompts = B // G
    plens = mask[::G, :Pc].sum(1)
    p_off = np.concatenate([[0], np.cumsum(plens)])
    full = ix.pack_batch(ids, mask, pos, R, image_token_id=7)
    pk = ix.pack_responses(ids, mask, pos, R, [r // G for r in range(B)], p_off)
    assert pk.T == int(mask[:, Pc:].sum())
    assert sorted(pk.out_index.tolist()) == sorted(full.out_index.tolist())
    # same (slot -> label) assignment as the full packing
    lab_full = dict(zip(full.out_index.tolist(), full.labels.tolist()))
    assert all(lab_full[i] == l for i, l in zip(pk.out_index.tolist(), pk.labels.tolist()))
    n_first = len(pk.first_prompt)
    assert np.all(pk.out_index[:n_first] % R == 0) and np.all(pk.out_index[n_first:] % R > 0)
    for (b_, e_, pb_, pe_) in zip(pk.seg_b, pk.seg_e, pk.pre_b, pk.pre_e):
        assert pe_ - pb_ in plens and 0 <= pb_ < pe_ <= p_off[-1]
