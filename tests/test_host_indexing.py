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
