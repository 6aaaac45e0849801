"""RLHFDataset / collate_fn against the reference class's own output rows (tests/golden/dataset.npz, produced by
make_golden.py gen_dataset from /root/reference/verl/utils/dataset.py with the stub tokenizer / processor of
tests/golden/stub_mm.py on the committed tiny parquet tests/golden/stvqa_tiny)."""
import os

import numpy as np
import pytest
import torch

from stub_mm import StubProcessor, StubTokenizer
from verl.utils.dataset import RLHFDataset, collate_fn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
Z = np.load(os.path.join(GOLD, "dataset.npz"))
ROOT = os.path.join(GOLD, "stvqa_tiny")

CASES = {
    "spatial": dict(split="train", prompt_key="problem", answer_key="answer_option_text", image_key="images", max_prompt_length=96,
                    truncation="right", format_prompt=None, shuffle=False, mixed_data=False, text_only=False),
    "vanilla": dict(split="val", prompt_key="question_with_options", answer_key="answer_option_text_only", image_key="images",
                    max_prompt_length=160, truncation="right", format_prompt="  You FIRST think. \n", shuffle=True, mixed_data=False, text_only=False),
    "mixed": dict(split="train", prompt_key="problem", answer_key="answer_option_text", image_key="images", max_prompt_length=128,
                  truncation="left", format_prompt=None, shuffle=True, mixed_data=True, text_only=False),
    "textonly": dict(split="train", prompt_key="problem", answer_key="answer_option_text", image_key="images", max_prompt_length=64,
                     truncation="right", format_prompt=None, shuffle=False, mixed_data=False, text_only=True),
}


def _ds(c):
    return RLHFDataset(f"{ROOT}@{c['split']}", StubTokenizer(), StubProcessor(), prompt_key=c["prompt_key"], answer_key=c["answer_key"],
                       image_key=c["image_key"], mixed_data=c["mixed_data"], text_only=c["text_only"], max_prompt_length=c["max_prompt_length"],
                       truncation=c["truncation"], format_prompt=c["format_prompt"], max_pixels=64 * 28 * 28 // 4, min_pixels=28 * 28 * 4,
                       shuffle=c["shuffle"], seed=5)


@pytest.mark.parametrize("name", list(CASES))
def test_rows_match_reference(name):
    ds = _ds(CASES[name])
    assert len(ds) == int(Z[f"{name}_len"][0])                       # @val reads the val shards, not train
    for i in range(len(ds)):
        r = ds[i]
        for k in ("input_ids", "attention_mask", "position_ids"):
            assert np.array_equal(r[k].numpy(), Z[f"{name}_{i}_{k}"]), (name, i, k)
        assert list(r["raw_prompt_ids"]) == Z[f"{name}_{i}_raw_prompt_ids"].tolist()
        assert r["ground_truth"] == str(Z[f"{name}_{i}_ground_truth"][0]) and r["extra"] == int(Z[f"{name}_{i}_extra"][0])
        if f"{name}_{i}_grid" in Z.files:
            mm = r["multi_modal_inputs"]
            assert np.array_equal(mm["image_grid_thw"].numpy(), Z[f"{name}_{i}_grid"])
            pv = mm["pixel_values"].numpy()
            st = Z[f"{name}_{i}_pix_stats"]
            assert pv.shape == (int(st[0]), int(st[1]))
            np.testing.assert_allclose([float(pv.sum()), float(np.abs(pv).sum()), float(pv[0, :8].sum())], st[2:], rtol=1e-6)
            assert tuple(r["multi_modal_data"]["image"][0].size) == tuple(Z[f"{name}_{i}_image_size"])
        else:
            assert "multi_modal_inputs" not in r and r["position_ids"].dim() == 1
    assert "problem" in ds[0] and "answer_option_text" not in ds[0] or CASES[name]["answer_key"] != "answer_option_text"


def test_collate_fn_stacks_tensors_and_wraps_objects():
    ds = _ds(CASES["spatial"])
    b = collate_fn([ds[0], ds[1]])
    assert sorted(b.keys()) == Z["collate_keys"].tolist()
    assert np.array_equal(b["input_ids"].numpy(), Z["collate_input_ids"])
    assert isinstance(b["multi_modal_inputs"], np.ndarray) and b["multi_modal_inputs"].dtype == object and len(b["ground_truth"]) == 2


def test_missing_split_raises_and_single_file_path_loads():
    with pytest.raises(ValueError):
        RLHFDataset(f"{ROOT}@test", StubTokenizer(), StubProcessor(), prompt_key="problem", answer_key="answer_option_text")
    one = RLHFDataset(os.path.join(ROOT, "val-00000-of-00001.parquet") + "@val", StubTokenizer(), StubProcessor(), prompt_key="problem",
                      answer_key="answer_option_text", max_prompt_length=96, truncation="right", shuffle=False)
    assert len(one) == 2
    with pytest.raises(NotImplementedError):                          # truncation="error" and a prompt longer than the limit
        RLHFDataset(f"{ROOT}@train", StubTokenizer(), StubProcessor(), prompt_key="problem", answer_key="answer_option_text",
                    max_prompt_length=16, truncation="error", shuffle=False)[0]
