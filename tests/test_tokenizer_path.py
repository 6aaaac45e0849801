"""The NON-synthetic branches of the tokenizer / processor / dataset path (reference verl/utils/tokenizer.py:21-50,
verl/utils/dataset.py:186-265, verl/workers/reward/custom.py:48-73) on a REAL tokenizer + processor built locally
(tests/golden/tiny_hf.py: a byte-level PreTrainedTokenizerFast with Qwen's special tokens and chat template, a real
Qwen2VLImageProcessor, tiny Qwen2.5-VL weights): AutoTokenizer / the Qwen2_5_VLProcessor loaders, pretrained.load_model(<dir>),
RLHFDataset rows on the committed STVQA-shaped parquet, decode -> spatial_sgg reward.  CPU only; the GPU end-to-end run of the same
directory through `python -m verl.trainer.main` is tests/test_gpu_e2e.py::test_main_with_a_real_tokenizer_and_processor."""
import os

import numpy as np
import pytest
import torch

import tiny
import tiny_hf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "stvqa_tiny")


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    return tiny_hf.build_model_dir(str(tmp_path_factory.mktemp("tiny_hf") / "model"))


def test_loaders_return_a_real_tokenizer_and_a_qwen25vl_processor(model_dir):
    from verl.utils.tokenizer import get_processor, get_tokenizer
    tok = get_tokenizer(model_dir, trust_remote_code=True, use_fast=True)
    proc = get_processor(model_dir, trust_remote_code=True, use_fast=True)
    assert tok.__class__.__name__ != "SyntheticTokenizer" and hasattr(tok, "apply_chat_template")
    assert proc.__class__.__name__ == "Qwen2_5_VLProcessor" and "ImageProcessor" in proc.image_processor.__class__.__name__
    assert tok.pad_token_id == tiny.PAD_ID and tok.eos_token_id == tiny.EOS_ID
    assert tok.convert_tokens_to_ids("<|image_pad|>") == tiny.TINY["image_token_id"]
    assert tok.convert_tokens_to_ids("<|vision_start|>") == tiny.TINY["vision_start_token_id"]
    s = "<scene>{\"objects\": []}</scene> ünï"
    assert tok.decode(tok.encode(s, add_special_tokens=False), skip_special_tokens=True) == s       # byte-level round trip


def test_load_model_reads_the_directory_and_matches_the_source_weights(model_dir):
    from spatialthinker_amd import model as mdl
    from spatialthinker_amd.pretrained import load_model
    cfg, store, special = load_model(model_dir, trainable=False, device="cpu")
    assert cfg == mdl.VLConfig(**tiny.TINY) and special == {"eos": tiny.EOS_ID, "pad": tiny.PAD_ID}
    back = store.export_hf()
    for k, v in tiny.make_params().items():
        assert torch.equal(back[k].float(), torch.from_numpy(v)), k


def test_rlhf_dataset_rows_through_the_real_processor(model_dir):
    from spatialthinker_amd import indexing as ix
    from verl.utils.dataset import RLHFDataset, collate_fn
    from verl.utils.tokenizer import get_processor, get_tokenizer
    tok, proc = get_tokenizer(model_dir, use_fast=True), get_processor(model_dir, use_fast=True)
    ds = RLHFDataset(f"{DATA}@train", tok, proc, prompt_key="problem", answer_key="answer_option_text", image_key="images", max_prompt_length=160,
                     truncation="right", min_pixels=4 * 28 * 28, max_pixels=64 * 28 * 28, shuffle=False)
    rows = [ds[i] for i in range(len(ds))]
    for r in rows:
        ids, am = r["input_ids"].numpy(), r["attention_mask"].numpy()
        assert ids.shape == (160,) and am[:160 - am.sum()].sum() == 0                               # left-padded to max_prompt_length
        grid = r["multi_modal_inputs"]["image_grid_thw"].numpy()
        n_img = int(grid.prod(1).sum()) // tiny.TINY["v_merge"] ** 2
        assert int((ids == tiny.TINY["image_token_id"]).sum()) == n_img > 0                         # <|image_pad|> expanded to the grid
        assert r["multi_modal_inputs"]["pixel_values"].shape == (int(grid.prod(1).sum()), 1176)
        text = tok.decode(ids[am == 1], skip_special_tokens=False)
        assert text.startswith("<|im_start|>user\n<|vision_start|>") and text.endswith("<|im_start|>assistant\n") and "Image size: (" in text
        # M-RoPE ids of the row = the engine's own get_rope_index on the row's tokens (golden-pinned against the reference's function)
        valid = am == 1
        want = ix.get_rope_index(ids[valid], grid, am[valid], image_token_id=tiny.TINY["image_token_id"],
                                 vision_start_token_id=tiny.TINY["vision_start_token_id"], spatial_merge_size=2)
        assert np.array_equal(r["position_ids"].numpy()[:, valid], want)
        # raw_prompt_ids = the chat text with ONE <|image_pad|> per image (dataset.py:259: what vLLM would expand itself)
        v_ids = ids[valid].tolist()
        collapsed = [t for j, t in enumerate(v_ids) if not (t == tiny.TINY["image_token_id"] and j > 0 and v_ids[j - 1] == t)]
        assert r["ground_truth"].startswith("<scene>") and r["raw_prompt_ids"] == collapsed
    batch = collate_fn(rows[:4])
    assert batch["input_ids"].shape == (4, 160) and batch["multi_modal_inputs"].dtype == object


def test_decode_then_spatial_reward_through_the_reward_manager(model_dir):
    """CustomRewardManager with the real tokenizer: response ids -> text (skip_special_tokens) -> spatial_sgg score at the last valid token."""
    import types
    from verl.protocol import DataProto
    from verl.utils.reward_score import spatial_sgg_compute_score
    from verl.utils.tokenizer import get_tokenizer
    from verl.workers.reward import CustomRewardManager
    tok = get_tokenizer(model_dir, use_fast=True)
    gt = '<scene>{"objects": [{"id": "cat.1", "bbox": [4, 4, 40, 30]}], "relationships": []}</scene>\n<answer>(A) a0</answer>'
    good = '<observe>a cat</observe>\n<scene>{"objects": [{"id": "cat.1", "bbox": [5, 4, 41, 31]}], "relationships": []}</scene>\n<think>left</think>\n<answer>(A) a0</answer>'
    bad = "no structure at all"
    R = 256
    resp = torch.full((2, R), tok.pad_token_id, dtype=torch.int64)
    mask = torch.zeros(2, R, dtype=torch.int64)
    for i, s in enumerate((good, bad)):
        ids = tok.encode(s, add_special_tokens=False) + [tok.eos_token_id]
        resp[i, :len(ids)] = torch.tensor(ids); mask[i, :len(ids)] = 1
    rm = CustomRewardManager(tok, types.SimpleNamespace(score_function="spatial_sgg", skip_special_tokens=True))
    problems = np.array(["Image size: (64 x 48)\nQ. what?"] * 2, dtype=object)
    data = DataProto.from_dict(tensors={"responses": resp, "response_mask": mask}, non_tensors={"ground_truth": np.array([gt, gt], dtype=object), "problem": problems})
    rewards, metrics = rm(data)
    want = [spatial_sgg_compute_score(s, gt, problems[0])["overall"] for s in (good, bad)]
    for i in range(2):
        last = int(mask[i].sum()) - 1
        assert abs(float(rewards[i, last]) - want[i]) < 1e-6 and float(rewards[i].abs().sum()) == pytest.approx(abs(want[i]), abs=1e-6)
    assert want[0] > want[1]
