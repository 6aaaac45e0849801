"""Datasets producing the prompt-side batch of the GRPO loop.

RLHFDataset mirrors the reference row pipeline (verl/utils/dataset.py:34-265): chat template ->
processor -> pixel_values / image_grid_thw -> M-RoPE ids -> left-pad / truncate to max_prompt_length; it needs the real
tokenizer + processor files and a parquet/HF dataset (a tiny parquet + stub processor pin it in tests).  SyntheticSTVQADataset emits
STVQA-7K-shaped rows without any of them (`data.train_files=synthetic:stvqa@train`)."""
from __future__ import annotations

import json
import math
from collections import defaultdict
from typing import Any, Dict, List, Optional

import numpy as np
import torch
from torch.utils.data import Dataset

from spatialthinker_amd import indexing as ix

from .torch_functional import postprocess_data


def collate_fn(features: List[Dict[str, Any]]) -> Dict[str, Any]:
    tensors, others = defaultdict(list), defaultdict(list)
    for feat in features:
        for k, v in feat.items():
            (tensors if isinstance(v, torch.Tensor) else others)[k].append(v)
    out = {k: torch.stack(v, 0) for k, v in tensors.items()}
    for k, v in others.items():
        arr = np.empty(len(v), dtype=object)
        for i, x in enumerate(v):
            arr[i] = x
        out[k] = arr
    return out


class SyntheticSTVQADataset(Dataset):
    """STVQA-7K-shaped rows: ~766 text tokens + one Visual-Genome-like image (588x448 -> 1344 patches -> 336 image tokens),
    a `problem` column ending in "Image size: (W x H)" and an `answer_option_text` scene-graph ground truth."""

    def __init__(self, model_cfg, tokenizer, size: int = 4096, max_prompt_length: int = 1152, seed: int = 1, grid=(1, 32, 42),
                 text_tokens=(200, 564), answer_key: str = "answer", response_lengths=None):
        """response_lengths = (mean, std, n, cap): synthetic-benchmark mode (`synthetic:stvqa:len=512,128@train`, bench.py --through-api) —
        every row carries `synthetic_response_lengths`, the n response lengths ~ clip(N(mean, std), min(64, cap), cap) its rollouts are cut
        to; the trainer hands them to the rollout as meta_info (random-init weights would otherwise never emit EOS)."""
        self.cfg, self.tok, self.size, self.P, self.seed, self.grid, self.text_tokens = model_cfg, tokenizer, size, max_prompt_length, seed, grid, text_tokens
        self.answer_key = answer_key
        self.response_lengths = response_lengths
        self.epoch_salt = 0

    def __len__(self):
        return self.size

    def __getitem__(self, index: int) -> Dict[str, Any]:
        c = self.cfg
        rs = np.random.RandomState(self.seed * 1_000_003 + index)
        t, h, w = self.grid
        n_img = t * h * w // (c.v_merge ** 2)
        hi = min(c.image_token_id, c.vocab_size) - 16
        ids = np.concatenate([rs.randint(0, hi, self.text_tokens[0]), [c.vision_start_token_id], np.full(n_img, c.image_token_id),
                              [c.vision_start_token_id + 1], rs.randint(0, hi, self.text_tokens[1])]).astype(np.int64)
        grid = np.asarray([self.grid], dtype=np.int64)
        mask = np.ones_like(ids)
        pos = ix.get_rope_index(ids, grid, mask, image_token_id=c.image_token_id, vision_start_token_id=c.vision_start_token_id,
                                spatial_merge_size=c.v_merge)
        input_ids, attention_mask, position_ids = postprocess_data(torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(pos),
                                                                   self.P, self.tok.pad_token_id, True, "right")
        k = rs.randint(2, 6)
        objs = [{"id": f"obj_{'abcde'[j]}.{j + 1}", "bbox": [int(v) for v in sorted(rs.randint(0, 400, 2)) + sorted(rs.randint(0, 300, 2))]} for j in range(k)]
        objs = [{"id": o["id"], "bbox": [o["bbox"][0], o["bbox"][2], o["bbox"][1] + 5, o["bbox"][3] + 5]} for o in objs]
        gt = f"<scene>{json.dumps({'objects': objs, 'relationships': []})}</scene>\n<answer>(A) yes</answer>"
        pix = torch.from_numpy(rs.standard_normal((t * h * w, c.patch_k)).astype(np.float32))
        row = {"input_ids": input_ids, "attention_mask": attention_mask, "position_ids": position_ids,
               "raw_prompt_ids": ids.tolist(), "multi_modal_data": {"image": [None]},
               "multi_modal_inputs": {"pixel_values": pix, "image_grid_thw": torch.from_numpy(grid)},
               "ground_truth": gt, "problem": "Synthetic scene. Image size: (588 x 448)\nQ. is it?\nOptions: (A) yes (B) no"}
        if self.response_lengths is not None:
            mu, sd, n, cap = self.response_lengths
            row["synthetic_response_lengths"] = np.clip(rs.normal(mu, sd, int(n)), min(64, cap), cap).astype(np.int64)
        return row


class ImageProcessMixin:
    """`process_image` for classes that carry max_pixels / min_pixels (reference verl/utils/dataset.py:52-75; a vLLM-side processor mixes it
    in too)."""
    max_pixels: Optional[int]
    min_pixels: Optional[int]

    def process_image(self, image):
        """bytes/dict -> PIL, down-scale above max_pixels, up-scale below min_pixels, RGB."""
        from io import BytesIO

        from PIL import Image
        if isinstance(image, dict):
            image = Image.open(BytesIO(image["bytes"]))
        elif isinstance(image, bytes):
            image = Image.open(BytesIO(image))
        if self.max_pixels and image.width * image.height > self.max_pixels:
            f = math.sqrt(self.max_pixels / (image.width * image.height))
            image = image.resize((int(image.width * f), int(image.height * f)))
        if self.min_pixels and image.width * image.height < self.min_pixels:
            f = math.sqrt(self.min_pixels / (image.width * image.height))
            image = image.resize((int(image.width * f), int(image.height * f)))
        return image.convert("RGB") if image.mode != "RGB" else image


class RLHFDataset(Dataset, ImageProcessMixin):
    """Real-data row pipeline of the reference (verl/utils/dataset.py:79-265), pinned against the reference
    class itself by tests/golden/dataset.npz (tests/test_dataset.py).  data_path = "<dir | parquet file | hub id>[@split]"."""

    def __init__(self, data_path: str, tokenizer, processor, prompt_key="prompt", answer_key="answer", image_key="images",
                 mixed_data: bool = False, text_only: bool = False, max_prompt_length=1024, truncation="error",
                 format_prompt: Optional[str] = None, max_pixels=None, min_pixels=None, shuffle: bool = True, seed: int = 42):
        import glob
        import os

        from datasets import load_dataset
        self.tokenizer, self.processor = tokenizer, processor
        self.prompt_key, self.answer_key, self.image_key = prompt_key, answer_key, image_key
        self.mixed_data, self.text_only = mixed_data, text_only
        self.max_prompt_length, self.truncation, self.format_prompt = max_prompt_length, truncation, format_prompt
        self.max_pixels, self.min_pixels = max_pixels, min_pixels
        split = "train"
        if "@" in data_path:
            data_path, split = data_path.split("@")
        if os.path.isdir(data_path):
            # a local parquet directory holds "<split>-*.parquet" shards (dataset.py:121-148); the train split also registers the
            # val files, exactly as the reference does
            files = sorted(glob.glob(os.path.join(data_path, f"{split}-*.parquet")))
            if not files:
                raise ValueError(f"No files found for split '{split}' at path '{data_path}'")
            data_files = {split: files}
            if split == "train":
                val = sorted(glob.glob(os.path.join(data_path, "val-*.parquet")))
                if val:
                    data_files["val"] = val
            self.dataset = load_dataset("parquet", data_files=data_files, split=split)
        elif os.path.isfile(data_path):
            self.dataset = load_dataset("parquet", data_files={split: [data_path]}, split=split)
        else:                                                   # hub dataset id, e.g. hunarbatra/STVQA-7K@train
            self.dataset = load_dataset(data_path, split=split)
        if mixed_data:                                          # :159-170 every even row loses its <image> marker (text-only half)
            def drop_image(example, idx):
                if idx % 2 == 0 and self.image_key in example and "<image>" in example[self.prompt_key]:
                    example[self.prompt_key] = example[self.prompt_key].replace("<image>", "").strip()
                return example
            self.dataset = self.dataset.map(drop_image, with_indices=True, desc="Removing <image> from prompt_key")
        if shuffle:
            self.dataset = self.dataset.shuffle(seed=seed)

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        row = dict(self.dataset[index])
        prompt = row[self.prompt_key]
        if self.format_prompt:
            prompt = self.format_prompt.strip() + " " + prompt          # prefix (:189-191)
        if self.text_only:
            prompt = prompt.replace("<image>", "").strip()
        if "<image>" in prompt and self.image_key in row and row[self.image_key] is not None:
            # exactly one image marker, moved to the front (:205-206); text pieces are passed on verbatim (with their spaces)
            prompt = "<image> " + prompt.replace("<image>", "").strip()
            content = []
            for i, piece in enumerate(prompt.split("<image>")):
                if i != 0:
                    content.append({"type": "image"})
                if piece:
                    content.append({"type": "text", "text": piece})
            text = self.processor.apply_chat_template([{"role": "user", "content": content}], add_generation_prompt=True, tokenize=False)
            images = row.pop(self.image_key)
            if not isinstance(images, list):
                images = [images]
            if any(im is None for im in images):
                raise ValueError(f"Image is None at index {index} despite <image> token present. Check data logic.")
            images = [self.process_image(im) for im in images]
            enc = dict(self.processor(images, [text], return_tensors="pt"))
            input_ids, attention_mask = enc.pop("input_ids")[0], enc.pop("attention_mask")[0]
            row["multi_modal_data"] = {"image": images}
            row["multi_modal_inputs"] = enc
            tok = self.processor.tokenizer
            position_ids = torch.from_numpy(ix.get_rope_index(
                input_ids.numpy(), enc["image_grid_thw"].numpy(), attention_mask.numpy(), image_token_id=tok.convert_tokens_to_ids("<|image_pad|>"),
                vision_start_token_id=tok.convert_tokens_to_ids("<|vision_start|>"), spatial_merge_size=self.processor.image_processor.merge_size))
        else:
            text = self.tokenizer.apply_chat_template([{"role": "user", "content": prompt}], add_generation_prompt=True, tokenize=False)
            enc = self.tokenizer([text], add_special_tokens=False, return_tensors="pt")
            input_ids, attention_mask = enc["input_ids"][0], enc["attention_mask"][0]
            position_ids = torch.clip(attention_mask.cumsum(0) - 1, min=0)
        input_ids, attention_mask, position_ids = postprocess_data(input_ids, attention_mask, position_ids, self.max_prompt_length,
                                                                   self.tokenizer.pad_token_id, True, self.truncation)
        row.update(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                   raw_prompt_ids=self.tokenizer.encode(text, add_special_tokens=False), ground_truth=row.pop(self.answer_key))
        return row
