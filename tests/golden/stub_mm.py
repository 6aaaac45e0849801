"""Deterministic stand-ins for the tokenizer / processor files that do not exist offline (no Qwen2.5-VL tokenizer.json in the
build container or on the GPU box).  Used on BOTH sides of the dataset fixture: make_golden.py feeds them to the reference's
RLHFDataset, tests/test_dataset.py feeds them to this repo's — so the fixture pins the row pipeline (prompt assembly, image
resizing, placeholder expansion, M-RoPE ids, padding / truncation, raw_prompt_ids), not a vocabulary.
The image side is transformers' real Qwen2VLImageProcessor (PIL backend, works offline)."""
import re

import numpy as np
import torch

SPECIAL = {"<|image_pad|>": 500, "<|vision_start|>": 502, "<|vision_end|>": 503, "<|im_start|>": 504, "<|im_end|>": 505, "<|video_pad|>": 501}
_SPLIT = re.compile("(" + "|".join(re.escape(k) for k in SPECIAL) + ")")


class StubTokenizer:
    pad_token_id, eos_token_id = 0, 505

    def convert_tokens_to_ids(self, tok):
        return SPECIAL[tok]

    def encode(self, text, add_special_tokens=False):
        ids = []
        for piece in _SPLIT.split(text):
            if piece in SPECIAL:
                ids.append(SPECIAL[piece])
            else:
                ids.extend(20 + (ord(c) % 400) for c in piece)
        return ids

    def __call__(self, texts, add_special_tokens=False, return_tensors="pt"):
        ids = [self.encode(t) for t in texts]
        assert len(ids) == 1
        return {"input_ids": torch.tensor(ids), "attention_mask": torch.ones(1, len(ids[0]), dtype=torch.long)}

    def apply_chat_template(self, messages, add_generation_prompt=True, tokenize=False):
        out = ""
        for m in messages:
            out += f"<|im_start|>{m['role']}\n"
            if isinstance(m["content"], str):
                out += m["content"]
            else:
                for c in m["content"]:
                    out += "<|vision_start|><|image_pad|><|vision_end|>" if c["type"] == "image" else c["text"]
            out += "<|im_end|>\n"
        return out + ("<|im_start|>assistant\n" if add_generation_prompt else "")

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(int(i)) for i in ids)


class StubProcessor:
    def __init__(self, min_pixels=28 * 28, max_pixels=64 * 28 * 28):
        from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil as IP
        self.image_processor = IP(min_pixels=min_pixels, max_pixels=max_pixels)
        self.tokenizer = StubTokenizer()

    def apply_chat_template(self, messages, add_generation_prompt=True, tokenize=False):
        return self.tokenizer.apply_chat_template(messages, add_generation_prompt, tokenize)

    def __call__(self, images, text, return_tensors="pt", **_):
        enc = self.image_processor(images=images, return_tensors="pt")
        grid = enc["image_grid_thw"]
        m2 = self.image_processor.merge_size ** 2
        t, k = text[0], 0
        while "<|image_pad|>" in t:
            t = t.replace("<|image_pad|>", "<|placeholder|>" * int(grid[k].prod() // m2), 1)
            k += 1
        t = t.replace("<|placeholder|>", "<|image_pad|>")
        ids = self.tokenizer.encode(t)
        return {"input_ids": torch.tensor([ids]), "attention_mask": torch.ones(1, len(ids), dtype=torch.long),
                "pixel_values": enc["pixel_values"], "image_grid_thw": grid}
