"""Tokenizer / processor loaders (reference: verl/utils/tokenizer.py:21-50) with an offline synthetic fallback.

`model_path = "random:<7b|3b|tiny>"` selects random-init weights at real shapes and the SyntheticTokenizer below
(no tokenizer files exist offline); any other path is loaded through transformers exactly as the reference does."""
from __future__ import annotations

from typing import List, Optional


class SyntheticTokenizer:
    """Byte-level stand-in: ids 0..255 are bytes; special ids follow the model config."""

    def __init__(self, vocab_size: int, eos_token_id: int, pad_token_id: int, image_token_id: int, vision_start_token_id: int):
        self.vocab_size, self.eos_token_id, self.pad_token_id = vocab_size, eos_token_id, pad_token_id
        self.bos_token_id = None
        self._special = {"<|image_pad|>": image_token_id, "<|vision_start|>": vision_start_token_id,
                         "<|vision_end|>": vision_start_token_id + 1, "<|video_pad|>": image_token_id + 1}

    def convert_tokens_to_ids(self, tok: str) -> int:
        return self._special[tok]

    def encode(self, text: str, add_special_tokens: bool = False) -> List[int]:
        return list(text.encode("utf-8"))

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        return bytes(int(i) for i in ids if 0 <= int(i) < 256).decode("utf-8", errors="replace")


def is_synthetic(model_path: Optional[str]) -> bool:
    return bool(model_path) and model_path.startswith("random:")


def get_tokenizer(model_path: str, **kwargs):
    if is_synthetic(model_path):
        from spatialthinker_amd.pretrained import synthetic_config
        cfg, special = synthetic_config(model_path)
        return SyntheticTokenizer(cfg.vocab_size, special["eos"], special["pad"], cfg.image_token_id, cfg.vision_start_token_id)
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(model_path, **kwargs)
    if tok.pad_token_id is None:
        tok.pad_token = tok.eos_token
    return tok


def get_processor(model_path: str, **kwargs):
    if is_synthetic(model_path):
        return None
    from transformers import AutoProcessor
    try:
        proc = AutoProcessor.from_pretrained(model_path, **kwargs)
    except Exception:
        proc = None
    if proc is not None and "Processor" not in proc.__class__.__name__:      # reference :46-48: a bare tokenizer is not a processor
        proc = None
    return proc
