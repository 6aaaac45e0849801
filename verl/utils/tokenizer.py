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


def _image_only_processor(model_path: str, **kwargs):
    """Qwen2(.5)-VL processor assembled from its parts when AutoProcessor cannot build it: transformers 5.x gives the processor a VIDEO
    sub-processor whose only implementation needs torchvision, which this image does not ship — the image + text pipeline the GRPO path
    uses (verl/utils/dataset.py:186-265) needs none of it.  The video slot gets an inert BaseVideoProcessor; the image processor and the
    tokenizer are the checkpoint's own (Qwen2VLImageProcessor falls back to its PIL backend when torchvision is missing)."""
    import json
    import os

    from transformers import AutoTokenizer
    from transformers.video_processing_utils import BaseVideoProcessor
    with open(os.path.join(model_path, "config.json")) as f:
        mtype = json.load(f).get("model_type", "")
    if mtype == "qwen2_5_vl":
        from transformers import Qwen2_5_VLProcessor as Proc
    elif mtype == "qwen2_vl":
        from transformers import Qwen2VLProcessor as Proc
    else:
        return None

    import transformers
    # without torchvision `transformers.BaseVideoProcessor` is a placeholder class, and it is the class ProcessorMixin type-checks the
    # video slot against — the slot inherits from the real base and, where they differ, from the placeholder as well
    bases = (BaseVideoProcessor,) if transformers.BaseVideoProcessor is BaseVideoProcessor else (BaseVideoProcessor, transformers.BaseVideoProcessor)

    class _PlainAttrs(type(bases[-1])):                      # the placeholder's metaclass refuses every class-attribute read
        def __getattribute__(cls, key):
            return type.__getattribute__(cls, key)

    class ImageOnlyVideoSlot(*bases, metaclass=_PlainAttrs):
        model_input_names = ["pixel_values_videos", "video_grid_thw"]

        def __call__(self, *a, **k):
            raise NotImplementedError("video inputs are outside the SpatialThinker path (image + text prompts only)")
    tok = AutoTokenizer.from_pretrained(model_path, **kwargs)
    try:
        from transformers import AutoImageProcessor
        ip = AutoImageProcessor.from_pretrained(model_path)
    except ImportError:                                     # AutoImageProcessor itself wants torchvision; the model's own class has a PIL backend
        from transformers import Qwen2VLImageProcessor
        ip = Qwen2VLImageProcessor.from_pretrained(model_path)
    tpl = getattr(tok, "chat_template", None)
    for name in ("chat_template.jinja", "chat_template.json"):
        fp = os.path.join(model_path, name)
        if tpl is None and os.path.exists(fp):
            raw = open(fp).read()
            tpl = json.loads(raw)["chat_template"] if name.endswith(".json") else raw
    return Proc(image_processor=ip, tokenizer=tok, video_processor=ImageOnlyVideoSlot(), chat_template=tpl)


def get_processor(model_path: str, **kwargs):
    if is_synthetic(model_path):
        return None
    from transformers import AutoProcessor
    try:
        proc = AutoProcessor.from_pretrained(model_path, **kwargs)
    except Exception as e:
        proc = None
        try:
            proc = _image_only_processor(model_path, **kwargs)
            if proc is not None:
                print(f"AutoProcessor could not load {model_path} ({type(e).__name__}); using the image + text processor assembled from its parts.")
        except Exception:
            proc = None
    if proc is not None and "Processor" not in proc.__class__.__name__:      # reference :46-48: a bare tokenizer is not a processor
        proc = None
    return proc
