"""A REAL (tiny) Hugging Face model directory for the tokenizer / processor path (verl/utils/tokenizer.py:21-50,
verl/utils/dataset.py:186-265): a byte-level `PreTrainedTokenizerFast` with Qwen's special tokens and chat template, a real
`Qwen2VLImageProcessor`, and the tiny Qwen2.5-VL weights written by this build's save_hf — everything `AutoTokenizer` /
`AutoImageProcessor` / load_model read from disk.  Built locally (no network): tokenizers + transformers are in the image."""
from __future__ import annotations

import os

import tiny

SPECIALS = ["<|image_pad|>", "<|vision_start|>", "<|vision_end|>", "<|endoftext|>", "<|im_end|>", "<|im_start|>", "<|video_pad|>"]
CHAT_TEMPLATE = ("{% for message in messages %}<|im_start|>{{ message['role'] }}\n{% if message['content'] is string %}{{ message['content'] }}"
                 "{% else %}{% for c in message['content'] %}{% if c['type'] == 'image' %}<|vision_start|><|image_pad|><|vision_end|>"
                 "{% elif c['type'] == 'text' %}{{ c['text'] }}{% endif %}{% endfor %}{% endif %}<|im_end|>\n{% endfor %}"
                 "{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}")


def build_tokenizer():
    """ids 0..255 = the byte-level alphabet, 256..1009 filler, then the specials at exactly the ids of tiny.TINY
    (image_pad 1010, vision_start 1011, vision_end 1012, pad <|endoftext|> 1013, eos <|im_end|> 1014)."""
    from tokenizers import AddedToken, Tokenizer, decoders, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {ch: i for i, ch in enumerate(sorted(pre_tokenizers.ByteLevel.alphabet()))}
    for i in range(len(vocab), tiny.TINY["image_token_id"]):
        vocab[f"<fill_{i}>"] = i
    tk = Tokenizer(models.BPE(vocab=vocab, merges=[]))
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tk.decoder = decoders.ByteLevel()
    tk.add_special_tokens([AddedToken(s, special=True) for s in SPECIALS])
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="<|im_end|>", pad_token="<|endoftext|>", additional_special_tokens=SPECIALS,
                                  chat_template=CHAT_TEMPLATE)
    assert tok.convert_tokens_to_ids("<|image_pad|>") == tiny.TINY["image_token_id"] and tok.eos_token_id == tiny.EOS_ID and tok.pad_token_id == tiny.PAD_ID
    return tok


def build_model_dir(path: str, device: str = "cpu") -> str:
    """weights (tiny.make_params) + config + generation config + tokenizer + image-processor files under `path`."""
    import torch
    from transformers import Qwen2VLImageProcessor

    from spatialthinker_amd import model as mdl
    from spatialthinker_amd.pretrained import hf_config_dict, save_hf
    cfg = mdl.VLConfig(**tiny.TINY)
    store = mdl.ParamStore(cfg, device=device, trainable=False)
    store.load_hf_state_dict({k: torch.from_numpy(v) for k, v in tiny.make_params().items()})
    store.hf_config = hf_config_dict(cfg, {"eos": tiny.EOS_ID, "pad": tiny.PAD_ID})
    store.generation_config = {"eos_token_id": tiny.EOS_ID, "pad_token_id": tiny.PAD_ID}
    store.source_dir = None
    tok = build_tokenizer()
    save_hf(store, path, tokenizer=tok)
    ip = Qwen2VLImageProcessor(min_pixels=4 * 28 * 28, max_pixels=64 * 28 * 28, patch_size=tiny.TINY["v_patch"], merge_size=tiny.TINY["v_merge"],
                               temporal_patch_size=tiny.TINY["v_temporal_patch"])
    ip.save_pretrained(path)
    with open(os.path.join(path, "chat_template.jinja"), "w") as f:
        f.write(CHAT_TEMPLATE)
    return path
