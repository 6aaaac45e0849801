"""Model construction: HF checkpoint directories (config.json + *.safetensors) or synthetic `random:<size>` models."""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Tuple

import torch

from .model import ParamStore, VLConfig

_TINY = dict(hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, vocab_size=1024, v_depth=3, v_hidden=320,
             v_heads=4, v_intermediate=200, v_window=56, v_fullatt=[1], image_token_id=1010, vision_start_token_id=1011)


def synthetic_config(model_path: str) -> Tuple[VLConfig, Dict[str, int]]:
    size = model_path.split(":", 1)[1].lower()
    if size == "7b":
        return VLConfig.qwen2_5_vl_7b(), {"eos": 151645, "pad": 151643}
    if size == "3b":
        return VLConfig.qwen2_5_vl_3b(), {"eos": 151645, "pad": 151643}
    if size == "tiny":
        return VLConfig(**_TINY), {"eos": 1014, "pad": 1013}
    raise ValueError(f"unknown synthetic model {model_path!r} (expected random:7b|3b|tiny)")


def load_model(model_path: str, trainable: bool, device="cuda", seed: int = 7) -> Tuple[VLConfig, ParamStore, Dict[str, int]]:
    if model_path.startswith("random:"):
        cfg, special = synthetic_config(model_path)
        store = ParamStore(cfg, device=device, trainable=trainable)
        store.init_random(seed=seed)
        return cfg, store, special
    with open(os.path.join(model_path, "config.json")) as f:
        hf = json.load(f)
    cfg = VLConfig.from_hf_dict(hf)
    store = ParamStore(cfg, device=device, trainable=trainable)
    from safetensors.torch import load_file
    sd: Dict[str, torch.Tensor] = {}
    for shard in sorted(glob.glob(os.path.join(model_path, "*.safetensors"))):
        sd.update(load_file(shard))
    # transformers < 4.52 checkpoints name the towers "visual.*" / "model.*"; normalise to the 5.x names used by ParamStore
    norm = {}
    for k, v in sd.items():
        if k.startswith("visual."):
            k = "model." + k
        elif k.startswith("model.") and not k.startswith(("model.visual.", "model.language_model.")):
            k = "model.language_model." + k[len("model."):]
        norm[k] = v
    store.load_hf_state_dict(norm)
    gen = {}
    gp = os.path.join(model_path, "generation_config.json")
    if os.path.exists(gp):
        gen = json.load(open(gp))
    eos = gen.get("eos_token_id", hf.get("eos_token_id", 151645))
    return cfg, store, {"eos": eos, "pad": gen.get("pad_token_id", hf.get("pad_token_id", 151643))}


def save_hf(store: ParamStore, path: str) -> None:
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    save_file({k: v.detach().cpu().contiguous() for k, v in store.export_hf().items()}, os.path.join(path, "model.safetensors"))
