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


def resolve_model_path(model_path: str) -> str:
    """A local directory is used as is; anything else is treated as a Hugging Face hub id (what every shipped script passes,
    e.g. scripts/spatialthinker_7b_grpo.sh:3 MODEL_PATH=Qwen/Qwen2.5-VL-7B-Instruct) and resolved to its local snapshot —
    downloaded once, or found in the hub cache when the machine is offline (HF_HUB_OFFLINE=1)."""
    if os.path.isdir(model_path):
        return model_path
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(model_path, allow_patterns=["*.json", "*.safetensors", "*.txt", "*.model", "*.jinja"])
    except Exception as e:
        raise FileNotFoundError(f"model_path {model_path!r} is neither a local directory nor a downloadable / cached hub "
                                f"repository ({type(e).__name__}: {e})") from e


def hf_config_dict(cfg: VLConfig, special: Dict[str, int]) -> dict:
    """A transformers-loadable config.json (model_type qwen2_5_vl) for a model built from a VLConfig (synthetic models, or as a
    fallback when the source directory's config is unavailable)."""
    eos = special.get("eos", 151645)
    return {
        "architectures": ["Qwen2_5_VLForConditionalGeneration"], "model_type": "qwen2_5_vl", "torch_dtype": "bfloat16",
        "tie_word_embeddings": cfg.tie_word_embeddings, "image_token_id": cfg.image_token_id, "video_token_id": cfg.image_token_id + 1,
        "vision_start_token_id": cfg.vision_start_token_id, "vision_end_token_id": cfg.vision_start_token_id + 1,
        "eos_token_id": eos if isinstance(eos, int) else eos[0], "pad_token_id": special.get("pad", 151643),
        "text_config": {"model_type": "qwen2_5_vl_text", "hidden_size": cfg.hidden_size, "intermediate_size": cfg.intermediate_size,
                        "num_hidden_layers": cfg.num_layers, "num_attention_heads": cfg.num_heads, "num_key_value_heads": cfg.num_kv_heads,
                        "vocab_size": cfg.vocab_size, "rms_norm_eps": cfg.rms_eps, "hidden_act": "silu", "max_position_embeddings": 128000,
                        "tie_word_embeddings": cfg.tie_word_embeddings, "use_sliding_window": False,
                        "rope_parameters": {"rope_type": "default", "rope_theta": cfg.rope_theta, "mrope_section": list(cfg.mrope_section)},
                        "rope_scaling": {"type": "mrope", "mrope_section": list(cfg.mrope_section)}, "rope_theta": cfg.rope_theta},
        "vision_config": {"model_type": "qwen2_5_vl", "depth": cfg.v_depth, "hidden_size": cfg.v_hidden, "num_heads": cfg.v_heads,
                          "intermediate_size": cfg.v_intermediate, "out_hidden_size": cfg.hidden_size, "patch_size": cfg.v_patch,
                          "temporal_patch_size": cfg.v_temporal_patch, "spatial_merge_size": cfg.v_merge, "window_size": cfg.v_window,
                          "fullatt_block_indexes": list(cfg.v_fullatt), "in_channels": cfg.v_in_channels, "hidden_act": "silu",
                          "tokens_per_second": 2},
    }


def load_model(model_path: str, trainable: bool, device="cuda", seed: int = 7, master_fp32: bool = False,
               value_head: bool = False) -> Tuple[VLConfig, ParamStore, Dict[str, int]]:
    """master_fp32: enable the fp32 master BEFORE the weights are loaded, so that an fp32 checkpoint's own values reach the master (the
    reference keeps them: fp32 shards under MixedPrecision(param_dtype=bf16)); enabling it afterwards would start from bf16-rounded weights.
    value_head: the critic's model (fsdp_workers.py:212-224 loads AutoModelForTokenClassification with num_labels = 1): the backbone with
    score = Linear(H, 1) instead of the lm_head; a checkpoint without `score.*` gets the head HF would initialise (normal(0, 0.02), zero bias)."""
    import dataclasses
    if model_path.startswith("random:"):
        cfg, special = synthetic_config(model_path)
        cfg = dataclasses.replace(cfg, value_head=True) if value_head else cfg
        store = ParamStore(cfg, device=device, trainable=trainable)
        store.init_random(seed=seed)
        if master_fp32:
            store.enable_fp32_master()
        store.hf_config, store.generation_config, store.source_dir = hf_config_dict(cfg, special), {"eos_token_id": special["eos"], "pad_token_id": special["pad"]}, None
        return cfg, store, special
    model_path = resolve_model_path(model_path)
    with open(os.path.join(model_path, "config.json")) as f:
        hf = json.load(f)
    cfg = VLConfig.from_hf_dict(hf)
    cfg = dataclasses.replace(cfg, value_head=True) if value_head else cfg
    store = ParamStore(cfg, device=device, trainable=trainable)
    if master_fp32:
        store.enable_fp32_master()
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
    if value_head and "score.weight" not in norm:
        g = torch.Generator().manual_seed(seed)
        norm["score.weight"] = torch.randn(1, cfg.hidden_size, generator=g) * 0.02
        norm["score.bias"] = torch.zeros(1)
    store.load_hf_state_dict(norm)
    gen = {}
    gp = os.path.join(model_path, "generation_config.json")
    if os.path.exists(gp):
        gen = json.load(open(gp))
    eos = gen.get("eos_token_id", hf.get("eos_token_id", 151645))
    special = {"eos": eos, "pad": gen.get("pad_token_id", hf.get("pad_token_id", 151643))}
    store.hf_config, store.generation_config, store.source_dir = hf, gen or {"eos_token_id": eos, "pad_token_id": special["pad"]}, model_path
    return cfg, store, special


# tokenizer / processor / template files that travel with a checkpoint (everything but the weights)
_AUX_FILES = ("tokenizer.json", "tokenizer_config.json", "vocab.json", "merges.txt", "special_tokens_map.json", "added_tokens.json",
              "preprocessor_config.json", "video_preprocessor_config.json", "processor_config.json", "chat_template.json", "chat_template.jinja")


def save_hf(store: ParamStore, path: str, tokenizer=None, processor=None, max_shard_bytes: int = 5 << 30) -> None:
    """A directory `from_pretrained` (and load_model above) can read: sharded safetensors + index, config.json,
    generation_config.json and the tokenizer / processor files (FSDPCheckpointManager.save_checkpoint writes the same set into
    actor/huggingface, verl/utils/checkpoint/fsdp_checkpoint_manager.py:96-131)."""
    import shutil

    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    tensors = {k: v.detach().cpu().contiguous() for k, v in store.export_hf().items()}
    shards, cur, size = [], {}, 0
    for k, t in tensors.items():
        nb = t.numel() * t.element_size()
        if cur and size + nb > max_shard_bytes:
            shards.append(cur); cur, size = {}, 0
        cur[k] = t; size += nb
    shards.append(cur)
    if len(shards) == 1:
        save_file(shards[0], os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    else:
        index = {"metadata": {"total_size": sum(t.numel() * t.element_size() for t in tensors.values())}, "weight_map": {}}
        for i, sh in enumerate(shards):
            name = f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file(sh, os.path.join(path, name), metadata={"format": "pt"})
            index["weight_map"].update({k: name for k in sh})
        json.dump(index, open(os.path.join(path, "model.safetensors.index.json"), "w"), indent=2)
    cfg = getattr(store, "hf_config", None) or hf_config_dict(store.cfg, {})
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"), indent=2)
    json.dump(getattr(store, "generation_config", None) or {}, open(os.path.join(path, "generation_config.json"), "w"), indent=2)
    saved_aux = False
    for obj in (processor, tokenizer):
        if obj is not None and hasattr(obj, "save_pretrained"):
            obj.save_pretrained(path)
            saved_aux = True
    src = getattr(store, "source_dir", None)
    if not saved_aux and src:
        for name in _AUX_FILES:
            if os.path.exists(os.path.join(src, name)):
                shutil.copy2(os.path.join(src, name), os.path.join(path, name))
