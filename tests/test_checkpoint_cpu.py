"""Checkpoint directories written by spatialthinker_amd.pretrained.save_hf: loadable by transformers'
from_pretrained AND by this repo's load_model, weights bit-identical; hub-id model paths resolve through the hub cache."""
import json
import os

import pytest
import torch

import tiny
from spatialthinker_amd import model as mdl
from spatialthinker_amd.pretrained import hf_config_dict, load_model, resolve_model_path, save_hf


def _store():
    cfg = mdl.VLConfig(**tiny.TINY)
    store = mdl.ParamStore(cfg, device="cpu", trainable=False)
    params = tiny.make_params()
    store.load_hf_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    store.hf_config = hf_config_dict(cfg, {"eos": tiny.EOS_ID, "pad": tiny.PAD_ID})
    store.generation_config = {"eos_token_id": [tiny.EOS_ID, 1015], "pad_token_id": tiny.PAD_ID}
    return cfg, store, params


@pytest.mark.parametrize("shard_bytes", [400_000, 1 << 40])
def test_saved_directory_loads_with_transformers_and_with_load_model(tmp_path, shard_bytes):
    cfg, store, params = _store()
    save_hf(store, str(tmp_path), max_shard_bytes=shard_bytes)
    files = sorted(os.listdir(tmp_path))
    assert "config.json" in files and "generation_config.json" in files
    assert ("model.safetensors" in files) == (shard_bytes > 1 << 30) and ("model.safetensors.index.json" in files) == (shard_bytes < 1 << 30)
    from transformers import Qwen2_5_VLForConditionalGeneration
    hf = Qwen2_5_VLForConditionalGeneration.from_pretrained(str(tmp_path), torch_dtype=torch.float32)
    sd = hf.state_dict()
    assert set(sd) == set(params)
    for k, v in params.items():
        assert torch.equal(sd[k].float(), torch.from_numpy(v)), k
    cfg2, store2, special = load_model(str(tmp_path), trainable=False, device="cpu")
    assert cfg2 == cfg and torch.equal(store2.flat, store.flat)
    assert special == {"eos": [tiny.EOS_ID, 1015], "pad": tiny.PAD_ID}                       # the EOS list of generation_config.json survives
    assert json.load(open(tmp_path / "config.json"))["model_type"] == "qwen2_5_vl"


def test_hub_id_resolves_through_the_hub_cache(tmp_path, monkeypatch):
    """What every shipped script passes (MODEL_PATH=Qwen/Qwen2.5-VL-7B-Instruct): a hub id is resolved with snapshot_download; here the
    'hub' is a pre-populated local cache in offline mode, and an unknown id fails with a clear error instead of open(config.json)."""
    cfg, store, _ = _store()
    repo = tmp_path / "hub" / "models--acme--tiny-vl"
    snap = repo / "snapshots" / "abc123"
    save_hf(store, str(snap))
    os.makedirs(repo / "refs", exist_ok=True)
    (repo / "refs" / "main").write_text("abc123")
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    monkeypatch.setenv("HF_HUB_CACHE", str(tmp_path / "hub"))
    import huggingface_hub.constants as C
    monkeypatch.setattr(C, "HF_HUB_CACHE", str(tmp_path / "hub"))
    monkeypatch.setattr(C, "HF_HUB_OFFLINE", True)
    path = resolve_model_path("acme/tiny-vl")
    assert os.path.samefile(path, snap)
    cfg2, store2, _ = load_model("acme/tiny-vl", trainable=False, device="cpu")
    assert cfg2 == cfg and torch.equal(store2.flat, store.flat)
    with pytest.raises(FileNotFoundError):
        resolve_model_path("acme/does-not-exist")
    assert resolve_model_path(str(snap)) == str(snap)
