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


# ------------------------------------------------------------------ the reference's sharded layout
def _export_worker(args):
    src_sd, optim, out, world = args
    from verl.utils.checkpoint import export_reference_layout
    export_reference_layout(src_sd, optim, out, world, opt_steps=7, sched_steps=3)


@pytest.mark.parametrize("world", [1, 4])
def test_reference_sharded_checkpoint_roundtrip(tmp_path, world):
    """model_/optim_/extra_state_world_size_W_rank_r.pt as FSDPCheckpointManager writes them (DTensor Shard(0) on an ("fsdp",) mesh,
    verl/utils/checkpoint/fsdp_checkpoint_manager.py:83-131), with the transformers-4.49 parameter names the reference's runs carry:
    the loader reassembles weights, AdamW moments, Kahan compensation, optimizer step and scheduler position bit for bit, without
    any process group.  (The files are produced by export_reference_layout in a child process: building DTensors needs one.)"""
    import multiprocessing as mp
    from verl.utils.checkpoint import find_reference_world_size, load_reference_checkpoint, read_reference_shards
    cfg, store, params = _store()
    hf = {k: torch.from_numpy(v).bfloat16() for k, v in params.items()}
    # the reference's checkpoints use the old tower names
    old = {}
    for k, v in hf.items():
        if k.startswith("model.visual."):
            old[k[len("model."):]] = v
        elif k.startswith("model.language_model."):
            old["model." + k[len("model.language_model."):]] = v
        else:
            old[k] = v
    g = torch.Generator().manual_seed(0)
    optim = {k: {"step": torch.tensor(7.0), "exp_avg": (torch.randn(v.shape, generator=g) * 1e-3).bfloat16(),
                 "exp_avg_sq": (torch.rand(v.shape, generator=g) * 1e-6).bfloat16(), "compensation": (torch.randn(v.shape, generator=g) * 1e-5).bfloat16()}
             for k, v in old.items()}
    out = str(tmp_path / "actor")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_export_worker, args=((old, optim, out, world),))
    p.start(); p.join()
    assert p.exitcode == 0
    assert find_reference_world_size(out) == world
    assert sorted(os.listdir(out)) == sorted(f"{k}_world_size_{world}_rank_{r}.pt" for k in ("model", "optim", "extra_state") for r in range(world))
    merged = read_reference_shards(out, "model")
    assert set(merged) == set(old) and all(torch.equal(merged[k], old[k]) for k in old)
    if world > 1:
        r0 = torch.load(os.path.join(out, f"model_world_size_{world}_rank_0.pt"), weights_only=False)
        k0 = "model.embed_tokens.weight"
        assert type(r0[k0]).__name__ == "DTensor" and r0[k0]._local_tensor.shape[0] < old[k0].shape[0]      # really sharded
    tgt = mdl.ParamStore(cfg, device="cpu", trainable=False)
    tgt.trainable = True
    tgt.m, tgt.v, tgt.c = (torch.zeros(tgt.numel, dtype=torch.bfloat16) for _ in range(3))
    tgt.refresh_transposes = lambda: None
    info = load_reference_checkpoint(tgt, out)
    assert info == {"world_size": world, "opt_steps": 7, "sched_steps": 3}
    assert torch.equal(tgt.flat, store.flat)
    back = tgt.export_hf({n: tgt._view(tgt.m, n) for n in tgt.layout})
    for k, v in hf.items():
        ko = k[len("model."):] if k.startswith("model.visual.") else ("model." + k[len("model.language_model."):] if k.startswith("model.language_model.") else k)
        assert torch.equal(back[k], optim[ko]["exp_avg"].reshape(back[k].shape)), k
    back_c = tgt.export_hf({n: tgt._view(tgt.c, n) for n in tgt.layout})
    assert torch.equal(back_c["lm_head.weight"], optim["lm_head.weight"]["compensation"])
