"""Checkpoint directories written by spatialthinker_amd.pretrained.save_hf: loadable by transformers'
from_pretrained AND by this repo's load_model, weights bit-identical; hub-id model paths resolve through the hub cache."""
import json

import numpy as np
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
    export_reference_layout(src_sd, optim, out, world, opt_steps=7, sched_steps=3, write_optim=True)


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
    assert info == {"world_size": world, "opt_steps": 7, "sched_steps": 3, "optimizer": "loaded"}
    extra = torch.load(os.path.join(out, f"extra_state_world_size_{world}_rank_0.pt"), weights_only=False)
    assert "rng" not in extra and extra["lr_scheduler"]["last_epoch"] == 3        # no empty rng dict for the reference to choke on
    assert torch.equal(tgt.flat, store.flat)
    back = tgt.export_hf({n: tgt._view(tgt.m, n) for n in tgt.layout})
    for k, v in hf.items():
        ko = k[len("model."):] if k.startswith("model.visual.") else ("model." + k[len("model.language_model."):] if k.startswith("model.language_model.") else k)
        assert torch.equal(back[k], optim[ko]["exp_avg"].reshape(back[k].shape)), k
    back_c = tgt.export_hf({n: tgt._view(tgt.c, n) for n in tgt.layout})
    assert torch.equal(back_c["lm_head.weight"], optim["lm_head.weight"]["compensation"])


def _export_plain_worker(args):
    hf5_sd, out, world = args
    from verl.utils.checkpoint import export_reference_layout
    export_reference_layout(hf5_sd, None, out, world, sched_steps=5)


def test_checkpoint_written_by_a_reference_run_resumes_without_its_optimizer_state(tmp_path, capsys):
    """What FSDPCheckpointManager.save_checkpoint really leaves on disk (fsdp_checkpoint_manager.py:83-131): DTensor model shards under the
    4.49 names, the rank's RAW optimizer.state_dict() — integer keys, one 1-D flat-parameter shard per FSDP unit — an extra_state with
    a filled rng dict, and next to it a torchdata StatefulDataLoader snapshot as dataloader.pt.  The loader restores weights and the
    scheduler position, says that the optimizer state starts from zero, and crashes on none of it."""
    import multiprocessing as mp
    import random
    from verl.utils.checkpoint import load_reference_checkpoint, read_reference_shards
    from verl.utils.dataloader import ResumableDataLoader
    world = 2
    cfg, store, params = _store()
    hf5 = {k: torch.from_numpy(v).bfloat16() for k, v in params.items()}
    out = str(tmp_path / "global_step_3" / "actor")
    p = mp.get_context("spawn").Process(target=_export_plain_worker, args=((hf5, out, world),))
    p.start(); p.join()
    assert p.exitcode == 0
    merged = read_reference_shards(out, "model")
    assert "visual.patch_embed.proj.weight" in merged and "model.embed_tokens.weight" in merged       # exported under the 4.49 names
    assert not any(k.startswith(("model.visual.", "model.language_model.")) for k in merged)
    assert not any(f.startswith("optim_") for f in os.listdir(out))                                   # no optimizer files by default
    # --- now make the directory look like one a reference run wrote
    g = torch.Generator().manual_seed(1)
    n_flat = [9_000, 4_096, 4_096, 1_234]                       # FSDP units: root + three wrapped layers; rank-local shard lengths
    for r in range(world):
        state = {i: {"step": torch.tensor(11.0), "exp_avg": torch.randn(n, generator=g).bfloat16(), "exp_avg_sq": torch.rand(n, generator=g).bfloat16(),
                     "compensation": torch.zeros(n, dtype=torch.bfloat16)} for i, n in enumerate(n_flat)}
        groups = [{"lr": 1e-6, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 1e-2, "use_kahan_summation": True,
                   "momentum_dtype": torch.bfloat16, "variance_dtype": torch.bfloat16, "compensation_buffer_dtype": torch.bfloat16,
                   "initial_lr": 1e-6, "params": list(range(len(n_flat)))}]
        torch.save({"state": state, "param_groups": groups}, os.path.join(out, f"optim_world_size_{world}_rank_{r}.pt"))
        sched = {"base_lrs": [1e-6], "last_epoch": 5, "verbose": False, "_step_count": 6, "_get_lr_called_within_step": False, "_last_lr": [1e-6],
                 "lr_lambdas": [None]}
        rng = {"cpu": torch.get_rng_state(), "cuda": torch.zeros(16, dtype=torch.uint8), "numpy": np.random.get_state(), "random": random.getstate()}
        torch.save({"lr_scheduler": sched, "rng": rng}, os.path.join(out, f"extra_state_world_size_{world}_rank_{r}.pt"))
    tgt = mdl.ParamStore(cfg, device="cpu", trainable=False)
    tgt.trainable = True
    tgt.m, tgt.v, tgt.c = (torch.ones(tgt.numel, dtype=torch.bfloat16) for _ in range(3))
    tgt.refresh_transposes = lambda: None
    info = load_reference_checkpoint(tgt, out)
    assert info == {"world_size": world, "opt_steps": 0, "sched_steps": 5, "optimizer": "reset"}
    assert torch.equal(tgt.flat, store.flat)
    assert float(tgt.m.abs().max()) == 0 and float(tgt.v.abs().max()) == 0 and float(tgt.c.abs().max()) == 0
    assert "start from zero" in capsys.readouterr().out
    # --- the trainer-level resume with a foreign dataloader.pt
    with pytest.raises(ValueError):
        ResumableDataLoader.load_state_dict(ResumableDataLoader.__new__(ResumableDataLoader), {"_snapshot": {"_main_snapshot": {}}, "_steps_since_snapshot": 0})


def test_fp32_master_enabled_before_the_load_keeps_the_checkpoints_own_precision(tmp_path):
    """worker.actor.fsdp.torch_dtype=fp32 with an fp32 checkpoint: the master must start from the checkpoint's fp32 values (the reference
    keeps them), not from their bf16 rounding — load_model(master_fp32=True) enables the master BEFORE loading."""
    from safetensors.torch import save_file
    cfg, store, params = _store()
    d = tmp_path / "fp32_ckpt"
    save_hf(store, str(d))
    rs = np.random.RandomState(3)
    off_grid = {k: torch.from_numpy((v * (1.0 + 1e-3 * rs.standard_normal(v.shape))).astype(np.float32)) for k, v in params.items()}
    for f in os.listdir(d):
        if f.endswith(".safetensors"):
            os.remove(d / f)
    save_file(off_grid, str(d / "model.safetensors"))
    cfg2, st2, _ = load_model(str(d), trainable=True, device="cpu", master_fp32=True)
    assert st2.master is not None and st2.c is None and st2.m.dtype == torch.float32
    back = st2.export_hf({n: st2._view(st2.master, n) for n in st2.layout})
    k = "model.language_model.layers.0.mlp.down_proj.weight"
    assert torch.equal(back[k], off_grid[k])                                   # the fp32 values themselves, not their bf16 rounding
    assert not torch.equal(back[k], off_grid[k].bfloat16().float())
    assert torch.equal(st2.flat, st2.master.bfloat16())                        # the working copy is the rounding of the master
