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


def _real_fsdp_worker(rank, world, hf_dir, out, steps):
    """What a reference run leaves on disk, produced by REAL torch FSDP (use_orig_params=True, frozen vision tower — fsdp_workers.py:227-229,
    268-280) + AdamW on `world` gloo ranks and saved exactly as FSDPCheckpointManager.save_checkpoint does (:83-131)."""
    import functools
    import warnings
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("ST_TEST_PORT", "29581"), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch.distributed.device_mesh import init_device_mesh
    from torch.distributed.fsdp import FullyShardedDataParallel as FSDP, ShardedOptimStateDictConfig, ShardedStateDictConfig, ShardingStrategy, StateDictType
    from torch.distributed.fsdp.wrap import transformer_auto_wrap_policy
    from transformers import Qwen2_5_VLForConditionalGeneration
    model = Qwen2_5_VLForConditionalGeneration.from_pretrained(hf_dir, torch_dtype=torch.float32).train()
    model.model.visual.requires_grad_(False)
    layer_cls = {type(model.model.language_model.layers[0]), type(model.model.visual.blocks[0])}
    fs = FSDP(model, sharding_strategy=ShardingStrategy.FULL_SHARD, device_id=torch.device("cpu"), use_orig_params=True,
              auto_wrap_policy=functools.partial(transformer_auto_wrap_policy, transformer_layer_cls=layer_cls),
              device_mesh=init_device_mesh("cpu", (world,), mesh_dim_names=("fsdp",)))
    opt = torch.optim.AdamW(fs.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2)
    for step in range(steps):
        ids = torch.randint(0, 900, (2, 12), generator=torch.Generator().manual_seed(1000 * step + rank))
        loss = fs(input_ids=ids, attention_mask=torch.ones_like(ids)).logits.float().pow(2).mean()
        opt.zero_grad(); loss.backward(); opt.step()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with FSDP.state_dict_type(fs, StateDictType.SHARDED_STATE_DICT, ShardedStateDictConfig(offload_to_cpu=True), ShardedOptimStateDictConfig(offload_to_cpu=True)):
            msd, osd = fs.state_dict(), opt.state_dict()
    os.makedirs(out, exist_ok=True)
    torch.save(msd, os.path.join(out, f"model_world_size_{world}_rank_{rank}.pt"))
    torch.save(osd, os.path.join(out, f"optim_world_size_{world}_rank_{rank}.pt"))
    torch.save({"lr_scheduler": {"last_epoch": steps, "_step_count": steps + 1, "base_lrs": [1e-3], "_last_lr": [1e-3], "lr_lambdas": [None]},
                "rng": {"cpu": torch.get_rng_state()}}, os.path.join(out, f"extra_state_world_size_{world}_rank_{rank}.pt"))
    dist.barrier(); dist.destroy_process_group()


def test_adamw_moments_of_a_real_fsdp_run_are_restored_per_parameter(tmp_path, capsys):
    """the reference's optim_world_size_W_rank_r.pt with use_orig_params=True (what freeze_vision_tower — every shipped
    script — switches on) holds per-PARAMETER 1-D pieces keyed by the parameter's index: the loader concatenates them in rank order and
    names them through the model file's key order.  Ground truth: the same two AdamW steps on ONE process with the two ranks' losses
    averaged (FSDP averages the gradients)."""
    import torch.multiprocessing as tmp
    from transformers import Qwen2_5_VLForConditionalGeneration
    from verl.utils.checkpoint import load_reference_checkpoint
    world, steps = 2, 2
    cfg, store, params = _store()
    hf_dir, out = str(tmp_path / "hf"), str(tmp_path / "global_step_2" / "actor")
    save_hf(store, hf_dir)
    tmp.spawn(_real_fsdp_worker, args=(world, hf_dir, out, steps), nprocs=world)
    # single-process ground truth
    model = Qwen2_5_VLForConditionalGeneration.from_pretrained(hf_dir, torch_dtype=torch.float32).train()
    model.model.visual.requires_grad_(False)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2)
    for step in range(steps):
        loss = 0.0
        for rank in range(world):
            ids = torch.randint(0, 900, (2, 12), generator=torch.Generator().manual_seed(1000 * step + rank))
            loss = loss + model(input_ids=ids, attention_mask=torch.ones_like(ids)).logits.float().pow(2).mean() / world
        opt.zero_grad(); loss.backward(); opt.step()
    tgt = mdl.ParamStore(cfg, device="cpu", trainable=False)
    tgt.trainable = True
    tgt.m, tgt.v, tgt.c = torch.zeros(tgt.numel, dtype=torch.float32), torch.zeros(tgt.numel, dtype=torch.float32), None
    tgt.refresh_transposes = lambda: None
    info = load_reference_checkpoint(tgt, out)
    assert info == {"world_size": world, "opt_steps": steps, "sched_steps": steps, "optimizer": "loaded-per-parameter"}
    assert "restored AdamW state" in capsys.readouterr().out
    got_m = tgt.export_hf({n: tgt._view(tgt.m, n) for n in tgt.layout})
    got_v = tgt.export_hf({n: tgt._view(tgt.v, n) for n in tgt.layout})
    got_w = tgt.export_hf()
    n_state = 0
    for name, p in model.named_parameters():
        st = opt.state.get(p)
        if not st:
            assert float(got_m[name].abs().max()) == 0.0, name             # frozen tower: no state, zeros
            continue
        n_state += 1
        scale = float(st["exp_avg"].abs().max()) + 1e-30
        assert float((got_m[name].float() - st["exp_avg"]).abs().max()) <= 2e-4 * scale + 1e-12, name
        assert float((got_v[name].float() - st["exp_avg_sq"]).abs().max()) <= 4e-4 * float(st["exp_avg_sq"].abs().max()) + 1e-20, name
        assert float((got_w[name].float() - p.detach().bfloat16().float()).abs().max()) <= 2.0 ** -7 * float(p.detach().abs().max()), name   # bf16 store
    assert n_state >= 20
