"""Interop with the reference's on-disk checkpoint layout (verl/utils/checkpoint/fsdp_checkpoint_manager.py:52-131):

    <ckpt>/actor/model_world_size_{W}_rank_{r}.pt         FSDP SHARDED_STATE_DICT: {hf_param_name: DTensor Shard(0) on mesh ("fsdp",)}
    <ckpt>/actor/optim_world_size_{W}_rank_{r}.pt         the rank's RAW `optimizer.state_dict()` (:95-96 — FSDP.optim_state_dict is never
                                                           called): {"state": {0: {step, exp_avg, exp_avg_sq, compensation}, 1: ...},
                                                           "param_groups": [{"params": [0, 1, ...], ...}]} — INTEGER keys, one entry per
                                                           FSDP FlatParameter, each tensor the rank-local 1-D shard of that flat parameter
    <ckpt>/actor/extra_state_world_size_{W}_rank_{r}.pt   {"lr_scheduler": LambdaLR.state_dict(), "rng": {cpu, cuda, numpy, random}}
    <ckpt>/actor/huggingface/                             config + generation config + tokenizer / processor files

This engine keeps full replicas, so its native checkpoint is ONE HF-loadable directory + ONE optimizer file
(verl/workers/fsdp_workers.py save_checkpoint).  To let a run checkpointed by the reference continue here (and vice versa):
  * load_reference_checkpoint  — reads all W model shard files (unpickling DTensors needs no process group), concatenates the local
    shards along their placement dimension, maps transformers-4.49 names to the 5.x names, fills the ParamStore's weights and the
    scheduler position.  OPTIMIZER STATE: integer-keyed raw `optimizer.state_dict()` files.  With FSDP use_orig_params=True (set by
    `freeze_vision_tower`, i.e. by every shipped script) an entry is one ORIGINAL parameter's rank-local 1-D piece: those are
    concatenated in rank order, mapped to names through the model file's key order and restored (_restore_per_parameter_state; pinned
    against real FSDP + AdamW on two gloo ranks in tests/test_checkpoint_cpu.py).  With use_orig_params=False an entry is the shard of a
    FlatParameter of a whole wrapped unit; mapping it back needs the wrap policy and flattening order of the run that wrote it, which
    this loader does not reconstruct — such files are recognised, reported, and the AdamW moments / Kahan buffers / step counter start
    from zero (it never crashes and never mixes up tensors).  Name-keyed optimizer files (what
    export_reference_layout(write_optim=True) writes: this build's own round-trip format) load bit for bit;
  * export_reference_layout    — writes this engine's weights as W rank files of DTensor shards under the transformers-4.49 names
    (`visual.*`, `model.*`: what FSDPCheckpointManager.load_checkpoint :52-81 and scripts/model_merger.py :37-164 expect), plus the
    scheduler position.  MERGER-ONLY: the export is an input for scripts/model_merger.py (which reads the model shards alone), NOT a
    resume point for a reference run — FSDPCheckpointManager.load_checkpoint (:57-62) `torch.load`s optim_world_size_W_rank_r.pt
    unconditionally, so without optimizer files it stops with FileNotFoundError, and the name-keyed files of write_optim=True are
    rejected by its Optimizer.load_state_dict (it accepts only its own integer-keyed per-rank layout, whose piece boundaries follow
    the FSDP wrap policy of the run that loads them).  No "rng" entry is written.  DTensor construction needs a process group of W
    ranks: a stand-alone process builds them on torch's in-process "fake" backend, one rank at a time
    (tools/export_reference_checkpoint.py).
The CUDA RNG state of the reference ("rng") has no counterpart: this engine's sampler is counter-based (seed, row, step)."""
from __future__ import annotations

import os
import re
from typing import Any, Dict, Optional, Tuple

import torch

from .checkpoint_manager import BaseCheckpointManager

_PAT = re.compile(r"model_world_size_(\d+)_rank_0\.pt$")


def find_reference_world_size(path: str) -> Optional[int]:
    for name in sorted(os.listdir(path)) if os.path.isdir(path) else []:
        m = _PAT.match(name)
        if m:
            return int(m.group(1))
    return None


def _local(t):
    """(local tensor, shard dim or None) of one rank's entry: DTensor (FSDP with a device mesh), ShardedTensor (FSDP without) or a
    plain tensor / scalar (replicated)."""
    try:
        from torch.distributed.tensor import DTensor
    except Exception:                                           # pragma: no cover
        DTensor = ()
    if DTensor and isinstance(t, DTensor):
        pl = t.placements[-1]
        return t._local_tensor, (pl.dim if pl.is_shard() else None)
    if hasattr(t, "local_shards"):                              # torch.distributed._shard.sharded_tensor.ShardedTensor
        sh = t.local_shards()
        return (sh[0].tensor if sh else None), 0
    return t, None


def _merge(per_rank: list):
    """One entry of every rank's dict -> the full value."""
    first = per_rank[0]
    if isinstance(first, dict):
        return {k: _merge([d[k] for d in per_rank]) for k in first}
    if not torch.is_tensor(first):
        return first
    locs = [_local(t) for t in per_rank]
    dim = locs[0][1]
    if dim is None:
        return locs[0][0]
    parts = [x for x, _ in locs if x is not None and x.numel() > 0]
    return torch.cat(parts, dim=dim).contiguous() if parts else locs[0][0]


def read_reference_shards(path: str, kind: str = "model", world_size: Optional[int] = None) -> Dict[str, Any]:
    """Merge `{kind}_world_size_W_rank_r.pt` for r = 0..W-1 into one dict of full tensors (kind: model | optim | extra_state)."""
    W = world_size or find_reference_world_size(path)
    if not W:
        raise FileNotFoundError(f"no model_world_size_*_rank_0.pt under {path}")
    per_rank = [torch.load(os.path.join(path, f"{kind}_world_size_{W}_rank_{r}.pt"), map_location="cpu", weights_only=False) for r in range(W)]
    if kind == "extra_state":
        return per_rank[0]
    return _merge(per_rank)


def normalise_hf_names(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """transformers < 4.52 (the reference pins >= 4.49) names the towers `visual.*` / `model.*`; 5.x (and ParamStore) use
    `model.visual.*` / `model.language_model.*`.  FSDP wrapper prefixes are dropped."""
    out = {}
    for k, v in sd.items():
        k = k.replace("_fsdp_wrapped_module.", "").replace("_checkpoint_wrapped_module.", "")
        if k.startswith("visual."):
            k = "model." + k
        elif k.startswith("model.") and not k.startswith(("model.visual.", "model.language_model.")):
            k = "model.language_model." + k[len("model."):]
        out[k] = v
    return out


def denormalise_hf_names(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The inverse of normalise_hf_names: 5.x / ParamStore names -> the transformers-4.49 names a reference run loads."""
    out = {}
    for k, v in sd.items():
        if k.startswith("model.visual."):
            k = k[len("model."):]
        elif k.startswith("model.language_model."):
            k = "model." + k[len("model.language_model."):]
        out[k] = v
    return out


def _restore_per_parameter_state(store, path: str, W: int) -> Optional[int]:
    """The reference's optimizer files when FSDP ran with use_orig_params=True (what `freeze_vision_tower` switches on,
    verl/workers/fsdp_workers.py:227-229, and what every shipped script sets): the optimizer then holds the ORIGINAL parameters, so the
    raw `optimizer.state_dict()` a rank saves (fsdp_checkpoint_manager.py:95-96) is {"state": {i: {step, exp_avg, exp_avg_sq
    [, compensation]}}} with i = the parameter's position in `module.parameters()` and every tensor the rank's 1-D piece of that
    parameter's flattened values (possibly empty, then the entry may be missing).  Pieces concatenated in rank order are the flattened
    parameter; the parameter order is the key order of the model shard file (a state dict lists parameters in `parameters()` order;
    Qwen2.5-VL has no persistent buffers; a tied lm_head appears once in `parameters()`).  Returns the step count, or None when the
    files are not of this form (flat-parameter shards of use_orig_params=False: sizes do not add up to parameter sizes)."""
    model0 = torch.load(os.path.join(path, f"model_world_size_{W}_rank_0.pt"), map_location="cpu", weights_only=False)
    names = [k.replace("_fsdp_wrapped_module.", "").replace("_checkpoint_wrapped_module.", "") for k in model0.keys()]
    numel = []
    for t in model0.values():
        loc, dim = _local(t)
        numel.append(int(torch.Size(t.shape).numel()) if hasattr(t, "shape") else int(loc.numel()))
    shapes = [tuple(t.shape) for t in model0.values()]
    per_rank = [torch.load(os.path.join(path, f"optim_world_size_{W}_rank_{r}.pt"), map_location="cpu", weights_only=False) for r in range(W)]
    n_params = sum(len(g["params"]) for g in per_rank[0]["param_groups"])
    if n_params == len(names) - 1 and "lm_head.weight" in names:      # tied head: one parameter, two state-dict names
        k = names.index("lm_head.weight")
        names.pop(k); numel.pop(k); shapes.pop(k)
    if n_params != len(names):
        return None
    ids = sorted({i for d in per_rank for i in d["state"].keys()})
    full: Dict[str, Dict[str, torch.Tensor]] = {}
    steps = []
    for i in ids:
        if not isinstance(i, int) or not 0 <= i < len(names):
            return None
        entries = [d["state"].get(i) for d in per_rank]
        keys = sorted({k for e in entries if e for k in e.keys()})
        st = {}
        for k in keys:
            vals = [e[k] for e in entries if e is not None and k in e]
            if k == "step" or not torch.is_tensor(vals[0]) or vals[0].dim() == 0:
                steps.append(int(float(vals[0])))
                continue
            flat = torch.cat([v.reshape(-1) for v in vals])
            if flat.numel() != numel[i]:
                return None                                               # flat-parameter shards (padding, several parameters per entry)
            st[k] = flat.reshape(shapes[i])
        full[names[i]] = st
    state = normalise_hf_names(full)
    views = store.export_hf()
    for buf, key in ((store.m, "exp_avg"), (store.v, "exp_avg_sq"), (store.c, "compensation")):
        if buf is None:
            continue
        have = {n: st[key] for n, st in state.items() if key in st}
        store.load_hf_state_dict({n: have[n] if n in have else torch.zeros(tuple(t.shape)) for n, t in views.items()}, target=buf)
    print(f"[checkpoint] restored AdamW state of {len(full)} parameters from the reference's per-parameter optimizer shards ({W} ranks, "
          f"step {max(steps) if steps else 0})")
    return max(steps) if steps else 0


def load_reference_checkpoint(store, path: str, engine=None) -> Dict[str, Any]:
    """Fill `store` (ParamStore) — and, when given, the PolicyEngine's optimizer / scheduler counters — from a checkpoint directory in
    the reference's layout.  Returns {"world_size", "opt_steps", "sched_steps", "optimizer"}; "optimizer" is "loaded", "reset" (files
    in the reference's own flat-shard layout, or unreadable: state starts from zero, with a printed notice) or "absent"."""
    W = find_reference_world_size(path)
    if not W:
        raise FileNotFoundError(f"{path} holds no reference-layout checkpoint (model_world_size_W_rank_r.pt)")
    store.load_hf_state_dict(normalise_hf_names(read_reference_shards(path, "model", W)))
    info = {"world_size": W, "opt_steps": 0, "sched_steps": 0, "optimizer": "absent"}
    opt_file = os.path.join(path, f"optim_world_size_{W}_rank_0.pt")
    if store.trainable and os.path.exists(opt_file):
        try:
            probe = torch.load(opt_file, map_location="cpu", weights_only=False)
            keys = list((probe.get("state") or {}).keys()) if isinstance(probe, dict) else None
        except Exception as e:                                        # a file this torch cannot unpickle must not end the run
            keys, probe = None, None
            print(f"[checkpoint] cannot read {opt_file} ({type(e).__name__}: {e}); the optimizer state starts from zero")
        if keys is None or any(not isinstance(k, str) for k in keys):
            restored = None
            if keys is not None:
                try:
                    restored = _restore_per_parameter_state(store, path, W)
                except Exception as e:                                    # a layout this reader does not know must not end the run
                    print(f"[checkpoint] {opt_file}: per-parameter restore failed ({type(e).__name__}: {e})")
            if restored is not None:
                info["optimizer"], info["opt_steps"] = "loaded-per-parameter", restored
            else:
                if keys is not None:
                    print(f"[checkpoint] {opt_file} holds the reference's raw per-rank optimizer.state_dict() ({len(keys)} flat-parameter shards keyed "
                          f"by integer, FSDP use_orig_params=False); they are not unflattened here — weights and scheduler position are restored, "
                          f"AdamW moments / Kahan buffers / step counter start from zero")
                info["optimizer"] = "reset"
                for buf in (store.m, store.v, store.c):
                    if buf is not None:
                        buf.zero_()
        else:
            info["optimizer"] = "loaded"
    if info["optimizer"] == "loaded":
        opt = read_reference_shards(path, "optim", W)
        state = normalise_hf_names(opt.get("state", {}))
        shapes = store.export_hf()                                   # HF name -> view with the parameter's shape
        for buf, key in ((store.m, "exp_avg"), (store.v, "exp_avg_sq"), (store.c, "compensation")):
            have = {n: st[key] for n, st in state.items() if key in st}
            if have and buf is not None:                             # parameters without state (frozen tower) keep zeros; no Kahan buffer in fp32-master mode
                full = {n: have[n] if n in have else torch.zeros(tuple(t.shape)) for n, t in shapes.items()}
                store.load_hf_state_dict(full, target=buf)
        steps = [int(float(st["step"])) for st in state.values() if "step" in st]
        info["opt_steps"] = max(steps) if steps else 0
    extra_file = os.path.join(path, f"extra_state_world_size_{W}_rank_0.pt")
    if os.path.exists(extra_file):
        extra = read_reference_shards(path, "extra_state", W)
        sched = (extra.get("lr_scheduler") if isinstance(extra, dict) else None) or {}
        info["sched_steps"] = int(sched.get("last_epoch", 0))
    if engine is not None:
        engine.opt_steps, engine.sched_steps = info["opt_steps"], info["sched_steps"]
        store.version = getattr(store, "version", 0) + 1
    return info


def export_reference_layout(hf_state: Dict[str, torch.Tensor], optim_state: Optional[Dict[str, Dict[str, torch.Tensor]]], out_dir: str,
                            world_size: int, opt_steps: int = 0, sched_steps: int = 0, base_lr: float = 1e-6, hyper: Optional[dict] = None,
                            write_optim: bool = False, names: str = "hf4") -> None:
    """Write full tensors as the reference's per-rank DTensor shards.  Must run in a process WITHOUT an initialised process group
    (it brings up torch's "fake" backend once per rank to build the device mesh).
    names: "hf4" (default) renames to the transformers-4.49 parameter names a reference run expects; "asis" keeps the given names.
    write_optim: also write NAME-keyed optimizer files — this build's own round-trip format (load_reference_checkpoint reads them back
    bit for bit).  The reference can load NEITHER form as a resume point (module docstring: merger-only) — to continue in the
    reference, merge the shards with scripts/model_merger.py and start it from the resulting HF directory."""
    if names == "hf4":
        hf_state = denormalise_hf_names(hf_state)
        if optim_state is not None:
            optim_state = denormalise_hf_names(optim_state)
    import torch.distributed as dist
    from torch.distributed.device_mesh import DeviceMesh
    from torch.distributed.tensor import DTensor, Shard
    from torch.testing._internal.distributed.fake_pg import FakeStore
    assert not dist.is_initialized(), "export_reference_layout needs a process without a process group (see tools/export_reference_checkpoint.py)"
    os.makedirs(out_dir, exist_ok=True)
    W = int(world_size)

    def shard(full: torch.Tensor, mesh, r: int):
        if full.dim() == 0:
            return full.clone()
        chunks = list(torch.chunk(full, W, dim=0))
        local = chunks[r].clone() if r < len(chunks) else full[:0].clone()
        return DTensor.from_local(local, mesh, [Shard(0)], run_check=False, shape=full.shape, stride=full.stride())

    for r in range(W):
        dist.init_process_group(backend="fake", store=FakeStore(), rank=r, world_size=W)
        try:
            mesh = DeviceMesh("cpu", torch.arange(W), mesh_dim_names=("fsdp",))
            torch.save({k: shard(v.detach().cpu().contiguous(), mesh, r) for k, v in hf_state.items()},
                       os.path.join(out_dir, f"model_world_size_{W}_rank_{r}.pt"))
            if optim_state is not None and write_optim:
                st = {n: {k: (shard(t.detach().cpu().contiguous(), mesh, r) if torch.is_tensor(t) and t.dim() > 0 else torch.tensor(float(opt_steps)))
                          for k, t in d.items()} for n, d in optim_state.items()}
                groups = [dict({"lr": base_lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 1e-2}, **(hyper or {}), params=list(optim_state.keys()))]
                torch.save({"state": st, "param_groups": groups}, os.path.join(out_dir, f"optim_world_size_{W}_rank_{r}.pt"))
            sched = {"last_epoch": int(sched_steps), "_step_count": int(sched_steps) + 1, "base_lrs": [base_lr], "_last_lr": [base_lr],
                     "lr_lambdas": [None]}
            # no "rng" key: FSDPCheckpointManager restores it only when present (:80-81), and an empty dict there raises KeyError('cpu')
            torch.save({"lr_scheduler": sched}, os.path.join(out_dir, f"extra_state_world_size_{W}_rank_{r}.pt"))
        finally:
            dist.destroy_process_group()


class FSDPCheckpointManager(BaseCheckpointManager):
    """The reference's class name (fsdp_checkpoint_manager.py:34-131: `FSDPCheckpointManager(model, optimizer, lr_scheduler,
    processing_class)`, `.save_checkpoint(path)`, `.load_checkpoint(path)`) over this build's engine: `model` is a PolicyEngine / CriticEngine
    (weights, AdamW moments, Kahan buffers and step counters live in its ParamStore; `optimizer` / `lr_scheduler` are unused),
    `processing_class` the tokenizer or processor whose files go next to the weights.

    save: replicas are identical, so rank 0 writes ONE HF-loadable directory `path/huggingface` (weights + config + generation config +
    tokenizer / processor files, what :96-131 puts there) and ONE `optim_world_size_1_rank_0.pt` (moments, counters, `extra`); the
    reference writes a shard per rank (:83-95).  load: that layout, or — recognised by its file names — a checkpoint written by the
    REFERENCE (per-rank DTensor shards; load_reference_checkpoint above).  Returns the `extra` dict that was saved (the worker keeps its
    rollout seed position there)."""

    def __init__(self, model, optimizer=None, lr_scheduler=None, processing_class=None, tokenizer=None):
        super().__init__(model, optimizer, lr_scheduler, processing_class)
        if not (hasattr(model, "store") and hasattr(model, "opt_steps")):
            raise TypeError("model must be a spatialthinker_amd.actor.PolicyEngine / CriticEngine (FSDPWorker.actor); this build has no FSDP nn.Module")
        self.tokenizer = tokenizer
        self.last_load_info: Optional[Dict[str, Any]] = None

    def save_checkpoint(self, path: str, extra: Optional[Dict[str, Any]] = None) -> None:
        from spatialthinker_amd.pretrained import save_hf
        if self.rank == 0:
            eng, st = self.model, self.model.store
            tok, proc = self.tokenizer, self.processing_class
            if tok is None and proc is not None and not hasattr(proc, "image_processor"):
                tok, proc = proc, None                      # the reference passes ONE processing_class: a bare tokenizer for text-only models
            save_hf(st, os.path.join(path, "huggingface"), tokenizer=tok, processor=proc)
            opt = {"m": st.m.cpu(), "v": st.v.cpu(), "opt_steps": eng.opt_steps, "sched_steps": eng.sched_steps, "strategy": eng.h.optim_strategy}
            opt.update(extra or {})
            if st.c is not None and eng.h.optim_strategy == "adamw_bf16":
                opt["c"] = st.c.cpu()
            if st.master is not None:
                opt["master"] = st.master.cpu()
            torch.save(opt, os.path.join(path, "optim_world_size_1_rank_0.pt"))
        if self.world_size > 1:
            torch.distributed.barrier()

    def load_checkpoint(self, path: Optional[str] = None) -> Dict[str, Any]:
        if path is None:
            return {}
        eng, st = self.model, self.model.store
        if find_reference_world_size(path):
            # every rank reassembles the full weights / optimizer state from all W shard files (whatever this run's world size is)
            self.last_load_info = load_reference_checkpoint(st, path, engine=eng)
            if self.world_size > 1:
                torch.distributed.barrier()
            return {}
        import glob
        from safetensors.torch import load_file
        sd = {}
        for shard in sorted(glob.glob(os.path.join(path, "huggingface", "*.safetensors"))):
            sd.update(load_file(shard))
        st.load_hf_state_dict(sd)
        st.version = getattr(st, "version", 0) + 1
        opt = torch.load(os.path.join(path, "optim_world_size_1_rank_0.pt"), map_location="cpu")
        st.m.copy_(opt["m"]); st.v.copy_(opt["v"])
        if "c" in opt and st.c is not None:
            st.c.copy_(opt["c"])
        if st.master is not None:
            st.master.copy_(opt["master"]) if "master" in opt else st.master.copy_(st.flat)
        eng.opt_steps, eng.sched_steps = opt["opt_steps"], opt["sched_steps"]
        self.last_load_info = None
        if self.world_size > 1:
            torch.distributed.barrier()
        return {k: v for k, v in opt.items() if k not in ("m", "v", "c", "master", "opt_steps", "sched_steps", "strategy")}
