"""FSDPWorker — the worker API surface of the reference (verl/workers/fsdp_workers.py:65-616) on the MI355X engine.

Same constructor, roles and `@register`-tagged methods (init_model, generate_sequences, compute_log_probs,
compute_ref_log_probs, update_actor, save_checkpoint, load_checkpoint) and the same DataProto key contract
(SURVEY.md §8b).  What changed underneath: no FSDP sharding (a 7B actor + optimizer + frozen reference is ~130 GB of the
288 GB HBM, so every rank keeps full replicas and gradients are all-reduced once per optimizer step over RCCL), no
vLLM (the generator reads the actor's own weight buffer, so the per-step weight hand-off of fsdp_vllm.py:76-116 is
free), no CPU offload.  The `fsdp.*`, `offload.*`, `rollout.tensor_parallel_size`, `gpu_memory_utilization` keys are
accepted and ignored."""
from __future__ import annotations

import os
import time
from typing import Literal

import numpy as np
import psutil
import torch
import torch.distributed as dist

from spatialthinker_amd.actor import ActorHyper, CriticEngine, PolicyEngine
from spatialthinker_amd.pretrained import load_model, save_hf
from spatialthinker_amd.rollout import Generator

from ..protocol import DataProto
from ..single_controller.base import Worker
from ..single_controller.base.decorator import Dispatch, register
from ..utils.flops_counter import FlopsCounter
from ..utils.tokenizer import get_processor, get_tokenizer
from .rollout import assemble_rollout_batch
from .sharding_manager import FSDPUlyssesShardingManager


class _SpDim:
    """The "sp" dimension of the reference's device mesh, as FSDPUlyssesShardingManager reads it."""

    def __init__(self, group, size: int, local_rank: int):
        self._g, self._n, self._r = group, size, local_rank

    def get_group(self):
        return self._g

    def size(self):
        return self._n

    def get_local_rank(self):
        return self._r


class FSDPWorker(Worker):
    _warned_padding = False

    def __init__(self, config, role: Literal["actor", "critic", "rollout", "ref", "actor_rollout", "actor_rollout_ref"]):
        super().__init__()
        self.config, self.role = config, role
        self.world_size = int(os.environ.get("WORLD_SIZE", 1))
        self.rank = int(os.environ.get("RANK", 0))
        if self.world_size > 1 and not dist.is_initialized():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
            dist.init_process_group(backend="nccl")           # "nccl" IS RCCL on ROCm
        self._is_actor = role in ("actor", "actor_rollout", "actor_rollout_ref")
        self._is_rollout = role in ("rollout", "actor_rollout", "actor_rollout_ref")
        self._is_ref = role in ("ref", "actor_rollout_ref")
        self._is_critic = role == "critic"
        # Ulysses sequence parallelism (fsdp_workers.py:107-125): the ranks form a ("dp", "sp") mesh, rank = dp_index * sp + sp_index; the sp
        # ranks of a group run the SAME rows, each on its slice of every packed pass (round 5: Qwen25VL.set_sequence_parallel, all-to-all
        # around the attention kernels, verl/utils/ulysses.py).  A 288 GB MI355X holds the shipped scripts' 8192-token sequences without
        # it (every script sets 1); it exists for longer sequences and for parity of the option.
        self.sp_size = 1 if role == "critic" else int(getattr(self.config.actor, "ulysses_sequence_parallel_size", 1) or 1)
        self.sp_group, self.ulysses_sharding_manager = None, FSDPUlyssesShardingManager(None)
        if self.sp_size > 1:
            if self.world_size % self.sp_size:
                raise ValueError(f"ulysses_sequence_parallel_size = {self.sp_size} does not divide the {self.world_size} ranks")
            for k in range(self.world_size // self.sp_size):                 # every rank creates every group (torch.distributed's contract)
                grp = dist.new_group(list(range(k * self.sp_size, (k + 1) * self.sp_size)))
                if k == self.rank // self.sp_size:
                    self.sp_group = grp
            self.ulysses_sharding_manager = FSDPUlyssesShardingManager({"sp": _SpDim(self.sp_group, self.sp_size, self.rank % self.sp_size)})
        if not bool(getattr(self.config.actor, "padding_free", True)) and self.rank == 0 and not FSDPWorker._warned_padding:
            FSDPWorker._warned_padding = True
            print("[FSDPWorker] worker.actor.padding_free=false: this engine always runs the padding-free (packed) formulation of "
                  "dp_actor.py:86-138; the log-probs are the same function of the same tokens, only the padded rows are never computed", flush=True)
        if self._is_actor:
            self._init_batch_sizes(self.config.actor)
        if self._is_critic:
            if int(getattr(self.config.critic, "ulysses_sequence_parallel_size", 1)) > 1:
                raise NotImplementedError("worker.critic.ulysses_sequence_parallel_size > 1: Ulysses sequence parallelism is not built — set it to 1")
            self._init_batch_sizes(self.config.critic)

    def _init_batch_sizes(self, cfg):
        """fsdp_workers.py:130-147: global batch is counted in rollouts, then split over the ranks."""
        if self.config.rollout.n > 1:
            cfg.global_batch_size *= self.config.rollout.n
        cfg.global_batch_size_per_device = cfg.global_batch_size // self.world_size * getattr(self, "sp_size", 1)     # :133-135
        if cfg.global_batch_size_per_device == 0:
            raise ValueError("actor global batch size * ulysses size must be larger than num gpus.")
        if cfg.global_batch_size_per_device % cfg.micro_batch_size_per_device_for_update != 0:
            raise ValueError("actor global batch size per device must be divisible by the micro batch size.")

    def print_rank0(self, *a):
        if self.rank == 0:
            print(*a, flush=True)

    # ------------------------------------------------------------------------------------------------
    @register(dispatch_mode=Dispatch.ONE_TO_ALL)
    def init_model(self):
        mc = self.config.critic.model if self._is_critic else self.config.actor.model
        if self._is_critic and not mc.model_path:               # the shipped configs leave worker.critic.model empty: the actor's backbone
            mc = self.config.actor.model
        self.tokenizer = get_tokenizer(mc.tokenizer_path or mc.model_path, trust_remote_code=mc.trust_remote_code, use_fast=True)
        self.processor = get_processor(mc.tokenizer_path or mc.model_path, trust_remote_code=mc.trust_remote_code, use_fast=True)
        if self._is_critic:
            # fsdp_workers.py:212-224 builds AutoModelForTokenClassification(num_labels = 1) — a mapping transformers does not have for
            # qwen2_5_vl, so the reference cannot construct this model; here the same backbone carries a score = Linear(H, 1) head
            cr = self.config.critic
            if cr.optim.strategy not in ("adamw_bf16", "adamw"):
                raise NotImplementedError(f"Optimizer {cr.optim.strategy} not supported.")
            dt = (cr.fsdp.torch_dtype or "fp32").lower()
            master = dt not in ("bf16", "bfloat16")
            if master and cr.optim.strategy != "adamw":
                raise NotImplementedError("worker.critic.fsdp.torch_dtype=fp32 needs optim.strategy=adamw (fp32 master weights); "
                                          "adamw_bf16 goes with torch_dtype=bf16")
            cfg, store, special = load_model(mc.model_path, trainable=True, master_fp32=master, value_head=True)
            hyper = ActorHyper(micro_batch_size_per_device_for_update=cr.micro_batch_size_per_device_for_update,
                               micro_batch_size_per_device_for_experience=cr.micro_batch_size_per_device_for_experience,
                               global_batch_size_per_device=cr.global_batch_size_per_device, max_grad_norm=cr.max_grad_norm,
                               ppo_epochs=cr.ppo_epochs, lr=cr.optim.lr, betas=tuple(cr.optim.betas), weight_decay=cr.optim.weight_decay,
                               lr_warmup_steps=int(cr.optim.lr_warmup_ratio * max(cr.optim.training_steps, 0)), optim_strategy=cr.optim.strategy,
                               freeze_vision_tower=bool(mc.freeze_vision_tower), cliprange_value=cr.cliprange_value,
                               grad_exchange_dtype="bf16" if str(cr.fsdp.mp_reduce_dtype).lower() in ("bf16", "bfloat16") else "fp32")
            self.model_config, self.special = cfg, special
            self.critic = self.actor = CriticEngine(cfg, store, hyper)      # (save / load_checkpoint address the trainable engine as self.actor)
            self.flops_counter = FlopsCounter(cfg)
            if self.world_size > 1:
                dist.broadcast(store.flat, src=0)
                if store.master is not None:
                    dist.broadcast(store.master, src=0)
                store.refresh_transposes()
            return
        if self._is_actor:
            a = self.config.actor
            if a.optim.strategy not in ("adamw_bf16", "adamw"):
                raise NotImplementedError(f"Optimizer {a.optim.strategy} not supported.")
            dt = (a.fsdp.torch_dtype or "fp32").lower()
            if dt not in ("bf16", "bfloat16", "fp32", "float32", "float"):
                raise NotImplementedError(f"worker.actor.fsdp.torch_dtype={a.fsdp.torch_dtype!r}: bf16 or fp32")
            master = dt not in ("bf16", "bfloat16")
            if master and a.optim.strategy != "adamw":
                raise NotImplementedError(
                    "worker.actor.fsdp.torch_dtype=fp32 (the reference's default when unset, fsdp_workers.py:186-189) keeps fp32 master "
                    "weights with torch.optim.AdamW (optim.strategy=adamw, the reference's default pair); AnyPrecisionAdamW on fp32 "
                    "parameters (strategy=adamw_bf16 without torch_dtype=bf16) is not built — every shipped STVQA script passes "
                    "torch_dtype=bf16 with adamw_bf16 (scripts/spatialthinker_7b_grpo.sh:25)")
            cfg, store, special = load_model(mc.model_path, trainable=True, master_fp32=master)      # master first: it takes the checkpoint's own fp32 values
            if master:
                self.print_rank0("Actor parameters: fp32 master weights + fp32 AdamW moments, bf16 compute copy (MixedPrecision param_dtype).")
            if mc.freeze_vision_tower:
                self.print_rank0("Vision tower is set to not trainable.")
            hyper = ActorHyper(micro_batch_size_per_device_for_update=a.micro_batch_size_per_device_for_update,
                               micro_batch_size_per_device_for_experience=a.micro_batch_size_per_device_for_experience,
                               global_batch_size_per_device=a.global_batch_size_per_device, max_grad_norm=a.max_grad_norm,
                               clip_ratio_low=a.clip_ratio_low, clip_ratio_high=a.clip_ratio_high, clip_ratio_dual=a.clip_ratio_dual,
                               ppo_epochs=a.ppo_epochs, use_kl_loss=a.use_kl_loss, disable_kl=a.disable_kl, kl_penalty=a.kl_penalty,
                               kl_coef=a.kl_coef, lr=a.optim.lr, betas=tuple(a.optim.betas), weight_decay=a.optim.weight_decay,
                               lr_warmup_steps=int(a.optim.lr_warmup_ratio * max(a.optim.training_steps, 0)),
                               optim_strategy=a.optim.strategy, freeze_vision_tower=bool(mc.freeze_vision_tower),
                               # FSDP's MixedPrecision(reduce_dtype=...) of the reference (fsdp_workers.py:238-243) = the payload dtype of the
                               # gradient exchange here; ST_GRAD_EXCHANGE=reduce_scatter selects SURVEY §5.8's direct schedule
                               grad_exchange_dtype="bf16" if str(a.fsdp.mp_reduce_dtype).lower() in ("bf16", "bfloat16") else "fp32")
            self.model_config, self.special = cfg, special
            self.actor = PolicyEngine(cfg, store, hyper, sp_group=self.sp_group)
            self.flops_counter = FlopsCounter(cfg)
            if self.world_size > 1:                          # sync_module_states: rank 0's weights everywhere (fsdp_workers.py:261-263)
                dist.broadcast(store.flat, src=0)
                if store.master is not None:
                    dist.broadcast(store.master, src=0)
                store.refresh_transposes()
        if self._is_rollout:
            self.generator = Generator(self.actor.model)
        if self._is_ref:
            cfg, store, special = load_model(mc.model_path, trainable=False)
            if self._is_actor:
                store.flat.copy_(self.actor.store.flat)
            elif self.world_size > 1:
                dist.broadcast(store.flat, src=0)
            self.model_config, self.special = cfg, special
            self.ref_policy = PolicyEngine(cfg, store, None, sp_group=self.sp_group)

    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def _as_dict(data: DataProto):
        d = {k: v for k, v in data.batch.items()}
        d.update(data.non_tensor_batch)
        return d

    @register(dispatch_mode=Dispatch.DP_COMPUTE_PROTO)
    def generate_sequences(self, prompts: DataProto) -> DataProto:
        assert self._is_rollout
        r = self.config.rollout
        over = {k: prompts.meta_info[k] for k in ("temperature", "n", "top_p", "top_k", "ignore_eos") if k in prompts.meta_info}
        n = int(over.get("n", r.n))
        temperature = float(over.get("temperature", r.temperature))
        top_p, top_k = float(over.get("top_p", r.top_p)), int(over.get("top_k", r.top_k))
        ids, mask, pos = prompts.batch["input_ids"], prompts.batch["attention_mask"], prompts.batch["position_ids"]
        mm = prompts.non_tensor_batch.get("multi_modal_inputs")
        px = gr = None
        if mm is not None:
            px = [m["pixel_values"] for m in mm]
            gr = [m["image_grid_thw"] for m in mm]
        eos = self.special["eos"]
        self._gen_calls = getattr(self, "_gen_calls", 0) + 1
        forced = prompts.meta_info.get("synthetic_response_lengths")       # set by a benchmark dataset only (verl/utils/dataset.py); None in training
        # the prefill's prompt K/V serves the old-policy log-prob pass that follows on the same weights (PolicyEngine checks that
        # the cache matches the rows it is handed, else it simply recomputes)
        resp, self._prompt_cache = self.generator.generate(
            ids, mask, pos, n=n, max_new_tokens=r.response_length, temperature=temperature, eos_token_id=eos,
            pad_token_id=self.special["pad"], seed=(self.rank + 1000) * 100003 + self._gen_calls, pixel_values=px, image_grid_thw=gr,
            ignore_eos=bool(over.get("ignore_eos", r.ignore_eos)), forced_lengths=forced,
            top_k=top_k, top_p=top_p, return_prompt_cache=True, emit_log_probs=self._old_from_rollout())
        batch = assemble_rollout_batch(ids, mask, pos, resp.cpu(), n, eos)          # vllm_rollout_spmd.py:144-188
        non_tensor = {}
        if mm is not None:
            non_tensor["multi_modal_inputs"] = np.repeat(mm, n, axis=0) if n > 1 else mm
        return DataProto.from_dict(batch, non_tensors=non_tensor)

    @register(dispatch_mode=Dispatch.DP_COMPUTE_PROTO)
    def compute_log_probs(self, data: DataProto) -> DataProto:
        assert self._is_actor
        t = self.config.rollout.temperature
        data.meta_info["temperature"] = t
        cache, self._prompt_cache = getattr(self, "_prompt_cache", None), None            # one use, then the K/V memory is released
        with self.ulysses_sharding_manager:                                               # sp > 1: the group's rows in, this rank's rows out (:514-520)
            data = self.ulysses_sharding_manager.preprocess_data(data)
            lp = self.actor.compute_log_prob(self._as_dict(data), t, prompt_cache=cache if self.sp_size == 1 else None,
                                             use_rollout_log_probs=self._old_from_rollout()).cpu()
            out = self.ulysses_sharding_manager.postprocess_data(DataProto.from_dict(tensors={"old_log_probs": lp}))
        out.meta_info = {"temperature": t, "prompt_cache_hit": bool(self.actor.last_prompt_cache_hit),
                         "old_log_probs_source": getattr(self.actor, "last_log_prob_source", "forward")}
        return out

    def _old_from_rollout(self) -> bool:
        """worker.rollout.old_log_probs_from_rollout (or ST_OLD_FROM_ROLLOUT=1): opt-in, see PolicyEngine.compute_log_prob."""
        return bool(getattr(self.config.rollout, "old_log_probs_from_rollout", False)) or os.environ.get("ST_OLD_FROM_ROLLOUT", "0") == "1"

    @register(dispatch_mode=Dispatch.DP_COMPUTE_PROTO)
    def compute_ref_log_probs(self, data: DataProto) -> DataProto:
        assert self._is_ref
        t = self.config.rollout.temperature
        with self.ulysses_sharding_manager:                                               # :541-545
            data = self.ulysses_sharding_manager.preprocess_data(data)
            lp = self.ref_policy.compute_log_prob(self._as_dict(data), t, self.config.ref.micro_batch_size_per_device_for_experience).cpu()
            return self.ulysses_sharding_manager.postprocess_data(DataProto.from_dict(tensors={"ref_log_probs": lp}))

    @register(dispatch_mode=Dispatch.DP_COMPUTE_PROTO)
    def update_actor(self, data: DataProto) -> DataProto:
        assert self._is_actor
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        with self.ulysses_sharding_manager:                                               # :434-436 (metrics are not re-sharded)
            data = self.ulysses_sharding_manager.preprocess_data(data)
            metrics = self.actor.update_policy(self._as_dict(data), data.meta_info["temperature"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        est, promised = self.flops_counter.estimate_flops(data.meta_info["global_token_num"], dt)
        # meta_info["global_token_num"] holds THIS rank's sequences (each rank drives its own shard), so the estimate is already
        # per GPU: no division by the world size (the reference divides a global count, fsdp_workers.py:447-449)
        metrics["perf/mfu_actor"] = est * self.config.actor.ppo_epochs / promised
        metrics["perf/max_memory_allocated_gb"] = torch.cuda.max_memory_allocated() / (1024 ** 3)
        metrics["perf/max_memory_reserved_gb"] = torch.cuda.max_memory_reserved() / (1024 ** 3)
        metrics["perf/cpu_memory_used_gb"] = psutil.virtual_memory().used / (1024 ** 3)
        return DataProto(non_tensor_batch={k: np.array([v] if np.isscalar(v) else v) for k, v in metrics.items()})

    @register(dispatch_mode=Dispatch.DP_COMPUTE_PROTO)
    def compute_values(self, data: DataProto) -> DataProto:
        """fsdp_workers.py:558-575."""
        assert self._is_critic
        v = self.critic.compute_values(self._as_dict(data)).cpu()
        return DataProto.from_dict(tensors={"values": v})

    @register(dispatch_mode=Dispatch.DP_COMPUTE_PROTO)
    def update_critic(self, data: DataProto) -> DataProto:
        """fsdp_workers.py:577-616."""
        assert self._is_critic
        t0 = time.perf_counter()
        metrics = self.critic.update_critic(self._as_dict(data))
        torch.cuda.synchronize()
        est, promised = self.flops_counter.estimate_flops(data.meta_info["global_token_num"], time.perf_counter() - t0)
        metrics["perf/mfu_critic"] = est * self.config.actor.ppo_epochs / promised          # (the reference multiplies by the ACTOR's epochs, :594-596)
        return DataProto(non_tensor_batch={k: np.array([v] if np.isscalar(v) else v) for k, v in metrics.items()})

    # ------------------------------------------------------------------------------------------------
    def _checkpoint_manager(self):
        from ..utils.checkpoint.fsdp_checkpoint_manager import FSDPCheckpointManager
        return FSDPCheckpointManager(model=self.actor, processing_class=getattr(self, "processor", None), tokenizer=getattr(self, "tokenizer", None))

    @register(dispatch_mode=Dispatch.ONE_TO_ALL)
    def save_checkpoint(self, path: str):
        """FSDPCheckpointManager.save_checkpoint (verl/utils/checkpoint/fsdp_checkpoint_manager.py): ONE HF-loadable directory + ONE optimizer /
        scheduler-position file written by rank 0 (replicas are identical; the reference writes a shard per rank, :83-131); the rollout's
        seed position travels as `gen_calls`."""
        assert self._is_actor or self._is_critic
        self._checkpoint_manager().save_checkpoint(path, extra={"gen_calls": getattr(self, "_gen_calls", 0)})

    @register(dispatch_mode=Dispatch.ONE_TO_ALL)
    def load_checkpoint(self, path: str):
        if path is None:
            return
        mgr = self._checkpoint_manager()
        extra = mgr.load_checkpoint(path)
        if mgr.last_load_info is not None:                    # a run checkpointed by the REFERENCE (model_/optim_/extra_state_world_size_W_rank_r.pt)
            info = mgr.last_load_info
            self.print_rank0(f"Loaded a reference-layout checkpoint written by {info['world_size']} ranks: optimizer state {info['optimizer']}, "
                             f"optimizer step {info['opt_steps']}, scheduler step {info['sched_steps']}.")
            return
        self._gen_calls = extra.get("gen_calls", 0)                   # the rollout seed stream continues where it stopped
