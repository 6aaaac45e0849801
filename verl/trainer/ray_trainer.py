"""GRPO trainer: the step loop of the reference's RayPPOTrainer.fit (verl/trainer/ray_trainer.py:543-721) without Ray.

One process per GPU (torchrun); every rank runs this loop on its own share of the rollout batch (SPMD).  Phases, timers
and metric names follow the reference: gen -> reward -> (balance) -> old -> ref -> adv -> update_actor.  Group-relative
advantages need all G rollouts of a prompt, which stay on the rank that generated them, so no activation or score ever
crosses the fabric; the only collective is the gradient all-reduce inside update_actor."""
from __future__ import annotations

import os
import time
import uuid
from contextlib import contextmanager
from enum import Enum
from typing import Any, Dict, Optional

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, RandomSampler, SequentialSampler

from ..protocol import DataProto
from ..utils.seqlen_balancing import get_seqlen_balanced_partitions, log_seqlen_unbalance
from . import core_algos
from .metrics import compute_data_metrics, compute_throughout_metrics, compute_timing_metrics, reduce_metrics


class AdvantageEstimator(str, Enum):
    GAE = "gae"
    GRPO = "grpo"
    REINFORCE_PLUS_PLUS = "reinforce_plus_plus"
    REMAX = "remax"
    RLOO = "rloo"


@contextmanager
def _timer(name: str, timing_raw: Dict[str, float]):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    yield
    torch.cuda.synchronize()
    timing_raw[name] = time.perf_counter() - t0


def compute_advantage(data: DataProto, adv_estimator: str):
    if adv_estimator != AdvantageEstimator.GRPO.value:
        raise NotImplementedError("only algorithm.adv_estimator=grpo is on the SpatialThinker path")
    adv, ret = core_algos.compute_grpo_outcome_advantage(data.batch["token_level_rewards"], data.batch["response_mask"],
                                                         data.non_tensor_batch["uid"])
    data.batch["advantages"], data.batch["returns"] = adv, ret
    return data


def _kl_tokens(old: torch.Tensor, ref: torch.Tensor, kind: str) -> torch.Tensor:
    """compute_kl (core_algos.py:394-436) on host tensors — used only by the reward-side KL penalty branch."""
    if kind == "kl":
        return old - ref
    if kind == "abs":
        return (old - ref).abs()
    if kind == "mse":
        return 0.5 * (old - ref).square()
    if kind == "low_var_kl":
        d = ref - old
        return torch.clamp(d.exp() - d - 1, min=-10, max=10)
    if kind == "chi2":
        return torch.clamp(((ref - old).exp() - 1) ** 2, min=0, max=20)
    raise NotImplementedError(kind)


def apply_kl_penalty(data: DataProto, kl_ctrl, kl_penalty="kl"):
    """token_level_rewards = scores - kl_coef * kl(old, ref)  (ray_trainer.py:125-145; only when use_kl_loss is off)."""
    scores, mask = data.batch["token_level_scores"], data.batch["response_mask"]
    if "ref_log_probs" in data.batch.keys():
        kld = _kl_tokens(data.batch["old_log_probs"].float(), data.batch["ref_log_probs"].float(), kl_penalty) * mask
    else:
        kld = torch.zeros_like(mask, dtype=torch.float32)
    data.batch["token_level_rewards"] = scores - kl_ctrl.kl_coef * kld
    cur = ((kld * mask).sum(-1) / (mask.sum(-1) + 1e-8)).mean().item()
    kl_ctrl.update(current_kl=cur, n_steps=len(data))
    return data, {"critic/kl": cur, "critic/kl_coef": kl_ctrl.kl_coef}


class ConsoleTracker:
    def __init__(self, loggers, config=None):
        self.rank = int(os.environ.get("RANK", 0))
        extra = [l for l in loggers if l != "console"]
        if extra and self.rank == 0:
            print(f"[logger] only the console logger is built; ignoring {extra}")

    def log(self, data: Dict[str, Any], step: int):
        if self.rank == 0:
            print(f"step {step}: " + " - ".join(f"{k}:{v:.4g}" if isinstance(v, (int, float)) else f"{k}:{v}" for k, v in sorted(data.items())), flush=True)


class RayPPOTrainer:
    """Name kept for drop-in use by verl.trainer.main; there is no Ray underneath."""

    def __init__(self, config, tokenizer, processor, worker_group, ref_worker_group, reward_fn, val_reward_fn, train_dataset, val_dataset=None):
        self.config, self.tokenizer, self.processor = config, tokenizer, processor
        self.actor_rollout_wg, self.ref_policy_wg = worker_group, ref_worker_group      # may be attached later (set_worker_groups)
        self.reward_fn, self.val_reward_fn = reward_fn, val_reward_fn
        self.world = int(os.environ.get("WORLD_SIZE", 1))
        self.rank = int(os.environ.get("RANK", 0))
        a = config.algorithm
        if a.adv_estimator != "grpo":
            raise NotImplementedError("only algorithm.adv_estimator=grpo is built")
        self.use_reference_policy = not a.disable_kl
        self.kl_ctrl = core_algos.get_kl_controller(a) if self.use_reference_policy else core_algos.FixedKLController(0.0)
        d, act = config.data, config.worker.actor
        # validation of ray_trainer.py:238-263
        if d.rollout_batch_size % act.global_batch_size != 0:
            raise ValueError("Rollout batch size must be divisible by global batch size.")
        if (d.rollout_batch_size * config.worker.rollout.n) % act.micro_batch_size_per_device_for_experience != 0:
            raise ValueError("Rollout batch size * rollout.n must be divisible by actor micro batch size for experience.")
        if config.worker.rollout.n == 1:
            raise ValueError("GRPO and RLOO algorithm need `config.worker.rollout.n > 1`.")
        if d.rollout_batch_size % self.world != 0:
            raise ValueError("rollout_batch_size must be divisible by the number of GPUs")
        self.local_prompts = d.rollout_batch_size // self.world
        gen = torch.Generator().manual_seed(d.seed)
        sampler = RandomSampler(train_dataset, generator=gen) if d.shuffle else SequentialSampler(train_dataset)
        from ..utils.dataset import collate_fn
        self.train_dataloader = DataLoader(train_dataset, batch_size=d.rollout_batch_size, sampler=sampler, num_workers=0,
                                           collate_fn=collate_fn, drop_last=True)
        self.val_dataloader = None
        if val_dataset is not None:
            vb = len(val_dataset) if d.val_batch_size == -1 else d.val_batch_size
            self.val_dataloader = DataLoader(val_dataset, batch_size=vb, shuffle=False, collate_fn=collate_fn, drop_last=False)
        t = config.trainer
        self.training_steps = t.max_steps if t.max_steps is not None else len(self.train_dataloader) * t.total_episodes
        act.optim.training_steps = self.training_steps
        self.global_step = 0

    def set_worker_groups(self, actor_rollout_wg, ref_policy_wg):
        self.actor_rollout_wg, self.ref_policy_wg = actor_rollout_wg, ref_policy_wg

    def init_workers(self):
        if self.use_reference_policy and self.ref_policy_wg is not self.actor_rollout_wg:
            self.ref_policy_wg.init_model()
        self.actor_rollout_wg.init_model()

    # ------------------------------------------------------------------------------------------------
    def _shard(self, batch_dict: Dict[str, Any]) -> Dict[str, Any]:
        """This rank's contiguous share of the global rollout batch (Dispatch.DP_COMPUTE_PROTO's chunk(world)[rank])."""
        lo, hi = self.rank * self.local_prompts, (self.rank + 1) * self.local_prompts
        return {k: v[lo:hi] for k, v in batch_dict.items()}

    def _balance_batch(self, batch: DataProto, metrics: Dict[str, Any], logging_prefix: str = "global_seqlen") -> None:
        """ray_trainer.py:526-541.  The partitioner is the reference's; rows stay on the rank that generated them (they never
        leave HBM-side locality), so with one rank per GPU the reorder is the identity and only the statistics are logged."""
        lens = batch.batch["attention_mask"].sum(-1).tolist()
        parts = get_seqlen_balanced_partitions(lens, k_partitions=1, equal_size=True)
        metrics.update(log_seqlen_unbalance(lens, parts, logging_prefix))

    def _validate(self) -> Dict[str, Any]:
        if self.val_dataloader is None:
            return {}
        scores = []
        for batch_dict in self.val_dataloader:
            test = DataProto.from_single_dict(batch_dict)
            keys = ["raw_prompt_ids", "multi_modal_data", "multi_modal_inputs"] if "multi_modal_inputs" in test.non_tensor_batch else ["raw_prompt_ids"]
            gen = test.pop(batch_keys=["input_ids", "attention_mask", "position_ids"], non_tensor_batch_keys=[k for k in keys if k in test.non_tensor_batch])
            gen.meta_info = dict(self.config.worker.rollout.val_override_config)
            out = self.actor_rollout_wg.generate_sequences(gen)
            n = int(gen.meta_info.get("n", 1))
            test = test.repeat(n, interleave=True).union(out) if n > 1 else test.union(out)
            reward, _ = self.val_reward_fn(test)
            scores.append(reward.sum(-1))
        return {"val/test_score": torch.cat(scores).mean().item()}

    def _save_checkpoint(self):
        path = os.path.join(self.config.trainer.save_checkpoint_path, f"global_step_{self.global_step}")
        os.makedirs(os.path.join(path, "actor"), exist_ok=True)
        self.actor_rollout_wg.save_checkpoint(os.path.join(path, "actor"))
        if self.rank == 0:
            with open(os.path.join(self.config.trainer.save_checkpoint_path, "latest_global_step.txt"), "w") as f:
                f.write(str(self.global_step))

    def _load_checkpoint(self):
        p = self.config.trainer.load_checkpoint_path
        if p is None:
            return
        if "global_step_" not in p.strip(os.path.sep).split(os.path.sep)[-1]:
            raise ValueError("`load_checkpoint_path` should end with `global_step_*`.")
        self.global_step = int(p.strip(os.path.sep).split("global_step_")[-1])
        self.actor_rollout_wg.load_checkpoint(os.path.join(p, "actor"))

    # ------------------------------------------------------------------------------------------------
    def fit(self):
        cfg = self.config
        self.logger = ConsoleTracker(cfg.trainer.logger, cfg.to_dict())
        self._load_checkpoint()
        val_metrics: Optional[Dict[str, Any]] = None
        if self.val_reward_fn is not None and cfg.trainer.val_before_train and self.val_dataloader is not None:
            val_metrics = self._validate()
            self.logger.log(val_metrics, self.global_step)
            if cfg.trainer.val_only:
                return
        n = cfg.worker.rollout.n
        for _ in range(cfg.trainer.total_episodes):
            for batch_dict in self.train_dataloader:
                self.global_step += 1
                if self.global_step > self.training_steps:
                    break
                metrics, timing_raw = {}, {}
                batch = DataProto.from_single_dict(self._shard(batch_dict))
                nt_keys = [k for k in ("raw_prompt_ids", "multi_modal_data", "multi_modal_inputs") if k in batch.non_tensor_batch]
                gen_batch = batch.pop(batch_keys=["input_ids", "attention_mask", "position_ids"], non_tensor_batch_keys=nt_keys)
                if "synthetic_response_lengths" in batch.meta_info:
                    gen_batch.meta_info["synthetic_response_lengths"] = batch.meta_info["synthetic_response_lengths"]
                with _timer("step", timing_raw):
                    with _timer("gen", timing_raw):
                        gen_out = self.actor_rollout_wg.generate_sequences(gen_batch)
                    batch.non_tensor_batch["uid"] = np.array([str(uuid.uuid4()) for _ in range(len(batch))], dtype=object)
                    batch = batch.repeat(repeat_times=n, interleave=True)
                    batch = batch.union(gen_out)
                    with _timer("reward", timing_raw):
                        reward_tensor, reward_metrics = self.reward_fn(batch)
                        batch.batch["token_level_scores"] = reward_tensor
                        metrics.update({f"reward/{k}": v for k, v in reduce_metrics(reward_metrics).items()})
                    self._balance_batch(batch, metrics)
                    batch.meta_info["global_token_num"] = torch.sum(batch.batch["attention_mask"], dim=-1).tolist()
                    with _timer("old", timing_raw):
                        batch = batch.union(self.actor_rollout_wg.compute_log_probs(batch))
                    if self.use_reference_policy:
                        with _timer("ref", timing_raw):
                            batch = batch.union(self.ref_policy_wg.compute_ref_log_probs(batch))
                    with _timer("adv", timing_raw):
                        if not cfg.algorithm.use_kl_loss and self.use_reference_policy:
                            batch, kl_metrics = apply_kl_penalty(batch, self.kl_ctrl, cfg.algorithm.kl_penalty)
                            metrics.update(kl_metrics)
                        else:
                            batch.batch["token_level_rewards"] = batch.batch["token_level_scores"]
                        batch = compute_advantage(batch, cfg.algorithm.adv_estimator)
                    if cfg.trainer.critic_warmup <= self.global_step:
                        with _timer("update_actor", timing_raw):
                            actor_out = self.actor_rollout_wg.update_actor(batch)
                        metrics.update(reduce_metrics(actor_out.non_tensor_batch))
                    if self.val_reward_fn is not None and cfg.trainer.val_freq > 0 and self.global_step % cfg.trainer.val_freq == 0:
                        with _timer("validation", timing_raw):
                            val_metrics = self._validate()
                        metrics.update(val_metrics)
                    if cfg.trainer.save_freq > 0 and self.global_step % cfg.trainer.save_freq == 0:
                        with _timer("save_checkpoint", timing_raw):
                            self._save_checkpoint()
                metrics.update(compute_data_metrics(batch))
                metrics.update(compute_timing_metrics(batch, timing_raw))
                metrics.update(compute_throughout_metrics(batch, timing_raw, n_gpus=1))      # per-rank tokens / per-rank time = per-GPU rate
                metrics["perf/samples_per_s_per_gpu"] = len(batch) / timing_raw["step"]
                self.logger.log(metrics, self.global_step)
            if self.global_step > self.training_steps:
                break
        # the reference always writes a final checkpoint (ray_trainer.py:718-719); ST_SKIP_FINAL_SAVE=1 is a test/bench knob for
        # multi-GB synthetic models whose final state nobody will read
        if (cfg.trainer.save_freq <= 0 or self.global_step % cfg.trainer.save_freq != 0) and os.environ.get("ST_SKIP_FINAL_SAVE") != "1":
            self._save_checkpoint()
