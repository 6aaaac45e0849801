"""GRPO trainer: the step loop of the reference's RayPPOTrainer.fit (verl/trainer/ray_trainer.py:543-721) without Ray.

One process per GPU (torchrun); every rank runs this loop on its own share of the rollout batch (SPMD).  Phases, timers
and metric names follow the reference: gen -> reward -> balance -> old -> ref -> adv -> update_actor.  Group-relative
advantages need all G rollouts of a prompt, which stay on the rank that generated them, so no activation or score ever
crosses the fabric; the only data-path collective is the gradient all-reduce inside update_actor.  What the reference's single
driver sees of the whole batch — reward / actor / data metrics, the balance statistics, validation scores — is rebuilt from
small per-rank python objects gathered over the process group (`SPMDWorkerGroup.gather_objects`), so rank 0 logs the same
numbers a one-controller run would."""
from __future__ import annotations

import os
import shutil
import time
import uuid
from collections import defaultdict
from contextlib import contextmanager
from copy import deepcopy
from enum import Enum, IntEnum
from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from ..protocol import DataProto
from ..utils.checkpoint.checkpoint_manager import CHECKPOINT_TRACKER, remove_obsolete_ckpt
from ..utils.dataloader import ForeignDataloaderState, ResumableDataLoader
from ..utils.logger import Tracker
from ..utils.seqlen_balancing import get_seqlen_balanced_partitions, log_seqlen_unbalance
from . import core_algos
from .metrics import compute_data_metrics, compute_throughout_metrics, compute_timing_metrics, reduce_metrics



class Role(IntEnum):
    """Worker roles of the reference's role -> worker-class mapping (ray_trainer.py:53-64; its values are `auto()` in this order).  This
    build's FSDPWorker takes the role as a string ("actor_rollout_ref", "critic", ...); the enum is kept for launchers that name roles by it."""
    Actor = 1
    Rollout = 2
    ActorRollout = 3
    Critic = 4
    RefPolicy = 5
    RewardModel = 6
    ActorRolloutRef = 7


class AdvantageEstimator(str, Enum):
    GAE = "gae"
    GRPO = "grpo"
    REINFORCE_PLUS_PLUS = "reinforce_plus_plus"
    REMAX = "remax"
    RLOO = "rloo"


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


@contextmanager
def _timer(name: str, timing_raw: Dict[str, float]):
    _sync()
    t0 = time.perf_counter()
    yield
    _sync()
    timing_raw[name] = time.perf_counter() - t0


def _dist_sum(t: torch.Tensor) -> torch.Tensor:
    """Sum a small host tensor over the ranks (identity on one rank): the batch-wide statistics of masked_whiten."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            d = t.to(torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(d)
            return d.to(t.device)
        dist.all_reduce(t)
    return t


def compute_advantage(data: DataProto, adv_estimator: str, gamma: float = 1.0, lam: float = 1.0, whole_batch: bool = False):
    """ray_trainer.py:148-175.  GRPO runs the HIP kernel; the other estimators are host math (core_algos).
    whole_batch: `data` already holds every rank's rows (migrate mode), so batch-wide statistics need no cross-rank sum."""
    rew, mask, index = data.batch["token_level_rewards"], data.batch["response_mask"], data.non_tensor_batch["uid"]
    est = AdvantageEstimator(adv_estimator).value
    _sum = (lambda t: t) if whole_batch else _dist_sum
    if est == "grpo":
        adv, ret = core_algos.compute_grpo_outcome_advantage(rew, mask, index)
    elif est == "gae":
        adv, ret = core_algos.compute_gae_advantage_return(rew, data.batch["values"], mask, gamma, lam, all_reduce=_sum)
    elif est == "reinforce_plus_plus":
        adv, ret = core_algos.compute_reinforce_plus_plus_outcome_advantage(rew, mask, gamma, all_reduce=_sum)
    elif est == "remax":
        adv, ret = core_algos.compute_remax_outcome_advantage(rew, data.batch["reward_baselines"], mask)
    else:
        adv, ret = core_algos.compute_rloo_outcome_advantage(rew, mask, index)
    data.batch["advantages"], data.batch["returns"] = adv, ret
    return data


def apply_kl_penalty(data: DataProto, kl_ctrl, kl_penalty="kl", gather=None):
    """token_level_rewards = scores - kl_coef * kl(old, ref)  (ray_trainer.py:125-145; only when use_kl_loss is off).
    gather: callable(list) -> the list concatenated over the ranks, so the controller sees the GLOBAL batch's mean KL and size."""
    scores, mask = data.batch["token_level_scores"], data.batch["response_mask"]
    if "ref_log_probs" in data.batch.keys():
        kld = core_algos.compute_kl(data.batch["old_log_probs"], data.batch["ref_log_probs"], kl_penalty) * mask
    else:
        kld = torch.zeros_like(mask, dtype=torch.float32)
    data.batch["token_level_rewards"] = scores - kl_ctrl.kl_coef * kld
    per_seq = core_algos.masked_mean(kld, mask, dim=-1)
    if gather is not None:
        per_seq = torch.tensor(gather(per_seq.tolist()), dtype=per_seq.dtype)
    cur = torch.mean(per_seq, dim=0).item()
    metrics = {"critic/kl": cur, "critic/kl_coef": kl_ctrl.kl_coef}
    kl_ctrl.update(current_kl=cur, n_steps=int(per_seq.numel()))
    return data, metrics


class _RewardJob:
    """reward_fn(batch) on a thread, on a SNAPSHOT of the batch (own dicts, shared tensors): the main thread keeps adding keys to the
    live batch (old / ref log-probs) while the scorer reads the rollout's."""

    def __init__(self, fn, batch: DataProto):
        import threading
        snap = DataProto(batch=None if batch.batch is None else type(batch.batch)(dict(batch.batch.items()), batch_size=batch.batch.batch_size),
                         non_tensor_batch=dict(batch.non_tensor_batch), meta_info=dict(batch.meta_info))
        self._out, self._err, self.seconds = None, None, 0.0

        def run():
            t0 = time.perf_counter()
            try:
                self._out = fn(snap)
            except BaseException as e:                      # re-raised on the main thread
                self._err = e
            self.seconds = time.perf_counter() - t0
        self._t = threading.Thread(target=run, name="reward", daemon=True)
        self._t.start()

    def result(self):
        self._t.join()
        if self._err is not None:
            raise self._err
        return self._out


class RayPPOTrainer:
    """Name kept for drop-in use by verl.trainer.main; there is no Ray underneath."""

    def __init__(self, config, tokenizer, processor, worker_group, ref_worker_group, reward_fn, val_reward_fn, train_dataset, val_dataset=None):
        self.config, self.tokenizer, self.processor = config, tokenizer, processor
        self.actor_rollout_wg, self.ref_policy_wg = worker_group, ref_worker_group      # may be attached later (set_worker_groups)
        self.reward_fn, self.val_reward_fn = reward_fn, val_reward_fn
        self.world = int(os.environ.get("WORLD_SIZE", 1))
        self.rank = int(os.environ.get("RANK", 0))
        a = config.algorithm
        AdvantageEstimator(a.adv_estimator)
        self.use_critic = a.adv_estimator == "gae"               # ray_trainer.py:230-233
        self.critic_wg = None
        self.use_reference_policy = not a.disable_kl
        self.kl_ctrl = core_algos.get_kl_controller(a) if self.use_reference_policy else core_algos.FixedKLController(0.0)
        d, act = config.data, config.worker.actor
        # validation of ray_trainer.py:238-263
        if d.rollout_batch_size % act.global_batch_size != 0:
            raise ValueError("Rollout batch size must be divisible by global batch size.")
        if (d.rollout_batch_size * config.worker.rollout.n) % act.micro_batch_size_per_device_for_experience != 0:
            raise ValueError("Rollout batch size * rollout.n must be divisible by actor micro batch size for experience.")
        if self.use_critic:                                       # ray_trainer.py:248-257
            cr = config.worker.critic
            if d.rollout_batch_size % cr.global_batch_size != 0:
                raise ValueError("Rollout batch size must be divisible by critic global batch size.")
            if (d.rollout_batch_size * config.worker.rollout.n) % cr.micro_batch_size_per_device_for_experience != 0:
                raise ValueError("Rollout batch size * rollout.n must be divisible by critic micro batch size for experience.")
        if a.adv_estimator in ("grpo", "rloo") and config.worker.rollout.n == 1:
            raise ValueError("GRPO and RLOO algorithm need `config.worker.rollout.n > 1`.")
        if d.rollout_batch_size % self.world != 0:
            raise ValueError("rollout_batch_size must be divisible by the number of GPUs")
        self.local_prompts = d.rollout_batch_size // self.world
        from ..utils.dataset import collate_fn
        workers = int(os.environ.get("ST_DATALOADER_WORKERS", "0"))
        self.train_dataloader = ResumableDataLoader(train_dataset, batch_size=d.rollout_batch_size, shuffle=d.shuffle, seed=d.seed,
                                                    collate_fn=collate_fn, drop_last=True, num_workers=workers, rank=self.rank,
                                                    world_size=self.world)
        self.val_dataloader = None
        self._val_rows = 0
        if val_dataset is not None:
            vb = len(val_dataset) if d.val_batch_size == -1 else d.val_batch_size
            vb = -(-vb // self.world) * self.world             # pad_dataproto_to_divisor(world) of the reference (:384)
            self.val_dataloader = ResumableDataLoader(val_dataset, batch_size=vb, shuffle=False, collate_fn=collate_fn, drop_last=False,
                                                      num_workers=workers, rank=self.rank, world_size=self.world)
            self._val_rows = len(val_dataset)
        t = config.trainer
        self.training_steps = t.max_steps if t.max_steps is not None else len(self.train_dataloader) * t.total_episodes
        act.optim.training_steps = self.training_steps
        config.worker.critic.optim.training_steps = self.training_steps
        self.global_step = 0
        self.logger = Tracker(loggers=t.logger, config=config.to_dict())          # ray_trainer.py:566 of the reference

    def set_worker_groups(self, actor_rollout_wg, ref_policy_wg, critic_wg=None):
        self.actor_rollout_wg, self.ref_policy_wg, self.critic_wg = actor_rollout_wg, ref_policy_wg, critic_wg
        if self.use_critic and critic_wg is None:
            raise ValueError("adv_estimator=gae needs a critic worker group (FSDPWorker(config.worker, 'critic'))")

    def init_workers(self):
        if self.use_critic:                                       # ray_trainer.py:467-469: the critic first
            self.critic_wg.init_model()
        if self.use_reference_policy and self.ref_policy_wg is not self.actor_rollout_wg:
            self.ref_policy_wg.init_model()
        self.actor_rollout_wg.init_model()

    # ------------------------------------------------------------------------------------------------ cross-rank helpers
    def _gather(self, obj):
        """[obj of rank 0, obj of rank 1, ...] (DP_COMPUTE_PROTO's collect side, decorator.py:118-123, for small objects)."""
        wg = self.actor_rollout_wg
        if wg is not None and hasattr(wg, "gather_objects"):
            return wg.gather_objects(obj)
        return [obj]

    def _gather_list(self, xs: list) -> list:
        return [x for part in self._gather(list(xs)) for x in part]

    def _gather_metric_lists(self, m: Dict[str, Any]) -> Dict[str, list]:
        out: Dict[str, list] = defaultdict(list)
        local = {k: ([v] if np.isscalar(v) else [float(x) for x in np.asarray(v).reshape(-1)]) for k, v in m.items()}
        for part in self._gather(local):
            for k, v in part.items():
                out[k].extend(v)
        return out

    # ------------------------------------------------------------------------------------------------
    def _balance_batch(self, batch: DataProto, metrics: Dict[str, Any], logging_prefix: str = "global_seqlen", reorder: bool = True):
        """ray_trainer.py:526-541 in two parts.
        (1) Statistics: the reference partitions the GLOBAL list of sequence lengths into world_size sets with Karmarkar-Karp and
            logs min/max/minmax_diff (contiguous chunks) vs balanced_min/max (its partition): same list, same partitioner, same keys.
        (2) Reorder: the reference then permutes rows so DP rank r receives partition r.  Here the G rollouts of a prompt stay on
            the GPU that generated them (one prompt copy per group in every pass, prompt K/V re-used for the old log-probs), so
            the partitioner is applied where this design has a choice: each rank's rollout GROUPS are spread over its optimizer
            steps (mini-batches of global_batch_size_per_device rows) with equal group counts and balanced token sums — ranks then
            reach each gradient all-reduce after similar amounts of work.  Only update_actor's mini-batch split depends on that
            order, so fit() applies the permutation (returned here when reorder=False) AFTER the old / ref log-prob passes: those
            see the rows in generation order, which is what lets the old-policy pass re-use the rollout's prompt K/V cache
            (PolicyEngine._cache_matches; perf/prompt_cache_hit reports it).
        trainer.balance_mode=migrate is the reference-faithful alternative (_migrate_batch)."""
        lens = batch.batch["attention_mask"].sum(-1).tolist()
        glob = self._gather_list(lens)
        world = max(1, len(glob) // max(1, len(lens)))
        parts = get_seqlen_balanced_partitions(glob, k_partitions=world, equal_size=True)
        metrics.update(log_seqlen_unbalance(glob, parts, logging_prefix))
        n = self.config.worker.rollout.n
        mini = getattr(self.config.worker.actor, "global_batch_size_per_device", 0) or len(batch)
        n_mini = len(batch) // mini if (mini and len(batch) % mini == 0) else 1
        n_groups = len(batch) // n if n else 0
        uid = batch.non_tensor_batch.get("uid")
        grouped = uid is not None and n > 0 and len(batch) % n == 0 and all(uid[i] == uid[i - i % n] for i in range(len(batch)))
        if n_mini > 1 and grouped and mini % n == 0 and n_groups % n_mini == 0:
            gl = [sum(lens[g * n:(g + 1) * n]) for g in range(n_groups)]
            gparts = get_seqlen_balanced_partitions(gl, k_partitions=n_mini, equal_size=True)
            idx = torch.tensor([g * n + j for p in gparts for g in p for j in range(n)])
            sums = [sum(gl[g] for g in p) for p in gparts]
            metrics.update({"minibatch_seqlen/balanced_min": min(sums), "minibatch_seqlen/balanced_max": max(sums)})
            if reorder:
                batch.reorder(idx)
            return idx
        return None

    def _migrate_batch(self, batch: DataProto, metrics: Dict[str, Any], logging_prefix: str = "global_seqlen") -> DataProto:
        """trainer.balance_mode=migrate: the reference's _balance_batch to the letter (ray_trainer.py:526-541 +
        verl/utils/seqlen_balancing.py:150-181).  The whole rollout batch is assembled in global row order (rank-major = the order
        the single driver holds it in), partitioned into world_size equal-size sets by Karmarkar-Karp over the per-row token counts,
        reordered partition by partition, and rank r continues with chunk r — so every rank's micro-batch composition (hence its
        token-mean loss terms) is the reference's.  Costs what the reference pays: every row (with its pixel values) crosses the
        host fabric once per step, the rollouts of a prompt scatter over the ranks (no shared-prompt packing across them) and the
        rollout's prompt K/V cache no longer matches the rows (perf/prompt_cache_hit = 0).  Off by default."""
        parts_in = self._gather(batch)
        world = len(parts_in)
        whole = DataProto.concat(parts_in) if world > 1 else batch
        lens = whole.batch["attention_mask"].sum(-1).tolist()
        parts = get_seqlen_balanced_partitions(lens, k_partitions=world, equal_size=True)
        metrics.update(log_seqlen_unbalance(lens, parts, logging_prefix))
        whole.reorder(torch.tensor([j for p in parts for j in p]))
        return whole.chunk(world)[self.rank] if world > 1 else whole

    def _validate(self) -> Dict[str, Any]:
        """ray_trainer.py:358-411, data-parallel: every rank generates and scores ITS rows of each validation batch (in chunks
        that bound the generator's KV allocation), the per-sample scores and reward components are gathered, rank-padding rows are
        dropped, and the reference's metric names are returned: val/reward_score + val/{k}_reward."""
        if self.val_dataloader is None:
            return {}
        chunk = int(os.environ.get("ST_VAL_CHUNK", "256"))
        over = dict(self.config.worker.rollout.val_override_config)
        # rows per prompt the worker will return: the override's n, else the rollout's own n (the reference unions the un-repeated batch
        # and would stop with a batch-size error for n > 1 without the shipped config's `val_override_config: {n: 1}`)
        n = int(over.get("n", self.config.worker.rollout.n))
        scores_all: List[float] = []
        comp_all: Dict[str, list] = defaultdict(list)
        samples: List[tuple] = []
        seen = 0
        for batch_dict in self.val_dataloader:
            test = DataProto.from_single_dict(batch_dict)
            local_scores, local_comp = [], defaultdict(list)
            for lo in range(0, len(test), chunk):
                part = test[lo:lo + chunk]
                nt_keys = [k for k in ("raw_prompt_ids", "multi_modal_data", "multi_modal_inputs") if k in part.non_tensor_batch]
                gen = part.pop(batch_keys=["input_ids", "attention_mask", "position_ids"], non_tensor_batch_keys=nt_keys)
                gen.meta_info = dict(over)
                in_texts = [self.tokenizer.decode(ids, skip_special_tokens=True) for ids in gen.batch["input_ids"]]
                out = self.actor_rollout_wg.generate_sequences(gen)
                part = part.repeat(n, interleave=True).union(out) if n > 1 else part.union(out)
                reward, rmet = self.val_reward_fn(part)
                sc = reward.sum(-1).tolist()
                local_scores.extend(sc)
                for k, v in rmet.items():
                    local_comp[k].extend(v)
                if self.config.trainer.val_generations_to_log > 0:
                    outs = [self.tokenizer.decode(ids, skip_special_tokens=True) for ids in out.batch["responses"]]
                    samples.extend(zip(np.repeat(np.array(in_texts, dtype=object), n).tolist(), outs,
                                       part.non_tensor_batch["ground_truth"].tolist(), sc))
            # global row order = rank-major shards of the batch; the tail of the LAST batch may be cyclic rank padding
            per_rank = self._gather((local_scores, dict(local_comp)))
            n_glob = sum(len(p[0]) for p in per_rank)
            n_keep = min(n_glob, max(0, self._val_rows * n - seen))
            scores_all.extend([s for p in per_rank for s in p[0]][:n_keep])
            for k in per_rank[0][1]:
                comp_all[k].extend([v for p in per_rank for v in p[1][k]][:n_keep])
            seen += n_keep
        if self.config.trainer.val_generations_to_log > 0:
            samples = [s for part in self._gather(samples) for s in part]
            samples.sort(key=lambda x: x[0])
            np.random.RandomState(42).shuffle(samples)
            self.logger.log_generation(samples[: self.config.trainer.val_generations_to_log], self.global_step)
        out = {"val/reward_score": float(np.mean(scores_all)) if scores_all else 0.0}
        out.update({f"val/{k}_reward": v for k, v in reduce_metrics(comp_all).items()})
        return out

    # ------------------------------------------------------------------------------------------------ checkpoints
    def _save_checkpoint(self):
        """global_step_N/{actor/, dataloader.pt} + latest_global_step.txt, older steps pruned to save_limit (ray_trainer.py:483-506)."""
        root = self.config.trainer.save_checkpoint_path
        if self.rank == 0:
            remove_obsolete_ckpt(root, self.global_step, self.config.trainer.save_limit)
        path = os.path.join(root, f"global_step_{self.global_step}")
        os.makedirs(os.path.join(path, "actor"), exist_ok=True)
        self.actor_rollout_wg.save_checkpoint(os.path.join(path, "actor"))
        if self.use_critic:
            os.makedirs(os.path.join(path, "critic"), exist_ok=True)
            self.critic_wg.save_checkpoint(os.path.join(path, "critic"))
        if self.rank == 0:
            # dataloader.pt holds the loader's bare state_dict, as the reference writes it (ray_trainer.py:498-500); the adaptive KL
            # coefficient (driver state the reference loses on resume) rides along as one extra key
            state = dict(self.train_dataloader.state_dict(), kl_coef=self.kl_ctrl.kl_coef)
            torch.save(state, os.path.join(path, "dataloader.pt"))
            with open(os.path.join(root, CHECKPOINT_TRACKER), "w") as f:
                f.write(str(self.global_step))

    def _load_checkpoint(self):
        p = self.config.trainer.load_checkpoint_path
        if p is None:
            return
        if "global_step_" not in p.strip(os.path.sep).split(os.path.sep)[-1]:
            raise ValueError("`load_checkpoint_path` should end with `global_step_*`.")
        print(f"Load from checkpoint: {p}.")
        self.global_step = int(p.strip(os.path.sep).split("global_step_")[-1])
        self.actor_rollout_wg.load_checkpoint(os.path.join(p, "actor"))
        if self.use_critic:
            self.critic_wg.load_checkpoint(os.path.join(p, "critic"))
        dl = os.path.join(p, "dataloader.pt")
        if os.path.exists(dl):
            try:
                try:
                    st = torch.load(dl, weights_only=False)  # an unreadable file of THIS loader is an error, not a fresh start
                except ModuleNotFoundError as e:             # pickled objects of a package this image lacks (torchdata): another implementation's file
                    raise ForeignDataloaderState(str(e)) from e
                self.train_dataloader.load_state_dict(st["dataloader"] if isinstance(st, dict) and "dataloader" in st else st)   # round-2 files nested it
            except ForeignDataloaderState as e:   # ONLY a reference run's dataloader.pt (StatefulDataLoader snapshot): the data order restarts;
                # a state of this loader saved for another dataset size / rollout_batch_size stays a ValueError and stops the resume
                print(f"Dataloader state at {dl} is not usable here ({type(e).__name__}: {e}); the data order starts from scratch.")
            else:
                self.kl_ctrl.kl_coef = st.get("kl_coef", self.kl_ctrl.kl_coef)        # only after the loader state has loaded
        else:
            print(f"No dataloader state found at {dl}, will start from scratch.")

    # ------------------------------------------------------------------------------------------------
    def fit(self):
        cfg = self.config
        self._load_checkpoint()
        val_metrics: Optional[Dict[str, Any]] = None
        if self.val_reward_fn is not None and cfg.trainer.val_before_train and self.val_dataloader is not None:
            val_metrics = self._validate()
            self.logger.log(val_metrics, self.global_step)
            if cfg.trainer.val_only:
                return
        n = cfg.worker.rollout.n
        est = cfg.algorithm.adv_estimator
        done = False
        for _ in range(cfg.trainer.total_episodes):
            for batch_dict in self.train_dataloader:
                self.global_step += 1
                if self.global_step > self.training_steps:
                    done = True
                    break
                metrics, timing_raw = {}, {}
                batch = DataProto.from_single_dict(batch_dict)                 # this rank's rows of the global rollout batch
                nt_keys = [k for k in ("raw_prompt_ids", "multi_modal_data", "multi_modal_inputs") if k in batch.non_tensor_batch]
                gen_batch = batch.pop(batch_keys=["input_ids", "attention_mask", "position_ids"], non_tensor_batch_keys=nt_keys)
                if "synthetic_response_lengths" in batch.non_tensor_batch:       # benchmark datasets only (SyntheticSTVQADataset(response_lengths=...))
                    per_row = batch.non_tensor_batch.pop("synthetic_response_lengths")
                    gen_batch.meta_info["synthetic_response_lengths"] = np.concatenate([np.asarray(v, dtype=np.int64) for v in per_row])
                elif "synthetic_response_lengths" in batch.meta_info:
                    gen_batch.meta_info["synthetic_response_lengths"] = batch.meta_info["synthetic_response_lengths"]
                with _timer("step", timing_raw):
                    with _timer("gen", timing_raw):
                        gen_out = self.actor_rollout_wg.generate_sequences(gen_batch)
                    if est == "remax":                                        # greedy baseline rollout (ray_trainer.py:590-604)
                        with _timer("gen_max", timing_raw):
                            base_in = deepcopy(gen_batch)
                            base_in.meta_info.update({"temperature": 0, "n": 1})
                            base_out = self.actor_rollout_wg.generate_sequences(base_in)
                            batch = batch.union(base_out)
                            base_reward, _ = self.reward_fn(batch)
                            batch.pop(batch_keys=list(base_out.batch.keys()))
                            batch.batch["reward_baselines"] = base_reward.sum(-1)
                    batch.non_tensor_batch["uid"] = np.array([str(uuid.uuid4()) for _ in range(len(batch))], dtype=object)
                    batch = batch.repeat(repeat_times=n, interleave=True)
                    batch = batch.union(gen_out)
                    migrated = getattr(cfg.trainer, "balance_mode", "local") == "migrate"
                    # The reward is host work (detokenise + score, reward/custom.py:48-73) that needs nothing but the rollout, and the
                    # log-prob passes that follow are device work that needs nothing of the reward: the scorer runs on a thread beside
                    # them (round 5) and is joined in front of the advantages.  The reference runs them one after the other
                    # (ray_trainer.py:606-640); same values either way.  Not with balance_mode=migrate (the scores travel with the rows).
                    reward_job = None
                    if migrated or os.environ.get("ST_REWARD_THREAD", "1") == "0":
                        with _timer("reward", timing_raw):
                            reward_tensor, reward_metrics = self.reward_fn(batch)
                            batch.batch["token_level_scores"] = reward_tensor
                            metrics.update({f"reward/{k}": v for k, v in reduce_metrics(self._gather_metric_lists(reward_metrics)).items()})
                    else:
                        reward_job = _RewardJob(self.reward_fn, batch)
                    pending_order = None
                    if migrated:
                        batch = self._migrate_batch(batch, metrics)
                    else:
                        pending_order = self._balance_batch(batch, metrics, reorder=False)
                    batch.meta_info["global_token_num"] = torch.sum(batch.batch["attention_mask"], dim=-1).tolist()
                    with _timer("old", timing_raw):
                        old = self.actor_rollout_wg.compute_log_probs(batch)
                        hit = old.meta_info.pop("prompt_cache_hit", None)
                        if hit is not None:                               # did the old-policy pass run on the rollout's prompt K/V?
                            metrics["perf/prompt_cache_hit"] = float(np.mean(self._gather_list([float(hit)])))
                        batch = batch.union(old)
                    if self.use_reference_policy:
                        with _timer("ref", timing_raw):
                            batch = batch.union(self.ref_policy_wg.compute_ref_log_probs(batch))
                    if self.use_critic:                                   # ray_trainer.py:644-648
                        with _timer("values", timing_raw):
                            batch = batch.union(self.critic_wg.compute_values(batch))
                    if reward_job is not None:
                        with _timer("reward_wait", timing_raw):               # what the scorer still needed after the log-prob passes
                            reward_tensor, reward_metrics = reward_job.result()
                        timing_raw["reward"] = reward_job.seconds              # the scorer's own wall time (overlapped with old / ref)
                        batch.batch["token_level_scores"] = reward_tensor
                        metrics.update({f"reward/{k}": v for k, v in reduce_metrics(self._gather_metric_lists(reward_metrics)).items()})
                    with _timer("adv", timing_raw):
                        if not cfg.algorithm.use_kl_loss and self.use_reference_policy:
                            batch, kl_metrics = apply_kl_penalty(batch, self.kl_ctrl, cfg.algorithm.kl_penalty, gather=self._gather_list)
                            metrics.update(kl_metrics)
                        else:
                            batch.batch["token_level_rewards"] = batch.batch["token_level_scores"]
                        if migrated and self.world > 1:
                            # after the migration a rank holds PARTS of rollout groups: the group statistics are formed over the whole
                            # batch, as on the reference's driver (ray_trainer.py:667-672), and every rank keeps its rows' share
                            keys = [k for k in ("token_level_rewards", "response_mask", "values", "reward_baselines") if k in batch.batch.keys()]
                            whole = DataProto.concat(self._gather(batch.select(batch_keys=keys, non_tensor_batch_keys=["uid"])))
                            mine = compute_advantage(whole, est, cfg.algorithm.gamma, cfg.algorithm.lam, whole_batch=True).chunk(self.world)[self.rank]
                            batch.batch["advantages"], batch.batch["returns"] = mine.batch["advantages"].clone(), mine.batch["returns"].clone()
                        else:
                            batch = compute_advantage(batch, est, cfg.algorithm.gamma, cfg.algorithm.lam)
                    if pending_order is not None:                         # mini-batch balance: only update_actor's split depends on it
                        batch.reorder(pending_order)
                    if self.use_critic:                                   # ray_trainer.py:669-675
                        with _timer("update_critic", timing_raw):
                            critic_out = self.critic_wg.update_critic(batch)
                        metrics.update(reduce_metrics(self._gather_metric_lists(critic_out.non_tensor_batch)))
                    if cfg.trainer.critic_warmup <= self.global_step:
                        with _timer("update_actor", timing_raw):
                            actor_out = self.actor_rollout_wg.update_actor(batch)
                        metrics.update(reduce_metrics(self._gather_metric_lists(actor_out.non_tensor_batch)))
                    if self.val_reward_fn is not None and cfg.trainer.val_freq > 0 and self.global_step % cfg.trainer.val_freq == 0:
                        with _timer("validation", timing_raw):
                            val_metrics = self._validate()
                        metrics.update(val_metrics)
                    if cfg.trainer.save_freq > 0 and self.global_step % cfg.trainer.save_freq == 0:
                        with _timer("save_checkpoint", timing_raw):
                            self._save_checkpoint()
                # the driver's view (ray_trainer.py:697-703): data metrics over every rank's rows, the slowest rank's phase times,
                # the token count over all GPUs
                metrics.update(compute_data_metrics(batch, use_critic=self.use_critic, gather=self._gather_list))
                timing_all = self._gather(timing_raw)
                timing_max = {k: max(t[k] for t in timing_all if k in t) for k in timing_raw}
                batch.meta_info["global_token_num"] = self._gather_list(batch.meta_info["global_token_num"])
                n_resp = sum(self._gather_list([int(batch.batch["response_mask"].sum().item())]))
                metrics.update(compute_timing_metrics(batch, timing_max, n_response_tokens=n_resp))
                metrics.update(compute_throughout_metrics(batch, timing_max, n_gpus=self.world))
                metrics["perf/samples_per_s"] = len(batch) * self.world / timing_max["step"]
                self.logger.log(metrics, self.global_step)
            if done:
                break
        if self.val_reward_fn is not None and self.val_dataloader is not None:
            if val_metrics is None or cfg.trainer.val_freq <= 0 or self.global_step % cfg.trainer.val_freq != 0:
                val_metrics = self._validate()
                self.logger.log(val_metrics, self.global_step)
            if self.rank == 0:
                print("Final validation metrics: " + ", ".join(f"{k}: {v}" for k, v in val_metrics.items()), flush=True)
        # the reference always writes a final checkpoint (ray_trainer.py:718-719); ST_SKIP_FINAL_SAVE=1 is a test/bench knob for
        # multi-GB synthetic models whose final state nobody will read
        if (cfg.trainer.save_freq <= 0 or self.global_step % cfg.trainer.save_freq != 0) and os.environ.get("ST_SKIP_FINAL_SAVE") != "1":
            self._save_checkpoint()
