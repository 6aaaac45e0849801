"""Configuration tree of `python3 -m verl.trainer.main` — same keys, defaults and derived fields as the reference's
dataclasses (verl/trainer/config.py:34-111, verl/workers/config.py:40-52, verl/workers/actor/config.py:22-103,
verl/workers/rollout/config.py:22-45, verl/workers/reward/config.py:21-25, verl/workers/critic/config.py:23-39), built
from ONE declarative schema below instead of hand-written classes, and merged without OmegaConf (absent offline):

    dataclass defaults  <-  config=<yaml>  <-  key=value dotlist        (verl/trainer/main.py:88-98)

Unknown keys are errors and values are coerced to the declared field type, as OmegaConf's structured mode does.
"""
from __future__ import annotations

import ast
import os
import typing
from dataclasses import asdict, field, fields, is_dataclass, make_dataclass
from typing import Any, Dict, Optional, Tuple

AUTO = object()          # marks "auto keys" (init=False in the reference): derived in post_init, not user-settable


def _build(name: str, spec: Dict[str, Any], methods: Optional[Dict[str, Any]] = None):
    cols = []
    for key, (tp, default) in spec.items():
        if isinstance(default, type) and is_dataclass(default):
            cols.append((key, tp, field(default_factory=default)))
        elif isinstance(default, (dict, list)):
            cols.append((key, tp, field(default_factory=lambda d=default: copy_of(d))))
        else:
            cols.append((key, tp, field(default=default)))
    return make_dataclass(name, cols, namespace=methods or {})


def copy_of(x):
    import copy
    return copy.deepcopy(x)


# ---- leaf groups --------------------------------------------------------------------------------------
def _model_post(self):
    if self.tokenizer_path is None:
        self.tokenizer_path = self.model_path


ModelConfig = _build("ModelConfig", {
    "model_path": (Optional[str], None), "tokenizer_path": (Optional[str], None), "override_config": (Dict[str, Any], {}),
    "enable_gradient_checkpointing": (bool, True), "trust_remote_code": (bool, True), "freeze_vision_tower": (bool, False),
}, {"post_init": _model_post})

OptimConfig = _build("OptimConfig", {
    "lr": (float, 1e-6), "betas": (Tuple[float, float], (0.9, 0.999)), "weight_decay": (float, 1e-2), "strategy": (str, "adamw"),
    "lr_warmup_ratio": (float, 0.0), "min_lr_ratio": (Optional[float], None), "warmup_style": (str, "constant"),
    "training_steps": (int, -1),
})

FSDPConfig = _build("FSDPConfig", {
    "enable_full_shard": (bool, True), "enable_cpu_offload": (bool, False), "enable_rank0_init": (bool, False),
    "use_orig_params": (bool, False), "torch_dtype": (Optional[str], None), "fsdp_size": (int, -1), "mp_param_dtype": (str, "bf16"),
    "mp_reduce_dtype": (str, "fp32"), "mp_buffer_dtype": (str, "fp32"),
})

OffloadConfig = _build("OffloadConfig", {"offload_params": (bool, False), "offload_optimizer": (bool, False)})

ActorConfig = _build("ActorConfig", {
    "strategy": (str, "fsdp"), "global_batch_size": (int, 256), "micro_batch_size_per_device_for_update": (int, 4),
    "micro_batch_size_per_device_for_experience": (int, 16), "max_grad_norm": (float, 1.0), "clip_ratio_low": (float, 0.2),
    "clip_ratio_high": (float, 0.3), "clip_ratio_dual": (float, 3.0), "ppo_epochs": (int, 1), "padding_free": (bool, False),
    "ulysses_sequence_parallel_size": (int, 1), "use_torch_compile": (bool, True), "model": (ModelConfig, ModelConfig),
    "optim": (OptimConfig, OptimConfig), "fsdp": (FSDPConfig, FSDPConfig), "offload": (OffloadConfig, OffloadConfig),
    "global_batch_size_per_device": (int, -1), "disable_kl": (bool, False), "use_kl_loss": (bool, False), "kl_penalty": (str, "kl"),
    "kl_coef": (float, 0.0),
})

RefConfig = _build("RefConfig", {
    "strategy": (str, "fsdp"), "fsdp": (FSDPConfig, FSDPConfig), "offload": (OffloadConfig, OffloadConfig),
    "micro_batch_size_per_device_for_experience": (int, -1), "padding_free": (bool, False), "ulysses_sequence_parallel_size": (int, 1),
    "use_torch_compile": (bool, True),
})

CriticConfig = _build("CriticConfig", {
    "strategy": (str, "fsdp"), "global_batch_size": (int, 256), "micro_batch_size_per_device_for_update": (int, 4),
    "micro_batch_size_per_device_for_experience": (int, 16), "max_grad_norm": (float, 1.0), "cliprange_value": (float, 0.5),
    "ppo_epochs": (int, 1), "padding_free": (bool, False), "ulysses_sequence_parallel_size": (int, 1), "model": (ModelConfig, ModelConfig),
    "optim": (OptimConfig, OptimConfig), "fsdp": (FSDPConfig, FSDPConfig), "offload": (OffloadConfig, OffloadConfig),
    "global_batch_size_per_device": (int, -1),
})

RolloutConfig = _build("RolloutConfig", {
    "name": (str, "vllm"), "n": (int, 1), "temperature": (float, 1.0), "top_p": (float, 1.0), "top_k": (int, -1), "limit_images": (int, 0),
    "dtype": (str, "bf16"), "gpu_memory_utilization": (float, 0.6), "ignore_eos": (bool, False), "enforce_eager": (bool, False),
    # extension (not a reference key; default off = the reference's behaviour): the rollout records log pi_old of every token it samples and
    # compute_log_probs returns those instead of running the old-policy forward again (spatialthinker_amd.rollout emit_log_probs)
    "old_log_probs_from_rollout": (bool, False),
    "enable_chunked_prefill": (bool, False), "tensor_parallel_size": (int, 2), "max_num_batched_tokens": (int, 8192),
    "max_num_seqs": (int, 1024), "disable_log_stats": (bool, True), "val_override_config": (Dict[str, Any], {}),
    "prompt_length": (int, -1), "response_length": (int, -1),
}, {"to_dict": lambda self: asdict(self)})

RewardConfig = _build("RewardConfig", {"reward_type": (str, "function"), "score_function": (str, "math"), "skip_special_tokens": (bool, True)})


def _worker_post(self):
    self.ref.micro_batch_size_per_device_for_experience = self.actor.micro_batch_size_per_device_for_experience
    self.ref.padding_free = self.actor.padding_free
    self.ref.ulysses_sequence_parallel_size = self.actor.ulysses_sequence_parallel_size
    self.ref.use_torch_compile = self.actor.use_torch_compile


WorkerConfig = _build("WorkerConfig", {
    "hybrid_engine": (bool, True), "actor": (ActorConfig, ActorConfig), "critic": (CriticConfig, CriticConfig), "ref": (RefConfig, RefConfig),
    "reward": (RewardConfig, RewardConfig), "rollout": (RolloutConfig, RolloutConfig),
}, {"post_init": _worker_post})

DataConfig = _build("DataConfig", {
    "train_files": (str, ""), "val_files": (str, ""), "prompt_key": (str, "prompt"), "answer_key": (str, "answer"), "image_key": (str, "images"),
    "mixed_data": (bool, False), "text_only": (bool, False), "max_prompt_length": (int, 512), "max_response_length": (int, 512),
    "rollout_batch_size": (int, 512), "val_batch_size": (int, -1), "format_prompt": (Optional[str], None), "shuffle": (bool, True),
    "seed": (int, 1), "max_pixels": (int, 4194304), "min_pixels": (int, 262144),
})

AlgorithmConfig = _build("AlgorithmConfig", {
    "gamma": (float, 1.0), "lam": (float, 1.0), "adv_estimator": (str, "grpo"), "disable_kl": (bool, False), "use_kl_loss": (bool, False),
    "kl_penalty": (str, "kl"), "kl_coef": (float, 1e-3), "kl_type": (str, "fixed"), "kl_horizon": (float, 0.0), "kl_target": (float, 0.0),
})


def _trainer_post(self):
    if self.save_checkpoint_path is None:
        self.save_checkpoint_path = os.path.join("checkpoints", self.project_name, self.experiment_name)


TrainerConfig = _build("TrainerConfig", {
    "total_episodes": (int, 10), "max_steps": (Optional[int], None), "project_name": (str, "easy_r1"), "experiment_name": (str, "demo"),
    "logger": (Tuple[str, ...], ("console", "wandb")), "nnodes": (int, 1), "n_gpus_per_node": (int, 8), "critic_warmup": (int, 0),
    "val_freq": (int, -1), "val_before_train": (bool, True), "val_only": (bool, False), "val_generations_to_log": (int, 0),
    "save_freq": (int, -1), "save_limit": (int, -1), "save_checkpoint_path": (Optional[str], None), "load_checkpoint_path": (Optional[str], None),
    # extension (not a reference key): "local" keeps a prompt's rollouts on the GPU that generated them and balances each rank's
    # mini-batches; "migrate" reproduces the reference's cross-rank row migration (ray_trainer.py:526-541) — RayPPOTrainer._migrate_batch
    "balance_mode": (str, "local"),
}, {"post_init": _trainer_post})

AUTO_KEYS = {"worker.actor.global_batch_size_per_device", "worker.actor.disable_kl", "worker.actor.use_kl_loss", "worker.actor.kl_penalty",
             "worker.actor.kl_coef", "worker.actor.optim.training_steps", "worker.critic.global_batch_size_per_device",
             "worker.critic.optim.training_steps", "worker.ref.micro_batch_size_per_device_for_experience", "worker.ref.padding_free",
             "worker.ref.ulysses_sequence_parallel_size", "worker.ref.use_torch_compile", "worker.rollout.prompt_length",
             "worker.rollout.response_length"}


def _ppo_post(self):
    self.worker.rollout.prompt_length = self.data.max_prompt_length
    self.worker.rollout.response_length = self.data.max_response_length
    self.worker.actor.disable_kl = self.algorithm.disable_kl
    self.worker.actor.use_kl_loss = self.algorithm.use_kl_loss
    self.worker.actor.kl_penalty = self.algorithm.kl_penalty
    self.worker.actor.kl_coef = self.algorithm.kl_coef


def recursive_post_init(dataclass_obj):
    obj = dataclass_obj
    if hasattr(obj, "post_init"):
        obj.post_init()
    for f in fields(obj):
        child = getattr(obj, f.name)
        if is_dataclass(child):
            recursive_post_init(child)


PPOConfig = _build("PPOConfig", {
    "data": (DataConfig, DataConfig), "worker": (WorkerConfig, WorkerConfig), "algorithm": (AlgorithmConfig, AlgorithmConfig),
    "trainer": (TrainerConfig, TrainerConfig),
}, {"post_init": _ppo_post, "deep_post_init": lambda self: recursive_post_init(self), "to_dict": lambda self: asdict(self)})


# ---- merging (OmegaConf structured-mode semantics) ---------------------------------------------------------
def _coerce(value: Any, tp: Any, path: str):
    origin = typing.get_origin(tp)
    args = typing.get_args(tp)
    if origin is typing.Union:                                   # Optional[T]
        if value is None or (isinstance(value, str) and value.lower() in ("null", "none", "~")):
            return None
        inner = [a for a in args if a is not type(None)]
        return _coerce(value, inner[0], path)
    if tp is bool:
        if isinstance(value, bool):
            return value
        if isinstance(value, str) and value.lower() in ("true", "false", "yes", "no", "on", "off", "1", "0"):
            return value.lower() in ("true", "yes", "on", "1")
        raise ValueError(f"{path}: cannot interpret {value!r} as bool")
    if tp is int:
        if isinstance(value, bool):
            raise ValueError(f"{path}: bool given for int")
        if isinstance(value, float) and value != int(value):
            raise ValueError(f"{path}: {value!r} is not an int")
        return int(value)
    if tp is float:
        return float(value)
    if tp is str:
        return value if isinstance(value, str) else str(value)
    if origin in (tuple, Tuple):
        if isinstance(value, str):
            value = ast.literal_eval(value)
        if not isinstance(value, (list, tuple)):
            raise ValueError(f"{path}: expected a list, got {value!r}")
        elem = args[0] if args else str
        return tuple(_coerce(v, elem if elem is not Ellipsis else str, path) for v in value)
    if origin in (dict, Dict):
        if isinstance(value, str):
            value = ast.literal_eval(value)
        if not isinstance(value, dict):
            raise ValueError(f"{path}: expected a mapping, got {value!r}")
        return dict(value)
    return value


def merge_into(cfg, updates: Dict[str, Any], prefix: str = "", from_cli: bool = False):
    """Apply a nested dict onto a config dataclass; unknown keys raise KeyError (structured mode)."""
    names = {f.name: f for f in fields(cfg)}
    for key, val in updates.items():
        path = f"{prefix}{key}"
        if key not in names:
            raise KeyError(f"Key '{path}' is not in the config schema")
        cur = getattr(cfg, key)
        if is_dataclass(cur):
            if not isinstance(val, dict):
                raise ValueError(f"{path}: expected a mapping")
            merge_into(cur, val, path + ".", from_cli)
        else:
            setattr(cfg, key, _coerce(val, names[key].type, path))
    return cfg


def _parse_scalar(text: str):
    """YAML-style scalar of a `key=value` CLI override (what OmegaConf.from_cli does)."""
    import yaml
    try:
        val = yaml.safe_load(text)
    except Exception:
        return text
    return text if isinstance(val, str) else val        # keep strings verbatim (multi-line format_prompt must not be folded)


def parse_dotlist(argv) -> Dict[str, Any]:
    out: Dict[str, Any] = {}
    for arg in argv:
        if "=" not in arg:
            raise ValueError(f"expected key=value, got {arg!r}")
        key, val = arg.split("=", 1)
        node = out
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _parse_scalar(val) if val != "" else ""
    return out


def load_config(argv) -> "PPOConfig":
    """defaults <- yaml (config=path) <- dotlist, exactly the precedence of verl/trainer/main.py:88-98."""
    import yaml
    cli = parse_dotlist(argv)
    cfg = PPOConfig()
    path = cli.pop("config", None)
    if path is not None:
        with open(path) as f:
            merge_into(cfg, yaml.safe_load(f) or {})
    merge_into(cfg, cli, from_cli=True)
    return cfg
