"""`verl.utils.checkpoint.checkpoint_manager` — the tracker-file helpers under the reference's import path
(verl/utils/checkpoint/checkpoint_manager.py:31,110-160).  The worker-side save / load lives in fsdp_checkpoint_manager.py and
FSDPWorker.save_checkpoint / load_checkpoint; RayPPOTrainer uses these helpers for `global_step_N/` + `latest_global_step.txt`."""
from __future__ import annotations

import os
import random
import shutil
from abc import ABC, abstractmethod
from typing import Any, Dict, Optional

import numpy as np
import torch

CHECKPOINT_TRACKER = "latest_global_step.txt"

__all__ = ["BaseCheckpointManager", "CHECKPOINT_TRACKER", "find_latest_ckpt_path", "get_checkpoint_tracker_filename", "remove_obsolete_ckpt"]


class BaseCheckpointManager(ABC):
    """save / load of one role's training state under <step dir>/<role>/ (reference: checkpoint_manager.py:34-107).  `model` is what the
    concrete manager knows how to serialise (the reference: an FSDP module; here: an engine with a ParamStore)."""

    def __init__(self, model, optimizer=None, lr_scheduler=None, processing_class=None):
        self.model, self.optimizer, self.lr_scheduler, self.processing_class = model, optimizer, lr_scheduler, processing_class
        dist = torch.distributed
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    @abstractmethod
    def load_checkpoint(self, *args, **kwargs): ...

    @abstractmethod
    def save_checkpoint(self, *args, **kwargs): ...

    @staticmethod
    def local_mkdir(path: str) -> str:
        path = path if os.path.isabs(path) else os.path.join(os.getcwd(), path)
        os.makedirs(path, exist_ok=True)                     # concurrent ranks: exist_ok instead of the reference's file lock
        return path

    @staticmethod
    def get_rng_state() -> Dict[str, Any]:
        state = {"cpu": torch.get_rng_state(), "numpy": np.random.get_state(), "random": random.getstate()}
        if torch.cuda.is_available():
            state["cuda"] = torch.cuda.get_rng_state()
        return state

    @staticmethod
    def load_rng_state(rng_state: Dict[str, Any]) -> None:
        torch.set_rng_state(rng_state["cpu"])
        if "cuda" in rng_state and torch.cuda.is_available():
            torch.cuda.set_rng_state(rng_state["cuda"])
        np.random.set_state(rng_state["numpy"])
        random.setstate(rng_state["random"])


def get_checkpoint_tracker_filename(root_path: str) -> str:
    """the file that names the newest complete step directory"""
    return os.path.join(root_path, CHECKPOINT_TRACKER)


def find_latest_ckpt_path(path: Optional[str] = None, directory_format: str = "global_step_{}") -> Optional[str]:
    """`path/global_step_N` for the N in the tracker file; None when there is no tracker or the directory is gone."""
    if path is None:
        return None
    tracker = get_checkpoint_tracker_filename(path)
    if not os.path.exists(tracker):
        print(f"Checkpoint tracker file does not exist: {tracker}")
        return None
    with open(tracker, "rb") as f:
        step = int(f.read().decode())
    ckpt = os.path.join(path, directory_format.format(step))
    if not os.path.exists(ckpt):
        print(f"Checkpoint does not exist: {ckpt}")
        return None
    print(f"Found checkpoint: {ckpt}")
    return ckpt


def remove_obsolete_ckpt(path: str, global_step: int, save_limit: int = -1, directory_format: str = "global_step_{}"):
    """Keep the newest save_limit - 1 step directories older than `global_step` (the one being written is the save_limit-th)."""
    if save_limit <= 0 or not os.path.exists(path):
        return
    head, _, tail = directory_format.partition("{}")
    steps = []
    for name in os.listdir(path):
        mid = name[len(head):len(name) - len(tail)] if tail else name[len(head):]
        if name.startswith(head) and name.endswith(tail) and mid.isdigit() and int(mid) < global_step:
            steps.append(int(mid))
    for s in sorted(steps, reverse=True)[save_limit - 1:]:
        folder = os.path.join(path, directory_format.format(s))
        shutil.rmtree(folder, ignore_errors=True)
        print(f"Removed obsolete checkpoint: {folder}")
