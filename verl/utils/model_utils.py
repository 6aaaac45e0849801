"""`verl.utils.model_utils` — rank / memory / size print helpers (reference: verl/utils/model_utils.py:27-75).  `print_model_size` accepts a
torch module or this build's ParamStore (anything with `.flat` holding all parameters)."""
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def is_rank0() -> bool:
    return (not dist.is_initialized()) or dist.get_rank() == 0


def print_gpu_memory_usage(prefix: str = "GPU memory usage") -> None:
    if is_rank0() and torch.cuda.is_available():
        free, total = torch.cuda.mem_get_info()
        print(f"{prefix}: {(total - free) / 1024 ** 3:.2f} GB / {total / 1024 ** 3:.2f} GB.")


def _get_model_size(model, scale: str = "auto") -> Tuple[float, str]:
    n = float(model.flat.numel()) if hasattr(model, "flat") else float(sum(p.numel() for p in model.parameters()))
    if scale == "auto":
        scale = "B" if n > 1e9 else "M" if n > 1e6 else "K" if n > 1e3 else ""
    div = {"B": 1e9, "M": 1e6, "K": 1e3, "": 1.0}
    if scale not in div:
        raise NotImplementedError(f"Unknown scale {scale}.")
    return n / div[scale], scale


def print_model_size(model, name: Optional[str] = None) -> None:
    if is_rank0():
        n, scale = _get_model_size(model)
        print(f"{name or model.__class__.__name__} contains {n:.2f}{scale} parameters.")
