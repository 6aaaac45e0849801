"""SPMD worker group: the local stand-in for RayWorkerGroup (verl/single_controller/ray/base.py:75-405).

`wg.<method>(data)` calls the colocated worker of THIS rank on THIS rank's shard.  `world_size` is the number of
ranks (= GPUs); helpers gather small python objects (metrics) across ranks for the driver on rank 0."""
from __future__ import annotations

import torch.distributed as dist

from .decorator import MAGIC_ATTR


class SPMDWorkerGroup:
    def __init__(self, worker):
        self.worker = worker
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world_size > 1 else 0
        for name in dir(worker):
            fn = getattr(worker, name)
            if callable(fn) and hasattr(fn, MAGIC_ATTR):
                setattr(self, name, fn)

    def gather_objects(self, obj):
        if self.world_size == 1:
            return [obj]
        out = [None] * self.world_size
        dist.all_gather_object(out, obj)
        return out
