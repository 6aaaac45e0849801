"""`Worker`: the base class the reference's workers derive from (verl/single_controller/base/worker.py:85-198).

The reference's Worker is configured by a Ray register center (master address / port discovery, WG_PREFIX actors).  Here a worker is one
torchrun rank: the same public attributes (`rank`, `world_size`, `get_master_addr_port`, `get_cuda_visible_devices`, `print_rank0`,
`execute_with_func_generator`, `execute_func_rank_zero`) are fed from the torchrun environment, nothing is registered anywhere."""
from __future__ import annotations

import os

from ..decorator import Dispatch, Execute, register


class Worker:
    """A (distributed) worker = one SPMD rank."""

    def __init__(self, cuda_visible_devices=None) -> None:
        self._world_size = int(os.getenv("WORLD_SIZE", "1"))
        self._rank = int(os.getenv("RANK", "0"))
        self._local_world_size = int(os.getenv("LOCAL_WORLD_SIZE", "1"))
        self._local_rank = int(os.getenv("LOCAL_RANK", "0"))
        self._master_addr = os.getenv("MASTER_ADDR")
        self._master_port = os.getenv("MASTER_PORT")
        if cuda_visible_devices is not None:
            self._cuda_visible_devices = cuda_visible_devices

    def get_master_addr_port(self):
        return self._master_addr, self._master_port

    def get_cuda_visible_devices(self):
        return os.getenv("CUDA_VISIBLE_DEVICES", os.getenv("HIP_VISIBLE_DEVICES", os.getenv("ROCR_VISIBLE_DEVICES", "not set")))

    def print_rank0(self, *args, **kwargs):
        if self.rank == 0:
            print(*args, **kwargs)

    # properties in the reference; settable here because FSDPWorker assigns them from the environment itself
    @property
    def world_size(self):
        return self._world_size

    @world_size.setter
    def world_size(self, v):
        self._world_size = int(v)

    @property
    def rank(self):
        return self._rank

    @rank.setter
    def rank(self, v):
        self._rank = int(v)

    @register(dispatch_mode=Dispatch.DP_COMPUTE_PROTO_WITH_FUNC)
    def execute_with_func_generator(self, func, *args, **kwargs):
        return func(self, *args, **kwargs)

    @register(dispatch_mode=Dispatch.ALL_TO_ALL, execute_mode=Execute.RANK_ZERO)
    def execute_func_rank_zero(self, func, *args, **kwargs):
        return func(*args, **kwargs)
