"""`ResourcePool`, `ClassWithInitArgs`, `WorkerGroup` under the reference's module path (verl/single_controller/base/worker_group.py:27-198).

Ray placement groups do not exist here: a ResourcePool only records the process counts a config asked for (its `world_size` is checked
against the torchrun world), ClassWithInitArgs is the same deferred constructor, and WorkerGroup is the SPMD group of
`verl/single_controller/worker_group.py` — `wg.method(data)` runs the colocated worker of this rank on this rank's shard."""
from __future__ import annotations

from typing import Any, List

from ..worker_group import SPMDWorkerGroup


class ResourcePool:
    def __init__(self, process_on_nodes=None, max_collocate_count: int = 10, n_gpus_per_node: int = 8) -> None:
        self._store = list(process_on_nodes or [])
        self.max_collocate_count = max_collocate_count
        self.n_gpus_per_node = n_gpus_per_node

    def add_node(self, process_count):
        self._store.append(process_count)

    @property
    def world_size(self):
        return sum(self._store)

    def __call__(self) -> Any:
        return self._store

    @property
    def store(self):
        return self._store

    def local_world_size_list(self) -> List[int]:
        return [n for n in self._store for _ in range(n)]

    def local_rank_list(self) -> List[int]:
        return [i for n in self._store for i in range(n)]


class ClassWithInitArgs:
    """A class constructor with the arguments to call it with later (on the worker's own process in the reference; here, in place)."""

    def __init__(self, cls, *args, **kwargs) -> None:
        self.cls, self.args, self.kwargs = cls, args, kwargs

    def __call__(self) -> Any:
        return self.cls(*self.args, **self.kwargs)


class WorkerGroup(SPMDWorkerGroup):
    """WorkerGroup(resource_pool, ray_cls_with_init=ClassWithInitArgs(...)) or WorkerGroup(worker=obj)."""

    def __init__(self, resource_pool: ResourcePool = None, ray_cls_with_init: ClassWithInitArgs = None, worker=None, **kwargs) -> None:
        if worker is None:
            if ray_cls_with_init is None:
                raise ValueError("WorkerGroup needs a worker or a ClassWithInitArgs to build one from")
            worker = ray_cls_with_init()
        super().__init__(worker)
        self.resource_pool = resource_pool
        if resource_pool is not None and resource_pool.world_size not in (0, self.world_size):
            raise ValueError(f"resource pool asks for {resource_pool.world_size} processes, torchrun started {self.world_size} ranks")
