"""The reference's `verl.single_controller.base` package surface (verl/single_controller/base/__init__.py:15-19): third-party worker code
does `from verl.single_controller.base import Worker` and `from verl.single_controller.base.decorator import Dispatch, register`
(verl/workers/fsdp_workers.py:41-42)."""
from .worker import Worker
from .worker_group import ClassWithInitArgs, ResourcePool, WorkerGroup

__all__ = ["ClassWithInitArgs", "ResourcePool", "Worker", "WorkerGroup"]
