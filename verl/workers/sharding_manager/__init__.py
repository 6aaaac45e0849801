from .base import BaseShardingManager  # noqa: F401
from .fsdp_ulysses import FSDPUlyssesShardingManager  # noqa: F401

__all__ = ["BaseShardingManager", "FSDPUlyssesShardingManager"]
