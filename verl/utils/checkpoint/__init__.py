"""`verl.utils.checkpoint` — the reference's package surface (CHECKPOINT_TRACKER, remove_obsolete_ckpt; verl/utils/checkpoint/__init__.py)
plus this build's readers / writers of the reference's on-disk layout."""
from .checkpoint_manager import CHECKPOINT_TRACKER, BaseCheckpointManager, find_latest_ckpt_path, remove_obsolete_ckpt
from .fsdp_checkpoint_manager import (FSDPCheckpointManager, export_reference_layout, find_reference_world_size, load_reference_checkpoint,  # noqa: F401
                                      read_reference_shards)

__all__ = ["BaseCheckpointManager", "CHECKPOINT_TRACKER", "FSDPCheckpointManager", "export_reference_layout", "find_latest_ckpt_path", "find_reference_world_size", "load_reference_checkpoint",
           "read_reference_shards", "remove_obsolete_ckpt"]
