from .fsdp_checkpoint_manager import (export_reference_layout, find_reference_world_size, load_reference_checkpoint,  # noqa: F401
                                      read_reference_shards)
