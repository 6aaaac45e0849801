"""Data re-sharding around a worker method under Ulysses sequence parallelism (reference
verl/workers/sharding_manager/fsdp_ulysses.py:26-68).

The driver deals the batch over ALL ranks (Dispatch.DP_COMPUTE_PROTO).  The `sp` ranks of a sequence-parallel group must see the SAME rows
— each computes a slice of every sequence — so the rows of the group are all-gathered on the way in (`preprocess_data`) and every rank
keeps its own chunk of the result on the way out (`postprocess_data`).  Inside the context the group is the process-wide Ulysses group
(verl.utils.ulysses.set_ulysses_sequence_parallel_group), which the engine's attention all-to-all reads.

`device_mesh`: anything with `mesh["sp"].get_group() / .size() / .get_local_rank()` (a torch DeviceMesh with an "sp" dimension), or None
(no sequence parallelism: everything is the identity)."""
from ...protocol import DataProto, all_gather_data_proto
from ...utils.ulysses import get_ulysses_sequence_parallel_group, set_ulysses_sequence_parallel_group
from .base import BaseShardingManager


class FSDPUlyssesShardingManager(BaseShardingManager):
    def __init__(self, device_mesh):
        self.device_mesh = device_mesh
        self.prev_sp_group = None

    def _sp(self):
        return self.device_mesh["sp"]

    def __enter__(self):
        if self.device_mesh is not None:
            self.prev_sp_group = get_ulysses_sequence_parallel_group()
            set_ulysses_sequence_parallel_group(self._sp().get_group())
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        if self.device_mesh is not None:
            set_ulysses_sequence_parallel_group(self.prev_sp_group)
        return False

    def preprocess_data(self, data: DataProto) -> DataProto:
        """The gathered rows go into a NEW DataProto: `all_gather_data_proto` works in place (as the reference's does), and the
        reference is only safe with that because Ray hands the worker its own copy of the driver's batch.  Here the worker is called
        on the trainer's own object (SPMDWorkerGroup), which must keep its N rows for the `union` that follows."""
        if self.device_mesh is not None:
            data = DataProto(batch=data.batch, non_tensor_batch=dict(data.non_tensor_batch), meta_info=dict(data.meta_info))
            all_gather_data_proto(data, size=self._sp().size(), group=self._sp().get_group())
        return data

    def postprocess_data(self, data: DataProto) -> DataProto:
        if self.device_mesh is not None:
            data = data.chunk(chunks=self._sp().size())[self._sp().get_local_rank()]
        return data
