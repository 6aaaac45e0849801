"""Sharding-manager protocol of the reference's workers (verl/workers/sharding_manager/base.py): a context that is entered around a
worker method and may re-shard the data going in and coming out."""
from ...protocol import DataProto


class BaseShardingManager:
    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        return False

    def preprocess_data(self, data: DataProto) -> DataProto:
        return data

    def postprocess_data(self, data: DataProto) -> DataProto:
        return data
