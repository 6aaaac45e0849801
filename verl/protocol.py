"""DataProto — the batch container every worker call takes and returns (API mirror of the reference's
verl/protocol.py:165-598, re-implemented without `tensordict`/`ray`).

`batch` is a TensorBatch (ordered dict of tensors sharing dim 0), `non_tensor_batch` a dict of numpy object
arrays with the same length, `meta_info` a free-form dict.  Semantics kept from the reference:
  * chunk(n) requires len % n == 0 and splits non-tensors with np.array_split (:497-518);
  * repeat(k, interleave=True) = repeat_interleave on tensors / np.repeat on arrays (:556-598);
  * union raises on conflicting keys whose values differ (:84-110);
  * reorder is in place; concat keeps the first meta_info (:521-545).
"""
from __future__ import annotations

import copy
import pickle
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, Iterable, List, Optional, Tuple, Union

import numpy as np
import torch

__all__ = ["DataProto", "TensorBatch", "pad_dataproto_to_divisor", "unpad_dataproto", "union_tensor_dict"]


class TensorBatch:
    """Minimal stand-in for tensordict.TensorDict with a 1-D batch size."""

    def __init__(self, source: Optional[Dict[str, torch.Tensor]] = None, batch_size: Optional[Union[int, Tuple[int, ...]]] = None):
        self._d: Dict[str, torch.Tensor] = dict(source or {})
        if batch_size is None:
            batch_size = next(iter(self._d.values())).shape[0] if self._d else 0
        if isinstance(batch_size, int):
            batch_size = (batch_size,)
        self.batch_size = torch.Size(batch_size)
        for k, v in self._d.items():
            assert v.shape[0] == self.batch_size[0], f"key {k}: dim0 {v.shape[0]} != batch size {self.batch_size[0]}"

    # mapping protocol
    def __getitem__(self, item):
        if isinstance(item, str):
            return self._d[item]
        sub = {k: v[item] for k, v in self._d.items()}
        if isinstance(item, int):
            return sub
        n = next(iter(sub.values())).shape[0] if sub else len(range(*item.indices(self.batch_size[0]))) if isinstance(item, slice) else 0
        return TensorBatch(sub, batch_size=n)

    def __setitem__(self, key: str, value: torch.Tensor):
        assert value.shape[0] == self.batch_size[0], f"key {key}: dim0 {value.shape[0]} != batch size {self.batch_size[0]}"
        self._d[key] = value

    def __contains__(self, key):
        return key in self._d

    def __len__(self):
        return self.batch_size[0]

    def keys(self):
        return self._d.keys()

    def values(self):
        return self._d.values()

    def items(self):
        return self._d.items()

    def pop(self, key: str):
        return self._d.pop(key)

    def select(self, *keys: str) -> "TensorBatch":
        return TensorBatch({k: self._d[k] for k in keys}, batch_size=self.batch_size)

    def to(self, device) -> "TensorBatch":
        return TensorBatch({k: v.to(device) for k, v in self._d.items()}, batch_size=self.batch_size)

    @property
    def device(self):
        return next(iter(self._d.values())).device if self._d else torch.device("cpu")

    def chunk(self, chunks: int, dim: int = 0) -> List["TensorBatch"]:
        n = self.batch_size[0] // chunks
        return [TensorBatch({k: v[i * n:(i + 1) * n] for k, v in self._d.items()}, batch_size=n) for i in range(chunks)]

    def rename_key_(self, old: str, new: str):
        self._d[new] = self._d.pop(old)

    def to_dict(self):
        return dict(self._d)

    @staticmethod
    def cat(batches: List["TensorBatch"]) -> "TensorBatch":
        keys = list(batches[0].keys())
        return TensorBatch({k: torch.cat([b[k] for b in batches], dim=0) for k in keys}, batch_size=sum(len(b) for b in batches))


def union_tensor_dict(tensor_dict1: TensorBatch, tensor_dict2: TensorBatch) -> TensorBatch:
    """tensor_dict1 updated with tensor_dict2 (same batch size; a key present in both must hold equal tensors) — protocol.py:84-97."""
    a, b = tensor_dict1, tensor_dict2
    if a.batch_size != b.batch_size:
        raise ValueError(f"Two tensor dict must have identical batch size. Got {a.batch_size} and {b.batch_size}")
    for key in b.keys():
        if key in a and not torch.equal(a[key].cpu(), b[key].cpu()):
            raise ValueError(f"Key already exists: {key}.")
        a[key] = b[key]
    return a


def union_numpy_dict(tensor_dict1: Dict[str, np.ndarray], tensor_dict2: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """the same for the non-tensor side (protocol.py:100-110)"""
    return _union_numpy(tensor_dict1, tensor_dict2)


def batch_collate(features: List[Dict[str, Any]]) -> Dict[str, List[Any]]:
    """list of per-sample dicts -> dict of lists (protocol.py:113-122)"""
    out: Dict[str, List[Any]] = {}
    for feature in features:
        for key, value in feature.items():
            out.setdefault(key, []).append(value)
    return out


def collate_fn(data_items: List["DataProtoItem"]) -> "DataProto":
    """DataProtoItems (what DataProto[i] returns) -> one DataProto: tensors stacked, the rest as object arrays (protocol.py:145-155)"""
    keys = list(data_items[0].batch.keys()) if data_items and data_items[0].batch is not None else []
    batch = TensorBatch({k: torch.stack([d.batch[k] for d in data_items]).contiguous() for k in keys}, batch_size=len(data_items)) if keys else None
    non = batch_collate([d.non_tensor_batch for d in data_items])
    return DataProto(batch=batch, non_tensor_batch={k: _object_array(v) for k, v in non.items()})


def _object_array(values: List[Any]) -> np.ndarray:
    out = np.empty(len(values), dtype=object)                # element-wise: np.array(list_of_equal_length_lists, dtype=object) would go 2-D
    for i, v in enumerate(values):
        out[i] = v
    return out


def fold_batch_dim(data: "DataProto", new_batch_size: int) -> "DataProto":
    """[bsz, ...] -> [new_bsz, bsz // new_bsz, ...] for every tensor and array (protocol.py:125-142)"""
    bsz = len(data)
    assert bsz % new_batch_size == 0
    batch = None
    if data.batch is not None:
        batch = TensorBatch({k: v.reshape(new_batch_size, -1, *v.shape[1:]) for k, v in data.batch.items()}, batch_size=new_batch_size)
    non = {k: np.reshape(v, (new_batch_size, -1, *v.shape[1:])) for k, v in data.non_tensor_batch.items()}
    return DataProto(batch=batch, non_tensor_batch=non, meta_info=data.meta_info)


def _union_numpy(a: Dict[str, np.ndarray], b: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    for key, val in b.items():
        if key in a:
            assert isinstance(val, np.ndarray) and isinstance(a[key], np.ndarray)
            same = len(a[key]) == len(val) and all(_eq(x, y) for x, y in zip(a[key], val))
            if not same:
                raise ValueError(f"Key already exists: {key}.")
        a[key] = val
    return a


def _eq(x, y) -> bool:
    if x is y:
        return True
    try:
        r = x == y
        return bool(r.all()) if hasattr(r, "all") else bool(r)
    except Exception:
        return False


def _union_meta(a: Dict[str, Any], b: Dict[str, Any]) -> Dict[str, Any]:
    for key, val in b.items():
        if key in a and a[key] != val:
            raise ValueError(f"{key} in meta_info are not the same: {a[key]} vs {val}")
        a[key] = val
    return a


@dataclass
class DataProtoItem:
    batch: Optional[Dict[str, torch.Tensor]] = None
    non_tensor_batch: Dict[str, Any] = field(default_factory=dict)
    meta_info: Dict[str, Any] = field(default_factory=dict)


@dataclass
class DataProto:
    batch: Optional[TensorBatch] = None
    non_tensor_batch: Dict[str, np.ndarray] = field(default_factory=dict)
    meta_info: Dict[str, Any] = field(default_factory=dict)

    def __post_init__(self):
        self.check_consistency()

    def __len__(self) -> int:
        if self.batch is not None:
            return self.batch.batch_size[0]
        if self.non_tensor_batch:
            return next(iter(self.non_tensor_batch.values())).shape[0]
        return 0

    def __getitem__(self, item):
        if isinstance(item, int):
            return DataProtoItem(batch=self.batch[item] if self.batch is not None else None,
                                 non_tensor_batch={k: v[item] for k, v in self.non_tensor_batch.items()}, meta_info=self.meta_info)
        return DataProto(batch=self.batch[item] if self.batch is not None else None,
                         non_tensor_batch={k: v[item] for k, v in self.non_tensor_batch.items()}, meta_info=self.meta_info)

    def print_size(self, prefix: str = "") -> None:
        """bytes held by the tensors and by the arrays (protocol.py:224-238)"""
        gb = 1024 ** 3
        t = sum(v.element_size() * v.numel() for v in self.batch.values()) / gb if self.batch is not None else 0.0
        n = sum(v.nbytes for v in self.non_tensor_batch.values()) / gb
        print(f"{prefix} Size of tensordict: {t} GB, size of non_tensor_batch: {n} GB.")

    def make_iterator(self, mini_batch_size: int, epochs: int, seed: Optional[int] = None, dataloader_kwargs: Optional[Dict[str, Any]] = None):
        """Iterator over mini-batches of this DataProto: `epochs` passes of a torch DataLoader over the rows (so `shuffle`, `drop_last` ...
        in dataloader_kwargs behave as in the reference, seeded by `seed`); every mini-batch carries this object's meta_info
        (protocol.py:447-486)."""
        assert len(self) % mini_batch_size == 0, f"{len(self)} % {mini_batch_size} != 0"
        from torch.utils.data import DataLoader
        kwargs = dict(dataloader_kwargs or {})
        assert isinstance(kwargs, dict)
        gen = None
        if seed is not None:
            gen = torch.Generator()
            gen.manual_seed(seed)
        loader = DataLoader(dataset=self, batch_size=mini_batch_size, collate_fn=collate_fn, generator=gen, **kwargs)

        def rows():
            for _ in range(epochs):
                for d in loader:
                    d.meta_info = self.meta_info
                    yield d
        return iter(rows())

    def check_consistency(self):
        if self.batch is not None:
            assert len(self.batch.batch_size) == 1, "only support num_batch_dims=1"
            n = self.batch.batch_size[0]
            for key, val in self.non_tensor_batch.items():
                assert isinstance(val, np.ndarray), f"non-tensor {key} must be a numpy array"
                assert len(val) == n, f"key {key} length {len(val)} is not equal to batch size {n}."

    # ---- construction -----------------------------------------------------------------------
    @classmethod
    def from_single_dict(cls, data: Dict[str, Union[torch.Tensor, np.ndarray]], meta_info: Optional[Dict[str, Any]] = None) -> "DataProto":
        tensors, others = {}, {}
        for key, val in data.items():
            if isinstance(val, torch.Tensor):
                tensors[key] = val
            elif isinstance(val, np.ndarray):
                others[key] = val
            else:
                raise ValueError(f"Unsupported type in data {type(val)}")
        return cls.from_dict(tensors=tensors, non_tensors=others, meta_info=meta_info)

    @classmethod
    def from_dict(cls, tensors: Dict[str, torch.Tensor], non_tensors: Optional[Dict[str, np.ndarray]] = None,
                  meta_info: Optional[Dict[str, Any]] = None, num_batch_dims: int = 1) -> "DataProto":
        assert len(tensors) > 0, "tensors must not be empty"
        assert num_batch_dims == 1, "only num_batch_dims=1 is supported"
        sizes = {k: v.shape[0] for k, v in tensors.items()}
        assert len(set(sizes.values())) == 1, f"Not all the tensor in tensors have the same batch size: {sizes}"
        return cls(batch=TensorBatch(tensors), non_tensor_batch=dict(non_tensors or {}), meta_info=dict(meta_info or {}))

    # ---- movement / persistence -------------------------------------------------------------
    def to(self, device) -> "DataProto":
        if self.batch is not None:
            self.batch = self.batch.to(device)
        return self

    def save_to_disk(self, filepath: str) -> None:
        with open(filepath, "wb") as f:
            pickle.dump(self, f)

    @staticmethod
    def load_from_disk(filepath: str) -> "DataProto":
        with open(filepath, "rb") as f:
            return pickle.load(f)

    # ---- key selection ----------------------------------------------------------------------
    def select(self, batch_keys: Optional[List[str]] = None, non_tensor_batch_keys: Optional[List[str]] = None,
               meta_info_keys: Optional[List[str]] = None, deepcopy: bool = False) -> "DataProto":
        sub = self.batch.select(*batch_keys) if batch_keys is not None else self.batch
        nt = {k: v for k, v in self.non_tensor_batch.items() if non_tensor_batch_keys is None or k in non_tensor_batch_keys}
        mi = {k: v for k, v in self.meta_info.items() if meta_info_keys is None or k in meta_info_keys}
        if deepcopy:
            nt, mi = copy.deepcopy(nt), copy.deepcopy(mi)
        return DataProto(batch=sub, non_tensor_batch=nt, meta_info=mi)

    def pop(self, batch_keys: Optional[List[str]] = None, non_tensor_batch_keys: Optional[List[str]] = None,
            meta_info_keys: Optional[List[str]] = None) -> "DataProto":
        assert batch_keys is not None
        tensors = {k: self.batch.pop(k) for k in batch_keys}
        others = {k: self.non_tensor_batch.pop(k) for k in (non_tensor_batch_keys or [])}
        meta = {k: self.meta_info.pop(k) for k in (meta_info_keys or [])}
        return DataProto.from_dict(tensors=tensors, non_tensors=others, meta_info=meta)

    def rename(self, old_keys=None, new_keys=None) -> "DataProto":
        old = [old_keys] if isinstance(old_keys, str) else list(old_keys)
        new = [new_keys] if isinstance(new_keys, str) else list(new_keys)
        if len(old) != len(new):
            raise ValueError(f"new_keys and old_keys must have the same length, but got {len(new)} and {len(old)}")
        for o, n in zip(old, new):
            self.batch.rename_key_(o, n)
        return self

    def union(self, other: "DataProto") -> "DataProto":
        self.batch = union_tensor_dict(self.batch, other.batch)
        self.non_tensor_batch = _union_numpy(self.non_tensor_batch, other.non_tensor_batch)
        self.meta_info = _union_meta(self.meta_info, other.meta_info)
        return self

    # ---- splitting / merging ----------------------------------------------------------------
    def chunk(self, chunks: int) -> List["DataProto"]:
        assert len(self) % chunks == 0, f"only support equal chunk. Got size of DataProto {len(self)} and chunk {chunks}."
        parts = self.batch.chunk(chunks) if self.batch is not None else [None] * chunks
        nts = [{} for _ in range(chunks)]
        for key, val in self.non_tensor_batch.items():
            for i, piece in enumerate(np.array_split(val, chunks)):
                nts[i][key] = piece
        return [DataProto(batch=parts[i], non_tensor_batch=nts[i], meta_info=self.meta_info) for i in range(chunks)]

    def split(self, split_size: int) -> List["DataProto"]:
        return self.chunk(len(self) // split_size)

    @staticmethod
    def concat(data: List["DataProto"]) -> "DataProto":
        new_batch = TensorBatch.cat([d.batch for d in data]) if data[0].batch is not None else None
        keys = list(data[0].non_tensor_batch.keys())
        nt = {k: np.concatenate([d.non_tensor_batch[k] for d in data], axis=0) for k in keys}
        return DataProto(batch=new_batch, non_tensor_batch=nt, meta_info=data[0].meta_info)

    def reorder(self, indices: torch.Tensor) -> None:
        idx_np = indices.detach().cpu().numpy()
        self.batch = self.batch[indices]
        self.non_tensor_batch = {k: v[idx_np] for k, v in self.non_tensor_batch.items()}

    def repeat(self, repeat_times: int = 2, interleave: bool = True) -> "DataProto":
        new_batch = None
        if self.batch is not None:
            if interleave:
                rep = {k: v.repeat_interleave(repeat_times, dim=0) for k, v in self.batch.items()}
            else:
                rep = {k: v.unsqueeze(0).expand(repeat_times, *v.shape).reshape(-1, *v.shape[1:]) for k, v in self.batch.items()}
            new_batch = TensorBatch(rep, batch_size=len(self.batch) * repeat_times)
        nt = {}
        for key, val in self.non_tensor_batch.items():
            nt[key] = np.repeat(val, repeat_times, axis=0) if interleave else np.tile(val, (repeat_times,) + (1,) * (val.ndim - 1))
        return DataProto(batch=new_batch, non_tensor_batch=nt, meta_info=self.meta_info)


def pad_dataproto_to_divisor(data: DataProto, size_divisor: int) -> Tuple[DataProto, int]:
    """Pad by cycling the leading rows until len % size_divisor == 0 (reference :48-73)."""
    assert isinstance(data, DataProto), "data must be a DataProto"
    rem = len(data) % size_divisor
    if rem == 0:
        return data, 0
    pad, pieces, left = size_divisor - rem, [data], size_divisor - rem
    while left > 0:
        take = min(left, len(data))
        pieces.append(data[:take])
        left -= take
    return DataProto.concat(pieces), pad


def unpad_dataproto(data: DataProto, pad_size: int) -> DataProto:
    return data[:-pad_size] if pad_size else data


def allgather_dict_tensors(tensors, size: int, group, dim: int = 0):
    """every tensor of a dict / TensorBatch all-gathered over `group` and concatenated along `dim`, keys in sorted order
    (protocol.py:651-678); a TensorBatch comes back as a TensorBatch of size x rows."""
    import torch.distributed as dist
    is_batch = isinstance(tensors, TensorBatch)
    src = tensors.to_dict() if is_batch else tensors
    out = {}
    for key in sorted(src.keys()):
        val = src[key].contiguous()
        parts = [torch.empty_like(val) for _ in range(size)]
        dist.all_gather(parts, val, group=group)
        out[key] = torch.cat(parts, dim=dim)
    return TensorBatch(out, batch_size=len(tensors) * size) if is_batch else out


def all_gather_data_proto(data: DataProto, size: int, group) -> None:
    """In place: every rank of `group` ends up with the rows of ALL its ranks, rank-major (reference verl/protocol.py:651-689) — the
    pre-processing step of Ulysses sequence parallelism: the sp ranks of a group must run the SAME rows (each computes a slice of every
    sequence).  Tensors travel through all_gather on their own device (gloo: host tensors; RCCL: device tensors), objects by
    all_gather_object."""
    import torch.distributed as dist
    if data.batch is not None:
        out = {}
        on_device = dist.get_backend(group) == "nccl"            # RCCL moves device tensors: host batches make the round trip the reference makes (:682-685)
        for key in sorted(data.batch.keys()):
            src = data.batch[key]
            t = (src.cuda() if (on_device and not src.is_cuda) else src).contiguous()
            parts = [torch.empty_like(t) for _ in range(size)]
            dist.all_gather(parts, t, group=group)
            out[key] = torch.cat(parts, dim=0).to(src.device)
        data.batch = TensorBatch(out, batch_size=len(data.batch) * size)
    gathered = [None] * size
    dist.all_gather_object(gathered, data.non_tensor_batch, group=group)
    data.non_tensor_batch = {k: np.concatenate([g[k] for g in gathered], axis=0) for k in data.non_tensor_batch}
