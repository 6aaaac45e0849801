"""Sequence-length balancing across DP ranks (reference: verl/utils/seqlen_balancing.py:24-215).

`get_seqlen_balanced_partitions(lens, k, equal_size=True)` = Karmarkar-Karp largest differencing with the
reference's exact tie-breaking, so the same samples land on the same rank."""
from __future__ import annotations

import heapq
from typing import Dict, List


def _set_order(s):
    total, items = s
    return (total, len(items), items)


class _KKState:
    __slots__ = ("sets",)

    def __init__(self, sets):
        self.sets = sorted(sets, key=_set_order, reverse=True)

    @property
    def spread(self) -> int:
        return self.sets[0][0] - self.sets[-1][0]

    def absorb(self, other: "_KKState") -> "_KKState":
        k = len(self.sets)
        return _KKState([(self.sets[i][0] + other.sets[k - 1 - i][0], self.sets[i][1] + other.sets[k - 1 - i][1]) for i in range(k)])

    def __lt__(self, other: "_KKState") -> bool:          # heapq pops the LARGEST spread first; then the larger leading set
        if self.spread != other.spread:
            return self.spread > other.spread
        return _set_order(self.sets[0]) > _set_order(other.sets[0])


def karmarkar_karp(seqlen_list: List[int], k_partitions: int, equal_size: bool) -> List[List[int]]:
    order = sorted((length, idx) for idx, length in enumerate(seqlen_list))
    heap: List[_KKState] = []
    if equal_size:
        assert len(seqlen_list) % k_partitions == 0, f"{len(seqlen_list)} % {k_partitions} != 0"
        for off in range(0, len(order), k_partitions):
            heapq.heappush(heap, _KKState([(length, [(idx, length)]) for length, idx in order[off:off + k_partitions]]))
    else:
        for length, idx in order:
            heapq.heappush(heap, _KKState([(length, [(idx, length)])] + [(0, []) for _ in range(k_partitions - 1)]))
    while len(heap) > 1:
        first, second = heapq.heappop(heap), heapq.heappop(heap)
        heapq.heappush(heap, first.absorb(second))
    parts = [[idx for idx, _ in s[1]] for s in heap[0].sets]
    if equal_size:
        for p in parts:
            assert len(p) * k_partitions == len(seqlen_list)
    return parts


def get_seqlen_balanced_partitions(seqlen_list: List[int], k_partitions: int, equal_size: bool) -> List[List[int]]:
    assert len(seqlen_list) >= k_partitions, f"number of items:[{len(seqlen_list)}] < k_partitions:[{k_partitions}]"
    parts = karmarkar_karp(seqlen_list, k_partitions, equal_size)
    assert len(parts) == k_partitions and all(len(p) > 0 for p in parts)
    assert {i for p in parts for i in p} == set(range(len(seqlen_list)))
    return [sorted(p) for p in parts]


def log_seqlen_unbalance(seqlen_list: List[int], partitions: List[List[int]], prefix: str) -> Dict[str, float]:
    k = len(partitions)
    per = len(seqlen_list) // k
    naive = [sum(seqlen_list[o:o + per]) for o in range(0, len(seqlen_list), per)]
    balanced = [sum(seqlen_list[i] for i in p) for p in partitions]
    return {f"{prefix}/min": min(naive), f"{prefix}/max": max(naive), f"{prefix}/minmax_diff": max(naive) - min(naive),
            f"{prefix}/balanced_min": min(balanced), f"{prefix}/balanced_max": max(balanced), f"{prefix}/mean": sum(naive) / k}


def ceildiv(a, b):
    return -(a // -b)


def greedy_partition(seqlen_list: List[int], k_partitions: int, equal_size: bool) -> List[List[int]]:
    """Items in input order, each to the currently lightest partition (first of the lightest); with equal_size every item carries a bias
    larger than the total so that the counts even out first (seqlen_balancing.py:130-147)."""
    bias = sum(seqlen_list) + 1 if equal_size else 0
    parts: List[List[int]] = [[] for _ in range(k_partitions)]
    sums = [0] * k_partitions
    for i, n in enumerate(seqlen_list):
        j = min(range(k_partitions), key=lambda q: sums[q])        # min() keeps the first of equals, as the reference's scan does
        parts[j].append(i)
        sums[j] += n + bias
    if equal_size:
        for p_ in parts:
            assert len(p_) * k_partitions == len(seqlen_list), f"{len(p_)} * {k_partitions} != {len(seqlen_list)}"
    return parts


def rearrange_micro_batches(batch, max_token_len: int, dp_group=None):
    """Split a batch (TensorBatch / dict of tensors with an `attention_mask`) into micro-batches of at most max_token_len valid tokens with
    balanced token counts: ceil(total / max_token_len) of them (the maximum over the ranks of dp_group, so every rank runs the same
    number), rows assigned by the Karmarkar-Karp partition.  Returns (micro_batches, row indices of each) — seqlen_balancing.py:220-258.
    (This build's engines plan their packed passes themselves — spatialthinker_amd.actor; the function is kept for code written against it.)"""
    import torch
    import torch.distributed as dist
    mask = batch["attention_mask"]
    max_seq_len = mask.shape[-1]
    assert max_token_len >= max_seq_len, f"max_token_len must be greater than the sequence length. Got {max_token_len=} and {max_seq_len=}"
    eff = mask.sum(dim=1)
    n_micro = ceildiv(int(eff.sum().item()), max_token_len)
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([n_micro], device="cuda" if dist.get_backend(dp_group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=dp_group)
        n_micro = int(t.item())
    eff = eff.tolist()
    assert n_micro <= len(eff)
    idx = get_seqlen_balanced_partitions(eff, n_micro, equal_size=False)
    keys = list(batch.keys())
    micro = []
    for part in idx:
        rows = {k: torch.cat([batch[k][i:i + 1] for i in part]) for k in keys}
        micro.append(type(batch)(rows, batch_size=len(part)) if not isinstance(batch, dict) else rows)
    return micro, idx


def get_reverse_idx(idx_map: List[int]) -> List[int]:
    """the inverse permutation: out[idx_map[i]] = i (seqlen_balancing.py:261-267)"""
    out = list(idx_map)
    for i, j in enumerate(idx_map):
        out[j] = i
    return out
