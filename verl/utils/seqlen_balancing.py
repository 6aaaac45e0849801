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
