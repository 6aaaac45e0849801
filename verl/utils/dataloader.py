"""Resumable, rank-sharded prompt loader — what `torchdata.stateful_dataloader.StatefulDataLoader` is to the reference
(verl/trainer/ray_trainer.py:267-299: RandomSampler(generator seeded with data.seed), batch_size = rollout_batch_size,
drop_last, `state_dict()` written to `dataloader.pt` by _save_checkpoint :498-500 and restored by _load_checkpoint :518-523).

Order: ONE torch.Generator seeded with `seed`; every epoch draws `torch.randperm(len(dataset), generator=g)` — the sequence
RandomSampler produces — and cuts it into global batches of `batch_size` (the tail is dropped).  One process per GPU: rank r
materialises only rows [r*B/W, (r+1)*B/W) of every global batch (Dispatch.DP_COMPUTE_PROTO's chunk(world)[rank],
decorator.py:106-108) — the other ranks' images are never decoded here.  State = (generator state at the start of the running
epoch, epochs finished, batches handed out in the running epoch): loading it continues with the very next batch."""
from __future__ import annotations

from typing import Any, Callable, Dict, Iterator, List, Optional

import torch
from torch.utils.data import DataLoader, Dataset


class ForeignDataloaderState(ValueError):
    """A dataloader.pt that another implementation wrote (the reference's torchdata StatefulDataLoader snapshot): the ONE resume error the
    trainer tolerates (data order restarts); a state of THIS loader saved for other sizes stays a plain ValueError and stops the run."""


class ResumableDataLoader:
    def __init__(self, dataset: Dataset, batch_size: int, shuffle: bool = True, seed: int = 1, collate_fn: Optional[Callable] = None,
                 drop_last: bool = True, num_workers: int = 0, rank: int = 0, world_size: int = 1):
        if batch_size % world_size:
            raise ValueError("rollout_batch_size must be divisible by the number of GPUs")
        self.dataset, self.batch_size, self.shuffle, self.collate_fn = dataset, batch_size, shuffle, collate_fn
        self.drop_last, self.num_workers, self.rank, self.world = drop_last, num_workers, rank, world_size
        self.gen = torch.Generator().manual_seed(seed)
        self._epoch_start_state = self.gen.get_state()
        self.epochs_done = 0
        self.batches_yielded = 0          # inside the running epoch
        self._resume_skip = 0

    def __len__(self) -> int:
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def _epoch_batches(self) -> List[List[int]]:
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.gen).tolist() if self.shuffle else list(range(n))
        out = [order[o:o + self.batch_size] for o in range(0, n, self.batch_size)]
        if out and len(out[-1]) < self.batch_size and self.drop_last:
            out.pop()
        return out

    def local_rows(self, global_batch: List[int]) -> List[int]:
        if len(global_batch) % self.world:          # a ragged last batch (drop_last=False): pad by cycling until the divisor is met,
            need = self.world - len(global_batch) % self.world       # as pad_dataproto_to_divisor (verl/protocol.py:48-66) — a batch of
            reps = -(-need // len(global_batch))                      # fewer than world/2 rows needs more than one lap
            global_batch = global_batch + (global_batch * reps)[:need]
        per = len(global_batch) // self.world
        return global_batch[self.rank * per:(self.rank + 1) * per]

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        self._epoch_start_state = self.gen.get_state()
        batches = self._epoch_batches()
        skip, self._resume_skip = self._resume_skip, 0
        self.batches_yielded = skip
        shards = [self.local_rows(b) for b in batches[skip:]]
        loader = DataLoader(self.dataset, batch_sampler=shards, num_workers=self.num_workers, collate_fn=self.collate_fn)
        for item in loader:
            self.batches_yielded += 1
            yield item
        self.epochs_done += 1
        self.batches_yielded = 0
        self._epoch_start_state = self.gen.get_state()     # a state saved between epochs resumes with a NEW permutation, not a replay

    # ---------------------------------------------------------------- checkpoint interface (StatefulDataLoader's names)
    def state_dict(self) -> Dict[str, Any]:
        return {"generator_state": self._epoch_start_state.clone(), "epochs_done": self.epochs_done,
                "batches_yielded": self.batches_yielded, "batch_size": self.batch_size, "num_rows": len(self.dataset)}

    def load_state_dict(self, state: Dict[str, Any]) -> None:
        if not isinstance(state, dict) or "generator_state" not in state:
            # e.g. the reference's dataloader.pt: a torchdata StatefulDataLoader snapshot (worker / sampler-iterator internals)
            raise ForeignDataloaderState("not a state of this loader (no `generator_state`): written by another dataloader implementation")
        if state.get("batch_size", self.batch_size) != self.batch_size or state.get("num_rows", len(self.dataset)) != len(self.dataset):
            raise ValueError("dataloader state was saved for a different dataset size or rollout_batch_size")
        self.gen.set_state(state["generator_state"])
        self.epochs_done = int(state["epochs_done"])
        self._resume_skip = int(state["batches_yielded"])
        if self._resume_skip >= len(self):            # saved exactly at an epoch end: start the next epoch
            self._epoch_batches()                     # advances the generator past the finished epoch
            self._resume_skip = 0
            self.epochs_done += 1
