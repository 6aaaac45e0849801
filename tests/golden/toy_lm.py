"""A toy causal LM (embedding + one-token look-back + tanh MLP) shared by the update-loop fixture generator (make_golden.py: it runs
INSIDE the reference's DataParallelPPOActor as `actor_module`) and by tests/test_oracle_update_loop.py (as the oracle loop's logp_fn).
Own code, pure torch; only its outputs are stored in tests/golden/update_loop.npz."""
from __future__ import annotations

import types

import numpy as np
import torch

V, H = 37, 16


def make_params(seed: int = 11) -> "dict[str, np.ndarray]":
    rs = np.random.RandomState(seed)
    return {"E": (0.5 * rs.standard_normal((V, H))).astype(np.float32), "W1": (0.4 * rs.standard_normal((H, H))).astype(np.float32),
            "b1": (0.1 * rs.standard_normal(H)).astype(np.float32), "W2": (0.6 * rs.standard_normal((V, H))).astype(np.float32)}


def logits_fn(p, input_ids: torch.Tensor) -> torch.Tensor:
    """(B, S) ids -> (B, S, V) logits; position t sees tokens t and t-1."""
    e = p["E"][input_ids]
    prev = torch.cat([torch.zeros_like(e[:, :1]), e[:, :-1]], 1)
    h = torch.tanh((e + 0.5 * prev) @ p["W1"].t() + p["b1"])
    return h @ p["W2"].t()


class ToyLM(torch.nn.Module):
    """nn.Module face of logits_fn with the call signature dp_actor.py:139-145 uses (padding_free = False)."""

    def __init__(self, params):
        super().__init__()
        self.E, self.W1, self.b1, self.W2 = (torch.nn.Parameter(torch.from_numpy(params[k]).clone()) for k in ("E", "W1", "b1", "W2"))

    def forward(self, input_ids, attention_mask=None, position_ids=None, use_cache=False):
        return types.SimpleNamespace(logits=logits_fn(dict(E=self.E, W1=self.W1, b1=self.b1, W2=self.W2), input_ids))


def make_data(seed: int = 5, N: int = 8, P: int = 5, R: int = 6):
    rs = np.random.RandomState(seed)
    ids = rs.randint(0, V, (N, P + R)).astype(np.int64)
    mask = np.ones((N, P + R), dtype=np.int64)
    for r in range(N):
        mask[r, P + int(rs.randint(2, R + 1)):] = 0
        mask[r, :int(rs.randint(0, 3))] = 0
    return ids, mask, P, R
