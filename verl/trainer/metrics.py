"""Step metrics with the reference's keys (verl/trainer/metrics.py:23-120)."""
from typing import Any, Dict, List

import numpy as np
import torch

from ..protocol import DataProto


def reduce_metrics(metrics: Dict[str, List[Any]]) -> Dict[str, Any]:
    return {k: float(np.mean(v)) for k, v in metrics.items()}


def _stats(prefix: str, t: torch.Tensor) -> Dict[str, float]:
    t = t.float()
    return {f"{prefix}/mean": t.mean().item(), f"{prefix}/max": t.max().item(), f"{prefix}/min": t.min().item()}


def compute_data_metrics(batch: DataProto, use_critic: bool = False, gather=None) -> Dict[str, Any]:
    """metrics.py:27-94.  gather(list) -> the list concatenated over the data-parallel ranks: the statistics then cover the
    global batch, as the reference's driver computes them (per-sample / per-valid-token scalars only are exchanged)."""
    b = batch.batch
    R = b["responses"].size(-1)
    mask = b["attention_mask"]
    prompt_mask, resp_mask = mask[:, :-R].bool(), mask[:, -R:].bool()
    g = (lambda t: t) if gather is None else (lambda t: torch.tensor(gather(t.tolist()), dtype=t.dtype))
    plen, rlen = g(prompt_mask.sum(-1).float()), g(resp_mask.sum(-1).float())
    out: Dict[str, Any] = {}
    out.update(_stats("critic/score", g(b["token_level_scores"].sum(-1))))
    out.update(_stats("critic/rewards", g(b["token_level_rewards"].sum(-1))))
    out.update(_stats("critic/advantages", g(torch.masked_select(b["advantages"], resp_mask))))
    valid_returns = g(torch.masked_select(b["returns"], resp_mask))
    out.update(_stats("critic/returns", valid_returns))
    if use_critic:                                             # metrics.py:46-50,69-80
        valid_values = g(torch.masked_select(b["values"], resp_mask))
        out.update(_stats("critic/values", valid_values))
        out["critic/vf_explained_var"] = (1.0 - torch.var(valid_returns - valid_values) / (torch.var(valid_returns) + 1e-5)).item()
    out.update(_stats("response_length", rlen))
    out["response_length/clip_ratio"] = (rlen == R).float().mean().item()
    out.update(_stats("prompt_length", plen))
    out["prompt_length/clip_ratio"] = (plen == prompt_mask.size(-1)).float().mean().item()
    return out


def compute_timing_metrics(batch: DataProto, timing_raw: Dict[str, float], n_response_tokens=None) -> Dict[str, Any]:
    n_resp = int(batch.batch["response_mask"].sum().item()) if n_response_tokens is None else int(n_response_tokens)
    n_all = sum(batch.meta_info["global_token_num"])
    per = {**dict.fromkeys(["gen", "reward"], n_resp), **dict.fromkeys(["ref", "old", "values", "adv", "update_critic", "update_actor"], n_all)}
    out = {f"timing_s/{k}": v for k, v in timing_raw.items()}
    out.update({f"timing_per_token_ms/{k}": timing_raw[k] * 1000 / per[k] for k in per.keys() & timing_raw.keys()})
    return out


def compute_throughout_metrics(batch: DataProto, timing_raw: Dict[str, float], n_gpus: int) -> Dict[str, Any]:
    total = sum(batch.meta_info["global_token_num"])
    t = timing_raw["step"]
    return {"perf/total_num_tokens": total, "perf/time_per_step": t, "perf/throughput": total / (t * n_gpus)}
