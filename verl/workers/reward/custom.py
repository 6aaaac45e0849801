"""CustomRewardManager — decodes each response, scores it with the configured plug-in and writes the scalar at the last
valid response token (reference: verl/workers/reward/custom.py:33-73)."""
from __future__ import annotations

from collections import defaultdict
from typing import Callable, Dict, List, Tuple

import torch

from ...protocol import DataProto
from ...utils.reward_score import math_compute_score, r1v_compute_score, r1v_scene_compute_score, spatial_sgg_compute_score

_SCORERS: Dict[str, Callable] = {"math": math_compute_score, "r1v": r1v_compute_score, "r1v_scene": r1v_scene_compute_score, "spatial_sgg": spatial_sgg_compute_score}


class CustomRewardManager:
    def __init__(self, tokenizer, config):
        self.tokenizer, self.config = tokenizer, config
        if config.score_function not in _SCORERS:
            raise NotImplementedError(f"Unknown score function {config.score_function}.")
        self.compute_score = _SCORERS[config.score_function]

    def __call__(self, data: DataProto) -> Tuple[torch.Tensor, Dict[str, List[float]]]:
        responses = data.batch["responses"]
        lengths = data.batch["response_mask"].sum(-1).tolist()
        reward = torch.zeros(responses.shape, dtype=torch.float32)
        metrics: Dict[str, List[float]] = defaultdict(list)
        for i in range(len(data)):
            n = int(lengths[i])
            text = self.tokenizer.decode(responses[i, :n], skip_special_tokens=self.config.skip_special_tokens)
            gt = data.non_tensor_batch["ground_truth"][i]
            if self.config.score_function == "spatial_sgg":
                score = self.compute_score(text, gt, data.non_tensor_batch["problem"][i])
            else:
                score = self.compute_score(text, gt)
            reward[i, n - 1] = score["overall"]
            for k, v in score.items():
                metrics[k].append(v)
        return reward, metrics
