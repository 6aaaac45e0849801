"""`r1v_scene` reward plug-in under the reference's module path and function names (verl/utils/reward_score/r1v_scene.py:27-61):
format = the whole response is <observe>…</observe> <scene>…</scene> <think>…</think> <answer>…</answer>; accuracy = the <answer> contents
agree after strip + lower-case; overall = 0.5 accuracy + 0.5 format, all zero when the format fails.  (`r1v_format_reward` and
`r1v_accuracy_reward` exist in r1v.py too, with the plain <think>/<answer> format: the names are per module, as in the reference.)"""
import re
from typing import Dict

from .r1v import _SCENE, r1v_scene_compute_score

__all__ = ["acc_reward", "extract_answer", "r1v_accuracy_reward", "r1v_format_reward", "r1v_scene_compute_score"]


def r1v_format_reward(predict_str: str) -> float:
    return 1.0 if _SCENE.fullmatch(predict_str) else 0.0


def extract_answer(text: str) -> str:
    m = re.search(r"<answer>(.*?)</answer>", text, re.DOTALL)
    return m.group(1).strip() if m else ""


def acc_reward(pred: str, gt: str) -> float:
    return float(pred.strip().lower() == gt.strip().lower())


def r1v_accuracy_reward(predict_str: str, ground_truth: str) -> float:
    return acc_reward(extract_answer(predict_str), extract_answer(ground_truth))
