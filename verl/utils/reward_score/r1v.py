"""`r1v` and `r1v_scene` reward plug-ins (vanilla-GRPO scripts): 0.5*accuracy + 0.5*format.
Reference behaviour: verl/utils/reward_score/r1v.py:21-59 and r1v_scene.py:27-61.  Answer grading in the
reference goes through `mathruler.grader.grade_answer` (absent here, unpinned); for the multiple-choice /
short-text answers of STVQA it reduces to a normalised string comparison, used as the fallback."""
from __future__ import annotations

import re
from typing import Dict

try:                                                    # pragma: no cover - only when mathruler is installed
    from mathruler.grader import grade_answer as _grade
except Exception:
    def _grade(pred: str, gt: str) -> bool:
        return pred.strip().lower() == gt.strip().lower()

_R1V = re.compile(r"<think>.*?</think>\s*<answer>.*?</answer>", re.DOTALL)
_SCENE = re.compile(r"<observe>.*?</observe>\s*<scene>.*?</scene>\s*<think>.*?</think>\s*<answer>.*?</answer>", re.DOTALL)


def r1v_format_reward(predict_str: str) -> float:
    return 1.0 if _R1V.fullmatch(predict_str) else 0.0


def r1v_accuracy_reward(predict_str: str, ground_truth: str) -> float:
    try:
        gt = ground_truth.strip()
        if "<answer>" in ground_truth and "</answer>" in ground_truth:
            m = re.search(r"<answer>(.*?)</answer>", ground_truth)
            gt = m.group(1).strip() if m else gt
        m = re.search(r"<answer>(.*?)</answer>", predict_str)
        pred = m.group(1).strip() if m else predict_str.strip()
        return 1.0 if _grade(pred, gt) else 0.0
    except Exception:
        return 0.0


def r1v_compute_score(predict_str: str, ground_truth: str) -> Dict[str, float]:
    f, a = r1v_format_reward(predict_str), r1v_accuracy_reward(predict_str, ground_truth)
    return {"overall": 0.5 * a + 0.5 * f, "format": f, "accuracy": a}


def _tag(text: str) -> str:
    m = re.search(r"<answer>(.*?)</answer>", text, re.DOTALL)
    return m.group(1).strip() if m else ""


def r1v_scene_compute_score(predict_str: str, ground_truth: str) -> Dict[str, float]:
    if not _SCENE.fullmatch(predict_str):
        return {"overall": 0.0, "format": 0.0, "accuracy": 0.0}
    a = float(_tag(predict_str).lower() == _tag(ground_truth).lower())
    return {"overall": 0.5 * a + 0.5, "format": 1.0, "accuracy": a}
