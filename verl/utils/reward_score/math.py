"""`math` reward plug-in (the reference's default `worker.reward.score_function`; text-only math extras, SURVEY.md: out of the hot path's scope):
0.9 * accuracy + 0.1 * format.  Reference behaviour: verl/utils/reward_score/math.py:21-40 — format = the whole response matches
`<think>…</think>…\\boxed{…}…`, accuracy = `mathruler.grader.grade_answer(extract_boxed_content(response), ground_truth)`.
mathruler is not in this image: when it is importable it is used (the reference's grading, symbolic equivalence included); otherwise the
content of the LAST `\\boxed{…}` (brace-matched) is compared with the ground truth as a normalised string — unpinned, stated here and
in the returned dict's absence of any extra key (same keys as the reference)."""
from __future__ import annotations

import re
from typing import Dict

try:                                                    # pragma: no cover - only when mathruler is installed
    from mathruler.grader import extract_boxed_content as _boxed, grade_answer as _grade
    _FALLBACK = False
except Exception:
    _FALLBACK = True

    def _boxed(text: str) -> str:
        """content of the last \\boxed{...} with balanced braces, "None" when there is none (mathruler's convention)"""
        start = text.rfind("\\boxed{")
        if start < 0:
            return "None"
        depth, i0 = 0, start + len("\\boxed{")
        for i in range(i0, len(text)):
            if text[i] == "{":
                depth += 1
            elif text[i] == "}":
                if depth == 0:
                    return text[i0:i]
                depth -= 1
        return "None"

    def _grade(pred: str, gt: str) -> bool:
        norm = lambda s: re.sub(r"\s+", "", s.strip().strip("$").lower())
        return norm(pred) == norm(gt)

_warned = False


def _warn_fallback_once() -> None:
    global _warned
    if not _warned and _FALLBACK:
        _warned = True
        print("[math reward] mathruler is not installed: answers are graded by a normalised STRING comparison of the last \\boxed{...}; "
              "the reference's grade_answer also accepts symbolic / numeric equivalents ('0.5' vs '\\frac12'), so accuracy rewards can be "
              "lower than the reference's on the same responses", flush=True)


_FORMAT = re.compile(r"<think>.*</think>.*\\boxed\{.*\}.*", re.DOTALL)


def math_format_reward(predict_str: str) -> float:
    return 1.0 if _FORMAT.fullmatch(predict_str) else 0.0


def math_acc_reward(predict_str: str, ground_truth: str) -> float:
    return 1.0 if _grade(_boxed(predict_str), ground_truth) else 0.0


def math_compute_score(predict_str: str, ground_truth: str) -> Dict[str, float]:
    _warn_fallback_once()
    predict_str = re.sub(r"\s*(<|>|/)\s*", r"\1", predict_str)      # "< think >" -> "<think>" (the reference's qwen2.5-vl-32b clean-up)
    fmt = math_format_reward(predict_str)
    acc = math_acc_reward(predict_str, ground_truth)
    return {"overall": 0.9 * acc + 0.1 * fmt, "format": fmt, "accuracy": acc}
