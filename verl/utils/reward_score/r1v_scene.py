"""`verl.utils.reward_score.r1v_scene` — the reference's module path (r1v_scene.py:27-61); the scorer lives in r1v.py."""
from .r1v import r1v_scene_compute_score

__all__ = ["r1v_scene_compute_score"]
