from .math import math_compute_score
from .r1v import r1v_compute_score, r1v_scene_compute_score
from .spatial_sgg import spatial_sgg_compute_score

__all__ = ["math_compute_score", "r1v_compute_score", "r1v_scene_compute_score", "spatial_sgg_compute_score"]
