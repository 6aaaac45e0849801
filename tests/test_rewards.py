"""The reward plug-ins against golden scores produced by the reference scorers (tests/golden/rewards.json)."""
import json
import os

import pytest

from verl.utils.reward_score import r1v_compute_score, r1v_scene_compute_score, spatial_sgg_compute_score
from verl.utils.reward_score import spatial_sgg as S


@pytest.fixture(scope="module")
def gold(golden_dir):
    S.set_similarity(S._exact_similarity)            # the stub recorded in the fixture
    return json.load(open(os.path.join(golden_dir, "rewards.json")))


def test_spatial_sgg_scores_are_float64_exact(gold):
    for case in gold["cases"]:
        got = spatial_sgg_compute_score(case["predict"], case["ground_truth"], case["problem"])
        assert got == case["spatial_sgg"], (case["name"], got, case["spatial_sgg"])
    assert len(gold["cases"]) >= 30 and any(c["spatial_sgg"]["spatial_score"] not in (0.0, 1.0) for c in gold["cases"])


def test_r1v_scene_and_r1v(gold):
    for case in gold["cases"]:
        assert r1v_scene_compute_score(case["predict"], case["ground_truth"]) == case["r1v_scene"], case["name"]
    for row in gold["r1v"]:
        assert r1v_compute_score(row["predict"], row["ground_truth"]) == row["r1v"]


def test_ciou_exact(gold):
    for row in gold["ciou"]:
        assert S.compute_ciou(row["a"], row["b"]) == row["ciou"]


def test_missing_image_size_raises_like_the_reference(gold):
    with pytest.raises(ValueError) as e:
        spatial_sgg_compute_score("x", "y", "no size here")
    assert str(e.value) == gold["missing_size_error"]
