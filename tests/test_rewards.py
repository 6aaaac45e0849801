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


# ---------------------------------------------------------------- fractional label similarity
@pytest.fixture(scope="module")
def graded(golden_dir):
    import sys
    sys.path.insert(0, golden_dir)
    from label_sim import trigram_jaccard
    return trigram_jaccard, json.load(open(os.path.join(golden_dir, "rewards_graded.json")))


def _scene(text):
    import re
    return json.loads(re.search(r"<scene>(.*?)</scene>", text, re.S).group(1))


def test_graded_similarity_scores_and_assignments_are_float64_exact(graded):
    """The reference scorer was run under a GRADED deterministic similarity (character-trigram Jaccard, tests/golden/label_sim.py) and
    under the binary stub; the Hungarian cost 2 (1 - sim) + (1 - ciou) (reference spatial_sgg.py:150-160) depends on the fraction.
    Both regimes must reproduce float64-exactly, and in >= 20 strings the graded assignment differs from the binary one."""
    sim, gold = graded
    differ = scored = 0
    try:
        for kind, fn in (("binary", S._exact_similarity), ("graded", sim)):
            S.set_similarity(fn)
            for case in gold["cases"]:
                got = spatial_sgg_compute_score(case["predict"], case["ground_truth"], case["problem"])
                assert got == case[f"spatial_sgg_{kind}"], (kind, case["name"], got, case[f"spatial_sgg_{kind}"])
                mapping = list(S.bi_match(_scene(case["ground_truth"])["objects"], _scene(case["predict"])["objects"]))
                assert mapping == case[f"mapping_{kind}"], (kind, case["name"], mapping)
                g_rel, p_rel = _scene(case["ground_truth"]).get("relationships", []), _scene(case["predict"]).get("relationships", [])
                if g_rel and p_rel:
                    assert S._triplet_matches(g_rel, p_rel) == len(case[f"triplets_{kind}"]), (kind, case["name"])
        for a, b, want in gold["sims"]:
            assert sim(a, b) == want
        for case in gold["cases"]:
            differ += case["mapping_binary"] != case["mapping_graded"]
            scored += case["spatial_sgg_binary"] != case["spatial_sgg_graded"]
    finally:
        S.set_similarity(S._exact_similarity)
    assert differ >= 20 and scored >= 8 and len(gold["cases"]) >= 50, (differ, scored)
    assert any(0.0 < c["spatial_sgg_graded"]["spatial_score"] < 1.0 for c in gold["cases"])


def test_math_plugin_equals_the_reference_module_on_its_stubbed_graders():
    """rewards_math.json = the reference's verl/utils/reward_score/math.py:21-40 run with mathruler's two functions stubbed by this build's
    documented fallbacks (make_golden.py math): the clean-up regex, the format regex and the 0.9 / 0.1 weighting, float64-exact."""
    import json
    import os
    from verl.utils.reward_score import math_compute_score
    from verl.workers.reward.custom import _SCORERS
    assert _SCORERS["math"] is math_compute_score                     # the reference's DEFAULT worker.reward.score_function resolves
    rows = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rewards_math.json")))["cases"]
    assert len(rows) >= 16
    for r in rows:
        assert math_compute_score(r["predict"], r["ground_truth"]) == r["score"], r


def test_public_reward_helpers_equal_the_reference_functions(graded, golden_dir):
    """rewards_helpers.json = the reference's other public spatial_sgg functions (make_golden.py helpers) under the graded stand-in
    similarity: box metrics on 29 box pairs, label / id normalisers, and per scene of the 50 graded cases the strict spatial_reward pair,
    compute_rel_score and the triplet alignment — float64-exact."""
    sim, _ = graded
    gold = json.load(open(os.path.join(golden_dir, "rewards_helpers.json")))
    for r in gold["boxes"]:
        a, b = r["a"], r["b"]
        assert S.compute_iou(a, b) == r["iou"] and S.compute_giou(a, b) == r["giou"] and S.compute_ciou(boxA=a, boxB=b) == r["ciou"], r
        assert S.box_L1(a, b) == r["l1"] and S.scale_box(a, (0.5, 2.0)) == r["scaled"]
    assert all(S.refine_node_edge(l) == want for l, want in gold["labels"]) and all(S.is_valid_id_format(i) == want for i, want in gold["ids"])
    try:
        S.set_similarity(sim)
        for r in gold["scenes"]:
            got = S.spatial_reward(r["pred_scene"], r["gt_scene"], r["w"], r["h"])
            assert [float(x) for x in got] == r["spatial_reward"], (r["name"], got, r["spatial_reward"])
            g_rel, p_rel = r["gt_scene"].get("relationships", []), r["pred_scene"].get("relationships", [])
            assert float(S.compute_rel_score(g_rel, p_rel)) == r["rel_score"], r["name"]
            assert [float(m["similarity"]) for m in S.bi_match_triplets(g_rel, p_rel)] == r["triplet_similarity"], r["name"]
        for r in gold["odd"]:
            assert [float(x) for x in S.spatial_reward(r["pred_scene"], r["gt_scene"], 100, 50)] == r["spatial_reward"], r
    finally:
        S.set_similarity(None)
    assert S.is_valid_object(obj={"id": "dog.1", "bbox": [0, 0, 1, 1]}) and not S.is_valid_relation(rel={"subject": "dog", "predicate": "on", "object": "mat.1"})


def test_r1v_scene_module_functions(gold):
    """verl.utils.reward_score.r1v_scene: the module-level helper names of the reference, consistent with the scorer pinned above."""
    from verl.utils.reward_score import r1v_scene as R
    for row in gold["cases"]:
        p, g, want = row["predict"], row["ground_truth"], row["r1v_scene"]
        assert R.r1v_scene_compute_score(p, g) == want and R.r1v_format_reward(p) == want["format"]
        if want["format"] == 1.0:
            assert R.r1v_accuracy_reward(p, g) == want["accuracy"]
    assert R.extract_answer("x <answer> B </answer>") == "B" and R.extract_answer("none") == "" and R.acc_reward(" a ", "A") == 1.0
