"""Dense spatial reward of SpatialThinker (reward plug-in `worker.reward.score_function=spatial_sgg`).

Same signature and scores as the reference scorer (verl/utils/reward_score/spatial_sgg.py:644-691 and the
helpers it reaches); written from the behaviour described in SURVEY.md §8(a6), not from its text:

    overall = 0.1*format + 0.2*count + 0.5*accuracy + 0.2*spatial
    count, accuracy only if format == 1;  spatial only if accuracy == 1.

Label similarity (`sem_sim`, spaCy `en_core_web_md` vectors in the reference) only steers the Hungarian
assignment; it is pluggable here (`set_similarity`).  Default: spaCy vectors if the model is installed,
otherwise exact match of the cleaned label (1.0 / 0.0) — the stub the golden fixtures were generated with.
"""
from __future__ import annotations

import json
import math
import re
from functools import lru_cache
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
from scipy.optimize import linear_sum_assignment

WEIGHTS = {"format": 0.1, "count": 0.2, "accuracy": 0.5, "spatial_score": 0.2}
SEM_WEIGHT, BOX_WEIGHT, DUMMY_COST = 2.0, 1.0, 1e5
_ID_RE = re.compile(r"[a-zA-Z_]+\.\d+")
_TAGS = ("observe", "think", "scene", "answer")


# ------------------------------------------------------------------ label similarity (pluggable)
def _clean(label: str) -> str:
    return label.replace("_", " ").replace("-", " ").strip().lower()


def _exact_similarity(a: str, b: str) -> float:
    return 1.0 if a == b else 0.0


SIMILARITY_KIND = "unset"        # which label similarity the scores of this process use: "spacy:en_core_web_md" | "exact-match" | "custom"


def _load_default_similarity() -> Callable[[str, str], float]:
    """The reference's choice (spaCy `en_core_web_md` vectors) when the package and the model are installed, else exact label match.
    The two give DIFFERENT object-matching scores, so the choice is announced once per process (stderr) and kept in SIMILARITY_KIND —
    a run must not change its reward scale silently with what happens to be installed."""
    global SIMILARITY_KIND
    import sys
    try:
        import spacy
        nlp = spacy.load("en_core_web_md", disable=["parser", "ner", "tagger"])
        doc = lru_cache(maxsize=4096)(nlp)
        SIMILARITY_KIND = "spacy:en_core_web_md"
        fn = lambda a, b: float(doc(a).similarity(doc(b)))
    except Exception as e:
        SIMILARITY_KIND = "exact-match"
        fn = _exact_similarity
        print(f"[spatial_sgg] spaCy en_core_web_md is not available ({type(e).__name__}): object labels are matched EXACTLY (similarity 1 / 0); "
              f"the reference's scores use spaCy vector similarity — install it or call set_similarity(fn) for comparable rewards", file=sys.stderr)
    else:
        print("[spatial_sgg] label similarity: spaCy en_core_web_md vectors (the reference's)", file=sys.stderr)
    return fn


_similarity: Optional[Callable[[str, str], float]] = None


def set_similarity(fn: Optional[Callable[[str, str], float]]) -> None:
    global _similarity, SIMILARITY_KIND
    _similarity = fn
    SIMILARITY_KIND = "custom" if fn is not None else "unset"
    _match_cached.cache_clear()


def sem_sim(a: str, b: str) -> float:
    global _similarity
    if _similarity is None:
        _similarity = _load_default_similarity()
    return _similarity(_clean(a.split(".")[0]), _clean(b.split(".")[0]))


# ------------------------------------------------------------------ geometry
def compute_ciou(boxA: Sequence[float], boxB: Sequence[float], eps: float = 1e-7) -> float:
    """Complete-IoU of two [x1,y1,x2,y2] boxes mapped to [0,1] by (ciou+1)/2."""
    a, b = boxA, boxB
    wa, ha, wb, hb = a[2] - a[0], a[3] - a[1], b[2] - b[0], b[3] - b[1]
    iw = max(0.0, min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iw * ih
    iou = inter / (wa * ha + wb * hb - inter + eps)
    dist2 = ((a[0] + a[2]) / 2 - (b[0] + b[2]) / 2) ** 2 + ((a[1] + a[3]) / 2 - (b[1] + b[3]) / 2) ** 2
    diag2 = (max(a[2], b[2]) - min(a[0], b[0])) ** 2 + (max(a[3], b[3]) - min(a[1], b[1])) ** 2 + eps
    v = (4 / (math.pi ** 2)) * (math.atan(wb / (hb + eps)) - math.atan(wa / (ha + eps))) ** 2
    denom = (1 - iou) + v
    alpha = v / denom if denom != 0 else 0.0
    return ((iou - (dist2 / diag2 + alpha * v)) + 1) / 2


# ------------------------------------------------------------------ parsing / validity
def extract_answer(text: str) -> str:
    m = re.search(r"<answer>(.*?)</answer>", text, re.DOTALL)
    return m.group(1).strip() if m else ""


def extract_scene(text: str) -> dict:
    m = re.search(r"<scene>(.*?)</scene>", text, re.DOTALL)
    if not m:
        return {}
    try:
        obj = json.loads(m.group(1).strip())
    except Exception:
        return {}
    return obj if isinstance(obj, dict) else {}


def extract_image_size(problem: str) -> Tuple[int, int]:
    m = re.search(r"Image size: \((.*?) x (.*?)\)", problem)
    if not m:
        raise ValueError("Image size not found in problem!!! Required for spatial_sgg reward scoring.")
    return int(m.group(1)), int(m.group(2))


def is_valid_object(obj) -> bool:
    o = obj
    if not isinstance(o, dict) or set(o.keys()) != {"id", "bbox"}:
        return False
    if not isinstance(o["id"], str) or not _ID_RE.fullmatch(o["id"]):
        return False
    box = o["bbox"]
    return isinstance(box, list) and len(box) == 4 and all(isinstance(x, (int, float)) for x in box)


def is_valid_relation(rel) -> bool:
    r = rel
    if not isinstance(r, dict) or not {"subject", "predicate", "object"} <= set(r.keys()):
        return False
    if not all(isinstance(r[k], str) for k in ("subject", "predicate", "object")):
        return False
    return bool(_ID_RE.fullmatch(r["subject"])) and bool(_ID_RE.fullmatch(r["object"]))


def format_reward(text: str) -> float:
    try:
        for tag in _TAGS:
            if not re.search(rf"<{tag}>.*?</{tag}>", text, re.DOTALL) or text.count(f"<{tag}>") != 1:
                return 0.0
        scene = extract_scene(text)
        if not scene:
            return 0.0
        objs, rels = scene.get("objects", []), scene.get("relationships", [])
        if not isinstance(objs, list) or not isinstance(rels, list):
            return 0.0
        if not all(is_valid_object(o) for o in objs) or not all(is_valid_relation(r) for r in rels):
            return 0.0
        ids = [o["id"] for o in objs]
        return 1.0 if len(ids) == len(set(ids)) else 0.0
    except Exception:
        return 0.0


def acc_reward(pred: str, gt: str) -> float:
    return float(pred.strip().lower() == gt.strip().lower())


def count_reward(pred_scene, gt_scene) -> float:
    if not isinstance(pred_scene, dict) or not isinstance(gt_scene, dict):
        return 0.0
    po, go = pred_scene.get("objects"), gt_scene.get("objects")
    if not isinstance(po, list) or not isinstance(go, list):
        return 0.0
    pr, gr = pred_scene.get("relationships") or [], gt_scene.get("relationships") or []
    closeness = lambda n_pred, n_gt: max(0.0, 1 - abs(n_pred - n_gt) / max(n_gt, 1))
    obj = closeness(len(po), len(go))
    return obj if not len(gr) else obj * 0.7 + closeness(len(pr), len(gr)) * 0.3


# ------------------------------------------------------------------ matching
@lru_cache(maxsize=4096)
def _match_cached(gt_key, pr_key) -> tuple:
    """Hungarian assignment GT j -> prediction i (or None) minimising 2*(1-sim) + (1-ciou); missing predictions are
    padded with rows of cost 1e5."""
    G, P = len(gt_key), len(pr_key)
    cost = np.zeros((P + max(0, G - P), G))
    for i, (pid, pbox) in enumerate(pr_key):
        for j, (gid, gbox) in enumerate(gt_key):
            # argument order follows the reference call chain (_hungarian -> _cost(p, g) -> compute_ciou(g, p))
            cost[i, j] = SEM_WEIGHT * (1.0 - sem_sim(gid, pid)) + BOX_WEIGHT * (1.0 - compute_ciou(list(gbox), list(pbox)))
    cost[P:, :] = DUMMY_COST
    rows, cols = linear_sum_assignment(cost)
    out = [None] * G
    for r, c in zip(rows, cols):
        if r < P:
            out[c] = int(r)
    return tuple(out)


def bi_match(gt_objs: List[dict], pr_objs: List[dict]) -> tuple:
    key = lambda objs: tuple((o["id"], tuple(o["bbox"])) for o in objs)
    return _match_cached(key(gt_objs), key(pr_objs))


def compute_obj_score(gt_objs: List[dict], pr_objs: List[dict]) -> float:
    if not gt_objs:
        return 1.0
    total = 0.0
    for j, i in enumerate(bi_match(gt_objs, pr_objs)):
        if i is not None:
            total += compute_ciou(gt_objs[j]["bbox"], pr_objs[i]["bbox"])
            sem_sim(gt_objs[j]["id"], pr_objs[i]["id"])
    return total / len(gt_objs)


def _triplet_matches(gt_rels: List[dict], pr_rels: List[dict]) -> int:
    G, P = len(gt_rels), len(pr_rels)
    cost = np.zeros((P + max(0, G - P), G))
    for i, p in enumerate(pr_rels):
        for j, g in enumerate(gt_rels):
            cost[i, j] = 1.0 - (0.3 * sem_sim(p["subject"], g["subject"]) + 0.3 * sem_sim(p["object"], g["object"])
                                + 0.4 * sem_sim(p["predicate"], g["predicate"]))
    cost[P:, :] = DUMMY_COST
    rows, _ = linear_sum_assignment(cost)
    return int(sum(1 for r in rows if r < P))


# ---- the reference module's other public helpers (spatial_sgg.py:21-27,41-72,146-149,209-246,248-386,416-420,507-508).  The shipped
# scorer (spatial_sgg_compute_score -> relaxed_spatial_reward) does not call spatial_reward / compute_rel_score / compute_giou; they are kept
# under their names for code that imports them, and pinned against the reference's functions (tests/golden/rewards_helpers.json).
IOU_W, L1_W = 1.0, 5.0


def scale_box(box: Sequence[float], scale: Sequence[float]) -> List[float]:
    sw, sh = scale
    return [box[0] * sw, box[1] * sh, box[2] * sw, box[3] * sh]


def refine_node_edge(label: str) -> str:
    """'fire-hydrant' == 'fire_hydrant' == 'Fire hydrant '"""
    return _clean(label)


def compute_iou(boxA: Sequence[float], boxB: Sequence[float]) -> float:
    iw = max(0, min(boxA[2], boxB[2]) - max(boxA[0], boxB[0]))
    ih = max(0, min(boxA[3], boxB[3]) - max(boxA[1], boxB[1]))
    inter = iw * ih
    union = (boxA[2] - boxA[0]) * (boxA[3] - boxA[1]) + (boxB[2] - boxB[0]) * (boxB[3] - boxB[1]) - inter
    return 0.0 if union == 0 else inter / union


def compute_giou(boxA: Sequence[float], boxB: Sequence[float]) -> float:
    """generalised IoU mapped to [0, 1] by (giou + 1) / 2; the plain IoU when the enclosing box is empty"""
    iw = max(0, min(boxA[2], boxB[2]) - max(boxA[0], boxB[0]))
    ih = max(0, min(boxA[3], boxB[3]) - max(boxA[1], boxB[1]))
    inter = iw * ih
    union = (boxA[2] - boxA[0]) * (boxA[3] - boxA[1]) + (boxB[2] - boxB[0]) * (boxB[3] - boxB[1]) - inter
    iou = inter / union if union > 0 else 0.0
    hull = (max(boxA[2], boxB[2]) - min(boxA[0], boxB[0])) * (max(boxA[3], boxB[3]) - min(boxA[1], boxB[1]))
    if hull == 0:
        return iou
    return ((iou - (hull - union) / hull) + 1.0) / 2.0


def box_L1(a: Sequence[float], b: Sequence[float]) -> float:
    return sum(abs(x - y) for x, y in zip(a, b))


def is_valid_id_format(s: str) -> bool:
    return bool(_ID_RE.fullmatch(s))


def bi_match_triplets(gt_rels: List[dict], pred_rels: List[dict]) -> List[dict]:
    """Hungarian alignment of predicted to ground-truth (subject, predicate, object) triplets on the cost 1 - (0.3 subject + 0.3 object +
    0.4 predicate similarity); missing predictions are padded with rows of cost 1e5 and dropped from the result."""
    G, P = len(gt_rels), len(pred_rels)
    cost = np.zeros((P + max(0, G - P), G))
    for i, p in enumerate(pred_rels):
        for j, g in enumerate(gt_rels):
            cost[i, j] = 1.0 - (0.3 * sem_sim(p["subject"], g["subject"]) + 0.3 * sem_sim(p["object"], g["object"])
                                + 0.4 * sem_sim(p["predicate"], g["predicate"]))
    cost[P:, :] = DUMMY_COST
    rows, cols = linear_sum_assignment(cost)
    return [{"groundtruth": gt_rels[c], "prediction": pred_rels[r], "cost": cost[r, c], "similarity": 1.0 - cost[r, c]}
            for r, c in zip(rows, cols) if r < P]


def compute_rel_score(gt_rels: List[dict], pr_rels: List[dict]) -> float:
    return sum(1.0 - m["cost"] for m in bi_match_triplets(gt_rels, pr_rels)) / len(gt_rels) if gt_rels else 1.0


def spatial_reward(pred_scene, gt_scene, w: int, h: int) -> Tuple[float, float]:
    """(object score, relation score) of the strict variant: per matched GT object 0.5 x (IoU + 5 exp(-L1)) / 6 + 0.5 x label similarity,
    averaged over the GT objects; relation score = mean triplet similarity over the GT relations."""
    if not isinstance(pred_scene, dict) or not isinstance(gt_scene, dict):
        return 0.0, 0.0
    go, po = gt_scene.get("objects") or [], pred_scene.get("objects") or []
    gr, pr = gt_scene.get("relationships") or [], pred_scene.get("relationships") or []
    if not all(isinstance(x, list) for x in (go, po, gr, pr)):
        return 0.0, 0.0
    if not all(is_valid_object(o) for o in po) or not all(is_valid_relation(r) for r in pr):
        return 0.0, 0.0
    norm = lambda objs: [{**o, "id": _clean(o["id"]), "bbox": scale_box(o["bbox"], (1.0 / w, 1.0 / h))} for o in objs]
    rel = lambda rs: [{**r, "subject": _clean(r["subject"]), "object": _clean(r["object"])} for r in rs]
    go, po, gt_trip, pr_trip = norm(go), norm(po), rel(gr), rel(pr)
    if not go:
        obj_score = 1.0 if not po else 0.0
    else:
        box, sim = [], []
        for j, i in enumerate(bi_match(go, po)):
            if i is None:
                box.append(0.0); sim.append(0.0)
                continue
            g, p = go[j], po[i]
            sim.append(sem_sim(g["id"], p["id"]))
            box.append((IOU_W * compute_iou(g["bbox"], p["bbox"]) + L1_W * math.exp(-box_L1(g["bbox"], p["bbox"]))) / (IOU_W + L1_W))
        obj_score = 0.5 * (sum(box) / len(go)) + 0.5 * (sum(sim) / len(go))
    if not gr:
        rel_score = 1.0 if not pr else 0.0
    else:
        rel_score = sum(1.0 - m["cost"] for m in bi_match_triplets(gt_trip, pr_trip)) / len(gt_trip)
    return obj_score, rel_score


def relaxed_spatial_reward(pred_scene, gt_scene, w: int, h: int, threshold: float = 0.0, rel_gating: bool = False) -> float:
    if not isinstance(pred_scene, dict) or not isinstance(gt_scene, dict):
        return 0.0
    go, po = gt_scene.get("objects") or [], pred_scene.get("objects") or []
    gr, pr = gt_scene.get("relationships") or [], pred_scene.get("relationships") or []
    if not all(isinstance(x, list) for x in (go, po, gr, pr)):
        return 0.0
    if not all(is_valid_object(o) for o in po) or not all(is_valid_relation(r) for r in pr):
        return 0.0
    sx, sy = 1.0 / w, 1.0 / h
    norm = lambda objs: [{**o, "id": _clean(o["id"]), "bbox": [o["bbox"][0] * sx, o["bbox"][1] * sy, o["bbox"][2] * sx, o["bbox"][3] * sy]}
                         for o in objs]
    go, po = norm(go), norm(po)
    if not gr:
        return (1.0 if not po else 0.0) if not go else compute_obj_score(go, po)
    rel = lambda rs: [{**r, "subject": _clean(r["subject"]), "object": _clean(r["object"])} for r in rs]
    n_matched = _triplet_matches(rel(gr), rel(pr))
    score = compute_obj_score(go, po)
    return 0.0 if (n_matched == 0 and rel_gating) else score


# ------------------------------------------------------------------ entry point
def spatial_sgg_compute_score(predict_str: str, ground_truth_str: str, problem: str) -> Dict[str, float]:
    pred_answer, gt_answer = extract_answer(predict_str), extract_answer(ground_truth_str)
    pred_scene, gt_scene = extract_scene(predict_str), extract_scene(ground_truth_str)
    width, height = extract_image_size(problem)
    fr = format_reward(predict_str)
    cr = ar = spatial = 0.0
    if fr == 1.0:
        cr = count_reward(pred_scene, gt_scene)
        ar = acc_reward(pred_answer, gt_answer)
        if ar == 1.0:
            spatial = relaxed_spatial_reward(pred_scene, gt_scene, width, height, 0.0, False)
    total = fr * WEIGHTS["format"] + cr * WEIGHTS["count"] + ar * WEIGHTS["accuracy"] + spatial * WEIGHTS["spatial_score"]
    return {"overall": total, "format": fr, "count": cr, "accuracy": ar, "spatial_score": spatial}
