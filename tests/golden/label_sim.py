"""Deterministic GRADED label similarity used to pin the fractional-similarity branch of the spatial reward
(reference: verl/utils/reward_score/spatial_sgg.py:33-39 `sem_sim`, :150-160 `_cost`, :209-246 `bi_match_triplets`).

spaCy's `en_core_web_md` vectors are absent from this image, so the golden generator (make_golden.py) and the tests install the SAME
stand-in on both sides — the reference scorer loaded by path and the build's `set_similarity` — once as a binary stub (1.0 / 0.0) and
once as this graded one: character-trigram Jaccard of the cleaned labels, an exact ratio of two small integers in float64.
"""


def trigrams(s: str) -> frozenset:
    p = "  " + s + " "
    return frozenset(p[i:i + 3] for i in range(len(p) - 2))


def trigram_jaccard(a: str, b: str) -> float:
    if a == b:
        return 1.0
    A, B = trigrams(a), trigrams(b)
    return len(A & B) / len(A | B)
