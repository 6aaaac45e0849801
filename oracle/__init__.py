"""CPU oracle for the GRPO hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Every function in this package is a plain numpy / torch-fp32 restatement of what the
reference (hunarbatra/SpatialThinker, a veRL/EasyR1 fork) or its third-party model
code (HF `transformers` Qwen2.5-VL) computes on the hot path, each citing the
reference file:line it follows.

Rules (enforced by tests/test_layout.py):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
    import anything under `oracle/`;
  * the product path (`spatialthinker_amd/`, `verl/`) never imports it and fails
    loudly when the HIP library is missing instead of falling back to this code.

Parity pin: the oracle itself is pinned by `tests/golden/*.npz|json`, generated in
the build container by `tests/golden/make_golden.py` from (i) the reference's own
importable pure functions and (ii) HF `Qwen2_5_VLForConditionalGeneration` tiny
random-init configs (the reference holds no tests or golden vectors of its own:
SURVEY.md §4).  `sem_sim` (spaCy vectors) and `mathruler.grade_answer` are absent
third-party pieces: parity for those two sub-terms is UNPINNED.
"""
