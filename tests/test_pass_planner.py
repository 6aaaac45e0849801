"""Token-budgeted passes (spatialthinker_amd/actor.py PolicyEngine._plan_passes): how many of the reference's micro-batches ride in
one forward(/backward) pass is decided from the packed-token count of the rows at hand — the reference's contract is that
micro_batch_size_per_device_* holds for ANY sequence length up to max_prompt_length + max_response_length
(verl/workers/actor/dp_actor.py:169-292; scripts/spatialthinker_7b_grpo.sh:33-34 allow 6144 + 2048)."""
import numpy as np

from spatialthinker_amd.actor import PolicyEngine

plan = PolicyEngine._plan_passes


def _rows(n_groups, G, P, resp):
    p_len = np.full(n_groups * G, P, dtype=np.int64)
    r_len = np.asarray(resp, dtype=np.int64) if not np.isscalar(resp) else np.full(n_groups * G, resp, dtype=np.int64)
    keys = [g for g in range(n_groups) for _ in range(G)]
    return p_len, r_len, keys


def test_bench_shaped_mini_batch_fits_one_pass():
    p_len, r_len, keys = _rows(4, 8, 1102, 512)                            # 4 x 1102 + 32 x 512 = 20.8k packed tokens
    assert plan(0, 32, 4, 8, 24576, p_len, r_len, keys) == [(0, 32)]
    assert plan(0, 32, 4, 4, 24576, p_len, r_len, keys) == [(0, 16), (16, 32)]          # the block cap binds first


def test_rows_at_the_response_cap_are_cut_by_the_token_budget():
    p_len, r_len, keys = _rows(4, 8, 1102, 2048)                           # a group of 8 = 1102 + 16384 tokens; a second group does not fit
    passes = plan(0, 32, 4, 8, 24576, p_len, r_len, keys)
    assert passes == [(0, 8), (8, 16), (16, 24), (24, 32)]
    for a, b in passes:
        assert len({keys[r] for r in range(a, b)}) * 1102 + int(r_len[a:b].sum()) <= 24576


def test_a_single_micro_batch_is_never_split_and_ragged_lengths_pack_greedily():
    p_len, r_len, keys = _rows(2, 4, 6144, 2048)                           # one micro-batch = 6144 + 8192 tokens > budget
    assert plan(0, 8, 4, 8, 10000, p_len, r_len, keys) == [(0, 4), (4, 8)]
    rs = np.random.RandomState(0)
    p_len, r_len, keys = _rows(8, 8, 1102, rs.randint(64, 2048, 64))
    passes = plan(0, 64, 4, 8, 24576, p_len, r_len, keys)
    assert passes[0][0] == 0 and passes[-1][1] == 64 and all(a[1] == b[0] for a, b in zip(passes, passes[1:]))
    for a, b in passes:
        assert (b - a) % 4 == 0 and (b - a) // 4 <= 8
        tok = sum({keys[r]: 1102 for r in range(a, b)}.values()) + int(r_len[a:b].sum())
        assert tok <= 24576 or b - a == 4
        if b < 64 and (b - a) // 4 < 8:                                    # greedy: the next block would not have fitted
            tok2 = sum({keys[r]: 1102 for r in range(a, b + 4)}.values()) + int(r_len[a:b + 4].sum())
            assert tok2 > 24576


def test_cached_prompts_count_responses_only():
    p_len, r_len, keys = _rows(8, 8, 1102, 512)
    assert plan(0, 64, 16, 16, 65536, p_len, r_len, keys, prompts_cached=True) == [(0, 64)]          # 32.8k response tokens
    assert plan(0, 64, 16, 16, 20000, p_len, r_len, keys, prompts_cached=True) == [(0, 32), (32, 64)]
    assert plan(0, 64, 16, 16, 65536, p_len, r_len, keys) == [(0, 64)]                               # + 8 prompts = 41.6k
