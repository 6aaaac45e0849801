"""Rollout generator parity: KV-cache decode with shared prompt KV vs teacher-forced fp32 oracle logits,
response post-processing layout, and the sampler's distribution."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tiny  # noqa: E402
from oracle import qwen25vl as Q  # noqa: E402
from oracle import positions as P  # noqa: E402


@pytest.fixture(scope="module")
def env(golden_dir):
    from spatialthinker_amd import model as mdl
    from spatialthinker_amd.rollout import Generator
    cfg = mdl.VLConfig(**tiny.TINY)
    params = tiny.make_params()
    store = mdl.ParamStore(cfg, trainable=False)
    store.load_hf_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    eng = mdl.Qwen25VL(cfg, store)
    return cfg, params, eng, Generator(eng)


def _prompts():
    b = tiny.make_batch()
    Pn = b["P"]
    ids, mask = b["input_ids"][:, :Pn], b["attention_mask"][:, :Pn]
    pos = np.stack([P.mrope_position_ids(ids[i], b["image_grid_thw"][i:i + 1], mask[i], image_token_id=tiny.TINY["image_token_id"],
                                         vision_start_token_id=tiny.TINY["vision_start_token_id"]) for i in range(2)])
    pos[:, :, :][np.repeat((mask == 0)[:, None, :], 3, 1)] = 0
    off = np.concatenate([[0], np.cumsum(b["patch_counts"])])
    pix = [torch.from_numpy(b["pixel_values"][off[i]:off[i + 1]]) for i in range(2)]
    grids = [b["image_grid_thw"][i:i + 1] for i in range(2)]
    return ids, mask, pos, pix, grids


@pytest.mark.parametrize("use_graph", [False, True])
def test_greedy_decode_matches_teacher_forced_oracle(env, use_graph):
    cfg, params, eng, gen = env
    ids, mask, pos, pix, grids = _prompts()
    R, n = 10, 2
    out = gen.generate(ids, mask, pos, n=n, max_new_tokens=R, temperature=0.0, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID,
                       pixel_values=pix, image_grid_thw=grids, ignore_eos=True, use_graph=use_graph).cpu().numpy()
    assert out.shape == (4, R)
    np.testing.assert_array_equal(out[0], out[1]); np.testing.assert_array_equal(out[2], out[3])     # greedy: rollouts of a prompt coincide
    p32 = {k: torch.from_numpy(v) for k, v in params.items()}
    ocfg = Q.VLConfig(**tiny.TINY)
    for b in range(2):
        sel = mask[b] == 1
        seq = np.concatenate([ids[b][sel], out[b * n]])
        ppos = np.concatenate([pos[b][:, sel], pos[b][:, -1:] + np.arange(1, R + 1)], 1)
        logits = Q.forward_logits(p32, ocfg, torch.from_numpy(seq), torch.from_numpy(ppos), [0, len(seq)], pix[b], grids[b])
        L0 = int(sel.sum())
        for j in range(R):
            row = logits[L0 - 1 + j]
            chosen = float(row[out[b * n][j]])
            assert chosen >= float(row.max()) - 0.08, (b, j, chosen, float(row.max()))    # argmax up to bf16 noise


def test_greedy_decode_vs_hf_generate_golden(env, golden_dir):
    """Greedy rollout vs transformers' own `generate(do_sample=False)` on the same tiny model (tests/golden/generate_tiny.npz, made by
    make_golden.py gen_generate; the oracle's KV-cache decode reproduces it exactly on CPU).  bf16 evaluation may flip an argmax whose
    fp32 margin is below the bf16 noise (0.08 on these logits): tokens must agree up to the first such near-tie, which the fp32
    oracle's teacher-forced logits identify."""
    cfg, params, eng, gen = env
    ids, mask, pos, pix, grids = _prompts()
    gold = np.load(os.path.join(golden_dir, "generate_tiny.npz"))
    R = 10
    out = gen.generate(ids, mask, pos, n=1, max_new_tokens=R, temperature=0.0, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID,
                       pixel_values=pix, image_grid_thw=grids, ignore_eos=True).cpu().numpy()
    p32 = {k: torch.from_numpy(v) for k, v in params.items()}
    ocfg = Q.VLConfig(**tiny.TINY)
    agreed = 0
    for b in range(2):
        want = gold[f"greedy{b}"]
        sel = mask[b] == 1
        seq = np.concatenate([ids[b][sel], want])
        ppos = np.concatenate([pos[b][:, sel], pos[b][:, -1:] + np.arange(1, R + 1)], 1)
        logits = Q.forward_logits(p32, ocfg, torch.from_numpy(seq), torch.from_numpy(ppos), [0, len(seq)], pix[b], grids[b])
        L0 = int(sel.sum())
        for j in range(R):
            if out[b, j] == want[j]:
                agreed += 1
                continue
            top2 = torch.topk(logits[L0 - 1 + j], 2).values
            assert float(top2[0] - top2[1]) < 0.08, (b, j, out[b].tolist(), want.tolist(), top2.tolist())   # a genuine near-tie
            break
    print(f"greedy tokens identical to HF generate: {agreed} of {2 * R}")
    assert agreed >= R                                                        # and at least half of them before any near-tie


def test_fused_decode_epilogues_are_bit_identical_to_unfused_chain(env):
    """The fused decode step (slab GEMM + finish/RoPE/KV-append, finish/RMSNorm, SwiGLU epilogue) keeps every bf16 rounding
    point of the unfused launch chain: sampled responses agree token for token."""
    from spatialthinker_amd.rollout import Generator
    cfg, params, eng, gen = env
    ids, mask, pos, pix, grids = _prompts()
    kw = dict(n=3, max_new_tokens=12, temperature=1.0, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID, seed=5,
              pixel_values=pix, image_grid_thw=grids, ignore_eos=True)
    a = gen.generate(ids, mask, pos, **kw).cpu().numpy()
    b = Generator(eng, fused_decode=False).generate(ids, mask, pos, **kw).cpu().numpy()
    np.testing.assert_array_equal(a, b)


def test_decode_batch_compaction_keeps_samples_and_layout(env):
    """Finished samples leave the decode batch (new, narrower phase on the survivors).  The RNG is keyed by the sample id, so a
    sample's tokens do not depend on who else is still running — up to last-bit logit differences between GEMM tile plans of
    different batch widths, which may flip a near-tie: most rows must agree exactly, all must keep the response layout."""
    from spatialthinker_amd.rollout import Generator
    cfg, params, eng, gen = env
    ids, mask, pos, pix, grids = _prompts()
    n, R = 40, 24
    rs = np.random.RandomState(0)
    lens = np.where(rs.rand(2 * n) < 0.6, rs.randint(2, 6, 2 * n), rs.randint(16, R + 1, 2 * n)).astype(np.int64)
    kw = dict(n=n, max_new_tokens=R, temperature=1.0, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID, seed=9,
              pixel_values=pix, image_grid_thw=grids, forced_lengths=lens, sync_every=4)
    g1 = Generator(eng); g1.compact = True
    g0 = Generator(eng); g0.compact = False
    a = g1.generate(ids, mask, pos, **kw).cpu().numpy()
    b = g0.generate(ids, mask, pos, **kw).cpu().numpy()
    for o in (a, b):
        for r in range(2 * n):
            L_ = int(lens[r])
            e = int(np.argmax(o[r] == tiny.EOS_ID))             # first EOS: forced at L_-1 unless the random model emitted one earlier
            assert o[r, e] == tiny.EOS_ID and e <= L_ - 1 and np.all(o[r, e + 1:] == tiny.PAD_ID)
    same = np.mean([np.array_equal(a[r], b[r]) for r in range(2 * n)])
    assert same >= 0.9, same


def test_waves_and_pooled_survivors_match_single_wave(env):
    """Rollout batches wider than max_decode_batch run as waves whose survivors are pooled and decoded together (rows at
    different response indices): same samples as one wide wave up to last-bit GEMM-plan differences."""
    from spatialthinker_amd.rollout import Generator
    cfg, params, eng, gen = env
    ids, mask, pos, pix, grids = _prompts()
    n, R = 48, 20
    rs = np.random.RandomState(1)
    lens = np.where(rs.rand(2 * n) < 0.5, rs.randint(2, 7, 2 * n), rs.randint(12, R + 1, 2 * n)).astype(np.int64)
    kw = dict(n=n, max_new_tokens=R, temperature=1.0, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID, seed=4,
              pixel_values=pix, image_grid_thw=grids, forced_lengths=lens, sync_every=4)
    g_waves = Generator(eng); g_waves.max_decode_batch = 64
    g_one = Generator(eng); g_one.compact = False
    a = g_waves.generate(ids, mask, pos, **kw).cpu().numpy()
    b = g_one.generate(ids, mask, pos, **kw).cpu().numpy()
    for o in (a, b):
        for r in range(2 * n):
            e = int(np.argmax(o[r] == tiny.EOS_ID))
            assert o[r, e] == tiny.EOS_ID and e <= int(lens[r]) - 1 and np.all(o[r, e + 1:] == tiny.PAD_ID)
    same = np.mean([np.array_equal(a[r], b[r]) for r in range(2 * n)])
    assert same >= 0.9, same


def test_512_row_decode_waves_match_256_row_waves_and_greedy_oracle(env):
    """Decode waves wider than one 256-row GEMM tile (the default since round 2: up to 512 rows = two row tiles per projection, the
    fused finishes and the decode attention over 512 rows): (a) greedy decode of 2 x 180 = 360 concurrent rollouts reproduces the
    256-row schedule token for token on almost every row (last-bit GEMM-plan differences may flip a near-tie) and every rollout of a
    prompt decodes the same greedy sequence; (b) sampled decode keeps the response layout and agrees with the 256-row schedule on
    >= 90 % of the rows (the RNG is keyed by sample id and response index, not by the batch)."""
    from spatialthinker_amd.rollout import Generator
    cfg, params, eng, gen = env
    ids, mask, pos, pix, grids = _prompts()
    n, R = 180, 14
    wide, narrow = Generator(eng), Generator(eng)
    wide.max_decode_batch, narrow.max_decode_batch = 512, 256
    kw = dict(n=n, max_new_tokens=R, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID, pixel_values=pix, image_grid_thw=grids)
    g_w = wide.generate(ids, mask, pos, temperature=0.0, ignore_eos=True, **kw).cpu().numpy()
    g_n = narrow.generate(ids, mask, pos, temperature=0.0, ignore_eos=True, **kw).cpu().numpy()
    assert g_w.shape == (2 * n, R) and wide.stats["decode_row_steps"] / wide.stats["decode_steps"] > 256       # the wide wave really ran > 256 rows
    for b in range(2):
        rows = g_w[b * n:(b + 1) * n]
        assert np.mean([np.array_equal(r, rows[0]) for r in rows]) >= 0.95                    # greedy: the rollouts of a prompt coincide
    assert np.mean([np.array_equal(g_w[r], g_n[r]) for r in range(2 * n)]) >= 0.95
    rs = np.random.RandomState(3)
    lens = np.where(rs.rand(2 * n) < 0.5, rs.randint(2, 6, 2 * n), rs.randint(8, R + 1, 2 * n)).astype(np.int64)
    s_w = wide.generate(ids, mask, pos, temperature=1.0, seed=4, forced_lengths=lens, sync_every=4, **kw).cpu().numpy()
    s_n = narrow.generate(ids, mask, pos, temperature=1.0, seed=4, forced_lengths=lens, sync_every=4, **kw).cpu().numpy()
    for o in (s_w, s_n):
        for r in range(2 * n):
            e = int(np.argmax(o[r] == tiny.EOS_ID))
            assert o[r, e] == tiny.EOS_ID and e <= int(lens[r]) - 1 and np.all(o[r, e + 1:] == tiny.PAD_ID)
    assert np.mean([np.array_equal(s_w[r], s_n[r]) for r in range(2 * n)]) >= 0.9


def test_eos_stops_and_pads_and_forced_lengths(env):
    cfg, params, eng, gen = env
    ids, mask, pos, pix, grids = _prompts()
    lens = np.array([3, 7, 1, 12])
    out = gen.generate(ids, mask, pos, n=2, max_new_tokens=12, temperature=1.0, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID,
                       pixel_values=pix, image_grid_thw=grids, forced_lengths=lens, seed=5).cpu().numpy()
    for i, L in enumerate(lens):
        assert out[i, L - 1] == tiny.EOS_ID and (out[i, L:] == tiny.PAD_ID).all()
        assert (out[i, :L - 1] != tiny.PAD_ID).all()
    # different seeds give different samples, same seed reproduces
    a = gen.generate(ids, mask, pos, n=2, max_new_tokens=8, temperature=1.0, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID,
                     pixel_values=pix, image_grid_thw=grids, ignore_eos=True, seed=1).cpu().numpy()
    b = gen.generate(ids, mask, pos, n=2, max_new_tokens=8, temperature=1.0, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID,
                     pixel_values=pix, image_grid_thw=grids, ignore_eos=True, seed=1).cpu().numpy()
    c = gen.generate(ids, mask, pos, n=2, max_new_tokens=8, temperature=1.0, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID,
                     pixel_values=pix, image_grid_thw=grids, ignore_eos=True, seed=2).cpu().numpy()
    np.testing.assert_array_equal(a, b)
    assert (a != c).any() and (a[0] != a[1]).any()


def test_sampler_distribution():
    from spatialthinker_amd import ops
    z = torch.tensor([2.0, 1.0, 0.0, -1.0, 0.5, -3.0, 1.5, 0.25]).bfloat16()
    B = 4096
    logits = z[None, :].repeat(B, 1).cuda()
    for temp in (1.0, 0.5):
        counts = np.zeros(8)
        for step in range(8):
            t = ops.sample(logits, temp, seed=11, step=step).cpu().numpy()
            counts += np.bincount(t, minlength=8)
        p = torch.softmax(z.float() / temp, 0).numpy()
        n = counts.sum()
        chi2 = ((counts - n * p) ** 2 / (n * p)).sum()
        assert chi2 < 30, (temp, chi2, counts / n, p)                   # 7 dof: P(chi2 > 30) ~ 1e-4
    assert (ops.sample(logits, 0.0, seed=3, step=0).cpu().numpy() == 0).all()
    forced = torch.full((B,), -1, dtype=torch.int32); forced[5] = 6
    t = ops.sample(logits, 1.0, seed=3, step=0, forced=forced.cuda()).cpu().numpy()
    assert t[5] == 6


@pytest.mark.parametrize("top_k,top_p", [(5, 1.0), (-1, 0.7), (20, 0.9), (1, 1.0)])
def test_top_k_top_p_sampling_support_and_distribution(top_k, top_p):
    """vLLM filter order (top-k, then top-p on the renormalised rest, ties with the cut value kept): every sample lies in the
    numpy-derived support and the empirical distribution matches the renormalised softmax (chi-square)."""
    from spatialthinker_amd import ops
    rs = np.random.RandomState(top_k * 7 + int(top_p * 100))
    V, N, T = 300, 6000, 0.8
    z = torch.from_numpy((rs.standard_normal(V) * 2.0).astype(np.float32)).bfloat16()
    zf = z.float().numpy().astype(np.float64) / T
    p = np.exp(zf - zf.max()); p /= p.sum()
    keep = np.ones(V, dtype=bool)
    if top_k > 0:
        kth = np.sort(zf)[::-1][top_k - 1]
        keep &= zf >= kth
    if top_p < 1.0:
        q = np.where(keep, p, 0.0); q /= q.sum()
        vals = np.unique(zf[keep])[::-1]                       # distinct values, descending; whole value-buckets are kept
        cum, cut = 0.0, vals[-1]
        for v_ in vals:
            cum += q[(zf == v_) & keep].sum()
            if cum >= top_p - 1e-12:
                cut = v_; break
        keep &= zf >= cut
    q = np.where(keep, p, 0.0); q /= q.sum()
    logits = z[None, :].repeat(N, 1).cuda().contiguous()
    tok = ops.sample(logits, T, seed=3, step=1, top_k=top_k, top_p=top_p).cpu().numpy()
    assert keep[tok].all(), (np.unique(tok[~keep[tok]]), keep.sum())
    cnt = np.bincount(tok, minlength=V).astype(np.float64)
    sel = q * N >= 5
    chi2 = ((cnt[sel] - q[sel] * N) ** 2 / (q[sel] * N)).sum()
    dof = max(int(sel.sum()) - 1, 1)
    assert chi2 < dof + 6 * np.sqrt(2 * dof) + 10, (chi2, dof)


def test_old_log_probs_from_prompt_cache_match_full_pass(env):
    """The rollout prefill leaves the prompt K/V and last hidden states; the old-policy log-prob pass run on the response tokens
    only (on top of that cache) must give what the full pass over prompt + image + response gives."""
    from spatialthinker_amd.actor import PolicyEngine
    cfg, params, eng, gen = env
    ids, mask, pos, pix, grids = _prompts()
    n, R = 3, 10
    resp, cache = gen.generate(ids, mask, pos, n=n, max_new_tokens=R, temperature=1.0, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID,
                               seed=2, pixel_values=pix, image_grid_thw=grids, forced_lengths=np.array([10, 4, 7, 10, 2, 9]),
                               return_prompt_cache=True)
    resp = resp.cpu()
    rmask = (torch.cumsum((resp == tiny.EOS_ID).long(), 1) - (resp == tiny.EOS_ID).long() == 0).long()
    ids_f = torch.cat([torch.from_numpy(ids).repeat_interleave(n, 0), resp], 1)
    mask_f = torch.cat([torch.from_numpy(mask).repeat_interleave(n, 0), rmask], 1)
    pos_p = torch.from_numpy(pos).repeat_interleave(n, 0)
    pos_f = torch.cat([pos_p, pos_p[..., -1:] + torch.arange(1, R + 1)], -1)
    mm = np.repeat(np.array([{"pixel_values": p_, "image_grid_thw": g_} for p_, g_ in zip(pix, grids)], dtype=object), n)
    data = dict(input_ids=ids_f, attention_mask=mask_f, position_ids=pos_f, responses=resp, multi_modal_inputs=mm)
    pe = PolicyEngine(cfg, eng.p, None)
    full = pe.compute_log_prob(data, 1.0)
    cached = pe.compute_log_prob(data, 1.0, prompt_cache=cache)
    m = rmask.bool()
    assert float((full.cpu() - cached.cpu())[m].abs().max()) < 2e-3
    assert torch.all(cached.cpu()[~m] == 0)
    # a cache built for other prompts (or other weights) is ignored, not misused
    bad = dict(cache); bad["prompt_ids"] = cache["prompt_ids"].copy(); bad["prompt_ids"][0, -1] += 1
    assert torch.equal(pe.compute_log_prob(data, 1.0, prompt_cache=bad), full)


def test_prompt_chunked_rollout_matches_the_unchunked_call_and_overflow_fails_cleanly(env):
    """Round 4 (the shipped scripts' worst case: 128 prompts x 8 rollouts, 6144-token prompts, 2048-token responses do not fit next to
    the training state): generate() cuts the prompts into chunks by a memory plan.  The counter RNG is keyed by the sample's GLOBAL row,
    so the chunked call draws the same samples (up to last-bit GEMM-plan differences of the different batch widths); a chunked call
    hands back no prompt cache; a single prompt that cannot fit raises a RuntimeError with the numbers instead of an OOM."""
    from spatialthinker_amd.rollout import Generator
    cfg, params, eng, gen = env
    ids, mask, pos, pix, grids = _prompts()
    n, R = 24, 16
    rs = np.random.RandomState(2)
    lens = np.where(rs.rand(2 * n) < 0.5, rs.randint(2, 7, 2 * n), rs.randint(9, R + 1, 2 * n)).astype(np.int64)
    kw = dict(n=n, max_new_tokens=R, temperature=1.0, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID, seed=9, pixel_values=pix,
              image_grid_thw=grids, forced_lengths=lens, sync_every=4)
    whole, parts = Generator(eng), Generator(eng)
    a, cache = whole.generate(ids, mask, pos, return_prompt_cache=True, **kw)
    assert whole.last_chunks == [(0, 2)] and cache is not None
    parts.plan_prompt_chunks = lambda lens_, n_, R_, budget_bytes=None: [(0, 1), (1, 2)]
    b, cache_b = parts.generate(ids, mask, pos, return_prompt_cache=True, **kw)
    assert parts.last_chunks == [(0, 1), (1, 2)] and cache_b is None
    a, b = a.cpu().numpy(), b.cpu().numpy()
    assert a.shape == b.shape == (2 * n, R)
    assert np.mean([np.array_equal(a[r], b[r]) for r in range(2 * n)]) >= 0.9
    # the plan itself: whole call when it fits, one chunk per prompt under a tight budget, a clean error when nothing fits
    plens = mask.sum(1)
    need_one = max(whole.rollout_bytes(int(plens[i]), 1, n, R) for i in range(2))
    assert whole.plan_prompt_chunks(plens, n, R, budget_bytes=1e15) == [(0, 2)]
    assert whole.plan_prompt_chunks(plens, n, R, budget_bytes=need_one * 1.01) == [(0, 1), (1, 2)]
    with pytest.raises(RuntimeError, match="rollout does not fit"):
        whole.plan_prompt_chunks(plens, n, R, budget_bytes=need_one * 0.5)


def test_rollout_recorded_log_probs_match_the_recomputed_old_policy_log_probs(env, measured):
    """Opt-in `emit_log_probs` (worker.rollout.old_log_probs_from_rollout): the decode loop records log softmax(logits / T)[token] of every
    token it samples.  Those ARE the old-policy log-probs of the rollout; the reference obtains the same quantity with a second forward
    (fsdp_workers.py compute_log_probs).  Checked here against that second forward (PolicyEngine.compute_log_prob, full pass) at two
    temperatures — the difference is the bf16 noise between the decode kernels and the packed-forward kernels, the same order as the
    engine's own error against fp32 (tests/test_gpu_model.py: 0.0158) — and through the identity checks of use_rollout_log_probs."""
    from spatialthinker_amd.actor import PolicyEngine
    cfg, params, eng, gen = env
    ids, mask, pos, pix, grids = _prompts()
    n, R = 4, 12
    lens = np.array([12, 4, 7, 12, 2, 9, 5, 11])
    pe = PolicyEngine(cfg, eng.p, None)
    worst = 0.0
    for temp in (1.0, 0.7):
        resp, cache = gen.generate(ids, mask, pos, n=n, max_new_tokens=R, temperature=temp, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID,
                                   seed=3, pixel_values=pix, image_grid_thw=grids, forced_lengths=lens, return_prompt_cache=True, emit_log_probs=True)
        assert cache["log_probs"].shape == (2 * n, R) and cache["temperature"] == temp
        resp_c = resp.cpu()
        rmask = (torch.cumsum((resp_c == tiny.EOS_ID).long(), 1) - (resp_c == tiny.EOS_ID).long() == 0).long()
        ids_f = torch.cat([torch.from_numpy(ids).repeat_interleave(n, 0), resp_c], 1)
        mask_f = torch.cat([torch.from_numpy(mask).repeat_interleave(n, 0), rmask], 1)
        pos_p = torch.from_numpy(pos).repeat_interleave(n, 0)
        pos_f = torch.cat([pos_p, pos_p[..., -1:] + torch.arange(1, R + 1)], -1)
        mm = np.repeat(np.array([{"pixel_values": p_, "image_grid_thw": g_} for p_, g_ in zip(pix, grids)], dtype=object), n)
        data = dict(input_ids=ids_f, attention_mask=mask_f, position_ids=pos_f, responses=resp_c, multi_modal_inputs=mm)
        full = pe.compute_log_prob(data, temp).cpu()
        m = rmask.bool()
        rec = cache["log_probs"].cpu()
        assert torch.all(rec[~m] == 0) and torch.all(rec[m] < 0)                    # nothing behind the end of a response
        d = float((rec - full)[m].abs().max())
        print(f"T = {temp}: rollout-recorded vs recomputed old log-probs, max |d| = {d:.4f} over {int(m.sum())} tokens (mean logp {float(full[m].mean()):.2f})")
        worst = max(worst, d)
        got = pe.compute_log_prob(data, temp, prompt_cache=cache, use_rollout_log_probs=True)
        assert pe.last_log_prob_source == "rollout" and torch.equal(got.cpu(), rec * rmask)
        # identity checks: other responses, another temperature, stale weights -> the forward pass runs (prompt K/V cache still usable)
        other = dict(data, responses=resp_c.clone()); other["responses"][0, 0] += 1
        pe.compute_log_prob(other, temp, prompt_cache=cache, use_rollout_log_probs=True)
        assert pe.last_log_prob_source == "forward"
        pe.compute_log_prob(data, temp * 0.5, prompt_cache=cache, use_rollout_log_probs=True)
        assert pe.last_log_prob_source == "forward"
        stale = dict(cache, weights_version=cache["weights_version"] + 1)
        pe.compute_log_prob(data, temp, prompt_cache=stale, use_rollout_log_probs=True)
        assert pe.last_log_prob_source == "forward"
    measured("rollout_recorded_vs_recomputed_old_logp_max_abs", worst)
    # measured on MI355X: 0.0233 at T = 1, 0.0312 at T = 0.7 — ONE bf16 step of a logit of magnitude 4..8 (2^-5) divided by T: the decode
    # GEMM and the packed-forward GEMM round a logit to neighbouring bf16 values; the bound is 1.3x that
    assert worst < 0.041


def test_early_old_log_probs_are_bit_identical_to_the_serial_order(env):
    """the old-policy log-prob pass of the samples that have FINISHED runs on a CU-range stream while the decode tail of
    the same rollout runs on the complementary compute units (actor.EarlyLogProb <- Generator.generate(on_finished=...)).
    (a) the tokens of the rollout do not depend on the hooks / the tail stream;  (b) the hook really computed rows DURING the rollout,
    in several sets;  (c) the result is BIT-IDENTICAL to the same sets processed serially after the rollout (compute_log_prob_in_sets)
    and (d) equal to the ordinary consecutive-rows pass up to the GEMM summation order (one bf16 step of a logit)."""
    from spatialthinker_amd import ops
    from spatialthinker_amd.actor import PolicyEngine
    from spatialthinker_amd.rollout import Generator
    from verl.workers.rollout.hip_rollout import assemble_rollout_batch
    cfg, params, eng, _ = env
    ids, mask, pos, pix, grids = _prompts()
    n, R, T = 96, 24, 1.0
    rs = np.random.RandomState(7)
    lens = np.clip(rs.normal(12, 5, 2 * n), 2, R).astype(np.int64)
    actor = PolicyEngine(cfg, eng.p, None)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    tail, side = ops.cu_range_stream(0, 64), ops.cu_range_stream(64, n_cu - 64)
    kw = dict(n=n, max_new_tokens=R, temperature=T, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID, seed=11, pixel_values=pix,
              image_grid_thw=grids, forced_lengths=lens, sync_every=2, return_prompt_cache=True)
    plain, cache0 = Generator(eng).generate(ids, mask, pos, **kw)
    early = actor.early_log_prob(ids, mask, pos, n, R, T, [tiny.EOS_ID], side_stream=side)
    gen = Generator(eng)
    resp, cache = gen.generate(ids, mask, pos, on_finished=early.feed, tail_stream=tail, tail_rows=128, **kw)
    assert torch.equal(resp, plain)                                                     # (a)
    fed = [len(s_) for s_ in early.sets]
    assert len(fed) >= 2 and sum(fed) >= n, fed                                         # (b) several sets, most rows during the rollout
    out = assemble_rollout_batch(torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(pos), resp.cpu(), n, [tiny.EOS_ID])
    data = dict(input_ids=out["input_ids"], attention_mask=out["attention_mask"], position_ids=out["position_ids"], responses=out["responses"])
    got = early.finish(data, cache)
    sets = [s_.copy() for s_ in early.sets]
    assert sorted(np.concatenate(sets).tolist()) == list(range(2 * n))
    hooked = [s_.tolist() for s_ in gen.last_finish_sets[:-1] if len(s_)]
    assert [s_.tolist() for s_ in sets[:len(hooked)]] == hooked                         # the hook saw every phase's finishers but the last's
    serial = actor.compute_log_prob_in_sets(data, T, cache, sets, [tiny.EOS_ID], n)
    assert torch.equal(got, serial)                                                     # (c)
    ordinary = actor.compute_log_prob(data, T, prompt_cache=cache)
    assert actor.last_prompt_cache_hit
    rm = out["response_mask"].cuda().bool()
    assert float((got - ordinary)[rm].abs().max()) < 0.04                               # (d)
    assert float(got[~rm].abs().max()) == 0.0
    # a response that does not match what the hook saw -> everything is recomputed by the ordinary pass (nothing stale is returned)
    early2 = actor.early_log_prob(ids, mask, pos, n, R, T, [tiny.EOS_ID], side_stream=side)
    resp2, cache2 = Generator(eng).generate(ids, mask, pos, on_finished=early2.feed, tail_stream=tail, tail_rows=128, **kw)
    tampered = dict(data)
    tampered["responses"] = data["responses"].clone()
    victim = int(early2.sets[0][0])
    tampered["responses"][victim, 0] = (tampered["responses"][victim, 0] + 1) % 900
    tampered["input_ids"] = torch.cat([data["input_ids"][:, :-R], tampered["responses"]], 1)
    got2 = early2.finish(tampered, cache2)
    want2 = actor.compute_log_prob(tampered, T, prompt_cache=cache2)
    assert torch.equal(got2, want2) and early2.sets == []


def test_fp8_mode_runs_the_wide_decode_gate_up_on_the_fp8_tile_within_fp8_noise(env, monkeypatch):
    """fp8 mode (BASELINE config #5's arithmetic, not a parity mode): at 257..512 decode rows the gate/up + SwiGLU product runs on the MX-fp8
    tile from the fp8 weight copy (round 6).  The next-token logits of the first decode iterations must stay within fp8 noise of the same
    engine with ST_FP8_DECODE=0 (bf16 decode tiles on the same fp8-mode prefill), and the switch must change nothing below 257 rows."""
    from spatialthinker_amd.rollout import Generator
    cfg, params, eng, gen = env
    ids, mask, pos, pix, grids = _prompts()
    eng.enable_fp8(True)
    try:
        def logits_of(n, flag):
            monkeypatch.setenv("ST_FP8_DECODE", flag)
            g = Generator(eng)
            taps = []
            g.tap = lambda S, live, idx, tok, lg: taps.append(lg.float().cpu())
            g.generate(ids, mask, pos, n=n, max_new_tokens=3, temperature=0.0, eos_token_id=[tiny.EOS_ID], pad_token_id=tiny.PAD_ID,
                       pixel_values=pix, image_grid_thw=grids, ignore_eos=True, use_graph=False)
            return taps
        wide_on, wide_off = logits_of(160, "1"), logits_of(160, "0")              # 320 rows: the fp8 tile is in play
        assert len(wide_on) == len(wide_off) >= 2
        a, b = wide_on[0], wide_off[0]                                           # same sampled token on both sides (greedy from the prefill's logits)
        rel = float((a - b).norm() / b.norm())
        print(f"fp8 vs bf16 decode gate/up, next-token logits: relative L2 {rel:.4f}, max abs {float((a - b).abs().max()):.4f}, logit std {float(b.std()):.3f}")
        assert 1e-4 < rel < 0.12, rel                                             # differs (the fp8 tile ran) and stays within fp8 noise (measured 0.074)
        narrow_on, narrow_off = logits_of(20, "1"), logits_of(20, "0")            # 40 rows: the switch is not consulted
        assert torch.equal(narrow_on[0], narrow_off[0])
    finally:
        eng.enable_fp8(False)
