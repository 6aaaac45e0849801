"""Training-TRAJECTORY parity of PolicyEngine.update_policy (dp_actor.py:155-167, 212-292 + AnyPrecisionAdamW + the
constant-with-warmup schedule): four consecutive update_actor calls — 2 mini-batches x 2 micro-batches each, shared-prompt rollout
pairs, the lr = 0 first call, a warm-up step, global-norm clipping active — on the tiny Qwen2.5-VL model, against
oracle.update_loop.UpdateLoop (fp32 CPU autograd through oracle.qwen25vl + oracle.rl_math.AdamWKahanBF16 in its "gpu" scalar mode;
the loop itself is pinned to the reference class by tests/test_oracle_update_loop.py).

What is compared, call by call: every micro-batch's pg_loss / clip fractions / ppo_kl / entropy, kl_loss, every optimizer step's
gradient norm, lr, and after the last call every weight tensor (distance in bf16 ulps and the relative L2 error of the accumulated update
w_final - w_initial).  Bounds are <= 1.3x what the MI355X produced (printed through `measured`)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tiny  # noqa: E402
from oracle import positions as P  # noqa: E402
from oracle import qwen25vl as Q  # noqa: E402
from oracle.update_loop import KahanBF16, UpdateLoop  # noqa: E402

LR, WARMUP, MAX_NORM, KL_COEF, TEMP, CALLS = 3e-5, 2, 0.5, 0.04, 1.0, 4


def _rollout_pairs(seed=31, R=12, Pc=64):
    """4 prompts (own image + text each) x 2 rollouts, prompt-major, in the reference's (N, P+R) layout."""
    c = tiny.TINY
    grids = ((1, 8, 8), (1, 4, 12), (1, 8, 4), (1, 4, 8))
    base = tiny.make_batch(c, seed=seed, grids=grids, text_lens=((5, 9), (3, 6), (7, 4), (2, 8)), response_lens=(7, 11, 5, 9), R=R, P=Pc)
    rs = np.random.RandomState(seed + 1)
    n = len(grids)
    ids = np.repeat(base["input_ids"], 2, 0).copy()
    mask = np.repeat(base["attention_mask"], 2, 0).copy()
    for r in range(1, 2 * n, 2):                                  # second rollout of every prompt: other tokens, other length
        L = int(rs.randint(3, R + 1))
        ids[r, Pc:] = tiny.PAD_ID; mask[r, Pc:] = 0
        ids[r, Pc:Pc + L] = rs.randint(0, 900, L - 1).tolist() + [tiny.EOS_ID]; mask[r, Pc:Pc + L] = 1
    off = np.concatenate([[0], np.cumsum(base["patch_counts"])])
    px = [torch.from_numpy(base["pixel_values"][off[i]:off[i + 1]]) for i in range(n)]
    gr = [base["image_grid_thw"][i:i + 1] for i in range(n)]
    pos = np.stack([P.mrope_position_ids(ids[r], gr[r // 2], mask[r], image_token_id=c["image_token_id"],
                                         vision_start_token_id=c["vision_start_token_id"]) for r in range(2 * n)])
    mm = np.array([{"pixel_values": px[r // 2], "image_grid_thw": gr[r // 2]} for r in range(2 * n)], dtype=object)   # same object per pair
    return ids, mask, pos, mm, px, gr, R


def _ulps(a: torch.Tensor, b: torch.Tensor, floor: float) -> torch.Tensor:
    """|a - b| in bf16 steps AT THE MAGNITUDE max(|b|, floor): floor = the tensor's RMS, so that a weight next to zero (whose own ulp is
    arbitrarily small while the optimizer moves it by ~lr like every other weight) is measured on the tensor's scale"""
    mag = torch.clamp(b.abs(), min=floor)
    ulp = torch.exp2(torch.floor(torch.log2(mag)) - 7.0)
    return (a - b).abs() / ulp


def test_three_update_actor_calls_follow_the_oracle_trajectory(measured):
    from spatialthinker_amd import model as mdl
    from spatialthinker_amd.actor import ActorHyper, PolicyEngine
    ids, mask, pos, mm, px, gr, R = _rollout_pairs()
    N = ids.shape[0]
    rmask = mask[:, -R:]
    params = tiny.make_params()
    ocfg = Q.VLConfig(**tiny.TINY)
    t_ids, t_mask, t_pos = torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(pos)

    def logp_fn(p, rows, temperature):
        pv = torch.cat([px[r // 2] for r in rows], 0)
        g = np.concatenate([gr[r // 2] for r in rows], 0)
        return Q.response_log_probs(p, ocfg, t_ids[rows], t_mask[rows], t_pos[rows], R, temperature, pv, g)

    loop = UpdateLoop({k: torch.from_numpy(v) for k, v in params.items()}, logp_fn, KahanBF16(scalar_mode="gpu"), lr=LR, lr_warmup_steps=WARMUP,
                      mini=4, micro=2, max_grad_norm=MAX_NORM, kl_kind="low_var_kl", kl_coef=KL_COEF)
    rs = np.random.RandomState(77)
    with torch.no_grad():
        lp0 = logp_fn(loop.p, np.arange(N), TEMP).numpy()
    # old_log_probs are re-derived before every call, as in training (each update_actor follows a fresh compute_log_probs of the current
    # policy): old = current oracle log-probs + a FIXED offset pattern, so the PPO ratio of a token is exp(-offset) +- what one optimizer
    # step moves it.  The offsets stay clear of the clip boundaries (ln 0.8 = -0.223, ln 1.3 = 0.262): three quarters within +-0.04, a
    # quarter at +-(0.55..0.7) — firmly clipped tokens, firmly unclipped tokens, none that bf16 noise could push across (the PPO gradient is
    # discontinuous there; the first version of this test had such tokens and its gradient norms differed by 45 % for that reason alone).
    off = np.where(rs.rand(*lp0.shape) < 0.25, rs.choice([-1.0, 1.0], lp0.shape) * rs.uniform(0.55, 0.7, lp0.shape),
                   rs.uniform(-0.04, 0.04, lp0.shape)).astype(np.float32)
    ref = (lp0 + 0.2 * rs.standard_normal(lp0.shape)).astype(np.float32)     # the frozen reference policy's log-probs: constant
    adv = (1.5 * rs.standard_normal((N, 1)).astype(np.float32)).repeat(R, 1) * rmask

    cfg = mdl.VLConfig(**tiny.TINY)
    store = mdl.ParamStore(cfg, trainable=True)
    store.load_hf_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    store.refresh_transposes()
    eng = PolicyEngine(cfg, store, ActorHyper(micro_batch_size_per_device_for_update=2, global_batch_size_per_device=4, lr=LR,
                                              lr_warmup_steps=WARMUP, max_grad_norm=MAX_NORM, kl_coef=KL_COEF))
    t = torch.from_numpy
    data = dict(input_ids=t(ids), attention_mask=t(mask), position_ids=t(pos), responses=t(ids[:, -R:].copy()), multi_modal_inputs=mm,
                ref_log_probs=t(ref), advantages=t(adv))
    odata = dict(ref_log_probs=ref, advantages=adv, response_mask=rmask)

    worst = dict(metric=0.0, norm=0.0)
    for call in range(CALLS):
        with torch.no_grad():
            old = (logp_fn(loop.p, np.arange(N), TEMP).numpy() + off).astype(np.float32)
        data["old_log_probs"], odata["old_log_probs"] = t(old), old
        w_before = store.flat.clone()
        got = eng.update_policy(data, TEMP)
        want = loop.update_policy(odata, TEMP)
        assert eng.last_plan["update"] == [(0, 4), (4, 8)]           # both micro-batches of a mini-batch ride in ONE pass (loss_rows = 2)
        for k in ("pg_loss", "pg_clipfrac_higher", "pg_clipfrac_lower", "ppo_kl", "entropy_loss"):
            g_, w_ = np.asarray(got["actor/" + k]), np.asarray(want["actor/" + k])
            assert g_.shape == w_.shape == (4,), (call, k)
            d = float(np.abs(g_ - w_).max())
            print(f"call {call} {k}: engine {np.round(g_, 5).tolist()} oracle {np.round(w_, 5).tolist()}")
            if "clipfrac" in k:
                # a fraction of 11..17 tokens; a token whose ratio sits within the bf16 noise of a clip boundary may land on either side
                # (measured: two such tokens in one micro-batch of call 1) — the losses next to it carry the quantitative comparison
                assert d <= 2.0 / 11 + 1e-6, (call, k, g_, w_)
            else:
                worst["metric"] = max(worst["metric"], d)
        worst["metric"] = max(worst["metric"], abs(got["actor/kl_loss"] - want["actor/kl_loss"]))
        gn, wn = np.asarray(got["actor/grad_norm"]), np.asarray(want["actor/grad_norm"])
        print(f"call {call} grad_norm: engine {gn.tolist()} oracle {wn.tolist()}  lr {got['actor/lr']}")
        assert gn.shape == wn.shape == (2,)
        worst["norm"] = max(worst["norm"], float(np.abs(gn / wn - 1).max()))
        assert abs(got["actor/lr"] - want["actor/lr"]) < 1e-12 and got["actor/kl_coef"] == KL_COEF
        if call == 0:                                                # the scheduler quirk: lr = 0, weights untouched, moments updated
            assert torch.equal(w_before, store.flat) and float(store.m.float().abs().max()) > 0
            assert got["actor/lr"] == LR / 2
        else:
            assert not torch.equal(w_before, store.flat)
        assert (wn > MAX_NORM).any() or call > 0                      # clipping is active somewhere in the run
    measured("trajectory_metric_max_abs", worst["metric"])
    measured("trajectory_grad_norm_max_rel", worst["norm"])
    assert worst["metric"] <= 0.0147 and worst["norm"] <= 0.0074       # measured 0.0113 (abs, losses of magnitude 0.3 .. 7.7) and 0.0057 (relative)

    final = store.export_hf()
    comp = store.export_hf({n: store._view(store.c, n) for n in store.layout})           # the Kahan compensation buffers, HF names
    ulp_max, same, moved, rels, eff = 0.0, [], [], [], []
    for name, w0 in params.items():
        a = final[name].float().cpu()
        b = loop.p[name].detach()
        w0 = torch.from_numpy(w0)
        rms = float(w0.pow(2).mean().sqrt())
        u = _ulps(a, b, rms)
        ulp_max = max(ulp_max, float(u.max()))
        same.append(float((a == b).float().mean()))
        moved.append(float((b != w0).float().mean()))
        # the EFFECTIVE weight p + c (what the bf16 grid hides): accumulated update of engine vs oracle
        ca = comp[name].float().cpu()
        cb = torch.from_numpy(loop.opt.state[name].c.reshape(w0.shape)) if name in loop.opt.state else torch.zeros_like(w0)
        da, db = (a + ca) - w0, (b + cb) - w0
        rels.append((float((da - db).norm() / (db.norm() + 1e-20)), name))
        eff.append(float(((a + ca) - (b + cb)).abs().max() / rms))
    rels.sort(reverse=True)
    # a key bias shifts every score of a query by the same amount, which the softmax ignores: the TRUE gradient of k_proj.bias (and of the
    # k third of the ViT's fused qkv bias) is zero, what both sides accumulate there is rounding noise — reported, not bounded
    noise = lambda n: n.endswith(("k_proj.bias", "attn.qkv.bias"))
    print("accumulated update (p + c - w_initial), relative L2 error engine vs oracle; zero-gradient (key-bias) tensors:", [r for r in rels if noise(r[1])][:3])
    rels = [r for r in rels if not noise(r[1])]
    print("... worst of the other tensors:", rels[:6])
    print(f"weights: max distance {ulp_max:.2f} bf16 ulps (at max(|w|, rms)), bit-identical fraction min {min(same):.4f} mean {np.mean(same):.4f}; "
          f"fraction of weights the oracle moved: mean {np.mean(moved):.3f}; max |d(p + c)| / rms(w) {max(eff):.2e}")
    measured("trajectory_weight_max_ulps", ulp_max)
    measured("trajectory_weight_identical_fraction_mean", float(np.mean(same)))
    measured("trajectory_update_rel_l2_worst", rels[0][0])
    measured("trajectory_update_rel_l2_median", float(np.median([r for r, _ in rels])))
    assert np.mean(moved) > 0.3                                       # the run is long enough to move a good part of the weights off their start
    # measured on MI355X: max distance 2.0 ulps, 96.2 % of all weights bit-identical to the oracle's after the 8 optimizer steps, median
    # relative error of the accumulated update 2.5 %
    assert ulp_max <= 3.0 and np.mean(same) >= 0.94 and float(np.median([r for r, _ in rels])) <= 0.033 and rels[0][0] <= 0.25
