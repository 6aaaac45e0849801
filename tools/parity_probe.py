"""Where does the engine's bf16 error come from, next to HF's own bf16 evaluation?  

Tiny Qwen2.5-VL config, N random batches (tests/golden/tiny.make_batch with other seeds), one sample per pass.  Three evaluations of the same
weights on the same inputs: HF transformers in fp32 on the GPU (the yardstick's zero), HF in bf16, and the HIP engine.  For every tap —
image features, hidden state after every LM layer, final response log-probs — the relative L2 error and max|err| / rms of both bf16
evaluations against fp32, pooled over all samples.  A max over the 18 response tokens of ONE batch (what tests/test_gpu_model.py compared
until round 6) is one draw of a heavy-tailed statistic; pooled over N batches the two evaluations can be compared.

    python tools/parity_probe.py [N=16] [--json out.json]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import tiny  # noqa: E402
from oracle import positions as P  # noqa: E402  (a tool, not the product path)


def hf_model(params, dtype):
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    c = tiny.TINY
    hc = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"], num_hidden_layers=c["num_layers"],
                         num_attention_heads=c["num_heads"], num_key_value_heads=c["num_kv_heads"], vocab_size=c["vocab_size"],
                         rms_norm_eps=c["rms_eps"], rope_parameters=dict(rope_type="default", rope_theta=c["rope_theta"], mrope_section=c["mrope_section"]),
                         tie_word_embeddings=False, max_position_embeddings=4096, bos_token_id=None, eos_token_id=tiny.EOS_ID, pad_token_id=tiny.PAD_ID),
        vision_config=dict(depth=c["v_depth"], hidden_size=c["v_hidden"], num_heads=c["v_heads"], intermediate_size=c["v_intermediate"],
                           out_hidden_size=c["hidden_size"], patch_size=c["v_patch"], spatial_merge_size=c["v_merge"],
                           temporal_patch_size=c["v_temporal_patch"], window_size=c["v_window"], fullatt_block_indexes=c["v_fullatt"],
                           in_channels=c["v_in_channels"]),
        image_token_id=c["image_token_id"], video_token_id=1009, vision_start_token_id=c["vision_start_token_id"],
        vision_end_token_id=tiny.VISION_END, tie_word_embeddings=False, bos_token_id=None, eos_token_id=tiny.EOS_ID, pad_token_id=tiny.PAD_ID)
    hc._attn_implementation = "sdpa"
    m = Qwen2_5_VLForConditionalGeneration(hc)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=False)
    return m.to(dtype).cuda().eval()


def position_ids(batch):
    ids, mask, Pn, R = batch["input_ids"], batch["attention_mask"], batch["P"], batch["R"]
    pos = np.zeros((ids.shape[0], 3, ids.shape[1]), dtype=np.int64)
    for i in range(ids.shape[0]):
        pp = P.mrope_position_ids(ids[i, :Pn], batch["image_grid_thw"][i:i + 1], mask[i, :Pn], image_token_id=tiny.TINY["image_token_id"],
                                  vision_start_token_id=tiny.TINY["vision_start_token_id"])
        pp[:, mask[i, :Pn] == 0] = 0
        pos[i, :, :Pn] = pp
        pos[i, :, Pn:] = pp[:, -1:] + np.arange(1, R + 1)
    return pos


def hf_taps(m, batch, pos, bi, off, n):
    ids, mask = batch["input_ids"], batch["attention_mask"]
    sel = mask[bi] == 1
    with torch.no_grad():
        o = m(input_ids=torch.from_numpy(ids[bi][sel])[None].cuda(), attention_mask=None,
              position_ids=torch.from_numpy(pos[bi][:, sel])[:, None, :].cuda(),
              pixel_values=torch.from_numpy(batch["pixel_values"][off:off + n]).cuda().to(next(m.parameters()).dtype),
              image_grid_thw=torch.from_numpy(batch["image_grid_thw"][bi:bi + 1]).cuda(), use_cache=False, output_hidden_states=True)
    hs = [h[0].float().cpu().numpy() for h in o.hidden_states]          # embeddings (image features merged in), then every layer; the last one is post-norm
    lg = o.logits[0].float().cpu()
    labels = torch.roll(torch.from_numpy(ids[bi][sel]), -1)
    lp = torch.log_softmax(lg, -1).gather(-1, labels[:, None])[:, 0].numpy()
    R = batch["R"]
    nr = int(mask[bi, -R:].sum())
    T = int(sel.sum())
    return hs, lp[T - nr - 1:T - 1]


def engine_taps(eng, batch, pos, bi, off, n):
    from spatialthinker_amd import ops
    R = batch["R"]
    b = eng.stage(batch["input_ids"][bi:bi + 1], batch["attention_mask"][bi:bi + 1], pos[bi:bi + 1], R, batch["pixel_values"][off:off + n],
                  batch["image_grid_thw"][bi:bi + 1])
    T = b.pk.T
    with torch.no_grad():
        x = eng._embed(b, None)
        hs = [x[:T].float().cpu().numpy()]
        for i in range(eng.cfg.num_layers):
            x = eng._lm_layer_fwd(i, x, b, None)
            hs.append(x[:T].float().cpu().numpy())
        hn, _ = ops.rmsnorm_fwd(x, eng.p.w["final_norm"], eng.cfg.rms_eps)
        hs[-1] = hn[:T].float().cpu().numpy()                              # HF's last hidden state is the normed one
        lp = eng.log_probs(b, 1.0).cpu().numpy()[0]
    nr = int(batch["attention_mask"][bi, -R:].sum())
    return hs, lp[:nr]


def hf_response_grads(m, batch, pos, glogp):
    """Gradients of sum(logp * glogp) over the response tokens through HF's own autograd (per-sample forward, gradients accumulate in the
    parameters' dtype — bf16 for the bf16 model, which is what a plain HF bf16 training step holds).  {hf name: fp32 numpy array}."""
    ids, mask, R = batch["input_ids"], batch["attention_mask"], batch["R"]
    for p_ in m.parameters():
        p_.grad = None
    off = 0
    for bi, n in enumerate(batch["patch_counts"]):
        n = int(n)
        sel = mask[bi] == 1
        o = m(input_ids=torch.from_numpy(ids[bi][sel])[None].cuda(), attention_mask=None,
              position_ids=torch.from_numpy(pos[bi][:, sel])[:, None, :].cuda(),
              pixel_values=torch.from_numpy(batch["pixel_values"][off:off + n]).cuda().to(next(m.parameters()).dtype),
              image_grid_thw=torch.from_numpy(batch["image_grid_thw"][bi:bi + 1]).cuda(), use_cache=False)
        off += n
        labels = torch.roll(torch.from_numpy(ids[bi][sel]), -1).cuda()
        lp = torch.log_softmax(o.logits[0].float(), -1).gather(-1, labels[:, None])[:, 0]
        nr, T = int(mask[bi, -R:].sum()), int(sel.sum())
        (lp[T - nr - 1:T - 1] * torch.from_numpy(np.ascontiguousarray(glogp[bi, :nr])).cuda()).sum().backward()
    return {k: p_.grad.float().cpu().numpy() for k, p_ in m.named_parameters() if p_.grad is not None}


def grad_yardstick(params, batch, pos, glogp, truth: dict) -> dict:
    """per-tensor relative L2 error of HF-bf16's gradients against `truth` (fp32 gradients of the same scalar, HF parameter names)"""
    g16 = hf_response_grads(hf_model(params, torch.bfloat16).train(False), batch, pos, glogp)
    out = {}
    for k, want in truth.items():
        if k in g16:
            out[k] = float(np.linalg.norm(g16[k] - want) / (np.linalg.norm(want) + 1e-12))
    return out


def run(n_batches: int = 16, verbose: bool = True) -> dict:
    """{taps: [...], logp: {engine: {...}, hf_bf16: {...}}, per_batch_max_ratio: [...]} — see the module docstring."""
    print_ = print if verbose else (lambda *a, **k: None)
    from spatialthinker_amd import model as mdl
    params = tiny.make_params()
    cfg = mdl.VLConfig(**tiny.TINY)
    store = mdl.ParamStore(cfg, trainable=False)
    store.load_hf_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    eng = mdl.Qwen25VL(cfg, store)
    m32, m16 = hf_model(params, torch.float32), hf_model(params, torch.bfloat16)
    L = cfg.num_layers
    names = ["embed+image"] + [f"layer {i}" for i in range(L - 1)] + ["final norm", "logp"]
    acc = {who: [dict(num=0.0, den=0.0, mx=0.0, n=0) for _ in names] for who in ("engine", "hf_bf16")}
    lp_err = {"engine": [], "hf_bf16": []}
    per_batch_max = {"engine": [], "hf_bf16": []}
    for s in range(n_batches):
        grids = ((1, 8, 8), (1, 4, 12)) if s % 2 == 0 else ((1, 6, 10), (1, 8, 4))
        batch = tiny.make_batch(seed=100 + s, grids=grids)
        pos = position_ids(batch)
        off = 0
        bm = {"engine": 0.0, "hf_bf16": 0.0}
        for bi, n in enumerate(batch["patch_counts"]):
            n = int(n)
            h32, lp32 = hf_taps(m32, batch, pos, bi, off, n)
            taps = {"hf_bf16": hf_taps(m16, batch, pos, bi, off, n), "engine": engine_taps(eng, batch, pos, bi, off, n)}
            off += n
            for who, (hs, lp) in taps.items():
                for j in range(L + 1):
                    d = hs[j] - h32[j]
                    a = acc[who][j]
                    a["num"] += float((d ** 2).sum()); a["den"] += float((h32[j] ** 2).sum()); a["n"] += d.size
                    a["mx"] = max(a["mx"], float(np.abs(d).max() / np.sqrt((h32[j] ** 2).mean())))
                d = lp - lp32
                a = acc[who][L + 1]
                a["num"] += float((d ** 2).sum()); a["den"] += float((lp32 ** 2).sum()); a["n"] += d.size
                a["mx"] = max(a["mx"], float(np.abs(d).max()))
                lp_err[who] += list(np.abs(d))
                bm[who] = max(bm[who], float(np.abs(d).max()))
        for who in bm:
            per_batch_max[who].append(bm[who])
    print_(f"{n_batches} batches x 2 samples, tiny Qwen2.5-VL; both columns are errors against HF-fp32 on the same GPU")
    print_(f"{'tap':14s} | {'engine rel L2':>13s} {'HF-bf16 rel L2':>14s} {'ratio':>6s} | {'engine max/rms':>14s} {'HF-bf16 max/rms':>15s}")
    rows = []
    for j, nm in enumerate(names):
        e, h = acc["engine"][j], acc["hf_bf16"][j]
        re, rh = np.sqrt(e["num"] / e["den"]), np.sqrt(h["num"] / h["den"])
        print_(f"{nm:14s} | {re:13.5f} {rh:14.5f} {re / rh:6.2f} | {e['mx']:14.4f} {h['mx']:15.4f}")
        rows.append(dict(tap=nm, engine_rel_l2=re, hf_bf16_rel_l2=rh, engine_max=e["mx"], hf_bf16_max=h["mx"]))
    le, lh = np.array(lp_err["engine"]), np.array(lp_err["hf_bf16"])
    stats = {}
    for who, v in (("engine", le), ("hf_bf16", lh)):
        stats[who] = dict(tokens=int(v.size), rms=float(np.sqrt((v ** 2).mean())), mean_abs=float(v.mean()), p99=float(np.percentile(v, 99)),
                          max=float(v.max()), mean_of_per_batch_max=float(np.mean(per_batch_max[who])),
                          per_batch_max=[round(x, 4) for x in per_batch_max[who]])
        print_(f"log-prob |err| {who:8s}: rms {stats[who]['rms']:.5f}  mean {stats[who]['mean_abs']:.5f}  p99 {stats[who]['p99']:.4f}  max {stats[who]['max']:.4f}  "
              f"mean of per-batch max {stats[who]['mean_of_per_batch_max']:.4f}")
    ratio = np.array(per_batch_max["engine"]) / np.array(per_batch_max["hf_bf16"])
    print_(f"per-batch max ratio engine / HF-bf16: min {ratio.min():.2f}  median {np.median(ratio):.2f}  max {ratio.max():.2f}   (the old test's statistic: one draw of this)")
    return dict(batches=n_batches, taps=rows, logp=stats, per_batch_max_ratio=[round(float(r), 3) for r in ratio])


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    res = run(n_batches)
    # gradient yardstick on the default tiny batch: HF-bf16's autograd gradients of sum(logp * g) against HF-fp32's, per tensor
    params = tiny.make_params()
    batch = tiny.make_batch()
    pos = position_ids(batch)
    rs = np.random.RandomState(5)
    g = (rs.standard_normal((batch["input_ids"].shape[0], batch["R"])) * 0.05).astype(np.float32) * batch["attention_mask"][:, -batch["R"]:]
    truth = hf_response_grads(hf_model(params, torch.float32), batch, pos, g)
    rel = grad_yardstick(params, batch, pos, g, truth)
    lm = [v for k, v in rel.items() if k.endswith(("q_proj.weight", "down_proj.weight")) and "language_model" in k]
    res["grad_hf_bf16_rel_l2"] = {"worst": max(rel.values()), "median": float(np.median(list(rel.values()))), "worst_lm_q_and_down_proj": max(lm),
                                  "per_tensor": {k: round(v, 5) for k, v in sorted(rel.items())}}
    print(f"HF-bf16 gradient error vs HF-fp32 ({len(rel)} tensors): worst {max(rel.values()):.4f}, median {np.median(list(rel.values())):.4f}, "
          f"worst over the LM q_proj / down_proj weights {max(lm):.4f}")
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
