#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json in the BUILD CONTAINER (needs /root/reference and HF
transformers; neither exists on the GPU box — only the outputs travel).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Sources of truth:
  * the reference's own importable pure functions (verl.trainer.core_algos,
    verl.utils.torch_functional, verl.models.transformers.qwen2_vl.get_rope_index,
    verl.utils.seqlen_balancing (tensordict stubbed), reward_score/*.py loaded by path with
    spacy / mathruler stubbed — the stub similarity is recorded in the fixture);
  * HF transformers Qwen2_5_VLForConditionalGeneration (fp32, CPU, sdpa) on the tiny config
    of tests/golden/tiny.py.
Only inputs + expected outputs are written; no reference source text is stored.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import tiny  # noqa: E402


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrays.items()})


# ------------------------------------------------------------------ 1. RL math
def gen_rl_math():
    from verl.trainer import core_algos
    from verl.utils import torch_functional as VF

    g = torch.Generator().manual_seed(0)
    out = {}
    # GRPO advantage: groups of 4/8/16 with shuffled uids, one zero-variance group
    for G in (4, 8, 16):
        n_prompt, R = 5, 9
        N = n_prompt * G
        scores = torch.rand(N, generator=g)
        uid = np.repeat(np.arange(n_prompt), G)
        perm = torch.randperm(N, generator=g).numpy()
        uid = uid[perm]
        scores[torch.from_numpy(uid == 0)] = 0.7           # zero-variance group -> advantage 0
        lens = torch.randint(1, R + 1, (N,), generator=g)
        mask = (torch.arange(R)[None, :] < lens[:, None]).long()
        rewards = torch.zeros(N, R)
        rewards[torch.arange(N), lens - 1] = scores
        adv, ret = core_algos.compute_grpo_outcome_advantage(rewards.clone(), mask, uid.astype(object))
        out[f"grpo{G}_rewards"], out[f"grpo{G}_mask"], out[f"grpo{G}_uid"] = rewards.numpy(), mask.numpy(), uid
        out[f"grpo{G}_adv"] = adv.numpy()
    # policy loss / KL, with forced clip branches
    B, R = 6, 40
    old = -torch.rand(B, R, generator=g) * 3
    new = old + torch.randn(B, R, generator=g) * 0.4
    new[0, :4] = old[0, :4] + torch.tensor([1.5, -1.5, 2.0, 0.0])      # ratio>3 & ratio<0.8
    adv = torch.randn(B, R, generator=g)
    adv[0, :4] = torch.tensor([-1.0, 1.0, -2.0, 0.5])
    ref = old + torch.randn(B, R, generator=g) * 0.3
    ref[1, :3] = old[1, :3] + torch.tensor([4.0, -12.0, 0.0])           # clamp edges of low_var_kl / chi2
    lens = torch.randint(1, R + 1, (B,), generator=g)
    mask = (torch.arange(R)[None, :] < lens[:, None]).long()
    new_ = new.clone().requires_grad_(True)
    pg, fh, fl, pk = core_algos.compute_policy_loss(old, new_, adv, mask, 0.2, 0.3, 3.0)
    pg.backward()
    out.update(pl_old=old.numpy(), pl_new=new.numpy(), pl_adv=adv.numpy(), pl_ref=ref.numpy(), pl_mask=mask.numpy(),
               pl_out=np.array([pg.item(), fh.item(), fl.item(), pk.item()], dtype=np.float32),
               pl_grad=new_.grad.numpy())
    for kind in ("kl", "abs", "mse", "low_var_kl", "chi2"):
        new_ = new.clone().requires_grad_(True)
        kld = core_algos.compute_kl(new_, ref, kind)
        VF.masked_mean(kld, mask).backward()
        out[f"kl_{kind}"] = kld.detach().numpy()
        out[f"kl_{kind}_mean_grad"] = new_.grad.numpy()
    # full micro-batch loss as dp_actor.update_policy composes it (dp_actor.py:252-278)
    new_ = new.clone().requires_grad_(True)
    pg, fh, fl, pk = core_algos.compute_policy_loss(old, new_, adv, mask, 0.2, 0.3, 3.0)
    kl_loss = VF.masked_mean(core_algos.compute_kl(new_, ref, "low_var_kl"), mask)
    loss = (pg + kl_loss * 1e-2) / 4
    loss.backward()
    out.update(mb_metrics=np.array([(pg + kl_loss * 1e-2).item(), kl_loss.item(), -VF.masked_mean(new, mask).item()], dtype=np.float32),
               mb_grad=new_.grad.numpy())
    # response mask with multi-EOS list; masked_mean
    resp = torch.randint(0, 20, (5, 16), generator=g)
    out.update(rm_ids=resp.numpy(), rm_single=VF.get_response_mask(resp, 3).numpy(),
               rm_multi=VF.get_response_mask(resp, [3, 7]).numpy())
    # log-probs with flash-attn CE semantics (= -cross_entropy) on bf16-rounded logits
    for V in (512, 152064):
        T = 5 if V > 1000 else 33
        z = (torch.randn(T, V, generator=g) * 3).bfloat16()
        lab = torch.randint(0, V, (T,), generator=g)
        lp = -torch.nn.functional.cross_entropy(z.float(), lab, reduction="none")
        out[f"lp{V}_seed_shape"] = np.array([T, V])
        out[f"lp{V}_logits_bf16_bits"] = z.view(torch.int16).numpy() if V <= 1000 else np.zeros(0)
        out[f"lp{V}_labels"], out[f"lp{V}_logp"] = lab.numpy(), lp.numpy()
        if V > 1000:   # too large to store: regenerate from the numpy stream below
            rs = np.random.RandomState(V)
            zz = torch.from_numpy((rs.standard_normal((T, V)) * 3).astype(np.float32)).bfloat16()
            lab = torch.from_numpy(rs.randint(0, V, size=T))
            out[f"lp{V}_labels"] = lab.numpy()
            out[f"lp{V}_logp"] = (-torch.nn.functional.cross_entropy(zz.float(), lab, reduction="none")).numpy()
    save("rl_math", **out)


# ------------------------------------------------------------------ 2. AdamW (bf16 + Kahan)
def gen_adamw():
    from verl.utils.torch_functional import AnyPrecisionAdamW, get_constant_schedule_with_warmup

    rs = np.random.RandomState(5)
    n = 4096
    p0 = torch.from_numpy((rs.standard_normal(n) * 0.02).astype(np.float32)).bfloat16()
    p = torch.nn.Parameter(p0.clone())
    opt = AnyPrecisionAdamW([p], lr=1e-6, betas=(0.9, 0.999), weight_decay=1e-2)
    sched = get_constant_schedule_with_warmup(opt, num_warmup_steps=0)
    out = {"p0": p0.float().numpy()}
    lrs = []
    for call in range(3):              # three update_actor calls, 2 optimizer steps each
        for s in range(2):
            gr = torch.from_numpy((rs.standard_normal(n) * (1e-3 if call < 2 else 1e-1)).astype(np.float32)).bfloat16()
            p.grad = gr.clone()
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            k = call * 2 + s
            st = opt.state[p]
            out[f"g{k}"] = gr.float().numpy()
            out[f"p{k + 1}"] = p.detach().float().numpy()
            out[f"m{k + 1}"] = st["exp_avg"].float().numpy()
            out[f"v{k + 1}"] = st["exp_avg_sq"].float().numpy()
            out[f"c{k + 1}"] = st["compensation"].float().numpy()
        sched.step()                   # fsdp_workers.py:453 — once per update_actor call
    out["lrs"] = np.asarray(lrs, dtype=np.float64)
    save("adamw", **out)


# ------------------------------------------------------------------ 3. positions / balancing
class _FakeTok:
    def __init__(self, m):
        self.m = m

    def convert_tokens_to_ids(self, t):
        return self.m[t]


def gen_positions():
    from verl.models.transformers.qwen2_vl import get_rope_index

    proc = types.SimpleNamespace(
        tokenizer=_FakeTok({"<|image_pad|>": 990, "<|video_pad|>": 989, "<|vision_start|>": 991}),
        image_processor=types.SimpleNamespace(merge_size=2))
    cases, out = [], {}
    rs = np.random.RandomState(3)

    def seq(text_a, grids, text_b, pad):
        toks = rs.randint(0, 900, size=text_a).tolist()
        for (t, h, w), gap in grids:
            toks += [991] + [990] * (t * h * w // 4) + [992] + rs.randint(0, 900, size=gap).tolist()
        toks += rs.randint(0, 900, size=text_b).tolist()
        ids = [993] * pad + toks
        mask = [0] * pad + [1] * len(toks)
        return np.asarray(ids), np.asarray(mask)

    specs = [(4, [], 6, 3), (2, [((1, 8, 8), 3)], 5, 0), (5, [((1, 4, 12), 2), ((1, 6, 4), 0)], 4, 7),
             (0, [((1, 2, 2), 1)], 1, 2)]
    for i, (a, grids, b, pad) in enumerate(specs):
        ids, mask = seq(a, grids, b, pad)
        thw = torch.tensor([g for g, _ in grids], dtype=torch.long) if grids else None
        pos = get_rope_index(proc, torch.from_numpy(ids), image_grid_thw=thw, attention_mask=torch.from_numpy(mask))
        out[f"rope{i}_ids"], out[f"rope{i}_mask"] = ids, mask
        out[f"rope{i}_thw"] = thw.numpy() if thw is not None else np.zeros((0, 3), dtype=np.int64)
        out[f"rope{i}_pos"] = pos.numpy()
    # video blocks (and image + video mixed), with and without second_per_grid_ts (:88-101, :117-118)
    def vseq(parts, pad):
        toks = []
        for kind, val in parts:
            if kind == "text":
                toks += rs.randint(0, 900, size=val).tolist()
            else:
                t, h, w = val
                toks += [991] + [990 if kind == "image" else 989] * (t * h * w // 4) + [992]
        return np.asarray([993] * pad + toks), np.asarray([0] * pad + [1] * len(toks))

    vspecs = [([("text", 3), ("video", (4, 4, 4)), ("text", 5)], 2, None),
              ([("text", 2), ("video", (3, 4, 8)), ("text", 1), ("image", (1, 4, 4)), ("text", 4)], 0, [0.5]),
              ([("image", (1, 8, 4)), ("text", 2), ("video", (2, 4, 4)), ("video", (5, 2, 4)), ("text", 3)], 5, [1.5, 0.4])]
    for i, (parts, pad, secs) in enumerate(vspecs):
        ids, mask = vseq(parts, pad)
        img = [v for k_, v in parts if k_ == "image"]
        vid = [v for k_, v in parts if k_ == "video"]
        pos = get_rope_index(proc, torch.from_numpy(ids), image_grid_thw=torch.tensor(img, dtype=torch.long) if img else None,
                             video_grid_thw=torch.tensor(vid, dtype=torch.long), second_per_grid_ts=torch.tensor(secs) if secs else None,
                             attention_mask=torch.from_numpy(mask))
        out[f"ropev{i}_ids"], out[f"ropev{i}_mask"], out[f"ropev{i}_pos"] = ids, mask, pos.numpy()
        out[f"ropev{i}_img"] = np.asarray(img, dtype=np.int64).reshape(-1, 3)
        out[f"ropev{i}_vid"] = np.asarray(vid, dtype=np.int64).reshape(-1, 3)
        out[f"ropev{i}_secs"] = np.asarray(secs if secs else [], dtype=np.float32)
    # HF vision index helpers
    from transformers.vision_utils import get_vision_position_ids, get_vision_window_index

    for i, grids in enumerate([[(1, 8, 8)], [(1, 4, 12), (1, 6, 4)], [(1, 32, 42)], [(1, 16, 16), (1, 10, 6)]]):
        thw = torch.tensor(grids)
        for win in (56, 112):
            wi, cu = get_vision_window_index(thw, spatial_merge_size=2, window_size=win, patch_size=14)
            out[f"vwin{i}_{win}_idx"], out[f"vwin{i}_{win}_cu"] = wi.numpy(), cu.numpy()
        out[f"vwin{i}_thw"] = thw.numpy()
        out[f"vwin{i}_pos"] = get_vision_position_ids(thw, 2).numpy()
    # Karmarkar-Karp (pure-python part of seqlen_balancing; tensordict import stubbed)
    sys.modules.setdefault("tensordict", types.SimpleNamespace(TensorDict=object))
    from verl.utils.seqlen_balancing import get_seqlen_balanced_partitions

    kk = []
    for n, k in ((8, 2), (16, 4), (64, 8), (32, 8), (24, 4)):
        lens = rs.randint(300, 2200, size=n).tolist()
        if n == 24:
            lens = [1000] * 12 + lens[:12]          # many ties
        kk.append({"lens": lens, "k": k, "parts": get_seqlen_balanced_partitions(lens, k, equal_size=True)})
    with open(os.path.join(HERE, "balance.json"), "w") as f:
        json.dump(kk, f)
    save("positions", **out)


# ------------------------------------------------------------------ 4. rewards
def _stub_similarity_modules():
    """spaCy `en_core_web_md` and `mathruler` are absent here: stand-ins so the
    reference files can be loaded by path.  Similarity stub = 1.0 for identical cleaned labels,
    else 0.0; grade_answer stub = whitespace/case-normalised string equality.  Parity for these
    two sub-terms is therefore UNPINNED; everything else in the scorers is pinned."""
    class _Doc:
        def __init__(self, t):
            self.t = t

        def similarity(self, o):
            return _SIM_FN[0](self.t, o.t)

    spacy = types.ModuleType("spacy")
    spacy.load = lambda *a, **k: (lambda tok: _Doc(tok))
    sys.modules["spacy"] = spacy
    mr = types.ModuleType("mathruler")
    gr = types.ModuleType("mathruler.grader")
    gr.grade_answer = lambda a, b: a.strip().lower() == b.strip().lower()
    gr.extract_boxed_content = lambda s: s
    mr.grader = gr
    sys.modules["mathruler"], sys.modules["mathruler.grader"] = mr, gr


_SIM_FN = [lambda a, b: 1.0 if a == b else 0.0]          # what the stubbed spaCy Doc.similarity evaluates (swapped by gen_rewards_graded)


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reward_cases():
    gt_scene = {"objects": [{"id": "dog.1", "bbox": [10, 20, 110, 220]}, {"id": "ball.2", "bbox": [200, 210, 260, 270]},
                            {"id": "tree.3", "bbox": [300, 5, 480, 330]}],
                "relationships": [{"subject": "dog.1", "predicate": "next to", "object": "ball.2"}]}
    gt = f"<scene>{json.dumps(gt_scene)}</scene>\n<answer>(B) left of the tree</answer>"
    gt_norel = f"<scene>{json.dumps({'objects': gt_scene['objects'][:2]})}</scene>\n<answer>(A) yes</answer>"
    problem = "Describe. Image size: (500 x 375)\nQ. where is the dog?\nOptions: (A) x (B) left of the tree"

    def resp(scene, answer="(B) left of the tree", observe="<observe>a dog</observe>", think="<think>hmm</think>", extra=""):
        sc = scene if isinstance(scene, str) else json.dumps(scene)
        return f"{observe}\n<scene>{sc}</scene>\n{think}\n<answer>{answer}</answer>{extra}"

    near = {"objects": [{"id": "dog.1", "bbox": [12, 22, 108, 215]}, {"id": "ball.2", "bbox": [190, 200, 250, 280]},
                        {"id": "tree.3", "bbox": [310, 0, 470, 300]}],
            "relationships": [{"subject": "dog.1", "predicate": "beside", "object": "ball.2"}]}
    cases = [
        ("exact", resp(gt_scene), gt, problem),
        ("near", resp(near), gt, problem),
        ("wrong_answer", resp(near, answer="(A) x"), gt, problem),
        ("case_answer", resp(near, answer="  (b) LEFT of the tree "), gt, problem),
        ("fewer_objs", resp({"objects": near["objects"][:1], "relationships": []}), gt, problem),
        ("more_objs", resp({"objects": near["objects"] + [{"id": "cat.9", "bbox": [1, 2, 3, 4]}], "relationships": near["relationships"]}), gt, problem),
        ("swapped_labels", resp({"objects": [{"id": "ball.7", "bbox": [12, 22, 108, 215]}, {"id": "dog.4", "bbox": [190, 200, 250, 280]}], "relationships": []}), gt, problem),
        ("dup_ids", resp({"objects": [near["objects"][0], near["objects"][0]], "relationships": []}), gt, problem),
        ("bad_json", resp("{not json"), gt, problem),
        ("scene_list", resp("[1,2]"), gt, problem),
        ("extra_key", resp({"objects": [{"id": "dog.1", "bbox": [1, 2, 3, 4], "color": "red"}], "relationships": []}), gt, problem),
        ("rel_extra_key", resp({"objects": near["objects"], "relationships": [{"subject": "dog.1", "predicate": "on", "object": "ball.2", "conf": 1}]}), gt, problem),
        ("bad_id", resp({"objects": [{"id": "dog1", "bbox": [1, 2, 3, 4]}], "relationships": []}), gt, problem),
        ("bbox3", resp({"objects": [{"id": "dog.1", "bbox": [1, 2, 3]}], "relationships": []}), gt, problem),
        ("bbox_str", resp({"objects": [{"id": "dog.1", "bbox": [1, 2, 3, "4"]}], "relationships": []}), gt, problem),
        ("no_observe", resp(near, observe=""), gt, problem),
        ("two_think", resp(near, think="<think>a</think><think>b</think>"), gt, problem),
        ("no_answer_tag", "<observe>x</observe><scene>{}</scene><think>t</think>(B)", gt, problem),
        ("empty_scene", resp({}), gt, problem),
        ("no_objects_key", resp({"relationships": []}), gt, problem),
        ("objs_not_list", resp({"objects": "dog", "relationships": []}), gt, problem),
        ("norel_gt_match", resp({"objects": near["objects"][:2], "relationships": []}, answer="(A) yes"), gt_norel, problem),
        ("norel_gt_pred_rel", resp(near, answer="(A) yes"), gt_norel, problem),
        ("degenerate_box", resp({"objects": [{"id": "dog.1", "bbox": [10, 10, 10, 10]}, {"id": "ball.2", "bbox": [0, 0, 0, 5]}], "relationships": []}), gt, problem),
        ("float_boxes", resp({"objects": [{"id": "dog_big.1", "bbox": [10.5, 20.25, 110.0, 220.75]}], "relationships": []}), gt, problem),
        ("hyphen_label", resp({"objects": [{"id": "fire_hydrant.1", "bbox": [10, 20, 110, 220]}], "relationships": []}),
         f"<scene>{json.dumps({'objects': [{'id': 'fire_hydrant.1', 'bbox': [10, 20, 110, 220]}]})}</scene><answer>(B) left of the tree</answer>", problem),
        ("multiline", resp(json.dumps(near, indent=2)), gt, problem),
        ("trailing_text", resp(near, extra=" and more text"), gt, problem),
        ("other_size", resp(near), gt, "Image size: (1024 x 768) Q."),
        ("empty", "", gt, problem),
    ]
    return cases


def gen_rewards():
    _stub_similarity_modules()
    sgg = _load(os.path.join(REF, "verl/utils/reward_score/spatial_sgg.py"), "ref_spatial_sgg")
    r1v = _load(os.path.join(REF, "verl/utils/reward_score/r1v.py"), "ref_r1v")
    r1vs = _load(os.path.join(REF, "verl/utils/reward_score/r1v_scene.py"), "ref_r1v_scene")
    rows = []
    for name, pred, gt, problem in reward_cases():
        rows.append({"name": name, "predict": pred, "ground_truth": gt, "problem": problem,
                     "spatial_sgg": sgg.spatial_sgg_compute_score(pred, gt, problem),
                     "r1v_scene": r1vs.r1v_scene_compute_score(pred, gt)})
    r1v_rows = []
    for pred, gt in [("<think>a</think> <answer>(B) cat</answer>", "(B) cat"), ("<think>a</think><answer>dog</answer>", "<answer>Dog</answer>"),
                     ("<answer>x</answer>", "x"), ("<think>t</think>\n\n<answer>3</answer> tail", "3"), ("junk", "junk"),
                     ("<think>multi\nline</think>\n<answer>a\nb</answer>", "a\nb")]:
        r1v_rows.append({"predict": pred, "ground_truth": gt, "r1v": r1v.r1v_compute_score(pred, gt)})
    boxes = []
    rs = np.random.RandomState(11)
    for _ in range(24):
        a = np.sort(rs.rand(2, 2), axis=0).T.reshape(-1)[[0, 2, 1, 3]].tolist()   # x1,y1,x2,y2
        b = np.sort(rs.rand(2, 2), axis=0).T.reshape(-1)[[0, 2, 1, 3]].tolist()
        boxes.append({"a": a, "b": b, "ciou": sgg.compute_ciou(a, b)})
    boxes.append({"a": [0.1, 0.1, 0.1, 0.1], "b": [0.2, 0.2, 0.2, 0.2], "ciou": sgg.compute_ciou([0.1] * 4, [0.2] * 4)})
    err = None
    try:
        sgg.spatial_sgg_compute_score("x", "y", "no size here")
    except ValueError as e:
        err = str(e)
    with open(os.path.join(HERE, "rewards.json"), "w") as f:
        json.dump({"similarity_stub": "1.0 iff cleaned labels identical else 0.0 (spaCy vectors absent: UNPINNED)",
                   "grade_answer_stub": "strip().lower() equality (mathruler absent: UNPINNED)",
                   "cases": rows, "r1v": r1v_rows, "ciou": boxes, "missing_size_error": err}, f, indent=1)
    print("wrote rewards.json", len(rows))


def graded_reward_cases():
    """Cases whose Hungarian assignment depends on a FRACTIONAL label similarity (reference spatial_sgg.py:150-160: cost =
    2 (1 - sim) + 1 (1 - ciou)): near-synonym / plural / compound labels and boxes that compete for the same ground-truth object."""
    problem = "Scene. Image size: (640 x 480)\nQ. what is left of the bench?\nOptions: (A) a dog (B) a tree"

    def scene(objs, rels=()):
        return {"objects": [{"id": i, "bbox": b} for i, b in objs],
                "relationships": [{"subject": s, "predicate": p, "object": o} for s, p, o in rels]}

    def doc(sc, answer="(A) a dog"):
        return f"<scene>{json.dumps(sc)}</scene>\n<answer>{answer}</answer>"

    def resp(sc, answer="(A) a dog"):
        return f"<observe>a park</observe>\n<scene>{json.dumps(sc)}</scene>\n<think>look</think>\n<answer>{answer}</answer>"

    A, B, C, D = [40, 60, 200, 300], [220, 80, 380, 320], [400, 40, 620, 440], [30, 320, 300, 460]
    Aj, Bj, Cj, Dj = [48, 70, 190, 290], [230, 90, 370, 300], [410, 60, 600, 420], [50, 330, 280, 450]
    gt1 = scene([("dog.1", A), ("dogs.2", B), ("tree.3", C)], [("dog.1", "next to", "tree.3")])
    gt2 = scene([("park_bench.1", D), ("bench.2", B), ("trash-can.3", C), ("man.4", A)],
                [("man.4", "sitting on", "bench.2"), ("trash-can.3", "beside", "park_bench.1")])
    gt3 = scene([("car.1", A), ("cart.2", B), ("card.3", C), ("cat.4", D)], [("cat.4", "under", "car.1"), ("cart.2", "behind", "card.3")])
    cases = []

    def add(name, pred_scene, gt_scene, answer="(A) a dog"):
        cases.append((name, resp(pred_scene, answer), doc(gt_scene), problem))

    # 1-6: plural / singular: the label term outweighs a better box
    add("plural_swap_boxes", scene([("dogs.5", Aj), ("dog.6", Bj), ("tree.7", Cj)], [("dog.6", "near", "tree.7")]), gt1)
    add("plural_same_boxes", scene([("dogs.5", Bj), ("dog.6", Aj), ("trees.7", Cj)], [("dog.6", "next to", "trees.7")]), gt1)
    add("puppy_vs_dog", scene([("puppy_dog.1", Aj), ("doge.2", Bj), ("tree.3", Cj)]), gt1)
    add("one_pred_two_similar_gt", scene([("dog.9", [130, 70, 290, 310])]), gt1)
    add("three_dogs", scene([("dog.1", Cj), ("dog.2", Bj), ("dog.3", Aj)]), gt1)
    add("tree_typos", scene([("treee.1", Aj), ("tre.2", Bj), ("street.3", Cj)]), gt1)
    # 7-13: compound labels (underscore / hyphen cleaning happens before the similarity)
    add("bench_vs_park_bench", scene([("bench.1", Dj), ("park-bench.2", Bj), ("trash_can.3", Cj), ("woman.4", Aj)],
                                     [("woman.4", "sitting on", "park-bench.2"), ("trash_can.3", "next to", "bench.1")]), gt2)
    add("bench_boxes_crossed", scene([("park bench.1", Bj), ("bench.2", Dj), ("can.3", Cj), ("man.4", Aj)],
                                     [("man.4", "sits on", "bench.2")]), gt2)
    add("trashcan_fused", scene([("trashcan.3", Cj), ("human.4", Aj), ("benches.7", Bj)], [("human.4", "on", "benches.7")]), gt2)
    add("only_bench", scene([("bench.1", [120, 200, 340, 400])]), gt2)
    add("more_preds_than_gt", scene([("bench.1", Dj), ("bench.2", Bj), ("bench.3", Cj), ("bench.4", Aj), ("park_bench.5", Dj), ("man.6", Aj)],
                                    [("man.6", "sitting on", "bench.2"), ("bench.3", "beside", "park_bench.5"), ("man.6", "beside", "bench.4")]), gt2)
    add("rel_predicate_graded", scene([("park_bench.1", Dj), ("bench.2", Bj), ("trash-can.3", Cj), ("man.4", Aj)],
                                      [("man.4", "sitting", "bench.2"), ("trash-can.3", "besides", "park_bench.1")]), gt2)
    add("rel_swapped_roles", scene([("park_bench.1", Dj), ("bench.2", Bj), ("trash-can.3", Cj), ("man.4", Aj)],
                                   [("bench.2", "sitting on", "man.4"), ("park_bench.1", "beside", "trash-can.3")]), gt2)
    # 14-22: one-letter neighbours car / cart / card / cat — every pair has a different fractional similarity
    add("c4_identity_jitter", scene([("car.1", Aj), ("cart.2", Bj), ("card.3", Cj), ("cat.4", Dj)], [("cat.4", "under", "car.1"), ("cart.2", "behind", "card.3")]), gt3)
    add("c4_rotated_labels", scene([("cart.1", Aj), ("card.2", Bj), ("cat.3", Cj), ("car.4", Dj)], [("car.4", "under", "cart.1")]), gt3)
    add("c4_reversed_boxes", scene([("car.1", Dj), ("cart.2", Cj), ("card.3", Bj), ("cat.4", Aj)], [("cat.4", "below", "car.1"), ("cart.2", "in front of", "card.3")]), gt3)
    add("c4_all_cars", scene([("cars.1", Aj), ("cars.2", Bj), ("cars.3", Cj), ("cars.4", Dj)]), gt3)
    add("c4_two_preds", scene([("carts.1", [100, 70, 300, 310]), ("cats.2", [200, 200, 500, 450])], [("cats.2", "under", "carts.1")]), gt3)
    add("c4_unrelated", scene([("zebra.1", Aj), ("lamp.2", Bj), ("sky.3", Cj), ("road.4", Dj)], [("zebra.1", "on", "road.4")]), gt3)
    add("c4_scart", scene([("scart.1", Bj), ("carton.2", Aj), ("cardboard.3", Cj), ("category.4", Dj)]), gt3)
    add("c4_rel_subject_graded", scene([("car.1", Aj), ("cart.2", Bj), ("card.3", Cj), ("cat.4", Dj)], [("cats.4", "underneath", "cars.1"), ("carts.2", "behind", "cards.3")]), gt3)
    add("c4_wrong_answer", scene([("card.1", Aj), ("car.2", Bj), ("cart.3", Cj), ("cat.4", Dj)]), gt3, answer="(B) a tree")
    # 23-26: empty / case / whitespace labels through the cleaning step
    add("case_and_space", scene([("Dog.1", Aj), ("DOGS.2", Bj), ("Tree_.3", Cj)]), gt1)
    add("underscore_only", scene([("_.1", Aj), ("dog_.2", Bj), ("_tree.3", Cj)]), gt1)
    add("far_boxes_right_labels", scene([("dog.1", [500, 400, 630, 470]), ("dogs.2", [10, 10, 60, 50]), ("tree.3", [300, 300, 330, 330])]), gt1)
    add("right_boxes_wrong_labels", scene([("tree.1", A), ("dog.2", B), ("dogs.3", C)]), gt1)
    # 27-50: labels never match exactly (the binary stub sees similarity 0 everywhere, so the BOXES decide its assignment); the labels are
    # near-forms of a PERMUTATION of the ground-truth labels and the boxes jittered copies of the unpermuted ground-truth boxes, so the
    # graded similarity (2 x (1 - sim)) pulls the assignment towards the permutation against the box term (1 x (1 - ciou))
    rs = np.random.RandomState(5)
    vocab = [("dog", "dogs"), ("tree", "trees"), ("bench", "benches"), ("car", "cars"), ("table", "tables"), ("lamp", "lamps"),
             ("window", "windows"), ("bottle", "bottles"), ("person", "persons"), ("bicycle", "bicycles")]
    variants = [lambda w, pl: pl, lambda w, pl: w + "_", lambda w, pl: "a-" + w, lambda w, pl: w + w[-1], lambda w, pl: w[:-1] if len(w) > 3 else pl,
                lambda w, pl: "big_" + w]
    for k in range(24):
        n = int(rs.randint(2, 6))
        words = [vocab[i] for i in rs.permutation(len(vocab))[:n]]
        boxes = []
        for i in range(n):
            x0, y0 = int(rs.randint(0, 400)), int(rs.randint(0, 280))
            boxes.append([x0, y0, x0 + int(rs.randint(60, 230)), y0 + int(rs.randint(60, 190))])
        perm = rs.permutation(n)
        if n > 1 and np.array_equal(perm, np.arange(n)):
            perm = np.roll(perm, 1)
        jit = int(rs.choice([2, 8, 20, 45]))
        preds = []
        for i in range(n):
            w, pl = words[perm[i]]
            b = [int(v + rs.randint(-jit, jit + 1)) for v in boxes[i]]
            b = [min(b[0], b[2] - 5), min(b[1], b[3] - 5), b[2], b[3]]
            preds.append((variants[int(rs.randint(len(variants)))](w, pl) + f".{i + 1}", b))
        gts = [(words[i][0] + f".{i + 1}", boxes[i]) for i in range(n)]
        g_rels = [(gts[0][0], "next to", gts[1][0])] + ([(gts[2][0], "behind", gts[0][0])] if n > 2 else [])
        p_rels = [(preds[int(np.where(perm == 0)[0][0])][0], "near to", preds[int(np.where(perm == 1)[0][0])][0])] + \
                 ([(preds[int(np.where(perm == 2)[0][0])][0], "in front of", preds[int(np.where(perm == 0)[0][0])][0]),
                   (preds[0][0], "behind", preds[1][0])] if n > 2 else [])
        add(f"perm{k}_n{n}_jit{jit}", scene(preds, p_rels), scene(gts, g_rels), answer="(A) a dog" if k % 8 else "(B) a tree")
    return cases


def gen_rewards_graded():
    """The same reference scorer under TWO label similarities: the binary stub and the graded one (tests/golden/label_sim.py).  For every
    case the fixture records both score dicts and both object assignments (reference `bi_match`), so the test can prove that the graded
    similarity really changes assignments — the branch the binary stub never reaches."""
    from label_sim import trigram_jaccard

    _stub_similarity_modules()
    sgg = _load(os.path.join(REF, "verl/utils/reward_score/spatial_sgg.py"), "ref_spatial_sgg_graded")
    r1vs = _load(os.path.join(REF, "verl/utils/reward_score/r1v_scene.py"), "ref_r1v_scene_graded")
    binary = _SIM_FN[0]
    import re as _re

    def scene_of(text):
        return json.loads(_re.search(r"<scene>(.*?)</scene>", text, _re.S).group(1))

    rows = []
    for name, pred, gt, problem in graded_reward_cases():
        row = {"name": name, "predict": pred, "ground_truth": gt, "problem": problem}
        for kind, fn in (("binary", binary), ("graded", trigram_jaccard)):
            _SIM_FN[0] = fn
            sgg._doc.cache_clear(); sgg._bi_match_cached.cache_clear()
            row[f"spatial_sgg_{kind}"] = sgg.spatial_sgg_compute_score(pred, gt, problem)
            row[f"mapping_{kind}"] = [None if v is None else int(v) for v in sgg.bi_match(scene_of(gt)["objects"], scene_of(pred)["objects"])]
            g_rel, p_rel = scene_of(gt).get("relationships", []), scene_of(pred).get("relationships", [])
            row[f"triplets_{kind}"] = [[g_rel.index(m["groundtruth"]), p_rel.index(m["prediction"]), float(m["cost"])]
                                       for m in sgg.bi_match_triplets(g_rel, p_rel)] if g_rel and p_rel else []
        row["r1v_scene"] = r1vs.r1v_scene_compute_score(pred, gt)
        rows.append(row)
    _SIM_FN[0] = binary
    sims = [[a, b, trigram_jaccard(a, b)] for a, b in [("dog", "dogs"), ("bench", "park bench"), ("car", "cart"), ("cart", "card"), ("cat", "car"),
                                                      ("trash can", "trashcan"), ("sitting on", "sits on"), ("", "dog"), ("", "")]]
    differ = sum(r["mapping_binary"] != r["mapping_graded"] for r in rows)
    with open(os.path.join(HERE, "rewards_graded.json"), "w") as f:
        json.dump({"similarity": "character-trigram Jaccard of the cleaned labels (tests/golden/label_sim.py); spaCy vectors absent: the "
                                 "reference's own numeric similarity stays UNPINNED, the code path that consumes a fractional similarity is pinned",
                   "assignments_that_differ_from_the_binary_stub": differ, "sims": sims, "cases": rows}, f, indent=1)
    print("wrote rewards_graded.json", len(rows), "cases,", differ, "with a different object assignment than under the binary stub")


def gen_rewards_helpers():
    """The other public functions of the reference's spatial_sgg.py (scale_box, refine_node_edge, compute_iou / giou / ciou, box_L1,
    is_valid_id_format, bi_match_triplets, compute_rel_score, spatial_reward — the strict variant the shipped scorer does not call),
    under the graded stand-in similarity, on the scenes of the graded reward cases and on seeded boxes."""
    from label_sim import trigram_jaccard
    import re as _re

    _stub_similarity_modules()
    sgg = _load(os.path.join(REF, "verl/utils/reward_score/spatial_sgg.py"), "ref_spatial_sgg_helpers")
    _SIM_FN[0] = trigram_jaccard
    sgg._doc.cache_clear(); sgg._bi_match_cached.cache_clear()
    rs = np.random.RandomState(33)
    boxes = []
    for _ in range(24):
        a = np.sort(rs.rand(2, 2), axis=0).T.reshape(-1)[[0, 2, 1, 3]].tolist()
        b = np.sort(rs.rand(2, 2), axis=0).T.reshape(-1)[[0, 2, 1, 3]].tolist()
        boxes.append((a, b))
    boxes += [([0.1, 0.1, 0.4, 0.4], [0.6, 0.6, 0.9, 0.9]), ([0.2, 0.2, 0.5, 0.5], [0.2, 0.2, 0.5, 0.5]), ([0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0]),
              ([0.1, 0.1, 0.1, 0.4], [0.1, 0.1, 0.1, 0.4]), ([0, 0, 1, 1], [0, 0, 2, 2])]
    box_rows = [{"a": a, "b": b, "iou": sgg.compute_iou(a, b), "giou": sgg.compute_giou(a, b), "ciou": sgg.compute_ciou(a, b), "l1": sgg.box_L1(a, b),
                 "scaled": sgg.scale_box(a, (0.5, 2.0))} for a, b in boxes]
    labels = ["Fire-Hydrant", "trash_can ", " park bench", "dog.3", "Dog", "a-b_c"]
    ids = ["dog.1", "dog", "fire_hydrant.12", "fire-hydrant.1", "Dog.07", "dog.1a", ".3", "d.3 "]
    scene_rows = []
    for name, pred, gt, problem in graded_reward_cases():
        ps = json.loads(_re.search(r"<scene>(.*?)</scene>", pred, _re.S).group(1))
        gs = json.loads(_re.search(r"<scene>(.*?)</scene>", gt, _re.S).group(1))
        w, h = sgg.extract_image_size(problem)
        o, r = sgg.spatial_reward(ps, gs, w, h)
        g_rel, p_rel = gs.get("relationships", []), ps.get("relationships", [])
        scene_rows.append({"name": name, "pred_scene": ps, "gt_scene": gs, "w": w, "h": h, "spatial_reward": [float(o), float(r)],
                           "rel_score": float(sgg.compute_rel_score(g_rel, p_rel)),
                           "triplet_similarity": [float(m["similarity"]) for m in sgg.bi_match_triplets(g_rel, p_rel)]})
    odd = [("not a dict", {"objects": []}), ({"objects": "x"}, {"objects": []}), ({"objects": [{"id": "dog", "bbox": [0, 0, 1, 1]}]}, {"objects": []}),
           ({"objects": [], "relationships": []}, {"objects": [], "relationships": []}),
           ({"objects": [{"id": "dog.1", "bbox": [0, 0, 10, 10]}]}, {"objects": [], "relationships": []})]
    odd_rows = [{"pred_scene": p_, "gt_scene": g_, "spatial_reward": [float(x) for x in sgg.spatial_reward(p_, g_, 100, 50)]} for p_, g_ in odd]
    _SIM_FN[0] = lambda a, b: 1.0 if a == b else 0.0
    with open(os.path.join(HERE, "rewards_helpers.json"), "w") as f:
        json.dump({"similarity": "character-trigram Jaccard (tests/golden/label_sim.py)", "boxes": box_rows,
                   "labels": [[l, sgg.refine_node_edge(l)] for l in labels], "ids": [[i, sgg.is_valid_id_format(i)] for i in ids],
                   "scenes": scene_rows, "odd": odd_rows}, f, indent=1)
    print("wrote rewards_helpers.json:", len(box_rows), "box pairs,", len(scene_rows), "scenes")


# ------------------------------------------------------------------ 5. HF tiny model
def _hf_tiny_model():
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration

    c = tiny.TINY
    cfg = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
                         num_hidden_layers=c["num_layers"], num_attention_heads=c["num_heads"],
                         num_key_value_heads=c["num_kv_heads"], vocab_size=c["vocab_size"], rms_norm_eps=c["rms_eps"],
                         rope_parameters=dict(rope_type="default", rope_theta=c["rope_theta"], mrope_section=c["mrope_section"]),
                         tie_word_embeddings=False, max_position_embeddings=4096, bos_token_id=None, eos_token_id=tiny.EOS_ID,
                         pad_token_id=tiny.PAD_ID),
        vision_config=dict(depth=c["v_depth"], hidden_size=c["v_hidden"], num_heads=c["v_heads"],
                           intermediate_size=c["v_intermediate"], out_hidden_size=c["hidden_size"], patch_size=c["v_patch"],
                           spatial_merge_size=c["v_merge"], temporal_patch_size=c["v_temporal_patch"],
                           window_size=c["v_window"], fullatt_block_indexes=c["v_fullatt"], in_channels=c["v_in_channels"]),
        image_token_id=c["image_token_id"], video_token_id=1009, vision_start_token_id=c["vision_start_token_id"],
        vision_end_token_id=tiny.VISION_END, tie_word_embeddings=False, bos_token_id=None, eos_token_id=tiny.EOS_ID,
        pad_token_id=tiny.PAD_ID)
    cfg._attn_implementation = "sdpa"
    model = Qwen2_5_VLForConditionalGeneration(cfg).float().eval()
    params = tiny.make_params(c)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("inv_freq" in m for m in missing), (missing, unexpected)
    return model


def gen_rewards_math():
    """The reference's `math` plug-in (verl/utils/reward_score/math.py:21-40) with mathruler's two functions stubbed by THIS build's
    documented fallbacks (last brace-matched \\boxed{}, normalised string equality): pins the clean-up regex, the format regex and the
    0.9 / 0.1 weighting; mathruler's own grading stays unpinned (not installed)."""
    # this build's module by file path: `verl` on sys.path is the REFERENCE's package in this script
    build_math = _load(os.path.join(os.path.dirname(os.path.dirname(HERE)), "verl/utils/reward_score/math.py"), "build_math")
    _stub_similarity_modules()
    sys.modules["mathruler.grader"].extract_boxed_content = build_math._boxed
    sys.modules["mathruler.grader"].grade_answer = build_math._grade
    ref = _load(os.path.join(REF, "verl/utils/reward_score/math.py"), "ref_math")
    cases = [
        ("<think>2+2</think> so \\boxed{4}", "4"), ("<think>x</think>\\boxed{ 4 }", "4"), ("<think>x</think> \\boxed{5}", "4"),
        ("\\boxed{4}", "4"), ("<think>no box</think> 4", "4"), ("< think >a< / think > \\boxed{\\frac{1}{2}}", "\\frac{1}{2}"),
        ("<think>a</think>\\boxed{1} then \\boxed{2}", "2"), ("<think>a</think>\\boxed{1} then \\boxed{2}", "1"),
        ("<think>a</think>\\boxed{x^{2}+1}", "x^{2} + 1"), ("<think>a</think>\\boxed{unclosed", "unclosed"), ("", ""),
        ("<think>\n multi\n line</think>\n\\boxed{A}\n", "a"), ("<think>a</think>\\boxed{$3$}", "3"), ("<think>a</think>\\boxed{}", ""),
        ("text <think>a</think>\\boxed{7}", "7"), ("<think>a</think>\\boxed{7} trailing words", "7"),
    ]
    cases = [(p_.replace("\\\\", "\\"), g_.replace("\\\\", "\\")) for p_, g_ in cases]
    rows = [{"predict": p_, "ground_truth": g_, "score": ref.math_compute_score(p_, g_)} for p_, g_ in cases]
    with open(os.path.join(HERE, "rewards_math.json"), "w") as f:
        json.dump({"stubs": "mathruler.grader.extract_boxed_content / grade_answer = verl.utils.reward_score.math._boxed / _grade (mathruler absent: UNPINNED)",
                   "cases": rows}, f, indent=1)
    print("rewards_math:", len(rows), "cases;", sum(r["score"]["accuracy"] for r in rows), "accurate,", sum(r["score"]["format"] for r in rows), "well-formed")


def gen_small_helpers():
    """Small helpers of verl/utils/torch_functional.py (masked_var :74-88, masked_whiten :91-94, pad_sequence_to_length :137-147) and
    core_algos.compute_rewards (:281-288), from the reference's own functions."""
    from verl.trainer import core_algos
    from verl.utils import torch_functional as VF
    g = torch.Generator().manual_seed(11)
    v = torch.randn(5, 9, generator=g)
    m = (torch.rand(5, 9, generator=g) > 0.35).float()
    one = torch.zeros(5, 9); one[2, 3] = 1.0
    out = {"v": v, "m": m, "one": one,
           "var_unbiased": VF.masked_var(v, m), "var_biased": VF.masked_var(v, m, unbiased=False), "var_one": VF.masked_var(v, one),
           "whiten": VF.masked_whiten(v, m), "whiten_eps": VF.masked_whiten(v, m, eps=1e-3)}
    ids = torch.randint(0, 50, (3, 4, 6), generator=g)
    out.update(ids=ids, pad_right=VF.pad_sequence_to_length(ids, 9, 77), pad_left=VF.pad_sequence_to_length(ids, 9, 77, left_pad=True),
               pad_noop=VF.pad_sequence_to_length(ids, 6, 77), pad_shorter=VF.pad_sequence_to_length(ids, 4, 77))
    sc, lp, rp = torch.randn(4, 7, generator=g), torch.randn(4, 7, generator=g), torch.randn(4, 7, generator=g)
    out.update(sc=sc, lp=lp, rp=rp, rewards=core_algos.compute_rewards(sc, lp, rp, 0.037))
    save("small_helpers", **{k: t.numpy() for k, t in out.items()})


def gen_balance_micro():
    """seqlen_balancing.{greedy_partition :130-147, rearrange_micro_batches :220-258, get_reverse_idx :261-267, ceildiv} from the reference
    (tensordict import stubbed; rearrange_micro_batches is fed a stand-in batch whose row slices are 1 x 1 tensors holding the row number, so
    the concatenated "micro-batches" spell out which rows went where)."""
    sys.modules.setdefault("tensordict", types.SimpleNamespace(TensorDict=object))
    from verl.utils import seqlen_balancing as SB

    class _Rows:
        def __init__(self, mask):
            self.mask = mask

        def __getitem__(self, item):
            if isinstance(item, str):
                assert item == "attention_mask"
                return self.mask
            return torch.arange(self.mask.shape[0])[item].view(-1, 1)

    rs = np.random.RandomState(21)
    cases = []
    for n, width, budget in ((8, 64, 128), (16, 96, 200), (12, 50, 50), (5, 40, 400), (20, 128, 777)):
        lens = rs.randint(1, width + 1, size=n)
        mask = (torch.arange(width)[None, :] < torch.from_numpy(lens)[:, None]).long()
        micro, idx = SB.rearrange_micro_batches(_Rows(mask), max_token_len=budget)
        assert [m.flatten().tolist() for m in micro] == [list(p) for p in idx]
        cases.append({"lens": lens.tolist(), "width": width, "max_token_len": budget, "idx": [list(map(int, p)) for p in idx]})
    greedy = []
    for n, k, eq in ((8, 2, True), (9, 3, False), (16, 4, True), (7, 3, False)):
        lens = rs.randint(10, 500, size=n).tolist()
        greedy.append({"lens": lens, "k": k, "equal_size": eq, "parts": SB.greedy_partition(lens, k, eq)})
    perm = rs.permutation(11).tolist()
    with open(os.path.join(HERE, "balance_micro.json"), "w") as f:
        json.dump({"micro": cases, "greedy": greedy, "perm": perm, "reverse": SB.get_reverse_idx(perm),
                   "ceildiv": [[a, b, SB.ceildiv(a, b)] for a, b in ((7, 2), (8, 2), (0, 5), (1, 5), (1000, 333))]}, f)
    print("balance_micro:", [len(c["idx"]) for c in cases], "micro-batches")


def gen_generate():
    """Greedy continuation by HF itself (`model.generate(do_sample=False)`, its own KV cache and rope-index bookkeeping) for the two
    tiny image+text prompts: pins the oracle's KV-cache decode (oracle.qwen25vl.generate_greedy) and, through it, the GPU rollout."""
    model = _hf_tiny_model()
    batch = tiny.make_batch(tiny.TINY)
    ids, mask, P = batch["input_ids"], batch["attention_mask"], batch["P"]
    out, off = {}, 0
    for b in range(ids.shape[0]):
        sel = mask[b, :P] == 1
        n_patch = int(batch["patch_counts"][b])
        px = torch.from_numpy(batch["pixel_values"][off:off + n_patch])
        off += n_patch
        prompt = torch.from_numpy(ids[b, :P][sel])[None]
        with torch.no_grad():
            # transformers 5.x derives the M-RoPE ids inside generate() from mm_token_type_ids (1 = image placeholder)
            seq = model.generate(input_ids=prompt, attention_mask=torch.ones_like(prompt), pixel_values=px,
                                 mm_token_type_ids=(prompt == tiny.TINY["image_token_id"]).int(),
                                 image_grid_thw=torch.from_numpy(batch["image_grid_thw"][b:b + 1]), max_new_tokens=10, do_sample=False,
                                 eos_token_id=None, pad_token_id=tiny.PAD_ID)
        out[f"greedy{b}"] = seq[0, prompt.shape[1]:].numpy()
    save("generate_tiny", **out)


def gen_model():
    from verl.models.transformers.qwen2_vl import get_rope_index

    c = tiny.TINY
    model = _hf_tiny_model()
    proc = types.SimpleNamespace(
        tokenizer=_FakeTok({"<|image_pad|>": c["image_token_id"], "<|video_pad|>": 1009, "<|vision_start|>": c["vision_start_token_id"]}),
        image_processor=types.SimpleNamespace(merge_size=2))
    batch = tiny.make_batch(c)
    ids, mask, P, R = batch["input_ids"], batch["attention_mask"], batch["P"], batch["R"]
    B = ids.shape[0]
    # position ids exactly as the dataset + rollout build them (dataset.py:233-238, vllm_rollout_spmd.py:159-170)
    pos = np.zeros((B, 3, P + R), dtype=np.int64)
    for b in range(B):
        pp = get_rope_index(proc, torch.from_numpy(ids[b, :P]), image_grid_thw=torch.from_numpy(batch["image_grid_thw"][b:b + 1]),
                            attention_mask=torch.from_numpy(mask[b, :P])).numpy()
        pp[:, mask[b, :P] == 0] = 0                         # postprocess_data left-pads position ids with 0 (torch_functional.py:150-184)
        pos[b, :, :P] = pp
        pos[b, :, P:] = pp[:, -1:] + np.arange(1, R + 1)
    # run HF per sample, un-padded (the padded-vs-packed equivalence is the build's own invariant)
    logps, logits_rows, hid = [], [], []
    taps = {}
    off = 0
    for b in range(B):
        sel = mask[b] == 1
        n_patch = int(batch["patch_counts"][b])
        px = torch.from_numpy(batch["pixel_values"][off:off + n_patch])
        off += n_patch
        with torch.no_grad():
            o = model(input_ids=torch.from_numpy(ids[b][sel])[None], attention_mask=None,
                      position_ids=torch.from_numpy(pos[b][:, sel])[:, None, :], pixel_values=px,
                      image_grid_thw=torch.from_numpy(batch["image_grid_thw"][b:b + 1]), use_cache=False,
                      output_hidden_states=True)
            vis = model.model.visual(px, grid_thw=torch.from_numpy(batch["image_grid_thw"][b:b + 1]))
        lg = o.logits[0]
        labels = torch.roll(torch.from_numpy(ids[b][sel]), -1)
        lp = torch.log_softmax(lg, -1).gather(-1, labels[:, None])[:, 0]
        full = np.zeros(P + R, dtype=np.float32)
        full[sel] = lp.numpy()
        logps.append(full[-R - 1:-1])
        logits_rows.append(lg[-3:].numpy())
        hid.append(np.stack([h[0, -1].numpy() for h in o.hidden_states]))
        taps[f"image_embeds{b}"] = vis.pooler_output.numpy()
        taps[f"vit_last{b}"] = vis.last_hidden_state.numpy()
    save("model_tiny", position_ids=pos, logp=np.stack(logps), logits_last3=np.stack(logits_rows),
         hidden_last_token=np.stack(hid), **taps)


def gen_values():
    """The critic's values (dp_critic.py:52-125) from HF itself: the tiny Qwen2.5-VL backbone's last hidden state (after the final norm, what
    a *ForTokenClassification model feeds its head — e.g. Qwen2ForTokenClassification.forward: `logits = self.score(dropout(sequence_output))`)
    through a torch.nn.Linear(H, 1) with fixed weights, scattered back to (B, S) and sliced [:, -R-1:-1].  transformers has no
    token-classification class for qwen2_5_vl (the reference's AutoModelForTokenClassification call fails for this family), so the head is
    applied here by hand exactly as those classes do.  Pins oracle.qwen25vl.response_values."""
    c = tiny.TINY
    model = _hf_tiny_model()
    z = np.load(os.path.join(HERE, "model_tiny.npz"))
    batch = tiny.make_batch(c)
    ids, mask, P, R, pos = batch["input_ids"], batch["attention_mask"], batch["P"], batch["R"], z["position_ids"]
    rs = np.random.RandomState(5)
    score = torch.nn.Linear(c["hidden_size"], 1)
    w = torch.from_numpy((rs.standard_normal((1, c["hidden_size"])) * 0.05).astype(np.float32)).bfloat16().float()
    with torch.no_grad():
        score.weight.copy_(w); score.bias.fill_(0.125)
    vals, off = [], 0
    for b in range(ids.shape[0]):
        sel = mask[b] == 1
        n_patch = int(batch["patch_counts"][b])
        px = torch.from_numpy(batch["pixel_values"][off:off + n_patch])
        off += n_patch
        with torch.no_grad():
            o = model(input_ids=torch.from_numpy(ids[b][sel])[None], attention_mask=None, position_ids=torch.from_numpy(pos[b][:, sel])[:, None, :],
                      pixel_values=px, image_grid_thw=torch.from_numpy(batch["image_grid_thw"][b:b + 1]), use_cache=False, output_hidden_states=True)
            v = score(o.hidden_states[-1][0])[:, 0]
        full = np.zeros(P + R, dtype=np.float32)
        full[sel] = v.numpy()
        vals.append(full[-R - 1:-1])
    save("values_tiny", score_weight=w.numpy(), score_bias=np.asarray([0.125], dtype=np.float32), values=np.stack(vals))


# ------------------------------------------------------------------ 6. trainer-side RL math outside the GRPO kernel + rollout post-processing
def gen_rl_extra():
    """apply_kl_penalty's pieces + KL controllers, the non-GRPO estimators, the value loss, FlopsCounter and the rollout
    post-processing, all evaluated by the reference's own importable functions (verl.trainer.core_algos,
    verl.utils.torch_functional, verl.utils.flops_counter).  verl.trainer.ray_trainer / verl.workers.rollout.vllm_rollout_spmd
    themselves need ray / vllm / tensordict: the few torch lines that glue those functions together there
    (ray_trainer.py:125-145, vllm_rollout_spmd.py:149-176) are re-executed here on the reference functions' outputs."""
    from types import SimpleNamespace

    from verl.trainer import core_algos
    from verl.utils import torch_functional as VF
    from verl.utils.flops_counter import FlopsCounter

    g = torch.Generator().manual_seed(11)
    out = {}
    # ---- apply_kl_penalty (ray_trainer.py:125-145) for every estimator kind, one AdaptiveKLController step each
    B, R = 7, 24
    old = -torch.rand(B, R, generator=g) * 3
    ref = old + torch.randn(B, R, generator=g) * 0.3
    ref[2, :3] = old[2, :3] + torch.tensor([4.0, -12.0, 0.0])
    lens = torch.randint(1, R + 1, (B,), generator=g)
    mask = (torch.arange(R)[None, :] < lens[:, None]).long()
    scores = torch.zeros(B, R)
    scores[torch.arange(B), lens - 1] = torch.rand(B, generator=g)
    out.update(klp_old=old.numpy(), klp_ref=ref.numpy(), klp_mask=mask.numpy(), klp_scores=scores.numpy())
    for kind in ("kl", "abs", "mse", "low_var_kl", "chi2"):
        ctrl = core_algos.AdaptiveKLController(init_kl_coef=0.05, target_kl=0.02, horizon=100.0)
        kld = core_algos.compute_kl(old, ref, kl_penalty=kind) * mask
        rewards = scores - ctrl.kl_coef * kld
        cur = torch.mean(VF.masked_mean(kld, mask=mask, dim=-1), dim=0).item()
        ctrl.update(current_kl=cur, n_steps=B)
        out[f"klp_{kind}_rewards"] = rewards.numpy()
        out[f"klp_{kind}_stats"] = np.array([cur, ctrl.kl_coef], dtype=np.float64)
    # controller trajectories
    ctrl = core_algos.AdaptiveKLController(init_kl_coef=0.01, target_kl=0.05, horizon=64.0)
    kls = [0.001, 0.2, 0.05, 0.049, 0.0, 1.0, 0.06]
    traj = []
    for i, k in enumerate(kls):
        ctrl.update(current_kl=k, n_steps=8 + i)
        traj.append(ctrl.kl_coef)
    fixed = core_algos.get_kl_controller(SimpleNamespace(kl_type="fixed", kl_coef=0.3, kl_horizon=0, kl_target=0))
    fixed.update(current_kl=5.0, n_steps=10)
    adaptive = core_algos.get_kl_controller(SimpleNamespace(kl_type="adaptive", kl_coef=0.3, kl_horizon=10, kl_target=0.1))
    out.update(klc_kls=np.array(kls), klc_traj=np.array(traj, dtype=np.float64), klc_fixed=np.array([fixed.kl_coef]),
               klc_adaptive_cls=np.array([type(adaptive).__name__ == "AdaptiveKLController"]))
    # ---- other estimators (core_algos.py:92-133,178-278) and the value loss (:356-391)
    N, R = 12, 10
    uid = np.repeat(np.arange(3), 4)[torch.randperm(12, generator=g).numpy()]
    lens = torch.randint(1, R + 1, (N,), generator=g)
    mask = (torch.arange(R)[None, :] < lens[:, None]).long()
    rew = torch.zeros(N, R)
    rew[torch.arange(N), lens - 1] = torch.rand(N, generator=g)
    dense_rew = torch.randn(N, R, generator=g) * mask
    values = torch.randn(N, R, generator=g)
    base = torch.rand(N, generator=g)
    adv, ret = core_algos.compute_gae_advantage_return(dense_rew, values, mask, torch.tensor(0.99), torch.tensor(0.95))
    out.update(est_uid=uid, est_mask=mask.numpy(), est_rew=rew.numpy(), est_dense_rew=dense_rew.numpy(), est_values=values.numpy(),
               est_base=base.numpy(), gae_adv=adv.numpy(), gae_ret=ret.numpy())
    adv, _ = core_algos.compute_rloo_outcome_advantage(rew.clone(), mask, uid.astype(object))
    out["rloo_adv"] = adv.numpy()
    adv, ret = core_algos.compute_reinforce_plus_plus_outcome_advantage(dense_rew, mask, torch.tensor(0.97))
    out.update(rpp_adv=adv.numpy(), rpp_ret=ret.numpy())
    adv, _ = core_algos.compute_remax_outcome_advantage(rew, base, mask)
    out["remax_adv"] = adv.numpy()
    vpred = values + torch.randn(N, R, generator=g) * 0.5
    vl, vc = core_algos.compute_value_loss(vpred, ret, values, mask, 0.3)
    out.update(vl_vpred=vpred.numpy(), vl_out=np.array([vl.item(), vc.item()], dtype=np.float32))
    out["whiten"] = VF.masked_whiten(dense_rew, mask).numpy()
    # ---- FlopsCounter._estimate_llama_flops (flops_counter.py:82-115) at the 7B and 3B text shapes
    seqlens = [1612, 1480, 1750, 1333, 1612, 1900, 1211, 1499]
    rows = []
    for (H, V, L, nkv, nq, I) in ((3584, 152064, 28, 4, 28, 18944), (2048, 151936, 36, 2, 16, 11008)):
        cfg = SimpleNamespace(model_type="qwen2_5_vl", hidden_size=H, vocab_size=V, num_hidden_layers=L, num_key_value_heads=nkv,
                              num_attention_heads=nq, intermediate_size=I)
        rows.append(FlopsCounter(cfg)._estimate_llama_flops(sum(seqlens), seqlens, 2.5))
    out.update(flops_seqlens=np.array(seqlens), flops_dt=np.array([2.5]), flops_achieved=np.array(rows, dtype=np.float64))
    # ---- rollout post-processing (vllm_rollout_spmd.py:144-188): n = 3 completions per prompt, ragged, EOS in the middle / absent
    b, P, R, n, pad, eos = 3, 9, 8, 3, 99, [7, 5]
    ids = torch.randint(10, 90, (b, P), generator=g)
    am = torch.ones(b, P, dtype=torch.long)
    for i, k in enumerate((0, 3, 5)):
        ids[i, :k] = pad
        am[i, :k] = 0
    pos = torch.clip(am.cumsum(-1) - 1, min=0)[:, None, :].repeat(1, 3, 1)
    pos[:, 1] += torch.randint(0, 3, (b, 1), generator=g) * am                       # distinct h/w rows as get_rope_index produces
    pos[:, 2] += torch.randint(0, 4, (b, 1), generator=g) * am
    comp = []
    for i in range(b * n):
        L_ = int(torch.randint(1, R + 1, (1,), generator=g))
        toks = torch.randint(10, 90, (L_,), generator=g).tolist()
        if i % 3 == 0 and L_ > 2:
            toks[L_ // 2] = 7                                                         # EOS in the middle: everything after is masked
        if i % 3 == 1:
            toks[-1] = 5                                                              # ends with the second EOS id
        comp.append(toks)
    resp = VF.pad_2d_list_to_length(comp, pad, max_length=R)
    ids_r, am_r, pos_r = (t.repeat_interleave(n, dim=0) for t in (ids, am, pos))
    delta = torch.arange(1, R + 1).view(1, -1).expand(b * n, -1).view(b * n, 1, -1).expand(b * n, 3, -1)
    pos_full = torch.cat([pos_r, pos_r[..., -1:] + delta], dim=-1)
    rmask = VF.get_response_mask(response_ids=resp, eos_token_id=eos, dtype=am.dtype)
    out.update(ro_ids=ids.numpy(), ro_mask=am.numpy(), ro_pos=pos.numpy(), ro_n=np.array([n]), ro_pad=np.array([pad]), ro_eos=np.array(eos),
               ro_completions=np.array([len(c) for c in comp]), ro_completion_tokens=np.concatenate([np.asarray(c) for c in comp]),
               ro_prompts=ids_r.numpy(), ro_responses=resp.numpy(), ro_input_ids=torch.cat([ids_r, resp], -1).numpy(),
               ro_attention_mask=torch.cat([am_r, rmask], -1).numpy(), ro_response_mask=rmask.numpy(), ro_position_ids=pos_full.numpy())
    save("rl_extra", **out)


# ------------------------------------------------------------------ 7. dataset row pipeline (verl/utils/dataset.py:34-265)
def gen_dataset():
    """Writes tests/golden/stvqa_tiny/{train,val}-00000.parquet (6 + 2 synthetic rows: STVQA-7K's columns, 3 random PNG images)
    and dataset.npz = the reference RLHFDataset's output rows for several configurations, produced with the stub tokenizer /
    processor of stub_mm.py (no real tokenizer files exist offline)."""
    import io

    import pandas as pd
    from PIL import Image

    from stub_mm import StubProcessor, StubTokenizer
    from verl.utils.dataset import RLHFDataset, collate_fn

    rs = np.random.RandomState(3)
    root = os.path.join(HERE, "stvqa_tiny")
    os.makedirs(root, exist_ok=True)

    def png(w, h):
        buf = io.BytesIO()
        Image.fromarray(rs.randint(0, 255, (h, w, 3), dtype=np.uint8)).save(buf, format="PNG")
        return {"bytes": buf.getvalue(), "path": None}

    def rows(n, tag):
        out = []
        for i in range(n):
            w, h = [(64, 48), (30, 20), (200, 120)][i % 3]                      # 30x20 is below min_pixels (up-scaled), 200x120 above max_pixels
            q = f"Image size: ({w} x {h})\nQ{tag}{i}. what is left of the {'dog' if i % 2 else 'cat'}?"
            out.append({"images": [png(w, h)], "problem": ("<image>" if i % 4 != 3 else "") + q if i % 2 == 0 else q + " <image> tail ",
                        "question_with_options": f"<image>{q}\nOptions: (A) a (B) b", "answer_option_text": f"<scene>{{}}</scene>\n<answer>(A) a{i}</answer>",
                        "answer_option_text_only": f"(A) a{i}", "extra": i})
        return out
    pd.DataFrame(rows(6, "t")).to_parquet(os.path.join(root, "train-00000-of-00001.parquet"))
    pd.DataFrame(rows(2, "v")).to_parquet(os.path.join(root, "val-00000-of-00001.parquet"))

    out = {}
    cases = {
        "spatial": dict(split="train", prompt_key="problem", answer_key="answer_option_text", image_key="images", max_prompt_length=96,
                        truncation="right", format_prompt=None, shuffle=False, mixed_data=False, text_only=False),
        "vanilla": dict(split="val", prompt_key="question_with_options", answer_key="answer_option_text_only", image_key="images",
                        max_prompt_length=160, truncation="right", format_prompt="  You FIRST think. \n", shuffle=True, mixed_data=False,
                        text_only=False),
        "mixed": dict(split="train", prompt_key="problem", answer_key="answer_option_text", image_key="images", max_prompt_length=128,
                      truncation="left", format_prompt=None, shuffle=True, mixed_data=True, text_only=False),
        "textonly": dict(split="train", prompt_key="problem", answer_key="answer_option_text", image_key="images", max_prompt_length=64,
                         truncation="right", format_prompt=None, shuffle=False, mixed_data=False, text_only=True),
    }
    for name, c in cases.items():
        ds = RLHFDataset(f"{root}@{c['split']}", StubTokenizer(), StubProcessor(), prompt_key=c["prompt_key"], answer_key=c["answer_key"],
                         image_key=c["image_key"], mixed_data=c["mixed_data"], text_only=c["text_only"], max_prompt_length=c["max_prompt_length"],
                         truncation=c["truncation"], format_prompt=c["format_prompt"], max_pixels=64 * 28 * 28 // 4, min_pixels=28 * 28 * 4,
                         shuffle=c["shuffle"], seed=5)
        out[f"{name}_len"] = np.array([len(ds)])
        for i in range(len(ds)):
            r = ds[i]
            out[f"{name}_{i}_input_ids"] = r["input_ids"].numpy()
            out[f"{name}_{i}_attention_mask"] = r["attention_mask"].numpy()
            out[f"{name}_{i}_position_ids"] = r["position_ids"].numpy()
            out[f"{name}_{i}_raw_prompt_ids"] = np.asarray(r["raw_prompt_ids"])
            out[f"{name}_{i}_ground_truth"] = np.array([r["ground_truth"]])
            out[f"{name}_{i}_extra"] = np.array([r["extra"]])
            if "multi_modal_inputs" in r:
                out[f"{name}_{i}_grid"] = r["multi_modal_inputs"]["image_grid_thw"].numpy()
                pv = r["multi_modal_inputs"]["pixel_values"].numpy()
                out[f"{name}_{i}_pix_stats"] = np.array([pv.shape[0], pv.shape[1], float(pv.sum()), float(np.abs(pv).sum()), float(pv[0, :8].sum())])
                out[f"{name}_{i}_image_size"] = np.array(r["multi_modal_data"]["image"][0].size)
        if name == "spatial":
            b = collate_fn([ds[0], ds[1]])
            out["collate_keys"] = np.array(sorted(b.keys()))
            out["collate_input_ids"] = b["input_ids"].numpy()
    save("dataset", **out)



# ------------------------------------------------------------------ 8. the actor update loop (dp_actor.update_policy on a toy LM)
def gen_update_loop():
    """The reference's DataParallelPPOActor.update_policy (verl/workers/actor/dp_actor.py:155-167, 212-292) run HERE on CPU around
    tests/golden/toy_lm.ToyLM, with torch.optim.SGD and the reference's get_constant_schedule_with_warmup stepped once per call as
    fsdp_workers.py:453-455 does.  Four consecutive calls: lr = 0 on the first (scheduler quirk), warm-up on the second, clipping active,
    and an infinite advantage in the fourth (non-finite gradient norm: the optimizer step is skipped).  Stand-ins (recorded here, none of
    them computes anything): ray / tensordict are stubbed for the import, the batch is a duck-typed DataProto (select / split by rows).
    No flash-attn in this container, so log_probs_from_logits takes the reference's torch fallback, which returns +cross-entropy
    (SURVEY.md 0.7); the test's logp_fn uses the same quantity — the loop is agnostic to it."""
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    stub("ray", ObjectRef=object, get=lambda x: x)
    stub("ray.experimental")
    stub("ray.experimental.tqdm_ray", tqdm=lambda it, **kw: it)
    stub("tensordict", TensorDict=dict)
    stub("tensordict.tensorclass", NonTensorData=object)
    from verl.utils.torch_functional import get_constant_schedule_with_warmup
    from verl.workers.actor.dp_actor import DataParallelPPOActor
    import toy_lm

    class Proto:
        def __init__(self, batch, meta):
            self.batch, self.non_tensor_batch, self.meta_info = batch, {}, meta

        def select(self, keys, nt_keys):
            return Proto({k: self.batch[k] for k in keys}, self.meta_info)

        def split(self, n):
            N = next(iter(self.batch.values())).shape[0]
            return [Proto({k: v[i:i + n] for k, v in self.batch.items()}, self.meta_info) for i in range(0, N, n)]

    cfg = types.SimpleNamespace(global_batch_size_per_device=4, micro_batch_size_per_device_for_update=2, ppo_epochs=1, padding_free=False,
                                ulysses_sequence_parallel_size=1, use_kl_loss=True, disable_kl=False, kl_penalty="low_var_kl", kl_coef=0.05,
                                clip_ratio_low=0.2, clip_ratio_high=0.3, clip_ratio_dual=3.0, max_grad_norm=0.05, use_torch_compile=False)
    params = toy_lm.make_params()
    model = toy_lm.ToyLM(params)
    opt = torch.optim.SGD(model.parameters(), lr=0.2)
    sched = get_constant_schedule_with_warmup(opt, num_warmup_steps=2)
    actor = DataParallelPPOActor(cfg, model, opt)
    ids, mask, P, R = toy_lm.make_data()
    T = 0.8
    rs = np.random.RandomState(9)
    with torch.no_grad():
        lg = toy_lm.logits_fn({k: torch.from_numpy(v) for k, v in params.items()}, torch.from_numpy(ids))[:, -R - 1:-1] / T
        lp0 = torch.nn.functional.cross_entropy(lg.reshape(-1, toy_lm.V), torch.from_numpy(ids[:, -R:]).reshape(-1), reduction="none").view(-1, R).numpy()
    old = (lp0 + 0.25 * rs.standard_normal(lp0.shape)).astype(np.float32)
    ref = (lp0 + 0.3 * rs.standard_normal(lp0.shape)).astype(np.float32)
    adv = (rs.standard_normal((ids.shape[0], 1)).astype(np.float32) * 2.0).repeat(R, 1) * mask[:, -R:]
    out = dict(input_ids=ids, attention_mask=mask, old_log_probs=old, ref_log_probs=ref, advantages=adv, temperature=np.float32(T),
               R=np.int64(R), lr=np.float32(0.2), warmup=np.int64(2))
    out.update({"p0_" + k: v for k, v in params.items()})
    t = torch.from_numpy
    for call in range(4):
        a = adv.copy()
        if call == 3:
            a[5, 0] = np.inf
        pos = np.broadcast_to(np.arange(ids.shape[1]), ids.shape).copy()
        data = Proto(dict(responses=t(ids[:, -R:].copy()), input_ids=t(ids), attention_mask=t(mask), position_ids=t(pos), old_log_probs=t(old),
                          advantages=t(a), ref_log_probs=t(ref)), {"temperature": T})
        met = actor.update_policy(data)
        sched.step()
        for k, v in met.items():
            out[f"c{call}_" + k.replace("/", "_")] = np.asarray(v, dtype=np.float64)
        out[f"c{call}_lr"] = np.float64(sched.get_last_lr()[0])
        for k, v in model.state_dict().items():
            out[f"c{call}_p_" + k] = v.detach().numpy().copy()
    save("update_loop", **out)

if __name__ == "__main__":
    which = sys.argv[1:] or ["rl", "adamw", "pos", "rewards", "graded", "helpers", "math", "small", "micro", "model", "extra", "dataset", "generate", "loop", "values"]
    if "rl" in which:
        gen_rl_math()
    if "adamw" in which:
        gen_adamw()
    if "pos" in which:
        gen_positions()
    if "rewards" in which:
        gen_rewards()
    if "graded" in which:
        gen_rewards_graded()
    if "helpers" in which:
        gen_rewards_helpers()
    if "math" in which:
        gen_rewards_math()
    if "small" in which:
        gen_small_helpers()
    if "micro" in which:
        gen_balance_micro()
    if "model" in which:
        gen_model()
    if "extra" in which:
        gen_rl_extra()
    if "dataset" in which:
        gen_dataset()
    if "generate" in which:
        gen_generate()
    if "loop" in which:
        gen_update_loop()
    if "values" in which:
        gen_values()
