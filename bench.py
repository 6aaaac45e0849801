#!/usr/bin/env python3
"""Headline benchmark: GRPO samples/sec (G = 8 rollouts per prompt) for Qwen2.5-VL-7B on N MI355X.

One "step" = one full GRPO iteration of the reference's fit() loop (verl/trainer/ray_trainer.py:565-706) over a
synthetic STVQA-7K-shaped batch: generate G rollouts per prompt -> dense spatial reward ->
old log-probs (actor, no grad) -> ref log-probs -> group-relative advantages -> update_actor (PPO-clip + low_var_kl,
micro-batch 4, four optimizer steps per rollout batch as rollout_batch_size/global_batch_size = 512/128 in
scripts/config.yaml).  Every rank owns `--prompts-per-gpu` prompts (weak scaling); the only collective is the
gradient all-reduce (RCCL).  Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus 8 ...
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16 = 2.5e15          # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_HBM = 8.0e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="7b", choices=["7b", "3b", "tiny"])
    ap.add_argument("--prompts-per-gpu", type=int, default=64,
                    help="rollout prompts per GPU and step (64 = the reference's 512-prompt rollout batch over 8 GPUs; SURVEY 8d' #3/#4)")
    ap.add_argument("--rollouts", type=int, default=8)
    ap.add_argument("--response-cap", type=int, default=1024, help="max_response_length of the synthetic batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--experience-micro-batch", type=int, default=16, help="rows per no-grad log-prob pass (reference: 16)")
    ap.add_argument("--fuse-micro-batches", type=int, default=None, help="reference micro-batches per forward/backward pass (default: engine default)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher/contract check without a GPU: ranks rendezvous over gloo, time K trivial steps, rank 0 prints the JSON line")
    return ap.parse_args()


def maybe_spawn(a):
    """`python bench.py --gpus N` started as ONE process: launch N ranks as a CHILD torch.distributed.run (nothing has touched the
    GPU yet — a process that has initialised HIP must never exec) and exit with its return code.  Under torchrun (RANK set) this is
    a no-op.  Counterpart of the reference's worker-group launch (verl/single_controller/ray/base.py:75-405)."""
    if "RANK" in os.environ or a.gpus <= 1:
        return
    port = os.environ.get("MASTER_PORT", "29541")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))


def dry_run(a):
    """The bench contract without the engine (CPU, gloo): same barrier / max-over-ranks timing / one JSON line from rank 0."""
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        dist.init_process_group("gloo")
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    for _ in range(a.warmup):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(0.002 * (1 + rank))
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "dry-run (launcher check)", "value": a.steps * world / float(el), "unit": "steps/s", "n_gpus": world,
                          "rccl_ranks": dist.get_world_size() if world > 1 else 1, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": float(el) / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "none", "data": "none", "config": {"workload": "dry-run"}}))
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------ synthetic STVQA-shaped data
def synth_prompts(cfg, n_prompts, rs, P, grid, text_before=200, text_after=564):
    from spatialthinker_amd import indexing as ix
    t, h, w = grid
    n_img_tok = t * h * w // (cfg.v_merge ** 2)
    hi = min(cfg.image_token_id, cfg.vocab_size) - 16
    ids = np.zeros((n_prompts, P), dtype=np.int64)
    mask = np.zeros((n_prompts, P), dtype=np.int64)
    pos = np.zeros((n_prompts, 3, P), dtype=np.int64)
    pix, grids = [], []
    for i in range(n_prompts):
        toks = np.concatenate([rs.randint(0, hi, text_before), [cfg.vision_start_token_id], np.full(n_img_tok, cfg.image_token_id),
                               [cfg.vision_start_token_id + 1], rs.randint(0, hi, text_after)]).astype(np.int64)
        L = len(toks)
        assert L <= P
        ids[i, P - L:] = toks
        mask[i, P - L:] = 1
        g = np.asarray([grid], dtype=np.int64)
        pp = ix.get_rope_index(ids[i], g, mask[i], image_token_id=cfg.image_token_id, vision_start_token_id=cfg.vision_start_token_id,
                               spatial_merge_size=cfg.v_merge)
        pp[:, mask[i] == 0] = 0
        pos[i] = pp
        pix.append(torch.from_numpy(rs.standard_normal((t * h * w, cfg.patch_k)).astype(np.float32)))
        grids.append(g)
    return ids, mask, pos, pix, grids


def synth_reward_strings(n, rs):
    """Templated <observe>/<scene>/<think>/<answer> responses (valid and invalid), 2-6 objects, for the dense reward."""
    gts, preds, problems = [], [], []
    labels = ["dog", "ball", "tree", "car", "person", "bench", "cat", "sign"]
    for _ in range(n):
        k = rs.randint(2, 7)
        objs = []
        for j in range(k):
            x1, y1 = rs.randint(0, 400), rs.randint(0, 300)
            objs.append({"id": f"{labels[rs.randint(len(labels))]}.{j + 1}", "bbox": [int(x1), int(y1), int(x1 + rs.randint(10, 180)), int(y1 + rs.randint(10, 140))]})
        rel = [{"subject": objs[0]["id"], "predicate": "next to", "object": objs[1]["id"]}]
        gts.append(f"<scene>{json.dumps({'objects': objs, 'relationships': rel})}</scene>\n<answer>(B) left</answer>")
        problems.append("Image size: (588 x 448)\nQ. where?")
        mode = rs.randint(0, 4)
        jit = [{"id": o["id"], "bbox": [int(v + rs.randint(-8, 9)) for v in o["bbox"]]} for o in objs[:k - (mode == 1)]]
        scene = json.dumps({"objects": jit, "relationships": rel}) if mode != 2 else "{bad json"
        ans = "(B) left" if mode != 3 else "(A) right"
        preds.append(f"<observe>things</observe>\n<scene>{scene}</scene>\n<think>reasoning</think>\n<answer>{ans}</answer>")
    return preds, gts, problems


# ------------------------------------------------------------------ CPU baseline (oracle, bounded sample)
def cpu_baseline(cfg, S_text, grid, R_mean):
    """Times the fp32 CPU oracle on ONE STVQA-shaped sequence at reduced depth — (1,1), (2,1), (1,2) LM/ViT layers —
    and extrapolates linearly in depth to the full model: cost/sample = 2 no-grad forwards (old, ref) + 1 forward/backward."""
    from oracle import qwen25vl as Q
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                    # honour the container's CPU quota (cgroup v2)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            threads = max(1, min(threads, int(int(quota) / int(period))))
    except Exception:
        pass
    torch.set_num_threads(threads)
    t, h, w = grid
    n_patch = t * h * w
    n_img = n_patch // 4
    S = S_text + n_img + R_mean
    rs = np.random.RandomState(0)

    def run(n_lm, n_vit, grad):
        oc = Q.VLConfig(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_layers=n_lm, num_heads=cfg.num_heads,
                        num_kv_heads=cfg.num_kv_heads, vocab_size=cfg.vocab_size, v_depth=n_vit, v_hidden=cfg.v_hidden, v_heads=cfg.v_heads,
                        v_intermediate=cfg.v_intermediate, v_fullatt=[0], image_token_id=cfg.image_token_id,
                        vision_start_token_id=cfg.vision_start_token_id, tie_word_embeddings=cfg.tie_word_embeddings,
                        mrope_section=cfg.mrope_section)
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import tiny as tiny_shapes  # shape table only
        shapes = tiny_shapes.param_shapes(dict(hidden_size=oc.hidden_size, intermediate_size=oc.intermediate_size, num_layers=n_lm,
                                        num_heads=oc.num_heads, num_kv_heads=oc.num_kv_heads, vocab_size=oc.vocab_size, v_depth=n_vit,
                                        v_hidden=oc.v_hidden, v_intermediate=oc.v_intermediate, v_in_channels=3, v_temporal_patch=2,
                                        v_patch=14, v_merge=2, tie_word_embeddings=oc.tie_word_embeddings))
        def init(shape):                                   # timing does not depend on the values: large tables get a cheap ramp
            n = int(np.prod(shape))
            if n < (1 << 24):
                return torch.randn(shape) * 0.02
            return (torch.arange(n, dtype=torch.float32).remainder_(997.0).mul_(4e-5).sub_(0.02)).view(shape)
        p = {k: init(s).requires_grad_(grad) for k, s in shapes.items()}
        ids = torch.from_numpy(np.concatenate([rs.randint(0, 1000, 200), [oc.vision_start_token_id], np.full(n_img, oc.image_token_id),
                                               rs.randint(0, 1000, S - 201 - n_img)]))
        pos = torch.arange(S)[None, :].repeat(3, 1)
        px = torch.randn(n_patch, 1176)
        rows = torch.arange(S - R_mean - 1, S - 1)
        t0 = time.perf_counter()
        with torch.set_grad_enabled(grad):
            lg = Q.forward_logits(p, oc, ids, pos, [0, S], px, np.asarray([grid]), rows=rows)
            lp = torch.log_softmax(lg, -1)[:, 0].sum()
            if grad:
                lp.backward()
        return time.perf_counter() - t0

    run(1, 1, False)                                        # warm-up: thread pool, allocator, first-touch of the big tables
    f11, f21, f12 = (min(run(*d, False), run(*d, False)) for d in ((1, 1), (2, 1), (1, 2)))
    b11, b21, b12 = run(1, 1, True), run(2, 1, True), run(1, 2, True)
    # per-layer slopes are clamped at zero: the ViT layer (~0.1 s) sits inside the timing noise of the shared part
    full = lambda a11, a21, a12: a11 + max(a21 - a11, 0.0) * (cfg.num_layers - 1) + max(a12 - a11, 0.0) * (cfg.v_depth - 1)
    fwd, fb = full(f11, f21, f12), full(b11, b21, b12)
    per_sample = 2 * fwd + fb
    return {"value": 1.0 / per_sample, "unit": "samples/s (actor path: old+ref forward, update fwd/bwd; generation excluded)",
            "cores": threads, "kind": "port",
            "sample": f"1 sequence of {S} tokens ({n_patch} patches) through the fp32 torch oracle at depths (LM,ViT) = (1,1),(2,1),(1,2), "
                      f"extrapolated linearly to ({cfg.num_layers},{cfg.v_depth}); measured fwd {fwd:.1f}s, fwd+bwd {fb:.1f}s per sample"}


# ------------------------------------------------------------------ main
def main():
    a = parse()
    maybe_spawn(a)
    if a.dry_run:
        return dry_run(a)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == max(1, a.gpus), f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from spatialthinker_amd import ops
    from spatialthinker_amd.actor import ActorHyper, PolicyEngine
    from spatialthinker_amd.model import ParamStore, VLConfig
    from spatialthinker_amd.rollout import Generator
    from verl.utils.reward_score import spatial_sgg_compute_score

    if a.model == "7b":
        cfg, name = VLConfig.qwen2_5_vl_7b(), "Qwen2.5-VL-7B"
    elif a.model == "3b":
        cfg, name = VLConfig.qwen2_5_vl_3b(), "Qwen2.5-VL-3B"
    else:
        cfg, name = VLConfig(hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, vocab_size=1024, v_depth=3,
                             v_hidden=320, v_heads=4, v_intermediate=200, v_window=56, v_fullatt=[1], image_token_id=1010,
                             vision_start_token_id=1011), "tiny"
    tiny = a.model == "tiny"
    grid = (1, 8, 8) if tiny else (1, 32, 42)                       # STVQA-shaped 588x448 -> 1344 patches -> 336 image tokens
    tb, ta = (8, 12) if tiny else (200, 564)
    n_img = grid[1] * grid[2] // 4
    P = (tb + ta + 2 + n_img + 63) // 64 * 64
    R = 64 if tiny else a.response_cap
    G, npr = a.rollouts, a.prompts_per_gpu
    B = G * npr
    micro = 4
    n_opt = 4 if B % (4 * micro) == 0 else 1
    hyper = ActorHyper(micro_batch_size_per_device_for_update=micro, micro_batch_size_per_device_for_experience=a.experience_micro_batch,
                       global_batch_size_per_device=B // n_opt)
    actor_store = ParamStore(cfg, trainable=True)
    actor_store.init_random(seed=7)
    ref_store = ParamStore(cfg, trainable=False)
    ref_store.flat.copy_(actor_store.flat)
    actor = PolicyEngine(cfg, actor_store, hyper)
    ref = PolicyEngine(cfg, ref_store, None)
    if a.fuse_micro_batches is not None:
        actor.fuse_micro_batches = a.fuse_micro_batches
    gen = Generator(actor.model)
    rs = np.random.RandomState(a.seed + rank)
    eos_id, pad_id = (1014, 1013) if tiny else (151645, 151643)
    temperature = 1.0
    flops = {"old": 0.0, "ref": 0.0, "update": 0.0}
    phase = {k: 0.0 for k in ("gen", "reward", "old", "ref", "adv", "update_actor")}
    tokens_total = [0]

    # the synthetic rollout batches (random images are ~0.4 GB of host RNG output per step) are drawn BEFORE the timed region: a
    # training job's dataloader workers prepare the next batch while the GPU runs the current step
    staged = []
    for _ in range(a.warmup + a.steps):
        batch_in = synth_prompts(cfg, npr, rs, P, grid, tb, ta)
        lens_in = np.clip(rs.normal(16 if tiny else 512, 4 if tiny else 128, B), 4 if tiny else 64, R).astype(np.int64)
        staged.append((batch_in, lens_in))

    def one_step(step_idx, timed):
        (ids, mask, pos, pix, grids), lens = staged[step_idx]
        tick = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
        t0 = tick()
        resp, prompt_cache = gen.generate(ids, mask, pos, n=G, max_new_tokens=R, temperature=temperature, eos_token_id=eos_id,
                                          pad_token_id=pad_id, seed=a.seed * 1000 + step_idx, pixel_values=pix, image_grid_thw=grids,
                                          forced_lengths=lens, return_prompt_cache=True)
        t1 = tick()
        # ---- assemble the (B, P+R) batch exactly as vllm_rollout_spmd.py:144-188 does
        resp_c = resp.cpu()
        rmask = (torch.cumsum((resp_c == eos_id).long(), 1) - (resp_c == eos_id).long() == 0).long()
        ids_f = torch.cat([torch.from_numpy(ids).repeat_interleave(G, 0), resp_c], 1)
        mask_f = torch.cat([torch.from_numpy(mask).repeat_interleave(G, 0), rmask], 1)
        pos_p = torch.from_numpy(pos).repeat_interleave(G, 0)
        pos_f = torch.cat([pos_p, pos_p[..., -1:] + torch.arange(1, R + 1)], -1)
        mm = np.repeat(np.array([{"pixel_values": p, "image_grid_thw": g} for p, g in zip(pix, grids)], dtype=object), G)
        data = dict(input_ids=ids_f, attention_mask=mask_f, position_ids=pos_f, responses=resp_c, multi_modal_inputs=mm)
        # ---- reward: dense spatial scorer on templated strings (no tokenizer offline), score at the last valid token
        preds, gts, problems = synth_reward_strings(B, rs)
        scores = torch.tensor([spatial_sgg_compute_score(p, g, q)["overall"] for p, g, q in zip(preds, gts, problems)], dtype=torch.float32)
        rewards = torch.zeros(B, R)
        rewards[torch.arange(B), rmask.sum(1) - 1] = scores
        t2 = tick()
        data["old_log_probs"] = actor.compute_log_prob(data, temperature, prompt_cache=prompt_cache)   # as FSDPWorker.compute_log_probs
        del prompt_cache
        t3 = tick()
        data["ref_log_probs"] = ref.compute_log_prob(data, temperature)
        t4 = tick()
        group = torch.arange(npr, dtype=torch.int32).repeat_interleave(G).cuda()
        adv, status = ops.grpo_advantage(rewards.cuda(), rmask.cuda(), group, npr)
        data["advantages"] = adv
        t5 = tick()
        metrics = actor.update_policy(data, temperature)
        t6 = tick()
        if timed:
            for k, v in zip(phase, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
                phase[k] += v
            seqlens = mask_f.sum(1).tolist()
            # FLOPs two ways: (i) as the reference's per-sequence packing would spend them (every rollout runs its own copy of the
            # prompt and image: the "model FLOPs" of the workload), (ii) as actually executed here (prompt + image once per
            # rollout group inside a micro-batch).  actor_mfu uses (ii): it is a hardware-utilisation number.
            n_patch = grid[0] * grid[1] * grid[2]
            f_ref = cfg.flops_forward(seqlens, [n_patch] * B, logit_rows=int(rmask.sum()))
            plen = mask[:, :].sum(1)                                         # valid prompt tokens per prompt
            rlen = rmask.sum(1).tolist()

            def executed(mb, cached=False):
                groups = []
                for s0 in range(0, B, mb):
                    rows = list(range(s0, min(B, s0 + mb)))
                    for pr in sorted(set(r // G for r in rows)):
                        groups.append((int(plen[pr]), [rlen[r] for r in rows if r // G == pr]))
                return cfg.flops_forward_grouped(groups, [] if cached else [n_patch] * len(groups), logit_rows=int(rmask.sum()),
                                                 prefix_cached=cached)
            f_exp = executed(hyper.micro_batch_size_per_device_for_experience)
            f_upd = executed(micro * max(1, min(actor.fuse_micro_batches, (B // n_opt) // micro)))
            f_old = executed(hyper.micro_batch_size_per_device_for_experience, cached=True)      # prompt K/V re-used from the rollout
            flops["old"] += f_old; flops["ref"] += f_exp; flops["update"] += 3 * f_upd
            flops["reference_formulation"] = flops.get("reference_formulation", 0.0) + 5 * f_ref
            tokens_total[0] += int(mask_f.sum())
        return metrics

    for w in range(a.warmup):
        one_step(w, False)
    ops.prof_enable(ops.K_GEMM, 400000)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_start = time.perf_counter()
    for s in range(a.steps):
        one_step(a.warmup + s, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    n_launch, gemm_ms, gemm_flops = ops.prof_read(ops.K_GEMM)
    ops.prof_disable(ops.K_GEMM)
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        samples = B * world * a.steps
        actor_t = phase["old"] + phase["ref"] + phase["update_actor"]
        out = {
            "metric": "GRPO samples/sec (G=8 rollouts/prompt) Qwen2.5-VL-7B at 1/2/4/8 MI355X",
            "value": samples / elapsed, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (random-init weights at real shapes; STVQA-7K-shaped prompts: 766 text + 336 image tokens from "
                                     "1344 random patches; response lengths ~ clip(N(512,128),64,cap) enforced by forcing EOS; templated reward strings)",
            "config": {"workload": f"{name} dense spatial-reward GRPO step (gen + reward + old/ref log-probs + advantage + update), G={G}, "
                                   f"{npr} prompts/GPU, micro-batch {micro}, {n_opt} optimizer steps/step, max_response_length {R}",
                       "global_batch": B * world, "seq_len": P + R, "parallelism": f"dp{world}"},
            "timing_s": {k: v / a.steps for k, v in phase.items()},
            "perf_throughput_tokens_per_s_per_gpu": tokens_total[0] / elapsed,
            "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
            "peak_reserved_gb": torch.cuda.max_memory_reserved() / 2 ** 30,
            "actor_mfu": (flops["old"] + flops["ref"] + flops["update"]) / actor_t / PEAK_BF16 if actor_t > 0 else None,
            "actor_mfu_reference_flops": flops.get("reference_formulation", 0.0) / actor_t / PEAK_BF16 if actor_t > 0 else None,
            "roofline": {"bound": "mfma", "kernel": "st_gemm_nt family: gemm_tile_kernel<256,256> / gemm_nt_kernel<128,128> (bf16 MFMA 16x16x32, LDS-DMA staged)", "achieved": gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else None,
                         "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                         "frac": gemm_flops / (gemm_ms * 1e-3) / PEAK_BF16 if gemm_ms > 0 else None, "traffic": None,
                         "launches": n_launch, "avg_launch_ms": gemm_ms / max(n_launch, 1)},
        }
        if world == 1 and not a.no_cpu_baseline and not tiny:
            out["cpu_baseline"] = cpu_baseline(cfg, 766, grid, 512)
        elif tiny and not a.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
