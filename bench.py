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

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL across processes)

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16 = 2.5e15          # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_FP8 = 5.0e15           # dense MFMA fp8 (block-scaled K=128 instructions), same table
PEAK_HBM = 8.0e12


class ChipTelemetry:
    """What the chip was doing while the timed region ran (the GEMM class ran 0.518 of peak on the driver's box and
    0.553 on the builder's with identical sources, and nothing in the line could say why).  A host thread wakes every `period` seconds and
      * launches st_clock_probe on a side stream: 8 one-wave workgroups (one per XCD) measure shader cycles per 100-MHz reference tick
        for 20 us WHILE the compute stream's kernels run — the clock the power management actually grants under this load;
      * reads the amdgpu sysfs / hwmon files that exist on the box (socket power, junction / memory temperature, sclk): host side only.
    Every sample carries the phase label the step loop has set (gen / old / ref / update_actor), so the clock under the GEMM-heavy
    phases is separable from the clock under the decode phase.  No sample is ever waited for inside the timed region."""
    HWMON = ("power1_average", "power1_input", "temp1_input", "temp2_input", "temp3_input", "freq1_input", "freq2_input")

    def __init__(self, period: float = 0.25, max_samples: int = 8192):
        import glob
        import threading
        from spatialthinker_amd import ops
        self.ops, self.period, self.max = ops, period, max_samples
        self.buf = torch.zeros(max_samples, 8, 2, dtype=torch.int64, device="cuda")
        self.stream = torch.cuda.Stream()
        self.labels, self.host = [], []
        self.phase = "idle"
        self.errors = 0
        self.files = {}
        # the hwmon directory of THIS process's GPU (a node exposes every GPU's sysfs files to the container; c2 of round 6 read an idle
        # neighbour): matched by PCI address; no match -> no sysfs figures rather than somebody else's
        self.pci = None
        try:
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            self.pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:
            pass
        for d in (sorted(glob.glob(f"/sys/bus/pci/devices/{self.pci}/hwmon/hwmon*")) if self.pci else []):
            for f in self.HWMON:
                fp = os.path.join(d, f)
                if os.path.exists(fp) and f not in self.files:
                    self.files[f] = fp
            if self.files:
                break
        self.power_cap_w = None                                    # the board's power limit (hwmon power1_cap, microwatts): the GEMM phases sit at it
        if self.files:
            cap = self._read(os.path.join(os.path.dirname(next(iter(self.files.values()))), "power1_cap"))
            self.power_cap_w = cap * 1e-6 if cap else None
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, name="telemetry", daemon=True)

    def start(self):
        self.ops.clock_probe(self.buf, self.max - 1, self.stream)          # first launch (module load) from the main thread, outside any capture
        self.stream.synchronize()
        self._thread.start()

    def _read(self, fp):
        try:
            with open(fp) as f:
                return float(f.read().split()[0])
        except Exception:
            return None

    def _run(self):
        while not self._stop.wait(self.period):
            k = len(self.labels)
            if k >= self.max:
                break
            try:
                self.ops.clock_probe(self.buf, k, self.stream)
            except Exception:
                self.errors += 1
                continue
            self.labels.append(self.phase)
            self.host.append({f: self._read(fp) for f, fp in self.files.items()})

    def stop(self):
        self._stop.set()
        self._thread.join(timeout=5.0)

    def summary(self, gemm_phases=("old", "ref", "update_actor")):
        self.stream.synchronize()
        n = len(self.labels)
        if n == 0:
            return None
        raw = self.buf[:n].cpu().numpy().astype(np.float64)
        mhz = 100.0 * raw[:, :, 0] / np.maximum(raw[:, :, 1], 1.0)            # (samples, XCDs)
        ok = raw[:, :, 1].min(axis=1) > 0
        lab = np.array(self.labels)

        def stats(sel):
            sel = sel & ok
            if not sel.any():
                return None
            m = mhz[sel]
            out = {"samples": int(sel.sum()), "clock_mhz_mean": float(m.mean()), "clock_mhz_min": float(m.min(axis=1).min()),
                   "clock_mhz_p10": float(np.percentile(m.mean(axis=1), 10)), "clock_mhz_max": float(m.max())}
            for f in self.files:
                v = np.array([h[f] for h, s_ in zip(self.host, sel) if s_ and h.get(f) is not None], dtype=np.float64)
                if v.size:
                    scale = 1e-6 if f.startswith(("power", "freq")) else 1e-3      # uW -> W, Hz -> MHz, millidegrees -> degrees C
                    key = {"power1_average": "socket_power_w", "power1_input": "socket_power_w", "freq1_input": "sclk_mhz_sysfs",
                           "freq2_input": "mclk_mhz_sysfs"}.get(f, f.replace("_input", "_c"))
                    out[key + "_mean"] = float(v.mean() * scale)
                    out[key + "_max"] = float(v.max() * scale)
            return out
        per_phase = {p: stats(lab == p) for p in sorted(set(self.labels))}
        return {"period_s": self.period, "probe": "st_clock_probe: s_memtime / s_memrealtime over 20 us on 8 one-wave workgroups (one per XCD), side stream",
                "sysfs_files": sorted(self.files), "pci": self.pci, "power_cap_w": getattr(self, "power_cap_w", None), "probe_errors": self.errors, "all": stats(np.ones(n, dtype=bool)),
                "gemm_phases": stats(np.isin(lab, gemm_phases)), "by_phase": per_phase}


def parity_record():
    """Both parity criteria side by side: north_star's literal 1e-3 against what a bf16 evaluation of this model family
    gives at all.  The figures are the committed measurement of tools/parity_probe.py on an MI355X (tiny Qwen2.5-VL config, 16 random batches,
    288 response tokens; tests/test_gpu_model.py re-measures them live and asserts engine <= HF-bf16 on every statistic) — not re-measured here."""
    try:
        y = json.load(open(os.path.join(ROOT, "tests", "golden", "bf16_yardstick.json")))
        keys = ("tokens", "rms", "mean_abs", "p99", "max", "mean_of_per_batch_max")
        return {"quantity": "|log-prob - HF-fp32 log-prob| over the response tokens, tiny Qwen2.5-VL config, 16 batches pooled",
                "engine_vs_fp32": {k: y["logp"]["engine"][k] for k in keys}, "hf_bf16_vs_fp32": {k: y["logp"]["hf_bf16"][k] for k in keys},
                "hidden_state_taps_engine_over_hf_bf16_rel_l2": {t["tap"]: t["engine_rel_l2"] / t["hf_bf16_rel_l2"] for t in y["taps"]},
                "north_star_literal_tolerance": 1e-3, "criterion_in_force": "SURVEY 8c (iii) at factor 1.0: engine error <= HF-bf16's own error, every pooled statistic and tap",
                "full_depth_7b": "tests/test_gpu_depth.py: engine max 0.105 / rms 0.045 vs plain torch-bf16 0.220 / 0.078 against the fp32 CPU oracle",
                "source": "tests/golden/bf16_yardstick.json (tools/parity_probe.py 16 on MI355X)"}
    except Exception as e:                                     # a missing fixture must not cost the measurement
        return {"error": f"{type(e).__name__}: {e}"}


def gemm_source_sha() -> str:
    """sha256 (16 hex digits) of the GEMM kernel sources: ties a committed PMC traffic figure to the kernels it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm_asm4.hip", "gemm_tile_kernel.h", "gemm.hip", "gemm_tiles_train.hip", "gemm_tiles_swiglu.hip", "gemm_tiles_layout.hip"):
        with open(os.path.join(ROOT, "spatialthinker_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="7b", choices=["7b", "3b", "tiny"])
    ap.add_argument("--prompts-per-gpu", type=int, default=64,
                    help="rollout prompts per GPU and step (64 = the reference's 512-prompt rollout batch over 8 GPUs; SURVEY 8d' #3/#4)")
    ap.add_argument("--rollouts", type=int, default=8)
    ap.add_argument("--image", default=None, help="WxH pixels of the synthetic image instead of the STVQA-shaped 588x448 (SURVEY 8d': 224x224 / "
                                                  "448x448 / 896x896 -> 256 / 1024 / 4096 patches, with 700 text tokens)")
    ap.add_argument("--response-cap", type=int, default=2048,
                    help="max_response_length of the synthetic batch (scripts/spatialthinker_7b_grpo.sh:34: 2048)")
    ap.add_argument("--prompt-tokens", type=int, default=None,
                    help="valid tokens per prompt (text + image tokens + 2 vision markers) instead of the STVQA-shaped ~1102")
    ap.add_argument("--responses-at-cap", action="store_true", help="every response runs to --response-cap (default: lengths ~ N(512, 128))")
    ap.add_argument("--worst-case", action="store_true",
                    help="the shipped scripts' worst-case shape (scripts/spatialthinker_7b_grpo.sh:23,33-34 + config.yaml:11,27-29): 128 prompts/GPU, "
                         "6144-token prompts, every response at the 2048 cap, ONE untimed-warm-up-free step, then an update_policy over a "
                         "micro-batch of 4 UNRELATED 8192-token rows; reports peak allocated / reserved memory, rollout chunks and passes per step, "
                         "and exits non-zero with a message (never an OOM traceback) if it does not fit")
    ap.add_argument("--old-from-rollout", action="store_true",
                    help="OPT-IN variant, not the headline configuration: the rollout records log pi_old of the tokens it samples (same logits the "
                         "sampler reads) and the old-policy pass returns them instead of a second forward (worker.rollout.old_log_probs_from_rollout)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--experience-micro-batch", type=int, default=16, help="rows per no-grad log-prob pass (reference: 16)")
    ap.add_argument("--master-fp32", action="store_true", help="the reference's default actor dtype pair: fp32 master weights + fp32 AdamW moments "
                    "(worker.actor.fsdp.torch_dtype unset, optim.strategy=adamw) instead of the shipped scripts' bf16 + AnyPrecisionAdamW")
    ap.add_argument("--fuse-micro-batches", type=int, default=None, help="reference micro-batches per forward/backward pass (default: engine default)")
    ap.add_argument("--no-recompute-light", action="store_true", help="keep the RMSNorm / SwiGLU outputs for the backward instead of recomputing them (+50 %% activation memory per packed token)")
    ap.add_argument("--tokens-grad", type=int, default=None, help="packed-token budget of an update pass (default: engine default, ST_TOKENS_GRAD)")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8 = BASELINE config #5's arithmetic: the LM projection GEMMs of every forward pass on the MX-fp8 (OCP e4m3, block-"
                         "scaled) MFMA path, everything else (attention, lm_head, ViT, backward, decode) bf16")
    ap.add_argument("--fp8-dgrad", action="store_true", help="with --dtype fp8: the input-gradient GEMMs dX = dY W of the LM layers in MX-fp8 too "
                    "(transposed fp8 weight copies, dY quantised on the fly); weight gradients stay bf16")
    ap.add_argument("--fp8-wgrad", action="store_true", help="with --dtype fp8: the weight-gradient GEMMs dW = dY^T X of the LM layers in MX-fp8 too "
                    "(both operands quantised token-minor on the fly)")
    ap.add_argument("--through-api", action="store_true",
                    help="the same synthetic step through the kept API: `python -m verl.trainer.main` (config merge, RLHF dataloader, FSDPWorker "
                         "methods, DataProto .cpu() round trips, tokenizer decode + CustomRewardManager, RayPPOTrainer.fit) instead of calling the "
                         "engine directly; the JSON line reports the step rate the trainer itself logged")
    ap.add_argument("--ranks-share-gpu", action="store_true",
                    help="TEST MODE for a 1-GPU box: the N ranks of --gpus N all use cuda:0 and exchange gradients over gloo (RCCL refuses two ranks on "
                         "one device).  Runs the real multi-rank step — rank-sharded batches, overlapped gradient exchange on device tensors, "
                         "barrier / max-over-ranks timing, the exchange statistics of the JSON line — with a small model; not a scaling measurement")
    ap.add_argument("--overlap-old", default="off", choices=["on", "off"],
                    help="on: the old-policy log-prob pass of the samples that have finished runs on a CU-range stream WHILE the decode tail of the "
                         "rollout (phases of <= --tail-rows rows) runs on the complementary compute units (actor.EarlyLogProb) — bit-identical "
                         "results, measured SLOWER on MI355X (profiles/r05_notes.md: the two streams share every XCD's L2 and the fabric; gen + old "
                         "12.35-12.40 s vs 11.99 s serial), hence off (default): the reference's serial order, rollout then compute_log_prob")
    ap.add_argument("--tail-cus", type=int, default=64, help="compute units of the decode tail's stream under --overlap-old on (multiple of 8)")
    ap.add_argument("--tail-rows", type=int, default=128, help="decode phases of at most this many rows run on the tail stream")
    ap.add_argument("--no-telemetry", action="store_true", help="do not sample the shader clock / socket power beside the timed region")
    ap.add_argument("--no-fp8-leg", action="store_true",
                    help="skip the short BASELINE-config-#5 leg (G = 16, 896x896, 32 prompts/GPU, MX-fp8 projections forward + dX + dW; 2 steps + 1 "
                         "warm-up in a child process before this process touches the GPU) that the default command reports as `cfg5_fp8`")
    ap.add_argument("--gemm-table", default=None, metavar="PATH",
                    help="write the per-(form, M, N, K) table of the event-timed GEMM launches of the timed region as JSON (the five costliest "
                         "shapes are in the line's roofline.by_shape either way)")
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


def gemm_shape_table(ev_ms, ev_units, ev_tags):
    """Event-timed GEMM launches of the timed region grouped by (form, epilogue flags, M, N, K): launches, total ms, TF/s.  Rows sorted by
    the time a shape would save at the class's best rate (where the in-situ loss sits)."""
    from spatialthinker_amd import ops
    rows = {}
    for ms, u, t in zip(ev_ms.tolist(), ev_units.tolist(), ev_tags.tolist()):
        r = rows.setdefault(int(t), [0, 0.0, 0.0])
        r[0] += 1; r[1] += ms; r[2] += u
    out = []
    for t, (n, ms, u) in rows.items():
        d = ops.gemm_tag_decode(t)
        d.update(launches=n, ms=round(ms, 3), tflops=round(u / (ms * 1e-3) / 1e12, 1) if ms > 0 else None)
        out.append(d)
    out.sort(key=lambda d: -d["ms"])
    return out


def fp8_leg(a):
    """BASELINE config #5's per-GPU shape in its own arithmetic (fp8 MFMA), as a short leg of the DEFAULT command so that the driver's bench
    witnesses it: a CHILD process — started before this process initialises the GPU, so it finds the whole HBM and is
    never an exec from a GPU process — runs `bench.py --dtype fp8 --fp8-dgrad --fp8-wgrad --rollouts 16 --prompts-per-gpu 32 --image
    896x896 --steps 1 --warmup 1`; its line is condensed into {value, dtype, roofline {frac, peak}, timing_s}.  A throughput mode, not
    a parity mode; the headline `value` stays the bf16 config #3 step.  Returns None when the leg does not apply."""
    default_workload = (a.model == "7b" and a.dtype == "bf16" and a.gpus <= 1 and not a.image and a.rollouts == 8 and a.prompts_per_gpu == 64
                        and not a.worst_case and not a.responses_at_cap and not a.prompt_tokens and not a.old_from_rollout and not a.master_fp32)
    if a.no_fp8_leg or not default_workload or "RANK" in os.environ:
        return None
    # under rocprofv3 the profiler's preloaded library has already initialised the GPU in THIS process: starting another program from it
    # is the exec the GPU boxes refuse — the leg is skipped there (the profiling tools pass --no-fp8-leg anyway)
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ):
        return None
    cmd = [sys.executable, os.path.abspath(__file__), "--dtype", "fp8", "--fp8-dgrad", "--fp8-wgrad", "--rollouts", "16", "--prompts-per-gpu", "32",
           "--image", "896x896", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-fp8-leg", "--no-telemetry"]      # (round 6: one timed step —
    # the leg cost 78-85 s of the driver's 860-870-s run with two)
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=420)      # ~2x the leg's usual 180-200 s
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        if p.returncode != 0 or not line:
            return {"error": f"child exited with {p.returncode}", "stderr_tail": p.stderr[-400:]}
        d = json.loads(line[-1])
    except Exception as e:                                    # the leg must never cost the headline measurement
        return {"error": f"{type(e).__name__}: {e}"}
    cls = [c for c in d.get("roofline_classes", []) if c and c.get("peak") == 5000.0]
    r8 = cls[0] if cls else (d.get("roofline") if (d.get("roofline") or {}).get("peak") == 5000.0 else None)
    return {"value": d.get("value"), "unit": d.get("unit"), "dtype": d.get("dtype"), "steps": d.get("steps"), "warmup": d.get("warmup"),
            "ms_per_step": d.get("ms_per_step"), "config": d.get("config"), "timing_s": d.get("timing_s"),
            "roofline": None if r8 is None else {k: r8.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches_timed", "avg_launch_ms")},
            "roofline_bf16_gemm_frac": (d.get("roofline") or {}).get("frac"), "peak_mem_gb": d.get("peak_mem_gb"),
            "command": " ".join(cmd[1:]), "wall_s": time.perf_counter() - t0,
            "note": "throughput mode (MX-fp8 LM projections forward + input / weight gradients; decode, attention, lm_head, ViT stay bf16), not a parity mode"}


def through_api(a):
    """Run K + W steps of `python -m verl.trainer.main` on the bench workload as a CHILD process (this one never touches the GPU) and
    report samples/s from the trainer's own timers (timing_s/step = gen + reward + balance + old + ref + adv + update_actor, the
    reference's `step` timer of ray_trainer.py:585-706)."""
    import re
    size = {"7b": "random:7b", "3b": "random:3b", "tiny": "random:tiny"}[a.model]
    tiny = a.model == "tiny"
    G, npr = a.rollouts, a.prompts_per_gpu
    R = 64 if tiny else a.response_cap
    spec = "synthetic:stvqa" + (f":{a.image}" if a.image else "") + (":len=16,4" if tiny else ":len=512,128") + "@train"
    n_opt = 4 if (G * npr) % 16 == 0 else 1
    cmd = [sys.executable, "-m", "verl.trainer.main", f"data.train_files={spec}", "data.val_files=", f"data.rollout_batch_size={npr * max(1, a.gpus)}",
           f"data.max_prompt_length={128 if tiny else 1152}", f"data.max_response_length={R}", f"worker.actor.model.model_path={size}",
           f"worker.actor.global_batch_size={npr * max(1, a.gpus) // n_opt}", "worker.actor.micro_batch_size_per_device_for_update=4",
           f"worker.actor.micro_batch_size_per_device_for_experience={a.experience_micro_batch}", "worker.actor.optim.strategy=adamw_bf16",
           "worker.actor.fsdp.torch_dtype=bf16", "worker.actor.padding_free=true", f"worker.rollout.n={G}", "worker.reward.score_function=spatial_sgg",
           "algorithm.use_kl_loss=true", "algorithm.kl_penalty=low_var_kl", "algorithm.kl_coef=1.0e-2", f"trainer.max_steps={a.steps + a.warmup}",
           "trainer.total_episodes=100", f"trainer.n_gpus_per_node={max(1, a.gpus)}", "trainer.val_before_train=false", "trainer.logger=['console']",
           "trainer.save_freq=-1", "trainer.save_checkpoint_path=/tmp/st_bench_api_ckpt"]
    env = dict(os.environ, PYTHONPATH=ROOT, ST_SKIP_FINAL_SAVE="1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stdout[-3000:] + p.stderr[-3000:])
        raise SystemExit(p.returncode)
    steps = []
    for line in p.stdout.splitlines():
        if line.startswith("step ") and "timing_s/step" in line:
            kv = dict(item.split(":", 1) for item in line.split(": ", 1)[1].split(" - ") if ":" in item)
            steps.append({k: float(v) for k, v in kv.items() if re.fullmatch(r"[-+0-9.eE]+|nan|inf", v)})
    timed = steps[a.warmup:]
    assert len(timed) == a.steps, (len(steps), a.steps, p.stdout[-2000:])
    B, world = G * npr, max(1, a.gpus)
    el = sum(s_["timing_s/step"] for s_ in timed)
    keys = ("gen", "reward", "old", "ref", "adv", "update_actor")
    print(json.dumps({
        "metric": "GRPO samples/sec (G=8 rollouts/prompt) Qwen2.5-VL-7B at 1/2/4/8 MI355X", "value": B * world * a.steps / el, "unit": "samples/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic (SyntheticSTVQADataset rows through the resumable dataloader; random-init weights; "
        "response lengths ~ clip(N(512,128),64,cap) forced by the worker; rewards scored on the DECODED random tokens)",
        "config": {"workload": f"python -m verl.trainer.main (kept API) on the bench workload: {size}, G={G}, {npr} prompts/GPU, micro-batch 4, {n_opt} optimizer "
                               f"steps/step, max_response_length {R}", "global_batch": B * world, "parallelism": f"dp{world}", "through_api": True},
        "timing_s": {k: sum(s_.get(f"timing_s/{k}", 0.0) for s_ in timed) / a.steps for k in keys},
        "timing_s_step_minus_phases": (el - sum(s_.get(f"timing_s/{k}", 0.0) for s_ in timed for k in keys)) / a.steps,
        "prompt_cache_hit": sum(s_.get("perf/prompt_cache_hit", 0.0) for s_ in timed) / a.steps,
        "perf_mfu_actor_reference_counter": sum(s_.get("perf/mfu_actor", 0.0) for s_ in timed) / a.steps,
        "trainer_samples_per_s": [s_.get("perf/samples_per_s") for s_ in timed]}))


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


def preflight(rank: int, world: int, local: int, need_gb: float, timeout_s: float = 120.0):
    """Before the first RCCL collective of an N-GPU run: everything that would otherwise show up as a HANG is checked and turned into a
    message + non-zero exit — enough visible devices, enough free HBM on this rank for the replica design (weights + fp32 grads + AdamW state + frozen reference),
    and one tiny all-gather under a watchdog (a rank that never arrives, a wedged transport); missing peer access between this rank's GPU
    and another one (xGMI / PCIe P2P: RCCL's direct transports) is a warning.  Rank 0 prints one summary line
    (RCCL version, free GB per rank).  The watchdog ends the PROCESS (os._exit) — it never re-execs anything: a process that has touched
    the GPU must not exec (the launcher's torchrun parent then tears the other ranks down)."""
    import threading
    problems = []
    n_dev = torch.cuda.device_count()
    if n_dev < world:
        problems.append(f"{world} ranks but only {n_dev} visible GPUs")
    else:
        # reported, not fatal: without peer access RCCL still has its shared-memory transport (slower, and a scaling number measured that
        # way says so in stderr); a transport that does wedge is caught by the watchdog around the first collective below
        no_peer = [j for j in range(world) if j != local and not torch.cuda.can_device_access_peer(local, j)]
        if no_peer:
            sys.stderr.write(f"[bench preflight] rank {rank}: WARNING GPU {local} reports no peer access to GPUs {no_peer} "
                             f"(no xGMI / PCIe P2P visible here: RCCL will stage through host memory)\n")
    free_b, total_b = torch.cuda.mem_get_info()
    if free_b / 2 ** 30 < need_gb:
        problems.append(f"only {free_b / 2 ** 30:.0f} GB free of {total_b / 2 ** 30:.0f} GB on GPU {local}, the workload needs ~{need_gb:.0f} GB")
    done = threading.Event()

    def watchdog():
        if not done.wait(timeout_s):
            sys.stderr.write(f"[bench preflight] rank {rank}: the first all-reduce did not complete within {timeout_s:.0f} s "
                             f"(a rank missing or an RCCL transport wedged) — aborting\n")
            sys.stderr.flush()
            os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    flag = torch.tensor([float(len(problems) > 0), free_b / 2 ** 30], device="cuda", dtype=torch.float64)
    gathered = [torch.zeros_like(flag) for _ in range(world)]
    dist.all_gather(gathered, flag)
    torch.cuda.synchronize()
    done.set()
    bad = [i for i, g in enumerate(gathered) if float(g[0]) > 0]
    if problems:
        sys.stderr.write(f"[bench preflight] rank {rank}: " + "; ".join(problems) + "\n")
    if rank == 0:
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = "unknown"
        sys.stderr.write(f"[bench preflight] RCCL {ver}, {world} ranks, free GB per rank: {[round(float(g[1]), 1) for g in gathered]}"
                         + (f", FAILED on ranks {bad}" if bad else ", ok") + "\n")
    if bad:
        dist.destroy_process_group()
        raise SystemExit(4)


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
def cpu_baseline():
    """SURVEY 8(d) CPU baseline through the build's CPU restatement of the actor path (oracle/, plain PyTorch fp32, all host cores),
    a BOUNDED SAMPLE of ~25 s in two legs, each timing WHOLE passes over a 4-layer model (not one layer multiplied out):
      leg 1 = BASELINE config #1 (Qwen2.5-VL-3B widths, 2 prompts x G=4, 224x224 image = 256 patches = 64 image tokens, 700 text
              tokens, 512-token responses): a 4-LM-layer / 4-ViT-block model on 1 of the 8 sequences — no-grad pass (old / ref),
              forward + backward (update), final norm + tied lm_head + log-softmax on the response rows, 6 KV-cache decode steps of
              the 4 layers for the 8 rollouts, AnyPrecisionAdamW's torch ops on 8M bf16 parameters; every component is the MEDIAN of 3
              runs after a warm-up; scaled by layers (36 / 4, 32 / 4), sequences (8 / 1),
              decode steps (512 / 16) and parameters to the full step;
      leg 2 = truncated BASELINE config #3 (7B widths): a no-grad pass of 4 LM layers over one 1614-token STVQA-shaped sequence, so
              the CPU number exists at the GPU line's own widths (reported as forward tokens/s per layer-normalised pass)."""
    import resource
    from oracle import positions as OP
    from oracle import qwen25vl as Q
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                    # honour the container's CPU quota (cgroup v2)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            threads = max(1, min(threads, int(int(quota) / int(period))))
    except Exception:
        pass
    # Two cores of the quota stay free for this process's other threads (the Python main thread's siblings, the GPU runtime's helpers): with
    # quota-many compute threads spinning at their barriers the cgroup gets THROTTLED (CFS bandwidth: the whole process sleeps until the next
    # 100-ms period) and the legs made of many small ops (decode step, AdamW) came out 4-10x slower on some boxes (round 3: the driver's
    # run; round 4: the 20-step run) while the large matmuls moved by 6 %.  The throttle counters are recorded per leg; a leg that was
    # throttled is measured again (up to 3 attempts, the least-throttled one counts).
    threads = max(1, threads - 2) if threads > 4 else threads
    torch.set_num_threads(threads)
    # The legs made of many small ops allocate their temporaries through glibc malloc.  At the end of a LONG bench run (the driver's 25 steps)
    # they came out 4-10x slower than after a 3-step run on the same kind of box, with no CPU throttling: every temporary above the mmap
    # threshold (the decode step's repeat_kv copies: 2 x 67 MB per layer) is mapped, page-faulted in and unmapped again — 65 000 minor faults
    # per layer and step — and whether the kernel still has transparent huge pages to hand out decides how much that costs (0.07 vs 0.26 s
    # per decode step).  Serve every size from the heap and keep freed memory mapped for the duration of the baseline: 0 faults per step
    # after the warm-up, the same time in both regimes — what a long-lived CPU trainer process with a caching allocator sees.
    malloc_tuned = False
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        M_TRIM_THRESHOLD, M_TOP_PAD, M_MMAP_THRESHOLD, M_MMAP_MAX = -1, -2, -3, -4
        malloc_tuned = bool(libc.mallopt(M_MMAP_THRESHOLD, 32 << 20) and libc.mallopt(M_TRIM_THRESHOLD, (1 << 31) - 1) and libc.mallopt(M_TOP_PAD, 256 << 20)
                            and libc.mallopt(M_MMAP_MAX, 0))
    except Exception:
        pass
    gen = torch.Generator().manual_seed(0)
    rnd = lambda *sh: torch.randn(*sh, generator=gen) * 0.02

    def throttle_counters():
        try:
            kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())
            return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
        except Exception:
            return 0, 0
    host = {"cgroup_cpu_max": (open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None),
            "cpu_count": os.cpu_count(), "loadavg_at_start": (open("/proc/loadavg").read().split()[:3] if os.path.exists("/proc/loadavg") else None),
            "throttled_usec_per_leg": {}, "attempts_per_leg": {}, "minor_faults_per_run_of_leg": {}, "malloc_keeps_freed_memory_mapped": malloc_tuned}
    thr0 = throttle_counters()

    def lm_params(c, L, with_head):
        H, I, D = c.hidden_size, c.intermediate_size, c.head_dim
        kv = c.num_kv_heads * D
        p = {}
        for i in range(L):
            b = f"model.language_model.layers.{i}."
            p.update({b + "input_layernorm.weight": torch.ones(H), b + "post_attention_layernorm.weight": torch.ones(H),
                      b + "self_attn.q_proj.weight": rnd(H, H), b + "self_attn.q_proj.bias": torch.zeros(H),
                      b + "self_attn.k_proj.weight": rnd(kv, H), b + "self_attn.k_proj.bias": torch.zeros(kv),
                      b + "self_attn.v_proj.weight": rnd(kv, H), b + "self_attn.v_proj.bias": torch.zeros(kv),
                      b + "self_attn.o_proj.weight": rnd(H, H), b + "mlp.gate_proj.weight": rnd(I, H), b + "mlp.up_proj.weight": rnd(I, H),
                      b + "mlp.down_proj.weight": rnd(H, I)})
        if with_head:
            n = c.vocab_size * H                            # a cheap ramp for the big table (timing does not depend on the values)
            p["model.language_model.embed_tokens.weight"] = torch.arange(n, dtype=torch.float32).remainder_(997.0).mul_(4e-5).sub_(0.02).view(c.vocab_size, H)
            p["model.language_model.norm.weight"] = torch.ones(H)
        return p

    REPS = 3
    spread = {}

    def timed(fn, name, reps=REPS, warmup=1):
        """median of `reps` runs after `warmup` untimed ones (round 3's single-rep timings moved 2x between boxes: first-touch page
        faults and the thread pool's start-up landed inside the one timed run); the min / max of the reps go into `spread`"""
        for _ in range(warmup):
            fn()
        best = None
        for attempt in range(3):
            before = throttle_counters()
            flt0 = resource.getrusage(resource.RUSAGE_SELF).ru_minflt
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            throttled = throttle_counters()[1] - before[1]
            host["minor_faults_per_run_of_leg"][name] = (resource.getrusage(resource.RUSAGE_SELF).ru_minflt - flt0) // reps
            ts.sort()
            if best is None or throttled < best[0]:
                best = (throttled, ts)
            if throttled == 0:
                break
        host["throttled_usec_per_leg"][name], host["attempts_per_leg"][name] = best[0], attempt + 1
        ts = best[1]
        spread[name] = [ts[0], ts[-1]]
        return ts[len(ts) // 2]

    # ---------------------------------------------------------------- leg 1: config #1 (3B widths)
    L_LM, L_VIT, G, n_prompt, P_txt, n_img_tok, R = 36, 32, 4, 2, 700, 64, 512
    LS, NSEQ = 4, 1                                          # layers and sequences of the timed sample
    c = Q.VLConfig(hidden_size=2048, intermediate_size=11008, num_layers=LS, num_heads=16, num_kv_heads=2, vocab_size=151936, v_depth=LS,
                   tie_word_embeddings=True)
    H, D, V = c.hidden_size, c.head_dim, c.vocab_size
    p = lm_params(c, LS, with_head=True)
    vh, vi = c.v_hidden, c.v_intermediate
    for i in range(LS):
        b = f"model.visual.blocks.{i}."
        p.update({b + "norm1.weight": torch.ones(vh), b + "norm2.weight": torch.ones(vh), b + "attn.qkv.weight": rnd(3 * vh, vh),
                  b + "attn.qkv.bias": torch.zeros(3 * vh), b + "attn.proj.weight": rnd(vh, vh), b + "attn.proj.bias": torch.zeros(vh),
                  b + "mlp.gate_proj.weight": rnd(vi, vh), b + "mlp.gate_proj.bias": torch.zeros(vi), b + "mlp.up_proj.weight": rnd(vi, vh),
                  b + "mlp.up_proj.bias": torch.zeros(vi), b + "mlp.down_proj.weight": rnd(vh, vi), b + "mlp.down_proj.bias": torch.zeros(vh)})
    S = P_txt + n_img_tok + R
    T = NSEQ * S
    pos = torch.arange(S).repeat(NSEQ)[None, :].repeat(3, 1)
    cos, sin = Q.mrope_cos_sin(pos, D, c.rope_theta, c.mrope_section)
    cu = [i * S for i in range(NSEQ + 1)]
    x = torch.randn(T, H, generator=gen) * 0.1

    def lm_pass(xx):
        for i in range(LS):
            xx = Q.lm_layer(p, c, i, xx, cos, sin, cu)
        return xx
    lm_names = [k for k in p if k.startswith("model.language_model.layers.")]

    def grad_on(names, on):
        for k in names:
            p[k].requires_grad_(on); p[k].grad = None

    with torch.no_grad():
        Q.lm_layer(p, c, 0, x[:S], cos[:S], sin[:S], [0, S])                          # warm-up: thread pool, allocator
        t_lm_f = timed(lambda: lm_pass(x), "lm_4_layers_fwd")
    grad_on(lm_names, True)
    xg = x.clone().requires_grad_(True)
    t_lm_fb = timed(lambda: lm_pass(xg).sum().backward(), "lm_4_layers_fwd_bwd")
    grad_on(lm_names, False)
    # ViT: the reference runs the tower per sequence (one 16x16-patch image each)
    grid = np.asarray([[1, 16, 16]] * NSEQ)
    _, cu_win = OP.vision_window_index(grid, merge_size=2, window_size=112, patch_size=14)
    N = 256 * NSEQ
    xv = torch.randn(N, vh, generator=gen) * 0.1
    ang = torch.randn(N, 1, c.v_head_dim, generator=gen)

    def vit_pass(xx):
        for i in range(LS):
            xx = Q.vit_block(p, c, i, xx, ang.cos(), ang.sin(), cu_win)
        return xx
    vit_names = [k for k in p if k.startswith("model.visual.blocks.")]
    with torch.no_grad():
        t_vit_f = timed(lambda: vit_pass(xv), "vit_4_blocks_fwd")
    grad_on(vit_names, True)
    xvg = xv.clone().requires_grad_(True)
    t_vit_fb = timed(lambda: vit_pass(xvg).sum().backward(), "vit_4_blocks_fwd_bwd")
    grad_on(vit_names, False)
    # final norm + tied lm_head + log-softmax on the response rows (dp_actor.py:126-153)
    rows = NSEQ * R
    xr = torch.randn(rows, H, generator=gen) * 0.1
    lab = torch.randint(0, V, (rows,), generator=gen)
    head_fn = lambda xx: torch.log_softmax(Q.lm_head(p, c, xx), -1).gather(-1, lab[:, None]).sum()
    with torch.no_grad():
        t_head_f = timed(lambda: head_fn(xr), "head_fwd")
    emb = p["model.language_model.embed_tokens.weight"]
    emb.requires_grad_(True)
    xrg = xr.clone().requires_grad_(True)
    t_head_fb = timed(lambda: head_fn(xrg).backward(), "head_fwd_bwd")
    emb.requires_grad_(False); emb.grad = None
    # generation: KV-cache decode of the 4 layers, all 8 rollouts of the step in one batch, 16 tokens at a ~1000-token context
    Bd, ctx, n_dec = n_prompt * G, P_txt + n_img_tok + R // 2, 6
    kc = [torch.randn(Bd, ctx + n_dec, c.num_kv_heads, D, generator=gen) for _ in range(LS)]
    vc = [torch.randn(Bd, ctx + n_dec, c.num_kv_heads, D, generator=gen) for _ in range(LS)]
    xd = torch.randn(Bd, H, generator=gen) * 0.1
    cd, sd = Q.mrope_cos_sin(torch.full((3, Bd), ctx), D, c.rope_theta, c.mrope_section)

    def dec():
        ln = torch.full((Bd,), ctx)
        for _ in range(n_dec):
            h = xd
            for i in range(LS):
                h = Q.lm_layer_decode(p, c, i, h, cd, sd, kc[i], vc[i], ln)
            ln = ln + 1
    with torch.no_grad():
        t_dec = timed(dec, "decode_step_4_layers") / n_dec   # per decode step of 4 layers
        spread["decode_step_4_layers"] = [v / n_dec for v in spread["decode_step_4_layers"]]
        t_dec_head = timed(lambda: Q.lm_head(p, c, xd).argmax(-1), "decode_head_step")
    # AdamW: AnyPrecisionAdamW's arithmetic (verl/utils/torch_functional.py:253-329: decoupled decay, bf16 exp_avg / exp_avg_sq, Kahan
    # compensation buffer) as torch ops on bf16 CPU tensors — what the reference's optimizer would execute on the host — over 16M
    # parameters (round 3 timed the oracle's numpy bf16 EMULATION here, 10 % of the CPU step for an artefact of the checker)
    n_par = 1 << 23                                         # 16-MiB tensors: below glibc's largest mmap threshold, so their temporaries are reused too
    pa = (torch.randn(n_par, generator=gen) * 0.02).bfloat16()
    ga = (torch.randn(n_par, generator=gen) * 1e-3).bfloat16()
    ma, va, ca = torch.zeros_like(pa), torch.zeros_like(pa), torch.zeros_like(pa)
    step_t = torch.tensor(0.0)

    def adam():
        b1, b2, lr, wd, eps = 0.9, 0.999, 1e-6, 1e-2, 1e-8
        step_t.add_(1)
        pa.mul_(1 - lr * wd)
        ma.mul_(b1).add_(ga, alpha=1 - b1)
        va.mul_(b2).addcmul_(ga, ga, value=1 - b2)
        step_size = lr / (1 - b1 ** step_t)
        denom = (va.sqrt() / (1 - b2 ** step_t) ** 0.5).add_(eps)
        ca.addcdiv_(ma, denom, value=-float(step_size))
        prev = pa.clone()
        pa.add_(ca)
        ca.add_(prev.sub_(pa))
    t_adam = timed(adam, "adamw_8M_params")
    n_params_3b = 3.75e9
    seq_scale, lm_scale, vit_scale = (n_prompt * G) / NSEQ, L_LM / LS, L_VIT / LS
    fwd = seq_scale * (lm_scale * t_lm_f + vit_scale * t_vit_f + t_head_f)             # one no-grad pass over the 8 sequences
    fb = seq_scale * (lm_scale * t_lm_fb + vit_scale * t_vit_fb + t_head_fb)
    prefill = fwd * (P_txt + n_img_tok) / S
    gen_s = prefill + R * (lm_scale * t_dec + t_dec_head)
    adam_s = t_adam * n_params_3b / n_par
    step_s = gen_s + 2 * fwd + fb + adam_s
    del p, kc, vc, emb
    # ---------------------------------------------------------------- leg 2: truncated config #3 (7B widths)
    c7 = Q.VLConfig(num_layers=LS, v_depth=0)
    p7 = lm_params(c7, LS, with_head=False)
    S7 = 1614
    pos7 = torch.arange(S7)[None, :].repeat(3, 1)
    cos7, sin7 = Q.mrope_cos_sin(pos7, c7.head_dim, c7.rope_theta, c7.mrope_section)
    x7 = torch.randn(S7, c7.hidden_size, generator=gen) * 0.1

    def lm7(xx):
        for i in range(LS):
            xx = Q.lm_layer(p7, c7, i, xx, cos7, sin7, [0, S7])
        return xx
    with torch.no_grad():
        t7 = timed(lambda: lm7(x7), "cfg3_7b_widths_4_layers_fwd_1614_tokens", reps=2)
    del p7
    fwd7_per_sample = t7 * 28 / LS                            # the 28 LM layers of one 1614-token sample (ViT + head excluded)
    return {"value": n_prompt * G / step_s, "unit": "samples/s (full GRPO step: gen + old + ref + update + AdamW)", "cores": threads, "kind": "port",
            "sample": f"EXTRAPOLATED from {LS} of {L_LM} LM layers and {LS} of {L_VIT} ViT blocks (measured legs scaled by depth, sequence and step counts; "
                      f"no full-depth CPU step was run).  leg 1 = config #1 shape (Qwen2.5-VL-3B widths, 2 prompts x G=4, 224x224 image -> 64 image tokens, 700 text tokens, 512-token "
                      f"responses), fp32 torch oracle, whole passes over a {LS}-LM-layer / {LS}-ViT-block model on {NSEQ} of the 8 sequences ({T} tokens): "
                      f"no-grad pass, forward+backward, final norm + tied lm_head + log-softmax on {rows} response rows, {n_dec} KV-cache decode steps "
                      f"of the {LS} layers for the 8 rollouts, AnyPrecisionAdamW's torch op sequence on 8M bf16 parameters; every component = median of {REPS} runs after a warm-up; scaled by layers ({L_LM}/{LS}, {L_VIT}/{LS}), sequences "
                      f"(8/{NSEQ}), 512 decode steps, 3.75B parameters.  leg 2 = truncated config #3: no-grad pass of {LS} LM layers at 7B widths over "
                      f"one {S7}-token sequence",
            "timing_s": {"gen": gen_s, "old": fwd, "ref": fwd, "update_actor": fb, "adamw": adam_s},
            "measured_s": {"lm_4_layers_fwd": t_lm_f, "lm_4_layers_fwd_bwd": t_lm_fb, "vit_4_blocks_fwd": t_vit_f, "vit_4_blocks_fwd_bwd": t_vit_fb,
                           "head_fwd": t_head_f, "head_fwd_bwd": t_head_fb, "decode_step_4_layers": t_dec, "decode_head_step": t_dec_head,
                           "adamw_8M_params": t_adam, "cfg3_7b_widths_4_layers_fwd_1614_tokens": t7},
            "reps": REPS, "statistic": "median after one warm-up run; a leg throttled by the cgroup's CPU quota is re-measured (<= 3 attempts)", "spread_min_max_s": spread,
            "host": dict(host, throttled_usec_total=throttle_counters()[1] - thr0[1], nr_throttled_total=throttle_counters()[0] - thr0[0]),
            "value_excl_generation_and_adamw": n_prompt * G / (2 * fwd + fb),
            "config3_truncated": {"widths": "Qwen2.5-VL-7B", "forward_s_per_sample_28_lm_layers": fwd7_per_sample,
                                  "forward_samples_per_s": 1.0 / fwd7_per_sample,
                                  "note": "LM layers of ONE no-grad pass over one STVQA-shaped sample; a GRPO sample costs ~5 such forwards (old + ref + 3x update) "
                                          "plus generation"}}


def unrelated_rows_probe(actor, cfg, rs, P, grid, tb, ta, R, eos_id, temperature):
    """update_policy over ONE micro-batch of 4 rows that share nothing (4 different prompts + images, each with one response at the cap:
    4 x (P + R) packed tokens with gradients) — the case `_plan_passes` cannot split (a reference micro-batch is never cut) and shared-prompt
    packing cannot shrink.  Returns the peak memory of that call."""
    import dataclasses
    from verl.workers.rollout import assemble_rollout_batch
    ids, mask, pos, pix, grids = synth_prompts(cfg, 4, rs, P, grid, tb, ta)
    resp = torch.from_numpy(rs.randint(0, min(cfg.image_token_id, cfg.vocab_size) - 16, (4, R)).astype(np.int64))
    out = assemble_rollout_batch(torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(pos), resp, 1, eos_id)
    rmask = out["response_mask"]
    data = dict(input_ids=out["input_ids"], attention_mask=out["attention_mask"], position_ids=out["position_ids"], responses=out["responses"],
                multi_modal_inputs=np.array([{"pixel_values": p, "image_grid_thw": g} for p, g in zip(pix, grids)], dtype=object),
                old_log_probs=torch.full((4, R), -11.9), ref_log_probs=torch.full((4, R), -11.9),
                advantages=torch.from_numpy(rs.standard_normal((4, 1)).astype(np.float32)).repeat(1, R) * rmask)
    h0 = actor.h
    actor.h = dataclasses.replace(h0, global_batch_size_per_device=4, micro_batch_size_per_device_for_update=4)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    try:
        met = actor.update_policy(data, temperature)
        torch.cuda.synchronize()
    finally:
        actor.h = h0
    return {"rows": 4, "packed_tokens_with_grad": int(out["attention_mask"].sum()), "passes": len(actor.last_plan.get("update", [])),
            "seconds": time.perf_counter() - t0, "grad_norm": met["actor/grad_norm"], "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
            "peak_reserved_gb": torch.cuda.max_memory_reserved() / 2 ** 30}


# ------------------------------------------------------------------ main
def main():
    a = parse()
    if a.worst_case:
        a.prompts_per_gpu, a.prompt_tokens, a.responses_at_cap, a.response_cap = 128, 6144, True, 2048
        a.steps, a.warmup, a.no_cpu_baseline = 1, 0, True
    if a.through_api:
        return through_api(a)
    maybe_spawn(a)
    if a.dry_run:
        return dry_run(a)
    cfg5 = fp8_leg(a)                         # child process, BEFORE this process touches the GPU
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == max(1, a.gpus), f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    if a.ranks_share_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import datetime
        if a.ranks_share_gpu:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))          # device tensors through gloo: correctness, not speed
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=600))
            preflight(rank, world, local, need_gb={"7b": 215.0, "3b": 120.0}.get(a.model, 1.0))
    from spatialthinker_amd import ops
    from spatialthinker_amd.actor import ActorHyper, PolicyEngine
    from spatialthinker_amd.model import ParamStore, VLConfig
    from spatialthinker_amd.rollout import Generator
    from verl.utils.reward_score import spatial_sgg_compute_score
    from verl.workers.rollout import assemble_rollout_batch

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
    if a.image and not tiny:
        w_px, h_px = (int(v) for v in a.image.lower().split("x"))
        grid, (tb, ta) = (1, h_px // cfg.v_patch, w_px // cfg.v_patch), (200, 498)
    n_img = grid[1] * grid[2] // 4
    if a.prompt_tokens and not tiny:
        ta = a.prompt_tokens - n_img - 2 - tb
        assert ta > 0, "--prompt-tokens is smaller than the image + 200 leading text tokens"
    P = (tb + ta + 2 + n_img + 63) // 64 * 64
    R = 64 if tiny else a.response_cap
    G, npr = a.rollouts, a.prompts_per_gpu
    B = G * npr
    micro = 4
    n_opt = 4 if B % (4 * micro) == 0 else 1
    hyper = ActorHyper(micro_batch_size_per_device_for_update=micro, micro_batch_size_per_device_for_experience=a.experience_micro_batch,
                       global_batch_size_per_device=B // n_opt, **({"optim_strategy": "adamw"} if a.master_fp32 else {}))
    actor_store = ParamStore(cfg, trainable=True)
    actor_store.init_random(seed=7)
    if a.master_fp32:
        actor_store.enable_fp32_master()
    ref_store = ParamStore(cfg, trainable=False)
    ref_store.flat.copy_(actor_store.flat)
    actor = PolicyEngine(cfg, actor_store, hyper)
    ref = PolicyEngine(cfg, ref_store, None)
    if a.fuse_micro_batches is not None:
        actor.fuse_micro_batches = a.fuse_micro_batches
    if a.no_recompute_light:
        actor.model.recompute_light = False
    if a.tokens_grad is not None:
        actor.tokens_per_pass_grad = a.tokens_grad
    if a.dtype == "fp8":
        actor.model.enable_fp8(True, dgrad=a.fp8_dgrad, wgrad=a.fp8_wgrad)
        ref.model.enable_fp8(True)
    gen = Generator(actor.model)
    rs = np.random.RandomState(a.seed + rank)
    eos_id, pad_id = (1014, 1013) if tiny else (151645, 151643)
    temperature = 1.0
    flops = {"old": 0.0, "ref": 0.0, "update": 0.0}
    phase = {k: 0.0 for k in ("gen", "reward", "old", "ref", "adv", "update_actor")}
    tokens_total = [0]
    reserved_trace = []          # reserved GB at the end of every step (warm-up included): allocator creep shows here

    # the synthetic rollout batches (random images are ~0.4 GB of host RNG output per step) are drawn BEFORE the timed region: a
    # training job's dataloader workers prepare the next batch while the GPU runs the current step
    staged = []
    for _ in range(a.warmup + a.steps):
        batch_in = synth_prompts(cfg, npr, rs, P, grid, tb, ta)
        lens_in = np.clip(rs.normal(16 if tiny else 512, 4 if tiny else 128, B), 4 if tiny else 64, R).astype(np.int64)
        if a.responses_at_cap:
            lens_in = np.full(B, R, dtype=np.int64)
        staged.append((batch_in, lens_in))

    overlap_old = a.overlap_old == "on" and not a.old_from_rollout and not a.ranks_share_gpu
    overlap_stats = {"early_rows": 0, "feed_host_s": 0.0}
    reward_cpu_s = [0.0]                                       # the scorer thread's own wall time over all steps (overlapped with old / ref)

    class _NoTelemetry:
        phase = "idle"
    tele = _NoTelemetry()

    def one_step(step_idx, timed):
        (ids, mask, pos, pix, grids), lens = staged[step_idx]
        tick = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
        t0 = tick()
        tele.phase = "gen"
        early, extra = None, {}
        if overlap_old:
            n_cu = torch.cuda.get_device_properties(0).multi_processor_count
            early = actor.early_log_prob(ids, mask, pos, G, R, temperature, eos_id, side_stream=ops.cu_range_stream(a.tail_cus, n_cu - a.tail_cus))
            extra = dict(on_finished=early.feed, tail_stream=ops.cu_range_stream(0, a.tail_cus), tail_rows=a.tail_rows)
        resp, prompt_cache = gen.generate(ids, mask, pos, n=G, max_new_tokens=R, temperature=temperature, eos_token_id=eos_id,
                                          pad_token_id=pad_id, seed=a.seed * 1000 + step_idx, pixel_values=pix, image_grid_thw=grids,
                                          forced_lengths=lens, return_prompt_cache=True, emit_log_probs=a.old_from_rollout, **extra)
        t1 = tick()
        # ---- the (B, P+R) batch of vllm_rollout_spmd.py:144-188, assembled by the worker's own post-processing
        out = assemble_rollout_batch(torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(pos), resp.cpu(), G, eos_id)
        rmask, mask_f = out["response_mask"], out["attention_mask"]
        mm = np.repeat(np.array([{"pixel_values": p, "image_grid_thw": g} for p, g in zip(pix, grids)], dtype=object), G)
        data = dict(input_ids=out["input_ids"], attention_mask=mask_f, position_ids=out["position_ids"], responses=out["responses"],
                    multi_modal_inputs=mm)
        # ---- reward: dense spatial scorer on templated strings (no tokenizer offline), score at the last valid token
        reward_out = {}

        def score_rewards():                                 # host work that needs only the rollout: runs beside the old / ref passes
            t_ = time.perf_counter()                         # on a thread (as RayPPOTrainer.fit does, _RewardJob), joined before `adv`
            preds, gts, problems = synth_reward_strings(B, rs)       # stands for the reward manager's tokenizer.batch_decode: scorer-side host work
            scores = torch.tensor([spatial_sgg_compute_score(p, g, q)["overall"] for p, g, q in zip(preds, gts, problems)], dtype=torch.float32)
            rw = torch.zeros(B, R)
            rw[torch.arange(B), rmask.sum(1) - 1] = scores
            reward_out.update(rewards=rw, seconds=time.perf_counter() - t_)
        import threading
        reward_thread = threading.Thread(target=score_rewards, name="reward", daemon=True)
        reward_thread.start()
        if os.environ.get("ST_REWARD_THREAD", "1") == "0":    # the reference's serial order
            reward_thread.join()
        t2 = tick()
        tele.phase = "old"
        if early is not None:
            data["old_log_probs"] = early.finish(data, prompt_cache)       # most of it ran beside the decode tail; the rest runs here
            overlap_stats["early_rows"] += sum(len(s_) for s_ in early.sets[:-1]) if len(early.sets) > 1 else 0
            overlap_stats["feed_host_s"] += early.fed_s
        else:
            data["old_log_probs"] = actor.compute_log_prob(data, temperature, prompt_cache=prompt_cache,
                                                           use_rollout_log_probs=a.old_from_rollout)   # as FSDPWorker.compute_log_probs
        del prompt_cache, early
        t3 = tick()
        tele.phase = "ref"
        data["ref_log_probs"] = ref.compute_log_prob(data, temperature)
        t4 = tick()
        tele.phase = "adv"
        reward_thread.join()
        rewards = reward_out["rewards"]
        reward_cpu_s[0] += reward_out["seconds"]
        group = torch.arange(npr, dtype=torch.int32).repeat_interleave(G).cuda()
        adv, status = ops.grpo_advantage(rewards.cuda(), rmask.cuda(), group, npr)
        data["advantages"] = adv
        t5 = tick()
        tele.phase = "update_actor"
        metrics = actor.update_policy(data, temperature)
        t6 = tick()
        tele.phase = "between_steps"
        reserved_trace.append(round(torch.cuda.memory_reserved() / 2 ** 30, 1))
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

            def executed(passes, cached=False):
                groups = []
                for (s0, s1) in passes:                                      # the row ranges the engine actually ran (token-budgeted)
                    rows = list(range(s0, s1))
                    for pr in sorted(set(r // G for r in rows)):
                        groups.append((int(plen[pr]), [rlen[r] for r in rows if r // G == pr]))
                return cfg.flops_forward_grouped(groups, [] if cached else [n_patch] * len(groups), logit_rows=int(rmask.sum()),
                                                 prefix_cached=cached)
            f_exp = executed(ref.last_plan["experience"])
            f_upd = executed(actor.last_plan["update"])
            if overlap_old:                                                  # sets of finished samples, prompts cached: the responses' tokens + context once
                f_old = cfg.flops_forward_grouped([(int(plen[r // G]), [rlen[r]]) for r in range(B)], [], logit_rows=int(rmask.sum()), prefix_cached=True)
            else:
                f_old = executed(actor.last_plan["experience"], cached=actor.last_prompt_cache_hit) if actor.last_plan["experience"] else 0.0   # prompt K/V re-used from the rollout; 0 when the rollout's own log-probs were used
            flops["old"] += f_old; flops["ref"] += f_exp; flops["update"] += 3 * f_upd
            flops["reference_formulation"] = flops.get("reference_formulation", 0.0) + 5 * f_ref
            tokens_total[0] += int(mask_f.sum())
        return metrics

    for w in range(a.warmup):
        one_step(w, False)
    # live per-class roofline: every PROF_STRIDE-th launch of each kernel class is bracketed by hipEvents on its launch stream over
    # the WHOLE timed region (a prime stride, so the periodic per-layer launch sequence is sampled uniformly); the event budget is
    # sized from the step count.  Launches replayed from the decode hipGraph cannot carry events: the decode loop is timed as a
    # whole by the generator (roofline_decode).
    PROF_STRIDE = 13
    classes = {  # class id: (kernel names, bound, peak per second, unit scale, unit)
        ops.K_GEMM: ("st_gemm_nt / st_gemm_nn / st_gemm_tn / st_gemm_swiglu: gemm_nt4_kernel (256x256x64, 4 waves x 128x128, hand-scheduled K loop; gemm_asm4.hip), gemm_nt_kernel<128,128> below 128 tiles (bf16 MFMA 16x16x32, LDS-DMA staged)", "mfma", PEAK_BF16, 1e12, "TFLOP/s"),
        ops.K_ATTN_FWD: ("attn_fwd128_kernel<causal> (shared-prefix segments, MFMA 32x32x16)", "mfma", PEAK_BF16, 1e12, "TFLOP/s"),
        ops.K_ATTN_BWD: ("attn_bwd128_dq/kv/reduce kernels", "mfma", PEAK_BF16, 1e12, "TFLOP/s"),
        ops.K_VIT_ATTN: ("attn_fwd_kernel<80> (ViT full-attention layers; the backward of images >= 512 patches runs zero-padded on the head-dim-128 kernels)", "mfma", PEAK_BF16, 1e12, "TFLOP/s"),
        ops.K_VIT_WIN: ("attn_win80_fwd/bwd_kernel (ViT windows of <= 64 tokens, one launch each way; bytes = q, k, v(, dO) in + o / dq, dk, dv out + lse, delta)", "hbm", PEAK_HBM, 1e9, "GB/s"),
        ops.K_LOGPROB: ("logprob_fwd/bwd_kernel", "hbm", PEAK_HBM, 1e9, "GB/s"),
        ops.K_RMSNORM: ("rmsnorm_fwd_kernel", "hbm", PEAK_HBM, 1e9, "GB/s"),
        ops.K_ADAMW: ("adamw_kahan_kernel", "hbm", PEAK_HBM, 1e9, "GB/s"),
        ops.K_GEMM_FP8: ("gemm_mx4_kernel (4-wave tile, v_mfma_scale_f32_32x32x64_f8f6f4, e4m3 + e8m0 block scales; gemm_mxfp8_kernel with ST_FP8_TILE=8)", "mfma", PEAK_FP8, 1e12, "TFLOP/s"),
    }
    for k in classes:
        ops.prof_enable(k, max(4096, 6000 * a.steps if k == ops.K_GEMM else 1500 * a.steps), PROF_STRIDE if k != ops.K_ADAMW else 1)
    gen.stats = {k: 0 for k in gen.stats}
    ops.gemm_bytes.update(on=True, bytes=0.0, launches=0)
    torch.cuda.synchronize()
    if actor.grad_reducer() is not None:
        actor._exchange_stats_at_start = actor.grad_reducer().stats()          # the warm-up steps' exchanges are not part of the timed region
    # (not under rocprofv3: a second host thread launching kernels while the profiler collects counters crashed the profiler, round 6)
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ)
    if rank == 0 and not a.no_telemetry and not under_profiler:
        tele = ChipTelemetry(period=max(0.1, min(0.5, 0.02 * a.steps)))     # ~120 samples per 30-s step at the driver's 20 steps
        tele.start()
    if world > 1:
        dist.barrier()
    t_start = time.perf_counter()
    probe = None
    try:
        for s in range(a.steps):
            one_step(a.warmup + s, True)
        torch.cuda.synchronize()
        update_passes = len(actor.last_plan.get("update", []))
        if a.worst_case:
            probe = unrelated_rows_probe(actor, cfg, rs, P, grid, tb, ta, R, eos_id, temperature)
    except (torch.OutOfMemoryError, RuntimeError) as e:
        if not a.worst_case and not isinstance(e, torch.OutOfMemoryError):
            raise
        msg = f"{type(e).__name__}: {str(e).splitlines()[0][:400]}"
        sys.stderr.write(f"[bench] the workload does not fit this GPU: {msg}\n")
        if rank == 0:
            print(json.dumps({"metric": "GRPO samples/sec (G=8 rollouts/prompt) Qwen2.5-VL-7B at 1/2/4/8 MI355X", "value": None, "error": msg,
                              "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "peak_reserved_gb": torch.cuda.max_memory_reserved() / 2 ** 30,
                              "config": {"workload": f"{name}, G={G}, {npr} prompts/GPU, prompt {tb + ta + 2 + n_img} tokens, max_response_length {R}"}}))
        raise SystemExit(5)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    telemetry = None
    if isinstance(tele, ChipTelemetry):
        tele.stop()
        telemetry = tele.summary()
    prof = {}
    gemm_shapes = None
    for k in classes:
        seen = ops.prof_seen(k)
        if k == ops.K_GEMM:                       # per-launch view: the per-(form, M, N, K) table behind `roofline`
            ev_ms, ev_units, ev_tags = ops.prof_read_events(k)
            n_launch, ms, units = len(ev_ms), float(ev_ms.sum(dtype=np.float64)), float(ev_units.sum())
            gemm_shapes = gemm_shape_table(ev_ms, ev_units, ev_tags)
        else:
            n_launch, ms, units = ops.prof_read(k)
        ops.prof_disable(k)
        prof[k] = (seen, n_launch, ms, units)
    phase_max = dict(phase)
    red = actor.grad_reducer()
    xs = red.stats() if red is not None else None
    xs0 = getattr(actor, "_exchange_stats_at_start", None)
    if xs is not None and xs0 is not None:
        xs = {k: xs[k] - xs0[k] for k in xs}
    exch = [xs["allreduce_s"], xs["allreduce_exposed_s"]] if xs else [0.0, 0.0]
    if world > 1:
        t = torch.tensor([elapsed] + [phase[k] for k in phase] + exch, device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        phase_max = {k: float(t[1 + i].item()) for i, k in enumerate(phase)}
        exch = [float(t[-2].item()), float(t[-1].item())]
    if rank == 0:
        samples = B * world * a.steps
        actor_t = phase["old"] + phase["ref"] + phase["update_actor"]

        def roof(k):
            names, bound, peak, scale, unit = classes[k]
            seen, n_launch, ms, units = prof[k]
            ach = units / (ms * 1e-3) if ms > 0 else None
            return {"bound": bound, "kernel": names, "achieved": ach / scale if ach else None, "peak": peak / scale, "unit": unit,
                    "frac": ach / peak if ach else None, "launches_timed": n_launch, "launches_seen": seen,
                    "avg_launch_ms": ms / max(n_launch, 1)}
        dec_traffic = None
        main_roof = roof(ops.K_GEMM)
        if gemm_shapes:
            main_roof["by_shape"] = gemm_shapes[:6]          # the costliest (form, M, N, K) groups of the sampled launches; --gemm-table writes all
            if a.gemm_table:
                os.makedirs(os.path.dirname(os.path.abspath(a.gemm_table)), exist_ok=True)
                with open(a.gemm_table, "w") as f:
                    json.dump({"command": " ".join(sys.argv), "stride": PROF_STRIDE, "class_tflops": main_roof["achieved"], "shapes": gemm_shapes}, f, indent=1)
        # the clock the chip ran the GEMM-heavy phases at (ChipTelemetry): the MFMA peak scales with the shader clock (2.5 PF/s is the
        # 2400-MHz figure), so `frac_of_clocked_peak` is what the kernel made of the cycles it was given, `frac` what it made of the datasheet
        if telemetry and telemetry.get("gemm_phases") and main_roof.get("achieved"):
            ck = telemetry["gemm_phases"]["clock_mhz_mean"]
            main_roof["clock_mhz"] = ck
            main_roof["clock_mhz_min"] = telemetry["gemm_phases"]["clock_mhz_min"]
            main_roof["socket_power_w"] = telemetry["gemm_phases"].get("socket_power_w_mean")
            main_roof["frac_of_clocked_peak"] = main_roof["achieved"] * 1e12 / (PEAK_BF16 * ck / 2400.0)
        main_roof["algorithmic_bytes_per_launch"] = ops.gemm_bytes["bytes"] / max(1, ops.gemm_bytes["launches"])
        main_roof["algorithmic_bytes"] = "operands once + result once (+ residual / fp32 read-modify-write), averaged over the class's launches"
        # HBM traffic per launch of the GEMM class: a PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE
        # doubled per the gfx950 note of MI355X_MICROARCH.md) over THIS command (`bench.py --steps 1`, tools/bench_traffic.sh) is kept
        # under profiles/; it is quoted only while the kernel sources it was taken on are the ones being run (sha of the GEMM sources),
        # otherwise the field is null — a measured number must not outlive the kernel it measured
        tp = os.path.join(ROOT, "profiles", "r05_gemm_traffic.json")
        main_roof["traffic"] = None
        if os.path.exists(tp) and a.model == "7b":
            tj = json.load(open(tp))
            if tj.get("kernel_source_sha16") == gemm_source_sha():
                main_roof["traffic"] = tj.get("hbm_bytes_per_launch")
                main_roof["traffic_source"] = "profiles/r05_gemm_traffic.json (PMC pass over bench.py --steps 1, time-weighted over the top GEMM instantiations)"
                main_roof["traffic_over_algorithmic_operand_bytes"] = tj.get("traffic_over_algorithmic")
                if dec_traffic is None:
                    dec_traffic = tj.get("decode_hbm_bytes_per_iteration")
            else:
                main_roof["traffic_stale"] = "profiles/r05_gemm_traffic.json was measured on other kernel sources"
        st = gen.stats
        dec = None
        if st["decode_s"] > 0:
            # at ~340 rows the iteration sits AT the ridge of the roofline (2 flop per weight byte and row = ~340 flop/B vs 312 flop/B):
            # both bounds are reported, `frac` is the larger of the two fractions (the bound that is closer to binding)
            bw = st["decode_bytes"] / st["decode_s"]
            tf = st.get("decode_flops", 0.0) / st["decode_s"]
            dec = {"bound": "hbm" if bw / PEAK_HBM >= tf / PEAK_BF16 else "mfma",
                   "kernel": "decode iteration (hipGraph replay: decode GEMM tiles x4 + attn_fwd128_kernel<false> + merge + fused finishes per layer, "
                             "lm_head, sampler + decode_step_kernel)",
                   "achieved": bw / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": max(bw / PEAK_HBM, tf / PEAK_BF16),
                   "hbm": {"achieved_GBps": bw / 1e9, "peak_GBps": PEAK_HBM / 1e9, "frac": bw / PEAK_HBM},
                   "mfma": {"achieved_TFLOPs": tf / 1e12, "peak_TFLOPs": PEAK_BF16 / 1e12, "frac": tf / PEAK_BF16},
                   "traffic": dec_traffic,
                   "traffic_source": ("profiles/r05_gemm_traffic.json: PMC FETCH_SIZE x2 + WRITE_SIZE per iteration of a 512-row decode phase "
                                      "(tools/gen_flat.py 6 64 8); this run's mean live rows are below that") if dec_traffic else None,
                   "iterations": st["decode_steps"], "ms_per_iteration": st["decode_s"] / st["decode_steps"] * 1e3,
                   "mean_rows_per_iteration": st["decode_row_steps"] / st["decode_steps"],
                   "algorithmic_bytes": "every LM weight once per iteration + K/V of the live context (prompt K/V once per prompt)",
                   "algorithmic_flops": "2 flop per LM weight and row + 4*D flop per (query head, cached key)"}
        out = {
            "metric": "GRPO samples/sec (G=8 rollouts/prompt) Qwen2.5-VL-7B at 1/2/4/8 MI355X",
            "value": samples / elapsed, "unit": "samples/s", "n_gpus": world, "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if a.dtype == "bf16" else ("fp8 (MX e4m3 forward" + (" + input-gradient" if a.fp8_dgrad else "") + (" + weight-gradient" if a.fp8_wgrad else "")
                                                            + " projection GEMMs; bf16 attention / lm_head / "
                                                            + ("rest of the backward" if (a.fp8_dgrad or a.fp8_wgrad) else "backward") + " / decode)"),
            "data": f"synthetic (random-init weights at real shapes; prompts: {tb + ta + 2} text + {n_img} image tokens from "
                                     f"{grid[1] * grid[2]} random patches; response lengths ~ clip(N(512,128),64,cap) enforced by forcing EOS; templated reward strings)",
            "config": {"workload": f"{name} dense spatial-reward GRPO step (gen + reward + old/ref log-probs + advantage + update), G={G}, "
                                   f"{npr} prompts/GPU, micro-batch {micro}, {n_opt} optimizer steps/step, max_response_length {R}",
                       "global_batch": B * world, "seq_len": P + R, "parallelism": f"dp{world}" + (" (ranks share ONE GPU, gloo exchange: test mode)" if a.ranks_share_gpu else "")},
            "timing_s": {k: v / a.steps for k, v in phase.items()},
            "reward_scorer": {"host_s_per_step": reward_cpu_s[0] / max(1, a.steps + a.warmup), "placement": "thread beside the old / ref log-prob passes, joined before adv "
                              "(timing_s.reward = what is left of it in front of `old`; ST_REWARD_THREAD=0: the reference's serial order)"},
            "timing_s_max_over_ranks": {k: v / a.steps for k, v in phase_max.items()},
            # gradient exchange per step (max over ranks; device events on the compute stream, actor.GradReducer): the whole exchange from the
            # first layer slice sent during the last backward pass to the averaged gradients, and the part of it NOT hidden behind the
            # backward (the compute stream's wait inside finish()); 0 at N = 1 (no process group, no exchange)
            "allreduce_s": exch[0] / a.steps, "allreduce_exposed_s": exch[1] / a.steps,
            "grad_exchange": ({"mode": red.mode, "payload": red.payload, "exchanges_per_step": xs["exchanges"] / a.steps,
                               "early_fraction": xs["early_fraction"] / max(1, xs["exchanges"]), "bytes_per_exchange": actor_store.grad.numel() * (4 if red.payload == "fp32" else 2),
                               "staging_allocations": red.pool.allocations, "staging_gb": red.pool.allocated_bytes / 2 ** 30} if red is not None else None),
            "perf_throughput_tokens_per_s_per_gpu": tokens_total[0] / elapsed,
            "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
            "peak_reserved_gb": torch.cuda.max_memory_reserved() / 2 ** 30,
            "reserved_gb_after_each_step": reserved_trace,
            "overlap_old": ({"mode": f"old-policy log-probs of finished samples on CUs [{a.tail_cus}, all) beside decode phases of <= {a.tail_rows} rows on CUs [0, {a.tail_cus})",
                             "rows_computed_during_the_rollout_per_step": overlap_stats["early_rows"] / max(1, a.steps + a.warmup),
                             "feed_host_s_per_step": overlap_stats["feed_host_s"] / max(1, a.steps + a.warmup)} if overlap_old else None),
            "passes_per_step": {"update": update_passes, "old": len(actor.last_plan.get("experience", [])),
                                "ref": len(ref.last_plan.get("experience", []))},
            "rollout_prompt_chunks": len(getattr(gen, "last_chunks", [(0, npr)])), "prompt_cache_hit": bool(actor.last_prompt_cache_hit),
            "prompt_tokens": tb + ta + 2 + n_img, "responses_at_cap": bool(a.responses_at_cap),
            "old_log_probs": getattr(actor, "last_log_prob_source", "forward") + (" (OPT-IN --old-from-rollout: not the reference-faithful headline configuration)" if a.old_from_rollout else " (second forward over the responses on the rollout's prompt K/V, as the reference recomputes them)"),
            **({"worst_case": True, "unrelated_rows_probe": probe} if a.worst_case else {}),
            "actor_mfu": (flops["old"] + flops["ref"] + flops["update"]) / actor_t / PEAK_BF16 if actor_t > 0 else None,
            "actor_mfu_reference_flops": flops.get("reference_formulation", 0.0) / actor_t / PEAK_BF16 if actor_t > 0 else None,
            "roofline": main_roof,
            "roofline_classes": [roof(k) for k in classes if k != ops.K_GEMM and prof[k][1] > 0],
            "roofline_decode": dec,
            "telemetry": telemetry,
            "parity": parity_record(),
        }
        if cfg5 is not None:
            out["cfg5_fp8"] = cfg5
        if world == 1 and not a.no_cpu_baseline and not tiny:
            out["cpu_baseline"] = cpu_baseline()
        elif tiny and not a.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
