"""Entry point: `python3 -m verl.trainer.main config=<yaml> a.b.c=value ...` — same command line as the reference
(verl/trainer/main.py:88-105; scripts/spatialthinker_*_grpo.sh drop in unchanged).

Launch: one process per GPU.  When started as a single process on a node with several GPUs and
trainer.n_gpus_per_node > 1, this module re-launches itself under torch.distributed.run (a CHILD process: nothing has
touched the GPU yet) so the shipped scripts need no edit."""
import json
import os
import subprocess
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL across processes)


def _maybe_spawn(n_gpus: int) -> bool:
    if "RANK" in os.environ or n_gpus <= 1:
        return False
    import torch
    n = min(n_gpus, torch.cuda.device_count())          # device_count() does not initialise the GPU on this image
    if n <= 1:
        return False
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", os.environ.get("MASTER_PORT", "29531"), "-m", "verl.trainer.main"] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    from .config import load_config
    cfg = load_config(sys.argv[1:])
    _maybe_spawn(cfg.trainer.n_gpus_per_node * cfg.trainer.nnodes)
    cfg.deep_post_init()
    rank = int(os.environ.get("RANK", 0))
    if rank == 0:
        print(json.dumps(cfg.to_dict(), indent=2))
    import torch
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
    from ..single_controller import SPMDWorkerGroup
    from ..utils.dataset import RLHFDataset, SyntheticSTVQADataset
    from ..utils.tokenizer import get_processor, get_tokenizer, is_synthetic
    from ..workers.fsdp_workers import FSDPWorker
    from ..workers.reward import CustomRewardManager
    from .ray_trainer import RayPPOTrainer

    mc = cfg.worker.actor.model
    tokenizer = get_tokenizer(mc.model_path, trust_remote_code=mc.trust_remote_code, use_fast=True)
    processor = get_processor(mc.model_path, trust_remote_code=mc.trust_remote_code, use_fast=True)
    reward_fn = CustomRewardManager(tokenizer, cfg.worker.reward)

    def dataset(spec, train: bool):
        if not spec:
            return None
        if spec.startswith("synthetic:") or is_synthetic(mc.model_path):
            from spatialthinker_amd.pretrained import synthetic_config
            mcfg, _ = synthetic_config(mc.model_path if is_synthetic(mc.model_path) else "random:7b")
            tiny = mcfg.hidden_size <= 512
            grid, text = ((1, 8, 8), (8, 12)) if tiny else ((1, 32, 42), (200, 564))
            # "synthetic:stvqa:224x224@train": square/rect images of that pixel size (SURVEY 8d': 224/448/896 -> 256/1024/4096
            # patches) with 700 text tokens, instead of the STVQA-shaped 588x448 default
            # "synthetic:stvqa:len=512,128@train": the rows also carry forced response lengths ~ clip(N(512, 128), 64, cap) for their
            # rollouts (benchmark mode: random-init weights never emit EOS on their own)
            name = spec.split("@")[0].split(":")
            lengths = None
            for field in name[2:]:
                if field.startswith("len="):
                    mu, sd = (float(v) for v in field[4:].split(","))
                    lengths = (mu, sd, int(cfg.worker.rollout.n), int(cfg.data.max_response_length))
                elif "x" in field:
                    w_px, h_px = (int(v) for v in field.lower().split("x"))
                    grid, text = (1, h_px // mcfg.v_patch, w_px // mcfg.v_patch), ((8, 12) if tiny else (200, 500))
            return SyntheticSTVQADataset(mcfg, tokenizer, size=max(4 * cfg.data.rollout_batch_size, 64), max_prompt_length=cfg.data.max_prompt_length,
                                         seed=cfg.data.seed + (0 if train else 1), grid=grid, text_tokens=text, response_lengths=lengths if train else None)
        # ray_trainer.py:267-313: the train set takes mixed_data / text_only, the validation set does not; both keep the dataset's own
        # shuffle (seed 42) — data.shuffle / data.seed drive the SAMPLER
        extra = dict(mixed_data=cfg.data.mixed_data, text_only=cfg.data.text_only) if train else {}
        return RLHFDataset(spec, tokenizer, processor, prompt_key=cfg.data.prompt_key, answer_key=cfg.data.answer_key, image_key=cfg.data.image_key,
                           max_prompt_length=cfg.data.max_prompt_length, truncation="right", format_prompt=cfg.data.format_prompt,
                           min_pixels=cfg.data.min_pixels, max_pixels=cfg.data.max_pixels, **extra)

    # the trainer validates the batch-size relations on the user's numbers BEFORE the worker scales global_batch_size by
    # rollout.n (ray_trainer.py:238-263 runs before fsdp_workers.py:130-136 in the reference too)
    trainer = RayPPOTrainer(cfg, tokenizer, processor, None, None, reward_fn, reward_fn, dataset(cfg.data.train_files, True), dataset(cfg.data.val_files, False))
    role = "actor_rollout" if cfg.algorithm.disable_kl else "actor_rollout_ref"      # colocated roles (ray/base.py:453-493)
    wg = SPMDWorkerGroup(FSDPWorker(cfg.worker, role))
    critic_wg = SPMDWorkerGroup(FSDPWorker(cfg.worker, "critic")) if cfg.algorithm.adv_estimator == "gae" else None      # ray_trainer.py:428-434
    trainer.set_worker_groups(wg, wg, critic_wg)
    trainer.init_workers()
    trainer.fit()


if __name__ == "__main__":
    main()
