"""Convert this engine's checkpoint (actor/huggingface/*.safetensors + actor/optim_world_size_1_rank_0.pt) into the reference's
layout — model_/extra_state_world_size_W_rank_r.pt with DTensor Shard(0) shards on an ("fsdp",) mesh — so that the reference's
scripts/model_merger.py can read it.  MERGER-ONLY: a reference run cannot RESUME from the export (FSDPCheckpointManager.load_checkpoint,
verl/utils/checkpoint/fsdp_checkpoint_manager.py:52-81, loads optimizer files in its own per-rank layout unconditionally).
    python tools/export_reference_checkpoint.py <global_step_N/actor> <out_dir> <world_size>
Runs stand-alone (no process group may exist: the device mesh is built on torch's in-process fake backend)."""
import glob
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(src: str, out: str, world: int):
    from safetensors.torch import load_file
    from spatialthinker_amd.model import ParamStore, VLConfig
    from verl.utils.checkpoint import export_reference_layout
    import json
    sd = {}
    for shard in sorted(glob.glob(os.path.join(src, "huggingface", "*.safetensors"))):
        sd.update(load_file(shard))
    optim_state, steps, sched = None, 0, 0
    op = os.path.join(src, "optim_world_size_1_rank_0.pt")
    if os.path.exists(op):
        opt = torch.load(op, map_location="cpu")
        cfg = VLConfig.from_hf_dict(json.load(open(os.path.join(src, "huggingface", "config.json"))))
        store = ParamStore(cfg, device="cpu", trainable=False)
        names = {"exp_avg": "m", "exp_avg_sq": "v", "compensation": "c"}
        per = {}
        for key, short in names.items():
            if short not in opt:
                continue
            views = {n: store._view(opt[short], n) for n in store.layout}
            for hf_name, t in store.export_hf(views).items():
                per.setdefault(hf_name, {"step": torch.tensor(float(opt["opt_steps"]))})[key] = t
        optim_state, steps, sched = per, int(opt["opt_steps"]), int(opt["sched_steps"])
    keep_optim = os.environ.get("ST_EXPORT_OPTIM", "0") == "1"      # name-keyed optimizer files: this build's own format (the reference cannot load them)
    export_reference_layout(sd, optim_state, out, world, opt_steps=steps, sched_steps=sched, write_optim=keep_optim)
    print(f"wrote {world} x (model, extra_state{', optim' if keep_optim else ''}) shard files to {out} (transformers-4.49 parameter names, no rng entry)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
