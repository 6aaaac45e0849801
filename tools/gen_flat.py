"""One decode phase at a fixed batch: 64 prompts x 8 rollouts that all stop after LEN tokens (no compaction, no tail) — per-kernel
times of the 512-row decode iteration under rocprofv3.      python tools/gen_flat.py [LEN] [prompts] [G]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_prompts
from spatialthinker_amd.model import ParamStore, VLConfig, Qwen25VL
from spatialthinker_amd.rollout import Generator
LEN = int(sys.argv[1]) if len(sys.argv) > 1 else 200
npr = int(sys.argv[2]) if len(sys.argv) > 2 else 64
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cfg = VLConfig.qwen2_5_vl_3b() if os.environ.get("ST_MODEL") == "3b" else VLConfig.qwen2_5_vl_7b()
st = ParamStore(cfg, trainable=False); st.init_random(1)
gen = Generator(Qwen25VL(cfg, st))
rs = np.random.RandomState(0)
ids, mask, pos, pix, grids = synth_prompts(cfg, npr, rs, 1152, (1, 32, 42))
lens = np.full(npr * G, LEN, dtype=np.int64)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gen.generate(ids, mask, pos, n=G, max_new_tokens=LEN, temperature=1.0, eos_token_id=151645, pad_token_id=151643, seed=it,
                 pixel_values=pix, image_grid_thw=grids, forced_lengths=lens)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    s = gen.stats
    print(f"rows {npr * G}: total {tot:.3f}s, prefill {s['prefill_s']:.3f}s, decode {s['decode_s']:.3f}s over {s['decode_steps']} iterations = "
          f"{1e3 * s['decode_s'] / max(1, s['decode_steps']):.2f} ms/iteration", flush=True)
    gen.stats = {k: 0 for k in gen.stats}
