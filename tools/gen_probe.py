"""Generation-only probe (7B shapes): time per decode step with and without hipGraph replay."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_prompts
from spatialthinker_amd.model import ParamStore, VLConfig, Qwen25VL
from spatialthinker_amd.rollout import Generator

npr = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R = int(sys.argv[2]) if len(sys.argv) > 2 else 128
graph = (sys.argv[3] == "graph") if len(sys.argv) > 3 else True
cfg = VLConfig.qwen2_5_vl_7b()
st = ParamStore(cfg, trainable=False); st.init_random(1)
gen = Generator(Qwen25VL(cfg, st), autotune=(os.environ.get("ST_TUNE", "0") == "1"))
gen.max_decode_batch = int(os.environ.get("ST_MAX_DECODE", "256"))
rs = np.random.RandomState(0)
ids, mask, pos, pix, grids = synth_prompts(cfg, npr, rs, 1152, (1, 32, 42))
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = gen.generate(ids, mask, pos, n=8, max_new_tokens=R, temperature=1.0, eos_token_id=151645, pad_token_id=151643, seed=it,
                       pixel_values=pix, image_grid_thw=grids, ignore_eos=True, use_graph=graph)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"npr={npr} R={R} graph={graph}: total {t1 - t0:.3f}s -> {(t1 - t0) / R * 1e3:.2f} ms/step (incl. prefill)")
