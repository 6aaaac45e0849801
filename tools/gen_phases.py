import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ST_GEN_DEBUG"] = "1"
from bench import synth_prompts
from spatialthinker_amd.model import ParamStore, VLConfig, Qwen25VL
from spatialthinker_amd.rollout import Generator
cfg = VLConfig.qwen2_5_vl_7b()
st = ParamStore(cfg, trainable=False); st.init_random(1)
gen = Generator(Qwen25VL(cfg, st))
rs = np.random.RandomState(0)
npr, G, R = 64, 8, 2048
ids, mask, pos, pix, grids = synth_prompts(cfg, npr, rs, 1152, (1, 32, 42))
for it in range(2):
    lens = np.clip(rs.normal(512, 128, npr * G), 64, R).astype(np.int64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = gen.generate(ids, mask, pos, n=G, max_new_tokens=R, temperature=1.0, eos_token_id=151645, pad_token_id=151643, seed=it,
                       pixel_values=pix, image_grid_thw=grids, forced_lengths=lens)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    st_ = gen.stats
    print(f"total {tot:.3f}s: prefill {st_['prefill_s']:.3f}s, decode loops {st_['decode_s']:.3f}s over {st_['decode_steps']} iterations in {st_['phases']} phases, "
          f"rest (graph capture, host bookkeeping, syncs) {tot - st_['prefill_s'] - st_['decode_s']:.3f}s", flush=True)
    gen.stats = {k: 0 for k in gen.stats}
