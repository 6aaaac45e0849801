"""FULL-DEPTH parity at the real Qwen2.5-VL-7B configuration: 28 LM layers + 32 ViT blocks, random-init bf16-exact weights, the
log-probs of one rollout group (2 rollouts behind one image + text prompt) from the HIP engine (shared-prompt packing, production GEMM /
attention dispatch) against the fp32 CPU oracle run sequence by sequence — with a plain torch bf16 evaluation of the SAME model on the
same GPU (the oracle's own torch code on bf16 CUDA tensors: hipBLASLt matmuls, fp32 softmax / norm statistics as HF's eager path) as
the yardstick of what 60 layers of bf16 rounding cost.  Criterion (as tests/test_gpu_model.py): the engine's error
against fp32 is no larger than (factor 1.0, round 6) the plain bf16 evaluation's own error.

Weights are generated once, tensor by tensor, on the GPU (8.3 B parameters: no host copy at all), into the engine's flat store and into a
bf16 CUDA dict; the CPU oracle pulls each tensor back as fp32 when it needs it."""
import time
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import tiny  # noqa: E402
from oracle import positions as P  # noqa: E402
from oracle import qwen25vl as Q  # noqa: E402

from fullsize import EOS, FULL, VISION_END  # noqa: E402

DEPTH = dict(FULL, num_layers=28, v_depth=32, v_fullatt=[7, 15, 23, 31])


def _gen(name: str, shape) -> torch.Tensor:
    """Deterministic bf16 CUDA tensor: norm weights 1 + noise, biases small, matrices ~U(-a, a) with a = sqrt(3) * 0.02 (std 0.02, the HF
    initializer_range) from an integer hash of the element index, computed ON THE GPU (8.3 B values: seconds there, half an hour in numpy)."""
    n = int(np.prod(shape))
    seed = zlib.crc32(name.encode()) & 0x7FFFFFFF
    if "norm" in name or "ln_q" in name:
        return torch.from_numpy((1.0 + 0.1 * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32)).to("cuda", torch.bfloat16)
    if name.endswith(".bias"):
        return torch.from_numpy((0.02 * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32)).to("cuda", torch.bfloat16)
    i = torch.arange(n, dtype=torch.int64, device="cuda")
    h = ((i * 2654435761 + seed) ^ ((i >> 7) * 40503)) % 65521
    return ((h.to(torch.float32) / 65521.0 - 0.5) * (2.0 * 0.02 * 3 ** 0.5)).to(torch.bfloat16).reshape(shape)


class _FromDevice(dict):
    """name -> fp32 CPU tensor, converted from the bf16 CUDA copy on access (nothing cached: one tensor of host memory at a time)"""

    def __init__(self, dev):
        super().__init__()
        self.dev = dev

    def __getitem__(self, k):
        return self.dev[k].float().cpu()


def _group(rs, n_roll=2, text=(24, 40), grid=(1, 16, 20), R=32, Pc=192):
    n_img = grid[0] * grid[1] * grid[2] // 4
    prompt = (rs.randint(0, 150000, text[0]).tolist() + [FULL["vision_start_token_id"]] + [FULL["image_token_id"]] * n_img + [VISION_END]
              + rs.randint(0, 150000, text[1]).tolist())
    ids = np.full((n_roll, Pc + R), 151643, dtype=np.int64)
    mask = np.zeros((n_roll, Pc + R), dtype=np.int64)
    for r, L in zip(range(n_roll), (R, R - 9)):
        ids[r, Pc - len(prompt):Pc] = prompt; mask[r, Pc - len(prompt):Pc] = 1
        ids[r, Pc:Pc + L] = rs.randint(0, 150000, L - 1).tolist() + [EOS]; mask[r, Pc:Pc + L] = 1
    px = rs.standard_normal((grid[0] * grid[1] * grid[2], 1176)).astype(np.float32)
    g = np.asarray([grid], dtype=np.int64)
    pos = np.stack([P.mrope_position_ids(ids[r], g, mask[r], image_token_id=FULL["image_token_id"],
                                         vision_start_token_id=FULL["vision_start_token_id"]) for r in range(n_roll)])
    return ids, mask, pos, px, g, R


def test_full_depth_7b_log_probs_vs_fp32_oracle_with_a_torch_bf16_yardstick(measured):
    from spatialthinker_amd import model as mdl
    t0 = time.time()
    cfg = mdl.VLConfig(**DEPTH)
    store = mdl.ParamStore(cfg, trainable=False)
    dev_params = {}

    class _Feed(dict):                                                # load_hf_state_dict pulls every HF name exactly once
        def __getitem__(self, k):
            dev_params[k] = _gen(k, shapes[k])
            return dev_params[k]
    shapes = tiny.param_shapes(DEPTH)
    store.load_hf_state_dict(_Feed())
    assert set(dev_params) == set(shapes)
    eng = mdl.Qwen25VL(cfg, store)
    t_load = time.time() - t0

    rs = np.random.RandomState(23)
    ids, mask, pos, px, g, R = _group(rs)
    k = ids.shape[0]
    m = mask[:, -R:].astype(bool)
    ocfg = Q.VLConfig(**DEPTH)

    # ---- engine: the group behind one prompt copy
    b = eng.stage(ids, mask, pos, R, px, g, groups=[0] * k)
    lp_e = eng.log_probs(b, temperature=1.0).float().cpu().numpy()
    # ---- yardstick: the oracle's torch code on bf16 CUDA tensors, every rollout as its own full sequence
    to = lambda a, dt=None: torch.from_numpy(a).to("cuda") if dt is None else torch.from_numpy(a).to("cuda", dt)
    with torch.no_grad():
        lp_b = Q.response_log_probs(dev_params, ocfg, to(ids), to(mask), to(pos), R, 1.0, to(np.concatenate([px] * k, 0), torch.bfloat16),
                                    np.concatenate([g] * k, 0)).float().cpu().numpy()
    t_gpu = time.time() - t0 - t_load
    # ---- fp32 CPU oracle, sequence by sequence
    with torch.no_grad():
        lp_o = Q.response_log_probs(_FromDevice(dev_params), ocfg, torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(pos), R, 1.0,
                                    torch.from_numpy(np.concatenate([px] * k, 0)), np.concatenate([g] * k, 0)).numpy()
    t_cpu = time.time() - t0 - t_load - t_gpu
    err_e = float(np.abs(lp_e[m] - lp_o[m]).max())
    err_b = float(np.abs(lp_b[m] - lp_o[m]).max())
    rms_e = float(np.sqrt(np.mean((lp_e[m] - lp_o[m]) ** 2)))
    rms_b = float(np.sqrt(np.mean((lp_b[m] - lp_o[m]) ** 2)))
    print(f"full-depth 7B (28 + 32 layers), {int(m.sum())} response tokens: engine max|dlogp| {err_e:.4f} (rms {rms_e:.4f}); "
          f"torch bf16 eager {err_b:.4f} (rms {rms_b:.4f}); mean logp {lp_o[m].mean():.3f}; load {t_load:.0f}s gpu {t_gpu:.0f}s cpu oracle {t_cpu:.0f}s")
    measured("depth7b_engine_max_abs_dlogp", err_e)
    measured("depth7b_torch_bf16_max_abs_dlogp", err_b)
    measured("depth7b_engine_rms_dlogp", rms_e)
    measured("depth7b_torch_bf16_rms_dlogp", rms_b)
    assert np.all(lp_e[~m] == 0)
    assert np.isfinite(lp_o[m]).all() and lp_o[m].std() > 0.05          # the comparison is not vacuous
    assert err_e <= 1.0 * err_b and rms_e <= 1.0 * rms_b          # round 6: factor 1.0 (SURVEY 8c iii as written); measured 0.105 vs 0.220 max, 0.045 vs 0.078 rms
