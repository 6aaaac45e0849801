#!/bin/bash
# transposing quantiser with column-contiguous output lanes: bit-exactness tests, timing of the two quantisers on the gradient shapes
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -k "quantiser" 2>&1 | tail -3
python3 - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from spatialthinker_amd import ops
def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (R, C) in ((16384, 3584), (16384, 18944), (16384, 37888)):
    x = torch.randn(R, C, device="cuda").bfloat16()
    t1 = timeit(lambda: ops.mxfp8_quantize(x)); t2 = timeit(lambda: ops.mxfp8_quantize_t(x)); t3 = timeit(lambda: ops.mxfp8_quantize_both(x))
    b = R * C
    print(f"{R}x{C}: row-wise {t1:7.1f} us ({3 * b / t1 / 1e6:5.2f} TB/s)  transposed {t2:7.1f} us ({3 * b / t2 / 1e6:5.2f} TB/s)  both {t3:7.1f} us ({4 * b / t3 / 1e6:5.2f} TB/s)")
PY
