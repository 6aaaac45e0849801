import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
for R, C in [(6528, 37888), (6528, 3584), (6528, 18944), (6528, 4608)]:
    x = torch.randn(R, C, device="cuda").bfloat16()
    y = ops.transpose(x)
    assert torch.equal(y, x.t().contiguous())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ops.transpose(x)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"{R}x{C}: {t*1e6:.0f} us, {R*C*4/t/1e12:.2f} TB/s (read+write)")
