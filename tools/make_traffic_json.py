"""profiles/r04_gemm_traffic.json (r03_… in round 3) from the PMC passes of tools/bench_traffic.sh.
    python tools/make_traffic_json.py <pmc FETCH json> <pmc WRITE json> <bench line of the FETCH pass> <out.json>
FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: on gfx950 it counts 128-byte
requests as 64 bytes for wide coalesced reads).  Training GEMM class = the 256x256 tiles (4-wave asm tile, 8-wave tile in its TN /
SwiGLU forms) and the 128x128 kernel; decode iteration = everything replayed from the decode hipGraph."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fetch, write, bench_line, out = sys.argv[1:5]
F, W = json.load(open(fetch)), json.load(open(write))
DF, DW = (json.load(open(sys.argv[5])), json.load(open(sys.argv[6]))) if len(sys.argv) > 6 else ({}, {})     # decode probe (tools/gen_flat.py 40 64 8)
line = json.loads(open(bench_line).read().strip().splitlines()[-1])


def is_decode_a4(k):
    """the decode entry's instantiation of the 4-wave tile: last template argument (DEC) true"""
    if not k.startswith("gemm_nt4_kernel<") or ">" not in k:
        return False
    args = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
    return len(args) == 9 and args[-1] == "true"


def is_train_gemm(k):
    if is_decode_a4(k) or k.startswith("gemm_a4_swiglu_finish_kernel"):
        return False
    return k.startswith(("gemm_nt4_kernel", "gemm_a4_finish_kernel", "gemm_nt_kernel")) or (k.startswith("gemm_tile_kernel<256, 256") and "true, false, true" not in k[:80] and True)


DECODE = ("gemm_tile_kernel<256, 160", "gemm_tile_kernel<256, 128", "gemm_tile_kernel<128,", "gemm_tile_kernel<64,", "attn_fwd128_kernel<false>",
          "attn_merge_kernel", "decode_finish", "decode_step_kernel", "sample_kernel", "sample_filter_kernel", "gemm_skinny_finish",
          "gemm_a4_swiglu_finish_kernel", "attn_decode128_kernel", "gemm_tile_kernel<256, 256", "gemm_swiglu512_kernel")      # the last one: the decode lm_head (the probe's prefill runs the 4-wave tile)
tr_bytes = tr_launch = 0.0
per_kernel = {}
dec_bytes = 0.0
for k in set(F) | set(W):
    f = F.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
    w = W.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
    n = max(F.get(k, {}).get("dispatches", 0), W.get(k, {}).get("dispatches", 0))
    if is_train_gemm(k):
        tr_bytes += f + w
        tr_launch += n if not k.startswith("gemm_a4_finish_kernel") else 0
        per_kernel[k[:90]] = {"launches": n, "hbm_bytes_per_launch": (f + w) / max(n, 1)}
dec_its = int(sys.argv[7]) if len(sys.argv) > 7 else 10        # decode iterations of the probe (tools/gen_flat.py 6 64 8: two calls x 5 iterations)
dec_kernels = {}
for k in set(DF) | set(DW):
    if not (k.startswith(DECODE) or is_decode_a4(k)):
        continue                                              # prefill / ViT kernels of the probe
    b = DF.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0 * 2.0 + DW.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
    dec_bytes += b
    dec_kernels[k[:70]] = b / dec_its
alg = line["roofline"].get("algorithmic_bytes_per_launch")
its = dec_its if dec_bytes else None
import bench
res = {"kernel_source_sha16": bench.gemm_source_sha(), "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE} --kernel-include-regex <training GEMM kernels> -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline",
       "hbm_bytes_per_launch": tr_bytes / max(tr_launch, 1), "launches": tr_launch, "algorithmic_bytes_per_launch": alg,
       "traffic_over_algorithmic": (tr_bytes / max(tr_launch, 1)) / alg if alg else None,
       "decode_hbm_bytes_per_iteration": dec_bytes / its if its else None, "decode_iterations": its,
       "decode_command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE} -- python3 tools/gen_flat.py 6 64 8   (from the repo root; 512 live rows, 1102-token prompts, 2 x 5 decode iterations)",
       "decode_top_kernels": dict(sorted(dec_kernels.items(), key=lambda kv: -kv[1])[:6]),
       "top_kernels": dict(sorted(per_kernel.items(), key=lambda kv: -kv[1]["launches"] * kv[1]["hbm_bytes_per_launch"])[:6])}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:600])
