# round 4, call 13: the row loop of the 257..512-row decode tiles — parity, iteration time A/B, decode traffic (PMC); + the two new e2e tests
mkdir -p gpurun_out/r04
python3 -m pytest tests/test_gpu_production_shapes.py tests/test_gpu_kernels.py tests/test_gpu_rollout.py "tests/test_gpu_e2e.py::test_bench_two_ranks_on_one_gpu_runs_the_real_multi_rank_step" "tests/test_gpu_e2e.py::test_grpo_loop_learns_a_dense_synthetic_reward" -q -s > gpurun_out/r04/tests_run13.log 2>&1
grep -n "share of sampled\|passed\|failed\|^E " gpurun_out/r04/tests_run13.log | head -12 | cut -c1-400
for m in 1 0; do
  echo "== ST_DECODE_ROWLOOP=$m"
  ST_DECODE_ROWLOOP=$m python3 tools/gen_flat.py 200 64 8 2>&1 | grep "^rows" | tail -1
  ST_DECODE_ROWLOOP=$m python3 tools/gen_flat.py 200 44 8 2>&1 | grep "^rows" | tail -1
done
python3 tools/decode512_probe.py 512 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_dec_$c
  rm -rf $d
  timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o x -- python3 tools/gen_flat.py 6 64 8 > gpurun_out/r04/dec_traffic_$c.log 2>&1 || tail -5 gpurun_out/r04/dec_traffic_$c.log
  python3 tools/pmc_summarize.py $d gpurun_out/r04/pmc_dec_rowloop_$c.json
  rm -rf $d
done
python3 - <<'PY'
import json
F = json.load(open("gpurun_out/r04/pmc_dec_rowloop_FETCH_SIZE.json")); W = json.load(open("gpurun_out/r04/pmc_dec_rowloop_WRITE_SIZE.json"))
DECODE = ("gemm_tile_kernel<256, 160", "gemm_tile_kernel<256, 128", "gemm_tile_kernel<128,", "gemm_tile_kernel<64,", "attn_fwd128_kernel<false>", "attn_decode128", "attn_merge_kernel", "decode_finish", "decode_step_kernel", "sample_kernel", "sample_filter_kernel", "gemm_skinny_finish", "gemm_tile_kernel<256, 256")
tot = 0.0; rows = []
for k in set(F) | set(W):
    if not k.startswith(DECODE): continue
    b = F.get(k, {}).get("FETCH_SIZE", 0.0) * 2048.0 + W.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
    n = max(F.get(k, {}).get("dispatches", 0), W.get(k, {}).get("dispatches", 0))
    tot += b; rows.append((b / 10 / 1e9, n, k[:80]))
rows.sort(reverse=True)
print(f"decode HBM traffic per 512-row iteration: {tot / 10 / 1e9:.2f} GB")
for r in rows[:8]: print(f"  {r[0]:6.2f} GB  {r[1]:6d} launches  {r[2]}")
PY
