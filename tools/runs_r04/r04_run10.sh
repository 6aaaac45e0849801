# round 4, call 10: the whole GPU suite on the final kernels, then the PMC traffic passes (training GEMM class + decode iteration)
mkdir -p gpurun_out/r04
python3 -m pytest tests -m gpu -q > gpurun_out/r04/tests_full_a.log 2>&1
tail -6 gpurun_out/r04/tests_full_a.log
bash tools/bench_traffic.sh > gpurun_out/r04/bench_traffic.log 2>&1
tail -3 gpurun_out/r04/bench_traffic.log
cp gpurun_out/r04_gemm_traffic.json gpurun_out/r04/ 2>/dev/null
python3 -c "
import json; d = json.load(open('gpurun_out/r04_gemm_traffic.json')); print({k: d[k] for k in ('hbm_bytes_per_launch', 'algorithmic_bytes_per_launch', 'traffic_over_algorithmic', 'decode_hbm_bytes_per_iteration', 'kernel_source_sha16')}); print(d['decode_top_kernels'])"
