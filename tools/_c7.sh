mkdir -p gpurun_out/r06
ST_DECODE_ROWS=1 bash tools/gen_flat_trace.sh 200 64 8 2>&1 | grep -E "^rows|attn|merge" > gpurun_out/r06/c7_trace_rows1_200.txt
ST_DECODE_ROWS=1 bash tools/gen_flat_trace.sh 600 64 8 2>&1 | grep -E "^rows|attn|merge" > gpurun_out/r06/c7_trace_rows1_600.txt
ST_DECODE_ROWS=0 bash tools/gen_flat_trace.sh 600 64 8 2>&1 | grep -E "^rows|attn|merge" > gpurun_out/r06/c7_trace_rows0_600.txt
cat gpurun_out/r06/c7_trace_*.txt
