#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_byname_api.py tests/test_gpu_production_shapes.py tests/test_gpu_e2e.py -m gpu -q > gpurun_out/r03_gputests_3.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r03_gputests_3.log
(cd /tmp && PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hbl -o x -- python3 $GRAFT_REPO_ROOT/tools/hipblaslt_probe.py both 28672 > gpurun_out/r03_hbl_probe_28k.log 2>&1)
cat $GRAFT_REPO_ROOT/gpurun_out/r03_hbl_probe_28k.log /tmp/gpurun_out/r03_hbl_probe_28k.log 2>/dev/null
cp /tmp/hbl/x_kernel_stats.csv gpurun_out/r03_hipblaslt_kernel_stats_28k.csv 2>/dev/null || find /tmp/hbl -name "*stats*" | head
bash tools/pmc_pass.sh hbl28_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" tools/hipblaslt_probe.py both 28672
bash tools/pmc_pass.sh hbl28_sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA" tools/hipblaslt_probe.py both 28672
bash tools/pmc_pass.sh hbl28_fetch "FETCH_SIZE" tools/hipblaslt_probe.py both 28672
python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_c.json 2> gpurun_out/r03_bench_c.err; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_c.json'));print('default', d['value'], d['timing_s'], d['peak_mem_gb'], d['peak_reserved_gb'], d['reserved_gb_after_each_step'], d['roofline_decode']['ms_per_iteration'], d['roofline']['frac'])"
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_c_exp.json 2> gpurun_out/r03_bench_c_exp.err; tail -3 gpurun_out/r03_bench_c_exp.err; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_c_exp.json'));print('expandable', d['value'], d['timing_s'], d['peak_mem_gb'], d['peak_reserved_gb'], d['reserved_gb_after_each_step'])"
