#!/bin/bash
# GPU run 1 of round 3: full GPU test suite (incl. the production-shape parity tests), vendor-GEMM probe (kernel names + PMC), short bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r03_gputests_1.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r03_gputests_1.log
python tools/hipblaslt_probe.py both > gpurun_out/r03_hbl_probe.log 2>&1; cat gpurun_out/r03_hbl_probe.log
(cd /tmp && PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hbl -o x -- python3 $GRAFT_REPO_ROOT/tools/hipblaslt_probe.py both > /tmp/hbl.log 2>&1)
cp /tmp/hbl/x_kernel_stats.csv gpurun_out/r03_hipblaslt_kernel_stats.csv 2>/dev/null || find /tmp/hbl -name "*stats*" | head
bash tools/pmc_pass.sh hbl_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" tools/hipblaslt_probe.py both
bash tools/pmc_pass.sh hbl_sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA" tools/hipblaslt_probe.py both
bash tools/pmc_pass.sh hbl_fetch "FETCH_SIZE" tools/hipblaslt_probe.py both
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err; tail -c 1500 gpurun_out/r03_bench_a.json
