#!/bin/bash
# whole GPU suite on the final sources (incl. the critic and fp8 additions), smoke(), the short default bench line (cpu_baseline after the page-fault fix)
mkdir -p gpurun_out/r04
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -4
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python3 bench.py > gpurun_out/r04/bench_h.json 2> gpurun_out/r04/bench_h.err
python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_h.json').read().strip().splitlines()[-1])
c = d['cpu_baseline']
print(d['value'], d['timing_s'], d['roofline']['frac'], 'cpu', c['value'], c['value_excl_generation_and_adamw'], c['cores'], c['measured_s']['decode_step_4_layers'])"
