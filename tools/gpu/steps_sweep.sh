#!/bin/bash
# GPU box helper: bench.py at increasing step counts (the scan's mean duration over a longer and longer run)
set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/steps_sweep.txt
for st in ${SWEEP:-8 20 50 200 1000 3000}; do
  python bench.py --steps $st --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('steps', d['steps'], 'scan_ms', r['kernel_ms'], 'frac', r['frac'], 'finish_ms', r['finish_order_ms'], 'ms_per_step', d['ms_per_step'], 'Msamples/s', d['value'], 'launches_timed', r['launches_timed'])" | tee -a gpurun_out/steps_sweep.txt
done
