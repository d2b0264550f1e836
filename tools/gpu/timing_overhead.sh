#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do
ADSB_BENCH_NO_TIMING=1 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('no-timing ms_per_step', d['ms_per_step'])"
python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('timing    ms_per_step', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['order_pass_ms'])"
done
