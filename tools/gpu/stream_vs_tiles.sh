#!/bin/bash
# GPU box helper: parity of both i8 kernels, then kernel time of each (bench.py, 1 GiB).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/parity.log; rc=$?
cat gpurun_out/parity.log
[ $rc -ne 0 ] && exit $rc
: > gpurun_out/kernels.txt
for k in ${KERNELS:-stream tiles stream tiles}; do
  ADSB_KERNEL=$k timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>gpurun_out/bench_$k.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$k', 'kernel_ms', r['kernel_ms'], 'order_ms', r['order_pass_ms'], 'ms_per_step', d['ms_per_step'], 'GB/s', r['achieved'], 'frames', d['config']['frames_per_step'])" >> gpurun_out/kernels.txt || exit 1
  tail -1 gpurun_out/kernels.txt
done
