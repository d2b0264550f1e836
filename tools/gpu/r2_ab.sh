#!/bin/bash
# GPU box helper (round 2): parity of the default build, then kernel time of library variants, alternating
# usage: tools/gpu/r2_ab.sh VARIANT...   (names under air_rs_amd/lib/variants/, or "default")
set -o pipefail
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/t_all.log
  grep -q "passed" gpurun_out/t_all.log || exit 1
  grep -q "failed" gpurun_out/t_all.log && exit 1
fi
: > gpurun_out/ab.txt
for rep in 1 2 ${REPS:-}; do
  for v in "$@"; do
    lib=$PWD/air_rs_amd/lib/variants/libadsb_hip_$v.so; [ "$v" = "default" ] && lib=$PWD/air_rs_amd/lib/libadsb_hip.so
    ADSB_HIP_LIB_LENIENT=1 ADSB_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', 'kernel_ms', r['kernel_ms'], 'order_ms', r['order_pass_ms'], 'ms_per_step', d['ms_per_step'], 'GB/s', r['achieved'], 'frac', r['frac'], 'frames', d['config']['frames_per_step'])" | tee -a gpurun_out/ab.txt
  done
done
