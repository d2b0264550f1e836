#!/bin/bash
# GPU box helper: compare library build variants (tools/build_variant.sh) on kernel time; parity first.
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/variants.txt
for v in "$@"; do
  lib=$PWD/air_rs_amd/lib/variants/libadsb_hip_$v.so
  [ "$v" = "default" ] && lib=$PWD/air_rs_amd/lib/libadsb_hip.so
  if [ "${SKIP_TESTS:-0}" != "1" ]; then
    ADSB_HIP_LIB=$lib python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 | tr '\n' ' ' >> gpurun_out/variants.txt
  fi
  ADSB_HIP_LIB=$lib python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', 'kernel_ms', r['kernel_ms'], 'order_ms', r['order_pass_ms'], 'ms_per_step', d['ms_per_step'], 'GB/s', r['achieved'], 'frames', d['config']['frames_per_step'])" >> gpurun_out/variants.txt
  tail -1 gpurun_out/variants.txt
done
