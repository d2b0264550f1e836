#!/bin/bash
# GPU box helper: finish_candidates + gather_tiles on their own (high-priority) stream beside the next launch's scan
# (ADSB_OVERLAP_ORDERING=1) against the in-order default, for library variants.  usage: overlap_ab.sh VARIANT...
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/overlap_ab.txt
for rep in 1 2 ${REPS:-}; do
  for v in "$@"; do
    lib=$PWD/air_rs_amd/lib/variants/libadsb_hip_$v.so; [ "$v" = "default" ] && lib=$PWD/air_rs_amd/lib/libadsb_hip.so
    for ov in 0 1; do
      ADSB_OVERLAP_ORDERING=$ov ADSB_HIP_LIB_LENIENT=1 ADSB_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 40 --warmup 4 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v overlap=$ov', 'scan_ms', r['kernel_ms'], 'finish_ms', r['finish_pass_ms'], 'order_ms', r['order_pass_ms'], 'ms_per_step', d['ms_per_step'], 'Msamples/s', d['value'], 'frames', d['config']['frames_per_step'])" | tee -a gpurun_out/overlap_ab.txt
    done
  done
done
