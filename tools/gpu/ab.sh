#!/bin/bash
# GPU box helper: kernel time of library variants / scan kernels, alternating.
# usage: tools/gpu/ab.sh OUTFILE SPEC...   SPEC = VARIANT[:SCAN]  (VARIANT: a name under air_rs_amd/lib/variants/, or "default";
#                                                                   SCAN: root | nsq, default root)
# env: REPS (default 2), BENCH_ARGS
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/$1; shift
: > $out
for rep in $(seq 1 ${REPS:-2}); do
  for spec in "$@"; do
    v=${spec%%:*}; scan=root; [[ "$spec" == *:* ]] && scan=${spec##*:}
    lib=$PWD/air_rs_amd/lib/variants/libadsb_hip_$v.so; [ "$v" = "default" ] && lib=$PWD/air_rs_amd/lib/libadsb_hip.so
    ADSB_SCAN=$scan ADSB_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['cold_start']
print('$spec', 'scan_ms', r['kernel_ms'], 'finish_ms', r['finish_order_ms'], 'ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'cold_scan_ms', c['kernel_ms'], 'frames', d['config']['frames_per_step'], 'ceil', r['read_ceiling_gbps'])" | tee -a $out
  done
done
