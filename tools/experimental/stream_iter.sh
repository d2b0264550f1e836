#!/bin/bash
# GPU box helper for iterating on the streaming kernel: parity (both kernels), stamps of a -DADSB_STAMPS=1
# build, then kernel time of the default library and of the named variants (stream kernel unless the
# variant name starts with "t", which is timed with ADSB_KERNEL=tiles).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/parity.log; rc=$?
cat gpurun_out/parity.log
[ $rc -ne 0 ] && exit $rc
if [ -f air_rs_amd/lib/variants/libadsb_hip_stamps.so ]; then
  ADSB_KERNEL=stream ADSB_HIP_LIB=$PWD/air_rs_amd/lib/variants/libadsb_hip_stamps.so timeout -k 10 200 python tools/gpu/stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/stamps.txt
fi
: > gpurun_out/kernels.txt
for v in default "$@" default; do
  lib=$PWD/air_rs_amd/lib/variants/libadsb_hip_$v.so
  [ "$v" = "default" ] && lib=$PWD/air_rs_amd/lib/libadsb_hip.so
  [ "$v" = "tdefault" ] && lib=$PWD/air_rs_amd/lib/libadsb_hip.so
  k=stream; case $v in t*) k=tiles;; esac
  ADSB_KERNEL=$k ADSB_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>gpurun_out/bench_$v.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', '$k', 'kernel_ms', r['kernel_ms'], 'order_ms', r['order_pass_ms'], 'ms_per_step', d['ms_per_step'], 'GB/s', r['achieved'], 'frames', d['config']['frames_per_step'])" >> gpurun_out/kernels.txt || { tail -5 gpurun_out/bench_$v.err; exit 1; }
  tail -1 gpurun_out/kernels.txt
done
