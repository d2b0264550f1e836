#!/bin/bash
# GPU box helper: stamps of several diagnostic builds (variant names given as arguments)
mkdir -p gpurun_out; : > gpurun_out/stamps_multi.txt
for v in "$@"; do
  echo "== $v" >> gpurun_out/stamps_multi.txt
  ADSB_HIP_LIB=$PWD/air_rs_amd/lib/variants/libadsb_hip_$v.so timeout -k 10 200 python tools/gpu/stamps.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/stamps_multi.txt || exit 1
done
cat gpurun_out/stamps_multi.txt
