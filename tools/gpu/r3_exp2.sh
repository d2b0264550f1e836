#!/bin/bash
# round 3, experiment 2: several tiles per workgroup with the next tile's loads in flight (nsq scan)
set -o pipefail
cd "$(dirname "$0")/../.."
V=air_rs_amd/lib/variants
: > gpurun_out/r3_exp2.log
for v in tpw4 tpw4s; do
  echo "== parity $v" | tee -a gpurun_out/r3_exp2.log
  ADSB_HIP_LIB=$PWD/$V/libadsb_hip_$v.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nsq and not large_streaming" 2>&1 | tail -3 | tee -a gpurun_out/r3_exp2.log
done
tools/gpu/ab.sh r3_ab2.txt default default:root tpw4 tpw4e2 tpw2 tpw8 tpw4s
echo "== new tests (default build)" | tee -a gpurun_out/r3_exp2.log
timeout -k 10 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_round2.py tests/test_gpu_streaming.py tests/test_tracker.py tests/test_golden.py -x -q -m gpu -k "not 16GiB" 2>&1 | tail -15 | tee -a gpurun_out/r3_exp2.log
