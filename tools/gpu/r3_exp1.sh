#!/bin/bash
# round 3, experiment 1: why is the nsq scan slower than the root scan at 8 workgroups per CU?
set -o pipefail
cd "$(dirname "$0")/../.."
V=air_rs_amd/lib/variants
for v in t128 pad5 t128pad5; do
  echo "== parity $v" | tee -a gpurun_out/r3_exp1.log
  ADSB_HIP_LIB=$PWD/$V/libadsb_hip_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nsq and (synthetic_sizes_i8 or ties or gate or full_scale or constant or truncation or multichannel or random)" 2>&1 | tail -2 | tee -a gpurun_out/r3_exp1.log
done
tools/gpu/ab.sh r3_ab1.txt default default:root t128 pad5 t128pad5 aux0 p1 p2 p1:root p2:root
PMC_OUT=gpurun_out/pmc_nsq tools/gpu/pmc_passes.sh > gpurun_out/r3_pmc_nsq.txt 2>&1
ADSB_SCAN=root PMC_OUT=gpurun_out/pmc_root tools/gpu/pmc_passes.sh > gpurun_out/r3_pmc_root.txt 2>&1
ADSB_HIP_LIB=$PWD/$V/libadsb_hip_t128.so PMC_OUT=gpurun_out/pmc_t128 tools/gpu/pmc_passes.sh > gpurun_out/r3_pmc_t128.txt 2>&1
tail -40 gpurun_out/r3_pmc_nsq.txt
