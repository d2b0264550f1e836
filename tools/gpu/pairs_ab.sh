#!/bin/bash
# GPU box helper: CS16 with the pair image (-DADSB_I16_PAIRS=1; variants p6 = 74 VGPRs / six workgroups per CU, p7 = look-ahead 28,
# 72 VGPRs / seven) against the plain u16 image: parity of each variant, then alternating bench runs at 1 GiB and 16 GiB.
set -o pipefail
mkdir -p gpurun_out
for v in ${VARIANTS:-p6 p7}; do
  ADSB_HIP_LIB=$PWD/air_rs_amd/lib/variants/libadsb_hip_$v.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round4.py tests/test_gpu_streaming.py -m gpu -x -q > gpurun_out/pairs_parity_$v.txt 2>&1 || { tail -30 gpurun_out/pairs_parity_$v.txt; exit 1; }
  echo $v $(tail -1 gpurun_out/pairs_parity_$v.txt)
done
BENCH_ARGS="--no-feed --sample-type i16" tools/gpu/ab.sh pairs_cs16.txt default ${VARIANTS:-p6 p7} || exit 1
REPS=1 BENCH_ARGS="--no-feed --samples 8589934592 --steps 10 --sample-type i16" tools/gpu/ab.sh pairs_cs16_16g.txt default ${VARIANTS:-p6 p7} || exit 1
