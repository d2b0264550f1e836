#!/bin/bash
# GPU box helper: the A/B parity cases (tests/ab_cases.py) under ADSB_SCAN=sieve for a library variant, then timing against root.
# usage: tools/gpu/sieve_cases.sh OUT VARIANT   (VARIANT: a name under air_rs_amd/lib/variants/, e.g. ab)
out=$1; v=$2
lib=$PWD/air_rs_amd/lib/variants/libadsb_hip_$v.so
ADSB_HIP_LIB=$lib ADSB_SCAN=sieve timeout -k 10 600 python -m pytest tests/ab_cases.py -x -q -m gpu -p no:cacheprovider > gpurun_out/${out}_cases.txt 2>&1
tail -4 gpurun_out/${out}_cases.txt
REPS=${REPS:-2} BENCH_ARGS=--no-feed bash tools/gpu/ab.sh ${out}.txt $v:sieve default:root
