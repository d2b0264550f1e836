#!/bin/bash
# first GPU contact: parity tests, then a quick look at kernel time
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/parity.log; rc=$?
cat gpurun_out/parity.log
exit $rc
