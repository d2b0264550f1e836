#!/bin/bash
# GPU box helper (round 2): typed-load phase 1 against the untyped one -- parity, kernel time, phase ablations
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 tools/gpu/compare_variants.sh default typed32 || exit 1
cp gpurun_out/variants.txt gpurun_out/r2_typed_a.txt
SKIP_TESTS=1 timeout -k 10 600 tools/gpu/compare_variants.sh typed16 typed48 base_p1 typed_p1 base_p2 typed_p2 default typed32 || exit 1
cp gpurun_out/variants.txt gpurun_out/r2_typed_b.txt
