#!/bin/bash
# GPU box helper: ablation variants of the streaming kernel (tools/build_variant.sh NAME -DADSB_ABL_NOLOOKUP=1 /
# -DADSB_ABL_NOGATE=1; no parity: results are wrong by design), then PMC passes.
set -o pipefail
mkdir -p gpurun_out
SKIP_TESTS=1 tools/gpu/compare_variants.sh default "$@" || exit 1
cp gpurun_out/variants.txt gpurun_out/ablate.txt
tools/gpu/pmc_passes.sh > gpurun_out/pmc_out.txt 2>&1; rc=$?
tail -60 gpurun_out/pmc_out.txt
exit $rc
