#!/bin/bash
# GPU box helper: bench line + rocprofv3 kernel stats of the same command (summaries -> gpurun_out/)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py --steps ${STEPS:-30} --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?
tail -5 gpurun_out/bench.err; cat gpurun_out/bench.json
[ $rc -ne 0 ] && exit $rc
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof.err; rc=$?
tail -3 gpurun_out/prof.err; cat gpurun_out/prof_bench.json
find gpurun_out/prof -name "*stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
exit $rc
