#!/bin/bash
# GPU box helper: parity tests, bench line, rocprofv3 kernel stats of the same command
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/parity.log; rc=$?
  cat gpurun_out/parity.log
  [ $rc -ne 0 ] && exit $rc
fi
python bench.py --steps ${STEPS:-30} --warmup 3 ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?
tail -5 gpurun_out/bench.err; cat gpurun_out/bench.json
[ $rc -ne 0 ] && exit $rc
if [ "${SKIP_PROF:-0}" != "1" ]; then
  rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
  # (--no-feed: the host-fed measurement behind the timed region launches the same kernels on 32 MiB buffers; they would
  # be averaged into demod_tiles' row of the stats file)
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o stats -- python3 bench.py --steps 20 --warmup 5 --no-feed > gpurun_out/prof_bench.json 2> gpurun_out/prof.err; rc=$?
  tail -3 gpurun_out/prof.err
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
fi
exit $rc
