#!/bin/bash
# GPU box helper: the round's measured artefacts in one call: parity, default bench line (+cpu baseline),
# rocprofv3 kernel stats, PMC passes, the streaming kernel's bench line and the 16 GiB configuration.
set -o pipefail
mkdir -p gpurun_out
tools/gpu/bench_and_profile.sh || exit 1
cp gpurun_out/bench.json gpurun_out/r_bench.json
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r_kernel_stats.csv
tools/gpu/pmc_passes.sh > gpurun_out/pmc_out.txt 2>&1 || { tail -20 gpurun_out/pmc_out.txt; exit 1; }
cp gpurun_out/pmc/summary.json gpurun_out/r_pmc_summary.json
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --kernel stream > gpurun_out/r_bench_stream.json 2>gpurun_out/r_bench_stream.err || exit 1
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --samples 8589934592 > gpurun_out/r_bench_16g.json 2>gpurun_out/r_bench_16g.err || exit 1
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --samples 8589934592 --kernel stream > gpurun_out/r_bench_16g_stream.json 2>gpurun_out/r_bench_16g_stream.err || exit 1
for f in r_bench r_bench_stream r_bench_16g r_bench_16g_stream; do python3 -c "
import json; d=json.load(open('gpurun_out/$f.json')); r=d['roofline']; print('$f', r['kernel'], 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'GB/s', r['achieved'], 'frac', r['frac'], 'ceil', r['read_ceiling_gbps'])"; done
