#!/bin/bash
# GPU box helper: the round's measured artefacts in one call: parity, default bench line (+cpu baseline),
# rocprofv3 kernel stats of the same command, PMC passes (i8 and CS16), the 16 GiB / CS16 / 64-channel configurations.
# Everything lands under gpurun_out/ (r_* files); copy what is to be judged into profiles/.
set -o pipefail
mkdir -p gpurun_out
PART=${PART:-123}
if [[ $PART == *1* ]]; then
tools/gpu/bench_and_profile.sh || exit 1
cp gpurun_out/bench.json gpurun_out/r_bench.json
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r_kernel_stats.csv
tools/gpu/pmc_passes.sh > gpurun_out/pmc_out.txt 2>&1 || { tail -20 gpurun_out/pmc_out.txt; exit 1; }
cp gpurun_out/pmc/summary.json gpurun_out/r_pmc_summary_i8.json
BENCH_ARGS="--sample-type i16" tools/gpu/pmc_passes.sh > gpurun_out/pmc_out_i16.txt 2>&1 || { tail -20 gpurun_out/pmc_out_i16.txt; exit 1; }
cp gpurun_out/pmc/summary.json gpurun_out/r_pmc_summary_i16.json
fi
if [[ $PART == *2* ]]; then
# the A/B scan kernels live in the -DADSB_AB_KERNELS=1 build of the library
AB=$PWD/air_rs_amd/lib/variants/libadsb_hip_ab.so
for k in code nsq reg sieve; do
  ADSB_HIP_LIB=$AB BENCH_ARGS="--scan $k --no-feed" tools/gpu/pmc_passes.sh > gpurun_out/pmc_out_$k.txt 2>&1 || { tail -20 gpurun_out/pmc_out_$k.txt; exit 1; }
  cp gpurun_out/pmc/summary.json gpurun_out/r_pmc_summary_$k.json
  ADSB_HIP_LIB=$AB python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-feed --scan $k > gpurun_out/r_bench_$k.json 2>gpurun_out/r_bench_$k.err || exit 1
done
fi
if [[ $PART == *3* ]]; then
python bench.py --steps 20 --warmup 5 > gpurun_out/r_bench_driver_flags.json 2>gpurun_out/r_bench_driver_flags.err || exit 1
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-feed --sample-type i16 > gpurun_out/r_bench_cs16.json 2>gpurun_out/r_bench_cs16.err || exit 1
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-feed --samples 8589934592 > gpurun_out/r_bench_16g.json 2>gpurun_out/r_bench_16g.err || exit 1
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-feed --samples 8589934592 --sample-type i16 > gpurun_out/r_bench_cs16_16g.json 2>gpurun_out/r_bench_cs16_16g.err || exit 1
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-feed --channels 64 > gpurun_out/r_bench_64ch.json 2>gpurun_out/r_bench_64ch.err || exit 1
rm -rf gpurun_out/prof16 && mkdir -p gpurun_out/prof16
TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof16 -o stats -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-feed --samples 8589934592 > gpurun_out/prof16_bench.json 2> gpurun_out/prof16.err
f=$(find gpurun_out/prof16 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r_kernel_stats_16g.csv
[ -x tools/bench/feed_bench ] && timeout -k 10 300 tools/bench/feed_bench > gpurun_out/r_feed_bench.txt 2>&1
fi
for f in r_bench r_bench_driver_flags r_bench_code r_bench_nsq r_bench_reg r_bench_sieve r_bench_cs16 r_bench_16g r_bench_cs16_16g r_bench_64ch; do [ -s gpurun_out/$f.json ] && python3 -c "
import json; d=json.load(open('gpurun_out/$f.json')); r=d['roofline']; fp=r.get('fused_pass',{})
print('$f', r['kernel'], 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'GB/s', r['achieved'], 'frac', r['frac'], 'ceil', r['read_ceiling_gbps'], 'finish_ms', r.get('finish_order_ms'))"; done
