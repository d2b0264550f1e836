#!/bin/bash
# GPU box helper: the XCD-contiguous tile mapping (-DADSB_XCD_MAP=1, variant "xcd") against tile = workgroup index: parity of the
# variant, then alternating bench runs (i8 and CS16, 1 GiB and 16 GiB) and one FETCH_SIZE pass each.
set -o pipefail
mkdir -p gpurun_out
V=$PWD/air_rs_amd/lib/variants/libadsb_hip_xcd.so
ADSB_HIP_LIB=$V timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/xcd_parity.txt 2>&1 || { tail -20 gpurun_out/xcd_parity.txt; exit 1; }
tail -1 gpurun_out/xcd_parity.txt
BENCH_ARGS="--no-feed" tools/gpu/ab.sh xcd_i8.txt default xcd || exit 1
BENCH_ARGS="--no-feed --sample-type i16" tools/gpu/ab.sh xcd_cs16.txt default xcd || exit 1
REPS=1 BENCH_ARGS="--no-feed --samples 8589934592 --steps 10" tools/gpu/ab.sh xcd_i8_16g.txt default xcd || exit 1
REPS=1 BENCH_ARGS="--no-feed --samples 8589934592 --steps 10 --sample-type i16" tools/gpu/ab.sh xcd_cs16_16g.txt default xcd || exit 1
export TMPDIR=/tmp ADSB_BENCH_SETTLE_S=0
for v in default xcd; do for st in i8 i16; do
  lib=$V; [ $v = default ] && lib=$PWD/air_rs_amd/lib/libadsb_hip.so
  d=gpurun_out/xcd_pmc/$v$st; rm -rf $d; mkdir -p $d
  ADSB_HIP_LIB=$lib timeout -k 5 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $d -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-feed --sample-type $st > $d/out.json 2> $d/err.txt || { tail -3 $d/err.txt; exit 1; }
  python3 - $v $st $d <<'PY' | tee -a gpurun_out/xcd_fetch.txt
import csv, glob, sys
v, st, d = sys.argv[1:4]
vals = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "demod_tiles" in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
            vals.append(float(row["Counter_Value"]))
print(v, st, "FETCH_SIZE mean", sum(vals) / max(1, len(vals)), "launches", len(vals), "bytes (x 2 x 1024)", 2048 * sum(vals) / max(1, len(vals)))
PY
done; done
