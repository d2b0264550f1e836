#!/bin/bash
# GPU box helper: VALU counters of demod_tiles<i8> under each forced magnitude mode (one PMC pass each)
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcm
for m in "$@"; do
  rm -rf gpurun_out/pmcm/m$m
  ADSB_FORCE_MAG_MODE=$m timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmcm/m$m -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pmcm/m$m.json 2> gpurun_out/pmcm/m$m.err || { tail -5 gpurun_out/pmcm/m$m.err; exit 1; }
  python3 - "$m" <<'PY'
import csv, glob, collections, sys
m = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/pmcm/m{m}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "demod_tiles" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("mode", m, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
done
