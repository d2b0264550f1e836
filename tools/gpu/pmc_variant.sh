#!/bin/bash
# tools/gpu/pmc_variant.sh LIBNAME... : a few SQ counters for library variants (ADSB_HIP_LIB), demod kernel only
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcv
for v in "$@"; do
  lib=$PWD/air_rs_amd/lib/variants/libadsb_hip_$v.so; [ "$v" = "default" ] && lib=$PWD/air_rs_amd/lib/libadsb_hip.so
  for pass in "a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "b SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
    set -- $pass; name=$1; shift
    rm -rf gpurun_out/pmcv/${v}_$name
    echo "$v $name" >> gpurun_out/pmcv/progress.txt
    ADSB_HIP_LIB=$lib timeout -k 5 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcv/${v}_$name -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmcv/${v}_$name.err || tail -3 gpurun_out/pmcv/${v}_$name.err
  done
done
python3 - <<'PY'
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcv/*/**/*counter_collection.csv", recursive=True):
    v = f.split("/")[2].rsplit("_", 1)[0]
    for row in csv.DictReader(open(f)):
        if "demod_tiles" in row["Kernel_Name"]:
            res[v][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for d in res.values() for c in d})
print("counter".ljust(26), *[v.rjust(16) for v in res])
for c in names:
    print(c.ljust(26), *[f"{sum(res[v][c])/max(len(res[v][c]),1):16.0f}" for v in res])
PY
