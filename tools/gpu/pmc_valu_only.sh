#!/bin/bash
# GPU box helper: one PMC pass (VALU / wave-cycle counters) per library variant.  usage: tools/gpu/pmc_valu_only.sh OUT VARIANT[:SCAN]...
set -o pipefail
export TMPDIR=/tmp ADSB_BENCH_SETTLE_S=0
out=gpurun_out/$1; shift
: > $out
for spec in "$@"; do
  v=${spec%%:*}; scan=code; [[ "$spec" == *:* ]] && scan=${spec##*:}
  lib=$PWD/air_rs_amd/lib/variants/libadsb_hip_$v.so; [ "$v" = "default" ] && lib=$PWD/air_rs_amd/lib/libadsb_hip.so
  d=gpurun_out/pmc_v/$v; rm -rf $d; mkdir -p $d
  ADSB_SCAN=$scan ADSB_HIP_LIB=$lib timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $d -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-feed > $d/out.json 2> $d/err.txt || { tail -3 $d/err.txt; continue; }
  python3 - "$spec" $d >> $out <<'PY'
import csv, glob, sys, collections
spec, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "demod_tiles" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m.get("SQ_WAVES", 1)
print(spec, " ".join(f"{k}/wave={m[k] / w:.1f}" for k in sorted(m) if k != "SQ_WAVES"), f"waves={w:.0f}")
PY
done
cat $out
