#!/bin/bash
# GPU box helper: kernel start/end timeline (rocprofv3 --kernel-trace) of a short bench run with the finish + gather
# kernels on their own stream; prints the last launches' kernels relative to the first of them.
# usage: overlap_trace.sh VARIANT [0|1]
set -o pipefail
v=${1:-default}; ov=${2:-1}
lib=$PWD/air_rs_amd/lib/variants/libadsb_hip_$v.so; [ "$v" = "default" ] && lib=$PWD/air_rs_amd/lib/libadsb_hip.so
export ADSB_OVERLAP_ORDERING=$ov ADSB_HIP_LIB_LENIENT=1 ADSB_HIP_LIB=$lib ADSB_BENCH_NO_TIMING=1 TMPDIR=/tmp
rm -rf gpurun_out/ovtrace && mkdir -p gpurun_out/ovtrace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovtrace -o t -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline > gpurun_out/ovtrace/bench.json 2> gpurun_out/ovtrace/err.txt || { tail -5 gpurun_out/ovtrace/err.txt; exit 1; }
f=$(find gpurun_out/ovtrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/overlap_trace_${v}_${ov}.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "?")) for r in rows
      if any(x in r["Kernel_Name"] for x in ("demod_tiles", "finish_candidates", "gather_tiles"))]
ks.sort()
ks = ks[-18:]
t0 = ks[0][0]
for s, e, n, q in ks:
    print(f"{(s - t0) / 1e3:9.1f} us -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f} us)  queue {q}  {n}")
PY
