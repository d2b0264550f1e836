#!/bin/bash
# GPU box helper: tools/gpu/scan_drift.py under rocprofv3 --kernel-trace; prints every scan's duration in launch order
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/drift && mkdir -p gpurun_out/drift
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/drift -o t -- python3 tools/gpu/scan_drift.py > gpurun_out/drift/out.txt 2> gpurun_out/drift/err.txt || { tail -5 gpurun_out/drift/err.txt; exit 1; }
f=$(find gpurun_out/drift -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/scan_drift.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "demod_tiles" in r["Kernel_Name"]]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
t0 = ks[0][0]
burst, prev_end = [], None
out = []
for s, e in ks:
    if prev_end is not None and s - prev_end > 20_000_000:  # > 20 ms of idle: a new burst
        out.append(burst); burst = []
    burst.append(((s - t0) / 1e6, (e - s) / 1e3))
    prev_end = e
out.append(burst)
for i, b in enumerate(out):
    print(f"burst {i}: {len(b)} scans, starts at {b[0][0]:.1f} ms; durations (us):")
    print("   " + " ".join(f"{d:.0f}" for _, d in b))
PY
