#!/bin/bash
# GPU box helper: PMC passes (each its own run, no tracing flags besides kernel-trace) for bench.py
set -o pipefail
export TMPDIR=/tmp
export ADSB_BENCH_SETTLE_S=0   # counters per launch do not depend on the clock state: keep the passes short
export PMC_OUT=${PMC_OUT:-gpurun_out/pmc}
PMC=$PMC_OUT
mkdir -p $PMC
run_pass() {
  name=$1; shift
  rm -rf $PMC/$name
  echo "pass $name" >> $PMC/progress.txt
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $PMC/$name -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-feed ${BENCH_ARGS:-} > $PMC/$name.json 2> $PMC/$name.err || { tail -5 $PMC/$name.err; return 1; }
}
run_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU || exit 1
run_pass sq2 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAVES || exit 1
run_pass sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_LDS_UNALIGNED_STALL || exit 1
run_pass tcc1 FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
python3 - <<'PY'
import csv, glob, collections, json, os
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.environ.get("PMC_OUT","gpurun_out/pmc") + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "demod_tiles" in k: k = "demod_tiles"
        elif "finish_order" in k: k = "finish_order"
        elif "read_only" in k: k = "read_only"
        else: continue
        res[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":  # the dispatch's own duration in that pass: effective clock = GRBM_GUI_ACTIVE / 8 / duration
            res[k]["KERNEL_NS_IN_GRBM_PASS"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
json.dump(out, open(os.environ.get("PMC_OUT","gpurun_out/pmc") + "/summary.json", "w"), indent=1)
for k, d in out.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:28s} {v:16.1f}")
PY
