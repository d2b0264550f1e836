#!/usr/bin/env python3
"""ISA-level issue-slot count of a gfx950 kernel, per source phase.

Compiles the kernel source with -gline-tables-only (line tables only: the code is the same as the
product build's), walks one kernel's assembly, attributes every instruction to the innermost source line
its `.loc` names and sums issue slots per phase (phases = line ranges of the source, read from
`// [phase:NAME]` ... `// [phase:end]` markers in the .hip/.h files).

Issue-slot model (tools/ubench/valu_rates*.hip): one wave64 VALU instruction holds
its SIMD for 4 cycles = 1 slot; v_sqrt_f32 and the other transcendentals 8 cycles = 2 slots.  SALU, LDS,
VMEM and branch instructions issue on other ports and are counted separately.

Blocks the hardware executes conditionally are listed with their static counts; `--cold LINES` names
source lines whose instructions are on a rarely taken path (kept out of the per-tile totals).

usage: tools/isa_slots.py SRC.hip KERNEL_MANGLED_NAME [-D...] [--out FILE]
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

TRANS = ("v_sqrt_f32", "v_rsq_f32", "v_rcp_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32",
         "v_cvt_pk_fp8_f32", "v_cvt_pk_bf8_f32")  # (the fp8 conversions issue at the quarter rate too: profiles/r04_fp8_probe.txt)


def phase_ranges(paths):
    """{file basename: [(first, last, name)]} from // [phase:NAME] ... // [phase:end] markers"""
    out = {}
    for p in paths:
        cur, start, rs = None, 0, []
        with open(p) as f:
            for i, line in enumerate(f, 1):
                m = re.search(r"\[phase:([^\]]+)\]", line)
                if not m:
                    continue
                if cur is not None:
                    rs.append((start, i, cur))
                cur = None if m.group(1) == "end" else m.group(1)
                start = i
        out[os.path.basename(p)] = rs
    return out


def classify(op):
    if op.startswith("v_"):
        if op.startswith(TRANS):
            return "valu_trans"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu_lane"
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "valu_cmp"
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("kernel")
    ap.add_argument("--hipcc", default=os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--out")
    ap.add_argument("--weight", action="append", default=[], metavar="PHASE=W",
                    help="how often a wave executes that phase's instructions per tile (default 1): gives the "
                         "'per tile' column, to be checked against SQ_INSTS_VALU per wave")
    ap.add_argument("--note", action="append", default=[], help="free text appended to the report")
    a = ap.parse_args()

    srcdir = os.path.dirname(os.path.abspath(a.src))
    markers = phase_ranges([os.path.join(srcdir, f) for f in os.listdir(srcdir) if f.endswith((".hip", ".h"))])

    with tempfile.TemporaryDirectory() as td:
        s_path = os.path.join(td, "k.s")
        cmd = [a.hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-gline-tables-only", "-S",
               "--cuda-device-only", os.path.abspath(a.src), "-o", s_path] + ["-D" + d for d in a.D]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        lines = open(s_path).read().split("\n")

    start = next(i for i, ln in enumerate(lines) if ln.startswith(a.kernel + ":"))
    counts = collections.defaultdict(lambda: collections.Counter())
    ops_by_phase = collections.defaultdict(lambda: collections.Counter())
    cur = ("?", 0)
    n_total = 0
    for ln in lines[start + 1:]:
        s = ln.strip()
        if s.startswith(".loc"):
            # innermost location first, then the inlined-at chain: take the first one inside the source directory
            for path, line in re.findall(r"([^\s:\[;]+):(\d+):\d+", s):
                if os.path.dirname(os.path.abspath(path)) == srcdir and int(line) > 0:
                    cur = (os.path.basename(path), int(line))
                    break
            continue
        if s.startswith(".Lfunc_end"):
            break
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        kind = classify(op)
        phase = "other"
        for (lo, hi, name) in markers.get(cur[0], []):
            if lo <= cur[1] <= hi:
                phase = name
                break
        counts[phase][kind] += 1
        ops_by_phase[phase][op] += 1
        n_total += 1

    weights = {}
    for kv in a.weight:
        k, v = kv.rsplit("=", 1)
        weights[k] = float(v)
    out = []
    w = out.append
    w(f"# ISA issue-slot count: {a.kernel}")
    w(f"# source {os.path.relpath(a.src)}  defines {a.D}  (static instruction counts from hipcc -S, gfx950)")
    w("# slots = VALU instructions x 1 + transcendentals x 2 (4-cycle issue units per wave64 instruction)")
    w(f"{'phase':38s} {'valu':>6s} {'cmp':>5s} {'trans':>6s} {'lane':>5s} {'SLOTS':>7s} | {'salu':>5s} {'lds':>5s} {'vmem':>5s} {'branch':>6s} {'wait':>5s} | {'x/tile':>6s} {'instr/tile':>10s} {'slots/tile':>10s}")
    tot = collections.Counter()
    dyn_i = dyn_s = 0.0
    for phase in sorted(counts):
        c = counts[phase]
        n_valu = c["valu"] + c["valu_cmp"] + c["valu_lane"] + c["valu_trans"]
        slots = n_valu + c["valu_trans"]
        wt = weights.get(phase, 1.0)
        dyn_i += wt * n_valu
        dyn_s += wt * slots
        w(f"{phase:38s} {c['valu']:6d} {c['valu_cmp']:5d} {c['valu_trans']:6d} {c['valu_lane']:5d} {slots:7d} | "
          f"{c['salu']:5d} {c['lds']:5d} {c['vmem']:5d} {c['branch']:6d} {c['wait']:5d} | {wt:6.2f} {wt * n_valu:10.0f} {wt * slots:10.0f}")
        tot.update(c)
        tot["slots"] += slots
    w(f"{'TOTAL':38s} {tot['valu']:6d} {tot['valu_cmp']:5d} {tot['valu_trans']:6d} {tot['valu_lane']:5d} {tot['slots']:7d} | "
      f"{tot['salu']:5d} {tot['lds']:5d} {tot['vmem']:5d} {tot['branch']:6d} {tot['wait']:5d} | {'':6s} {dyn_i:10.0f} {dyn_s:10.0f}")
    for n in a.note:
        w("# " + n)
    w("")
    for phase in sorted(ops_by_phase):
        top = ", ".join(f"{op} {n}" for op, n in ops_by_phase[phase].most_common(14))
        w(f"[{phase}] {top}")
    text = "\n".join(out) + "\n"
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
