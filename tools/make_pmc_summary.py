#!/usr/bin/env python3
"""Builds profiles/pmc_summary.json from the PMC passes tools/gpu/round_profiles.sh left in gpurun_out/
(r_pmc_summary_i8.json, r_pmc_summary_i16.json) and copies the round's bench lines / kernel stats into profiles/."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def derived(d, samples, bps):
    w = d["SQ_WAVES"]
    return {
        "valu_instructions_per_wave": round(d["SQ_INSTS_VALU"] / w, 1),
        "valu_active_quad_cycles_per_wave": round(d["SQ_ACTIVE_INST_VALU"] / w, 1),
        "valu_lane_slots_per_sample": round(d["SQ_ACTIVE_INST_VALU"] * 64 / samples, 2),
        "wave_life_quad_cycles": round(d["SQ_WAVE_CYCLES"] / w, 1),
        "wait_any_share_of_wave_life": round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3),
        "wait_inst_any_share_of_wave_life": round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 3),
        "valu_active_over_busy_cu_cycles": round(d["SQ_ACTIVE_INST_VALU"] / d["SQ_BUSY_CU_CYCLES"], 3),
        "lds_bank_conflict_share_of_lds_active": round(d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1), 3),
        "hbm_bytes_per_launch_fetch_size_x2": int(d["FETCH_SIZE"] * 1024 * 2),
        "algorithmic_bytes_per_launch": samples * bps,
        # the clock the chip held in the counter pass: GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS)
        "kernel_ms_in_counter_pass": (round(d["KERNEL_NS_IN_GRBM_PASS"] * 1e-6, 4) if d.get("KERNEL_NS_IN_GRBM_PASS") else None),
        "effective_clock_ghz": (round(d["GRBM_GUI_ACTIVE"] / 8.0 / d["KERNEL_NS_IN_GRBM_PASS"], 3) if d.get("KERNEL_NS_IN_GRBM_PASS") else None),
        # share of the SIMDs' VALU issue capacity (one wave64 instruction per 4 cycles per SIMD, 1024 SIMDs) the kernel used
        "valu_issue_utilisation": (round(d["SQ_ACTIVE_INST_VALU"] * 4.0 / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 3) if d.get("GRBM_GUI_ACTIVE") else None),
    }


i8 = json.load(open(os.path.join(G, "r_pmc_summary_i8.json")))
i16 = json.load(open(os.path.join(G, "r_pmc_summary_i16.json")))
nsq_path = os.path.join(G, "r_pmc_summary_nsq.json")
nsq = json.load(open(nsq_path)) if os.path.exists(nsq_path) else None
reg_path = os.path.join(G, "r_pmc_summary_reg.json")
reg = json.load(open(reg_path)) if os.path.exists(reg_path) else None
code_path = os.path.join(G, "r_pmc_summary_code.json")
code = json.load(open(code_path)) if os.path.exists(code_path) else None
sieve_path = os.path.join(G, "r_pmc_summary_sieve.json")
sieve = json.load(open(sieve_path)) if os.path.exists(sieve_path) else None
old = json.load(open(os.path.join(P, "pmc_summary.json")))
before = old.get("before_the_split") or {"note": "demod_tiles with the whole decode inside (round 1 .. mid round 2)",
                                         "i8": old.get("i8", {}).get("demod_tiles", {}).get("derived"),
                                         "i8_fused_pass_only": (old.get("i8", {}).get("demod_tiles_fused_pass") or {}).get("derived"),
                                         "cs16": old.get("cs16", {}).get("demod_tiles", {}).get("derived"),
                                         "round_1": old.get("round_1")}
new = {
    "round": "round 4",
    "note": "rocprofv3 --pmc passes over `bench.py --steps 4 --warmup 1` (tools/gpu/pmc_passes.sh; `--sample-type i16` for CS16), "
            "mean per launch of the named kernel on the 1 GiB workload; FETCH_SIZE is in KiB and is doubled per MI355X_MICROARCH.md "
            "(gfx950 reports half of a wide streaming read); SQ_* cycle counters are in quad-cycles summed over waves, GRBM_GUI_ACTIVE "
            "is summed over the 8 XCDs.  demod_tiles = the scan kernel (magnitude + gate + PPM slice of survivors; i8: the root scan on "
            "16384-offset tiles, 8 workgroups per CU), finish_order = CRC-24 / repair of the survivors + the ordered list.",
    "demod_tiles_hbm_bytes_per_launch": int(i8["demod_tiles"]["FETCH_SIZE"] * 1024 * 2),
    "demod_tiles_i16_hbm_bytes_per_launch": int(i16["demod_tiles"]["FETCH_SIZE"] * 1024 * 2),
    "algorithmic_bytes_per_launch": 1073741824,
    "i8": {"demod_tiles": {"derived": derived(i8["demod_tiles"], 1 << 29, 2), "raw": i8["demod_tiles"]},
           "finish_order": i8.get("finish_order"),
           "read_only_kernel": i8.get("read_only")},
    "cs16": {"demod_tiles": {"derived": derived(i16["demod_tiles"], 1 << 28, 4), "raw": i16["demod_tiles"]},
             "finish_order": i16.get("finish_order")},
    # the round-3 A/B kernel (ADSB_SCAN=nsq: gate on I^2+Q^2, no root per sample, 2 bytes of LDS per sample -> 4 workgroups
    # per CU instead of 8): fewer VALU slots, more waiting
    "i8_nsq_scan": ({"demod_tiles": {"derived": derived(nsq["demod_tiles"], 1 << 29, 2), "raw": nsq["demod_tiles"]}} if nsq else None),
    # the register scan (ADSB_SCAN=reg: the same gate from registers, no LDS image)
    "i8_reg_scan": ({"demod_tiles": {"derived": derived(reg["demod_tiles"], 1 << 29, 2), "raw": reg["demod_tiles"]}} if reg else None),
    # the code scan (round 4, ADSB_SCAN=code in the -DADSB_AB_KERNELS=1 build: the gate on an 8-bit log code of I^2+Q^2, no root per sample)
    "i8_code_scan": ({"demod_tiles": {"derived": derived(code["demod_tiles"], 1 << 29, 2), "raw": code["demod_tiles"]}} if code else None),
    # the sieve scan (round 4, ADSB_SCAN=sieve in the A/B build: two relation bits per sample, the gate's adjacent taps on 64-bit words,
    # candidates decided exactly from a raw image in LDS -- four workgroups per CU; adsb_sieve.inc)
    "i8_sieve_scan": ({"demod_tiles": {"derived": derived(sieve["demod_tiles"], 1 << 29, 2), "raw": sieve["demod_tiles"]}} if sieve else None),
    "before_the_split": before,
    # the scan kernel's PMC rows as it was trimmed after the split (each measured by the same passes, one MI355X box each)
    "demod_tiles_i8_history": [
        {"what": "whole decode inside demod_tiles (round 1 .. mid round 2)", "valu_instructions_per_wave": 1539, "valu_slots_per_wave": 1674},
        {"what": "the split: CRC / repair / ordering in finish_candidates", "valu_instructions_per_wave": 1432, "valu_slots_per_wave": 1567.1},
        {"what": "n_valid masking out of the DF17 block", "valu_instructions_per_wave": 1411.7, "valu_slots_per_wave": 1546.8},
        {"what": "tile loads: sweep constant in the SGPR offset; one register for the DF17 bit constant", "valu_instructions_per_wave": 1386.3, "valu_slots_per_wave": 1521.4},
        {"what": "slicer: SDWA byte compare + add-with-carry per bit", "valu_instructions_per_wave": 1367.4, "valu_slots_per_wave": 1502.5},
        {"what": "round 3: 16384-offset tiles (runs of 32; per wave per tile, i.e. per HALF as many samples as the rows above), 19 KB of LDS, "
                 "8 workgroups per CU", "valu_instructions_per_wave": round(i8["demod_tiles"]["SQ_INSTS_VALU"] / i8["demod_tiles"]["SQ_WAVES"], 1),
         "valu_slots_per_wave": round(i8["demod_tiles"]["SQ_ACTIVE_INST_VALU"] / i8["demod_tiles"]["SQ_WAVES"], 1)},
    ],
    "demod_tiles_cs16_history": old.get("demod_tiles_cs16_history") or [
        {"what": "whole decode inside demod_tiles", "valu_instructions_per_wave": 1155, "valu_slots_per_wave": 1223},
        {"what": "the split", "valu_instructions_per_wave": 1085, "valu_slots_per_wave": 1155},
        {"what": "per-tile f16 gate", "valu_instructions_per_wave": 999.8, "valu_slots_per_wave": 1069.5},
        {"what": "n_valid masking, SGPR load offsets, DF17 constant", "valu_instructions_per_wave": 972.9, "valu_slots_per_wave": 1042.6},
    ],
}
json.dump(new, open(os.path.join(P, "pmc_summary.json"), "w"), indent=1)
for src, dst in (("r_bench.json", "bench.json"), ("r_bench_driver_flags.json", "bench_driver_flags.json"), ("r_bench_nsq.json", "bench_nsq_scan.json"), ("r_bench_reg.json", "bench_reg_scan.json"), ("r_bench_code.json", "bench_code_scan.json"), ("r_bench_sieve.json", "bench_sieve_scan.json"),
                 ("r_feed_bench.txt", "feed_bench.txt"), ("r_bench_cs16.json", "bench_cs16.json"), ("r_bench_16g.json", "bench_16GiB.json"),
                 ("r_bench_cs16_16g.json", "bench_cs16_16GiB.json"), ("r_bench_64ch.json", "bench_64_channels.json"),
                 ("r_kernel_stats.csv", "kernel_stats.csv"), ("prof_bench.json", "bench_under_rocprofv3.json"), ("r_kernel_stats_16g.csv", "kernel_stats_16GiB.csv")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, f"{tag}_{dst}"))
for k in ("i8", "cs16"):
    print(k, json.dumps(new[k]["demod_tiles"]["derived"]))
