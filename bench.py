#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s (+ decoded Mode-S msgs/s) of the HIP demodulation path on MI355X.

Workload (BASELINE.json configs[1]): 2 MSPS-format i8 IQ, a 1 GiB synthetic buffer per GPU, resident
in HBM before the timed region; one step = one pass of the fused magnitude + preamble/DF17 gate + PPM
slice (the scan kernel) + CRC-24 / repair + ordered frame list (the finishing kernel) over that buffer.  With N > 1 ranks the stream is time-sharded: rank
g owns offsets [g*(n-240), (g+1)*(n-240)) of one long stream and generates its own slice plus the
240-sample read halo (no input exchange); frames carry absolute stream offsets (adsb_set_stream_base); every
launch writes its ordered frame list into a slot of an 8-launch bucket and one RCCL gather per bucket moves
the lists to rank 0 from a side stream, overlapped with the next bucket's kernels (air_rs_amd/sharding.py
BucketGather -- the class tests/test_sharding_gloo.py drives over gloo).  After the timed
region rank 0 checks the last launch's gathered lists: globally ordered once concatenated in rank order, and
every rank's first and last frame equal to what the synthetic source planted there.  Weak scaling: per-GPU
work is fixed.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PRESETS = {  # --preset NAME: the BASELINE.json configs by name (samples per GPU, extra flags)
    "config2": {"samples": 1 << 29},                  # 1 GiB of i8 IQ on one GPU (the metric's workload; the default)
    "config3": {"samples": 1 << 33},                  # 16 GiB of i8 IQ on one GPU
    "config4": {"samples": 1 << 29, "channels": 64},  # 64 channels batched in one launch
    "config5": {"samples": 1 << 33},                  # 8 GPUs x 16 GiB: run with --gpus 8
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--samples", type=int, default=None, help="IQ samples per GPU (2 B each for i8); default 2^29 = 1 GiB")
    ap.add_argument("--preset", choices=sorted(PRESETS), default=None,
                    help="a BASELINE.json config by name (sets --samples / --channels): config3 and config5 = 16 GiB per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sample-type", choices=["i8", "i16"], default="i8",
                    help="i8 = BASELINE metric (2 B/sample); i16 = the reference's Complex<i16> (4 B/sample)")
    ap.add_argument("--channels", type=int, default=None,
                    help="split the per-GPU buffer into this many independent channels handled by ONE launch "
                         "(BASELINE configs[3]: 64); the default 1 is the metric's workload")
    ap.add_argument("--scan", choices=["default", "code", "root", "nsq", "reg", "sieve"], default="default",
                    help="i8 scan kernel: root = floor(sqrt) per sample (the product's); code / nsq / reg / sieve = the A/B kernels of "
                         "rounds 3-4 (-DADSB_AB_KERNELS=1 builds only: point ADSB_HIP_LIB at air_rs_amd/lib/variants/libadsb_hip_ab.so); "
                         "default = the library's default (ADSB_SCAN in the environment is honoured)")
    ap.add_argument("--force-gather", action="store_true",
                    help="exercise the multi-rank frame-list gather even with one rank (testing)")
    ap.add_argument("--no-feed", action="store_true", help="skip the PCIe-inclusive feed measurement after the timed region")
    args = ap.parse_args(argv)
    pre = PRESETS.get(args.preset, {})
    if args.samples is None:
        args.samples = pre.get("samples", 1 << 29)
    if args.channels is None:
        args.channels = pre.get("channels", 1)
    return args


def spawn_argv(argv, gpus, port):
    """The command line of the child that runs `bench.py --gpus N` as N ranks (one per GPU) of ONE node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` with no launcher environment: start the N ranks as a CHILD process (torch.distributed.run),
    relay its stdout (rank 0's one JSON line) and exit code.  Nothing in this parent has touched HIP (torch and the library are
    imported further down), and the child is a child, never an exec."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(spawn_argv(argv, args.gpus, port), env=env, stdout=subprocess.PIPE)
    sys.stdout.buffer.write(proc.stdout)
    sys.stdout.flush()
    return proc.returncode


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _a = parse_args()
    if _a.gpus > 1:
        sys.exit(self_launch(_a, sys.argv[1:]))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import air_rs_amd as A  # noqa: E402
from air_rs_amd import sharding  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(iq_host_i8, gpu_frames=None, target_seconds=15.0):
    """The CPU oracle (a plain-C port of the reference's thread 2; the Rust original cannot be built
    here) timed on this box's host cores, single thread like the reference, on a bounded prefix of
    the same buffer.  The frame list of its last pass is also the checker of the GPU list of the same buffer
    (`gpu_frames`: adsb_fetch of the last timed launch): offsets, bytes, status and repaired bit must be identical
    over the offsets the oracle covered -> "parity_check" (full output compared)."""
    from tests.oracle_binding import Oracle
    orc = Oracle()
    probe = min(len(iq_host_i8), 1 << 22)
    t0 = time.perf_counter()
    orc.process_buffer(iq_host_i8[:probe], max_out=1 << 16)
    dt = time.perf_counter() - t0
    rate = probe / dt
    n = int(min(len(iq_host_i8), max(probe, rate * target_seconds)))  # bounded: <= target_seconds per pass
    passes, dt, found = 0, 0.0, 0
    while passes < 4 and dt < 10.0:       # a 1 GiB buffer is only ~6 s of CPU work: repeat it to reach >= 10 s
        t0 = time.perf_counter()
        rc, frames, found = orc.process_buffer(iq_host_i8[:n], max_out=1 << 20)
        dt += time.perf_counter() - t0
        passes += 1
    out = {"value": round(passes * n / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": f"first {n} samples of the same buffer x {passes} passes, {found} frames per pass, {dt:.1f} s, 1 thread",
           "msgs_per_s": round(passes * found / dt, 1)}
    parity = None
    if gpu_frames is not None:
        try:
            got = gpu_frames[gpu_frames["offset"] < np.uint64(n - 240)]   # the offsets the oracle's prefix covers
            same = len(got) == len(frames) == found and bool((got == frames).all())
            parity = {"ok": bool(same), "frames": int(len(frames)), "gpu_frames": int(len(got)), "samples": int(n),
                      "what": "adsb_fetch of the last timed launch vs the CPU oracle over the same samples: offset, 14 bytes, status, fixed_bit"}
            if not same and len(got) == len(frames):
                parity["first_mismatch"] = int(np.nonzero(got != frames)[0][0])
        except Exception as e:  # noqa: BLE001
            parity = {"ok": False, "error": f"{type(e).__name__}: {e}"}
    # Courtesy number: the same port on every host core this process may use, the buffer
    # time-sharded with the 240-sample overlap the multi-GPU path uses.  Not the reference's configuration
    # (its thread 2 is one thread); "value" above stays the single-thread rate.
    try:
        from concurrent.futures import ThreadPoolExecutor
        cores = min(len(os.sched_getaffinity(0)), 16)   # a one-GPU box's CPU share is 16 cores
        if cores > 1:
            own = (n - 240) // cores
            shards = [iq_host_i8[k * own: k * own + own + 240] for k in range(cores)]
            work = lambda sh: orc.process_buffer(sh, max_out=1 << 18)[2]   # ctypes releases the GIL
            with ThreadPoolExecutor(cores) as pool:
                reps, t_all, got = 0, 0.0, 0
                while reps < 8 and t_all < 4.0:
                    t0 = time.perf_counter()
                    got = sum(pool.map(work, shards))
                    t_all += time.perf_counter() - t0
                    reps += 1
            out["all_cores"] = {"value": round(reps * own * cores / t_all / 1e6, 3), "unit": "Msamples/s",
                                "cores": cores, "frames_per_pass": int(got), "passes": reps, "seconds": round(t_all, 1)}
    except Exception as e:  # the courtesy number must never cost the bench line
        out["all_cores"] = {"error": str(e)}
    return out, parity


def main():
    args = parse_args()

    # The contract is ONE JSON line on stdout.  RCCL prints a five-line version banner to fd 1 when the first
    # communicator comes up (native code, not Python): keep the real stdout aside for the JSON line and point
    # fd 1 at stderr for everything else.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    if args.scan != "default":
        os.environ["ADSB_SCAN"] = args.scan  # read by adsb_create
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: run `python bench.py --gpus N` by itself (it starts its own "
                         "ranks) or under torch.distributed.run with --nproc-per-node N")
    # Rehearsal (ADSB_BENCH_REHEARSAL=1; tests/test_gpu_round2.py): N ranks SHARE GPU 0 -- RCCL refuses a communicator
    # with two ranks on one device, so the frame lists travel over gloo through pinned host memory.  Everything else
    # (shard plan, stream base, result targets in device buckets, rank-0 checks) is the N-GPU code; the throughput of
    # such a run says nothing.
    rehearsal = os.environ.get("ADSB_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_gather:
        import torch.distributed as dist_mod
        dist = dist_mod
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    st = A.ADSB_SAMPLE_I8 if args.sample_type == "i8" else A.ADSB_SAMPLE_I16
    bps = 2 if args.sample_type == "i8" else 4
    n = args.samples if args.sample_type == "i8" else args.samples // 2  # same bytes per GPU
    nch = max(1, args.channels)
    n_ch = (n // nch) & ~7                  # samples per channel (stride = length: contiguous, 16-byte aligned)
    if nch > 1:
        n = n_ch * nch
    cfg = A.synth_default()
    cfg_slot = cfg.slot_len
    own = n - A.WINDOW                      # offsets this rank owns
    first = rank * own                      # its slice of the long stream (240-sample read halo)
    cap = n // cfg_slot + 8192              # frame capacity: at most one frame per slot, + margin for noise
    stream = torch.cuda.current_stream()
    dem = A.AdsbDemod(device=local_rank, sample_type=st, max_samples=n_ch if nch > 1 else n, max_out=cap,
                      max_channels=nch, stream=stream.cuda_stream, host_staging=False)
    if st == A.ADSB_SAMPLE_I16:
        cfg.amp_shift = 6
    iq = torch.empty(n * bps, dtype=torch.int8, device="cuda")
    if nch == 1:
        dem.synth_fill_device(cfg, 0, first, n, iq.data_ptr())
    else:
        for c in range(nch):                # every channel is its own stream (its own generator channel)
            dem.synth_fill_device(cfg, c, first, n_ch, iq.data_ptr() + c * n_ch * bps)
    torch.cuda.synchronize()

    multi = dist is not None
    # Multi-rank: every launch writes its ordered frame list ([32-byte header | frames]) straight into a slot of
    # a bucket (adsb_set_result_target: no device-to-device copy); one RCCL gather to rank 0 per BUCKET launches,
    # issued from a side stream and double-buffered, so the exchange of bucket k overlaps the demodulation of
    # bucket k+1 (sharding.BucketGather).  Per-step cross-stream synchronisation was measured at ~40-60 us
    # (15-20 % of a step); per bucket it is noise.
    BUCKET = 8
    bg = sharding.BucketGather(dist, cap, bucket=BUCKET, device="cuda", host_staged=rehearsal) if multi else None
    if multi:
        dem.set_stream_base(first)        # absolute stream offsets: rank 0 concatenates, nothing to rebase
    state = {"launched": False}

    def step():
        if multi:
            ptr, _ = bg.begin_launch()
            dem.set_result_target(ptr, bg.payload)
        if nch == 1:
            dem.demod_device_async(iq.data_ptr(), n)
        else:
            dem.demod_device_async(iq.data_ptr(), n_ch, n_channels=nch, channel_stride=n_ch)
        state["launched"] = True
        if multi:
            bg.end_launch(dem.stream_wait_results)

    def drain():
        if multi:
            bg.drain(dem.stream_wait_results)
        if state["launched"]:
            dem.fetch_counts()  # waits for the last ordering pass
        torch.cuda.synchronize()

    def allmax(x):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        if dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def measure(W, K):
        """The contract's procedure: W untimed steps, then exactly K steps between barrier + synchronize on both
        sides; wall time = max over ranks."""
        for _ in range(W):
            step()
        drain()
        # kernel durations are sampled on every 4th launch of the timed region (every launch for very short
        # runs): event-carrying dispatches cost a few us each; sampling keeps the loop within 1 % of untimed
        every = 4 if K >= 16 else (2 if K >= 4 else 1)
        dem.timing_enable(0 if os.environ.get("ADSB_BENCH_NO_TIMING") == "1" else every)
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        t_enq = time.perf_counter() - t0  # host time to enqueue all steps (host-bound if ~= total)
        drain()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        demod_ms, decode_ms, order_ms, n_timed = dem.timing_read3()
        dem.timing_enable(False)
        return {"dt": allmax(dt), "t_enq": t_enq, "demod_ms": demod_ms, "decode_ms": decode_ms, "order_ms": order_ms,
                "n_timed": n_timed}

    # Two measurements by the same procedure.  `cold`: straight from an idle GPU -- the first few milliseconds of
    # load run at boost clock, then the power controller overshoots and throttles (launches 8..40 of a burst are the
    # slowest of all), and the clock converges on its sustained value over the next ~0.1-0.2 s
    # (profiles/r02_scan_drift.txt, r02_steps_sweep.txt).  A 25-launch run from idle (5 ms) measures that trough.
    # `settled`: the same W + K after `settle_s` seconds of the same steps, untimed -- what a stream that keeps
    # arriving gets, and what `value` reports.  The cold figures are in the JSON line as `cold_start`.
    settle_s = float(os.environ.get("ADSB_BENCH_SETTLE_S", "0.4"))
    cold = measure(args.warmup, args.steps)
    settle_launches = 0
    if settle_s > 0:
        settle_launches = int(min(4000, max(32, settle_s / (cold["dt"] / args.steps))))  # same count on every rank
        for _ in range(settle_launches):
            step()
        drain()
    m = measure(args.warmup, args.steps) if settle_s > 0 else cold
    # untimed launches in front of the reported K steps: the cold run's W + K, the settle phase, the W of the reported run
    warmup_total = args.warmup + ((args.warmup + args.steps + settle_launches) if settle_s > 0 else 0)
    dt, t_enq = m["dt"], m["t_enq"]
    demod_ms, decode_ms, order_ms, n_timed = m["demod_ms"], m["decode_ms"], m["order_ms"], m["n_timed"]

    n_out, total, flags = dem.fetch_counts()

    red_dev = "cpu" if rehearsal else "cuda"
    cnt = torch.tensor([float(n_out)], dtype=torch.float64, device=red_dev)
    if dist:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    frames_per_step = float(cnt.item())
    # what the process group itself reports (a reader can see that N ranks took part) and every rank's own kernel times
    ranks_seen = {"world_size": dist.get_world_size() if dist else 1, "backend": dist.get_backend() if dist else None,
                  "launched_by": os.environ.get("TORCHELASTIC_RUN_ID") and "torch.distributed.run" or "direct"}
    mine = torch.tensor([demod_ms, decode_ms, float(local_rank)], dtype=torch.float64, device=red_dev)
    per_rank = [mine]
    if dist:
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
    per_rank = [[float(x) for x in t.cpu()] for t in per_rank]

    if rank == 0:
        gather_check = None
        def check_gather():  # what rank 0 holds after the last gather: every rank's list of the last launch
            per_rank = bg.lists_of_launch(bg.last_launch())
            got = sum(n_r for (n_r, _, _, _) in per_rank)
            assert got == int(frames_per_step), (got, frames_per_step)
            merged = sharding.merge_rank_lists(per_rank)      # asserts global order after plain concatenation
            planted_hits = 0
            for r, (n_r, tot_r, fl_r, fr) in enumerate(per_rank):
                assert fl_r == 0 and n_r == tot_r, (r, n_r, tot_r, fl_r)
                lo, hi = r * own, r * own + own               # the offsets rank r owns
                assert n_r == 0 or (lo <= int(fr["offset"][0]) and int(fr["offset"][-1]) < hi), (r, lo, hi)
                for f in ([fr[0], fr[-1]] if n_r else []):   # spot check against the host-side generator
                    slot_idx = int(f["offset"]) // cfg.slot_len
                    for sl in (slot_idx, slot_idx - 1):
                        if sl < 0:
                            continue
                        present, start, clean, sent, kind = A.synth_slot(cfg, 0, sl)
                        if present and start == int(f["offset"]):
                            assert kind in (0, 1) and bytes(f["bytes"]) == bytes(clean), (r, sl, kind)
                            assert int(f["status"]) == kind
                            planted_hits += 1
            assert planted_hits >= world, "no gathered frame could be matched with a planted one"
            return {"ok": True, "launch": bg.last_launch(), "frames": int(len(merged)), "globally_ordered": True,
                    "spot_checked_frames": planted_hits}
        if multi:  # a failed check is reported in the line, it does not cost the measurement
            try:
                gather_check = check_gather()
            except Exception as e:  # noqa: BLE001
                gather_check = {"ok": False, "error": f"{type(e).__name__}: {e}"}
        ms_per_step = dt / args.steps * 1e3
        value = world * n * args.steps / dt / 1e6
        algo_bytes = float(bps) * n
        achieved = algo_bytes / (demod_ms * 1e-3) / 1e9 if demod_ms > 0 else 0.0
        ceil_ms = dem.time_read_ceiling(iq.data_ptr(), n * bps, 10)
        # HBM bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE x 2, gfx950 correction), which cannot run
        # inside this process: the figure is the committed one for this exact workload and says so; null otherwise.
        traffic, traffic_source, valu_issue = None, None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc) and n == 1 << (29 if bps == 2 else 28) and nch == 1:
            try:
                pj = json.load(open(pmc))
                rows = {("i8", "root"): pj["i8"], ("i16", "root"): pj["cs16"], ("i8", "code"): pj.get("i8_code_scan"),
                        ("i8", "nsq"): pj.get("i8_nsq_scan"), ("i8", "reg"): pj.get("i8_reg_scan"), ("i8", "sieve"): pj.get("i8_sieve_scan")}
                row = rows.get((args.sample_type, dem.scan))
                drv = row["demod_tiles"]["derived"] if row else None
                if drv:
                    traffic = drv.get("hbm_bytes_per_launch_fetch_size_x2")
                    traffic_source = f"profiles/pmc_summary.json ({pj.get('round', 'committed')} rocprofv3 --pmc FETCH_SIZE pass of this workload; not measured in this run)"
                    # What bounds the kernel (profiles/pmc_summary.json, the counter passes of this workload): the SIMDs issue one
                    # wave64 VALU instruction per 4 cycles (8 for a transcendental); slots = those 4-cycle units per wave, counted
                    # by SQ_ACTIVE_INST_VALU.  frac_of_valu_peak = what this run's kernel time leaves of that budget at the clock
                    # the chip held in the counter pass.
                    clock = drv.get("effective_clock_ghz")
                    slots = drv["valu_active_quad_cycles_per_wave"]
                    waves_per_simd = row["demod_tiles"]["raw"]["SQ_WAVES"] / 1024.0
                    valu_issue = {"slots_per_wave_tile": slots, "slots_per_sample": drv["valu_lane_slots_per_sample"],
                                  "valu_instructions_per_wave_tile": drv["valu_instructions_per_wave"],
                                  "effective_clock_ghz": clock, "clock_source": "GRBM_GUI_ACTIVE / 8 / kernel time of the counter pass",
                                  "valu_issue_ms_at_that_clock": round(slots * waves_per_simd * 4 / (clock * 1e6), 4) if clock else None,
                                  "frac_of_valu_peak": round(slots * waves_per_simd * 4 / (clock * 1e6) / demod_ms, 3) if clock and demod_ms > 0 else None,
                                  "frac_of_valu_peak_in_counter_pass": drv.get("valu_issue_utilisation"),
                                  "source": "profiles/pmc_summary.json (SQ_ACTIVE_INST_VALU, SQ_WAVES, GRBM_GUI_ACTIVE; not measured in this run)"}
            except Exception:
                traffic, traffic_source, valu_issue = None, None, None
        # hbm: the algorithmic bytes against the HBM peak (what `frac` reports, as the contract asks).  The i8 scan is not
        # limited by HBM but by VALU issue (frac_of_valu_peak ~ 0.9 while the same bytes stream 20 % faster through a kernel
        # that only reads them: read_ceiling_gbps): `bound` names the binding resource, `frac` stays the HBM fraction.
        binding = "hbm"
        if valu_issue and valu_issue.get("frac_of_valu_peak") and ceil_ms > 0 and valu_issue["frac_of_valu_peak"] > (ceil_ms / demod_ms if demod_ms > 0 else 0):
            binding = "valu-issue"
        out = {
            "metric": f"IQ Msamples/s (decoded Mode-S msgs/s alongside), 2 MSPS {args.sample_type} stream",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": warmup_total, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.sample_type, "data": "synthetic",
            "config": {"workload": f"2 MSPS {args.sample_type} IQ, {bps * n / 2**30:g} GiB synthetic buffer per GPU, fused magnitude+preamble/DF17 gate+PPM+CRC-24",
                       "samples_per_gpu": n, "bytes_per_gpu": bps * n, "frames_per_step": int(frames_per_step),
                       "channels": nch,
                       "sharding": "single buffer" if world == 1 else f"time-sharded x{world}, 240-sample read halo, {'gloo (host-staged) gather' if rehearsal else 'RCCL gather'} of frame lists",
                       "synth": {"seed": cfg.seed, "slot_len": cfg.slot_len, "noise_div": cfg.noise_div}},
            "msgs_per_s": round(frames_per_step * args.steps / dt, 1),
            "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 4),
            "roofline": {"bound": "hbm", "binding_resource": binding, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": f"adsbk::demod_tiles<{args.sample_type}, {dem.scan}>", "kernel_ms": round(demod_ms, 4),
                         "frac_of_read_ceiling": round(ceil_ms / demod_ms, 4) if demod_ms > 0 else None,
                         "kernel_does": "scan kernel: reads every IQ byte once; fused magnitude + preamble/DF17 gate + PPM slice "
                                        "of the gate survivors (their CRC-24 / repair / ordering is finish_order, the launch's second kernel)",
                         "finish_order_ms": round(decode_ms, 4), "launches_timed": n_timed,
                         "algorithmic_bytes_per_launch": int(algo_bytes),
                         "read_ceiling_gbps": round(float(bps) * n / (ceil_ms * 1e-3) / 1e9, 1)},
        }
        # The dominant kernel IS the fused magnitude + preamble/DF17 pass BASELINE.json's ">= 90 % of HBM-read roofline"
        # names (the decode of its survivors is a kernel of its own): the same figures, against both denominators.
        ceil_gbps = float(bps) * n / (ceil_ms * 1e-3) / 1e9
        out["roofline"]["fused_pass"] = {
            "what": "the fused magnitude + preamble/DF17 pass is the dominant kernel above (demod_tiles, which also slices the gate's survivors)",
            "kernel_ms": round(demod_ms, 4), "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "frac_of_read_ceiling": round(achieved / ceil_gbps, 4) if ceil_gbps > 0 else None}
        if valu_issue is not None:
            out["roofline"]["valu_issue"] = valu_issue
        if gather_check is not None:
            out["gather_check"] = gather_check
        out["ranks_seen"] = ranks_seen
        out["per_rank"] = {"scan_kernel_ms": [round(r[0], 4) for r in per_rank], "finish_order_ms": [round(r[1], 4) for r in per_rank],
                           "device": [int(r[2]) for r in per_rank]}
        out["warmup_requested"] = args.warmup
        out["warmup_launches_total"] = warmup_total  # every untimed launch before the K timed ones (= "warmup")
        out["warmup_breakdown"] = {"cold_run_warmup": args.warmup, "cold_run_timed_steps": args.steps, "settle_launches": settle_launches,
                                   "reported_run_warmup": args.warmup} if settle_s > 0 else {"reported_run_warmup": args.warmup}
        out["settle"] = {"seconds": settle_s, "launches": settle_launches,
                         "why": "untimed steps between the cold measurement and the reported one: the power controller needs ~0.1-0.2 s of "
                                "load to converge (boost, overshoot, recovery); ADSB_BENCH_SETTLE_S=0 reports the cold run as value"}
        out["cold_start"] = {"what": f"the same {args.warmup} + {args.steps} steps straight from an idle GPU",
                             "value": round(world * n * args.steps / cold["dt"] / 1e6, 1), "unit": "Msamples/s",
                             "ms_per_step": round(cold["dt"] / args.steps * 1e3, 4), "kernel_ms": round(cold["demod_ms"], 4),
                             "frac": round(algo_bytes / (cold["demod_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if cold["demod_ms"] > 0 else None}
        if rehearsal:
            out["rehearsal"] = f"{world} ranks sharing one GPU, frame lists over gloo through pinned host memory: a functional run of the N-GPU path, its throughput is not a measurement"

        if world == 1 and not args.no_feed:
            # The PCIe-inclusive rates (never `value`): buffers that arrive in HOST memory, through the streaming front end
            # (adsb_feed_*: pinned ring, two in flight), timed in C after the timed region -- the reference's own playback
            # buffer (20 000 samples, src/adsb.rs:77-79) and 16 Mi-sample buffers, against the box's pinned-copy rate.
            try:
                us_small, fr_small, nb_small = A.measure_feed(local_rank, st, 20000, 0.15)
                big = 1 << 24
                us_big, fr_big, nb_big = A.measure_feed(local_rank, st, big, 0.25)
                pin = A.measure_pinned_copy(local_rank, 64 << 20, 8)
                out["feed"] = {"what": "host-fed (PCIe-inclusive) rate of adsb_feed_*: in-place producer, two buffers in flight, every frame list popped; measured after the timed region, not part of value",
                               "us_per_20000_sample_buffer": round(us_small, 2), "msamples_per_s_20000": round(20000 / us_small, 1),
                               "frames_per_20000_sample_buffer": round(fr_small, 2), "buffers_20000": int(nb_small),
                               "gsamples_per_s": round(big / us_big / 1e3, 3), "gbytes_per_s": round(big * bps / us_big / 1e3, 2),
                               "large_buffer_samples": big, "buffers_large": int(nb_big),
                               "pinned_copy_ceiling_gbps": round(pin, 2), "frac_of_pinned_copy": round(big * bps / us_big / 1e3 / pin, 3) if pin > 0 else None}
            except Exception as e:  # noqa: BLE001  (the courtesy figures must never cost the bench line)
                out["feed"] = {"error": f"{type(e).__name__}: {e}"}

        if world == 1 and not args.no_cpu_baseline and bps == 2 and nch == 1:
            sample = iq.cpu().numpy().reshape(-1, 2)  # the whole buffer: ~6-10 s on one host core
            gpu_frames = dem.fetch()[0]               # the last timed launch's list (same buffer every step)
            out["cpu_baseline"], parity = cpu_baseline(sample, gpu_frames)
            if parity is not None:
                out["parity_check"] = parity
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    dem.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
