#!/usr/bin/env python3
"""GPU box helper: what the drop-in actually does in the reference's own configuration -- one blocking
adsb_demod() per received buffer of 20 000 CS16 samples (playback_thread, src/adsb.rs:75-89; 10 ms of
signal at 2 MSPS): mean wall time per call (H2D copy + kernels + D2H of the frames) and the real-time margin."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import air_rs_amd as A

# the last two rows are the PCIe-inclusive rate of the host-buffer entry point on a large buffer (pageable numpy
# memory; the copy, not the kernel, sets it -- never bench.py's `value`)
for st, name, n in ((A.ADSB_SAMPLE_I16, "CS16", 20000), (A.ADSB_SAMPLE_I8, "i8", 20000), (A.ADSB_SAMPLE_I16, "CS16", 2000000),
                    (A.ADSB_SAMPLE_I8, "i8", 1 << 27), (A.ADSB_SAMPLE_I16, "CS16", 1 << 26)):
    cfg = A.synth_default(seed=99)
    if st == A.ADSB_SAMPLE_I16:
        cfg.amp_shift = 6
    iq = A.synth_fill_host(cfg, st, 0, 0, n)
    with A.AdsbDemod(sample_type=st, max_samples=n, max_out=n) as d:
        for _ in range(20 if n <= 2000000 else 2):
            d.demod(iq)
        reps = 300 if n <= 20000 else (50 if n <= 2000000 else 5)
        t0 = time.perf_counter()
        for _ in range(reps):
            frames, _ = d.demod(iq)
        dt = (time.perf_counter() - t0) / reps
    print(f"{name:5s} {n:8d} samples per buffer: {dt * 1e6:8.1f} us per adsb_demod() call, {len(frames)} frames; "
          f"{n / dt / 1e6:8.1f} Msamples/s = {n / dt / 2e6:7.1f} x real time at 2 MSPS")
