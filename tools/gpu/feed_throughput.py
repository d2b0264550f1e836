#!/usr/bin/env python3
"""GPU box helper: host-fed throughput of the streaming front end (adsb_feed_*) next to the pinned host-to-device
copy ceiling of this box, and next to one blocking adsb_demod() per buffer (pageable memory, the round-1 path)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import air_rs_amd as A  # noqa: E402


def h2d_ceiling(nbytes=256 << 20, reps=10):
    src = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e9


def run_feed(st, chunk, n_buf, carry, zero_copy):
    bps = 2 if st == A.ADSB_SAMPLE_I8 else 4
    cfg = A.synth_default(seed=9)
    if st == A.ADSB_SAMPLE_I16:
        cfg.amp_shift = 5
    data = A.synth_fill_host(cfg, st, 0, 0, chunk * min(n_buf, 8))
    bufs = [data[k * chunk:(k + 1) * chunk] for k in range(min(n_buf, 8))]
    with A.AdsbDemod(sample_type=st, max_samples=chunk + 240, max_out=chunk // 200 + 4096, host_staging=False) as d:
        with A.Feed(d, max_chunk=chunk, carry=carry, ring_slots=3) as f:
            frames = 0

            def one(k):
                nonlocal frames
                if zero_copy:
                    slot = f.acquire()   # a real producer (SDR driver, file reader) writes its samples here;
                    if k < 4:            # the fill is the producer's cost and is not timed: the three ring slots
                        slot[:chunk] = bufs[k % len(bufs)]   # are filled once, during the warm-up pushes
                    f.push_acquired(chunk)
                else:
                    f.push(bufs[k % len(bufs)])
                if f.in_flight == 2:
                    frames += len(f.pop()[0])
            for k in range(4):
                one(k)
            while f.in_flight:
                f.pop()
            frames = 0
            t0 = time.perf_counter()
            for k in range(n_buf):
                one(k + 4)
            while f.in_flight:
                frames += len(f.pop()[0])
            dt = time.perf_counter() - t0
    return chunk * n_buf / dt / 1e6, chunk * n_buf * bps / dt / 1e9, dt / n_buf * 1e6, frames


def run_blocking(st, chunk, n_buf):
    cfg = A.synth_default(seed=9)
    data = A.synth_fill_host(cfg, st, 0, 0, chunk)
    with A.AdsbDemod(sample_type=st, max_samples=chunk, max_out=chunk // 200 + 4096) as d:
        for _ in range(3):
            d.demod(data)
        t0 = time.perf_counter()
        for _ in range(n_buf):
            d.demod(data)
        dt = time.perf_counter() - t0
    return chunk * n_buf / dt / 1e6, dt / n_buf * 1e6


print(f"pinned host -> device copy ceiling (256 MiB, torch pinned tensor): {h2d_ceiling():.1f} GB/s")
print("streaming front end, frames popped one buffer behind; Msamples/s | GB/s of IQ | us per buffer")
for st, name in ((A.ADSB_SAMPLE_I8, "i8"), (A.ADSB_SAMPLE_I16, "cs16")):
    for chunk, n_buf in ((20_000, 2000), (1 << 20, 200), (1 << 24, 24)):
        for carry in (False, True):
            for zc in (False, True):
                ms, gb, us, fr = run_feed(st, chunk, n_buf, carry, zc)
                print(f"  {name:4s} chunk {chunk:9d} {'carry ' if carry else 'parity'} {'in-place producer' if zc else 'push (host memcpy)'}:"
                      f" {ms:10.1f} Msamples/s {gb:7.2f} GB/s {us:9.1f} us/buffer  ({fr} frames)")
print("one blocking adsb_demod() per buffer (pageable host memory, synchronous copy + kernels + fetch):")
for chunk, n_buf in ((20_000, 1000), (1 << 20, 50)):
    ms, us = run_blocking(A.ADSB_SAMPLE_I8, chunk, n_buf)
    print(f"  i8   chunk {chunk:9d}: {ms:10.1f} Msamples/s {us:9.1f} us/buffer")
