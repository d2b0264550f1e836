#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz: small IQ inputs + the frame list the reference algorithm yields.

The reference (Rust) cannot be executed in this image, so the expected outputs come from the CPU
oracle -- a literal restatement pinned by the reference's own KATs (tests/test_oracle_kats.py).
These files pin the oracle AND the HIP path against regressions; the crafted cases additionally
encode behaviours read from the reference source that its own tests do not cover.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import air_rs_amd as A  # noqa: E402
from tests.oracle_binding import Oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF_FRAMES = ["8d7c6b3020293532d70820fc8090", "8d7c6b30581304f388bb4455896f", "8D40621D58C386435CC412692AD6",
              "8D40621D58C382D690C8AC2863A7", "8d7c6b30580d107903b3cabf62ab", "8d7c6b30580d24eeaebb2dfea5bb",
              "8D406B902015A678D4D220AA4BDA"]


def modulate(frame14, hi, lo):
    """240 samples: preamble pulses at 0,2,7,9 (demod.rs:20-22), then 112 PPM bits (1 = pulse first)."""
    s = [lo] * 240
    for p in (0, 2, 7, 9):
        s[p] = hi
    bits = int.from_bytes(frame14, "big")
    for k in range(112):
        one = (bits >> (111 - k)) & 1
        s[16 + 2 * k + (0 if one else 1)] = hi
    return s


def place(n, items, dtype, floor=3, seed=1):
    """Frames dropped onto a non-degenerate noise floor (a constant floor would emit an all-zero
    frame at every offset).  `None` entries of a frame keep the floor sample."""
    rng = np.random.default_rng(seed)
    iq = rng.integers(-floor, floor + 1, size=(n, 2)).astype(dtype)
    for start, samples in items:
        for j, v in enumerate(samples):
            if v is not None:
                iq[start + j] = v
    return iq


def main():
    orc = Oracle()
    cases = {}

    # 1. the seven frames the reference's tests carry, modulated cleanly on a quiet floor (i8 and i16)
    items = [(300 + 400 * k, modulate(bytes.fromhex(h), (90, 20), None)) for k, h in enumerate(REF_FRAMES)]
    cases["ref_frames_i8"] = place(3400, items, np.int8)
    items16 = [(300 + 400 * k, modulate(bytes.fromhex(h), (9000, -2000), None)) for k, h in enumerate(REF_FRAMES)]
    cases["ref_frames_i16"] = place(3400, items16, np.int16, floor=400)

    # 2. floor(sqrt) ties: highs (3,4) -> 25 -> 5, lows (5,1) -> 26 -> 5.  Squared
    #    magnitudes would order them the other way; the truncated ones tie, ties pass the gate
    #    (demod.rs:29 is a strict <), the slicer's strict > yields all-zero bits, CRC(0) == 0.
    tie = modulate(bytes(14), (3, 4), (5, 1))
    cases["sqrt_ties_i8"] = place(1200, [(100, tie)], np.int8, floor=40, seed=2)

    # 3. constant input: one all-zero frame per offset
    cases["constant_i8"] = np.full((500, 2), -7, dtype=np.int8)

    # 4. exactly 240 / 241 samples (adsb.rs:98 runs 0 / 1 iterations)
    one = place(241, [(0, modulate(bytes.fromhex(REF_FRAMES[6]), (60, 0), None))], np.int8)
    cases["len241_i8"] = one
    cases["len240_i8"] = one[:240].copy()

    # 5. single-bit errors: data bit (repaired), CRC-field bit (rejected), two data bits (rejected)
    f = bytearray(bytes.fromhex(REF_FRAMES[4]))
    a = bytearray(f); a[5] ^= 0x10
    b = bytearray(f); b[12] ^= 0x01
    c = bytearray(f); c[2] ^= 0x80; c[9] ^= 0x02
    cases["bit_errors_i8"] = place(1500, [(50, modulate(bytes(a), (70, 70), None)),
                                          (500, modulate(bytes(b), (70, 70), None)),
                                          (950, modulate(bytes(c), (70, 70), None))], np.int8)

    # 6. synthetic noise + frames straddling the 20 000-sample playback chunks (lost)
    cfg = A.synth_default(seed=0xC0FFEE, slot_len=1900)
    cases["synth_i8"] = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 6000)

    for name, iq in cases.items():
        rc, frames, n = orc.process_buffer(iq)
        assert rc == 0, name
        np.savez_compressed(os.path.join(OUT, name + ".npz"), iq=iq, frames=frames)
        print(f"{name:18s} {iq.shape[0]:6d} samples ({iq.dtype}) -> {n} frames")

    # 7. behind the channel (aircraft.rs / cpr.rs): a time-ordered frame list and what the oracle's tracker
    #    makes of it -- per frame: did it complete an even/odd pair, and the position; per aircraft: the summary
    from tests.traffic import random_traffic
    traffic = random_traffic(orc, seed=77, n_aircraft=12, n_frames=600, span_s=45.0)
    trk = orc.tracker()
    new = np.zeros(len(traffic), dtype=np.uint8)
    pos = np.zeros((len(traffic), 2), dtype=np.float64)
    for k, (t, fr) in enumerate(traffic):
        got, s = trk.update(fr, t)
        new[k] = got
        if got:
            pos[k] = (s.latitude, s.longitude)
    table = sorted(trk.aircraft(), key=lambda s: s.icao)
    np.savez_compressed(
        os.path.join(OUT, "tracker_traffic.npz"),
        times=np.array([t for t, _ in traffic]), frames=np.frombuffer(b"".join(fr for _, fr in traffic), dtype=np.uint8).reshape(-1, 14),
        new_position=new, position=pos,
        icao=np.array([s.icao for s in table], dtype=np.uint32), callsign=np.array([s.callsign for s in table]),
        altitude=np.array([s.altitude for s in table], dtype=np.int32),
        has_position=np.array([s.has_position for s in table], dtype=np.uint8),
        latitude=np.array([s.latitude for s in table]), longitude=np.array([s.longitude for s in table]))
    print(f"tracker_traffic    {len(traffic)} frames, {len(table)} aircraft, {int(new.sum())} positions")


if __name__ == "__main__":
    main()
