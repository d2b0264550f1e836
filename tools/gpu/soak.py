#!/usr/bin/env python3
"""GPU box helper: randomized parity soak.  For `--seconds` of wall time: random buffer length, synthetic
source parameters (frame density, noise level, error mix), sample type, and now and then a dense input on a
small-capacity context (slot-pool overflow, host re-plan); the HIP path's frame
list must equal the CPU oracle's.  Prints one line per failure and a summary; exit code 1 on any mismatch."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import air_rs_amd as A
from tests.oracle_binding import Oracle

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--scans", default="root", help="i8 scan kernels to soak (the A/B build also has code,nsq,reg,sieve)")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
orc = Oracle()
def make(st, max_samples, max_out, scan="root", small_path="1"):
    os.environ["ADSB_SCAN"], os.environ["ADSB_SMALL_PATH"] = scan, small_path   # read by adsb_create
    return A.AdsbDemod(sample_type=st, max_samples=max_samples, max_out=max_out)


# every kernel the loaded library carries: i8 root scan (the product's; with ADSB_HIP_LIB pointing at the -DADSB_AB_KERNELS=1
# build and --scans root,code,nsq,reg,sieve also the A/B kernels), CS16; the one-dispatch path for small buffers on (default) and
# off; and small frame capacities (dense inputs overflow the slot pool and are re-planned)
SCANS = args.scans.split(",")
ctx = {(A.ADSB_SAMPLE_I16, "root"): make(A.ADSB_SAMPLE_I16, 1 << 22, 1 << 19),
       (A.ADSB_SAMPLE_I16, "root-3k"): make(A.ADSB_SAMPLE_I16, 1 << 22, 1 << 19, small_path="0"),
       (A.ADSB_SAMPLE_I16, "small"): make(A.ADSB_SAMPLE_I16, 1 << 20, 3000)}
for sc in SCANS:
    ctx[(A.ADSB_SAMPLE_I8, sc)] = make(A.ADSB_SAMPLE_I8, 1 << 22, 1 << 19, scan=sc)
    ctx[(A.ADSB_SAMPLE_I8, sc + "-3k")] = make(A.ADSB_SAMPLE_I8, 1 << 22, 1 << 19, scan=sc, small_path="0")
    ctx[(A.ADSB_SAMPLE_I8, "small-" + sc)] = make(A.ADSB_SAMPLE_I8, 1 << 20, 3000, scan=sc)
ctx[(A.ADSB_SAMPLE_I8, "small-root-3k")] = make(A.ADSB_SAMPLE_I8, 1 << 20, 3000, small_path="0")
t0 = time.time()
runs = fails = frames_total = 0
t_note = t0
while time.time() - t0 < args.seconds:
    if time.time() - t_note > 30:      # a silent run is taken for a hung one on the GPU box
        t_note = time.time()
        print(f"  ... {runs} buffers, {fails} mismatches after {t_note - t0:.0f} s", flush=True)
    st = A.ADSB_SAMPLE_I8 if rng.random() < 0.7 else A.ADSB_SAMPLE_I16
    kern = str(rng.choice([k for sc in SCANS for k in (sc, sc + "-3k")])) if st == A.ADSB_SAMPLE_I8 else str(rng.choice(["root", "root-3k"]))
    n = int(rng.choice([rng.integers(240, 4000), rng.integers(4000, 200000), rng.integers(200000, 3000000)]))
    dense = rng.random() < 0.06
    if dense:
        kern = str(rng.choice(["small-" + sc for sc in SCANS] + ["small-root-3k"])) if st == A.ADSB_SAMPLE_I8 else "small"
        n = min(n, 1 << 20)
    cfg = A.synth_default(seed=int(rng.integers(1, 1 << 40)), slot_len=int(rng.choice([300, 600, 2000, 9000])))
    cfg.noise_div = int(rng.choice([3, 8, 18, 60, 200]))
    level = int(rng.integers(0, 4))  # now and then: samples at full scale (i8: |I|, |Q| >= 125 switch a tile's gate to integer compares)
    cfg.pct_flip_data = int(rng.integers(0, 30)); cfg.pct_flip_crc = int(rng.integers(0, 10)); cfg.pct_flip_two = int(rng.integers(0, 10))
    cfg.frame_pct = int(rng.choice([0, 30, 100]))
    if st == A.ADSB_SAMPLE_I16:
        cfg.amp_shift = int(rng.integers(0, 8))
    iq = A.synth_fill_host(cfg, st, int(rng.integers(0, 4)), int(rng.integers(0, 1 << 30)), n)
    if level == 0 and st == A.ADSB_SAMPLE_I8 and n > 1000:
        pos = rng.integers(0, n, size=max(1, n // 5000))
        iq[pos] = rng.choice(np.array([-128, -127, -126, -125, 125, 126, 127], dtype=np.int8), size=(len(pos), 2))
    if dense:  # every offset of a constant stretch is a frame; a two-level alphabet makes many ties
        mode = int(rng.integers(0, 3))
        if mode == 0:
            a, b = sorted(int(x) for x in rng.integers(0, n + 1, size=2))
            iq[a:b] = int(rng.integers(-3, 4))
        elif mode == 1:
            iq = rng.choice(np.array([-1, 0, 1], dtype=iq.dtype), size=iq.shape)
        else:
            iq[:] = 0
    d = ctx[(st, kern)]
    got, flags = d.demod(iq)
    rc, want, found = orc.process_buffer(iq, max_out=d.max_out)
    ok = rc == 0 and len(got) == len(want) and (got == want).all() and bool(flags & A.ADSB_FLAG_TRUNCATED) == (found > d.max_out)
    runs += 1
    frames_total += len(want)
    if not ok:
        fails += 1
        print(f"MISMATCH st={st} kernel={kern} n={n} seed={cfg.seed} slot={cfg.slot_len} noise_div={cfg.noise_div} "
              f"got {len(got)} want {len(want)}", flush=True)
print(f"soak: {runs} buffers, {frames_total} frames compared, {fails} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
