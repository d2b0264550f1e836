#!/usr/bin/env python3
"""GPU box helper: where a tile's time goes inside demod_tiles (a -DADSB_TILE_STAMPS=1 build selected with
ADSB_HIP_LIB): shader cycles per tile and segment as seen by lane 0 of waves 0 and 3 of every workgroup, last of a
few launches on the 1 GiB i8 bench buffer.  Measurement only (that build waits for all loads at once)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import air_rs_amd as A  # noqa: E402

n = int(os.environ.get("SAMPLES", 1 << 29))
cfg = A.synth_default()
dem = A.AdsbDemod(device=0, max_samples=n, max_out=n // cfg.slot_len + 8192,
                  stream=torch.cuda.current_stream().cuda_stream, host_staging=False)
iq = torch.empty(n * 2, dtype=torch.int8, device="cuda")
dem.synth_fill_device(cfg, 0, 0, n, iq.data_ptr())
for _ in range(4):
    dem.demod_device_async(iq.data_ptr(), n)
dem.fetch_counts()
st = dem.tile_stamps().astype(np.float64)
names = ["prologue -> loads issued", "phase 1 arithmetic + LDS stores", "barrier 1", "phase 2 (gate)", "barrier 2",
         "phase 3 (decode) -> tile end", "wait for the loads (vmcnt 0)"]
t0 = st[:, 7]
print(f"{len(st)} tiles; workgroup start times span {(t0.max() - t0.min()) / 100.0:.1f} us (s_memrealtime, 100 MHz)")
for w, base in (("wave 0", 0), ("wave 3", 8)):
    seg = st[:, base:base + 7]
    tot = seg.sum(axis=1)
    print(f"{w}: {tot.mean():.0f} cycles per tile inside the stamps (p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f})")
    for k in (0, 6, 1, 2, 3, 4, 5):
        print(f"   {names[k]:34s} mean {seg[:, k].mean():8.0f}  p10 {np.percentile(seg[:, k], 10):8.0f}  p90 {np.percentile(seg[:, k], 90):8.0f}  {100.0 * seg[:, k].sum() / tot.sum():5.1f} %")
# steady state only: tiles started in the middle half of the launch
mid = (t0 > np.percentile(t0, 25)) & (t0 < np.percentile(t0, 75))
seg = st[mid, 0:7]
print("wave 0, tiles started in the middle half of the launch:")
for k in (0, 6, 1, 2, 3, 4, 5):
    print(f"   {names[k]:34s} mean {seg[:, k].mean():8.0f}")
print(f"   total {seg.sum(axis=1).mean():.0f}")
# shader clock under this load: s_memtime against s_memrealtime (100 MHz) at the start of wave 3 of every tile;
# the 32-bit words may wrap: differences from the earliest tile, modulo 2^32
raw = dem.tile_stamps()
rt = (raw[:, 15] - raw[:, 15].min()).astype(np.int64)          # (no wrap expected within a launch: 43 s period)
k0 = int(np.argmin(rt))
mt = (raw[:, 14] - raw[k0, 14]).astype(np.uint32).astype(np.int64)   # cycles since the earliest tile, mod 2^32
keep = (rt > np.percentile(rt, 10)) & (rt < np.percentile(rt, 90))
slope = np.polyfit(rt[keep].astype(np.float64), mt[keep].astype(np.float64), 1)[0]
print(f"shader clock during the launch: {slope * 100.0:.0f} MHz (s_memtime ticks per s_memrealtime tick x 100 MHz)")
out = os.environ.get("STAMPS_OUT")
if out:
    np.save(out, raw)
