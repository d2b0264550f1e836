#!/usr/bin/env python3
"""GPU box helper: where a tile's time goes inside demod_tiles (a -DADSB_TILE_STAMPS=1 build selected with
ADSB_HIP_LIB): mean shader cycles per tile and segment, summed by lane 0 of waves 0 and 3 of every workgroup
over a few launches on the 1 GiB i8 bench buffer.  Measurement only (the build waits for all loads at once)."""
import os
import sys

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
s = [int(x) for x in dem.stamps()]
names = ["prologue -> loads issued", "phase 1 arithmetic + LDS stores", "barrier 1", "phase 2 (gate)", "barrier 2",
         "phase 3 (decode) -> tile end", "wait for the loads (vmcnt 0)"]
for w, base in (("wave 0", 0), ("wave 3", 8)):
    tiles = max(s[base + 7], 1)
    tot = sum(s[base:base + 7])
    print(f"{w}: {tiles} tiles, {tot / tiles:.0f} cycles per tile inside the stamps")
    for k in (0, 6, 1, 2, 3, 4, 5):
        print(f"   {names[k]:34s} {s[base + k] / tiles:9.0f} cycles  {100.0 * s[base + k] / max(tot, 1):5.1f} %")
