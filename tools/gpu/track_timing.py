#!/usr/bin/env python3
"""GPU box helper: time of the device field decode + tracker (adsb_track_device) over the frame list of
the bench workload (1 GiB synthetic i8 buffer, ~253 k frames; every synthetic frame has its own ICAO, so
this measures the sort + per-frame kernels, not long partner walks)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import air_rs_amd as A

n = 1 << 29
cfg = A.synth_default()
cap = n // cfg.slot_len + 8192
dem = A.AdsbDemod(device=0, max_samples=n, max_out=cap, stream=torch.cuda.current_stream().cuda_stream,
                  host_staging=False)
iq = torch.empty(n * 2, dtype=torch.int8, device="cuda")
dem.synth_fill_device(cfg, 0, 0, n, iq.data_ptr())
lib = dem._lib
for rep in range(3):
    dem.demod_device_async(iq.data_ptr(), n)
    n_out, _, _ = dem.fetch_counts()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = lib.adsb_track_device(dem.handle, 0.5e-6)
    assert rc == 0, rc
    torch.cuda.synchronize()
    t1 = time.perf_counter()
print(f"frames {n_out}, field decode + tracker {1e3 * (t1 - t0):.3f} ms "
      f"({n_out / (t1 - t0) / 1e6:.1f} M frames/s; {n_out * (24 + 32 + 24) / (t1 - t0) / 1e9:.2f} GB/s of records)")
pts, acs = dem.track(0.5e-6)
print("aircraft", len(acs), "new positions", int((pts["flags"] & 1).sum()))
