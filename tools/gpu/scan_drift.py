#!/usr/bin/env python3
"""GPU box helper (run under rocprofv3 --kernel-trace): how does the scan kernel's duration develop over a run of
back-to-back launches?  Three bursts of 40 launches of the 1 GiB i8 bench buffer, 100 ms of idle before each:
  burst 0: the product's launch (scan + finish + gather); burst 1: scan + gather only (adsb_debug_fused_pass_only);
  burst 2: the product's launch again."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import air_rs_amd as A

n = 1 << 29
cfg = A.synth_default()
dem = A.AdsbDemod(max_samples=n, max_out=n // cfg.slot_len + 8192, host_staging=False,
                  stream=torch.cuda.current_stream().cuda_stream)
iq = torch.empty(2 * n, dtype=torch.int8, device="cuda")
dem.synth_fill_device(cfg, 0, 0, n, iq.data_ptr())
torch.cuda.synchronize()
for burst in range(3):
    time.sleep(0.1)
    dem.fused_pass_only(burst == 1)
    for _ in range(40):
        dem.demod_device_async(iq.data_ptr(), n)
    dem.fetch_counts()
    torch.cuda.synchronize()
dem.close()
