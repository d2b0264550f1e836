#!/usr/bin/env python3
"""GPU box helper: per-segment cycle counts of the streaming kernel's workgroup 0 (a -DADSB_STAMPS=1
build, selected with ADSB_HIP_LIB), one launch over a 1 GiB synthetic buffer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import air_rs_amd as A

n = 1 << 29
cfg = A.synth_default()
dem = A.AdsbDemod(device=0, max_samples=n, max_out=n // cfg.slot_len + 8192,
                  stream=torch.cuda.current_stream().cuda_stream, host_staging=False)
iq = torch.empty(n * 2, dtype=torch.int8, device="cuda")
dem.synth_fill_device(cfg, 0, 0, n, iq.data_ptr())
for _ in range(3):
    dem.demod_device_async(iq.data_ptr(), n)
dem.fetch_counts()
s = dem.stamps()
rounds = max(int(s[15]), 1)
names = ["gate", "gate waves: wait at R", "-", "-", "-", "-", "-", "-",
         "lookup: conversion pass", "lookup: wait at R", "-", "-", "decode (tile i-1)", "decode waves: wait at R",
         "record (tile i-2; zero on decode wave 0 unless it is the record wave)"]
print(f"kernel {dem.kernel}, rounds {rounds}")
for k, nm in enumerate(names):
    if nm != "-":
        print(f"  {nm:32s} {int(s[k]) / rounds:10.0f} cycles/round")
print(f"  gate-wave round total            {sum(int(x) for x in s[:7]) / rounds:10.0f}")
print(f"  lookup-wave round total          {sum(int(x) for x in s[8:12]) / rounds:10.0f}")
print(f"  decode-wave round total          {sum(int(x) for x in s[12:15]) / rounds:10.0f}")
w = dem.stamps_waves()
roles = ["gate"] * 7 + ["lookup"] * 5 + ["decode"] * 4
print("  busy cycles per round, by wave (barrier exit -> next barrier arrival):")
print("   " + "  ".join(f"{roles[k][0]}{k}:{int(w[k]) / rounds:.0f}" for k in range(16)))
