#!/usr/bin/env python3
"""Replay an IQ capture through the GPU path and print what `air_rs adsb -p FILE -m stream` prints
(reference: src/main.rs:19-23 -> launch_adsb, src/adsb.rs:126-173; text format src/adsb/packet.rs:77-99).

  tools/replay.py capture.c16                 # the reference's format (utils.rs:22-43), reference semantics
  tools/replay.py capture.bin --format u8     # raw rtl_sdr capture (unsigned bytes)
  tools/replay.py capture.c16 --carry --tail  # also decode frames straddling buffers and the last chunk

Everything below the argument parsing is one call through the C ABI (adsb_replay_file, include/adsb_host.h).
The "Processed Time" line carries no value (the reference prints the wall clock there)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import air_rs_amd as A  # noqa: E402
from air_rs_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("file")
    ap.add_argument("--format", choices=["c16", "u8"], default=None, help="default: by extension (.c16 -> c16, else u8)")
    ap.add_argument("--chunk", type=int, default=20000, help="samples per buffer (adsb.rs:77-79: 20000)")
    ap.add_argument("--carry", action="store_true", help="carry the last 240 samples over (not reference behaviour)")
    ap.add_argument("--tail", action="store_true", help="also send the last chunk (not reference behaviour)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--summary", action="store_true", help="print counts to stderr")
    a = ap.parse_args()
    fmt = a.format or ("c16" if a.file.endswith(".c16") else "u8")
    st = A.ADSB_SAMPLE_I16 if fmt == "c16" else A.ADSB_SAMPLE_I8
    n_max = os.path.getsize(a.file) // (4 if fmt == "c16" else 2)
    with A.AdsbDemod(device=a.device, sample_type=st, max_samples=a.chunk + 240, max_out=a.chunk + 240,
                     host_staging=False) as d:
        frames, n_buf, n_samp, text = d.replay_file(a.file, L.ADSB_FILE_C16 if fmt == "c16" else L.ADSB_FILE_U8,
                                                    chunk_len=a.chunk, carry=a.carry, send_tail=a.tail,
                                                    max_frames=max(n_max // 200, 1 << 16))
    sys.stdout.write(text)
    if a.summary:
        print(f"{n_samp} samples, {n_buf} buffers of {a.chunk}, {len(frames)} packets", file=sys.stderr)


if __name__ == "__main__":
    main()
