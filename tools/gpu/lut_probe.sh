#!/bin/bash
# GPU box helper: LDS-table feasibility probe (tools/ubench/lut_probe.hip) on the bench's synthetic samples
set -o pipefail
mkdir -p gpurun_out
python3 - <<'PY'
import numpy as np, air_rs_amd as A
cfg = A.synth_default(seed=1)
iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 1 << 20)
np.asarray(iq).tofile("gpurun_out/iq_probe.bin")
m = np.sqrt(iq[0::2].astype(np.int32) ** 2 + iq[1::2].astype(np.int32) ** 2)
print("synthetic i8 magnitudes: mean %.1f, p50 %.0f, p90 %.0f, p99 %.0f, max %.0f; share >= 90: %.4f" % (m.mean(), *np.percentile(m, [50, 90, 99]), m.max(), (m >= 90).mean()))
PY
timeout -k 10 120 tools/ubench/lut_probe gpurun_out/iq_probe.bin | tee gpurun_out/lut_probe.txt
rm -f gpurun_out/iq_probe.bin
