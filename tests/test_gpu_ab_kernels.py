"""One parity run per A/B scan kernel (code, nsq, reg, sieve) through the -DADSB_AB_KERNELS=1 build of the library.

The product library (air_rs_amd/lib/libadsb_hip.so) carries ONE i8 scan kernel; the kernels round 3-4 measured against it
are compiled into air_rs_amd/lib/variants/libadsb_hip_ab.so only (build.sh).  A process loads one library, so each kernel's
cases (tests/ab_cases.py: sizes around the tile edges, constant / saturated / coarse input, error mixes, both launch paths,
slot-pool loss, channels; for the code scan its table and the levels where codes tie) run in a child pytest whose
ADSB_HIP_LIB points at that build."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB_LIB = os.path.join(ROOT, "air_rs_amd", "lib", "variants", "libadsb_hip_ab.so")


@pytest.mark.parametrize("scan", ["code", "nsq", "reg", "sieve"])
def test_ab_kernel_parity(gpu, scan):
    assert os.path.exists(AB_LIB), "build.sh builds it next to the product library"
    env = dict(os.environ, ADSB_HIP_LIB=AB_LIB, ADSB_SCAN=scan)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "ab_cases.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]
