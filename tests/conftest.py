import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def _ensure_built():
    need = [os.path.join(ROOT, "air_rs_amd", "lib", "libadsb_hip.so"),
            os.path.join(ROOT, "oracle", "libadsb_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call([os.path.join(ROOT, "build.sh")], cwd=ROOT)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only; the product never loads it)."""
    _ensure_built()
    from tests import oracle_binding
    return oracle_binding.Oracle()


@pytest.fixture(scope="session")
def lib():
    _ensure_built()
    import air_rs_amd
    return air_rs_amd


def _have_gpu():
    # Ask the product library itself (adsb_create fails with ADSB_E_NODEVICE without a HIP device).  Not
    # torch.cuda.is_available(): the torch wheel carries its own HIP runtime, and initialising it AFTER
    # libadsb_hip.so has loaded the system one (CPU-tier tests of the same session do) leaves the library
    # without a device.
    try:
        import air_rs_amd
        air_rs_amd.AdsbDemod(max_samples=1024, max_out=16).close()
        return True
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu(lib):
    if not _have_gpu():
        pytest.skip("no GPU visible")
    return lib
