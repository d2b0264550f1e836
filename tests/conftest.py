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
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu(lib):
    if not _have_gpu():
        pytest.skip("no GPU visible")
    return lib
