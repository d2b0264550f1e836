"""The C-ABI shared library loads and exports every symbol include/*.h declares (no compute calls:
there is no GPU in this tier), and fails loudly without a device instead of falling back."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(adsb_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(lib):
    so = ctypes.CDLL(os.path.join(ROOT, "air_rs_amd", "lib", "libadsb_hip.so"))
    names = _declared("adsb_hip.h") + _declared("adsb_host.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(so, n), f"{n} declared in include/ but not exported"
    # and the Python binding covers all of them
    from air_rs_amd import _lib
    assert set(names) == set(_lib.PROTOTYPES)


def test_frame_layout_is_24_byte_pod(lib):
    from air_rs_amd import _lib
    assert ctypes.sizeof(_lib.AdsbFrame) == 24
    assert _lib.AdsbFrame.offset.offset == 0 and _lib.AdsbFrame.bytes.offset == 8
    assert _lib.AdsbFrame.status.offset == 22 and _lib.AdsbFrame.fixed_bit.offset == 23
    assert lib.FRAME_DTYPE.itemsize == 24


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.AdsbError) as e:
        lib.AdsbDemod()
    assert e.value.code == lib.ADSB_E_NODEVICE


def test_bad_arguments(lib):
    from air_rs_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    assert L.adsb_create(None, ctypes.byref(h)) == lib.ADSB_E_ARG
    cfg = _lib.AdsbCfg(99, 0, 0, 1, 1000, 10, None, 1, 0)  # wrong ABI version
    assert L.adsb_create(ctypes.byref(cfg), ctypes.byref(h)) == lib.ADSB_E_ARG
    assert b"240" in L.adsb_strerror(lib.ADSB_E_SHORT)
    assert L.adsb_demod(None, None, 0, None, 0, None, None) == lib.ADSB_E_ARG


def test_product_does_not_reference_the_oracle():
    # oracle/ is test infrastructure: nothing under air_rs_amd/ or include/ may include, link or load it
    bad = []
    for base in ("air_rs_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".so", ".pyc")):
                    continue
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if "adsb_oracle" in txt or "oracle/" in txt or "libadsb_oracle" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
    import subprocess
    out = subprocess.run(["ldd", os.path.join(ROOT, "air_rs_amd", "lib", "libadsb_hip.so")],
                         capture_output=True, text=True).stdout
    assert "oracle" not in out
