"""Committed golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py): the oracle
must reproduce them on CPU, and the HIP path must reproduce them on the GPU."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
# IQ fixtures (tracker_traffic.npz is a frame-list fixture: tests/test_tracker.py)
FILES = sorted(f for f in glob.glob(os.path.join(HERE, "golden", "*.npz")) if "tracker" not in os.path.basename(f))
REF_FRAMES = ["8d7c6b3020293532d70820fc8090", "8d7c6b30581304f388bb4455896f", "8d40621d58c386435cc412692ad6",
              "8d40621d58c382d690c8ac2863a7", "8d7c6b30580d107903b3cabf62ab", "8d7c6b30580d24eeaebb2dfea5bb",
              "8d406b902015a678d4d220aa4bda"]


def _load(path):
    z = np.load(path, allow_pickle=False)
    return z["iq"], z["frames"]


def test_fixture_set_is_complete():
    names = {os.path.basename(f)[:-4] for f in FILES}
    assert {"ref_frames_i8", "ref_frames_i16", "sqrt_ties_i8", "constant_i8", "len240_i8", "len241_i8",
            "bit_errors_i8", "synth_i8"} <= names


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_reproduces_golden(oracle, path):
    iq, want = _load(path)
    rc, got, n = oracle.process_buffer(iq)
    assert rc == 0 and n == len(want) and (got == want.astype(got.dtype)).all()


def test_golden_contents_mean_what_they_should():
    iq, fr = _load(os.path.join(HERE, "golden", "ref_frames_i8.npz"))
    assert [bytes(f["bytes"]).hex() for f in fr] == REF_FRAMES and (fr["status"] == 0).all()
    assert list(fr["offset"]) == [300 + 400 * k for k in range(7)]
    iq, fr = _load(os.path.join(HERE, "golden", "ref_frames_i16.npz"))
    assert [bytes(f["bytes"]).hex() for f in fr] == REF_FRAMES
    iq, fr = _load(os.path.join(HERE, "golden", "sqrt_ties_i8.npz"))
    assert 100 in fr["offset"] and not fr[fr["offset"] == 100]["bytes"].any()
    iq, fr = _load(os.path.join(HERE, "golden", "constant_i8.npz"))
    assert len(fr) == 500 - 240 and not fr["bytes"].any()
    assert len(_load(os.path.join(HERE, "golden", "len240_i8.npz"))[1]) == 0
    iq, fr = _load(os.path.join(HERE, "golden", "len241_i8.npz"))
    assert len(fr) == 1 and fr[0]["offset"] == 0
    iq, fr = _load(os.path.join(HERE, "golden", "bit_errors_i8.npz"))
    assert len(fr) == 1 and fr[0]["offset"] == 50 and fr[0]["status"] == 1 and fr[0]["fixed_bit"] == 43
    assert bytes(fr[0]["bytes"]).hex() == REF_FRAMES[4]  # data-bit flip repaired; CRC-bit and double flips dropped


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_hip_reproduces_golden(gpu, path):
    iq, want = _load(path)
    st = gpu.ADSB_SAMPLE_I8 if iq.dtype == np.int8 else gpu.ADSB_SAMPLE_I16
    with gpu.AdsbDemod(sample_type=st, max_samples=max(len(iq), 240), max_out=4096) as d:
        got, flags = d.demod(iq)
    assert flags == 0 and len(got) == len(want) and (got == want.astype(got.dtype)).all()
