"""Parity cases for the A/B scan kernels (code, nsq, reg, sieve), run by tests/test_gpu_ab_kernels.py in a subprocess whose
ADSB_HIP_LIB points at the -DADSB_AB_KERNELS=1 build (air_rs_amd/lib/variants/libadsb_hip_ab.so) and whose ADSB_SCAN names
the kernel under test.  Not collected by the default run (the file name does not match test_*.py): the product library has
none of these kernels.  Same bar as everywhere: bit-exact against the CPU oracle, through the C ABI."""
import os

import numpy as np
import pytest

import air_rs_amd as A

pytestmark = pytest.mark.gpu
SCAN = os.environ.get("ADSB_SCAN", "")
TILE = 16128 if SCAN == "reg" else 16384


def _eq(got, want):
    assert len(got) == len(want), (len(got), len(want))
    if len(got):
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (bad[:5], got[bad[:3]], want[bad[:3]])


@pytest.fixture(scope="module")
def dem(gpu):
    assert SCAN in ("code", "nsq", "reg", "sieve"), "run through tests/test_gpu_ab_kernels.py"
    with A.AdsbDemod(sample_type=A.ADSB_SAMPLE_I8, max_samples=1 << 22, max_out=1 << 18) as d:
        assert d.scan == SCAN
        yield d


def _check(dem, oracle, iq):
    frames, flags = dem.demod(iq)
    rc, want, n = oracle.process_buffer(iq, max_out=dem.max_out)
    assert rc == 0 and flags == 0
    _eq(frames, want)
    return frames


@pytest.mark.parametrize("n", [241, 1000, TILE + 239, TILE + 240, TILE + 241, 2 * TILE + 240, 3 * TILE + 777, 200_001,
                               32 * TILE + 240, 32 * TILE + 241, 1_000_003])
def test_synthetic_sizes(dem, oracle, n):
    cfg = A.synth_default(seed=1000 + n % 97, slot_len=700)
    frames = _check(dem, oracle, A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, n))
    assert n < 5000 or len(frames) > 0


def test_constant_saturated_coarse(dem, oracle):
    for v in ((0, 0), (5, -3), (-128, -128), (127, 127)):
        iq = np.empty((40_000, 2), dtype=np.int8)
        iq[:] = v
        _check(dem, oracle, iq)                     # one all-zero frame per offset
    rng = np.random.default_rng(5)
    for amp in (1, 2, 3, 6, 127):
        _check(dem, oracle, rng.integers(-amp, amp + 1, size=(150_000, 2), dtype=np.int8))


def test_error_mix_and_planted_frames(dem, oracle):
    cfg = A.synth_default(seed=77, slot_len=600, pct_flip_data=20, pct_flip_crc=10, pct_flip_two=10)
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 2_000_000)
    frames = _check(dem, oracle, iq)
    assert (frames["status"] == 1).sum() > 100 and (frames["status"] == 0).sum() > 1000


def test_small_path_and_two_kernel_path(gpu, oracle, monkeypatch):
    cfg = A.synth_default(seed=61, slot_len=500)
    data = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 600_000)
    for mode in ("1", "0"):
        monkeypatch.setenv("ADSB_SMALL_PATH", mode)
        with A.AdsbDemod(max_samples=600_000, max_out=1 << 15) as d:
            for n in (241, 20_000, TILE + 240, 3 * TILE + 777, 600_000):
                frames, flags = d.demod(data[:n])
                rc, want, _ = oracle.process_buffer(data[:n])
                assert rc == 0 and flags == 0
                _eq(frames, want)
            dense = d.demod(np.zeros((50_000, 2), dtype=np.int8))      # 49 760 frames > max_out: truncated
            assert dense[1] & A.ADSB_FLAG_TRUNCATED and len(dense[0]) == 1 << 15
            assert (dense[0]["offset"] == np.arange(1 << 15)).all() and not dense[0]["bytes"].any()


def test_slot_pool_loss_is_repaired(gpu, oracle):
    """tiles over their 32 slots lose them (adsb_debug_pool_limit): counted in place, re-run by the host"""
    cfg = A.synth_default(seed=19, slot_len=600)
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 300_000).copy()
    iq[40_000:95_000] = (3, 4)            # constant: one frame per offset, three whole tiles and two part ones
    iq[200_000:200_300] = 0
    rc, want, n = oracle.process_buffer(iq, max_out=1 << 18)
    assert rc == 0 and n > 50_000
    with A.AdsbDemod(max_samples=300_000, max_out=1 << 18) as d:
        d.pool_limit(True)
        frames, flags = d.demod(iq)
        d.pool_limit(False)
        assert flags == 0
        _eq(frames, want)
        frames, flags = d.demod(iq)
        assert flags == 0
        _eq(frames, want)


def test_multi_channel(gpu, oracle):
    cfg = A.synth_default(seed=3, slot_len=800)
    nch, n = 5, 70_000 + 8
    chans = [A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, c, 0, n) for c in range(nch)]
    with A.AdsbDemod(max_samples=n, max_out=1 << 15, max_channels=nch, host_staging=False) as d:
        import torch
        buf = torch.from_numpy(np.concatenate(chans)).cuda()
        d.demod_device_async(buf.data_ptr(), n, n_channels=nch, channel_stride=n)
        frames, counts, total, flags = d.fetch()
        assert flags == 0
        pos = 0
        for c in range(nch):
            rc, want, _ = oracle.process_buffer(chans[c])
            assert rc == 0 and counts[c] == len(want)
            _eq(frames[pos:pos + len(want)], want)
            pos += len(want)


# ---- the code scan ---------------------------------------------------------------------------------------------------
CODE_ONLY = pytest.mark.skipif(SCAN != "code", reason="the code scan's own cases")


@CODE_ONLY
def test_code_table_is_a_superset_table(gpu, monkeypatch):
    """The code scan's gate passes wherever the reference's does only if, for EVERY n = I^2+Q^2 an i8 sample can give, the
    threshold code of n reaches the code of the largest n' with the same floor(sqrt) -- computed by the device through
    the kernel's own v_cvt_pk_fp8_f32 / v_pk_fma_f16 (adsb_create checks the same and fails otherwise)."""
    monkeypatch.setenv("ADSB_SCAN", "code")
    with A.AdsbDemod(max_samples=4096, max_out=64) as d:
        assert d.scan == "code"
        tab = d.code_table().astype(np.int64)
    code, th = tab & 0xFF, tab >> 8
    n = np.arange(32769)
    root = np.floor(np.sqrt(n)).astype(np.int64)
    top = np.minimum((root + 1) ** 2 - 1, 32768)
    assert (np.diff(code) >= 0).all() and code.max() < 0x7C             # monotone, an ordered f16 pattern in the high byte
    assert (th >= code[top]).all()
    assert (th >= code).all()
    # what makes it selective: the threshold is at most a few codes above the code itself where noise lives
    assert (th[64:] - code[64:]).max() <= 6 and np.median(th[256:] - code[256:]) <= 3


@CODE_ONLY
@pytest.mark.parametrize("div,slot", [(72, 2000), (36, 900), (18, 400), (9, 300), (4, 260)])
def test_code_scan_at_every_level(gpu, oracle, monkeypatch, div, slot):
    """Noise from sigma ~ 2 (codes tie almost everywhere: most survivors of the code gate are decided from the samples
    themselves) to sigma ~ 36 (clipping), dense frames: the code scan's list equals the oracle's."""
    monkeypatch.setenv("ADSB_SCAN", "code")
    cfg = A.synth_default(seed=100 + div, noise_div=div, slot_len=slot)
    n = 9 * TILE + 321
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, n)
    rc, want, _ = oracle.process_buffer(iq, max_out=1 << 17)
    assert rc == 0
    for small in ("1", "0"):
        monkeypatch.setenv("ADSB_SMALL_PATH", small)
        with A.AdsbDemod(max_samples=n, max_out=1 << 17) as d:
            frames, flags = d.demod(iq)
            assert flags == 0
            _eq(frames, want)


@CODE_ONLY
def test_code_scan_equal_codes_different_roots(gpu, oracle, monkeypatch):
    """Windows built from pairs of n that share a code but not a root, and share a root but not a code: the gate's and
    the slicer's uncertain cases (the samples decide), at every alignment of the window in the image."""
    monkeypatch.setenv("ADSB_SCAN", "code")
    with A.AdsbDemod(max_samples=1 << 20, max_out=1 << 17) as d:
        tab = d.code_table().astype(np.int64) & 0xFF
        # (I, Q) with I^2 + Q^2 = n for the n we want: brute force over the i8 square
        i, q = np.meshgrid(np.arange(0, 128), np.arange(0, 128), indexing="ij")
        nn = (i * i + q * q).ravel()
        first = {}
        for k in np.argsort(nn, kind="stable"):
            first.setdefault(int(nn[k]), (int(i.ravel()[k]), int(q.ravel()[k])))
        ns = np.array(sorted(first))
        roots = np.floor(np.sqrt(ns)).astype(np.int64)
        rng = np.random.default_rng(8)
        bufs = []
        for _ in range(400):
            k = int(rng.integers(1, len(ns) - 8))
            near = ns[max(0, k - 6):k + 7]                               # neighbours in n: same / adjacent code, same / adjacent root
            lo_n, hi_n = rng.choice(near, 2)
            w = np.zeros((240, 2), dtype=np.int8)
            vals = rng.choice(near, 240)
            for p in range(240):
                w[p] = first[int(vals[p])]
            for p in (0, 2, 7, 9):
                w[p] = first[int(max(lo_n, hi_n))]
            for p in (16, 19, 21, 23, 24):
                w[p] = first[int(max(lo_n, hi_n))]
            for p in (1, 3, 4, 5, 6, 8, 10, 11, 12, 13, 14, 15, 17, 18, 20, 22, 25):
                w[p] = first[int(min(lo_n, hi_n))] if rng.random() < 0.8 else first[int(rng.choice(near))]
            pad = np.zeros((int(rng.integers(0, 40)), 2), dtype=np.int8)
            bufs += [pad, w]
        iq = np.concatenate(bufs + [np.zeros((300, 2), dtype=np.int8)])
        assert len(np.unique(roots)) > 100 and len(np.unique(tab[ns])) > 40
        rc, want, n_found = oracle.process_buffer(iq, max_out=1 << 17)
        assert rc == 0 and n_found > 100
        frames, flags = d.demod(iq)
        assert flags == 0
        _eq(frames, want)


