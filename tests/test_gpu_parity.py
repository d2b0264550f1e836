"""Parity of the HIP path (through the C ABI) with the CPU oracle.  Bit-exact: frames, offsets,
status and repaired-bit index must all be identical (integer/byte work, no tolerance)."""
import os

import numpy as np
import pytest

import air_rs_amd as A

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["root"], autouse=True)
def scan_kind(request):
    """Every i8 test of this module runs once per i8 scan kernel the PRODUCT library carries: "root" (floor(sqrt) per sample,
    u8 magnitudes in LDS).  The kernels measured against it ("code", "nsq", "reg": bit-exact, none faster) live in the
    -DADSB_AB_KERNELS=1 build only; tests/test_gpu_ab_kernels.py runs this module's core cases through that library, once per
    kernel.  ADSB_SCAN is read by adsb_create."""
    old = os.environ.get("ADSB_SCAN")
    os.environ["ADSB_SCAN"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("ADSB_SCAN", None)
    else:
        os.environ["ADSB_SCAN"] = old


def _eq(got, want):
    assert len(got) == len(want), (len(got), len(want))
    if len(got):
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (bad[:5], got[bad[:3]], want[bad[:3]])


def _check(dem, oracle, iq, max_out=None):
    frames, flags = dem.demod(iq, max_out)
    rc, want, n = oracle.process_buffer(iq, max_out=dem.max_out if max_out is None else max_out)
    assert rc == 0
    cap = dem.max_out if max_out is None else max_out
    assert bool(flags & A.ADSB_FLAG_TRUNCATED) == (n > cap)
    _eq(frames, want)
    return frames


@pytest.fixture(scope="module")
def dem8(gpu, scan_kind):
    with A.AdsbDemod(sample_type=A.ADSB_SAMPLE_I8, max_samples=1 << 22, max_out=1 << 18) as d:
        assert d.scan == scan_kind
        yield d


@pytest.fixture(scope="module")
def dem16(gpu):
    with A.AdsbDemod(sample_type=A.ADSB_SAMPLE_I16, max_samples=1 << 21, max_out=1 << 18) as d:
        yield d


def test_magnitude_i8_exhaustive(dem8, oracle):
    # every (I, Q) an i8 stream can carry: the device floor(sqrt) equals utils.rs:46-52
    i, q = np.meshgrid(np.arange(-128, 128), np.arange(-128, 128), indexing="ij")
    iq = np.stack([i.ravel(), q.ravel()], axis=1).astype(np.int8)
    got = dem8.magnitudes(iq)
    want = oracle.get_magnitude(iq.astype(np.int16))
    assert (got == want).all(), (dem8.mag_mode, np.nonzero(got != want)[0][:10])


def test_nsq_values_exhaustive(dem8):
    # every (I, Q) an i8 stream can carry through the nsq kernel's packing code (dot4 for the low half, perm + dot2
    # for the high half): I^2 + Q^2 + 72 in both halves
    i, q = np.meshgrid(np.arange(-128, 128), np.arange(-128, 128), indexing="ij")
    iq = np.stack([i.ravel(), q.ravel()], axis=1).astype(np.int8)
    got = dem8.nsq_values(iq).astype(np.int64)
    want = (iq.astype(np.int64) ** 2).sum(axis=1) + 72
    assert (got == want).all(), np.nonzero(got != want)[0][:10]


def test_magnitude_i16(dem16, oracle):
    rng = np.random.default_rng(11)
    iq = rng.integers(-32768, 32768, size=(1 << 20, 2)).astype(np.int16)
    ks = np.arange(1, 32768, 7)
    edge = np.concatenate([np.stack([ks, np.zeros_like(ks)], 1), np.stack([-ks, ks], 1),
                           np.array([[-32768, -32768], [32767, 32767], [0, 0], [-32768, 0], [181, 181]])])
    # every perfect square and its neighbour the i16 range can carry: (k, 0) -> k^2, (k, 1) -> k^2 + 1, and the
    # k^2 - 1 cases that are sums of two squares are met by the random draw (the biased float estimate in
    # mag_i16 is exactly wrong there without its integer correction)
    k = np.arange(0, 32768)
    sq = np.concatenate([np.stack([k, np.zeros_like(k)], 1), np.stack([k, np.ones_like(k)], 1),
                         np.stack([-k, k], 1), np.stack([k, -32768 * np.ones_like(k)], 1)])
    iq = np.concatenate([iq, edge.astype(np.int16), sq.astype(np.int16)])
    assert (dem16.magnitudes(iq) == oracle.get_magnitude(iq)).all()


@pytest.mark.parametrize("n", [240, 241, 255, 271, 272, 1000, 2016 + 239, 2016 + 241, 4032 + 240, 4032 + 241, 8192 + 239, 8192 + 240,
                               8192 + 241, 16128 + 239, 16128 + 240, 16128 + 241, 16384 + 239, 16384 + 240, 16384 + 241, 20000,
                               32256 + 240, 32256 + 241, 32768 + 239, 32768 + 240, 32768 + 241, 65536 + 240, 100003])
def test_synthetic_sizes_i8(dem8, oracle, n):
    cfg = A.synth_default(seed=100 + n, slot_len=600)
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, n)
    fr = _check(dem8, oracle, iq)
    if n >= 20000:
        assert len(fr) > 5


def test_short_buffer(dem8):
    with pytest.raises(A.AdsbError) as e:
        dem8.demod(np.zeros((239, 2), dtype=np.int8))
    assert e.value.code == A.ADSB_E_SHORT


def test_constant_input_emits_every_offset(dem8, oracle):
    # all-equal magnitudes pass the gate, slice to zeros, CRC(0) == 0
    for val in (0, 127, -128):
        iq = np.full((40000, 2), val, dtype=np.int8)
        fr = _check(dem8, oracle, iq)
        assert len(fr) == 40000 - 240


def test_truncation_returns_first_frames(dem8, oracle):
    iq = np.zeros((5000, 2), dtype=np.int8)
    fr = _check(dem8, oracle, iq, max_out=100)
    assert (fr["offset"] == np.arange(100)).all()


def test_slot_store_overflow_is_replanned(gpu, oracle):
    # far more survivors than max_out + one tile: the slot store overflows and the needed tiles are
    # redone in batches; the first max_out frames must still come back, in order
    with A.AdsbDemod(max_samples=400000, max_out=50000) as d:
        iq = np.zeros((400000, 2), dtype=np.int8)
        fr, flags = d.demod(iq)
        assert flags & A.ADSB_FLAG_TRUNCATED
        assert len(fr) == 50000 and (fr["offset"] == np.arange(50000)).all() and not fr["bytes"].any()
        # and the ctx still works afterwards
        cfg = A.synth_default(seed=5, slot_len=700)
        x = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 150000)
        _check(d, oracle, x)


@pytest.mark.parametrize("noise_div", [18, 60, 200, 1020])
def test_ties_and_noise_levels(dem8, oracle, noise_div):
    # coarse noise -> many equal magnitudes: exercises `>=` in the gate and strict `>` in the slicer
    cfg = A.synth_default(seed=noise_div, slot_len=900, noise_div=noise_div)
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 300000)
    _check(dem8, oracle, iq)


def test_random_and_saturated(dem8, oracle):
    rng = np.random.default_rng(7)
    iq = rng.integers(-128, 128, size=(200000, 2)).astype(np.int8)
    _check(dem8, oracle, iq)
    iq = rng.choice(np.array([-128, -127, 126, 127], dtype=np.int8), size=(100000, 2))
    _check(dem8, oracle, iq)
    iq = rng.integers(-2, 3, size=(100000, 2)).astype(np.int8)
    _check(dem8, oracle, iq)


def test_error_mix(dem8, oracle):
    # 40 % data-bit flips (repaired), 30 % crc-bit flips and 30 % double flips (rejected or, rarely,
    # mis-repaired exactly as the reference would)
    cfg = A.synth_default(seed=77, slot_len=500, pct_flip_data=40, pct_flip_crc=30, pct_flip_two=30)
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 500000)
    fr = _check(dem8, oracle, iq)
    assert (fr["status"] == 1).sum() > 100
    # every planted data-bit flip is repaired to the clean frame at the planted offset
    by_off = {int(f["offset"]): f for f in fr}
    seen = 0
    for slot in range(500000 // 500 - 1):
        present, start, clean, sent, kind = A.synth_slot(cfg, 0, slot)
        if present and kind == 1 and start in by_off:
            assert bytes(by_off[start]["bytes"]) == clean and by_off[start]["status"] == 1
            seen += 1
    assert seen > 100


def test_large_buffer_i8(dem8, oracle):
    cfg = A.synth_default(seed=2024)
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 1 << 22)
    fr = _check(dem8, oracle, iq)
    assert len(fr) > 1500


def test_device_generator_matches_host(dem8, gpu):
    import torch
    cfg = A.synth_default(seed=99)
    n, first = 300001, 123457
    t = torch.empty(n * 2, dtype=torch.int8, device="cuda")
    dem8.synth_fill_device(cfg, 3, first, n, t.data_ptr())
    torch.cuda.synchronize()
    import ctypes
    got = t.cpu().numpy().reshape(n, 2)
    want = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 3, first, n)
    assert (got == want).all()


def test_multichannel_batch(gpu, oracle):
    import torch
    nch, n, stride = 5, 70001, 70008
    with A.AdsbDemod(max_samples=n, max_out=1 << 16, max_channels=nch, host_staging=False) as d:
        cfg = A.synth_default(seed=31, slot_len=800)
        host = np.zeros((nch, stride, 2), dtype=np.int8)
        for c in range(nch):
            host[c, :n] = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, c, 0, n)
        host[:, n:] = 77  # padding between channels must never be looked at
        t = torch.from_numpy(host).cuda()
        d.demod_device_async(t.data_ptr(), n, nch, stride)
        frames, counts, total, flags = d.fetch(n_channels=nch)
        pos = 0
        for c in range(nch):
            rc, want, cnt = oracle.process_buffer(host[c, :n])
            assert counts[c] == cnt
            _eq(frames[pos:pos + cnt], want)
            pos += cnt
        assert pos == len(frames) == total and flags == 0


@pytest.mark.parametrize("n", [240, 241, 1000, 20000, 32768 + 241, 150001])
def test_synthetic_sizes_i16(dem16, oracle, n):
    cfg = A.synth_default(seed=400 + n, slot_len=600, amp_shift=5)
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I16, 0, 0, n)
    _check(dem16, oracle, iq)


def test_i16_equals_widened_i8(dem8, dem16, oracle):
    cfg = A.synth_default(seed=8)
    iq8 = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 250000)
    a, _ = dem8.demod(iq8)
    b, _ = dem16.demod(iq8.astype(np.int16))
    _eq(a, b)


def test_i16_extremes(dem16, oracle):
    rng = np.random.default_rng(5)
    iq = rng.integers(-32768, 32768, size=(150000, 2)).astype(np.int16)
    _check(dem16, oracle, iq)
    iq = rng.choice(np.array([-32768, 32767, 0, 1], dtype=np.int16), size=(60000, 2))
    _check(dem16, oracle, iq)
    _check(dem16, oracle, np.full((3000, 2), -32768, dtype=np.int16))


def test_playback_pipeline_matches_reference_chunking(dem16, oracle):
    # config 1: 1 s of 2 MSPS CS16 through playback_thread -> process_sdr_data_thread -> stream text
    cfg = A.synth_default(seed=1, amp_shift=4)
    data = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I16, 0, 0, 2_000_000)
    with A.AdsbDemod(sample_type=A.ADSB_SAMPLE_I16, max_samples=20000, max_out=20000) as d:
        frames, n_buf, text = d.pipeline_playback(data, chunk_len=20000)
    chunks, want, n = oracle.playback(data, 20000)
    assert n_buf == chunks == 99  # the 100th buffer is never sent (adsb.rs:77)
    _eq(frames, want)
    assert len(frames) > 800
    expect_text = "".join("\n" + oracle.packet_display(bytes(f["bytes"]), "") + "\n" for f in want)
    assert text == expect_text


def test_64_channel_batch(gpu, oracle):
    # BASELINE config 4: 64 parallel channels batched in one launch, per-channel ordered lists
    import torch
    nch, n = 64, 40_008
    with A.AdsbDemod(max_samples=n, max_out=1 << 16, max_channels=nch, host_staging=False) as d:
        cfg = A.synth_default(seed=64, slot_len=900)
        host = np.stack([A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, c, 1000 * c, n) for c in range(nch)])
        t = torch.from_numpy(host).cuda()
        d.demod_device_async(t.data_ptr(), n, nch, n)
        frames, counts, total, flags = d.fetch(n_channels=nch)
        assert flags == 0 and sum(counts) == len(frames) == total
        pos = 0
        for c in range(nch):
            rc, want, cnt = oracle.process_buffer(host[c])
            assert counts[c] == cnt
            _eq(frames[pos:pos + cnt], want)
            pos += cnt
        assert total > 64 * 30


def test_large_streaming_buffer_sampled(gpu, oracle, scan_kind):
    # BASELINE config 3 regime (buffer much larger than the caches, > 4 GiB of offsets arithmetic):
    # 5 GiB of i8 IQ generated on the device, whole-buffer demod, parity on sampled sub-ranges via
    # the size-independent property that any sub-range demodulated alone gives the same frames
    # (every offset is independent).
    import torch
    if scan_kind != "root":
        pytest.skip("the 5 GiB buffer runs once, through the product's scan kernel")
    n = 5 * (1 << 29)  # 2.68 G samples = 5 GiB: byte offsets cross 2^32
    cfg = A.synth_default(seed=333)
    t = torch.empty(n * 2, dtype=torch.int8, device="cuda")
    with A.AdsbDemod(max_samples=n, max_out=1 << 21, host_staging=False) as d:
        d.synth_fill_device(cfg, 0, 0, n, t.data_ptr())
        d.demod_device_async(t.data_ptr(), n)
        frames, counts, total, flags = d.fetch()
    assert flags == 0 and total == len(frames)
    assert abs(total - n / 2000 * 0.944) < 0.02 * n / 2000  # ~94 % of the planted frames decode
    off = frames["offset"].astype(np.int64)
    assert (np.diff(off) > 0).all()
    for start in (0, (1 << 31) - 50_000, (1 << 31) + (1 << 28) + 12345, n - 300_000):
        assert start + 300_000 <= n
        m = 300_000
        part = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, start, m)
        rc, want, cnt = oracle.process_buffer(part)
        sel = frames[(off >= start) & (off < start + m - 240)].copy()
        sel["offset"] -= np.uint64(start)
        _eq(sel, want)
    del t
    torch.cuda.empty_cache()


def test_caller_owned_result_target(gpu, oracle):
    # adsb_set_result_target: the ordered list lands in caller memory as [32-byte header | frames]
    import torch
    n = 180_000
    cfg = A.synth_default(seed=17, slot_len=600)
    host = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, n)
    rc, want, cnt = oracle.process_buffer(host)
    t = torch.from_numpy(host).cuda()
    cap = 1000
    blob = torch.zeros(2, 32 + cap * 24, dtype=torch.uint8, device="cuda")
    with A.AdsbDemod(max_samples=n, max_out=4096, host_staging=False) as d:
        for k in range(2):  # two launches into two different slots, then back to the internal buffers
            d.set_result_target(blob[k].data_ptr(), blob[k].numel())
            d.demod_device_async(t.data_ptr(), n)
        d.set_result_target(None, 0)
        d.demod_device_async(t.data_ptr(), n)
        frames, counts, total, flags = d.fetch()
        _eq(frames, want)
    torch.cuda.synchronize()
    for k in range(2):
        raw = blob[k].cpu().numpy()
        hdr = raw[:32].view(np.uint64)
        assert hdr[0] == cnt and hdr[1] == cnt and hdr[2] == 0
        _eq(raw[32:32 + cnt * 24].view(A.FRAME_DTYPE), want)
    # a target smaller than the frame count truncates and says so
    small = torch.zeros(32 + 10 * 24, dtype=torch.uint8, device="cuda")
    with A.AdsbDemod(max_samples=n, max_out=4096, host_staging=False) as d:
        d.set_result_target(small.data_ptr(), small.numel())
        d.demod_device_async(t.data_ptr(), n)
        n_out, total, flags = d.fetch_counts()
        assert n_out == 10 and total == cnt and flags & A.ADSB_FLAG_TRUNCATED
    raw = small.cpu().numpy()
    assert raw[:32].view(np.uint64)[0] == 10
    _eq(raw[32:].view(A.FRAME_DTYPE), want[:10])


def test_carry_over_mode_recovers_boundary_frames(gpu, oracle):
    # SURVEY §8f-1 (switchable, not reference behaviour): with the last 240 samples carried into the
    # next buffer, the chunked stream decodes exactly like one long buffer of the samples that were sent
    cfg = A.synth_default(seed=5150, slot_len=700)
    data = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 400_000)
    chunk = 20000
    sent = (len(data) - 1) // chunk * chunk  # playback_thread never sends the last buffer (adsb.rs:77)
    with A.AdsbDemod(max_samples=chunk + 240, max_out=chunk + 240) as d:
        plain, nb = d.pipeline_playback(data, chunk_len=chunk, want_text=False)[:2]
        carried, nb2 = d.pipeline_playback_carry(data, chunk_len=chunk)
    rc, whole, n = oracle.process_buffer(data[:sent])
    _eq(carried, whole)
    assert nb == nb2 == sent // chunk
    assert len(carried) > len(plain)  # the reference loses the frames that straddle buffers
    lost = set(carried["offset"].tolist()) - set(plain["offset"].tolist())
    assert all((o % chunk) >= chunk - 240 for o in lost)


def test_on_device_field_decode(gpu, oracle):
    # AdsbPacket::new's field split on the GPU == the host mirror == the oracle, per frame
    cfg = A.synth_default(seed=808, slot_len=500)
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 600_000)
    with A.AdsbDemod(max_samples=len(iq), max_out=1 << 14) as d:
        frames, flags = d.demod(iq)
        fields = d.decode_fields()
    assert len(fields) == len(frames) > 800
    kinds = set()
    for f, g in zip(frames, fields):
        o = oracle.packet_new(bytes(f["bytes"]))
        kinds.add(o.msg_kind)
        assert (g["icao"], g["downlink_format"], g["capability"], g["msg_type"], g["msg_kind"]) == \
               (o.icao, o.downlink_format, o.capability, o.msg_type, o.msg_kind)
        assert (g["altitude"], g["cpr_latitude"], g["cpr_longitude"], g["surveillance_status"], g["nic_supplement"],
                g["cpr_time"], g["cpr_odd"]) == (o.altitude, o.cpr_latitude, o.cpr_longitude, o.surveillance_status,
                                                 o.nic_supplement, o.cpr_time, o.cpr_odd)
        assert g["callsign"] == o.callsign
    assert kinds == {0, 1, 2}
    # the reference's own whole-frame vectors, modulated, through the whole GPU path
    import json, os
    kats = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))["frames"]
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_frames_i8.npz"))
    with A.AdsbDemod(max_samples=len(z["iq"]), max_out=64) as d:
        frames, _ = d.demod(z["iq"])
        fields = d.decode_fields()
    for k, g in zip(kats, fields):
        assert f"{g['icao']:06X}" == k["icao"]
        for name in ("altitude", "cpr_latitude", "cpr_longitude", "msg_type", "capability", "downlink_format"):
            if name in k:
                assert g[name] == k[name], name
        if "callsign" in k:
            assert g["callsign"].decode() == k["callsign"]


# ---- the nsq gate: ties after truncation, decided on n = I^2 + Q^2 ------------------------------------------------
def _gate_buf(hi_iq, lo_iq, df_hi=None, df_lo=None, n=241):
    """241 samples: preamble highs / lows as given (I, Q) pairs, DF17 highs / lows (default: all equal), then zeros."""
    buf = np.zeros((n, 2), dtype=np.int8)
    buf[[0, 2, 7, 9]] = hi_iq
    buf[[1, 3, 4, 5, 6, 8, 10, 11, 12, 13, 14, 15]] = lo_iq
    if df_hi is not None:
        buf[[16 + k for k in (0, 3, 5, 7, 8)]] = df_hi
        buf[[16 + k for k in (1, 2, 4, 6, 9)]] = df_lo
    return buf


def test_gate_ties_after_truncation(dem8, oracle):
    """The reference compares floor(sqrt(I^2+Q^2)) (utils.rs:46-52, demod.rs:27-36, 48-54): a "high" with a SMALLER
    n than a "low" still passes when their truncated roots are equal.  Sums of two squares around every root
    boundary an i8 sample can reach: (high n, low n) with high < low inside one root class (passes), across a class
    boundary (fails), and equal / reversed; in the preamble group and in the DF17 group; including the nine values
    with |I|, |Q| >= 125 that are no ordered f16 patterns (the tile then takes the integer gate)."""
    # representable n -> one (I, Q)
    rep = {}
    for i in range(0, 129):
        for q in range(i, 129):
            rep.setdefault(i * i + q * q, (-i if i == 128 else i, -q if q == 128 else q))
    ns = np.array(sorted(rep))
    roots = np.floor(np.sqrt(ns)).astype(int)
    cases = []
    rng = np.random.default_rng(3)
    for r in list(range(0, 182)):
        cls = ns[roots == r]
        if len(cls) >= 2:
            cases.append((int(cls[0]), int(cls[-1]), True))    # smallest high, largest low of one class: tie, passes
            cases.append((int(cls[-1]), int(cls[0]), True))
        nxt = ns[roots == r + 1]
        if len(cls) and len(nxt):
            cases.append((int(cls[-1]), int(nxt[0]), False))   # adjacent classes: high root < low root
            cases.append((int(nxt[0]), int(cls[-1]), True))
    cases = [cases[k] for k in rng.permutation(len(cases))]
    n_pass = 0
    for hi_n, lo_n, expect in cases:
        for where in ("preamble", "df17"):
            if where == "preamble":
                buf = _gate_buf(rep[hi_n], rep[lo_n])
                buf[16:] = 0
                ok_rest = True  # DF17 region all zero: ties pass
            else:
                buf = _gate_buf((90, 0), (10, 0), rep[hi_n], rep[lo_n])
                ok_rest = True
            frames, flags = dem8.demod(buf)
            rc, want, n = oracle.process_buffer(buf)
            assert rc == 0 and flags == 0
            _eq(frames, want)
            # the gate verdict itself (a passing gate need not yield a frame: the CRC decides)
            m = oracle.get_magnitude(buf.astype(np.int16))
            assert (oracle.check_for_adsb_packet(m[:32]) is not None) == (expect and ok_rest), (hi_n, lo_n, where)
            n_pass += len(want)
    assert n_pass > 50


def test_gate_band_random_windows(dem8, oracle):
    """Dense random windows drawn so that highs and lows sit within a few units of each other at many amplitude
    levels (the band the nsq gate has to resolve with roots), and pairs of the slicer within one root class: every
    buffer against the oracle."""
    rng = np.random.default_rng(99)
    for level in (0, 1, 2, 3, 5, 8, 13, 20, 35, 60, 90, 110, 124, 126):
        spread = max(1, level // 6)
        iq = rng.integers(max(-128, level - spread - 1), min(127, level + spread + 1) + 1, size=(60_000, 2)).astype(np.int8)
        sign = rng.choice(np.array([-1, 1], dtype=np.int8), size=iq.shape)
        iq = (iq.astype(np.int16) * sign).clip(-128, 127).astype(np.int8)
        frames, flags = dem8.demod(iq)
        rc, want, n = oracle.process_buffer(iq)
        assert rc == 0 and flags == 0
        _eq(frames, want)


def test_full_scale_samples_in_some_tiles(dem8, oracle):
    """A realistic stream with |I|, |Q| >= 125 samples (n + 72 >= 0x7C00: not an ordered f16 pattern) dropped into
    tiles 1, 4 and 7 only: those tiles run the integer gate, the others the three-input f16 gate, in one launch."""
    rng = np.random.default_rng(21)
    cfg = A.synth_default(seed=271, slot_len=700)
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 300_000)
    frames, flags = dem8.demod(iq)
    rc, want, n_clean = oracle.process_buffer(iq)
    _eq(frames, want)
    for t in (1, 4, 7):
        pos = rng.integers(t * 16384, (t + 1) * 16384, size=40)
        iq[pos] = rng.choice(np.array([-128, -127, -126, -125, 125, 126, 127], dtype=np.int8), size=(40, 2))
    frames, flags = dem8.demod(iq)
    rc, want, n = oracle.process_buffer(iq)
    assert rc == 0 and flags == 0 and n > 0.9 * n_clean
    _eq(frames, want)
    # and a whole buffer of nothing but such samples (every tile in integer mode; many ties)
    iq = rng.choice(np.array([-128, -127, -126, -125, 125, 126, 127], dtype=np.int8), size=(120_000, 2))
    frames, flags = dem8.demod(iq)
    rc, want, n = oracle.process_buffer(iq)
    _eq(frames, want)


def test_zero_tile_launch_clears_flags(gpu):
    """a 240-sample launch (zero offsets, adsb.rs:98 iterates 0..0) runs no scan kernel; its result set's
    flag words must not keep TRUNCATED from the launch two before it -- in the ctx header and in a caller-owned blob."""
    import torch
    with A.AdsbDemod(max_samples=5000, max_out=100, host_staging=False) as d:
        zeros = torch.zeros(5000 * 2, dtype=torch.int8, device="cuda")
        blob = torch.full((32 + 100 * 24,), 0xEE, dtype=torch.uint8, device="cuda")
        for use_blob in (False, True):
            d.set_result_target(blob.data_ptr(), blob.numel()) if use_blob else d.set_result_target(None, 0)
            d.demod_device_async(zeros.data_ptr(), 5000)          # 4760 all-zero frames: truncated to 100
            assert d.fetch_counts()[2] & A.ADSB_FLAG_TRUNCATED
            d.demod_device_async(zeros.data_ptr(), 5000)          # the other result set
            d.demod_device_async(zeros.data_ptr(), 240)           # same set as the first launch, no tiles
            n_out, total, flags = d.fetch_counts()
            assert (n_out, total, flags) == (0, 0, 0)
            if use_blob:
                torch.cuda.synchronize()
                hdr = blob[:32].cpu().numpy().view(np.uint64)
                assert tuple(hdr[:3]) == (0, 0, 0)
        d.set_result_target(None, 0)
