"""Parity of the HIP path (through the C ABI) with the CPU oracle.  Bit-exact: frames, offsets,
status and repaired-bit index must all be identical (integer/byte work, no tolerance)."""
import os

import numpy as np
import pytest

import air_rs_amd as A

pytestmark = pytest.mark.gpu


def _kernel_kinds():
    # The product build has one kernel (demod_tiles).  A library built with the experimental streaming kernel
    # (tools/build_variant.sh stream -DADSB_WITH_STREAM_KERNEL=1 -Itools/experimental, used via ADSB_HIP_LIB)
    # runs the whole module a second time with ADSB_KERNEL=stream when ADSB_TEST_STREAM_KERNEL=1 is set.
    return ["tiles", "stream"] if os.environ.get("ADSB_TEST_STREAM_KERNEL") == "1" else ["tiles"]


@pytest.fixture(scope="module", params=_kernel_kinds(), autouse=True)
def kernel_kind(request):
    """Every test of this module runs once per i8 kernel under test: demod_tiles (the default and only kernel
    of the product build), and the experimental streaming kernel when asked for (see _kernel_kinds)."""
    old = os.environ.get("ADSB_KERNEL")
    os.environ["ADSB_KERNEL"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("ADSB_KERNEL", None)
    else:
        os.environ["ADSB_KERNEL"] = old


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
def dem8(gpu, kernel_kind):
    with A.AdsbDemod(sample_type=A.ADSB_SAMPLE_I8, max_samples=1 << 22, max_out=1 << 18) as d:
        assert d.kernel == kernel_kind
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


def test_magnitude_table_exhaustive(dem8, oracle, kernel_kind):
    # the streaming kernel's 64 KB table: floor(sqrt(I^2+Q^2)) for every raw sample (Q << 8) | I
    if kernel_kind != "stream":
        pytest.skip("the tile kernel computes magnitudes arithmetically (test above)")
    table = dem8.magnitude_table()
    r = np.arange(65536, dtype=np.uint32)
    iq = np.stack([(r & 0xFF).astype(np.uint8).view(np.int8), (r >> 8).astype(np.uint8).view(np.int8)], axis=1)
    want = oracle.get_magnitude(iq.astype(np.int16))
    assert (table == want).all(), np.nonzero(table != want)[0][:10]


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


@pytest.mark.parametrize("n", [240, 241, 255, 271, 272, 1000, 20000, 32768 + 239, 32768 + 240, 32768 + 241,
                               65536 + 240, 100003])
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


def test_large_streaming_buffer_sampled(gpu, oracle):
    # BASELINE config 3 regime (buffer much larger than the caches, > 4 GiB of offsets arithmetic):
    # 5 GiB of i8 IQ generated on the device, whole-buffer demod, parity on sampled sub-ranges via
    # the size-independent property that any sub-range demodulated alone gives the same frames
    # (every offset is independent).
    import torch
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


@pytest.mark.parametrize("grid", [1, 3, 7])
def test_stream_kernel_many_rounds_per_workgroup(gpu, oracle, kernel_kind, grid):
    """The streaming kernel's persistent workgroups walk tiles b, b + G, b + 2G, ...: with the default grid
    (one workgroup per CU) the buffers of this suite give every workgroup a single tile, so pin the grid to a
    few workgroups and make each run many rounds (double-buffered magnitudes, parity-buffered lists, deferred
    records) -- sparse rounds, dense rounds (constant stretch: a frame per offset) and the ragged last tile."""
    if kernel_kind != "stream":
        pytest.skip("streaming kernel only")
    os.environ["ADSB_STREAM_GRID"] = str(grid)
    try:
        cfg = A.synth_default(seed=4000 + grid, slot_len=700)
        n = 32768 * 21 + 12345
        iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, n)
        iq[32768 * 5 + 100:32768 * 5 + 2100] = 17        # constant stretch: > 64 survivors in tile 5 (dense round)
        iq[32768 * 6 - 50:32768 * 6 + 400] = -3          # ... and one straddling a tile edge
        with A.AdsbDemod(sample_type=A.ADSB_SAMPLE_I8, max_samples=n, max_out=1 << 16) as d:
            assert d.kernel == "stream"
            fr = _check(d, oracle, iq)
            assert len(fr) > 2500
            fr2 = _check(d, oracle, iq[: 32768 * 9 + 241])   # same context, shorter buffer
            assert 0 < len(fr2) < len(fr)
    finally:
        os.environ.pop("ADSB_STREAM_GRID", None)
