"""SURVEY 8f-1 / 8f-4 on the GPU: the streaming front end (adsb_feed_*: pinned ring, asynchronous DMA, two buffers in
flight, the 240-sample tail carried on the device) and the replay entry (file -> playback chunking -> GPU thread 2 ->
stream-mode text), against the CPU oracle.  Parity mode must equal the reference's per-buffer semantics; carry mode
must equal one long buffer of the same samples."""
import os

import numpy as np
import pytest

import air_rs_amd as A

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["small-path", "three-kernels"], autouse=True)
def small_path(request):
    """Every test of this module runs twice: with the one-dispatch path for small buffers (adsbk::demod_small: buffers of
    up to 32 tiles go through ONE kernel that reads the pinned ring and writes the frames into pinned memory) and with
    it switched off (ADSB_SMALL_PATH=0 at adsb_create: copy + scan + finish + result copies, as for large buffers)."""
    old = os.environ.get("ADSB_SMALL_PATH")
    os.environ["ADSB_SMALL_PATH"] = "1" if request.param == "small-path" else "0"
    yield request.param
    if old is None:
        os.environ.pop("ADSB_SMALL_PATH", None)
    else:
        os.environ["ADSB_SMALL_PATH"] = old


def _eq(got, want):
    assert len(got) == len(want), (len(got), len(want))
    if len(got):
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (bad[:5], got[bad[:3]], want[bad[:3]])


def _ragged_cuts(n, rng, lo, hi):
    cuts, pos = [], 0
    while pos < n:
        step = int(rng.integers(lo, hi))
        cuts.append((pos, min(pos + step, n)))
        pos += step
    return cuts


@pytest.mark.parametrize("st", [A.ADSB_SAMPLE_I8, A.ADSB_SAMPLE_I16])
def test_feed_parity_mode_is_per_buffer(gpu, oracle, st):
    """carry = 0: every pushed buffer is its own reference buffer (adsb.rs:95-98): frames and buffer-relative offsets
    equal the oracle's on that buffer alone; buffers of ragged length (a live SDR sends MTU-sized reads), two in flight."""
    cfg = A.synth_default(seed=41, slot_len=600)
    if st == A.ADSB_SAMPLE_I16:
        cfg.amp_shift = 5
    data = A.synth_fill_host(cfg, st, 0, 0, 900_000)
    rng = np.random.default_rng(3)
    cuts = _ragged_cuts(len(data), rng, 240, 70_000)
    with A.AdsbDemod(sample_type=st, max_samples=70_000 + 240, max_out=1 << 15, host_staging=False) as d:
        with A.Feed(d, max_chunk=70_000, carry=False) as f:
            got = []
            for (a, b) in cuts:
                f.push(data[a:b])
                if f.in_flight == 2:
                    got.append(f.pop())
            while f.in_flight:
                got.append(f.pop())
            with pytest.raises(A.AdsbError) as e:       # nothing in flight
                f.pop()
            assert e.value.code == A.ADSB_E_STATE
            with pytest.raises(A.AdsbError) as e:       # the reference panics below 240 samples (adsb.rs:98)
                f.push(data[:100])
            assert e.value.code == A.ADSB_E_SHORT and f.in_flight == 0
    assert len(got) == len(cuts)
    total = 0
    for (a, b), (frames, flags, first) in zip(cuts, got):
        rc, want, n = oracle.process_buffer(data[a:b])
        assert rc == 0 and flags == 0 and first == a
        _eq(frames, want)
        total += n
    assert total > 1000


@pytest.mark.parametrize("st,lo,hi", [(A.ADSB_SAMPLE_I8, 1, 50_000), (A.ADSB_SAMPLE_I16, 3000, 40_000), (A.ADSB_SAMPLE_I8, 1, 600)])
def test_feed_carry_mode_equals_one_long_buffer(gpu, oracle, st, lo, hi):
    """carry = 1: the chunked stream decodes exactly like one long buffer (absolute offsets), whatever the cuts -- incl.
    buffers shorter than 240 samples (nothing decodable yet / tiny tails) and frames straddling several buffers."""
    cfg = A.synth_default(seed=43, slot_len=500)
    if st == A.ADSB_SAMPLE_I16:
        cfg.amp_shift = 4
    n = 600_000 if hi > 1000 else 60_000
    data = A.synth_fill_host(cfg, st, 0, 0, n)
    rng = np.random.default_rng(hi)
    cuts = _ragged_cuts(n, rng, lo, hi)
    with A.AdsbDemod(sample_type=st, max_samples=hi + 240, max_out=1 << 15, host_staging=False) as d:
        with A.Feed(d, max_chunk=hi, carry=True) as f:
            parts = []
            for (a, b) in cuts:
                f.push(data[a:b])
                if f.in_flight == 2:
                    parts.append(f.pop()[0])
            while f.in_flight:
                parts.append(f.pop()[0])
        assert d.demod is not None
    merged = np.concatenate(parts)
    rc, want, cnt = oracle.process_buffer(data)
    _eq(merged, want)
    assert cnt > 50


def test_feed_zero_copy_producer_and_backpressure(gpu, oracle):
    """adsb_feed_acquire: the producer writes into the pinned ring itself; a third push with two buffers in flight is
    refused (ADSB_E_STATE) and changes nothing."""
    cfg = A.synth_default(seed=45, slot_len=700)
    chunk = 20_000
    data = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 10 * chunk)
    with A.AdsbDemod(max_samples=chunk + 240, max_out=chunk, host_staging=False) as d:
        with A.Feed(d, max_chunk=chunk, carry=True, ring_slots=2) as f:
            parts = []
            for k in range(10):
                slot = f.acquire()
                slot[:chunk] = data[k * chunk:(k + 1) * chunk]
                f.push_acquired(chunk)
                if f.in_flight == 2:
                    with pytest.raises(A.AdsbError) as e:
                        f.push(data[:chunk])
                    assert e.value.code == A.ADSB_E_STATE and f.in_flight == 2
                    parts.append(f.pop()[0])
            while f.in_flight:
                parts.append(f.pop()[0])
    rc, want, cnt = oracle.process_buffer(data)
    _eq(np.concatenate(parts), want)


def test_pipeline_send_tail_switch(gpu, oracle):
    """adsb.rs:77 never sends the last chunk; ADSB_REPLAY_SEND_TAIL does.  Per-buffer mode + tail = the oracle on every
    chunk incl. the last, partial one; carry + tail = the oracle on the whole file as one buffer."""
    cfg = A.synth_default(seed=5150, slot_len=700)
    chunk = 20_000
    data = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 7 * chunk + 12_345)
    with A.AdsbDemod(max_samples=chunk + 240, max_out=chunk + 240, host_staging=False) as d:
        ref, nb_ref, _ = d.pipeline_run(data, chunk, want_text=False)
        tail, nb_tail, _ = d.pipeline_run(data, chunk, send_tail=True, want_text=False)
        both, nb_both, _ = d.pipeline_run(data, chunk, carry=True, send_tail=True, want_text=False)
    assert nb_ref == 7 and nb_tail == nb_both == 8
    want = []
    for k in range(8):
        rc, w, n = oracle.process_buffer(data[k * chunk:(k + 1) * chunk])
        w["offset"] += np.uint64(k * chunk)
        want.append(w)
    _eq(ref, np.concatenate(want[:7]))
    _eq(tail, np.concatenate(want))
    rc, whole, n = oracle.process_buffer(data)
    _eq(both, whole)
    assert len(both) >= len(tail) > len(ref)


def test_replay_c16_file_text_equals_reference_playback(gpu, oracle, tmp_path):
    """a `.c16` file (utils.rs:6-43) -> playback chunking -> GPU thread 2 -> stream-mode text, byte for
    byte what the oracle's restatement of `air_rs adsb -p FILE -m stream` prints (Processed Time blanked)."""
    cfg = A.synth_default(seed=49, amp_shift=4, slot_len=1500)
    data = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I16, 0, 0, 1_000_000)
    path = tmp_path / "capture.c16"
    data.astype("<i2").tofile(path)                       # raw little-endian i16 I,Q pairs, no header
    assert os.path.getsize(path) == 4 * len(data)
    with A.AdsbDemod(sample_type=A.ADSB_SAMPLE_I16, max_samples=20_240, max_out=20_240, host_staging=False) as d:
        frames, n_buf, n_samp, text = d.replay_file(str(path))
        carried = d.replay_file(str(path), carry=True, send_tail=True)[0]
    chunks, want, n = oracle.playback(data, 20000)
    assert n_samp == len(data) and n_buf == chunks == 49
    _eq(frames, want)
    expect = "".join("\n" + oracle.packet_display(bytes(f["bytes"]), "") + "\n" for f in want)
    assert text == expect and len(want) > 400
    rc, whole, cnt = oracle.process_buffer(data)
    _eq(carried, whole)
    # a file that is not a whole number of samples is refused like utils.rs:28-30
    bad = tmp_path / "bad.c16"
    bad.write_bytes(b"\x00" * 6)
    with A.AdsbDemod(sample_type=A.ADSB_SAMPLE_I16, max_samples=20_240, max_out=64, host_staging=False) as d:
        with pytest.raises(A.AdsbError) as e:
            d.replay_file(str(bad))
        assert e.value.code == A.ADSB_E_ARG


def test_replay_rtlsdr_u8_file(gpu, oracle, tmp_path):
    """Raw rtl_sdr capture (unsigned bytes, zero level 128): re-centred as x - 128 and replayed on an i8 context."""
    cfg = A.synth_default(seed=51, slot_len=900)
    data = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 300_000)
    path = tmp_path / "capture.bin"
    (data.astype(np.int16) + 128).astype(np.uint8).tofile(path)
    with A.AdsbDemod(max_samples=20_240, max_out=20_240, host_staging=False) as d:
        frames, n_buf, n_samp, text = d.replay_file(str(path))
    want = []
    for k in range(14):                                   # 15 chunks of 20 000; the last is never sent (adsb.rs:77)
        rc, w, n = oracle.process_buffer(data[k * 20000:(k + 1) * 20000])
        w["offset"] += np.uint64(k * 20000)
        want.append(w)
    want = np.concatenate(want)
    assert n_buf == 14 and n_samp == 300_000
    _eq(frames, want)
    assert text.count("== ") == len(want)


@pytest.mark.parametrize("carry", [False, True])
def test_feed_slot_pool_overflow_with_two_launches_in_flight(gpu, oracle, carry):
    """the feed's slow path -- a launch whose tiles lost their slots (every offset of a constant stretch is
    a survivor; the pool is switched off so that it happens deterministically) is re-planned while a
    NEWER launch is in flight: adsb_feed_pop views the older launch, re-runs it and restores the newest.  Constant
    buffers make every pop take that path back to back; real-looking buffers in between must not be disturbed.
    Also: a context with a caller-owned result target cannot open a feed (two launches would share the blob)."""
    import torch
    chunk = 40_000
    cfg = A.synth_default(seed=515, slot_len=700)
    live = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 3 * chunk)
    zeros = np.zeros((chunk, 2), dtype=np.int8)
    sevens = np.full((chunk, 2), 7, dtype=np.int8)
    bufs = [zeros, live[:chunk], sevens, zeros, live[chunk:2 * chunk], live[2 * chunk:], sevens]
    with A.AdsbDemod(max_samples=chunk + 240, max_out=chunk + 240, host_staging=False) as d:
        blob = torch.zeros(32 + 24 * 16, dtype=torch.uint8, device="cuda")
        d.set_result_target(blob.data_ptr(), blob.numel())
        with pytest.raises(A.AdsbError) as e:
            A.Feed(d, max_chunk=chunk, carry=carry)
        assert e.value.code == A.ADSB_E_STATE
        d.set_result_target(None, 0)
        d.pool_limit(True)
        got = []
        with A.Feed(d, max_chunk=chunk, carry=carry) as f:
            for b in bufs:
                f.push(b)
                if f.in_flight == 2:
                    got.append(f.pop())
            while f.in_flight:
                got.append(f.pop())
        d.pool_limit(False)
    assert len(got) == len(bufs)
    if carry:
        whole = np.concatenate(bufs)
        rc, want, n = oracle.process_buffer(whole, max_out=1 << 19)
        assert rc == 0
        merged = np.concatenate([fr for fr, _, _ in got])
        assert all(fl == 0 for _, fl, _ in got)
        _eq(merged, want)
    else:
        pos = 0
        for b, (fr, fl, first) in zip(bufs, got):
            rc, want, n = oracle.process_buffer(b, max_out=1 << 17)
            assert rc == 0 and fl == 0 and first == pos
            _eq(fr, want)
            pos += len(b)
