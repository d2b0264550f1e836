"""GPU-tier tests added in round 2 (all through the C ABI, all against the CPU oracle or the reference's own KATs):
the reference's gate KAT as IQ, the time-shard plan run through the HIP path on one device, the zero-copy
consumer's INCOMPLETE flag, bench.py's RCCL gather path at world size 1, and BASELINE config 3 at full size."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import air_rs_amd as A
from air_rs_amd import sharding

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _eq(got, want):
    assert len(got) == len(want), (len(got), len(want))
    if len(got):
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (bad[:5], got[bad[:3]], want[bad[:3]])


# ---- the reference's own gate KAT (src/adsb/demod.rs:250-278) through the HIP path --------------------------
@pytest.mark.parametrize("st,hi,lo", [(A.ADSB_SAMPLE_I16, 1000, 500), (A.ADSB_SAMPLE_I16, (600, 800), (300, 400)),
                                       (A.ADSB_SAMPLE_I8, 100, 50)])
def test_reference_gate_kat_through_hip(gpu, oracle, st, hi, lo):
    """test_check_for_adsb_packet_valid: highs 1000 at {0,2,7,9}, lows 500, everything after sample 15 zero.
    As IQ with exactly those magnitudes ((1000,0), (600,800), ...) in a 241-sample buffer -- one offset exists --
    both gates pass (ties pass in the DF17 region), the slicer reads 112 zero bits, CRC(0) = 0: one all-zero frame
    at offset 0.  test_check_for_adsb_packet_invalid (highs 500, lows 1000): nothing."""
    dt = np.int16 if st == A.ADSB_SAMPLE_I16 else np.int8
    iqv = lambda v: v if isinstance(v, tuple) else (v, 0)
    highs, lows = [0, 2, 7, 9], [1, 3, 4, 5, 6, 8, 10, 11, 12, 13, 14, 15]
    with A.AdsbDemod(sample_type=st, max_samples=4096, max_out=64) as d:
        for valid in (True, False):
            iq = np.zeros((241, 2), dtype=dt)
            iq[highs] = iqv(hi if valid else lo)
            iq[lows] = iqv(lo if valid else hi)
            mags = d.magnitudes(iq[:16])
            want_hi = 1000 if st == A.ADSB_SAMPLE_I16 else 100
            assert set(mags[highs]) == {want_hi if valid else want_hi // 2}
            frames, flags = d.demod(iq)
            rc, want, n = oracle.process_buffer(iq.astype(np.int16))
            assert rc == 0 and flags == 0
            _eq(frames, want)
            if valid:
                assert len(frames) == 1 and frames[0]["offset"] == 0 and bytes(frames[0]["bytes"]) == bytes(14)
                assert frames[0]["status"] == 0
            else:
                assert len(frames) == 0


# ---- multi-GPU readiness on one device: the shard plan through the HIP path ---------------------------------
@pytest.mark.parametrize("st,total,world", [(A.ADSB_SAMPLE_I8, 3_000_017, 8), (A.ADSB_SAMPLE_I8, 700_001, 3),
                                            (A.ADSB_SAMPLE_I16, 1_200_003, 8)])
def test_shard_plan_through_hip_equals_single_buffer(gpu, oracle, st, total, world):
    """sharding.plan(total, world) run rank after rank on ONE device, each rank with adsb_set_stream_base(first
    sample of its slice): the per-rank lists, merely concatenated, are the single-buffer HIP list and the oracle's."""
    cfg = A.synth_default(seed=77, slot_len=900)
    if st == A.ADSB_SAMPLE_I16:
        cfg.amp_shift = 5
    whole = A.synth_fill_host(cfg, st, 0, 0, total)
    with A.AdsbDemod(sample_type=st, max_samples=total, max_out=1 << 16) as d:
        single, flags = d.demod(whole)
        assert flags == 0
        parts = []
        for sh in sharding.plan(total, world):
            if not sh.n_samples:
                continue
            d.set_stream_base(sh.first_sample)
            fr, fl = d.demod(whole[sh.first_sample: sh.first_sample + sh.n_samples])
            assert fl == 0
            assert len(fr) == 0 or (fr["offset"][0] >= sh.first_offset and fr["offset"][-1] < sh.first_offset + sh.n_offsets)
            parts.append(fr)
        d.set_stream_base(0)
    merged = np.concatenate(parts)
    assert (np.diff(merged["offset"].astype(np.int64)) > 0).all()
    _eq(merged, single)
    rc, want, n = oracle.process_buffer(whole)
    _eq(single, want)
    assert len(want) > 500


# ---- the zero-copy path must not hand out a list with holes silently --------------------------------
def test_zero_copy_consumer_sees_incomplete_flag_and_repair(gpu, oracle):
    """All-zero input: every offset is a valid all-zero frame, so every tile has 16384 survivors and asks
    the shared pool for slots.  With the pool switched off (adsb_debug_pool_limit: deterministic, no race for the
    pool) every tile loses its slots, the wanted frames are missing after the first pass, and a device-side consumer
    (adsb_set_result_target blob) must see ADSB_FLAG_INCOMPLETE in the blob header; adsb_fetch_counts() re-plans
    (the knob is off again by then), completes the blob IN PLACE and clears the flag."""
    import torch
    n, cap = 1 << 20, 1000
    iq = np.zeros((n, 2), dtype=np.int8)
    rc, want, cnt = oracle.process_buffer(iq[:4096], max_out=cap)   # the first `cap` offsets, all-zero frames
    assert len(want) == cap
    with A.AdsbDemod(max_samples=n, max_out=cap, host_staging=False) as d:
        t = torch.from_numpy(iq).cuda()
        blob = torch.full((sharding.payload_bytes(cap),), 0xEE, dtype=torch.uint8, device="cuda")
        side = torch.cuda.Stream()
        for attempt in range(2):
            blob.fill_(0xEE)
            torch.cuda.synchronize()
            d.set_result_target(blob.data_ptr(), blob.numel())
            d.pool_limit(True)
            d.demod_device_async(t.data_ptr(), n)
            d.pool_limit(False)
            d.stream_wait_results(side.cuda_stream)
            side.synchronize()
            n_out, total, flags, frames = sharding.parse_payload(blob.cpu().numpy())
            assert n_out == cap and total == n - 240 and flags & A.ADSB_FLAG_TRUNCATED
            assert flags & A.ADSB_FLAG_INCOMPLETE, "every tile lost its slots: the blob must say so"
            n2, total2, flags2 = d.fetch_counts()       # the host entry point re-plans what is missing ...
            assert (n2, total2) == (cap, n - 240)
            assert not (flags2 & A.ADSB_FLAG_INCOMPLETE) and flags2 & A.ADSB_FLAG_TRUNCATED
            n_out, total, flags, frames = sharding.parse_payload(blob.cpu().numpy())
            assert not (flags & A.ADSB_FLAG_INCOMPLETE) and flags & A.ADSB_FLAG_TRUNCATED
            _eq(frames.copy(), want)                    # ... and the blob is whole, in place
        d.set_result_target(None, 0)
        # the knob left no trace: a plain launch gives a whole list straight away
        d.demod_device_async(t.data_ptr(), n)
        frames, counts, total, flags = d.fetch()
        assert not (flags & A.ADSB_FLAG_INCOMPLETE)
        _eq(frames, want)


def test_small_buffer_path_equals_three_kernel_path(gpu, oracle, monkeypatch):
    """adsb_demod() on buffers of up to 32 tiles takes ONE dispatch (scan + finish + ordered list into pinned memory);
    ADSB_SMALL_PATH=0 keeps the copy + two kernels + fetch.  Same frames either way, on ragged sizes around the tile
    and 32-tile edges, for i8 (both scan kernels) and CS16, incl. dense input (every offset a frame) and a capacity
    smaller than the frame count."""
    # (tile = 16384 offsets for i8, 8192 for CS16: the 32-tile edge of the one-dispatch path lies at both)
    sizes = [241, 1000, 8192 + 240, 16128 + 240, 16128 + 241, 16384 + 240, 16384 + 241, 20000, 3 * 16384 + 777, 32 * 8192 + 240,
             32 * 8192 + 241, 32 * 16128 + 240, 32 * 16128 + 241, 32 * 16384 + 240, 32 * 16384 + 241, 600_000]
    for st, scan in ((A.ADSB_SAMPLE_I8, "root"), (A.ADSB_SAMPLE_I16, "root")):
        monkeypatch.setenv("ADSB_SCAN", scan)
        cfg = A.synth_default(seed=61, slot_len=500)
        if st == A.ADSB_SAMPLE_I16:
            cfg.amp_shift = 5
        data = A.synth_fill_host(cfg, st, 0, 0, max(sizes))
        got = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("ADSB_SMALL_PATH", mode)
            with A.AdsbDemod(sample_type=st, max_samples=max(sizes), max_out=1 << 15) as d:
                got[mode] = [d.demod(data[:n]) for n in sizes]
                dense = d.demod(np.zeros((50_000, 2), dtype=data.dtype))          # 49 760 frames > max_out: truncated
                assert dense[1] & A.ADSB_FLAG_TRUNCATED and len(dense[0]) == 1 << 15
                assert (dense[0]["offset"] == np.arange(1 << 15)).all() and not dense[0]["bytes"].any()
        for n, (a, fa), (b, fb) in zip(sizes, got["1"], got["0"]):
            rc, want, cnt = oracle.process_buffer(data[:n])
            assert fa == fb == 0 and rc == 0
            _eq(a, want)
            _eq(b, want)


def test_scan_kernel_selection(gpu, monkeypatch):
    """ADSB_SCAN at adsb_create: default and "root" = the product's i8 scan kernel; the A/B kernels ("code", "nsq", "reg") are
    not in the product library and asking for one is an error, like any unknown name; CS16 has one kernel."""
    monkeypatch.delenv("ADSB_SCAN", raising=False)
    with A.AdsbDemod(max_samples=4096, max_out=16) as d:
        assert d.scan == "root"
    monkeypatch.setenv("ADSB_SCAN", "root")
    with A.AdsbDemod(max_samples=4096, max_out=16) as d:
        assert d.scan == "root"
    for name in ("code", "nsq", "reg", "stream"):
        monkeypatch.setenv("ADSB_SCAN", name)
        with pytest.raises(A.AdsbError) as e:
            A.AdsbDemod(max_samples=4096, max_out=16)
        assert e.value.code == A.ADSB_E_ARG
    monkeypatch.setenv("ADSB_SCAN", "root")
    with A.AdsbDemod(sample_type=A.ADSB_SAMPLE_I16, max_samples=4096, max_out=16) as d:
        assert d.scan == "root"


def test_fetch_counts_array_is_sized_for_the_launch(gpu, oracle):
    """fetch() after a 7-channel launch, called with the default arguments, must not write past its array."""
    import torch
    nch, n = 7, 30_001
    stride = (n + 7) & ~7
    with A.AdsbDemod(max_samples=n, max_out=1 << 14, max_channels=nch, host_staging=False) as d:
        cfg = A.synth_default(seed=3, slot_len=500)
        host = np.zeros((nch, stride, 2), dtype=np.int8)
        for c in range(nch):
            host[c, :n] = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, c, 0, n)
        t = torch.from_numpy(host).cuda()
        d.demod_device_async(t.data_ptr(), n, nch, stride)
        frames, counts, total, flags = d.fetch()
        assert len(counts) == nch and sum(counts) == len(frames) == total
        pos = 0
        for c in range(nch):
            rc, want, cnt = oracle.process_buffer(host[c, :n])
            assert counts[c] == cnt
            _eq(frames[pos:pos + cnt], want)
            pos += cnt


def test_pipeline_rejects_mismatched_sample_type(gpu):
    with A.AdsbDemod(sample_type=A.ADSB_SAMPLE_I16, max_samples=1 << 16, max_out=1 << 10) as d:
        data = np.zeros((50_000, 2), dtype=np.int8)
        import ctypes as C
        lib = A.load()
        nf, nb, tl = C.c_size_t(), C.c_uint64(), C.c_size_t()
        rc = lib.adsb_pipeline_playback(d.handle, A.ADSB_SAMPLE_I8, data.ctypes.data, len(data), 20000, None, 0,
                                        C.byref(nf), C.byref(nb), None, 0, C.byref(tl))
        assert rc == A.ADSB_E_ARG


# ---- bench.py's multi-rank code on RCCL (world size 1: the only size one GPU can run) ---------------------------
def test_bench_gather_path_world1(gpu):
    """bench.py --force-gather: BucketGather over an nccl (RCCL) group of one rank, 11 steps with a bucket of 8
    (one full bucket + a partial one); rank 0's checks (global order, planted frames) are part of the JSON line."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), RANK="0",
               LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-gather", "--steps", "11",
                        "--warmup", "3", "--samples", str(1 << 24), "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    g = d["gather_check"]
    assert g["ok"], g
    assert g["globally_ordered"] and g["spot_checked_frames"] >= 1 and g["frames"] == d["config"]["frames_per_step"]
    assert d["dtype"] == "i8" and d["roofline"]["kernel_ms"] > 0
    # the line carries both regimes: the reported (settled) run and the same W + K steps from an idle GPU
    assert d["settle"]["launches"] >= 32 and d["cold_start"]["kernel_ms"] > 0 and d["cold_start"]["value"] > 0


def test_bench_three_ranks_share_one_gpu(gpu):
    """The N-GPU path of bench.py with three REAL ranks, launched the way the driver launches it (torch.distributed.run,
    one process per rank), all on the one GPU a box has: shard plan, stream base, result targets in device buckets, the
    bucketed gather (11 steps: a full bucket of 8 and a partial one), rank 0's global-order and planted-frame checks.
    RCCL refuses two ranks on one device, so the lists travel over gloo through pinned host memory
    (ADSB_BENCH_REHEARSAL=1); a single-rank run of the same stream slices gives the expected frame count."""
    world, n = 3, 1 << 24
    env = dict(os.environ, ADSB_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    port = 29900 + os.getpid() % 90
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", str(world), "--steps", "11", "--warmup", "3", "--samples", str(n), "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    g = d["gather_check"]
    assert g["ok"], g
    assert d["n_gpus"] == world and "rehearsal" in d
    assert g["globally_ordered"] and g["spot_checked_frames"] >= world and g["frames"] == d["config"]["frames_per_step"]
    # the same three slices of the stream through one context in this process
    own = n - A.WINDOW
    cfg = A.synth_default()
    want = 0
    with A.AdsbDemod(max_samples=n, max_out=n // cfg.slot_len + 8192) as dm:
        for rk in range(world):
            iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, rk * own, n)
            frames, _ = dm.demod(iq)
            want += len(frames)
    assert g["frames"] == want, (g["frames"], want)


# ---- BASELINE.json configs[1] at full size: the WHOLE 1 GiB list against the oracle -------------
def test_config2_1GiB_full_output_equals_oracle(gpu, oracle):
    """The bench workload itself (536 870 912 i8 samples generated on the device, one launch): every frame of the HIP
    list -- offset, 14 bytes, status, repaired bit -- against the CPU oracle over the same samples.  (~6-10 s of one
    host core; bench.py repeats the comparison in its own JSON line as `parity_check`.)"""
    import torch
    n = 1 << 29
    cfg = A.synth_default()
    with A.AdsbDemod(max_samples=n, max_out=n // cfg.slot_len + 8192, host_staging=False,
                     stream=torch.cuda.current_stream().cuda_stream) as d:
        iq = torch.empty(2 * n, dtype=torch.int8, device="cuda")
        d.synth_fill_device(cfg, 0, 0, n, iq.data_ptr())
        d.demod_device_async(iq.data_ptr(), n)
        frames, counts, total, flags = d.fetch()
        host = iq.cpu().numpy().reshape(n, 2)
        del iq
    torch.cuda.empty_cache()
    assert flags == 0 and total == len(frames)
    rc, want, found = oracle.process_buffer(host, max_out=1 << 20)
    assert rc == 0 and found == len(want) > 250_000
    _eq(frames, want)


# ---- BASELINE.json configs[2] at full size: 16 GiB of i8 IQ resident on one MI355X ------------------------------
def test_config3_16GiB_whole_buffer(gpu, oracle):
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 20 * (1 << 30):
        pytest.skip(f"needs 20 GiB of free HBM, {free / 2**30:.1f} GiB available")
    n = 1 << 33                      # samples: 16 GiB at 2 bytes each
    cfg = A.synth_default(seed=2024)
    slots = n // cfg.slot_len
    cap = slots + 8192
    with A.AdsbDemod(max_samples=n, max_out=cap, host_staging=False,
                     stream=torch.cuda.current_stream().cuda_stream) as d:
        iq = torch.empty(2 * n, dtype=torch.int8, device="cuda")
        d.synth_fill_device(cfg, 0, 0, n, iq.data_ptr())
        d.demod_device_async(iq.data_ptr(), n)
        frames, counts, total, flags = d.fetch()
        assert flags == 0 and total == len(frames)
        off = frames["offset"].astype(np.int64)
        assert (np.diff(off) > 0).all() and off[0] >= 0 and off[-1] < n - 240
        # planted: one frame per slot; 90 % clean + 5 % with one data bit flipped (repaired) can come out, 2 % + 3 %
        # cannot; noise costs a little more (the 1 GiB bench buffer yields 94.33 % of its slots)
        assert 0.935 * slots < len(frames) < 0.955 * slots, (len(frames), slots)
        assert 0.040 * slots < int((frames["status"] == 1).sum()) < 0.060 * slots
        # sampled sub-ranges against the oracle, incl. the one that crosses byte offset 2^33 (sample 2^32)
        L = 1 << 20
        for a in (0, (1 << 32) - L // 2, 3 * (1 << 31) + 12_345, n - L):
            sub = iq[2 * a: 2 * (a + L)].cpu().numpy().reshape(L, 2)
            rc, want, cnt = oracle.process_buffer(sub)
            assert rc == 0 and cnt > 400
            lo, hi = np.searchsorted(off, a), np.searchsorted(off, a + L - 240)
            got = frames[lo:hi].copy()
            got["offset"] -= np.uint64(a)
            _eq(got, want)
        del iq
    torch.cuda.empty_cache()


def test_16GiB_cs16_whole_buffer(gpu, oracle):
    """The 16 GiB size in the reference's own sample format (Complex<i16>: 4 294 967 296 samples): strictly ascending
    offsets, the planted-frame band, sampled windows equal to the oracle incl. one across byte 2^33."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 20 * (1 << 30):
        pytest.skip(f"needs 20 GiB of free HBM, {free / 2**30:.1f} GiB available")
    n = 1 << 32                      # samples: 16 GiB at 4 bytes each
    cfg = A.synth_default(seed=77)
    cfg.amp_shift = 6
    slots = n // cfg.slot_len
    cap = slots + 8192
    with A.AdsbDemod(sample_type=A.ADSB_SAMPLE_I16, max_samples=n, max_out=cap, host_staging=False,
                     stream=torch.cuda.current_stream().cuda_stream) as d:
        iq = torch.empty(2 * n, dtype=torch.int16, device="cuda")
        d.synth_fill_device(cfg, 0, 0, n, iq.data_ptr())
        d.demod_device_async(iq.data_ptr(), n)
        frames, counts, total, flags = d.fetch()
        assert flags == 0 and total == len(frames)
        off = frames["offset"].astype(np.int64)
        assert (np.diff(off) > 0).all() and off[0] >= 0 and off[-1] < n - 240
        assert 0.935 * slots < len(frames) < 0.955 * slots, (len(frames), slots)
        assert 0.040 * slots < int((frames["status"] == 1).sum()) < 0.060 * slots
        L = 1 << 20
        for a in (0, (1 << 31) - L // 2, 3 * (1 << 30) + 54_321, n - L):  # (sample 2^31 = byte 2^33)
            sub = iq[2 * a: 2 * (a + L)].cpu().numpy().reshape(L, 2)
            rc, want, cnt = oracle.process_buffer(sub)
            assert rc == 0 and cnt > 400
            lo, hi = np.searchsorted(off, a), np.searchsorted(off, a + L - 240)
            got = frames[lo:hi].copy()
            got["offset"] -= np.uint64(a)
            _eq(got, want)
        del iq
    torch.cuda.empty_cache()


# ---- the opt-in mode that runs the finishing kernel on its own stream beside the next launch's scan ------------------
def test_overlapped_small_kernels_mode_matches_oracle(gpu, oracle, monkeypatch):
    """ADSB_OVERLAP_ORDERING=1 (read at adsb_create): finish_order of launch k runs on a second,
    high-priority stream behind the scan's own completion event while the scan of launch k+1 already runs.  Three
    device-resident buffers are launched back to back, twice, with no host synchronisation in between; every launch's
    list must be the oracle's (result sets alternate: a missing dependency shows as a mixed or stale list)."""
    import torch
    monkeypatch.setenv("ADSB_OVERLAP_ORDERING", "1")
    n = 1 << 21
    with A.AdsbDemod(max_samples=n, max_out=1 << 16, host_staging=False,
                     stream=torch.cuda.current_stream().cuda_stream) as d:
        bufs, wants = [], []
        for k in range(3):
            cfg = A.synth_default(seed=900 + k, slot_len=700)
            host = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, n)
            rc, want, cnt = oracle.process_buffer(host)
            assert rc == 0 and cnt > 1000
            bufs.append(torch.from_numpy(np.ascontiguousarray(host)).cuda())
            wants.append(want)
        torch.cuda.synchronize()
        for rnd in range(2):
            for k in (0, 1, 2, 1, 0, 2):
                d.demod_device_async(bufs[k].data_ptr(), n)
                if k == 2 or rnd == 1:   # (some launches are never fetched: their results are simply overwritten)
                    frames, counts, total, flags = d.fetch()
                    assert flags == 0
                    _eq(frames, wants[k])


# ---- CS16: the per-tile choice between the f16 3-input gate and the integer gate ---------------------------------
def test_cs16_gate_paths_agree_with_the_oracle(gpu, oracle):
    """A CS16 tile whose magnitudes all lie below 31744 (0x7C00: ordered f16 bit patterns) runs the 3-input f16 gate,
    any other tile the integer gate.  Both must be the reference's ordering test: full-range random input (integer
    gate everywhere), a realistic stream with a few full-scale samples dropped into some tiles only (both gates in
    one launch), and gate patterns built around the 31743 / 31744 boundary."""
    rng = np.random.default_rng(12)
    with A.AdsbDemod(sample_type=A.ADSB_SAMPLE_I16, max_samples=400_000, max_out=1 << 16) as d:
        def check(iq):
            frames, flags = d.demod(iq)
            rc, want, n = oracle.process_buffer(iq)
            assert rc == 0 and flags == 0
            _eq(frames, want)
            return len(want)
        # 1. full range: magnitudes up to 46340
        check(rng.integers(-32768, 32768, size=(200_000, 2)).astype(np.int16))
        check(rng.choice(np.array([-32768, -32767, 32766, 32767, 0], dtype=np.int16), size=(100_000, 2)))
        # 2. a stream of frames, with full-scale spikes in three 16384-offset stretches only (two 8192-offset CS16 tiles each)
        cfg = A.synth_default(seed=314, slot_len=700, amp_shift=7)
        iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I16, 0, 0, 300_000)
        n_clean = check(iq)
        for t in (1, 4, 7):
            pos = rng.integers(t * 16384, (t + 1) * 16384, size=40)
            iq[pos] = rng.choice(np.array([-32768, 32767, 23000, -23000], dtype=np.int16), size=(40, 2))
        assert check(iq) > 0.9 * n_clean
        # 3. the reference's gate KAT shape (demod.rs:250-278) around the boundary: highs >= lows by one, equal, reversed
        highs, lows = [0, 2, 7, 9], [1, 3, 4, 5, 6, 8, 10, 11, 12, 13, 14, 15]
        for hi, lo, expect in ((31744, 31743, 1), (31743, 31743, 1), (31743, 31744, 0), (31744, 31744, 1),
                               (46340, 31744, 1), (31743, 46340, 0), (31743, 31742, 1)):
            buf = np.zeros((241, 2), dtype=np.int16)
            mk = lambda m: (32767, 32767) if m == 46340 else (m, 0)   # |(32767, 32767)| = 46339.9 -> 46339; only order matters
            buf[highs] = mk(hi)
            buf[lows] = mk(lo)
            frames, flags = d.demod(buf)
            rc, want, n = oracle.process_buffer(buf)
            _eq(frames, want)
            assert len(frames) == expect, (hi, lo, len(frames))
