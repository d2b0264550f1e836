"""GPU-tier tests added in round 4 (all through the C ABI, against the CPU oracle):

* the far end of BASELINE config 5's stream on one GPU (rank 7 of 8 x 16 GiB starts at sample 7 (2^33 - 240); a base
  near 2^40): absolute 64-bit offsets through both launch paths, with a dense tile and a tile of more than 16 survivors;
* finish_order's give-up path: a workgroup that never publishes its exchange word costs ~0.1 s and ADSB_E_STATE, not a
  hang, and the context stays usable; the exchange words survive the wrap of their 30-bit epoch;
* the streaming front end with two ring slots and two buffers in flight (a producer that acquires the next slot while a
  one-dispatch kernel may still be reading it);
* `python bench.py --gpus 3` with no launcher environment (bench.py starts its own ranks)."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import air_rs_amd as A

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILE = 16384


def _eq(got, want):
    assert len(got) == len(want), (len(got), len(want))
    if len(got):
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (bad[:5], got[bad[:3]], want[bad[:3]])


def _mixed_buffer(n_tiles=6, seed=77):
    """synthetic stream | a constant stretch (every offset emits: SURVEY F8) | coarse noise (ties: > 16 survivors in a tile)"""
    cfg = A.synth_default(seed=seed, slot_len=700)
    n = n_tiles * TILE + 1000
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, n).copy()
    iq[2 * TILE + 100: 3 * TILE + 700] = (5, -3)                      # constant: one frame per offset, across a tile edge
    rng = np.random.default_rng(seed)
    iq[4 * TILE: 5 * TILE] = rng.integers(-1, 2, size=(TILE, 2), dtype=np.int8)   # magnitudes 0/1: ties everywhere
    return iq


@pytest.mark.parametrize("scan", ["root"])
@pytest.mark.parametrize("small", ["1", "0"])
@pytest.mark.parametrize("base", [7 * ((1 << 33) - 240), (1 << 40) - 12345, (1 << 32) - 5000])
def test_far_end_stream_base(gpu, oracle, monkeypatch, scan, small, base):
    """adsb_set_stream_base at the far end of a 128 GiB stream (rank 7 of BASELINE config 5; near 2^40; across 2^32): the
    kernels' 64-bit absolute offsets and finish_order's 32-bit RELATIVE rank keys (adsb_kernels.hip: row_rank keys,
    finish_big_tile) on a buffer with a dense tile and a > 16-survivor tile, through the one-dispatch path and the
    two-kernel path, equal the oracle's offsets + base (src/adsb.rs:98: ascending i)."""
    monkeypatch.setenv("ADSB_SCAN", scan)
    monkeypatch.setenv("ADSB_SMALL_PATH", small)
    iq = _mixed_buffer()
    rc, want, n = oracle.process_buffer(iq, max_out=1 << 17)
    assert rc == 0 and n == len(want) and n > TILE                      # the constant stretch alone gives > 16384 frames
    want = want.copy()
    want["offset"] += np.uint64(base)
    with A.AdsbDemod(max_samples=len(iq), max_out=1 << 17) as d:
        assert d.scan == scan
        d.set_stream_base(base)
        frames, flags = d.demod(iq)
        assert flags == 0
        _eq(frames, want)
        assert (np.diff(frames["offset"].astype(np.int64)) > 0).all()
        d.set_stream_base(0)
        frames0, _ = d.demod(iq)
        assert (frames0["offset"] + np.uint64(base) == frames["offset"]).all()


def test_finish_order_gives_up_instead_of_hanging(gpu, oracle, monkeypatch):
    """A finish_order workgroup that never publishes its exchange word: the workgroups behind it wait ~0.1 s, give up,
    the host gets ADSB_E_STATE and the header ADSB_FLAG_INCOMPLETE -- and the NEXT launch on the same context (a new
    epoch: the stale words read as "not there yet") is whole again.  The reference's only failure mode is a closed
    channel (src/adsb.rs:108-111): the replacement must not add a hang."""
    monkeypatch.setenv("ADSB_SMALL_PATH", "0")                          # the two-kernel path: several finish_order workgroups
    cfg = A.synth_default(seed=5)
    n = 40 * TILE * 3                                                   # 120 tiles = 4 workgroups of 32 tiles
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, n)
    rc, want, _ = oracle.process_buffer(iq, max_out=1 << 16)
    assert rc == 0
    with A.AdsbDemod(max_samples=n, max_out=1 << 16) as d:
        frames, _ = d.demod(iq)
        _eq(frames, want)
        d.finish_stall(1)                                               # workgroup 1 of 4 withholds its word
        t0 = time.perf_counter()
        with pytest.raises(A.AdsbError) as e:
            d.demod(iq)
        dt = time.perf_counter() - t0
        assert e.value.code == A.ADSB_E_STATE
        assert dt < 20.0, dt                                            # (bounded spin: no hang; ~0.1-1 s in practice)
        d.finish_stall()                                                # back to normal: the same context, the next epoch
        for _ in range(3):
            frames, flags = d.demod(iq)
            assert flags == 0
            _eq(frames, want)


def test_exchange_words_survive_the_epoch_wrap(gpu, oracle, monkeypatch):
    """finish_order's exchange words are tagged with (launch index + 1) mod 2^30 and never cleared -- except when that
    epoch wraps: launches 2^30 - 3 .. 2^30 + 2 on a context whose words still carry the tags of launches 1 .. 6."""
    monkeypatch.setenv("ADSB_SMALL_PATH", "0")
    cfg = A.synth_default(seed=9)
    n = 70 * TILE
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, n)
    rc, want, _ = oracle.process_buffer(iq, max_out=1 << 16)
    assert rc == 0
    with A.AdsbDemod(max_samples=n, max_out=1 << 16) as d:
        for _ in range(6):                                              # epochs 1 .. 6 are in the words now
            frames, _ = d.demod(iq)
        _eq(frames, want)
        d.set_launch_index((1 << 30) - 3)
        for _ in range(10):                                             # ... across the wrap, past epoch 6 again
            frames, flags = d.demod(iq)
            assert flags == 0
            _eq(frames, want)


@pytest.mark.parametrize("small", ["1", "0"])
def test_feed_two_ring_slots_two_in_flight(gpu, oracle, monkeypatch, small):
    """ring_slots = 2 and a producer that fills the NEXT slot in place while two buffers are in flight: the slot it gets
    back belongs to the oldest buffer, whose one-dispatch kernel reads its samples from that very slot -- acquire must
    not hand it out before that kernel has finished (the slot is scribbled over at once here)."""
    monkeypatch.setenv("ADSB_SMALL_PATH", small)
    cfg = A.synth_default(seed=13, slot_len=500)
    chunk = 20_000
    data = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, 60 * chunk)
    with A.AdsbDemod(max_samples=chunk + 240, max_out=1 << 14, host_staging=False) as d:
        with A.Feed(d, max_chunk=chunk, carry=False, ring_slots=2) as f:
            got = []
            for k in range(60):
                if f.in_flight == 2:
                    # two in flight: take the next slot BEFORE popping, scribble, then pop the oldest and fill for real
                    slot = f.acquire()
                    slot[:] = 77
                    got.append(f.pop())
                    slot[:chunk] = data[k * chunk:(k + 1) * chunk]
                    f.push_acquired(chunk)
                else:
                    f.push(data[k * chunk:(k + 1) * chunk])
            while f.in_flight:
                got.append(f.pop())
    assert len(got) == 60
    for k, (frames, flags, first) in enumerate(got):
        rc, want, _ = oracle.process_buffer(data[k * chunk:(k + 1) * chunk])
        assert rc == 0 and flags == 0 and first == k * chunk
        _eq(frames, want)


def test_bench_starts_its_own_ranks(gpu):
    """`python bench.py --gpus 3` with NO launcher environment (the way the driver runs BENCH / SCALE): bench.py starts
    its three ranks itself (a child torch.distributed.run), relays rank 0's one JSON line and the exit code; the line says
    what the process group saw.  Three ranks share the box's one GPU (ADSB_BENCH_REHEARSAL=1: lists over gloo)."""
    env = dict(os.environ, ADSB_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "11", "--warmup", "3",
                        "--samples", str(1 << 24), "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["ranks_seen"]["world_size"] == 3 and d["ranks_seen"]["backend"] == "gloo"
    assert len(d["per_rank"]["scan_kernel_ms"]) == 3 and all(x > 0 for x in d["per_rank"]["scan_kernel_ms"])
    assert d["gather_check"]["ok"], d["gather_check"]
