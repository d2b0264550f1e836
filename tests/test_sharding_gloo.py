"""The N > 1 path on CPU: world_size 2 and 3 gloo processes time-shard one stream, each rank demodulates its
slice (+ the 239/240-sample read halo) and rank 0 receives the frame lists.  There is no GPU in this tier, so
the per-rank demodulator is the CPU oracle standing in as a *checker stub* for the HIP path.  Under test:
  * sharding.plan / weak_plan (the halo, disjoint ownership),
  * sharding.BucketGather -- the bucketed, double-buffered gather bench.py runs unchanged over RCCL (same class,
    same calls; only the backend and the device of the tensors differ) -- incl. partial buckets, a step count
    that is not a multiple of the bucket size, and frame CONTENT against the single-buffer result,
  * sharding.gather_frame_lists (the one-shot variable-length gather, a convenience for callers without buckets).
The `-m gpu` tier runs the same shard plans through the HIP path on one device (tests/test_gpu_round2.py::test_shard_plan_through_hip_equals_single_buffer and tests/test_gpu_group.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import air_rs_amd as A
    from air_rs_amd import sharding
    from tests.oracle_binding import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = A.synth_default(seed=seed, slot_len=700)
        sh = sharding.plan(total, world)[rank]
        orc = Oracle()
        if sh.n_samples:
            # each rank generates only its own slice of the stream (no input exchange)
            iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, sh.first_sample, sh.n_samples)
            rc, frames, n = orc.process_buffer(iq)
            assert rc == 0
        else:
            frames = np.zeros(0, dtype=A.FRAME_DTYPE)
        merged = sharding.gather_frame_lists(frames, sh.first_offset, dist)
        if rank == 0:
            whole = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, total)
            rc, want, n = orc.process_buffer(whole)
            ok = len(merged) == len(want) and bool((merged == want).all())
            q.put((ok, len(want)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,world", [(200_003, 2), (100_000, 3), (241, 2), (240, 2)])
def test_time_sharded_stream_equals_single_buffer(total, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + total) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, 11, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    ok, n = q.get(timeout=10)
    assert ok
    if total > 50_000:
        assert n > 50


def test_plan_covers_every_offset_once():
    from air_rs_amd import sharding
    for total in (240, 241, 1000, 65536 + 240, 10**6 + 7):
        for world in (1, 2, 3, 8):
            shards = sharding.plan(total, world)
            assert sum(s.n_offsets for s in shards) == total - 240
            pos = 0
            for s in shards:
                assert s.first_offset == pos or s.n_offsets == 0
                pos += s.n_offsets
                if s.n_offsets:
                    assert s.first_sample + s.n_samples <= total
                    assert s.n_samples == s.n_offsets + 240
    w = sharding.weak_plan(1 << 20, 8)
    assert w[3].first_sample == 3 * ((1 << 20) - 240) and w[3].n_samples == 1 << 20


def _bucket_worker(rank, world, port, steps, bucket, keep, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import air_rs_amd as A
    from air_rs_amd import sharding
    from tests.oracle_binding import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = A.synth_default(seed=5, slot_len=600)
        orc = Oracle()
        seg = 30_011                      # samples per step; every step is another piece of one long stream
        cap = 256
        bg = sharding.BucketGather(dist, cap, bucket=bucket, device="cpu", keep=keep)
        for s in range(steps):
            sh = sharding.plan(seg, world)[rank]
            base = s * seg + sh.first_sample          # what adsb_set_stream_base() is given on the HIP path
            _, slot = bg.begin_launch()
            if sh.n_samples:
                iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, base, sh.n_samples)
                rc, frames, n = orc.process_buffer(iq)
                assert rc == 0
                frames = sharding.rebase(frames, base)
            else:
                frames = np.zeros(0, dtype=A.FRAME_DTYPE)
            sharding.write_payload(slot, frames)
            bg.end_launch()
        bg.drain()
        if rank == 0:
            check = range(steps) if keep else range(max(0, bg.last_launch() - (steps - 1) % bucket), bg.last_launch() + 1)
            ok, n_frames = True, 0
            assert bg.last_launch() == steps - 1
            for s in check:
                merged = sharding.merge_rank_lists(bg.lists_of_launch(s))
                whole = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, s * seg, seg)
                rc, want, n = orc.process_buffer(whole)
                want = sharding.rebase(want, s * seg)
                ok = ok and len(merged) == len(want) and bool((merged == want).all())
                n_frames += len(want)
            q.put((ok, n_frames, len(list(check))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,steps,bucket,keep", [(2, 11, 4, True), (3, 7, 8, True), (2, 9, 2, False), (2, 3, 8, False)])
def test_bucket_gather_delivers_every_launch(world, steps, bucket, keep):
    """steps % bucket != 0 (partial last bucket), steps < bucket (only a partial bucket), several full buckets
    through both halves of the double buffer; rank 0 compares rebased frame contents with the oracle on the
    unsharded stream piece -- all launches with keep=True, the launches of the last bucket otherwise."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() * 7 + world * 100 + steps * 10 + bucket) % 2000
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, steps, bucket, keep, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    ok, n_frames, n_checked = q.get(timeout=10)
    assert ok and n_checked >= 1
    assert n_frames > 30 * n_checked
