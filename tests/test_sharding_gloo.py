"""The N > 1 path on CPU: world_size-2 gloo processes time-shard one stream, each rank demodulates
its slice (+239/240-sample read halo) and rank 0 gathers the frame lists.  The per-rank demodulator
here is the CPU oracle standing in as a *checker stub* for the HIP path (no GPU in this tier); what
is under test is the sharding plan, the halo, the offset rebasing and the gather -- the code
bench.py and a multi-GPU deployment use unchanged."""
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
