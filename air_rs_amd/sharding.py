"""Time-sharding of one long IQ stream across ranks.

Every offset of the reference loop (src/adsb.rs:98) is independent and no state crosses buffers,
so rank g can demodulate offsets [lo_g, hi_g) on its own as long as it reads the samples
[lo_g, hi_g + 239]: a 239-sample read halo (the window is 16 + 224 = 240 samples, adsb.rs:98,106;
BASELINE.json's "16-sample overlap" would only cover the preamble).  The only exchange is the
final gather of the per-rank frame lists (24 bytes per frame), rebased to stream offsets;
concatenated in rank order they are already globally sorted.
"""
from dataclasses import dataclass

import numpy as np

from .demod import FRAME_DTYPE, WINDOW


@dataclass(frozen=True)
class Shard:
    rank: int
    first_offset: int    # first stream offset this rank owns
    n_offsets: int       # how many it owns
    first_sample: int    # == first_offset
    n_samples: int       # n_offsets + 240 (0 if the rank owns nothing)


def plan(total_samples: int, world: int):
    """Even split of the offsets [0, total_samples - 240) over `world` ranks."""
    if total_samples < WINDOW:
        raise ValueError("stream shorter than 240 samples (the reference panics, adsb.rs:98)")
    n_off = total_samples - WINDOW
    per = -(-n_off // world) if n_off else 0
    shards = []
    for g in range(world):
        lo = min(g * per, n_off)
        hi = min(lo + per, n_off)
        cnt = hi - lo
        shards.append(Shard(g, lo, cnt, lo, cnt + WINDOW if cnt else 0))
    return shards


def weak_plan(samples_per_rank: int, world: int):
    """Fixed per-rank buffer (bench.py): rank g reads samples_per_rank samples starting at
    g * (samples_per_rank - 240); the stream is world*(n-240)+240 samples long."""
    own = samples_per_rank - WINDOW
    return [Shard(g, g * own, own, g * own, samples_per_rank) for g in range(world)]


def rebase(frames: np.ndarray, first_offset: int) -> np.ndarray:
    out = frames.copy()
    out["offset"] += np.uint64(first_offset)
    return out


def gather_frame_lists(local_frames: np.ndarray, first_offset: int, dist, device=None, dst: int = 0):
    """Gather every rank's (rebased) frame list on rank `dst` through torch.distributed (`dist`):
    an all_gather of the counts, then a gather of the lists padded to the largest count.
    Returns the merged, globally ordered list on dst and None elsewhere.  Works with gloo (CPU
    tests) and nccl == RCCL (device tensors)."""
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    mine = rebase(local_frames, first_offset)
    dev = device if device is not None else "cpu"
    count = torch.tensor([len(mine)], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    buf = np.zeros(cap, dtype=FRAME_DTYPE)
    buf[:len(mine)] = mine
    send = torch.from_numpy(buf.view(np.uint8).copy()).to(dev)
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst)
    if rank != dst:
        return None
    parts = [r.cpu().numpy().view(FRAME_DTYPE)[:c] for r, c in zip(recv, counts)]
    merged = np.concatenate(parts) if parts else np.zeros(0, dtype=FRAME_DTYPE)
    # concatenation in rank order is already sorted because shards are disjoint and ascending
    assert (np.diff(merged["offset"].astype(np.int64)) > 0).all() if len(merged) > 1 else True
    return merged


# ---- per-launch frame lists to rank 0, one collective per BUCKET launches ---------------------------------
# Payload of one launch = what adsb_set_result_target() makes the ordering pass write (include/adsb_hip.h):
#   [ u64 n_out | u64 total_found | u64 flags | u64 0 | adsb_frame[cap] ]   (16-byte aligned)
HEADER_BYTES = 32
FRAME_BYTES = FRAME_DTYPE.itemsize  # 24


def payload_bytes(cap_frames: int) -> int:
    return (HEADER_BYTES + int(cap_frames) * FRAME_BYTES + 15) // 16 * 16


def write_payload(slot, frames: np.ndarray, total_found=None, flags: int = 0):
    """Fill one payload slot (a uint8 torch tensor on the CPU) the way the HIP ordering pass fills it on the
    device.  Used by CPU-tier tests, whose per-rank demodulator is a stand-in for the HIP path."""
    cap = (slot.numel() - HEADER_BYTES) // FRAME_BYTES
    n = min(len(frames), cap)
    total = len(frames) if total_found is None else int(total_found)
    raw = slot.numpy()
    raw[:HEADER_BYTES].view(np.uint64)[:] = (n, total, flags | (1 if total > n else 0), 0)
    raw[HEADER_BYTES:HEADER_BYTES + n * FRAME_BYTES] = np.ascontiguousarray(frames[:n]).view(np.uint8)


def parse_payload(raw: np.ndarray):
    """(n_out, total_found, flags, frames) of one payload given as a uint8 numpy array."""
    n_out, total, flags, _ = (int(x) for x in raw[:HEADER_BYTES].view(np.uint64))
    frames = raw[HEADER_BYTES:HEADER_BYTES + n_out * FRAME_BYTES].view(FRAME_DTYPE)
    return n_out, total, flags, frames


class BucketGather:
    """Moves every rank's per-launch frame lists to rank 0 with ONE torch.distributed gather per `bucket`
    launches (backend `nccl` = RCCL over xGMI on the GPUs, `gloo` on the CPU for tests; the code is the same).

    A cross-stream hand-off per launch costs 40-60 us on MI355X (15-20 % of a 1 GiB step),
    so launches write their ordered list straight into slot i % bucket of a bucket (zero copies:
    adsb_set_result_target), a full bucket is gathered from a side stream while the next bucket fills, and two
    buckets alternate.  Offsets are absolute (adsb_set_stream_base), so rank 0 concatenates the lists of one
    launch in rank order and has the globally ordered list.

    Stream contract (device tensors): the launches that fill the slots must be enqueued on torch's CURRENT stream at
    begin_launch() / end_launch() time -- i.e. the AdsbDemod context must have been created on that stream
    (`stream=torch.cuda.current_stream().cuda_stream`, as bench.py does).  Re-use of a bucket half is ordered behind
    the gather that still reads it by `pending.wait()`, which makes the current stream wait; a context with a stream
    of its own would overwrite a slot RCCL is still reading.

    Use, per launch:   ptr = bg.begin_launch()         # slot to write; waits (stream-side) for the gather
                       ...enqueue the launch...        #   that still reads this half of the double buffer
                       bg.end_launch(wait_results)     # full bucket -> gather
    then once:         bg.drain()
    Rank 0 reads the gathered lists with lists_of_launch(); with keep=True every launch's lists are parsed as
    the buckets complete (tests), otherwise only the last two buckets are available (bench.py: no host work
    inside the timed region).
    """

    def __init__(self, dist, cap_frames: int, bucket: int = 8, device="cpu", keep: bool = False, dst: int = 0,
                 host_staged: bool = False):
        import torch
        self.torch = torch
        self.dist, self.rank, self.world, self.dst = dist, dist.get_rank(), dist.get_world_size(), dst
        self.bucket, self.payload, self.keep = int(bucket), payload_bytes(cap_frames), keep
        self.cuda = str(device).startswith("cuda")
        # host_staged (rehearsals only: several ranks sharing ONE GPU, where RCCL refuses to form a communicator):
        # the launches still write into device buckets, a flushed bucket is copied to pinned host memory on the side
        # stream and gathered with a CPU backend (gloo).  Blocking; not for timing.
        self.host_staged = bool(host_staged) and self.cuda
        size = self.bucket * self.payload
        self.buf = [torch.zeros(size, dtype=torch.uint8, device=device) for _ in range(2)]
        rdev = "cpu" if self.host_staged else device
        self.recv = [[torch.empty(size, dtype=torch.uint8, device=rdev) for _ in range(self.world)]
                     if self.rank == dst else None for _ in range(2)]
        self.stage = [torch.empty(size, dtype=torch.uint8).pin_memory() for _ in range(2)] if self.host_staged else None
        self.side = torch.cuda.Stream() if self.cuda else None
        self.pending = [None, None]       # outstanding gather per half
        self.filled = [0, 0]              # launches in the gather outstanding / last completed per half
        self.first = [0, 0]               # index of the first launch of that bucket
        self.i = 0                        # launches begun
        self.open = False                 # a launch has begun and not ended
        self.kept = {}                    # keep=True, rank dst: launch index -> [per-rank (n_out, total, flags, frames)]

    # -- per launch ------------------------------------------------------------------------------------
    def _half_slot(self, i):
        return (i // self.bucket) & 1, i % self.bucket

    def begin_launch(self):
        """Returns (data_ptr, uint8 tensor view) of the payload slot the next launch must fill."""
        assert not self.open
        bk, slot = self._half_slot(self.i)
        if slot == 0:
            self._retire(bk)              # the gather that still reads this half (two buckets ago)
            self.first[bk] = self.i
        view = self.buf[bk][slot * self.payload:(slot + 1) * self.payload]
        self.open = True
        return view.data_ptr(), view

    def end_launch(self, wait_results=None):
        """The launch that fills the slot handed out by begin_launch() has been enqueued (or, on the CPU, has
        written it).  wait_results(stream_handle) must make that stream wait for the launch's results
        (AdsbDemod.stream_wait_results); it is called only when a bucket is flushed."""
        assert self.open
        self.open = False
        bk, slot = self._half_slot(self.i)
        self.i += 1
        if slot == self.bucket - 1:
            self._flush(bk, self.bucket, wait_results)

    def _flush(self, bk, n_launches, wait_results):
        torch = self.torch
        self.filled[bk] = n_launches
        if self.cuda:
            if wait_results is not None:
                wait_results(self.side.cuda_stream)   # side stream waits for the last ordering pass
            if self.host_staged:
                with torch.cuda.stream(self.side):
                    self.stage[bk].copy_(self.buf[bk], non_blocking=True)
                self.side.synchronize()
                self.pending[bk] = self.dist.gather(self.stage[bk], self.recv[bk], dst=self.dst, async_op=True)
                return
            with torch.cuda.stream(self.side):
                self.pending[bk] = self.dist.gather(self.buf[bk], self.recv[bk], dst=self.dst, async_op=True)
        else:
            self.pending[bk] = self.dist.gather(self.buf[bk], self.recv[bk], dst=self.dst, async_op=True)

    def _retire(self, bk):
        if self.pending[bk] is not None:
            self.pending[bk].wait()       # (CUDA: the current stream waits; the host does not block)
            self.pending[bk] = None
            if self.keep and self.rank == self.dst:
                if self.cuda:
                    self.torch.cuda.current_stream().synchronize()
                for k in range(self.filled[bk]):
                    self.kept[self.first[bk] + k] = self._parse(bk, k)

    def drain(self, wait_results=None):
        """Flush a partly filled bucket and wait for every outstanding gather."""
        assert not self.open
        bk, slot = self._half_slot(self.i)
        if slot != 0:                      # partial bucket: slots >= slot hold older launches and are ignored
            self._flush(bk, slot, wait_results)
            self.i += self.bucket - slot   # the next launch starts a fresh bucket
        for h in ((bk + 1) & 1, bk):       # oldest first
            if self.pending[h] is not None:
                self.pending[h].wait()
                if self.side is not None:
                    self.side.synchronize()
                self.pending[h] = None
                if self.keep and self.rank == self.dst:
                    for k in range(self.filled[h]):
                        self.kept[self.first[h] + k] = self._parse(h, k)

    # -- rank dst: what arrived --------------------------------------------------------------------------
    def _parse(self, bk, slot):
        out = []
        for r in range(self.world):
            raw = self.recv[bk][r][slot * self.payload:(slot + 1) * self.payload].cpu().numpy().copy()
            out.append(parse_payload(raw))
        return out

    def lists_of_launch(self, launch: int):
        """Rank dst, after drain(): [(n_out, total_found, flags, frames)] per rank for launch index `launch`
        (any launch with keep=True; otherwise only launches of the two most recent buckets)."""
        assert self.rank == self.dst
        if launch in self.kept:
            return self.kept[launch]
        for bk in (0, 1):
            if self.first[bk] <= launch < self.first[bk] + self.filled[bk]:
                return self._parse(bk, launch - self.first[bk])
        raise KeyError(f"launch {launch} is no longer held (keep=False keeps two buckets)")

    def last_launch(self):
        bk = max((0, 1), key=lambda h: self.first[h] + self.filled[h] if self.filled[h] else -1)
        return self.first[bk] + self.filled[bk] - 1


ADSB_FLAG_INCOMPLETE = 0x2  # include/adsb_hip.h


def merge_rank_lists(per_rank):
    """Concatenate one launch's per-rank lists (absolute offsets) in rank order; asserts global order.  A list
    whose payload header carries ADSB_FLAG_INCOMPLETE has holes (slot-pool overflow on the producing rank, which
    must call fetch_counts() before its bucket is flushed): refused, never merged silently."""
    for r, (_, _, fl, _) in enumerate(per_rank):
        if fl & ADSB_FLAG_INCOMPLETE:
            raise ValueError(f"rank {r}'s list is incomplete (ADSB_FLAG_INCOMPLETE): the producer must repair it with fetch_counts() first")
    parts = [f for (_, _, _, f) in per_rank]
    merged = np.concatenate(parts) if parts else np.zeros(0, dtype=FRAME_DTYPE)
    if len(merged) > 1:
        assert (np.diff(merged["offset"].astype(np.int64)) > 0).all(), "per-rank lists are not globally ordered"
    return merged
