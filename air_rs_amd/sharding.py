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
