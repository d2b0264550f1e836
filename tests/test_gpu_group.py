"""adsb_group_*: one buffer time-sharded over several contexts behind ONE native call (no torch, no launcher;
SURVEY section 8e).  A box has one GPU, so every member sits on device 0 -- the code path (per-member contexts and
streams, slices with the 239-sample halo, absolute offsets, hipMemcpyPeerAsync merge on the root's device) is the
N-device one; what one GPU cannot show is the copy BETWEEN devices and any scaling."""
import numpy as np
import pytest

import air_rs_amd as A

pytestmark = pytest.mark.gpu


def _eq(got, want):
    assert len(got) == len(want), (len(got), len(want))
    if len(got):
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (bad[:5], got[bad[:3]], want[bad[:3]])


@pytest.mark.parametrize("st,total,world", [(A.ADSB_SAMPLE_I8, 3_000_017, 8), (A.ADSB_SAMPLE_I8, 700_001, 3),
                                            (A.ADSB_SAMPLE_I8, 500_000, 2), (A.ADSB_SAMPLE_I16, 1_200_003, 8)])
def test_group_equals_single_context_and_oracle(gpu, oracle, st, total, world):
    cfg = A.synth_default(seed=78, slot_len=900)
    if st == A.ADSB_SAMPLE_I16:
        cfg.amp_shift = 5
    whole = A.synth_fill_host(cfg, st, 0, 0, total)
    rc, want, n = oracle.process_buffer(whole)
    assert rc == 0 and n > 400
    with A.AdsbDemod(sample_type=st, max_samples=total, max_out=1 << 16) as d:
        single, flags = d.demod(whole)
        assert flags == 0
    _eq(single, want)
    with A.AdsbGroup([0] * world, sample_type=st, max_samples=total, max_out=1 << 16) as g:
        frames, flags = g.demod(whole)           # host buffer -> per-member H2D + launch -> merge -> host
        assert flags == 0
        _eq(frames, want)
        # same group, a shorter buffer (members own fewer offsets), then one with fewer offsets than members
        frames, flags = g.demod(whole[: total // 3 + 5])
        rc, want2, n2 = oracle.process_buffer(whole[: total // 3 + 5])
        _eq(frames, want2)
        tiny = np.zeros((240 + 3, 2), dtype=whole.dtype)   # three offsets, all-zero input: three all-zero frames
        frames, flags = g.demod(tiny)
        assert len(frames) == 3 and (frames["offset"] == np.arange(3)).all()
        frames, flags = g.demod(tiny[:240])                # exactly 240 samples: zero offsets (adsb.rs:98)
        assert len(frames) == 0 and flags == 0
        with pytest.raises(A.AdsbError) as e:
            g.demod(tiny[:239])
        assert e.value.code == A.ADSB_E_SHORT


def test_group_device_resident_and_truncation(gpu, oracle):
    """Device-resident entry: the members' slices are views of ONE device buffer (the plan's slice starts are
    multiples of 8 samples = 16 bytes); the merged blob on the root's device; truncation keeps the FIRST frames."""
    import torch
    total, world = 2_000_003, 5
    cfg = A.synth_default(seed=79, slot_len=700)
    whole = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, total)
    rc, want, n = oracle.process_buffer(whole)
    t = torch.from_numpy(whole).cuda()
    plan = A.group_plan(total, world)
    ptrs = [t.data_ptr() + first * 2 if ns else None for first, ns, noff in plan]
    with A.AdsbGroup([0] * world, max_samples=total, max_out=1 << 14, host_staging=False) as g:
        g.demod_device_async(ptrs, total)
        frames, tot, flags = g.fetch()
        assert flags == 0 and tot == n
        _eq(frames, want)
        blob_ptr, stream = g.result_device()
        torch.cuda.synchronize()
        class _Blob:   # the device blob as a torch tensor (array-interface import: no copy, no second HIP runtime)
            __cuda_array_interface__ = {"shape": (32 + n * 24,), "typestr": "|u1", "data": (blob_ptr, False), "version": 2}
        host = torch.as_tensor(_Blob(), device="cuda").cpu().numpy()
        assert tuple(host[:32].view(np.uint64)) == (n, n, 0, 0)
        _eq(host[32:].view(A.FRAME_DTYPE), want)
    cap = 100
    with A.AdsbGroup([0] * world, max_samples=total, max_out=cap, host_staging=False) as g:
        g.demod_device_async(ptrs, total)
        frames, tot, flags = g.fetch()
        assert flags & A.ADSB_FLAG_TRUNCATED and tot == n and len(frames) == cap
        _eq(frames, want[:cap])


def test_group_constant_input_every_offset(gpu):
    """SURVEY F8 through a group: all-zero input emits a frame per offset; the members' slot pools overflow, the
    per-member re-plan runs inside adsb_group_fetch, and the merged list is still the first max_out offsets."""
    n = 300_000
    with A.AdsbGroup([0, 0, 0], max_samples=n, max_out=50_000) as g:
        frames, flags = g.demod(np.zeros((n, 2), dtype=np.int8))
        assert flags & A.ADSB_FLAG_TRUNCATED
        assert len(frames) == 50_000 and (frames["offset"] == np.arange(50_000)).all() and not frames["bytes"].any()
