"""adsb_group_plan (include/adsb_hip.h): the native time-shard plan of adsb_group_* -- a pure function of the C ABI, so it
is checked on the CPU tier: every offset of the reference loop (adsb.rs:98) owned exactly once, 240-sample overlap, slices
16-byte aligned; and the same cut as air_rs_amd.sharding.plan up to that alignment."""
import pytest

import air_rs_amd as A
from air_rs_amd import sharding


def test_group_plan_covers_every_offset_once():
    for n in (240, 241, 247, 248, 1000, 20000, 3_000_017):
        for world in (1, 2, 3, 8, 64):
            sh = A.group_plan(n, world)
            assert sum(s[2] for s in sh) == n - 240
            pos = 0
            for first, ns, noff in sh:
                if noff == 0:
                    assert ns == 0
                    continue
                assert first == pos and first % 8 == 0 and ns == noff + 240 and first + ns <= n
                pos += noff
    with pytest.raises(A.AdsbError) as e:
        A.group_plan(239, 2)
    assert e.value.code == A.ADSB_E_SHORT




def test_group_plan_agrees_with_the_python_plan_where_aligned():
    """sharding.plan (what bench.py --gpus N uses) splits evenly; the native plan rounds the slice length up to a multiple
    of 8 samples so that every slice of one 16-byte aligned buffer starts aligned.  Both own every offset exactly once."""
    for n in (100_000 + 240, 3_000_017, 1 << 24):
        for world in (2, 3, 8):
            native = A.group_plan(n, world)
            py = sharding.plan(n, world)
            assert sum(s[2] for s in native) == n - 240 == sum(sh.n_offsets for sh in py)
            for sh in py:   # the same reading rule on both sides: a slice is its offsets plus the 240-sample window
                assert sh.n_samples == sh.n_offsets + 240 and sh.first_sample == sh.first_offset
