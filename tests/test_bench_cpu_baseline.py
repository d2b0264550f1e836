"""bench.py's cpu_baseline leg on a small buffer (CPU only): the time-sharded all-cores courtesy run must find
exactly the frames the single-thread pass finds (240-sample overlap, like the multi-GPU sharding)."""
import numpy as np


def test_cpu_baseline_sharded_equals_single(lib, oracle):
    import bench
    cfg = lib.synth_default()
    iq = lib.synth_fill_host(cfg, lib.ADSB_SAMPLE_I8, 0, 0, 1 << 20)
    rc, frames, found = oracle.process_buffer(iq)
    # the list handed over as "the GPU's" is the checker's own here (no GPU in this tier): the comparison leg itself
    out, parity = bench.cpu_baseline(iq, gpu_frames=frames, target_seconds=0.2)
    assert out["kind"] == "port" and out["cores"] == 1 and out["value"] > 0
    assert parity["ok"] and parity["frames"] == found
    bad = frames.copy()
    bad["bytes"][7, 3] ^= 1
    assert not bench.cpu_baseline(iq, gpu_frames=bad, target_seconds=0.2)[1]["ok"]
    assert not bench.cpu_baseline(iq, gpu_frames=frames[:-1], target_seconds=0.2)[1]["ok"]
    assert rc == 0 and f"{found} frames per pass" in out["sample"]
    if "all_cores" in out:  # single-core machines skip the courtesy number
        assert "error" not in out["all_cores"], out["all_cores"]
        assert out["all_cores"]["frames_per_pass"] == found
        assert 1 < out["all_cores"]["cores"] <= 16
