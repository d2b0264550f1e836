"""The deterministic synthetic IQ source: reproducible, sliceable, and what it plants
is what the reference algorithm (oracle) finds.  No GPU needed."""
import numpy as np


def test_slices_equal_the_whole(lib):
    cfg = lib.synth_default(seed=7)
    whole = lib.synth_fill_host(cfg, lib.ADSB_SAMPLE_I8, 2, 1000, 50_000)
    for a, b in ((0, 1), (123, 4567), (49_000, 50_000), (7, 50_000)):
        part = lib.synth_fill_host(cfg, lib.ADSB_SAMPLE_I8, 2, 1000 + a, b - a)
        assert (part == whole[a:b]).all()
    other = lib.synth_fill_host(cfg, lib.ADSB_SAMPLE_I8, 3, 1000, 50_000)
    assert (other != whole).any()


def test_noise_statistics(lib):
    cfg = lib.synth_default(seed=1, frame_pct=0)
    x = lib.synth_fill_host(cfg, lib.ADSB_SAMPLE_I8, 0, 0, 400_000).astype(np.float64)
    assert abs(x.mean()) < 0.1
    assert 7.5 < x.std() < 9.0  # Irwin-Hall(4 bytes)/18: sigma ~ 8.2 LSB


def test_planted_frames_are_found_by_the_oracle(lib, oracle):
    cfg = lib.synth_default(seed=3, slot_len=1000)
    n = 400_000
    iq = lib.synth_fill_host(cfg, lib.ADSB_SAMPLE_I8, 0, 0, n)
    rc, frames, found = oracle.process_buffer(iq)
    assert rc == 0
    by_off = {int(f["offset"]): f for f in frames}
    # Noise (sigma ~ 8 LSB against pulses of 40..110) breaks the all-pairs gate or a PPM decision now
    # and then, exactly as it would in the reference, so the planted frames are checked statistically.
    planted = [0, 0, 0, 0]
    good = [0, 0, 0, 0]
    for slot in range(n // 1000 - 1):
        present, start, clean, sent, kind = lib.synth_slot(cfg, 0, slot)
        assert present and start + 240 <= (slot + 1) * 1000
        planted[kind] += 1
        f = by_off.get(start)
        if kind == 0:    # clean frame: decoded as sent
            good[0] += f is not None and bytes(f["bytes"]) == clean and f["status"] == 0
        elif kind == 1:  # one flipped data bit: repaired (crc.rs:49-65), and the repaired bit is the planted one
            diff = int.from_bytes(clean, "big") ^ int.from_bytes(sent, "big")
            good[1] += (f is not None and bytes(f["bytes"]) == clean and f["status"] == 1
                        and f["fixed_bit"] == 111 - (diff.bit_length() - 1))
        elif kind == 2:  # one flipped CRC-field bit: never repaired
            good[2] += f is None
        else:            # two flipped data bits: not repaired to the clean frame
            good[3] += f is None or bytes(f["bytes"]) != clean
    assert all(k > 0 for k in planted)
    assert good[0] > 0.85 * planted[0] and good[1] > 0.8 * planted[1]
    assert good[2] >= planted[2] - 1 and good[3] == planted[3]
    # the clean frames carry a valid Mode-S CRC and DF17
    present, start, clean, sent, kind = lib.synth_slot(cfg, 0, 5)
    assert clean[0] == 0x8D and oracle.get_adsb_crc(clean[:11]) == int.from_bytes(clean[11:], "big")


def test_i16_generator_scales_i8(lib):
    cfg = lib.synth_default(seed=9, amp_shift=4)
    a = lib.synth_fill_host(cfg, lib.ADSB_SAMPLE_I16, 0, 0, 20_000).astype(np.int32)
    cfg0 = lib.synth_default(seed=9, amp_shift=0)
    b = lib.synth_fill_host(cfg0, lib.ADSB_SAMPLE_I16, 0, 0, 20_000).astype(np.int32)
    assert (a == b * 16).all()
