"""Host-side mirror of AdsbPacket / msgs (air_rs_amd/csrc/host) against the reference's KATs and
against the oracle's independent restatement of the same decode.  No GPU needed."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


def _frame_from_me(me_hex):
    return bytes([0x8D, 0, 0, 0]) + bytes.fromhex(me_hex) + bytes(3)


@pytest.mark.parametrize("k", KATS["aircraft_id"], ids=lambda k: k["src"])
def test_aircraft_id(lib, k):
    v = lib.packet_new(_frame_from_me(k["me_hex"]))
    assert v.msg_kind == 0 and v.callsign.decode() == k["callsign"] and v.msg_type == k["msg_type"]


@pytest.mark.parametrize("k", KATS["aircraft_position"], ids=lambda k: k["src"])
def test_aircraft_position(lib, k):
    v = lib.packet_new(_frame_from_me(k["me_hex"]))
    assert v.msg_kind == 1
    for f in ("altitude", "msg_type", "surveillance_status", "nic_supplement", "cpr_time", "cpr_odd",
              "cpr_latitude", "cpr_longitude"):
        if f in k:
            assert getattr(v, f) == k[f], f


@pytest.mark.parametrize("k", KATS["frames"], ids=lambda k: k["hex"])
def test_new_from_string(lib, k):
    v = lib.packet_new_from_string(k["hex"])
    assert f"{v.icao:06X}" == k["icao"]
    assert v.msg_kind == {"id": 0, "position": 1, "unknown": 2}[k["kind"]]
    for f in ("downlink_format", "capability", "msg_type", "altitude", "cpr_odd", "cpr_latitude", "cpr_longitude"):
        if f in k:
            assert getattr(v, f) == k[f], f
    if "callsign" in k:
        assert v.callsign.decode() == k["callsign"]


def test_matches_oracle_on_random_frames(lib, oracle):
    rng = np.random.default_rng(5)
    for _ in range(3000):
        b = bytes(rng.integers(0, 256, size=14, dtype=np.uint8))
        v, o = lib.packet_new(b), oracle.packet_new(b)
        for f in ("downlink_format", "capability", "icao", "msg_type", "msg_kind", "surveillance_status",
                  "nic_supplement", "altitude", "cpr_time", "cpr_odd", "cpr_latitude", "cpr_longitude"):
            assert getattr(v, f) == getattr(o, f), (f, b.hex())
        assert v.callsign == o.callsign and bytes(v.raw_msg) == bytes(o.raw_msg)
        assert lib.packet_display(b, "T") == oracle.packet_display(b, "T")


def test_capability_mask_is_5_like_the_reference(lib):
    # packet.rs:27 masks with 5 (not 7): CA=7 reads back as 5, CA=2 as 0
    assert lib.packet_new(bytes([0x8F]) + bytes(13)).capability == 5
    assert lib.packet_new(bytes([0x8A]) + bytes(13)).capability == 0


def test_unknown_message_keeps_ten_raw_bytes(lib):
    raw = bytes.fromhex("8d4840d6ea8f0885a73f9700a1b2")  # TC 29
    v = lib.packet_new(raw)
    assert v.msg_kind == 2 and bytes(v.raw_msg) == raw[4:]
    assert "Raw Msg :  [234, 143, 8, 133, 167, 63, 151, 0, 161, 178]" in lib.packet_display(raw)


def test_c16_round_trip(lib, tmp_path):
    import ctypes as C
    from air_rs_amd import _lib
    L = _lib.load()
    data = np.array([[1, -2], [32767, -32768], [0, 255], [-256, 3]], dtype="<i2")
    p = str(tmp_path / "x.c16").encode()
    assert L.adsb_save_c16(p, data.ctypes.data, len(data)) == 0
    assert open(p, "rb").read() == data.tobytes()  # raw LE i16 I,Q pairs, no header (utils.rs:6-20)
    ptr, n = C.POINTER(C.c_int16)(), C.c_size_t()
    assert L.adsb_load_c16(p, C.byref(ptr), C.byref(n)) == 0 and n.value == 4
    back = np.ctypeslib.as_array(ptr, shape=(8,)).copy().reshape(4, 2)
    L.adsb_free(ptr)
    assert (back == data).all()
    open(p, "ab").write(b"\x00")  # length not divisible by 4 is rejected (utils.rs:28-30)
    assert L.adsb_load_c16(p, C.byref(ptr), C.byref(n)) != 0


def test_raw_u8_capture_loads_as_i8(lib, tmp_path):
    """rtl_sdr's unsigned-byte capture (not a reference format): x - 128 into the ADSB_SAMPLE_I8 layout."""
    import ctypes as C
    from air_rs_amd import _lib
    L = _lib.load()
    raw = np.arange(256, dtype=np.uint8)
    p = str(tmp_path / "x.bin").encode()
    open(p, "wb").write(raw.tobytes())
    ptr, n = C.POINTER(C.c_int8)(), C.c_size_t()
    assert L.adsb_load_u8(p, C.byref(ptr), C.byref(n)) == 0 and n.value == 128
    back = np.ctypeslib.as_array(ptr, shape=(256,)).copy()
    L.adsb_free(ptr)
    assert (back.astype(np.int16) == raw.astype(np.int16) - 128).all()
    open(p, "ab").write(b"\x00")  # half a sample
    assert L.adsb_load_u8(p, C.byref(ptr), C.byref(n)) != 0
    assert L.adsb_load_u8(str(tmp_path / "missing.bin").encode(), C.byref(ptr), C.byref(n)) != 0
