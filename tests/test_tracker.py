"""Behind the channel: global CPR decode (src/adsb/cpr.rs) and the per-ICAO tracker
(src/adsb/aircraft.rs).  CPU tier: the oracle restatement against the reference's own known answers
(cpr.rs:149-206, aircraft.rs:167-263), then the C++ host mirror against the oracle."""
import math

import numpy as np
import pytest

import air_rs_amd as A
from tests.traffic import random_traffic

# the reference's tests compare positions with 1e-4 degrees (cpr.rs:159,187; aircraft.rs:209-210,260-261)
REF_TOL = 1e-4


def test_kats_json_matches_oracle(oracle):
    """tests/golden/reference_kats.json carries the same reference vectors (transcribed values)."""
    import json, os
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
    for lat, want in k["cpr"]["zones"]["cases"]:
        assert oracle.calc_num_zones(lat) == want
    c = k["cpr"]["latitude"]
    assert abs(oracle.calculate_latitude(c["even_cpr_lat"], c["odd_cpr_lat"], c["first"] == "Odd")[0] - c["latitude"]) < c["tolerance"]
    c = k["cpr"]["longitude"]
    assert oracle.calculate_longitude(c["even_cpr_long"], c["odd_cpr_long"], c["latitude"], c["first"] == "Odd") == c["code_yields"]
    for case in k["aircraft"]:
        t = oracle.tracker()
        for n, h in enumerate(case["frames"]):
            _, s = t.update(bytes.fromhex(h), 0.5 * n)
        if "callsign" in case:
            assert s.callsign.decode() == case["callsign"]
        if "altitude" in case:
            assert s.altitude == case["altitude"]
        if "latitude" in case:
            assert abs(s.latitude - case["latitude"]) < case["tolerance"]
        if "longitude" in case:
            assert abs(s.longitude - case["longitude"]) < case["tolerance"]
        if "longitude_code_yields" in case:
            assert s.longitude == case["longitude_code_yields"]


def test_cpr_reference_kats(oracle):
    # cpr.rs:152-160 test_latitude_calculation
    lat = oracle.calculate_latitude(93000, 74158, first_is_odd=True)
    assert abs(lat[0] - 52.25720) < REF_TOL
    # cpr.rs:162-177 test_zone_calcuation
    for la, want in [(0.0, 59), (87.0, 2), (-87.0, 2), (90.0, 1), (-90.0, 1), (10.0, 59), (52.25720214843750, 36)]:
        assert oracle.calc_num_zones(la) == want
    # cpr.rs:179-189 test_longitude_calculation is WRONG AS WRITTEN (like the two tests of SURVEY F9): it
    # expects 3.829498291015625 = 10 * odd_cpr_lon, but for `first == Odd` the code (cpr.rs:120-122) returns
    # divisions * (m % num_zones + lon_cpr_e) = 10 * 51372/131072 = 3.91937255859375 -- which is also the
    # published answer for this message pair (the worked example of "The 1090 MHz Riddle").  The oracle
    # restates the code, so the code's value is what is pinned here; the reference's expectation is what
    # the other format would give from the same zone:
    lon = oracle.calculate_longitude(51372, 50194, 52.25720214843750, first_is_odd=True)
    assert lon == 3.91937255859375
    assert abs(10.0 * 50194 / 131072 - 3.829498291015625) < 1e-12
    # cpr.rs:191-205 test_identify_issue_with_latitude: the two zone counts agree
    l3 = oracle.calculate_latitude(23868, 38688, first_is_odd=True)
    assert oracle.calc_num_zones(l3[1]) == oracle.calc_num_zones(l3[2])


def test_tracker_reference_kats(oracle):
    # aircraft.rs:186-199: callsign and altitude from single packets
    t = oracle.tracker()
    _, s = t.update(bytes.fromhex("8d7c6b3020293532d70820fc8090"), 0.0)
    assert s.callsign.decode() == "JST250__"
    _, s = t.update(bytes.fromhex("8d7c6b30581304f388bb4455896f"), 1.0)
    assert s.altitude == 2600
    # aircraft.rs:201-212: even then odd? (formats as decoded) -> 52.2572 / 3.8295, 38000 ft
    t = oracle.tracker()
    new1, _ = t.update(bytes.fromhex("8D40621D58C386435CC412692AD6"), 0.0)
    new2, s = t.update(bytes.fromhex("8D40621D58C382D690C8AC2863A7"), 0.5)
    assert (new1, new2) == (False, True) and s.altitude == 38000 and s.has_position
    # latitude as the reference expects; its longitude expectation (3.8295) contradicts its own code, see
    # test_cpr_reference_kats
    assert abs(s.latitude - 52.25720) < REF_TOL and s.longitude == 3.91937255859375
    # aircraft.rs:214-262: -41.28965 / 174.80927, 1450 ft
    t = oracle.tracker()
    t.update(bytes.fromhex("8d7c6b30580d107903b3cabf62ab"), 0.0)
    new2, s = t.update(bytes.fromhex("8d7c6b30580d24eeaebb2dfea5bb"), 1.4)
    assert new2 and s.altitude == 1450
    # (the restatement reproduces the reference's expected values to the last digit)
    assert s.latitude == -41.28964698920816 and s.longitude == 174.80927207253197
    # aircraft.rs:68-70: a partner older than 10 s is not used
    t = oracle.tracker()
    t.update(bytes.fromhex("8d7c6b30580d107903b3cabf62ab"), 0.0)
    new2, s = t.update(bytes.fromhex("8d7c6b30580d24eeaebb2dfea5bb"), 10.5)
    assert not new2 and not s.has_position and s.altitude == 1450


def test_host_cpr_equals_oracle(oracle):
    rng = np.random.default_rng(5)
    some = 0
    for _ in range(20000):
        el, eo, ol, oo = (int(x) for x in rng.integers(0, 1 << 17, size=4))
        first_is_odd = bool(rng.integers(0, 2))
        want = oracle.geographic_position(el, eo, ol, oo, first_is_odd)
        got = A.cpr_position(el, eo, ol, oo, first_is_odd)
        assert (want is None) == (got is None)
        if want is not None:
            some += 1
            assert got == pytest.approx(want, abs=1e-12)
    assert some > 1000
    lib = A.load()
    for la in (0.0, 87.0, -87.0, 90.0, 10.0, 52.2572021484375, -41.3, 86.999):
        assert lib.adsb_cpr_num_zones(la) == oracle.calc_num_zones(la)


def test_host_tracker_equals_oracle(oracle):
    traffic = random_traffic(oracle, seed=9)
    ot, ht = oracle.tracker(), A.Tracker()
    n_new = 0
    for t, fr in traffic:
        new_o, so = ot.update(fr, t)
        new_h, sh = ht.update(fr, t)
        assert new_o == new_h
        n_new += new_o
        assert (so.icao, so.callsign, so.altitude, so.has_position) == (sh.icao, sh.callsign, sh.altitude, sh.has_position)
        if so.has_position:
            assert (sh.latitude, sh.longitude) == pytest.approx((so.latitude, so.longitude), abs=1e-12)
    assert n_new > 500 and len(ht) == len(ot.aircraft()) == 40
    for so in ot.aircraft():
        sh = ht.get(so.icao)
        assert so.callsign == sh.callsign and so.altitude == sh.altitude and so.has_position == sh.has_position


def test_golden_tracker_fixture(oracle):
    """tests/golden/tracker_traffic.npz (written by make_golden.py from the oracle): the oracle and the C++ host
    mirror both reproduce it -- per-frame new-position flags, positions, the aircraft table."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "tracker_traffic.npz"))
    assert len(z["frames"]) == 600 and int(z["new_position"].sum()) > 100
    ot, ht = oracle.tracker(), A.Tracker()
    for k, (t, fr) in enumerate(zip(z["times"], z["frames"])):
        new_o, so = ot.update(bytes(fr), float(t))
        new_h, sh = ht.update(bytes(fr), float(t))
        assert new_o == new_h == bool(z["new_position"][k])
        if new_o:
            assert (so.latitude, so.longitude) == tuple(z["position"][k])
            assert (sh.latitude, sh.longitude) == pytest.approx(tuple(z["position"][k]), abs=1e-12)
    table = sorted(ot.aircraft(), key=lambda s: s.icao)
    assert [s.icao for s in table] == list(z["icao"])
    for s, cs, alt, hp, la, lo in zip(table, z["callsign"], z["altitude"], z["has_position"], z["latitude"], z["longitude"]):
        sh = ht.get(s.icao)
        assert s.callsign == bytes(cs) == sh.callsign and s.altitude == alt == sh.altitude
        assert bool(s.has_position) == bool(hp) == bool(sh.has_position)
        if hp:
            assert (s.latitude, s.longitude) == (la, lo)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [21, 22])
def test_device_tracker_equals_oracle(gpu, oracle, seed):
    """The device tracker (adsb_track_device: rocPRIM sort by ICAO + partner search + CPR in f64) over the
    frame list the demodulator itself produced from modulated traffic, against the oracle's sequential
    restatement of aircraft.rs run over the same frames.  Positions by tolerance (device vs host libm)."""
    from tests.golden.make_golden import modulate, place
    traffic = random_traffic(oracle, seed=seed, n_aircraft=25, n_frames=2500, span_s=80.0)
    gap = 400                                    # samples between frame starts
    n = 300 + gap * len(traffic) + 600
    sps = 80.0 / (gap * len(traffic))            # the buffer spans the 80 s of traffic: 10 s = 312 frames
    items = [(300 + gap * k, modulate(fr, (80, 30), None)) for k, (_, fr) in enumerate(traffic)]
    iq = place(n, items, np.int8, floor=3, seed=seed)
    with A.AdsbDemod(max_samples=n, max_out=1 << 13) as d:
        frames, flags = d.demod(iq)
        assert flags == 0 and len(frames) >= len(traffic)
        points, aircraft = d.track(sps)
    assert len(points) == len(frames)
    ot = oracle.tracker()
    n_new = 0
    for k, fr in enumerate(frames):
        new, s = ot.update(bytes(fr["bytes"]), float(fr["offset"]) * sps)
        n_new += new
        assert bool(points[k]["flags"] & A.ADSB_TRACK_NEW_POSITION) == new, k
        assert points[k]["icao"] == s.icao
        if new:
            assert (points[k]["latitude"], points[k]["longitude"]) == pytest.approx((s.latitude, s.longitude), abs=1e-9)
    assert n_new > 400
    want = sorted(ot.aircraft(), key=lambda s: s.icao)
    assert len(aircraft) == len(want) >= 25
    counts = {}
    for fr in frames:
        b = fr["bytes"]
        icao = (int(b[1]) << 16) | (int(b[2]) << 8) | int(b[3])
        counts[icao] = counts.get(icao, 0) + 1
    for rec, s in zip(aircraft, want):
        assert rec["icao"] == s.icao and rec["n_frames"] == counts[s.icao]
        assert rec["callsign"].decode() == s.callsign.decode() and rec["altitude"] == s.altitude
        assert bool(rec["has_position"]) == bool(s.has_position)
        if s.has_position:
            assert (rec["latitude"], rec["longitude"]) == pytest.approx((s.latitude, s.longitude), abs=1e-9)
        assert (math.isnan(rec["last_contact"]) and math.isnan(s.last_contact)) or \
            rec["last_contact"] == pytest.approx(s.last_contact, abs=1e-9)


@pytest.mark.gpu
def test_device_tracker_reference_frames(gpu, oracle):
    # the reference's own message pairs (aircraft.rs:201-262), through the whole device path
    from tests.golden.make_golden import modulate, place
    pairs = ["8D40621D58C386435CC412692AD6", "8D40621D58C382D690C8AC2863A7",
             "8d7c6b30580d107903b3cabf62ab", "8d7c6b30580d24eeaebb2dfea5bb", "8d7c6b3020293532d70820fc8090"]
    items = [(300 + 500 * k, modulate(bytes.fromhex(h), (90, 20), None)) for k, h in enumerate(pairs)]
    iq = place(3400, items, np.int8)
    with A.AdsbDemod(max_samples=len(iq), max_out=256) as d:
        frames, _ = d.demod(iq)
        points, aircraft = d.track(0.5e-6)
    assert [bytes(f["bytes"]).hex() for f in frames] == [h.lower() for h in pairs]
    assert [int(p["flags"]) for p in points] == [0, 1, 0, 1, 0]
    assert abs(points[1]["latitude"] - 52.25720) < REF_TOL and abs(points[1]["longitude"] - 3.91937255859375) < 1e-9
    assert abs(points[3]["latitude"] - -41.28964698920816) < 1e-9 and abs(points[3]["longitude"] - 174.80927207253197) < 1e-9
    assert [int(a["icao"]) for a in aircraft] == [0x40621D, 0x7C6B30]
    assert aircraft[1]["callsign"] == b"JST250__" and aircraft[1]["altitude"] == 1450 and aircraft[0]["altitude"] == 38000
