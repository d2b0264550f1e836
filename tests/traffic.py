"""Synthetic Mode-S traffic for the tracker tests and fixtures: DF17 position / identification frames with
correct CRCs, from a few aircraft that move slowly, time-ordered."""
import math

import numpy as np


def position_frame(oracle, icao, odd, cpr_lat, cpr_lon, alt_code=0x3A8, tc=11):
    """A DF17 airborne-position frame (msgs.rs:70-102 layout) with a correct CRC (crc.rs:10-40)."""
    me = bytearray(7)
    me[0] = (tc << 3)
    me[1] = (alt_code >> 4) & 0xFF          # 12-bit altitude code: bits 7..1 of m1 + q bit, high nibble of m2
    me[2] = ((alt_code & 0xF) << 4) | (int(odd) << 2) | ((cpr_lat >> 15) & 0x3)
    me[3] = (cpr_lat >> 7) & 0xFF
    me[4] = ((cpr_lat & 0x7F) << 1) | ((cpr_lon >> 16) & 0x1)
    me[5] = (cpr_lon >> 8) & 0xFF
    me[6] = cpr_lon & 0xFF
    data = bytes([0x8D, (icao >> 16) & 0xFF, (icao >> 8) & 0xFF, icao & 0xFF]) + bytes(me)
    crc = oracle.get_adsb_crc(data)
    return data + bytes([(crc >> 16) & 0xFF, (crc >> 8) & 0xFF, crc & 0xFF])


def ident_frame(oracle, icao, chars6):
    """A DF17 identification frame (TC 4): eight 6-bit characters (msgs.rs:150-177)."""
    bits = 0
    for c in chars6:
        bits = (bits << 6) | (int(c) & 0x3F)
    me = bytes([4 << 3]) + bits.to_bytes(6, "big")
    data = bytes([0x8D, (icao >> 16) & 0xFF, (icao >> 8) & 0xFF, icao & 0xFF]) + me
    crc = oracle.get_adsb_crc(data)
    return data + bytes([(crc >> 16) & 0xFF, (crc >> 8) & 0xFF, crc & 0xFF])


def random_traffic(oracle, seed, n_aircraft=40, n_frames=3000, span_s=60.0):
    """A time-ordered list of (time_s, frame) from a few aircraft: mostly position messages with plausible
    (consistent) even/odd CPR pairs, some identification messages, some long silences."""
    rng = np.random.default_rng(seed)
    icaos = rng.choice(np.arange(0x400000, 0x800000), size=n_aircraft, replace=False)
    lat = rng.uniform(-80, 80, n_aircraft)
    lon = rng.uniform(-180, 180, n_aircraft)
    out = []
    times = np.sort(rng.uniform(0, span_s, n_frames))
    for t in times:
        a = int(rng.integers(0, n_aircraft))
        if rng.random() < 0.1:
            out.append((float(t), ident_frame(oracle, int(icaos[a]), list(rng.integers(1, 27, size=8)))))
            continue
        odd = bool(rng.integers(0, 2))
        # CPR encode (the inverse of cpr.rs, ICAO Doc 9871): enough to make pairs decode to sensible places
        dlat = 360.0 / (59 if odd else 60)
        yz = math.floor(131072 * ((lat[a] % dlat) / dlat) + 0.5)
        rlat = dlat * (yz / 131072 + math.floor(lat[a] / dlat))
        nl = max(oracle.calc_num_zones(rlat) - (1 if odd else 0), 1)
        dlon = 360.0 / nl
        xz = math.floor(131072 * ((lon[a] % dlon) / dlon) + 0.5)
        out.append((float(t), position_frame(oracle, int(icaos[a]), odd, int(yz) & 0x1FFFF, int(xz) & 0x1FFFF,
                                             alt_code=int(rng.integers(0, 1 << 12)))))
        lat[a] += rng.normal(0, 0.002)
        lon[a] += rng.normal(0, 0.002)
    return out
