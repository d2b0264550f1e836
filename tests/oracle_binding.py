"""ctypes binding of the CPU oracle (oracle/adsb_oracle.h).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libadsb_oracle.so")

FRAME_DTYPE = np.dtype([("offset", "<u8"), ("bytes", "u1", (14,)), ("status", "u1"), ("fixed_bit", "u1")])


class OraclePacket(C.Structure):
    _fields_ = [("packet", C.c_uint8 * 14), ("downlink_format", C.c_uint8), ("capability", C.c_uint8),
                ("icao", C.c_uint32), ("msg_type", C.c_uint8), ("msg_kind", C.c_int32),
                ("callsign", C.c_char * 9), ("surveillance_status", C.c_uint8),
                ("nic_supplement", C.c_uint8), ("altitude", C.c_int32), ("cpr_time", C.c_uint8),
                ("cpr_odd", C.c_uint8), ("cpr_latitude", C.c_uint32), ("cpr_longitude", C.c_uint32),
                ("raw_msg", C.c_uint8 * 10)]


class OracleAircraftSummary(C.Structure):
    _fields_ = [("icao", C.c_uint32), ("callsign", C.c_char * 9), ("altitude", C.c_int32),
                ("has_position", C.c_int32), ("latitude", C.c_double), ("longitude", C.c_double),
                ("last_contact", C.c_double)]

    def as_tuple(self):
        return (self.icao, self.callsign.decode(), self.altitude, bool(self.has_position),
                self.latitude if self.has_position else None, self.longitude if self.has_position else None,
                self.last_contact)


class OracleTracker:
    """aircraft.rs: HashMap<u32, Aircraft> + handle_aircraft_update, restated in oracle/adsb_oracle.c."""

    def __init__(self, lib):
        self.lib = lib
        self.h = lib.oracle_tracker_create()

    def update(self, frame_bytes, time_s):
        b = np.frombuffer(bytes(frame_bytes), dtype=np.uint8).copy()
        out = OracleAircraftSummary()
        r = self.lib.oracle_tracker_update(self.h, b.ctypes.data, float(time_s), C.byref(out))
        assert r >= 0
        return bool(r), out

    def aircraft(self):
        res = []
        for k in range(self.lib.oracle_tracker_count(self.h)):
            s = OracleAircraftSummary()
            assert self.lib.oracle_tracker_get(self.h, k, C.byref(s)) == 0
            res.append(s)
        return res

    def close(self):
        if self.h:
            self.lib.oracle_tracker_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Oracle:
    E_SHORT = -1

    def __init__(self):
        self.lib = C.CDLL(LIB)
        L = self.lib
        L.oracle_get_magnitude.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_check_for_adsb_packet.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.oracle_check_for_adsb_packet.restype = C.c_int
        L.oracle_get_adsb_crc.argtypes = [C.c_void_p, C.c_size_t]
        L.oracle_get_adsb_crc.restype = C.c_uint32
        L.oracle_try_crc_recovery.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p,
                                              C.POINTER(C.c_int)]
        L.oracle_try_crc_recovery.restype = C.c_int
        L.oracle_extract_packet.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint8),
                                            C.POINTER(C.c_uint8)]
        L.oracle_extract_packet.restype = C.c_int
        L.oracle_extract_manchester_relative.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
        L.oracle_extract_manchester_relative.restype = C.c_int
        L.oracle_decode_packet.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_decode_packet.restype = C.c_int
        for name in ("oracle_process_buffer_i16", "oracle_process_buffer_i8"):
            fn = getattr(L, name)
            fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
            fn.restype = C.c_int
        L.oracle_playback_i16.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                          C.POINTER(C.c_uint64)]
        L.oracle_playback_i16.restype = C.c_int64
        L.oracle_packet_new.argtypes = [C.c_void_p, C.POINTER(OraclePacket)]
        L.oracle_packet_display.argtypes = [C.POINTER(OraclePacket), C.c_char_p, C.c_char_p, C.c_size_t]
        L.oracle_packet_display.restype = C.c_size_t
        # cpr.rs / aircraft.rs
        L.oracle_calc_num_zones.argtypes = [C.c_double]
        L.oracle_calc_num_zones.restype = C.c_uint32
        L.oracle_calculate_latitude.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_double)]
        L.oracle_calculate_latitude.restype = None
        L.oracle_calculate_longitude.argtypes = [C.c_uint32, C.c_uint32, C.c_double, C.c_int]
        L.oracle_calculate_longitude.restype = C.c_double
        L.oracle_calculate_geographic_position.argtypes = [C.c_uint32] * 4 + [C.c_int, C.POINTER(C.c_double),
                                                                              C.POINTER(C.c_double)]
        L.oracle_calculate_geographic_position.restype = C.c_int
        L.oracle_tracker_create.restype = C.c_void_p
        L.oracle_tracker_destroy.argtypes = [C.c_void_p]
        L.oracle_tracker_update.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(OracleAircraftSummary)]
        L.oracle_tracker_update.restype = C.c_int
        L.oracle_tracker_count.argtypes = [C.c_void_p]
        L.oracle_tracker_count.restype = C.c_size_t
        L.oracle_tracker_get.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(OracleAircraftSummary)]
        L.oracle_tracker_get.restype = C.c_int

    # utils.rs:46-52
    def get_magnitude(self, iq_i16):
        iq = np.ascontiguousarray(iq_i16, dtype=np.int16)
        n = iq.shape[0]
        out = np.empty(n, dtype=np.uint32)
        self.lib.oracle_get_magnitude(iq.ctypes.data, n, out.ctypes.data)
        return out

    # demod.rs:17-57 -> None or high
    def check_for_adsb_packet(self, buf32):
        b = np.ascontiguousarray(buf32, dtype=np.uint32)
        assert b.size == 32
        high = C.c_uint32()
        ok = self.lib.oracle_check_for_adsb_packet(b.ctypes.data, C.byref(high))
        return high.value if ok else None

    def get_adsb_crc(self, data):
        b = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        return self.lib.oracle_get_adsb_crc(b.ctypes.data, b.size)

    def try_crc_recovery(self, packet, calc_crc, packet_crc):
        b = np.frombuffer(bytes(packet), dtype=np.uint8).copy()
        out = np.zeros(b.size, dtype=np.uint8)
        bit = C.c_int(-1)
        ok = self.lib.oracle_try_crc_recovery(b.ctypes.data, b.size, calc_crc, packet_crc, out.ctypes.data,
                                              C.byref(bit))
        return (bytes(out), bit.value) if ok else None

    # demod.rs:65-82 -> None or (bytes, status, fixed_bit)
    def extract_packet(self, mags224, high=0):
        m = np.ascontiguousarray(mags224, dtype=np.uint32)
        assert m.size == 224
        out = np.zeros(14, dtype=np.uint8)
        st, fx = C.c_uint8(), C.c_uint8()
        ok = self.lib.oracle_extract_packet(m.ctypes.data, high, out.ctypes.data, C.byref(st), C.byref(fx))
        return (bytes(out), st.value, fx.value) if ok else None

    def extract_manchester_relative(self, mags, high=0):
        m = np.ascontiguousarray(mags, dtype=np.uint32)
        out = np.zeros(m.size // 16, dtype=np.uint16)
        ok = self.lib.oracle_extract_manchester_relative(m.ctypes.data, m.size, high, out.ctypes.data)
        return out if ok else None

    def decode_packet(self, symbols):
        s = np.ascontiguousarray(symbols, dtype=np.uint16)
        out = np.zeros(s.size, dtype=np.uint8)
        ok = self.lib.oracle_decode_packet(s.ctypes.data, s.size, out.ctypes.data)
        return bytes(out) if ok else None

    # adsb.rs:95-116 for one buffer; returns (rc, frames, n_found)
    def process_buffer(self, iq, max_out=1 << 20):
        iq = np.ascontiguousarray(iq)
        assert iq.dtype in (np.int8, np.int16)
        n = iq.shape[0] if iq.ndim == 2 else iq.size // 2
        out = np.zeros(max(max_out, 1), dtype=FRAME_DTYPE)
        found = C.c_uint64()
        fn = self.lib.oracle_process_buffer_i8 if iq.dtype == np.int8 else self.lib.oracle_process_buffer_i16
        rc = fn(iq.ctypes.data, n, out.ctypes.data, max_out, C.byref(found))
        return rc, out[:min(found.value, max_out)].copy(), found.value

    # adsb.rs:75-89 + 92-122
    def playback(self, iq_i16, chunk_len=20000, max_out=1 << 20):
        iq = np.ascontiguousarray(iq_i16, dtype=np.int16)
        n = iq.shape[0]
        out = np.zeros(max(max_out, 1), dtype=FRAME_DTYPE)
        found = C.c_uint64()
        chunks = self.lib.oracle_playback_i16(iq.ctypes.data, n, chunk_len, out.ctypes.data, max_out,
                                              C.byref(found))
        return chunks, out[:min(found.value, max_out)].copy(), found.value

    def packet_new(self, frame_bytes):
        b = np.frombuffer(bytes(frame_bytes), dtype=np.uint8).copy()
        p = OraclePacket()
        self.lib.oracle_packet_new(b.ctypes.data, C.byref(p))
        return p

    def packet_display(self, frame_bytes, time_text=""):
        p = self.packet_new(frame_bytes)
        buf = C.create_string_buffer(2048)
        n = self.lib.oracle_packet_display(C.byref(p), time_text.encode(), buf, 2048)
        return buf.value.decode()[:n]

    # cpr.rs
    def calc_num_zones(self, lat):
        return self.lib.oracle_calc_num_zones(float(lat))

    def calculate_latitude(self, even, odd, first_is_odd):
        out = (C.c_double * 3)()
        self.lib.oracle_calculate_latitude(even, odd, int(first_is_odd), out)
        return tuple(out)

    def calculate_longitude(self, even, odd, latitude, first_is_odd):
        return self.lib.oracle_calculate_longitude(even, odd, float(latitude), int(first_is_odd))

    def geographic_position(self, even_lat, even_lon, odd_lat, odd_lon, first_is_odd):
        lat, lon = C.c_double(), C.c_double()
        ok = self.lib.oracle_calculate_geographic_position(even_lat, even_lon, odd_lat, odd_lon, int(first_is_odd),
                                                           C.byref(lat), C.byref(lon))
        return (lat.value, lon.value) if ok else None

    def tracker(self):
        return OracleTracker(self.lib)
