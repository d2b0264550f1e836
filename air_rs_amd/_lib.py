"""ctypes binding of libadsb_hip.so (include/adsb_hip.h, include/adsb_host.h).

There is no CPU fallback: if the HIP library has not been built this module raises, and without
a HIP device ``adsb_create`` returns ADSB_E_NODEVICE.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ADSB_HIP_LIB lets tuning experiments point at another build of the same library
LIB_PATH = os.environ.get("ADSB_HIP_LIB") or os.path.join(_HERE, "lib", "libadsb_hip.so")

ADSB_ABI_VERSION = 1
ADSB_OK = 0
ADSB_E_SHORT = -1
ADSB_E_ARG = -2
ADSB_E_CAPACITY = -3
ADSB_E_NOMEM = -4
ADSB_E_NODEVICE = -5
ADSB_E_STATE = -6
ADSB_FLAG_TRUNCATED = 0x1
ADSB_FLAG_INCOMPLETE = 0x2
ADSB_SAMPLE_I8 = 0
ADSB_SAMPLE_I16 = 1
ADSB_MSG_AIRCRAFT_ID, ADSB_MSG_AIRCRAFT_POSITION, ADSB_MSG_UNKNOWN = 0, 1, 2
ADSB_REPLAY_CARRY, ADSB_REPLAY_SEND_TAIL = 0x1, 0x2
ADSB_FILE_C16, ADSB_FILE_U8 = 0, 1


class AdsbFrame(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("bytes", C.c_uint8 * 14), ("status", C.c_uint8),
                ("fixed_bit", C.c_uint8)]


class AdsbCfg(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("sample_type", C.c_int32),
                ("max_channels", C.c_uint32), ("max_samples", C.c_uint64), ("max_out", C.c_uint64),
                ("stream", C.c_void_p), ("host_staging", C.c_uint32), ("reserved", C.c_uint32)]


class AdsbFeedCfg(C.Structure):
    _fields_ = [("max_chunk", C.c_size_t), ("carry", C.c_uint32), ("ring_slots", C.c_uint32)]


class AdsbGroupCfg(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("sample_type", C.c_int32), ("n_members", C.c_uint32), ("root", C.c_uint32),
                ("devices", C.POINTER(C.c_int32)), ("max_samples", C.c_uint64), ("max_out", C.c_uint64),
                ("host_staging", C.c_uint32), ("reserved", C.c_uint32)]


class AdsbGroupShard(C.Structure):
    _fields_ = [("first_sample", C.c_uint64), ("n_samples", C.c_uint64), ("n_offsets", C.c_uint64)]


class AdsbSynthCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("slot_len", C.c_uint32), ("frame_pct", C.c_uint32),
                ("pct_flip_data", C.c_uint32), ("pct_flip_crc", C.c_uint32),
                ("pct_flip_two", C.c_uint32), ("noise_div", C.c_uint32), ("amp_shift", C.c_uint32),
                ("reserved", C.c_uint32)]


class AdsbPacketFields(C.Structure):
    _fields_ = [("icao", C.c_uint32), ("altitude", C.c_int32), ("cpr_latitude", C.c_uint32),
                ("cpr_longitude", C.c_uint32), ("downlink_format", C.c_uint8), ("capability", C.c_uint8),
                ("msg_type", C.c_uint8), ("msg_kind", C.c_uint8), ("surveillance_status", C.c_uint8),
                ("nic_supplement", C.c_uint8), ("cpr_time", C.c_uint8), ("cpr_odd", C.c_uint8),
                ("callsign", C.c_char * 8)]


class AdsbTrackPoint(C.Structure):
    _fields_ = [("latitude", C.c_double), ("longitude", C.c_double), ("icao", C.c_uint32), ("flags", C.c_uint32)]


class AdsbAircraftRecord(C.Structure):
    _fields_ = [("latitude", C.c_double), ("longitude", C.c_double), ("last_contact", C.c_double),
                ("icao", C.c_uint32), ("altitude", C.c_int32), ("has_position", C.c_uint32),
                ("n_frames", C.c_uint32), ("callsign", C.c_char * 8)]


class AdsbAircraftSummary(C.Structure):
    _fields_ = [("icao", C.c_uint32), ("callsign", C.c_char * 9), ("altitude", C.c_int32),
                ("has_position", C.c_int32), ("latitude", C.c_double), ("longitude", C.c_double),
                ("last_contact", C.c_double)]


ADSB_TRACK_NEW_POSITION = 0x1


class AdsbPacketView(C.Structure):
    _fields_ = [("packet", C.c_uint8 * 14), ("downlink_format", C.c_uint8), ("capability", C.c_uint8),
                ("icao", C.c_uint32), ("msg_type", C.c_uint8), ("msg_kind", C.c_int32),
                ("callsign", C.c_char * 9), ("surveillance_status", C.c_uint8),
                ("nic_supplement", C.c_uint8), ("altitude", C.c_int32), ("cpr_time", C.c_uint8),
                ("cpr_odd", C.c_uint8), ("cpr_latitude", C.c_uint32), ("cpr_longitude", C.c_uint32),
                ("raw_msg", C.c_uint8 * 10)]


# name -> (restype, argtypes); every function include/*.h declares
_P = C.POINTER
PROTOTYPES = {
    "adsb_create": (C.c_int, [_P(AdsbCfg), _P(C.c_void_p)]),
    "adsb_destroy": (None, [C.c_void_p]),
    "adsb_strerror": (C.c_char_p, [C.c_int]),
    "adsb_demod": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, _P(AdsbFrame), C.c_size_t,
                             _P(C.c_size_t), _P(C.c_uint32)]),
    "adsb_demod_device_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.c_size_t]),
    "adsb_fetch": (C.c_int, [C.c_void_p, _P(AdsbFrame), C.c_size_t, _P(C.c_size_t), _P(C.c_uint64),
                             _P(C.c_uint64), _P(C.c_uint32)]),
    "adsb_fetch_counts": (C.c_int, [C.c_void_p, _P(C.c_uint64), _P(C.c_uint64), _P(C.c_uint32)]),
    "adsb_result_device": (C.c_int, [C.c_void_p, _P(C.c_void_p), _P(C.c_void_p)]),
    "adsb_decode_fields_device_async": (C.c_int, [C.c_void_p]),
    "adsb_fetch_fields": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "adsb_fields_device": (C.c_int, [C.c_void_p, _P(C.c_void_p)]),
    "adsb_track_device": (C.c_int, [C.c_void_p, C.c_double]),
    "adsb_fetch_track": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, _P(C.c_size_t), C.c_void_p, C.c_size_t,
                                   _P(C.c_size_t)]),
    "adsb_cpr_num_zones": (C.c_uint32, [C.c_double]),
    "adsb_cpr_position": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _P(C.c_double),
                                    _P(C.c_double)]),
    "adsb_tracker_create": (C.c_void_p, []),
    "adsb_tracker_destroy": (None, [C.c_void_p]),
    "adsb_tracker_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, _P(AdsbAircraftSummary)]),
    "adsb_tracker_count": (C.c_size_t, [C.c_void_p]),
    "adsb_tracker_get": (C.c_int, [C.c_void_p, C.c_uint32, _P(AdsbAircraftSummary)]),
    "adsb_stream": (C.c_void_p, [C.c_void_p]),
    "adsb_sample_type": (C.c_int, [C.c_void_p]),
    "adsb_stream_wait_results": (C.c_int, [C.c_void_p, C.c_void_p]),
    "adsb_feed_open": (C.c_int, [C.c_void_p, _P(AdsbFeedCfg), _P(C.c_void_p)]),
    "adsb_feed_acquire": (C.c_int, [C.c_void_p, _P(C.c_void_p)]),
    "adsb_feed_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "adsb_feed_pop": (C.c_int, [C.c_void_p, _P(AdsbFrame), C.c_size_t, _P(C.c_size_t), _P(C.c_uint32), _P(C.c_uint64)]),
    "adsb_feed_in_flight": (C.c_int, [C.c_void_p]),
    "adsb_feed_ready": (C.c_int, [C.c_void_p]),
    "adsb_feed_close": (None, [C.c_void_p]),
    "adsb_group_create": (C.c_int, [_P(AdsbGroupCfg), _P(C.c_void_p)]),
    "adsb_group_destroy": (None, [C.c_void_p]),
    "adsb_group_size": (C.c_uint32, [C.c_void_p]),
    "adsb_group_member": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "adsb_group_plan": (C.c_int, [C.c_uint64, C.c_uint32, _P(AdsbGroupShard)]),
    "adsb_group_demod": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, _P(AdsbFrame), C.c_size_t, _P(C.c_size_t),
                                   _P(C.c_uint32)]),
    "adsb_group_demod_host_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "adsb_group_demod_device_async": (C.c_int, [C.c_void_p, _P(C.c_void_p), C.c_size_t]),
    "adsb_group_fetch": (C.c_int, [C.c_void_p, _P(AdsbFrame), C.c_size_t, _P(C.c_size_t), _P(C.c_uint64), _P(C.c_uint32)]),
    "adsb_group_result_device": (C.c_int, [C.c_void_p, _P(C.c_void_p), _P(C.c_void_p)]),
    "adsb_set_result_target": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "adsb_set_stream_base": (C.c_int, [C.c_void_p, C.c_uint64]),
    "adsb_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "adsb_timing_read": (C.c_int, [C.c_void_p, _P(C.c_double), _P(C.c_double), _P(C.c_uint32)]),
    "adsb_timing_read3": (C.c_int, [C.c_void_p, _P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_uint32)]),
    "adsb_time_read_ceiling": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, _P(C.c_double)]),
    "adsb_debug_magnitudes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "adsb_debug_mag_mode": (C.c_int, [C.c_void_p]),
    "adsb_debug_fused_pass_only": (C.c_int, [C.c_void_p, C.c_int]),
    "adsb_debug_nsq_values": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "adsb_debug_scan": (C.c_int, [C.c_void_p]),
    "adsb_debug_code_table": (C.c_int, [C.c_void_p, C.c_void_p]),
    "adsb_measure_feed": (C.c_int, [C.c_int, C.c_int, C.c_size_t, C.c_double, _P(C.c_double), _P(C.c_double), _P(C.c_uint64)]),
    "adsb_measure_pinned_copy": (C.c_int, [C.c_int, C.c_size_t, C.c_int, _P(C.c_double)]),
    "adsb_debug_set_launch_index": (C.c_int, [C.c_void_p, C.c_uint32]),
    "adsb_debug_finish_stall": (C.c_int, [C.c_void_p, C.c_uint32]),
    "adsb_debug_pool_limit": (C.c_int, [C.c_void_p, C.c_int]),
    "adsb_debug_tile_stamps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "adsb_synth_default": (None, [_P(AdsbSynthCfg)]),
    "adsb_synth_fill_host": (C.c_int, [_P(AdsbSynthCfg), C.c_int, C.c_uint32, C.c_uint64, C.c_size_t,
                                       C.c_void_p]),
    "adsb_synth_fill_device": (C.c_int, [C.c_void_p, _P(AdsbSynthCfg), C.c_uint32, C.c_uint64,
                                         C.c_size_t, C.c_void_p]),
    "adsb_synth_slot": (C.c_int, [_P(AdsbSynthCfg), C.c_uint32, C.c_uint64, _P(C.c_uint64),
                                  _P(C.c_uint8 * 14), _P(C.c_uint8 * 14), _P(C.c_int)]),
    # include/adsb_host.h
    "adsb_packet_new": (C.c_int, [_P(C.c_uint8 * 14), _P(AdsbPacketView)]),
    "adsb_packet_new_from_string": (C.c_int, [C.c_char_p, _P(AdsbPacketView)]),
    "adsb_packet_display": (C.c_size_t, [_P(C.c_uint8 * 14), C.c_char_p, C.c_char_p, C.c_size_t]),
    "adsb_pipeline_playback": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t,
                                         _P(AdsbFrame), C.c_size_t, _P(C.c_size_t), _P(C.c_uint64),
                                         C.c_char_p, C.c_size_t, _P(C.c_size_t)]),
    "adsb_pipeline_playback_carry": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t,
                                               _P(AdsbFrame), C.c_size_t, _P(C.c_size_t), _P(C.c_uint64)]),
    "adsb_pipeline_run": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32,
                                    _P(AdsbFrame), C.c_size_t, _P(C.c_size_t), _P(C.c_uint64),
                                    C.c_char_p, C.c_size_t, _P(C.c_size_t)]),
    "adsb_replay_file": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_size_t, C.c_uint32, _P(AdsbFrame), C.c_size_t,
                                   _P(C.c_size_t), _P(C.c_uint64), _P(C.c_uint64), C.c_char_p, C.c_size_t, _P(C.c_size_t)]),
    "adsb_load_c16": (C.c_int, [C.c_char_p, _P(_P(C.c_int16)), _P(C.c_size_t)]),
    "adsb_save_c16": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t]),
    "adsb_load_u8": (C.c_int, [C.c_char_p, _P(_P(C.c_int8)), _P(C.c_size_t)]),
    "adsb_free": (None, [C.c_void_p]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """One process must not hold two HIP runtimes: the second one to initialise finds no device.

    A PyTorch-ROCm wheel ships its own libamdhip64.so (SONAME libamdhip64.so.7, the same as /opt/rocm's) and
    loads it by file name, so `import torch` AFTER libadsb_hip.so has pulled in the system copy gives the process
    a second runtime ("No HIP GPUs are available"), while the other order is fine: the loader then binds
    libadsb_hip.so's NEEDED libamdhip64.so.7 to the copy that is already there.  When a torch wheel with a
    bundled runtime is installed, load that copy first, so that whichever of the two is imported first they
    share one runtime.  torch itself is not imported.  ADSB_HIP_SYSTEM_RUNTIME=1 opts out.
    """
    if os.environ.get("ADSB_HIP_SYSTEM_RUNTIME") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(bundled):
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
    except Exception:
        pass  # best effort: without it the import order decides, as before


def load():
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with ./build.sh (or __graft_entry__.build()). "
            "air_rs_amd has no CPU fallback for the demodulation path.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    lenient = os.environ.get("ADSB_HIP_LIB_LENIENT") == "1"  # A/B runs against libraries built from older sources
    for name, (res, args) in PROTOTYPES.items():
        if lenient and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def strerror(code):
    return load().adsb_strerror(int(code)).decode()


class AdsbError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = int(code)
        super().__init__(f"{where}: {strerror(code)} (code {int(code)})")


def check(code, where=""):
    if code != ADSB_OK:
        raise AdsbError(code, where)
