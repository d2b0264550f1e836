"""air_rs_amd -- MI355X-native drop-in for air_rs's IQ -> packet thread (src/adsb.rs:92-122).

The product is the HIP library ``air_rs_amd/lib/libadsb_hip.so`` behind the C ABI in
``include/adsb_hip.h``; this package is the thin Python face used by the tests and bench.py.
"""
from ._lib import (ADSB_TRACK_NEW_POSITION, ADSB_E_ARG, ADSB_E_CAPACITY, ADSB_E_NODEVICE, ADSB_E_SHORT, ADSB_E_STATE,
                   ADSB_FLAG_INCOMPLETE, ADSB_FLAG_TRUNCATED, ADSB_OK, ADSB_SAMPLE_I8, ADSB_SAMPLE_I16, AdsbError, load)
from .demod import (AIRCRAFT_DTYPE, FIELDS_DTYPE, FRAME_DTYPE, TRACK_POINT_DTYPE, WINDOW, AdsbDemod, AdsbGroup, Feed, Tracker,
                    group_plan,
                    cpr_position, packet_display, packet_new,
                    packet_new_from_string, synth_default, synth_fill_host, synth_slot, measure_feed, measure_pinned_copy)

__all__ = [
    "ADSB_OK", "ADSB_E_SHORT", "ADSB_E_ARG", "ADSB_E_CAPACITY", "ADSB_E_NODEVICE", "ADSB_E_STATE",
    "ADSB_FLAG_INCOMPLETE", "ADSB_FLAG_TRUNCATED", "ADSB_SAMPLE_I8", "ADSB_SAMPLE_I16", "AdsbError", "load", "FRAME_DTYPE",
    "FIELDS_DTYPE", "TRACK_POINT_DTYPE", "AIRCRAFT_DTYPE", "ADSB_TRACK_NEW_POSITION", "Tracker", "cpr_position",
    "WINDOW", "AdsbDemod", "AdsbGroup", "group_plan", "Feed", "packet_display", "packet_new", "packet_new_from_string",
    "synth_default", "synth_fill_host", "synth_slot", "measure_feed", "measure_pinned_copy",
]
