"""Python face of the demodulator context (one per GPU, one per calling thread).

Mirrors the C ABI one to one; numpy arrays carry host buffers, integer device pointers (for
example ``torch.Tensor.data_ptr()``) carry HBM-resident ones.  Frames come back as a numpy
structured array with the adsb_frame layout.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L

FRAME_DTYPE = np.dtype([("offset", "<u8"), ("bytes", "u1", (14,)), ("status", "u1"),
                        ("fixed_bit", "u1")])
assert FRAME_DTYPE.itemsize == C.sizeof(L.AdsbFrame) == 24

FIELDS_DTYPE = np.dtype([("icao", "<u4"), ("altitude", "<i4"), ("cpr_latitude", "<u4"), ("cpr_longitude", "<u4"),
                         ("downlink_format", "u1"), ("capability", "u1"), ("msg_type", "u1"), ("msg_kind", "u1"),
                         ("surveillance_status", "u1"), ("nic_supplement", "u1"), ("cpr_time", "u1"),
                         ("cpr_odd", "u1"), ("callsign", "S8")])
assert FIELDS_DTYPE.itemsize == C.sizeof(L.AdsbPacketFields) == 32

WINDOW = 240  # 16 preamble + 112*2 samples (reference src/adsb.rs:98)


def synth_default(**overrides):
    cfg = L.AdsbSynthCfg()
    L.load().adsb_synth_default(C.byref(cfg))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def synth_fill_host(cfg, sample_type, channel, first_sample, n_samples):
    """Host copy of the synthetic stream: int8/int16 array of shape (n_samples, 2) = (I, Q)."""
    dt = np.int8 if sample_type == L.ADSB_SAMPLE_I8 else np.int16
    out = np.empty((n_samples, 2), dtype=dt)
    L.check(L.load().adsb_synth_fill_host(C.byref(cfg), sample_type, channel, first_sample,
                                          n_samples, out.ctypes.data), "adsb_synth_fill_host")
    return out


def measure_feed(device, sample_type, chunk, seconds=0.15):
    """(us per buffer, frames per buffer, buffers) of the streaming front end fed from pinned HOST memory, timed in C."""
    us, fr, nb = C.c_double(), C.c_double(), C.c_uint64()
    L.check(L.load().adsb_measure_feed(int(device), int(sample_type), int(chunk), float(seconds), C.byref(us), C.byref(fr),
                                       C.byref(nb)), "adsb_measure_feed")
    return us.value, fr.value, nb.value


def measure_pinned_copy(device, nbytes=64 << 20, iters=8):
    """Pinned host -> device copy rate of this box in GB/s."""
    g = C.c_double()
    L.check(L.load().adsb_measure_pinned_copy(int(device), int(nbytes), int(iters), C.byref(g)), "adsb_measure_pinned_copy")
    return g.value


def synth_slot(cfg, channel, slot):
    start = C.c_uint64()
    clean = (C.c_uint8 * 14)()
    sent = (C.c_uint8 * 14)()
    kind = C.c_int()
    r = L.load().adsb_synth_slot(C.byref(cfg), channel, slot, C.byref(start), C.byref(clean),
                                 C.byref(sent), C.byref(kind))
    if r < 0:
        raise L.AdsbError(r, "adsb_synth_slot")
    return bool(r), start.value, bytes(clean), bytes(sent), kind.value


TRACK_POINT_DTYPE = np.dtype([("latitude", "<f8"), ("longitude", "<f8"), ("icao", "<u4"), ("flags", "<u4")])
AIRCRAFT_DTYPE = np.dtype([("latitude", "<f8"), ("longitude", "<f8"), ("last_contact", "<f8"), ("icao", "<u4"),
                           ("altitude", "<i4"), ("has_position", "<u4"), ("n_frames", "<u4"), ("callsign", "S8")])


class Tracker:
    """Host mirror of the reference's `HashMap<u32, Aircraft>` + handle_aircraft_update (aircraft.rs:158-165)."""

    def __init__(self):
        self._lib = L.load()
        self._h = self._lib.adsb_tracker_create()

    def update(self, frame_bytes, time_s):
        b = np.frombuffer(bytes(frame_bytes), dtype=np.uint8).copy()
        out = L.AdsbAircraftSummary()
        r = self._lib.adsb_tracker_update(self._h, b.ctypes.data, float(time_s), C.byref(out))
        if r < 0:
            raise L.AdsbError(r, "adsb_tracker_update")
        return bool(r), out

    def get(self, icao):
        out = L.AdsbAircraftSummary()
        L.check(self._lib.adsb_tracker_get(self._h, icao, C.byref(out)), "adsb_tracker_get")
        return out

    def __len__(self):
        return self._lib.adsb_tracker_count(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.adsb_tracker_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cpr_position(even_lat, even_lon, odd_lat, odd_lon, first_is_odd):
    """cpr.rs:135-147 through the host mirror: (lat, lon) or None."""
    lib = L.load()
    lat, lon = C.c_double(), C.c_double()
    r = lib.adsb_cpr_position(even_lat, even_lon, odd_lat, odd_lon, int(first_is_odd), C.byref(lat), C.byref(lon))
    return (lat.value, lon.value) if r == 1 else None


class AdsbDemod:
    """adsb_ctx wrapper.  ``stream`` is a hipStream_t as int (e.g. torch's current stream)."""

    def __init__(self, device=0, sample_type=L.ADSB_SAMPLE_I8, max_samples=1 << 20, max_out=1 << 16,
                 max_channels=1, stream=None, host_staging=True):
        self._lib = L.load()
        cfg = L.AdsbCfg(L.ADSB_ABI_VERSION, device, sample_type, max_channels, max_samples, max_out,
                        stream, 1 if host_staging else 0, 0)
        h = C.c_void_p()
        L.check(self._lib.adsb_create(C.byref(cfg), C.byref(h)), "adsb_create")
        self._h = h
        self.sample_type = sample_type
        self.max_out = max_out
        self.max_channels = max_channels
        self._last_channels = 1   # channels of the last launch (adsb_demod is single-channel)
        self._np_dtype = np.int8 if sample_type == L.ADSB_SAMPLE_I8 else np.int16

    def close(self):
        if getattr(self, "_h", None):
            self._lib.adsb_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def stream(self):
        return self._lib.adsb_stream(self._h)

    @property
    def mag_mode(self):
        return self._lib.adsb_debug_mag_mode(self._h)

    @property
    def scan(self):
        """Which scan kernel this context launches (fixed at adsb_create by ADSB_SCAN): 'root' (floor(sqrt) per sample, u8
        magnitudes in LDS: the product's, the default; CS16's only one), or one of the A/B kernels of rounds 3-4, in
        -DADSB_AB_KERNELS=1 builds only: 'nsq', 'reg', 'code', 'sieve' (DESIGN.md sections 4.1b-d)."""
        return {0: "nsq", 1: "root", 2: "reg", 3: "code", 4: "sieve"}[self._lib.adsb_debug_scan(self._h)]

    def code_table(self):
        """The code scan's table as the device computes it: uint16[32769], c(n) | th(n) << 8."""
        out = np.empty(32769, dtype=np.uint16)
        L.check(self._lib.adsb_debug_code_table(self._h, out.ctypes.data), "adsb_debug_code_table")
        return out

    def set_launch_index(self, idx):
        """Test knob: the next launch counts as launch number idx (epoch of finish_order's exchange words)."""
        L.check(self._lib.adsb_debug_set_launch_index(self._h, int(idx) & 0xFFFFFFFF), "adsb_debug_set_launch_index")

    def finish_stall(self, blk=0xFFFFFFFF):
        """Test knob: finish_order's workgroup blk withholds its exchange word (default: none does)."""
        L.check(self._lib.adsb_debug_finish_stall(self._h, int(blk) & 0xFFFFFFFF), "adsb_debug_finish_stall")

    def pool_limit(self, on=True):
        """Test knob: the shared slot pool hands out nothing (tiles over their quota lose their slots)."""
        L.check(self._lib.adsb_debug_pool_limit(self._h, 1 if on else 0), "adsb_debug_pool_limit")

    # -- behind the channel: tracker + CPR on the device (aircraft.rs, cpr.rs) ------------------------
    def track(self, seconds_per_sample=0.5e-6):
        """Runs the tracker over the last (single-channel) launch's frame list.  Returns (points, aircraft):
        structured arrays, one point per frame (frame order) and one record per ICAO (ascending)."""
        L.check(self._lib.adsb_track_device(self._h, float(seconds_per_sample)), "adsb_track_device")
        pts = np.zeros(max(self.max_out, 1), dtype=TRACK_POINT_DTYPE)
        acs = np.zeros(max(self.max_out, 1), dtype=AIRCRAFT_DTYPE)
        npts, nac = C.c_size_t(), C.c_size_t()
        L.check(self._lib.adsb_fetch_track(self._h, pts.ctypes.data, len(pts), C.byref(npts), acs.ctypes.data,
                                           len(acs), C.byref(nac)), "adsb_fetch_track")
        return pts[:npts.value].copy(), acs[:min(nac.value, len(acs))].copy()

    def fused_pass_only(self, on=True):
        """Measurement only: following launches stop after the fused magnitude + gate pass (no frames)."""
        L.check(self._lib.adsb_debug_fused_pass_only(self._h, 1 if on else 0), "adsb_debug_fused_pass_only")

    def tile_stamps(self, max_tiles=1 << 20):
        """Diagnostic builds (-DADSB_TILE_STAMPS=1): (n_tiles, 16) uint32 of the last launch."""
        out = np.zeros((max_tiles, 16), dtype=np.uint32)
        n = C.c_size_t()
        L.check(self._lib.adsb_debug_tile_stamps(self._h, out.ctypes.data, max_tiles, C.byref(n)), "adsb_debug_tile_stamps")
        return out[:n.value].copy()

    # -- one received buffer (reference adsb.rs:95-116) -------------------------------------------
    def demod(self, iq, max_out=None):
        """iq: array of shape (n, 2) [I, Q] of the ctx sample dtype.  Returns (frames, flags)."""
        iq = np.ascontiguousarray(iq, dtype=self._np_dtype)
        n = iq.shape[0] if iq.ndim == 2 else iq.size // 2
        cap = self.max_out if max_out is None else max_out
        out = np.zeros(max(cap, 1), dtype=FRAME_DTYPE)
        n_out = C.c_size_t()
        flags = C.c_uint32()
        L.check(self._lib.adsb_demod(self._h, iq.ctypes.data, n,
                                     out.ctypes.data_as(C.POINTER(L.AdsbFrame)), cap, C.byref(n_out),
                                     C.byref(flags)), "adsb_demod")
        self._last_channels = 1
        return out[:n_out.value].copy(), flags.value

    # -- HBM-resident, asynchronous -----------------------------------------------------------------
    def demod_device_async(self, dev_ptr, n_samples, n_channels=1, channel_stride=None):
        stride = n_samples if channel_stride is None else channel_stride
        L.check(self._lib.adsb_demod_device_async(self._h, dev_ptr, n_channels, n_samples, stride),
                "adsb_demod_device_async")
        self._last_channels = int(n_channels)

    def fetch(self, max_out=None, n_channels=None):
        """Frames of the last launch: (frames, per-channel counts, total_found, flags).  adsb_fetch writes one
        count per channel of the LAST LAUNCH, so the array handed to it always holds max_channels entries; the
        first `n_channels` (default: the last launch's) are returned."""
        cap = self.max_out if max_out is None else max_out
        out = np.zeros(max(cap, 1), dtype=FRAME_DTYPE)
        n_out = C.c_size_t()
        total = C.c_uint64()
        flags = C.c_uint32()
        counts = (C.c_uint64 * max(self.max_channels, 1))()
        L.check(self._lib.adsb_fetch(self._h, out.ctypes.data_as(C.POINTER(L.AdsbFrame)), cap,
                                     C.byref(n_out), counts, C.byref(total), C.byref(flags)),
                "adsb_fetch")
        n = self._last_channels if n_channels is None else min(int(n_channels), self.max_channels)
        return out[:n_out.value].copy(), list(counts)[:n], total.value, flags.value

    def fetch_counts(self):
        n_out, total, flags = C.c_uint64(), C.c_uint64(), C.c_uint32()
        L.check(self._lib.adsb_fetch_counts(self._h, C.byref(n_out), C.byref(total), C.byref(flags)),
                "adsb_fetch_counts")
        return n_out.value, total.value, flags.value

    def result_device(self):
        frames, hdr = C.c_void_p(), C.c_void_p()
        L.check(self._lib.adsb_result_device(self._h, C.byref(frames), C.byref(hdr)),
                "adsb_result_device")
        return frames.value, hdr.value

    def decode_fields(self, max_out=None):
        """On-device AdsbPacket field decode of the last launch's frames -> structured array."""
        L.check(self._lib.adsb_decode_fields_device_async(self._h), "adsb_decode_fields_device_async")
        cap = self.max_out if max_out is None else max_out
        out = np.zeros(max(cap, 1), dtype=FIELDS_DTYPE)
        n = C.c_size_t()
        L.check(self._lib.adsb_fetch_fields(self._h, out.ctypes.data, cap, C.byref(n)), "adsb_fetch_fields")
        return out[:n.value].copy()

    def set_result_target(self, dev_ptr, nbytes):
        """Next launches write [32-byte header | frames] straight into caller-owned HBM (None: reset)."""
        L.check(self._lib.adsb_set_result_target(self._h, dev_ptr, nbytes), "adsb_set_result_target")

    def set_stream_base(self, first_sample_index):
        """Frames of the following launches carry offset = first_sample_index + index inside the buffer."""
        L.check(self._lib.adsb_set_stream_base(self._h, int(first_sample_index)), "adsb_set_stream_base")

    def stream_wait_results(self, stream):
        """Make `stream` (hipStream_t as int) wait for the last launch's ordered frame list."""
        L.check(self._lib.adsb_stream_wait_results(self._h, stream), "adsb_stream_wait_results")

    # -- measurement / test helpers -------------------------------------------------------------------
    def timing_enable(self, every=1):
        """every = N > 0: attach timing events to every N-th launch; 0/False: off."""
        L.check(self._lib.adsb_timing_enable(self._h, int(every)), "adsb_timing_enable")

    def timing_read(self):
        a, b, n = C.c_double(), C.c_double(), C.c_uint32()
        L.check(self._lib.adsb_timing_read(self._h, C.byref(a), C.byref(b), C.byref(n)),
                "adsb_timing_read")
        return a.value, b.value, n.value

    def timing_read3(self):
        """(scan_ms, decode_ms, order_ms, n_launches): the three kernels of a launch, mean per timed launch."""
        a, b, c, n = C.c_double(), C.c_double(), C.c_double(), C.c_uint32()
        L.check(self._lib.adsb_timing_read3(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)), "adsb_timing_read3")
        return a.value, b.value, c.value, n.value

    def time_read_ceiling(self, dev_ptr, nbytes, iters=10):
        ms = C.c_double()
        L.check(self._lib.adsb_time_read_ceiling(self._h, dev_ptr, nbytes, iters, C.byref(ms)),
                "adsb_time_read_ceiling")
        return ms.value

    def magnitudes(self, iq):
        iq = np.ascontiguousarray(iq, dtype=self._np_dtype)
        n = iq.shape[0]
        out = np.empty(n, dtype=np.uint16)
        L.check(self._lib.adsb_debug_magnitudes(self._h, iq.ctypes.data, n, out.ctypes.data),
                "adsb_debug_magnitudes")
        return out

    def nsq_values(self, iq):
        """I^2 + Q^2 + 72 per i8 sample through the scan kernel's packing code (0xFFFF: the two paths disagree)."""
        iq = np.ascontiguousarray(iq, dtype=np.int8)
        n = iq.shape[0]
        out = np.empty(n, dtype=np.uint16)
        L.check(self._lib.adsb_debug_nsq_values(self._h, iq.ctypes.data, n, out.ctypes.data), "adsb_debug_nsq_values")
        return out

    def synth_fill_device(self, cfg, channel, first_sample, n_samples, dev_ptr):
        L.check(self._lib.adsb_synth_fill_device(self._h, C.byref(cfg), channel, first_sample,
                                                 n_samples, dev_ptr), "adsb_synth_fill_device")

    # -- reference thread structure (playback + stream mode) ------------------------------------------
    def pipeline_playback(self, data, chunk_len=20000, max_frames=1 << 20, want_text=True):
        data = np.ascontiguousarray(data, dtype=self._np_dtype)
        n = data.shape[0]
        frames = np.zeros(max_frames, dtype=FRAME_DTYPE)
        n_frames, n_buf, text_len = C.c_size_t(), C.c_uint64(), C.c_size_t()
        cap = 1 << 26 if want_text else 0
        text = C.create_string_buffer(cap) if want_text else None
        L.check(self._lib.adsb_pipeline_playback(self._h, self.sample_type, data.ctypes.data, n,
                                                 chunk_len,
                                                 frames.ctypes.data_as(C.POINTER(L.AdsbFrame)),
                                                 max_frames, C.byref(n_frames), C.byref(n_buf), text,
                                                 cap, C.byref(text_len)), "adsb_pipeline_playback")
        txt = text.value.decode() if want_text else None
        return frames[:min(n_frames.value, max_frames)].copy(), n_buf.value, txt

    def pipeline_run(self, data, chunk_len=20000, carry=False, send_tail=False, max_frames=1 << 20, want_text=True):
        """adsb_pipeline_run: playback thread -> GPU thread 2 (streaming front end) -> stream-mode text."""
        data = np.ascontiguousarray(data, dtype=self._np_dtype)
        frames = np.zeros(max_frames, dtype=FRAME_DTYPE)
        n_frames, n_buf, text_len = C.c_size_t(), C.c_uint64(), C.c_size_t()
        cap = 1 << 26 if want_text else 0
        text = C.create_string_buffer(cap) if want_text else None
        flags = (L.ADSB_REPLAY_CARRY if carry else 0) | (L.ADSB_REPLAY_SEND_TAIL if send_tail else 0)
        L.check(self._lib.adsb_pipeline_run(self._h, self.sample_type, data.ctypes.data, data.shape[0], chunk_len, flags,
                                            frames.ctypes.data_as(C.POINTER(L.AdsbFrame)), max_frames, C.byref(n_frames),
                                            C.byref(n_buf), text, cap, C.byref(text_len)), "adsb_pipeline_run")
        return frames[:min(n_frames.value, max_frames)].copy(), n_buf.value, (text.value.decode() if want_text else None)

    def replay_file(self, path, file_format=None, chunk_len=20000, carry=False, send_tail=False, max_frames=1 << 20):
        """adsb_replay_file: `.c16` (ctx i16) or raw rtl_sdr u8 (ctx i8) file -> (frames, n_buffers, n_samples, text)."""
        if file_format is None:
            file_format = L.ADSB_FILE_C16 if self.sample_type == L.ADSB_SAMPLE_I16 else L.ADSB_FILE_U8
        frames = np.zeros(max_frames, dtype=FRAME_DTYPE)
        n_frames, n_buf, n_samp, text_len = C.c_size_t(), C.c_uint64(), C.c_uint64(), C.c_size_t()
        cap = 1 << 26
        text = C.create_string_buffer(cap)
        flags = (L.ADSB_REPLAY_CARRY if carry else 0) | (L.ADSB_REPLAY_SEND_TAIL if send_tail else 0)
        L.check(self._lib.adsb_replay_file(self._h, os.fsencode(path), file_format, chunk_len, flags,
                                           frames.ctypes.data_as(C.POINTER(L.AdsbFrame)), max_frames, C.byref(n_frames),
                                           C.byref(n_buf), C.byref(n_samp), text, cap, C.byref(text_len)),
                "adsb_replay_file")
        return frames[:min(n_frames.value, max_frames)].copy(), n_buf.value, n_samp.value, text.value.decode()

    def pipeline_playback_carry(self, data, chunk_len=20000, max_frames=1 << 20):
        """Like pipeline_playback but thread 2 carries the last 240 samples over (not reference behaviour)."""
        data = np.ascontiguousarray(data, dtype=self._np_dtype)
        frames = np.zeros(max_frames, dtype=FRAME_DTYPE)
        n_frames, n_buf = C.c_size_t(), C.c_uint64()
        L.check(self._lib.adsb_pipeline_playback_carry(self._h, self.sample_type, data.ctypes.data, data.shape[0],
                                                       chunk_len, frames.ctypes.data_as(C.POINTER(L.AdsbFrame)),
                                                       max_frames, C.byref(n_frames), C.byref(n_buf)),
                "adsb_pipeline_playback_carry")
        return frames[:min(n_frames.value, max_frames)].copy(), n_buf.value


def group_plan(n_samples, n_members):
    """adsb_group_plan: [(first_sample, n_samples, n_offsets)] per member."""
    sh = (L.AdsbGroupShard * n_members)()
    L.check(L.load().adsb_group_plan(int(n_samples), int(n_members), sh), "adsb_group_plan")
    return [(s.first_sample, s.n_samples, s.n_offsets) for s in sh]


class AdsbGroup:
    """adsb_group_*: one buffer time-sharded over several contexts / devices behind one call (native: no torch,
    no launcher).  `devices` may repeat an ordinal (several contexts on one GPU)."""

    def __init__(self, devices, sample_type=L.ADSB_SAMPLE_I8, max_samples=1 << 20, max_out=1 << 16, root=0,
                 host_staging=True):
        self._lib = L.load()
        self._devs = (C.c_int32 * len(devices))(*devices)
        cfg = L.AdsbGroupCfg(L.ADSB_ABI_VERSION, sample_type, len(devices), root, self._devs, max_samples, max_out,
                             1 if host_staging else 0, 0)
        h = C.c_void_p()
        L.check(self._lib.adsb_group_create(C.byref(cfg), C.byref(h)), "adsb_group_create")
        self._h, self.n, self.max_out, self.sample_type = h, len(devices), max_out, sample_type
        self._np_dtype = np.int8 if sample_type == L.ADSB_SAMPLE_I8 else np.int16

    def close(self):
        if getattr(self, "_h", None):
            self._lib.adsb_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def demod(self, iq, max_out=None):
        """One host buffer through all members: (frames, flags)."""
        iq = np.ascontiguousarray(iq, dtype=self._np_dtype)
        n = iq.shape[0] if iq.ndim == 2 else iq.size // 2
        cap = self.max_out if max_out is None else max_out
        out = np.zeros(max(cap, 1), dtype=FRAME_DTYPE)
        n_out, flags = C.c_size_t(), C.c_uint32()
        L.check(self._lib.adsb_group_demod(self._h, iq.ctypes.data, n, out.ctypes.data_as(C.POINTER(L.AdsbFrame)), cap,
                                           C.byref(n_out), C.byref(flags)), "adsb_group_demod")
        return out[:n_out.value].copy(), flags.value

    def demod_device_async(self, dev_ptrs, n_samples):
        arr = (C.c_void_p * self.n)(*[int(p) if p else None for p in dev_ptrs])
        L.check(self._lib.adsb_group_demod_device_async(self._h, arr, int(n_samples)), "adsb_group_demod_device_async")

    def fetch(self, max_out=None):
        cap = self.max_out if max_out is None else max_out
        out = np.zeros(max(cap, 1), dtype=FRAME_DTYPE)
        n_out, total, flags = C.c_size_t(), C.c_uint64(), C.c_uint32()
        L.check(self._lib.adsb_group_fetch(self._h, out.ctypes.data_as(C.POINTER(L.AdsbFrame)), cap, C.byref(n_out),
                                           C.byref(total), C.byref(flags)), "adsb_group_fetch")
        return out[:n_out.value].copy(), total.value, flags.value

    def result_device(self):
        """(blob device pointer, hipStream_t the merge was enqueued on)."""
        blob, stream = C.c_void_p(), C.c_void_p()
        L.check(self._lib.adsb_group_result_device(self._h, C.byref(blob), C.byref(stream)), "adsb_group_result_device")
        return blob.value, stream.value


class Feed:
    """Streaming front end (adsb_feed_*): push host buffers, pop their frames in order; two may be in flight."""

    def __init__(self, dem, max_chunk, carry=False, ring_slots=3):
        self._lib, self._dem = dem._lib, dem
        cfg = L.AdsbFeedCfg(int(max_chunk), 1 if carry else 0, int(ring_slots))
        h = C.c_void_p()
        L.check(self._lib.adsb_feed_open(dem.handle, C.byref(cfg), C.byref(h)), "adsb_feed_open")
        self._h, self.max_chunk, self.carry = h, int(max_chunk), bool(carry)

    def close(self):
        if self._h:
            self._lib.adsb_feed_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def in_flight(self):
        return self._lib.adsb_feed_in_flight(self._h)

    @property
    def ready(self):
        """True when pop() would not wait for the GPU."""
        r = self._lib.adsb_feed_ready(self._h)
        if r < 0 or r > 1:
            raise L.AdsbError(r, "adsb_feed_ready")
        return bool(r)

    def push(self, iq):
        iq = np.ascontiguousarray(iq, dtype=self._dem._np_dtype)
        n = iq.shape[0] if iq.ndim == 2 else iq.size // 2
        L.check(self._lib.adsb_feed_push(self._h, iq.ctypes.data, n), "adsb_feed_push")

    def acquire(self):
        """A pinned ring slot as a numpy array of shape (max_chunk, 2) to fill in place; then push_acquired(n)."""
        p = C.c_void_p()
        L.check(self._lib.adsb_feed_acquire(self._h, C.byref(p)), "adsb_feed_acquire")
        itemsize = np.dtype(self._dem._np_dtype).itemsize
        buf = (C.c_char * (self.max_chunk * 2 * itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=self._dem._np_dtype).reshape(self.max_chunk, 2)

    def push_acquired(self, n):
        L.check(self._lib.adsb_feed_push(self._h, None, int(n)), "adsb_feed_push")

    def pop(self, max_out=None):
        cap = self._dem.max_out if max_out is None else max_out
        out = np.zeros(max(cap, 1), dtype=FRAME_DTYPE)
        n_out, flags, first = C.c_size_t(), C.c_uint32(), C.c_uint64()
        L.check(self._lib.adsb_feed_pop(self._h, out.ctypes.data_as(C.POINTER(L.AdsbFrame)), cap, C.byref(n_out),
                                        C.byref(flags), C.byref(first)), "adsb_feed_pop")
        return out[:n_out.value].copy(), flags.value, first.value


def packet_new(frame_bytes):
    """AdsbPacket::new (packet.rs:25-49) -> AdsbPacketView."""
    b = (C.c_uint8 * 14)(*bytes(frame_bytes))
    v = L.AdsbPacketView()
    L.check(L.load().adsb_packet_new(C.byref(b), C.byref(v)), "adsb_packet_new")
    return v


def packet_new_from_string(hexstr):
    v = L.AdsbPacketView()
    L.check(L.load().adsb_packet_new_from_string(hexstr.encode(), C.byref(v)),
            "adsb_packet_new_from_string")
    return v


def packet_display(frame_bytes, time_text=""):
    b = (C.c_uint8 * 14)(*bytes(frame_bytes))
    buf = C.create_string_buffer(2048)
    n = L.load().adsb_packet_display(C.byref(b), time_text.encode(), buf, 2048)
    return buf.value.decode()[:n]
