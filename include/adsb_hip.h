/*
 * adsb_hip.h -- C ABI of the MI355X-native ADS-B demodulator (libadsb_hip.so).
 *
 * This is the drop-in boundary for air_rs's thread 2, `process_sdr_data_thread`
 * (reference: src/adsb.rs:92-122).  The reference has no FFI of its own: thread 2 is a
 * private Rust fn fed by `mpsc::Receiver<Vec<Complex<i16>>>` (adsb.rs:131) and feeding
 * `mpsc::Sender<AdsbPacket>` (adsb.rs:146).  A maintainer keeps both channels and replaces the
 * body of the `while let Ok(buf) = rx.recv()` loop (adsb.rs:95-116) with one call to
 * adsb_demod() per received Vec, then builds `AdsbPacket::new(frame.bytes.to_vec())`
 * (adsb.rs:107) for every returned frame, in the order returned.  INTEGRATION.md shows the
 * Rust `extern "C"` block and the replacement loop.
 *
 * Everything here is plain C: opaque handle, pointers and sizes, POD structs, int return
 * codes.  No exceptions cross the boundary.  A context is NOT thread-safe: one context per
 * calling thread (the reference has exactly one consumer thread, adsb.rs:147) and one per GPU.
 */
#ifndef ADSB_HIP_H
#define ADSB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADSB_ABI_VERSION 1

/* ---- return codes ----------------------------------------------------------------------- */
#define ADSB_OK 0
/* n_samples < 240: the reference panics at adsb.rs:98 (`mags.len() - 240` underflows).
 * The caller decides whether to mimic the panic. */
#define ADSB_E_SHORT (-1)
#define ADSB_E_ARG (-2)      /* NULL / misaligned / inconsistent argument                      */
#define ADSB_E_CAPACITY (-3) /* n_samples or n_channels exceeds what the ctx was created for    */
#define ADSB_E_NOMEM (-4)
#define ADSB_E_NODEVICE (-5) /* no HIP device / HIP runtime failure at create                   */
#define ADSB_E_STATE (-6)    /* fetch without a launch, etc.                                    */
/* > 0 : a hipError_t from the HIP runtime */

/* ---- flags returned by fetch/demod ------------------------------------------------------- */
/* More than max_out frames exist; the first max_out (in offset order) were returned.
 * The reference has no cap (unbounded mpsc); see SURVEY F8 for why a cap is needed. */
#define ADSB_FLAG_TRUNCATED 0x1u
/* Only ever seen by device-side consumers (adsb_result_device's header, adsb_set_result_target's blob):
 * the launch ran out of temporary frame slots (far more gate survivors than max_out + one tile: constant or
 * all-zero input), so the list holds n_out entries of which some are not written yet.  The host
 * entry points that wait for a launch (adsb_fetch, adsb_fetch_counts, adsb_fetch_fields, adsb_track_device)
 * re-run the affected tiles, complete the list IN PLACE (blob included) and clear the flag; a consumer
 * that reads the device copy directly must check it and call adsb_fetch_counts() first when it is set. */
#define ADSB_FLAG_INCOMPLETE 0x2u

/* ---- sample formats ---------------------------------------------------------------------- */
/* ADSB_SAMPLE_I16 is the reference's `Complex<i16>` memory layout: interleaved {re, im},
 * 4 bytes per sample (adsb.rs:131; file format utils.rs:22-43).
 * ADSB_SAMPLE_I8 is interleaved {re, im} int8, 2 bytes per sample (RTL-SDR class radios,
 * BASELINE.json's metric).  i8 results are by definition those of the reference path on the
 * exactly widened i16 values. */
#define ADSB_SAMPLE_I8 0
#define ADSB_SAMPLE_I16 1

/* One decoded Mode-S extended squitter.  24-byte POD, identical on host and device. */
typedef struct adsb_frame {
    uint64_t offset;    /* index i of the first preamble sample inside its buffer/channel        */
    uint8_t  bytes[14]; /* what extract_packet returns (demod.rs:65-82): 11 data + 3 CRC bytes   */
    uint8_t  status;    /* 0: CRC matched (demod.rs:81); 1: one data bit repaired (crc.rs:49-65) */
    uint8_t  fixed_bit; /* status==1: repaired bit 0..87, MSB-first; otherwise 0xFF              */
} adsb_frame;

typedef struct adsb_cfg {
    uint32_t abi_version;  /* ADSB_ABI_VERSION                                                    */
    int32_t  device;       /* HIP device ordinal                                                  */
    int32_t  sample_type;  /* ADSB_SAMPLE_I8 / ADSB_SAMPLE_I16                                    */
    uint32_t max_channels; /* >= 1: how many independent buffers one launch may carry             */
    uint64_t max_samples;  /* per channel; sizes the segment table (and the H2D staging buffer)   */
    uint64_t max_out;      /* frames kept per launch, all channels together                       */
    void    *stream;       /* hipStream_t to enqueue on; NULL: the ctx creates and owns one       */
    uint32_t host_staging; /* 1: allocate a device staging buffer so adsb_demod() (host pointers)
                              works; 0: device-resident entry points only                         */
    uint32_t reserved;
} adsb_cfg;

typedef struct adsb_ctx adsb_ctx;

/* Replaces nothing in the reference (it has no setup step); owns device buffers and stream. */
int adsb_create(const adsb_cfg *cfg, adsb_ctx **out_ctx);
void adsb_destroy(adsb_ctx *ctx);
/* Static string for a return code of this library (HIP codes: hipGetErrorString). */
const char *adsb_strerror(int code);

/*
 * adsb_demod -- one iteration of the reference loop (adsb.rs:95-116) for one received buffer.
 *   iq        : host pointer, n_samples interleaved samples of cfg.sample_type
 *   out       : host array of max_out frames, filled in ascending offset order
 *   n_out     : number of frames written
 *   flags     : ADSB_FLAG_*
 * Returns ADSB_E_SHORT for n_samples < 240 (reference panics), ADSB_OK with *n_out = 0 for
 * n_samples == 240 (reference: zero iterations).  Blocking.  Requires cfg.host_staging.
 */
int adsb_demod(adsb_ctx *ctx, const void *iq, size_t n_samples, adsb_frame *out, size_t max_out,
               size_t *n_out, uint32_t *flags);

/*
 * Device-resident, asynchronous form (roofline configs, multi-channel batch).
 *   iq_dev          : device pointer, 16-byte aligned
 *   n_channels      : independent buffers; each is its own reference buffer (offsets
 *                     0..n_samples-240 per channel; no window crosses a channel edge)
 *   n_samples       : per channel
 *   channel_stride  : samples between channel starts (>= n_samples, multiple of 8)
 * Enqueues the kernels on the ctx stream and returns; results stay on the device until
 * adsb_fetch()/adsb_result_device().
 */
int adsb_demod_device_async(adsb_ctx *ctx, const void *iq_dev, uint32_t n_channels,
                            size_t n_samples, size_t channel_stride);

/*
 * Waits for the last launch and copies the frame list to the host.
 *   out               : host array of max_out frames: channel 0's frames in ascending offset,
 *                       then channel 1's, ...
 *   n_out             : frames written (all channels)
 *   per_channel_counts: optional array of n_channels uint64 (frames per channel in `out`)
 *   total_found       : optional; number of frames that exist (> *n_out when TRUNCATED)
 */
int adsb_fetch(adsb_ctx *ctx, adsb_frame *out, size_t max_out, size_t *n_out,
               uint64_t *per_channel_counts, uint64_t *total_found, uint32_t *flags);

/* Waits for the last launch and returns only the counters (8+8+4 bytes of D2H). */
int adsb_fetch_counts(adsb_ctx *ctx, uint64_t *n_out, uint64_t *total_found, uint32_t *flags);

/*
 * Zero-copy access for device-side consumers (e.g. an RCCL gather of the packet list):
 *   frames_dev : adsb_frame[ ] in device memory
 *   header_dev : device pointer to { uint64 n_out; uint64 total_found; uint32 flags; ... }
 * Order a consumer stream behind the launch that fills them with adsb_stream_wait_results().
 * If header.flags has ADSB_FLAG_INCOMPLETE the list has holes: call adsb_fetch_counts() (it re-runs what
 * is missing and clears the flag) before using it.
 * A context alternates between two result sets, so these pointers stay valid (and unchanged) until
 * the second-next adsb_demod_device_async() on this context.
 */
int adsb_result_device(adsb_ctx *ctx, const adsb_frame **frames_dev, const void **header_dev);
/*
 * Redirects the ordered frame list of the following launches into caller-owned device memory laid
 * out as [ uint64 n_out | uint64 total_found | uint64 flags | uint64 0 | adsb_frame[...] ]
 * (16-byte aligned; capacity = (blob_bytes - 32) / 24 frames, further capped by cfg.max_out).
 * Lets a consumer fill a multi-launch bucket in place (e.g. one RCCL gather per N launches) with
 * no device-to-device copy.  blob_dev == NULL returns to the context's own buffers.
 */
int adsb_set_result_target(adsb_ctx *ctx, void *blob_dev, size_t blob_bytes);
/*
 * Position of the next launches' sample 0 inside a longer stream: every frame of the following launches is
 * reported with offset = first_sample_index + (index of its first preamble sample inside the buffer), in
 * every channel.  0 (the default) gives the reference's per-buffer offsets (adsb.rs:98).  A rank that owns
 * the slice [first, first + n) of a time-sharded stream sets this to `first`, and the per-rank lists
 * concatenate into one globally ordered list with no host-side rebasing.
 */
int adsb_set_stream_base(adsb_ctx *ctx, uint64_t first_sample_index);
/* Makes `stream` (hipStream_t) wait for the results of the last launch; does not block the host.
 * (It cannot repair an ADSB_FLAG_INCOMPLETE list: that takes the host, see adsb_fetch_counts.) */
int adsb_stream_wait_results(adsb_ctx *ctx, void *stream);

/*
 * ---- streaming front end -------------------------------------------------------------
 * The reference's thread 2 receives one Vec per recv() (src/adsb.rs:95) from the SDR reader (adsb.rs:54-73) or
 * the playback thread (adsb.rs:75-89) and treats each as an island.  A feed takes the same sequence of host
 * buffers and keeps the GPU busy across them: each buffer goes through a pinned host ring to one of two device
 * staging slots by asynchronous DMA on a copy stream, overlapped with the previous buffer's kernels; up to two
 * buffers are in flight, results come back in order from adsb_feed_pop().  The ctx must have been created with
 * max_samples >= max_chunk (+ 240 in carry mode; cfg.host_staging is not needed) and must not be used for other
 * launches while the feed is open.
 *   carry = 0 (default, the reference's behaviour): every buffer is its own reference buffer: offsets
 *     0 .. n-241 of each are examined, frames straddling two buffers are lost (adsb.rs:98); frame
 *     offsets are buffer-relative (*first_sample of adsb_feed_pop says where the buffer began in the stream); a
 *     buffer shorter than 240 samples makes adsb_feed_push return ADSB_E_SHORT (the reference panics).
 *   carry = 1 (NOT reference behaviour): the last 240 samples seen so far stay on the device and are copied,
 *     device to device, in front of the next buffer: the chunked stream decodes exactly like one long buffer of
 *     the same samples; frame offsets are absolute stream positions.
 */
typedef struct adsb_feed adsb_feed;
typedef struct adsb_feed_cfg {
    size_t   max_chunk;   /* largest buffer (samples) that will be pushed                          */
    uint32_t carry;       /* 0 = per-buffer semantics (reference), 1 = carry the 240-sample tail    */
    uint32_t ring_slots;  /* pinned host buffers of max_chunk samples (0: 3)                        */
} adsb_feed_cfg;
int adsb_feed_open(adsb_ctx *ctx, const adsb_feed_cfg *cfg, adsb_feed **out_feed);
/* Optional zero-copy producer path: a pinned ring slot (max_chunk samples) to fill in place; hand it over with
 * adsb_feed_push(feed, NULL, n).  Blocks only if the DMA out of that slot has not finished yet. */
int adsb_feed_acquire(adsb_feed *feed, void **host_slot);
/* Enqueues one buffer (copied into the ring unless it was acquired) and returns without waiting for the GPU.
 * ADSB_E_STATE when two buffers are already in flight (pop first). */
int adsb_feed_push(adsb_feed *feed, const void *iq_host, size_t n_samples);
/* Waits for the OLDEST buffer in flight and returns its frames in ascending offset order; *first_sample
 * (optional) = stream position of that buffer's first sample.  ADSB_E_STATE when nothing is in flight. */
int adsb_feed_pop(adsb_feed *feed, adsb_frame *out, size_t max_out, size_t *n_out, uint32_t *flags,
                  uint64_t *first_sample);
int adsb_feed_in_flight(const adsb_feed *feed); /* 0, 1 or 2 */
/* 1: adsb_feed_pop() would return the oldest buffer's frames without waiting for the GPU; 0: it would wait;
 * ADSB_E_STATE: nothing is in flight.  Lets a consumer hand packets on as soon as they exist instead of one buffer
 * late (the reference sends a buffer's packets before its next recv(), src/adsb.rs:95-116). */
int adsb_feed_ready(adsb_feed *feed);
void adsb_feed_close(adsb_feed *feed);

/*
 * On-device field decode of the last launch's frame list: what AdsbPacket::new
 * computes per frame (src/adsb/packet.rs:25-49, src/adsb/msgs.rs:70-102,150-201), as one 32-byte
 * record per frame, in frame order.  Lets the host wrapper only wrap when output volumes are large.
 */
typedef struct adsb_packet_fields {
    uint32_t icao;                /* packet.rs:28 */
    int32_t  altitude;            /* AircraftPosition (msgs.rs:70-75), feet; 0 otherwise */
    uint32_t cpr_latitude;        /* msgs.rs:84-86 */
    uint32_t cpr_longitude;       /* msgs.rs:87-89 */
    uint8_t  downlink_format;     /* packet.rs:26 */
    uint8_t  capability;          /* packet.rs:27 (mask 5, as in the reference) */
    uint8_t  msg_type;            /* packet.rs:29 */
    uint8_t  msg_kind;            /* 0 AircraftID, 1 AircraftPosition, 2 Uknown (msgs.rs:6-11) */
    uint8_t  surveillance_status; /* msgs.rs:78 */
    uint8_t  nic_supplement;      /* msgs.rs:79 */
    uint8_t  cpr_time;            /* msgs.rs:80 */
    uint8_t  cpr_odd;             /* msgs.rs:81-82: 1 = CprFormat::Odd */
    char     callsign[8];         /* AircraftID (msgs.rs:180-201), not NUL terminated; zeros otherwise */
} adsb_packet_fields;
/* Enqueues the decode after the last launch (ctx stream); needs cfg.max_out records of ctx memory
 * (allocated on first use). */
int adsb_decode_fields_device_async(adsb_ctx *ctx);
/* Waits and copies the records to the host; *n_out = number of frames decoded. */
int adsb_fetch_fields(adsb_ctx *ctx, adsb_packet_fields *out, size_t max_out, size_t *n_out);
/* Device pointer to the records (valid until the next decode on this ctx); does not synchronise. */
int adsb_fields_device(adsb_ctx *ctx, const adsb_packet_fields **fields_dev);

/*
 * Tracker + global CPR position decode on the device: what the reference's display
 * threads do with every AdsbPacket, `handle_aircraft_update` (src/adsb/aircraft.rs:158-165 ->
 * Aircraft::handle_packet, aircraft.rs:48-111 -> cpr::calculate_geographic_position, cpr.rs:135-147),
 * applied to the ordered frame list of the last single-channel launch, starting from an empty aircraft
 * map (like the demodulation itself, no state is carried between launches).  Packet time = frame offset
 * x seconds_per_sample (the reference stamps the wall clock; excluded from parity).  f64 arithmetic;
 * parity with the reference is by tolerance (its own tests use 1e-4 degrees).
 */
#define ADSB_TRACK_NEW_POSITION 0x1u /* this frame completed an even/odd pair: latitude/longitude valid */
typedef struct adsb_track_point {   /* one per frame, in frame order */
    double   latitude, longitude;   /* degrees; 0 unless ADSB_TRACK_NEW_POSITION */
    uint32_t icao;
    uint32_t flags;
} adsb_track_point;
typedef struct adsb_aircraft_record { /* AircraftSummary (aircraft.rs:14-23), one per ICAO, ascending ICAO */
    double   latitude, longitude;   /* geo_position, valid if has_position */
    double   last_contact;          /* seconds: time of the last position message (aircraft.rs:56); NaN if none */
    uint32_t icao;
    int32_t  altitude;              /* feet, of the last position message; 0 if none */
    uint32_t has_position;
    uint32_t n_frames;              /* frames of this aircraft in the list */
    char     callsign[8];           /* of the last identification message; zeros if none (not NUL terminated) */
} adsb_aircraft_record;
/* Runs adsb_decode_fields_device_async if needed, waits for the list's length, then enqueues the tracker
 * on the ctx stream.  ADSB_E_ARG for multi-channel launches. */
int adsb_track_device(adsb_ctx *ctx, double seconds_per_sample);
/* Waits and copies: up to max_points per-frame points (frame order) and up to max_aircraft records
 * (ascending ICAO); either array may be NULL with a zero count.  *n_aircraft is the number of distinct
 * ICAO addresses in the list even if fewer records were copied. */
int adsb_fetch_track(adsb_ctx *ctx, adsb_track_point *points, size_t max_points, size_t *n_points,
                     adsb_aircraft_record *aircraft, size_t max_aircraft, size_t *n_aircraft);

/*
 * ---- several GPUs behind one call ---------------------------------------------------------
 * The reference's thread 2 is one function on one thread (src/adsb.rs:92, spawned at adsb.rs:147); a group is the
 * drop-in for that function when the buffer should be spread over N devices: one context per member, the offsets
 * [0, n - 240) of the buffer split evenly in member order, every member reading its own offsets plus a 239-sample
 * halo (neighbouring slices overlap by 240 samples: the window is 16 + 224, adsb.rs:98,106).  Every offset is
 * independent (adsb.rs:113 skips nothing), so the members' lists, which carry absolute offsets, concatenated in
 * member order ARE the single-context list: same frames, same order.  The lists meet in the root member's device
 * memory (hipMemcpyPeerAsync) as [ uint64 n_out | uint64 total_found | uint64 flags | uint64 0 | adsb_frame[...] ].
 * Members may name the same device more than once (several contexts on one GPU).  Like a context, a group is
 * not thread-safe: one calling thread.
 */
typedef struct adsb_group adsb_group;
typedef struct adsb_group_cfg {
    uint32_t       abi_version;  /* ADSB_ABI_VERSION                                                       */
    int32_t        sample_type;  /* ADSB_SAMPLE_I8 / ADSB_SAMPLE_I16                                       */
    uint32_t       n_members;    /* 1..64 contexts                                                         */
    uint32_t       root;         /* index of the member whose device receives the merged list              */
    const int32_t *devices;      /* [n_members] HIP device ordinal of each member                          */
    uint64_t       max_samples;  /* of the WHOLE buffer                                                    */
    uint64_t       max_out;      /* frames kept per launch, whole buffer                                   */
    uint32_t       host_staging; /* 1: per-member device staging so the host-pointer entry points work     */
    uint32_t       reserved;
} adsb_group_cfg;
typedef struct adsb_group_shard { /* what one member works on */
    uint64_t first_sample;       /* its slice starts here (a multiple of 8 samples) ...                    */
    uint64_t n_samples;          /* ... and is this long: n_offsets + 240; 0 = the member has nothing      */
    uint64_t n_offsets;          /* it owns the offsets [first_sample, first_sample + n_offsets)           */
} adsb_group_shard;
int adsb_group_create(const adsb_group_cfg *cfg, adsb_group **out_group);
void adsb_group_destroy(adsb_group *group);
uint32_t adsb_group_size(const adsb_group *group);
adsb_ctx *adsb_group_member(adsb_group *group, uint32_t index); /* e.g. for adsb_synth_fill_device on its device */
/* The split a group of n_members makes of an n_samples buffer (ADSB_E_SHORT below 240 samples). */
int adsb_group_plan(uint64_t n_samples, uint32_t n_members, adsb_group_shard *shards);
/* One iteration of the reference loop (adsb.rs:95-116) for one received buffer in HOST memory, spread over the
 * members: like adsb_demod().  Blocking.  Requires cfg.host_staging. */
int adsb_group_demod(adsb_group *group, const void *iq_host, size_t n_samples, adsb_frame *out, size_t max_out,
                     size_t *n_out, uint32_t *flags);
/* The same without waiting (copies and kernels are enqueued on the members' streams). */
int adsb_group_demod_host_async(adsb_group *group, const void *iq_host, size_t n_samples);
/* Device-resident form: iq_dev[i] = member i's slice (adsb_group_plan: samples [first_sample, first_sample +
 * n_samples) of the buffer) in ITS device's memory, 16-byte aligned; NULL where n_samples is 0. */
int adsb_group_demod_device_async(adsb_group *group, const void *const *iq_dev, size_t n_samples);
/* Waits for the members, merges, copies the list to the host (ascending offset). */
int adsb_group_fetch(adsb_group *group, adsb_frame *out, size_t max_out, size_t *n_out, uint64_t *total_found,
                     uint32_t *flags);
/* Waits for the members' counts and enqueues the merge; *blob_dev = the merged [header | frames] blob on the root
 * member's device, complete once *stream (hipStream_t, on that device) has drained. */
int adsb_group_result_device(adsb_group *group, const void **blob_dev, void **stream);

/* The stream the ctx enqueues on (hipStream_t as void*). */
void *adsb_stream(adsb_ctx *ctx);
/* cfg.sample_type the ctx was created with (ADSB_SAMPLE_*); ADSB_E_ARG for NULL. */
int adsb_sample_type(const adsb_ctx *ctx);

/* ---- measurement / test helpers (bench.py, tests; not part of the reference's surface) ----- */
/*
 * With timing on (on = N > 0), every N-th adsb_demod_device_async() attaches HIP events to the
 * scan kernel's dispatch and to the finishing kernel's (on the stream they run on).  adsb_timing_read()
 * waits for the stream, returns the mean milliseconds per launch of each since the last read
 * (at most the 512 most recent launches) and clears the log.
 */
int adsb_timing_enable(adsb_ctx *ctx, int on);
/* The two kernels of a launch separately: the scan kernel (demod_tiles: magnitude + preamble/DF17 gate over every
 * sample + PPM slice of the gate's survivors -- the kernel that reads the IQ bytes) and the finishing kernel
 * (finish_order: CRC-24, single-bit repair and the ordered frame list in one pass over the survivors; reported as
 * decode_ms_mean).  order_ms_mean is 0 since round 3 (the separate ordering pass was fused into finish_order).
 * adsb_timing_read reports the scan and the finishing kernel. */
int adsb_timing_read3(adsb_ctx *ctx, double *scan_ms_mean, double *decode_ms_mean, double *order_ms_mean,
                      uint32_t *n_launches);
int adsb_timing_read(adsb_ctx *ctx, double *demod_ms_mean, double *order_ms_mean,
                     uint32_t *n_launches);
/* Pure-read HBM ceiling on this device: streams `bytes` from `buf_dev` `iters` times with 16-byte loads in each of three
 * access shapes (4 / 8 / 16 loads in flight per lane; no shape is the fastest on every box and size) and returns the mean
 * milliseconds per pass of the fastest. */
int adsb_time_read_ceiling(adsb_ctx *ctx, const void *buf_dev, size_t bytes, int iters,
                           double *ms_per_pass);
/* Measurement: what one buffer costs when it comes from HOST memory through the streaming front end (adsb_feed_*, defined
 * further down; reference: src/adsb.rs:75-89 sends 20 000-sample buffers, adsb.rs:59-64 MTU-sized ones), timed from C with a
 * context and a feed of its own on `device`: in-place producer, two buffers in flight, every list popped, for about
 * `seconds`.  PCIe-inclusive by construction; bench.py prints it next to (never as) the HBM-resident `value`. */
int adsb_measure_feed(int device, int sample_type, size_t chunk_samples, double seconds, double *us_per_buffer,
                      double *frames_per_buffer, uint64_t *buffers);
/* Measurement: pinned host -> device copy rate of this box (GB/s), the ceiling of any host-fed path. */
int adsb_measure_pinned_copy(int device, size_t bytes, int iters, double *gbytes_per_s);
/* floor(sqrt(I^2+Q^2)) of n host samples through the device magnitude code (utils.rs:46-52). */
int adsb_debug_magnitudes(adsb_ctx *ctx, const void *iq_host, size_t n_samples,
                          uint16_t *mags_host);
/* How v_cvt_pk_u8_f32 was found to round on this device: 0 truncates, 1 truncates under
 * MODE.fp_round = toward-zero, 2 rounds to nearest (kernel subtracts 0.5 first). */
int adsb_debug_mag_mode(adsb_ctx *ctx);
/* v = I^2 + Q^2 + 72 of n host i8 samples through the packing code of the scan kernel's phase 1 (the gate and the
 * slicer of the i8 path work on these; utils.rs:46-52's root is only taken where a comparison needs it).  0xFFFF marks
 * a sample whose two packing paths disagree.  ADSB_E_STATE for a CS16 context. */
int adsb_debug_nsq_values(adsb_ctx *ctx, const void *iq_host, size_t n_samples, uint16_t *vals_host);
/* Measurement only: with on != 0 the following launches run the scan kernel (magnitude + preamble/DF17 gate + slice
 * of the survivors) but not the finishing kernel -- no survivor is CRC-checked, the header reports an empty list, so
 * NO frames come out.  on = 0 restores the full path. */
int adsb_debug_fused_pass_only(adsb_ctx *ctx, int on);
/* Which scan kernel an i8 context launches: 1 = floor(sqrt) per sample, the product's and the default (ADSB_SCAN=root or
 * unset in the environment at adsb_create); 0 / 2 / 3 / 4 = the A/B kernels of rounds 3-4 (ADSB_SCAN=nsq / reg / code / sieve;
 * only in builds with -DADSB_AB_KERNELS=1: the gate on I^2+Q^2, the same from registers, on an 8-bit log code, on two relation
 * bits per sample).  Same results; DESIGN.md sections 4.1b-d have the measurements.  Always 1 for CS16 (one kernel). */
int adsb_debug_scan(adsb_ctx *ctx);
/* The code scan's table as this device computes it, for n = I^2+Q^2 = 0 .. 32768: out[n] = c(n) | th(n) << 8 (c = the
 * 8-bit code of n, th = the threshold code of the gate's slack for a "high" of that code).  Returns ADSB_E_STATE if the
 * table does not have the properties the kernel's superset test rests on (adsb_create checks the same). */
int adsb_debug_code_table(adsb_ctx *ctx, uint16_t *out32769);
/* Test knob: with on != 0 the shared slot pool of the following launches hands out nothing, so every tile with
 * more gate survivors than its own 32 slots loses them: the launch's list comes out with ADSB_FLAG_INCOMPLETE (for
 * device-side consumers) and the host entry points take their re-run path -- deterministically. */
int adsb_debug_pool_limit(adsb_ctx *ctx, int on);
/* Test knob: the next launch counts as launch number `idx`.  finish_order tags its exchange words with the launch's epoch
 * ((index + 1) mod 2^30) and the library zeroes them whenever the epoch wraps: this lets a test cross the wrap. */
int adsb_debug_set_launch_index(adsb_ctx *ctx, uint32_t idx);
/* Test knob: workgroup `blk` of finish_order withholds its exchange word in the following launches (0xFFFFFFFF: none).
 * The workgroups behind it give up after ~0.1 s: adsb_fetch / adsb_fetch_counts return ADSB_E_STATE, the header carries
 * ADSB_FLAG_INCOMPLETE, nothing hangs, and the context stays usable (the reference's only failure mode is a closed
 * channel, src/adsb.rs:108-111: the replacement must not add a hang). */
int adsb_debug_finish_stall(adsb_ctx *ctx, uint32_t blk);
/* Diagnostic builds of demod_tiles (-DADSB_TILE_STAMPS=1) only: 16 uint32 per tile of the last launch (waves 0
 * and 3 of the tile's workgroup, 8 each: shader cycles in prologue, phase 1, barrier, phase 2, barrier, phase 3,
 * wait for the loads; s_memrealtime at start).  ADSB_E_STATE in a normal build. */
int adsb_debug_tile_stamps(adsb_ctx *ctx, uint32_t *out16_per_tile, size_t max_tiles, size_t *n_tiles);
/* ---- deterministic synthetic IQ source --------------------------------------- */
typedef struct adsb_synth_cfg {
    uint64_t seed;
    uint32_t slot_len;     /* one frame slot per slot_len samples (>= 256); 2000 = ~1000 msg/s */
    uint32_t frame_pct;    /* 0..100: share of slots that carry a frame                        */
    uint32_t pct_flip_data;/* of the frames: one flipped data bit  (must be repaired)          */
    uint32_t pct_flip_crc; /* of the frames: one flipped CRC bit   (must be rejected)          */
    uint32_t pct_flip_two; /* of the frames: two flipped data bits (must be rejected)          */
    uint32_t noise_div;    /* noise = (sum of 4 hash bytes - 510) / noise_div; 18 -> sigma~8   */
    uint32_t amp_shift;    /* i16 only: left shift applied to the i8-scale value (0..7)        */
    uint32_t reserved;
} adsb_synth_cfg;

void adsb_synth_default(adsb_synth_cfg *cfg);
/* Sample k of channel `channel` depends only on (cfg, channel, k): any slice of the stream can be
 * generated independently (time-sharding across GPUs needs no input exchange). */
int adsb_synth_fill_host(const adsb_synth_cfg *cfg, int sample_type, uint32_t channel,
                         uint64_t first_sample, size_t n_samples, void *iq_host);
int adsb_synth_fill_device(adsb_ctx *ctx, const adsb_synth_cfg *cfg, uint32_t channel,
                           uint64_t first_sample, size_t n_samples, void *iq_dev);
/* The frame (and what the demodulator must make of it) planted in slot `slot`:
 * returns 1 if the slot carries a frame; start = first preamble sample (absolute),
 * clean14 = the error-free frame, kind: 0 clean, 1 data-bit flip, 2 crc-bit flip, 3 two flips. */
int adsb_synth_slot(const adsb_synth_cfg *cfg, uint32_t channel, uint64_t slot, uint64_t *start,
                    uint8_t clean14[14], uint8_t sent14[14], int *kind);

#ifdef __cplusplus
}
#endif
#endif /* ADSB_HIP_H */
