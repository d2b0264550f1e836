/*
 * adsb_host.h -- C ABI over the host-side mirror of air_rs's thread structure (libadsb_hip.so).
 *
 * The reference is a Rust binary with no Rust toolchain in this image, so the host side above
 * include/adsb_hip.h is written in C++ (air_rs_amd/csrc/host/) with the reference's names:
 * AdsbPacket (src/adsb/packet.rs:9-99), AircraftID / AircraftPosition / UknownMsg
 * (src/adsb/msgs.rs), playback_thread (src/adsb.rs:75-89), process_sdr_data_thread
 * (src/adsb.rs:92-122), load_data/save_data (src/utils.rs:6-43).  These C entry points exist so
 * tests (ctypes) and other languages can drive that C++.
 */
#ifndef ADSB_HOST_H
#define ADSB_HOST_H

#include "adsb_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ADSB_MSG_AIRCRAFT_ID = 0, ADSB_MSG_AIRCRAFT_POSITION = 1, ADSB_MSG_UNKNOWN = 2 };

/* Flat view of AdsbPacket (packet.rs:9-18) + its AdsbMsgType variant (msgs.rs:6-11). */
typedef struct adsb_packet_view {
    uint8_t  packet[14];
    uint8_t  downlink_format;      /* packet.rs:26 */
    uint8_t  capability;           /* packet.rs:27 (mask 5, as in the reference) */
    uint32_t icao;                 /* packet.rs:28 */
    uint8_t  msg_type;             /* packet.rs:29 */
    int32_t  msg_kind;             /* ADSB_MSG_* */
    char     callsign[9];          /* AircraftID, NUL terminated */
    uint8_t  surveillance_status;  /* AircraftPosition ... */
    uint8_t  nic_supplement;
    int32_t  altitude;
    uint8_t  cpr_time;
    uint8_t  cpr_odd;
    uint32_t cpr_latitude;
    uint32_t cpr_longitude;
    uint8_t  raw_msg[10];          /* UknownMsg: packet[4..14] */
} adsb_packet_view;

/* AdsbPacket::new (packet.rs:25-49). */
int adsb_packet_new(const uint8_t bytes[14], adsb_packet_view *out);
/* AdsbPacket::_new_from_string (packet.rs:56-68): 28 hex digits. */
int adsb_packet_new_from_string(const char *hex, adsb_packet_view *out);
/* `impl Display for AdsbPacket` (packet.rs:77-99).  time_text replaces the wall-clock value of
 * the "Processed Time" line.  Returns the text length (without NUL); writes only if cap suffices. */
size_t adsb_packet_display(const uint8_t bytes[14], const char *time_text, char *dst, size_t cap);

/*
 * launch_adsb in playback + stream mode (adsb.rs:126-173): thread 1 = playback_thread over
 * `data`, thread 2 = process_sdr_data_thread on `ctx` (GPU), thread 3 collects what the stream
 * printer would print ("\n{packet}\n" per packet, adsb.rs:157).
 *   sample_type : layout of `data`; ADSB_E_ARG unless it is the sample type the ctx was created with
 *   chunk_len   : samples per buffer (20000 in the reference); the tail is dropped as in adsb.rs:77
 *   frames/max_frames/n_frames : the frames behind the packets, offsets absolute in `data`
 *   text/text_cap/text_len     : optional stream-mode text, "Processed Time" lines blanked
 * Returns ADSB_OK or the first error thread 2 met.
 */
int adsb_pipeline_playback(adsb_ctx *ctx, int sample_type, const void *data, size_t n_samples,
                           size_t chunk_len, adsb_frame *frames, size_t max_frames, size_t *n_frames,
                           uint64_t *n_buffers, char *text, size_t text_cap, size_t *text_len);

/*
 * The same three threads with carry-over switched on in thread 2 (NOT reference
 * behaviour): the last 240 samples of every buffer are prepended to the next, so frames straddling
 * two buffers -- which the reference loses (adsb.rs:95-98) -- are decoded, and the chunked stream
 * yields exactly what one long buffer of the samples actually sent would.  The ctx must have been
 * created for max_samples >= chunk_len + 240.  (Both entry points run thread 2 on the streaming front end,
 * adsb_feed_*: pinned ring, asynchronous DMA, the 240-sample tail kept on the device.)
 */
int adsb_pipeline_playback_carry(adsb_ctx *ctx, int sample_type, const void *data, size_t n_samples,
                                 size_t chunk_len, adsb_frame *frames, size_t max_frames,
                                 size_t *n_frames, uint64_t *n_buffers);


/*
 * The general form of the two entry points above, and the replay entry: what
 * `air_rs adsb -p FILE -m stream` does (main.rs:19-23 -> launch_adsb, adsb.rs:126-173) with thread 2 on the GPU.
 *   flags: ADSB_REPLAY_CARRY      thread 2 carries the last 240 samples over (not reference behaviour)
 *          ADSB_REPLAY_SEND_TAIL  the playback thread also sends the last full or partial chunk, which
 *                                 adsb.rs:77's strict `<` never sends (not reference behaviour)
 *          0 reproduces the reference: per-buffer demodulation, tail dropped.
 * adsb_replay_file reads the whole file like utils.rs:22-43 (ADSB_FILE_C16: raw little-endian i16 I,Q pairs, the
 * reference's `.c16`; the ctx must be ADSB_SAMPLE_I16) or as a raw rtl_sdr capture (ADSB_FILE_U8: unsigned bytes
 * re-centred as x - 128; the ctx must be ADSB_SAMPLE_I8; not a format the reference reads), cuts it into
 * chunk_len-sample buffers (20000 in the reference) and returns the frames (offsets absolute in the file) and the
 * stream-mode text ("\n{packet}\n" per packet, adsb.rs:157, "Processed Time" values blanked).  The ctx must have
 * been created for max_samples >= chunk_len + 240.
 */
#define ADSB_REPLAY_CARRY 0x1u
#define ADSB_REPLAY_SEND_TAIL 0x2u
#define ADSB_FILE_C16 0
#define ADSB_FILE_U8 1
int adsb_pipeline_run(adsb_ctx *ctx, int sample_type, const void *data, size_t n_samples, size_t chunk_len,
                      uint32_t flags, adsb_frame *frames, size_t max_frames, size_t *n_frames, uint64_t *n_buffers,
                      char *text, size_t text_cap, size_t *text_len);
int adsb_replay_file(adsb_ctx *ctx, const char *path, int file_format, size_t chunk_len, uint32_t flags,
                     adsb_frame *frames, size_t max_frames, size_t *n_frames, uint64_t *n_buffers,
                     uint64_t *n_samples, char *text, size_t text_cap, size_t *text_len);

/* ---- behind the channel: tracker + CPR ------------------------------------------- */

/* cpr.rs:39-54 calc_num_zones; cpr.rs:135-147 calculate_geographic_position (returns 1 = Some, 0 = None).
 * first_is_odd: the OLDER message's CPR format (`first`), 0 = Even, 1 = Odd. */
uint32_t adsb_cpr_num_zones(double latitude);
int adsb_cpr_position(uint32_t even_lat, uint32_t even_lon, uint32_t odd_lat, uint32_t odd_lon, int first_is_odd,
                      double *latitude, double *longitude);

/* AircraftSummary (aircraft.rs:14-23) */
typedef struct adsb_aircraft_summary {
    uint32_t icao;
    char     callsign[9];   /* "" while None */
    int32_t  altitude;
    int32_t  has_position;  /* geo_position.is_some() */
    double   latitude, longitude;
    double   last_contact;  /* seconds on the caller's clock (the reference: wall clock) */
} adsb_aircraft_summary;

/* The `HashMap<u32, Aircraft>` of the display threads + handle_aircraft_update (aircraft.rs:158-165). */
typedef struct adsb_tracker adsb_tracker;
adsb_tracker *adsb_tracker_create(void);
void adsb_tracker_destroy(adsb_tracker *t);
/* Feeds one packet (time_s stands in for AdsbPacket::time_processed).  Returns 1 if the packet produced a
 * new geographic position, 0 if not, < 0 on error; *out (optional) = the aircraft's summary afterwards. */
int adsb_tracker_update(adsb_tracker *t, const uint8_t bytes[14], double time_s, adsb_aircraft_summary *out);
size_t adsb_tracker_count(const adsb_tracker *t);
/* Summary of one aircraft; ADSB_E_ARG if the ICAO address has not been seen. */
int adsb_tracker_get(const adsb_tracker *t, uint32_t icao, adsb_aircraft_summary *out);

/* utils.rs:22-43 / 6-20.  adsb_load_c16 allocates *data with malloc (free with adsb_free). */
int adsb_load_c16(const char *path, int16_t **data, size_t *n_samples);
int adsb_save_c16(const char *path, const int16_t *data, size_t n_samples);
/* Raw rtl_sdr capture: interleaved unsigned bytes I,Q (zero level 127.5).  Not a reference format; samples
 * are re-centred as x - 128 into the ADSB_SAMPLE_I8 layout.  *data is malloc'ed (free with adsb_free);
 * ADSB_E_ARG for an unreadable file or an odd length. */
int adsb_load_u8(const char *path, int8_t **data, size_t *n_samples);
void adsb_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
