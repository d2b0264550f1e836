/*
 * adsb_oracle.h -- CPU ORACLE for the air_rs IQ->packet path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a literal plain-C restatement of the reference's thread 2
 * (jaxsonpd/air_rs src/adsb.rs:92-122 and its callees).  It exists so the HIP
 * path can be checked bit-for-bit.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product library
 * (air_rs_amd/csrc) never links, includes or calls anything in oracle/.
 *
 * Parity status ("what pins this oracle"):
 *   - The reference is Rust; there is no Rust toolchain in this image, so the
 *     reference cannot be compiled or run here (oracle/_ref is unbuildable).
 *   - PINNED by the reference's own known-answer tests (tests/test_oracle_kats.py):
 *     CRC-24 (demod.rs:337-367), preamble/DF gate (demod.rs:250-278),
 *     slicer+CRC+recovery negative case (demod.rs:369-380), message field
 *     decode (msgs.rs:229-320), seven whole frames (aircraft.rs:188-254,
 *     demod.rs:339-342) and the Display dump in aircraft.rs:216-251.
 *   - PARITY UNPINNED by any reference test (pinned by source reading only):
 *     get_magnitude truncation (utils.rs:46-52), the live *_relative slicer
 *     (demod.rs:92-131), successful CRC recovery (crc.rs:49-65) and the
 *     per-buffer loop bounds / no-skip behaviour (adsb.rs:98,113).
 *
 * Each function cites the reference file:line it follows.
 */
#ifndef ADSB_ORACLE_H
#define ADSB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same 24-byte POD as the product's adsb_frame (include/adsb_hip.h). */
typedef struct {
    uint64_t offset;     /* sample index i of the preamble start inside the buffer */
    uint8_t  bytes[14];  /* the 112-bit frame as emitted by extract_packet */
    uint8_t  status;     /* 0 = CRC matched, 1 = one data bit flipped by try_crc_recovery */
    uint8_t  fixed_bit;  /* status==1: index 0..87 of the flipped bit (MSB-first); else 0xFF */
} oracle_frame;

/* utils.rs:46-52 -- floor(sqrt(re^2+im^2)) through f64, then `as u32`. */
void oracle_get_magnitude(const int16_t *iq_interleaved, size_t n, uint32_t *mags);

/* demod.rs:17-57 -- returns 1 and *high for Some((high,0,0)), 0 for None. */
int oracle_check_for_adsb_packet(const uint32_t buf[32], uint32_t *high);

/* demod.rs:92-131 -- always Some; 224 mags -> 14 u16 symbols. */
int oracle_extract_manchester_relative(const uint32_t *buf, size_t buf_len, uint32_t high,
                                       uint16_t *symbols);

/* demod.rs:180-201 -- always Some; symbols -> bytes. */
int oracle_decode_packet(const uint16_t *symbols, size_t n, uint8_t *bytes);

/* crc.rs:10-40 -- bit-vector long division by 0x1FFF409. */
uint32_t oracle_get_adsb_crc(const uint8_t *buf, size_t len);

/* crc.rs:49-65 -- returns 1 and writes augmented packet (+ flipped bit index) or 0. */
int oracle_try_crc_recovery(const uint8_t *buf, size_t len, uint32_t calc_crc,
                            uint32_t packet_crc, uint8_t *out, int *flipped_bit);

/* demod.rs:65-82 -- returns 1 for Some(packet). status/fixed_bit as in oracle_frame. */
int oracle_extract_packet(const uint32_t *buf224, uint32_t high, uint8_t out[14],
                          uint8_t *status, uint8_t *fixed_bit);

/*
 * adsb.rs:95-116 -- the body of one `while let Ok(buf) = rx.recv()` iteration:
 * every offset 0..len-240 independently (the `_i += 240` at :113 has no effect).
 * Returns ADSB_ORACLE_E_SHORT (-1) where the reference would panic (len < 240).
 * Writes at most max_out frames, *n_found gets the total number found.
 */
#define ADSB_ORACLE_E_SHORT (-1)
int oracle_process_buffer_i16(const int16_t *iq_interleaved, size_t n_samples,
                              oracle_frame *out, size_t max_out, uint64_t *n_found);
/* i8 input: widened exactly to i16 then the same path. */
int oracle_process_buffer_i8(const int8_t *iq_interleaved, size_t n_samples,
                             oracle_frame *out, size_t max_out, uint64_t *n_found);

/*
 * adsb.rs:75-89 + 92-122 -- playback chunking: `while i < len-20000` sends
 * data[i..i+20000]; each chunk is processed as its own buffer; the tail is never
 * sent.  Offsets reported are absolute (chunk_start + i).  chunk_len==0 means
 * "one buffer".  Returns number of chunks processed, or <0 on error.
 */
int64_t oracle_playback_i16(const int16_t *iq_interleaved, size_t n_samples, size_t chunk_len,
                            oracle_frame *out, size_t max_out, uint64_t *n_found);

/* ---- packet.rs / msgs.rs field decode -------------------------------------------------- */

enum { ORACLE_MSG_AIRCRAFT_ID = 0, ORACLE_MSG_AIRCRAFT_POSITION = 1, ORACLE_MSG_UNKNOWN = 2 };

typedef struct {
    uint8_t  packet[14];
    uint8_t  downlink_format;     /* packet.rs:26 */
    uint8_t  capability;          /* packet.rs:27  (mask 5, sic) */
    uint32_t icao;                /* packet.rs:28 */
    uint8_t  msg_type;            /* packet.rs:29 */
    int32_t  msg_kind;            /* which AdsbMsgType variant */
    /* AircraftID (msgs.rs:180-201) */
    char     callsign[9];
    /* AircraftPosition (msgs.rs:70-102) */
    uint8_t  surveillance_status;
    uint8_t  nic_supplement;
    int32_t  altitude;
    uint8_t  cpr_time;
    uint8_t  cpr_odd;             /* 0 = Even, 1 = Odd */
    uint32_t cpr_latitude;
    uint32_t cpr_longitude;
    /* UknownMsg (packet.rs:37): packet[4..14] */
    uint8_t  raw_msg[10];
} oracle_packet;

/* packet.rs:25-49 (time_processed excluded: wall clock). */
void oracle_packet_new(const uint8_t bytes[14], oracle_packet *p);

/* packet.rs:77-99 + msgs.rs Display impls.  `time_str` replaces the wall-clock field.
 * Returns the number of bytes written (excluding NUL), or the needed size if cap too small. */
size_t oracle_packet_display(const oracle_packet *p, const char *time_str, char *dst, size_t cap);


/* ---- cpr.rs: global CPR decode (f64; parity by tolerance: the reference's own tests use 1e-4 deg) --- */

/* cpr.rs:39-54 calc_num_zones */
uint32_t oracle_calc_num_zones(double lat);
/* cpr.rs:63-88 calculate_latitude: out[0] = selected latitude, out[1] = even, out[2] = odd.
 * first_is_odd: the OLDER message's format (CprFormat `first`): 0 = Even, 1 = Odd. */
void oracle_calculate_latitude(uint32_t even_cpr_lat, uint32_t odd_cpr_lat, int first_is_odd, double out[3]);
/* cpr.rs:90-127 calculate_longitude */
double oracle_calculate_longitude(uint32_t even_cpr_long, uint32_t odd_cpr_long, double latitude, int first_is_odd);
/* cpr.rs:135-147 calculate_geographic_position: returns 1 for Some (lat/lon written), 0 for None */
int oracle_calculate_geographic_position(uint32_t even_lat, uint32_t even_lon, uint32_t odd_lat, uint32_t odd_lon,
                                         int first_is_odd, double *latitude, double *longitude);

/* ---- aircraft.rs: per-ICAO tracker (Aircraft::handle_packet + handle_aircraft_update) ---------- */

typedef struct {
    uint32_t icao;
    char     callsign[9];     /* "" while None (aircraft.rs:119-126) */
    int32_t  altitude;
    int32_t  has_position;    /* geo_position.is_some() */
    double   latitude, longitude;
    double   last_contact;    /* seconds; the reference stores a wall-clock DateTime (set on Position messages
                                 only, aircraft.rs:56); NaN until the first one here */
} oracle_aircraft_summary;

typedef struct oracle_tracker oracle_tracker;
oracle_tracker *oracle_tracker_create(void);
void oracle_tracker_destroy(oracle_tracker *t);
/* aircraft.rs:158-165 handle_aircraft_update: `time_s` stands in for packet.time_processed (the reference
 * takes the wall clock at AdsbPacket::new; here the caller supplies seconds, e.g. sample offset / 2e6).
 * Writes the updated aircraft's summary to *out (may be NULL).  Returns 1 if this packet produced a new
 * geographic position, 0 otherwise. */
int oracle_tracker_update(oracle_tracker *t, const uint8_t bytes[14], double time_s, oracle_aircraft_summary *out);
size_t oracle_tracker_count(const oracle_tracker *t);
/* aircraft in order of first appearance */
int oracle_tracker_get(const oracle_tracker *t, size_t index, oracle_aircraft_summary *out);

#ifdef __cplusplus
}
#endif
#endif
