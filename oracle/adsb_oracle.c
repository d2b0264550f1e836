/*
 * adsb_oracle.c -- CPU ORACLE (test infrastructure, never shipped in the product path).
 * See adsb_oracle.h for the parity status.  Plain C99, libm only.
 *
 * The code below keeps the reference's structure on purpose (f64 sqrt + truncation,
 * all-pairs ordering loops with early exit, 16-bit "manchester" symbols, bit-vector
 * CRC long division, ordered single-bit brute force), so that it is a restatement a
 * reviewer can diff against the Rust by eye, and so that timing it is a fair stand-in
 * for the reference's own CPU path.
 */
#include "adsb_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PREAMBLE_LEN 16
#define PACKET_BITS 112
#define WINDOW (PREAMBLE_LEN + PACKET_BITS * 2) /* adsb.rs:98 : 16 + 112*2 = 240 */

/* utils.rs:46-52 */
void oracle_get_magnitude(const int16_t *iq, size_t n, uint32_t *mags)
{
    for (size_t k = 0; k < n; ++k) {
        double re = (double)iq[2 * k];
        double im = (double)iq[2 * k + 1];
        double m = sqrt(re * re + im * im); /* powi(2) == x*x exactly for these ranges */
        mags[k] = (uint32_t)m;              /* `as u32` truncates; value <= 46341 */
    }
}

/* demod.rs:17-57 */
int oracle_check_for_adsb_packet(const uint32_t buf[32], uint32_t *high_out)
{
    static const int lows[12] = {1, 3, 4, 5, 6, 8, 10, 11, 12, 13, 14, 15}; /* demod.rs:23 */
    static const int highs[4] = {0, 2, 7, 9};                               /* demod.rs:24 */
    uint32_t min = UINT32_MAX;

    for (int h = 0; h < 4; ++h) {
        for (int l = 0; l < 12; ++l) {
            if (buf[highs[h]] < buf[lows[l]]) return 0; /* demod.rs:29-31 */
        }
        if (buf[highs[h]] < min) min = buf[highs[h]];   /* demod.rs:33-35 */
    }

    static const int df_lows[5] = {1, 2, 4, 6, 9};  /* demod.rs:45 */
    static const int df_highs[5] = {0, 3, 5, 7, 8}; /* demod.rs:46 */
    for (int h = 0; h < 5; ++h) {
        for (int l = 0; l < 5; ++l) {
            if (buf[df_highs[h] + 16] < buf[df_lows[l] + 16]) return 0; /* demod.rs:50-52 */
        }
    }

    if (high_out) *high_out = (uint32_t)((float)min * 0.9f); /* demod.rs:56 */
    return 1;
}

/* demod.rs:92-131 */
int oracle_extract_manchester_relative(const uint32_t *buf, size_t buf_len, uint32_t high,
                                       uint16_t *symbols)
{
    (void)high; /* `_high` is unused in the reference */
    int errors = 0;
    size_t n_out = 0;
    for (size_t block_start = 0; block_start < buf_len; block_start += 16) {
        uint16_t symbol = 0;
        for (int bit = 0; bit < 8; ++bit) {
            size_t i = block_start + (size_t)bit * 2;
            int first, second;
            if (buf[i] > buf[i + 1]) { first = 1; second = 0; }
            else                     { first = 0; second = 1; }
            if (first != second) {
                symbol |= (uint16_t)(first << (14 - bit * 2));
                symbol |= (uint16_t)(second << (15 - bit * 2));
            } else {
                errors += 1;                 /* unreachable: first != second always */
                if (errors > 2) return 0;
            }
        }
        symbols[n_out++] = symbol;
        errors = 0;
    }
    return 1;
}

/* demod.rs:180-201 */
int oracle_decode_packet(const uint16_t *symbols, size_t n, uint8_t *bytes)
{
    for (size_t s = 0; s < n; ++s) {
        uint16_t encoded = symbols[s];
        uint8_t byte = 0;
        for (int i = 0; i < 8; ++i) {
            int hi = (encoded >> (15 - i * 2)) & 1;
            int lo = (encoded >> (14 - i * 2)) & 1;
            if (hi == 0 && lo == 1) byte |= (uint8_t)(1u << (7 - i));
            /* (1,0) and the invalid pairs contribute nothing */
        }
        bytes[s] = byte;
    }
    return 1;
}

/* crc.rs:10-40 */
uint32_t oracle_get_adsb_crc(const uint8_t *buf, size_t len)
{
    const uint32_t GENERATOR = 0x1FFF409u; /* 0b1_1111_1111_1111_0100_0000_1001 */
    const size_t GENERATOR_LEN = 24;
    size_t nbits = len * 8 + GENERATOR_LEN;
    unsigned char stack_bits[14 * 8 + 24];
    unsigned char *bits = nbits <= sizeof(stack_bits) ? stack_bits : (unsigned char *)malloc(nbits);
    size_t p = 0;
    for (size_t b = 0; b < len; ++b)
        for (int i = 7; i >= 0; --i) bits[p++] = (unsigned char)((buf[b] >> i) & 1);
    for (size_t i = 0; i < GENERATOR_LEN; ++i) bits[p++] = 0;

    for (size_t i = 0; i < nbits - GENERATOR_LEN; ++i) {
        if (bits[i]) {
            for (size_t j = 0; j <= GENERATOR_LEN; ++j)
                bits[i + j] ^= (unsigned char)((GENERATOR >> (GENERATOR_LEN - j)) & 1);
        }
    }

    uint32_t remainder = 0;
    for (size_t i = 0; i < GENERATOR_LEN; ++i)
        if (bits[nbits - GENERATOR_LEN + i]) remainder |= 1u << (GENERATOR_LEN - 1 - i);
    if (bits != stack_bits) free(bits);
    return remainder;
}

/* crc.rs:49-65 */
int oracle_try_crc_recovery(const uint8_t *buf, size_t len, uint32_t calc_crc,
                            uint32_t packet_crc, uint8_t *out, int *flipped_bit)
{
    (void)calc_crc; /* `_calc_crc` is unused in the reference */
    uint8_t augmented[64];
    if (len > sizeof(augmented)) return 0;
    for (size_t num = 0; num < len; ++num) {
        memcpy(augmented, buf, len);                 /* buf.clone() per byte */
        for (int i = 0; i < 8; ++i) {
            uint8_t augmented_byte = buf[num];
            augmented_byte ^= (uint8_t)(1u << (7 - i));
            augmented[num] = augmented_byte;
            uint32_t crc = oracle_get_adsb_crc(augmented, len - 3);
            if (crc == packet_crc) {
                memcpy(out, augmented, len);
                if (flipped_bit) *flipped_bit = (int)(num * 8 + (size_t)i);
                return 1;
            }
        }
    }
    return 0;
}

/* demod.rs:65-82 */
int oracle_extract_packet(const uint32_t *buf224, uint32_t high, uint8_t out[14],
                          uint8_t *status, uint8_t *fixed_bit)
{
    uint16_t symbols[14];
    uint8_t packet[14];
    if (!oracle_extract_manchester_relative(buf224, 224, (uint32_t)((double)high * 0.9), symbols))
        return 0;
    if (!oracle_decode_packet(symbols, 14, packet)) return 0;

    const size_t len = 14;
    uint32_t calced_crc = oracle_get_adsb_crc(packet, len - 3);
    uint32_t packet_crc = ((uint32_t)packet[len - 1] << 0) | ((uint32_t)packet[len - 2] << 8) |
                          ((uint32_t)packet[len - 3] << 16);
    if (calced_crc != packet_crc) {
        int bit = -1;
        if (!oracle_try_crc_recovery(packet, len, calced_crc, packet_crc, out, &bit)) return 0;
        if (status) *status = 1;
        if (fixed_bit) *fixed_bit = (uint8_t)bit;
        return 1;
    }
    memcpy(out, packet, len);
    if (status) *status = 0;
    if (fixed_bit) *fixed_bit = 0xFF;
    return 1;
}

/* adsb.rs:95-116, one received buffer */
static int process_mags(const uint32_t *mags, size_t n, uint64_t base_offset, oracle_frame *out,
                        size_t max_out, uint64_t *n_found)
{
    if (n < WINDOW) return ADSB_ORACLE_E_SHORT; /* `mags.len() - 240` underflows: panic */
    for (size_t i = 0; i < n - WINDOW; ++i) {    /* adsb.rs:98 */
        uint32_t check_mags[32];
        memcpy(check_mags, mags + i, sizeof(check_mags)); /* adsb.rs:99-101 */
        uint32_t high;
        if (oracle_check_for_adsb_packet(check_mags, &high)) {
            uint8_t bytes[14], status, fixed;
            if (oracle_extract_packet(mags + i + 16, high, bytes, &status, &fixed)) { /* :106 */
                if (*n_found < max_out) {
                    oracle_frame *f = &out[*n_found];
                    f->offset = base_offset + i;
                    memcpy(f->bytes, bytes, 14);
                    f->status = status;
                    f->fixed_bit = fixed;
                }
                *n_found += 1;
                /* adsb.rs:113 `_i += 240` rebinds the loop variable only: no skip. */
            }
        }
    }
    return 0;
}

int oracle_process_buffer_i16(const int16_t *iq, size_t n, oracle_frame *out, size_t max_out,
                              uint64_t *n_found)
{
    *n_found = 0;
    if (n < WINDOW) return ADSB_ORACLE_E_SHORT;
    uint32_t *mags = (uint32_t *)malloc(n * sizeof(uint32_t));
    if (!mags) return -2;
    oracle_get_magnitude(iq, n, mags); /* adsb.rs:96 */
    int rc = process_mags(mags, n, 0, out, max_out, n_found);
    free(mags);
    return rc;
}

int oracle_process_buffer_i8(const int8_t *iq, size_t n, oracle_frame *out, size_t max_out,
                             uint64_t *n_found)
{
    *n_found = 0;
    if (n < WINDOW) return ADSB_ORACLE_E_SHORT;
    /* Stream through in slabs so a 1 GiB buffer does not need 2+4 GiB of scratch:
     * the loop has no cross-offset state, so slabs with a 239-sample
     * read halo are the same computation as one pass. */
    const size_t SLAB = 1u << 20;
    int16_t *wide = (int16_t *)malloc((SLAB + WINDOW) * 2 * sizeof(int16_t));
    uint32_t *mags = (uint32_t *)malloc((SLAB + WINDOW) * sizeof(uint32_t));
    if (!wide || !mags) { free(wide); free(mags); return -2; }
    size_t n_offsets = n - WINDOW;
    int rc = 0;
    for (size_t start = 0; start < n_offsets; start += SLAB) {
        size_t offs = n_offsets - start < SLAB ? n_offsets - start : SLAB;
        size_t need = offs + WINDOW; /* the slab's own loop then runs exactly `offs` times */
        for (size_t k = 0; k < need * 2; ++k) wide[k] = (int16_t)iq[start * 2 + k];
        oracle_get_magnitude(wide, need, mags);
        rc = process_mags(mags, need, start, out, max_out, n_found);
        if (rc != 0) break;
    }
    free(wide);
    free(mags);
    return rc;
}

/* adsb.rs:75-89 feeding adsb.rs:92-122 */
int64_t oracle_playback_i16(const int16_t *iq, size_t n, size_t chunk_len, oracle_frame *out,
                            size_t max_out, uint64_t *n_found)
{
    *n_found = 0;
    if (chunk_len == 0) {
        int rc = oracle_process_buffer_i16(iq, n, out, max_out, n_found);
        return rc < 0 ? rc : 1;
    }
    if (n < chunk_len) return ADSB_ORACLE_E_SHORT; /* `data.len()-20000` underflows */
    int64_t chunks = 0;
    size_t i = 0;
    uint32_t *mags = (uint32_t *)malloc(chunk_len * sizeof(uint32_t));
    if (!mags) return -2;
    while (i < n - chunk_len) { /* strict `<`: the final chunk is never sent */
        oracle_get_magnitude(iq + 2 * i, chunk_len, mags);
        int rc = process_mags(mags, chunk_len, i, out, max_out, n_found);
        if (rc < 0) { free(mags); return rc; }
        i += chunk_len;
        chunks += 1;
    }
    free(mags);
    return chunks;
}

/* ---- packet.rs / msgs.rs --------------------------------------------------------------- */

/* msgs.rs:150-170 */
static size_t to_6bit_chunks(const uint8_t *input, size_t n, uint8_t *out)
{
    size_t n_out = 0;
    uint32_t acc = 0;
    int bits = 0;
    for (size_t k = 0; k < n; ++k) {
        acc = (acc << 8) | input[k];
        bits += 8;
        while (bits >= 6) {
            bits -= 6;
            out[n_out++] = (uint8_t)((acc >> bits) & 0x3F);
        }
    }
    if (bits > 0) out[n_out++] = (uint8_t)((acc << (6 - bits)) & 0x3F);
    return n_out;
}

/* msgs.rs:172-177 */
static const char CHAR_CONVERT[64] = {
    '#', 'A', 'B', 'C', 'D', 'E', 'F', 'G', 'H', 'I', 'J', 'K', 'L', 'M', 'N', 'O',
    'P', 'Q', 'R', 'S', 'T', 'U', 'V', 'W', 'X', 'Y', 'Z', '#', '#', '#', '#', '#',
    '_', '#', '#', '#', '#', '#', '#', '#', '#', '#', '#', '#', '#', '#', '#', '#',
    '0', '1', '2', '3', '4', '5', '6', '7', '8', '9', '#', '#', '#', '#', '#', '#'};

void oracle_packet_new(const uint8_t packet[14], oracle_packet *p)
{
    memset(p, 0, sizeof(*p));
    memcpy(p->packet, packet, 14);
    p->downlink_format = packet[0] >> 3;                               /* packet.rs:26 */
    p->capability = packet[0] & 5;                                     /* packet.rs:27 */
    p->icao = ((uint32_t)packet[1] << 16) | ((uint32_t)packet[2] << 8) | packet[3]; /* :28 */
    p->msg_type = packet[4] >> 3;                                      /* packet.rs:29 */
    const uint8_t *msg = packet + 4;                                   /* packet[4..11] */

    if (1 <= p->msg_type && p->msg_type <= 4) {                        /* msgs.rs:210-212 */
        p->msg_kind = ORACLE_MSG_AIRCRAFT_ID;
        uint8_t six[16];
        size_t n6 = to_6bit_chunks(msg + 1, 6, six);                   /* msgs.rs:181 */
        size_t c = 0;
        for (size_t k = 0; k < n6 && c < 8; ++k) p->callsign[c++] = CHAR_CONVERT[six[k] & 63];
        p->callsign[c] = 0;
    } else if (9 <= p->msg_type && p->msg_type <= 18) {                /* msgs.rs:122-124 */
        p->msg_kind = ORACLE_MSG_AIRCRAFT_POSITION;
        int alt_mode_25 = (msg[1] & 1) == 1;                           /* msgs.rs:71 */
        int32_t altitude = ((int32_t)((msg[1] & 0xFE) >> 1) << 4) | ((int32_t)(msg[2] & 0xF0) >> 4);
        altitude *= alt_mode_25 ? 25 : 100;
        altitude -= 1000;
        p->altitude = altitude;
        p->surveillance_status = (msg[0] & 0x06) >> 1;
        p->nic_supplement = msg[0] & 0x01;
        p->cpr_time = (msg[2] & 0x08) >> 3;
        p->cpr_odd = (msg[2] & 0x04) >> 2;
        p->cpr_latitude = ((uint32_t)(msg[2] & 0x3) << 15) | ((uint32_t)msg[3] << 7) |
                          (((uint32_t)msg[4] & 0xFE) >> 1);            /* msgs.rs:84-86 */
        p->cpr_longitude = ((uint32_t)(msg[4] & 0x1) << 16) | ((uint32_t)msg[5] << 8) |
                           (uint32_t)msg[6];                           /* msgs.rs:87-89 */
    } else {
        p->msg_kind = ORACLE_MSG_UNKNOWN;
        memcpy(p->raw_msg, packet + 4, 10);                            /* packet.rs:37 */
    }
}

size_t oracle_packet_display(const oracle_packet *p, const char *time_str, char *dst, size_t cap)
{
    char tmp[1024];
    size_t n = 0;
#define EMIT(...) n += (size_t)snprintf(tmp + n, sizeof(tmp) - n, __VA_ARGS__)
    EMIT("== ");
    for (int k = 0; k < 14; ++k) EMIT("%02x", p->packet[k]);
    EMIT(" ==\n");
    EMIT("Decoded Information:\n");
    EMIT("Downlink Format : %u\n", p->downlink_format);
    EMIT("Capability      : %u\n", p->capability);
    EMIT("ICAO            : %06X\n", p->icao);
    EMIT("Processed Time  : %s\n", time_str ? time_str : "");
    EMIT("Message Type    : %u\n", p->msg_type);
    if (p->msg_kind == ORACLE_MSG_AIRCRAFT_ID) {                       /* msgs.rs:215-223 */
        EMIT("Message:\n");
        EMIT("Type                : %u (ID)\n", p->msg_type);
        EMIT("Callsign            : %s\n", p->callsign);
    } else if (p->msg_kind == ORACLE_MSG_AIRCRAFT_POSITION) {          /* msgs.rs:127-140 */
        EMIT("Message:\n");
        EMIT("Type                : %u (Position)\n", p->msg_type);
        EMIT("Surveillance Status : %u\n", p->surveillance_status);
        EMIT("NIC Supplement      : %u\n", p->nic_supplement);
        EMIT("Altitude (ft)       : %d\n", p->altitude);
        EMIT("CPR Time            : %u\n", p->cpr_time);
        EMIT("CPR Format          : %s\n", p->cpr_odd ? "Odd" : "Even");
        EMIT("Raw Latitude        : %u\n", p->cpr_latitude);
        EMIT("Raw Longitude       : %u\n", p->cpr_longitude);
    } else {                                                           /* msgs.rs:36-44 */
        EMIT("Message:\n");
        EMIT("Type    : Unknown\n");
        EMIT("Raw Msg :  [");
        for (int k = 0; k < 10; ++k) EMIT(k ? ", %u" : "%u", p->raw_msg[k]);
        EMIT("]\n");
    }
#undef EMIT
    if (dst && cap > n) memcpy(dst, tmp, n + 1);
    return n;
}

/* ======================================================================================== */
/* src/adsb/cpr.rs                                                                          */
/* ======================================================================================== */

#define ORACLE_PI 3.14159265358979323846264338327950288 /* std::f64::consts::PI */
static const double NUM_ZONES = 15.0;                   /* cpr.rs:19 */

static double convert_cpr_to_float(uint32_t cpr)        /* cpr.rs:22-25 */
{
    return (double)cpr / 131072.0;
}

static double normalize_longitude(double lon)           /* cpr.rs:27-31 */
{
    while (lon < -180.0) lon += 360.0;
    while (lon > 180.0) lon -= 360.0;
    return lon;
}

/* `x.floor() as u32`: Rust float->int casts saturate and map NaN to 0 */
static uint32_t floor_as_u32(double x)
{
    double f = floor(x);
    if (!(f >= 0.0)) return 0u;
    if (f >= 4294967295.0) return 4294967295u;
    return (uint32_t)f;
}

uint32_t oracle_calc_num_zones(double lat)               /* cpr.rs:39-54 */
{
    if (lat == 0.0) return 59;
    else if (lat == 87.0 || lat == -87.0) return 2;
    else if (lat < -87.0 || lat > 87.0) return 1;
    const double pi = ORACLE_PI;
    double int1 = 1.0 - cos(pi / (2.0 * NUM_ZONES));
    double int2 = cos(pi / 180.0 * lat);
    double int3 = (2.0 * pi) / acos(1.0 - (int1 / (int2 * int2)));
    return floor_as_u32(int3);
}

void oracle_calculate_latitude(uint32_t even_cpr_lat_u, uint32_t odd_cpr_lat_u, int first_is_odd, double out[3])
{                                                        /* cpr.rs:63-88 */
    const double EVEN_LAT_DIVISIONS = 360.0 / (4.0 * NUM_ZONES);
    const double ODD_LAT_DIVISIONS = 360.0 / (4.0 * NUM_ZONES - 1.0);
    double even_cpr_lat = convert_cpr_to_float(even_cpr_lat_u);
    double odd_cpr_lat = convert_cpr_to_float(odd_cpr_lat_u);
    double latitude_index = floor(59.0 * even_cpr_lat - 60.0 * odd_cpr_lat + 0.5);
    /* Rust's % on f64 is fmod (result takes the sign of the dividend) */
    double even_latitude = EVEN_LAT_DIVISIONS * (fmod(latitude_index, 60.0) + even_cpr_lat);
    double odd_latitude = ODD_LAT_DIVISIONS * (fmod(latitude_index, 59.0) + odd_cpr_lat);
    /* "Use the newest format to determine the latitude": first Even -> odd, first Odd -> even */
    double latitude = first_is_odd ? even_latitude : odd_latitude;
    if (latitude > 270.0) latitude -= 360.0;
    out[0] = latitude;
    out[1] = even_latitude;
    out[2] = odd_latitude;
}

double oracle_calculate_longitude(uint32_t even_cpr_long, uint32_t odd_cpr_long, double latitude, int first_is_odd)
{                                                        /* cpr.rs:90-127 */
    double lon_cpr_e = convert_cpr_to_float(even_cpr_long);
    double lon_cpr_o = convert_cpr_to_float(odd_cpr_long);
    uint32_t nl = oracle_calc_num_zones(latitude);
    uint32_t nz = first_is_odd ? oracle_calc_num_zones(latitude)          /* later is even */
                               : oracle_calc_num_zones(latitude - 1.0);   /* later is odd (sic: latitude - 1.0) */
    if (nz < 1) nz = 1;                                                   /* .max(1) */
    double num_zones = (double)nz;
    double divisions = 360.0 / num_zones;
    /* (nl - 1) as f64: u32 arithmetic; nl >= 1 on every path of calc_num_zones except NaN -> 0, where the
     * reference would overflow (panic in debug, wrap in release); wrap like release */
    double m = floor(lon_cpr_e * (double)(uint32_t)(nl - 1u) - lon_cpr_o * (double)nl + 0.5);
    double longitude = first_is_odd ? divisions * (fmod(m, num_zones) + lon_cpr_e)
                                    : divisions * (fmod(m, num_zones) + lon_cpr_o);
    return normalize_longitude(longitude);
}

int oracle_calculate_geographic_position(uint32_t even_lat, uint32_t even_lon, uint32_t odd_lat, uint32_t odd_lon,
                                         int first_is_odd, double *latitude, double *longitude)
{                                                        /* cpr.rs:135-147 */
    double l[3];
    oracle_calculate_latitude(even_lat, odd_lat, first_is_odd, l);
    if (oracle_calc_num_zones(l[1]) != oracle_calc_num_zones(l[2])) return 0; /* (the reference also prints) */
    *longitude = oracle_calculate_longitude(even_lon, odd_lon, l[0], first_is_odd);
    *latitude = l[0];
    return 1;
}

/* ======================================================================================== */
/* src/adsb/aircraft.rs                                                                     */
/* ======================================================================================== */

typedef struct {
    uint32_t icao;
    int has_callsign;
    char callsign[9];
    int32_t altitude;
    int has_geo;
    double lat, lon;
    double last_contact;
    int has_odd, has_even;               /* last_odd_packet / last_even_packet */
    uint32_t odd_lat, odd_lon, even_lat, even_lon;
    double odd_processed, even_processed;
} oracle_aircraft;

struct oracle_tracker {
    oracle_aircraft *a;
    size_t n, cap;
};

oracle_tracker *oracle_tracker_create(void)
{
    return (oracle_tracker *)calloc(1, sizeof(oracle_tracker));
}

void oracle_tracker_destroy(oracle_tracker *t)
{
    if (!t) return;
    free(t->a);
    free(t);
}

static void aircraft_summary(const oracle_aircraft *a, oracle_aircraft_summary *s)   /* aircraft.rs:142-152 */
{
    memset(s, 0, sizeof(*s));
    s->icao = a->icao;
    if (a->has_callsign) memcpy(s->callsign, a->callsign, 9);
    s->altitude = a->altitude;
    s->has_position = a->has_geo;
    s->latitude = a->lat;
    s->longitude = a->lon;
    s->last_contact = a->last_contact;
}

/* Aircraft::handle_packet (aircraft.rs:48-111); returns 1 when geo_position was (re)computed */
static int aircraft_handle_packet(oracle_aircraft *self, const oracle_packet *msg, double time_processed)
{
    if (msg->icao != self->icao) return 0;                                   /* :49-51 */
    if (msg->msg_kind == ORACLE_MSG_AIRCRAFT_POSITION) {
        self->altitude = msg->altitude;                                      /* :55 */
        self->last_contact = time_processed;                                 /* :56 */
        uint32_t cpr_odd_lat, cpr_odd_lon, cpr_even_lat, cpr_even_lon;
        int first_is_odd;
        if (!msg->cpr_odd) {                                                 /* CprFormat::Even, :63-77 */
            self->has_even = 1;
            self->even_lat = msg->cpr_latitude;
            self->even_lon = msg->cpr_longitude;
            self->even_processed = time_processed;
            if (!self->has_odd) return 0;
            if (fabs(time_processed - self->odd_processed) > 10.0) return 0; /* :68-70 */
            cpr_odd_lat = self->odd_lat; cpr_odd_lon = self->odd_lon;
            cpr_even_lat = msg->cpr_latitude; cpr_even_lon = msg->cpr_longitude;
            first_is_odd = 1;
        } else {                                                             /* CprFormat::Odd, :79-94 */
            self->has_odd = 1;
            self->odd_lat = msg->cpr_latitude;
            self->odd_lon = msg->cpr_longitude;
            self->odd_processed = time_processed;
            if (!self->has_even) return 0;
            if (fabs(time_processed - self->even_processed) > 10.0) return 0;
            cpr_odd_lat = msg->cpr_latitude; cpr_odd_lon = msg->cpr_longitude;
            cpr_even_lat = self->even_lat; cpr_even_lon = self->even_lon;
            first_is_odd = 0;
        }
        double lat, lon;
        if (oracle_calculate_geographic_position(cpr_even_lat, cpr_even_lon, cpr_odd_lat, cpr_odd_lon,
                                                 first_is_odd, &lat, &lon)) { /* :97-102 */
            self->has_geo = 1;
            self->lat = lat;
            self->lon = lon;
            return 1;
        }
        return 0;
    } else if (msg->msg_kind == ORACLE_MSG_AIRCRAFT_ID) {                    /* :105-107 */
        self->has_callsign = 1;
        memcpy(self->callsign, msg->callsign, 9);
    }
    return 0;                                                                /* Uknown: :108-110 */
}

int oracle_tracker_update(oracle_tracker *t, const uint8_t bytes[14], double time_s, oracle_aircraft_summary *out)
{                                                                            /* aircraft.rs:158-165 */
    oracle_packet pk;
    oracle_packet_new(bytes, &pk);
    size_t k = 0;
    while (k < t->n && t->a[k].icao != pk.icao) ++k;                         /* HashMap entry(icao) */
    if (k == t->n) {                                                         /* or_insert(Aircraft::new(icao)) */
        if (t->n == t->cap) {
            size_t nc = t->cap ? 2 * t->cap : 64;
            oracle_aircraft *na = (oracle_aircraft *)realloc(t->a, nc * sizeof(*na));
            if (!na) return -1;
            t->a = na;
            t->cap = nc;
        }
        memset(&t->a[k], 0, sizeof(t->a[k]));
        t->a[k].icao = pk.icao;
        t->a[k].last_contact = NAN;
        t->n++;
    }
    int r = aircraft_handle_packet(&t->a[k], &pk, time_s);
    if (out) aircraft_summary(&t->a[k], out);
    return r;
}

size_t oracle_tracker_count(const oracle_tracker *t) { return t->n; }

int oracle_tracker_get(const oracle_tracker *t, size_t index, oracle_aircraft_summary *out)
{
    if (index >= t->n) return -1;
    aircraft_summary(&t->a[index], out);
    return 0;
}
