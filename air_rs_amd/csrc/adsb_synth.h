// adsb_synth.h -- deterministic, integer-only synthetic 2 MSPS IQ source.
//
// One definition shared by the host generator (adsb_synth_fill_host) and the HIP generator
// kernel (adsb_synth_fill_device): sample k of channel c is a pure function of (cfg, c, k), so
// any slice of the stream can be produced on any rank without exchanging input (time-sharding,
// SURVEY §8e) and the host copy used by the parity tests is identical to what the GPU generated.
//
// The reference ships no IQ capture (its author's file is git-ignored), so this is
// what the tests and bench.py feed both the HIP path and the CPU oracle.  Modulation follows the
// pulse positions documented in the reference's gate (src/adsb/demod.rs:20-22, 41-44):
// preamble pulses at half-microsecond slots 0,2,7,9; a 1 bit is a pulse in the first half of its
// microsecond, a 0 bit a pulse in the second half (demod.rs:180-201 decodes exactly that).
#pragma once
#include <stdint.h>

#include "../../include/adsb_hip.h"

#if defined(__HIPCC__)
#define ADSB_HD __host__ __device__ inline
#else
#define ADSB_HD static inline
#endif

namespace adsb_synth {

ADSB_HD uint64_t mix64(uint64_t x)
{ // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Mode-S CRC-24 (generator 0x1FFF409, reference src/adsb/crc.rs:10-40) of 11 bytes, shift form.
ADSB_HD uint32_t crc24_11(const uint8_t *d)
{
    uint32_t r = 0;
    for (int b = 0; b < 11; ++b) {
        r ^= (uint32_t)d[b] << 16;
        for (int i = 0; i < 8; ++i) {
            r <<= 1;
            if (r & 0x1000000u) r ^= 0x1FFF409u;
        }
    }
    return r & 0xFFFFFFu;
}

struct Slot {
    int      present;
    uint32_t jitter;    // frame start inside the slot
    int      kind;      // 0 clean, 1 one data bit flipped, 2 one crc bit flipped, 3 two data bits
    int      amp_i, amp_q;
    uint8_t  clean[14];
    uint8_t  sent[14];
};

ADSB_HD void slot_params(const adsb_synth_cfg &c, uint32_t channel, uint64_t slot, Slot &s)
{
    const uint64_t chmix = (uint64_t)channel * 0xD1B54A32D192ED03ull;
    uint64_t h = mix64(c.seed ^ chmix ^ (slot * 0x9FB21C651E98DF25ull) ^ 0x5157415453594E54ull);
    uint64_t h2 = mix64(h);
    uint64_t h3 = mix64(h2);

    s.present = (uint32_t)(h % 100u) < c.frame_pct;
    s.jitter = (uint32_t)((h >> 8) % (uint64_t)(c.slot_len - 240u));
    const int amp_tab[8][2] = {{40, 0}, {0, 50}, {45, 45}, {60, -30},
                               {-70, 20}, {80, 40}, {-64, -64}, {100, 45}};
    int ai = (int)((h >> 40) & 7);
    s.amp_i = amp_tab[ai][0];
    s.amp_q = amp_tab[ai][1];

    uint32_t kroll = (uint32_t)((h >> 44) % 100u);
    if (kroll < c.pct_flip_data) s.kind = 1;
    else if (kroll < c.pct_flip_data + c.pct_flip_crc) s.kind = 2;
    else if (kroll < c.pct_flip_data + c.pct_flip_crc + c.pct_flip_two) s.kind = 3;
    else s.kind = 0;

    // DF17, CA5 (0x8D), 24-bit ICAO, 56-bit ME with a type code that walks through the three
    // AdsbMsgType variants (ID: TC 1-4, position: TC 9-18, other).
    uint8_t *p = s.clean;
    p[0] = 0x8D;
    p[1] = (uint8_t)(h2 >> 0);
    p[2] = (uint8_t)(h2 >> 8);
    p[3] = (uint8_t)(h2 >> 16);
    uint32_t tcsel = (uint32_t)((h2 >> 24) % 3u);
    uint32_t tc = tcsel == 0 ? 1u + (uint32_t)((h2 >> 28) & 3u)
                : tcsel == 1 ? 9u + (uint32_t)((h2 >> 28) % 10u)
                             : 19u + (uint32_t)((h2 >> 28) % 10u);
    p[4] = (uint8_t)((tc << 3) | ((h2 >> 36) & 7u));
    for (int b = 0; b < 6; ++b) p[5 + b] = (uint8_t)(h3 >> (8 * b));
    uint32_t crc = crc24_11(p);
    p[11] = (uint8_t)(crc >> 16);
    p[12] = (uint8_t)(crc >> 8);
    p[13] = (uint8_t)crc;

    for (int b = 0; b < 14; ++b) s.sent[b] = s.clean[b];
    uint32_t e0 = (uint32_t)((h3 >> 48) % 88u);
    uint32_t e1 = (uint32_t)((h3 >> 56) % 87u);
    if (e1 >= e0) e1 += 1; // distinct second position
    if (s.kind == 1) {
        s.sent[e0 >> 3] ^= (uint8_t)(0x80u >> (e0 & 7));
    } else if (s.kind == 2) {
        uint32_t e = 88u + (uint32_t)((h3 >> 48) % 24u);
        s.sent[e >> 3] ^= (uint8_t)(0x80u >> (e & 7));
    } else if (s.kind == 3) {
        s.sent[e0 >> 3] ^= (uint8_t)(0x80u >> (e0 & 7));
        s.sent[e1 >> 3] ^= (uint8_t)(0x80u >> (e1 & 7));
    }
}

// Is there a pulse at position p (0..239) of a frame with these bytes?
ADSB_HD int pulse_at(const uint8_t *sent, uint32_t p)
{
    if (p < 16) return p == 0 || p == 2 || p == 7 || p == 9;
    uint32_t q = p - 16;
    uint32_t bit = q >> 1;
    int one = (sent[bit >> 3] >> (7 - (bit & 7))) & 1;
    return one ? ((q & 1) == 0) : ((q & 1) == 1);
}

ADSB_HD void noise_iq(const adsb_synth_cfg &c, uint32_t channel, uint64_t k, int &ni, int &nq)
{
    const uint64_t chmix = (uint64_t)channel * 0xD1B54A32D192ED03ull;
    uint64_t h = mix64(c.seed ^ chmix ^ (k * 0xC2B2AE3D27D4EB4Full));
    int si = (int)(h & 255) + (int)((h >> 8) & 255) + (int)((h >> 16) & 255) + (int)((h >> 24) & 255);
    int sq = (int)((h >> 32) & 255) + (int)((h >> 40) & 255) + (int)((h >> 48) & 255) +
             (int)((h >> 56) & 255);
    ni = (si - 510) / (int)c.noise_div; // C division: truncates toward zero on both sides
    nq = (sq - 510) / (int)c.noise_div;
}

// i8-scale value of sample k (before clipping).
ADSB_HD void sample_iq(const adsb_synth_cfg &c, uint32_t channel, uint64_t k, const Slot &s,
                       uint64_t slot, int &vi, int &vq)
{
    noise_iq(c, channel, k, vi, vq);
    if (s.present) {
        uint64_t start = slot * (uint64_t)c.slot_len + s.jitter;
        if (k >= start && k < start + 240 && pulse_at(s.sent, (uint32_t)(k - start))) {
            vi += s.amp_i;
            vq += s.amp_q;
        }
    }
}

ADSB_HD int clip8(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }

} // namespace adsb_synth
