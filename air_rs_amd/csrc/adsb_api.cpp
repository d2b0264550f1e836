// adsb_api.cpp -- the extern "C" boundary (include/adsb_hip.h) over the gfx950 kernels.
//
// Replaces, per received buffer, the body of the reference's thread 2 loop
// (src/adsb.rs:95-116).  There is NO CPU fallback: without a HIP device adsb_create() fails with
// ADSB_E_NODEVICE and every other entry point needs a context.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "adsb_kernels.h"
#include "adsb_synth.h"

using adsbk::kTile;
using adsbk::kWindow;

namespace {
constexpr int kTimingRing = 512;
}

struct adsb_ctx {
    adsb_cfg cfg{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int mag_mode = 0;
    uint32_t bps = 2; // bytes per IQ sample

    // device buffers
    void *staging = nullptr;        // host-fed input (cfg.host_staging)
    // Two result sets, used alternately: the ordering pass of launch i runs on `aux` while the
    // demod kernel of launch i+1 already runs on `stream` (they touch different sets).
    struct ResultSet {
        adsbk::Seg *seg = nullptr;       // [n_tiles_max]
        adsb_frame *slots = nullptr;     // [n_tiles_max * kQuota] fixed region, then the pool [cap_slots]
        adsb_frame *out = nullptr;       // [max_out]
        adsbk::Header *hdr = nullptr;
        uint64_t *chan_prefix = nullptr; // [max_channels + 1]: frames before each channel's first tile; last = total
        hipEvent_t k_done = nullptr, g_done = nullptr;
        bool g_pending = false;
        // the launch whose results this set holds (the streaming front end fetches the older of two launches
        // in flight: view_launch() makes it the one the fetch / re-plan code below works on)
        struct Launch {
            const void *iq = nullptr;
            uint32_t channels = 0, tpc = 0, tiles = 0, cap = 0, idx = 0;
            uint64_t samples = 0, stride = 0, base = 0;
            adsb_frame *out = nullptr;
            bool valid = false;
        } li;
    } rs[2];
    hipStream_t aux = nullptr;      // ordering pass + result copies (== stream unless ADSB_OVERLAP_ORDERING=1)
    bool own_aux = false;
    adsb_packet_fields *fields = nullptr; // [max_out], allocated on first adsb_decode_fields_device_async
    bool fields_current = false;    // fields[] belongs to the last launch
    // tracker (allocated on first adsb_track_device)
    uint32_t *trk_u32 = nullptr;    // 4 x [max_out]: keys, vals, sorted keys, sorted vals
    void *trk_temp = nullptr;
    size_t trk_temp_bytes = 0;
    adsb_track_point *trk_points = nullptr;      // [max_out]
    adsb_aircraft_record *trk_aircraft = nullptr; // [max_out]
    uint64_t *trk_n_aircraft = nullptr;
    uint32_t trk_n = 0;             // frames the last tracker run covered
    bool trk_done = false;
    void *ext_blob = nullptr;       // caller-owned [32-byte header | frames] target for the next launches
    size_t ext_frames = 0;          // frame capacity of ext_blob
    adsb_frame *last_out = nullptr; // where the last launch's ordered list went
    uint32_t last_cap = 0;
    bool fused_pass_only = false;   // adsb_debug_fused_pass_only (measurement)
    uint64_t stream_base = 0;       // adsb_set_stream_base: added to the offsets of the following launches
    uint64_t last_base = 0;         // ... of the last launch (re-runs of its tiles use the same)
    uint32_t launch_idx = 0;        // launches so far
    uint32_t last = 0;              // result set of the last launch
    uint32_t *out_start = nullptr;  // [n_tiles_max + 1]  (slot-overflow re-run path only)
    uint64_t *lb = nullptr;         // finish_order's exchange words: one per workgroup, then one per 64 workgroups
    uint32_t lb_groups_at = 0;
    uint32_t *scratch = nullptr;    // 16 dwords: probe result, read-kernel sink
    unsigned long long *stamps = nullptr; // cycle counters of diagnostic builds (64 bytes per tile with -DADSB_TILE_STAMPS=1)
    size_t stamps_bytes = 0;
    int scan = adsbk::kScanRoot;    // which i8 scan kernel (ADSB_SCAN=nsq selects the A/B kernel at adsb_create)
    // The one-dispatch path for small buffers (adsbk::launch_small): per result set a pinned, device-writable blob
    // [32-byte header | frames | u64 sequence number] and a device counter; a pinned input buffer for adsb_demod().
    struct Small {
        bool enabled = true, ready = false;
        char *blob[2] = {nullptr, nullptr};
        uint32_t cap = 0;            // frames per blob
        uint32_t *done = nullptr;    // device: 2 words
        char *in_host = nullptr;     // pinned copy of adsb_demod()'s buffer (allocated on first use)
        uint64_t seq = 0;
        uint64_t max_samples = 0;    // longest buffer the path takes
    } sm;
    bool pool_off = false;          // adsb_debug_pool_limit: the shared slot pool hands out nothing (test knob)
    uint32_t stall_blk = 0xFFFFFFFFu; // adsb_debug_finish_stall: this workgroup of finish_order withholds its exchange word (test knob)
    uint32_t cap_slots = 0;
    uint32_t n_tiles_max = 0;

    // pinned host mirrors
    adsbk::Header *hdr_host = nullptr;

    // last launch
    bool launched = false;
    const void *last_iq = nullptr;
    uint32_t last_channels = 0;
    uint64_t last_samples = 0, last_stride = 0;
    uint32_t last_tpc = 0, last_tiles = 0;

    // timing
    int timing = 0;                 // 0 off; N: events on every N-th launch
    hipEvent_t ev[kTimingRing][4] = {}; // scan kernel, finishing kernel: start/end each
    bool ev_made = false;
    uint32_t ev_count = 0;
};

#define HIPCHK(x)                                  \
    do {                                           \
        hipError_t e_ = (x);                       \
        if (e_ != hipSuccess) return (int)e_;      \
    } while (0)

static uint32_t tiles_for(uint64_t n_samples, int sample_type, int scan)
{
    if (n_samples <= (uint64_t)kWindow) return 0;
    const uint64_t n_off = n_samples - kWindow, tile = (uint64_t)adsbk::tile_offsets_of(sample_type, scan);
    return (uint32_t)((n_off + tile - 1) / tile);
}

extern "C" const char *adsb_strerror(int code)
{
    switch (code) {
    case ADSB_OK: return "ok";
    case ADSB_E_SHORT: return "buffer shorter than 240 samples (reference panics, adsb.rs:98)";
    case ADSB_E_ARG: return "bad argument";
    case ADSB_E_CAPACITY: return "exceeds the capacity the context was created with";
    case ADSB_E_NOMEM: return "out of memory";
    case ADSB_E_NODEVICE: return "no usable HIP device (there is no CPU fallback)";
    case ADSB_E_STATE: return "call sequence error";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

extern "C" void adsb_destroy(adsb_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->ev_made)
        for (auto &e : c->ev)
            for (auto &x : e)
                if (x) (void)hipEventDestroy(x);
    if (c->own_aux && c->aux) (void)hipStreamSynchronize(c->aux);
    (void)hipFree(c->staging);
    for (auto &r : c->rs) {
        (void)hipFree(r.seg);
        (void)hipFree(r.slots);
        (void)hipFree(r.out);
        (void)hipFree(r.hdr);
        (void)hipFree(r.chan_prefix);
        if (r.k_done) (void)hipEventDestroy(r.k_done);
        if (r.g_done) (void)hipEventDestroy(r.g_done);
    }
    (void)hipFree(c->out_start);
    (void)hipFree(c->fields);
    (void)hipFree(c->trk_u32);
    (void)hipFree(c->trk_temp);
    (void)hipFree(c->trk_points);
    (void)hipFree(c->trk_aircraft);
    (void)hipFree(c->trk_n_aircraft);
    (void)hipFree(c->scratch);
    (void)hipFree(c->stamps);
    (void)hipFree(c->lb);
    (void)hipFree(c->sm.done);
    for (char *b : c->sm.blob) if (b) (void)hipHostFree(b);
    if (c->sm.in_host) (void)hipHostFree(c->sm.in_host);
    if (c->own_aux && c->aux) (void)hipStreamDestroy(c->aux);
    if (c->hdr_host) (void)hipHostFree(c->hdr_host);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// The code scan's table through the device (adsbk::launch_code_probe): tab[n] = c(n) | th(n) << 8 for n = I^2+Q^2 in
// 0 .. 32768, c = the 8-bit code the gate slides over, th = the code a "low" may reach while floor(sqrt) of a "high" n can still
// be >= its own.  What the kernel relies on (adsb_kernels.hip, "the code scan"): c is monotone, stays below 0x7C (an
// ordered f16 pattern in the high byte), and th(n) >= c(top(n)), top(n) = the largest n' with floor(sqrt(n')) = floor(sqrt(n)).
// ADSB_E_STATE if the device disagrees (no fallback: adsb_create fails).
static int code_table_check(adsb_ctx *c, uint16_t *tab)
{
    uint16_t *dev = nullptr;
    if (hipMalloc((void **)&dev, 32769 * sizeof(uint16_t)) != hipSuccess) return ADSB_E_NOMEM;
    hipError_t e = adsbk::launch_code_probe(c->stream, dev);
    if (e == hipSuccess) e = hipMemcpyAsync(tab, dev, 32769 * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dev);
    if (e != hipSuccess) return (int)e;
    uint32_t root = 0;
    for (uint32_t n = 0; n <= 32768; ++n) {
        while ((root + 1) * (root + 1) <= n) ++root;
        const uint32_t top = std::min<uint32_t>((root + 1) * (root + 1) - 1, 32768);
        const uint32_t code = tab[n] & 0xFFu, th = tab[n] >> 8, code_top = tab[top] & 0xFFu;
        if (code >= 0x7Cu || (n && code < (tab[n - 1] & 0xFFu)) || th < code_top) return ADSB_E_STATE;
    }
    return ADSB_OK;
}

extern "C" int adsb_debug_code_table(adsb_ctx *c, uint16_t *out32769)
{
    if (!c || !out32769) return ADSB_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    return code_table_check(c, out32769);
}

extern "C" int adsb_create(const adsb_cfg *cfg, adsb_ctx **out_ctx)
{
    if (!cfg || !out_ctx) return ADSB_E_ARG;
    *out_ctx = nullptr;
    if (cfg->abi_version != ADSB_ABI_VERSION) return ADSB_E_ARG;
    if (cfg->sample_type != ADSB_SAMPLE_I8 && cfg->sample_type != ADSB_SAMPLE_I16) return ADSB_E_ARG;
    if (cfg->max_channels == 0 || cfg->max_samples == 0 || cfg->max_out == 0) return ADSB_E_ARG;
    if (cfg->max_out > 0x7FFFFFFFull - kTile) return ADSB_E_ARG;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
        return ADSB_E_NODEVICE;
    if (hipSetDevice(cfg->device) != hipSuccess) return ADSB_E_NODEVICE;

    adsb_ctx *c = new (std::nothrow) adsb_ctx();
    if (!c) return ADSB_E_NOMEM;
    c->cfg = *cfg;
    c->bps = cfg->sample_type == ADSB_SAMPLE_I8 ? 2 : 4;
    // i8 has two scan kernels: the product's (floor(sqrt) per sample, u8 magnitudes in LDS: eight workgroups per
    // CU) and the round-3 A/B kernel whose gate works on n = I^2+Q^2 (no root per sample, but 2 bytes of LDS per
    // sample: four workgroups per CU; 10 % fewer VALU slots, 12 % slower -- DESIGN.md section 5.3): ADSB_SCAN=nsq in
    // the environment at adsb_create selects it.
    if (const char *sp = getenv("ADSB_SMALL_PATH")) c->sm.enabled = !(sp[0] == '0');
    c->scan = adsbk::kScanRoot;
    if (const char *sc = getenv("ADSB_SCAN")) {
        if (strcmp(sc, "nsq") == 0) c->scan = adsbk::kScanNsq;
        else if (strcmp(sc, "reg") == 0) c->scan = adsbk::kScanReg;
        else if (strcmp(sc, "root") == 0 || sc[0] == 0) c->scan = adsbk::kScanRoot;
        else if (strcmp(sc, "code") == 0) c->scan = adsbk::kScanCode;
        else if (strcmp(sc, "sieve") == 0) c->scan = adsbk::kScanSieve;
        else { delete c; return ADSB_E_ARG; }
        // (the A/B kernels exist only in -DADSB_AB_KERNELS=1 builds: asking this library for one it does not have is an error)
        if (c->scan != adsbk::kScanRoot && !adsbk::ab_kernels_built()) { delete c; return ADSB_E_ARG; }
    }
    if (cfg->sample_type != ADSB_SAMPLE_I8) c->scan = adsbk::kScanRoot; // (CS16 has one scan kernel)
    uint64_t tiles = (uint64_t)tiles_for(cfg->max_samples, cfg->sample_type, c->scan) * cfg->max_channels;
    if (tiles == 0) tiles = 1;
    if (tiles * adsbk::kQuota + cfg->max_out + kTile > 0xFFFFFFF0ull) { delete c; return ADSB_E_CAPACITY; }
    c->n_tiles_max = (uint32_t)tiles;
    c->cap_slots = (uint32_t)(cfg->max_out + kTile);

    int rc = ADSB_OK;
    auto fail = [&](int code) { rc = code; };
    do {
        if (cfg->stream) {
            c->stream = (hipStream_t)cfg->stream;
        } else {
            if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { fail(ADSB_E_NODEVICE); break; }
            c->own_stream = true;
        }
        hipError_t e = hipSuccess;
        if (cfg->host_staging) {
            // +64 bytes so the staging base keeps 16-byte alignment for any channel stride rounding
            uint64_t stride = (cfg->max_samples + 7) & ~7ull;
            e = hipMalloc(&c->staging, stride * cfg->max_channels * c->bps + 64);
            if (e != hipSuccess) { fail(ADSB_E_NOMEM); break; }
        }
        // The ordering pass normally follows the demod kernel on the same stream.  Running it on its
        // own stream (ADSB_OVERLAP_ORDERING=1) lets it overlap the next launch's kernel, but measured
        // slower on MI355X: the tiny gather then competes with 16 k workgroups for CUs (26 us
        // instead of 9) and the step does not get shorter.
        const char *ov = getenv("ADSB_OVERLAP_ORDERING");
        if (ov && ov[0] == '1') {
            // (highest priority: its two small kernels should take the CU slots the scan's workgroups free, not wait
            // for the next launch's whole grid)
            int pr_lo = 0, pr_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi);
            if (hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, pr_hi) != hipSuccess) { fail(ADSB_E_NODEVICE); break; }
            c->own_aux = true;
        } else {
            c->aux = c->stream;
        }
        const size_t n_slots = (size_t)c->n_tiles_max * adsbk::kQuota + c->cap_slots;
        bool ok = true;
        for (auto &r : c->rs) {
            ok = ok && hipMalloc((void **)&r.seg, sizeof(adsbk::Seg) * (size_t)c->n_tiles_max) == hipSuccess &&
                 hipMalloc((void **)&r.slots, sizeof(adsb_frame) * n_slots) == hipSuccess &&
                 hipMalloc((void **)&r.out, sizeof(adsb_frame) * (size_t)cfg->max_out) == hipSuccess &&
                 hipMalloc((void **)&r.hdr, sizeof(adsbk::Header)) == hipSuccess &&
                 hipMalloc((void **)&r.chan_prefix, sizeof(uint64_t) * ((size_t)cfg->max_channels + 1)) == hipSuccess &&
                 hipMemsetAsync(r.chan_prefix, 0, sizeof(uint64_t) * ((size_t)cfg->max_channels + 1), c->stream) == hipSuccess &&
                 hipEventCreateWithFlags(&r.k_done, hipEventDisableTiming | hipEventReleaseToDevice) == hipSuccess &&
                 hipEventCreateWithFlags(&r.g_done, hipEventDisableTiming | hipEventReleaseToDevice) == hipSuccess &&
                 hipMemsetAsync(r.hdr, 0, sizeof(adsbk::Header), c->stream) == hipSuccess;
        }
        c->lb_groups_at = (uint32_t)(((size_t)c->n_tiles_max / adsbk::kFinishTilesPerWg + 2 + 63) / 64 * 64);
        const size_t lb_words = (size_t)c->lb_groups_at + c->lb_groups_at / 64 + 64;
        ok = ok && hipMalloc((void **)&c->out_start, sizeof(uint32_t) * ((size_t)c->n_tiles_max + 1)) == hipSuccess &&
             hipMalloc((void **)&c->scratch, 64) == hipSuccess &&
             hipMalloc((void **)&c->lb, sizeof(uint64_t) * lb_words) == hipSuccess &&
             hipMemsetAsync(c->lb, 0, sizeof(uint64_t) * lb_words, c->stream) == hipSuccess &&
             hipMemsetAsync(c->scratch, 0, 64, c->stream) == hipSuccess;
        if (!ok) { fail(ADSB_E_NOMEM); break; }
        if (hipHostMalloc((void **)&c->hdr_host, sizeof(adsbk::Header), hipHostMallocDefault) != hipSuccess) { fail(ADSB_E_NOMEM); break; }
        uint32_t probe[4] = {0, 0, 0, 0};
        e = adsbk::probe_cvt(c->stream, c->scratch, probe);
        if (e != hipSuccess) { fail((int)e); break; }
        // expected truncation of {0.75, 2.5, 180.9986} = {0, 2, 180}
        const uint32_t want = 0u | (2u << 8) | (180u << 16);
        if (probe[0] == want) c->mag_mode = 0;
        else if (probe[1] == want) c->mag_mode = 1;
        else c->mag_mode = 2;
        if (const char *force = getenv("ADSB_FORCE_MAG_MODE")) c->mag_mode = atoi(force) % 3;
        if (c->scan == adsbk::kScanCode) {
            // The code scan's gate is a SUPERSET test only if this device's conversion and f16 multiply-add behave as the
            // kernel assumes: check it through the kernel's own instructions, for every n an i8 sample can give.
            std::vector<uint16_t> tab(32769);
            const int prc = code_table_check(c, tab.data());
            if (prc != ADSB_OK) { fail(prc); break; }
        }
        // cycle counters of diagnostic builds (-DADSB_TILE_STAMPS=1); zeros otherwise
        c->stamps_bytes = adsbk::tile_stamps_built() ? (size_t)c->n_tiles_max * 64 + 512 : 512;
        if (hipMalloc((void **)&c->stamps, c->stamps_bytes) != hipSuccess || hipMemsetAsync(c->stamps, 0, c->stamps_bytes, c->stream) != hipSuccess) { fail(ADSB_E_NOMEM); break; }
    } while (0);
    if (rc != ADSB_OK) { adsb_destroy(c); return rc; }
    *out_ctx = c;
    return ADSB_OK;
}

extern "C" void *adsb_stream(adsb_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int adsb_sample_type(const adsb_ctx *c) { return c ? c->cfg.sample_type : ADSB_E_ARG; }
extern "C" int adsb_debug_fused_pass_only(adsb_ctx *c, int on)
{
    if (!c) return ADSB_E_ARG;
    c->fused_pass_only = on != 0;
    return ADSB_OK;
}
extern "C" int adsb_debug_mag_mode(adsb_ctx *c) { return c ? c->mag_mode : ADSB_E_ARG; }
extern "C" int adsb_debug_scan(adsb_ctx *c) { return c ? (c->cfg.sample_type == ADSB_SAMPLE_I8 ? c->scan : adsbk::kScanRoot) : ADSB_E_ARG; }
// Test knobs.  adsb_debug_set_launch_index: the next launch counts as launch number `idx` (the exchange words' epoch is
// derived from it: lets a test cross the 2^30 wrap).  adsb_debug_finish_stall: finish_order's workgroup `blk` of the
// following launches withholds its exchange word (0xFFFFFFFF: none) -- the workgroups behind it give up after ~0.1 s.
extern "C" int adsb_debug_set_launch_index(adsb_ctx *c, uint32_t idx)
{
    if (!c) return ADSB_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->own_aux) HIPCHK(hipStreamSynchronize(c->aux));
    c->launch_idx = idx;
    return ADSB_OK;
}
extern "C" int adsb_debug_finish_stall(adsb_ctx *c, uint32_t blk)
{
    if (!c) return ADSB_E_ARG;
    c->stall_blk = blk;
    return ADSB_OK;
}

extern "C" int adsb_debug_pool_limit(adsb_ctx *c, int on)
{
    if (!c) return ADSB_E_ARG;
    c->pool_off = on != 0;
    return ADSB_OK;
}

extern "C" int adsb_debug_tile_stamps(adsb_ctx *c, uint32_t *out, size_t max_tiles, size_t *n_tiles)
{
    if (!c || !n_tiles || (!out && max_tiles)) return ADSB_E_ARG;
    if (!adsbk::tile_stamps_built() || !c->stamps) return ADSB_E_STATE;
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    const size_t n = std::min<size_t>(max_tiles, c->last_tiles);
    if (n) HIPCHK(hipMemcpy(out, c->stamps, n * 64, hipMemcpyDeviceToHost));
    *n_tiles = n;
    return ADSB_OK;
}

static adsbk::DemodArgs demod_args(adsb_ctx *c, adsb_ctx::ResultSet &r, uint32_t tile_first, uint32_t tile_count,
                                   bool first_pass)
{
    adsbk::DemodArgs a{};
    a.iq = c->last_iq;
    a.n_samples = c->last_samples;
    a.channel_stride = c->last_stride;
    a.tiles_per_channel = c->last_tpc;
    a.tile_first = tile_first;
    a.tile_count = tile_count;
    a.count_groups = first_pass ? 1u : 0u;
    a.offset_base = c->last_base;
    a.fused_pass_only = c->fused_pass_only ? 1u : 0u;
    a.seg = r.seg;
    a.slots = r.slots;
    a.pool_first = c->n_tiles_max * adsbk::kQuota;
    a.cap_slots = c->cap_slots;
    a.hdr = r.hdr;
    a.hdr_pub = (first_pass && c->ext_blob) ? static_cast<uint64_t *>(c->ext_blob) : nullptr;
    a.pool_off = (c->pool_off && first_pass) ? 1u : 0u; // (first passes only: the re-run of lost tiles needs the pool)
    a.stamps = c->stamps;
    return a;
}

// launch_epoch: the index of the launch the pass belongs to (tags the look-back words)
static adsbk::FinishArgs finish_args(adsb_ctx *c, adsb_ctx::ResultSet &r, uint32_t launch_epoch, uint32_t tile_first,
                                     uint32_t tile_count, bool rerun)
{
    adsbk::FinishArgs a{};
    a.seg = r.seg;
    a.slots = r.slots;
    a.out_start = rerun ? c->out_start : nullptr;
    a.lb = c->lb;
    a.lb_groups_at = c->lb_groups_at;
    a.chan_prefix = rerun ? nullptr : r.chan_prefix;
    a.epoch = (launch_epoch + 1u) & 0x3FFFFFFFu;
    const bool ext = c->ext_blob && !rerun;
    a.out = ext ? reinterpret_cast<adsb_frame *>(static_cast<char *>(c->ext_blob) + 32) : r.out;
    a.hdr_pub = ext ? static_cast<uint64_t *>(c->ext_blob) : nullptr;
    a.tiles_per_channel = c->last_tpc;
    a.n_channels = c->last_channels;
    a.max_out = ext ? (uint32_t)std::min<size_t>(c->ext_frames, c->cfg.max_out) : (uint32_t)c->cfg.max_out;
    if (rerun) { a.out = c->last_out; a.max_out = c->last_cap; }
    a.tile_first = tile_first;
    a.tile_count = tile_count;
    a.hdr = r.hdr;
    a.stall_blk = rerun ? 0xFFFFFFFFu : c->stall_blk;
    return a;
}

// finish_order's exchange words are never cleared: they carry the launch's 30-bit epoch ((launch index + 1) mod 2^30) and a
// word of another epoch reads as "not there yet".  A word written exactly 2^30 launches ago would read as this launch's:
// whenever the epoch wraps (every ~4 hours of back-to-back 13 us launches) the words are zeroed on the stream first.
static hipError_t clear_exchange_words_at_wrap(adsb_ctx *c, uint32_t launch_idx)
{
    if (((launch_idx + 1u) & 0x3FFFFFFFu) != 0u) return hipSuccess;
    const size_t lb_words = (size_t)c->lb_groups_at + c->lb_groups_at / 64 + 64;
    return hipMemsetAsync(c->lb, 0, sizeof(uint64_t) * lb_words, c->aux);
}

extern "C" int adsb_demod_device_async(adsb_ctx *c, const void *iq_dev, uint32_t n_channels,
                                       size_t n_samples, size_t channel_stride)
{
    if (!c || !iq_dev || n_channels == 0) return ADSB_E_ARG;
    if (n_samples < (size_t)kWindow) return ADSB_E_SHORT;
    if (n_channels > c->cfg.max_channels || n_samples > c->cfg.max_samples) return ADSB_E_CAPACITY;
    if (((uintptr_t)iq_dev & 15u) != 0) return ADSB_E_ARG;
    if (n_channels > 1 && (channel_stride < n_samples || (channel_stride & 7u) != 0)) return ADSB_E_ARG;
    if (n_channels == 1) channel_stride = n_samples;
    HIPCHK(hipSetDevice(c->cfg.device));

    c->last_iq = iq_dev;
    c->last_channels = n_channels;
    c->last_samples = n_samples;
    c->last_stride = channel_stride;
    c->last_tpc = tiles_for(n_samples, c->cfg.sample_type, c->scan);
    c->last_tiles = c->last_tpc * n_channels;
    c->last_base = c->stream_base;
    c->launched = true;
    c->fields_current = false;
    c->trk_done = false;

    // Launch i uses result set i & 1.  (ADSB_OVERLAP_ORDERING=1: the finishing kernel of launch i runs on `aux` beside
    // the scan of launch i+1, which only has to wait for the finishing kernel of launch i-2: same result set.)
    const uint32_t i = c->launch_idx;
    adsb_ctx::ResultSet &r = c->rs[i & 1u];
    if (c->own_aux && r.g_pending) HIPCHK(hipStreamWaitEvent(c->stream, r.g_done, 0));
    HIPCHK(clear_exchange_words_at_wrap(c, i));

    hipEvent_t *ev = nullptr;
    if (c->timing && (c->launch_idx % (uint32_t)c->timing) == 0) {
        if (!c->ev_made) {
            for (auto &e : c->ev)
                for (auto &x : e) HIPCHK(hipEventCreateWithFlags(&x, hipEventReleaseToDevice)); // no system-scope flush
            c->ev_made = true;
        }
        ev = c->ev[c->ev_count % kTimingRing];
    }
    const adsbk::DemodArgs da = demod_args(c, r, 0, c->last_tiles, true);
    // (ADSB_OVERLAP_ORDERING=1: the event the other stream waits for rides on the scan's own dispatch packet: no
    // barrier packet between two scans.)
    hipEvent_t scan_done = ev ? ev[1] : (c->own_aux ? r.k_done : nullptr);
    HIPCHK(adsbk::launch_demod(c->stream, c->cfg.sample_type, c->mag_mode, c->scan, da, ev ? ev[0] : nullptr, scan_done));
    if (c->own_aux && scan_done && c->last_tiles) HIPCHK(hipStreamWaitEvent(c->aux, scan_done, 0));
    // second kernel: CRC-24 / repair of the survivors the scan kernel sliced, and the ordered list.  Without tiles
    // (exactly 240 samples: adsb.rs:98 iterates 0..0) or in measurement mode (scan only) the list is empty.
    if (c->last_tiles == 0 || c->fused_pass_only) {
        adsbk::Header *hdr = r.hdr;
        HIPCHK(adsbk::launch_empty_result(c->aux, hdr, c->ext_blob ? static_cast<uint64_t *>(c->ext_blob) : nullptr,
                                          r.chan_prefix, c->last_channels, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr));
    } else {
        HIPCHK(adsbk::launch_finish(c->aux, finish_args(c, r, i, 0, c->last_tiles, false), ev ? ev[2] : nullptr, ev ? ev[3] : nullptr));
    }
    // (same stream: in-order already; adsb_stream_wait_results records the event when somebody asks for it)
    if (c->own_aux) {
        HIPCHK(hipEventRecord(r.g_done, c->aux));
        r.g_pending = true;
    }
    c->last_out = c->ext_blob ? reinterpret_cast<adsb_frame *>(static_cast<char *>(c->ext_blob) + 32) : r.out;
    c->last_cap = c->ext_blob ? (uint32_t)std::min<size_t>(c->ext_frames, c->cfg.max_out) : (uint32_t)c->cfg.max_out;
    c->last = i & 1u;
    r.li.iq = c->last_iq; r.li.channels = c->last_channels; r.li.samples = c->last_samples; r.li.stride = c->last_stride;
    r.li.base = c->last_base; r.li.tpc = c->last_tpc; r.li.tiles = c->last_tiles; r.li.out = c->last_out;
    r.li.cap = c->last_cap; r.li.idx = i; r.li.valid = true;
    c->launch_idx = i + 1u;
    if (ev && c->last_tiles) c->ev_count++;
    return ADSB_OK;
}

// Makes result set `set` (launch li.idx) the launch every fetch / re-plan function below refers to.  Only the
// streaming front end looks at anything but the newest launch; it restores the newest before enqueueing again.
static void view_launch(adsb_ctx *c, uint32_t set)
{
    const adsb_ctx::ResultSet::Launch &li = c->rs[set].li;
    c->last = set;
    c->last_iq = li.iq; c->last_channels = li.channels; c->last_samples = li.samples; c->last_stride = li.stride;
    c->last_base = li.base; c->last_tpc = li.tpc; c->last_tiles = li.tiles; c->last_out = li.out; c->last_cap = li.cap;
    c->fields_current = false;
    c->trk_done = false;
}

// ---- the one-dispatch path for small buffers --------------------------------------------------------------------------
// (adsbk::demod_small: every workgroup scans its tile, the last one to finish checks and orders all survivors and writes
// header + frames straight into pinned host memory, then a sequence number the host polls.)  One HIP call per buffer
// instead of a copy, three launches, two result copies and their synchronisations: what the reference's own buffer sizes
// (20 000 samples, adsb.rs:77-79; MTU-sized reads, adsb.rs:59-64) need.
static size_t small_blob_bytes(uint32_t cap) { return 32 + sizeof(adsb_frame) * (size_t)cap + 16; }
static uint64_t *small_seq_word(adsb_ctx *c, uint32_t set)
{
    return reinterpret_cast<uint64_t *>(c->sm.blob[set] + 32 + sizeof(adsb_frame) * (size_t)c->sm.cap);
}

static int small_init(adsb_ctx *c)
{
    if (c->sm.ready) return ADSB_OK;
    if (!c->sm.enabled) return ADSB_E_STATE;
    const uint64_t tile = (uint64_t)adsbk::tile_offsets_of(c->cfg.sample_type, c->scan);
    c->sm.max_samples = std::min<uint64_t>(c->cfg.max_samples, tile * adsbk::kFinishTilesPerWg + kWindow);
    // a buffer of n samples has at most n - 240 frames (one per offset: SURVEY F8)
    c->sm.cap = (uint32_t)std::min<uint64_t>(c->cfg.max_out, c->sm.max_samples);
    if (c->sm.cap == 0) return ADSB_E_STATE;
    HIPCHK(hipSetDevice(c->cfg.device));
    for (int k = 0; k < 2; ++k) {
        // coherent (fine-grained) host memory: the device's writes are visible to the polling host while the stream runs
        if (hipHostMalloc((void **)&c->sm.blob[k], small_blob_bytes(c->sm.cap), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            (void)hipGetLastError();
            c->sm.enabled = false;
            return ADSB_E_NOMEM;
        }
        std::memset(c->sm.blob[k], 0, 32);
        *small_seq_word(c, (uint32_t)k) = 0;
    }
    if (hipMalloc((void **)&c->sm.done, 2 * sizeof(uint32_t)) != hipSuccess ||
        hipMemsetAsync(c->sm.done, 0, 2 * sizeof(uint32_t), c->stream) != hipSuccess) {
        c->sm.enabled = false;
        return ADSB_E_NOMEM;
    }
    c->sm.ready = true;
    return ADSB_OK;
}

// Can this buffer take the one-dispatch path?  (Not while the caller redirects results into a blob of its own.)
static bool small_fits(adsb_ctx *c, size_t n_samples)
{
    return c->sm.enabled && !c->ext_blob && !c->fused_pass_only && !c->own_aux && n_samples > (size_t)kWindow &&
           small_init(c) == ADSB_OK && n_samples <= c->sm.max_samples;
}

// Enqueues one buffer (device-visible memory: pinned host or device, 16-byte aligned) as launch `launch_idx`; like
// adsb_demod_device_async for one channel, in one dispatch.  *seq_out = the value the blob's sequence word will carry.
static int small_launch(adsb_ctx *c, const void *iq, size_t n_samples, uint64_t *seq_out)
{
    HIPCHK(hipSetDevice(c->cfg.device));
    c->last_iq = iq;
    c->last_channels = 1;
    c->last_samples = n_samples;
    c->last_stride = n_samples;
    c->last_tpc = tiles_for(n_samples, c->cfg.sample_type, c->scan);
    c->last_tiles = c->last_tpc;
    c->last_base = c->stream_base;
    c->launched = true;
    c->fields_current = false;
    c->trk_done = false;
    const uint32_t i = c->launch_idx, set = i & 1u;
    adsb_ctx::ResultSet &r = c->rs[set];
    HIPCHK(clear_exchange_words_at_wrap(c, i));
    adsbk::DemodArgs da = demod_args(c, r, 0, c->last_tiles, true);
    da.hdr_pub = reinterpret_cast<uint64_t *>(c->sm.blob[set]);
    adsbk::FinishArgs fa = finish_args(c, r, i, 0, c->last_tiles, false);
    fa.out = reinterpret_cast<adsb_frame *>(c->sm.blob[set] + 32);
    fa.hdr_pub = reinterpret_cast<uint64_t *>(c->sm.blob[set]);
    fa.max_out = c->sm.cap;
    adsbk::SmallArgs sa{};
    sa.done = c->sm.done + set;
    sa.seq_host = small_seq_word(c, set);
    sa.seq = ++c->sm.seq;
    HIPCHK(adsbk::launch_small(c->stream, c->cfg.sample_type, c->mag_mode, c->scan, da, fa, sa));
    c->last_out = fa.out;
    c->last_cap = fa.max_out;
    c->last = set;
    r.li.iq = iq; r.li.channels = 1; r.li.samples = n_samples; r.li.stride = n_samples;
    r.li.base = c->last_base; r.li.tpc = c->last_tpc; r.li.tiles = c->last_tiles; r.li.out = c->last_out;
    r.li.cap = c->last_cap; r.li.idx = i; r.li.valid = true;
    c->launch_idx = i + 1u;
    *seq_out = sa.seq;
    return ADSB_OK;
}

// Waits (polling pinned memory, no HIP call on the usual path) until result set `set` carries sequence number `seq`.
// (at_least: a LATER launch on the same result set has finished -- the stream is in order -- will do as well: used where
// only "the device no longer reads that launch's input" matters)
static int small_wait(adsb_ctx *c, uint32_t set, uint64_t seq, bool at_least = false)
{
    volatile uint64_t *w = small_seq_word(c, set);
    auto done = [&]() { const uint64_t v = __atomic_load_n(w, __ATOMIC_ACQUIRE); return at_least ? v >= seq : v == seq; };
    for (uint32_t spins = 0; !done(); ++spins) {
        if (spins > (1u << 22)) { // ~ tens of milliseconds of polling: let the runtime tell what happened
            HIPCHK(hipSetDevice(c->cfg.device));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (!done()) return ADSB_E_STATE;
            break;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    return ADSB_OK;
}

static int sync_header(adsb_ctx *c);
// Hands the finished list of result set `set` (a small launch that small_wait has seen complete) to the caller.
static int small_collect(adsb_ctx *c, uint32_t set, adsb_frame *out, size_t max_out, size_t *n_out, uint64_t *total_found,
                         uint32_t *flags)
{
    const uint64_t *hdr = reinterpret_cast<const uint64_t *>(c->sm.blob[set]);
    if (hdr[2] & ADSB_FLAG_INCOMPLETE) {
        // slot-pool overflow (pathological input): the standard re-run path completes the blob in place
        const uint32_t newest = c->last;
        view_launch(c, set);
        int rc = sync_header(c);
        view_launch(c, newest);
        if (rc != ADSB_OK) return rc;
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    uint64_t n = hdr[0];
    uint32_t fl = (uint32_t)hdr[2] & ~ADSB_FLAG_INCOMPLETE;
    if (n > max_out) { n = max_out; fl |= ADSB_FLAG_TRUNCATED; }
    if (n) std::memcpy(out, c->sm.blob[set] + 32, sizeof(adsb_frame) * (size_t)n);
    *n_out = (size_t)n;
    if (total_found) *total_found = hdr[1];
    if (flags) *flags = fl;
    return ADSB_OK;
}

// Slot-pool overflow (far more gate survivors than max_out + one tile): redo the tiles that feed
// the first max_out frames in batches whose survivors fit.  Counts from the first pass are exact,
// so the plan is made on the host.  Only pathological inputs get here.
static int rerun_in_batches(adsb_ctx *c, adsb_ctx::ResultSet &r)
{
    HIPCHK(hipStreamSynchronize(c->aux));
    HIPCHK(hipStreamSynchronize(c->stream));
    const uint32_t n = c->last_tiles;
    std::vector<adsbk::Seg> seg(n);
    HIPCHK(hipMemcpyAsync(seg.data(), r.seg, sizeof(adsbk::Seg) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<uint32_t> start((size_t)n + 1);
    uint64_t run = 0;
    for (uint32_t t = 0; t <= n; ++t) {
        start[t] = run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)run;
        if (t < n) run += seg[t].valid;
    }
    HIPCHK(hipMemcpyAsync(c->out_start, start.data(), sizeof(uint32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream));
    uint32_t limit = 0;
    while (limit < n && start[limit] < c->last_cap) ++limit;
    uint32_t t0 = 0;
    int rc = ADSB_OK;
    while (t0 < limit && rc == ADSB_OK) {
        uint64_t used = 0;
        uint32_t t1 = t0;
        auto pool_need = [&](uint32_t t) { return seg[t].cand > adsbk::kQuota ? (uint64_t)seg[t].cand : 0ull; };
        while (t1 < limit && used + pool_need(t1) <= c->cap_slots) used += pool_need(t1++);
        if (t1 == t0) { rc = ADSB_E_STATE; break; } // a single tile never exceeds cap_slots (>= kTile)
        hipError_t e;
        if ((e = hipMemsetAsync(&r.hdr->alloc, 0, sizeof(unsigned long long), c->stream)) != hipSuccess ||
            (e = adsbk::launch_demod(c->stream, c->cfg.sample_type, c->mag_mode, c->scan,
                                     demod_args(c, r, t0, t1 - t0, false))) != hipSuccess ||
            (e = adsbk::launch_finish(c->stream, finish_args(c, r, r.li.idx, t0, t1 - t0, true))) != hipSuccess)
            rc = (int)e;
        t0 = t1;
    }
    HIPCHK(hipMemsetAsync(&r.hdr->alloc, 0, sizeof(unsigned long long), c->stream));
    HIPCHK(hipMemsetAsync(&r.hdr->retry, 0, sizeof(uint32_t), c->stream));
    if (rc == ADSB_OK) { // the list is whole now: drop ADSB_FLAG_INCOMPLETE where device-side consumers read it
        c->hdr_host->flags &= ~ADSB_FLAG_INCOMPLETE;
        const uint64_t pub_flags = c->hdr_host->flags;
        HIPCHK(hipMemcpyAsync(&r.hdr->flags, &c->hdr_host->flags, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        if (c->last_out != r.out) // the launch wrote into a caller-owned blob: [n_out | total | flags | 0 | frames]
            HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(c->last_out) - 16, &pub_flags, sizeof(uint64_t), hipMemcpyDefault, c->stream)); // (the blob may be pinned host memory: the small-buffer path)
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return rc;
}

static int sync_header(adsb_ctx *c)
{
    if (!c->launched) return ADSB_E_STATE;
    HIPCHK(hipSetDevice(c->cfg.device));
    adsb_ctx::ResultSet &r = c->rs[c->last];
    HIPCHK(hipMemcpyAsync(c->hdr_host, r.hdr, sizeof(adsbk::Header), hipMemcpyDeviceToHost, c->aux));
    HIPCHK(hipStreamSynchronize(c->aux)); // the ordering pass of the last launch ran on aux before this copy
    if (c->hdr_host->retry & 4u) return ADSB_E_STATE; // finish_order gave up waiting for a workgroup (never seen; see its comment)
    if (c->hdr_host->retry) {
        int rc = rerun_in_batches(c, r);
        if (rc != ADSB_OK) return rc;
        c->hdr_host->retry = 0;
        c->fields_current = false; // the list was rebuilt: decoded fields / tracker output are stale
        c->trk_done = false;
    }
    return ADSB_OK;
}

extern "C" int adsb_fetch_counts(adsb_ctx *c, uint64_t *n_out, uint64_t *total_found, uint32_t *flags)
{
    if (!c) return ADSB_E_ARG;
    int rc = sync_header(c);
    if (rc != ADSB_OK) return rc;
    if (n_out) *n_out = c->hdr_host->n_out;
    if (total_found) *total_found = c->hdr_host->total_found;
    if (flags) *flags = c->hdr_host->flags;
    return ADSB_OK;
}

extern "C" int adsb_fetch(adsb_ctx *c, adsb_frame *out, size_t max_out, size_t *n_out,
                          uint64_t *per_channel_counts, uint64_t *total_found, uint32_t *flags)
{
    if (!c || !n_out || (!out && max_out)) return ADSB_E_ARG;
    int rc = sync_header(c);
    if (rc != ADSB_OK) return rc;
    uint64_t n = c->hdr_host->n_out;
    uint32_t fl = c->hdr_host->flags;
    if (n > max_out) { n = max_out; fl |= ADSB_FLAG_TRUNCATED; }
    adsb_ctx::ResultSet &r = c->rs[c->last];
    if (n) HIPCHK(hipMemcpyAsync(out, c->last_out, sizeof(adsb_frame) * n, hipMemcpyDefault, c->aux)); // (device memory, or the small-buffer path's pinned blob)
    std::vector<uint64_t> pre;
    if (per_channel_counts) {
        pre.resize((size_t)c->last_channels + 1);
        HIPCHK(hipMemcpyAsync(pre.data(), r.chan_prefix, sizeof(uint64_t) * pre.size(), hipMemcpyDeviceToHost, c->aux));
    }
    HIPCHK(hipStreamSynchronize(c->aux));
    if (per_channel_counts) { // frames of channel k in `out` = its share of the first n frames of the list
        for (uint32_t k = 0; k < c->last_channels; ++k)
            per_channel_counts[k] = std::min<uint64_t>(pre[k + 1], n) - std::min<uint64_t>(pre[k], n);
    }
    *n_out = (size_t)n;
    if (total_found) *total_found = c->hdr_host->total_found;
    if (flags) *flags = fl;
    return ADSB_OK;
}

extern "C" int adsb_result_device(adsb_ctx *c, const adsb_frame **frames_dev, const void **header_dev)
{
    if (!c) return ADSB_E_ARG;
    if (frames_dev) *frames_dev = c->last_out ? c->last_out : c->rs[c->last].out;
    if (header_dev) *header_dev = c->rs[c->last].hdr;
    return ADSB_OK;
}

extern "C" int adsb_decode_fields_device_async(adsb_ctx *c)
{
    if (!c) return ADSB_E_ARG;
    if (!c->launched) return ADSB_E_STATE;
    HIPCHK(hipSetDevice(c->cfg.device));
    if (!c->fields && hipMalloc((void **)&c->fields, sizeof(adsb_packet_fields) * (size_t)c->cfg.max_out) != hipSuccess)
        return ADSB_E_NOMEM;
    // same stream as the ordering pass, so it sees the finished list and header
    HIPCHK(adsbk::launch_decode_fields(c->aux, c->last_out, c->rs[c->last].hdr, c->last_cap, c->fields));
    c->fields_current = true;
    return ADSB_OK;
}

extern "C" int adsb_fields_device(adsb_ctx *c, const adsb_packet_fields **fields_dev)
{
    if (!c || !fields_dev) return ADSB_E_ARG;
    *fields_dev = c->fields;
    return c->fields ? ADSB_OK : ADSB_E_STATE;
}

extern "C" int adsb_fetch_fields(adsb_ctx *c, adsb_packet_fields *out, size_t max_out, size_t *n_out)
{
    if (!c || !n_out || (!out && max_out)) return ADSB_E_ARG;
    if (!c->fields) return ADSB_E_STATE;
    int rc = sync_header(c);
    if (rc != ADSB_OK) return rc;
    uint64_t n = std::min<uint64_t>(c->hdr_host->n_out, c->last_cap);
    if (n > max_out) n = max_out;
    if (n) HIPCHK(hipMemcpyAsync(out, c->fields, sizeof(adsb_packet_fields) * n, hipMemcpyDeviceToHost, c->aux));
    HIPCHK(hipStreamSynchronize(c->aux));
    *n_out = (size_t)n;
    return ADSB_OK;
}


extern "C" int adsb_track_device(adsb_ctx *c, double seconds_per_sample)
{
    if (!c || !(seconds_per_sample > 0.0)) return ADSB_E_ARG;
    if (!c->launched) return ADSB_E_STATE;
    if (c->last_channels != 1) return ADSB_E_ARG;
    int rc = sync_header(c); // the list's length (and the rebuild after a slot-pool overflow)
    if (rc != ADSB_OK) return rc;
    if (!c->fields_current && (rc = adsb_decode_fields_device_async(c)) != ADSB_OK) return rc;
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t cap = (size_t)c->cfg.max_out;
    if (!c->trk_u32) {
        c->trk_temp_bytes = adsbk::track_sort_temp_bytes(cap);
        if (hipMalloc((void **)&c->trk_u32, sizeof(uint32_t) * 4 * cap) != hipSuccess ||
            hipMalloc(&c->trk_temp, c->trk_temp_bytes) != hipSuccess ||
            hipMalloc((void **)&c->trk_points, sizeof(adsb_track_point) * cap) != hipSuccess ||
            hipMalloc((void **)&c->trk_aircraft, sizeof(adsb_aircraft_record) * cap) != hipSuccess ||
            hipMalloc((void **)&c->trk_n_aircraft, sizeof(uint64_t)) != hipSuccess)
            return ADSB_E_NOMEM;
    }
    const uint64_t n = std::min<uint64_t>(c->hdr_host->n_out, c->last_cap);
    adsbk::TrackArgs a{};
    a.frames = c->last_out;
    a.fields = c->fields;
    a.n = (uint32_t)n;
    a.seconds_per_sample = seconds_per_sample;
    a.keys = c->trk_u32;
    a.vals = c->trk_u32 + cap;
    a.skeys = c->trk_u32 + 2 * cap;
    a.svals = c->trk_u32 + 3 * cap;
    a.temp = c->trk_temp;
    a.temp_bytes = c->trk_temp_bytes;
    a.points = c->trk_points;
    a.aircraft = c->trk_aircraft;
    a.max_aircraft = (uint32_t)cap;
    a.n_aircraft = c->trk_n_aircraft;
    HIPCHK(adsbk::launch_track(c->aux, a)); // same stream as the ordering pass and the field decode
    c->trk_n = (uint32_t)n;
    c->trk_done = true;
    return ADSB_OK;
}

extern "C" int adsb_fetch_track(adsb_ctx *c, adsb_track_point *points, size_t max_points, size_t *n_points,
                                adsb_aircraft_record *aircraft, size_t max_aircraft, size_t *n_aircraft)
{
    if (!c || (!points && max_points) || (!aircraft && max_aircraft)) return ADSB_E_ARG;
    if (!c->trk_done) return ADSB_E_STATE;
    HIPCHK(hipSetDevice(c->cfg.device));
    uint64_t na = 0;
    HIPCHK(hipMemcpyAsync(&na, c->trk_n_aircraft, sizeof(uint64_t), hipMemcpyDeviceToHost, c->aux));
    HIPCHK(hipStreamSynchronize(c->aux));
    const size_t np = std::min<size_t>(c->trk_n, max_points);
    const size_t nac = std::min<size_t>((size_t)na, max_aircraft);
    if (np) HIPCHK(hipMemcpyAsync(points, c->trk_points, sizeof(adsb_track_point) * np, hipMemcpyDeviceToHost, c->aux));
    if (nac) HIPCHK(hipMemcpyAsync(aircraft, c->trk_aircraft, sizeof(adsb_aircraft_record) * nac, hipMemcpyDeviceToHost, c->aux));
    HIPCHK(hipStreamSynchronize(c->aux));
    if (n_points) *n_points = np;
    if (n_aircraft) *n_aircraft = (size_t)na;
    return ADSB_OK;
}

extern "C" int adsb_set_result_target(adsb_ctx *c, void *blob_dev, size_t blob_bytes)
{
    if (!c) return ADSB_E_ARG;
    if (!blob_dev) { c->ext_blob = nullptr; c->ext_frames = 0; return ADSB_OK; }
    if (((uintptr_t)blob_dev & 15u) || blob_bytes < 32 + sizeof(adsb_frame)) return ADSB_E_ARG;
    c->ext_blob = blob_dev;
    c->ext_frames = (blob_bytes - 32) / sizeof(adsb_frame);
    return ADSB_OK;
}

extern "C" int adsb_set_stream_base(adsb_ctx *c, uint64_t first_sample_index)
{
    if (!c) return ADSB_E_ARG;
    c->stream_base = first_sample_index;
    return ADSB_OK;
}

extern "C" int adsb_stream_wait_results(adsb_ctx *c, void *stream)
{
    if (!c) return ADSB_E_ARG;
    if (!c->launched) return ADSB_E_STATE;
    HIPCHK(hipSetDevice(c->cfg.device));
    if (!c->own_aux) HIPCHK(hipEventRecord(c->rs[c->last].g_done, c->aux)); // tail of the in-order stream
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->rs[c->last].g_done, 0));
    return ADSB_OK;
}

extern "C" int adsb_demod(adsb_ctx *c, const void *iq, size_t n_samples, adsb_frame *out,
                          size_t max_out, size_t *n_out, uint32_t *flags)
{
    if (!c || !iq || !n_out) return ADSB_E_ARG;
    *n_out = 0;
    if (flags) *flags = 0;
    if (!c->staging) return ADSB_E_STATE;
    if (n_samples < (size_t)kWindow) return ADSB_E_SHORT;
    if (n_samples > c->cfg.max_samples) return ADSB_E_CAPACITY;
    HIPCHK(hipSetDevice(c->cfg.device));
    if (small_fits(c, n_samples)) {
        // a small buffer (the reference's own sizes): copy it into pinned memory the device reads directly, ONE dispatch,
        // the frames come back through pinned memory too -- no copy commands, no stream synchronisation
        if (!c->sm.in_host &&
            hipHostMalloc((void **)&c->sm.in_host, (size_t)c->sm.max_samples * c->bps + 64, hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            c->sm.enabled = false;
        } else {
            std::memcpy(c->sm.in_host, iq, n_samples * c->bps);
            uint64_t seq = 0;
            int rc = small_launch(c, c->sm.in_host, n_samples, &seq);
            if (rc != ADSB_OK) return rc;
            const uint32_t set = c->last;
            rc = small_wait(c, set, seq);
            if (rc != ADSB_OK) return rc;
            return small_collect(c, set, out, max_out, n_out, nullptr, flags);
        }
    }
    HIPCHK(hipMemcpyAsync(c->staging, iq, n_samples * c->bps, hipMemcpyHostToDevice, c->stream));
    int rc = adsb_demod_device_async(c, c->staging, 1, n_samples, n_samples);
    if (rc != ADSB_OK) return rc;
    return adsb_fetch(c, out, max_out, n_out, nullptr, nullptr, flags);
}

// ---- measurement / test helpers -----------------------------------------------------------------
extern "C" int adsb_timing_enable(adsb_ctx *c, int on)
{
    if (!c) return ADSB_E_ARG;
    c->timing = on > 0 ? on : 0;
    c->ev_count = 0;
    return ADSB_OK;
}

static int timing_read(adsb_ctx *c, double *demod_ms, double *order_ms, double *decode_ms, uint32_t *n_launches);
extern "C" int adsb_timing_read(adsb_ctx *c, double *demod_ms, double *order_ms, uint32_t *n_launches)
{
    return timing_read(c, demod_ms, order_ms, nullptr, n_launches);
}
extern "C" int adsb_timing_read3(adsb_ctx *c, double *scan_ms, double *decode_ms, double *order_ms, uint32_t *n_launches)
{
    return timing_read(c, scan_ms, order_ms, decode_ms, n_launches);
}
static int timing_read(adsb_ctx *c, double *demod_ms, double *order_ms, double *decode_ms, uint32_t *n_launches)
{
    if (!c) return ADSB_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->aux));
    uint32_t n = std::min<uint32_t>(c->ev_count, kTimingRing);
    double a = 0, b = 0;
    for (uint32_t k = 0; k < n; ++k) {
        float x = 0, y = 0;
        HIPCHK(hipEventElapsedTime(&x, c->ev[k][0], c->ev[k][1]));
        HIPCHK(hipEventElapsedTime(&y, c->ev[k][2], c->ev[k][3]));
        a += x;
        b += y;
    }
    // a launch is two kernels since round 3: the scan and finish_order (CRC / repair + the ordered list in one); the
    // latter is reported as the "decode" figure, the separate ordering pass no longer exists (0)
    if (demod_ms) *demod_ms = n ? a / n : 0.0;
    if (decode_ms) *decode_ms = n ? b / n : 0.0;
    if (order_ms) *order_ms = decode_ms ? 0.0 : (n ? b / n : 0.0);
    if (n_launches) *n_launches = n;
    c->ev_count = 0;
    return ADSB_OK;
}

extern "C" int adsb_time_read_ceiling(adsb_ctx *c, const void *buf_dev, size_t bytes, int iters,
                                      double *ms_per_pass)
{
    if (!c || !buf_dev || iters <= 0 || !ms_per_pass || ((uintptr_t)buf_dev & 15u)) return ADSB_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    // the fastest of the read shapes (adsb_kernels.hip, read_only_kernel): none of them wins on every box and size
    double best = 0;
    for (int shape = 0; shape < adsbk::kReadShapes; ++shape) {
        HIPCHK(adsbk::launch_read_only(c->stream, buf_dev, bytes, c->scratch + 8, shape)); // warm-up
        HIPCHK(hipEventRecord(e0, c->stream));
        for (int k = 0; k < iters; ++k) HIPCHK(adsbk::launch_read_only(c->stream, buf_dev, bytes, c->scratch + 8, shape));
        HIPCHK(hipEventRecord(e1, c->stream));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (shape == 0 || (double)ms / iters < best) best = (double)ms / iters;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_pass = best;
    return ADSB_OK;
}

// ---- measurement: the host-fed (PCIe-inclusive) rate of the streaming front end ---------------------------------------------
// What the reference's thread 1 -> thread 2 hand-over costs per buffer through adsb_feed_* (src/adsb.rs:75-89: playback
// sends 20 000-sample buffers; adsb.rs:59-64: MTU-sized reads), measured from C (no interpreter in the loop): a context
// and a feed of its own on `device`, synthetic samples written into the pinned ring by an in-place producer (acquire /
// push), two buffers in flight, every list popped; runs for about `seconds`.  bench.py reports it next to `value`
// (which is the HBM-resident rate and never includes PCIe).
extern "C" int adsb_measure_feed(int device, int sample_type, size_t chunk, double seconds, double *us_per_buffer,
                                 double *frames_per_buffer, uint64_t *buffers)
{
    if (!us_per_buffer || chunk < (size_t)kWindow || seconds <= 0 || (sample_type != ADSB_SAMPLE_I8 && sample_type != ADSB_SAMPLE_I16))
        return ADSB_E_ARG;
    const size_t bps = sample_type == ADSB_SAMPLE_I8 ? 2 : 4;
    adsb_synth_cfg sc;
    adsb_synth_default(&sc);
    sc.seed = 9;
    if (sample_type == ADSB_SAMPLE_I16) sc.amp_shift = 5;
    const int n_src = chunk <= (1u << 20) ? 8 : 2;
    std::vector<char> data(chunk * bps * n_src);
    int rc = adsb_synth_fill_host(&sc, sample_type, 0, 0, chunk * n_src, data.data());
    if (rc != ADSB_OK) return rc;
    adsb_cfg cfg{};
    cfg.abi_version = ADSB_ABI_VERSION;
    cfg.device = device;
    cfg.sample_type = sample_type;
    cfg.max_channels = 1;
    cfg.max_samples = chunk + kWindow;
    cfg.max_out = chunk / 200 + 4096;
    adsb_ctx *ctx = nullptr;
    if ((rc = adsb_create(&cfg, &ctx)) != ADSB_OK) return rc;
    adsb_feed_cfg fc{};
    fc.max_chunk = chunk;
    fc.carry = 0;
    fc.ring_slots = 3;
    adsb_feed *feed = nullptr;
    if ((rc = adsb_feed_open(ctx, &fc, &feed)) != ADSB_OK) { adsb_destroy(ctx); return rc; }
    std::vector<adsb_frame> frames(cfg.max_out);
    uint64_t total = 0, n_buf = 0;
    auto one = [&](uint64_t k, bool fill) {
        void *slot = nullptr;
        int r = adsb_feed_acquire(feed, &slot);
        // (a real producer -- SDR driver, file reader -- writes its samples straight into the slot: that is its cost, not
        // the hand-over's; the slots are filled during the warm-up rounds)
        if (r == ADSB_OK && fill) std::memcpy(slot, data.data() + (size_t)(k % n_src) * chunk * bps, chunk * bps);
        if (r == ADSB_OK) r = adsb_feed_push(feed, nullptr, chunk);
        if (r == ADSB_OK && adsb_feed_in_flight(feed) == 2) {
            size_t n = 0;
            r = adsb_feed_pop(feed, frames.data(), frames.size(), &n, nullptr, nullptr);
            total += n;
        }
        return r;
    };
    auto drain = [&]() {
        int r = ADSB_OK;
        while (r == ADSB_OK && adsb_feed_in_flight(feed) > 0) {
            size_t n = 0;
            r = adsb_feed_pop(feed, frames.data(), frames.size(), &n, nullptr, nullptr);
            total += n;
        }
        return r;
    };
    for (uint64_t k = 0; k < 6 && rc == ADSB_OK; ++k) rc = one(k, true);
    if (rc == ADSB_OK) rc = drain();
    total = 0;
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    double dt = 0;
    while (rc == ADSB_OK) {
        for (int k = 0; k < 16 && rc == ADSB_OK; ++k, ++n_buf) rc = one(n_buf + 6, false);
        dt = std::chrono::duration<double>(clk::now() - t0).count();
        if (dt >= seconds) break;
    }
    if (rc == ADSB_OK) rc = drain();
    dt = std::chrono::duration<double>(clk::now() - t0).count();
    adsb_feed_close(feed);
    adsb_destroy(ctx);
    if (rc != ADSB_OK) return rc;
    *us_per_buffer = n_buf ? dt * 1e6 / (double)n_buf : 0.0;
    if (frames_per_buffer) *frames_per_buffer = n_buf ? (double)total / (double)n_buf : 0.0;
    if (buffers) *buffers = n_buf;
    return ADSB_OK;
}

// The box's pinned host -> device copy rate (one stream, `bytes` per copy): the ceiling of any host-fed path.
extern "C" int adsb_measure_pinned_copy(int device, size_t bytes, int iters, double *gbytes_per_s)
{
    if (!gbytes_per_s || bytes == 0 || iters <= 0) return ADSB_E_ARG;
    HIPCHK(hipSetDevice(device));
    void *h = nullptr, *d = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipHostMalloc(&h, bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&d, bytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms = 0;
    if (e == hipSuccess) {
        std::memset(h, 1, bytes);
        e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); // warm-up
        if (e == hipSuccess) e = hipEventRecord(e0, s);
        for (int k = 0; k < iters && e == hipSuccess; ++k) e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s) (void)hipStreamDestroy(s);
    (void)hipFree(d);
    if (h) (void)hipHostFree(h);
    if (e != hipSuccess) return (int)e;
    *gbytes_per_s = ms > 0 ? (double)bytes * iters / (ms * 1e-3) / 1e9 : 0.0;
    return ADSB_OK;
}

extern "C" int adsb_debug_magnitudes(adsb_ctx *c, const void *iq_host, size_t n, uint16_t *mags_host)
{
    if (!c || !iq_host || !mags_host) return ADSB_E_ARG;
    if (n == 0) return ADSB_OK;
    HIPCHK(hipSetDevice(c->cfg.device));
    void *d_in = nullptr;
    uint16_t *d_out = nullptr;
    HIPCHK(hipMalloc(&d_in, n * c->bps + 16));
    hipError_t e = hipMalloc((void **)&d_out, n * sizeof(uint16_t));
    if (e != hipSuccess) { (void)hipFree(d_in); return (int)e; }
    int rc = ADSB_OK;
    do {
        if ((e = hipMemcpyAsync(d_in, iq_host, n * c->bps, hipMemcpyHostToDevice, c->stream)) != hipSuccess) break;
        if ((e = adsbk::launch_magnitudes(c->stream, c->cfg.sample_type, c->mag_mode, d_in, n, d_out)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(mags_host, d_out, n * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) break;
        e = hipStreamSynchronize(c->stream);
    } while (0);
    if (e != hipSuccess) rc = (int)e;
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
}

extern "C" int adsb_debug_nsq_values(adsb_ctx *c, const void *iq_host, size_t n, uint16_t *vals_host)
{
    if (!c || !iq_host || !vals_host) return ADSB_E_ARG;
    if (c->cfg.sample_type != ADSB_SAMPLE_I8) return ADSB_E_STATE;
    if (n == 0) return ADSB_OK;
    HIPCHK(hipSetDevice(c->cfg.device));
    void *d_in = nullptr;
    uint16_t *d_out = nullptr;
    HIPCHK(hipMalloc(&d_in, n * 2 + 16));
    hipError_t e = hipMalloc((void **)&d_out, n * sizeof(uint16_t));
    if (e != hipSuccess) { (void)hipFree(d_in); return (int)e; }
    do {
        if ((e = hipMemcpyAsync(d_in, iq_host, n * 2, hipMemcpyHostToDevice, c->stream)) != hipSuccess) break;
        if ((e = adsbk::launch_nsq_values(c->stream, d_in, n, d_out)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(vals_host, d_out, n * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) break;
        e = hipStreamSynchronize(c->stream);
    } while (0);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return e == hipSuccess ? ADSB_OK : (int)e;
}

// ---- synthetic source ------------------------------------------------------------------------------
extern "C" void adsb_synth_default(adsb_synth_cfg *s)
{
    if (!s) return;
    std::memset(s, 0, sizeof(*s));
    s->seed = 0x0AD5B0001ull;
    s->slot_len = 2000;  // one frame slot per millisecond of 2 MSPS signal
    s->frame_pct = 100;
    s->pct_flip_data = 5;
    s->pct_flip_crc = 2;
    s->pct_flip_two = 3;
    s->noise_div = 18;   // Irwin-Hall(4 bytes)/18: sigma ~ 8.2 LSB
    s->amp_shift = 0;
}

static bool synth_ok(const adsb_synth_cfg *s)
{
    return s && s->slot_len >= 256 && s->frame_pct <= 100 && s->noise_div >= 1 && s->amp_shift <= 7 &&
           s->pct_flip_data + s->pct_flip_crc + s->pct_flip_two <= 100;
}

extern "C" int adsb_synth_slot(const adsb_synth_cfg *cfg, uint32_t channel, uint64_t slot,
                               uint64_t *start, uint8_t clean14[14], uint8_t sent14[14], int *kind)
{
    if (!synth_ok(cfg)) return ADSB_E_ARG;
    adsb_synth::Slot s;
    adsb_synth::slot_params(*cfg, channel, slot, s);
    if (start) *start = slot * (uint64_t)cfg->slot_len + s.jitter;
    if (clean14) std::memcpy(clean14, s.clean, 14);
    if (sent14) std::memcpy(sent14, s.sent, 14);
    if (kind) *kind = s.kind;
    return s.present ? 1 : 0;
}

extern "C" int adsb_synth_fill_host(const adsb_synth_cfg *cfg, int sample_type, uint32_t channel,
                                    uint64_t first, size_t n, void *iq)
{
    if (!synth_ok(cfg) || (!iq && n) || (sample_type != ADSB_SAMPLE_I8 && sample_type != ADSB_SAMPLE_I16))
        return ADSB_E_ARG;
    int8_t *o8 = (int8_t *)iq;
    int16_t *o16 = (int16_t *)iq;
    uint64_t cur_slot = ~0ull;
    adsb_synth::Slot s{};
    for (size_t j = 0; j < n; ++j) {
        const uint64_t k = first + j;
        const uint64_t slot = k / cfg->slot_len;
        if (slot != cur_slot) {
            adsb_synth::slot_params(*cfg, channel, slot, s);
            cur_slot = slot;
        }
        int vi, vq;
        adsb_synth::sample_iq(*cfg, channel, k, s, slot, vi, vq);
        if (sample_type == ADSB_SAMPLE_I8) {
            o8[2 * j] = (int8_t)adsb_synth::clip8(vi);
            o8[2 * j + 1] = (int8_t)adsb_synth::clip8(vq);
        } else {
            int wi = vi << cfg->amp_shift, wq = vq << cfg->amp_shift;
            wi = wi < -32768 ? -32768 : (wi > 32767 ? 32767 : wi);
            wq = wq < -32768 ? -32768 : (wq > 32767 ? 32767 : wq);
            o16[2 * j] = (int16_t)wi;
            o16[2 * j + 1] = (int16_t)wq;
        }
    }
    return ADSB_OK;
}

extern "C" int adsb_synth_fill_device(adsb_ctx *c, const adsb_synth_cfg *cfg, uint32_t channel,
                                      uint64_t first, size_t n, void *iq_dev)
{
    if (!c || !synth_ok(cfg) || (!iq_dev && n)) return ADSB_E_ARG;
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(adsbk::launch_synth(c->stream, *cfg, c->cfg.sample_type, channel, first, n, iq_dev));
    return ADSB_OK;
}

// ---- streaming front end (reference: the Vec-per-recv loop of src/adsb.rs:95-98 and its feeders,
// adsb.rs:54-89) --------------------------------------------------------------------------------------------
// Buffers arrive on the host one after the other.  Each is copied into a slot of a PINNED host ring (or was
// written there by the producer: adsb_feed_acquire), goes to one of two device staging slots by asynchronous DMA
// on a copy stream, and is demodulated on the ctx stream -- so the copy of buffer k+1 overlaps the kernels of
// buffer k, and the host only blocks in adsb_feed_pop() for results that are not ready yet.  Two buffers may be
// in flight (the ctx alternates between two result sets).
//   parity mode (carry = 0, the reference's behaviour): every buffer is its own reference buffer; the last 240
//     offsets of each are never looked at (adsb.rs:98); offsets are buffer-relative.
//   carry mode (carry = 1): the last 240 samples of the stream so far are kept ON THE DEVICE and copied, device
//     to device, in front of the next buffer before it is demodulated: the chunked stream decodes exactly like one
//     long buffer (frames straddling two buffers are found); offsets are absolute stream positions.
// Staging slot layout (samples): [ head: up to 240 carried samples | the buffer ], the buffer's copy lands behind
// the carried samples and the demodulated region starts at a multiple of 8 samples (16-byte alignment).
struct adsb_feed {
    adsb_ctx *c = nullptr;
    adsb_feed_cfg cfg{};
    uint32_t bps = 2;
    hipStream_t copy = nullptr;
    char *dev[2] = {nullptr, nullptr};
    hipEvent_t h2d_done[2] = {nullptr, nullptr}, kern_done[2] = {nullptr, nullptr};
    // results of the launch from staging slot k are complete and visible to the copy engine (a default event: its
    // record releases to system scope, unlike the ctx's device-scope events)
    bool kern_pending[2] = {false, false};
    std::vector<char *> ring;
    std::vector<hipEvent_t> ring_done; // H2D out of that ring slot has completed
    std::vector<char> ring_busy;
    // a one-dispatch launch reads its samples straight from the ring slot: the slot may be overwritten once the launch's
    // sequence word says so (ring_seq != 0: result set ring_set carries that number when the kernel has finished)
    std::vector<uint64_t> ring_seq;
    std::vector<uint32_t> ring_set;
    uint32_t ring_next = 0;
    int acquired = -1;
    uint64_t pushed = 0, popped = 0; // buffers
    uint64_t consumed = 0;           // samples pushed so far
    size_t prev_start = 0, prev_len = 0; // demodulated region of the previous buffer's slot (samples)
    struct Entry {
        bool launched = false;
        bool small = false;        // went through the one-dispatch path: results in the ctx's pinned blob `set`
        uint64_t seq = 0;          // ... complete when its sequence word carries this
        uint32_t set = 0;
        uint64_t first_sample = 0; // stream position of the buffer's first own sample
    } q[2];
    adsbk::Header *hdr_host = nullptr;
    size_t headroom = 0;           // samples in front of every ring slot's data (carry mode: room for the 240-sample tail)
    const char *prev_host = nullptr; // where the previous buffer's demodulated region starts in HOST memory (nullptr: it
                                     // only exists in a device staging slot)
    uint32_t prev_ring = 0;          // ... and the ring slot that holds it
};

extern "C" void adsb_feed_close(adsb_feed *f)
{
    if (!f) return;
    if (f->c) (void)hipSetDevice(f->c->cfg.device);
    if (f->copy) (void)hipStreamSynchronize(f->copy);
    if (f->c && f->c->stream) (void)hipStreamSynchronize(f->c->stream);
    for (int k = 0; k < 2; ++k) {
        (void)hipFree(f->dev[k]);
        if (f->h2d_done[k]) (void)hipEventDestroy(f->h2d_done[k]);
        if (f->kern_done[k]) (void)hipEventDestroy(f->kern_done[k]);
    }
    for (char *p : f->ring) (void)hipHostFree(p);
    for (hipEvent_t e : f->ring_done) if (e) (void)hipEventDestroy(e);
    if (f->hdr_host) (void)hipHostFree(f->hdr_host);
    if (f->copy) (void)hipStreamDestroy(f->copy);
    if (f->c) { (void)adsb_set_stream_base(f->c, 0); }
    delete f;
}

extern "C" int adsb_feed_open(adsb_ctx *c, const adsb_feed_cfg *cfg, adsb_feed **out)
{
    if (!c || !cfg || !out || cfg->max_chunk == 0) return ADSB_E_ARG;
    *out = nullptr;
    // one launch covers the buffer, in carry mode with the 240 carried samples in front of it
    if (cfg->max_chunk + (cfg->carry ? (size_t)kWindow : 0) > c->cfg.max_samples) return ADSB_E_CAPACITY;
    if (c->cfg.max_channels < 1) return ADSB_E_ARG;
    // two launches are in flight: they must not share one caller-owned result blob
    if (c->ext_blob) return ADSB_E_STATE;
    HIPCHK(hipSetDevice(c->cfg.device));
    adsb_feed *f = new (std::nothrow) adsb_feed();
    if (!f) return ADSB_E_NOMEM;
    f->c = c;
    f->cfg = *cfg;
    if (f->cfg.ring_slots < 2) f->cfg.ring_slots = 3;
    f->bps = c->bps;
    bool ok = hipStreamCreateWithFlags(&f->copy, hipStreamNonBlocking) == hipSuccess;
    f->headroom = kWindow; // (240 samples = 480 / 960 bytes: keeps the data 16-byte aligned)
    const size_t slot_bytes = (cfg->max_chunk + (size_t)kWindow + 8) * f->bps;
    for (int k = 0; k < 2 && ok; ++k)
        ok = hipMalloc((void **)&f->dev[k], slot_bytes) == hipSuccess &&
             hipEventCreateWithFlags(&f->h2d_done[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&f->kern_done[k], hipEventDisableTiming) == hipSuccess;
    for (uint32_t k = 0; k < f->cfg.ring_slots && ok; ++k) {
        char *p = nullptr;
        hipEvent_t e = nullptr;
        ok = hipHostMalloc((void **)&p, (cfg->max_chunk + f->headroom) * f->bps, hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (p) f->ring.push_back(p);
        if (e) f->ring_done.push_back(e);
        f->ring_busy.push_back(0);
        f->ring_seq.push_back(0);
        f->ring_set.push_back(0);
    }
    ok = ok && hipHostMalloc((void **)&f->hdr_host, sizeof(adsbk::Header), hipHostMallocDefault) == hipSuccess;
    if (!ok) { adsb_feed_close(f); return ADSB_E_NOMEM; }
    *out = f;
    return ADSB_OK;
}

extern "C" int adsb_feed_in_flight(const adsb_feed *f) { return f ? (int)(f->pushed - f->popped) : ADSB_E_ARG; }

// 1: adsb_feed_pop() would not wait for the GPU (the oldest buffer's kernels have finished, or it launched none);
// 0: it would; ADSB_E_STATE: nothing in flight.
extern "C" int adsb_feed_ready(adsb_feed *f)
{
    if (!f) return ADSB_E_ARG;
    if (f->popped == f->pushed) return ADSB_E_STATE;
    const uint32_t s = (uint32_t)(f->popped & 1u);
    if (!f->q[s].launched) return 1;
    if (f->q[s].small) return __atomic_load_n(small_seq_word(f->c, f->q[s].set), __ATOMIC_ACQUIRE) == f->q[s].seq ? 1 : 0;
    HIPCHK(hipSetDevice(f->c->cfg.device));
    const hipError_t e = hipEventQuery(f->kern_done[s]);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return (int)e;
}

static int feed_ring_slot(adsb_feed *f, uint32_t *slot)
{
    const uint32_t r = f->ring_next;
    if (f->ring_busy[r]) { // the DMA out of this slot must have finished before the host overwrites it
        HIPCHK(hipEventSynchronize(f->ring_done[r]));
        f->ring_busy[r] = 0;
    }
    if (f->ring_seq[r]) { // ... and so must the one-dispatch kernel that reads its samples from the slot itself
        const int rc = small_wait(f->c, f->ring_set[r], f->ring_seq[r], true);
        if (rc != ADSB_OK) return rc;
        f->ring_seq[r] = 0;
    }
    *slot = r;
    return ADSB_OK;
}

extern "C" int adsb_feed_acquire(adsb_feed *f, void **host_slot)
{
    if (!f || !host_slot) return ADSB_E_ARG;
    if (f->acquired >= 0) return ADSB_E_STATE;
    HIPCHK(hipSetDevice(f->c->cfg.device));
    uint32_t r = 0;
    int rc = feed_ring_slot(f, &r);
    if (rc != ADSB_OK) return rc;
    f->acquired = (int)r;
    *host_slot = f->ring[r] + f->headroom * f->bps;
    return ADSB_OK;
}

extern "C" int adsb_feed_push(adsb_feed *f, const void *iq_host, size_t n)
{
    if (!f || n == 0 || n > f->cfg.max_chunk) return ADSB_E_ARG;
    if (!iq_host && f->acquired < 0) return ADSB_E_ARG;
    if (f->pushed - f->popped >= 2) return ADSB_E_STATE; // two buffers in flight: pop first
    // parity mode: the reference panics on a buffer shorter than 240 samples (adsb.rs:98); nothing is consumed
    if (!f->cfg.carry && n < (size_t)kWindow) return ADSB_E_SHORT;
    adsb_ctx *c = f->c;
    HIPCHK(hipSetDevice(c->cfg.device));
    uint32_t r = 0;
    char *data = nullptr; // the buffer's samples in the pinned ring
    if (f->acquired >= 0) {
        r = (uint32_t)f->acquired;
        data = f->ring[r] + f->headroom * f->bps;
        if (iq_host && iq_host != data) std::memcpy(data, iq_host, n * f->bps);
        f->acquired = -1;
    } else {
        int rc = feed_ring_slot(f, &r);
        if (rc != ADSB_OK) return rc;
        data = f->ring[r] + f->headroom * f->bps;
        std::memcpy(data, iq_host, n * f->bps);
    }
    f->ring_next = (r + 1) % (uint32_t)f->ring.size();

    const uint32_t s = (uint32_t)(f->pushed & 1u);
    const size_t tail = (f->cfg.carry && f->pushed) ? std::min<size_t>(kWindow, f->prev_len) : 0;
    const size_t len = tail + n;
    adsb_feed::Entry &e = f->q[s];
    e.first_sample = f->consumed;
    e.launched = false;
    e.small = false;
    const uint64_t base = f->cfg.carry ? f->consumed - tail : 0;

    // ---- small buffers: one dispatch, samples read from the ring and frames written to pinned memory by the device ----
    // (carry mode: the tail of the previous buffer is copied in front of this one on the host -- 480 or 960 bytes --
    // which needs that buffer in host memory and a 16-byte aligned start, i.e. the usual 240-sample tail)
    if (len >= (size_t)kWindow && small_fits(c, len) && (tail == 0 || (f->prev_host && (tail * f->bps) % 16 == 0))) {
        char *start = data - tail * f->bps;
        if (tail) std::memcpy(start, f->prev_host + (f->prev_len - tail) * f->bps, tail * f->bps);
        int rc = adsb_set_stream_base(c, base);
        if (rc == ADSB_OK) rc = small_launch(c, start, len, &e.seq);
        if (rc != ADSB_OK) return rc;
        e.launched = true;
        e.small = true;
        e.set = c->last;
        f->ring_busy[r] = 0;
        f->ring_seq[r] = e.seq; // the kernel reads the slot until its sequence word is out (feed_ring_slot waits for it)
        f->ring_set[r] = e.set;
        f->prev_host = start;
        f->prev_ring = r;
        f->prev_start = 0;
        f->prev_len = len;
        f->consumed += n;
        f->pushed++;
        return ADSB_OK;
    }

    const size_t start = ((size_t)kWindow - tail) / 8 * 8; // 16-byte aligned start of the demodulated region
    // this staging slot was last read by the launch two buffers ago
    if (f->kern_pending[s]) HIPCHK(hipStreamWaitEvent(f->copy, f->kern_done[s], 0));
    HIPCHK(hipMemcpyAsync(f->dev[s] + (start + tail) * f->bps, data, n * f->bps, hipMemcpyHostToDevice, f->copy));
    HIPCHK(hipEventRecord(f->ring_done[r], f->copy));
    f->ring_busy[r] = 1;
    if (tail) { // the last `tail` samples of what the previous launch saw
        if (f->prev_host) { // ... which went through the one-dispatch path: they are in the ring (host memory)
            HIPCHK(hipMemcpyAsync(f->dev[s] + start * f->bps, f->prev_host + (f->prev_len - tail) * f->bps, tail * f->bps,
                                  hipMemcpyHostToDevice, f->copy));
            HIPCHK(hipEventRecord(f->ring_done[f->prev_ring], f->copy)); // that slot is read once more: not reusable before
            f->ring_busy[f->prev_ring] = 1;
        } else // ... device to device (its H2D is earlier on this stream)
            HIPCHK(hipMemcpyAsync(f->dev[s] + start * f->bps, f->dev[s ^ 1u] + (f->prev_start + f->prev_len - tail) * f->bps,
                                  tail * f->bps, hipMemcpyDeviceToDevice, f->copy));
    }
    HIPCHK(hipEventRecord(f->h2d_done[s], f->copy));

    if (len >= (size_t)kWindow) {
        HIPCHK(hipStreamWaitEvent(c->stream, f->h2d_done[s], 0));
        // after a pop of an older launch the ctx may "view" that launch: the next enqueue starts from the newest state
        int rc = adsb_set_stream_base(c, base);
        if (rc == ADSB_OK) rc = adsb_demod_device_async(c, f->dev[s] + start * f->bps, 1, len, len);
        if (rc != ADSB_OK) return rc;
        e.launched = true;
        e.set = c->last;
        HIPCHK(hipEventRecord(f->kern_done[s], c->aux)); // THIS launch's kernels and results (not the stream's tail)
        f->kern_pending[s] = true;
    }
    f->prev_host = nullptr;
    f->prev_start = start;
    f->prev_len = len;
    f->consumed += n;
    f->pushed++;
    return ADSB_OK;
}

extern "C" int adsb_feed_pop(adsb_feed *f, adsb_frame *out, size_t max_out, size_t *n_out, uint32_t *flags,
                             uint64_t *first_sample)
{
    if (!f || !n_out || (!out && max_out)) return ADSB_E_ARG;
    *n_out = 0;
    if (flags) *flags = 0;
    if (f->popped == f->pushed) return ADSB_E_STATE;
    adsb_ctx *c = f->c;
    HIPCHK(hipSetDevice(c->cfg.device));
    adsb_feed::Entry &e = f->q[f->popped & 1u];
    if (first_sample) *first_sample = e.first_sample;
    f->popped++;
    if (!e.launched) return ADSB_OK; // (carry mode, fewer than 240 samples so far: nothing decodable yet)
    if (e.small) { // the device wrote header and frames into pinned memory: poll its sequence word, copy, done
        int rc = small_wait(c, e.set, e.seq);
        if (rc != ADSB_OK) return rc;
        return small_collect(c, e.set, out, max_out, n_out, nullptr, flags);
    }
    adsb_ctx::ResultSet &r = c->rs[e.set];
    // results travel on the copy stream, behind this launch's ordering pass only -- not behind the kernels of the
    // buffer pushed after it, which share the ctx stream
    HIPCHK(hipStreamWaitEvent(f->copy, f->kern_done[(f->popped - 1) & 1u], 0));
    HIPCHK(hipMemcpyAsync(f->hdr_host, r.hdr, sizeof(adsbk::Header), hipMemcpyDeviceToHost, f->copy));
    HIPCHK(hipStreamSynchronize(f->copy));
    if (f->hdr_host->retry) { // slot-pool overflow (pathological input): the slow, fully synchronous path
        const uint32_t newest = c->last;
        view_launch(c, e.set);
        size_t got = 0;
        uint32_t fl = 0;
        int rc = adsb_fetch(c, out, max_out, &got, nullptr, nullptr, &fl);
        view_launch(c, newest);
        if (rc != ADSB_OK) return rc;
        *n_out = got;
        if (flags) *flags = fl;
        return ADSB_OK;
    }
    uint64_t n = f->hdr_host->n_out;
    uint32_t fl = f->hdr_host->flags;
    if (n > max_out) { n = max_out; fl |= ADSB_FLAG_TRUNCATED; }
    if (n) {
        HIPCHK(hipMemcpyAsync(out, r.li.out, sizeof(adsb_frame) * n, hipMemcpyDeviceToHost, f->copy));
        HIPCHK(hipStreamSynchronize(f->copy));
    }
    *n_out = (size_t)n;
    if (flags) *flags = fl;
    return ADSB_OK;
}
