// adsb_kernels.h -- launch interface between the C-ABI layer (adsb_api.cpp) and the gfx950
// kernels (adsb_kernels.hip).  Internal; the public boundary is include/adsb_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/adsb_hip.h"

namespace adsbk {

// ---- tiling constants (see DESIGN.md "Data layout") ---------------------------------------
#ifndef ADSB_KRUN
#define ADSB_KRUN 32
#endif
#ifndef ADSB_THREADS
#define ADSB_THREADS 256
#endif
constexpr int kThreads = ADSB_THREADS; // 4 waves per workgroup (a multiple of 64; the tile length scales with it)
constexpr int kRun = ADSB_KRUN; // consecutive offsets one lane slides over, per packed half (<= 64)
constexpr int kTile = 2 * kThreads * kRun; // offsets owned by one workgroup (16384 at kRun 32)
constexpr int kHalo = 256;      // >= 239 extra samples so PPM never leaves the tile; 16-aligned
constexpr int kMag = kTile + kHalo;
// CS16 keeps u16 magnitudes in LDS.  Runs of 16 offsets = 8192-offset tiles, 19 KB: the kernel's 85 VGPRs then allow
// five workgroups per CU (20 waves) where 16384-offset tiles (35 KB) allowed four -- and the finer tiles overlap better:
// 0.177 -> 0.161 ms per GiB (0.76 -> 0.83 of 8 TB/s; profiles/r03_ab_cs16_tile8192.txt).  Holding the kernel to 80 or
// 64 VGPRs for six or eight workgroups spills and is slower (0.171 / 0.188 ms).
#ifndef ADSB_KRUN_I16
#define ADSB_KRUN_I16 16
#endif
constexpr int kRunI16 = ADSB_KRUN_I16;
template <int ST> struct TileCfg {
    static constexpr int kRunT = ST == ADSB_SAMPLE_I8 ? kRun : kRunI16;
    static constexpr int kTileT = 2 * kThreads * kRunT;
    static constexpr int kMagT = kTileT + kHalo;
};
constexpr int tile_offsets(int sample_type)
{
    return sample_type == ADSB_SAMPLE_I8 ? kTile : 2 * kThreads * kRunI16;
}
constexpr int kTileMax = kTile > 2 * kThreads * kRunI16 ? kTile : 2 * kThreads * kRunI16;
// Which i8 scan kernel a context launches (adsb_create reads ADSB_SCAN from the environment; default root):
//   kScanRoot: floor(sqrt(n)) per sample (v_sqrt_f32), u8 magnitudes in LDS -- the product's kernel
//   kScanNsq : the gate runs on n = I^2+Q^2 (no root per sample; exact: DESIGN.md section 4.1b), 2 bytes of LDS per
//              sample; the round-3 A/B kernel (fewer VALU slots, half the resident workgroups: slower)
constexpr int kScanNsq = 0, kScanRoot = 1, kScanReg = 2, kScanCode = 3, kScanSieve = 4;
//   kScanCode: the gate slides over an 8-bit LOG code of n = I^2+Q^2 (one quarter-rate v_cvt_pk_fp8_f32 per pair of samples
//              instead of a root per sample); a superset test on codes, the few uncertain survivors are decided from the
//              samples themselves (adsb_kernels.hip, "the code scan"): a round-4 A/B kernel (bit-exact, not faster)
//   kScanSieve: one pair of relation bits per sample (neighbouring samples compared, no root), the gate's fourteen adjacent taps as
//              shifts and ANDs of 64-bit words, the few candidates decided exactly from the raw samples kept in LDS
//              (adsb_sieve.inc): the round-4 A/B kernel (bit-exact, 0.200 ms against the root scan's 0.190)
//   kScanReg : the nsq gate from registers, no LDS image (every wave a chunk of 4032 offsets; window overlap by DPP from
//              the neighbouring lane); tiles of 16128 offsets
constexpr int kRegTile = 4 * 2 * 63 * 32; // offsets per tile of the register scan: four waves x 4032
#ifndef ADSB_SV_SWEEPS
#define ADSB_SV_SWEEPS 8                  // 16-byte loads per lane and tile of the sieve scan (8: 16384-offset tiles, four workgroups
#endif                                    // per CU; 6: 12288, five -- measured no faster, profiles/r04_ab_sieve.txt)
constexpr int kSieveTile = ADSB_SV_SWEEPS * kThreads * 8; // offsets per tile of the sieve scan (adsb_sieve.inc)
// offsets per tile of a context (the scan kind is fixed at adsb_create)
constexpr int tile_offsets_of(int sample_type, int scan)
{
    return (sample_type == ADSB_SAMPLE_I8 && scan == kScanReg) ? kRegTile
           : (sample_type == ADSB_SAMPLE_I8 && scan == kScanSieve) ? kSieveTile : tile_offsets(sample_type);
}
constexpr int kListCap = 128;   // candidate offsets staged per decode chunk
constexpr int kSparseCap = 64;  // up to this many gate survivors per tile take the cheap (rank-sort) path
constexpr int kWindow = 240;    // 16 + 112*2  (reference src/adsb.rs:98)
constexpr uint32_t kNoBase = 0xFFFFFFFFu;
// Every tile owns kQuota frame slots at a fixed place (slot = tile * kQuota + i), so the normal
// case needs no allocation at all; only a tile with more gate survivors than that draws from the
// shared pool with one returning atomic.  (One atomic per tile on a single address measured as the
// bottleneck: ~80 atomics/us per address vs 62 tiles/us.)
constexpr uint32_t kQuota = 32;
// One entry per tile, written unconditionally by the demod kernel.
struct Seg {
    uint32_t base;    // first temp slot of this tile, kNoBase if the slot store was full
    uint32_t cand;    // offsets that passed the preamble+DF17 gate (slots reserved; the scan kernel writes each survivor's offset there)
    uint32_t valid;   // of those, frames that passed CRC / single-bit repair (written by the finishing kernel)
    uint32_t decoded; // 1: `valid` is already final when the scan kernel ends (a tile without slots, counted in place)
};

// Device-resident result header (adsb_result_device).
struct Header {
    uint64_t n_out;        // frames in the final list (<= max_out)
    uint64_t total_found;  // frames that exist
    uint32_t flags;        // ADSB_FLAG_*
    uint32_t retry;        // internal: a needed tile lost its slots (slot store overflow)
    unsigned long long alloc; // pool allocator for tiles over their quota (reset by the gather kernel)
};

struct DemodArgs {
    const void *iq;            // channel 0, sample 0
    uint64_t n_samples;        // per channel
    uint64_t channel_stride;   // samples
    uint32_t tiles_per_channel;
    uint32_t tile_first;       // global tile id of blockIdx.x == 0
    uint32_t tile_count;       // workgroups in this launch
    uint32_t count_groups;     // 1: first pass of a launch (clears the header's flags); 0: re-run of known tiles
    uint32_t fused_pass_only;  // measurement (adsb_debug_fused_pass_only): magnitude + gate only, survivors counted but not decoded
    uint64_t offset_base;      // added to every frame's offset (adsb_set_stream_base: position of sample 0 in a longer stream)
    Seg *seg;
    adsb_frame *slots;         // [n_tiles_max * kQuota] fixed region, then the pool
    uint32_t pool_first;       // index of the pool's first slot
    uint32_t cap_slots;        // pool capacity
    Header *hdr;
    uint64_t *hdr_pub;         // optional caller-owned header copy (adsb_set_result_target): its flags word is cleared here
    uint32_t pool_off;         // test knob (adsb_debug_pool_limit): 1 = the shared slot pool hands out nothing
    unsigned long long *stamps; // diagnostic builds (-DADSB_TILE_STAMPS=1) only: 64 bytes of cycle counters per tile
};

// finish_order: CRC-24 / repair of the survivors + the ordered list, in one kernel (see adsb_kernels.hip)
struct FinishArgs {
    Seg *seg;
    adsb_frame *slots;
    adsb_frame *out;
    const uint32_t *out_start; // optional [n_tiles]: host-planned positions (re-run of lost tiles): no look-back, no header
    uint64_t *lb;              // exchange words (value | flag | epoch), never cleared, tagged with `epoch`: one per
    uint32_t lb_groups_at;     // workgroup from lb[0], one per 64 workgroups from lb[lb_groups_at]
    uint64_t *chan_prefix;     // optional [n_channels + 1]: frames before each channel's first tile; [n_channels] = total
    uint32_t epoch;            // launch index + 1 (30 bits)
    uint32_t tiles_per_channel;
    uint32_t n_channels;
    uint32_t max_out;
    uint32_t tile_first, tile_count;
    Header *hdr;
    uint64_t *hdr_pub;         // optional caller-owned copy of {n_out, total_found, flags} (4 x u64)
    uint32_t stall_blk;        // test knob (adsb_debug_finish_stall): this workgroup withholds its exchange word; 0xFFFFFFFF: none
};

// demod_small: the one-dispatch path for buffers of at most kFinishTilesPerWg tiles
struct SmallArgs {
    uint32_t *done;      // device word, zero between launches: workgroups that have finished their tile
    uint64_t *seq_host;  // host-visible (pinned) word the last workgroup writes `seq` to when header and list are complete
    uint64_t seq;
};

// mag_mode: how v_cvt_pk_u8_f32 rounds on this device (decided once per ctx by probe_cvt):
//   0: truncates as is; 1: truncates once MODE.fp_round(f32) is set to round-toward-zero;
//   2: rounds to nearest regardless -> subtract 0.5 first.
hipError_t probe_cvt(hipStream_t s, uint32_t *dev_scratch4, uint32_t host_out[4]);
// the code scan's arithmetic on every n = 0 .. 32768: out[n] = c(n) | byte1(S(c(n) << 8)) << 8 (device buffer of 32769 u16)
hipError_t launch_code_probe(hipStream_t s, uint16_t *dev_out32769);

// e0/e1: optional events recorded at the start / end of the dispatch itself (nullptr: none)
// scan: kScanNsq / kScanRoot (i8 only; CS16 has one kernel)
hipError_t launch_demod(hipStream_t s, int sample_type, int mag_mode, int scan, const DemodArgs &a,
                        hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// second kernel of a launch (finish_order): CRC-24 + single-bit repair of the survivors the scan kernel sliced into
// their slots, and the ordered frame list
hipError_t launch_finish(hipStream_t s, const FinishArgs &a, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// the header of an empty list (no tiles, or a measurement launch without the finishing kernel)
hipError_t launch_empty_result(hipStream_t s, Header *hdr, uint64_t *hdr_pub, uint64_t *chan_prefix, uint32_t n_channels,
                               hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
constexpr int kFinishTilesPerWg = 32;
// scan + finish in one dispatch for 1..kFinishTilesPerWg tiles; the list goes to f.out / f.hdr_pub (host-visible memory)
hipError_t launch_small(hipStream_t s, int sample_type, int mag_mode, int scan, const DemodArgs &p, const FinishArgs &f,
                        const SmallArgs &sm);

bool ab_kernels_built();  // -DADSB_AB_KERNELS=1: the A/B scan kernels (nsq, reg, code) are in this library
bool tile_stamps_built(); // -DADSB_TILE_STAMPS=1 diagnostic build: DemodArgs::stamps holds 64 bytes per tile

// field decode of an ordered frame list (count read from hdr->n_out on the device)
hipError_t launch_decode_fields(hipStream_t s, const adsb_frame *frames, const Header *hdr, uint32_t cap,
                                adsb_packet_fields *out);

// tracker + CPR position decode over an ordered frame list (adsb_track.hip)
struct TrackArgs {
    const adsb_frame *frames;
    const adsb_packet_fields *fields;
    uint32_t n;                  // frames in the list (host value)
    double seconds_per_sample;
    uint32_t *keys, *vals, *skeys, *svals; // [n] each (keys/vals are reused as tail flags / positions)
    void *temp;
    size_t temp_bytes;
    adsb_track_point *points;    // [n], frame order
    adsb_aircraft_record *aircraft; // [max_aircraft], ascending ICAO
    uint32_t max_aircraft;
    uint64_t *n_aircraft;        // device word
};
size_t track_sort_temp_bytes(size_t n);
hipError_t launch_track(hipStream_t s, const TrackArgs &a);

// test / measurement kernels
hipError_t launch_magnitudes(hipStream_t s, int sample_type, int mag_mode, const void *iq,
                             size_t n, uint16_t *out);
// i8: the biased squared magnitudes I^2+Q^2+72 as the nsq scan kernel's phase 1 packs them (test hook)
hipError_t launch_nsq_values(hipStream_t s, const void *iq, size_t n, uint16_t *out);
hipError_t launch_read_only(hipStream_t s, const void *buf, size_t bytes, uint32_t *sink, int shape); // shape 0..2 (kReadShapes)
constexpr int kReadShapes = 3;
hipError_t launch_synth(hipStream_t s, const adsb_synth_cfg &cfg, int sample_type,
                        uint32_t channel, uint64_t first, size_t n, void *iq);

} // namespace adsbk
