// adsb_kernels.h -- launch interface between the C-ABI layer (adsb_api.cpp) and the gfx950
// kernels (adsb_kernels.hip).  Internal; the public boundary is include/adsb_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/adsb_hip.h"

namespace adsbk {

// ---- tiling constants (see DESIGN.md "Data layout") ---------------------------------------
constexpr int kTile = 32768;    // offsets owned by one workgroup
constexpr int kThreads = 256;   // 4 waves; 4 workgroups per CU for i8 (LDS-bound)
constexpr int kRun = 64;        // consecutive offsets one lane slides over, per packed half
constexpr int kHalo = 256;      // >= 239 extra samples so PPM never leaves the tile; 16-aligned
constexpr int kMag = kTile + kHalo;
constexpr int kListCap = 512;   // candidate offsets staged per decode chunk
constexpr int kWindow = 240;    // 16 + 112*2  (reference src/adsb.rs:98)
constexpr uint32_t kNoBase = 0xFFFFFFFFu;

// One entry per tile, written unconditionally by the demod kernel.
struct Seg {
    uint32_t base;   // first temp slot of this tile, kNoBase if the slot store was full
    uint32_t cand;   // offsets that passed the preamble+DF17 gate (slots reserved)
    uint32_t valid;  // of those, frames that passed CRC / single-bit repair
    uint32_t pad;
};

// Device-resident result header (adsb_result_device).
struct Header {
    uint64_t n_out;        // frames in the final list (<= max_out)
    uint64_t total_found;  // frames that exist
    uint32_t flags;        // ADSB_FLAG_*
    uint32_t retry;        // internal: a needed tile lost its slots (slot store overflow)
    unsigned long long alloc; // slot allocator (reset by the scan kernel)
};

struct DemodArgs {
    const void *iq;            // channel 0, sample 0
    uint64_t n_samples;        // per channel
    uint64_t channel_stride;   // samples
    uint32_t tiles_per_channel;
    uint32_t tile_first;       // global tile id of blockIdx.x == 0
    Seg *seg;
    adsb_frame *slots;
    uint32_t cap_slots;
    Header *hdr;
};

struct CompactArgs {
    const Seg *seg;
    const adsb_frame *slots;
    uint32_t *out_start;       // [n_tiles + 1] exclusive scan of Seg::valid
    uint64_t *chan_counts;     // [n_channels]
    adsb_frame *out;
    uint32_t n_tiles;
    uint32_t tiles_per_channel;
    uint32_t n_channels;
    uint32_t max_out;
    uint32_t tile_first, tile_count; // gather range
    Header *hdr;
};

// mag_mode: how v_cvt_pk_u8_f32 rounds on this device (decided once per ctx by probe_cvt):
//   0: truncates as is; 1: truncates once MODE.fp_round(f32) is set to round-toward-zero;
//   2: rounds to nearest regardless -> subtract 0.5 first.
hipError_t probe_cvt(hipStream_t s, uint32_t *dev_scratch4, uint32_t host_out[4]);

hipError_t launch_demod(hipStream_t s, int sample_type, int mag_mode, const DemodArgs &a,
                        uint32_t n_tiles_launch);
hipError_t launch_scan(hipStream_t s, const CompactArgs &a);
hipError_t launch_gather(hipStream_t s, const CompactArgs &a);

// test / measurement kernels
hipError_t launch_magnitudes(hipStream_t s, int sample_type, int mag_mode, const void *iq,
                             size_t n, uint16_t *out);
hipError_t launch_read_only(hipStream_t s, const void *buf, size_t bytes, uint32_t *sink);
hipError_t launch_synth(hipStream_t s, const adsb_synth_cfg &cfg, int sample_type,
                        uint32_t channel, uint64_t first, size_t n, void *iq);

} // namespace adsbk
