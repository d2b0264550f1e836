// adsb_stream_kernel.h -- the streaming form of the i8 tile kernel (included by adsb_kernels.hip inside
// namespace adsbk; it uses that file's helpers: gate_phase, kSyn, tile_pos, tile_rsrc).
//
// Same function as demod_tiles<ADSB_SAMPLE_I8> (same closed form, same per-tile outputs: Seg, frame
// slots, group counters), different mapping to the machine.  demod_tiles is VALU-issue-bound: 20 of
// its ~54 issue cycles per 64 samples go into floor(sqrt(I^2+Q^2)) (v_dot4, v_add_f32, v_sqrt_f32,
// v_cvt_pk_u8_f32), and a workgroup alternates between waiting for HBM and computing.  Here:
//   * one persistent 1024-thread workgroup per CU (it owns 158 KB of the CU's 160 KB LDS) walks tiles
//     blockIdx.x, blockIdx.x + gridDim.x, ... (tiles of 28 672 offsets here);
//   * the magnitude is a table lookup: an i8 IQ sample is 16 bits, so floor(sqrt(I^2+Q^2)) for every
//     possible sample is a 64 KB byte table in LDS, indexed by the raw sample (index bits swizzled so
//     that receiver noise spreads over all LDS banks).  Exact by construction (the table is built
//     with integer arithmetic), no floating point anywhere on the path; the work moves from the VALU
//     to the otherwise idle LDS pipe;
//   * wave specialisation instead of co-resident workgroups: three roles work on three consecutive
//     tiles at the same time, through three LDS magnitude buffers, with ONE workgroup barrier per round:
//     5 "lookup waves" stream tile i+2 from HBM (each consumed 16-byte register quad is immediately
//     re-loaded, in place, with the next tile's data: ~58 KB per CU always in flight) and convert tile
//     i+1; 7 "gate waves" run the preamble/DF17 gate of tile i; 4 "decode waves" slice, CRC-check and
//     repair tile i-1's survivors into frame records staged in LDS and write tile i-2's out.
#pragma once

// ablation switches for measurements (tools/gpu): results are wrong when set
#ifndef ADSB_ABL_NOLOOKUP
#define ADSB_ABL_NOLOOKUP 0
#endif
#ifndef ADSB_ABL_NOGATE
#define ADSB_ABL_NOGATE 0
#endif

// Diagnostic build (-DADSB_STAMPS=1): workgroup 0 accumulates s_memtime deltas per segment of a round
// (gate wave 0: slots 0-7, lookup wave 0: slots 8-11, rounds in slot 15) into DemodArgs::stamps.
#ifndef ADSB_STAMPS
#define ADSB_STAMPS 0
#endif
#if ADSB_STAMPS
#define STAMP_DECL unsigned long long st_prev = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP_START()                                                                         \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
#define STAMP(k)                                                                              \
    do {                                                                                      \
        unsigned long long st_now;                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        st_acc[k] += st_now - st_prev;                                                        \
        st_prev = st_now;                                                                     \
    } while (0)
// per-wave busy time: from leaving a round barrier to arriving at the next
#define WSTAMP_DECL unsigned long long ws_prev = 0, ws_busy = 0
#define WSTAMP_LEAVE()                                                                        \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ws_prev)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
#define WSTAMP_ARRIVE()                                                                       \
    do {                                                                                      \
        unsigned long long ws_now;                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ws_now)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        ws_busy += ws_now - ws_prev;                                                          \
    } while (0)
#define WSTAMP_STORE(wave_global)                                                             \
    do {                                                                                      \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && p.stamps) p.stamps[16 + (wave_global)] = ws_busy; \
    } while (0)
#else
#define WSTAMP_DECL
#define WSTAMP_LEAVE() do { } while (0)
#define WSTAMP_ARRIVE() do { } while (0)
#define WSTAMP_STORE(w) do { } while (0)
#define STAMP_DECL
#define STAMP_START() do { } while (0)
#define STAMP(k) do { } while (0)
#endif

// Three roles in one 1024-thread workgroup, one barrier per round:
//   7 gate waves    preamble/DF17 gate of tile i (2 runs of 32 offsets per lane -> 28 672-offset tiles)
//   5 lookup waves  stream tile i+2 from HBM, convert tile i+1 into a magnitude buffer by table lookup
//   4 decode waves  PPM slice + CRC-24 + repair of tile i-1's survivors (16 groups: one pass for the usual
//                   ~15 per tile), frames staged in LDS
// Three magnitude buffers (28 928 B each) + the 64 KB table fill the CU's LDS; that is what sets the
// tile length.
constexpr int kSGateWaves = 7, kSLookupWaves = 5, kSDecodeWaves = 4;
constexpr int kSGateThreads = 64 * kSGateWaves;
constexpr int kSRun = 32;
constexpr int kSTile = 2 * kSGateThreads * kSRun; // 28 672 offsets per tile
constexpr int kSMag = kSTile + kHalo;             // 28 928 magnitudes per buffer
static_assert(kSTile == kStreamTile, "the host's tile length for this kernel");
constexpr int kSLookupThreads = 64 * kSLookupWaves;
constexpr int kSSweep = kSLookupThreads * 8;                  // samples per sweep of all lookup waves
constexpr int kSIters = (kSMag + kSSweep - 1) / kSSweep;      // 12 sweeps (the last one partial)
#define ADSB_STR2(x) #x
#define ADSB_STR(x) ADSB_STR2(x)
// when a sweep's quad is consumed, the kSIters - 1 loads issued after its own may still be in flight
#define ADSB_STREAM_VMCNT 11
static_assert(ADSB_STREAM_VMCNT == kSIters - 1, "wait count matches the number of sweeps");
constexpr int kSDecodeThreads = 64 * kSDecodeWaves;
constexpr int kSThreads = kSGateThreads + kSLookupThreads + kSDecodeThreads;
static_assert(kSThreads == 1024, "16 waves");
constexpr int kSDenseGroups = kSGateThreads / 16; // 16-lane groups of the gate waves (dense interlude)
constexpr int kLutBytes = 65536;

// Table index of a raw sample r = (Q << 8) | I (little-endian i8 pair).  The LDS bank of a byte
// address is bits 6:2; unswizzled those are I[6:2], which take ~13 values on receiver noise.  XOR-ing
// Q << 2 into bits 9:2 makes them I[6:2] ^ Q[4:0] (uniform on noise) and stays a bijection on 16 bits.
// On a packed dword of two samples: x = v ^ ((v >> 6) & 0x03FC03FC).
__host__ __device__ constexpr uint32_t lut_swizzle(uint32_t raw16)
{
    return raw16 ^ ((raw16 >> 6) & 0x03FCu);
}

// floor(sqrt(I^2+Q^2)) for all 65536 samples, integer arithmetic only (utils.rs:46-52 computes the same
// value through f64 sqrt + truncation; tests/test_gpu_parity.py compares the table with the oracle).
__global__ void build_lut_kernel(uint8_t *lut)
{
    const uint32_t raw = blockIdx.x * blockDim.x + threadIdx.x;
    if (raw >= (uint32_t)kLutBytes) return;
    const int i = (int)(int8_t)(raw & 0xFFu), q = (int)(int8_t)(raw >> 8);
    const uint32_t n = (uint32_t)(i * i + q * q); // <= 32768
    uint32_t r = 0;
    while ((r + 1u) * (r + 1u) <= n) ++r;
    lut[lut_swizzle(raw)] = (uint8_t)r;
}

hipError_t launch_build_lut(hipStream_t s, uint8_t *lut_dev)
{
    hipLaunchKernelGGL(build_lut_kernel, dim3(kLutBytes / 256), dim3(256), 0, s, lut_dev);
    return hipGetLastError();
}

#ifndef ADSB_STREAM_RECORD_WAVE
#define ADSB_STREAM_RECORD_WAVE 3 // which decode wave writes the per-tile records (the last one: its groups are the least often busy)
#endif
#ifndef ADSB_STREAM_GATE_GROUP
#define ADSB_STREAM_GATE_GROUP 1 // gate steps per wave-uniform test (see gate_phase; 1 measured fastest)
#endif

// Table indices of the 8 raw samples of one 16-byte load (both samples of a dword share the swizzle) ...
__device__ __forceinline__ void lookup_indices(u32x4 v, uint32_t x[4])
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) x[d] = w[d] ^ ((w[d] >> 6) & 0x03FC03FCu);
}
// ... the eight byte reads ...
__device__ __forceinline__ void lookup_reads(const unsigned char *lut, const uint32_t x[4], uint32_t m[8])
{
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        m[2 * d] = lut[x[d] & 0xFFFFu];
        m[2 * d + 1] = lut[x[d] >> 16];
    }
}
// In-place re-load of a raw quad whose eight table indices idx[] have been extracted: one asm statement
// with the quad and the indices as tied operands, so that (a) the quad's old value is dead afterwards --
// hipcc would otherwise re-derive the indices from it later, keep it alive across the load and move
// the load to another quad -- and (b) the load lands in the registers it is later consumed from.
__device__ __forceinline__ void reload_in_place(u32x4 &quad, uint32_t idx[8], uint32_t voff, u32x4 rsrc)
{
    asm volatile("buffer_load_dwordx4 %0, %9, %10, 0 offen nt"
                 : "+v"(quad), "+v"(idx[0]), "+v"(idx[1]), "+v"(idx[2]), "+v"(idx[3]), "+v"(idx[4]), "+v"(idx[5]),
                   "+v"(idx[6]), "+v"(idx[7])
                 : "v"(voff), "s"(rsrc));
}
__device__ __forceinline__ void lookup_indices8(u32x4 v, uint32_t idx[8])
{
    uint32_t x[4];
    lookup_indices(v, x);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        idx[2 * d] = x[d] & 0xFFFFu;
        idx[2 * d + 1] = x[d] >> 16;
    }
}
__device__ __forceinline__ void lookup_reads8(const unsigned char *lut, const uint32_t idx[8], uint32_t m[8])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = lut[idx[k]];
}
// ... and the packing of the eight magnitudes into two dwords.
__device__ __forceinline__ uint2 lookup_pack(const uint32_t m[8])
{
    const uint32_t p0 = m[0] | (m[1] << 16), p1 = m[2] | (m[3] << 16), p2 = m[4] | (m[5] << 16), p3 = m[6] | (m[7] << 16);
    return make_uint2(__builtin_amdgcn_perm(p1, p0, 0x06040200u), __builtin_amdgcn_perm(p3, p2, 0x06040200u));
}

// Table indices with the 4-instruction-per-dword form (shift, bitop3, and, shift): the intermediate is made
// opaque so that hipcc does not re-derive the low index from the raw dword with a fifth instruction.
// (The packing stays hipcc's 3 instructions per 4 bytes: the d16 byte loads that would deliver two
// magnitudes per register do not preserve the other half on this part -- SRAM ECC -- measured: wrong data.)
__device__ __forceinline__ void lookup_indices8_lean(u32x4 v, uint32_t idx[8])
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t x = w[d] ^ ((w[d] >> 6) & 0x03FC03FCu);
        asm("" : "+v"(x));
        idx[2 * d] = x & 0xFFFFu;
        idx[2 * d + 1] = x >> 16;
    }
}

// Round i of a workgroup (tiles t_k = tile_first + blockIdx.x + k * gridDim.x), all three at the same time:
//   gate waves     gate(i):    tile i's magnitudes (buffer i % 3) -> survivors in list[i % 2], count[i % 3]
//   lookup waves   pass i+2:   raw registers (tile i+1) -> magnitudes in buffer (i+1) % 3, registers re-loaded
//                              with tile i+2
//   decode waves   decode(i-1): survivors of tile i-1 (buffer (i-1) % 3, list[(i-1) % 2]) -> frame records in
//                              stage[(i-1) % 2], valid[(i-1) % 3]; then one of them writes the record of tile i-2
//                              (Seg, group counters, its staged frames -> global frame slots)
//   R(i)           ONE workgroup barrier ends the round.
// If gate(i) found more than kSparseCap survivors (pathological inputs) a "dense interlude"
// follows R(i): the gate waves decode tile i in place, straight to global memory, with the ordered
// bitmap compaction of demod_tiles; the other waves only take part in its barriers (every wave derives
// their number from the same LDS words).  Barriers before the first round: P1 (table in LDS), P2 (tile 0
// converted).  Rounds run i = 0 .. n_my + 1 so that the last tiles drain.
struct SLds {
    static constexpr int kOffLut = 0;
    static constexpr int kOffMag = kLutBytes;                      // 3 buffers of kSMag bytes
    static constexpr int kOffCand = kOffMag + 3 * kSMag;           // survivor bitmap of the tile being gated: kSTile / 32 words
    static constexpr int kOffList = kOffCand + kSTile / 8;         // 2 (round parity) x kListCap x u16
    static constexpr int kOffSyn = kOffList + 2 * kListCap * 2;
    static constexpr int kOffStage = kOffSyn + 112 * 4;            // 2 (round parity) x kSparseCap x 24 B frame records
    static constexpr int kOffRes = kOffStage + 2 * kSparseCap * 24; // dense interlude: per-group record staging
    static constexpr int kOffNib = kOffRes + kSDenseGroups * 24;   // 28 nibble positions x 16 syndrome sums (u32)
    static constexpr int kOffMisc = kOffNib + 28 * 16 * 4;
    static constexpr int kTotal = kOffMisc + 64;
    // misc words: [0..3] dense partial sums, [4] dense pool allocation, [5 + i%3] slot base of a dense tile i,
    // [8 + i%3] valid frames of tile i, [12 + i%3] gate survivors of tile i
    static constexpr int kAlloc = 4, kDenseBase = 5, kValid = 8, kCount = 12;
    __device__ static constexpr int mag_off(uint32_t k) { return kOffMag + (int)k * kSMag; }
};
static_assert(SLds::kTotal <= 160 * 1024, "LDS budget");
static_assert(kSMag % 16 == 0 && SLds::kOffStage % 8 == 0 && SLds::kOffCand % 16 == 0, "alignment of the LDS regions");

// barriers of a dense interlude for the waves that only take part in them
__device__ __forceinline__ void dense_interlude_mirror(const uint32_t *misc)
{
    __syncthreads(); // BD
    const uint32_t total = misc[0] + misc[1] + misc[2] + misc[3];
    __syncthreads(); // BA
    for (uint32_t chunk = 0; chunk < total; chunk += kListCap) {
        __syncthreads(); // BC1
        __syncthreads(); // BC2
    }
    __syncthreads(); // BX
}

// ---- role: lookup waves -----------------------------------------------------------------------------------
__device__ __forceinline__ void stream_lookup_role(const DemodArgs &p, unsigned char *smem, const uint32_t tg,
                                                   const uint32_t tile0, const uint32_t G, const uint32_t n_my)
{
    typedef SLds L;
    const unsigned char *lut = smem + L::kOffLut;
    const uint32_t *misc = reinterpret_cast<const uint32_t *>(smem + L::kOffMisc);

    // Pass r converts tile r-1 (if 1 <= r <= n_my) from the raw registers into magnitude buffer (r-1) % 3 and
    // re-loads every consumed register, in place, with the same piece of tile r (if r < n_my; otherwise
    // through an empty descriptor: zeros, no traffic).  Pass 0 and the passes after n_my convert zeros into a
    // buffer nobody reads: one uniform, branch-free body.  All loads are issued unconditionally: lanes past
    // the tile's samples are clipped by the descriptor and cost no traffic.
    //
    // The loads and their waits are inline asm.  With the builtin, every formulation tried made hipcc
    // either drain (s_waitcnt vmcnt(0)) or rotate the register quads through one v_mov each per pass,
    // because it cannot keep a load that is in flight across the loop's back edge in the register it
    // will be consumed from.  Tied operands ("+v") pin each quad; the wait is explicit: when sweep `it`
    // is consumed, the kSIters - 1 loads issued after its own (the rest of the previous pass, the start of
    // this one) may still be in flight -> s_waitcnt vmcnt(kSIters - 1).  These waves issue no other vector
    // memory instruction, so the count is exact.  (hipcc must not copy a quad between its load and its
    // wait: checked in the ISA.)
    u32x4 raw[kSIters];
#pragma unroll
    for (int it = 0; it < kSIters; ++it) raw[it] = u32x4{0u, 0u, 0u, 0u};

    STAMP_DECL;
    WSTAMP_DECL;
    STAMP_START();
    WSTAMP_LEAVE();
    for (uint32_t r = 0; r <= n_my + 3; ++r) {
        unsigned char *dst = smem + L::mag_off((r + 2) % 3); // buffer (r-1) % 3
        const bool more = r < n_my;
        const TilePos tpn = tile_pos<kSTile>(p, more ? tile0 + r * G : tile0);
        const u32x4 rn = tile_rsrc_words<2, kSMag>(p, tpn, more);
        // Software pipeline: the byte reads of sweep `it` are issued before the magnitudes of sweep
        // `it - 1` are packed and stored, so a wave always has 8-16 table reads in flight.
        uint32_t mprev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int it = 0; it < kSIters; ++it) {
            uint32_t idx[8], m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            asm volatile("s_waitcnt vmcnt(" ADSB_STR(ADSB_STREAM_VMCNT) ")" : "+v"(raw[it]));
            lookup_indices8_lean(raw[it], idx);
            reload_in_place(raw[it], idx, (uint32_t)it * (kSLookupThreads * 16) + tg * 16, rn);
            if (!ADSB_ABL_NOLOOKUP) lookup_reads8(lut, idx, m);
            if (it > 0 && !ADSB_ABL_NOLOOKUP)
                *reinterpret_cast<uint2 *>(dst + (uint32_t)(it - 1) * (kSLookupThreads * 8) + tg * 8) = lookup_pack(mprev);
#pragma unroll
            for (int k = 0; k < 8; ++k) mprev[k] = m[k];
            // keep the sweeps in program order (hipcc would otherwise hoist all index computations, and
            // with them the waits, to the top of the pass)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!ADSB_ABL_NOLOOKUP) { // the last sweep is partial: only lanes inside the buffer store
            const uint32_t s = (uint32_t)(kSIters - 1) * (kSLookupThreads * 8) + tg * 8;
            if (s < (uint32_t)kSMag) *reinterpret_cast<uint2 *>(dst + s) = lookup_pack(mprev);
        }
        STAMP(0); // conversion pass
        WSTAMP_ARRIVE();
        __syncthreads(); // P1 (r = 0), P2 (r = 1), R(r - 2)
        WSTAMP_LEAVE();
        STAMP(1); // wait for the other roles
        if (r >= 2 && r - 2 < n_my && misc[L::kCount + (r - 2) % 3] > (uint32_t)kSparseCap) dense_interlude_mirror(misc);
    }
    // nothing may be in flight into registers when the wave ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WSTAMP_STORE(threadIdx.x >> 6);
#if ADSB_STAMPS
    if (blockIdx.x == 0 && tg == 0 && p.stamps)
        for (int k = 0; k < 4; ++k) p.stamps[8 + k] = st_acc[k];
#endif
}

// ---- role: decode waves -----------------------------------------------------------------------------------
__device__ __forceinline__ void stream_decode_role(const DemodArgs &p, unsigned char *smem, const uint32_t tg,
                                                   const uint32_t tile0, const uint32_t G, const uint32_t n_my)
{
    typedef SLds L;
    const uint16_t *lists = reinterpret_cast<const uint16_t *>(smem + L::kOffList);
    const uint32_t *syn = reinterpret_cast<const uint32_t *>(smem + L::kOffSyn);
    const uint32_t *nib = reinterpret_cast<const uint32_t *>(smem + L::kOffNib);
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + L::kOffMisc);
    const uint32_t lane = tg & 63;
    const uint32_t dwave = __builtin_amdgcn_readfirstlane(tg >> 6);
    const uint32_t g = tg >> 4, l = tg & 15; // 16-lane groups: one candidate each, one lane per frame byte

    __syncthreads(); // P1
    __syncthreads(); // P2
    STAMP_DECL;
    WSTAMP_DECL;
    STAMP_START();
    WSTAMP_LEAVE();

    // Record of tile j (local index), written by one decode wave two rounds after the tile's gate: the
    // frames the decode waves staged in LDS go to the tile's slots (its own kQuota slots, or a block of the
    // shared pool), then Seg + group counters, and the tile's counters are re-armed for tile j + 3.  A dense
    // tile's frames were stored by its interlude, which also left its slot base in LDS.
    auto record = [&](const uint32_t j) {
        if (dwave != ADSB_STREAM_RECORD_WAVE) return;
        const uint32_t tile = tile0 + j * G, k3 = j % 3;
        const uint32_t total = misc[L::kCount + k3];
        const bool dense = total > (uint32_t)kSparseCap;
        uint32_t base = tile * kQuota;
        if (dense) {
            base = misc[L::kDenseBase + k3];
        } else {
            if (total > kQuota) {
                uint32_t b = 0;
                if (lane == 0) {
                    const unsigned long long b64 = atomicAdd(&p.hdr->alloc, (unsigned long long)total);
                    b = (b64 + total <= (unsigned long long)p.cap_slots) ? p.pool_first + (uint32_t)b64 : kNoBase;
                }
                base = __builtin_amdgcn_readfirstlane(b);
            }
            if (base != kNoBase) {
                const uint32_t *src = reinterpret_cast<const uint32_t *>(smem + L::kOffStage + (j & 1u) * (kSparseCap * 24));
                uint32_t *dst = reinterpret_cast<uint32_t *>(p.slots + (size_t)base);
                for (uint32_t k = lane; k < total * 6; k += 64) dst[k] = src[k];
            }
        }
        if (lane == 0) {
            Seg e;
            e.base = base;
            e.cand = total;
            e.valid = misc[L::kValid + k3];
            e.decoded = 1; // this kernel slices and CRC-checks in place: the decode kernel skips its tiles
            p.seg[tile] = e; // (finish_candidates, which runs after every tile kernel, sums the group counters)
            misc[L::kValid + k3] = 0;
            misc[L::kCount + k3] = 0;
        }
    };

    for (uint32_t i = 0; i <= n_my + 1; ++i) {
        if (i >= 1 && i <= n_my) {
            const uint32_t j = i - 1; // tile index (local) being decoded
            const uint32_t total = misc[L::kCount + j % 3];
            if (total && total <= (uint32_t)kSparseCap) { // (a dense tile was decoded in its interlude)
                const uint16_t *list = lists + (j & 1u) * kListCap;
                const unsigned char *mag = smem + L::mag_off(j % 3);
                unsigned char *stage = smem + L::kOffStage + (j & 1u) * (kSparseCap * 24);
                const TilePos tpd = tile_pos<kSTile>(p, tile0 + j * G);
                // The list is unordered; a candidate's slot is its rank: the number of listed offsets below
                // its own.  total <= kSparseCap = 64 = 16 lanes x 4.
                for (uint32_t c0 = 0; c0 < total; c0 += kSDecodeThreads / 16) {
                    if (c0 + 4 * dwave >= total) break; // none of this wave's four groups has a candidate
                    const uint32_t ci = c0 + g;
                    const bool have = ci < total; // uniform within the 16-lane group
                    const uint32_t off = have ? list[ci] : 0u;
                    uint32_t below = 0;
#pragma unroll
                    for (int k = 0; k < kSparseCap / 16; ++k) {
                        const uint32_t q = l + 16 * k;
                        const uint32_t e = q < total ? (uint32_t)list[q] : 0xFFFFFFFFu;
                        below += e < off ? 1u : 0u;
                    }
                    below = row16_sum(below);
                    const bool valid = decode_candidate<ADSB_SAMPLE_I8, true>(mag, syn, stage + (have ? below : 0u) * 24, have, off,
                                                                              tpd.sample0 + p.offset_base, l, lane, nib);
                    if (valid && l == 0) atomicAdd(&misc[L::kValid + j % 3], 1u);
                }
            }
        }
        STAMP(0); // decode
        if (i >= 2) record(i - 2);
        STAMP(2); // record
        WSTAMP_ARRIVE();
        __syncthreads(); // R(i)
        WSTAMP_LEAVE();
        STAMP(1);
        if (i < n_my && misc[L::kCount + i % 3] > (uint32_t)kSparseCap) dense_interlude_mirror(misc);
    }
#if ADSB_STAMPS
    if (blockIdx.x == 0 && tg == 0 && p.stamps)
        for (int k = 0; k < 3; ++k) p.stamps[12 + k] = st_acc[k];
#endif
    WSTAMP_STORE(threadIdx.x >> 6);
}

// ---- role: gate waves -------------------------------------------------------------------------------------
__device__ __forceinline__ void stream_gate_role(const DemodArgs &p, unsigned char *smem, const uint32_t tid,
                                                 const uint32_t tile0, const uint32_t G, const uint32_t n_my)
{
    typedef SLds L;
    constexpr int ST = ADSB_SAMPLE_I8;
    uint32_t *cand = reinterpret_cast<uint32_t *>(smem + L::kOffCand);
    uint16_t *lists = reinterpret_cast<uint16_t *>(smem + L::kOffList);
    const uint32_t *syn = reinterpret_cast<const uint32_t *>(smem + L::kOffSyn);
    unsigned char *res = smem + L::kOffRes;
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + L::kOffMisc);
    const uint32_t lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t g = tid >> 4, l = tid & 15; // dense interlude: kSDenseGroups groups of 16 lanes

    __syncthreads(); // P1
    __syncthreads(); // P2: tile 0's magnitudes are in buffer 0
    STAMP_DECL;
    WSTAMP_DECL;
    STAMP_START();
    WSTAMP_LEAVE();

    for (uint32_t i = 0; i <= n_my + 1; ++i) {
        const bool live = i < n_my;
        const uint32_t tile = tile0 + (live ? i : 0u) * G;
        const TilePos tp = tile_pos<kSTile>(p, tile);
        const unsigned char *mag = smem + L::mag_off(i % 3);
        uint16_t *list = lists + (i & 1u) * kListCap;

        if (live && !ADSB_ABL_NOGATE)
            gate_phase<ST, ADSB_STREAM_GATE_GROUP, kSRun, kSGateThreads>(mag, cand, list, &misc[L::kCount + i % 3], tid, tp.n_valid);
        STAMP(0); // gate
        WSTAMP_ARRIVE();
        __syncthreads(); // R(i)
        WSTAMP_LEAVE();
        STAMP(1);
        if (!live) continue;

        uint32_t total = misc[L::kCount + i % 3];
        if (total > (uint32_t)kSparseCap) {
            // Dense interlude: ordered compaction of the bitmap (kSTile / 32 words; offset 32 w + b is bit b of
            // word w) by prefix sums over the first four gate waves, four words per lane; decode in place by
            // the gate waves, straight to global memory.
            u32x4 cw = {0, 0, 0, 0};
            uint32_t cnt = 0, incl = 0;
            if (wave < 4) {
                if (4 * tid < (uint32_t)(kSTile / 32)) cw = reinterpret_cast<const u32x4 *>(cand)[tid];
                cnt = __builtin_popcount(cw.x) + __builtin_popcount(cw.y) + __builtin_popcount(cw.z) +
                      __builtin_popcount(cw.w);
                incl = cnt;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    uint32_t t = __shfl_up(incl, d, 64);
                    if ((int)lane >= d) incl += t;
                }
                if (lane == 63) misc[wave] = incl;
            }
            __syncthreads(); // BD
            uint32_t wbase = 0;
            total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t t = misc[w];
                wbase += (w < (int)wave) ? t : 0u;
                total += t;
            }
            const uint32_t my_first = wbase + incl - cnt;
            if (tid == 0) {
                const unsigned long long b64 = atomicAdd(&p.hdr->alloc, (unsigned long long)total);
                misc[L::kAlloc] = (b64 + total <= (unsigned long long)p.cap_slots) ? p.pool_first + (uint32_t)b64 : kNoBase;
            }
            __syncthreads(); // BA
            const uint32_t base_slot = misc[L::kAlloc];
            if (tid == 0) misc[L::kDenseBase + i % 3] = base_slot;
            for (uint32_t chunk = 0; chunk < total; chunk += kListCap) {
                if (cnt) {
                    uint32_t idx = my_first;
                    const uint32_t words[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        uint32_t bits = words[k];
                        while (bits) {
                            const uint32_t b = __builtin_ctz(bits);
                            bits &= bits - 1;
                            if (idx >= chunk && idx < chunk + kListCap)
                                list[idx - chunk] = (uint16_t)((4 * tid + k) * 32 + b);
                            ++idx;
                        }
                    }
                }
                __syncthreads(); // BC1
                const uint32_t ncl = (total - chunk) < (uint32_t)kListCap ? (total - chunk) : (uint32_t)kListCap;
                for (uint32_t r = 0; r < ncl; r += kSDenseGroups) {
                    if (r + 4 * wave >= ncl) break;
                    const uint32_t ci = r + g;
                    const bool have = ci < ncl;
                    unsigned char *rec = res + g * 24;
                    const bool valid = decode_candidate<ADSB_SAMPLE_I8>(mag, syn, rec, have, have ? list[ci] : 0u, tp.sample0 + p.offset_base, l, lane);
                    if (valid && l == 0) atomicAdd(&misc[L::kValid + i % 3], 1u);
                    if (have && base_slot != kNoBase && l < 6) {
                        uint32_t *dst = reinterpret_cast<uint32_t *>(p.slots + (size_t)base_slot + chunk + ci);
                        dst[l] = reinterpret_cast<const uint32_t *>(rec)[l];
                    }
                }
                __syncthreads(); // BC2
            }
            // the record must see the compacted total, not the (saturating) survivor count of the sparse path
            if (tid == 0) misc[L::kCount + i % 3] = total;
            __syncthreads(); // BX: in-place decode done before the buffer, the bitmap and the list are reused
        }
    }
#if ADSB_STAMPS
    if (blockIdx.x == 0 && tid == 0 && p.stamps) {
        for (int k = 0; k < 8; ++k) p.stamps[k] = st_acc[k];
        p.stamps[15] = n_my;
    }
#endif
    WSTAMP_STORE(threadIdx.x >> 6);
}

__global__ __launch_bounds__(kSThreads, 1) void demod_stream_i8(DemodArgs p)
{
    typedef SLds L;
    __shared__ __attribute__((aligned(16))) unsigned char smem[L::kTotal];
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- prologue: table and constants into LDS ---------------------------------------------------
    {
        unsigned char *lut = smem + L::kOffLut;
        uint32_t *syn = reinterpret_cast<uint32_t *>(smem + L::kOffSyn);
        uint32_t *misc = reinterpret_cast<uint32_t *>(smem + L::kOffMisc);
        for (uint32_t k = tid; k < (uint32_t)(kLutBytes / 16); k += kSThreads)
            reinterpret_cast<u32x4 *>(lut)[k] = reinterpret_cast<const u32x4 *>(p.lut)[k];
        if (tid < 112) syn[tid] = kSyn.v[tid];
        if (tid < 28 * 16) { // nibble sums of the syndrome table (decode_candidate<.., true>)
            uint32_t e = 0;
            for (int b = 0; b < 4; ++b) e ^= ((tid >> (3 - b)) & 1u) ? kSyn.v[4 * (tid >> 4) + b] : 0u;
            reinterpret_cast<uint32_t *>(smem + L::kOffNib)[tid] = e;
        }
        if (tid < 16) misc[tid] = 0;
        if (blockIdx.x == 0 && tid == 0) {
            p.hdr->retry = 0;
            if (p.count_groups) { p.hdr->flags = 0; if (p.hdr_pub) p.hdr_pub[2] = 0; } // as demod_tiles
        }
    }
    const uint32_t G = gridDim.x;
    const uint32_t n_my = (p.tile_count - blockIdx.x + G - 1) / G; // >= 1: the grid is at most tile_count
    const uint32_t tile0 = p.tile_first + blockIdx.x;

    // Three roles, three loops (scalar branches: whole waves).  All execute the same sequence of barriers.
    // (s_setprio for the gate waves measured no effect: per-wave stamps show the second gate wave of a SIMD
    // taking ~900 cycles longer per round than the first; raising its priority only swaps the two -- the SIMD's
    // issue slots are what is exhausted.)
    if (wave < (uint32_t)kSGateWaves) stream_gate_role(p, smem, tid, tile0, G, n_my);
    else if (wave < (uint32_t)(kSGateWaves + kSLookupWaves)) stream_lookup_role(p, smem, tid - kSGateThreads, tile0, G, n_my);
    else stream_decode_role(p, smem, tid - kSGateThreads - kSLookupThreads, tile0, G, n_my);
}

hipError_t launch_demod_stream(hipStream_t s, const DemodArgs &a, uint32_t n_cu, hipEvent_t e0, hipEvent_t e1)
{
    if (a.tile_count == 0) return hipSuccess;
    const uint32_t grid = a.tile_count < n_cu ? a.tile_count : n_cu;
    hipExtLaunchKernelGGL(demod_stream_i8, dim3(grid), dim3(kSThreads), 0, s, e0, e1, 0, a);
    return hipGetLastError();
}
