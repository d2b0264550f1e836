// adsb_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for the air_rs IQ -> packet path.
//
// What the reference computes per received buffer (src/adsb.rs:95-116), in closed form:
//   m[k]   = floor(sqrt(I^2+Q^2))                                  (src/utils.rs:46-52)
//   gate(i)= min m[i+{0,2,7,9}] >= max m[i+{1,3,4,5,6,8,10..15}]   (src/adsb/demod.rs:23-36)
//            && min m[i+16+{0,3,5,7,8}] >= max m[i+16+{1,2,4,6,9}] (src/adsb/demod.rs:45-54)
//   bit_k  = m[i+16+2k] > m[i+17+2k], k = 0..111, MSB first        (demod.rs:92-131,180-201)
//   s      = CRC24(bytes[0..11]) ^ bytes[11..14]                   (src/adsb/crc.rs:10-40)
//   emit at every i in [0, N-240) where gate(i) and (s == 0 or s is the syndrome of one of the
//   88 data bits, which is then flipped)                            (src/adsb/crc.rs:49-65)
// Every offset is independent (the `_i += 240` at adsb.rs:113 has no effect).
//
// Mapping to the machine (no MFMA: an elementwise / stencil path; HBM-bound by design, VALU-issue-bound as measured --
// DESIGN.md section 5).  A launch is two kernels:
//   demod_tiles (the SCAN: every IQ byte is read once)
//   * one workgroup = one tile of kTile offsets; the tile's raw IQ is read coalesced, 16 B per lane, all loads in flight
//     before the first use, through a bounds-checked buffer descriptor (tails read as zero); which tile a workgroup takes is
//     XCD-aware (tile_of_workgroup: eight contiguous ranges of tiles, one per XCD, so a tile's halo is an L2 hit);
//   * phase 1: magnitudes in registers (v_dot4_i32_i8 -> v_sqrt_f32 -> v_cvt_pk_u8_f32), parked in LDS as u8 (i8 input)
//     or u16 (CS16);
//   * phase 2: the gate runs "transposed": each lane slides along its own run of kRun consecutive offsets, two runs
//     packed in the halves of one VGPR so every min/max is one packed instruction for two offsets (3-input
//     v_pk_maximum3_f16 / v_pk_minimum3_f16 on the magnitudes as f16 bit patterns); running maxima / minima are shared
//     between neighbouring offsets: 8 VALU per step of two offsets; the DF17 part only where some lane of the wave passes
//     the preamble part;
//   * phase 3: the few survivors (LDS bitmap -> unordered list; prefix sums when dense) are sliced by 16-lane groups, one
//     lane per frame byte, straight from the LDS magnitudes; offset + 14 bytes go to the survivor's slot (every tile owns
//     kQuota slots; more come from a shared pool).  No CRC here.
//   finish_order (latency-bound, ~10 us): one LANE per survivor -- CRC-24 by byte table, single-bit repair by binary
//   search of the 88 sorted syndromes, rank inside the tile by DPP rotations, the tile's place from an exchange of
//   per-workgroup counts inside the kernel -- writes the frames in ascending (channel, offset) order, the order the
//   reference's mpsc channel would deliver them in.
// Buffers of at most 32 tiles (the reference's own 20 000-sample buffers) take ONE dispatch (demod_small: the tile body
// per workgroup, the last workgroup to arrive runs the finishing block and writes into pinned host memory).
// The product library carries one i8 scan (kScanRoot) and CS16's; the kernels measured against it (kScanNsq, kScanReg,
// kScanCode, kScanSieve: bit-exact, none faster) are compiled with -DADSB_AB_KERNELS=1 only.
#include <hip/hip_ext.h>
#include <utility>
#include <cstdlib>

#include "adsb_kernels.h"
#include "adsb_synth.h"

namespace adsbk {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// [phase:2 gate: min/max (helpers)]
// ---- small device helpers -------------------------------------------------------------------
__device__ __forceinline__ uint32_t pkmin(uint32_t a, uint32_t b)
{
    u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pkmax(uint32_t a, uint32_t b)
{
    u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
// Three-input packed max/min.  F16 = every value is below 0x7C00, i.e. a non-negative finite f16 BIT PATTERN
// (denormal or normal), whose numeric order is its integer order: gfx950's v_pk_maximum3_f16 /
// v_pk_minimum3_f16 then reduce three packed pairs in one instruction.  True for i8 input always (magnitudes
// <= 181) and for a CS16 tile whose largest magnitude is below 31744 (checked per tile in phase 1); otherwise
// (CS16 magnitudes reach 46340: infinities, NaNs, negative patterns) two integer v_pk_max_u16 / v_pk_min_u16.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <bool F16> __device__ __forceinline__ uint32_t pkmax3(uint32_t a, uint32_t b, uint32_t c)
{
    if (F16) {
        f16x2 r = __builtin_elementwise_maximum(
            __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)),
            __builtin_bit_cast(f16x2, c));
        return __builtin_bit_cast(uint32_t, r);
    }
    return pkmax(pkmax(a, b), c);
}
template <bool F16> __device__ __forceinline__ uint32_t pkmin3(uint32_t a, uint32_t b, uint32_t c)
{
    if (F16) {
        f16x2 r = __builtin_elementwise_minimum(
            __builtin_elementwise_minimum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)),
            __builtin_bit_cast(f16x2, c));
        return __builtin_bit_cast(uint32_t, r);
    }
    return pkmin(pkmin(a, b), c);
}

// [phase:1 magnitude (helpers)]  (markers read by tools/isa_slots.py)
// Eight I^2+Q^2 sums (one 16-byte load = 8 samples) as VOP3P v_dot4_i32_i8 with the accumulator in
// an SGPR.  Why asm: for the builtin hipcc picks the VOP2 v_dot4c form, which needs a v_mov per
// call to preload the constant accumulator.  gfx950 needs 3 wait states between a DOT writing a
// VGPR and a different VALU reading it; hipcc pads nothing for asm, so the block ends in s_nop 2
// (the eight dots themselves may issue back to back).
__device__ __forceinline__ void dot4x8_sacc(u32x4 v, int c, int n[8])
{
    const uint32_t a0 = v.x & 0xFFFFu, a1 = v.x & 0xFFFF0000u, a2 = v.y & 0xFFFFu, a3 = v.y & 0xFFFF0000u,
                   a4 = v.z & 0xFFFFu, a5 = v.z & 0xFFFF0000u, a6 = v.w & 0xFFFFu, a7 = v.w & 0xFFFF0000u;
    asm("v_dot4_i32_i8 %0, %8, %12, %20\n\t"
        "v_dot4_i32_i8 %1, %8, %13, %20\n\t"
        "v_dot4_i32_i8 %2, %9, %14, %20\n\t"
        "v_dot4_i32_i8 %3, %9, %15, %20\n\t"
        "v_dot4_i32_i8 %4, %10, %16, %20\n\t"
        "v_dot4_i32_i8 %5, %10, %17, %20\n\t"
        "v_dot4_i32_i8 %6, %11, %18, %20\n\t"
        "v_dot4_i32_i8 %7, %11, %19, %20\n\t"
        "s_nop 2"
        : "=&v"(n[0]), "=&v"(n[1]), "=&v"(n[2]), "=&v"(n[3]), "=&v"(n[4]), "=&v"(n[5]), "=&v"(n[6]), "=&v"(n[7])
        : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5),
          "v"(a6), "v"(a7), "s"(c));
}

// CRC-24 syndrome table (used by count_candidate only -- tiles that lost their slots are counted in place, a cold
// path; finish_order has its own byte table and sorted syndromes): kSyn[j] = x^(111-j) mod 0x1FFF409, j = 0..111 (bit j
// MSB-first of the 112-bit frame).  XOR over the set bits of a frame is CRC24(data) ^ crc_field; for j < 88 it is
// also the syndrome of a single error in data bit j (the 88 values are distinct and non-zero,
// which is why the reference's ordered brute force, crc.rs:49-65, has at most one match).
struct SynTable {
    uint32_t v[112];
};
constexpr SynTable make_syn()
{
    SynTable t{};
    uint32_t r = 1; // x^0
    for (int e = 0; e < 112; ++e) {
        t.v[111 - e] = r;
        r <<= 1;
        if (r & 0x1000000u) r ^= 0x1FFF409u;
    }
    return t;
}
__constant__ SynTable kSyn = make_syn();

// ---- magnitude ------------------------------------------------------------------------------
// floor(sqrt(n)) for n = I^2+Q^2 <= 32768 (i8 input), exact:
// sqrt(n + 0.5) is at least 1.3e-3 away from every integer for n <= 32768, far more than the
// 1-ulp error of v_sqrt_f32, so truncating it gives floor(sqrt(n)) -- the value the reference
// gets from f64 sqrt + `as u32` (utils.rs:48).  n + 0.5 is formed without an int->float
// convert: the dot product accumulates onto 0x4B000000 (2^23 as float bits), so the integer
// result reinterpreted as float is 2^23 + n, and one subtraction of (2^23 - 0.5) is exact.
template <int MAGMODE> __device__ __forceinline__ float mag_root_i8(int n_plus_2p23)
{
    float f = __builtin_bit_cast(float, n_plus_2p23) - 8388607.5f;
    float r = __builtin_amdgcn_sqrtf(f);
    if (MAGMODE == 2) r -= 0.5f; // converter rounds to nearest: land in (k-0.5, k+0.5)
    return r;
}

// 8 consecutive i8 IQ samples (16 bytes) -> 8 magnitudes packed as bytes in two dwords.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MAGMODE>
__device__ __forceinline__ void mags8_i8(u32x4 v, uint32_t &lo, uint32_t &hi)
{
    int n[8];
    dot4x8_sacc(v, 0x4B000000, n);
    // n + 0.5 for two samples per instruction (v_pk_add_f32; see mag_root_i8 for the constant)
    float f[8];
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        f32x2 t = {__builtin_bit_cast(float, n[k]), __builtin_bit_cast(float, n[k + 1])};
        t = t - (f32x2){8388607.5f, 8388607.5f};
        f[k] = t.x;
        f[k + 1] = t.y;
    }
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        r[k] = __builtin_amdgcn_sqrtf(f[k]);
        if (MAGMODE == 2) r[k] -= 0.5f; // converter rounds to nearest: land in (k-0.5, k+0.5)
    }
    lo = __builtin_amdgcn_cvt_pk_u8_f32(r[0], 0, 0u);
    lo = __builtin_amdgcn_cvt_pk_u8_f32(r[1], 1, lo);
    lo = __builtin_amdgcn_cvt_pk_u8_f32(r[2], 2, lo);
    lo = __builtin_amdgcn_cvt_pk_u8_f32(r[3], 3, lo);
    hi = __builtin_amdgcn_cvt_pk_u8_f32(r[4], 0, 0u);
    hi = __builtin_amdgcn_cvt_pk_u8_f32(r[5], 1, hi);
    hi = __builtin_amdgcn_cvt_pk_u8_f32(r[6], 2, hi);
    hi = __builtin_amdgcn_cvt_pk_u8_f32(r[7], 3, hi);
}

// floor(sqrt(I^2+Q^2)) for one i16 sample (n <= 2^31): n by one v_dot2_i32_i16, a float estimate rounded to
// the nearest integer, and one exact integer correction.
// Error budget of sqrtf((float)n) at s = sqrt(n) <= 46341: conversion to float 2^-24 relative (2^-25 after
// the root), v_sqrt_f32 one ulp (2^-23): |e| <= 46341 * 1.5e-7 < 0.007.  v_cvt_rpi_i32_f32 is floor(x + 0.5),
// so the estimate is floor(s) or floor(s) + 1 (either of them when the fraction of s is within 0.007 of one
// half), and r * r > n tells which (r <= 46341: r * r < 2^32).
// gfx950 does not interlock a transcendental result against the next VALU instruction (one wait state is
// required) and hipcc pads nothing inside or around inline asm: the s_nop between v_sqrt_f32 and its reader
// below is load-bearing.
__device__ __forceinline__ uint32_t mag_i16_fix(uint32_t r, uint32_t n)
{
    return r - ((__umul24(r, r) > n) ? 1u : 0u); // r < 2^24: the 24-bit multiply is exact and full rate
}
__device__ __forceinline__ uint32_t mag_i16(uint32_t iq)
{
    uint32_t n, r; // n = 2^31 for (-32768, -32768): the i32 result wraps to the right bits
    // (VOP3P form with the inline constant 0 as accumulator: for the builtin hipcc picks v_dot2c, which needs
    // a v_mov per call to preload it; gfx950 wants 3 wait states between a DOT and a VALU reading it)
    asm("v_dot2_i32_i16 %0, %2, %2, 0\n\t"
        "s_nop 2\n\t"
        "v_cvt_f32_u32 %1, %0\n\t"
        "v_sqrt_f32 %1, %1\n\t"
        "s_nop 0\n\t"
        "v_cvt_rpi_i32_f32 %1, %1"
        : "=&v"(n), "=&v"(r)
        : "v"(iq));
    return mag_i16_fix(r, n);
}
// four samples (one 16-byte load) -> two words of packed u16 magnitudes; each group of four like instructions
// issues back to back, which also covers the wait states between a group and the next
__device__ __forceinline__ void mags4_i16(u32x4 v, uint32_t &lo, uint32_t &hi)
{
    uint32_t n0, n1, n2, n3, r0, r1, r2, r3;
    asm("v_dot2_i32_i16 %0, %8, %8, 0\n\t"
        "v_dot2_i32_i16 %1, %9, %9, 0\n\t"
        "v_dot2_i32_i16 %2, %10, %10, 0\n\t"
        "v_dot2_i32_i16 %3, %11, %11, 0\n\t"
        "s_nop 2\n\t"
        "v_cvt_f32_u32 %4, %0\n\t"
        "v_cvt_f32_u32 %5, %1\n\t"
        "v_cvt_f32_u32 %6, %2\n\t"
        "v_cvt_f32_u32 %7, %3\n\t"
        "v_sqrt_f32 %4, %4\n\t"
        "v_sqrt_f32 %5, %5\n\t"
        "v_sqrt_f32 %6, %6\n\t"
        "v_sqrt_f32 %7, %7\n\t"
        "v_cvt_rpi_i32_f32 %4, %4\n\t"
        "v_cvt_rpi_i32_f32 %5, %5\n\t"
        "v_cvt_rpi_i32_f32 %6, %6\n\t"
        "s_nop 0\n\t"
        "v_cvt_rpi_i32_f32 %7, %7\n\t"
        "s_nop 0"
        : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    // low halves of two registers into one (one v_perm; the magnitudes are < 2^16)
    lo = __builtin_amdgcn_perm(mag_i16_fix(r1, n1), mag_i16_fix(r0, n0), 0x05040100u);
    hi = __builtin_amdgcn_perm(mag_i16_fix(r3, n3), mag_i16_fix(r2, n2), 0x05040100u);
}

// [phase:end]
// ---- probe: how does v_cvt_pk_u8_f32 round here? --------------------------------------------
__global__ void probe_cvt_kernel(uint32_t *out)
{
    if (threadIdx.x != 0) return;
    float a = 0.75f, b = 2.5f, c = 180.9986f;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
    uint32_t r0 = __builtin_amdgcn_cvt_pk_u8_f32(a, 0, 0u);
    r0 = __builtin_amdgcn_cvt_pk_u8_f32(b, 1, r0);
    r0 = __builtin_amdgcn_cvt_pk_u8_f32(c, 2, r0);
    out[0] = r0;
    // MODE.fp_round: bits [1:0] = f32 rounding; 3 = toward zero
    __builtin_amdgcn_s_setreg((1 | (0 << 6) | ((2 - 1) << 11)), 3);
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
    uint32_t r1 = __builtin_amdgcn_cvt_pk_u8_f32(a, 0, 0u);
    r1 = __builtin_amdgcn_cvt_pk_u8_f32(b, 1, r1);
    r1 = __builtin_amdgcn_cvt_pk_u8_f32(c, 2, r1);
    out[1] = r1;
    __builtin_amdgcn_s_setreg((1 | (0 << 6) | ((2 - 1) << 11)), 0);
    out[2] = 0xC0DEu;
    out[3] = 0;
}

hipError_t probe_cvt(hipStream_t s, uint32_t *dev_scratch4, uint32_t host_out[4])
{
    hipLaunchKernelGGL(probe_cvt_kernel, dim3(1), dim3(64), 0, s, dev_scratch4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(host_out, dev_scratch4, 16, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(s);
}

// ---- the fused tile kernel ------------------------------------------------------------------
template <int ST> struct MagT;
template <> struct MagT<ADSB_SAMPLE_I8> { typedef uint8_t type; };
template <> struct MagT<ADSB_SAMPLE_I16> { typedef uint16_t type; };

// ---- the nsq image (i8, kScanNsq): what phase 1 leaves in LDS for the gate and the slicer ------------------------
// One dword per PAIR of samples half a tile apart: logical dword q in [0, kNsqLog) holds v(q) in its low half and
// v(q + kNsqHalf) in its high half, v(k) = I_k^2 + Q_k^2 + 72 (<= 32840).  That pair is exactly what lane L's
// two runs (offsets 32 L + o and kNsqHalf + 32 L + o) need in one VGPR at step o: the gate reads it as it is, no
// unpacking.  Dwords kNsqHalf .. kNsqHalf+255 repeat samples as low halves that dwords 0..255 hold as high halves
// (the halo of run A's last lanes).  Physical dword = q + 4 (q >> 6): four pad dwords per 64 put the 16-byte reads
// of a ds_read_b128 lane group (lane L starts at 32 L) on sixteen different slots of the 64 banks.
constexpr int kNsqBias = 72;                 // 9 * 8: (v >> 3) = (n >> 3) + 9 exactly
constexpr int kNsqHalf = kTile / 2;          // 8192
constexpr int kNsqLog = kNsqHalf + kHalo;    // logical dwords
#ifndef ADSB_NSQ_PAD_SHIFT
#define ADSB_NSQ_PAD_SHIFT 6 // four pad dwords per 2^6 logical dwords (5: per 32 -- also conflict-free for the stores, 6 % more LDS)
#endif
constexpr int kNsqPadShift = ADSB_NSQ_PAD_SHIFT;
__host__ __device__ constexpr uint32_t nsq_phys(uint32_t q) { return q + 4u * (q >> kNsqPadShift); }
constexpr int kNsqPhys = (int)nsq_phys(kNsqLog);  // 8976 dwords = 35904 bytes
static_assert(kNsqHalf % 64 == 0 && kHalo % 64 == 0 && kRun == 32 && (kNsqPadShift == 5 || kNsqPadShift == 6),
              "nsq image: pads every 32 or 64 dwords, runs of 32");

// The u8 magnitude image of the i8 root scan, optionally padded (-DADSB_MAG_PAD=1): 16 bytes after every 256.  The gate's
// ds_read_b128 has lane L start at byte 32 L: the sixteen lanes of a read group span 512 bytes, two passes over the 64
// banks (lanes L and L + 8 on the same four banks: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.27 for the kernel).  With
// the pad lanes 8-15 sit four banks further: conflict-free.  Physical byte = b + 16 (b >> 8).
#ifndef ADSB_MAG_PAD
#define ADSB_MAG_PAD 0
#endif
template <int ST, int SCAN> struct MagPad {
    static constexpr bool on = ADSB_MAG_PAD != 0 && ST == ADSB_SAMPLE_I8 && SCAN == kScanRoot;
};
template <bool PAD> __host__ __device__ constexpr uint32_t mag_phys(uint32_t b) { return PAD ? b + 16u * (b >> 8) : b; }

template <int ST, int SCAN = kScanRoot> struct Lds {
    typedef typename MagT<ST>::type mag_t;
    static constexpr bool kNsq = ST == ADSB_SAMPLE_I8 && SCAN == kScanNsq;
    static constexpr int kMagBytes = kNsq ? kNsqPhys * 4
                                          : (int)((mag_phys<MagPad<ST, SCAN>::on>((uint32_t)TileCfg<ST>::kMagT * (uint32_t)sizeof(mag_t)) + 15u) & ~15u);
    static constexpr int kOffCand = kMagBytes;                 // one word per run of 32 offsets: survivor bitmap
    static constexpr int kOffList = kOffCand + 2 * kThreads * 4 * ((TileCfg<ST>::kRunT + 31) / 32); // kListCap x u16
    static constexpr int kOffMisc = kOffList + kListCap * 2;   // 16 x u32
    static constexpr int kTotal = kOffMisc + 64;
};

// [phase:2 gate: unpack (helpers)]
// Packed pair for sample k of a lane's two runs: low half = run A, high half = run B.
template <int ST>
__device__ __forceinline__ uint32_t pair_at(const uint32_t *ra, const uint32_t *rb, int k)
{
    if (ST == ADSB_SAMPLE_I8) {
        // bytes: [A.k, 0, B.k, 0]; selectors 0-3 pick from the 2nd operand, 4-7 from the 1st
        const uint32_t sel = 0x0C000C00u | (uint32_t)(k & 3) | ((uint32_t)(4 + (k & 3)) << 16);
        return __builtin_amdgcn_perm(rb[k >> 2], ra[k >> 2], sel);
    } else {
        const uint32_t sel = (k & 1) ? 0x07060302u : 0x05040100u;
        return __builtin_amdgcn_perm(rb[k >> 1], ra[k >> 1], sel);
    }
}

// Same value as pair_at with the two source operands exchanged (selectors adjusted): used on the
// rarely taken DF17 path so that hipcc does not merge these with the main path's next-step pair
// and grow a phi (extra exec juggling on every step).
template <int ST>
__device__ __forceinline__ uint32_t pair_at_cold(const uint32_t *ra, const uint32_t *rb, int k)
{
    if (ST == ADSB_SAMPLE_I8) {
        const uint32_t sel = 0x0C000C00u | (uint32_t)(4 + (k & 3)) | ((uint32_t)(k & 3) << 16);
        return __builtin_amdgcn_perm(ra[k >> 2], rb[k >> 2], sel);
    } else {
        const uint32_t sel = (k & 1) ? 0x03020706u : 0x01000504u;
        return __builtin_amdgcn_perm(ra[k >> 1], rb[k >> 1], sel);
    }
}

// [phase:3 decode_candidate: DPP helpers]
// XOR / sum over each row of 16 lanes (a decode group), result in every lane: four DPP steps (VALU
// latency each) instead of four ds_bpermute round trips through the LDS.
#define ADSB_DPP(v, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ uint32_t row16_xor(uint32_t v)
{
    v ^= ADSB_DPP(v, 0xB1);  // quad_perm [1,0,3,2]
    v ^= ADSB_DPP(v, 0x4E);  // quad_perm [2,3,0,1]
    v ^= ADSB_DPP(v, 0x141); // row_half_mirror
    v ^= ADSB_DPP(v, 0x140); // row_mirror
    return v;
}
__device__ __forceinline__ uint32_t row16_sum(uint32_t v)
{
    v += ADSB_DPP(v, 0xB1);
    v += ADSB_DPP(v, 0x4E);
    v += ADSB_DPP(v, 0x141);
    v += ADSB_DPP(v, 0x140);
    return v;
}

// ---- PPM slice + CRC-24 + single-bit repair of one candidate by a 16-lane group --------------------------
// Lane l slices frame byte l (magnitudes off+16+16l .. +15, demod.rs:97-101).  The 24-byte record
// {offset, bytes[14], status, fixed_bit} is written to `rec` (LDS); returns (on every lane) whether the
// frame is valid (CRC matched, or one data bit repaired: crc.rs:49-65).
// [phase:3 slice_byte (helper; inlined twice)]
// The PPM slice of one frame byte (demod.rs:92-131 + 180-201 in closed form): bit (7-k) = m[16 lb + 2k] > m[16 lb + 2k + 1]
// over the magnitudes off+16+16*lb .. +15 of the tile in LDS; strict, a tie gives 0.
template <int ST, bool PAD = false>
__device__ __forceinline__ uint32_t slice_byte(const typename MagT<ST>::type *mag, const uint32_t off, const uint32_t lb)
{
    uint32_t byte = 0;
    if (ST == ADSB_SAMPLE_I16) {
        const typename MagT<ST>::type *mp = mag + off + 16 + 16 * lb;
#pragma unroll
        for (int k = 0; k < 8; ++k) // b - a is negative exactly when a > b (magnitudes < 2^16)
            byte |= (((uint32_t)mp[2 * k + 1] - (uint32_t)mp[2 * k]) >> 31) << (7 - k);
    } else {
        const uint32_t pidx = off + 16 + 16 * lb;
        const uint32_t *mw = reinterpret_cast<const uint32_t *>(mag) + (pidx >> 2);
        const uint32_t sh = pidx & 3;
        uint32_t d0, d1, d2, d3, d4;
        if constexpr (PAD) { // (the five dwords may lie either side of a pad: physical dword = d + 4 (d >> 6))
            const uint32_t *m0 = reinterpret_cast<const uint32_t *>(mag);
            const uint32_t di = pidx >> 2;
            d0 = m0[di + 4u * (di >> 6)];
            d1 = m0[di + 1u + 4u * ((di + 1u) >> 6)];
            d2 = m0[di + 2u + 4u * ((di + 2u) >> 6)];
            d3 = m0[di + 3u + 4u * ((di + 3u) >> 6)];
            d4 = m0[di + 4u + 4u * ((di + 4u) >> 6)];
        } else {
            d0 = mw[0], d1 = mw[1], d2 = mw[2], d3 = mw[3], d4 = mw[4];
        }
        uint32_t w[4] = {__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                         __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh)};
        // dword k = [a0, b0, a1, b1] holds pairs 2k and 2k+1, bit = (a > b), MSB first: one SDWA byte compare per
        // pair into its own SGPR pair, then byte = byte + byte + carry-in per pair (v_addc): 16 VALU instead of 33 for
        // the packed-subtract form, no v_cndmask (which issues four times slower here).  All eight compares come
        // first: gfx950 wants 2 wait states between a VALU writing an SGPR and a VALU reading it, and hipcc pads
        // nothing inside asm.
        uint64_t m0, m1, m2, m3, m4, m5, m6, m7;
        asm("v_cmp_gt_u32_sdwa %1, %9, %9 src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
            "v_cmp_gt_u32_sdwa %2, %9, %9 src0_sel:BYTE_2 src1_sel:BYTE_3\n\t"
            "v_cmp_gt_u32_sdwa %3, %10, %10 src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
            "v_cmp_gt_u32_sdwa %4, %10, %10 src0_sel:BYTE_2 src1_sel:BYTE_3\n\t"
            "v_cmp_gt_u32_sdwa %5, %11, %11 src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
            "v_cmp_gt_u32_sdwa %6, %11, %11 src0_sel:BYTE_2 src1_sel:BYTE_3\n\t"
            "v_cmp_gt_u32_sdwa %7, %12, %12 src0_sel:BYTE_0 src1_sel:BYTE_1\n\t"
            "v_cmp_gt_u32_sdwa %8, %12, %12 src0_sel:BYTE_2 src1_sel:BYTE_3\n\t"
            "v_addc_co_u32_e64 %0, vcc, %0, %0, %1\n\t"
            "v_addc_co_u32_e64 %0, vcc, %0, %0, %2\n\t"
            "v_addc_co_u32_e64 %0, vcc, %0, %0, %3\n\t"
            "v_addc_co_u32_e64 %0, vcc, %0, %0, %4\n\t"
            "v_addc_co_u32_e64 %0, vcc, %0, %0, %5\n\t"
            "v_addc_co_u32_e64 %0, vcc, %0, %0, %6\n\t"
            "v_addc_co_u32_e64 %0, vcc, %0, %0, %7\n\t"
            "v_addc_co_u32_e64 %0, vcc, %0, %0, %8"
            : "+v"(byte), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(m7)
            : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3])
            : "vcc");
    }
    return byte;
}

// [phase:3 slice_byte (nsq image)]
// The same slice from the nsq image (i8, kScanNsq).  The reference compares truncated roots (demod.rs:106 on the
// output of utils.rs:46-52): bit = floor(sqrt(x)) > floor(sqrt(y)) = (r * r > y) with r = floor(sqrt(x)) -- r * r is the
// largest square <= x, so a square lies in (y, x] exactly when r * r > y.  One root per PAIR, survivors only
// (224 samples per survivor ~ 0.1 roots per sample of the stream).  r = trunc(sqrtf(x + 0.5)) is exact for x <= 32768
// (sqrt(x + 0.5) is >= 1.3e-3 from every integer there; v_sqrt_f32 errs by 1 ulp ~ 1e-5).
// A 16-lane group works on one survivor at tile offset `off`; the group's window of 224 samples starts at logical
// dword q + 16 of its half (half = off >= kNsqHalf).  Lane l reads the 16 consecutive samples of 16-aligned chunk
// (q + 16) / 16 + l, moved up by one sample when q + 16 is odd (so that pairs never straddle two lanes), slices its
// 8 pairs, and frame byte l is put together from the chunks of lanes l and l + 1 (one DPP row shift):
// chunk bit j of lane l is frame bit 8 l + j - sh, sh = ((q + 16) % 16) / 2.  All 16 lanes of a group must be active.
__device__ __forceinline__ uint32_t nsq_slice_byte(const uint32_t *img, const uint32_t off, const uint32_t l)
{
    const uint32_t half = off >= (uint32_t)kNsqHalf ? 1u : 0u;
    const uint32_t base = off - half * (uint32_t)kNsqHalf + 16u; // logical dword of the first data sample
    const uint32_t e = base & 15u, par = e & 1u, sh = e >> 1;
    const uint32_t v = (base >> 4) + l;                            // this lane's 16-sample chunk
    const uint32_t a1 = nsq_phys(16u * v) + par;                  // physical dword of its first sample
    // its last sample sits behind a pad when the chunk ends a padded block and was moved up by one
    constexpr uint32_t kChunksPerBlock = (1u << kNsqPadShift) / 16u;
    const uint32_t a2 = a1 + 15u + (((v & (kChunksPerBlock - 1u)) == kChunksPerBlock - 1u ? 4u : 0u) & (0u - par));
    uint32_t d[16];
#pragma unroll
    for (int j = 0; j < 15; ++j) d[j] = img[a1 + j];
    d[15] = img[a2];
    // (x, y) of a pair into one register: x in the low half, y in the high half (selectors 0-3: 2nd operand)
    const uint32_t sel = half ? 0x07060302u : 0x05040100u;
    uint32_t w[8], r2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        w[j] = __builtin_amdgcn_perm(d[2 * j + 1], d[2 * j], sel);
        const float fx = (float)(w[j] & 0xFFFFu) - ((float)kNsqBias - 0.5f); // n + 0.5
        const uint32_t r = (uint32_t)__builtin_amdgcn_sqrtf(fx);
        r2[j] = __umul24(r, r) + (uint32_t)kNsqBias; // (r <= 181; one v_mad_u32_u24)
    }
    // bit j = r2 > y, MSB first: one SDWA compare per pair into its own SGPR pair, then chunk = chunk + chunk +
    // carry-in per pair (v_addc): no v_cndmask.  All eight compares come first: gfx950 wants 2 wait states between a
    // VALU writing an SGPR and a VALU reading it, and hipcc pads nothing inside asm.
    uint32_t chunk = 0;
    uint64_t m0, m1, m2, m3, m4, m5, m6, m7;
    asm("v_cmp_gt_u32_sdwa %1, %9, %17 src0_sel:DWORD src1_sel:WORD_1\n\t"
        "v_cmp_gt_u32_sdwa %2, %10, %18 src0_sel:DWORD src1_sel:WORD_1\n\t"
        "v_cmp_gt_u32_sdwa %3, %11, %19 src0_sel:DWORD src1_sel:WORD_1\n\t"
        "v_cmp_gt_u32_sdwa %4, %12, %20 src0_sel:DWORD src1_sel:WORD_1\n\t"
        "v_cmp_gt_u32_sdwa %5, %13, %21 src0_sel:DWORD src1_sel:WORD_1\n\t"
        "v_cmp_gt_u32_sdwa %6, %14, %22 src0_sel:DWORD src1_sel:WORD_1\n\t"
        "v_cmp_gt_u32_sdwa %7, %15, %23 src0_sel:DWORD src1_sel:WORD_1\n\t"
        "v_cmp_gt_u32_sdwa %8, %16, %24 src0_sel:DWORD src1_sel:WORD_1\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %1\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %2\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %3\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %4\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %5\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %6\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %7\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %8"
        : "+v"(chunk), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(m7)
        : "v"(r2[0]), "v"(r2[1]), "v"(r2[2]), "v"(r2[3]), "v"(r2[4]), "v"(r2[5]), "v"(r2[6]), "v"(r2[7]),
          "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7])
        : "vcc");
    const uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)chunk, 0x101 /* row_shl:1 */, 0xF, 0xF, true);
    return (((chunk << 8) | next) >> (8u - sh)) & 0xFFu;
}

// [phase:3 count_candidate (tiles without slots only: cold)]
// Slot-store overflow (pathological input): the host re-plans from exact counts, so such a tile's
// survivors are decoded in place only to be COUNTED.  A 16-lane group, one lane per frame byte (`byte` = this
// lane's sliced byte, lanes 14/15 contribute nothing): syndrome = XOR of kSyn over the set bits; valid when it is
// zero or the syndrome of one of the 88 data bits (crc.rs:49-65).  Returns the verdict on every lane of the group.
__device__ __forceinline__ bool count_candidate(const bool have, const uint32_t byte, const uint32_t l, const uint32_t lane)
{
    const uint32_t lb = l < 14 ? l : 13;
    uint32_t s = 0;
    const uint32_t *sy = kSyn.v + 8 * lb;
    const int sb = (int)(l < 14 ? byte : 0u);
#pragma unroll
    for (int k = 0; k < 8; ++k) s ^= sy[k] & (uint32_t)((sb << (24 + k)) >> 31); // mask = -bit k (MSB first)
    s = row16_xor(s);
    int found = -1;
    if (s != 0 && l < 11) {
#pragma unroll
        for (int k = 0; k < 8; ++k) found = (sy[k] == s) ? k : found;
    }
    const unsigned long long fm = __ballot(found >= 0);
    const uint32_t gbits = (uint32_t)(fm >> (lane & 48u)) & 0xFFFFu;
    return have && (s == 0 || gbits != 0);
}

// [phase:end]
// Where a tile sits: global tile id -> channel, first sample, number of valid offsets.
struct TilePos {
    uint32_t ch;
    uint64_t sample0;
    uint32_t n_valid;
};
template <int TILE = kTile>
__device__ __forceinline__ TilePos tile_pos(const DemodArgs &p, uint32_t tile)
{
    TilePos t;
    t.ch = tile / p.tiles_per_channel;
    const uint32_t tch = tile - t.ch * p.tiles_per_channel;
    t.sample0 = (uint64_t)tch * TILE;
    const uint64_t left = (p.n_samples - kWindow) - t.sample0; // offsets 0..n-241 exist (adsb.rs:98)
    t.n_valid = left < (uint64_t)TILE ? (uint32_t)left : (uint32_t)TILE;
    return t;
}

// Bounds-checked descriptor over one tile's samples (+halo): reads past the channel end return 0,
// so ragged tails need no branches.  `tile` must be wave-uniform.
template <int BPS, int MAG = kMag>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const DemodArgs &p, const TilePos &t, bool live = true)
{
    const char *base = (const char *)p.iq + ((uint64_t)t.ch * p.channel_stride + t.sample0) * BPS;
    const uint64_t remain = (p.n_samples - t.sample0) * BPS; // bytes to the end of this channel
    const uint32_t nrec = remain > (uint64_t)(MAG * BPS) ? (uint32_t)(MAG * BPS) : (uint32_t)remain;
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, live ? (int)nrec : 0, 0x00020000);
}

// Diagnostic build (-DADSB_TILE_STAMPS=1; measurement only): lane 0 of waves 0 and 3 of every workgroup store the
// shader cycles (s_memtime) they spent in each segment of a tile to DemodArgs::stamps: 16 u32 per tile (8 per
// wave: prologue up to the loads issued, phase 1 arithmetic, barrier, phase 2, barrier, phase 3, wait for the
// loads, and the low word of s_memrealtime at the start), plain stores at the end of the tile.
#ifndef ADSB_TILE_STAMPS
#define ADSB_TILE_STAMPS 0
#endif
#if ADSB_TILE_STAMPS
#define TSTAMP(k)                                                                                              \
    do {                                                                                                       \
        if ((tid & 63u) == 0 && (wave == 0 || wave == 3)) {                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                      \
            if ((k) >= 0) ts_seg[(k)] = (uint32_t)(now_ - ts_prev);                                            \
            ts_prev = now_;                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
        }                                                                                                      \
    } while (0)
#else
#define TSTAMP(k) do { } while (0)
#endif

#ifndef ADSB_ABL_PHASES
#define ADSB_ABL_PHASES 3 // measurement only (no frames come out below 3): 1 = magnitudes only, 2 = magnitudes + gate
#endif
#ifndef ADSB_ABL_NOCMP
#define ADSB_ABL_NOCMP 0 // measurement only (wrong results): gate without compares and branches
#endif
// The product library carries ONE i8 scan kernel (kScanRoot; x its three magnitude modes) and CS16's.  The kernels round 3 and
// round 4 measured against it -- kScanNsq, kScanReg, kScanCode: all bit-exact, none faster -- are compiled only with
// -DADSB_AB_KERNELS=1 (build.sh puts that build in air_rs_amd/lib/variants/libadsb_hip_ab.so; tests/test_gpu_ab_kernels.py
// runs one parity smoke per kernel through it; tools/gpu/ab.sh times them).
#ifndef ADSB_AB_KERNELS
#define ADSB_AB_KERNELS 0
#endif
bool ab_kernels_built() { return ADSB_AB_KERNELS != 0; }
#ifndef ADSB_PRIO
#define ADSB_PRIO 0 // A/B (s_setprio): 1 = prologue + loads at priority 3, 2 = phase 3 at priority 2, 3 = both
#endif
#ifndef ADSB_GATE_GROUP
#define ADSB_GATE_GROUP 1 // steps per wave-uniform test in demod_tiles (see gate_phase)
#endif

// survivor bitmap words -> the lane's survivors appended (unordered) to the LDS list; shared by both gates
template <int RUN, int NT>
__device__ __forceinline__ void gate_collect(uint32_t *candA, uint32_t *candB, uint16_t *list, uint32_t *count,
                                             const uint32_t tid, const uint32_t n_valid)
{
    constexpr int WPR = (RUN + 31) / 32; // (runs of 16: one word per run, its low half used)
    const uint32_t sa = tid * RUN, sb = (tid + NT) * RUN;
    const uint32_t va = n_valid > sa ? (n_valid - sa) : 0u, vb = n_valid > sb ? (n_valid - sb) : 0u;
    uint32_t words[2 * WPR];
    uint32_t nz = 0, c = 0;
#pragma unroll
    for (int k = 0; k < WPR; ++k) {
        words[k] = candA[k];
        words[WPR + k] = candB[k];
    }
    if (n_valid < (uint32_t)(2 * NT * RUN)) { // (wave-uniform) the ragged last tile of a channel
#pragma unroll
        for (int k = 0; k < WPR; ++k) {
            const uint32_t la = va > 32u * k ? va - 32u * k : 0u, lb = vb > 32u * k ? vb - 32u * k : 0u;
            words[k] &= la >= 32u ? 0xFFFFFFFFu : ((1u << la) - 1u);
            words[WPR + k] &= lb >= 32u ? 0xFFFFFFFFu : ((1u << lb) - 1u);
            candA[k] = words[k]; // the dense path of phase 3 reads the bitmap itself
            candB[k] = words[WPR + k];
        }
    }
#pragma unroll
    for (int k = 0; k < 2 * WPR; ++k) {
        nz |= words[k];
        c += __builtin_popcount(words[k]);
    }
    // Survivors are rare (a handful per tile): the few lanes that have any append their offsets,
    // unordered, to the list (ordering happens later).  The bitmap above is only read if there
    // turn out to be more than kSparseCap.
    if (nz) {
        uint32_t pos = atomicAdd(count, c);
#pragma unroll
        for (int k = 0; k < 2 * WPR; ++k) {
            uint32_t bits = words[k];
            const uint32_t base = (k < WPR ? sa : sb) + (k % WPR) * 32;
            while (bits) {
                const uint32_t b = __builtin_ctz(bits);
                bits &= bits - 1;
                if (pos < (uint32_t)kSparseCap) list[pos] = (uint16_t)(base + b);
                ++pos;
            }
        }
    }
}

// ---- the gate (phase 2 of both tile kernels) --------------------------------------------------
// Preamble + DF17 ordering test (demod.rs:17-57) for the 2 x kRun offsets this lane owns:
// run A = offsets [tid*RUN, +RUN), run B = [(tid+NT)*RUN, +RUN) of the tile whose magnitudes
// are in `mag`.  Survivors are OR-ed into the lane's words of the LDS bitmap `cand` and appended,
// unordered, to `list` (their number is added to *count).  No barriers inside (except what `hook`, called
// before step HOOK_AT, does); `tid` < NT; 2 NT RUN = kTile.
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};

// [phase:2 gate: set-up]
template <int ST, int GROUP, int RUN, int NT, int HOOK_AT = -1, class HOOK = NoHook, bool F16OK = (ST == ADSB_SAMPLE_I8), bool PAD = false>
__device__ __forceinline__ void gate_phase(const typename MagT<ST>::type *mag, uint32_t *cand, uint16_t *list,
                                           uint32_t *count, const uint32_t tid, const uint32_t n_valid,
                                           HOOK hook = HOOK())
{
    static_assert(RUN % GROUP == 0 && (RUN == 16 || RUN == 32 || RUN == 64), "steps are taken GROUP at a time; 1 or 2 bitmap words per run");
    constexpr int WPR = (RUN + 31) / 32; // bitmap words per run (a run of 16 uses the low half of its word)
    constexpr int SPG = 16 / (int)sizeof(typename MagT<ST>::type); // magnitudes per 16-byte LDS granule
    // Survivor bitmap: WPR words per run (offset min(RUN, 32) w + b of the tile is bit b of word w), owned by this
    // lane.  The rare path ORs bits straight into LDS so the unrolled steps carry no mask registers.
    uint32_t *candA = cand + WPR * tid, *candB = cand + WPR * (tid + NT);
#pragma unroll
    for (int k = 0; k < WPR; ++k) candA[k] = candB[k] = 0u;
    // (offsets at or beyond n_valid do not exist in the reference loop, adsb.rs:98: masked out in gate_collect)
    constexpr int kGran = (RUN + 26 + SPG - 1) / SPG + 1; // granules a run may touch
    static_assert(!PAD || (ST == ADSB_SAMPLE_I8 && RUN == 32 && (NT * RUN) % 256 == 0), "the padded image: u8 magnitudes, runs of 32 bytes");
    // (padded image: a run starts at physical byte 32 k + 16 (k >> 3); its granules 2 and 3 lie behind the next pad for
    // the last run of a 256-byte row, k % 8 == 7 -- the same lanes for both runs, NT being a multiple of 8: a second
    // base pointer, no arithmetic per granule)
    const u32x4 *ga = reinterpret_cast<const u32x4 *>(mag + mag_phys<PAD>(tid * RUN));
    const u32x4 *gb = reinterpret_cast<const u32x4 *>(mag + mag_phys<PAD>((tid + NT) * RUN));
    const uint32_t hop = PAD && (tid & 7u) == 7u ? 1u : 0u;
    const u32x4 *ga2 = ga + hop, *gb2 = gb + hop;
    uint32_t ra[kGran * 4], rb[kGran * 4];
    constexpr int kAhead = 48 / SPG; // granules resident ahead of the current block
#pragma unroll
    for (int g = 0; g < kAhead; ++g) {
        u32x4 a = (PAD && g >= 2) ? ga2[g] : ga[g], b = (PAD && g >= 2) ? gb2[g] : gb[g];
        ra[4 * g] = a.x; ra[4 * g + 1] = a.y; ra[4 * g + 2] = a.z; ra[4 * g + 3] = a.w;
        rb[4 * g] = b.x; rb[4 * g + 1] = b.y; rb[4 * g + 2] = b.z; rb[4 * g + 3] = b.w;
    }

    // Sliding state shared by neighbouring offsets (all indices are compile-time after
    // unrolling): N[j] sample pair, H2[j] = min(N[j], N[j+2]), W3[j] = max(N[j..j+2]).
    // Sliding state shared by neighbouring offsets (all indices are compile-time after
    // unrolling).  With F[j] = max N[j + {0,2,3,4,5}] the twelve low slots of offset o are
    // F[o+1] u F[o+8] u {o+13,14,15}: one new W3, one new F and one 3-input max per offset.
    //   N[j]  sample pair             H2[j] = min(N[j], N[j+2])
    //   W3[j] = max(N[j..j+2])        F[j]  = max(N[j], W3[j+2], N[j+5])
    uint32_t N[RUN + 26], H2[RUN + 8], W3[RUN + 16], F[RUN + 9];
#pragma unroll
    for (int k = 0; k < 25; ++k) N[k] = pair_at<ST>(ra, rb, k);
#pragma unroll
    for (int j = 0; j < 7; ++j) H2[j] = pkmin(N[j], N[j + 2]);
#pragma unroll
    for (int j = 3; j < 13; ++j) W3[j] = pkmax3<F16OK>(N[j], N[j + 1], N[j + 2]);
#pragma unroll
    for (int j = 1; j < 8; ++j) F[j] = pkmax3<F16OK>(N[j], W3[j + 2], N[j + 5]);

    // [phase:2 gate: steps]
    // GROUP consecutive steps share one wave-uniform test: their 8 x GROUP min/max instructions form one
    // basic block (independent chains the scheduler can interleave) and the common path takes one
    // scalar branch per GROUP steps.  GROUP = 1 is what the many-waves-per-SIMD tile kernel uses; the
    // streaming kernel, whose gate waves are alone on their SIMD's VALU, needs the larger blocks.
#if ADSB_ABL_NOCMP
    uint32_t abl_acc = 0;
#endif
#pragma unroll
    for (int o0 = 0; o0 < RUN; o0 += GROUP) {
        if (o0 == HOOK_AT) hook(); // the streaming kernel places a workgroup barrier inside the gate
        bool pa[GROUP], pb[GROUP];
        bool any = false; // per lane; the wave-wide OR is one ballot below (an s_or chain of the compare masks)
#pragma unroll
        for (int gi = 0; gi < GROUP; ++gi) {
            const int o = o0 + gi;
            if (o % SPG == 0) { // keep 48 samples resident ahead of the block that starts here
                const int g = o / SPG + kAhead;
                if (g * SPG < RUN + 26) {
                    u32x4 a = (PAD && g >= 2) ? ga2[g] : ga[g], b = (PAD && g >= 2) ? gb2[g] : gb[g];
                    ra[4 * g] = a.x; ra[4 * g + 1] = a.y; ra[4 * g + 2] = a.z; ra[4 * g + 3] = a.w;
                    rb[4 * g] = b.x; rb[4 * g + 1] = b.y; rb[4 * g + 2] = b.z; rb[4 * g + 3] = b.w;
                }
            }
            // pairs are unpacked 26 samples ahead so the (rare) DF17 check below finds its ten
            // samples already in registers
            N[o + 25] = pair_at<ST>(ra, rb, o + 25);
            W3[o + 13] = pkmax3<F16OK>(N[o + 13], N[o + 14], N[o + 15]);
            F[o + 8] = pkmax3<F16OK>(N[o + 8], W3[o + 10], N[o + 13]);           // lows 8,10,11,12,13
            const uint32_t lo = pkmax3<F16OK>(F[o + 1], F[o + 8], W3[o + 13]);   // + 1,3,4,5,6 + 13,14,15
            H2[o + 7] = pkmin(N[o + 7], N[o + 9]);
            const uint32_t hi = pkmin(H2[o], H2[o + 7]);                      // highs 0,2,7,9
#if ADSB_ABL_NOCMP
            abl_acc ^= hi ^ lo; // measurement only: keeps the min/max chain alive without compares/branches
            pa[gi] = pb[gi] = false;
#else
            pa[gi] = (uint16_t)hi >= (uint16_t)lo;
            pb[gi] = (hi >> 16) >= (lo >> 16);
            any |= pa[gi] | pb[gi];
#endif
        }
        // [phase:2 gate: DF17 (cold)]
        // wave-uniform test (a scalar branch, no exec juggling): the block below is entered by
        // the whole wave when any lane passes; its effects are masked by pa/pb anyway
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(any) != 0, 0)) {
#pragma unroll
            for (int gi = 0; gi < GROUP; ++gi) {
                const int o = o0 + gi;
                if (GROUP == 1 || __builtin_amdgcn_ballot_w64(pa[gi] | pb[gi]) != 0) {
                    // DF17 part of the gate (demod.rs:45-54)
                    const uint32_t dh = pkmin3<F16OK>(pkmin3<F16OK>(N[o + 16], N[o + 19], N[o + 21]), N[o + 23], N[o + 24]);
                    const uint32_t dl = pkmax3<F16OK>(pkmax3<F16OK>(N[o + 17], N[o + 18], N[o + 20]), N[o + 22], N[o + 25]);
                    const bool da = (uint16_t)dh >= (uint16_t)dl;
                    const bool db = (dh >> 16) >= (dl >> 16);
                    // (offsets at or beyond n_valid are masked out of the bitmap words after the loop, in the one
                    // tile per channel that has any, instead of two compares here)
                    uint32_t bit = 1u << (o & 31);
                    asm("" : "+v"(bit)); // one v_mov for both stores (hipcc rematerialises the constant per exec region)
                    if (pa[gi] & da) atomicOr(candA + (o >> 5), bit);
                    if (pb[gi] & db) atomicOr(candB + (o >> 5), bit);
                }
            }
        }
    }
    // [phase:2 gate: survivor list]
#if ADSB_ABL_NOCMP
    if (abl_acc == 0x12345678u) atomicOr(candA, 1u);
#endif
    gate_collect<RUN, NT>(candA, candB, list, count, tid, n_valid);
}

// [phase:end]
// The same descriptor as four dwords (for inline asm): base, base_hi (stride 0), num_records, flags.
template <int BPS, int MAG = kMag>
__device__ __forceinline__ u32x4 tile_rsrc_words(const DemodArgs &p, const TilePos &t, bool live)
{
    const uint64_t base = (uint64_t)(uintptr_t)((const char *)p.iq + ((uint64_t)t.ch * p.channel_stride + t.sample0) * BPS);
    const uint64_t remain = (p.n_samples - t.sample0) * BPS;
    const uint32_t nrec = remain > (uint64_t)(MAG * BPS) ? (uint32_t)(MAG * BPS) : (uint32_t)remain;
    u32x4 w;
    w.x = __builtin_amdgcn_readfirstlane((uint32_t)base);
    w.y = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32) & 0xFFFFu);
    w.z = __builtin_amdgcn_readfirstlane(live ? nrec : 0u);
    w.w = 0x00020000u;
    return w;
}

// Cache policy of the streaming IQ loads: 2 = nt (read once, do not keep): measured 3 % faster than the
// default policy on the 1 GiB buffer (0.233 vs 0.241 ms).
#ifndef ADSB_LOAD_AUX
#define ADSB_LOAD_AUX 2
#endif
// [phase:1 magnitude (loads, stores: helpers)]
// Phase-1 geometry: one 16-byte load = 8 i8 samples or 4 i16 samples per lane; kIters sweeps of the workgroup cover
// the tile + halo (17 for both sample types at the default tile lengths).
template <int ST> struct P1 {
    static constexpr int kSPL = ST == ADSB_SAMPLE_I8 ? 8 : 4;
    static constexpr int kIters = (TileCfg<ST>::kMagT + kThreads * kSPL - 1) / (kThreads * kSPL);
};

// All of a tile's loads are issued at once (nothing is waited for here): 17 x 16 bytes per lane = the whole tile
// in flight.  `live == false` (there is no next tile) clips the descriptor to zero records: the loads return zeros
// without touching memory.  The last sweep only covers the halo: whole waves past it skip it (scalar branch).
template <int ST>
__device__ __forceinline__ void issue_tile_loads(const DemodArgs &p, const TilePos &tp, bool live, uint32_t tid,
                                                 u32x4 (&raw)[P1<ST>::kIters])
{
    constexpr int BPS = (ST == ADSB_SAMPLE_I8) ? 2 : 4;
    __amdgpu_buffer_rsrc_t rsrc = tile_rsrc<BPS, TileCfg<ST>::kMagT>(p, tp, live);
    const uint32_t wave_s0 = __builtin_amdgcn_readfirstlane(tid & ~63u) * P1<ST>::kSPL;
#pragma unroll
    for (int it = 0; it < P1<ST>::kIters; ++it)
        if ((uint32_t)it * (kThreads * P1<ST>::kSPL) + wave_s0 < (uint32_t)TileCfg<ST>::kMagT)
            // (the sweep's constant goes into the SGPR offset -- no per-load VALU address; gfx950 counts it in the
            // descriptor's bounds check: tools/ubench/soffset_probe.hip)
            raw[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, (uint32_t)it * (kThreads * 16), ADSB_LOAD_AUX);
}

// raw IQ -> magnitudes in LDS (u8 for i8 input, u16 for i16).  Returns (wave-uniform, CS16 only) whether this wave saw
// a magnitude that is not an ordered f16 bit pattern (>= 0x7C00 = 31744): the gate then takes its integer form.
template <int ST, int MAGMODE, bool PAD = false>
__device__ __forceinline__ bool magnitudes_to_lds(const u32x4 (&raw)[P1<ST>::kIters], typename MagT<ST>::type *mag, uint32_t tid)
{
    uint32_t mx = 0;
    const uint32_t wave_s0 = __builtin_amdgcn_readfirstlane(tid & ~63u) * P1<ST>::kSPL;
#pragma unroll
    for (int it = 0; it < P1<ST>::kIters; ++it) {
        if ((uint32_t)it * (kThreads * P1<ST>::kSPL) + wave_s0 < (uint32_t)TileCfg<ST>::kMagT) {
            const uint32_t s = (uint32_t)it * (kThreads * P1<ST>::kSPL) + tid * P1<ST>::kSPL;
            uint32_t lo, hi;
            if (ST == ADSB_SAMPLE_I8) mags8_i8<MAGMODE>(raw[it], lo, hi);
            else {
                mags4_i16(raw[it], lo, hi);
                mx = pkmax(mx, pkmax(lo, hi)); // (samples past the channel end read as zero)
            }
            // (padded image: s = it * 2048 + 8 tid, so s >> 8 = 8 it + (tid >> 5): the pad is a constant per sweep plus a
            // per-lane term -- written out so that the sweep's part folds into the store's immediate offset)
            const uint32_t sp = PAD ? s + 16u * (tid >> 5) + (uint32_t)it * (16u * (kThreads * P1<ST>::kSPL / 256)) : s;
            if (s < (uint32_t)TileCfg<ST>::kMagT) *reinterpret_cast<uint2 *>(mag + sp) = make_uint2(lo, hi);
        }
    }
    return ST == ADSB_SAMPLE_I16 && __builtin_amdgcn_ballot_w64(((mx & 0xFFFFu) >= 0x7C00u) || ((mx >> 16) >= 0x7C00u)) != 0;
}

// [phase:1 nsq (loads, dots, stores)]
// ---- phase 1 of the nsq scan: raw i8 IQ -> the nsq image --------------------------------------------------------
// One sweep of a lane = 16 bytes at sample q0 (eight "A" samples, low halves) and 16 bytes at sample q0 + kNsqHalf
// (eight "B" samples, high halves) -> eight packed dwords -> two ds_write_b128.  kNsqIters sweeps of the workgroup
// cover the image; the last one is the 256-dword halo (lanes 0-31 only).
constexpr int kNsqIters = (kNsqLog + kThreads * 8 - 1) / (kThreads * 8);
constexpr int kNsqFull = kNsqLog / (kThreads * 8);  // sweeps every lane takes part in
constexpr int kNsqTail = kNsqLog % (kThreads * 8);  // logical dwords of the last, partial sweep (the halo: 256)
static_assert(kNsqIters - kNsqFull <= 1 && kNsqTail % 8 == 0, "at most one partial sweep of whole lanes");

__device__ __forceinline__ void nsq_issue_loads(const DemodArgs &p, const TilePos &tp, uint32_t tid,
                                                u32x4 (&ra)[kNsqIters], u32x4 (&rb)[kNsqIters])
{
    __amdgpu_buffer_rsrc_t rsrc = tile_rsrc<2, kMag>(p, tp, true);
    // (the sweep's constant goes into the SGPR offset, which the descriptor's bounds check covers:
    // tools/ubench/soffset_probe.hip; reads past the channel end return zeros)
#pragma unroll
    for (int it = 0; it < kNsqFull; ++it) {
        ra[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, (uint32_t)it * (kThreads * 16), ADSB_LOAD_AUX);
        rb[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, (uint32_t)it * (kThreads * 16) + 2 * kNsqHalf, ADSB_LOAD_AUX);
    }
    if (kNsqTail && __builtin_amdgcn_readfirstlane(tid & ~63u) * 8 < (uint32_t)kNsqTail) { // (whole waves past the halo skip it)
        ra[kNsqFull] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, (uint32_t)kNsqFull * (kThreads * 16), ADSB_LOAD_AUX);
        rb[kNsqFull] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, (uint32_t)kNsqFull * (kThreads * 16) + 2 * kNsqHalf, ADSB_LOAD_AUX);
    }
}

// 8 A samples + 8 B samples -> 8 dwords (B's v << 16) | A's v, v = I^2 + Q^2 + 72.
//   A: v_and (the sample's two bytes) + v_dot4_i32_i8 accumulating onto the SGPR constant 0x00480048 (the bias of
//      both halves at once);
//   B: v_perm to the i16 pair (I * 256, Q * 256) + v_dot2_i32_i16 accumulating onto A's result: (I^2 + Q^2) << 16.
//      (n = 32768, I = Q = -128, wraps to the right bits.)
// 2 VALU per sample, packing included.  gfx950 wants 3 wait states between a DOT and a different VALU reading its
// result and hipcc pads nothing inside asm: the dot2 reads its dot4 eight instructions later, and the block ends in
// s_nop 2.
__device__ __forceinline__ void nsq_pack16(u32x4 a, u32x4 b, uint32_t (&d)[8])
{
    const uint32_t a0 = a.x & 0xFFFFu, a1 = a.x & 0xFFFF0000u, a2 = a.y & 0xFFFFu, a3 = a.y & 0xFFFF0000u,
                   a4 = a.z & 0xFFFFu, a5 = a.z & 0xFFFF0000u, a6 = a.w & 0xFFFFu, a7 = a.w & 0xFFFF0000u;
    // bytes [0, I, 0, Q] of the even / odd sample of a dword (selector 0x0C = a zero byte)
    const uint32_t h0 = __builtin_amdgcn_perm(b.x, b.x, 0x010C000Cu), h1 = __builtin_amdgcn_perm(b.x, b.x, 0x030C020Cu),
                   h2 = __builtin_amdgcn_perm(b.y, b.y, 0x010C000Cu), h3 = __builtin_amdgcn_perm(b.y, b.y, 0x030C020Cu),
                   h4 = __builtin_amdgcn_perm(b.z, b.z, 0x010C000Cu), h5 = __builtin_amdgcn_perm(b.z, b.z, 0x030C020Cu),
                   h6 = __builtin_amdgcn_perm(b.w, b.w, 0x010C000Cu), h7 = __builtin_amdgcn_perm(b.w, b.w, 0x030C020Cu);
    const uint32_t bias2 = (uint32_t)kNsqBias * 0x00010001u;
    asm("v_dot4_i32_i8 %0, %8, %12, %28\n\t"
        "v_dot4_i32_i8 %1, %8, %13, %28\n\t"
        "v_dot4_i32_i8 %2, %9, %14, %28\n\t"
        "v_dot4_i32_i8 %3, %9, %15, %28\n\t"
        "v_dot4_i32_i8 %4, %10, %16, %28\n\t"
        "v_dot4_i32_i8 %5, %10, %17, %28\n\t"
        "v_dot4_i32_i8 %6, %11, %18, %28\n\t"
        "v_dot4_i32_i8 %7, %11, %19, %28\n\t"
        "v_dot2_i32_i16 %0, %20, %20, %0\n\t"
        "v_dot2_i32_i16 %1, %21, %21, %1\n\t"
        "v_dot2_i32_i16 %2, %22, %22, %2\n\t"
        "v_dot2_i32_i16 %3, %23, %23, %3\n\t"
        "v_dot2_i32_i16 %4, %24, %24, %4\n\t"
        "v_dot2_i32_i16 %5, %25, %25, %5\n\t"
        "v_dot2_i32_i16 %6, %26, %26, %6\n\t"
        "v_dot2_i32_i16 %7, %27, %27, %7\n\t"
        "s_nop 2"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
        : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7),
          "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(h4), "v"(h5), "v"(h6), "v"(h7), "s"(bias2));
}

// raw IQ -> the nsq image.  Returns (wave-uniform) whether this wave saw a value that is not an ordered f16 bit
// pattern: v >= 0x7C00, i.e. |I| and |Q| both >= 125 (nine values of n, 31752 .. 32768).  The running
// v_pk_minimum3_f16 finds them all: 0x7C01..0x7FFF are NaNs, which minimum3 propagates, 0x8048 (n = 32768) is a
// negative number, and 0x7C00 (+infinity, which a minimum would not see) is no sum of two squares of i8 values.
__device__ __forceinline__ bool nsq_image_to_lds(const u32x4 (&ra)[kNsqIters], const u32x4 (&rb)[kNsqIters], uint32_t *img, uint32_t tid)
{
    uint32_t lo = 0x7BFF7BFFu; // largest finite pattern
    const uint32_t q_t = tid * 8;
    uint32_t *dst = img + nsq_phys(q_t);
    auto sweep = [&](const int it, const bool store) {
        uint32_t d[8];
        nsq_pack16(ra[it], rb[it], d);
#pragma unroll
        for (int k = 0; k < 8; k += 2) lo = pkmin3<true>(lo, d[k], d[k + 1]);
        if (store) { // (8 kThreads is a multiple of 64: the sweep is a constant offset)
            u32x4 *w = reinterpret_cast<u32x4 *>(dst + nsq_phys(it * kThreads * 8));
            w[0] = u32x4{d[0], d[1], d[2], d[3]};
            w[1] = u32x4{d[4], d[5], d[6], d[7]};
        }
    };
#pragma unroll
    for (int it = 0; it < kNsqFull; ++it) sweep(it, true);
    if (kNsqTail && __builtin_amdgcn_readfirstlane(tid & ~63u) * 8 < (uint32_t)kNsqTail) sweep(kNsqFull, q_t < (uint32_t)kNsqTail);
    return __builtin_amdgcn_ballot_w64(((lo & 0xFFFFu) >= 0x7C00u) || ((lo >> 16) >= 0x7C00u)) != 0;
}

// [phase:2 nsq gate: set-up]
// ---- the gate on the nsq image ------------------------------------------------------------------------------------
// The reference orders truncated roots s(.) = floor(sqrt(.)) (demod.rs:27-36, 48-54 on utils.rs:46-52): pass when
// s(a) >= s(b), a = the smallest "high" n, b = the largest "low" n (s is monotone, so the minimum / maximum of the
// roots are the roots of the minimum / maximum).  On n itself:  a >= b passes outright;  a < b passes only if
// s(a) = s(b), which forces b - a <= 2 s(a) <= 2 sqrt(a) <= a / 8 + 8 (AM-GM).  So with the biased values v = n + 72
//     b' <= t(a'),   t(x) = x + (x >> 3)          [ = n_a + (n_a >> 3) + 9 + 72 ]
// is an exact SUPERSET test in two packed instructions; lanes that pass it for the preamble AND the DF17 group are
// survivors at once when a' >= b' in both, and only the rest (a < b inside the band: a handful per million offsets)
// take two roots per group in a cold block.  Per step (two offsets) the common path is 3 three-input max, 2 min,
// shift, add, 2 compares = 9 VALU on values that need no unpacking.
// F16OK: every value of the tile is below 0x7C00, an ordered f16 pattern (v_pk_maximum3_f16 / v_pk_minimum3_f16);
// otherwise pairs of integer v_pk_max_u16 / v_pk_min_u16.
__device__ __forceinline__ uint32_t nsq_band(uint32_t x)
{
    const u16x2 v = __builtin_bit_cast(u16x2, x);
    return __builtin_bit_cast(uint32_t, (u16x2)(v + (v >> 3)));
}
__device__ __forceinline__ uint32_t nsq_root(uint32_t v) // floor(sqrt(v - 72)), exact for v - 72 <= 32768
{
    return (uint32_t)__builtin_amdgcn_sqrtf((float)v - ((float)kNsqBias - 0.5f));
}

#ifndef ADSB_NSQ_AHEAD
#define ADSB_NSQ_AHEAD 12 // granules of four pairs resident ahead of the current block in the nsq gate (>= 8)
#endif
template <bool F16OK>
__device__ __forceinline__ void gate_phase_nsq(const uint32_t *img, uint32_t *cand, uint16_t *list, uint32_t *count,
                                               const uint32_t tid, const uint32_t n_valid)
{
    constexpr int RUN = kRun, NT = kThreads;
    uint32_t *candA = cand + tid, *candB = cand + (tid + NT);
    *candA = 0u;
    *candB = 0u;
    // run A = offsets 32 tid + o, run B = kNsqHalf + 32 tid + o: logical dwords 32 tid + j, j < RUN + 26: this lane's
    // 32 and the first 28 of the next lane's (which may lie behind a pad)
    const u32x4 *g0 = reinterpret_cast<const u32x4 *>(img + nsq_phys(32 * tid));
    const u32x4 *g1 = reinterpret_cast<const u32x4 *>(img + nsq_phys(32 * tid + 32));
    constexpr int kGran = (RUN + 26 + 3) / 4; // 15 granules of four pairs
    constexpr int kAhead = ADSB_NSQ_AHEAD;    // 48 pairs resident ahead of the current block
    uint32_t N[kGran * 4];
    auto fetch = [&](int g) {
        const u32x4 x = g < 8 ? g0[g] : g1[g - 8];
        N[4 * g] = x.x; N[4 * g + 1] = x.y; N[4 * g + 2] = x.z; N[4 * g + 3] = x.w;
    };
#pragma unroll
    for (int g = 0; g < kAhead; ++g) fetch(g);
    //   N[j]  pair of values               H2[j] = min(N[j], N[j+2])
    //   W3[j] = max(N[j..j+2])             F[j]  = max(N[j], W3[j+2], N[j+5])
    // highs of offset o: min(H2[o], H2[o+7]);  lows: max(F[o+1], F[o+8], W3[o+13])
    uint32_t H2[RUN + 8], W3[RUN + 16], F[RUN + 9];
#pragma unroll
    for (int j = 0; j < 7; ++j) H2[j] = pkmin(N[j], N[j + 2]);
#pragma unroll
    for (int j = 3; j < 13; ++j) W3[j] = pkmax3<F16OK>(N[j], N[j + 1], N[j + 2]);
#pragma unroll
    for (int j = 1; j < 8; ++j) F[j] = pkmax3<F16OK>(N[j], W3[j + 2], N[j + 5]);

    // [phase:2 nsq gate: steps]
#pragma unroll
    for (int o = 0; o < RUN; ++o) {
        if (o % 4 == 0) { // keep 48 pairs resident ahead of the block that starts here
            const int g = o / 4 + kAhead;
            if (g < kGran) fetch(g);
        }
        W3[o + 13] = pkmax3<F16OK>(N[o + 13], N[o + 14], N[o + 15]);
        F[o + 8] = pkmax3<F16OK>(N[o + 8], W3[o + 10], N[o + 13]);           // lows 8,10,11,12,13
        const uint32_t lo = pkmax3<F16OK>(F[o + 1], F[o + 8], W3[o + 13]);   // + 1,3,4,5,6 + 13,14,15
        H2[o + 7] = pkmin(N[o + 7], N[o + 9]);
        const uint32_t hi = pkmin(H2[o], H2[o + 7]);                      // highs 0,2,7,9
        const uint32_t t = nsq_band(hi);
        const bool pa = (uint16_t)t >= (uint16_t)lo;
        const bool pb = (t >> 16) >= (lo >> 16);
        // [phase:2 nsq gate: DF17 (cold)]
        // wave-uniform tests (scalar branches, no exec juggling): a block is entered by the whole wave when any
        // lane needs it; its effects are masked by the lanes' own flags
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(pa | pb) != 0, 0)) {
            // DF17 part of the gate (demod.rs:45-54), the same superset test
            const uint32_t dh = pkmin3<F16OK>(pkmin3<F16OK>(N[o + 16], N[o + 19], N[o + 21]), N[o + 23], N[o + 24]);
            const uint32_t dl = pkmax3<F16OK>(pkmax3<F16OK>(N[o + 17], N[o + 18], N[o + 20]), N[o + 22], N[o + 25]);
            const uint32_t t2 = nsq_band(dh);
            bool sa = pa & ((uint16_t)t2 >= (uint16_t)dl);
            bool sb = pb & ((t2 >> 16) >= (dl >> 16));
            if (__builtin_amdgcn_ballot_w64(sa | sb) != 0) {
                // inside both bands.  Exact at once where both groups are ordered on n itself ...
                const bool ea = ((uint16_t)hi >= (uint16_t)lo) & ((uint16_t)dh >= (uint16_t)dl);
                const bool eb = ((hi >> 16) >= (lo >> 16)) & ((dh >> 16) >= (dl >> 16));
                // [phase:2 nsq gate: roots (cold)]
                if (__builtin_expect(__builtin_amdgcn_ballot_w64((sa & !ea) | (sb & !eb)) != 0, 0)) {
                    // ... the rest by the truncated roots themselves (utils.rs:46-52): ties after truncation pass
                    const bool ra = nsq_root(hi & 0xFFFFu) >= nsq_root(lo & 0xFFFFu) && nsq_root(dh & 0xFFFFu) >= nsq_root(dl & 0xFFFFu);
                    const bool rb = nsq_root(hi >> 16) >= nsq_root(lo >> 16) && nsq_root(dh >> 16) >= nsq_root(dl >> 16);
                    sa = sa && (ea || ra);
                    sb = sb && (eb || rb);
                }
                // (offsets at or beyond n_valid are masked out of the bitmap words afterwards, in the one tile per
                // channel that has any)
                uint32_t bit = 1u << o;
                asm("" : "+v"(bit)); // one v_mov for both stores
                if (sa) atomicOr(candA, bit);
                if (sb) atomicOr(candB, bit);
            }
        }
    }
    // [phase:2 nsq gate: survivor list]
    gate_collect<RUN, NT>(candA, candB, list, count, tid, n_valid);
}

// [phase:end]
// demod_tiles: one workgroup = one tile; the hardware dispatcher starts the next tile as soon as one retires, which
// staggers the phases of a CU's co-resident workgroups.  Per tile:
//   phase 1  the tile's raw IQ (16-byte loads, all in flight at once) becomes the LDS image: i8/kScanRoot and CS16:
//            floor(sqrt) magnitudes (u8 / u16); i8/kScanNsq (the A/B kernel): pairs of biased I^2+Q^2;     ... barrier
//   phase 2  preamble + DF17 gate over the tile's offsets (gate_phase / gate_phase_nsq);                    ... barrier
//   phase 3  every survivor gets a frame slot, its offset and its 14 sliced bytes; CRC-24, repair and ordering are
//            finish_order's (below).
// Measured alternatives: persistent workgroups drawing tiles from per-XCD ticket counters with
// the next tile's loads issued before phase 3 run the same tile in ~10 % more VALU instructions (loop-carried
// registers, SGPR spills) and come out 7 % slower; several tiles per workgroup with the next tile's loads in flight
// need 147 VGPRs (3 waves per SIMD): 0.25-0.27 ms against 0.182.  The instruction count is what bounds this kernel.
// Waves per SIMD the register allocation is held to = workgroups per CU the LDS image allows (4 waves per workgroup):
// 8 for the i8 root scan (u8 magnitudes: 19 KB), 4 for the 16-bit images (nsq, CS16: 35-38 KB).
#ifndef ADSB_SCAN_WAVES
#define ADSB_SCAN_WAVES 4
#endif
#ifndef ADSB_REG_ABL
#define ADSB_REG_ABL 0
#endif
#ifndef ADSB_REG_WAVES
#define ADSB_REG_WAVES 5 // waves per SIMD the register scan's allocation is held to (86-96 VGPRs)
#endif
// The tile body: everything one workgroup does for one tile (`first` = it is the launch's first workgroup: it clears
// the result header's flags).  smem: Lds<ST, SCAN>::kTotal bytes, 16-byte aligned.
template <int ST, int MAGMODE, int SCAN>
__device__ __forceinline__ void scan_tile(const DemodArgs &p, const uint32_t tile, const bool first, unsigned char *smem)
{
    typedef Lds<ST, SCAN> L;
    typedef TileCfg<ST> TC; // tile length of this sample type
    typedef typename L::mag_t mag_t;
    constexpr bool NSQ = L::kNsq;
    static_assert(kThreads / 64 <= 4, "misc[4 + wave] must stay below misc[8]");

    mag_t *mag = reinterpret_cast<mag_t *>(smem);
    uint32_t *img = reinterpret_cast<uint32_t *>(smem); // (nsq) the same bytes as pairs of biased squared magnitudes
    uint32_t *cand = reinterpret_cast<uint32_t *>(smem + L::kOffCand);
    uint16_t *list = reinterpret_cast<uint16_t *>(smem + L::kOffList);
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + L::kOffMisc);

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;

#if ADSB_PRIO & 1 // (A/B: a new workgroup's prologue and loads go out ahead of the resident waves' arithmetic)
    __builtin_amdgcn_s_setprio(3);
#endif
    if (!NSQ && MAGMODE == 1) __builtin_amdgcn_s_setreg((1 | (0 << 6) | ((2 - 1) << 11)), 3);
#if ADSB_TILE_STAMPS
    unsigned long long ts_prev = 0;
    uint32_t ts_seg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    ts_seg[7] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    TSTAMP(-1);
    ts_seg[8] = (uint32_t)ts_prev; // s_memtime next to the s_memrealtime above: the shader clock under this load
#endif
    {
        const TilePos tp = tile_pos<TC::kTileT>(p, tile);
        const uint64_t sample0 = tp.sample0;
        const uint32_t n_valid = tp.n_valid;
        // [phase:1 magnitude (loads, stores)]
        // ---- phase 1, first half: the loads go out before anything else --------------------------------------------
        u32x4 raw[NSQ ? 1 : P1<ST>::kIters];
        u32x4 raw_a[NSQ ? kNsqIters : 1], raw_b[NSQ ? kNsqIters : 1];
        if constexpr (NSQ) nsq_issue_loads(p, tp, tid, raw_a, raw_b);
        else issue_tile_loads<ST>(p, tp, true, tid, raw);
#if ADSB_PRIO & 1
        __builtin_amdgcn_s_setprio(0);
#endif
        TSTAMP(0); // prologue, loads issued
        if (tid == 0 && first) {
            p.hdr->retry = 0;
            if (p.count_groups) { // first pass of a launch: the finishing kernel ORs this launch's flags in
                p.hdr->flags = 0;
                if (p.hdr_pub) p.hdr_pub[2] = 0;
            }
        }
        if (tid == 0) {
            misc[8] = 0;  // valid-frame counter
            misc[12] = 0; // survivor counter
        }
#if ADSB_TILE_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (diagnostic build only) the whole wait for the loads ...
        TSTAMP(6);                                       // ... as its own segment
#endif
        // ---- phase 1, second half: raw IQ -> the LDS image ---------------------------------------------------
        bool wave_big;
        if constexpr (NSQ) wave_big = nsq_image_to_lds(raw_a, raw_b, img, tid);
        else wave_big = magnitudes_to_lds<ST, MAGMODE, MagPad<ST, SCAN>::on>(raw, mag, tid);
        if ((NSQ || ST == ADSB_SAMPLE_I16) && lane == 0) misc[4 + wave] = wave_big ? 1u : 0u; // (every wave writes its own word)
        TSTAMP(1); // phase 1 arithmetic
        __syncthreads();
        TSTAMP(2); // barrier
#if ADSB_ABL_PHASES < 2
        // (keeps the LDS stores of phase 1 alive; never true for real data)
        if (smem[tid * 64] == 0xFD && smem[tid * 64 + 1] == 0xFE && n_valid == 7) misc[12] = 1;
#else

        // [phase:2 gate: call]
        // ---- phase 2: preamble + DF17 gate, two runs per lane, packed u16x2 --------------------
        if constexpr (NSQ || ST == ADSB_SAMPLE_I16) {
            // the 3-input f16 gate whenever every value of the tile is an ordered f16 pattern (nsq: unless some sample
            // has |I| and |Q| >= 125; CS16: any signal below 2/3 of full scale); the integer gate otherwise
            uint32_t any_big = 0;
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) any_big |= misc[4 + w];
            const bool big = any_big != 0; // (workgroup-uniform)
            if constexpr (NSQ) {
                if (!big) gate_phase_nsq<true>(img, cand, list, &misc[12], tid, n_valid);
                else gate_phase_nsq<false>(img, cand, list, &misc[12], tid, n_valid);
            } else {
                if (!big) gate_phase<ST, ADSB_GATE_GROUP, TC::kRunT, kThreads, -1, NoHook, true>(mag, cand, list, &misc[12], tid, n_valid);
                else gate_phase<ST, ADSB_GATE_GROUP, TC::kRunT, kThreads, -1, NoHook, false>(mag, cand, list, &misc[12], tid, n_valid);
            }
        } else {
            gate_phase<ST, ADSB_GATE_GROUP, TC::kRunT, kThreads, -1, NoHook, (ST == ADSB_SAMPLE_I8), MagPad<ST, SCAN>::on>(mag, cand, list, &misc[12], tid, n_valid);
        }
#endif
        TSTAMP(3); // phase 2
        __syncthreads();
        TSTAMP(4); // barrier

#if ADSB_PRIO & 2 // (A/B: the tail of a tile ahead of other waves' arithmetic: the workgroup retires, the next one's loads start)
        __builtin_amdgcn_s_setprio(2);
#endif
        // [phase:3 hand-over: slots, offsets, sliced bytes]
        // ---- phase 3: PPM slice of the gate survivors; the CRC stage is a kernel of its own --------------------
        // Every survivor gets a frame slot, its absolute offset and its 14 sliced bytes (the image is here, in
        // LDS).  CRC-24, repair, ordering inside the tile and the valid-frame count are finish_order's work, one
        // LANE per survivor instead of sixteen.  (With the whole decode in this kernel a tile's 33 KB of LDS were
        // held through a latency-bound epilogue: 19 % of the kernel time for 13 % of its instructions.)
        uint32_t total = misc[12];
#if ADSB_ABL_PHASES < 3
        if (total != 0x7FFFFFFFu) total = 0; // survivors are counted (phase 2 stays alive) but not handed over
#endif
        const bool dense = total > (uint32_t)kSparseCap;
        constexpr int kBitsPerWord = TC::kRunT < 32 ? TC::kRunT : 32; // offsets per survivor-bitmap word (gate_phase)
        u32x4 cw = {0, 0, 0, 0};
        uint32_t cnt = 0, my_first = 0;
        if (dense) {
            // dense fallback: ordered compaction of the bitmap by workgroup-wide prefix sums
            // bitmap: offset kBitsPerWord w + b of the tile is bit b of word w; words 4*tid .. 4*tid+3 per thread
            if (4 * tid < (uint32_t)(TC::kTileT / kBitsPerWord)) cw = reinterpret_cast<const u32x4 *>(cand)[tid];
            cnt = __builtin_popcount(cw.x) + __builtin_popcount(cw.y) + __builtin_popcount(cw.z) +
                  __builtin_popcount(cw.w);
            uint32_t incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t t = __shfl_up(incl, d, 64);
                if ((int)lane >= d) incl += t;
            }
            if (lane == 63) misc[wave] = incl;
            __syncthreads();
            uint32_t wbase = 0;
            total = 0;
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) {
                uint32_t t = misc[w];
                wbase += (w < (int)wave) ? t : 0u;
                total += t;
            }
            my_first = wbase + incl - cnt; // index of this thread's first candidate
        }

        // Frame slots: the tile's own fixed region when the survivors fit (no atomics), otherwise
        // one allocation from the shared pool.  In the usual case (a handful of survivors) every thread knows
        // the base without asking tid 0, and the list is complete since the barrier above: no further barrier.
        const bool simple = !dense && total <= kQuota;
        const uint64_t abs0 = sample0 + p.offset_base; // absolute offset of this tile's offset 0
        uint32_t base_slot = tile * kQuota;
        // 16-lane groups slice one survivor each, one lane per frame byte, and store offset + 14 raw bytes (no CRC
        // verdict yet) into the survivor's slot: 16 survivors per workgroup round
        const uint32_t g = tid >> 4, l = tid & 15;
        auto slice_one = [&](uint32_t off) {
            if constexpr (NSQ) return nsq_slice_byte(img, off, l);
            else return slice_byte<ST, MagPad<ST, SCAN>::on>(mag, off, l < 14 ? l : 13);
        };
        auto slice_round = [&](uint32_t slot0, uint32_t ncl) {
            for (uint32_t r = 0; r < ncl; r += kThreads / 16) {
                if (r + 4 * wave >= ncl) break; // none of this wave's four groups has a survivor
                const uint32_t ci = r + g;
                const bool have = ci < ncl; // uniform within the 16-lane group
                const uint32_t off = have ? list[ci] : 0u;
                const uint32_t byte = slice_one(off);
                if (have) {
                    unsigned char *rec = reinterpret_cast<unsigned char *>(p.slots + (size_t)slot0 + ci);
                    const uint64_t o64 = abs0 + off;
                    if (l < 14) rec[8 + l] = (unsigned char)byte;
                    else reinterpret_cast<uint32_t *>(rec)[l - 14] = l == 14 ? (uint32_t)o64 : (uint32_t)(o64 >> 32);
                }
            }
        };
        if (simple) {
            slice_round(base_slot, total); // unordered list (finish_order ranks it): survivor j -> slot j
        } else {
            if (tid == 0) {
                const unsigned long long b64 = atomicAdd(&p.hdr->alloc, (unsigned long long)total);
                // (pool_off: test knob, adsb_debug_pool_limit -- every tile over its quota loses its slots)
                misc[9] = (!p.pool_off && b64 + total <= (unsigned long long)p.cap_slots) ? p.pool_first + (uint32_t)b64 : kNoBase;
            }
            __syncthreads();
            base_slot = misc[9];
            for (uint32_t chunk = 0; chunk < total; chunk += kListCap) {
                if (dense && cnt) {
                    uint32_t idx = my_first;
                    const uint32_t words[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        uint32_t bits = words[k];
                        while (bits) {
                            const uint32_t b = __builtin_ctz(bits);
                            bits &= bits - 1;
                            if (idx >= chunk && idx < chunk + kListCap)
                                list[idx - chunk] = (uint16_t)((4 * tid + k) * kBitsPerWord + b);
                            ++idx;
                        }
                    }
                }
                __syncthreads();
                const uint32_t ncl = (total - chunk) < (uint32_t)kListCap ? (total - chunk) : (uint32_t)kListCap;
                if (base_slot != kNoBase) {
                    slice_round(base_slot + chunk, ncl); // (ordered when dense; 33..64 survivors: unordered like the simple case)
                } else {
                    // The slot store is full (pathological input: SURVEY F8).  The host re-plans from exact counts,
                    // so this tile's survivors are decoded HERE, from the image in LDS, only to be counted.
                    for (uint32_t r = 0; r < ncl; r += kThreads / 16) {
                        if (r + 4 * wave >= ncl) break; // none of this wave's four groups has a candidate
                        const uint32_t ci = r + g;
                        const bool have = ci < ncl; // uniform within the 16-lane group
                        const uint32_t off = have ? list[ci] : 0u;
                        const bool valid = count_candidate(have, slice_one(off), l, lane);
                        if (valid && l == 0) atomicAdd(&misc[8], 1u);
                    }
                }
                __syncthreads();
            }
        }
        TSTAMP(5); // phase 3
#if ADSB_TILE_STAMPS
        if ((tid & 63u) == 0 && (wave == 0 || wave == 3)) {
            uint32_t *dst = reinterpret_cast<uint32_t *>(p.stamps) + (size_t)tile * 16 + (wave ? 8 : 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[k] = ts_seg[k];
            if (wave) dst[6] = ts_seg[8]; // wave 3's load wait is wave 0's: its slot carries the s_memtime stamp
        }
#endif
        // Seg::valid is written by the decode kernel, except for a tile that lost its slots (counted above)
        if (tid == 0) {
            Seg e;
            e.base = base_slot;
            e.cand = total;
            e.valid = misc[8];
            e.decoded = base_slot == kNoBase ? 1u : 0u;
            p.seg[tile] = e; // (finish_order reads it)
        }
    }
    // [phase:end]
    if (!NSQ && MAGMODE == 1) __builtin_amdgcn_s_setreg((1 | (0 << 6) | ((2 - 1) << 11)), 0);
}

// Which tile a workgroup takes.  The dispatcher deals workgroups to the eight XCDs round-robin (workgroup b runs on XCD b mod 8:
// tools/ubench/xcc_probe.hip reads HW_REG_XCC_ID in every workgroup, profiles/r04_xcc_probe.txt)
// and each XCD has its own L2: with tile = b, a tile's 240-sample halo -- the first samples of the NEXT tile -- is fetched by two
// different XCDs, i.e. twice from HBM (FETCH_SIZE = 1.016 x the buffer for i8, 1.031 x for CS16: exactly the halos).  With the
// launch's tiles cut into eight contiguous ranges, XCD x walking range x in order, neighbouring tiles run on the same XCD at about
// the same time and the halo is an L2 hit.  (n = 8 q + r tiles: XCD x owns q + (x < r) of them, starting at x q + min(x, r); b =
// 8 j + x < n picks the j-th.)
#ifndef ADSB_XCD_MAP
#define ADSB_XCD_MAP 1
#endif
__device__ __forceinline__ uint32_t tile_of_workgroup(uint32_t b, uint32_t n)
{
#if ADSB_XCD_MAP
    const uint32_t q = n >> 3, r = n & 7u, x = b & 7u, j = b >> 3;
    return x * q + (x < r ? x : r) + j;
#else
    (void)n;
    return b;
#endif
}

template <int ST, int MAGMODE, int SCAN>
__global__ __launch_bounds__(kThreads, (ST == ADSB_SAMPLE_I8 && SCAN == kScanRoot) ? 8 : ADSB_SCAN_WAVES) void demod_tiles(DemodArgs p)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[Lds<ST, SCAN>::kTotal];
    scan_tile<ST, MAGMODE, SCAN>(p, p.tile_first + tile_of_workgroup(blockIdx.x, p.tile_count), blockIdx.x == 0, smem);
}

// ---- CRC-24 + single-bit repair of the sliced survivors (demod.rs:71-81; crc.rs:10-65) --------------------------
// Byte-wise table of the Mode-S CRC-24 (generator 0x1FFF409, MSB first, init 0, no final XOR: crc.rs:10-40):
// kCrcTab[v] = (v * x^24) mod G.  crc' = (crc << 8) ^ kCrcTab[(crc >> 16) ^ byte] over the 11 data bytes.
struct CrcTable {
    uint32_t v[256];
};
constexpr CrcTable make_crc_table()
{
    CrcTable t{};
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t r = i << 16;
        for (int k = 0; k < 8; ++k) {
            r <<= 1;
            if (r & 0x1000000u) r ^= 0x1FFF409u;
        }
        t.v[i] = r & 0xFFFFFFu;
    }
    return t;
}
__constant__ CrcTable kCrcTab = make_crc_table();
// The 88 data-bit syndromes sorted, (syndrome << 7) | bit index, padded to 128 entries: the single-bit repair of
// crc.rs:49-65 (at most one of the 88 distinct non-zero values matches) as a 7-step binary search.
struct SynSorted {
    uint32_t v[128];
};
constexpr SynSorted make_syn_sorted()
{
    SynSorted t{};
    SynTable sy = make_syn();
    for (int j = 0; j < 128; ++j) t.v[j] = j < 88 ? ((sy.v[j] << 7) | (uint32_t)j) : 0xFFFFFFFFu;
    for (int i = 1; i < 128; ++i) { // insertion sort (constant evaluation)
        uint32_t x = t.v[i];
        int k = i - 1;
        while (k >= 0 && t.v[k] > x) {
            t.v[k + 1] = t.v[k];
            --k;
        }
        t.v[k + 1] = x;
    }
    return t;
}
__constant__ SynSorted kSynSorted = make_syn_sorted();

// finish_order: the second (and last) kernel of a launch.  The scan kernel (demod_tiles) left, per tile, `Seg{base,
// cand}` and, in the slots base .. base+cand-1, every gate survivor's absolute offset and 14 sliced bytes (unordered when
// cand <= kSparseCap, ascending otherwise).  This kernel checks them (CRC-24 over the 11 data bytes, byte-wise with a
// 1 KB table in LDS; a non-zero syndrome is looked up among the 88 data-bit syndromes and that bit flipped, crc.rs:49-65:
// a flip in the CRC field itself never matches) AND puts the valid frames into the final list in ascending (channel,
// offset) order -- the order the reference's mpsc channel delivers them in (adsb.rs:98-111).  Rounds 1-2 did this in two
// kernels (finish_candidates + gather_tiles: 16.5 + 5.9 us at 32 768 tiles, plus a dispatch gap); fused, every record is
// read once, finished in registers and written once, to its final place:
//   * one workgroup (4 waves) takes 32 consecutive tiles; a wave takes 4 tiles per pass, 16 lanes each, ONE LANE per
//     survivor (a tile has ~8 survivors; tiles with more than 16 take the whole wave afterwards);
//   * the lane's place inside its tile = its rank by offset among the tile's valid frames (15 DPP row rotations);
//   * the tile's place in the list = valid frames in all earlier tiles.  Counts are reduced per workgroup; every
//     workgroup publishes its aggregate in one 8-byte word {value | flag | epoch of the launch} (never cleared: a word
//     of another epoch reads as "not there yet"), the last workgroup of every 64 also publishes the sum of its 64, and
//     a workgroup's start is the sum of the (at most 63) aggregates before it in its own 64 plus the sums of all earlier
//     64s: two dependent round trips of wave-wide loads, whatever the grid size; up to 1024 workgroups (1 GiB of i8 IQ)
//     every workgroup simply sums all aggregates before it, one round trip (a chained look-back degenerates here:
//     all workgroups reach it at the same moment, and the last one would walk N/64 windows one after the other; a
//     ticket counter for arrival order costs 12 us by itself at 88 tickets per microsecond on one address);
//   * a workgroup only ever waits for workgroups with a LOWER block index, and the lowest unfinished block is always
//     resident (each XCD's dispatcher walks its share of the grid in order), so the waits end; a lane that still
//     finds nothing after ~0.1 s gives up and reports it (Header::retry bit 2 -> ADSB_E_STATE on the host) instead of
//     hanging the device;
//   * the last workgroup knows the total: it writes the header and re-arms the pool.
// A re-run of lost tiles (slot-pool overflow, host-planned positions in `out_start`) uses the same kernel without the
// exchange.
constexpr int kFinTiles = 32;          // tiles per workgroup
constexpr int kFinThreads = 256;
constexpr int kFinFan = 64;            // workgroups per second-level sum
constexpr int kFinFlat = 1024;         // up to this many workgroups exchange their aggregates in one level
constexpr uint32_t kLbReady = 1u;      // exchange word: value[31:0] | flag[33:32] | epoch[63:34]
constexpr uint32_t kLbSpinLimit = 1u << 21; // polls (with s_sleep) before a lane gives up: ~0.1 s

// CRC-24 + single-bit repair of one record per lane (w = the record as loaded; straight-line code, no branches, so
// that the chains of a wave's four tiles interleave): returns whether the frame is valid; w comes back finished
// (repaired bit flipped, status and fixed_bit filled in).
__device__ __forceinline__ bool finish_record(uint32_t (&w)[6], const bool have, const uint32_t *crc_tab, const uint32_t *syn_sorted)
{
    // frame byte i is record byte 8 + i: dword 2 + (i >> 2), byte (i & 3)
    uint32_t crc = 0;
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const uint32_t b = (w[2 + (i >> 2)] >> (8 * (i & 3))) & 0xFFu;
        crc = ((crc << 8) ^ crc_tab[((crc >> 16) ^ b) & 0xFFu]) & 0xFFFFFFu;
    }
    const uint32_t rx = ((w[4] >> 24) << 16) | ((w[5] & 0xFFu) << 8) | ((w[5] >> 8) & 0xFFu); // frame bytes 11, 12, 13
    const uint32_t sm = crc ^ rx;
    uint32_t pos = 0; // binary search: first entry whose syndrome is >= sm
#pragma unroll
    for (int step = 64; step >= 1; step >>= 1)
        pos += ((syn_sorted[pos + step - 1] >> 7) < sm) ? (uint32_t)step : 0u;
    const uint32_t hit = syn_sorted[pos];
    const bool found = sm != 0 && (hit >> 7) == sm;
    const bool valid = have && (sm == 0 || found);
    const bool fix = valid && sm != 0;
    const uint32_t fixed = fix ? (hit & 0x7Fu) : 0xFFu;          // data bit 0..87, MSB first
    const uint32_t status = valid ? (sm == 0 ? 0u : 1u) : 0xFFu;
    const uint32_t bi = 8u + ((hit & 0x7Fu) >> 3);               // record byte of that bit
    const uint32_t flip = fix ? ((0x80u >> (hit & 7u)) << (8u * (bi & 3u))) : 0u;
    const uint32_t wi = bi >> 2;                                 // 2, 3 or 4
    w[2] ^= wi == 2 ? flip : 0u;
    w[3] ^= wi == 3 ? flip : 0u;
    w[4] ^= wi == 4 ? flip : 0u;
    w[5] = (w[5] & 0xFFFFu) | (status << 16) | (fixed << 24);
    return valid;
}


template <int N> __device__ __forceinline__ uint32_t row_ror(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N /* row_ror:N */, 0xF, 0xF, true);
}
// number of lanes of this lane's 16-lane row whose key is smaller than its own (keys of lanes that do not count are
// >= 0x10000, above every key that does)
__device__ __forceinline__ uint32_t row_rank(uint32_t key)
{
    uint32_t below = 0;
#define ADSB_RANK_STEP(N) below += row_ror<N>(key) < key ? 1u : 0u;
    ADSB_RANK_STEP(1) ADSB_RANK_STEP(2) ADSB_RANK_STEP(3) ADSB_RANK_STEP(4) ADSB_RANK_STEP(5)
    ADSB_RANK_STEP(6) ADSB_RANK_STEP(7) ADSB_RANK_STEP(8) ADSB_RANK_STEP(9) ADSB_RANK_STEP(10)
    ADSB_RANK_STEP(11) ADSB_RANK_STEP(12) ADSB_RANK_STEP(13) ADSB_RANK_STEP(14) ADSB_RANK_STEP(15)
#undef ADSB_RANK_STEP
    return below;
}

__device__ __forceinline__ void store_record(adsb_frame *dst, const uint32_t (&w)[6])
{
    uint2 *d = reinterpret_cast<uint2 *>(dst);
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = make_uint2(w[2 * k], w[2 * k + 1]);
}
__device__ __forceinline__ void load_record(const adsb_frame *src, uint32_t (&w)[6])
{
    const uint2 *s2 = reinterpret_cast<const uint2 *>(src);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint2 v = s2[k];
        w[2 * k] = v.x;
        w[2 * k + 1] = v.y;
    }
}

// A tile with more than 16 survivors, by a whole wave: finished records go back to the tile's slots in offset order
// (ranked when they arrived unordered, i.e. cand <= kSparseCap; more arrive in order); returns its valid count.
__device__ __forceinline__ uint32_t finish_big_tile(const FinishArgs &a, const Seg &e, const uint32_t *crc_tab,
                                                    const uint32_t *syn_sorted, uint32_t lane)
{
    uint32_t n_good = 0;
    for (uint32_t chunk = 0; chunk < e.cand; chunk += 64) {
        const uint32_t nc = (e.cand - chunk) < 64u ? (e.cand - chunk) : 64u;
        uint32_t w[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0};
        if (lane < nc) load_record(a.slots + (size_t)e.base + chunk + lane, w);
        // (offset all ones: a survivor of the code gate that the samples themselves rejected -- no record)
        const bool valid = finish_record(w, lane < nc && w[1] != 0xFFFFFFFFu, crc_tab, syn_sorted);
        uint32_t slot = lane;
        if (e.cand <= (uint32_t)kSparseCap) { // one chunk, unordered: rank by offset (wrap-safe: a tile spans < 2^15)
            // keys: offset relative to some record's (within 2^15 either way) for records; above them and distinct for rejected
            // ones, so that every lane gets a slot of its own
            const bool rec = lane < nc && w[1] != 0xFFFFFFFFu;
            const unsigned long long rm = __ballot(rec);
            const uint32_t ref = rm ? (uint32_t)__builtin_amdgcn_readlane((int)w[0], (int)__builtin_ctzll(rm)) : 0u;
            const int32_t mine = rec ? (int32_t)(w[0] - ref) : (int32_t)(0x40000000u + lane);
            uint32_t below = 0;
            for (uint32_t k = 0; k < nc; ++k) {
                const int32_t other = (int32_t)__builtin_amdgcn_readlane((int)mine, (int)k);
                below += other < mine ? 1u : 0u;
            }
            slot = below;
        }
        if (lane < nc) store_record(a.slots + (size_t)e.base + chunk + slot, w);
        n_good += (uint32_t)__builtin_popcountll(__ballot(valid));
    }
    return n_good;
}

#ifndef ADSB_FIN_ABL
#define ADSB_FIN_ABL 0 // measurement only (wrong results): 1 no exchange, 2 no CRC / search, 3 no stores of the list, 4 empty kernel
#endif
constexpr int kFinLdsWords = 256 + 128 + 2 * kFinTiles; // tables, per-tile counts, per-tile positions
// What workgroup `blk` of `n_blk` does (lds: kFinLdsWords words).
// PUB_ATOMIC: the caller-owned header copy's flags word is updated with 64-bit atomic ORs (device memory: finish_order).  The
// small-buffer kernel, whose copy lives in pinned HOST memory, passes false and publishes the flags itself with one plain
// store at the end (atomics over PCIe are a platform option, not a given).
template <bool PUB_ATOMIC = true>
__device__ __forceinline__ void finish_block(const FinishArgs &a, const uint32_t blk, const uint32_t n_blk, uint32_t *lds)
{
    uint32_t *crc_tab = lds, *syn_sorted = lds + 256;
    uint32_t *counts = lds + 384;            // valid frames per tile of this workgroup
    uint32_t *tpos = lds + 384 + kFinTiles;  // position of each tile's first frame in the final list
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, sub = lane & 15;
    const uint32_t tile0 = a.tile_first + blk * kFinTiles, t_end = a.tile_first + a.tile_count;

    // ---- this wave's 8 tiles, 4 per pass: one lane per survivor ------------------------------------------------------
    // One memory round trip for everything the lane needs: the tile's Seg AND, without waiting for it, the record at
    // the place a small tile's survivor `sub` is known to be (a tile with at most kQuota survivors keeps them in its
    // own fixed slots, tile * kQuota + i: demod_tiles), next to the tables' words.
    Seg e[2];
    uint32_t w[2][6], rank[2];
    bool valid[2], small_[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const uint32_t idx = wave * 8 + p * 4 + g, tile = tile0 + idx;
        e[p].base = kNoBase; e[p].cand = 0; e[p].valid = 0; e[p].decoded = 1;
        w[p][0] = w[p][1] = 0xFFFFFFFFu;
        w[p][2] = w[p][3] = w[p][4] = w[p][5] = 0;
        if (tile < t_end) {
            e[p] = a.seg[tile];
            load_record(a.slots + (size_t)tile * kQuota + sub, w[p]);
        }
    }
    for (uint32_t i = tid; i < 256 + 128; i += kFinThreads) {
        if (i < 256) crc_tab[i] = kCrcTab.v[i];
        else syn_sorted[i - 256] = kSynSorted.v[i - 256];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const bool live = e[p].cand != 0 && e[p].base != kNoBase && !e[p].decoded;
        small_[p] = live && e[p].cand <= 16u; // (then base == tile * kQuota: what was loaded above is its record)
        if (!(small_[p] && sub < e[p].cand)) {
            w[p][0] = w[p][1] = 0xFFFFFFFFu;
            w[p][2] = w[p][3] = w[p][4] = w[p][5] = 0;
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const uint32_t idx = wave * 8 + p * 4 + g;
#if ADSB_FIN_ABL == 2
        valid[p] = small_[p] && sub < e[p].cand && (w[p][2] & 1u) == 0;
#else
        // (offset all ones: a survivor of the code gate that the samples themselves rejected -- no record)
        valid[p] = finish_record(w[p], small_[p] && sub < e[p].cand && w[p][1] != 0xFFFFFFFFu, crc_tab, syn_sorted);
#endif
        // place inside the tile: rank by offset among the valid frames of the 16-lane row.  A tile's offsets lie within
        // 2^15 of each other: relative to the row's first survivor they fit 16 bits (wrap-safe); bit 16 = does not count.
        // (relative to the row's first RECORD: a rejected survivor's all-ones offset is no reference)
        const uint32_t recs = (uint32_t)(__ballot(w[p][1] != 0xFFFFFFFFu) >> (16u * g)) & 0xFFFFu;
        const uint32_t ref = (uint32_t)__shfl((int)w[p][0], (int)((lane & 48u) + (recs ? (uint32_t)__builtin_ctz(recs) : 0u)), 64);
        const uint32_t key = valid[p] ? ((w[p][0] - ref + 0x8000u) & 0xFFFFu) : 0x10000u;
        rank[p] = row_rank(key);
        const uint32_t cnt = (uint32_t)__builtin_popcount((uint32_t)(__ballot(valid[p]) >> (16u * g)) & 0xFFFFu);
        if (sub == 0) counts[idx] = small_[p] ? cnt : (e[p].decoded ? e[p].valid : 0u); // (a tile the scan had to count itself keeps its count)
    }
    // tiles with more than 16 survivors (coarse or constant input): the whole wave, one tile after the other.  Which
    // ones: a 2 x 4-bit mask from the Seg entries the rows already hold (no further loads on the usual path)
    uint32_t big_mask = 0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const bool big = e[p].cand > 16u && e[p].base != kNoBase && !e[p].decoded;
        const unsigned long long m = __ballot(big && sub == 0); // bits 0, 16, 32, 48: rows 0..3
        big_mask |= (uint32_t)(((m >> 0) & 1u) | ((m >> 15) & 2u) | ((m >> 30) & 4u) | ((m >> 45) & 8u)) << (4 * p);
    }
    for (uint32_t left = big_mask; left; left &= left - 1) { // (wave-uniform)
        const uint32_t i = (uint32_t)__builtin_ctz(left), idx = wave * 8 + i;
        const Seg eb = a.seg[tile0 + idx];
        const uint32_t n_good = finish_big_tile(a, eb, crc_tab, syn_sorted, lane);
        if (lane == 0) counts[idx] = n_good;
    }
    __syncthreads();

    // ---- where this workgroup's frames start: prefix inside the workgroup + look-back over the earlier ones ----------
    if (wave == 0) {
        const uint32_t c = lane < (uint32_t)kFinTiles ? counts[lane] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < kFinTiles; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += t;
        }
        const uint32_t total = __shfl(incl, kFinTiles - 1, 64);
        unsigned long long before = 0; // valid frames in all earlier workgroups
        if (a.out_start == nullptr && ADSB_FIN_ABL != 1) {
            const unsigned long long tag = ((unsigned long long)a.epoch << 34) | ((unsigned long long)kLbReady << 32);
            uint64_t *lb_a = a.lb, *lb_s = a.lb + a.lb_groups_at; // one word per workgroup | one per 64 workgroups
            if (lane == 0 && blk != a.stall_blk) __hip_atomic_store(lb_a + blk, tag | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t grp = blk / kFinFan, r = blk % kFinFan;
            bool failed = false;
            auto wave_sum = [&](unsigned long long x) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const uint32_t lo = __shfl_xor((uint32_t)x, d, 64), hi32 = __shfl_xor((uint32_t)(x >> 32), d, 64);
                    x += ((unsigned long long)hi32 << 32) | lo;
                }
                return x;
            };
            if (n_blk <= (uint32_t)kFinFlat) {
                // small grids (up to 1024 workgroups = 1 GiB of i8 IQ): one level -- every aggregate before this one, 16
                // independent loads per lane in flight at once, re-read until all carry the tag: ONE exchange round trip
                unsigned long long acc;
                uint32_t spins = 0;
                bool ok;
                do {
                    unsigned long long v[kFinFlat / 64];
#pragma unroll
                    for (int u = 0; u < kFinFlat / 64; ++u) {
                        const uint32_t k = (uint32_t)u * 64u + lane;
                        v[u] = k < blk ? __hip_atomic_load(lb_a + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
                    }
                    acc = 0;
                    ok = true;
#pragma unroll
                    for (int u = 0; u < kFinFlat / 64; ++u) {
                        ok = ok && (v[u] >> 32) == (tag >> 32);
                        acc += v[u] & 0xFFFFFFFFull;
                    }
                    if (!__all(ok)) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > kLbSpinLimit) { failed = true; ok = true; acc = 0; }
                    }
                } while (!__all(ok));
                before = wave_sum(acc);
            } else {
                // two levels: the aggregates before this one inside its 64 (one per lane) and the sums of the earlier 64s
                // (lane l takes l, l + 64, ...; published by workgroups with lower indices than this one).  Both sets of
                // loads are in flight together and re-read until they carry the tag: one round trip when the others are
                // ahead, as they mostly are -- not one per level and per 64 earlier sums.  The last of a 64 publishes its
                // 64's sum as soon as its own in-group words are complete (it does not wait for the earlier 64s: no chain).
                unsigned long long in_grp = 0, earlier = 0;
                bool in_done = false, e_done = grp == 0;
                uint32_t spins = 0;
                do {
                    unsigned long long vi = tag, e_acc = 0;
                    bool e_ok = true;
                    if (!in_done && lane < r) vi = __hip_atomic_load(lb_a + (size_t)grp * kFinFan + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!e_done) {
                        for (uint32_t k0 = 0; k0 < grp; k0 += 64u * 8u) { // (wave-uniform trip count; eight loads per lane at a time)
                            unsigned long long v[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const uint32_t k = k0 + (uint32_t)u * 64u + lane;
                                v[u] = k < grp ? __hip_atomic_load(lb_s + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                e_ok = e_ok && (v[u] >> 32) == (tag >> 32);
                                e_acc += v[u] & 0xFFFFFFFFull;
                            }
                        }
                    }
                    if (!in_done && __all((vi >> 32) == (tag >> 32))) {
                        in_grp = wave_sum(vi & 0xFFFFFFFFull);
                        in_done = true;
                        if (r == kFinFan - 1 && lane == 0) { // the last of its 64 publishes their sum (a 64's frames fit 32 bits)
                            const unsigned long long sum64 = in_grp + total;
                            __hip_atomic_store(lb_s + grp, tag | (sum64 > 0xFFFFFFFFull ? 0xFFFFFFFFull : sum64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    if (!e_done && __all(e_ok)) {
                        earlier = wave_sum(e_acc);
                        e_done = true;
                    }
                    if (!(in_done && e_done)) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > kLbSpinLimit) { failed = true; in_done = e_done = true; in_grp = earlier = 0; }
                    }
                } while (!(in_done && e_done));
                before = in_grp + earlier;
            }
            if (__any(failed) && lane == 0) { // gave up waiting: the host returns ADSB_E_STATE; device-side consumers see a list with holes
                atomicOr(&a.hdr->retry, 4u);
                atomicOr(&a.hdr->flags, ADSB_FLAG_INCOMPLETE);
                if (PUB_ATOMIC && a.hdr_pub) atomicOr(reinterpret_cast<unsigned long long *>(a.hdr_pub + 2), (unsigned long long)ADSB_FLAG_INCOMPLETE);
            }
            if (blk == n_blk - 1 && lane == 0) { // the last workgroup: the whole launch's total is known here
                const unsigned long long tot = before + total, n_out = tot < a.max_out ? tot : a.max_out;
                a.hdr->total_found = tot;
                a.hdr->n_out = n_out;
                // flags were cleared by the scan kernel; bits are OR-ed in (another workgroup may add INCOMPLETE)
                if (tot > a.max_out) atomicOr(&a.hdr->flags, ADSB_FLAG_TRUNCATED);
                a.hdr->alloc = 0; // pool allocator: ready for the next launch
                if (a.hdr_pub) {
                    a.hdr_pub[0] = n_out;
                    a.hdr_pub[1] = tot;
                    if (PUB_ATOMIC && tot > a.max_out) atomicOr(reinterpret_cast<unsigned long long *>(a.hdr_pub + 2), (unsigned long long)ADSB_FLAG_TRUNCATED);
                    a.hdr_pub[3] = 0;
                }
                if (a.chan_prefix) a.chan_prefix[a.n_channels] = tot;
            }
        }
        if (lane < (uint32_t)kFinTiles) {
            const uint32_t tile = tile0 + lane;
            unsigned long long pos = before + (incl - c);
            if (a.out_start) pos = tile < t_end ? a.out_start[tile] : 0xFFFFFFFFu;
            const uint32_t p32 = pos > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)pos;
            tpos[lane] = p32;
            if (tile < t_end) {
                // frames before each channel's first tile (adsb_fetch turns them into per-channel counts)
                if (a.chan_prefix && tile % a.tiles_per_channel == 0) a.chan_prefix[tile / a.tiles_per_channel] = pos;
            }
        }
    }
    __syncthreads();

    // ---- every frame to its place -------------------------------------------------------------------------------------
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const uint32_t idx = wave * 8 + p * 4 + g, tile = tile0 + idx;
        const uint32_t dst = tpos[idx] + rank[p];
        if (valid[p] && dst < a.max_out && tpos[idx] != 0xFFFFFFFFu && ADSB_FIN_ABL != 3) store_record(a.out + dst, w[p]);
        if (sub == 0 && tile < t_end) {
            if (small_[p]) a.seg[tile].valid = counts[idx];
            // a tile that lost its slots (the pool was full) but whose frames are wanted: the host re-plans
            if (e[p].base == kNoBase && e[p].cand != 0 && counts[idx] != 0 && tpos[idx] < a.max_out && a.out_start == nullptr) {
                atomicOr(&a.hdr->retry, 1u);
                atomicOr(&a.hdr->flags, ADSB_FLAG_INCOMPLETE); // visible to device-side consumers: the list has holes
                if (PUB_ATOMIC && a.hdr_pub) atomicOr(reinterpret_cast<unsigned long long *>(a.hdr_pub + 2), (unsigned long long)ADSB_FLAG_INCOMPLETE);
            }
        }
    }
    for (uint32_t left = big_mask; left; left &= left - 1) { // the big tiles' valid frames, from their slots (offset order), by the whole wave
        const uint32_t i = (uint32_t)__builtin_ctz(left), idx = wave * 8 + i, tile = tile0 + idx;
        const Seg eb = a.seg[tile];
        if (lane == 0) a.seg[tile].valid = counts[idx];
        uint32_t pos = tpos[idx];
        if (pos >= a.max_out) continue;
        for (uint32_t i0 = 0; i0 < eb.cand; i0 += 64) {
            const uint32_t k = i0 + lane;
            uint32_t x[6] = {0, 0, 0, 0, 0, 0xFF0000u};
            if (k < eb.cand) load_record(a.slots + (size_t)eb.base + k, x);
            const bool ok = k < eb.cand && ((x[5] >> 16) & 0xFFu) != 0xFFu;
            const unsigned long long m = __ballot(ok);
            const uint32_t dst = pos + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (ok && dst < a.max_out) store_record(a.out + dst, x);
            pos += (uint32_t)__builtin_popcountll(m);
        }
    }
}

__global__ __launch_bounds__(kFinThreads) void finish_order(FinishArgs a)
{
#if ADSB_FIN_ABL == 4
    if (a.max_out != 0xFFFFFFF1u) return;
#endif
    __shared__ uint32_t lds[kFinLdsWords];
    finish_block<true>(a, blockIdx.x, gridDim.x, lds);
}

hipError_t launch_finish(hipStream_t s, const FinishArgs &a, hipEvent_t e0, hipEvent_t e1)
{
    const uint32_t blocks = (a.tile_count + kFinTiles - 1) / kFinTiles;
    if (blocks == 0) return hipSuccess;
    hipExtLaunchKernelGGL(finish_order, dim3(blocks), dim3(kFinThreads), 0, s, e0, e1, 0, a);
    return hipGetLastError();
}

// ---- the register scan (i8, kScanReg): the nsq gate without an LDS image -----------------------------------------------
// The nsq scan needs 10 % fewer instructions than the root scan and loses, because its image
// takes 2 bytes of LDS per sample and halves the resident workgroups.  Here the image never exists.  Every WAVE takes a
// chunk of 4032 offsets on its own: lane L slides along run A = offsets 32 L .. 32 L + 31 and run B = run A + 2016, packed
// in the halves of one VGPR as in the nsq scan.  The wave reads its chunk fully coalesced (lane i takes granule
// i + 64 g), turns the granules round in a wave-private 4 KB of LDS so that every lane holds ITS OWN 32 + 32 samples, and
// packs them into 32 VGPRs of v = I^2 + Q^2 + 72.  The 26 samples of window beyond a lane's run are its right
// neighbour's first 26 values -- the same registers one lane up: ONE DPP move each (wave_shl:1), where an image costs a
// store and a load per value and a workgroup barrier.  Lane 63 only supplies them (its run A is lane 0's run B, its run
// B belongs to the next chunk): 63 of 64 lanes produce offsets.  No barrier between loads and gate, no per-sample root,
// no unpacking: 128 (v) + 26 (DPP) + 312 (gate) VALU per 64 offsets where the root scan takes ~ 770, and 18 KB of LDS per
// workgroup.  Survivors are sliced from the IQ bytes themselves (L2-hot), roots only for their 112 pairs.
// A tile (= workgroup = Seg entry = 32 frame slots) is four chunks: 16128 offsets.
constexpr int kRegB = 63 * 32, kRegChunk = 2 * kRegB;
template <bool F16OK>
__device__ __forceinline__ void reg_gate(const uint32_t (&N)[32], uint32_t &bitsA, uint32_t &bitsB)
{
    uint32_t NX[26];
#pragma unroll
    for (int j = 0; j < 26; ++j) NX[j] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)N[j], 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
#define RN(j) ((j) < 32 ? N[(j) < 32 ? (j) : 0] : NX[(j) >= 32 ? (j) - 32 : 0])
    uint32_t H2[32 + 8], W3[32 + 16], F[32 + 9];
#pragma unroll
    for (int j = 0; j < 7; ++j) H2[j] = pkmin(RN(j), RN(j + 2));
#pragma unroll
    for (int j = 3; j < 13; ++j) W3[j] = pkmax3<F16OK>(RN(j), RN(j + 1), RN(j + 2));
#pragma unroll
    for (int j = 1; j < 8; ++j) F[j] = pkmax3<F16OK>(RN(j), W3[j + 2], RN(j + 5));
#pragma unroll
    for (int o = 0; o < 32; ++o) {
        W3[o + 13] = pkmax3<F16OK>(RN(o + 13), RN(o + 14), RN(o + 15));
        F[o + 8] = pkmax3<F16OK>(RN(o + 8), W3[o + 10], RN(o + 13));
        const uint32_t lo = pkmax3<F16OK>(F[o + 1], F[o + 8], W3[o + 13]);
        H2[o + 7] = pkmin(RN(o + 7), RN(o + 9));
        const uint32_t hi = pkmin(H2[o], H2[o + 7]);
        const uint32_t t = nsq_band(hi);
        const bool pa = (uint16_t)t >= (uint16_t)lo;
        const bool pb = (t >> 16) >= (lo >> 16);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(pa | pb) != 0, 0)) {
            const uint32_t dh = pkmin3<F16OK>(pkmin3<F16OK>(RN(o + 16), RN(o + 19), RN(o + 21)), RN(o + 23), RN(o + 24));
            const uint32_t dl = pkmax3<F16OK>(pkmax3<F16OK>(RN(o + 17), RN(o + 18), RN(o + 20)), RN(o + 22), RN(o + 25));
            const uint32_t t2 = nsq_band(dh);
            bool sa = pa & ((uint16_t)t2 >= (uint16_t)dl);
            bool sb = pb & ((t2 >> 16) >= (dl >> 16));
            if (__builtin_amdgcn_ballot_w64(sa | sb) != 0) {
                const bool ea = ((uint16_t)hi >= (uint16_t)lo) & ((uint16_t)dh >= (uint16_t)dl);
                const bool eb = ((hi >> 16) >= (lo >> 16)) & ((dh >> 16) >= (dl >> 16));
                if (__builtin_expect(__builtin_amdgcn_ballot_w64((sa & !ea) | (sb & !eb)) != 0, 0)) {
                    const bool ra = nsq_root(hi & 0xFFFFu) >= nsq_root(lo & 0xFFFFu) && nsq_root(dh & 0xFFFFu) >= nsq_root(dl & 0xFFFFu);
                    const bool rb = nsq_root(hi >> 16) >= nsq_root(lo >> 16) && nsq_root(dh >> 16) >= nsq_root(dl >> 16);
                    sa = sa && (ea || ra);
                    sb = sb && (eb || rb);
                }
                if (sa) bitsA |= 1u << o;
                if (sb) bitsB |= 1u << o;
            }
        }
    }
#undef RN
}

// (wave-private staging in LDS: lane i wrote granule i + 64 g and gets the granules of its own run back, 4 L .. 4 L + 3;
// the granule index is XOR-swizzled so that both the writes and the 64-byte-strided reads of a 16-lane group fall on
// distinct banks.  Lines read with 64-byte strides straight from memory, 2 lanes per 128-byte line and instruction,
// measured 0.174 ms per GiB against 0.168 this way and 0.156 for the coalesced read alone.)
__device__ __forceinline__ uint32_t reg_swz(uint32_t q) { return q ^ ((q >> 4) & 3u); }
__device__ __forceinline__ void reg_transpose(u32x4 *stage, const u32x4 (&in)[4], u32x4 (&out)[4], uint32_t lane)
{
#pragma unroll
    for (int g = 0; g < 4; ++g) stage[reg_swz(lane + 64u * g)] = in[g];
    // (same wave: LDS operations complete in order; no barrier)
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = stage[reg_swz(4u * lane + k)];
}

static_assert(kRegTile == 4 * kRegChunk && kThreads == 256, "four waves, one chunk each");
struct RegLds {
#ifndef ADSB_REG_STAGE_BYTES
#define ADSB_REG_STAGE_BYTES 4096
#endif
    static constexpr int kOffCand = 4 * ADSB_REG_STAGE_BYTES;   // wave-private staging: 4 KB per wave
    static constexpr int kOffList = kOffCand + 2048;            // survivor bitmap: 504 words (offset 32 w + b = bit b of word w)
    static constexpr int kOffMisc = kOffList + kListCap * 2;
    static constexpr int kTotal = kOffMisc + 64;
};

// Frame byte l of the survivor at tile offset `off`, sliced from the IQ bytes (one lane per byte, 16 samples = 8 pairs
// each; lanes 14/15 repeat byte 13).  The reference compares truncated roots (demod.rs:106 on utils.rs:46-52):
// bit = floor(sqrt(x)) > floor(sqrt(y)) = (r * r > y), r = floor(sqrt(x)) -- r * r is the largest square <= x, so a
// square lies in (y, x] exactly when r * r > y.  r = trunc(sqrtf(x + 0.5)) is exact for x <= 32768.
__device__ __forceinline__ uint32_t reg_slice_byte(__amdgpu_buffer_rsrc_t rsrc, const uint32_t off, const uint32_t l)
{
    const uint32_t b = 2u * (off + 16u + 16u * (l < 14u ? l : 13u)); // byte of the lane's first sample in the tile
    const uint32_t base = b & ~3u, sh = b & 3u;
    uint32_t d[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) d[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, base + 4u * k, 0, 0);
    uint32_t byte = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t w = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh); // [I_a, Q_a, I_b, Q_b]
        const uint32_t na = (uint32_t)__builtin_amdgcn_sdot4((int)(w & 0xFFFFu), (int)w, 0, false);
        const uint32_t nb = (uint32_t)__builtin_amdgcn_sdot4((int)(w & 0xFFFF0000u), (int)w, 0, false);
        const uint32_t r = (uint32_t)__builtin_amdgcn_sqrtf((float)na + 0.5f);
        byte |= (r * r > nb ? 1u : 0u) << (7 - k);
    }
    return byte;
}

__device__ __forceinline__ void scan_tile_reg(const DemodArgs &p, const uint32_t tile, const bool first, unsigned char *smem)
{
    typedef RegLds L;
    uint32_t *cand = reinterpret_cast<uint32_t *>(smem + L::kOffCand);
    uint16_t *list = reinterpret_cast<uint16_t *>(smem + L::kOffList);
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + L::kOffMisc);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const TilePos tp = tile_pos<kRegTile>(p, tile);
    const uint64_t sample0 = tp.sample0;
    const uint32_t n_valid = tp.n_valid;
    __amdgpu_buffer_rsrc_t rsrc = tile_rsrc<2, kRegTile + kHalo>(p, tp, true);
    // ---- the chunk's samples: eight coalesced 16-byte loads per lane, all in flight -------------------------------------
    u32x4 la[4], lb[4];
    const uint32_t chunk_byte = __builtin_amdgcn_readfirstlane(wave) * (2u * kRegChunk);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        la[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16u, chunk_byte + 1024u * g, ADSB_LOAD_AUX);
        lb[g] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16u, chunk_byte + 2u * kRegB + 1024u * g, ADSB_LOAD_AUX);
    }
    if (tid == 0 && first) {
        p.hdr->retry = 0;
        if (p.count_groups) { // first pass of a launch: the finishing kernel ORs this launch's flags in
            p.hdr->flags = 0;
            if (p.hdr_pub) p.hdr_pub[2] = 0;
        }
    }
    if (tid == 0) {
        misc[8] = 0;  // valid-frame counter
        misc[12] = 0; // survivor counter
    }
    uint32_t bitsA = 0, bitsB = 0;
    {
        u32x4 ra[4], rb[4];
        u32x4 *stage = reinterpret_cast<u32x4 *>(smem) + (ADSB_REG_STAGE_BYTES / 16) * wave;
        reg_transpose(stage, la, ra, lane);
        reg_transpose(stage, lb, rb, lane);
        uint32_t N[32];
        uint32_t lo = 0x7BFF7BFFu; // (the same detection of values that are no ordered f16 patterns as nsq_image_to_lds)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t d[8];
            nsq_pack16(ra[g], rb[g], d);
#pragma unroll
            for (int k = 0; k < 8; ++k) N[8 * g + k] = d[k];
#pragma unroll
            for (int k = 0; k < 8; k += 2) lo = pkmin3<true>(lo, d[k], d[k + 1]);
        }
        const bool big = __builtin_amdgcn_ballot_w64(((lo & 0xFFFFu) >= 0x7C00u) || ((lo >> 16) >= 0x7C00u)) != 0; // (per wave)
        if (!big) reg_gate<true>(N, bitsA, bitsB);
        else reg_gate<false>(N, bitsA, bitsB);
    }
#if ADSB_REG_ABL == 2 // (measurement: loads + gate only, as the prototype)
    if ((bitsA | bitsB) == 0x12345678u && n_valid == 7) misc[8] = 1;
    return;
#endif
    // offsets that do not exist (adsb.rs:98: the channel's last 240 samples start no window), and lane 63
    const uint32_t oa = wave * (uint32_t)kRegChunk + 32u * lane, ob = oa + (uint32_t)kRegB;
    const uint32_t va = (lane < 63u && n_valid > oa) ? n_valid - oa : 0u, vb = (lane < 63u && n_valid > ob) ? n_valid - ob : 0u;
    bitsA &= va >= 32u ? 0xFFFFFFFFu : ((1u << va) - 1u);
    bitsB &= vb >= 32u ? 0xFFFFFFFFu : ((1u << vb) - 1u);
    // the tile's survivor bitmap (read by the dense path only) and, unordered, its survivor list
    if (lane < 63u) {
        cand[wave * 126u + lane] = bitsA;
        cand[wave * 126u + 63u + lane] = bitsB;
    }
    if (bitsA | bitsB) {
        uint32_t pos = atomicAdd(&misc[12], (uint32_t)(__builtin_popcount(bitsA) + __builtin_popcount(bitsB)));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            uint32_t bits = k ? bitsB : bitsA;
            const uint32_t o0 = k ? ob : oa;
            while (bits) {
                const uint32_t bpos = __builtin_ctz(bits);
                bits &= bits - 1;
                if (pos < (uint32_t)kSparseCap) list[pos] = (uint16_t)(o0 + bpos);
                ++pos;
            }
        }
    }
    __syncthreads();

    // ---- hand-over: every survivor gets a frame slot, its absolute offset and its 14 sliced bytes (as scan_tile's phase 3) --
    uint32_t total = p.fused_pass_only ? 0u : misc[12];
#if ADSB_REG_ABL == 1 // (measurement: survivors counted, not handed over)
    if (total != 0x7FFFFFFFu) total = 0;
#endif
    const bool dense = total > (uint32_t)kSparseCap;
    u32x4 cw = {0, 0, 0, 0};
    uint32_t cnt = 0, my_first = 0;
    if (dense) { // ordered compaction of the bitmap by workgroup-wide prefix sums; words 4 tid .. 4 tid + 3 per thread
        if (4 * tid < (uint32_t)(kRegTile / 32)) cw = reinterpret_cast<const u32x4 *>(cand)[tid];
        cnt = __builtin_popcount(cw.x) + __builtin_popcount(cw.y) + __builtin_popcount(cw.z) + __builtin_popcount(cw.w);
        uint32_t incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += t;
        }
        if (lane == 63) misc[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        total = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) {
            uint32_t t = misc[w];
            wbase += (w < (int)wave) ? t : 0u;
            total += t;
        }
        my_first = wbase + incl - cnt;
    }
    const bool simple = !dense && total <= kQuota;
    const uint64_t abs0 = sample0 + p.offset_base; // absolute offset of this tile's offset 0
    uint32_t base_slot = tile * kQuota;
    const uint32_t g = tid >> 4, l = tid & 15;
    auto slice_round = [&](uint32_t slot0, uint32_t ncl) {
        for (uint32_t r = 0; r < ncl; r += kThreads / 16) {
            if (r + 4 * wave >= ncl) break; // none of this wave's four groups has a survivor
            const uint32_t ci = r + g;
            const bool have = ci < ncl; // uniform within the 16-lane group
            const uint32_t off = have ? list[ci] : 0u;
            const uint32_t byte = reg_slice_byte(rsrc, off, l);
            if (have) {
                unsigned char *rec = reinterpret_cast<unsigned char *>(p.slots + (size_t)slot0 + ci);
                const uint64_t o64 = abs0 + off;
                if (l < 14) rec[8 + l] = (unsigned char)byte;
                else reinterpret_cast<uint32_t *>(rec)[l - 14] = l == 14 ? (uint32_t)o64 : (uint32_t)(o64 >> 32);
            }
        }
    };
    if (simple) {
        slice_round(base_slot, total);
    } else {
        if (tid == 0) {
            const unsigned long long b64 = atomicAdd(&p.hdr->alloc, (unsigned long long)total);
            misc[9] = (!p.pool_off && b64 + total <= (unsigned long long)p.cap_slots) ? p.pool_first + (uint32_t)b64 : kNoBase;
        }
        __syncthreads();
        base_slot = misc[9];
        for (uint32_t chunk = 0; chunk < total; chunk += kListCap) {
            if (dense && cnt) {
                uint32_t idx = my_first;
                const uint32_t words[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t bits = words[k];
                    while (bits) {
                        const uint32_t bpos = __builtin_ctz(bits);
                        bits &= bits - 1;
                        if (idx >= chunk && idx < chunk + kListCap) list[idx - chunk] = (uint16_t)((4 * tid + k) * 32 + bpos);
                        ++idx;
                    }
                }
            }
            __syncthreads();
            const uint32_t ncl = (total - chunk) < (uint32_t)kListCap ? (total - chunk) : (uint32_t)kListCap;
            if (base_slot != kNoBase) {
                slice_round(base_slot + chunk, ncl);
            } else { // the slot store is full: this tile's survivors are decoded here only to be counted
                for (uint32_t r = 0; r < ncl; r += kThreads / 16) {
                    if (r + 4 * wave >= ncl) break;
                    const uint32_t ci = r + g;
                    const bool have = ci < ncl;
                    const uint32_t off = have ? list[ci] : 0u;
                    const bool valid = count_candidate(have, reg_slice_byte(rsrc, off, l), l, lane);
                    if (valid && l == 0) atomicAdd(&misc[8], 1u);
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        Seg e;
        e.base = base_slot;
        e.cand = total;
        e.valid = misc[8];
        e.decoded = base_slot == kNoBase ? 1u : 0u;
        p.seg[tile] = e;
    }
}

#if ADSB_AB_KERNELS
__global__ __launch_bounds__(kThreads, ADSB_REG_WAVES) void demod_tiles_reg(DemodArgs p)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[RegLds::kTotal];
    scan_tile_reg(p, p.tile_first + tile_of_workgroup(blockIdx.x, p.tile_count), blockIdx.x == 0, smem);
}
#endif

// ---- the code scan (i8, kScanCode): the gate on an 8-bit LOG code of n = I^2 + Q^2, no root per sample --------------------
// The root scan spends 3.5 of its 5.5 phase-1 issue slots per sample on floor(sqrt(n)) (v_sqrt_f32 alone holds the SIMD
// for 8 cycles) and is VALU-issue-bound.  The gate (demod.rs:17-57) only ORDERS truncated roots, and every survivor is
// checked again before it leaves the kernel, so the image the gate slides over may be any monotone 8-bit code c(n) as
// long as the test on codes passes wherever the reference's test on floor(sqrt) passes (a SUPERSET test) and whatever
// it lets through is decided exactly afterwards:
//   * c(n) = e4m3((n + 16) / 128): one v_pk_fma_f32 + one v_cvt_pk_fp8_f32 per PAIR of samples (a quarter-rate
//     conversion: 1 slot per sample where root + pack take 3.5).  Monotone in n; 8 codes per octave of n + 16.
//   * the gate's values are 16-bit lanes [other sample's code | code << 8]: the code in the HIGH byte of an f16 bit
//     pattern below 0x7C00, so v_pk_maximum3_f16 / v_pk_min_u16 order them by code (the low byte only breaks ties
//     between equal codes and never reaches a decision: all compares are on byte 1 / byte 3).  An image dword holds the
//     codes of samples 2q, 2q+1 of the tile's first half and of its second half: [A(2q), A(2q+1), B(2q), B(2q+1)] -- the
//     dword as it is serves sample 2q+1 of a lane's two runs, shifted left by 8 bits sample 2q: half an unpacking
//     instruction per step where the root image needs one v_perm.
//   * floor(sqrt(hi)) >= floor(sqrt(lo)) holds exactly when lo <= top(hi), the largest n with hi's root.  On codes:
//     c(lo) <= byte1(S(pattern(hi))) with S(x) = 1.5 x + 2^-7 in f16 ARITHMETIC on the pattern (one v_pk_fma_f16): the
//     f16 value of a pattern grows like (n + 16)^2, so one multiply-add bends the slack the way 2 sqrt(n) needs -- wide
//     (in codes) at low levels, one code at high ones.  That byte1(S(c(n) << 8)) >= c(top(n)) for EVERY n is checked on
//     the device, through these very instructions, when a context is created (adsb_create fails otherwise), and again
//     by tests/test_gpu_code_scan.py.
//   * a survivor of the code gate is CERTAIN when the codes themselves are strictly ordered (then n is) in both groups,
//     and a sliced bit is certain when c(x) > byte1(S(c(y))) (bit 1) or c(x) < c(y) (bit 0).  Anything else (0.6 per
//     tile on the synthetic stream) is decided from the samples themselves: the 16-lane group re-reads its 240 samples
//     (L2 / Infinity Cache; 480 bytes) and runs the reference's arithmetic -- floor(sqrt) by v_sqrt_f32 of n + 0.5 --
//     on them.  A survivor that fails there leaves a record whose offset is all ones; finish_order skips it.
// Everything downstream (slots, Seg, finish_order, the small-buffer kernel) is the root scan's.
constexpr int kCodeHalf = kTile / 2;                    // samples in the half a lane's run A / run B slides over
constexpr int kCodeLog = (kCodeHalf + kHalo) / 2;       // logical dwords of the image (two samples of each half per dword)
// The gate's ds_read_b128 has lane L start at dword 16 L: lanes L, L+4, L+8, L+12 of a 16-lane read group fall on the same
// banks (4-way).  -DADSB_CODE_PAD=1 puts 4 pad dwords after every 64 (conflict-free; the slicer then pays for the address
// arithmetic): measured no faster (profiles/r04_ab_code_pad.txt) -- the LDS is 20 % busy either way.
#ifndef ADSB_CODE_PAD
#define ADSB_CODE_PAD 0
#endif
#ifndef ADSB_CODE_ABL
#define ADSB_CODE_ABL 0
#endif
__host__ __device__ constexpr uint32_t code_phys(uint32_t q) { return ADSB_CODE_PAD ? q + 4u * (q >> 6) : q; }
constexpr int kCodePhys = (int)code_phys(kCodeLog);
constexpr int kCodeBias = 16;                           // c(n) = e4m3((n + kCodeBias) * 2^-kCodeShift)
constexpr int kCodeShift = 7;
constexpr uint32_t kCodeSlackMul = 0x3E003E00u;         // 1.5    (f16 x 2)
constexpr uint32_t kCodeSlackAdd = 0x20002000u;         // 2^-7   (f16 x 2)
static_assert(kRun == 32 && kThreads == 256 && kCodeHalf % (kThreads * 8) == 0 && kHalo == 256, "code scan geometry");

struct CodeLds {
    static constexpr int kOffCand = kCodePhys * 4;                 // survivor bitmap: word w = offsets 32 w .. 32 w + 31
    static constexpr int kOffList = kOffCand + 2 * kThreads * 4;   // kListCap x u16
    static constexpr int kOffMisc = kOffList + kListCap * 2;
    static constexpr int kTotal = kOffMisc + 64;
};
static_assert(CodeLds::kTotal <= 20480, "eight workgroups per CU");

// the threshold pattern of a pair of code patterns (see above): byte 1 / byte 3 of the result are what lows compare with
__device__ __forceinline__ uint32_t code_slack(uint32_t x, uint32_t add = kCodeSlackAdd)
{
    const f16x2 r = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, x), __builtin_bit_cast(f16x2, kCodeSlackMul),
                                              __builtin_bit_cast(f16x2, add));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t byte1(uint32_t x) { return (x >> 8) & 0xFFu; }
__device__ __forceinline__ uint32_t byte3(uint32_t x) { return x >> 24; }

// (2^23 + n) as float bits (what the dot4 leaves) -> (n + kCodeBias) * 2^-kCodeShift, two samples per v_pk_fma_f32 (exact)
__device__ __forceinline__ f32x2 code_arg(int n0, int n1)
{
    constexpr float s = 1.0f / (float)(1 << kCodeShift), t = ((float)kCodeBias - 8388608.0f) / (float)(1 << kCodeShift);
    const f32x2 f = {__builtin_bit_cast(float, n0), __builtin_bit_cast(float, n1)};
    return __builtin_elementwise_fma(f, (f32x2){s, s}, (f32x2){t, t});
}

// probe (adsb_create, tests): out[n] = c(n) | byte1(S(c(n) << 8)) << 8 for n = 0 .. 32768, through the scan's own code
__global__ void code_probe_kernel(uint16_t *out)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n > 32768u) return;
    const f32x2 x = code_arg((int)(0x4B000000u + n), (int)(0x4B000000u + n));
    const uint32_t c = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(x.x, x.y, 0, false) & 0xFFu;
    out[n] = (uint16_t)(c | (byte1(code_slack(c << 8)) << 8));
}
hipError_t launch_code_probe(hipStream_t s, uint16_t *dev_out32769)
{
    hipLaunchKernelGGL(code_probe_kernel, dim3(129), dim3(256), 0, s, dev_out32769);
    return hipGetLastError();
}

// [phase:1 code (loads, dots, conversions, stores)]
constexpr int kCodeFull = kCodeHalf / (kThreads * 8);   // sweeps every lane takes part in (4); one more covers the halo
__device__ __forceinline__ void code_issue_loads(__amdgpu_buffer_rsrc_t rsrc, uint32_t tid, u32x4 (&ra)[kCodeFull + 1], u32x4 (&rb)[kCodeFull + 1])
{
    // (the sweep's constant goes into the SGPR offset, which the descriptor's bounds check covers; reads past the channel
    // end return zeros)
#pragma unroll
    for (int it = 0; it < kCodeFull; ++it) {
        ra[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, (uint32_t)it * (kThreads * 16), ADSB_LOAD_AUX);
        rb[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, (uint32_t)it * (kThreads * 16) + 2 * kCodeHalf, ADSB_LOAD_AUX);
    }
    if (__builtin_amdgcn_readfirstlane(tid & ~63u) * 8 < (uint32_t)kHalo) { // the halo: 256 samples of each half (wave 0, lanes 0-31)
        ra[kCodeFull] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, (uint32_t)kCodeFull * (kThreads * 16), ADSB_LOAD_AUX);
        rb[kCodeFull] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, (uint32_t)kCodeFull * (kThreads * 16) + 2 * kCodeHalf, ADSB_LOAD_AUX);
    }
}
// 8 samples of the first half + the 8 samples half a tile further -> four image dwords
__device__ __forceinline__ u32x4 code_pack16(u32x4 a, u32x4 b)
{
    int n[8];
    uint32_t d[4];
    dot4x8_sacc(a, 0x4B000000, n);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 x = code_arg(n[2 * j], n[2 * j + 1]);
        // (asm: the builtin ties the destination -- the conversion keeps its other half -- and costs a v_mov per dword; that
        // half is overwritten below, so whatever the register held will do)
        asm("v_cvt_pk_fp8_f32 %0, %1, %2" : "=v"(d[j]) : "v"(x.x), "v"(x.y));
    }
    dot4x8_sacc(b, 0x4B000000, n);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 x = code_arg(n[2 * j], n[2 * j + 1]);
        d[j] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(x.x, x.y, (int)d[j], true);
    }
    return u32x4{d[0], d[1], d[2], d[3]};
}
__device__ __forceinline__ void code_image_to_lds(const u32x4 (&ra)[kCodeFull + 1], const u32x4 (&rb)[kCodeFull + 1], uint32_t *img, uint32_t tid)
{
    // logical dword of a lane's four in sweep `it`: it * 1024 + 4 tid; its pad 4 (it * 16 + (tid >> 4)): a constant per sweep
    u32x4 *dst = reinterpret_cast<u32x4 *>(img + code_phys(4 * tid));
#pragma unroll
    for (int it = 0; it < kCodeFull; ++it) dst[(int)code_phys(it * kThreads * 4) / 4] = code_pack16(ra[it], rb[it]);
    if (__builtin_amdgcn_readfirstlane(tid & ~63u) * 8 < (uint32_t)kHalo) {
        const u32x4 d = code_pack16(ra[kCodeFull], rb[kCodeFull]);
        if (tid * 8 < (uint32_t)kHalo) dst[(int)code_phys(kCodeFull * kThreads * 4) / 4] = d;
    }
}

// [phase:2 code gate]
// Preamble + DF17 superset test for the 2 x 32 offsets this lane owns: run A = offsets 32 tid + o, run B = kCodeHalf +
// 32 tid + o.  Offsets that pass both groups on codes are OR-ed into the lane's words of the LDS bitmap (candA / candB).
__device__ __forceinline__ void gate_phase_code(const uint32_t *img, uint32_t *candA, uint32_t *candB, const uint32_t tid, const uint32_t abl_key = 0)
{
    constexpr int RUN = kRun;
    // (survivors are OR-ed straight into the lane's two bitmap words in LDS: accumulators in registers cost four copies
    // per step at every join of the unrolled steps)
    *candA = 0u;
    *candB = 0u;
    uint32_t slack_add = kCodeSlackAdd; // (VOP3P takes one scalar operand: the other constant lives in a VGPR, once)
    asm volatile("" : "+v"(slack_add));
    // logical dwords 16 tid + k, k < 29: this lane's 16 and the first 13 of the next lane's (which may lie behind a pad)
    const u32x4 *g0 = reinterpret_cast<const u32x4 *>(img + code_phys(16 * tid));
    const u32x4 *g1 = reinterpret_cast<const u32x4 *>(img + code_phys(16 * tid + 16));
    constexpr int kGran = (RUN + 26 + 7) / 8; // granules of four dwords = eight samples of each run
    constexpr int kAhead = 6;                 // 48 samples resident ahead of the current step
    uint32_t W[kGran * 4];
    auto fetch = [&](int g) {
        const u32x4 x = g < 4 ? g0[g] : g1[g - 4];
        W[4 * g] = x.x; W[4 * g + 1] = x.y; W[4 * g + 2] = x.z; W[4 * g + 3] = x.w;
    };
#pragma unroll
    for (int g = 0; g < kAhead; ++g) fetch(g);
    //   N[j]  the pair of code patterns of sample j  H2[j] = min(N[j], N[j+2])
    //   W3[j] = max(N[j..j+2])                       F[j]  = max(N[j], W3[j+2], N[j+5])
    // highs of offset o: min(H2[o], H2[o+7]);  lows: max(F[o+1], F[o+8], W3[o+13])
    uint32_t N[RUN + 26], H2[RUN + 8], W3[RUN + 16], F[RUN + 9];
#define ADSB_CODE_N(j) (((j) & 1) ? W[(j) >> 1] : (W[(j) >> 1] << 8))
#pragma unroll
    for (int k = 0; k < 25; ++k) N[k] = ADSB_CODE_N(k);
#pragma unroll
    for (int j = 0; j < 7; ++j) H2[j] = pkmin(N[j], N[j + 2]);
#pragma unroll
    for (int j = 3; j < 13; ++j) W3[j] = pkmax3<true>(N[j], N[j + 1], N[j + 2]);
#pragma unroll
    for (int j = 1; j < 8; ++j) F[j] = pkmax3<true>(N[j], W3[j + 2], N[j + 5]);
#pragma unroll
    for (int o = 0; o < RUN; ++o) {
        if (o % 8 == 0) {
            const int g = o / 8 + kAhead;
            if (g < kGran) fetch(g);
        }
        N[o + 25] = ADSB_CODE_N(o + 25);
        W3[o + 13] = pkmax3<true>(N[o + 13], N[o + 14], N[o + 15]);
        F[o + 8] = pkmax3<true>(N[o + 8], W3[o + 10], N[o + 13]);           // lows 8,10,11,12,13
        const uint32_t lo = pkmax3<true>(F[o + 1], F[o + 8], W3[o + 13]);   // + 1,3,4,5,6 + 13,14,15
        H2[o + 7] = pkmin(N[o + 7], N[o + 9]);
        const uint32_t hi = pkmin(H2[o], H2[o + 7]);                        // highs 0,2,7,9
        const uint32_t th = code_slack(hi, slack_add);
        const bool pa = byte1(th) >= byte1(lo);
        const bool pb = byte3(th) >= byte3(lo);
        // wave-uniform tests (scalar branches): a block is entered by the whole wave when any lane needs it
#if ADSB_CODE_ABL == 1 // (measurement only, wrong results: the hot path alone -- the cold block is never entered)
        if (__builtin_expect(__builtin_amdgcn_ballot_w64((pa | pb) && abl_key == 7u) != 0, 0)) { // (n_valid == 7: never)
#else
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(pa | pb) != 0, 0)) {
#endif
            // DF17 part of the gate (demod.rs:45-54), the same superset test
            const uint32_t dh = pkmin3<true>(pkmin3<true>(N[o + 16], N[o + 19], N[o + 21]), N[o + 23], N[o + 24]);
            const uint32_t dl = pkmax3<true>(pkmax3<true>(N[o + 17], N[o + 18], N[o + 20]), N[o + 22], N[o + 25]);
            const uint32_t t2 = code_slack(dh, slack_add);
            const bool sa = pa & (byte1(t2) >= byte1(dl));
            const bool sb = pb & (byte3(t2) >= byte3(dl));
            uint32_t bit = 1u << o;
            asm("" : "+v"(bit)); // one v_mov for both stores
            if (sa) atomicOr(candA, bit);
            if (sb) atomicOr(candB, bit);
        }
    }
#undef ADSB_CODE_N
}

// [phase:3 code slicer]
// One survivor (tile offset `off`) by its 16-lane group, from the codes.  Lane l < 14: frame byte l -- bit = 1 where
// c(x) > byte1(S(c(y))) (then floor(sqrt(x)) > floor(sqrt(y)): x lies above every n that shares y's root), 0 where
// c(x) < c(y) (then x < y).  Lane 14 looks at the preamble's 16 samples, lane 15 at the ten DF17 samples: the gate's
// verdict is CERTAIN where the codes themselves are strictly ordered (then n is).  Returns the lane's byte; `unc_mask` =
// the wave's lanes that saw a pair / a group which is neither: the samples themselves decide (exact_from_raw).
__device__ __forceinline__ uint32_t code_slice_byte(const uint32_t *img, const uint32_t off, const uint32_t l, unsigned long long &unc_mask)
{
    const uint32_t h = off >= (uint32_t)kCodeHalf ? 1u : 0u;
    // the lane's first sample, counted inside its half: byte l's sixteen, the preamble's (lane 14), DF17's (lane 15)
    const uint32_t s0 = off - h * (uint32_t)kCodeHalf + (l < 14u ? 16u + 16u * l : (l == 14u ? 0u : 16u));
    const uint32_t e = s0 & 1u, q = s0 >> 1;
    uint32_t D[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) D[i] = img[code_phys(q + (uint32_t)i)];
    // [0, c(y), 0, c(x)] of the lane's pair k (x = its sample 2k, y = 2k+1) by one v_perm over dwords k, k+1 (selectors
    // 0-3: 2nd operand, 4-7: 1st, 0x0C: zero): e = 0: bytes 2h, 2h+1 of dword k;  e = 1: byte 2h+1 of dword k, byte 2h of k+1
    const uint32_t cx_sel = 2u * h + e, cy_sel = e ? 4u + 2u * h : 2u * h + 1u;
    const uint32_t sel = 0x000C000Cu | (cy_sel << 8) | (cx_sel << 24);
    uint32_t xy[8], t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        xy[k] = __builtin_amdgcn_perm(D[k + 1], D[k], sel);
        t[k] = code_slack(xy[k]);
    }
    // bit k = c(x) > byte1(S(c(y))), MSB first: one SDWA compare per pair into its own SGPR pair, then byte = byte + byte +
    // carry-in per pair (v_addc): no v_cndmask.  All eight compares come first: gfx950 wants 2 wait states between a VALU
    // writing an SGPR and a VALU reading it, and hipcc pads nothing inside asm.
    uint32_t byte = 0;
    uint64_t m0, m1, m2, m3, m4, m5, m6, m7;
    asm("v_cmp_gt_u32_sdwa %1, %9, %17 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t"
        "v_cmp_gt_u32_sdwa %2, %10, %18 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t"
        "v_cmp_gt_u32_sdwa %3, %11, %19 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t"
        "v_cmp_gt_u32_sdwa %4, %12, %20 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t"
        "v_cmp_gt_u32_sdwa %5, %13, %21 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t"
        "v_cmp_gt_u32_sdwa %6, %14, %22 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t"
        "v_cmp_gt_u32_sdwa %7, %15, %23 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t"
        "v_cmp_gt_u32_sdwa %8, %16, %24 src0_sel:BYTE_3 src1_sel:BYTE_1\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %1\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %2\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %3\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %4\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %5\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %6\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %7\n\t"
        "v_addc_co_u32_e64 %0, vcc, %0, %0, %8"
        : "+v"(byte), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(m7)
        : "v"(xy[0]), "v"(xy[1]), "v"(xy[2]), "v"(xy[3]), "v"(xy[4]), "v"(xy[5]), "v"(xy[6]), "v"(xy[7]),
          "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7])
        : "vcc");
    // a pair is decided when bit 1 is certain (above) or c(x) < c(y); the lane masks stay in scalar registers
    const uint64_t ms[8] = {m0, m1, m2, m3, m4, m5, m6, m7};
    unsigned long long decided = ~0ull;
#pragma unroll
    for (int k = 0; k < 8; ++k) decided &= ms[k] | __builtin_amdgcn_ballot_w64(byte3(xy[k]) < byte1(xy[k]));
    // even samples sit in the high halves, odd ones in the low halves (patterns c << 8)
    // preamble (demod.rs:20-22): highs 0, 2, 7, 9; lows the other twelve
    const uint32_t p1 = pkmin(xy[0], xy[1]), p2 = pkmin(xy[3], xy[4]);                       // hi: {0, 2} | lo: {7, 9}
    const uint32_t ma = pkmax3<true>(xy[2], xy[3], xy[4]), mb = pkmax3<true>(xy[5], xy[6], xy[7]); // hi: {4,6,8}, {10,12,14}
    const uint32_t mc = pkmax3<true>(xy[0], xy[1], xy[2]);                                    // lo: {1,3,5}; mb lo: {11,13,15}
    const uint32_t pre_hi = min(p1 >> 16, p2 & 0xFFFFu);
    const uint32_t pre_lo = max(max(ma >> 16, mb >> 16), max(mc & 0xFFFFu, mb & 0xFFFFu));
    // DF17 (demod.rs:41-44) on the lane's samples 0 .. 9: highs 0, 3, 5, 7, 8; lows 1, 2, 4, 6, 9
    const uint32_t q1 = pkmin(xy[0], xy[4]), q2 = pkmin3<true>(xy[1], xy[2], xy[3]);         // hi: {0, 8} | lo: {3, 5, 7}
    const uint32_t r1 = pkmax3<true>(xy[1], xy[2], xy[3]), r2 = pkmax(xy[0], xy[4]);         // hi: {2, 4, 6} | lo: {1, 9}
    const uint32_t df_hi = min(q1 >> 16, q2 & 0xFFFFu), df_lo = max(r1 >> 16, r2 & 0xFFFFu);
    const unsigned long long pre_ok = __builtin_amdgcn_ballot_w64(pre_hi > pre_lo), df_ok = __builtin_amdgcn_ballot_w64(df_hi > df_lo);
    constexpr unsigned long long k14 = 0x4000400040004000ull, k15 = 0x8000800080008000ull; // lane 14 / 15 of every group
    unc_mask = (~decided & ~(k14 | k15)) | (~pre_ok & k14) | (~df_ok & k15);
    return byte;
}

// [phase:3 exact_from_raw (uncertain survivors only: cold)]
// The reference's own arithmetic on a survivor's 240 samples, by its 16-lane group (ALL 16 lanes active): lane l < 14
// takes frame byte l (samples off + 16 + 16 l .. + 15 of the tile), lane 14 the preamble (samples off .. off + 15), lane 15
// repeats lane 14.  m = floor(sqrt(I^2+Q^2)) as utils.rs:46-52 (v_sqrt_f32 of n + 0.5, truncated: exact for n <= 32768).
// Returns the lane's byte; gate_ok (group-uniform) = the preamble test on lane 14's magnitudes (demod.rs:23-36) and the
// DF17 test on the first ten of lane 0's (demod.rs:45-54).
__device__ __forceinline__ uint32_t exact_from_raw(__amdgpu_buffer_rsrc_t rsrc, const uint32_t off, const uint32_t l, const uint32_t lane, bool &gate_ok)
{
    const uint32_t s = off + (l < 14u ? 16u + 16u * l : 0u);
    const uint32_t a = 2u * s, base = a & ~3u, sh = a & 3u; // (sh = 0 or 2)
    const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base, 0, 0), v1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + 16u, 0, 0);
    const uint32_t v2 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, base + 32u, 0, 0);
    const uint32_t d[9] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2};
    uint32_t m[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t w = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh); // [I, Q, I', Q'] of samples 2k, 2k+1
        const int n0 = __builtin_amdgcn_sdot4((int)(w & 0xFFFFu), (int)w, 0x4B000000, false);
        const int n1 = __builtin_amdgcn_sdot4((int)(w & 0xFFFF0000u), (int)w, 0x4B000000, false);
        m[2 * k] = (uint32_t)__builtin_amdgcn_sqrtf(__builtin_bit_cast(float, n0) - 8388607.5f);
        m[2 * k + 1] = (uint32_t)__builtin_amdgcn_sqrtf(__builtin_bit_cast(float, n1) - 8388607.5f);
    }
    uint32_t byte = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) byte |= (m[2 * k] > m[2 * k + 1] ? 1u : 0u) << (7 - k);
    auto mn = [](uint32_t x, uint32_t y) { return x < y ? x : y; };
    auto mx = [](uint32_t x, uint32_t y) { return x > y ? x : y; };
    const uint32_t hi = mn(mn(m[0], m[2]), mn(m[7], m[9]));
    const uint32_t lo = mx(mx(mx(mx(m[1], m[3]), mx(m[4], m[5])), mx(mx(m[6], m[8]), mx(m[10], m[11]))), mx(mx(m[12], m[13]), mx(m[14], m[15])));
    const uint32_t dh = mn(mn(mn(m[0], m[3]), mn(m[5], m[7])), m[8]);
    const uint32_t dl = mx(mx(mx(m[1], m[2]), mx(m[4], m[6])), m[9]);
    const unsigned long long pm = __builtin_amdgcn_ballot_w64(hi >= lo), dm = __builtin_amdgcn_ballot_w64(dh >= dl);
    const uint32_t g0 = lane & 48u;
    gate_ok = (((pm >> (g0 + 14u)) & (dm >> g0)) & 1ull) != 0;
    return byte;
}

// [phase:end]
__device__ __forceinline__ void scan_tile_code(const DemodArgs &p, const uint32_t tile, const bool first, unsigned char *smem)
{
    typedef CodeLds L;
    uint32_t *img = reinterpret_cast<uint32_t *>(smem);
    uint32_t *cand = reinterpret_cast<uint32_t *>(smem + L::kOffCand);
    uint16_t *list = reinterpret_cast<uint16_t *>(smem + L::kOffList);
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + L::kOffMisc);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const TilePos tp = tile_pos<kTile>(p, tile);
    const uint64_t sample0 = tp.sample0;
    const uint32_t n_valid = tp.n_valid;
    __amdgpu_buffer_rsrc_t rsrc = tile_rsrc<2, kMag>(p, tp, true);
    // [phase:1 code (loads, dots, conversions, stores)]
    u32x4 ra[kCodeFull + 1], rb[kCodeFull + 1];
    code_issue_loads(rsrc, tid, ra, rb);
    if (tid == 0 && first) {
        p.hdr->retry = 0;
        if (p.count_groups) { // first pass of a launch: the finishing kernel ORs this launch's flags in
            p.hdr->flags = 0;
            if (p.hdr_pub) p.hdr_pub[2] = 0;
        }
    }
    if (tid == 0) {
        misc[8] = 0;  // valid-frame counter (tiles without slots only)
        misc[12] = 0; // survivor counter
    }
    code_image_to_lds(ra, rb, img, tid);
    __syncthreads();
#if ADSB_ABL_PHASES < 2
    if (smem[tid * 64] == 0xFD && smem[tid * 64 + 1] == 0xFE && n_valid == 7) misc[12] = 1;
#else
    // [phase:2 code gate]
    gate_phase_code(img, cand + tid, cand + kThreads + tid, tid, n_valid);
    {
        // (word w of the bitmap = offsets 32 w .. 32 w + 31: the lane's run A is word tid, its run B word kThreads + tid)
        uint32_t bitsA = cand[tid], bitsB = cand[kThreads + tid];
        const uint32_t oa = tid * (uint32_t)kRun, ob = oa + (uint32_t)kCodeHalf;
        // offsets at or beyond n_valid do not exist in the reference loop (adsb.rs:98): the ragged last tile of a channel
        if (n_valid < (uint32_t)kTile) { // (wave-uniform)
            const uint32_t va = n_valid > oa ? n_valid - oa : 0u, vb = n_valid > ob ? n_valid - ob : 0u;
            bitsA &= va >= 32u ? 0xFFFFFFFFu : ((1u << va) - 1u);
            bitsB &= vb >= 32u ? 0xFFFFFFFFu : ((1u << vb) - 1u);
            cand[tid] = bitsA; // (the dense path reads the bitmap itself)
            cand[kThreads + tid] = bitsB;
        }
        // survivors are rare (a handful per tile): the few lanes that have any append their offsets, unordered, to the list
        if (bitsA | bitsB) {
            uint32_t pos = atomicAdd(&misc[12], (uint32_t)(__builtin_popcount(bitsA) + __builtin_popcount(bitsB)));
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                uint32_t bits = k ? bitsB : bitsA;
                const uint32_t o0 = k ? ob : oa;
                while (bits) {
                    const uint32_t b = (uint32_t)__builtin_ctz(bits);
                    bits &= bits - 1;
                    if (pos < (uint32_t)kSparseCap) list[pos] = (uint16_t)(o0 + b);
                    ++pos;
                }
            }
        }
    }
#endif
    __syncthreads();

    // [phase:3 hand-over: slots, offsets, sliced bytes]
    // Every survivor of the code gate gets a frame slot, its absolute offset and its 14 sliced bytes; one that the samples
    // themselves reject gets an all-ones offset (finish_order skips it).  CRC-24, repair, ordering: finish_order.
    uint32_t total = p.fused_pass_only ? 0u : misc[12];
#if ADSB_ABL_PHASES < 3
    if (total != 0x7FFFFFFFu) total = 0;
#endif
    const bool dense = total > (uint32_t)kSparseCap;
    u32x4 cw = {0, 0, 0, 0};
    uint32_t cnt = 0, my_first = 0;
    if (dense) { // ordered compaction of the bitmap by workgroup-wide prefix sums; words 4 tid .. 4 tid + 3 per thread
        if (4 * tid < (uint32_t)(kTile / 32)) cw = reinterpret_cast<const u32x4 *>(cand)[tid];
        cnt = __builtin_popcount(cw.x) + __builtin_popcount(cw.y) + __builtin_popcount(cw.z) + __builtin_popcount(cw.w);
        uint32_t incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += t;
        }
        if (lane == 63) misc[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        total = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) {
            uint32_t t = misc[w];
            wbase += (w < (int)wave) ? t : 0u;
            total += t;
        }
        my_first = wbase + incl - cnt;
    }
    const bool simple = !dense && total <= kQuota;
    const uint64_t abs0 = sample0 + p.offset_base; // absolute offset of this tile's offset 0
    uint32_t base_slot = tile * kQuota;
    const uint32_t g = tid >> 4, l = tid & 15;
    // one survivor per 16-lane group: its byte from the codes, or -- gate or some pair uncertain -- from the samples
    auto slice_one = [&](const bool have, const uint32_t off, bool &dropped) {
        unsigned long long um;
        uint32_t byte = code_slice_byte(img, off, l, um);
        um &= __builtin_amdgcn_ballot_w64(have);
        const bool grp_unc = ((um >> (lane & 48u)) & 0xFFFFull) != 0;
        dropped = false;
        if (um != 0) { // (wave-uniform) some group of this wave needs the samples themselves
            bool ok;
            const uint32_t eb = exact_from_raw(rsrc, off, l, lane, ok);
            if (grp_unc) {
                byte = eb;
                dropped = !ok;
            }
        }
        return byte;
    };
    auto slice_round = [&](uint32_t slot0, uint32_t ncl) {
        for (uint32_t r = 0; r < ncl; r += kThreads / 16) {
            if (r + 4 * wave >= ncl) break; // none of this wave's four groups has a survivor
            const uint32_t ci = r + g;
            const bool have = ci < ncl; // uniform within the 16-lane group
            const uint32_t off = have ? list[ci] : 0u;
            bool dropped;
            const uint32_t byte = slice_one(have, off, dropped);
            if (have) {
                unsigned char *rec = reinterpret_cast<unsigned char *>(p.slots + (size_t)slot0 + ci);
                const uint64_t o64 = dropped ? ~0ull : abs0 + off;
                if (l < 14) rec[8 + l] = (unsigned char)byte;
                else reinterpret_cast<uint32_t *>(rec)[l - 14] = l == 14 ? (uint32_t)o64 : (uint32_t)(o64 >> 32);
            }
        }
    };
    if (simple) {
        slice_round(base_slot, total); // unordered list (finish_order ranks it): survivor j -> slot j
    } else {
        if (tid == 0) {
            const unsigned long long b64 = atomicAdd(&p.hdr->alloc, (unsigned long long)total);
            misc[9] = (!p.pool_off && b64 + total <= (unsigned long long)p.cap_slots) ? p.pool_first + (uint32_t)b64 : kNoBase;
        }
        __syncthreads();
        base_slot = misc[9];
        for (uint32_t chunk = 0; chunk < total; chunk += kListCap) {
            if (dense && cnt) {
                uint32_t idx = my_first;
                const uint32_t words[4] = {cw.x, cw.y, cw.z, cw.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t bits = words[k];
                    while (bits) {
                        const uint32_t b = (uint32_t)__builtin_ctz(bits);
                        bits &= bits - 1;
                        if (idx >= chunk && idx < chunk + kListCap) list[idx - chunk] = (uint16_t)((4 * tid + k) * 32 + b);
                        ++idx;
                    }
                }
            }
            __syncthreads();
            const uint32_t ncl = (total - chunk) < (uint32_t)kListCap ? (total - chunk) : (uint32_t)kListCap;
            if (base_slot != kNoBase) {
                slice_round(base_slot + chunk, ncl);
            } else { // the slot store is full: this tile's survivors are decoded here only to be counted
                for (uint32_t r = 0; r < ncl; r += kThreads / 16) {
                    if (r + 4 * wave >= ncl) break;
                    const uint32_t ci = r + g;
                    const bool have = ci < ncl;
                    const uint32_t off = have ? list[ci] : 0u;
                    bool dropped;
                    const uint32_t byte = slice_one(have, off, dropped);
                    const bool valid = count_candidate(have && !dropped, byte, l, lane);
                    if (valid && l == 0) atomicAdd(&misc[8], 1u);
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        Seg e;
        e.base = base_slot;
        e.cand = total;
        e.valid = misc[8];
        e.decoded = base_slot == kNoBase ? 1u : 0u;
        p.seg[tile] = e;
    }
}

#if ADSB_AB_KERNELS
__global__ __launch_bounds__(kThreads, 8) void demod_tiles_code(DemodArgs p)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[CodeLds::kTotal];
    scan_tile_code(p, p.tile_first + tile_of_workgroup(blockIdx.x, p.tile_count), blockIdx.x == 0, smem);
}
#endif

#include "adsb_sieve.inc"

// ---- small buffers: scan + finish in ONE dispatch, results straight into host memory --------------------------------------
// A buffer of at most kFinTiles tiles (the reference's own buffers: 20 000 samples = 2 tiles, adsb.rs:77-79; an SDR's MTU-
// sized reads, adsb.rs:59-64) is not worth three host calls per kernel and a copy each way: one workgroup per tile runs the
// tile body, the workgroup that finishes LAST (a counter in device memory; agent-scope release / acquire around it) runs
// the finishing block over all tiles, writes header and frames through the caller's pointer -- pinned host memory the
// device can write -- and then, behind a system-scope fence, a sequence number the host polls.  The samples are read
// from pinned host memory the same way.  Per buffer the host makes ONE call (the launch).
static_assert(kFinThreads == kThreads, "the small-buffer kernel runs both bodies in one workgroup shape");
template <int ST, int MAGMODE, int SCAN>
__global__ __launch_bounds__(kThreads, 4) void demod_small(DemodArgs p, FinishArgs f, SmallArgs sm)
{
    constexpr int kScanBytes = SCAN == kScanSieve ? SieveLds::kTotal : SCAN == kScanReg ? RegLds::kTotal : SCAN == kScanCode ? CodeLds::kTotal : Lds<ST, (SCAN == kScanReg || SCAN == kScanCode || SCAN == kScanSieve) ? kScanRoot : SCAN>::kTotal, kFinBytes = kFinLdsWords * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kScanBytes > kFinBytes ? kScanBytes : kFinBytes];
    __shared__ uint32_t last_flag;
    if constexpr (SCAN == kScanSieve) scan_tile_sieve(p, p.tile_first + blockIdx.x, blockIdx.x == 0, smem);
    else if constexpr (SCAN == kScanReg) scan_tile_reg(p, p.tile_first + blockIdx.x, blockIdx.x == 0, smem);
    else if constexpr (SCAN == kScanCode) scan_tile_code(p, p.tile_first + blockIdx.x, blockIdx.x == 0, smem);
    else scan_tile<ST, MAGMODE, SCAN>(p, p.tile_first + blockIdx.x, blockIdx.x == 0, smem);
    // hand-off to whichever workgroup arrives last (cdna_hip_programming.md Guideline 16: every storing wave drains its
    // stores, the workgroup's barrier, one lane's agent-scope release, then the counter; the reader acquires)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t prev = __hip_atomic_fetch_add(sm.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = prev == gridDim.x - 1 ? 1u : 0u;
        if (last_flag) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!last_flag) return;
    finish_block<false>(f, 0, 1, reinterpret_cast<uint32_t *>(smem));
    // everything is written (list and header, through f.out / f.hdr_pub): tell the host
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        // the flags this workgroup's lanes OR-ed into the device header, to the host copy by ONE plain store
        if (f.hdr_pub) f.hdr_pub[2] = (uint64_t)__hip_atomic_load(&f.hdr->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *sm.done = 0; // re-armed for the next launch on this result set
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // system scope: the host reads what this kernel wrote
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(sm.seq_host, sm.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_small(hipStream_t s, int sample_type, int mag_mode, int scan, const DemodArgs &p, const FinishArgs &f,
                        const SmallArgs &sm)
{
    if (p.tile_count == 0 || p.tile_count > (uint32_t)kFinTiles) return hipErrorInvalidValue;
    dim3 grid(p.tile_count), block(kThreads);
    if (sample_type == ADSB_SAMPLE_I16) hipLaunchKernelGGL((demod_small<ADSB_SAMPLE_I16, 0, kScanRoot>), grid, block, 0, s, p, f, sm);
#if ADSB_AB_KERNELS
    else if (scan == kScanSieve) hipLaunchKernelGGL((demod_small<ADSB_SAMPLE_I8, 0, kScanSieve>), grid, block, 0, s, p, f, sm);
    else if (scan == kScanNsq) hipLaunchKernelGGL((demod_small<ADSB_SAMPLE_I8, 0, kScanNsq>), grid, block, 0, s, p, f, sm);
    else if (scan == kScanReg) hipLaunchKernelGGL((demod_small<ADSB_SAMPLE_I8, 0, kScanReg>), grid, block, 0, s, p, f, sm);
    else if (scan == kScanCode) hipLaunchKernelGGL((demod_small<ADSB_SAMPLE_I8, 0, kScanCode>), grid, block, 0, s, p, f, sm);
#else
    else if (scan != kScanRoot) return hipErrorInvalidValue; // (the A/B kernels are not in this build)
#endif
    else if (mag_mode == 0) hipLaunchKernelGGL((demod_small<ADSB_SAMPLE_I8, 0, kScanRoot>), grid, block, 0, s, p, f, sm);
    else if (mag_mode == 1) hipLaunchKernelGGL((demod_small<ADSB_SAMPLE_I8, 1, kScanRoot>), grid, block, 0, s, p, f, sm);
    else hipLaunchKernelGGL((demod_small<ADSB_SAMPLE_I8, 2, kScanRoot>), grid, block, 0, s, p, f, sm);
    return hipGetLastError();
}

// No tiles at all (a 240-sample buffer: adsb.rs:98 iterates 0..0), or a measurement launch without the finishing
// kernel: the header of an empty list.
__global__ void empty_result_kernel(Header *hdr, uint64_t *hdr_pub, uint64_t *chan_prefix, uint32_t n_channels)
{
    if (threadIdx.x == 0) {
        hdr->n_out = 0;
        hdr->total_found = 0;
        hdr->flags = 0;
        hdr->retry = 0;
        hdr->alloc = 0;
        if (hdr_pub) { hdr_pub[0] = 0; hdr_pub[1] = 0; hdr_pub[2] = 0; hdr_pub[3] = 0; }
    }
    if (chan_prefix)
        for (uint32_t k = threadIdx.x; k <= n_channels; k += blockDim.x) chan_prefix[k] = 0;
}
hipError_t launch_empty_result(hipStream_t s, Header *hdr, uint64_t *hdr_pub, uint64_t *chan_prefix, uint32_t n_channels,
                               hipEvent_t e0, hipEvent_t e1)
{
    hipExtLaunchKernelGGL(empty_result_kernel, dim3(1), dim3(64), 0, s, e0, e1, 0, hdr, hdr_pub, chan_prefix, n_channels);
    return hipGetLastError();
}

template <int ST>
static hipError_t launch_demod_st(hipStream_t s, int mag_mode, const DemodArgs &a, uint32_t grid_x,
                                  hipEvent_t e0, hipEvent_t e1)
{
    dim3 grid(grid_x), block(kThreads);
    if (ST == ADSB_SAMPLE_I16) mag_mode = 0; // the CS16 magnitude chain does not depend on the converter's rounding
    switch (mag_mode) {
    case 0: hipExtLaunchKernelGGL((demod_tiles<ST, 0, kScanRoot>), grid, block, 0, s, e0, e1, 0, a); break;
    case 1: hipExtLaunchKernelGGL((demod_tiles<ST, ST == ADSB_SAMPLE_I8 ? 1 : 0, kScanRoot>), grid, block, 0, s, e0, e1, 0, a); break;
    default: hipExtLaunchKernelGGL((demod_tiles<ST, ST == ADSB_SAMPLE_I8 ? 2 : 0, kScanRoot>), grid, block, 0, s, e0, e1, 0, a); break;
    }
    return hipGetLastError();
}

bool tile_stamps_built() { return ADSB_TILE_STAMPS != 0; }

hipError_t launch_demod(hipStream_t s, int sample_type, int mag_mode, int scan, const DemodArgs &a,
                        hipEvent_t e0, hipEvent_t e1)
{
    if (a.tile_count == 0) return hipSuccess;
#if ADSB_AB_KERNELS
    if (sample_type == ADSB_SAMPLE_I8 && scan == kScanSieve) {
        uint32_t grid = a.tile_count;
        if (kSievePersistent) { // (A/B) four workgroups per CU, each loops over its share of the tiles
            static int sieve_slots = 0;
            if (sieve_slots == 0) {
                int dev = 0, cus = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
                sieve_slots = cus * 4;
            }
            if (grid > (uint32_t)sieve_slots) grid = (uint32_t)sieve_slots;
        }
        hipExtLaunchKernelGGL(demod_tiles_sieve, dim3(grid), dim3(kThreads), 0, s, e0, e1, 0, a);
        return hipGetLastError();
    }
    if (sample_type == ADSB_SAMPLE_I8 && scan == kScanNsq) {
        hipExtLaunchKernelGGL((demod_tiles<ADSB_SAMPLE_I8, 0, kScanNsq>), dim3(a.tile_count), dim3(kThreads), 0, s, e0, e1, 0, a);
        return hipGetLastError();
    }
    if (sample_type == ADSB_SAMPLE_I8 && scan == kScanReg) {
        hipExtLaunchKernelGGL(demod_tiles_reg, dim3(a.tile_count), dim3(kThreads), 0, s, e0, e1, 0, a);
        return hipGetLastError();
    }
    if (sample_type == ADSB_SAMPLE_I8 && scan == kScanCode) {
        hipExtLaunchKernelGGL(demod_tiles_code, dim3(a.tile_count), dim3(kThreads), 0, s, e0, e1, 0, a);
        return hipGetLastError();
    }
#else
    if (sample_type == ADSB_SAMPLE_I8 && scan != kScanRoot) return hipErrorInvalidValue; // (the A/B kernels are not in this build)
#endif
    if (sample_type == ADSB_SAMPLE_I8) return launch_demod_st<ADSB_SAMPLE_I8>(s, mag_mode, a, a.tile_count, e0, e1);
    return launch_demod_st<ADSB_SAMPLE_I16>(s, 0, a, a.tile_count, e0, e1);
}

// ---- field decode (what AdsbPacket::new computes, src/adsb/packet.rs:25-49) -----------------------
// One thread per frame; 32-byte records.  Integer bit-field work, bound by the 24 + 32 bytes moved
// per frame (a few MB per launch): a latency-bound epilogue, not a hot kernel.
__constant__ char kIcaoCharset[65] = "#ABCDEFGHIJKLMNOPQRSTUVWXYZ#####_###############0123456789######"; // msgs.rs:172-177

__global__ __launch_bounds__(256) void decode_fields_kernel(const adsb_frame *frames, const Header *hdr,
                                                           uint32_t cap, adsb_packet_fields *out)
{
    const uint64_t n64 = hdr->n_out;
    const uint32_t n = n64 < cap ? (uint32_t)n64 : cap;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(frames + i); // bytes[] start at byte 8
    uint8_t b[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t v = w[2 + k];
        b[4 * k] = v & 0xFF; b[4 * k + 1] = (v >> 8) & 0xFF; b[4 * k + 2] = (v >> 16) & 0xFF; b[4 * k + 3] = v >> 24;
    }
    adsb_packet_fields f;
    f.icao = ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
    f.downlink_format = b[0] >> 3;
    f.capability = b[0] & 5;          // sic: the reference masks with 5 (packet.rs:27)
    f.msg_type = b[4] >> 3;
    f.altitude = 0; f.cpr_latitude = 0; f.cpr_longitude = 0;
    f.surveillance_status = 0; f.nic_supplement = 0; f.cpr_time = 0; f.cpr_odd = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) f.callsign[k] = 0;
    const uint8_t *m = b + 4;         // the 7-byte ME field, packet[4..11]
    if (f.msg_type >= 1 && f.msg_type <= 4) {           // msgs.rs:210-212
        f.msg_kind = 0;
        unsigned long long bits = 0;
#pragma unroll
        for (int k = 1; k < 7; ++k) bits = (bits << 8) | m[k];
#pragma unroll
        for (int c = 0; c < 8; ++c) f.callsign[c] = kIcaoCharset[(bits >> (42 - 6 * c)) & 0x3F];
    } else if (f.msg_type >= 9 && f.msg_type <= 18) {   // msgs.rs:122-124
        f.msg_kind = 1;
        const int code = ((int)(m[1] >> 1) << 4) | (m[2] >> 4);
        f.altitude = code * ((m[1] & 1) ? 25 : 100) - 1000;
        f.surveillance_status = (m[0] >> 1) & 3;
        f.nic_supplement = m[0] & 1;
        f.cpr_time = (m[2] >> 3) & 1;
        f.cpr_odd = (m[2] >> 2) & 1;
        f.cpr_latitude = ((uint32_t)(m[2] & 3) << 15) | ((uint32_t)m[3] << 7) | (m[4] >> 1);
        f.cpr_longitude = ((uint32_t)(m[4] & 1) << 16) | ((uint32_t)m[5] << 8) | m[6];
    } else {
        f.msg_kind = 2;
    }
    out[i] = f;
}

hipError_t launch_decode_fields(hipStream_t s, const adsb_frame *frames, const Header *hdr, uint32_t cap,
                                adsb_packet_fields *out)
{
    if (cap == 0) return hipSuccess;
    hipLaunchKernelGGL(decode_fields_kernel, dim3((cap + 255) / 256), dim3(256), 0, s, frames, hdr, cap, out);
    return hipGetLastError();
}

// ---- test / measurement kernels -----------------------------------------------------------------
template <int ST, int MAGMODE>
__global__ void magnitudes_kernel(const void *iq, size_t n, uint16_t *out)
{
    if (MAGMODE == 1) __builtin_amdgcn_s_setreg((1 | (0 << 6) | ((2 - 1) << 11)), 3);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (ST == ADSB_SAMPLE_I8) {
        // 8 samples per thread-step through the same code path as the tile kernel
        const size_t groups = (n + 7) / 8;
        for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
            u32x4 v = {0, 0, 0, 0};
            const uint16_t *src = reinterpret_cast<const uint16_t *>(iq) + g * 8;
            uint16_t tmp[8];
            for (int k = 0; k < 8; ++k) tmp[k] = (g * 8 + k < n) ? src[k] : (uint16_t)0;
            v.x = tmp[0] | ((uint32_t)tmp[1] << 16);
            v.y = tmp[2] | ((uint32_t)tmp[3] << 16);
            v.z = tmp[4] | ((uint32_t)tmp[5] << 16);
            v.w = tmp[6] | ((uint32_t)tmp[7] << 16);
            uint32_t lo, hi;
            mags8_i8<MAGMODE>(v, lo, hi);
            for (int k = 0; k < 8; ++k)
                if (g * 8 + k < n) out[g * 8 + k] = (uint16_t)(((k < 4 ? lo : hi) >> (8 * (k & 3))) & 0xFFu);
        }
    } else {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(iq);
        for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride)
            out[k] = (uint16_t)mag_i16(src[k]);
    }
    if (MAGMODE == 1) __builtin_amdgcn_s_setreg((1 | (0 << 6) | ((2 - 1) << 11)), 0);
}

hipError_t launch_magnitudes(hipStream_t s, int sample_type, int mag_mode, const void *iq, size_t n,
                             uint16_t *out)
{
    if (n == 0) return hipSuccess;
    dim3 grid(1024), block(256);
    if (sample_type == ADSB_SAMPLE_I16) {
        hipLaunchKernelGGL((magnitudes_kernel<ADSB_SAMPLE_I16, 0>), grid, block, 0, s, iq, n, out);
    } else if (mag_mode == 0) {
        hipLaunchKernelGGL((magnitudes_kernel<ADSB_SAMPLE_I8, 0>), grid, block, 0, s, iq, n, out);
    } else if (mag_mode == 1) {
        hipLaunchKernelGGL((magnitudes_kernel<ADSB_SAMPLE_I8, 1>), grid, block, 0, s, iq, n, out);
    } else {
        hipLaunchKernelGGL((magnitudes_kernel<ADSB_SAMPLE_I8, 2>), grid, block, 0, s, iq, n, out);
    }
    return hipGetLastError();
}

// nsq test hook: v = I^2 + Q^2 + 72 of n i8 samples through the scan kernel's own packing code (every group of 8
// samples is packed once as the "A" AND the "B" operand: both halves must agree, else 0xFFFF is reported).
__global__ void nsq_values_kernel(const void *iq, size_t n, uint16_t *out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t groups = (n + 7) / 8;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        const uint16_t *src = reinterpret_cast<const uint16_t *>(iq) + g * 8;
        uint16_t tmp[8];
        for (int k = 0; k < 8; ++k) tmp[k] = (g * 8 + k < n) ? src[k] : (uint16_t)0;
        u32x4 v;
        v.x = tmp[0] | ((uint32_t)tmp[1] << 16);
        v.y = tmp[2] | ((uint32_t)tmp[3] << 16);
        v.z = tmp[4] | ((uint32_t)tmp[5] << 16);
        v.w = tmp[6] | ((uint32_t)tmp[7] << 16);
        uint32_t d[8];
        nsq_pack16(v, v, d);
        for (int k = 0; k < 8; ++k)
            if (g * 8 + k < n) out[g * 8 + k] = (d[k] & 0xFFFFu) == (d[k] >> 16) ? (uint16_t)(d[k] & 0xFFFFu) : (uint16_t)0xFFFFu;
    }
}

hipError_t launch_nsq_values(hipStream_t s, const void *iq, size_t n, uint16_t *out)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(nsq_values_kernel, dim3(1024), dim3(256), 0, s, iq, n, out);
    return hipGetLastError();
}

// Pure streaming read: LOADS loads of 16 bytes per lane, all in flight before the first use, `nt` policy, one workgroup per
// LOADS x 4 KB; the values are only XOR-ed.  What this box's HBM delivers to a kernel that does nothing else with the bytes.  No
// single shape is the fastest on every box and size (tools/ubench/read_shapes.hip, profiles/r04_read_shapes.txt: 6.8-7.1 TB/s
// at 1 GiB, 7.0-7.2 at 16 GiB; the 16-load shape in plain workgroup order, the only one until round 4, is the slowest at 1 GiB
// by 3-4 %), so adsb_time_read_ceiling times three -- shape 0: 4 loads, 1: 8 loads, 2: 16 loads with the chunks dealt to
// workgroups in eight contiguous ranges like the scan's tiles -- and reports the fastest.
template <int LOADS, bool XCD>
__global__ __launch_bounds__(256) void read_only_kernel(const u32x4 *buf, size_t n16, uint32_t n_wg, uint32_t *sink)
{
    const uint32_t b = XCD ? tile_of_workgroup(blockIdx.x, n_wg) : blockIdx.x;
    const size_t base = (size_t)b * (256 * LOADS) + threadIdx.x;
    u32x4 v[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) {
        const size_t i = base + (size_t)k * 256;
        v[k] = i < n16 ? __builtin_nontemporal_load(buf + i) : u32x4{0u, 0u, 0u, 0u};
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < LOADS; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    if (acc == 0x9E3779B9u) *sink = acc; // practically never: keeps the loads alive
}

hipError_t launch_read_only(hipStream_t s, const void *buf, size_t bytes, uint32_t *sink, int shape)
{
    const size_t n16 = bytes / 16;
    if (n16 == 0) return hipSuccess;
    const int loads = shape == 0 ? 4 : shape == 1 ? 8 : 16;
    const size_t per_wg = (size_t)256 * loads;
    const uint32_t n_wg = (uint32_t)((n16 + per_wg - 1) / per_wg);
    const u32x4 *b = reinterpret_cast<const u32x4 *>(buf);
    if (shape == 0) hipLaunchKernelGGL((read_only_kernel<4, false>), dim3(n_wg), dim3(256), 0, s, b, n16, n_wg, sink);
    else if (shape == 1) hipLaunchKernelGGL((read_only_kernel<8, false>), dim3(n_wg), dim3(256), 0, s, b, n16, n_wg, sink);
    else if (shape == 2) hipLaunchKernelGGL((read_only_kernel<16, true>), dim3(n_wg), dim3(256), 0, s, b, n16, n_wg, sink);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// Synthetic source: one thread per sample (untimed; clarity over speed).
template <int ST>
__global__ void synth_kernel(adsb_synth_cfg cfg, uint32_t channel, uint64_t first, size_t n, void *iq)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint64_t k = first + j;
        const uint64_t slot = k / cfg.slot_len;
        int vi, vq;
        adsb_synth::noise_iq(cfg, channel, k, vi, vq);
        // only samples inside a frame's 240-sample span need the slot's frame
        adsb_synth::Slot s;
        adsb_synth::slot_params(cfg, channel, slot, s);
        if (s.present) {
            const uint64_t start = slot * (uint64_t)cfg.slot_len + s.jitter;
            if (k >= start && k < start + 240 && adsb_synth::pulse_at(s.sent, (uint32_t)(k - start))) {
                vi += s.amp_i;
                vq += s.amp_q;
            }
        }
        if (ST == ADSB_SAMPLE_I8) {
            reinterpret_cast<int8_t *>(iq)[2 * j] = (int8_t)adsb_synth::clip8(vi);
            reinterpret_cast<int8_t *>(iq)[2 * j + 1] = (int8_t)adsb_synth::clip8(vq);
        } else {
            int wi = vi << cfg.amp_shift, wq = vq << cfg.amp_shift;
            wi = wi < -32768 ? -32768 : (wi > 32767 ? 32767 : wi);
            wq = wq < -32768 ? -32768 : (wq > 32767 ? 32767 : wq);
            reinterpret_cast<int16_t *>(iq)[2 * j] = (int16_t)wi;
            reinterpret_cast<int16_t *>(iq)[2 * j + 1] = (int16_t)wq;
        }
    }
}

hipError_t launch_synth(hipStream_t s, const adsb_synth_cfg &cfg, int sample_type, uint32_t channel,
                        uint64_t first, size_t n, void *iq)
{
    if (n == 0) return hipSuccess;
    dim3 grid(256 * 16), block(256);
    if (sample_type == ADSB_SAMPLE_I8)
        hipLaunchKernelGGL((synth_kernel<ADSB_SAMPLE_I8>), grid, block, 0, s, cfg, channel, first, n, iq);
    else
        hipLaunchKernelGGL((synth_kernel<ADSB_SAMPLE_I16>), grid, block, 0, s, cfg, channel, first, n, iq);
    return hipGetLastError();
}

} // namespace adsbk
