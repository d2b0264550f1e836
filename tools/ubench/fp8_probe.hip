// fp8_probe.hip -- (1) the 8-bit code the code scan would store for every n = I^2+Q^2 in 0..32768:
// e4m3(fma(float_bits(2^23 + n), s, t)) through v_cvt_pk_fp8_f32 itself; (2) issue rates of the instructions the
// code scan's phase 1 / gate lean on (chip-wide, 8 waves per SIMD, as valu_rates2.hip measures them).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>

__global__ void codes_kernel(uint8_t *out, float s, float t)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n > 32768) return;
    const float f = __builtin_bit_cast(float, 0x4B000000u + n); // 2^23 + n
    const float x = __builtin_fmaf(f, s, t);
    uint32_t r = __builtin_amdgcn_cvt_pk_fp8_f32(x, 0.0f, 0u, false);
    out[n] = (uint8_t)(r & 0xFFu);
}

#define ITERS 1000
#define DEFK(NAME, ASM3)                                                                          \
    __global__ __launch_bounds__(1024) void k_##NAME(uint64_t *out, uint32_t seed)               \
    {                                                                                             \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,   \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed * 31 + 7, c = seed ^ 0x55aa;    \
        unsigned long long p0 = a0, p1 = a1, p2 = a2, p3 = a3, q = b;                             \
        for (int i = 0; i < ITERS; ++i) {                                                         \
            asm volatile(ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5")        \
                         ASM3("%6") ASM3("%7")                                                    \
                         ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5")        \
                         ASM3("%6") ASM3("%7")                                                    \
                         ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5")        \
                         ASM3("%6") ASM3("%7")                                                    \
                         ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5")        \
                         ASM3("%6") ASM3("%7")                                                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),   \
                           "+v"(a7)                                                               \
                         : "v"(b), "v"(c)                                                         \
                         : "vcc");                                                                \
        }                                                                                         \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345) out[1] = a0 + p0 + p1 + p2 + p3 + q; \
    }
// 64-bit operands (packed f32): register PAIRS
#define DEFK2(NAME, ASM3)                                                                         \
    __global__ __launch_bounds__(1024) void k_##NAME(uint64_t *out, uint32_t seed)               \
    {                                                                                             \
        unsigned long long a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed * 31 + 7, \
                           c = seed ^ 0x55aa;                                                     \
        for (int i = 0; i < ITERS; ++i) {                                                         \
            asm volatile(ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") \
                         ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") \
                         ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") \
                         ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                                  \
                         : "v"(b), "v"(c));                                                       \
        }                                                                                         \
        if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345) out[1] = a0;                                          \
    }

#define I_cvtpkfp8(R) "v_cvt_pk_fp8_f32 " R ", " R ", %8\n"
#define I_cvtpkfp8hi(R) "v_cvt_pk_fp8_f32 " R ", " R ", %8 op_sel:[0,0,1]\n"
#define I_cvtpkbf8(R) "v_cvt_pk_bf8_f32 " R ", " R ", %8\n"
#define I_cvtpkrtz(R) "v_cvt_pkrtz_f16_f32 " R ", " R ", %8\n"
#define I_cvtpknormu16(R) "v_cvt_pknorm_u16_f32 " R ", " R ", %8\n"
#define I_lshl8(R) "v_lshlrev_b32 " R ", 8, " R "\n"
#define I_lshladd(R) "v_lshl_add_u32 " R ", " R ", 8, %8\n"
#define I_pkaddu16(R) "v_pk_add_u16 " R ", " R ", %8\n"
#define I_pkmax3f16(R) "v_pk_maximum3_f16 " R ", " R ", %8, %9\n"
#define I_pkminf16(R) "v_pk_min_f16 " R ", " R ", %8\n"
#define I_pkminu16(R) "v_pk_min_u16 " R ", " R ", %8\n"
#define I_cmpsdwab1(R) "v_cmp_ge_u32_sdwa vcc, " R ", %8 src0_sel:BYTE_1 src1_sel:BYTE_1\n"
#define I_and(R) "v_and_b32 " R ", " R ", %8\n"
#define I_dot4(R) "v_dot4_i32_i8 " R ", " R ", %8, %9\n"
#define I_sqrt(R) "v_sqrt_f32 " R ", " R "\n"
#define I_cvtpku8(R) "v_cvt_pk_u8_f32 " R ", %8, 1, " R "\n"
#define I_mix_cvt_and(R) "v_cvt_pk_fp8_f32 " R ", " R ", %8\nv_and_b32 " R ", " R ", %9\n"
#define I_mix_dot4_and(R) "v_dot4_i32_i8 " R ", " R ", %8, %9\nv_and_b32 " R ", " R ", %9\n"
#define I_mix_pkmax3_and(R) "v_pk_maximum3_f16 " R ", " R ", %8, %9\nv_and_b32 " R ", " R ", %9\n"
#define I_mix_pkmax3_lshr(R) "v_pk_maximum3_f16 " R ", " R ", %8, %9\nv_lshrrev_b32 " R ", 8, " R "\n"
#define I_mix_and_add(R) "v_and_b32 " R ", " R ", %8\nv_add_u32 " R ", " R ", %9\n"
#define I_pkfmaf32(R) "v_pk_fma_f32 " R ", " R ", %4, %5\n"
#define I_pkmulf32(R) "v_pk_mul_f32 " R ", " R ", %4\n"
#define I_pkaddf32(R) "v_pk_add_f32 " R ", " R ", %4\n"
#define LIST(X) X(cvtpkfp8) X(cvtpkfp8hi) X(cvtpkbf8) X(cvtpkrtz) X(cvtpknormu16) X(lshl8) X(lshladd) X(pkaddu16) X(pkmax3f16) X(pkminf16) \
    X(pkminu16) X(cmpsdwab1) X(and) X(dot4) X(sqrt) X(cvtpku8) X(mix_cvt_and) X(mix_dot4_and) X(mix_pkmax3_and) X(mix_pkmax3_lshr) X(mix_and_add)
#define LIST2(X) X(pkfmaf32) X(pkmulf32) X(pkaddf32)
#define MK(N) DEFK(N, I_##N)
LIST(MK)
#define MK2(N) DEFK2(N, I_##N)
LIST2(MK2)
typedef void (*kfn)(uint64_t *, uint32_t);
struct Ent { const char *name; kfn fn; int per; };
#define EN(N) {#N, k_##N, 32},
#define EN2(N) {#N, k_##N, 32},
static Ent ents[] = {LIST(EN) LIST2(EN2)};

int main()
{
    uint8_t *dcodes;
    hipMalloc(&dcodes, 32800);
    std::vector<uint8_t> codes(32769);
    const struct { int B; int sc; } cfgs[] = {{256, 8}, {272, 8}, {288, 8}, {240, 8}, {224, 8}, {192, 8}, {128, 8}, {64, 7}};
    for (auto &c : cfgs) {
        const float s = ldexpf(1.0f, -c.sc), t = -(8388608.0f - (float)c.B) * s;
        hipLaunchKernelGGL(codes_kernel, dim3(129), dim3(256), 0, 0, dcodes, s, t);
        hipMemcpy(codes.data(), dcodes, 32769, hipMemcpyDeviceToHost);
        int mono = 1, span = 0, cmin = 255, cmax = 0;
        for (int n = 1; n <= 32768; ++n) mono &= codes[n] >= codes[n - 1];
        for (int k = 0; k <= 181; ++k) {
            int hi = (k + 1) * (k + 1) - 1;
            if (hi > 32768) hi = 32768;
            int d = codes[hi] - codes[k * k];
            if (d > span) span = d;
        }
        for (int n = 0; n <= 32768; ++n) { if (codes[n] < cmin) cmin = codes[n]; if (codes[n] > cmax) cmax = codes[n]; }
        // host emulation: OCP e4m3fn, round to nearest even
        int mism = 0;
        for (int n = 0; n <= 32768; ++n) {
            double x = ((double)n + c.B) * ldexp(1.0, -c.sc);
            int e = (int)floor(log2(x));
            if (e < -6) e = -6;
            double q = nearbyint(x / ldexp(1.0, e - 3));
            if (q >= 16) { q /= 2; e += 1; }
            int code = q >= 8 ? ((e + 7) << 3) + ((int)q - 8) : (int)q;
            mism += code != codes[n];
        }
        printf("B=%d scale=2^-%d: codes %d..%d monotone=%d max codes spanned by one m-interval minus one=%d host-emulation mismatches=%d  c(0)=%d c(128)=%d c(1024)=%d c(32768)=%d\n",
               c.B, c.sc, cmin, cmax, mono, span, mism, codes[0], codes[128], codes[1024], codes[32768]);
    }
    uint64_t *d;
    hipMalloc(&d, 64);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    for (auto &e : ents) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int blocks = ncu * 2 * 8;
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(1024), 0, 0, d, 12345u);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(1024), 0, 0, d, 12345u);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double winstr = (double)blocks * 16 * ITERS * e.per;
        const double per_simd_per_us = winstr / (ncu * 4.0) / (ms * 1e3);
        printf("%-16s chip: %8.1f asm-units/us/SIMD (%.2f cycles each at 2.4 GHz)\n", e.name, per_simd_per_us, 2400.0 / per_simd_per_us);
    }
    return 0;
}
