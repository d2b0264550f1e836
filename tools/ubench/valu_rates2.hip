// valu_rates.hip -- how many cycles does one wave64 VALU instruction occupy a gfx950 SIMD for?
// Each kernel runs a long stream of independent copies of ONE instruction; 1, 2 and 4 waves per
// SIMD.  Output: cycles per wave-instruction per SIMD (= elapsed / (instructions per wave *
// waves per SIMD)).  Used to budget the demod kernel's instruction mix.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define ITERS 1000

#define DEFK(NAME, ASM3)                                                                          \
    __global__ __launch_bounds__(1024) void k_##NAME(uint64_t *out, uint32_t seed)               \
    {                                                                                             \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,   \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed * 31 + 7, c = seed ^ 0x55aa;    \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                               \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                     \
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
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                               \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                     \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345) out[1] = a0;                      \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                \
    }

// each ASM3(R) is one instruction with destination R and sources (R, %8, %9) as appropriate
#define I_and(R) "v_and_b32 " R ", " R ", %8\n"
#define I_pkmax(R) "v_pk_max_u16 " R ", " R ", %8\n"
#define I_pkmin(R) "v_pk_min_u16 " R ", " R ", %8\n"
#define I_perm(R) "v_perm_b32 " R ", " R ", %8, %9\n"
#define I_maxu32(R) "v_max_u32 " R ", " R ", %8\n"
#define I_max3u32(R) "v_max3_u32 " R ", " R ", %8, %9\n"
#define I_maxf32(R) "v_max_f32 " R ", " R ", %8\n"
#define I_max3f32(R) "v_max3_f32 " R ", " R ", %8, %9\n"
#define I_pkmaxf16(R) "v_pk_max_f16 " R ", " R ", %8\n"
#define I_pkmax3f16(R) "v_pk_maximum3_f16 " R ", " R ", %8, %9\n"
#define I_pkmin3f16(R) "v_pk_minimum3_f16 " R ", " R ", %8, %9\n"
#define I_addf32(R) "v_add_f32 " R ", " R ", %8\n"
#define I_fmaf32(R) "v_fma_f32 " R ", " R ", %8, %9\n"
#define I_sqrt(R) "v_sqrt_f32 " R ", " R "\n"
#define I_cvtpku8(R) "v_cvt_pk_u8_f32 " R ", %8, 1, " R "\n"
#define I_dot4(R) "v_dot4_i32_i8 " R ", " R ", %8, %9\n"
#define I_cmpsdwa(R) "v_cmp_ge_u32_sdwa vcc, " R ", %8 src0_sel:WORD_1 src1_sel:WORD_1\n"
#define I_cmpu16(R) "v_cmp_ge_u16 vcc, " R ", %8\n"
#define I_lshlor(R) "v_lshl_or_b32 " R ", " R ", 8, %8\n"
#define I_pksubi16(R) "v_pk_sub_i16 " R ", " R ", %8\n"
#define I_pkaddu16(R) "v_pk_add_u16 " R ", " R ", %8\n"
#define I_pkmullo(R) "v_pk_mul_lo_u16 " R ", " R ", %8\n"
#define I_pkmad(R) "v_pk_mad_u16 " R ", " R ", %8, %9\n"
#define I_sad(R) "v_sad_u8 " R ", " R ", %8, %9\n"
#define I_cvtf32ub(R) "v_cvt_f32_ubyte1 " R ", " R "\n"
#define I_andor(R) "v_and_or_b32 " R ", " R ", %8, %9\n"
#define I_max3u16(R) "v_max3_u16 " R ", " R ", %8, %9\n"
#define I_mov(R) "v_mov_b32 " R ", %8\n"
#define I_pkfmaf16(R) "v_pk_fma_f16 " R ", " R ", %8, %9\n"
#define I_madu32u24(R) "v_mad_u32_u24 " R ", " R ", %8, %9\n"
#define I_mulu32u24(R) "v_mul_u32_u24 " R ", " R ", %8\n"
#define I_bfe(R) "v_bfe_u32 " R ", " R ", 8, 8\n"
#define I_alignbit(R) "v_alignbit_b32 " R ", " R ", %8, 16\n"
#define I_cndmask(R) "v_cndmask_b32 " R ", " R ", %8, vcc\n"
#define I_addco(R) "v_addc_co_u32 " R ", vcc, " R ", " R ", vcc\n"
#define I_minf16(R) "v_min_f16 " R ", " R ", %8\n"
#define I_rsq(R) "v_rsq_f32 " R ", " R "\n"
#define I_cvtf32u32(R) "v_cvt_f32_u32 " R ", " R "\n"
#define I_cvtu32f32(R) "v_cvt_u32_f32 " R ", " R "\n"
#define I_sqrtf16(R) "v_sqrt_f16 " R ", " R "\n"
#define I_pkmovb32(R) "v_pk_mov_b32 " R ", %8, %9\n"


#define I_or(R) "v_or_b32 " R ", " R ", %8\n"
#define I_xor(R) "v_xor_b32 " R ", " R ", %8\n"
#define I_addu32(R) "v_add_u32 " R ", " R ", %8\n"
#define I_subu32(R) "v_sub_u32 " R ", " R ", %8\n"
#define I_lshr(R) "v_lshrrev_b32 " R ", 4, " R "\n"
#define I_lshl(R) "v_lshlrev_b32 " R ", 4, " R "\n"
#define I_maxu16(R) "v_max_u16 " R ", " R ", %8\n"
#define I_minu16(R) "v_min_u16 " R ", " R ", %8\n"
#define I_maxf16(R) "v_max_f16 " R ", " R ", %8\n"
#define I_cmpu32(R) "v_cmp_ge_u32 vcc, " R ", %8\n"
#define I_cmpf16(R) "v_cmp_ge_f16 vcc, " R ", %8\n"
#define I_cmpf32(R) "v_cmp_ge_f32 vcc, " R ", %8\n"
#define I_mulf32(R) "v_mul_f32 " R ", " R ", %8\n"
#define I_subf32(R) "v_sub_f32 " R ", " R ", %8\n"
#define I_pkaddf32(R) "v_pk_add_f32 " R ", " R ", " R "\n"
#define I_bfi(R) "v_bfi_b32 " R ", " R ", %8, %9\n"
#define I_add3(R) "v_add3_u32 " R ", " R ", %8, %9\n"
#define I_or3(R) "v_or3_b32 " R ", " R ", %8, %9\n"
#define I_cndv(R) "v_cndmask_b32 " R ", " R ", %8, vcc\n"
#define I_minf32(R) "v_min_f32 " R ", " R ", %8\n"
#define I_addf16(R) "v_add_f16 " R ", " R ", %8\n"
#define I_pkaddf16(R) "v_pk_add_f16 " R ", " R ", %8\n"
#define I_mix_sqrt_pkmax(R) "v_sqrt_f32 " R ", " R "\nv_pk_max_u16 " R ", " R ", %8\nv_pk_max_u16 " R ", " R ", %9\nv_pk_max_u16 " R ", " R ", %8\n"
#define I_mix_sqrt_and(R) "v_sqrt_f32 " R ", " R "\nv_and_b32 " R ", " R ", %8\nv_and_b32 " R ", " R ", %9\nv_and_b32 " R ", " R ", %8\n"
#define I_mix_pkmax_and(R) "v_pk_max_u16 " R ", " R ", %8\nv_and_b32 " R ", " R ", %9\n"
#define I_mix_pkmax_add(R) "v_pk_max_u16 " R ", " R ", %8\nv_add_f32 " R ", " R ", %9\n"
#define I_cmpsdwab(R) "v_cmp_ge_u32_sdwa vcc, " R ", %8 src0_sel:BYTE_0 src1_sel:BYTE_1\n"
#define I_maxu16sdwa(R) "v_max_u16_sdwa " R ", " R ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
#define I_dot2(R) "v_dot2_i32_i16 " R ", " R ", %8, %9\n"
#define I_dot4u(R) "v_dot4_u32_u8 " R ", " R ", %8, %9\n"
#define I_addsdwapres(R) "v_add_f32_sdwa " R ", " R ", %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
#define I_addsdwapad(R) "v_add_f32_sdwa " R ", " R ", %8 dst_sel:BYTE_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
#define I_addsdwadw(R) "v_add_f32_sdwa " R ", " R ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define I_mix_addpres_and(R) "v_add_f32_sdwa " R ", " R ", %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\nv_and_b32 " R ", " R ", %9\n"
#define I_addsdwapres3(R) "v_add_f32_sdwa " R ", %9, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
#define I_addsdwapad3(R) "v_add_f32_sdwa " R ", %9, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
#define I_mix_pres3_pkmax(R) "v_add_f32_sdwa " R ", %9, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\nv_pk_max_u16 " R ", " R ", %8\n"
#define LIST(X) X(or) X(xor) X(addu32) X(subu32) X(lshr) X(lshl) X(maxu16) X(minu16) X(maxf16) X(cmpu32) \
    X(cmpf16) X(cmpf32) X(mulf32) X(subf32) X(bfi) X(add3) X(or3) X(cndv) X(minf32) X(addf16) X(pkaddf16) \
    X(mix_sqrt_pkmax) X(mix_sqrt_and) X(mix_pkmax_and) X(mix_pkmax_add) X(cmpsdwab) X(maxu16sdwa) X(dot2) X(dot4u) X(addsdwapres) X(addsdwapad) X(addsdwadw) X(mix_addpres_and) X(addsdwapres3) X(addsdwapad3) X(mix_pres3_pkmax)

#define MK(N) DEFK(N, I_##N)
LIST(MK)

typedef void (*kfn)(uint64_t *, uint32_t);
struct Ent { const char *name; kfn fn; };
#define EN(N) {#N, k_##N},
static Ent ents[] = {LIST(EN)};


// Do transcendentals of one wave overlap the ordinary VALU work of ANOTHER wave on the same SIMD?  512 threads:
// waves w and w + 4 share SIMD w.  Waves 0-3 run v_sqrt_f32 only, waves 4-7 v_pk_max_u16 only (same count);
// out[2] / out[3] = cycles of a sqrt wave / a pk_max wave.  Alone: 8 and 4 cycles per instruction.
__global__ __launch_bounds__(512) void k_mixwaves(uint64_t *out, uint32_t seed, int mode)
{
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed * 31 + 7;
    const bool trans = (threadIdx.x >> 8) == 0; // waves 0-3
    const bool idle = (mode == 1 && !trans) || (mode == 2 && trans);
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (!idle) {
        if (trans) {
            for (int i = 0; i < ITERS; ++i)
                asm volatile("v_sqrt_f32 %0, %0\nv_sqrt_f32 %1, %1\nv_sqrt_f32 %2, %2\nv_sqrt_f32 %3, %3\n"
                             "v_sqrt_f32 %0, %0\nv_sqrt_f32 %1, %1\nv_sqrt_f32 %2, %2\nv_sqrt_f32 %3, %3\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else {
            for (int i = 0; i < ITERS; ++i)
                asm volatile("v_pk_max_u16 %0, %0, %4\nv_pk_max_u16 %1, %1, %4\nv_pk_max_u16 %2, %2, %4\nv_pk_max_u16 %3, %3, %4\n"
                             "v_pk_max_u16 %0, %0, %4\nv_pk_max_u16 %1, %1, %4\nv_pk_max_u16 %2, %2, %4\nv_pk_max_u16 %3, %3, %4\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345) out[1] = a0;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[2] = t1 - t0;
    if (blockIdx.x == 0 && threadIdx.x == 256) out[3] = t1 - t0;
}

int main()
{
    uint64_t *d;
    hipMalloc(&d, 64);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, ncu);
    printf("%-12s %10s %10s %10s   (cycles per wave-instruction per SIMD at 1/2/4 waves per SIMD)\n", "instr", "w1", "w2", "w4");
    for (auto &e : ents) {
        double r[3];
        int wi = 0;
        for (int waves : {1, 2, 4}) {
            uint64_t h = 0;
            hipLaunchKernelGGL(e.fn, dim3(ncu), dim3(256 * waves), 0, 0, d, 12345u); // warm
            hipLaunchKernelGGL(e.fn, dim3(ncu), dim3(256 * waves), 0, 0, d, 12345u);
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            r[wi++] = (double)h / ((double)ITERS * 32 * waves);
        }
        // chip-wide: fill every CU with 8 waves per SIMD (2 x 1024-thread blocks), wall clock
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
        double winstr = (double)blocks * 16 * ITERS * 32; // wave-instructions executed
        double per_simd_per_us = winstr / (ncu * 4.0) / (ms * 1e3);
        printf("%-12s %10.2f %10.2f %10.2f   chip: %8.1f wave-instr/us/SIMD (%.2f cycles each at 2.4 GHz)\n", e.name, r[0], r[1], r[2],
               per_simd_per_us, 2400.0 / per_simd_per_us);
    }
    for (int mode = 0; mode < 3; ++mode) {
        uint64_t h[4] = {0, 0, 0, 0};
        hipMemset(d, 0, 64);
        hipLaunchKernelGGL(k_mixwaves, dim3(ncu), dim3(512), 0, 0, d, 12345u, mode);
        hipLaunchKernelGGL(k_mixwaves, dim3(ncu), dim3(512), 0, 0, d, 12345u, mode);
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        printf("mixwaves mode %d (%s): sqrt wave %.2f cycles per instruction, pk_max wave %.2f\n", mode,
               mode == 0 ? "both kinds on every SIMD" : mode == 1 ? "sqrt waves only" : "pk_max waves only",
               (double)h[2] / (ITERS * 8.0), (double)h[3] / (ITERS * 8.0));
    }
    return 0;
}
