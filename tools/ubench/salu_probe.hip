// salu_probe.hip -- issue rates the sieve scan leans on (chip-wide, 8 waves per SIMD, as fp8_probe.hip measures them):
// scalar-ALU throughput alone and beside VALU work, VALU compares writing SGPR pairs, ds_read_u16 beside dot4, and whether
// grouping the 2-cycle VALU class (and / or / add) next to 4-cycle instructions keeps its rate.
// Each BODY is one asm string over: %0-%7 VGPR (in/out), %8 %9 VGPR (in), %10-%13 SGPR pairs (in/out), %14 SGPR pair (in).
// `units` = how many instructions of the kind being priced one BODY holds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 500
#define DEFK(NAME, BODY)                                                                                       \
    __global__ __launch_bounds__(1024) void k_##NAME(uint64_t *out, uint32_t seed)                            \
    {                                                                                                          \
        __shared__ uint32_t lds[4096];                                                                         \
        lds[threadIdx.x] = seed + threadIdx.x;                                                                 \
        lds[threadIdx.x + 1024] = seed;                                                                        \
        __syncthreads();                                                                                       \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13,   \
                 a6 = a0 * 17, a7 = a0 * 19, b = seed * 31 + 7, c = (threadIdx.x * 2u) & 4094u;                \
        unsigned long long p0 = __builtin_amdgcn_readfirstlane(seed) | 1ull, p1 = p0 * 3, p2 = p0 * 5,         \
                           p3 = p0 * 7, q = p0 * 11;                                                           \
        for (int i = 0; i < ITERS; ++i) {                                                                      \
            asm volatile(BODY BODY BODY BODY                                                                   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)      \
                         : "v"(b), "v"(c), "s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(q)                          \
                         : "vcc", "scc", "memory", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",     \
                           "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75");                            \
        }                                                                                                      \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345) out[1] = a0 + p0 + p1 + p2 + p3 + q;           \
    }

#define SAND4 "s_and_b64 s[60:61], s[60:61], %14\ns_and_b64 s[62:63], s[62:63], %14\ns_and_b64 s[64:65], s[64:65], %14\ns_and_b64 s[66:67], s[66:67], %14\n"
#define SAND8 SAND4 "s_and_b64 s[68:69], s[68:69], %14\ns_and_b64 s[70:71], s[70:71], %14\ns_and_b64 s[72:73], s[72:73], %14\ns_and_b64 s[74:75], s[74:75], %14\n"
#define DOT(R) "v_dot4_i32_i8 " R ", " R ", %8, %9\n"
#define AND(R) "v_and_b32 " R ", " R ", %8\n"
#define PKM(R) "v_pk_max_u16 " R ", " R ", %8\n"
#define CMPS(S, R) "v_cmp_ge_f32_e64 " S ", " R ", %8\n"
#define CMPU(S, R) "v_cmp_ge_u32_e64 " S ", " R ", %8\n"

// --- scalar ALU alone
DEFK(sand, SAND8)
// --- VALU (dot4) : SALU 1:1, 1:2, 1:4 -- priced per dot4 (8 per body)
DEFK(dot_sand11, DOT("%0") "s_and_b64 s[60:61], s[60:61], %14\n" DOT("%1") "s_and_b64 s[62:63], s[62:63], %14\n" DOT("%2") "s_and_b64 s[64:65], s[64:65], %14\n"
     DOT("%3") "s_and_b64 s[66:67], s[66:67], %14\n" DOT("%4") "s_and_b64 s[68:69], s[68:69], %14\n" DOT("%5") "s_and_b64 s[70:71], s[70:71], %14\n"
     DOT("%6") "s_and_b64 s[72:73], s[72:73], %14\n" DOT("%7") "s_and_b64 s[74:75], s[74:75], %14\n")
DEFK(dot_sand12, DOT("%0") "s_and_b64 s[60:61], s[60:61], %14\ns_and_b64 s[62:63], s[62:63], %14\n" DOT("%1") "s_and_b64 s[64:65], s[64:65], %14\ns_and_b64 s[66:67], s[66:67], %14\n"
     DOT("%2") "s_and_b64 s[68:69], s[68:69], %14\ns_and_b64 s[70:71], s[70:71], %14\n" DOT("%3") "s_and_b64 s[72:73], s[72:73], %14\ns_and_b64 s[74:75], s[74:75], %14\n"
     DOT("%4") "s_and_b64 s[60:61], s[60:61], %14\ns_and_b64 s[62:63], s[62:63], %14\n" DOT("%5") "s_and_b64 s[64:65], s[64:65], %14\ns_and_b64 s[66:67], s[66:67], %14\n"
     DOT("%6") "s_and_b64 s[68:69], s[68:69], %14\ns_and_b64 s[70:71], s[70:71], %14\n" DOT("%7") "s_and_b64 s[72:73], s[72:73], %14\ns_and_b64 s[74:75], s[74:75], %14\n")
DEFK(dot_sand14, DOT("%0") SAND4 DOT("%1") SAND4 DOT("%2") SAND4 DOT("%3") SAND4 DOT("%4") SAND4 DOT("%5") SAND4 DOT("%6") SAND4 DOT("%7") SAND4)
DEFK(dot_sand18, DOT("%0") SAND8 DOT("%1") SAND8 DOT("%2") SAND8 DOT("%3") SAND8 DOT("%4") SAND8 DOT("%5") SAND8 DOT("%6") SAND8 DOT("%7") SAND8)
// --- compares into SGPR pairs (8 per body)
DEFK(cmpf_s, CMPS("s[60:61]", "%0") CMPS("s[62:63]", "%1") CMPS("s[64:65]", "%2") CMPS("s[66:67]", "%3") CMPS("s[68:69]", "%4") CMPS("s[70:71]", "%5") CMPS("s[72:73]", "%6") CMPS("s[74:75]", "%7"))
DEFK(cmpu_s, CMPU("s[60:61]", "%0") CMPU("s[62:63]", "%1") CMPU("s[64:65]", "%2") CMPU("s[66:67]", "%3") CMPU("s[68:69]", "%4") CMPU("s[70:71]", "%5") CMPU("s[72:73]", "%6") CMPU("s[74:75]", "%7"))
// --- the sieve's inner shape: per position 1 dot4 + 2 compares + N dependent s_and (the compare results feed the ands)
#define POS(R, N_ANDS) DOT(R) CMPS("s[60:61]", R) CMPS("s[62:63]", R) N_ANDS
#define A2 "s_and_b64 s[64:65], s[60:61], s[62:63]\ns_and_b64 s[66:67], s[66:67], s[64:65]\n"
#define A4 A2 "s_and_b64 s[68:69], s[68:69], s[60:61]\ns_and_b64 s[70:71], s[70:71], s[62:63]\n"
#define A8 A4 "s_and_b64 s[72:73], s[72:73], s[60:61]\ns_and_b64 s[74:75], s[74:75], s[62:63]\ns_and_b64 s[64:65], s[64:65], s[66:67]\ns_and_b64 s[68:69], s[68:69], s[70:71]\n"
#define A12 A8 A4
#define A16 A8 A8
DEFK(pos_a2, POS("%0", A2) POS("%1", A2) POS("%2", A2) POS("%3", A2) POS("%4", A2) POS("%5", A2) POS("%6", A2) POS("%7", A2))
DEFK(pos_a4, POS("%0", A4) POS("%1", A4) POS("%2", A4) POS("%3", A4) POS("%4", A4) POS("%5", A4) POS("%6", A4) POS("%7", A4))
DEFK(pos_a8, POS("%0", A8) POS("%1", A8) POS("%2", A8) POS("%3", A8) POS("%4", A8) POS("%5", A8) POS("%6", A8) POS("%7", A8))
DEFK(pos_a12, POS("%0", A12) POS("%1", A12) POS("%2", A12) POS("%3", A12) POS("%4", A12) POS("%5", A12) POS("%6", A12) POS("%7", A12))
DEFK(pos_a16, POS("%0", A16) POS("%1", A16) POS("%2", A16) POS("%3", A16) POS("%4", A16) POS("%5", A16) POS("%6", A16) POS("%7", A16))
// the same with a never-taken branch per position (s_cbranch_scc1 after the last and: scc = result non-zero; seed makes it zero)
#define BR "s_cmp_eq_u64 s[66:67], 0x7b\ns_cbranch_scc1 1f\n"
DEFK(pos_a8_br, POS("%0", A8 BR) POS("%1", A8 BR) POS("%2", A8 BR) POS("%3", A8 BR) POS("%4", A8 BR) POS("%5", A8 BR) POS("%6", A8 BR) POS("%7", A8 BR) "1:\n")
// --- ds_read_u16 feeding dot4 (8 per body); %9 = byte address (even, < 8188)
#define LD(R) "ds_read_u16 " R ", %9 offset:%c0\n"
DEFK(ldsu16_dot, "ds_read_u16 %0, %9\nds_read_u16 %1, %9 offset:2\nds_read_u16 %2, %9 offset:4\nds_read_u16 %3, %9 offset:6\n"
     "ds_read_u16 %4, %9 offset:8\nds_read_u16 %5, %9 offset:10\nds_read_u16 %6, %9 offset:12\nds_read_u16 %7, %9 offset:14\ns_waitcnt lgkmcnt(0)\n"
     "v_dot4_i32_i8 %0, %0, %0, %8\nv_dot4_i32_i8 %1, %1, %1, %8\nv_dot4_i32_i8 %2, %2, %2, %8\nv_dot4_i32_i8 %3, %3, %3, %8\n"
     "v_dot4_i32_i8 %4, %4, %4, %8\nv_dot4_i32_i8 %5, %5, %5, %8\nv_dot4_i32_i8 %6, %6, %6, %8\nv_dot4_i32_i8 %7, %7, %7, %8\n")
DEFK(ldsu16, "ds_read_u16 %0, %9\nds_read_u16 %1, %9 offset:2\nds_read_u16 %2, %9 offset:4\nds_read_u16 %3, %9 offset:6\n"
     "ds_read_u16 %4, %9 offset:8\nds_read_u16 %5, %9 offset:10\nds_read_u16 %6, %9 offset:12\nds_read_u16 %7, %9 offset:14\ns_waitcnt lgkmcnt(0)\n")
// --- grouping of the 2-cycle class next to 4-cycle instructions (16 VALU per body)
DEFK(alt_and_pk, AND("%0") PKM("%1") AND("%2") PKM("%3") AND("%4") PKM("%5") AND("%6") PKM("%7") AND("%1") PKM("%0") AND("%3") PKM("%2") AND("%5") PKM("%4") AND("%7") PKM("%6"))
DEFK(grp2_and_pk, AND("%0") AND("%2") PKM("%1") PKM("%3") AND("%4") AND("%6") PKM("%5") PKM("%7") AND("%1") AND("%3") PKM("%0") PKM("%2") AND("%5") AND("%7") PKM("%4") PKM("%6"))
DEFK(grp4_and_pk, AND("%0") AND("%2") AND("%4") AND("%6") PKM("%1") PKM("%3") PKM("%5") PKM("%7") AND("%1") AND("%3") AND("%5") AND("%7") PKM("%0") PKM("%2") PKM("%4") PKM("%6"))
DEFK(grp8_and_pk, AND("%0") AND("%1") AND("%2") AND("%3") AND("%4") AND("%5") AND("%6") AND("%7") PKM("%0") PKM("%1") PKM("%2") PKM("%3") PKM("%4") PKM("%5") PKM("%6") PKM("%7"))
DEFK(and16, AND("%0") AND("%1") AND("%2") AND("%3") AND("%4") AND("%5") AND("%6") AND("%7") AND("%0") AND("%1") AND("%2") AND("%3") AND("%4") AND("%5") AND("%6") AND("%7"))
DEFK(pk16, PKM("%0") PKM("%1") PKM("%2") PKM("%3") PKM("%4") PKM("%5") PKM("%6") PKM("%7") PKM("%0") PKM("%1") PKM("%2") PKM("%3") PKM("%4") PKM("%5") PKM("%6") PKM("%7"))
// --- v_cmp (VOPC, vcc) + v_addc (shift-in): the VGPR shift-register form, 8 pairs per body
#define CA(R) "v_cmp_ge_f32_e32 vcc, " R ", %8\nv_addc_co_u32_e32 " R ", vcc, " R ", " R ", vcc\n"
DEFK(cmp_addc, CA("%0") CA("%1") CA("%2") CA("%3") CA("%4") CA("%5") CA("%6") CA("%7"))

typedef void (*kfn)(uint64_t *, uint32_t);
struct Ent { const char *name; kfn fn; int units; const char *what; };
static Ent ents[] = {
    {"sand", k_sand, 8, "s_and_b64 alone"},
    {"dot_sand11", k_dot_sand11, 8, "per dot4, 1 s_and each"},
    {"dot_sand12", k_dot_sand12, 8, "per dot4, 2 s_and each"},
    {"dot_sand14", k_dot_sand14, 8, "per dot4, 4 s_and each"},
    {"dot_sand18", k_dot_sand18, 8, "per dot4, 8 s_and each"},
    {"cmpf_s", k_cmpf_s, 8, "v_cmp_ge_f32 -> sgpr pair"},
    {"cmpu_s", k_cmpu_s, 8, "v_cmp_ge_u32 -> sgpr pair"},
    {"pos_a2", k_pos_a2, 8, "per position: dot4 + 2 cmp + 2 s_and"},
    {"pos_a4", k_pos_a4, 8, "per position: dot4 + 2 cmp + 4 s_and"},
    {"pos_a8", k_pos_a8, 8, "per position: dot4 + 2 cmp + 8 s_and"},
    {"pos_a12", k_pos_a12, 8, "per position: dot4 + 2 cmp + 12 s_and"},
    {"pos_a16", k_pos_a16, 8, "per position: dot4 + 2 cmp + 16 s_and"},
    {"pos_a8_br", k_pos_a8_br, 8, "per position: dot4 + 2 cmp + 8 s_and + s_cmp + branch (not taken)"},
    {"ldsu16", k_ldsu16, 8, "ds_read_u16 alone"},
    {"ldsu16_dot", k_ldsu16_dot, 8, "per (ds_read_u16 + dot4)"},
    {"and16", k_and16, 16, "v_and"},
    {"pk16", k_pk16, 16, "v_pk_max_u16"},
    {"alt_and_pk", k_alt_and_pk, 16, "and, pk alternating (per instruction)"},
    {"grp2_and_pk", k_grp2_and_pk, 16, "and x2, pk x2 (per instruction)"},
    {"grp4_and_pk", k_grp4_and_pk, 16, "and x4, pk x4 (per instruction)"},
    {"grp8_and_pk", k_grp8_and_pk, 16, "and x8, pk x8 (per instruction)"},
    {"cmp_addc", k_cmp_addc, 8, "per (v_cmp vcc + v_addc)"},
};

int main(int argc, char **argv)
{
    uint64_t *d;
    hipMalloc(&d, 64);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    for (int wps = 8; wps >= 4; wps -= 4) {
        printf("---- %d waves per SIMD (%d blocks of 1024 threads per CU)\n", wps, wps / 4);
        for (auto &e : ents) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            const int blocks = ncu * (wps / 4) * 8;
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(1024), 0, 0, d, 12345u);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(1024), 0, 0, d, 12345u);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double units = (double)blocks * 16 * ITERS * 4 * e.units;
            const double per_simd_per_us = units / (ncu * 4.0) / (ms * 1e3);
            printf("%-12s %8.1f units/us/SIMD  %6.2f cycles per unit at 2.4 GHz   (%s)\n", e.name, per_simd_per_us,
                   2400.0 / per_simd_per_us, e.what);
            if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return 1; }
        }
    }
    return 0;
}
