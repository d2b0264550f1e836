// lut_probe.hip -- can floor(sqrt(I^2+Q^2)) of i8 IQ come from an LDS table at less VALU cost than the root?
//  1. semantics of v_mqsad_pk_u16_u8 with the reference 0x00000080 (one instruction -> 128-|b| for four
//     signed bytes) and whether ds_read_u8_d16_hi keeps the low half of its destination on this device
//  2. issue cost of the index/merge instructions
//  3. LDS byte-gather rate (cycles per wave-instruction per CU, 16 waves) for key streams taken from a real
//     sample file: key A = I^2+Q^2 (32 KB table), key B = (128-|I|)*129 + (128-|Q|) (16.6 KB table)
// usage: lut_probe IQ_FILE(i8 interleaved)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void k_sem(const uint32_t *in, uint32_t *out, int n)
{
    __shared__ uint8_t t[256];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    t[threadIdx.x & 255] = (uint8_t)(threadIdx.x * 3 + 1);
    __syncthreads();
    if (i >= n) return;
    uint64_t s0 = in[i]; // high dword 0
    uint64_t acc = 0, d;
    uint32_t ref = 0x00000080u;
    asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %3" : "=&v"(d) : "v"(s0), "v"(ref), "v"(acc));
    out[4 * i + 0] = (uint32_t)d;
    out[4 * i + 1] = (uint32_t)(d >> 32);
    uint32_t m;
    asm volatile("v_msad_u8 %0, %1, %2, 0" : "=v"(m) : "v"(in[i]), "v"(0x00008000u));
    out[4 * i + 2] = m;
    // d16_hi: does the low half survive?
    uint32_t r = 0x00001234u, a0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)t; // LDS address of t[0]
    uint32_t a = a0 + (threadIdx.x & 255);
    asm volatile("ds_read_u8_d16_hi %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(a));
    out[4 * i + 3] = r;
}

#define ITERS 400
#define DEFK(NAME, BODY, DECL)                                                                    \
    __global__ __launch_bounds__(1024) void k_##NAME(uint64_t *out, uint32_t seed)               \
    {                                                                                             \
        DECL                                                                                      \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                               \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                     \
        for (int i = 0; i < ITERS; ++i) { BODY BODY BODY BODY }                                   \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                               \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                     \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                \
    }

#define DECL32 uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed * 31 + 7, c = seed ^ 0x55aa; \
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, w0 = a4, w1 = a5, w2 = a6, w3 = a7;
#define B8(I) asm volatile(I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define B4Q(I) asm volatile(I("%0", "%4") I("%1", "%5") I("%2", "%6") I("%3", "%7") I("%0", "%4") I("%1", "%5") I("%2", "%6") I("%3", "%7") : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(b));

#define I_and(R) "v_and_b32 " R ", " R ", %8\n"
#define I_msad(R) "v_msad_u8 " R ", " R ", %8, %9\n"
#define I_dot2(R) "v_dot2_i32_i16 " R ", " R ", %8, %9\n"
#define I_dot4(R) "v_dot4_i32_i8 " R ", " R ", %8, %9\n"
#define I_lshlor(R) "v_lshl_or_b32 " R ", " R ", 8, %8\n"
#define I_perm(R) "v_perm_b32 " R ", " R ", %8, %9\n"
#define I_mad24(R) "v_mad_u32_u24 " R ", " R ", %8, %9\n"
#define I_mqsad(D, S) "v_mqsad_pk_u16_u8 " D ", " S ", %8, " D "\n"
#define I_qsad(D, S) "v_qsad_pk_u16_u8 " D ", " S ", %8, " D "\n"

DEFK(and, B8(I_and), DECL32)
DEFK(msad, B8(I_msad), DECL32)
DEFK(dot2, B8(I_dot2), DECL32)
DEFK(dot4, B8(I_dot4), DECL32)
DEFK(lshlor, B8(I_lshlor), DECL32)
DEFK(perm, B8(I_perm), DECL32)
DEFK(mad24, B8(I_mad24), DECL32)
DEFK(mqsad, B4Q(I_mqsad), DECL32)
DEFK(qsad, B4Q(I_qsad), DECL32)

// LDS byte gather: each lane holds 8 keys (from the sample file), table of `tab_bytes` in LDS; 32 reads per iteration
template <int WIDE>
__global__ __launch_bounds__(1024) void k_gather(const uint32_t *keys, uint64_t *out, uint32_t tab_bytes, int iters)
{
    extern __shared__ uint8_t tab[];
    for (uint32_t i = threadIdx.x; i < tab_bytes; i += blockDim.x) tab[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    uint32_t k[8];
    for (int j = 0; j < 8; ++j) k[j] = keys[(blockIdx.x * blockDim.x + threadIdx.x) * 8 + j];
    uint32_t s = 0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    for (int i = 0; i < iters; ++i) {
        uint32_t r[8];
        for (int j = 0; j < 8; ++j) {
            if (WIDE) asm volatile("ds_read_b32 %0, %1" : "=v"(r[j]) : "v"(k[j] & ~3u));
            else asm volatile("ds_read_u8 %0, %1" : "=v"(r[j]) : "v"(k[j]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
        for (int j = 0; j < 8; ++j) s += r[j];
        // walk the key set so that successive iterations do not repeat the very same bank pattern
        uint32_t t = k[0];
        for (int j = 0; j < 7; ++j) k[j] = k[j + 1];
        k[7] = t;
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (s == 0x12345678u) out[1] = s;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}

static double run_rate(void (*k)(uint64_t *, uint32_t), int waves_per_simd, uint64_t *dout, int instr_per_body)
{
    uint64_t h = 0;
    hipLaunchKernelGGL(k, dim3(1), dim3(256 * waves_per_simd), 0, 0, dout, 12345u);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k, dim3(1), dim3(256 * waves_per_simd), 0, 0, dout, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&h, dout, 8, hipMemcpyDeviceToHost));
    // s_memtime counts at 100 MHz-ish?  report raw ticks per instruction and let the `and` row calibrate
    return (double)h / ((double)ITERS * 4 * instr_per_body * waves_per_simd);
}

int main(int argc, char **argv)
{
    uint64_t *dout;
    CHECK(hipMalloc(&dout, 64));
    // ---- 1. semantics
    {
        const int n = 65536;
        std::vector<uint32_t> in(n), out(4 * n);
        for (int i = 0; i < n; ++i) in[i] = (uint32_t)i * 2654435761u ^ ((uint32_t)i << 16);
        in[0] = 0x80808080u; in[1] = 0x7F7F7F7Fu; in[2] = 0; in[3] = 0x0180FF7Fu;
        uint32_t *din, *dres;
        CHECK(hipMalloc(&din, n * 4)); CHECK(hipMalloc(&dres, n * 16));
        CHECK(hipMemcpy(din, in.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_sem, dim3(n / 256), dim3(256), 0, 0, din, dres, n);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(out.data(), dres, n * 16, hipMemcpyDeviceToHost));
        int bad_mq = 0, bad_ms = 0, keep = 0, zero = 0, other = 0;
        for (int i = 0; i < n; ++i) {
            int8_t b[4]; for (int j = 0; j < 4; ++j) b[j] = (int8_t)(in[i] >> (8 * j));
            uint16_t e[4]; for (int j = 0; j < 4; ++j) e[j] = (uint16_t)(128 - abs((int)b[j]));
            uint32_t lo = e[0] | ((uint32_t)e[1] << 16), hi = e[2] | ((uint32_t)e[3] << 16);
            if (out[4 * i] != lo || out[4 * i + 1] != hi) { if (bad_mq++ < 3) printf("mqsad in %08x got %08x %08x want %08x %08x\n", in[i], out[4 * i], out[4 * i + 1], lo, hi); }
            if (out[4 * i + 2] != e[1]) { if (bad_ms++ < 3) printf("msad in %08x got %u want %u\n", in[i], out[4 * i + 2], e[1]); }
            uint32_t tv = (uint8_t)(((i & 255)) * 3 + 1);
            uint32_t r = out[4 * i + 3];
            if (r == ((tv << 16) | 0x1234u)) keep++; else if (r == (tv << 16)) zero++; else { if (other++ < 3) printf("d16_hi got %08x tv %02x\n", r, tv); }
        }
        printf("semantics: v_mqsad_pk_u16_u8(x, 0x80) == 128-|b_i|: %d mismatches; v_msad_u8: %d mismatches\n", bad_mq, bad_ms);
        printf("ds_read_u8_d16_hi: low half kept %d, zeroed %d, other %d of %d\n", keep, zero, other, n);
    }
    // ---- 2. issue rates (ticks per wave-instruction per SIMD; s_memtime ticks, `and` = 4 shader cycles)
    struct { const char *name; void (*k)(uint64_t *, uint32_t); int per_body; } ks[] = {
        {"v_and_b32", k_and, 8}, {"v_msad_u8", k_msad, 8}, {"v_dot2_i32_i16", k_dot2, 8}, {"v_dot4_i32_i8", k_dot4, 8},
        {"v_lshl_or_b32", k_lshlor, 8}, {"v_perm_b32", k_perm, 8}, {"v_mad_u32_u24", k_mad24, 8},
        {"v_mqsad_pk_u16_u8", k_mqsad, 8}, {"v_qsad_pk_u16_u8", k_qsad, 8}};
    double base = 0;
    for (auto &e : ks) {
        double r1 = run_rate(e.k, 1, dout, e.per_body), r4 = run_rate(e.k, 4, dout, e.per_body);
        if (base == 0) base = r4;
        printf("rate %-20s 1 wave/SIMD %.3f ticks, 4 waves/SIMD %.3f ticks  (= %.2f x v_and)\n", e.name, r1, r4, r4 / base);
    }
    // ---- 3. gather
    if (argc > 1) {
        FILE *f = fopen(argv[1], "rb");
        if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
        const int nkeys = 1024 * 8;
        std::vector<int8_t> iq(2 * nkeys * 64);
        size_t got = fread(iq.data(), 1, iq.size(), f);
        fclose(f);
        printf("read %zu bytes of IQ\n", got);
        uint32_t *dk; CHECK(hipMalloc(&dk, nkeys * 4));
        // lane l of wave w reads samples the way phase 1 does: one 16-byte load = 8 consecutive samples per lane
        for (int region = 0; region < 8; ++region) {
            for (int scheme = 0; scheme < 3; ++scheme) {
                std::vector<uint32_t> keys(nkeys);
                uint32_t tab = scheme == 0 ? 32772 : scheme == 1 ? 16644 : 16644;
                size_t s0 = (size_t)region * nkeys; // sample index of this region's first key
                for (int t = 0; t < 1024; ++t) for (int j = 0; j < 8; ++j) {
                    size_t s = s0 + (size_t)t * 8 + j;
                    int I = iq[2 * s], Q = iq[2 * s + 1];
                    uint32_t k = scheme == 0 ? (uint32_t)(I * I + Q * Q) : scheme == 1 ? (uint32_t)((128 - abs(I)) * 129 + (128 - abs(Q))) : (uint32_t)(rand() % 16640);
                    keys[t * 8 + j] = k;
                }
                CHECK(hipMemcpy(dk, keys.data(), nkeys * 4, hipMemcpyHostToDevice));
                for (int wide = 0; wide < 1; ++wide) {
                    uint64_t h;
                    const int iters = 500;
                    for (int rep = 0; rep < 2; ++rep) {
                        hipLaunchKernelGGL(k_gather<0>, dim3(1), dim3(1024), tab, 0, dk, dout, tab, iters);
                        CHECK(hipDeviceSynchronize());
                    }
                    CHECK(hipMemcpy(&h, dout, 8, hipMemcpyDeviceToHost));
                    printf("gather region %d key %-22s: %.2f ticks per wave-instruction per CU (16 waves; x %.2f shader cycles per tick)\n", region,
                           scheme == 0 ? "n = I^2+Q^2" : scheme == 1 ? "(128-|I|)*129+(128-|Q|)" : "uniform random", (double)h / (iters * 8.0 * 16), 4.0 / base);
                }
            }
        }
    }
    return 0;
}
