// read_shapes.hip -- what shape of a pure streaming read gets the most out of this box's HBM (the "read ceiling" bench.py quotes
// next to every kernel time comes from ONE shape, adsbk::read_only_kernel: 16 loads of 16 bytes per lane, one workgroup per 64 KB).
// Shapes: LOADS in {4, 8, 16} 16-byte nt loads per lane all in flight (16 / 32 / 64 KB per 256-thread workgroup), workgroups per CU
// held to WGS in {4, 5, 8, 16} by dynamic LDS, chunk order = workgroup index or XCD-contiguous (eight ranges, as the scan's tiles).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/read_shapes.hip -o tools/ubench/read_shapes ; run: read_shapes [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int LOADS, bool XCD>
__global__ __launch_bounds__(256) void k_read(const u32x4 *buf, size_t n16, uint32_t n_wg, uint32_t *sink)
{
    extern __shared__ unsigned char lds_pad[]; // (only to hold the workgroups per CU down)
    uint32_t b = blockIdx.x;
    if (XCD) {
        const uint32_t q = n_wg >> 3, r = n_wg & 7u, x = b & 7u, j = b >> 3;
        b = x * q + (x < r ? x : r) + j;
    }
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
    if (acc == 0x9E3779B9u) { *sink = acc; lds_pad[threadIdx.x] = 1; }
}

template <int LOADS, bool XCD>
static double run(const void *buf, size_t bytes, int wgs_per_cu, uint32_t *sink)
{
    const size_t n16 = bytes / 16;
    const uint32_t n_wg = (uint32_t)((n16 + 256 * LOADS - 1) / (256 * LOADS));
    const size_t lds = wgs_per_cu >= 16 ? 0 : (size_t)(160 * 1024 / wgs_per_cu) - 1024; // > 1/(wgs+1) of the CU's LDS for every wgs used here
    CHECK(hipFuncSetAttribute((const void *)k_read<LOADS, XCD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((k_read<LOADS, XCD>), dim3(n_wg), dim3(256), lds, 0, (const u32x4 *)buf, n16, n_wg, sink);
    CHECK(hipDeviceSynchronize());
    const int reps = bytes > (4ull << 30) ? 10 : 60;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_read<LOADS, XCD>), dim3(n_wg), dim3(256), lds, 0, (const u32x4 *)buf, n16, n_wg, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return (double)bytes * reps / (ms * 1e-3) / 1e9;
}

int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 1;
    const size_t bytes = gib << 30;
    void *buf; uint32_t *sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, bytes));
    printf("pure read of %zu GiB, back-to-back launches (GB/s, 8000 = the peak the roofline uses)\n", gib);
    const int wgs[4] = {4, 5, 8, 16};
    for (int w : wgs) {
        printf("workgroups per CU <= %2d:  4 loads %7.1f (xcd %7.1f)   8 loads %7.1f (xcd %7.1f)  16 loads %7.1f (xcd %7.1f)\n", w,
               run<4, false>(buf, bytes, w, sink), run<4, true>(buf, bytes, w, sink), run<8, false>(buf, bytes, w, sink),
               run<8, true>(buf, bytes, w, sink), run<16, false>(buf, bytes, w, sink), run<16, true>(buf, bytes, w, sink));
    }
    return 0;
}
