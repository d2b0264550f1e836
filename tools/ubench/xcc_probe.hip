// xcc_probe.hip -- which XCD runs workgroup b?  tile_of_workgroup (adsb_kernels.hip) assumes the dispatcher deals the workgroups of
// a 1-D grid to the eight XCDs round-robin (b mod 8).  Every workgroup reads the hardware's XCC_ID (s_getreg_b32, hwreg 20 on gfx940+)
// and the probe counts, per XCD, how many workgroups ran there and how many of them had b mod 8 equal to one fixed value.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/xcc_probe.hip -o tools/ubench/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(256, 8) void k_probe(uint8_t *out, int spin)
{
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // some work, so that the grid does not drain faster than it is dispatched (the mapping must hold under back-pressure too)
    uint32_t a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1664525u + 1013904223u;
    if (threadIdx.x == 0) out[blockIdx.x] = (uint8_t)((xcc & 0xFu) | (a == 0x12345u ? 0x80u : 0u));
}

int main()
{
    const int grids[3] = {2048, 32768, 131079};
    for (int n : grids) {
        uint8_t *d;
        if (hipMalloc(&d, n) != hipSuccess) return 2;
        hipLaunchKernelGGL(k_probe, dim3(n), dim3(256), 0, 0, d, 2000);
        std::vector<uint8_t> h(n);
        if (hipMemcpy(h.data(), d, n, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        long per[16] = {0}, match = 0;
        int shift = (h[0] & 0xF); // XCD of workgroup 0
        for (int b = 0; b < n; ++b) {
            per[h[b] & 0xF]++;
            if (((h[b] & 0xF) + 8 - shift) % 8 == b % 8) ++match;
        }
        printf("grid %6d: workgroup 0 ran on XCC %d; workgroups whose XCC == (b + that) mod 8: %ld of %d; per XCC:", n, shift, match, n);
        for (int x = 0; x < 8; ++x) printf(" %ld", per[x]);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
