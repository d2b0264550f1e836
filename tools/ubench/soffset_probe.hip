// soffset_probe.hip -- is a raw buffer load's SGPR offset part of the hardware bounds check on gfx950?
// A 2 MiB allocation holds a pattern; the descriptor covers only its first 1 MiB.  Lanes read at voffset + soffset
// positions on both sides of the 1 MiB mark (always inside the allocation): a 0 beyond the mark means the check
// counts the SGPR offset (the tile kernel may then move its per-sweep constant there), the pattern means it does not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const uint32_t *buf, uint32_t nrec, uint32_t *out)
{
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, (int)nrec, 0x00020000);
    const uint32_t v = threadIdx.x * 16;
    // soffset as a run-time SGPR value (blockIdx-dependent so that it cannot be folded into the immediate)
    const uint32_t s = (blockIdx.x + 1) * 0x40000u; // 256 KiB, 512 KiB, ..., 1.75 MiB
    u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v, s, 0);
    u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, v + s, 0, 0);
    out[(blockIdx.x * 64 + threadIdx.x) * 2 + 0] = a.x;
    out[(blockIdx.x * 64 + threadIdx.x) * 2 + 1] = b.x;
}

int main()
{
    const size_t bytes = 2u << 20;
    std::vector<uint32_t> h(bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0xA0000000u | (uint32_t)i;
    uint32_t *d, *o;
    CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&o, 7 * 64 * 2 * 4));
    CHECK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(7), dim3(64), 0, 0, d, 1u << 20, o);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> r(7 * 64 * 2);
    CHECK(hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < 7; ++b) {
        const uint32_t s = (b + 1) * 0x40000u;
        int zs = 0, zv = 0, ok_s = 0, ok_v = 0;
        for (int t = 0; t < 64; ++t) {
            const uint32_t want = 0xA0000000u | ((t * 16 + s) / 4);
            const uint32_t a = r[(b * 64 + t) * 2], v = r[(b * 64 + t) * 2 + 1];
            zs += a == 0; ok_s += a == want; zv += v == 0; ok_v += v == want;
        }
        printf("offset %7u (%s the 1 MiB descriptor): via soffset: %2d data %2d zero | via voffset: %2d data %2d zero\n", s,
               s < (1u << 20) ? "inside " : "outside", ok_s, zs, ok_v, zv);
    }
    return 0;
}
