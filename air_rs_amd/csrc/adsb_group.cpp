// adsb_group.cpp -- one long IQ buffer time-sharded over several contexts / GPUs behind the C ABI
// (include/adsb_hip.h, adsb_group_*).
//
// The reference's thread 2 is one Rust function on one thread (src/adsb.rs:92, spawned at adsb.rs:147): what a
// maintainer can swap in is a call, not a launcher with one process per GPU.  A group is that call for N devices:
// one adsb_ctx (own stream, own buffers) per member, the offsets [0, n - 240) of the buffer split evenly in member
// order, every member reading its offsets plus a 239-sample halo (240 samples of overlap between neighbours: the
// window is 16 + 224 samples, adsb.rs:98,106).  Every offset of the reference loop is independent and no state
// crosses offsets, so the members' lists -- each kernel writes ABSOLUTE offsets (adsb_set_stream_base) -- simply
// concatenate, in member order, into the list one context would have produced: no rebasing, no sort.  The lists
// meet in the root member's device memory by hipMemcpyPeerAsync (24 bytes per frame: a latency question, not a
// bandwidth one), laid out like adsb_set_result_target's blob.  Members may share a device (that is how one GPU
// tests the path).  No torch, no Python, no launcher; bench.py --gpus N stays on torch.distributed because the
// driver launches it that way (one process per GPU).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/adsb_hip.h"

namespace {
constexpr uint64_t kWindow = 240; // 16 + 112 * 2 (reference src/adsb.rs:98)

#define GHIP(x)                                    \
    do {                                           \
        hipError_t e_ = (x);                       \
        if (e_ != hipSuccess) return (int)e_;      \
    } while (0)
} // namespace

struct adsb_group {
    std::vector<adsb_ctx *> ctx;
    std::vector<int> dev;
    std::vector<void *> staging;       // per member: device copy of its slice (host-fed entry point)
    std::vector<hipEvent_t> done;      // per member: its ordered list is complete
    std::vector<adsb_group_shard> plan;
    uint32_t root = 0;
    int sample_type = ADSB_SAMPLE_I8;
    uint32_t bps = 2;
    uint64_t max_samples = 0, max_out = 0, member_samples = 0;
    char *merged = nullptr;            // root device: [u64 n_out | u64 total_found | u64 flags | u64 0 | frames[max_out]]
    hipStream_t root_stream = nullptr;
    uint64_t *hdr_host = nullptr;      // pinned, 4 words
    bool launched = false, merged_current = false;
    uint64_t n_out = 0, total = 0;
    uint32_t flags = 0;
};

// The group calls switch the calling thread's current HIP device (every member lives on its own): they put it back.
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

extern "C" void adsb_group_destroy(adsb_group *g)
{
    DeviceGuard restore_device;
    if (!g) return;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        (void)hipSetDevice(g->dev[i]);
        if (g->ctx[i]) adsb_destroy(g->ctx[i]);
        if (i < g->staging.size()) (void)hipFree(g->staging[i]);
        if (i < g->done.size() && g->done[i]) (void)hipEventDestroy(g->done[i]);
    }
    if (!g->dev.empty()) {
        (void)hipSetDevice(g->dev[g->root]);
        if (g->root_stream) { (void)hipStreamSynchronize(g->root_stream); (void)hipStreamDestroy(g->root_stream); }
        (void)hipFree(g->merged);
    }
    if (g->hdr_host) (void)hipHostFree(g->hdr_host);
    delete g;
}

// Even split of the offsets [0, n - 240) in member order (the same plan as air_rs_amd/sharding.py plan()).
static void make_plan(uint64_t n_samples, uint32_t n_members, adsb_group_shard *out)
{
    const uint64_t n_off = n_samples - kWindow;
    // (a multiple of 8 samples, so that every slice of ONE 16-byte aligned buffer starts 16-byte aligned)
    const uint64_t per = n_off ? ((n_off + n_members - 1) / n_members + 7) & ~7ull : 0;
    for (uint32_t i = 0; i < n_members; ++i) {
        const uint64_t lo = std::min<uint64_t>((uint64_t)i * per, n_off), hi = std::min<uint64_t>(lo + per, n_off);
        out[i].first_sample = lo;
        out[i].n_offsets = hi - lo;
        out[i].n_samples = hi > lo ? (hi - lo) + kWindow : 0;
    }
}

extern "C" int adsb_group_plan(uint64_t n_samples, uint32_t n_members, adsb_group_shard *shards)
{
    if (!shards || n_members == 0) return ADSB_E_ARG;
    if (n_samples < kWindow) return ADSB_E_SHORT;
    make_plan(n_samples, n_members, shards);
    return ADSB_OK;
}

extern "C" int adsb_group_create(const adsb_group_cfg *cfg, adsb_group **out)
{
    DeviceGuard restore_device;
    if (!cfg || !out) return ADSB_E_ARG;
    *out = nullptr;
    if (cfg->abi_version != ADSB_ABI_VERSION || cfg->n_members == 0 || cfg->n_members > 64 || !cfg->devices ||
        cfg->root >= cfg->n_members || cfg->max_samples < kWindow || cfg->max_out == 0)
        return ADSB_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ADSB_E_NODEVICE;
    for (uint32_t i = 0; i < cfg->n_members; ++i)
        if (cfg->devices[i] < 0 || cfg->devices[i] >= ndev) return ADSB_E_NODEVICE;
    adsb_group *g = new (std::nothrow) adsb_group();
    if (!g) return ADSB_E_NOMEM;
    g->root = cfg->root;
    g->sample_type = cfg->sample_type;
    g->bps = cfg->sample_type == ADSB_SAMPLE_I8 ? 2 : 4;
    g->max_samples = cfg->max_samples;
    g->max_out = cfg->max_out;
    // the largest slice a member can get: ceil((max_samples - 240) / N) offsets + the halo
    g->member_samples = (((cfg->max_samples - kWindow + cfg->n_members - 1) / cfg->n_members + 7) & ~7ull) + kWindow;
    g->plan.resize(cfg->n_members);
    int rc = ADSB_OK;
    for (uint32_t i = 0; i < cfg->n_members && rc == ADSB_OK; ++i) {
        g->dev.push_back(cfg->devices[i]);
        g->ctx.push_back(nullptr);
        g->staging.push_back(nullptr);
        g->done.push_back(nullptr);
        adsb_cfg c{};
        c.abi_version = ADSB_ABI_VERSION;
        c.device = cfg->devices[i];
        c.sample_type = cfg->sample_type;
        c.max_channels = 1;
        c.max_samples = g->member_samples;
        c.max_out = cfg->max_out; // all frames may sit in one member's slice
        c.stream = nullptr;
        c.host_staging = 0;
        rc = adsb_create(&c, &g->ctx[i]);
        if (rc != ADSB_OK) break;
        if (hipSetDevice(cfg->devices[i]) != hipSuccess ||
            hipEventCreateWithFlags(&g->done[i], hipEventDisableTiming) != hipSuccess) { rc = ADSB_E_NODEVICE; break; }
        if (cfg->host_staging &&
            hipMalloc(&g->staging[i], (size_t)g->member_samples * g->bps + 64) != hipSuccess) rc = ADSB_E_NOMEM;
    }
    if (rc == ADSB_OK) {
        if (hipSetDevice(g->dev[g->root]) != hipSuccess ||
            hipStreamCreateWithFlags(&g->root_stream, hipStreamNonBlocking) != hipSuccess)
            rc = ADSB_E_NODEVICE;
        else if (hipMalloc((void **)&g->merged, 32 + sizeof(adsb_frame) * (size_t)cfg->max_out) != hipSuccess ||
                 hipHostMalloc((void **)&g->hdr_host, 32, hipHostMallocDefault) != hipSuccess)
            rc = ADSB_E_NOMEM;
        else if (hipMemsetAsync(g->merged, 0, 32, g->root_stream) != hipSuccess) rc = ADSB_E_NODEVICE;
    }
    // peer access between the root's device and the others (a no-op between contexts on one device; where the
    // platform offers none, hipMemcpyPeerAsync stages through the host by itself)
    if (rc == ADSB_OK)
        for (uint32_t i = 0; i < cfg->n_members; ++i)
            if (g->dev[i] != g->dev[g->root]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, g->dev[g->root], g->dev[i]) == hipSuccess && can) {
                    (void)hipSetDevice(g->dev[g->root]);
                    hipError_t e = hipDeviceEnablePeerAccess(g->dev[i], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                }
            }
    if (rc != ADSB_OK) { adsb_group_destroy(g); return rc; }
    *out = g;
    return ADSB_OK;
}

extern "C" uint32_t adsb_group_size(const adsb_group *g) { return g ? (uint32_t)g->ctx.size() : 0u; }
extern "C" adsb_ctx *adsb_group_member(adsb_group *g, uint32_t i) { return (g && i < g->ctx.size()) ? g->ctx[i] : nullptr; }

extern "C" int adsb_group_demod_device_async(adsb_group *g, const void *const *iq_dev, size_t n_samples)
{
    DeviceGuard restore_device;
    if (!g || !iq_dev) return ADSB_E_ARG;
    if (n_samples < kWindow) return ADSB_E_SHORT;
    if (n_samples > g->max_samples) return ADSB_E_CAPACITY;
    const uint32_t n = (uint32_t)g->ctx.size();
    make_plan(n_samples, n, g->plan.data());
    g->launched = false;
    g->merged_current = false;
    for (uint32_t i = 0; i < n; ++i) {
        const adsb_group_shard &sh = g->plan[i];
        if (sh.n_samples == 0) continue; // (fewer offsets than members, or exactly 240 samples: nothing to do here)
        if (!iq_dev[i]) return ADSB_E_ARG;
        int rc = adsb_set_stream_base(g->ctx[i], sh.first_sample);
        if (rc == ADSB_OK) rc = adsb_demod_device_async(g->ctx[i], iq_dev[i], 1, (size_t)sh.n_samples, (size_t)sh.n_samples);
        if (rc != ADSB_OK) return rc;
    }
    g->launched = true;
    return ADSB_OK;
}

extern "C" int adsb_group_demod_host_async(adsb_group *g, const void *iq_host, size_t n_samples)
{
    DeviceGuard restore_device;
    if (!g || !iq_host) return ADSB_E_ARG;
    if (n_samples < kWindow) return ADSB_E_SHORT;
    if (n_samples > g->max_samples) return ADSB_E_CAPACITY;
    const uint32_t n = (uint32_t)g->ctx.size();
    std::vector<adsb_group_shard> plan(n);
    make_plan(n_samples, n, plan.data());
    std::vector<const void *> ptr(n, nullptr);
    for (uint32_t i = 0; i < n; ++i) {
        if (plan[i].n_samples == 0) continue;
        if (!g->staging[i]) return ADSB_E_STATE; // created without host_staging
        GHIP(hipSetDevice(g->dev[i]));
        // each member's slice travels on its own stream, in front of its own kernels.  With PINNED iq_host (hipHostMalloc /
        // hipHostRegister by the caller) the copies of different devices overlap each other and the kernels of the members
        // that already have their samples; from pageable memory the runtime stages them and they run one after the other
        GHIP(hipMemcpyAsync(g->staging[i], static_cast<const char *>(iq_host) + plan[i].first_sample * g->bps,
                            (size_t)plan[i].n_samples * g->bps, hipMemcpyHostToDevice, (hipStream_t)adsb_stream(g->ctx[i])));
        ptr[i] = g->staging[i];
    }
    return adsb_group_demod_device_async(g, ptr.data(), n_samples);
}

// Waits for every member's count (20 bytes each), then lets the root device pull the lists into place.
static int merge_lists(adsb_group *g)
{
    if (!g->launched) return ADSB_E_STATE;
    if (g->merged_current) return ADSB_OK;
    const uint32_t n = (uint32_t)g->ctx.size();
    uint64_t pos = 0, total = 0;
    uint32_t flags = 0;
    struct Part { uint32_t member; uint64_t at, count; };
    std::vector<Part> parts;
    for (uint32_t i = 0; i < n; ++i) {
        if (g->plan[i].n_samples == 0) continue;
        uint64_t n_out = 0, tot = 0;
        uint32_t fl = 0;
        int rc = adsb_fetch_counts(g->ctx[i], &n_out, &tot, &fl); // (also completes a list with holes: ADSB_FLAG_INCOMPLETE)
        if (rc != ADSB_OK) return rc;
        total += tot;
        flags |= fl & ~ADSB_FLAG_INCOMPLETE;
        const uint64_t take = std::min<uint64_t>(n_out, g->max_out - pos);
        if (take) parts.push_back({i, pos, take});
        pos += take;
    }
    if (total > pos) flags |= ADSB_FLAG_TRUNCATED;
    GHIP(hipSetDevice(g->dev[g->root]));
    // (the pinned header words below may still be the source of the previous merge's copy, which
    // adsb_group_result_device does not wait for)
    GHIP(hipStreamSynchronize(g->root_stream));
    for (const Part &p : parts) {
        const adsb_frame *src = nullptr;
        int rc = adsb_result_device(g->ctx[p.member], &src, nullptr);
        if (rc != ADSB_OK) return rc;
        // (adsb_fetch_counts has waited for the member's stream: its list is complete)
        GHIP(hipMemcpyPeerAsync(g->merged + 32 + p.at * sizeof(adsb_frame), g->dev[g->root], src, g->dev[p.member],
                                (size_t)p.count * sizeof(adsb_frame), g->root_stream));
    }
    g->hdr_host[0] = pos;
    g->hdr_host[1] = total;
    g->hdr_host[2] = flags;
    g->hdr_host[3] = 0;
    GHIP(hipMemcpyAsync(g->merged, g->hdr_host, 32, hipMemcpyHostToDevice, g->root_stream));
    g->n_out = pos;
    g->total = total;
    g->flags = flags;
    g->merged_current = true;
    return ADSB_OK;
}

extern "C" int adsb_group_result_device(adsb_group *g, const void **blob_dev, void **stream)
{
    DeviceGuard restore_device;
    if (!g) return ADSB_E_ARG;
    int rc = merge_lists(g);
    if (rc != ADSB_OK) return rc;
    if (blob_dev) *blob_dev = g->merged;
    if (stream) *stream = (void *)g->root_stream;
    return ADSB_OK;
}

extern "C" int adsb_group_fetch(adsb_group *g, adsb_frame *out, size_t max_out, size_t *n_out, uint64_t *total_found,
                                uint32_t *flags)
{
    DeviceGuard restore_device;
    if (!g || !n_out || (!out && max_out)) return ADSB_E_ARG;
    *n_out = 0;
    int rc = merge_lists(g);
    if (rc != ADSB_OK) return rc;
    uint64_t n = g->n_out;
    uint32_t fl = g->flags;
    if (n > max_out) { n = max_out; fl |= ADSB_FLAG_TRUNCATED; }
    GHIP(hipSetDevice(g->dev[g->root]));
    if (n) GHIP(hipMemcpyAsync(out, g->merged + 32, (size_t)n * sizeof(adsb_frame), hipMemcpyDeviceToHost, g->root_stream));
    GHIP(hipStreamSynchronize(g->root_stream));
    *n_out = (size_t)n;
    if (total_found) *total_found = g->total;
    if (flags) *flags = fl;
    return ADSB_OK;
}

extern "C" int adsb_group_demod(adsb_group *g, const void *iq_host, size_t n_samples, adsb_frame *out, size_t max_out,
                                size_t *n_out, uint32_t *flags)
{
    if (!g || !n_out) return ADSB_E_ARG;
    *n_out = 0;
    if (flags) *flags = 0;
    int rc = adsb_group_demod_host_async(g, iq_host, n_samples);
    if (rc != ADSB_OK) return rc;
    return adsb_group_fetch(g, out, max_out, n_out, nullptr, flags);
}
