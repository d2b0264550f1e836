// feed_bench.cpp -- host-fed cost per buffer of the drop-in in the reference's own buffer sizes, measured from C
// (no interpreter in the loop): the streaming front end (adsb_feed_push / adsb_feed_pop, two buffers in flight) and one
// blocking adsb_demod() per buffer, each with the one-dispatch path for small buffers on and off (ADSB_SMALL_PATH).
// Reference: playback_thread sends 20 000-sample buffers (src/adsb.rs:77-79), the SDR reader MTU-sized ones
// (adsb.rs:59-64); thread 2 handles them one by one (adsb.rs:95-116).
// build: g++ -O2 -std=c++17 tools/bench/feed_bench.cpp -Iinclude -Lair_rs_amd/lib -ladsb_hip -Wl,-rpath,$PWD/air_rs_amd/lib -o tools/bench/feed_bench
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "adsb_hip.h"

static double now_us()
{
    using namespace std::chrono;
    return duration_cast<duration<double, std::micro>>(steady_clock::now().time_since_epoch()).count();
}

struct Result { double us_per_buffer; size_t frames; };

static Result run_feed(int st, size_t chunk, int n_buf, bool carry, bool in_place)
{
    const size_t bps = st == ADSB_SAMPLE_I8 ? 2 : 4;
    adsb_synth_cfg sc;
    adsb_synth_default(&sc);
    sc.seed = 9;
    if (st == ADSB_SAMPLE_I16) sc.amp_shift = 5;
    const int n_src = 8;
    std::vector<char> data(chunk * bps * n_src);
    adsb_synth_fill_host(&sc, st, 0, 0, chunk * n_src, data.data());
    adsb_cfg cfg{};
    cfg.abi_version = ADSB_ABI_VERSION;
    cfg.device = 0;
    cfg.sample_type = st;
    cfg.max_channels = 1;
    cfg.max_samples = chunk + 240;
    cfg.max_out = chunk / 200 + 4096;
    adsb_ctx *ctx = nullptr;
    if (adsb_create(&cfg, &ctx) != ADSB_OK) { std::fprintf(stderr, "adsb_create failed\n"); std::exit(1); }
    adsb_feed_cfg fc{};
    fc.max_chunk = chunk;
    fc.carry = carry ? 1u : 0u;
    fc.ring_slots = 3;
    adsb_feed *feed = nullptr;
    if (adsb_feed_open(ctx, &fc, &feed) != ADSB_OK) { std::fprintf(stderr, "adsb_feed_open failed\n"); std::exit(1); }
    std::vector<adsb_frame> frames(cfg.max_out);
    size_t total = 0;
    auto one = [&](int k, bool fill) {
        const char *src = data.data() + (size_t)(k % n_src) * chunk * bps;
        int rc;
        if (in_place) { // a real producer (SDR driver, file reader) writes its samples here; the fill is its cost
            void *slot = nullptr;
            rc = adsb_feed_acquire(feed, &slot);
            if (rc == ADSB_OK && fill) std::memcpy(slot, src, chunk * bps);
            if (rc == ADSB_OK) rc = adsb_feed_push(feed, nullptr, chunk);
        } else {
            rc = adsb_feed_push(feed, src, chunk);
        }
        if (rc != ADSB_OK) { std::fprintf(stderr, "push failed: %s\n", adsb_strerror(rc)); std::exit(1); }
        if (adsb_feed_in_flight(feed) == 2) {
            size_t n = 0;
            uint32_t fl = 0;
            rc = adsb_feed_pop(feed, frames.data(), frames.size(), &n, &fl, nullptr);
            if (rc != ADSB_OK) { std::fprintf(stderr, "pop failed: %s\n", adsb_strerror(rc)); std::exit(1); }
            total += n;
        }
    };
    for (int k = 0; k < 6; ++k) one(k, true);
    while (adsb_feed_in_flight(feed) > 0) { size_t n; adsb_feed_pop(feed, frames.data(), frames.size(), &n, nullptr, nullptr); }
    total = 0;
    const double t0 = now_us();
    for (int k = 0; k < n_buf; ++k) one(k + 6, false);
    while (adsb_feed_in_flight(feed) > 0) { size_t n = 0; adsb_feed_pop(feed, frames.data(), frames.size(), &n, nullptr, nullptr); total += n; }
    const double dt = now_us() - t0;
    adsb_feed_close(feed);
    adsb_destroy(ctx);
    return {dt / n_buf, total};
}

static Result run_blocking(int st, size_t chunk, int n_buf)
{
    const size_t bps = st == ADSB_SAMPLE_I8 ? 2 : 4;
    adsb_synth_cfg sc;
    adsb_synth_default(&sc);
    sc.seed = 9;
    if (st == ADSB_SAMPLE_I16) sc.amp_shift = 5;
    std::vector<char> data(chunk * bps);
    adsb_synth_fill_host(&sc, st, 0, 0, chunk, data.data());
    adsb_cfg cfg{};
    cfg.abi_version = ADSB_ABI_VERSION;
    cfg.sample_type = st;
    cfg.max_channels = 1;
    cfg.max_samples = chunk;
    cfg.max_out = chunk / 200 + 4096;
    cfg.host_staging = 1;
    adsb_ctx *ctx = nullptr;
    if (adsb_create(&cfg, &ctx) != ADSB_OK) { std::fprintf(stderr, "adsb_create failed\n"); std::exit(1); }
    std::vector<adsb_frame> frames(cfg.max_out);
    size_t n = 0, total = 0;
    for (int k = 0; k < 5; ++k) adsb_demod(ctx, data.data(), chunk, frames.data(), frames.size(), &n, nullptr);
    const double t0 = now_us();
    for (int k = 0; k < n_buf; ++k) {
        if (adsb_demod(ctx, data.data(), chunk, frames.data(), frames.size(), &n, nullptr) != ADSB_OK) { std::fprintf(stderr, "adsb_demod failed\n"); std::exit(1); }
        total += n;
    }
    const double dt = now_us() - t0;
    adsb_destroy(ctx);
    return {dt / n_buf, total};
}

int main()
{
    const struct { int st; const char *name; } types[] = {{ADSB_SAMPLE_I8, "i8"}, {ADSB_SAMPLE_I16, "cs16"}};
    for (int small = 1; small >= 0; --small) {
        setenv("ADSB_SMALL_PATH", small ? "1" : "0", 1);
        std::printf("== one-dispatch path for small buffers %s (ADSB_SMALL_PATH=%d)\n", small ? "ON" : "OFF", small);
        for (const auto &t : types) {
            for (size_t chunk : {(size_t)20000, (size_t)131072, (size_t)500000}) {
                const int n_buf = chunk <= 20000 ? 4000 : 1000;
                for (int carry = 0; carry < 2; ++carry)
                    for (int inpl = 0; inpl < 2; ++inpl) {
                        const Result r = run_feed(t.st, chunk, n_buf, carry != 0, inpl != 0);
                        std::printf("  feed  %-4s chunk %7zu %-6s %-18s: %8.1f us/buffer = %9.1f Msamples/s  (%zu frames)\n", t.name, chunk,
                                    carry ? "carry" : "parity", inpl ? "in-place producer" : "push (host memcpy)", r.us_per_buffer,
                                    chunk / r.us_per_buffer, r.frames);
                    }
                const Result b = run_blocking(t.st, chunk, chunk <= 20000 ? 2000 : 500);
                std::printf("  adsb_demod %-4s chunk %7zu (blocking, pageable memory): %8.1f us/buffer = %9.1f Msamples/s  (%zu frames)\n", t.name,
                            chunk, b.us_per_buffer, chunk / b.us_per_buffer, b.frames);
            }
        }
    }
    return 0;
}
