// adsb_threads.cpp -- see adsb_threads.hpp.
#include "adsb_threads.hpp"

#include <chrono>
#include <cstdio>
#include <fstream>
#include <thread>

namespace air_rs_amd {

bool load_data(const std::string &filename, IqBufI16 &out, std::string &err)
{
    std::ifstream f(filename, std::ios::binary | std::ios::ate);
    if (!f) { err = "cannot open " + filename; return false; }
    std::streamsize len = f.tellg();
    if (len % 4 != 0) { err = "Invalid file length (not divisible by 4)"; return false; } // utils.rs:28-30
    f.seekg(0);
    std::vector<unsigned char> bytes((size_t)len);
    if (len && !f.read(reinterpret_cast<char *>(bytes.data()), len)) { err = "short read"; return false; }
    out.resize((size_t)len / 4);
    for (size_t k = 0; k < out.size(); ++k) {
        out[k].re = (int16_t)(bytes[4 * k] | (bytes[4 * k + 1] << 8));
        out[k].im = (int16_t)(bytes[4 * k + 2] | (bytes[4 * k + 3] << 8));
    }
    return true;
}

bool save_data(const IqBufI16 &data, const std::string &filename, std::string &err)
{
    std::ofstream f(filename, std::ios::binary);
    if (!f) { err = "cannot create " + filename; return false; }
    std::vector<unsigned char> bytes(data.size() * 4);
    for (size_t k = 0; k < data.size(); ++k) {
        uint16_t re = (uint16_t)data[k].re, im = (uint16_t)data[k].im;
        bytes[4 * k] = re & 0xFF; bytes[4 * k + 1] = re >> 8;
        bytes[4 * k + 2] = im & 0xFF; bytes[4 * k + 3] = im >> 8;
    }
    f.write(reinterpret_cast<const char *>(bytes.data()), (std::streamsize)bytes.size());
    return (bool)f;
}

template <typename T>
void playback_thread(Sender<std::vector<Complex<T>>> tx, std::vector<Complex<T>> data, size_t chunk_len, bool pace,
                     bool send_tail, size_t min_tail)
{
    // The reference computes `data.len()-20000` in usize: a shorter file underflows (panic in
    // debug builds).  Here that case sends nothing (or, with send_tail, the whole file as one buffer).
    size_t i = 0;
    while (data.size() >= chunk_len && i < data.size() - chunk_len) {
        std::vector<Complex<T>> buf(data.begin() + i, data.begin() + i + chunk_len);
        i += chunk_len;
        if (!tx.send(std::move(buf))) {
            std::printf("Raw sdr receiver is dropped\n");
            return;
        }
        if (pace) std::this_thread::sleep_for(std::chrono::duration<double>(1e4 / 2e6));
    }
    // what adsb.rs:77's strict `<` never sends.  (A tail too short to be a buffer of its own for the consumer -- fewer
    // than 240 samples when thread 2 runs with the reference's per-buffer semantics, where it would be the panic of
    // adsb.rs:98 -- is not sent: there is no offset in it to examine.)
    if (send_tail && i < data.size() && data.size() - i >= min_tail) {
        std::vector<Complex<T>> buf(data.begin() + i, data.end());
        if (!tx.send(std::move(buf))) {
            std::printf("Raw sdr receiver is dropped\n");
            return;
        }
    }
    tx.drop();
}

template <typename T>
Thread2Stats process_sdr_data_thread(adsb_ctx *ctx, Receiver<std::vector<Complex<T>>> rx,
                                     Sender<AdsbPacket> tx, std::vector<adsb_frame> *frames_log,
                                     size_t max_frames, bool carry_over, size_t max_chunk)
{
    Thread2Stats st;
    // A buffer of n samples has n-240 offsets, so max_frames >= the largest buffer never truncates
    // (the reference's channel is unbounded).
    std::vector<adsb_frame> frames(max_frames);
    if (max_chunk == 0) max_chunk = max_frames;
    adsb_feed *feed = nullptr;
    adsb_feed_cfg fc{};
    fc.max_chunk = max_chunk;
    fc.carry = carry_over ? 1u : 0u;
    fc.ring_slots = 3;
    int rc = adsb_feed_open(ctx, &fc, &feed);
    if (rc != ADSB_OK) {
        st.last_error = rc;
        std::fprintf(stderr, "adsb_feed_open failed: %s\n", adsb_strerror(rc));
        tx.drop();
        return st;
    }
    bool closed = false, pop_failed = false;
    // one finished buffer: AdsbPacket::new per frame, in order (adsb.rs:107-111)
    auto pop_and_send = [&]() {
        size_t n_out = 0;
        uint32_t flags = 0;
        uint64_t first = 0;
        int prc = adsb_feed_pop(feed, frames.data(), frames.size(), &n_out, &flags, &first);
        if (prc != ADSB_OK) {
            if (st.last_error == ADSB_OK) st.last_error = prc;
            std::fprintf(stderr, "adsb_feed_pop failed: %s\n", adsb_strerror(prc));
            pop_failed = true; // (that buffer's frames are gone: any error ends the thread, like a closed channel)
            return false;
        }
        if (flags & ADSB_FLAG_TRUNCATED) st.truncated_buffers++;
        for (size_t k = 0; k < n_out && !closed; ++k) {
            if (frames_log) {
                adsb_frame f = frames[k];
                if (!carry_over) f.offset += first; // parity mode: offsets are buffer-relative
                frames_log->push_back(f);
            }
            AdsbPacket packet(std::vector<uint8_t>(frames[k].bytes, frames[k].bytes + 14)); // adsb.rs:107
            if (!tx.send(std::move(packet))) {
                std::printf("Adsb msg receiver is dropped\n");
                closed = true;
                break;
            }
            st.frames++;
        }
        return !closed;
    };
    // The reference sends a buffer's packets before its next recv() (adsb.rs:95-116).  Here up to two buffers are in
    // flight on the GPU, but nothing is held back: while the input channel is empty the thread waits for the GPU
    // instead (a stalled source cannot strand the last buffer's frames), and after every push whatever has already
    // finished is handed on.
    for (;;) {
        std::vector<Complex<T>> buf;
        int got = rx.try_recv(&buf);
        if (got == 1) { // Err(Empty)
            if (adsb_feed_in_flight(feed) > 0) {
                if (!pop_and_send()) break; // (blocks on the GPU, not on the source)
                continue;
            }
            auto b = rx.recv(); // nothing in flight: block on the source like the reference
            if (!b) break;
            buf = std::move(*b);
        } else if (got == 2) {
            break; // every sender dropped and the queue drained (adsb.rs:95 ends the loop)
        }
        if (buf.size() > max_chunk) rc = ADSB_E_CAPACITY;
        else rc = adsb_feed_push(feed, buf.data(), buf.size());
        if (rc != ADSB_OK) {
            // ADSB_E_SHORT is where the reference panics (adsb.rs:98); any error ends the thread.
            st.last_error = rc;
            std::fprintf(stderr, "adsb_feed_push failed: %s\n", adsb_strerror(rc));
            break;
        }
        st.buffers++;
        while (!closed && (adsb_feed_in_flight(feed) == 2 || (adsb_feed_in_flight(feed) > 0 && adsb_feed_ready(feed) == 1)))
            if (!pop_and_send()) break;
        if (closed || pop_failed) break;
    }
    while (!closed && !pop_failed && adsb_feed_in_flight(feed) > 0)
        if (!pop_and_send()) break;
    adsb_feed_close(feed);
    if (closed) return st;
    tx.drop();
    return st;
}

template void playback_thread<int16_t>(Sender<IqBufI16>, IqBufI16, size_t, bool, bool, size_t);
template void playback_thread<int8_t>(Sender<IqBufI8>, IqBufI8, size_t, bool, bool, size_t);
template Thread2Stats process_sdr_data_thread<int16_t>(adsb_ctx *, Receiver<IqBufI16>, Sender<AdsbPacket>, std::vector<adsb_frame> *, size_t, bool, size_t);
template Thread2Stats process_sdr_data_thread<int8_t>(adsb_ctx *, Receiver<IqBufI8>, Sender<AdsbPacket>, std::vector<adsb_frame> *, size_t, bool, size_t);

} // namespace air_rs_amd
