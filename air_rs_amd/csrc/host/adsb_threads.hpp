// adsb_threads.hpp -- the reference's thread structure (src/adsb.rs:75-173) with thread 2's body
// running on the GPU through the C ABI (include/adsb_hip.h).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/adsb_hip.h"
#include "adsb_packet.hpp"
#include "channel.hpp"

namespace air_rs_amd {

// num_complex::Complex<T> is #[repr(C)] {re, im}; Complex<int16_t> is the reference's sample type
// (adsb.rs:131), Complex<int8_t> the 2-byte RTL-SDR form BASELINE.json's metric is quoted on.
template <typename T> struct Complex {
    T re, im;
};
using IqBufI16 = std::vector<Complex<int16_t>>;
using IqBufI8 = std::vector<Complex<int8_t>>;

// utils.rs:22-43 / utils.rs:6-20: raw little-endian i16 I,Q pairs, no header.
bool load_data(const std::string &filename, IqBufI16 &out, std::string &err);
bool save_data(const IqBufI16 &data, const std::string &filename, std::string &err);

// adsb.rs:75-89: 20 000-sample buffers, `while i < len - 20000` (the tail is never sent), then
// drop(tx).  pace=true keeps the reference's 5 ms sleep per buffer.  send_tail=true (NOT reference behaviour,
// SURVEY 8f-1) also sends what the reference's strict `<` leaves behind: the last full or partial chunk -- unless it
// is shorter than min_tail samples (240 when the consumer demodulates buffer by buffer: such a tail holds no offset).
template <typename T>
void playback_thread(Sender<std::vector<Complex<T>>> tx, std::vector<Complex<T>> data,
                     size_t chunk_len = 20000, bool pace = false, bool send_tail = false, size_t min_tail = 0);

// adsb.rs:92-122: for every received buffer, demodulate and send one AdsbPacket per frame, in
// ascending offset order; return when either channel closes; drop(tx) at the end.
// `frames_log`, when given, also receives the raw frames with absolute offsets
// (buffer start + offset) -- test instrumentation, not part of the reference.
//
// carry_over (NOT reference behaviour, off by default): the reference never looks at
// the last 240 offsets of a buffer, so frames straddling two buffers are lost.  With
// carry_over the last 240 samples of the stream stay on the device and are put in front of the next buffer
// there, which makes the chunked stream decode exactly like one long buffer.
// Either way the buffers go through the streaming front end of the C ABI (adsb_feed_*: pinned host ring,
// asynchronous DMA overlapped with the previous buffer's kernels, two buffers in flight); packets leave in
// buffer order, ascending offset inside a buffer, one buffer behind the newest one received.
// max_chunk: the largest buffer the source may send (the ctx needs max_samples >= max_chunk + 240).
struct Thread2Stats {
    uint64_t buffers = 0, frames = 0, truncated_buffers = 0;
    int last_error = ADSB_OK; // first non-OK code returned by the C ABI, if any
};
template <typename T>
Thread2Stats process_sdr_data_thread(adsb_ctx *ctx, Receiver<std::vector<Complex<T>>> rx,
                                     Sender<AdsbPacket> tx,
                                     std::vector<adsb_frame> *frames_log = nullptr,
                                     size_t max_frames = 65536, bool carry_over = false, size_t max_chunk = 0);

} // namespace air_rs_amd
