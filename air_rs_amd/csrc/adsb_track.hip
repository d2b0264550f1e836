// adsb_track.hip -- what the reference does right behind the AdsbPacket channel, for one launch's ordered
// frame list, on the device: the per-ICAO tracker (src/adsb/aircraft.rs:48-165) and the global CPR position
// decode it calls (src/adsb/cpr.rs:22-147).  SURVEY section 8(f) rank 3.
//
// The reference runs a sequential state machine per packet: per ICAO it remembers the last even and the
// last odd position message; a new position message pairs with the last one of the other format if that
// one is at most 10 s old, and the pair yields a latitude/longitude (`first` = the older format).  In
// closed form, for position frame i:   j = the latest frame before i with the same ICAO, a position
// message, the other CPR format;  if j exists and t_i - t_j <= 10 s: position(i) = cpr(pair, first = j).
// That is a "previous matching element in my segment" query:
//   1. stable radix sort of frame indices by ICAO (24 bits; rocPRIM) -> one segment per aircraft, list
//      order (= time order) inside;
//   2. one thread per frame walks back in its segment to its partner (bounded by the 10 s window: an
//      older partner could not be used anyway) and evaluates the CPR formulas in f64;
//   3. one thread per segment tail gathers the aircraft's record (callsign of the last ID message,
//      altitude and time of the last position message, last position that was computed).
// Time is sample offset x seconds_per_sample (the reference stamps packets with the wall clock, which
// is excluded from parity).  O(frames) work, a few MB: a latency-bound epilogue.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "adsb_kernels.h"

namespace adsbk {

namespace {

__device__ __forceinline__ double cpr_to_float(uint32_t cpr) { return (double)cpr / 131072.0; } // cpr.rs:22-25

__device__ __forceinline__ uint32_t floor_as_u32(double x) // Rust `x.floor() as u32`: saturating, NaN -> 0
{
    const double f = floor(x);
    if (!(f >= 0.0)) return 0u;
    if (f >= 4294967295.0) return 4294967295u;
    return (uint32_t)f;
}

__device__ uint32_t calc_num_zones(double lat) // cpr.rs:39-54
{
    if (lat == 0.0) return 59;
    if (lat == 87.0 || lat == -87.0) return 2;
    if (lat < -87.0 || lat > 87.0) return 1;
    const double pi = 3.14159265358979323846264338327950288;
    const double int1 = 1.0 - cos(pi / 30.0);
    const double int2 = cos(pi / 180.0 * lat);
    const double int3 = (2.0 * pi) / acos(1.0 - (int1 / (int2 * int2)));
    return floor_as_u32(int3);
}

// cpr.rs:135-147 (+ 63-88, 90-127); first_is_odd: the older message's format
__device__ bool geographic_position(uint32_t even_lat_u, uint32_t even_lon_u, uint32_t odd_lat_u, uint32_t odd_lon_u,
                                    bool first_is_odd, double &latitude, double &longitude)
{
    const double even_cpr_lat = cpr_to_float(even_lat_u), odd_cpr_lat = cpr_to_float(odd_lat_u);
    const double latitude_index = floor(59.0 * even_cpr_lat - 60.0 * odd_cpr_lat + 0.5);
    const double even_latitude = (360.0 / 60.0) * (fmod(latitude_index, 60.0) + even_cpr_lat);
    const double odd_latitude = (360.0 / 59.0) * (fmod(latitude_index, 59.0) + odd_cpr_lat);
    double lat = first_is_odd ? even_latitude : odd_latitude; // the newest format decides
    if (lat > 270.0) lat -= 360.0;
    if (calc_num_zones(even_latitude) != calc_num_zones(odd_latitude)) return false;

    const double lon_cpr_e = cpr_to_float(even_lon_u), lon_cpr_o = cpr_to_float(odd_lon_u);
    const uint32_t nl = calc_num_zones(lat);
    uint32_t nz = first_is_odd ? calc_num_zones(lat) : calc_num_zones(lat - 1.0); // sic: latitude - 1.0
    if (nz < 1) nz = 1;
    const double num_zones = (double)nz;
    const double divisions = 360.0 / num_zones;
    const double m = floor(lon_cpr_e * (double)(uint32_t)(nl - 1u) - lon_cpr_o * (double)nl + 0.5);
    double lon = first_is_odd ? divisions * (fmod(m, num_zones) + lon_cpr_e)
                              : divisions * (fmod(m, num_zones) + lon_cpr_o);
    while (lon < -180.0) lon += 360.0;
    while (lon > 180.0) lon -= 360.0;
    latitude = lat;
    longitude = lon;
    return true;
}

__global__ __launch_bounds__(256) void track_keys_kernel(const adsb_packet_fields *fields, uint32_t n, uint32_t *keys,
                                                         uint32_t *vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = fields[i].icao & 0xFFFFFFu;
    vals[i] = i;
}

__global__ __launch_bounds__(256) void track_pairs_kernel(const adsb_frame *frames, const adsb_packet_fields *fields,
                                                          const uint32_t *skeys, const uint32_t *svals, uint32_t n,
                                                          double seconds_per_sample, adsb_track_point *points,
                                                          uint32_t *tail_flag)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t icao = skeys[s], i = svals[s];
    tail_flag[s] = (s + 1 == n || skeys[s + 1] != icao) ? 1u : 0u;
    const adsb_packet_fields f = fields[i];
    adsb_track_point pt;
    pt.latitude = 0.0;
    pt.longitude = 0.0;
    pt.icao = icao;
    pt.flags = 0;
    if (f.msg_kind == 1) { // AircraftPosition (aircraft.rs:54)
        const double t_i = (double)frames[i].offset * seconds_per_sample;
        for (uint32_t w = s; w > 0;) {
            --w;
            if (skeys[w] != icao) break;
            const uint32_t j = svals[w];
            const double t_j = (double)frames[j].offset * seconds_per_sample;
            if (fabs(t_i - t_j) > 10.0) break; // aircraft.rs:68-70, 84-86: too old (and so is anything before it)
            const adsb_packet_fields g = fields[j];
            if (g.msg_kind != 1 || g.cpr_odd == f.cpr_odd) continue;
            // the partner: last_odd_packet / last_even_packet at the time frame i arrives
            const bool i_odd = f.cpr_odd != 0;
            const uint32_t e_lat = i_odd ? g.cpr_latitude : f.cpr_latitude, e_lon = i_odd ? g.cpr_longitude : f.cpr_longitude;
            const uint32_t o_lat = i_odd ? f.cpr_latitude : g.cpr_latitude, o_lon = i_odd ? f.cpr_longitude : g.cpr_longitude;
            double lat, lon;
            if (geographic_position(e_lat, e_lon, o_lat, o_lon, /*first_is_odd=*/!i_odd, lat, lon)) {
                pt.latitude = lat;
                pt.longitude = lon;
                pt.flags = ADSB_TRACK_NEW_POSITION;
            }
            break;
        }
    }
    points[i] = pt;
}

__global__ __launch_bounds__(256) void track_summary_kernel(const adsb_frame *frames, const adsb_packet_fields *fields,
                                                            const adsb_track_point *points, const uint32_t *skeys,
                                                            const uint32_t *svals, const uint32_t *tail_flag,
                                                            const uint32_t *tail_pos, uint32_t n,
                                                            double seconds_per_sample, adsb_aircraft_record *out,
                                                            uint32_t max_aircraft, uint64_t *n_aircraft)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n || !tail_flag[s]) return;
    if (s + 1 == n) *n_aircraft = (uint64_t)tail_pos[s] + 1u;
    const uint32_t a = tail_pos[s];
    if (a >= max_aircraft) return;
    const uint32_t icao = skeys[s];
    adsb_aircraft_record r;
    r.icao = icao;
    r.altitude = 0;
    r.latitude = 0.0;
    r.longitude = 0.0;
    r.last_contact = __builtin_nan("");
    r.has_position = 0;
    r.n_frames = 0;
    for (int k = 0; k < 8; ++k) r.callsign[k] = 0;
    bool have_id = false, have_pos_msg = false;
    for (uint32_t w = s + 1; w > 0;) { // newest to oldest
        --w;
        if (skeys[w] != icao) break;
        const uint32_t j = svals[w];
        const adsb_packet_fields g = fields[j];
        ++r.n_frames;
        if (g.msg_kind == 0 && !have_id) { // aircraft.rs:105-107
            have_id = true;
            for (int k = 0; k < 8; ++k) r.callsign[k] = g.callsign[k];
        } else if (g.msg_kind == 1) {
            if (!have_pos_msg) { // aircraft.rs:55-56
                have_pos_msg = true;
                r.altitude = g.altitude;
                r.last_contact = (double)frames[j].offset * seconds_per_sample;
            }
            if (!r.has_position && (points[j].flags & ADSB_TRACK_NEW_POSITION)) { // aircraft.rs:97-102
                r.has_position = 1;
                r.latitude = points[j].latitude;
                r.longitude = points[j].longitude;
            }
        }
    }
    out[a] = r;
}

} // namespace

size_t track_sort_temp_bytes(size_t n)
{
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0, 24, (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u, n,
                                  rocprim::plus<uint32_t>(), (hipStream_t)0);
    return (sort_bytes > scan_bytes ? sort_bytes : scan_bytes) + 256;
}

hipError_t launch_track(hipStream_t st, const TrackArgs &a)
{
    if (a.n == 0) return hipMemsetAsync(a.n_aircraft, 0, sizeof(uint64_t), st);
    const uint32_t n = a.n, blocks = (n + 255) / 256;
    hipLaunchKernelGGL(track_keys_kernel, dim3(blocks), dim3(256), 0, st, a.fields, n, a.keys, a.vals);
    size_t tb = a.temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(a.temp, tb, (const uint32_t *)a.keys, a.skeys, (const uint32_t *)a.vals,
                                             a.svals, (size_t)n, 0, 24, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(track_pairs_kernel, dim3(blocks), dim3(256), 0, st, a.frames, a.fields, a.skeys, a.svals, n,
                       a.seconds_per_sample, a.points, a.keys /* reused: tail flags */);
    tb = a.temp_bytes;
    e = rocprim::exclusive_scan(a.temp, tb, (const uint32_t *)a.keys, a.vals /* reused: tail positions */, 0u, (size_t)n,
                                rocprim::plus<uint32_t>(), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(track_summary_kernel, dim3(blocks), dim3(256), 0, st, a.frames, a.fields, a.points, a.skeys,
                       a.svals, a.keys, a.vals, n, a.seconds_per_sample, a.aircraft, a.max_aircraft, a.n_aircraft);
    return hipGetLastError();
}

} // namespace adsbk
