// adsb_aircraft.cpp -- see adsb_aircraft.hpp.  Reference: src/adsb/cpr.rs, src/adsb/aircraft.rs.
#include "adsb_aircraft.hpp"

#include <cmath>

namespace air_rs_amd {

namespace {
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double NUM_ZONES = 15.0; // cpr.rs:19

double convert_cpr_to_float(uint32_t cpr) { return static_cast<double>(cpr) / 131072.0; } // cpr.rs:22-25

double normalize_longitude(double lon) // cpr.rs:27-31
{
    while (lon < -180.0) lon += 360.0;
    while (lon > 180.0) lon -= 360.0;
    return lon;
}

uint32_t floor_as_u32(double x) // Rust `x.floor() as u32`: saturating, NaN -> 0
{
    const double f = std::floor(x);
    if (!(f >= 0.0)) return 0u;
    if (f >= 4294967295.0) return 4294967295u;
    return static_cast<uint32_t>(f);
}
} // namespace

uint32_t calc_num_zones(double lat)
{
    if (lat == 0.0) return 59;
    if (lat == 87.0 || lat == -87.0) return 2;
    if (lat < -87.0 || lat > 87.0) return 1;
    const double int1 = 1.0 - std::cos(kPi / (2.0 * NUM_ZONES));
    const double int2 = std::cos(kPi / 180.0 * lat);
    const double int3 = (2.0 * kPi) / std::acos(1.0 - (int1 / (int2 * int2)));
    return floor_as_u32(int3);
}

void calculate_latitude(uint32_t even_cpr_lat_u, uint32_t odd_cpr_lat_u, CprFormat first, double &latitude,
                        double &even_latitude, double &odd_latitude)
{
    constexpr double EVEN_LAT_DIVISIONS = 360.0 / (4.0 * NUM_ZONES);
    constexpr double ODD_LAT_DIVISIONS = 360.0 / (4.0 * NUM_ZONES - 1.0);
    const double even_cpr_lat = convert_cpr_to_float(even_cpr_lat_u);
    const double odd_cpr_lat = convert_cpr_to_float(odd_cpr_lat_u);
    const double latitude_index = std::floor(59.0 * even_cpr_lat - 60.0 * odd_cpr_lat + 0.5);
    even_latitude = EVEN_LAT_DIVISIONS * (std::fmod(latitude_index, 60.0) + even_cpr_lat); // Rust % = fmod
    odd_latitude = ODD_LAT_DIVISIONS * (std::fmod(latitude_index, 59.0) + odd_cpr_lat);
    latitude = first == CprFormat::Even ? odd_latitude : even_latitude; // the newest format decides
    if (latitude > 270.0) latitude -= 360.0;
}

double calculate_longitude(uint32_t even_cpr_long, uint32_t odd_cpr_long, double latitude, CprFormat first)
{
    const double lon_cpr_e = convert_cpr_to_float(even_cpr_long);
    const double lon_cpr_o = convert_cpr_to_float(odd_cpr_long);
    const uint32_t nl = calc_num_zones(latitude);
    uint32_t nz = first == CprFormat::Even ? calc_num_zones(latitude - 1.0) // later is odd (sic: latitude - 1.0)
                                           : calc_num_zones(latitude);      // later is even
    if (nz < 1) nz = 1;
    const double num_zones = static_cast<double>(nz);
    const double divisions = 360.0 / num_zones;
    const double m = std::floor(lon_cpr_e * static_cast<double>(static_cast<uint32_t>(nl - 1u)) -
                                lon_cpr_o * static_cast<double>(nl) + 0.5);
    const double longitude = first == CprFormat::Even ? divisions * (std::fmod(m, num_zones) + lon_cpr_o)
                                                      : divisions * (std::fmod(m, num_zones) + lon_cpr_e);
    return normalize_longitude(longitude);
}

std::optional<GeographicPosition> calculate_geographic_position(uint32_t even_lat, uint32_t even_lon, uint32_t odd_lat,
                                                                uint32_t odd_lon, CprFormat first)
{
    double latitude, even_latitude, odd_latitude;
    calculate_latitude(even_lat, odd_lat, first, latitude, even_latitude, odd_latitude);
    if (calc_num_zones(even_latitude) != calc_num_zones(odd_latitude)) return std::nullopt; // cpr.rs:138-141
    GeographicPosition g;
    g.latitude = latitude;
    g.longitude = calculate_longitude(even_lon, odd_lon, latitude, first);
    return g;
}

Aircraft::Aircraft(uint32_t icao_, double now)
    : icao(icao_), last_contact(now), last_odd_processed(now), last_even_processed(now)
{
}

bool Aircraft::handle_packet(const AdsbPacket &msg, double time_processed)
{
    if (msg.get_icao() != icao) return false;
    if (const auto *pos = std::get_if<AircraftPosition>(&msg.msg)) {
        altitude = pos->get_altitude_ft();
        last_contact = time_processed;
        Cpr cpr_odd, cpr_even;
        CprFormat first;
        if (pos->get_cpr_format() == CprFormat::Even) {
            last_even_packet = Cpr{pos->cpr_latitude, pos->cpr_longitude};
            last_even_processed = time_processed;
            if (!last_odd_packet) return false;
            if (std::fabs(time_processed - last_odd_processed) > 10.0) return false;
            cpr_odd = *last_odd_packet;
            cpr_even = Cpr{pos->cpr_latitude, pos->cpr_longitude};
            first = CprFormat::Odd;
        } else {
            last_odd_packet = Cpr{pos->cpr_latitude, pos->cpr_longitude};
            last_odd_processed = time_processed;
            if (!last_even_packet) return false;
            if (std::fabs(time_processed - last_even_processed) > 10.0) return false;
            cpr_odd = Cpr{pos->cpr_latitude, pos->cpr_longitude};
            cpr_even = *last_even_packet;
            first = CprFormat::Even;
        }
        if (auto g = calculate_geographic_position(cpr_even.lat, cpr_even.lon, cpr_odd.lat, cpr_odd.lon, first)) {
            geo_position = g;
            return true;
        }
        return false;
    }
    if (const auto *id = std::get_if<AircraftID>(&msg.msg)) callsign = id->get_callsign();
    return false;
}

AircraftSummary Aircraft::get_summary() const
{
    AircraftSummary s;
    s.icao = icao;
    s.callsign = get_callsign();
    s.altitude = altitude;
    s.geo_position = geo_position;
    s.last_contact = last_contact;
    return s;
}

Aircraft handle_aircraft_update(const AdsbPacket &packet, double time_processed,
                                std::unordered_map<uint32_t, Aircraft> &aircrafts, bool *new_position)
{
    const uint32_t icao = packet.get_icao();
    auto it = aircrafts.find(icao);
    if (it == aircrafts.end()) it = aircrafts.emplace(icao, Aircraft(icao, time_processed)).first;
    const bool np = it->second.handle_packet(packet, time_processed);
    if (new_position) *new_position = np;
    return it->second;
}

} // namespace air_rs_amd
