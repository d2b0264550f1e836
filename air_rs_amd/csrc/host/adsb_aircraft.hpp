// adsb_aircraft.hpp -- C++ mirror of what sits right behind the AdsbPacket channel in the reference:
// the per-ICAO tracker (src/adsb/aircraft.rs) and the global CPR position decode it calls
// (src/adsb/cpr.rs).  Same names, same state machine, same arithmetic (f64), so that the display thread
// of the reference would see the same AircraftSummary values.  SURVEY section 8(f) rank 3.
#pragma once
#include <cstdint>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>

#include "adsb_packet.hpp"

namespace air_rs_amd {

// cpr.rs:9-15
struct GeographicPosition {
    double latitude = 0.0;
    double longitude = 0.0;
};

// cpr.rs:39-54 / 63-88 / 90-127 / 135-147
uint32_t calc_num_zones(double lat);
void calculate_latitude(uint32_t even_cpr_lat, uint32_t odd_cpr_lat, CprFormat first, double &latitude,
                        double &even_latitude, double &odd_latitude);
double calculate_longitude(uint32_t even_cpr_long, uint32_t odd_cpr_long, double latitude, CprFormat first);
std::optional<GeographicPosition> calculate_geographic_position(uint32_t even_lat, uint32_t even_lon, uint32_t odd_lat,
                                                                uint32_t odd_lon, CprFormat first);

// aircraft.rs:14-23
struct AircraftSummary {
    uint32_t icao = 0;
    std::string callsign;
    int32_t altitude = 0;
    std::optional<GeographicPosition> geo_position;
    double last_contact = 0.0; // seconds (the reference: Unix timestamp of a wall-clock DateTime)
};

// aircraft.rs:26-37.  Times are seconds on whatever clock the caller stamps packets with (the
// reference uses AdsbPacket::time_processed, a wall clock; the device path uses sample offset / rate).
class Aircraft {
public:
    explicit Aircraft(uint32_t icao, double now = 0.0); // aircraft.rs:40-46
    // aircraft.rs:48-111; returns true when geo_position was recomputed by this packet
    bool handle_packet(const AdsbPacket &msg, double time_processed);
    uint32_t get_icao() const { return icao; }
    std::string get_callsign() const { return callsign.value_or(""); } // aircraft.rs:117-124
    int32_t get_altitude_ft() const { return altitude; }
    std::optional<GeographicPosition> get_geo_position() const { return geo_position; }
    AircraftSummary get_summary() const; // aircraft.rs:142-152

private:
    struct Cpr {
        uint32_t lat = 0, lon = 0;
    };
    uint32_t icao;
    std::optional<std::string> callsign;
    int32_t altitude = 0;
    std::optional<GeographicPosition> geo_position;
    double last_contact;
    std::optional<Cpr> last_odd_packet, last_even_packet;
    double last_odd_processed, last_even_processed;
};

// aircraft.rs:158-165: inserts the aircraft if new, lets it handle the packet, returns a copy
Aircraft handle_aircraft_update(const AdsbPacket &packet, double time_processed,
                                std::unordered_map<uint32_t, Aircraft> &aircrafts, bool *new_position = nullptr);

} // namespace air_rs_amd
