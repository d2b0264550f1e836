// adsb_packet.hpp -- C++ mirror of the struct that crosses the channel between thread 2 and the
// display thread in the reference: AdsbPacket (src/adsb/packet.rs:9-18) and its message variants
// (src/adsb/msgs.rs:6-11).  Same field names, same decode rules, same Display text, so code
// downstream of the channel (tracker, stream printer) sees what it saw before.
#pragma once
#include <array>
#include <chrono>
#include <cstdint>
#include <string>
#include <variant>
#include <vector>

namespace air_rs_amd {

// msgs.rs:54-59
enum class CprFormat { Even, Odd };

// msgs.rs:61-76, decode at msgs.rs:70-102
struct AircraftPosition {
    std::array<uint8_t, 7> raw_msg{};
    uint8_t msg_type = 0;
    uint8_t surveillance_status = 0;
    uint8_t nic_supplement = 0;
    int32_t altitude = 0; // feet
    uint8_t cpr_time = 0;
    CprFormat cpr_format = CprFormat::Even;
    uint32_t cpr_latitude = 0;
    uint32_t cpr_longitude = 0;

    explicit AircraftPosition(const std::array<uint8_t, 7> &msg);
    static bool msg_id_match(uint8_t id) { return id >= 9 && id <= 18; } // msgs.rs:122-124
    int32_t get_altitude_ft() const { return altitude; }
    CprFormat get_cpr_format() const { return cpr_format; }
    std::string to_string() const; // msgs.rs:127-140
};

// msgs.rs:143-148, decode at msgs.rs:180-201
struct AircraftID {
    std::array<uint8_t, 7> raw_msg{};
    uint8_t msg_type = 0;
    std::string callsign;

    explicit AircraftID(const std::array<uint8_t, 7> &msg);
    static bool msg_id_match(uint8_t id) { return id >= 1 && id <= 4; } // msgs.rs:210-212
    std::string get_callsign() const { return callsign; }
    std::string to_string() const; // msgs.rs:215-223
};

// msgs.rs:31-34 (the reference spells it "Uknown")
struct UknownMsg {
    std::vector<uint8_t> raw_msg;
    std::string to_string() const; // msgs.rs:36-44
};

using AdsbMsgType = std::variant<AircraftID, AircraftPosition, UknownMsg>;

// packet.rs:9-18
class AdsbPacket {
public:
    // packet.rs:25-49; time_processed = now (excluded from parity: wall clock)
    explicit AdsbPacket(const std::vector<uint8_t> &packet);
    // packet.rs:56-68
    static AdsbPacket new_from_string(const std::string &hex);

    uint32_t get_icao() const { return icao; } // packet.rs:72-74
    // packet.rs:77-99.  time_text: what to print on the "Processed Time" line (the reference
    // prints chrono's Local::now()); nullptr prints the packet's own timestamp.
    std::string to_string(const char *time_text = nullptr) const;

    const std::vector<uint8_t> &bytes() const { return packet; }
    uint8_t get_downlink_format() const { return downlink_format; }
    uint8_t get_capability() const { return capability; }

    uint32_t icao = 0;
    uint8_t msg_type = 0;
    AdsbMsgType msg;
    std::chrono::system_clock::time_point time_processed;

private:
    std::vector<uint8_t> packet;
    uint8_t downlink_format = 0;
    uint8_t capability = 0;
};

} // namespace air_rs_amd
