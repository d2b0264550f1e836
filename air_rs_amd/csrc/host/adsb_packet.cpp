// adsb_packet.cpp -- field decode and Display text of AdsbPacket (reference:
// src/adsb/packet.rs:25-49,77-99 and src/adsb/msgs.rs:70-102,150-201).
#include "adsb_packet.hpp"

#include <cstdio>
#include <ctime>
#include <stdexcept>

namespace air_rs_amd {

namespace {

// msgs.rs:172-177: 6-bit ICAO character set, '#' marks the unused codes, '_' is the space.
const char kCharset[65] = "#ABCDEFGHIJKLMNOPQRSTUVWXYZ#####_###############0123456789######";

std::array<uint8_t, 7> me_field(const std::vector<uint8_t> &p)
{
    std::array<uint8_t, 7> m{};
    for (int k = 0; k < 7; ++k) m[k] = p.at(4 + k); // packet[4..4+7]
    return m;
}

std::string line(const char *label, const std::string &v) { return std::string(label) + v + "\n"; }

} // namespace

AircraftPosition::AircraftPosition(const std::array<uint8_t, 7> &msg) : raw_msg(msg)
{
    // 12-bit altitude code with the Q bit (bit 0 of msg[1]) squeezed out; Q=1: 25 ft steps.
    const bool step25 = (msg[1] & 0x01) == 1;
    int32_t code = (static_cast<int32_t>(msg[1] >> 1) << 4) | (msg[2] >> 4);
    altitude = code * (step25 ? 25 : 100) - 1000;

    msg_type = msg[0] >> 3;
    surveillance_status = (msg[0] >> 1) & 0x3;
    nic_supplement = msg[0] & 0x1;
    cpr_time = (msg[2] >> 3) & 0x1;
    cpr_format = ((msg[2] >> 2) & 0x1) ? CprFormat::Odd : CprFormat::Even;
    cpr_latitude = (static_cast<uint32_t>(msg[2] & 0x3) << 15) | (static_cast<uint32_t>(msg[3]) << 7) |
                   (msg[4] >> 1);
    cpr_longitude = (static_cast<uint32_t>(msg[4] & 0x1) << 16) | (static_cast<uint32_t>(msg[5]) << 8) |
                    msg[6];
}

std::string AircraftPosition::to_string() const
{
    std::string s = "Message:\n";
    s += line("Type                : ", std::to_string(msg_type) + " (Position)");
    s += line("Surveillance Status : ", std::to_string(surveillance_status));
    s += line("NIC Supplement      : ", std::to_string(nic_supplement));
    s += line("Altitude (ft)       : ", std::to_string(altitude));
    s += line("CPR Time            : ", std::to_string(cpr_time));
    s += line("CPR Format          : ", cpr_format == CprFormat::Odd ? "Odd" : "Even");
    s += line("Raw Latitude        : ", std::to_string(cpr_latitude));
    s += line("Raw Longitude       : ", std::to_string(cpr_longitude));
    return s;
}

AircraftID::AircraftID(const std::array<uint8_t, 7> &msg) : raw_msg(msg)
{
    // 48 bits after the type byte -> eight 6-bit characters (msgs.rs:150-170 regroups the same
    // bits with a running accumulator).
    uint64_t bits = 0;
    for (int k = 1; k < 7; ++k) bits = (bits << 8) | msg[k];
    for (int c = 0; c < 8; ++c) callsign.push_back(kCharset[(bits >> (42 - 6 * c)) & 0x3F]);
    msg_type = msg[0] >> 3;
}

std::string AircraftID::to_string() const
{
    std::string s = "Message:\n";
    s += line("Type                : ", std::to_string(msg_type) + " (ID)");
    s += line("Callsign            : ", callsign);
    return s;
}

std::string UknownMsg::to_string() const
{
    std::string s = "Message:\nType    : Unknown\nRaw Msg :  [";
    for (size_t k = 0; k < raw_msg.size(); ++k) {
        if (k) s += ", ";
        s += std::to_string(raw_msg[k]);
    }
    s += "]\n";
    return s;
}

static AdsbMsgType decode_msg(const std::vector<uint8_t> &p, uint8_t tc)
{
    if (AircraftID::msg_id_match(tc)) return AircraftID(me_field(p));
    if (AircraftPosition::msg_id_match(tc)) return AircraftPosition(me_field(p));
    return UknownMsg{std::vector<uint8_t>(p.begin() + 4, p.end())}; // packet.rs:37
}

AdsbPacket::AdsbPacket(const std::vector<uint8_t> &p)
    : icao((static_cast<uint32_t>(p.at(1)) << 16) | (static_cast<uint32_t>(p.at(2)) << 8) | p.at(3)),
      msg_type(p.at(4) >> 3),
      msg(decode_msg(p, p.at(4) >> 3)),
      time_processed(std::chrono::system_clock::now()),
      packet(p),
      downlink_format(p.at(0) >> 3),
      capability(p.at(0) & 5) // sic: the reference masks with 5, not 7 (packet.rs:27)
{
}

AdsbPacket AdsbPacket::new_from_string(const std::string &hex)
{
    if (hex.size() % 2) throw std::invalid_argument("Invalid hex string in packet");
    std::vector<uint8_t> bytes;
    for (size_t k = 0; k < hex.size(); k += 2) {
        size_t used = 0;
        int v = std::stoi(hex.substr(k, 2), &used, 16);
        if (used != 2) throw std::invalid_argument("Invalid hex string in packet");
        bytes.push_back(static_cast<uint8_t>(v));
    }
    return AdsbPacket(bytes);
}

std::string AdsbPacket::to_string(const char *time_text) const
{
    char buf[64];
    std::string s = "== ";
    for (uint8_t b : packet) {
        std::snprintf(buf, sizeof buf, "%02x", b);
        s += buf;
    }
    s += " ==\n";
    s += "Decoded Information:\n";
    s += line("Downlink Format : ", std::to_string(downlink_format));
    s += line("Capability      : ", std::to_string(capability));
    std::snprintf(buf, sizeof buf, "%06X", icao);
    s += line("ICAO            : ", buf);
    if (time_text) {
        s += line("Processed Time  : ", time_text);
    } else {
        // chrono's `impl Display for DateTime<Local>` (what packet.rs:94 prints, e.g.
        // "2025-07-26 07:47:16.818387100 +12:00"): date, time, the fraction in 0 / 3 / 6 / 9 digits -- the fewest
        // that lose nothing -- and the UTC offset with a colon
        using namespace std::chrono;
        const auto since = time_processed.time_since_epoch();
        std::time_t t = system_clock::to_time_t(time_processed);
        const long long ns = duration_cast<nanoseconds>(since - duration_cast<seconds>(since)).count();
        std::tm tmv{};
        localtime_r(&t, &tmv);
        size_t k = std::strftime(buf, sizeof buf, "%Y-%m-%d %H:%M:%S", &tmv);
        if (ns % 1000000 == 0 && ns != 0) k += (size_t)std::snprintf(buf + k, sizeof buf - k, ".%03lld", ns / 1000000);
        else if (ns % 1000 == 0 && ns != 0) k += (size_t)std::snprintf(buf + k, sizeof buf - k, ".%06lld", ns / 1000);
        else if (ns != 0) k += (size_t)std::snprintf(buf + k, sizeof buf - k, ".%09lld", ns);
        char off[8];
        std::strftime(off, sizeof off, "%z", &tmv); // +hhmm
        std::snprintf(buf + k, sizeof buf - k, " %.3s:%.2s", off, off + 3);
        s += line("Processed Time  : ", buf);
    }
    s += line("Message Type    : ", std::to_string(msg_type));
    s += std::visit([](const auto &m) { return m.to_string(); }, msg);
    return s;
}

} // namespace air_rs_amd
