// derive.cpp -- DerivedKey assembly (src/derive.rs:94-165) from what the device produced, and
// Matcher (src/matcher.rs:45-108).
#include "vuke_host.hpp"

#include <cstring>
#include <thread>

#include <cctype>
#include <cstdio>

#include <cstring>
#include <fstream>
#include <stdexcept>

namespace vuke {

Network parse_network(const std::string &raw)
{ // src/network.rs:6-17
    std::string s = raw;
    for (char &c : s) c = (char)tolower((unsigned char)c);
    if (s == "bitcoin" || s == "mainnet" || s == "main") return Network::Bitcoin;
    if (s == "testnet" || s == "test") return Network::Testnet;
    if (s == "signet") return Network::Signet;
    if (s == "regtest" || s == "reg") return Network::Regtest;
    fprintf(stderr, "Unknown network: %s. Defaulting to Bitcoin.\n", raw.c_str());
    return Network::Bitcoin;
}

DerivedKey KeyDeriver::from_device(const uint8_t k[32], const uint8_t hc[20], const uint8_t hu[20], Network net)
{
    // version bytes and bech32 prefix of the network (rust-bitcoin: PrivateKey::to_wif, Address::p2pkh / p2wpkh)
    const bool main = net == Network::Bitcoin;
    const uint8_t wif_ver = main ? 0x80 : 0xEF, p2pkh_ver = main ? 0x00 : 0x6F;
    const char *hrp = main ? "bc" : net == Network::Regtest ? "bcrt" : "tb";
    DerivedKey d;
    memcpy(d.raw.data(), k, 32);
    memcpy(d.h160_c, hc, 20);
    memcpy(d.h160_u, hu, 20);
    d.private_key_hex = hex_encode(k, 32);                   // :147
    d.private_key_decimal = decimal_of_be(k);                // :121
    d.private_key_binary.resize(256);                        // :124-131
    int lz = -1, hw = 0;
    for (int i = 0; i < 256; i++) {
        int bit = (k[i / 8] >> (7 - i % 8)) & 1;
        d.private_key_binary[i] = bit ? '1' : '0';
        hw += bit;
        if (bit && lz < 0) lz = i;
    }
    d.bit_length = (uint16_t)(256 - (lz < 0 ? 256 : lz));    // :134-141
    d.hamming_weight = (uint16_t)hw;                         // :144
    size_t l0 = 0;                                           // :148
    while (l0 < 64 && d.private_key_hex[l0] == '0') l0++;
    d.leading_zeros = (uint8_t)l0;
    uint8_t buf[40];
    buf[0] = wif_ver;                                        // :102-105,159-160
    memcpy(buf + 1, k, 32);
    buf[33] = 0x01;
    d.wif_compressed = base58check_encode(buf, 34);
    d.wif_uncompressed = base58check_encode(buf, 33);
    buf[0] = p2pkh_ver;                                      // :112-113
    memcpy(buf + 1, hc, 20);
    d.p2pkh_compressed = base58check_encode(buf, 21);
    memcpy(buf + 1, hu, 20);
    d.p2pkh_uncompressed = base58check_encode(buf, 21);
    d.p2wpkh = segwit_encode(hrp, 0, hc, 20);                // :116-118
    return d;
}

DerivedKey KeyDeriver::from_device_compact(const uint8_t k[32], const uint8_t hc[20], const uint8_t hu[20])
{
    DerivedKey d;
    memcpy(d.raw.data(), k, 32);
    memcpy(d.h160_c, hc, 20);
    memcpy(d.h160_u, hu, 20);
    d.private_key_hex = hex_encode(k, 32);
    uint8_t buf[21];
    buf[0] = 0x00;
    memcpy(buf + 1, hc, 20);
    d.p2pkh_compressed = base58check_encode(buf, 21);
    return d;
}

void KeyDeriver::set_pubkeys(DerivedKey &d, const uint8_t xy[64])
{ // src/derive.rs:98-99,150-151: serialize() / serialize_uncompressed() as hex
    d.pubkey_compressed = ((xy[63] & 1) ? "03" : "02") + hex_encode(xy, 32);
    d.pubkey_uncompressed = "04" + hex_encode(xy, 64);
}

std::vector<DerivedKey> KeyDeriver::derive(const std::vector<Key> &keys) const
{
    std::vector<DerivedKey> out;
    if (keys.empty()) return out;
    const size_t n = keys.size();
    std::vector<uint8_t> in(n * 32), norm(n * 32), hc(n * 20), hu(n * 20), xy(n * 64);
    for (size_t i = 0; i < n; i++) memcpy(&in[32 * i], keys[i].data(), 32);
    check_rc(engine_, vk_derive_batch_pub(engine_, in.data(), n, norm.data(), hc.data(), hu.data(), xy.data()));
    out.resize(n);
    for (size_t i = 0; i < n; i++) {
        out[i] = from_device(&norm[32 * i], &hc[20 * i], &hu[20 * i], net_);
        set_pubkeys(out[i], &xy[64 * i]);
    }
    return out;
}

const char *address_type_str(AddressType t)
{
    switch (t) {
    case AddressType::P2pkhCompressed: return "p2pkh_compressed";
    case AddressType::P2pkhUncompressed: return "p2pkh_uncompressed";
    default: return "p2wpkh";
    }
}

Matcher Matcher::load(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open targets file: " + path);
    Matcher m;
    {   // a million-line file: size the set once instead of rehashing it twenty times on the way up
        f.seekg(0, std::ios::end);
        const std::streamoff bytes = f.tellg();
        f.seekg(0, std::ios::beg);
        if (bytes > 0) m.targets_.reserve((size_t)bytes / 30 + 16);
    }
    std::string line;
    while (std::getline(f, line)) { // BufRead::lines: split at '\n'; `line?` propagates the error of a line that is not UTF-8
        if (!utf8_valid(line.data(), line.size())) throw std::runtime_error("stream did not contain valid UTF-8");
        const char *a = line.data(), *e = a + line.size();
        trim_unicode(a, e); // str::trim()
        if (a != e && *a != '#') m.targets_.insert(std::string(a, e));
    }
    return m;
}

Matcher Matcher::from_addresses(const std::vector<std::string> &a)
{
    Matcher m;
    for (auto &s : a) m.targets_.insert(s);
    return m;
}

std::optional<MatchInfo> Matcher::check(const DerivedKey &d) const
{
    if (targets_.count(d.p2pkh_compressed)) return MatchInfo{AddressType::P2pkhCompressed, d.p2pkh_compressed};
    if (targets_.count(d.p2pkh_uncompressed)) return MatchInfo{AddressType::P2pkhUncompressed, d.p2pkh_uncompressed};
    if (targets_.count(d.p2wpkh)) return MatchInfo{AddressType::P2wpkh, d.p2wpkh};
    return std::nullopt;
}

// One target line -> (hash160, kind) if it is the canonical spelling of a mainnet P2PKH / P2WPKH address.  The reference compares
// strings: only the canonical spelling can equal a derived address, so decode, re-encode and require identity (drops e.g.
// upper-case bech32).
static bool decode_target(const std::string &s, uint8_t out[20], uint8_t &kind)
{
    std::vector<uint8_t> raw;
    if (base58check_decode(s, raw) && raw.size() == 21 && raw[0] == 0x00) {
        if (base58check_encode(raw.data(), raw.size()) != s) return false;
        memcpy(out, raw.data() + 1, 20);
        kind = VK_TARGET_P2PKH;
        return true;
    }
    int ver = -1;
    std::vector<uint8_t> prog;
    if (segwit_decode("bc", s, ver, prog) && ver == 0 && prog.size() == 20) {
        if (segwit_encode("bc", 0, prog.data(), 20) != s) return false;
        memcpy(out, prog.data(), 20);
        kind = VK_TARGET_P2WPKH;
        return true;
    }
    return false;
}

void Matcher::prepare_device_targets()
{
    decode_all(dev_h160_, dev_kind_);
    dev_ready_ = true;
}

void Matcher::device_targets(std::vector<uint8_t> &h160, std::vector<uint8_t> &kind) const
{
    if (dev_ready_) {
        h160 = dev_h160_;
        kind = dev_kind_;
        return;
    }
    decode_all(h160, kind);
}

void Matcher::decode_all(std::vector<uint8_t> &h160, std::vector<uint8_t> &kind) const
{
    // a million lines cost two SHA-256 pairs and two base conversions each: decoded on every host core (round 3; one thread took
    // 1.6 of the 3.0 s of BASELINE config 3 through the CLI)
    std::vector<const std::string *> lines;
    lines.reserve(targets_.size());
    for (const std::string &s : targets_) lines.push_back(&s);
    const size_t n = lines.size();
    std::vector<uint8_t> all(n * 20), kinds(n), ok(n, 0);
    unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    if (n < 4096) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t]() {
            for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) ok[i] = decode_target(*lines[i], &all[20 * i], kinds[i]) ? 1 : 0;
        });
    for (auto &x : th) x.join();
    h160.clear();
    kind.clear();
    for (size_t i = 0; i < n; i++)
        if (ok[i]) {
            h160.insert(h160.end(), all.begin() + 20 * i, all.begin() + 20 * i + 20);
            kind.push_back(kinds[i]);
        }
}

} // namespace vuke
