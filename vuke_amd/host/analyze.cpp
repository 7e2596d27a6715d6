// analyze.cpp -- host mirror of the reference's analyzer interface for the generators whose seed space the
// GPU can sweep (src/analyze/mod.rs:36-129, milksad.rs, lcg.rs, mt64.rs, xorshift.rs, sha256_chain.rs,
// key_parser.rs:32-215, output.rs:3-25).
// The search runs on the device (vk_search); this file parses keys / cascades and words the results the way the
// reference does.  Where several seeds match (masked / cascade), the reference reports whichever a worker
// thread finds first; here it is the smallest.
#include "vuke_host.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cerrno>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace vuke {

const char *AnalysisResult::symbol() const
{
    switch (status) {
    case AnalysisStatus::Confirmed: return "\xE2\x9C\x93"; // ✓
    case AnalysisStatus::NotFound: return "\xE2\x9C\x97";  // ✗
    default: return "?";
    }
}
const char *AnalysisResult::status_str() const
{
    switch (status) {
    case AnalysisStatus::Confirmed: return "confirmed";
    case AnalysisStatus::Possible: return "possible";
    case AnalysisStatus::NotFound: return "not_found";
    default: return "unknown";
    }
}

static std::string hexu(uint64_t v)
{
    char b[32];
    snprintf(b, sizeof b, "%llx", (unsigned long long)v);
    return b;
}

static vk_search_result run(vk_engine *e, const vk_transform_spec &sp, uint64_t lo, uint64_t hi, const vk_search_target &tg)
{
    vk_search_result r;
    check_rc(e, vk_search(e, &sp, lo, hi, &tg, &r));
    return r;
}

static vk_search_target target_of(const Key &key, const AnalysisConfig &cfg)
{
    vk_search_target tg;
    memset(&tg, 0, sizeof tg);
    if (cfg.cascade_targets) {
        tg.mode = VK_SEARCH_CASCADE;
        if (cfg.cascade_targets->size() > 8) throw std::invalid_argument("cascade of more than 8 targets is not supported on the device path");
        tg.n_cascade = (uint32_t)cfg.cascade_targets->size();
        for (size_t j = 0; j < cfg.cascade_targets->size(); j++) {
            tg.bits[j] = (*cfg.cascade_targets)[j].first;
            tg.targets[j] = (*cfg.cascade_targets)[j].second;
        }
    } else if (cfg.mask_bits) {
        tg.mode = VK_SEARCH_MASKED;
        tg.mask_bits = *cfg.mask_bits;
        memcpy(tg.key, key.data(), 32);
    } else {
        tg.mode = VK_SEARCH_EXACT;
        memcpy(tg.key, key.data(), 32);
    }
    return tg;
}

static std::string cascade_desc(const AnalysisConfig &cfg)
{
    std::string s;
    for (auto &bt : *cfg.cascade_targets) s += (s.empty() ? "" : ",") + std::string("P") + std::to_string(bt.first) + ":0x" + hexu(bt.second);
    return s;
}

static std::string masked_details(const std::string &prefix, const vk_search_result &r, uint8_t bits, const std::string &suffix)
{
    const uint64_t mask = bits >= 64 ? ~0ull : (1ull << bits) - 1, high = 1ull << (bits - 1);
    uint64_t k = 0;
    for (int i = 24; i < 32; i++) k = (k << 8) | r.keys[0][i];
    return prefix + "full_key=" + hex_encode(r.keys[0], 32) + ", masked=0x" + hexu((k & mask) | high) + ", mask_bits=" + std::to_string(bits) + suffix +
           ", formula=(key & 0x" + hexu(mask) + ") | 0x" + hexu(high);
}

// ---------------------------------------------------------------- milksad (src/analyze/milksad.rs)
AnalysisResult MilksadAnalyzer::analyze(const Key &key, const AnalysisConfig &cfg) const
{
    const vk_transform_spec sp{VK_T_MILKSAD, VK_VARIANT_ALL, 0, 0};
    const uint64_t total = 1ull << 32;
    const vk_search_result r = run(engine_, sp, 0, total - 1, target_of(key, cfg));
    if (cfg.cascade_targets) {
        if (!r.found) return {name(), AnalysisStatus::NotFound, "checked " + std::to_string(total) + " seeds, cascade=[" + cascade_desc(cfg) + "]"};
        char head[64];
        snprintf(head, sizeof head, "seed=%llu (0x%08llx)", (unsigned long long)r.seed, (unsigned long long)r.seed);
        std::string d = head; // format_cascade_result, :368-381
        for (size_t j = 0; j < cfg.cascade_targets->size(); j++)
            d += "\n  P" + std::to_string((*cfg.cascade_targets)[j].first) + ": target=0x" + hexu((*cfg.cascade_targets)[j].second) + ", full_key=" +
                 hex_encode(r.keys[j], 32);
        return {name(), AnalysisStatus::Confirmed, d};
    }
    if (cfg.mask_bits) {
        if (!r.found) return {name(), AnalysisStatus::NotFound, "checked " + std::to_string(total) + " seeds with " + std::to_string(*cfg.mask_bits) + "-bit mask"};
        return {name(), AnalysisStatus::Confirmed, masked_details("seed = " + std::to_string(r.seed) + ", ", r, *cfg.mask_bits, "")};
    }
    if (!r.found) return {name(), AnalysisStatus::NotFound, "checked " + std::to_string(total) + " seeds"};
    return {name(), AnalysisStatus::Confirmed, "seed = " + std::to_string(r.seed)};
}

// ---------------------------------------------------------------- lcg (src/analyze/lcg.rs)
AnalysisResult LcgAnalyzer::analyze(const Key &key, const AnalysisConfig &cfg) const
{
    static const char *NAMES[4] = {"glibc", "minstd", "msvc", "borland"};
    static const uint64_t MAXSEED[4] = {(1ull << 31) - 1, (1ull << 31) - 2, 0xFFFFFFFFull, 0xFFFFFFFFull};
    AnalysisConfig c = cfg;
    c.cascade_targets.reset(); // the lcg analyzer looks at mask_bits only (:36-47)
    if (c.mask_bits && *c.mask_bits == 0) c.mask_bits.reset();
    uint64_t checked = 0;
    std::string names;
    for (uint32_t v = 0; v < 4; v++) {
        if (variant_ != VK_VARIANT_ALL && variant_ != v) continue;
        const vk_transform_spec sp{VK_T_LCG, v, endian_, 0};
        const vk_search_result r = run(engine_, sp, 0, MAXSEED[v], target_of(key, c));
        const char *en = endian_ ? "le" : "be";
        if (r.found) {
            if (c.mask_bits)
                return {name(), AnalysisStatus::Confirmed,
                        masked_details(std::string("variant=") + NAMES[v] + ", seed=" + std::to_string(r.seed) + ", ", r, *c.mask_bits, std::string(", endian=") + en)};
            return {name(), AnalysisStatus::Confirmed, std::string("variant=") + NAMES[v] + ", seed=" + std::to_string(r.seed) + ", endian=" + en};
        }
        checked += MAXSEED[v] + 1;
        names += (names.empty() ? "" : ", ") + std::string(NAMES[v]);
    }
    return {name(), AnalysisStatus::NotFound, "checked " + std::to_string(checked) + " seeds across variants: " + names};
}

// ---------------------------------------------------------------- mt64 (src/analyze/mt64.rs)
AnalysisResult Mt64Analyzer::analyze(const Key &key, const AnalysisConfig &cfg) const
{
    if (!cfg.cascade_targets)
        return {name(), AnalysisStatus::Unknown,
                "64-bit seed space (2^64) requires --cascade filter. Without cascade, brute-force is infeasible. Example: --cascade 5:0x15,10:0x202,20:0xd2c55"};
    if (cfg.cascade_targets->empty()) return {name(), AnalysisStatus::Unknown, "Cascade filter is empty"};
    const vk_transform_spec sp{VK_T_MT64, VK_VARIANT_ALL, 0, 0};
    const auto t0 = std::chrono::steady_clock::now();
    const vk_search_result r = run(engine_, sp, 0, limit_, target_of(key, cfg));
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    char buf[128];
    if (!r.found) {
        snprintf(buf, sizeof buf, "searched %llu seeds in %.2fs, cascade=[", (unsigned long long)r.seeds_checked, secs);
        return {name(), AnalysisStatus::NotFound, std::string(buf) + cascade_desc(cfg) + "]"};
    }
    snprintf(buf, sizeof buf, "seed=%llu (0x%016llx)\nsearched %llu seeds in %.2fs", (unsigned long long)r.seed, (unsigned long long)r.seed,
             (unsigned long long)r.seeds_checked, secs);
    std::string d = buf; // format_cascade_result, :217-240
    for (size_t j = 0; j < cfg.cascade_targets->size(); j++)
        d += "\n  P" + std::to_string((*cfg.cascade_targets)[j].first) + ": target=0x" + hexu((*cfg.cascade_targets)[j].second) + ", full_key=" +
             hex_encode(r.keys[j], 32);
    return {name(), AnalysisStatus::Confirmed, d};
}

// ---------------------------------------------------------------- xorshift (src/analyze/xorshift.rs)
AnalysisResult XorshiftAnalyzer::analyze(const Key &key, const AnalysisConfig &cfg) const
{
    static const char *NAMES[4] = {"xorshift64", "xorshift128", "xorshift128+", "xoroshiro128**"};
    if (!cfg.cascade_targets)
        return {name(), AnalysisStatus::Unknown, "64-bit seed space requires --cascade filter. Example: --cascade 5:0x15,10:0x202,20:0xd2c55"};
    if (cfg.cascade_targets->empty()) return {name(), AnalysisStatus::Unknown, "Cascade filter is empty"};
    std::string names;
    for (uint32_t v = 0; v < 4; v++) {
        if (variant_ != VK_VARIANT_ALL && variant_ != v) continue;
        const vk_transform_spec sp{VK_T_XORSHIFT, v, 0, 0};
        const auto t0 = std::chrono::steady_clock::now();
        const vk_search_result r = run(engine_, sp, 0, limit_, target_of(key, cfg));
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (r.found) { // format_cascade_result, :208-232 (cascade_hits is a progress statistic the device search does not keep)
            char buf[192];
            snprintf(buf, sizeof buf, "variant=%s, seed=%llu (0x%016llx)\nsearched %llu seeds in %.2fs", NAMES[v], (unsigned long long)r.seed,
                     (unsigned long long)r.seed, (unsigned long long)r.seeds_checked, secs);
            std::string d = buf;
            for (size_t j = 0; j < cfg.cascade_targets->size(); j++)
                d += "\n  P" + std::to_string((*cfg.cascade_targets)[j].first) + ": target=0x" + hexu((*cfg.cascade_targets)[j].second) + ", full_key=" +
                     hex_encode(r.keys[j], 32);
            return {name(), AnalysisStatus::Confirmed, d};
        }
        names += (names.empty() ? "" : ", ") + std::string(NAMES[v]);
    }
    return {name(), AnalysisStatus::NotFound, "checked variants=[" + names + "], cascade=[" + cascade_desc(cfg) + "]"};
}

// ---------------------------------------------------------------- sha256_chain (src/analyze/sha256_chain.rs)
AnalysisResult Sha256ChainAnalyzer::analyze(const Key &key, const AnalysisConfig &cfg) const
{
    static const char *NAMES[4] = {"iterated", "indexed:be", "indexed:le", "counter"};
    const vk_transform_spec sp{VK_T_SHA256_CHAIN, variant_, 0, depth_};
    const uint64_t total = 1ull << 32;
    const vk_search_result r = run(engine_, sp, 0, total - 1, target_of(key, cfg));
    std::string names;
    for (uint32_t v = 0; v < 4; v++)
        if (variant_ == VK_VARIANT_ALL || variant_ == v) names += (names.empty() ? "" : ", ") + std::string(NAMES[v]);
    char head[96];
    snprintf(head, sizeof head, "seed=%llu (0x%08llx), variant=%s", (unsigned long long)r.seed, (unsigned long long)r.seed, NAMES[r.variant & 3]);
    if (cfg.cascade_targets) {
        if (!r.found)
            return {name(), AnalysisStatus::NotFound, "checked " + std::to_string(total) + " seeds, variants=[" + names + "], cascade=[" + cascade_desc(cfg) + "]"};
        std::string d = head; // format_cascade_result, :427-448
        for (size_t j = 0; j < cfg.cascade_targets->size(); j++)
            d += "\n  P" + std::to_string((*cfg.cascade_targets)[j].first) + ": target=0x" + hexu((*cfg.cascade_targets)[j].second) + ", full_key=" +
                 hex_encode(r.keys[j], 32);
        return {name(), AnalysisStatus::Confirmed, d};
    }
    if (cfg.mask_bits) {
        if (!r.found)
            return {name(), AnalysisStatus::NotFound, "checked " + std::to_string(total) + " seeds with " + std::to_string(*cfg.mask_bits) + "-bit mask, variants=[" + names +
                                                          "], chain_depth=" + std::to_string(depth_)};
        return {name(), AnalysisStatus::Confirmed, std::string(head) + ", chain_index=" + std::to_string(r.index) + ", full_key=" + hex_encode(r.keys[0], 32) +
                                                       ", mask_bits=" + std::to_string(*cfg.mask_bits)};
    }
    if (!r.found)
        return {name(), AnalysisStatus::NotFound, "checked " + std::to_string(total) + " seeds, variants=[" + names + "], chain_depth=" + std::to_string(depth_)};
    return {name(), AnalysisStatus::Confirmed, std::string(head) + ", chain_index=" + std::to_string(r.index)};
}

// ---------------------------------------------------------------- multibit-hd (src/analyze/multibit.rs)
AnalysisResult MultibitAnalyzer::analyze(const Key &key, const AnalysisConfig &) const
{
    if (!mnemonic_ && !file_) return {name(), AnalysisStatus::Unknown, "requires --mnemonic or --mnemonic-file to check"};
    check_rc(engine_, vk_multibit_passphrase_set(engine_, (const uint8_t *)passphrase_.data(), passphrase_.size()));
    struct Reset { vk_engine *e; ~Reset() { vk_multibit_passphrase_set(e, nullptr, 0); } } reset{engine_};
    const vk_transform_spec sp{VK_T_MULTIBIT, 0, 0, depth_};
    const Transform t(sp);
    // index of the first derivation of `m` that equals the key (check_mnemonic, :47-61), over a batch of mnemonics
    auto first_hit = [&](const std::vector<std::string> &batch, size_t &which, uint32_t &index) {
        std::vector<uint8_t> data;
        std::vector<uint32_t> offs{0};
        for (const std::string &m : batch) {
            const std::string u = m.size() <= 256 ? m : std::string(); // longer than any 24-word mnemonic of the list: cannot validate
            data.insert(data.end(), u.begin(), u.end());
            offs.push_back((uint32_t)data.size());
        }
        std::vector<uint8_t> keys(batch.size() * depth_ * 32), valid(batch.size() * depth_);
        check_rc(engine_, vk_transform_strings(engine_, &sp, data.data(), offs.data(), batch.size(), keys.data(), valid.data()));
        for (size_t i = 0; i < batch.size(); i++)
            for (uint32_t j = 0; j < depth_; j++)
                if (valid[i * depth_ + j] && memcmp(&keys[32 * (i * depth_ + j)], key.data(), 32) == 0) {
                    which = i;
                    index = j;
                    return true;
                }
        return false;
    };
    auto shown = [&](const std::string &m) { // truncate_mnemonic: the label without its "[m/0'/0/i]" suffix
        const std::string l = t.label(m, 0);
        return l.substr(0, l.rfind('['));
    };
    size_t which = 0;
    uint32_t index = 0;
    if (mnemonic_) { // analyze_single_mnemonic, :63-88
        if (first_hit({*mnemonic_}, which, index))
            return {name(), AnalysisStatus::Confirmed, "mnemonic=\"" + shown(*mnemonic_) + "\", path=m/0'/0/" + std::to_string(index) + ", passphrase=\"" +
                                                           (passphrase_.empty() ? "<empty>" : "<set>") + "\""};
        return {name(), AnalysisStatus::NotFound, "mnemonic does not produce this key (checked " + std::to_string(depth_) + " derivations)"};
    }
    std::ifstream f(*file_); // analyze_mnemonic_file, :90-150
    if (!f) return {name(), AnalysisStatus::Unknown, "Failed to open mnemonic file: " + std::string(strerror(errno))};
    std::vector<std::string> batch;
    std::vector<size_t> line_no;
    size_t lines = 0;
    std::string line;
    auto run = [&]() -> std::optional<AnalysisResult> {
        if (batch.empty()) return std::nullopt;
        if (first_hit(batch, which, index))
            return AnalysisResult{name(), AnalysisStatus::Confirmed,
                                  "mnemonic=\"" + shown(batch[which]) + "\", path=m/0'/0/" + std::to_string(index) + ", line=" + std::to_string(line_no[which])};
        batch.clear();
        line_no.clear();
        return std::nullopt;
    };
    auto ws = [](unsigned char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f' || c == '\v'; };
    while (std::getline(f, line)) {
        lines++;
        size_t a = 0, b = line.size();
        while (a < b && ws((unsigned char)line[a])) a++;
        while (b > a && ws((unsigned char)line[b - 1])) b--;
        if (a == b || line[a] == '#') continue;
        batch.emplace_back(line, a, b - a);
        line_no.push_back(lines);
        if (batch.size() == 65536)
            if (auto r = run()) return *r;
    }
    if (auto r = run()) return *r;
    return {name(), AnalysisStatus::NotFound, "checked " + std::to_string(lines) + " mnemonics \xC3\x97 " + std::to_string(depth_) + " derivations"};
}

// ---------------------------------------------------------------- direct / heuristic (host only)
uint16_t calculate_bit_length(const Key &key)
{
    for (int i = 0; i < 32; i++)
        if (key[i]) return (uint16_t)(256 - 8 * i - __builtin_clz((unsigned)key[i]) + 24);
    return 0;
}

AnalysisResult DirectAnalyzer::analyze(const Key &key, const AnalysisConfig &cfg) const
{ // src/analyze/direct.rs:11-66
    if (cfg.mask_bits) return {name(), AnalysisStatus::Unknown, "masked analysis not supported"};
    std::vector<std::string> obs;
    int lead = 0, trail = 0;
    while (lead < 32 && key[lead] == 0) lead++;
    while (trail < 32 && key[31 - trail] == 0) trail++;
    const uint16_t bits = calculate_bit_length(key);
    if (bits <= 64) obs.push_back("bit_length=" + std::to_string(bits) + ", fits in u64");
    if (lead >= 24) obs.push_back("BE padding detected (" + std::to_string(lead) + " leading zero bytes)");
    if (trail >= 24) obs.push_back("LE padding detected (" + std::to_string(trail) + " trailing zero bytes)");
    size_t nn = 0; // is_ascii_string, :69-79
    while (nn < 32 && key[nn] != 0) nn++;
    bool ascii = nn >= 3 && nn <= 31;
    for (size_t i = nn; i < 32 && ascii; i++) ascii = key[i] == 0;
    for (size_t i = 0; i < nn && ascii; i++) ascii = key[i] >= 0x20 && key[i] <= 0x7e;
    if (ascii) obs.push_back("ASCII string: \"" + std::string(key.begin(), key.begin() + (long)nn) + "\"");
    if (obs.empty()) return {name(), AnalysisStatus::NotFound, "no direct patterns detected"};
    std::string d;
    for (auto &o : obs) d += (d.empty() ? "" : ", ") + o;
    return {name(), AnalysisStatus::Possible, d};
}

AnalysisResult HeuristicAnalyzer::analyze(const Key &key, const AnalysisConfig &cfg) const
{ // src/analyze/heuristic.rs:11-52
    if (cfg.mask_bits) return {name(), AnalysisStatus::Unknown, "masked analysis not supported"};
    uint32_t counts[256] = {0};
    unsigned hamming = 0;
    for (uint8_t b : key) {
        counts[b]++;
        hamming += (unsigned)__builtin_popcount(b);
    }
    double entropy = 0.0; // Shannon entropy of the byte histogram, :55-69
    for (uint32_t c : counts)
        if (c) {
            const double p = c / 32.0;
            entropy -= p * std::log2(p);
        }
    bool repeating = false; // :71-82
    for (int len = 1; len <= 8 && !repeating; len++) {
        if (32 % len) continue;
        bool rep = true, same = true;
        for (int i = 0; i < 32 && rep; i++) rep = key[i] == key[i % len];
        for (int i = 1; i < len; i++) same = same && key[i] == key[0];
        repeating = rep && !same;
    }
    char buf[64];
    std::vector<std::string> obs;
    if (entropy < 4.0) {
        snprintf(buf, sizeof buf, "low entropy (%.2f)", entropy);
        obs.push_back(buf);
    }
    if (hamming < 104 || hamming > 152) obs.push_back("unusual hamming weight (" + std::to_string(hamming) + ")");
    if (repeating) obs.push_back("repeating byte pattern");
    if (obs.empty()) {
        snprintf(buf, sizeof buf, "entropy=%.2f, hamming=%u", entropy, hamming);
        return {name(), AnalysisStatus::Unknown, buf};
    }
    std::string d;
    for (auto &o : obs) d += (d.empty() ? "" : ", ") + o;
    return {name(), AnalysisStatus::Possible, d};
}

// ---------------------------------------------------------------- key_parser.rs
static bool valid_scalar(const Key &k)
{
    static const uint8_t N[32] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFE,
                                  0xBA, 0xAE, 0xDC, 0xE6, 0xAF, 0x48, 0xA0, 0x3B, 0xBF, 0xD2, 0x5E, 0x8C, 0xD0, 0x36, 0x41, 0x41};
    bool zero = true;
    for (uint8_t b : k) zero = zero && b == 0;
    return !zero && memcmp(k.data(), N, 32) < 0;
}

Key parse_private_key(const std::string &raw)
{ // WIF, then hex, then decimal (src/analyze/key_parser.rs:32-48)
    size_t a = raw.find_first_not_of(" \t\r\n"), b = raw.find_last_not_of(" \t\r\n");
    const std::string in = a == std::string::npos ? "" : raw.substr(a, b - a + 1);
    Key key{};
    if (!in.empty() && strchr("5KL9c", in[0])) { // try_parse_wif
        std::vector<uint8_t> v;
        if (base58check_decode(in, v) && (v[0] == 0x80 || v[0] == 0xEF) && (v.size() == 33 || (v.size() == 34 && v[33] == 0x01))) {
            memcpy(key.data(), v.data() + 1, 32);
            if (valid_scalar(key)) return key;
        }
    }
    std::string h = in.rfind("0x", 0) == 0 ? in.substr(2) : in; // try_parse_hex
    if (!h.empty() && h.size() <= 64 && std::all_of(h.begin(), h.end(), [](unsigned char c) { return std::isxdigit(c); })) {
        h = std::string(64 - h.size(), '0') + h;
        for (int i = 0; i < 32; i++) key[i] = (uint8_t)std::stoul(h.substr(2 * i, 2), nullptr, 16);
        return key;
    }
    if (!in.empty() && std::all_of(in.begin(), in.end(), [](unsigned char c) { return std::isdigit(c); })) { // try_parse_decimal
        uint32_t w[8] = {0};
        bool overflow = false;
        for (char c : in) {
            uint64_t carry = (uint64_t)(c - '0');
            for (int i = 7; i >= 0; i--) {
                uint64_t cur = (uint64_t)w[i] * 10 + carry;
                w[i] = (uint32_t)cur;
                carry = cur >> 32;
            }
            overflow = overflow || carry != 0;
        }
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 4; j++) key[4 * i + j] = (uint8_t)(w[i] >> (24 - 8 * j));
        if (!overflow && valid_scalar(key)) return key;
    }
    throw std::invalid_argument("Unknown key format (expected hex, WIF, or decimal)");
}

std::vector<std::pair<uint8_t, uint64_t>> parse_cascade(const std::string &raw)
{ // src/analyze/key_parser.rs:131-215
    std::vector<std::pair<uint8_t, uint64_t>> t;
    size_t pos = 0;
    auto trim = [](std::string s) {
        size_t a = s.find_first_not_of(" \t"), b = s.find_last_not_of(" \t");
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    };
    if (trim(raw).empty()) throw std::invalid_argument("Invalid cascade: empty input");
    for (;;) {
        size_t q = raw.find(',', pos);
        std::string part = trim(raw.substr(pos, q == std::string::npos ? std::string::npos : q - pos));
        if (!part.empty()) {
            size_t c = part.find(':');
            if (c == std::string::npos || part.find(':', c + 1) != std::string::npos)
                throw std::invalid_argument("Invalid cascade: expected 'bits:target', got '" + part + "'");
            std::string bs = trim(part.substr(0, c)), ts = trim(part.substr(c + 1));
            char *end = nullptr;
            unsigned long bits = strtoul(bs.c_str(), &end, 10);
            if (bs.empty() || *end || bits > 255) throw std::invalid_argument("Invalid cascade: invalid bits '" + bs + "': must be 1-64");
            if (bits == 0 || bits > 64) throw std::invalid_argument("Invalid cascade: bits must be 1-64, got " + std::to_string(bits));
            const bool ishex = ts.rfind("0x", 0) == 0 || ts.rfind("0X", 0) == 0;
            errno = 0;
            unsigned long long target = strtoull(ishex ? ts.c_str() + 2 : ts.c_str(), &end, ishex ? 16 : 10);
            if (ts.empty() || (ishex && ts.size() == 2) || *end || errno)
                throw std::invalid_argument(std::string("Invalid cascade: invalid ") + (ishex ? "hex" : "decimal") + " target '" + ts + "'");
            const uint64_t maxv = bits >= 64 ? ~0ull : (1ull << bits) - 1;
            if (target > maxv)
                throw std::invalid_argument("Invalid cascade: target 0x" + hexu(target) + " exceeds " + std::to_string(bits) + "-bit maximum (0x" + hexu(maxv) + ")");
            if (!((target >> (bits - 1)) & 1))
                throw std::invalid_argument("Invalid cascade: target 0x" + hexu(target) + " does not have bit " + std::to_string(bits - 1) + " set");
            for (auto &bt : t)
                if (bt.first == bits && bt.second == target) throw std::invalid_argument("Invalid cascade: duplicate target " + std::to_string(bits) + ":" + std::to_string(target));
            t.emplace_back((uint8_t)bits, (uint64_t)target);
        }
        if (q == std::string::npos) break;
        pos = q + 1;
    }
    if (t.size() < 2) throw std::invalid_argument("Invalid cascade: cascade requires at least 2 targets (use --mask for single target)");
    std::stable_sort(t.begin(), t.end(), [](auto &x, auto &y) { return x.first < y.first; });
    return t;
}

std::string format_results(const Key &key, const std::vector<AnalysisResult> &results)
{ // src/analyze/output.rs:3-25 with KeyMetadata::from_key (src/analyze/mod.rs:266-281)
    int lz = 0, hw = 0;
    bool seen = false;
    for (int i = 0; i < 256; i++) {
        int bit = (key[i / 8] >> (7 - i % 8)) & 1;
        hw += bit;
        if (bit) seen = true;
        if (!seen) lz++;
    }
    std::string out = "Private Key: " + hex_encode(key.data(), 32) + "\nBit Length:  " + std::to_string(256 - lz) + "\nHamming Weight: " + std::to_string(hw) +
                      "\n---\nAnalysis:\n";
    for (const AnalysisResult &r : results) {
        std::string st = r.status_str();
        std::transform(st.begin(), st.end(), st.begin(), [](unsigned char c) { return (char)std::toupper(c); });
        out += std::string("  ") + r.symbol() + " " + r.analyzer + ": " + st + " " + (r.details.empty() ? "" : "(" + r.details + ")") + "\n";
    }
    return out;
}

std::string format_results_json(const Key &key, const std::vector<AnalysisResult> &results)
{ // src/analyze/output.rs:27-78
    int hw = 0;
    for (uint8_t b : key) hw += __builtin_popcount(b);
    const std::string hex = hex_encode(key.data(), 32);
    size_t lz = 0;
    while (lz < 64 && hex[lz] == '0') lz++; // KeyMetadata::leading_zeros: zero hex digits (src/analyze/mod.rs:266-281)
    auto esc = [](const std::string &s) {
        std::string r;
        for (unsigned char c : s) {
            if (c == '\\') r += "\\\\";
            else if (c == '"') r += "\\\"";
            else if (c == '\n') r += "\\n";
            else if (c == '\r') r += "\\r";
            else if (c == '\t') r += "\\t";
            else if (c < 0x20 || c == 0x7f) {
                char b[8];
                snprintf(b, sizeof b, "\\u%04x", c);
                r += b;
            } else r += (char)c;
        }
        return r;
    };
    std::string rows;
    for (size_t i = 0; i < results.size(); i++) {
        const AnalysisResult &r = results[i];
        rows += std::string("    {\"analyzer\": \"") + r.analyzer + "\", \"status\": \"" + r.status_str() + "\"" +
                (!r.details.empty() ? ", \"details\": \"" + esc(r.details) + "\"" : "") + "}" + (i + 1 < results.size() ? ",\n" : "");
    }
    return "{\n  \"private_key\": \"" + hex + "\",\n  \"bit_length\": " + std::to_string(calculate_bit_length(key)) + ",\n  \"hamming_weight\": " +
           std::to_string(hw) + ",\n  \"leading_zeros\": " + std::to_string(lz) + ",\n  \"results\": [\n" + rows + "\n  ]\n}";
}

} // namespace vuke
