// transform.cpp -- TransformType::from_str grammar, Transform::name(), labels and the device
// apply_batch.  Mirrors src/transform/mod.rs:170-215, src/lcg.rs:247-277, src/xorshift.rs:334-357,
// src/sha256_chain.rs:187-217 and the label formats of src/transform/{lcg,xorshift,sha256_chain}.rs.
#include "vuke_host.hpp"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace vuke {

static std::string lower(std::string s)
{
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    return s;
}
static std::vector<std::string> split_colon(std::string s)
{
    while (!s.empty() && s.back() == ':') s.pop_back(); // trim_end_matches(':')
    std::vector<std::string> parts;
    size_t pos = 0;
    for (;;) {
        size_t q = s.find(':', pos);
        parts.push_back(s.substr(pos, q == std::string::npos ? std::string::npos : q - pos));
        if (q == std::string::npos) break;
        pos = q + 1;
    }
    return parts;
}

static const char *LCG_NAMES[4] = {"glibc", "minstd", "msvc", "borland"};
static const char *XS_NAMES[4] = {"xorshift64", "xorshift128", "xorshift128+", "xoroshiro128**"};
static const char *CH_NAMES[4] = {"iterated", "indexed:be", "indexed:le", "counter"};

static int lcg_variant(const std::string &v)
{
    for (int i = 0; i < 4; i++)
        if (v == LCG_NAMES[i]) return i;
    return -1;
}
static int lcg_endian(const std::string &v)
{
    if (v == "be" || v == "big") return 0;
    if (v == "le" || v == "little") return 1;
    return -1;
}
static int xs_variant(const std::string &v)
{ // XorshiftVariant::from_str, src/xorshift.rs:46-56
    if (v == "64" || v == "xorshift64") return 0;
    if (v == "128" || v == "xorshift128") return 1;
    if (v == "128plus" || v == "128+" || v == "plus" || v == "xorshift128+") return 2;
    if (v == "xoroshiro" || v == "starstar" || v == "xoroshiro128**" || v == "xoroshiro128starstar") return 3;
    return -1;
}

Transform TransformType::from_str(const std::string &raw)
{
    const std::string s = lower(raw);
    vk_transform_spec sp{0, VK_VARIANT_ALL, 0, 10};
    if (s == "sha256") { sp.kind = VK_T_SHA256; return Transform(sp); }
    if (s == "milksad") { sp.kind = VK_T_MILKSAD; return Transform(sp); }
    if (s == "mt64") { sp.kind = VK_T_MT64; return Transform(sp); }
    if (s == "electrum" || s == "electrum:change") { // src/transform/mod.rs:180-181; 20 indices (src/transform/electrum.rs:12)
        sp.kind = VK_T_ELECTRUM;
        sp.variant = s == "electrum" ? 0u : 1u;
        sp.chain_depth = 20;
        return Transform(sp);
    }
    if (s == "multibit") { // src/transform/mod.rs:178; 20 indices, empty passphrase (src/transform/multibit.rs:15-21)
        sp.kind = VK_T_MULTIBIT;
        sp.variant = 0;
        sp.chain_depth = 20;
        return Transform(sp);
    }
    if (s == "double_sha256") { sp.kind = VK_T_DOUBLE_SHA256; return Transform(sp); }
    if (s == "md5") { sp.kind = VK_T_MD5; return Transform(sp); }
    if (s == "direct") { sp.kind = VK_T_DIRECT; return Transform(sp); }
    if (s == "armory") { sp.kind = VK_T_ARMORY; return Transform(sp); }
    if (s == "bitimage") { sp.kind = VK_T_BITIMAGE; return Transform(sp); } // src/transform/mod.rs:204-209: defaults, configured later
    if (s == "lcg" || s.rfind("lcg:", 0) == 0) { // LcgConfig::parse
        auto parts = split_colon(s);
        sp.kind = VK_T_LCG;
        if (parts.size() == 1 && parts[0] == "lcg") return Transform(sp);
        if (parts.size() == 2 && parts[0] == "lcg") {
            int e = lcg_endian(parts[1]), v = lcg_variant(parts[1]);
            if (e >= 0) { sp.endian = (uint32_t)e; return Transform(sp); }
            if (v >= 0) { sp.variant = (uint32_t)v; return Transform(sp); }
            throw std::invalid_argument("Invalid LCG variant or endian: '" + parts[1] +
                                        "'. Valid variants: glibc, minstd, msvc, borland. Valid endian: be, le");
        }
        if (parts.size() == 3 && parts[0] == "lcg") {
            int v = lcg_variant(parts[1]);
            if (v < 0) throw std::invalid_argument("Invalid LCG variant: '" + parts[1] + "'. Valid: glibc, minstd, msvc, borland");
            int e = lcg_endian(parts[2]);
            if (e < 0) throw std::invalid_argument("Invalid endian: '" + parts[2] + "'. Valid: be, le");
            sp.variant = (uint32_t)v;
            sp.endian = (uint32_t)e;
            return Transform(sp);
        }
        throw std::invalid_argument("Invalid LCG format. Use: lcg, lcg:variant, lcg:endian, lcg:variant:endian");
    }
    if (s == "xorshift" || s.rfind("xorshift:", 0) == 0) { // XorshiftConfig::parse
        auto parts = split_colon(s);
        sp.kind = VK_T_XORSHIFT;
        if (parts.size() == 1 && parts[0] == "xorshift") return Transform(sp);
        if (parts.size() == 2 && parts[0] == "xorshift") {
            int v = xs_variant(parts[1]);
            if (v < 0) throw std::invalid_argument("Invalid xorshift variant: '" + parts[1] + "'. Valid: 64, 128, 128plus, xoroshiro");
            sp.variant = (uint32_t)v;
            return Transform(sp);
        }
        throw std::invalid_argument("Invalid xorshift format. Use: xorshift, xorshift:64, xorshift:128, xorshift:128plus, xorshift:xoroshiro");
    }
    if (s == "sha256_chain" || s.rfind("sha256_chain:", 0) == 0) { // Sha256ChainConfig::parse
        std::string t = s;
        while (!t.empty() && t.back() == ':') t.pop_back();
        auto parts = split_colon(s);
        sp.kind = VK_T_SHA256_CHAIN;
        if (parts.size() == 1 && parts[0] == "sha256_chain") return Transform(sp);
        if (parts.size() == 2 && parts[0] == "sha256_chain") {
            // Sha256ChainVariant::from_str sees a single segment here: "indexed:be" cannot occur
            const std::string &v = parts[1];
            if (v == "iterated") sp.variant = 0;
            else if (v == "indexed") sp.variant = 1;
            else if (v == "counter") sp.variant = 3;
            else throw std::invalid_argument("Invalid SHA256 chain variant: '" + v + "'. Valid: iterated, indexed, counter");
            return Transform(sp);
        }
        if (parts.size() == 3 && parts[0] == "sha256_chain" && parts[1] == "indexed") {
            int e = lcg_endian(parts[2]);
            if (e < 0) throw std::invalid_argument("Invalid endian: '" + parts[2] + "'. Valid: be, le");
            sp.variant = e == 0 ? 1u : 2u;
            return Transform(sp);
        }
        throw std::invalid_argument("Invalid SHA256 chain format: '" + t +
                                    "'. Valid: sha256_chain, sha256_chain:iterated, sha256_chain:indexed[:be|:le], sha256_chain:counter");
    }
    throw std::invalid_argument("Unknown transform: " + raw +
                                ". Valid: direct, sha256, double_sha256, md5, milksad, mt64, armory, multibit, electrum[:change], "
                                "lcg[:variant][:endian], xorshift[:variant], sha256_chain[:variant], bitimage");
}

const char *Transform::name() const
{
    switch (spec_.kind) {
    case VK_T_BITIMAGE: return "bitimage";
    case VK_T_SHA256: return "sha256";
    case VK_T_DOUBLE_SHA256: return "double_sha256";
    case VK_T_MD5: return "md5";
    case VK_T_DIRECT: return "direct";
    case VK_T_ARMORY: return "armory";
    case VK_T_MILKSAD: return "milksad";
    case VK_T_MT64: return "mt64";
    case VK_T_LCG: return "lcg";
    case VK_T_XORSHIFT: return "xorshift";
    case VK_T_ELECTRUM: return spec_.variant ? "electrum:change" : "electrum"; // src/transform/electrum.rs:36-42
    case VK_T_MULTIBIT: return "multibit";
    case VK_T_SHA256_CHAIN: // src/transform/sha256_chain.rs:46-58
        switch (spec_.variant) {
        case 0: return "sha256_chain:iterated";
        case 1: return "sha256_chain:indexed:be";
        case 2: return "sha256_chain:indexed:le";
        case 3: return "sha256_chain:counter";
        default: return "sha256_chain";
        }
    default: return "?";
    }
}

std::string Transform::label(uint64_t seed, uint32_t sub) const
{
    const std::string s = std::to_string(seed); // Input::from_u64 string_val
    const bool all = spec_.variant == VK_VARIANT_ALL;
    switch (spec_.kind) {
    case VK_T_LCG: // src/transform/lcg.rs:54-66
        if (!all) return s;
        return spec_.endian ? s + ":" + LCG_NAMES[sub & 3] + ":le" : s + ":" + LCG_NAMES[sub & 3];
    case VK_T_XORSHIFT: // src/transform/xorshift.rs:43-47
        return all ? s + ":" + XS_NAMES[sub & 3] : s;
    case VK_T_SHA256_CHAIN: { // src/transform/sha256_chain.rs:67,88
        const uint32_t nv = all ? 4u : 1u, d = spec_.chain_depth;
        const uint32_t r = sub % (nv * d);
        const uint32_t v = all ? r / d : spec_.variant, idx = r % d;
        return s + "[" + CH_NAMES[v & 3] + ":" + std::to_string(idx) + "]";
    }
    default: return s;
    }
}

std::string Transform::label(const std::string &input, uint32_t sub) const
{
    if (spec_.kind == VK_T_ELECTRUM) { // "{truncate_seed(seed)}[{chain}/{i}]", src/transform/electrum.rs:58-62, src/electrum.rs:199-211
        std::string shown = input;
        if (input.size() > 20) shown = input.substr(0, 8) + "..." + input.substr(input.size() - 8); // hex seeds are ASCII
        return shown + "[" + (spec_.variant ? "1" : "0") + "/" + std::to_string(sub) + "]";
    }
    if (spec_.kind == VK_T_MULTIBIT) { // "{truncate_mnemonic(m)}[m/0'/0/{i}]", src/transform/multibit.rs:52, src/multibit.rs:352-360
        std::vector<std::string> w;
        size_t i = 0;
        auto ws = [](unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c; };
        while (i < input.size()) {
            while (i < input.size() && ws((unsigned char)input[i])) i++;
            size_t s0 = i;
            while (i < input.size() && !ws((unsigned char)input[i])) i++;
            if (i > s0) w.push_back(input.substr(s0, i - s0));
        }
        std::string shown = w.size() <= 4 ? input : w[0] + " " + w[1] + "..." + w[w.size() - 2] + " " + w[w.size() - 1];
        return shown + "[m/0'/0/" + std::to_string(sub) + "]";
    }
    if (spec_.kind == VK_T_SHA256_CHAIN) { // "{seed}[{variant}:{idx}]", src/transform/sha256_chain.rs:67
        const bool all = spec_.variant == VK_VARIANT_ALL;
        const uint32_t d = spec_.chain_depth ? spec_.chain_depth : 1;
        return input + "[" + CH_NAMES[(all ? sub / d : spec_.variant) & 3] + ":" + std::to_string(sub % d) + "]";
    }
    return input;
}

// ---------------------------------------------------------------------------------------------- bitimage
static bool rust_ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

// u32::from_str: optional '+', at least one ASCII digit, no overflow
static bool parse_u32(const std::string &t, uint32_t &out)
{
    size_t d = 0;
    if (d < t.size() && t[d] == '+') d++;
    if (d == t.size()) return false;
    uint64_t v = 0;
    for (; d < t.size(); d++) {
        if (t[d] < '0' || t[d] > '9') return false;
        v = v * 10 + (uint64_t)(t[d] - '0');
        if (v > 0xFFFFFFFFull) return false;
    }
    out = (uint32_t)v;
    return true;
}

std::vector<uint32_t> parse_derivation_path(const std::string &raw)
{
    size_t b = 0, e = raw.size();
    while (b < e && rust_ws((unsigned char)raw[b])) b++;
    while (e > b && rust_ws((unsigned char)raw[e - 1])) e--;
    const std::string path = raw.substr(b, e - b);
    if (path == "m") return {};
    if (path.rfind("m/", 0) != 0) throw std::invalid_argument("Invalid derivation path: Path must start with 'm/' or be 'm'");
    std::vector<uint32_t> out;
    size_t i = 2;
    while (i <= path.size()) {
        size_t s0 = i;
        while (i < path.size() && path[i] != '/') i++;
        std::string comp = path.substr(s0, i - s0);
        i++;
        if (comp.empty()) continue;
        std::string idx = comp;
        bool hardened = false;
        if (comp.back() == '\'' || comp.back() == 'h') {
            hardened = true;
            idx.pop_back();
        }
        uint32_t v;
        if (!parse_u32(idx, v)) throw std::invalid_argument("Invalid derivation path: Invalid index: " + comp);
        if (v >= 0x80000000u) throw std::invalid_argument("Invalid derivation path: Index " + std::to_string(v) + " exceeds BIP32 limit (must be < 2^31)");
        out.push_back(v | (hardened ? 0x80000000u : 0u));
    }
    return out;
}

std::string increment_path_index(const std::string &path)
{
    const size_t slash = path.rfind('/');
    if (slash != std::string::npos) {
        const std::string prefix = path.substr(0, slash + 1);
        std::string suffix = path.substr(slash + 1), marker;
        if (!suffix.empty() && (suffix.back() == '\'' || suffix.back() == 'h')) {
            marker = suffix.substr(suffix.size() - 1);
            suffix.pop_back();
        }
        uint32_t v;
        if (parse_u32(suffix, v)) return prefix + std::to_string((uint32_t)(v + 1)) + marker;
    }
    return path;
}

std::vector<std::string> BitimageConfig::load_passphrase_wordlist(const std::string &file)
{
    std::ifstream f(file, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open passphrase wordlist: " + file);
    std::vector<std::string> words;
    std::string line;
    while (std::getline(f, line)) {
        size_t b = 0, e = line.size();
        while (b < e && rust_ws((unsigned char)line[b])) b++;
        while (e > b && rust_ws((unsigned char)line[e - 1])) e--;
        if (e > b) words.push_back(line.substr(b, e - b));
    }
    return words;
}

void Transform::apply_batch_blobs(vk_engine *e, const std::vector<std::array<uint8_t, 32>> &entropies, const std::vector<std::string> &labels,
                                  std::vector<std::pair<std::string, Key>> &output) const
{
    if (!is_bitimage() || entropies.empty()) return; // every other transform skips Input::blob
    const BitimageConfig &c = bitimage;
    // the passphrases one file is tried with, in order (src/transform/bitimage.rs:120-126)
    std::vector<std::string> pws = c.passphrase_wordlist ? *c.passphrase_wordlist : std::vector<std::string>{c.passphrase};
    if (pws.empty() || c.derive_count == 0) return;
    // the path texts derive_keys_for_passphrase walks (src/transform/bitimage.rs:80-96); a text that does not parse ends
    // the walk (Err => continue without advancing: every later attempt repeats it)
    std::vector<std::string> texts{c.path};
    std::vector<std::vector<uint32_t>> comps;
    for (uint32_t i = 0; i < c.derive_count; i++) {
        try {
            comps.push_back(parse_derivation_path(texts[i]));
        } catch (const std::invalid_argument &) {
            break;
        }
        if (i + 1 < c.derive_count) texts.push_back(increment_path_index(texts[i]));
    }
    const uint32_t count = (uint32_t)comps.size();
    if (count == 0) return;
    // runs the device walks in one call: same parent, last index going up by one ("m" repeats the master key).  A last
    // component increment_path_index cannot read ("m/0/", "m/0/1 ") leaves the text as it is: runs of one.
    struct Run { uint32_t first, len; };
    std::vector<Run> runs;
    for (uint32_t i = 0; i < count;) {
        uint32_t len = 1;
        while (i + len < count) {
            const auto &a = comps[i], &n = comps[i + len];
            bool next = a.size() == n.size();
            if (next && !a.empty()) {
                next = std::equal(a.begin(), a.end() - 1, n.begin()) && (a.back() >> 31) == (n.back() >> 31) &&
                       (uint64_t)(a.back() & 0x7FFFFFFFu) + len == (n.back() & 0x7FFFFFFFu);
            }
            if (!next) break;
            len++;
        }
        runs.push_back({i, len});
        i += len;
    }
    std::vector<uint8_t> pdata;
    std::vector<uint32_t> poffs{0};
    for (const std::string &pw : pws) {
        pdata.insert(pdata.end(), pw.begin(), pw.end());
        poffs.push_back((uint32_t)pdata.size());
    }
    const size_t per_call = std::max<size_t>(1, (1u << 22) / (pws.size() * count)); // files per device call
    std::vector<uint8_t> keys, valid, rk, rv;
    for (size_t f0 = 0; f0 < entropies.size(); f0 += per_call) {
        const size_t nf = std::min(per_call, entropies.size() - f0), items = nf * pws.size();
        keys.assign(items * count * 32, 0);
        valid.assign(items * count, 0);
        for (const Run &r : runs) {
            rk.assign(items * r.len * 32, 0);
            rv.assign(items * r.len, 0);
            check_rc(e, vk_bitimage_keys(e, entropies[f0].data(), nf, pdata.data(), poffs.data(), pws.size(), comps[r.first].data(),
                                         (uint32_t)comps[r.first].size(), r.len, rk.data(), rv.data()));
            for (size_t it = 0; it < items; it++)
                for (uint32_t q = 0; q < r.len; q++) {
                    valid[it * count + r.first + q] = rv[it * r.len + q];
                    memcpy(&keys[(it * count + r.first + q) * 32], &rk[(it * r.len + q) * 32], 32);
                }
        }
        for (size_t f = 0; f < nf; f++)
            for (size_t j = 0; j < pws.size(); j++)
                for (uint32_t i = 0; i < count; i++) {
                    const size_t idx = (f * pws.size() + j) * count + i;
                    if (!valid[idx]) break; // a failed derivation is repeated for every later index
                    Key key;
                    std::copy(keys.begin() + 32 * idx, keys.begin() + 32 * idx + 32, key.begin());
                    const std::string &lab = labels[f0 + f];
                    output.emplace_back(pws[j].empty() ? lab + "[" + texts[i] + "]" : lab + ":" + pws[j] + "[" + texts[i] + "]", key);
                }
    }
}

void load_bip39_wordlist(vk_engine *e, const std::string &path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open BIP-39 word list: " + path);
    std::vector<uint8_t> blob(2048 * 8, 0);
    std::string w;
    size_t n = 0;
    while (f >> w) {
        if (n >= 2048 || w.size() > 8) throw std::invalid_argument("BIP-39 word list must hold 2048 words of at most 8 bytes");
        memcpy(&blob[8 * n++], w.data(), w.size());
    }
    if (n != 2048) throw std::invalid_argument("BIP-39 word list must hold exactly 2048 words");
    check_rc(e, vk_bip39_wordlist_set(e, blob.data()));
}

void Transform::apply_batch(vk_engine *e, const std::vector<std::string> &inputs, std::vector<std::pair<std::string, Key>> &output) const
{
    const uint32_t k = string_slots();
    if (inputs.empty() || k == 0) return; // k == 0: a transform that needs Input::u64_val pushes nothing
    // strings of any length: no transform has a device limit any more (ABI 5)
    std::vector<uint8_t> data;
    std::vector<uint32_t> offs{0};
    for (const std::string &s : inputs) {
        data.insert(data.end(), s.begin(), s.end());
        offs.push_back((uint32_t)data.size());
    }
    std::vector<uint8_t> keys(inputs.size() * k * 32), valid(inputs.size() * k);
    check_rc(e, vk_transform_strings(e, &spec_, data.data(), offs.data(), inputs.size(), keys.data(), valid.data()));
    for (size_t i = 0; i < inputs.size(); i++)
        for (uint32_t s = 0; s < k; s++) {
            const size_t idx = i * k + s;
            if (!valid[idx]) continue;
            Key key;
            std::copy(keys.begin() + 32 * idx, keys.begin() + 32 * idx + 32, key.begin());
            output.emplace_back(label(inputs[i], s), key);
        }
}

void Transform::apply_batch(vk_engine *e, const std::vector<uint64_t> &inputs, std::vector<std::pair<std::string, Key>> &output) const
{
    if (takes_strings()) { // numeric sources hand string transforms the decimal string (Input::from_u64)
        std::vector<std::string> strs;
        for (uint64_t v : inputs) strs.push_back(std::to_string(v));
        apply_batch(e, strs, output);
        return;
    }
    const uint32_t k = slots();
    if (inputs.empty() || k == 0) return;
    std::vector<uint8_t> keys(inputs.size() * k * 32), valid(inputs.size() * k);
    check_rc(e, vk_transform_batch(e, &spec_, inputs.data(), inputs.size(), keys.data(), valid.data()));
    for (size_t i = 0; i < inputs.size(); i++)
        for (uint32_t s = 0; s < k; s++) {
            const size_t idx = i * k + s;
            if (!valid[idx]) continue;
            Key key;
            std::copy(keys.begin() + 32 * idx, keys.begin() + 32 * idx + 32, key.begin());
            output.emplace_back(label(inputs[i], s), key);
        }
}

void check_rc(vk_engine *e, int rc)
{
    if (rc != VK_OK) throw EngineError(rc, vk_last_error(e));
}

std::string last_warning(vk_engine *e)
{
    char buf[512];
    buf[0] = 0;
    vk_last_warning(e, buf, sizeof buf);
    return buf;
}

} // namespace vuke
