// codecs.cpp -- text encodings of the reference's DerivedKey, on the host: hex, Base58Check
// (P2PKH, WIF), BIP-173 segwit v0 addresses, decimal.  The reference gets them from the crates
// base58ck / bech32 / hex / num-bigint via src/derive.rs:102-160.  Only checksums are hashed here
// (SHA-256 of <= 38 bytes per string); key derivation itself stays on the GPU.
#include "vuke_host.hpp"

#include <array>
#include <cstring>
#include <stdexcept>

namespace vuke {

bool utf8_valid(const char *p, size_t n)
{ // the well-formed byte sequences of Unicode table 3-7 (what core::str::from_utf8 accepts)
    const unsigned char *s = (const unsigned char *)p, *e = s + n;
    while (s < e) {
        const unsigned char c = *s;
        if (c < 0x80) { s++; continue; }
        int len;
        unsigned char lo = 0x80, hi = 0xBF;
        if (c >= 0xC2 && c <= 0xDF) len = 1;
        else if (c == 0xE0) { len = 2; lo = 0xA0; }
        else if ((c >= 0xE1 && c <= 0xEC) || c == 0xEE || c == 0xEF) len = 2;
        else if (c == 0xED) { len = 2; hi = 0x9F; }
        else if (c == 0xF0) { len = 3; lo = 0x90; }
        else if (c >= 0xF1 && c <= 0xF3) len = 3;
        else if (c == 0xF4) { len = 3; hi = 0x8F; }
        else return false;
        if (e - s <= len) return false;
        if (s[1] < lo || s[1] > hi) return false;
        for (int k = 2; k <= len; k++)
            if (s[k] < 0x80 || s[k] > 0xBF) return false;
        s += len + 1;
    }
    return true;
}

// length of a White_Space code point starting at p (0 if none); input is valid UTF-8
static int ws_at(const unsigned char *p, const unsigned char *e)
{
    const unsigned char c = p[0];
    if (c == ' ' || (c >= 9 && c <= 13)) return 1;
    if (c == 0xC2 && e - p >= 2 && (p[1] == 0x85 || p[1] == 0xA0)) return 2;
    if (e - p >= 3) {
        if (c == 0xE1 && p[1] == 0x9A && p[2] == 0x80) return 3;                                   // U+1680
        if (c == 0xE2 && p[1] == 0x80 && ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 || p[2] == 0xAF)) return 3; // U+2000-200A, 2028, 2029, 202F
        if (c == 0xE2 && p[1] == 0x81 && p[2] == 0x9F) return 3;                                   // U+205F
        if (c == 0xE3 && p[1] == 0x80 && p[2] == 0x80) return 3;                                   // U+3000
    }
    return 0;
}

void trim_unicode(const char *&a, const char *&e)
{
    const unsigned char *s = (const unsigned char *)a, *t = (const unsigned char *)e;
    for (int k; s < t && (k = ws_at(s, t)) != 0;) s += k;
    for (;;) {
        if (t == s) break;
        // step back one code point
        const unsigned char *q = t - 1;
        while (q > s && (*q & 0xC0) == 0x80) q--;
        const int k = ws_at(q, t);
        if (k == 0 || q + k != t) break;
        t = q;
    }
    a = (const char *)s;
    e = (const char *)t;
}


namespace {
inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

struct Sha256 {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    void block(const uint8_t *p)
    {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
            0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
            0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
            0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
            0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
            0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
            0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; i++)
            w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
        uint32_t v[8];
        memcpy(v, h, sizeof v);
        for (int i = 0; i < 64; i++) {
            uint32_t t1 = v[7] + (ror(v[4], 6) ^ ror(v[4], 11) ^ ror(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K[i] + w[i];
            uint32_t t2 = (ror(v[0], 2) ^ ror(v[0], 13) ^ ror(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
            for (int j = 7; j > 0; j--) v[j] = v[j - 1];
            v[4] += t1;
            v[0] = t1 + t2;
        }
        for (int i = 0; i < 8; i++) h[i] += v[i];
    }
};
} // namespace

void sha256(const uint8_t *msg, size_t len, uint8_t out[32])
{
    Sha256 s;
    size_t off = 0;
    for (; off + 64 <= len; off += 64) s.block(msg + off);
    uint8_t tail[128] = {0};
    size_t rem = len - off;
    if (rem) memcpy(tail, msg + off, rem);
    tail[rem] = 0x80;
    size_t tl = rem < 56 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    s.block(tail);
    if (tl == 128) s.block(tail + 64);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(s.h[i] >> 24);
        out[4 * i + 1] = (uint8_t)(s.h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(s.h[i] >> 8);
        out[4 * i + 3] = (uint8_t)s.h[i];
    }
}

void bitimage_entropy(const uint8_t *data, size_t len, uint8_t out[32])
{
    static const char A[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    Sha256 s;
    uint8_t blk[128];
    size_t off = 0;
    uint64_t text_len = 0;
    for (; off + 48 <= len; off += 48) { // 48 bytes -> 64 characters = one SHA-256 block
        for (int g = 0; g < 16; g++) {
            const uint32_t v = ((uint32_t)data[off + 3 * g] << 16) | ((uint32_t)data[off + 3 * g + 1] << 8) | data[off + 3 * g + 2];
            blk[4 * g] = (uint8_t)A[v >> 18];
            blk[4 * g + 1] = (uint8_t)A[(v >> 12) & 63];
            blk[4 * g + 2] = (uint8_t)A[(v >> 6) & 63];
            blk[4 * g + 3] = (uint8_t)A[v & 63];
        }
        s.block(blk);
        text_len += 64;
    }
    size_t tl = 0; // the tail: fewer than 64 characters, then the SHA-256 padding
    memset(blk, 0, sizeof blk);
    for (; off + 3 <= len; off += 3) {
        const uint32_t v = ((uint32_t)data[off] << 16) | ((uint32_t)data[off + 1] << 8) | data[off + 2];
        blk[tl++] = (uint8_t)A[v >> 18];
        blk[tl++] = (uint8_t)A[(v >> 12) & 63];
        blk[tl++] = (uint8_t)A[(v >> 6) & 63];
        blk[tl++] = (uint8_t)A[v & 63];
    }
    if (len - off == 1) {
        const uint32_t v = (uint32_t)data[off] << 16;
        blk[tl++] = (uint8_t)A[v >> 18];
        blk[tl++] = (uint8_t)A[(v >> 12) & 63];
        blk[tl++] = '=';
        blk[tl++] = '=';
    } else if (len - off == 2) {
        const uint32_t v = ((uint32_t)data[off] << 16) | ((uint32_t)data[off + 1] << 8);
        blk[tl++] = (uint8_t)A[v >> 18];
        blk[tl++] = (uint8_t)A[(v >> 12) & 63];
        blk[tl++] = (uint8_t)A[(v >> 6) & 63];
        blk[tl++] = '=';
    }
    text_len += tl;
    blk[tl] = 0x80;
    const size_t padded = tl < 56 ? 64 : 128;
    const uint64_t bits = text_len * 8;
    for (int i = 0; i < 8; i++) blk[padded - 1 - i] = (uint8_t)(bits >> (8 * i));
    s.block(blk);
    if (padded == 128) s.block(blk + 64);
    for (int i = 0; i < 8; i++) {
        out[4 * i] = (uint8_t)(s.h[i] >> 24);
        out[4 * i + 1] = (uint8_t)(s.h[i] >> 16);
        out[4 * i + 2] = (uint8_t)(s.h[i] >> 8);
        out[4 * i + 3] = (uint8_t)s.h[i];
    }
}

std::string hex_encode(const uint8_t *b, size_t n)
{
    static const char H[] = "0123456789abcdef";
    std::string s(2 * n, '0');
    for (size_t i = 0; i < n; i++) {
        s[2 * i] = H[b[i] >> 4];
        s[2 * i + 1] = H[b[i] & 15];
    }
    return s;
}

static const char B58[] = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz";

std::string base58check_encode(const uint8_t *payload, size_t len)
{
    if (len > 60) throw std::invalid_argument("base58check payload too long");
    uint8_t buf[64];
    memcpy(buf, payload, len);
    uint8_t d1[32], d2[32];
    sha256(payload, len, d1);
    sha256(d1, 32, d2);
    memcpy(buf + len, d2, 4);
    const size_t total = len + 4;
    size_t zeros = 0;
    while (zeros < total && buf[zeros] == 0) zeros++;
    // big-endian u32 limbs of the non-zero tail, divided by 58^5 per pass (five characters per division sweep)
    uint32_t limb[16];
    const size_t nb = total - zeros, nl = (nb + 3) / 4;
    for (size_t i = 0; i < nl; i++) limb[i] = 0;
    for (size_t i = 0; i < nb; i++) {
        const size_t pos = nb - 1 - i; // byte significance
        limb[nl - 1 - pos / 4] |= (uint32_t)buf[zeros + i] << (8 * (pos % 4));
    }
    char rev[96];
    size_t n = 0, start = 0;
    while (start < nl) {
        uint64_t rem = 0;
        for (size_t i = start; i < nl; i++) {
            const uint64_t cur = (rem << 32) | limb[i];
            limb[i] = (uint32_t)(cur / 656356768u); // 58^5
            rem = cur % 656356768u;
        }
        while (start < nl && limb[start] == 0) start++;
        uint32_t r = (uint32_t)rem;
        for (int k = 0; k < 5; k++) {
            if (start >= nl && r == 0) break; // most significant group: no leading '1's from the number itself
            rev[n++] = B58[r % 58];
            r /= 58;
        }
    }
    for (size_t i = 0; i < zeros; i++) rev[n++] = '1';
    std::string out(n, '1');
    for (size_t i = 0; i < n; i++) out[i] = rev[n - 1 - i];
    return out;
}

bool base58check_decode(const std::string &s, std::vector<uint8_t> &out)
{
    if (s.empty() || s.size() > 120) return false;
    size_t zeros = 0;
    while (zeros < s.size() && s[zeros] == '1') zeros++;
    std::vector<uint8_t> num; // big-endian base-256
    for (size_t i = zeros; i < s.size(); i++) {
        const char *p = strchr(B58, s[i]);
        if (!p || !s[i]) return false;
        uint32_t carry = (uint32_t)(p - B58);
        for (size_t j = num.size(); j-- > 0;) {
            carry += (uint32_t)num[j] * 58;
            num[j] = (uint8_t)carry;
            carry >>= 8;
        }
        while (carry) {
            num.insert(num.begin(), (uint8_t)carry);
            carry >>= 8;
        }
    }
    std::vector<uint8_t> full(zeros, 0);
    full.insert(full.end(), num.begin(), num.end());
    if (full.size() < 5) return false;
    uint8_t d1[32], d2[32];
    sha256(full.data(), full.size() - 4, d1);
    sha256(d1, 32, d2);
    if (memcmp(d2, full.data() + full.size() - 4, 4) != 0) return false;
    out.assign(full.begin(), full.end() - 4);
    return true;
}

// ------------------------------------------------------------------------------ bech32
static const char BC[] = "qpzry9x8gf2tvdw0s3jn54khce6mua7l";

static uint32_t polymod(const std::vector<uint8_t> &v)
{
    static const uint32_t G[5] = {0x3b6a57b2, 0x26508e6d, 0x1ea119fa, 0x3d4233dd, 0x2a1462b3};
    uint32_t chk = 1;
    for (uint8_t x : v) {
        uint8_t top = (uint8_t)(chk >> 25);
        chk = ((chk & 0x1ffffff) << 5) ^ x;
        for (int i = 0; i < 5; i++)
            if ((top >> i) & 1) chk ^= G[i];
    }
    return chk;
}

static std::vector<uint8_t> hrp_expand(const std::string &hrp)
{
    std::vector<uint8_t> v;
    for (char c : hrp) v.push_back((uint8_t)c >> 5);
    v.push_back(0);
    for (char c : hrp) v.push_back((uint8_t)c & 31);
    return v;
}

std::string segwit_encode(const std::string &hrp, int witver, const uint8_t *prog, size_t len)
{
    std::vector<uint8_t> data{(uint8_t)witver};
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < len; i++) {
        acc = (acc << 8) | prog[i];
        bits += 8;
        while (bits >= 5) {
            bits -= 5;
            data.push_back((acc >> bits) & 31);
        }
    }
    if (bits) data.push_back((acc << (5 - bits)) & 31);
    std::vector<uint8_t> v = hrp_expand(hrp);
    v.insert(v.end(), data.begin(), data.end());
    v.insert(v.end(), 6, 0);
    uint32_t pm = polymod(v) ^ 1; // bech32 constant (witness version 0)
    std::string out = hrp + "1";
    for (uint8_t d : data) out.push_back(BC[d]);
    for (int i = 0; i < 6; i++) out.push_back(BC[(pm >> (5 * (5 - i))) & 31]);
    return out;
}

bool segwit_decode(const std::string &hrp, const std::string &addr, int &witver, std::vector<uint8_t> &prog)
{
    if (addr.size() < hrp.size() + 8 || addr.size() > 90) return false;
    bool lower = false, upper = false;
    std::string a = addr;
    for (char &c : a) {
        if (c >= 'a' && c <= 'z') lower = true;
        if (c >= 'A' && c <= 'Z') {
            upper = true;
            c = (char)(c - 'A' + 'a');
        }
        if (c < 33 || c > 126) return false;
    }
    if (lower && upper) return false;
    if (a.compare(0, hrp.size(), hrp) != 0 || a[hrp.size()] != '1') return false;
    std::vector<uint8_t> data;
    for (size_t i = hrp.size() + 1; i < a.size(); i++) {
        const char *p = strchr(BC, a[i]);
        if (!p || !a[i]) return false;
        data.push_back((uint8_t)(p - BC));
    }
    std::vector<uint8_t> v = hrp_expand(hrp);
    v.insert(v.end(), data.begin(), data.end());
    if (polymod(v) != 1 || data.size() < 7) return false;
    data.resize(data.size() - 6);
    witver = data[0];
    prog.clear();
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 1; i < data.size(); i++) {
        acc = (acc << 5) | data[i];
        bits += 5;
        if (bits >= 8) {
            bits -= 8;
            prog.push_back((acc >> bits) & 0xff);
        }
    }
    if (bits >= 5 || ((acc << (8 - bits)) & 0xff)) return false;
    if (prog.size() < 2 || prog.size() > 40) return false;
    if (witver == 0 && prog.size() != 20 && prog.size() != 32) return false;
    return witver <= 16;
}

std::string decimal_of_be(const uint8_t be[32])
{
    uint32_t w[8];
    for (int i = 0; i < 8; i++) w[i] = (uint32_t)be[4 * i] << 24 | (uint32_t)be[4 * i + 1] << 16 | (uint32_t)be[4 * i + 2] << 8 | be[4 * i + 3];
    std::string out;
    for (;;) {
        uint64_t rem = 0;
        bool nz = false;
        for (int i = 0; i < 8; i++) {
            uint64_t cur = (rem << 32) | w[i];
            w[i] = (uint32_t)(cur / 1000000000u);
            rem = cur % 1000000000u;
            nz |= w[i] != 0;
        }
        char buf[16];
        snprintf(buf, sizeof buf, nz ? "%09u" : "%u", (unsigned)rem);
        out.insert(0, buf);
        if (!nz) break;
    }
    return out;
}

} // namespace vuke
