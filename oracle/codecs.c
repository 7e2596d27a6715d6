/*
 * codecs.c -- ORACLE (test infrastructure, see vko.h): Base58Check and BIP-173 segwit
 * addresses.  The reference gets these from the un-vendored crates base58ck 0.1.0 and
 * bech32 0.11.1 through Address::p2pkh / Address::p2wpkh / PrivateKey::to_wif
 * (src/derive.rs:112-118,159-160).  Restated from the Bitcoin wiki Base58Check text
 * and BIP-173; pinned by src/derive.rs:190-197 and BIP-173's key-1 address.
 */
#include "vko.h"
#include <ctype.h>
#include <string.h>

static const char B58[] = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz";

int vko_base58check_encode(const uint8_t *data, size_t len, char *out, size_t cap)
{
    uint8_t buf[128], d1[32], d2[32];
    if (len + 4 > sizeof buf) return -1;
    memcpy(buf, data, len);
    vko_sha256(data, len, d1);
    vko_sha256(d1, 32, d2);
    memcpy(buf + len, d2, 4);
    size_t n = len + 4, zeros = 0;
    while (zeros < n && buf[zeros] == 0) zeros++;
    uint8_t digits[192]; /* base58 digits, little-endian */
    size_t nd = 0;
    for (size_t i = zeros; i < n; i++) {
        uint32_t carry = buf[i];
        for (size_t j = 0; j < nd; j++) {
            carry += (uint32_t)digits[j] << 8;
            digits[j] = (uint8_t)(carry % 58);
            carry /= 58;
        }
        while (carry) { digits[nd++] = (uint8_t)(carry % 58); carry /= 58; }
    }
    if (zeros + nd + 1 > cap) return -1;
    size_t o = 0;
    for (size_t i = 0; i < zeros; i++) out[o++] = '1';
    for (size_t i = 0; i < nd; i++) out[o++] = B58[digits[nd - 1 - i]];
    out[o] = 0;
    return (int)o;
}

int vko_base58check_decode(const char *s, uint8_t *out, size_t cap)
{
    uint8_t bytes[128]; /* little-endian base-256 */
    size_t nb = 0, zeros = 0, sl = strlen(s);
    if (sl == 0 || sl > 110) return -1;
    while (zeros < sl && s[zeros] == '1') zeros++;
    for (size_t i = zeros; i < sl; i++) {
        const char *p = strchr(B58, s[i]);
        if (!p || !s[i]) return -1;
        uint32_t carry = (uint32_t)(p - B58);
        for (size_t j = 0; j < nb; j++) {
            carry += (uint32_t)bytes[j] * 58;
            bytes[j] = (uint8_t)carry;
            carry >>= 8;
        }
        while (carry) { if (nb >= sizeof bytes) return -1; bytes[nb++] = (uint8_t)carry; carry >>= 8; }
    }
    size_t n = zeros + nb;
    if (n < 5 || n > 120) return -1;
    uint8_t full[128];
    memset(full, 0, zeros);
    for (size_t i = 0; i < nb; i++) full[zeros + i] = bytes[nb - 1 - i];
    uint8_t d1[32], d2[32];
    vko_sha256(full, n - 4, d1);
    vko_sha256(d1, 32, d2);
    if (memcmp(d2, full + n - 4, 4) != 0) return -1;
    if (n - 4 > cap) return -1;
    memcpy(out, full, n - 4);
    return (int)(n - 4);
}

/* ------------------------------------------------------------------ bech32 */
static const char BC[] = "qpzry9x8gf2tvdw0s3jn54khce6mua7l";

static uint32_t polymod_step(uint32_t pre)
{
    uint8_t b = (uint8_t)(pre >> 25);
    return ((pre & 0x1FFFFFF) << 5) ^ (-((b >> 0) & 1) & 0x3b6a57b2UL) ^ (-((b >> 1) & 1) & 0x26508e6dUL) ^
           (-((b >> 2) & 1) & 0x1ea119faUL) ^ (-((b >> 3) & 1) & 0x3d4233ddUL) ^ (-((b >> 4) & 1) & 0x2a1462b3UL);
}

static int convert_bits(uint8_t *out, size_t *outlen, int outbits, const uint8_t *in, size_t inlen, int inbits, int pad)
{
    uint32_t val = 0;
    int bits = 0;
    uint32_t maxv = (1u << outbits) - 1;
    *outlen = 0;
    while (inlen--) {
        val = (val << inbits) | *in++;
        bits += inbits;
        while (bits >= outbits) { bits -= outbits; out[(*outlen)++] = (uint8_t)((val >> bits) & maxv); }
    }
    if (pad) {
        if (bits) out[(*outlen)++] = (uint8_t)((val << (outbits - bits)) & maxv);
    } else if (((val << (outbits - bits)) & maxv) || bits >= inbits) {
        return 0;
    }
    return 1;
}

int vko_segwit_encode(const char *hrp, int witver, const uint8_t *prog, size_t proglen, char *out, size_t cap)
{
    uint8_t data[72];
    size_t dl = 0;
    if (witver < 0 || witver > 16 || proglen < 2 || proglen > 40) return -1;
    data[0] = (uint8_t)witver;
    convert_bits(data + 1, &dl, 5, prog, proglen, 8, 1);
    dl += 1;
    size_t hl = strlen(hrp);
    if (hl + 1 + dl + 6 + 1 > cap) return -1;
    uint32_t chk = 1;
    for (size_t i = 0; i < hl; i++) chk = polymod_step(chk) ^ ((uint32_t)hrp[i] >> 5);
    chk = polymod_step(chk);
    size_t o = 0;
    for (size_t i = 0; i < hl; i++) { chk = polymod_step(chk) ^ ((uint32_t)hrp[i] & 0x1f); out[o++] = hrp[i]; }
    out[o++] = '1';
    for (size_t i = 0; i < dl; i++) { chk = polymod_step(chk) ^ data[i]; out[o++] = BC[data[i]]; }
    for (int i = 0; i < 6; i++) chk = polymod_step(chk);
    chk ^= 1; /* bech32 (not bech32m) constant: witness v0 */
    for (int i = 0; i < 6; i++) out[o++] = BC[(chk >> ((5 - i) * 5)) & 0x1f];
    out[o] = 0;
    return (int)o;
}

int vko_segwit_decode(const char *hrp, const char *addr, int *witver, uint8_t *prog, size_t *proglen)
{
    size_t al = strlen(addr), hl = strlen(hrp);
    if (al < hl + 1 + 7 || al > 90) return 0;
    int has_lower = 0, has_upper = 0;
    for (size_t i = 0; i < al; i++) {
        if (addr[i] >= 'a' && addr[i] <= 'z') has_lower = 1;
        if (addr[i] >= 'A' && addr[i] <= 'Z') has_upper = 1;
        if (addr[i] < 33 || addr[i] > 126) return 0;
    }
    if (has_lower && has_upper) return 0;
    for (size_t i = 0; i < hl; i++) if (tolower((unsigned char)addr[i]) != hrp[i]) return 0;
    if (addr[hl] != '1') return 0;
    uint32_t chk = 1;
    for (size_t i = 0; i < hl; i++) chk = polymod_step(chk) ^ ((uint32_t)hrp[i] >> 5);
    chk = polymod_step(chk);
    for (size_t i = 0; i < hl; i++) chk = polymod_step(chk) ^ ((uint32_t)hrp[i] & 0x1f);
    uint8_t data[90];
    size_t dl = 0;
    for (size_t i = hl + 1; i < al; i++) {
        const char *p = strchr(BC, tolower((unsigned char)addr[i]));
        if (!p) return 0;
        uint8_t v = (uint8_t)(p - BC);
        chk = polymod_step(chk) ^ v;
        data[dl++] = v;
    }
    if (chk != 1 || dl < 7) return 0;
    dl -= 6;
    *witver = data[0];
    if (!convert_bits(prog, proglen, 8, data + 1, dl - 1, 5, 0)) return 0;
    if (*proglen < 2 || *proglen > 40) return 0;
    if (*witver == 0 && *proglen != 20 && *proglen != 32) return 0;
    return 1;
}
