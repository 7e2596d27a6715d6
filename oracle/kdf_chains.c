/*
 * kdf_chains.c -- ORACLE (test infrastructure, see vko.h): Electrum v1 deterministic wallet keys
 * (reference src/electrum.rs:57-195, src/transform/electrum.rs:43-67).
 */
#include "vko.h"
#include <stdio.h>
#include <string.h>

static int is_hex_string(const uint8_t *s, size_t n)
{ /* char::is_ascii_hexdigit over every char (src/electrum.rs:59) */
    for (size_t i = 0; i < n; i++) {
        uint8_t c = s[i];
        if (!((c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F'))) return 0;
    }
    return 1;
}

/* src/electrum.rs:150-168: x = SHA256(seed||seed); 99 999 x: x = SHA256(x||seed) */
void vko_electrum_stretch(const uint8_t *seed, size_t len, uint32_t rounds, uint8_t out[32])
{
    uint8_t buf[32 + 256];
    if (len > 128) len = 128;
    memcpy(buf, seed, len);
    memcpy(buf + len, seed, len);
    vko_sha256(buf, 2 * len, out);
    for (uint32_t i = 1; i < rounds; i++) {
        memcpy(buf, out, 32);
        memcpy(buf + 32, seed, len);
        vko_sha256(buf, 32 + len, out);
    }
}

/* ElectrumTransform::apply_batch for one input: keys[i] valid iff valid[i]; returns 0 when the
 * whole input is skipped (non-hex seed, or stretched key not a valid scalar). */
int vko_electrum_keys(const uint8_t *seed, size_t len, uint32_t for_change, uint32_t count, uint32_t rounds,
                      uint8_t *keys /* count*32 */, uint8_t *valid /* count */, uint8_t mpk_out[64])
{
    memset(valid, 0, count);
    if (!is_hex_string(seed, len)) return 0;
    uint8_t master[32], p65[65];
    vko_electrum_stretch(seed, len, rounds, master);
    if (!vko_scalar_valid(master)) return 0; /* SecretKey::from_slice, src/electrum.rs:80-82 */
    vko_pubkey(master, NULL, p65);
    if (mpk_out) memcpy(mpk_out, p65 + 1, 64);
    for (uint32_t i = 0; i < count; i++) {
        uint8_t data[96], d1[32], seq[32];
        int pl = snprintf((char *)data, 32, "%u:%u:", i, for_change ? 1u : 0u); /* src/electrum.rs:178-185 */
        memcpy(data + pl, p65 + 1, 64);
        vko_sha256(data, (size_t)pl + 64, d1);
        vko_sha256(d1, 32, seq);
        /* Scalar::from_be_bytes rejects seq >= n (zero is allowed); add_tweak rejects a zero sum */
        static const uint8_t N[32] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFE,
                                      0xBA, 0xAE, 0xDC, 0xE6, 0xAF, 0x48, 0xA0, 0x3B, 0xBF, 0xD2, 0x5E, 0x8C, 0xD0, 0x36, 0x41, 0x41};
        if (memcmp(seq, N, 32) >= 0) continue;
        if (!vko_scalar_add(master, seq, keys + 32 * i)) continue;
        valid[i] = 1;
    }
    return 1;
}

/* truncate_seed (src/electrum.rs:199-211) + label of src/transform/electrum.rs:61 */
int vko_electrum_label(const char *seed, uint32_t for_change, uint32_t index, char *out, size_t cap)
{
    size_t n = strlen(seed);
    if (n <= 20) return snprintf(out, cap, "%s[%u/%u]", seed, for_change ? 1u : 0u, index);
    return snprintf(out, cap, "%.8s...%s[%u/%u]", seed, seed + n - 8, for_change ? 1u : 0u, index);
}

/* ============================================================================================
 * MultiBit HD seed-as-entropy chain (reference src/multibit.rs:87-350, src/transform/multibit.rs:45-61).
 * The BIP-39 word list is an INPUT (2048 NUL-terminated words): nothing under oracle/ embeds it.
 * ============================================================================================ */
static int is_ws(uint8_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c; }

static int word_index(const char *const *wl, const uint8_t *w, size_t n)
{
    for (int i = 0; i < 2048; i++)
        if (strlen(wl[i]) == n && memcmp(wl[i], w, n) == 0) return i;
    return -1;
}

/* entropy_to_mnemonic for 64 bytes -> 48 words joined by single spaces (src/multibit.rs:206-239,104) */
size_t vko_multibit_buggy_mnemonic(const char *const *wl, const uint8_t seed[64], char *out, size_t cap)
{
    uint8_t bits[66], h[32];
    vko_sha256(seed, 64, h);
    memcpy(bits, seed, 64);
    bits[64] = h[0]; /* 512 / 32 = 16 checksum bits */
    bits[65] = h[1];
    size_t o = 0;
    for (int i = 0; i < 48; i++) {
        uint32_t idx = 0;
        for (int j = 0; j < 11; j++) {
            int b = i * 11 + j;
            idx = (idx << 1) | ((bits[b / 8] >> (7 - b % 8)) & 1u);
        }
        size_t wlen = strlen(wl[idx]);
        if (o + wlen + 2 > cap) return 0;
        if (i) out[o++] = ' ';
        memcpy(out + o, wl[idx], wlen);
        o += wlen;
    }
    out[o] = 0;
    return o;
}

/* BIP32 CKDpriv as the reference spells it (:261-350): returns 0 on any of its error paths */
static int ckd(const uint8_t k[32], const uint8_t c[32], uint32_t index, int hardened, uint8_t ck[32], uint8_t cc[32])
{
    uint8_t data[37], I[64];
    if (hardened) {
        index |= 0x80000000u;
        data[0] = 0;
        memcpy(data + 1, k, 32);
    } else {
        if (!vko_scalar_valid(k)) return 0; /* private_to_public: SecretKey::from_slice */
        vko_pubkey(k, data, NULL);
    }
    data[33] = (uint8_t)(index >> 24); data[34] = (uint8_t)(index >> 16); data[35] = (uint8_t)(index >> 8); data[36] = (uint8_t)index;
    vko_hmac_sha512(c, 32, data, 37, I);
    /* scalar_add(il, parent): parent must be a valid key, il < n (zero allowed), sum != 0 */
    static const uint8_t N[32] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFE,
                                  0xBA, 0xAE, 0xDC, 0xE6, 0xAF, 0x48, 0xA0, 0x3B, 0xBF, 0xD2, 0x5E, 0x8C, 0xD0, 0x36, 0x41, 0x41};
    if (!vko_scalar_valid(k)) return 0;
    if (memcmp(I, N, 32) >= 0) return 0;
    if (!vko_scalar_add(k, I, ck)) return 0;
    memcpy(cc, I + 32, 32);
    return 1;
}

/* MultibitTransform::apply_batch for one input.  Returns 0 if the input is skipped. */
int vko_multibit_keys(const char *const *wl, const uint8_t *mnemonic, size_t len, uint32_t count, uint8_t *keys, uint8_t *valid,
                      uint8_t seed1_out[64], uint8_t seed2_out[64])
{
    return vko_multibit_keys_pass(wl, mnemonic, len, (const uint8_t *)"", 0, count, keys, valid, seed1_out, seed2_out);
}

/* The same with a BIP-39 passphrase: salt = "mnemonic" || passphrase for both stretches (src/multibit.rs:87-101,165-168);
 * the transform always passes "", the multibit-hd analyzer takes --passphrase (src/analyze/multibit.rs:47-52). */
int vko_multibit_keys_pass(const char *const *wl, const uint8_t *mnemonic, size_t len, const uint8_t *pass, size_t passlen, uint32_t count,
                           uint8_t *keys, uint8_t *valid, uint8_t seed1_out[64], uint8_t seed2_out[64])
{
    uint8_t salt[8 + 256];
    if (passlen > 256) return 0;
    memcpy(salt, "mnemonic", 8);
    memcpy(salt + 8, pass, passlen);
    memset(valid, 0, count);
    /* split_whitespace + validate_mnemonic (:143-160) */
    size_t nwords = 0, i = 0;
    while (i < len) {
        while (i < len && is_ws(mnemonic[i])) i++;
        size_t s = i;
        while (i < len && !is_ws(mnemonic[i])) i++;
        if (i > s) {
            if (word_index(wl, mnemonic + s, i - s) < 0) return 0;
            nwords++;
        }
    }
    if (!(nwords == 12 || nwords == 15 || nwords == 18 || nwords == 21 || nwords == 24)) return 0;
    /* NB the reference checks the count first; both are errors, so the order does not matter */
    uint8_t seed1[64], seed2[64], I[64];
    vko_pbkdf2_hmac_sha512(mnemonic, len, salt, 8 + passlen, 2048, seed1, 64); /* password = the string as given */
    char buggy[512];
    size_t bl = vko_multibit_buggy_mnemonic(wl, seed1, buggy, sizeof buggy);
    if (!bl) return 0;
    vko_pbkdf2_hmac_sha512((const uint8_t *)buggy, bl, salt, 8 + passlen, 2048, seed2, 64);
    if (seed1_out) memcpy(seed1_out, seed1, 64);
    if (seed2_out) memcpy(seed2_out, seed2, 64);
    vko_hmac_sha512((const uint8_t *)"Bitcoin seed", 12, seed2, 64, I); /* :242-254 */
    uint8_t k0[32], c0[32], k1[32], c1[32];
    if (!ckd(I, I + 32, 0, 1, k0, c0)) return 1; /* m/0' : every index fails */
    if (!ckd(k0, c0, 0, 0, k1, c1)) return 1;    /* m/0'/0 */
    for (uint32_t j = 0; j < count; j++) {
        uint8_t cc[32];
        if (ckd(k1, c1, j, 0, keys + 32 * j, cc)) valid[j] = 1; /* m/0'/0/j */
    }
    return 1;
}

/* truncate_mnemonic + label (src/multibit.rs:352-360, src/transform/multibit.rs:52) */
int vko_multibit_label(const char *mnemonic, uint32_t index, char *out, size_t cap)
{
    const char *w[64];
    size_t wl_[64], n = 0, i = 0, len = strlen(mnemonic);
    while (i < len && n < 64) {
        while (i < len && is_ws((uint8_t)mnemonic[i])) i++;
        size_t s = i;
        while (i < len && !is_ws((uint8_t)mnemonic[i])) i++;
        if (i > s) { w[n] = mnemonic + s; wl_[n] = i - s; n++; }
    }
    if (n <= 4) return snprintf(out, cap, "%s[m/0'/0/%u]", mnemonic, index);
    return snprintf(out, cap, "%.*s %.*s...%.*s %.*s[m/0'/0/%u]", (int)wl_[0], w[0], (int)wl_[1], w[1], (int)wl_[n - 2], w[n - 2],
                    (int)wl_[n - 1], w[n - 1], index);
}
