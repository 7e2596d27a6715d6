/*
 * kdf_chains.c -- ORACLE (test infrastructure, see vko.h): Electrum v1 deterministic wallet keys
 * (reference src/electrum.rs:57-195, src/transform/electrum.rs:43-67).
 */
#include "vko.h"
#include <stdio.h>
#include <stdlib.h>
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
    /* a seed of any length (the reference hashes the whole slice; round 3: the fixed buffer of rounds 1-2 cut seeds at 128 bytes,
     * unreachable while the device took at most 64) */
    uint8_t small[32 + 2 * 128];
    uint8_t *buf = 32 + 2 * len <= sizeof small ? small : (uint8_t *)malloc(32 + 2 * len);
    memcpy(buf, seed, len);
    memcpy(buf + len, seed, len);
    vko_sha256(buf, 2 * len, out);
    for (uint32_t i = 1; i < rounds; i++) {
        memcpy(buf, out, 32);
        memcpy(buf + 32, seed, len);
        vko_sha256(buf, 32 + len, out);
    }
    if (buf != small) free(buf);
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

static int vko_multibit_keys_salt(const char *const *wl, const uint8_t *mnemonic, size_t len, const uint8_t *salt, size_t saltlen, uint32_t count,
                                  uint8_t *keys, uint8_t *valid, uint8_t seed1_out[64], uint8_t seed2_out[64]);

/* The same with a BIP-39 passphrase: salt = "mnemonic" || passphrase for both stretches (src/multibit.rs:87-101,165-168);
 * the transform always passes "", the multibit-hd analyzer takes --passphrase (src/analyze/multibit.rs:47-52). */
int vko_multibit_keys_pass(const char *const *wl, const uint8_t *mnemonic, size_t len, const uint8_t *pass, size_t passlen, uint32_t count,
                           uint8_t *keys, uint8_t *valid, uint8_t seed1_out[64], uint8_t seed2_out[64])
{
    /* a passphrase of any length, as the reference takes it (round 5; 256 bytes at most before) */
    uint8_t salt_small[8 + 256], *salt = passlen <= 256 ? salt_small : (uint8_t *)malloc(8 + passlen);
    if (!salt) return 0;
    memcpy(salt, "mnemonic", 8);
    if (passlen) memcpy(salt + 8, pass, passlen);
    memset(valid, 0, count);
    int rc = vko_multibit_keys_salt(wl, mnemonic, len, salt, 8 + passlen, count, keys, valid, seed1_out, seed2_out);
    if (salt != salt_small) free(salt);
    return rc;
}

static int vko_multibit_keys_salt(const char *const *wl, const uint8_t *mnemonic, size_t len, const uint8_t *salt, size_t saltlen, uint32_t count,
                                  uint8_t *keys, uint8_t *valid, uint8_t seed1_out[64], uint8_t seed2_out[64])
{
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
    vko_pbkdf2_hmac_sha512(mnemonic, len, salt, saltlen, 2048, seed1, 64); /* password = the string as given */
    char buggy[512];
    size_t bl = vko_multibit_buggy_mnemonic(wl, seed1, buggy, sizeof buggy);
    if (!bl) return 0;
    vko_pbkdf2_hmac_sha512((const uint8_t *)buggy, bl, salt, saltlen, 2048, seed2, 64);
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

/* ==============================================================================================
 * Bitimage: file -> base64 -> SHA-256 -> BIP-39 mnemonic -> BIP-32 path (reference src/bitimage.rs:38-293,
 * src/transform/bitimage.rs:71-128).  The word list is an input, as above.
 * ============================================================================================ */
/* base64 STANDARD engine: RFC 4648 alphabet with '=' padding.  out needs 4 * ceil(n / 3) bytes. */
size_t vko_base64(const uint8_t *data, size_t n, char *out)
{
    static const char A[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    size_t o = 0, i = 0;
    for (; i + 3 <= n; i += 3) {
        uint32_t v = ((uint32_t)data[i] << 16) | ((uint32_t)data[i + 1] << 8) | data[i + 2];
        out[o++] = A[v >> 18]; out[o++] = A[(v >> 12) & 63]; out[o++] = A[(v >> 6) & 63]; out[o++] = A[v & 63];
    }
    if (n - i == 1) {
        uint32_t v = (uint32_t)data[i] << 16;
        out[o++] = A[v >> 18]; out[o++] = A[(v >> 12) & 63]; out[o++] = '='; out[o++] = '=';
    } else if (n - i == 2) {
        uint32_t v = ((uint32_t)data[i] << 16) | ((uint32_t)data[i + 1] << 8);
        out[o++] = A[v >> 18]; out[o++] = A[(v >> 12) & 63]; out[o++] = A[(v >> 6) & 63]; out[o++] = '=';
    }
    return o;
}

/* from_file_bytes, first half (:38-44): entropy = SHA-256 of the base64 text of the file */
int vko_bitimage_entropy(const uint8_t *data, size_t n, uint8_t out[32])
{
    char *b = (char *)malloc(4 * ((n + 2) / 3) + 1);
    if (!b) return 0;
    size_t bl = vko_base64(data, n, b);
    vko_sha256((const uint8_t *)b, bl, out);
    free(b);
    return 1;
}

/* entropy_to_mnemonic (:119-148): elen * 8 bits + elen / 4 checksum bits, 11 per word, words joined by one space */
size_t vko_bip39_mnemonic(const char *const *wl, const uint8_t *entropy, size_t elen, char *out, size_t cap)
{
    uint8_t bits[64 + 4], h[32];
    if (elen > 64 || elen % 4) return 0;
    vko_sha256(entropy, elen, h);
    memcpy(bits, entropy, elen);
    memcpy(bits + elen, h, 4);
    size_t nbits = elen * 8 + elen * 8 / 32, o = 0;
    for (size_t i = 0; i < nbits / 11; i++) {
        uint32_t idx = 0;
        for (int j = 0; j < 11; j++) {
            size_t b = i * 11 + j;
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

/* parse_derivation_path (:77-117).  comps[i] = index | 0x80000000 when hardened.  Returns the depth, -1 for
 * "Path must start with 'm/' or be 'm'", -2 for "Invalid index", -3 for "exceeds BIP32 limit", -4 if cap is too small. */
int vko_bip32_parse_path(const char *path, uint32_t *comps, int cap)
{
    /* str::trim: Unicode White_Space; the ASCII members are enough for a command-line argument */
    size_t b = 0, e = strlen(path);
    while (b < e && (path[b] == ' ' || (path[b] >= 9 && path[b] <= 13))) b++;
    while (e > b && (path[e - 1] == ' ' || (path[e - 1] >= 9 && path[e - 1] <= 13))) e--;
    size_t n = e - b;
    const char *p = path + b;
    if (n == 1 && p[0] == 'm') return 0;
    if (n < 2 || p[0] != 'm' || p[1] != '/') return -1;
    int depth = 0;
    size_t i = 2;
    while (i <= n) { /* split('/') over p[2..n): a trailing '/' yields one empty component */
        size_t s = i;
        while (i < n && p[i] != '/') i++;
        size_t len = i - s;
        i++;
        if (len == 0) continue;
        int hardened = p[s + len - 1] == '\'' || p[s + len - 1] == 'h';
        if (hardened) len--;
        /* u32::from_str: optional '+', then at least one ASCII digit, no overflow */
        size_t d = s;
        if (d < s + len && p[d] == '+') d++;
        if (d == s + len) return -2;
        uint64_t v = 0;
        for (; d < s + len; d++) {
            if (p[d] < '0' || p[d] > '9') return -2;
            v = v * 10 + (uint64_t)(p[d] - '0');
            if (v > 0xFFFFFFFFull) return -2;
        }
        if (v >= 0x80000000ull) return -3;
        if (depth >= cap) return -4;
        comps[depth++] = (uint32_t)v | (hardened ? 0x80000000u : 0u);
    }
    return depth;
}

/* increment_path_index (:277-293): text in, text out (the label carries it) */
size_t vko_bip32_increment_path(const char *path, char *out, size_t cap)
{
    size_t n = strlen(path);
    const char *slash = strrchr(path, '/');
    if (slash) {
        size_t pre = (size_t)(slash - path) + 1, sl = n - pre;
        const char *suf = path + pre;
        const char *marker = "";
        char mk[2] = {0, 0};
        if (sl && (suf[sl - 1] == '\'' || suf[sl - 1] == 'h')) {
            mk[0] = suf[sl - 1];
            marker = mk;
            sl--;
        }
        size_t d = 0;
        if (d < sl && suf[d] == '+') d++;
        int ok = d < sl;
        uint64_t v = 0;
        for (; ok && d < sl; d++) {
            if (suf[d] < '0' || suf[d] > '9') ok = 0;
            else {
                v = v * 10 + (uint64_t)(suf[d] - '0');
                if (v > 0xFFFFFFFFull) ok = 0;
            }
        }
        if (ok) return (size_t)snprintf(out, cap, "%.*s%u%s", (int)pre, path, (uint32_t)(v + 1), marker); /* wraps like release-mode u32 */
    }
    return (size_t)snprintf(out, cap, "%s", path);
}

/* seed_to_master_key + derive_path (:50-74, 180-275): 1 and key = the private key, 0 on any error of the walk */
int vko_bip32_derive(const uint8_t seed[64], const uint32_t *comps, int depth, uint8_t key[32])
{
    uint8_t I[64], k[32], c[32], ck[32], cc[32];
    vko_hmac_sha512((const uint8_t *)"Bitcoin seed", 12, seed, 64, I);
    memcpy(k, I, 32);
    memcpy(c, I + 32, 32);
    for (int i = 0; i < depth; i++) {
        if (!ckd(k, c, comps[i] & 0x7FFFFFFFu, (comps[i] >> 31) != 0, ck, cc)) return 0;
        memcpy(k, ck, 32);
        memcpy(c, cc, 32);
    }
    memcpy(key, k, 32);
    return 1;
}

/* BitimageTransform::derive_keys_for_passphrase for one (entropy, passphrase) (src/transform/bitimage.rs:71-97): keys =
 * count * 32, valid = count.  path = the text form; labels are not built here (vko_bip32_increment_path gives the text).
 * Returns the number of keys emitted. */
uint32_t vko_bitimage_keys(const char *const *wl, const uint8_t entropy[32], const uint8_t *pass, size_t passlen, const char *path,
                           uint32_t count, uint8_t *keys, uint8_t *valid)
{
    char mn[512], cur[512], nxt[512];
    uint8_t seed[64];
    uint32_t emitted = 0;
    memset(valid, 0, count);
    if (strlen(path) >= sizeof cur - 16) return 0;
    size_t ml = vko_bip39_mnemonic(wl, entropy, 32, mn, sizeof mn);
    if (!ml) return 0;
    uint8_t *salt = (uint8_t *)malloc(8 + passlen); /* a passphrase of any length (1024 bytes at most until round 5) */
    if (!salt) return 0;
    memcpy(salt, "mnemonic", 8);
    if (passlen) memcpy(salt + 8, pass, passlen);
    vko_pbkdf2_hmac_sha512((const uint8_t *)mn, ml, salt, 8 + passlen, 2048, seed, 64);
    free(salt);
    snprintf(cur, sizeof cur, "%s", path);
    for (uint32_t i = 0; i < count; i++) {
        uint32_t comps[256]; /* more components than a 512-character path text can hold */
        int depth = vko_bip32_parse_path(cur, comps, 256);
        if (depth < 0 || !vko_bip32_derive(seed, comps, depth, keys + 32 * i)) continue; /* Err(_) => continue: the path is NOT advanced */
        valid[i] = 1;
        emitted++;
        if (i + 1 < count) {
            vko_bip32_increment_path(cur, nxt, sizeof nxt);
            memcpy(cur, nxt, sizeof cur);
        }
    }
    return emitted;
}
