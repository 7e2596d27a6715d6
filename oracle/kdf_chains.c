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
