/*
 * transforms.c -- ORACLE (test infrastructure, see vko.h): seed -> 32-byte private key.
 *
 * Each function names the reference lines it restates.  MT19937 / MT19937-64 live in the
 * un-vendored crate rand_mt 4.2.2 (call sites src/transform/milksad.rs:22-24,
 * src/mt64.rs:16-18); they are restated from Matsumoto & Nishimura's init_genrand /
 * init_genrand64 + genrand, with the crate's fill_bytes byte order (little-endian words,
 * confirmed by the reference's own GPU host code src/gpu/mt19937.rs:216-226).
 */
#include "vko.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- MT19937 */
/* src/transform/milksad.rs:17-29: only inputs <= u32::MAX; key = first 32 bytes of
 * Mt::new(seed).fill_bytes = 8 tempered outputs, each little-endian.  Full 624-word
 * state on purpose (the device uses a rolling window; this is the independent form). */
int vko_milksad_key(uint64_t seed, uint8_t key[32])
{
    if (seed > 0xFFFFFFFFull) return 0;
    static const int N = 624, M = 397;
    uint32_t mt[624];
    mt[0] = (uint32_t)seed;
    for (int i = 1; i < N; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    /* one full twist */
    for (int k = 0; k < N; k++) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % N] & 0x7fffffffu);
        mt[k] = mt[(k + M) % N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (int i = 0; i < 8; i++) {
        uint32_t y = mt[i];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        key[4 * i] = (uint8_t)y; key[4 * i + 1] = (uint8_t)(y >> 8);
        key[4 * i + 2] = (uint8_t)(y >> 16); key[4 * i + 3] = (uint8_t)(y >> 24);
    }
    return 1;
}

/* ------------------------------------------------------------- MT19937-64 */
/* src/mt64.rs:15-20: Mt64::new(seed).fill_bytes(32) = 4 outputs, each little-endian. */
int vko_mt64_key(uint64_t seed, uint8_t key[32])
{
    static const int NN = 312, MM = 156;
    uint64_t mt[312];
    mt[0] = seed;
    for (int i = 1; i < NN; i++) mt[i] = 6364136223846793005ULL * (mt[i - 1] ^ (mt[i - 1] >> 62)) + (uint64_t)i;
    for (int k = 0; k < NN; k++) {
        uint64_t x = (mt[k] & 0xFFFFFFFF80000000ULL) | (mt[(k + 1) % NN] & 0x7FFFFFFFULL);
        mt[k] = mt[(k + MM) % NN] ^ (x >> 1) ^ ((x & 1ULL) ? 0xB5026F5AA96619E9ULL : 0ULL);
    }
    for (int i = 0; i < 4; i++) {
        uint64_t x = mt[i];
        x ^= (x >> 29) & 0x5555555555555555ULL;
        x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
        x ^= (x << 37) & 0xFFF7EEE000000000ULL;
        x ^= x >> 43;
        for (int j = 0; j < 8; j++) key[8 * i + j] = (uint8_t)(x >> (8 * j));
    }
    return 1;
}

/* -------------------------------------------------------------------- LCG */
/* src/lcg.rs:68-103 constants, :107-109 lcg_next, :114-128 generate_key;
 * seed filter `val <= variant.max_seed()` src/transform/lcg.rs:52. */
static const struct { const char *name; uint64_t a, c, m; } LCG[4] = {
    {"glibc", 1103515245ull, 12345ull, 1ull << 31},
    {"minstd", 16807ull, 0ull, (1ull << 31) - 1},
    {"msvc", 214013ull, 2531011ull, 1ull << 32},
    {"borland", 22695477ull, 1ull, 1ull << 32},
};

int vko_lcg_key(uint64_t seed, uint32_t variant, uint32_t little_endian, uint8_t key[32])
{
    if (variant >= 4) return 0;
    if (seed > LCG[variant].m - 1) return 0;
    uint64_t state = (uint32_t)seed;
    for (int i = 0; i < 8; i++) {
        state = (LCG[variant].a * state + LCG[variant].c) % LCG[variant].m; /* wrapping u64, then mod */
        uint32_t w = (uint32_t)state;
        if (little_endian) {
            key[4 * i] = (uint8_t)w; key[4 * i + 1] = (uint8_t)(w >> 8);
            key[4 * i + 2] = (uint8_t)(w >> 16); key[4 * i + 3] = (uint8_t)(w >> 24);
        } else {
            key[4 * i] = (uint8_t)(w >> 24); key[4 * i + 1] = (uint8_t)(w >> 16);
            key[4 * i + 2] = (uint8_t)(w >> 8); key[4 * i + 3] = (uint8_t)w;
        }
    }
    return 1;
}

/* --------------------------------------------------------------- xorshift */
static inline uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

/* src/xorshift.rs:286-291 */
static uint64_t splitmix64(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}

static void put_be64(uint8_t *p, uint64_t v) { for (int j = 0; j < 8; j++) p[j] = (uint8_t)(v >> (56 - 8 * j)); }
static void put_be32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

/* variants in ALL_VARIANTS order (src/xorshift.rs:69-74):
 * 0 xorshift64 (:83-119), 1 xorshift128 (:134-172), 2 xorshift128+ (:186-224),
 * 3 xoroshiro128** (:238-276).  Every variant maps seed 0 -> 1 and fills big-endian. */
int vko_xorshift_key(uint64_t seed, uint32_t variant, uint8_t key[32])
{
    if (seed == 0) seed = 1;
    switch (variant) {
    case 0: {
        uint64_t x = seed;
        for (int i = 0; i < 4; i++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            put_be64(key + 8 * i, x);
        }
        return 1;
    }
    case 1: {
        uint32_t x = (uint32_t)seed, y = (uint32_t)(seed >> 32);
        uint32_t z = (uint32_t)seed ^ 0x12345678u, w = (uint32_t)((seed >> 32) ^ 0x87654321ull);
        for (int i = 0; i < 8; i++) {
            uint32_t t = x ^ (x << 11);
            x = y; y = z; z = w;
            w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
            put_be32(key + 4 * i, w);
        }
        return 1;
    }
    case 2:
    case 3: {
        uint64_t s0 = splitmix64(seed), s1 = splitmix64(s0);
        for (int i = 0; i < 4; i++) {
            uint64_t r = (variant == 2) ? s0 + s1 : rotl64(s0 * 5, 7) * 9;
            uint64_t t = s1 ^ s0;
            s0 = rotl64(s0, 24) ^ t ^ (t << 16);
            s1 = rotl64(t, 37);
            put_be64(key + 8 * i, r);
        }
        return 1;
    }
    default: return 0;
    }
}

/* ----------------------------------------------------------- sha256_chain */
/* variants in ALL_VARIANTS order (src/sha256_chain.rs:74-79): 0 iterated, 1 indexed:be,
 * 2 indexed:le, 3 counter.  generate_key_at_index: src/sha256_chain.rs:88-118. */
int vko_sha256_chain_key(const uint8_t *seed, size_t seedlen, uint32_t variant, uint32_t index, uint8_t key[32])
{
    if (variant > 3) return 0;
    if (variant == 0) {
        vko_sha256(seed, seedlen, key);
        for (uint32_t i = 0; i < index; i++) { uint8_t t[32]; vko_sha256(key, 32, t); memcpy(key, t, 32); }
        return 1;
    }
    uint8_t stack[128], *buf = seedlen + 16 <= sizeof stack ? stack : malloc(seedlen + 16);
    if (!buf) return 0;
    size_t n = seedlen;
    memcpy(buf, seed, seedlen);
    if (variant == 1) { put_be32(buf + n, index); n += 4; }
    else if (variant == 2) { for (int j = 0; j < 4; j++) buf[n + j] = (uint8_t)(index >> (8 * j)); n += 4; }
    else n += (size_t)snprintf((char *)buf + n, 16, "%u", index);
    vko_sha256(buf, n, key);
    if (buf != stack) free(buf);
    return 1;
}

/* ------------------------------------------------------------------ armory */
/* ArmoryTransform::generate_key + advance_key (src/transform/armory.rs:21-70) for the 32-byte seeds apply_batch
 * builds (:84-101).  Returns 0 where the reference returns None: seed not a valid secret key, a tweak >= n
 * (Scalar::from_be_bytes) or a zero tweak (mul_tweak). */
int vko_armory_key(const uint8_t seed[32], uint8_t key[32])
{
    static const char MSG[] = "Derive Chaincode from Root Key";
    uint8_t h1[32], h2[32], cc[32], k[32];
    vko_sha256(seed, 32, h1);
    vko_sha256(h1, 32, h2);
    vko_hmac_sha256(h2, 32, (const uint8_t *)MSG, sizeof MSG - 1, cc);
    if (!vko_scalar_valid(seed)) return 0; /* seed.len() == 32: SecretKey::from_slice(seed) */
    memcpy(k, seed, 32);
    for (int step = 0; step < 4; step++) {
        uint8_t p33[33], p65[65], d1[32], d2[32], s[32], nk[32];
        vko_pubkey(k, p33, p65);
        vko_sha256(p65, 65, d1);
        vko_sha256(d1, 32, d2);
        for (int i = 0; i < 32; i++) s[i] = d2[i] ^ cc[i];
        if (!vko_scalar_valid(s)) return 0;
        if (!vko_scalar_mul(k, s, nk)) return 0;
        memcpy(k, nk, 32);
    }
    memcpy(key, k, 32);
    return 1;
}

/* --------------------------------------------------- generic (spec, seed, slot) */
static const char *XS_NAME[4] = {"xorshift64", "xorshift128", "xorshift128+", "xoroshiro128**"};
static const char *CH_NAME[4] = {"iterated", "indexed:be", "indexed:le", "counter"};

uint32_t vko_slots_per_input(const vko_spec *s)
{
    switch (s->kind) {
    case VKO_T_SHA256:
    case VKO_T_DOUBLE_SHA256:
    case VKO_T_DIRECT: return 3;
    case VKO_T_MILKSAD:
    case VKO_T_MD5:
    case VKO_T_MT64: return 1;
    case VKO_T_ARMORY: return 2;
    case VKO_T_LCG:
    case VKO_T_XORSHIFT: return s->variant == VKO_VARIANT_ALL ? 4 : 1;
    case VKO_T_SHA256_CHAIN: return 2u * (s->variant == VKO_VARIANT_ALL ? 4u : 1u) * s->chain_depth;
    default: return 0;
    }
}

/* Input::from_string (src/transform/input.rs:33-41): string_val only, so every `if let Some(be)` arm is skipped
 * and the transforms that need u64_val emit nothing. */
uint32_t vko_slots_per_string(const vko_spec *s)
{
    switch (s->kind) {
    case VKO_T_SHA256:
    case VKO_T_DOUBLE_SHA256:
    case VKO_T_MD5:
    case VKO_T_DIRECT:
    case VKO_T_ARMORY: return 1;
    case VKO_T_SHA256_CHAIN: return (s->variant == VKO_VARIANT_ALL ? 4u : 1u) * s->chain_depth;
    default: return 0;
    }
}

int vko_transform_string(const vko_spec *s, const uint8_t *str, size_t len, uint32_t sub, uint8_t key[32])
{
    uint8_t t[32];
    switch (s->kind) {
    case VKO_T_SHA256: if (sub) return 0; vko_sha256(str, len, key); return 1;                       /* sha256.rs:16-18 */
    case VKO_T_DOUBLE_SHA256: if (sub) return 0; vko_sha256(str, len, t); vko_sha256(t, 32, key); return 1; /* double_sha256.rs:15-17 */
    case VKO_T_MD5: if (sub) return 0; vko_md5(str, len, key); memcpy(key + 16, key, 16); return 1;  /* md5.rs:15-19 */
    case VKO_T_DIRECT:                                                                               /* direct.rs:28-32 */
        if (sub || len > 32) return 0;
        memset(key, 0, 32);
        memcpy(key, str, len);
        return 1;
    case VKO_T_ARMORY: if (sub) return 0; vko_sha256(str, len, t); return vko_armory_key(t, key);        /* armory.rs:86-89 */
    case VKO_T_SHA256_CHAIN: { /* sha256_chain.rs:63-72; generate_chain_from_string = chain over the string's bytes */
        uint32_t nv = s->variant == VKO_VARIANT_ALL ? 4u : 1u, d = s->chain_depth;
        if (d == 0 || sub >= nv * d) return 0;
        return vko_sha256_chain_key(str, len, s->variant == VKO_VARIANT_ALL ? sub / d : s->variant, sub % d, key);
    }
    default: return 0;
    }
}

/* Emit order inside one input == ascending sub_index (the order apply_batch pushes). */
int vko_transform_key(const vko_spec *s, uint64_t seed, uint32_t sub, uint8_t key[32])
{
    char dec[24];
    int dl = snprintf(dec, sizeof dec, "%llu", (unsigned long long)seed); /* Input::from_u64, src/transform/input.rs:22-30 */
    switch (s->kind) {
    case VKO_T_SHA256: { /* src/transform/sha256.rs:14-28: string, BE8, LE8 */
        uint8_t b[8];
        if (sub == 0) { vko_sha256((const uint8_t *)dec, (size_t)dl, key); return 1; }
        if (sub == 1) { for (int j = 0; j < 8; j++) b[j] = (uint8_t)(seed >> (56 - 8 * j)); vko_sha256(b, 8, key); return 1; }
        if (sub == 2) { for (int j = 0; j < 8; j++) b[j] = (uint8_t)(seed >> (8 * j)); vko_sha256(b, 8, key); return 1; }
        return 0;
    }
    case VKO_T_DOUBLE_SHA256: { /* src/transform/double_sha256.rs:13-32: same three forms, hashed twice */
        vko_spec one = *s;
        uint8_t t[32];
        one.kind = VKO_T_SHA256;
        if (!vko_transform_key(&one, seed, sub, t)) return 0;
        vko_sha256(t, 32, key);
        return 1;
    }
    case VKO_T_MD5: /* src/transform/md5.rs:13-21 */
        if (sub) return 0;
        vko_md5((const uint8_t *)dec, (size_t)dl, key);
        memcpy(key + 16, key, 16);
        return 1;
    case VKO_T_DIRECT: /* src/transform/direct.rs:13-34: BE right-aligned, LE left-aligned, string zero padded */
        memset(key, 0, 32);
        if (sub == 0) { for (int j = 0; j < 8; j++) key[24 + j] = (uint8_t)(seed >> (56 - 8 * j)); return 1; }
        if (sub == 1) { for (int j = 0; j < 8; j++) key[j] = (uint8_t)(seed >> (8 * j)); return 1; }
        if (sub == 2) { memcpy(key, dec, (size_t)dl); return 1; } /* a u64 has at most 20 digits */
        return 0;
    case VKO_T_ARMORY: { /* src/transform/armory.rs:84-101: SHA256(string) as the seed, then 24 zero bytes || BE8 */
        uint8_t sd[32];
        if (sub == 0) vko_sha256((const uint8_t *)dec, (size_t)dl, sd);
        else if (sub == 1) { memset(sd, 0, 24); for (int j = 0; j < 8; j++) sd[24 + j] = (uint8_t)(seed >> (56 - 8 * j)); }
        else return 0;
        return vko_armory_key(sd, key);
    }
    case VKO_T_MILKSAD: return sub == 0 ? vko_milksad_key(seed, key) : 0;
    case VKO_T_MT64: return sub == 0 ? vko_mt64_key(seed, key) : 0;
    case VKO_T_LCG: {
        uint32_t v = s->variant == VKO_VARIANT_ALL ? sub : s->variant;
        if (s->variant != VKO_VARIANT_ALL && sub != 0) return 0;
        return vko_lcg_key(seed, v, s->endian, key);
    }
    case VKO_T_XORSHIFT: {
        uint32_t v = s->variant == VKO_VARIANT_ALL ? sub : s->variant;
        if (s->variant != VKO_VARIANT_ALL && sub != 0) return 0;
        return vko_xorshift_key(seed, v, key);
    }
    case VKO_T_SHA256_CHAIN: { /* src/transform/sha256_chain.rs:60-94 */
        uint32_t nv = s->variant == VKO_VARIANT_ALL ? 4u : 1u, d = s->chain_depth;
        if (d == 0 || sub >= 2 * nv * d) return 0;
        uint32_t form = sub / (nv * d), r = sub % (nv * d);
        uint32_t v = s->variant == VKO_VARIANT_ALL ? r / d : s->variant, idx = r % d;
        if (form == 0) return vko_sha256_chain_key((const uint8_t *)dec, (size_t)dl, v, idx, key);
        if (seed > 0xFFFFFFFFull) return 0;
        uint8_t be[4];
        put_be32(be, (uint32_t)seed);
        return vko_sha256_chain_key(be, 4, v, idx, key);
    }
    default: return 0;
    }
}

int vko_transform_label(const vko_spec *s, uint64_t seed, uint32_t sub, char *out, size_t cap)
{
    unsigned long long sd = seed;
    switch (s->kind) {
    case VKO_T_SHA256:
    case VKO_T_DOUBLE_SHA256:
    case VKO_T_MD5:
    case VKO_T_DIRECT:
    case VKO_T_ARMORY:
    case VKO_T_MILKSAD:
    case VKO_T_MT64: return snprintf(out, cap, "%llu", sd);
    case VKO_T_LCG: /* src/transform/lcg.rs:54-66 */
        if (s->variant != VKO_VARIANT_ALL) return snprintf(out, cap, "%llu", sd);
        if (sub >= 4) return -1;
        return s->endian ? snprintf(out, cap, "%llu:%s:le", sd, LCG[sub].name) : snprintf(out, cap, "%llu:%s", sd, LCG[sub].name);
    case VKO_T_XORSHIFT: /* src/transform/xorshift.rs:43-47 */
        if (s->variant != VKO_VARIANT_ALL) return snprintf(out, cap, "%llu", sd);
        if (sub >= 4) return -1;
        return snprintf(out, cap, "%llu:%s", sd, XS_NAME[sub]);
    case VKO_T_SHA256_CHAIN: { /* src/transform/sha256_chain.rs:67,88 */
        uint32_t nv = s->variant == VKO_VARIANT_ALL ? 4u : 1u, d = s->chain_depth;
        if (d == 0 || sub >= 2 * nv * d) return -1;
        uint32_t r = sub % (nv * d);
        uint32_t v = s->variant == VKO_VARIANT_ALL ? r / d : s->variant, idx = r % d;
        return snprintf(out, cap, "%llu[%s:%u]", sd, CH_NAME[v & 3], idx);
    }
    default: return -1;
    }
}

const char *vko_transform_name(const vko_spec *s)
{
    switch (s->kind) {
    case VKO_T_SHA256: return "sha256";
    case VKO_T_DOUBLE_SHA256: return "double_sha256";
    case VKO_T_MD5: return "md5";
    case VKO_T_DIRECT: return "direct";
    case VKO_T_ARMORY: return "armory";
    case VKO_T_MILKSAD: return "milksad";
    case VKO_T_MT64: return "mt64";
    case VKO_T_LCG: return "lcg";
    case VKO_T_XORSHIFT: return "xorshift";
    case VKO_T_SHA256_CHAIN: /* src/transform/sha256_chain.rs:46-58 */
        switch (s->variant) {
        case 0: return "sha256_chain:iterated";
        case 1: return "sha256_chain:indexed:be";
        case 2: return "sha256_chain:indexed:le";
        case 3: return "sha256_chain:counter";
        default: return "sha256_chain";
        }
    default: return "?";
    }
}

/* ---------------------------------------------------------------- analyzer search (oracle)
 * Successive 32-byte draws of one generator instance, as the cascade analyzers consume them
 * (src/analyze/milksad.rs:269-288: rng.fill_bytes(&mut key) per target; src/analyze/mt64.rs:175-192). */
void vko_milksad_draws(uint32_t seed, uint32_t ndraws, uint8_t *out /* ndraws*32 */)
{
    uint32_t mt[624];
    mt[0] = seed;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    int idx = 624;
    for (uint32_t o = 0; o < ndraws * 8; o++) {
        if (idx >= 624) {
            for (int k = 0; k < 624; k++) {
                uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        out[4 * o] = (uint8_t)y; out[4 * o + 1] = (uint8_t)(y >> 8); out[4 * o + 2] = (uint8_t)(y >> 16); out[4 * o + 3] = (uint8_t)(y >> 24);
    }
}

void vko_mt64_draws(uint64_t seed, uint32_t ndraws, uint8_t *out)
{
    uint64_t mt[312];
    mt[0] = seed;
    for (int i = 1; i < 312; i++) mt[i] = 6364136223846793005ULL * (mt[i - 1] ^ (mt[i - 1] >> 62)) + (uint64_t)i;
    int idx = 312;
    for (uint32_t o = 0; o < ndraws * 4; o++) {
        if (idx >= 312) {
            for (int k = 0; k < 312; k++) {
                uint64_t x = (mt[k] & 0xFFFFFFFF80000000ULL) | (mt[(k + 1) % 312] & 0x7FFFFFFFULL);
                mt[k] = mt[(k + 156) % 312] ^ (x >> 1) ^ ((x & 1ULL) ? 0xB5026F5AA96619E9ULL : 0ULL);
            }
            idx = 0;
        }
        uint64_t x = mt[idx++];
        x ^= (x >> 29) & 0x5555555555555555ULL; x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
        x ^= (x << 37) & 0xFFF7EEE000000000ULL; x ^= x >> 43;
        for (int j = 0; j < 8; j++) out[8 * o + j] = (uint8_t)(x >> (8 * j));
    }
}

static uint64_t last8_be(const uint8_t *k) { uint64_t v = 0; for (int i = 24; i < 32; i++) v = (v << 8) | k[i]; return v; }
static int masked_eq(const uint8_t *k, uint32_t bits, uint64_t target)
{ /* (key & (2^bits - 1)) | 2^(bits-1), src/analyze/milksad.rs:147-150,191-192 */
    uint64_t mask = bits >= 64 ? ~0ULL : (1ULL << bits) - 1, high = 1ULL << (bits - 1);
    return ((last8_be(k) & mask) | high) == target;
}

/* nd successive 32-byte fills of one xorshift-family generator instance (src/analyze/xorshift.rs:187-206):
 * fill_bytes continues the stream, so draw j is outputs [4j, 4j+4) (u64 variants) or [8j, 8j+8) (xorshift128). */
void vko_xorshift_draws(uint64_t seed, uint32_t variant, uint32_t nd, uint8_t *out)
{
    if (seed == 0) seed = 1;
    if (variant == 0) {
        uint64_t x = seed;
        for (uint32_t i = 0; i < 4 * nd; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; put_be64(out + 8 * i, x); }
    } else if (variant == 1) {
        uint32_t x = (uint32_t)seed, y = (uint32_t)(seed >> 32);
        uint32_t z = (uint32_t)seed ^ 0x12345678u, w = (uint32_t)(seed >> 32) ^ 0x87654321u;
        for (uint32_t i = 0; i < 8 * nd; i++) {
            uint32_t t = x ^ (x << 11);
            x = y; y = z; z = w;
            w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
            put_be32(out + 4 * i, w);
        }
    } else {
        uint64_t s0 = splitmix64(seed), s1 = splitmix64(s0);
        for (uint32_t i = 0; i < 4 * nd; i++) {
            uint64_t r = (variant == 2) ? s0 + s1 : rotl64(s0 * 5, 7) * 9;
            uint64_t t = s1 ^ s0;
            s0 = rotl64(s0, 24) ^ t ^ (t << 16);
            s1 = rotl64(t, 37);
            put_be64(out + 8 * i, r);
        }
    }
}

/* sha256_chain analyzer for one u32 seed (src/analyze/sha256_chain.rs:48-425): variants in order, chain index
 * ascending, first match wins.  exact / masked: any key of generate_chain(seed_be4, v, depth); cascade: chain[j] vs
 * target j with depth = n.  On a match: *variant, *index (0 for cascade) and the key(s) the reference prints. */
static int chain_seed_match(const vko_spec *spec, uint32_t seed, uint32_t mode, const uint8_t key[32], uint32_t n, const uint8_t *bits,
                            const uint64_t *targets, uint32_t *variant, uint32_t *index, uint8_t *keys_out)
{
    uint8_t be[4];
    put_be32(be, seed);
    uint32_t v0 = spec->variant == VKO_VARIANT_ALL ? 0 : spec->variant, v1 = spec->variant == VKO_VARIANT_ALL ? 3 : spec->variant;
    uint32_t depth = mode == 2 ? n : spec->chain_depth;
    for (uint32_t v = v0; v <= v1; v++) {
        uint8_t chain[8 * 32];
        int all = 1;
        for (uint32_t idx = 0; idx < depth; idx++) {
            uint8_t k[32];
            vko_sha256_chain_key(be, 4, v, idx, k);
            if (mode == 0 ? memcmp(k, key, 32) == 0 : mode == 1 ? masked_eq(k, bits[0], last8_be(key)) : 0) {
                *variant = v; *index = idx;
                if (keys_out) memcpy(keys_out, k, 32);
                return 1;
            }
            if (mode == 2) {
                if (!masked_eq(k, bits[idx], targets[idx])) { all = 0; break; }
                memcpy(chain + 32 * idx, k, 32);
            }
        }
        if (mode == 2 && all) {
            *variant = v; *index = 0;
            if (keys_out) memcpy(keys_out, chain, 32 * n);
            return 1;
        }
    }
    return 0;
}

/* smallest seed in [lo, hi] matching; mode 0 exact (key), 1 masked (key, bits[0]), 2 cascade (n, bits, targets).
 * kind: VKO_T_MILKSAD, VKO_T_MT64, VKO_T_LCG (spec->variant single, spec->endian), VKO_T_XORSHIFT (single variant),
 * VKO_T_SHA256_CHAIN (variant single or all; vi[0] = matching variant, vi[1] = chain index).  Returns 1 and *seed. */
int vko_search(const vko_spec *spec, uint64_t lo, uint64_t hi, uint32_t mode, const uint8_t key[32], uint32_t n,
               const uint8_t *bits, const uint64_t *targets, uint64_t *seed_out, uint8_t *keys_out /* n*32 */, uint32_t vi[2])
{
    uint32_t nd = mode == 2 ? n : 1;
    if (nd > 8) return -1;
    for (uint64_t s = lo;; s++) {
        uint8_t d[8 * 32];
        int have = 1;
        if (spec->kind == VKO_T_SHA256_CHAIN) {
            uint32_t v = 0, ix = 0;
            if (s <= 0xFFFFFFFFULL && chain_seed_match(spec, (uint32_t)s, mode, key, n, bits, targets, &v, &ix, keys_out)) {
                *seed_out = s;
                if (vi) { vi[0] = v; vi[1] = ix; }
                return 1;
            }
            have = 0;
        }
        else if (spec->kind == VKO_T_MILKSAD) { if (s > 0xFFFFFFFFULL) have = 0; else vko_milksad_draws((uint32_t)s, nd, d); }
        else if (spec->kind == VKO_T_MT64) vko_mt64_draws(s, nd, d);
        else if (spec->kind == VKO_T_LCG) have = vko_lcg_key(s, spec->variant, spec->endian, d);
        else if (spec->kind == VKO_T_XORSHIFT) { if (spec->variant > 3) return -1; vko_xorshift_draws(s, spec->variant, nd, d); }
        else return -1;
        if (have) {
            int ok = 1;
            if (mode == 0) ok = memcmp(d, key, 32) == 0;
            else if (mode == 1) ok = masked_eq(d, bits[0], last8_be(key));
            else for (uint32_t j = 0; j < n && ok; j++) ok = masked_eq(d + 32 * j, bits[j], targets[j]);
            if (ok) { *seed_out = s; if (keys_out) memcpy(keys_out, d, 32 * nd); if (vi) { vi[0] = spec->variant; vi[1] = 0; } return 1; }
        }
        if (s == hi) break;
    }
    return 0;
}
