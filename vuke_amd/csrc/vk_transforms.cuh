// vk_transforms.cuh -- seed -> 32-byte private key, one (seed, slot) per lane, all in VGPRs.
//
// Every function fills kw[8] = the key's bytes as eight big-endian words (kw[0] = bytes 0..3)
// and returns false where the reference emits nothing for that slot.  Reference behaviour
// restated (see also oracle/transforms.c for the CPU form the tests compare against):
//   sha256        src/transform/sha256.rs:13-29
//   milksad       src/transform/milksad.rs:17-29 (rand_mt Mt::new + fill_bytes, LE words)
//   mt64          src/mt64.rs:15-20
//   lcg           src/lcg.rs:68-128, src/transform/lcg.rs:43-72
//   xorshift      src/xorshift.rs:83-317, src/transform/xorshift.rs:33-52
//   sha256_chain  src/sha256_chain.rs:88-145, src/transform/sha256_chain.rs:60-94
//   double_sha256 src/transform/double_sha256.rs:13-32, md5 src/transform/md5.rs:13-21, direct src/transform/direct.rs:13-34
#pragma once
#include "vk_hash.cuh"

#define VK_T_SHA256 0u
#define VK_T_MILKSAD 1u
#define VK_T_MT64 2u
#define VK_T_LCG 3u
#define VK_T_XORSHIFT 4u
#define VK_T_SHA256_CHAIN 5u
#define VK_T_ELECTRUM 6u
#define VK_T_MULTIBIT 7u
#define VK_T_DOUBLE_SHA256 8u
#define VK_T_MD5 9u
#define VK_T_DIRECT 10u
#define VK_T_RAWKEYS 100u /* keys come from memory (vk_derive_batch) */
#define VK_VARIANT_ALL 0xFFFFFFFFu

struct vk_spec_dev {
    u32 kind, variant, endian, chain_depth;
};

// ---- byte-stream builder for single-block SHA-256 messages (<= 55 bytes) -------------
// OR byte `b` into big-endian message position `pos` of w[16] without dynamic indexing.
VK_DEV void msg_put(u32 w[16], u32 pos, u32 b)
{
    u32 wi = pos >> 2, val = b << (24 - 8 * (pos & 3));
#pragma unroll
    for (int i = 0; i < 14; i++) w[i] |= (wi == (u32)i) ? val : 0u;
}

// append the decimal digits of v at byte position pos; returns new position
VK_DEV u32 msg_put_decimal(u32 w[16], u32 pos, u64 v)
{
    // count digits
    u32 nd = 1;
    u64 p = 10;
#pragma unroll 1
    for (int i = 0; i < 19 && v >= p; i++) {
        nd++;
        p *= 10;
    }
    u32 at = pos + nd;
#pragma unroll 1
    for (u32 i = 0; i < nd; i++) {
        u64 q = v / 10;
        msg_put(w, --at, (u32)('0' + (u32)(v - q * 10)));
        v = q;
    }
    return pos + nd;
}

// The decimal digits of v as the FIRST bytes of a big-endian packed message whose words the caller has zeroed, optionally followed
// by the 0x80 padding marker; returns the number of digits.  Digits come out least significant first and each is pushed in front of
// what is already there: a byte-wise right shift of a register window (one v_alignbit per word) with the new character entering at
// the top -- no byte position to decode, no selects (msg_put_decimal above costs ~45 instructions per digit, this ~8).  The marker
// starts out as byte 0 and ends up behind the last digit.  When every lane of the wave holds a value below 2^32 (every BASELINE
// config) the arithmetic is 32-bit and the window three words; otherwise six words and 64-bit divisions.
VK_DEV u32 msg_set_decimal(u32 w[16], u64 v, bool marker)
{
    u32 nd = 0;
    bool more = true;
    // one set of window registers for both paths, stored to w[] once behind them: with a store per path the compiler merged the two
    // tails into one store through a dynamic index and put the words into scratch
    u32 r0 = marker ? 0x80000000u : 0u, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0;
    if (__builtin_amdgcn_ballot_w64((v >> 32) != 0) == 0) {
        u32 x = (u32)v; // at most 10 digits + the marker: three words
        do {
            if (more) {
                const u32 q = __umulhi(x, 0xCCCCCCCDu) >> 3; // x / 10
                const u32 c = x - q * 10u + (u32)'0';
                r2 = __builtin_amdgcn_alignbit(r1, r2, 8);
                r1 = __builtin_amdgcn_alignbit(r0, r1, 8);
                r0 = __builtin_amdgcn_alignbit(c, r0, 8);
                x = q;
                nd++;
                more = x != 0;
            }
        } while (__builtin_amdgcn_ballot_w64(more) != 0);
    } else {
        u64 x = v; // at most 20 digits + the marker: six words
        do {
            if (more) {
                const u64 q = __umul64hi(x, 0xCCCCCCCCCCCCCCCDull) >> 3; // x / 10
                const u32 c = (u32)x - (u32)q * 10u + (u32)'0';
                r5 = __builtin_amdgcn_alignbit(r4, r5, 8);
                r4 = __builtin_amdgcn_alignbit(r3, r4, 8);
                r3 = __builtin_amdgcn_alignbit(r2, r3, 8);
                r2 = __builtin_amdgcn_alignbit(r1, r2, 8);
                r1 = __builtin_amdgcn_alignbit(r0, r1, 8);
                r0 = __builtin_amdgcn_alignbit(c, r0, 8);
                x = q;
                nd++;
                more = x != 0;
            }
        } while (__builtin_amdgcn_ballot_w64(more) != 0);
    }
    w[0] = r0;
    w[1] = r1;
    w[2] = r2;
    w[3] = r3;
    w[4] = r4;
    w[5] = r5;
    return nd;
}

VK_DEV void msg_finish(u32 w[16], u32 len)
{
    msg_put(w, len, 0x80u);
    w[14] = 0;
    w[15] = len * 8;
}

// ---- MT19937 (Milksad) ----------------------------------------------------------------
// one init_genrand step: 1812433253 * (prev ^ (prev >> 30)) + i as a single v_mad_u64_u32 (multiplier in a VGPR,
// the index as a 64-bit scalar operand); the low half of the result is the new state word
VK_DEV u32 mt_init_step(u32 prev, u32 i)
{
    const u32 x = prev ^ (prev >> 30);
    u64 r;
    const u32 mult = 1812433253u;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(x), "v"(mult), "s"((u64)i) : "vcc");
    return (u32)r;
}

// Only state words 0..8 and 397..404 of init_genrand(seed) feed the first 8 outputs.
VK_DEV bool transform_milksad(u32 kw[8], u64 seed)
{
    if (seed > 0xFFFFFFFFull) return false;
    u32 lo[9], hi[8];
    u32 m = (u32)seed;
    lo[0] = m;
#pragma unroll
    for (int i = 1; i <= 8; i++) {
        m = mt_init_step(m, (u32)i);
        lo[i] = m;
    }
#pragma unroll 4
    for (u32 i = 9; i < 397; i++) m = mt_init_step(m, i);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        m = mt_init_step(m, 397u + (u32)i);
        hi[i] = m;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 y = (lo[i] & 0x80000000u) | (lo[i + 1] & 0x7fffffffu);
        u32 v = hi[i] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        v ^= v >> 11;
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= v >> 18;
        kw[i] = bswap32(v); // key bytes 4i..4i+3 = LE32(v)
    }
    return true;
}

// Two seeds at once, the two init_genrand chains interleaved step by step.  A chain is 1212 dependent instructions; with two waves
// on a SIMD a second wave adds nothing to such a chain (the older wave's cadence leaves no room for the other's instructions),
// a second chain in the same lane makes a step 22 % cheaper (tools/ubench_chain.hip, profiles/r05_ubench_chain.txt) -- and no
// statement is followed by its own consumer, so the hazard nop behind every single-step statement goes as well.  The sweep kernel
// transforms its items in pairs (vk_kernels.cuh, pass 1).
VK_DEV void transform_milksad2(u32 kwa[8], u32 kwb[8], u32 sa, u32 sb)
{
    u32 loa[9], hia[8], lob[9], hib[8];
    u32 a = sa, b = sb;
    loa[0] = a;
    lob[0] = b;
#pragma unroll
    for (int i = 1; i <= 8; i++) {
        a = mt_init_step(a, (u32)i);
        b = mt_init_step(b, (u32)i);
        loa[i] = a;
        lob[i] = b;
    }
#pragma unroll 4
    for (u32 i = 9; i < 397; i++) {
        a = mt_init_step(a, i);
        b = mt_init_step(b, i);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a = mt_init_step(a, 397u + (u32)i);
        b = mt_init_step(b, 397u + (u32)i);
        hia[i] = a;
        hib[i] = b;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 y = (loa[i] & 0x80000000u) | (loa[i + 1] & 0x7fffffffu);
        u32 v = hia[i] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        v ^= v >> 11;
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= v >> 18;
        kwa[i] = bswap32(v);
        y = (lob[i] & 0x80000000u) | (lob[i + 1] & 0x7fffffffu);
        v = hib[i] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        v ^= v >> 11;
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= v >> 18;
        kwb[i] = bswap32(v);
    }
}

// ---- MT19937-64 -------------------------------------------------------------------------
VK_DEV u64 mt64_init_step(u64 prev, u64 i) { return 6364136223846793005ull * (prev ^ (prev >> 62)) + i; }

VK_DEV bool transform_mt64(u32 kw[8], u64 seed)
{
    u64 lo[5], hi[4];
    u64 m = seed;
    lo[0] = m;
#pragma unroll
    for (int i = 1; i <= 4; i++) {
        m = mt64_init_step(m, (u64)i);
        lo[i] = m;
    }
#pragma unroll 4
    for (u32 i = 5; i < 156; i++) m = mt64_init_step(m, (u64)i);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        m = mt64_init_step(m, 156ull + (u64)i);
        hi[i] = m;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 x = (lo[i] & 0xFFFFFFFF80000000ull) | (lo[i + 1] & 0x7FFFFFFFull);
        u64 v = hi[i] ^ (x >> 1) ^ ((x & 1ull) ? 0xB5026F5AA96619E9ull : 0ull);
        v ^= (v >> 29) & 0x5555555555555555ull;
        v ^= (v << 17) & 0x71D67FFFEDA60000ull;
        v ^= (v << 37) & 0xFFF7EEE000000000ull;
        v ^= v >> 43;
        // key bytes 8i..8i+7 = LE64(v)
        kw[2 * i] = bswap32((u32)v);
        kw[2 * i + 1] = bswap32((u32)(v >> 32));
    }
    return true;
}

// ---- LCG ------------------------------------------------------------------------------------
VK_DEV bool transform_lcg(u32 kw[8], u64 seed, u32 variant, u32 little_endian)
{
    u64 a, c, m;
    switch (variant) {
    case 0: a = 1103515245ull; c = 12345ull; m = 1ull << 31; break;
    case 1: a = 16807ull; c = 0ull; m = (1ull << 31) - 1; break;
    case 2: a = 214013ull; c = 2531011ull; m = 1ull << 32; break;
    case 3: a = 22695477ull; c = 1ull; m = 1ull << 32; break;
    default: return false;
    }
    if (seed > m - 1) return false;
    u64 st = (u32)seed;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 t = a * st + c; // < 2^63, no wrap
        if (variant == 1) {
            // mod 2^31 - 1 by folding (t < 2^46)
            t = (t & 0x7FFFFFFFull) + (t >> 31);
            t = (t & 0x7FFFFFFFull) + (t >> 31);
            if (t >= 0x7FFFFFFFull) t -= 0x7FFFFFFFull;
        } else {
            t &= (m - 1);
        }
        st = t;
        u32 w = (u32)st;
        kw[i] = little_endian ? bswap32(w) : w;
    }
    return true;
}

// ---- xorshift family ---------------------------------------------------------------------
VK_DEV u64 rotl64(u64 x, int n) { return (x << n) | (x >> (64 - n)); }
VK_DEV u64 splitmix64(u64 x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

VK_DEV bool transform_xorshift(u32 kw[8], u64 seed, u32 variant)
{
    if (seed == 0) seed = 1;
    if (variant == 0) {
        u64 x = seed;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            x ^= x << 13;
            x ^= x >> 7;
            x ^= x << 17;
            kw[2 * i] = (u32)(x >> 32);
            kw[2 * i + 1] = (u32)x;
        }
        return true;
    }
    if (variant == 1) {
        u32 x = (u32)seed, y = (u32)(seed >> 32);
        u32 z = (u32)seed ^ 0x12345678u, w = (u32)(seed >> 32) ^ 0x87654321u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            u32 t = x ^ (x << 11);
            x = y;
            y = z;
            z = w;
            w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
            kw[i] = w;
        }
        return true;
    }
    if (variant == 2 || variant == 3) {
        u64 s0 = splitmix64(seed), s1 = splitmix64(s0);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            u64 r = (variant == 2) ? s0 + s1 : rotl64(s0 * 5, 7) * 9;
            u64 t = s1 ^ s0;
            s0 = rotl64(s0, 24) ^ t ^ (t << 16);
            s1 = rotl64(t, 37);
            kw[2 * i] = (u32)(r >> 32);
            kw[2 * i + 1] = (u32)r;
        }
        return true;
    }
    return false;
}

// ---- sha256 ------------------------------------------------------------------------------------
VK_DEV bool transform_sha256(u32 kw[8], u64 seed, u32 sub)
{
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = 0;
    if (sub == 0) {
        const u32 len = msg_set_decimal(w, seed, true); // digits + 0x80; w[14] stays 0
        w[15] = len * 8;
    } else if (sub == 1) {
        w[0] = (u32)(seed >> 32);
        w[1] = (u32)seed;
        w[2] = 0x80000000u;
        w[15] = 64;
    } else if (sub == 2) {
        w[0] = bswap32((u32)seed);
        w[1] = bswap32((u32)(seed >> 32));
        w[2] = 0x80000000u;
        w[15] = 64;
    } else {
        return false;
    }
    sha256_oneblock(kw, w);
    return true;
}

// ---- double_sha256: SHA256(SHA256(x)) of the same three forms, same order -----------------------
VK_DEV bool transform_double_sha256(u32 kw[8], u64 seed, u32 sub)
{
    u32 t[8];
    if (!transform_sha256(t, seed, sub)) return false;
    sha256_of_digest(kw, t);
    return true;
}

// ---- md5: MD5(decimal string) written twice ------------------------------------------------------
VK_DEV bool transform_md5(u32 kw[8], u64 seed)
{
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = 0;
    const u32 len = msg_set_decimal(w, seed, true); // <= 20 bytes + the marker, big-endian packed
    u32 m[16];
#pragma unroll
    for (int i = 0; i < 14; i++) m[i] = bswap32(w[i]);
    m[14] = len * 8;
    m[15] = 0;
    u32 st[4] = VK_MD5_IV;
    md5_compress(st, m);
#pragma unroll
    for (int i = 0; i < 4; i++) kw[i] = kw[4 + i] = bswap32(st[i]);
    return true;
}

// ---- direct: slot 0 = 8 BE bytes right-aligned, 1 = 8 LE bytes left-aligned, 2 = the decimal string, zero padded
VK_DEV bool transform_direct(u32 kw[8], u64 seed, u32 sub)
{
#pragma unroll
    for (int i = 0; i < 8; i++) kw[i] = 0;
    if (sub == 0) {
        kw[6] = (u32)(seed >> 32);
        kw[7] = (u32)seed;
    } else if (sub == 1) {
        kw[0] = bswap32((u32)seed);
        kw[1] = bswap32((u32)(seed >> 32));
    } else if (sub == 2) {
        u32 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = 0;
        msg_set_decimal(w, seed, false);
#pragma unroll
        for (int i = 0; i < 8; i++) kw[i] = w[i];
    } else {
        return false;
    }
    return true;
}

// ---- sha256_chain ---------------------------------------------------------------------------
// slot = form*(V*D) + v*D + idx; form 0: seed bytes = decimal string, form 1: 4-byte BE (u32 only)
VK_DEV bool transform_sha256_chain(u32 kw[8], u64 seed, u32 sub, u32 spec_variant, u32 depth)
{
    u32 nv = (spec_variant == VK_VARIANT_ALL) ? 4u : 1u;
    if (depth == 0 || sub >= 2u * nv * depth) return false;
    u32 form = sub / (nv * depth), r = sub % (nv * depth);
    u32 v = (spec_variant == VK_VARIANT_ALL) ? r / depth : spec_variant, idx = r % depth;
    if (form == 1 && seed > 0xFFFFFFFFull) return false;
    if (v > 3) return false;
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = 0;
    u32 len;
    if (form == 0) {
        len = msg_set_decimal(w, seed, false);
    } else {
        w[0] = (u32)seed;
        len = 4;
    }
    if (v == 1 || v == 2) {
        u32 iw = (v == 1) ? idx : bswap32(idx);
        msg_put(w, len, iw >> 24);
        msg_put(w, len + 1, (iw >> 16) & 0xff);
        msg_put(w, len + 2, (iw >> 8) & 0xff);
        msg_put(w, len + 3, iw & 0xff);
        len += 4;
    } else if (v == 3) {
        len = msg_put_decimal(w, len, (u64)idx);
    }
    msg_finish(w, len);
    sha256_oneblock(kw, w);
    if (v == 0) {
#pragma unroll 1
        for (u32 i = 0; i < idx; i++) {
            u32 t[8];
            sha256_of_digest(t, kw);
#pragma unroll
            for (int j = 0; j < 8; j++) kw[j] = t[j];
        }
    }
    return true;
}

// ---- dispatch ------------------------------------------------------------------------------
VK_DEV u32 spec_slots(const vk_spec_dev &s)
{
    switch (s.kind) {
    case VK_T_SHA256:
    case VK_T_DOUBLE_SHA256:
    case VK_T_DIRECT: return 3;
    case VK_T_MILKSAD:
    case VK_T_MD5:
    case VK_T_MT64: return 1;
    case 11u /* VK_T_ARMORY */: return 2;
    case VK_T_LCG:
    case VK_T_XORSHIFT: return s.variant == VK_VARIANT_ALL ? 4u : 1u;
    case VK_T_SHA256_CHAIN: return 2u * (s.variant == VK_VARIANT_ALL ? 4u : 1u) * s.chain_depth;
    default: return 1;
    }
}

template <u32 KIND>
VK_DEV bool transform_key(u32 kw[8], const vk_spec_dev &s, u64 seed, u32 sub)
{
    if (KIND == VK_T_SHA256) return transform_sha256(kw, seed, sub);
    if (KIND == VK_T_MILKSAD) return transform_milksad(kw, seed);
    if (KIND == VK_T_MT64) return transform_mt64(kw, seed);
    if (KIND == VK_T_LCG) return transform_lcg(kw, seed, s.variant == VK_VARIANT_ALL ? sub : s.variant, s.endian);
    if (KIND == VK_T_XORSHIFT) return transform_xorshift(kw, seed, s.variant == VK_VARIANT_ALL ? sub : s.variant);
    if (KIND == VK_T_SHA256_CHAIN) return transform_sha256_chain(kw, seed, sub, s.variant, s.chain_depth);
    if (KIND == VK_T_DOUBLE_SHA256) return transform_double_sha256(kw, seed, sub);
    if (KIND == VK_T_MD5) return transform_md5(kw, seed);
    if (KIND == VK_T_DIRECT) return transform_direct(kw, seed, sub);
    return false;
}

// ---- string inputs (Input::from_string: no numeric forms): one key per string for the hash transforms,
// variants x depth for sha256_chain (src/transform/sha256_chain.rs:60-72).  Returns false where nothing is emitted.
VK_DEV u32 string_slots(const vk_spec_dev &s)
{
    switch (s.kind) {
    case VK_T_SHA256:
    case VK_T_DOUBLE_SHA256:
    case VK_T_MD5:
    case VK_T_DIRECT:
    case 11u /* VK_T_ARMORY */: return 1;
    case VK_T_SHA256_CHAIN: return (s.variant == VK_VARIANT_ALL ? 4u : 1u) * s.chain_depth;
    default: return 0;
    }
}

VK_DEV bool transform_string(u32 kw[8], const vk_spec_dev &s, const unsigned char *str, u32 len, u32 sub)
{
    vk_bytes b;
    b.p = str;
    b.len = len;
    b.suf[0] = b.suf[1] = b.suf[2] = 0;
    b.suf_len = 0;
    if (s.kind == VK_T_SHA256) {
        sha256_stream(kw, b);
        return true;
    }
    if (s.kind == VK_T_DOUBLE_SHA256) {
        u32 t[8];
        sha256_stream(t, b);
        sha256_of_digest(kw, t);
        return true;
    }
    if (s.kind == VK_T_MD5) {
        u32 st[4];
        md5_stream(st, b);
#pragma unroll
        for (int i = 0; i < 4; i++) kw[i] = kw[4 + i] = bswap32(st[i]);
        return true;
    }
    if (s.kind == VK_T_DIRECT) {
        if (len > 32) return false;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            u32 w = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) w |= ((u32)(4 * i + k) < len ? (u32)str[4 * i + k] : 0u) << (24 - 8 * k);
            kw[i] = w;
        }
        return true;
    }
    if (s.kind == VK_T_SHA256_CHAIN) {
        const u32 depth = s.chain_depth;
        if (depth == 0) return false;
        const u32 v = s.variant == VK_VARIANT_ALL ? sub / depth : s.variant, idx = sub % depth;
        if (v > 3 || (s.variant != VK_VARIANT_ALL && sub >= depth)) return false;
        if (v == 1 || v == 2) {
            b.suf[0] = v == 1 ? idx : bswap32(idx);
            b.suf_len = 4;
        } else if (v == 3) {
            u32 w[16];
#pragma unroll
            for (int i = 0; i < 16; i++) w[i] = 0;
            b.suf_len = msg_put_decimal(w, 0, (u64)idx);
            b.suf[0] = w[0];
            b.suf[1] = w[1];
            b.suf[2] = w[2];
        }
        sha256_stream(kw, b);
        if (v == 0) {
#pragma unroll 1
            for (u32 i = 0; i < idx; i++) {
                u32 t[8];
                sha256_of_digest(t, kw);
#pragma unroll
                for (int j = 0; j < 8; j++) kw[j] = t[j];
            }
        }
        return true;
    }
    return false;
}
