// vk_search.cuh -- analyzer brute force on gfx950: the smallest seed of a range whose generated key(s)
// satisfy a target.  This is the reference's `vuke analyze` search loop -- its only GPU user (wgpu,
// src/gpu/mt19937.rs, exact match only) -- for the generators whose seed space can be swept:
//   milksad  exact / masked / cascade   src/analyze/milksad.rs:69-366
//   lcg      exact / masked             src/analyze/lcg.rs:49-210 (one variant per launch)
//   mt64     cascade                    src/analyze/mt64.rs:60-215
//   xorshift cascade (one variant per launch)             src/analyze/xorshift.rs:30-206
//   sha256_chain exact / masked / cascade, u32 seeds as 4 BE bytes    src/analyze/sha256_chain.rs:48-425
// masked:  ((last 8 key bytes as BE u64) & (2^bits - 1)) | 2^(bits-1) == target
// cascade: successive 32-byte draws of one generator instance, draw j masked with bits[j] == targets[j].
// No EC, no hashing: a few hundred to ~1.8 k integer instructions per seed.  The reference returns
// whichever matching seed a thread reports first; this kernel returns the smallest one (atomicMin).
#pragma once
#include "vk_transforms.cuh"

#ifndef VK_BLOCK
#define VK_BLOCK 256
#endif

#define VK_SEARCH_EXACT 0u
#define VK_SEARCH_MASKED 1u
#define VK_SEARCH_CASCADE 2u
#define VK_SEARCH_MAX_DRAWS 8

struct vk_search_params {
    vk_spec_dev spec;
    u64 lo, span;
    u32 mode, ndraws;
    u32 key[8];                       // exact: target key as big-endian words
    u32 bits[VK_SEARCH_MAX_DRAWS];    // masked: bits[0]
    u64 targets[VK_SEARCH_MAX_DRAWS]; // masked: targets[0] = last 8 bytes of the target key
    u64 *best;                        // smallest matching seed (initialised to ~0)
};

// ND successive 32-byte draws of MT19937 (8 outputs each, little-endian bytes)
template <int ND>
VK_DEV void milksad_draws(u32 kw[ND][8], u32 seed)
{
    u32 lo[8 * ND + 1], hi[8 * ND];
    u32 m = seed;
    lo[0] = m;
#pragma unroll
    for (int i = 1; i <= 8 * ND; i++) {
        m = mt_init_step(m, (u32)i);
        lo[i] = m;
    }
#pragma unroll 4
    for (u32 i = 8 * ND + 1; i < 397; i++) m = mt_init_step(m, i);
#pragma unroll
    for (int i = 0; i < 8 * ND; i++) {
        m = mt_init_step(m, 397u + (u32)i);
        hi[i] = m;
    }
#pragma unroll
    for (int i = 0; i < 8 * ND; i++) {
        u32 y = (lo[i] & 0x80000000u) | (lo[i + 1] & 0x7fffffffu);
        u32 v = hi[i] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        v ^= v >> 11;
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= v >> 18;
        kw[i / 8][i % 8] = bswap32(v);
    }
}

// ND successive 32-byte draws of MT19937-64 (4 outputs each)
template <int ND>
VK_DEV void mt64_draws(u32 kw[ND][8], u64 seed)
{
    u64 lo[4 * ND + 1], hi[4 * ND];
    u64 m = seed;
    lo[0] = m;
#pragma unroll
    for (int i = 1; i <= 4 * ND; i++) {
        m = mt64_init_step(m, (u64)i);
        lo[i] = m;
    }
#pragma unroll 4
    for (u32 i = 4 * ND + 1; i < 156; i++) m = mt64_init_step(m, (u64)i);
#pragma unroll
    for (int i = 0; i < 4 * ND; i++) {
        m = mt64_init_step(m, 156ull + (u64)i);
        hi[i] = m;
    }
#pragma unroll
    for (int i = 0; i < 4 * ND; i++) {
        u64 x = (lo[i] & 0xFFFFFFFF80000000ull) | (lo[i + 1] & 0x7FFFFFFFull);
        u64 v = hi[i] ^ (x >> 1) ^ ((x & 1ull) ? 0xB5026F5AA96619E9ull : 0ull);
        v ^= (v >> 29) & 0x5555555555555555ull;
        v ^= (v << 17) & 0x71D67FFFEDA60000ull;
        v ^= (v << 37) & 0xFFF7EEE000000000ull;
        v ^= v >> 43;
        kw[i / 4][2 * (i % 4)] = bswap32((u32)v);
        kw[i / 4][2 * (i % 4) + 1] = bswap32((u32)(v >> 32));
    }
}

// ND successive 32-byte draws of one xorshift-family generator (big-endian fill, src/xorshift.rs:108-118 etc.)
template <int ND>
VK_DEV void xorshift_draws(u32 kw[ND][8], u64 seed, u32 variant)
{
    if (seed == 0) seed = 1;
    if (variant == 0) {
        u64 x = seed;
#pragma unroll
        for (int i = 0; i < 4 * ND; i++) {
            x ^= x << 13;
            x ^= x >> 7;
            x ^= x << 17;
            kw[i / 4][2 * (i % 4)] = (u32)(x >> 32);
            kw[i / 4][2 * (i % 4) + 1] = (u32)x;
        }
    } else if (variant == 1) {
        u32 x = (u32)seed, y = (u32)(seed >> 32);
        u32 z = (u32)seed ^ 0x12345678u, w = (u32)(seed >> 32) ^ 0x87654321u;
#pragma unroll
        for (int i = 0; i < 8 * ND; i++) {
            u32 t = x ^ (x << 11);
            x = y;
            y = z;
            z = w;
            w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
            kw[i / 8][i % 8] = w;
        }
    } else {
        u64 s0 = splitmix64(seed), s1 = splitmix64(s0);
#pragma unroll
        for (int i = 0; i < 4 * ND; i++) {
            u64 r = (variant == 2) ? s0 + s1 : rotl64(s0 * 5, 7) * 9;
            u64 t = s1 ^ s0;
            s0 = rotl64(s0, 24) ^ t ^ (t << 16);
            s1 = rotl64(t, 37);
            kw[i / 4][2 * (i % 4)] = (u32)(r >> 32);
            kw[i / 4][2 * (i % 4) + 1] = (u32)r;
        }
    }
}

VK_DEV bool masked_eq(const u32 kw[8], u32 bits, u64 target)
{
    const u64 k = ((u64)kw[6] << 32) | kw[7];
    const u64 mask = bits >= 64 ? ~0ull : (1ull << bits) - 1;
    const u64 high = 1ull << (bits - 1);
    return ((k & mask) | high) == target;
}

template <int ND>
VK_DEV bool search_match(const u32 kw[ND][8], const vk_search_params &p)
{
    if (p.mode == VK_SEARCH_EXACT) {
        u32 d = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d |= kw[0][i] ^ p.key[i];
        return d == 0;
    }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < ND; j++)
        if ((u32)j < p.ndraws) ok = ok && masked_eq(kw[j], p.bits[j], p.targets[j]);
    return ok;
}

// draws of (spec, seed): returns false where the transform emits nothing for that seed
template <u32 KIND, int ND>
VK_DEV bool search_draws(u32 kw[ND][8], const vk_spec_dev &s, u64 seed)
{
    if (KIND == VK_T_MILKSAD) {
        if (seed > 0xFFFFFFFFull) return false;
        milksad_draws<ND>(kw, (u32)seed);
        return true;
    }
    if (KIND == VK_T_MT64) {
        mt64_draws<ND>(kw, seed);
        return true;
    }
    if (KIND == VK_T_LCG) return transform_lcg(kw[0], seed, s.variant, s.endian); // ND == 1
    if (KIND == VK_T_XORSHIFT) {
        xorshift_draws<ND>(kw, seed, s.variant);
        return true;
    }
    return false;
}

// ---- sha256_chain: every key of generate_chain(seed.to_be_bytes(), variant, depth) (src/sha256_chain.rs:88-145)
VK_DEV void chain_key(u32 kw[8], u32 seed, u32 v, u32 idx, const u32 prev[8])
{
    if (v == 0 && idx > 0) { // iterated: k_i = SHA256(k_{i-1})
        u32 t[8];
        sha256_of_digest(t, prev);
#pragma unroll
        for (int j = 0; j < 8; j++) kw[j] = t[j];
        return;
    }
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = 0;
    w[0] = seed;
    u32 len = 4;
    if (v == 1 || v == 2) {
        w[1] = (v == 1) ? idx : bswap32(idx);
        len = 8;
    } else if (v == 3) {
        len = msg_put_decimal(w, 4, (u64)idx);
    }
    msg_finish(w, len);
    sha256_oneblock(kw, w);
}

// reference iteration order: variants in ALL_VARIANTS order, chain index ascending; the first match wins.
// exact / masked: any key of the chain; cascade: chain[j] against target j for every j (depth = ndraws).
VK_DEV bool chain_search_seed(const vk_search_params &p, u32 seed, u32 &hit_variant, u32 &hit_index)
{
    const bool all = p.spec.variant == VK_VARIANT_ALL;
    const u32 v0 = all ? 0u : p.spec.variant, v1 = all ? 3u : p.spec.variant;
    const u32 depth = p.mode == VK_SEARCH_CASCADE ? p.ndraws : p.spec.chain_depth;
    for (u32 v = v0; v <= v1; v++) {
        u32 kw[8], prev[8];
#pragma unroll
        for (int j = 0; j < 8; j++) prev[j] = 0;
        bool ok = true;
#pragma unroll 1
        for (u32 idx = 0; idx < depth; idx++) {
            chain_key(kw, seed, v, idx, prev);
#pragma unroll
            for (int j = 0; j < 8; j++) prev[j] = kw[j];
            if (p.mode == VK_SEARCH_EXACT) {
                u32 d = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) d |= kw[j] ^ p.key[j];
                if (d == 0) {
                    hit_variant = v;
                    hit_index = idx;
                    return true;
                }
            } else if (p.mode == VK_SEARCH_MASKED) {
                if (masked_eq(kw, p.bits[0], p.targets[0])) {
                    hit_variant = v;
                    hit_index = idx;
                    return true;
                }
            } else if (!masked_eq(kw, p.bits[idx], p.targets[idx])) {
                ok = false;
                break;
            }
        }
        if (p.mode == VK_SEARCH_CASCADE && ok) {
            hit_variant = v;
            hit_index = 0;
            return true;
        }
    }
    return false;
}

template <u32 KIND, int ND>
__global__ void __launch_bounds__(VK_BLOCK) vk_search_kernel(const vk_search_params p)
{
    const u64 stride = (u64)gridDim.x * VK_BLOCK;
    u64 best = ~0ull;
    for (u64 off = (u64)blockIdx.x * VK_BLOCK + threadIdx.x; off <= p.span; off += stride) {
        const u64 seed = p.lo + off;
        bool hit;
        if (KIND == VK_T_SHA256_CHAIN) {
            u32 hv, hx;
            hit = seed <= 0xFFFFFFFFull && chain_search_seed(p, (u32)seed, hv, hx);
        } else {
            u32 kw[ND][8];
            hit = search_draws<KIND, ND>(kw, p.spec, seed) && search_match<ND>(kw, p);
        }
        if (hit) {
            best = seed; // offsets grow within a lane: the first match is this lane's smallest
            break;
        }
        if (off + stride < off) break; // u64 wrap guard
    }
    if (best != ~0ull) atomicMin((unsigned long long *)p.best, (unsigned long long)best);
}

// Any other numeric transform (sha256, double_sha256, md5, direct): exact / masked against every key the transform
// emits for a seed (slot order = emit order).  Not a reference analyzer: it is the transform-only sweep that the
// reference's `vuke bench` harness times (src/benchmark.rs:11-67), and an exact-key lookup for those transforms.
template <u32 KIND>
VK_DEV bool slots_search_seed(const vk_search_params &p, u64 seed, u32 &hit_sub, u32 kw_out[8])
{
    const u32 slots = spec_slots(p.spec);
#pragma unroll 1
    for (u32 sub = 0; sub < slots; sub++) {
        u32 kw[1][8];
        if (transform_key<KIND>(kw[0], p.spec, seed, sub) && search_match<1>(kw, p)) {
            hit_sub = sub;
#pragma unroll
            for (int j = 0; j < 8; j++) kw_out[j] = kw[0][j];
            return true;
        }
    }
    return false;
}

template <u32 KIND>
__global__ void __launch_bounds__(VK_BLOCK) vk_search_slots_kernel(const vk_search_params p)
{
    const u64 stride = (u64)gridDim.x * VK_BLOCK;
    u64 best = ~0ull;
    for (u64 off = (u64)blockIdx.x * VK_BLOCK + threadIdx.x; off <= p.span; off += stride) {
        u32 sub, kw[8];
        if (slots_search_seed<KIND>(p, p.lo + off, sub, kw)) {
            best = p.lo + off;
            break;
        }
        if (off + stride < off) break;
    }
    if (best != ~0ull) atomicMin((unsigned long long *)p.best, (unsigned long long)best);
}

template <u32 KIND>
__global__ void vk_search_emit_slots_kernel(const vk_search_params p, u64 seed, u32 *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u32 sub = 0, kw[8];
    if (!slots_search_seed<KIND>(p, seed, sub, kw)) return;
#pragma unroll
    for (int j = 0; j < 8; j++) out[j] = bswap32(kw[j]);
    out[64] = 0;
    out[65] = sub;
}

// the ND draws of one seed, as raw key bytes (for the result record)
template <u32 KIND, int ND>
__global__ void vk_search_emit_kernel(vk_spec_dev spec, u64 seed, u32 *out /* ND x 8 words */)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u32 kw[ND][8];
    if (!search_draws<KIND, ND>(kw, spec, seed)) return;
#pragma unroll
    for (int j = 0; j < ND; j++)
#pragma unroll
        for (int i = 0; i < 8; i++) out[8 * j + i] = bswap32(kw[j][i]);
}

// sha256_chain: which (variant, index) matched for `seed`, and the key(s) the reference prints:
// exact / masked: the matching key; cascade: chain[0 .. ndraws).  out = 8 x 8 words, then variant, index.
__global__ void vk_search_emit_chain_kernel(const vk_search_params p, u64 seed, u32 *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u32 hv = 0, hx = 0;
    if (seed > 0xFFFFFFFFull || !chain_search_seed(p, (u32)seed, hv, hx)) return;
    out[64] = hv;
    out[65] = hx;
    const u32 last = p.mode == VK_SEARCH_CASCADE ? p.ndraws - 1 : hx;
    u32 kw[8], prev[8];
#pragma unroll
    for (int j = 0; j < 8; j++) prev[j] = 0;
#pragma unroll 1
    for (u32 idx = 0; idx <= last; idx++) {
        chain_key(kw, (u32)seed, hv, idx, prev);
#pragma unroll
        for (int j = 0; j < 8; j++) prev[j] = kw[j];
        if (p.mode == VK_SEARCH_CASCADE) {
#pragma unroll
            for (int j = 0; j < 8; j++) out[8 * idx + j] = bswap32(kw[j]);
        }
    }
    if (p.mode != VK_SEARCH_CASCADE)
#pragma unroll
        for (int j = 0; j < 8; j++) out[j] = bswap32(kw[j]);
}
