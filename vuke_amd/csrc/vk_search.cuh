// vk_search.cuh -- analyzer brute force on gfx950: the smallest seed of a range whose generated key(s)
// satisfy a target.  This is the reference's `vuke analyze` search loop -- its only GPU user (wgpu,
// src/gpu/mt19937.rs, exact match only) -- for the generators whose seed space can be swept:
//   milksad  exact / masked / cascade   src/analyze/milksad.rs:69-366
//   lcg      exact / masked             src/analyze/lcg.rs:49-210 (one variant per launch)
//   mt64     cascade                    src/analyze/mt64.rs:60-215
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
    return false;
}

template <u32 KIND, int ND>
__global__ void __launch_bounds__(VK_BLOCK) vk_search_kernel(const vk_search_params p)
{
    const u64 stride = (u64)gridDim.x * VK_BLOCK;
    u64 best = ~0ull;
    for (u64 off = (u64)blockIdx.x * VK_BLOCK + threadIdx.x; off <= p.span; off += stride) {
        const u64 seed = p.lo + off;
        u32 kw[ND][8];
        if (search_draws<KIND, ND>(kw, p.spec, seed) && search_match<ND>(kw, p)) {
            best = seed; // offsets grow within a lane: the first match is this lane's smallest
            break;
        }
        if (off + stride < off) break; // u64 wrap guard
    }
    if (best != ~0ull) atomicMin((unsigned long long *)p.best, (unsigned long long)best);
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
