// vk_field.cuh -- secp256k1 base-field and scalar arithmetic for gfx950 lanes.
//
// One field element = 8 x u32 little-endian limbs held in VGPRs.  Values are kept
// "weakly reduced": any 256-bit residue (possibly >= p); carries out of 2^256 are folded
// back with 2^256 == C (mod p), C = 2^32 + 977.  fe_canon() produces the unique
// representative before bytes leave the lane.  The multiplier is operand-scanning
// schoolbook on v_mad_u64_u32 (32x32+64 -> 64), the only wide multiplier CDNA4's VALU has.
//
// Replaces (behaviourally) the field code of libsecp256k1 that the reference reaches via
// src/derive.rs:97; written from SEC2 parameters, not from that library.
#pragma once
#ifndef VK_HOST_EMU
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

#ifndef VK_HOST_EMU
#define VK_DEV __device__ __forceinline__
#else
// ---- VK_HOST_EMU: the field layer compiled for the host, one lane, every generated asm statement replaced by calls into the
// instruction emulator below (vk_field_gen_host.h, written by tools/gen_field.py next to the device header).  TEST
// INFRASTRUCTURE: tests/test_host_field.py builds it with g++ and holds the arithmetic to python integers on a machine without
// a GPU.  Nothing in the device build sees this branch.
#define VK_DEV static inline
#define __device__
#define __noinline__ __attribute__((noinline))
#define __constant__
#define __builtin_amdgcn_ballot_w64(x) ((x) ? 1ull : 0ull)
// semantics of the ten instructions the generator emits, for one lane; carries are the lane's bit 0 of a 64-bit mask
static inline void emu_mad(u64 &d, u64 &c, u32 a, u32 b, u64 add)
{
    const unsigned __int128 t = (unsigned __int128)a * b + add;
    d = (u64)t;
    c = (u64)(t >> 64) & 1u;
}
static inline void emu_add_co(u32 &d, u64 &c, u32 x, u32 y) { const u64 t = (u64)x + y; d = (u32)t; c = t >> 32; }
static inline void emu_addc_co(u32 &d, u64 &co, u32 x, u32 y, u64 ci) { const u64 t = (u64)x + y + (ci & 1u); d = (u32)t; co = t >> 32; }
static inline void emu_sub_co(u32 &d, u64 &c, u32 x, u32 y) { c = x < y; d = x - y; }
static inline void emu_subb_co(u32 &d, u64 &co, u32 x, u32 y, u64 ci) { const u64 s = (u64)y + (ci & 1u); const u64 b = (u64)x < s; d = (u32)((u64)x - s); co = b; }
static inline void emu_cndmask(u32 &d, u32 x, u32 y, u64 m) { d = (m & 1u) ? y : x; }
static inline void emu_add(u32 &d, u32 x, u32 y) { d = x + y; }
static inline void emu_mul_u24(u32 &d, u32 x, u32 y) { d = (x & 0xFFFFFFu) * (y & 0xFFFFFFu); }
static inline void emu_lshlrev(u32 &d, u32 sh, u32 x) { d = x << (sh & 31u); }
static inline void emu_alignbit(u32 &d, u32 hi, u32 lo, u32 sh) { d = (u32)((((u64)hi << 32) | lo) >> (sh & 31u)); }
static inline void emu_s_or(u64 &d, u64 x, u64 y) { d = x | y; }
#endif

struct fe {
    u32 v[8];
};

#define VK_PC0 977u /* C = 2^32 + 977 */

VK_DEV void fe_set_zero(fe &r)
{
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
}
VK_DEV void fe_set_one(fe &r)
{
    fe_set_zero(r);
    r.v[0] = 1;
}

// ---- wide multiply on v_mad_u64_u32 -----------------------------------------------------
// Product scanning over the 15 columns k = i + j.  Even and odd columns are kept as two separate
// multi-precision numbers E and O: column k accumulates its 64-bit products into the register
// pair (limb k, limb k+1); an overflow of that pair (the multiplier's carry-out, vcc) is counted
// into cy[k], which is limb k+2 -- exactly where the next column of the same parity starts, so
// no register shuffling is needed.  One carry chain E + O then yields the 16-limb product.
// Every wide instruction here costs one 4-cycle issue slot (profiles/r01_ubench_valu_rates.txt):
// 64 multiplies + 49 carry counts + 15 merge adds, versus ~320 instructions (two thirds of them
// v_mov / 64-bit adds) for the same product written in plain C++.

#ifdef VK_HOST_EMU
typedef unsigned long vk_carry;
VK_DEV void mad_nc(u64 &acc, u32 a, u32 b) { u64 c; emu_mad(acc, c, a, b, acc); }
VK_DEV void mad_cy(u64 &acc, u32 &cy, u32 a, u32 b) { u64 c; emu_mad(acc, c, a, b, acc); emu_addc_co(cy, c, 0u, cy, c); }
VK_DEV u32 add_co(vk_carry &co, u32 x, u32 y) { u32 r; u64 c; emu_add_co(r, c, x, y); co = c; return r; }
VK_DEV u32 addc_co(vk_carry &co, u32 x, u32 y, vk_carry ci) { u32 r; u64 c; emu_addc_co(r, c, x, y, ci); co = c; return r; }
VK_DEV u32 addc(u32 x, u32 y, vk_carry ci) { u32 r; u64 c; emu_addc_co(r, c, x, y, ci); return r; }
VK_DEV u64 mul_wide_s(u32 a, u32 s) { u64 r, c; emu_mad(r, c, a, s, 0ull); return r; }
VK_DEV bool lane_in_mask(vk_carry mask) { return mask & 1ull; }
#else
// acc = a * b + acc; the sum is known not to overflow 64 bits
VK_DEV void mad_nc(u64 &acc, u32 a, u32 b)
{
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
// acc = a * b + acc (mod 2^64); cy += carry out
VK_DEV void mad_cy(u64 &acc, u32 &cy, u32 a, u32 b)
{
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(cy) : "v"(a), "v"(b) : "vcc");
}
// carry chains with the carry in an SGPR pair, so the compiler can schedule freely between links
typedef unsigned long vk_carry;
VK_DEV u32 add_co(vk_carry &co, u32 x, u32 y)
{
    u32 r;
    asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(r), "=s"(co) : "v"(x), "v"(y));
    return r;
}
VK_DEV u32 addc_co(vk_carry &co, u32 x, u32 y, vk_carry ci)
{
    u32 r;
    asm("v_addc_co_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(co) : "v"(x), "v"(y), "s"(ci));
    return r;
}
VK_DEV u32 addc(u32 x, u32 y, vk_carry ci) // last link: carry out is known to be zero
{
    u32 r;
    vk_carry co;
    asm("v_addc_co_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(co) : "v"(x), "v"(y), "s"(ci));
    return r;
}

// a * s (s in an SGPR), full 64-bit product
VK_DEV u64 mul_wide_s(u32 a, u32 s)
{
    u64 r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a), "s"(s) : "vcc");
    return r;
}

VK_DEV bool lane_in_mask(vk_carry mask)
{
    u32 lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return (mask >> lane) & 1ull;
}
#endif // VK_HOST_EMU
// rare paths of the generated add / sub / reduce: one more +-C where a lane wrapped 2^256 twice
VK_DEV void fe_wrap_add(fe &r, vk_carry mask)
{
    if (!lane_in_mask(mask)) return;
    u64 c = (u64)r.v[0] + VK_PC0;
    r.v[0] = (u32)c;
    c = (c >> 32) + r.v[1] + 1u;
    r.v[1] = (u32)c;
#pragma unroll
    for (int i = 2; i < 8; i++) {
        c = (c >> 32) + r.v[i];
        r.v[i] = (u32)c;
    }
}
VK_DEV void fe_wrap_sub(fe &r, vk_carry mask)
{
    if (!lane_in_mask(mask)) return;
    u64 d = (u64)r.v[0] - VK_PC0;
    r.v[0] = (u32)d;
    d = (u64)r.v[1] - 1u - ((d >> 32) & 1);
    r.v[1] = (u32)d;
#pragma unroll
    for (int i = 2; i < 8; i++) {
        d = (u64)r.v[i] - ((d >> 32) & 1);
        r.v[i] = (u32)d;
    }
}

// The fix-ups of the generated add / sub / reduce touch limbs 0..1 (0..2 in the reduction) and almost never carry further
// (one operation in 2^31), so their carry chains stop there and the lanes that do carry come here: +-1 from limb `from`
// upwards; a carry out of limb 7 is one more wrap of 2^256, i.e. one more +-C (the value is tiny then: no third wrap).
VK_DEV void fe_ripple_add(fe &r, int from, vk_carry mask)
{
    if (!lane_in_mask(mask)) return;
    u32 c = 1;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i < from) continue;
        const u32 v = r.v[i] + c;
        c = (v < c) ? 1u : 0u;
        r.v[i] = v;
    }
    if (c) {
        u64 t = (u64)r.v[0] + VK_PC0;
        r.v[0] = (u32)t;
        t = (t >> 32) + r.v[1] + 1u;
        r.v[1] = (u32)t;
#pragma unroll
        for (int i = 2; i < 8; i++) {
            t = (t >> 32) + r.v[i];
            r.v[i] = (u32)t;
        }
    }
}
VK_DEV void fe_ripple_sub(fe &r, int from, vk_carry mask)
{
    if (!lane_in_mask(mask)) return;
    u32 b = 1;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i < from) continue;
        const u32 v = r.v[i] - b;
        b = (r.v[i] < b) ? 1u : 0u;
        r.v[i] = v;
    }
    if (b) {
        u64 d = (u64)r.v[0] - VK_PC0;
        r.v[0] = (u32)d;
        d = (u64)r.v[1] - 1u - ((d >> 32) & 1);
        r.v[1] = (u32)d;
#pragma unroll
        for (int i = 2; i < 8; i++) {
            d = (u64)r.v[i] - ((d >> 32) & 1);
            r.v[i] = (u32)d;
        }
    }
}

#ifdef VK_HOST_EMU
#include "vk_field_gen_host.h" // the same statements for one lane on the host (tests/test_host_field.py)
#else
#include "vk_field_gen.cuh" // fe_mul_cols, fe_sqr_cols, merge_eo (generated by tools/gen_field.py)
#endif

// How the 16-limb product is reduced (tools/gen_field.py):
//   1  fe_reduce16_flags: 19 VALU instructions, no precondition test; every carry the short form can lose is flagged in an
//      SGPR pair, OR-ed on the scalar unit; a flagged wave (about one multiply in 10^4) recomputes the product exactly
//   0  fe_reduce16: v_max3 test of the high limbs up front, 32 VALU instructions (round 1)
#ifndef VK_REDUCE_FLAGS
#define VK_REDUCE_FLAGS 1
#endif
#ifndef VK_TIMING_ONLY_NO_RECOMPUTE
#define VK_TIMING_ONLY_NO_RECOMPUTE 0 /* 1 = experiment: flags ignored (wrong results one product in 10^4); never shipped */
#endif

// exact product / square for the flagged case: same columns and merge, the three-chain reduction that has no preconditions.
// Real calls (one copy per translation unit) so that the rare path costs the hot path a compare and a branch only.
__device__ __noinline__ fe fe_mul_exact(fe a, fe b)
{
    fe r;
    u64 acc[15];
    u32 top = fe_mul_cols(acc, a, b);
    u32 t[16];
    merge_eo(t, acc, top);
    fe_reduce16_slow(r, t);
    return r;
}
__device__ __noinline__ fe fe_sqr_exact(fe a)
{
    fe r;
    u64 acc[15];
    u32 top = fe_sqr_cols(acc, a);
    u32 od[16];
    merge_eo(od, acc, top);
    u32 t[16];
    fe_sqr_tail(t, od, a);
    fe_reduce16_slow(r, t);
    return r;
}

VK_DEV void fe_mul(fe &r, const fe &a, const fe &b)
{
#if VK_REDUCE_FLAGS
    const fe a0 = a, b0 = b; // r may alias an operand: the recomputation needs the values the product was taken of
    u64 acc[15];
    u32 top = fe_mul_cols(acc, a0, b0);
    u32 t[16];
    merge_eo(t, acc, top);
    const vk_carry lost = fe_reduce16_flags(r, t);
#if !VK_TIMING_ONLY_NO_RECOMPUTE
    if (__builtin_expect(lost != 0, 0)) r = fe_mul_exact(a0, b0);
#endif
#else
    u64 acc[15];
    u32 top = fe_mul_cols(acc, a, b);
    u32 t[16];
    merge_eo(t, acc, top);
    fe_reduce16(r, t);
#endif
}

// r = a^2: the 28 off-diagonal products in E/O columns, doubled while merging, plus the diagonal
VK_DEV void fe_sqr(fe &r, const fe &a)
{
#if VK_REDUCE_FLAGS
    const fe a0 = a;
    u64 acc[15];
    u32 top = fe_sqr_cols(acc, a0);
    u32 od[16];
    merge_eo(od, acc, top); // off-diagonal sum < 2^511
    u32 t[16];
    fe_sqr_tail(t, od, a0); // t = 2 * od + sum a_i^2 2^(64 i)
    const vk_carry lost = fe_reduce16_flags(r, t);
#if !VK_TIMING_ONLY_NO_RECOMPUTE
    if (__builtin_expect(lost != 0, 0)) r = fe_sqr_exact(a0);
#endif
#else
    u64 acc[15];
    u32 top = fe_sqr_cols(acc, a);
    u32 od[16];
    merge_eo(od, acc, top); // off-diagonal sum < 2^511
    u32 t[16];
    fe_sqr_tail(t, od, a); // t = 2 * od + sum a_i^2 2^(64 i)
    fe_reduce16(r, t);
#endif
}

VK_DEV void fe_sqr_n(fe &r, const fe &a, int n)
{
    r = a;
#pragma unroll 1
    for (int i = 0; i < n; i++) fe_sqr(r, r);
}

// canonical representative in [0, p)
VK_DEV void fe_canon(fe &r)
{
    // r >= p  <=>  r + C overflows 2^256
    u64 c = (u64)r.v[0] + VK_PC0;
    u32 s[8];
    s[0] = (u32)c;
    c >>= 32;
    c += (u64)r.v[1] + 1u;
    s[1] = (u32)c;
    c >>= 32;
#pragma unroll
    for (int i = 2; i < 8; i++) {
        c += r.v[i];
        s[i] = (u32)c;
        c >>= 32;
    }
    bool ge = c != 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = ge ? s[i] : r.v[i];
}

VK_DEV bool fe_is_zero_canon(const fe &a)
{
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}

// a^(p-2).  Chain over the run structure of p-2 = 1^223 0 1^22 0000 1 0 1 1 0 1.
// A real call (one copy per translation unit), operand and result BY VALUE: both travel in VGPRs.  With reference
// parameters the caller's field elements had their address taken, stayed in scratch (80 bytes per lane in the sweep) and were
// accessed as 64/128-bit register tuples inside the hot loop -- the shape in which LLVM's register coalescer dropped a lane
// of the running inverse in the "short fold" experiment (profiles/r02_short_fold_root_cause.txt).
__device__ __noinline__ fe fe_inv_val(fe a)
{
    fe r;
    fe x2, x3, x6, x9, x11, x22, x44, x88, x176, x220, x223, t;
    fe_sqr(x2, a);
    fe_mul(x2, x2, a);
    fe_sqr(x3, x2);
    fe_mul(x3, x3, a);
    fe_sqr_n(x6, x3, 3);
    fe_mul(x6, x6, x3);
    fe_sqr_n(x9, x6, 3);
    fe_mul(x9, x9, x3);
    fe_sqr_n(x11, x9, 2);
    fe_mul(x11, x11, x2);
    fe_sqr_n(x22, x11, 11);
    fe_mul(x22, x22, x11);
    fe_sqr_n(x44, x22, 22);
    fe_mul(x44, x44, x22);
    fe_sqr_n(x88, x44, 44);
    fe_mul(x88, x88, x44);
    fe_sqr_n(x176, x88, 88);
    fe_mul(x176, x176, x88);
    fe_sqr_n(x220, x176, 44);
    fe_mul(x220, x220, x44);
    fe_sqr_n(x223, x220, 3);
    fe_mul(x223, x223, x3);
    fe_sqr_n(t, x223, 23);
    fe_mul(t, t, x22);
    fe_sqr_n(t, t, 5);
    fe_mul(t, t, a);
    fe_sqr_n(t, t, 3);
    fe_mul(t, t, x2);
    fe_sqr_n(t, t, 2);
    fe_mul(r, t, a);
    return r;
}
VK_DEV void fe_inv(fe &r, const fe &a)
{
    const fe t = a;
    r = fe_inv_val(t);
}

// ---------------------------------------------------------------- scalars mod n
// n = FFFFFFFF FFFFFFFF FFFFFFFF FFFFFFFE BAAEDCE6 AF48A03B BFD25E8C D0364141
__device__ __constant__ const u32 VK_N[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u,
                                              0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};

// KeyDeriver::derive normalisation (src/derive.rs:79-93): k >= n -> k - n (2^256 < 2n), 0 -> 1.
VK_DEV void scalar_normalize(u32 k[8])
{
    const u32 n[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    u32 d[8];
    u64 bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 x = (u64)k[i] - n[i] - bw;
        d[i] = (u32)x;
        bw = (x >> 32) & 1;
    }
    bool ge = bw == 0; // no borrow: k >= n
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        k[i] = ge ? d[i] : k[i];
        o |= k[i];
    }
    if (o == 0) k[0] = 1;
}
