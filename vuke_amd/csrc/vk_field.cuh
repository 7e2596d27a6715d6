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
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

#define VK_DEV __device__ __forceinline__

struct fe {
    u32 v[8];
};

#define VK_PC0 977u /* C = 2^32 + 977 */

VK_DEV u64 mad64(u32 a, u32 b, u64 c) { return (u64)a * (u64)b + c; }

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

// r = a + b (mod p), weakly reduced
VK_DEV void fe_add(fe &r, const fe &a, const fe &b)
{
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)a.v[i] + b.v[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    // carry out of 2^256: add C; can carry once more only if the sum wrapped into [2^256-C, 2^256)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        u32 m = (u32)0 - (u32)c;
        c = (u64)r.v[0] + (VK_PC0 & m);
        r.v[0] = (u32)c;
        c >>= 32;
        c += (u64)r.v[1] + (1u & m);
        r.v[1] = (u32)c;
        c >>= 32;
#pragma unroll
        for (int i = 2; i < 8; i++) {
            c += r.v[i];
            r.v[i] = (u32)c;
            c >>= 32;
        }
    }
}

// r = a - b (mod p), weakly reduced
VK_DEV void fe_sub(fe &r, const fe &a, const fe &b)
{
    u64 bw = 0; // borrow as 0/1
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 d = (u64)a.v[i] - b.v[i] - bw;
        r.v[i] = (u32)d;
        bw = (d >> 32) & 1;
    }
    // borrow out: the lanes hold (a - b + 2^256); subtract C to land on a - b + p
#pragma unroll
    for (int k = 0; k < 2; k++) {
        u32 m = (u32)0 - (u32)bw;
        u64 d = (u64)r.v[0] - (VK_PC0 & m);
        r.v[0] = (u32)d;
        bw = (d >> 32) & 1;
        d = (u64)r.v[1] - (1u & m) - bw;
        r.v[1] = (u32)d;
        bw = (d >> 32) & 1;
#pragma unroll
        for (int i = 2; i < 8; i++) {
            d = (u64)r.v[i] - bw;
            r.v[i] = (u32)d;
            bw = (d >> 32) & 1;
        }
    }
}

// fold a 16-limb product into 8 limbs: t_lo + t_hi * (2^32 + 977)
VK_DEV void fe_reduce16(fe &r, const u32 t[16])
{
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c = mad64(t[8 + i], VK_PC0, c + t[i]);
        if (i > 0) c += t[8 + i - 1];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    u64 o = c + t[15];    // < 2^33: limb 8 of the intermediate
    u64 m = o * VK_PC0;   // < 2^43
    c = (u64)r.v[0] + (u32)m;
    r.v[0] = (u32)c;
    c >>= 32;
    c += (u64)r.v[1] + (m >> 32) + (u32)o;
    r.v[1] = (u32)c;
    c >>= 32;
    c += (u64)r.v[2] + (o >> 32);
    r.v[2] = (u32)c;
    c >>= 32;
#pragma unroll
    for (int i = 3; i < 8; i++) {
        c += r.v[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    // final wrap (value was 2^256 + small): add C, cannot carry again
    u32 msk = (u32)0 - (u32)c;
    c = (u64)r.v[0] + (VK_PC0 & msk);
    r.v[0] = (u32)c;
    c >>= 32;
    c += (u64)r.v[1] + (1u & msk);
    r.v[1] = (u32)c;
    c >>= 32;
    c += r.v[2];
    r.v[2] = (u32)c;
    c >>= 32;
    r.v[3] += (u32)c;
}

VK_DEV void fe_mul(fe &r, const fe &a, const fe &b)
{
    u32 t[16];
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        c = mad64(a.v[0], b.v[j], c);
        t[j] = (u32)c;
        c >>= 32;
    }
    t[8] = (u32)c;
#pragma unroll
    for (int i = 1; i < 8; i++) {
        c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c = mad64(a.v[i], b.v[j], c + t[i + j]);
            t[i + j] = (u32)c;
            c >>= 32;
        }
        t[i + 8] = (u32)c;
    }
    fe_reduce16(r, t);
}

// r = a^2: off-diagonal products once, doubled, plus the diagonal (36 wide multiplies)
VK_DEV void fe_sqr(fe &r, const fe &a)
{
    u32 t[16];
    u64 c;
    // off-diagonal: sum_{i<j} a_i a_j 2^(32(i+j))
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        c = 0;
#pragma unroll
        for (int j = i + 1; j < 8; j++) {
            c = mad64(a.v[i], a.v[j], c + t[i + j]);
            t[i + j] = (u32)c;
            c >>= 32;
        }
        t[i + 8] = (u32)c;
    }
    // double
    u32 top = 0;
#pragma unroll
    for (int i = 1; i < 16; i++) {
        u32 nt = t[i] >> 31;
        t[i] = (t[i] << 1) | top;
        top = nt;
    }
    // add the diagonal a_i^2 at limb 2i
    c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 sq = (u64)a.v[i] * a.v[i];
        c += (u64)t[2 * i] + (u32)sq;
        t[2 * i] = (u32)c;
        c >>= 32;
        c += (u64)t[2 * i + 1] + (sq >> 32);
        t[2 * i + 1] = (u32)c;
        c >>= 32;
    }
    fe_reduce16(r, t);
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

// r = a^(p-2).  Chain over the run structure of p-2 = 1^223 0 1^22 0000 1 0 1 1 0 1.
__device__ __noinline__ void fe_inv(fe &r, const fe &a)
{
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
