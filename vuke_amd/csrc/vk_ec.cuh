// vk_ec.cuh -- secp256k1 group operations for gfx950 lanes.
//
// Fixed-base k*G with a precomputed table of (d * 2^off_w) * G affine points that lives in HBM (vk_win_desc below:
// uniform W-bit windows -- library default W = 24, 11 windows, 11.8 GB -- or signed-digit windows: the 9 of the bench
// geometry, 4 x 29 + 5 x 28 bits, 120 GB of the 288 GB part, or 10 / 11 narrower ones at 18 / 4 GB).  One lane owns one
// key: one 64-byte table gather per window and windows - 1 point additions, no doublings.  The sweep uses
// ecmult_table_lds_xyzz (XYZZ accumulator, entries fetched by lane quads straight into LDS); ecmult_table is the Jacobian
// form with per-lane loads kept for the small kernels.
//
// Why no special cases are needed in the adders (P = +-Q never occurs): k is normalised to [1, n).  After windows 0..w-1 the
// accumulator is s*G and window w adds T*G with T = t * 2^off_w, t != 0, where k's bits below off_w + bits_w equal s + T as
// INTEGERS.  Unsigned windows: 0 <= s < 2^off_w <= T, so s != T, and s + T <= k < n, so s != -T (mod n).  Signed windows: a
// digit above half its range is replaced by its negative and a carry, so |s| <= 2^(off_w - 1) + 2^(off_w - bits_{w-1}) < 2^off_w
// <= |T| and the integers differ; modulo n they could only meet as s = T - n (or -T + n), i.e. n - |T| = |s| < 2^off_w, which
// needs |T| > n - 2^off_w: only the last (unsigned) window reaches that far, with T = 2^256 (then k = s + T > n: not a
// normalised key) or T = 2^256 - 2^off_w, where n - T = 2^off_w - (2^256 - n) > 2^(off_w - 1) + 2^(off_w - bits) because
// 2^256 - n < 2^129 and off_w >= 200 in every layout the engine accepts (window_bits <= 29).
//
// Stands in for libsecp256k1's ecmult_gen reached from src/derive.rs:97 (reference).
#pragma once
#include "vk_field.cuh"

struct jac {
    fe x, y, z;
};

// r = p + (qx, qy), p Jacobian (not infinity), q affine, p != +-q.   8M + 3S
VK_DEV void jac_madd(jac &r, const jac &p, const fe &qx, const fe &qy)
{
    fe z2, u2, s2, h, rr, h2, h3, v, t;
    fe_sqr(z2, p.z);
    fe_mul(u2, qx, z2);
    fe_mul(s2, qy, z2);
    fe_mul(s2, s2, p.z);
    fe_sub(h, u2, p.x);
    fe_sub(rr, s2, p.y);
    fe_sqr(h2, h);
    fe_mul(h3, h2, h);
    fe_mul(v, p.x, h2);
    fe_mul(r.z, p.z, h); // last use of p.z
    fe_sqr(t, rr);
    fe_sub3(t, t, h3, v);     // x3
    fe_mul(h3, h3, p.y); // y1 * h^3 (last use of p.y)
    fe_sub(v, v, t);
    fe_mul(v, v, rr);
    fe_sub(r.y, v, h3);
    r.x = t;
}

// r = p + q for two affine points with different x: the general formula with z1 = 1.   4M + 2S
// (the second window of every k*G: the accumulator is still the first table entry)
VK_DEV void jac_aadd(jac &r, const fe &px, const fe &py, const fe &qx, const fe &qy)
{
    fe h, rr, h2, h3, v, t;
    fe_sub(h, qx, px);
    fe_sub(rr, qy, py);
    fe_sqr(h2, h);
    fe_mul(h3, h2, h);
    fe_mul(v, px, h2);
    fe_sqr(t, rr);
    fe_sub3(t, t, h3, v); // x3
    fe_mul(h3, h3, py);
    fe_sub(v, v, t);
    fe_mul(v, v, rr);
    fe_sub(r.y, v, h3);
    r.x = t;
    r.z = h;
}

// r = 2p (a = 0), p.y != 0.   2M + 5S  (used only by the table builder)
VK_DEV void jac_double(jac &r, const jac &p)
{
    fe a, b, c, d, e, f, t;
    fe_sqr(a, p.x);
    fe_sqr(b, p.y);
    fe_sqr(c, b);
    fe_add(t, p.x, b);
    fe_sqr(t, t);
    fe_sub(t, t, a);
    fe_sub(t, t, c);
    fe_add(d, t, t);
    fe_add(e, a, a);
    fe_add(e, e, a);
    fe_sqr(f, e);
    fe_mul(r.z, p.y, p.z);
    fe_add(r.z, r.z, r.z);
    fe_sub(f, f, d);
    fe_sub(f, f, d); // x3
    fe_sub(t, d, f);
    fe_mul(t, e, t);
    fe_add(c, c, c);
    fe_add(c, c, c);
    fe_add(c, c, c);
    fe_sub(r.y, t, c);
    r.x = f;
}

// table entry = 16 u32: x[8] then y[8], canonical little-endian limbs
VK_DEV void table_load(fe &x, fe &y, const uint4 *__restrict__ p)
{
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w;
    x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w;
    y.v[0] = c.x; y.v[1] = c.y; y.v[2] = c.z; y.v[3] = c.w;
    y.v[4] = d.x; y.v[5] = d.y; y.v[6] = d.z; y.v[7] = d.w;
}

// ---- window layout -------------------------------------------------------------------------
// Window i covers scalar bits [off, off + bits).  Unsigned windows use the digit as it is; in a
// signed layout a digit above 2^(bits-1) is replaced by its negative (table entry 2^bits - d, y
// negated) and a carry of one into the next window, so a window needs only 2^(bits-1) + 1 entries.
// The last window is always unsigned (it absorbs the final carry; its entry 2^bits, i.e. the
// scalar 2^256, is stored as (2^256 mod n) * G).  Entry d of window i lives at table[base + d].
#define VK_MAX_WIN 64
struct vk_win_desc {
    u64 base;
    u32 off, bits, sgn, tb; // tb = index bits of the window's table = bits - sgn
};

VK_DEV u32 scalar_digit(const u32 k[8], u32 bitpos, u32 W)
{
    // W <= 31 bits starting at bitpos (may straddle two limbs).  The window geometry is the same for every lane, so the
    // limb pair is picked by a scalar branch (two register copies) instead of a chain of sixteen per-lane selects.
    const u32 limb = __builtin_amdgcn_readfirstlane(bitpos >> 5), sh = __builtin_amdgcn_readfirstlane(bitpos & 31);
    u32 lo, hi;
    switch (limb) {
    case 0: lo = k[0]; hi = k[1]; break;
    case 1: lo = k[1]; hi = k[2]; break;
    case 2: lo = k[2]; hi = k[3]; break;
    case 3: lo = k[3]; hi = k[4]; break;
    case 4: lo = k[4]; hi = k[5]; break;
    case 5: lo = k[5]; hi = k[6]; break;
    case 6: lo = k[6]; hi = k[7]; break;
    case 7: lo = k[7]; hi = 0; break;
    default: lo = 0; hi = 0; break;
    }
    return __builtin_amdgcn_alignbit(hi, lo, sh) & ((1u << W) - 1u);
}

// The scalar parked in LDS ([limb][lane] u32 columns of the wave, VK_PARK): the window geometry is wave-uniform, so the limb pair is two
// ds_read_b32 at a uniform offset; eight registers less across the whole window loop of the three-wave kernels.
struct vk_kpark {
    const u32 *col; // this lane's column, stride 64
};
VK_DEV u32 scalar_digit(const vk_kpark &k, u32 bitpos, u32 W)
{
    const u32 limb = __builtin_amdgcn_readfirstlane(bitpos >> 5), sh = __builtin_amdgcn_readfirstlane(bitpos & 31);
    const u32 lo = k.col[limb * 64], hi = limb < 7 ? k.col[(limb + 1) * 64] : 0u;
    return __builtin_amdgcn_alignbit(hi, lo, sh) & ((1u << W) - 1u);
}

// digit of window wd (magnitude), updating the running carry; neg = take the point's negative
template <typename K>
VK_DEV u32 window_digit(const K &k, const vk_win_desc &wd, u32 &carry, bool &neg)
{
    u32 raw = scalar_digit(k, wd.off, wd.bits) + carry;
    const u32 half = 1u << (wd.bits - 1);
    neg = (wd.sgn != 0) && (raw > half);
    carry = neg ? 1u : 0u;
    return neg ? (1u << wd.bits) - raw : raw;
}

// y = p - y for lanes with neg (y canonical and non-zero for table points).
// p - y = (2^256 - 1 - y) - (C - 1) with C = 2^256 - p = 2^32 + 977: the complement is a lane-masked XOR (plain VOP2, the fast
// issue class), the subtraction of C - 1 = 0x1000003D0 touches limbs 0 and 1 and borrows beyond them only when the low 64 bits
// of ~y are below C - 1 (one table point in 2^31: a divergent ripple).  ~y >= C because y < p, so the result is p - y itself,
// canonical.  (Until round 4 this was a 256-bit subtraction p - y + eight selects: 40 instructions per window as compiled;
// with the leaner address arithmetic of lds_gather_issue: -2 % instructions per key, +0.6 % keys/s same box, profiles/r05_lean_glue_experiment.txt.)
VK_DEV void fe_cond_negate(fe &y, bool neg)
{
    const u32 m = neg ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) y.v[i] ^= m;
    const u32 s0 = m & 0x3D0u, s1 = m & 1u;
    const u32 l0 = y.v[0], l1 = y.v[1];
    y.v[0] = l0 - s0;
    const u32 b0 = l0 < s0 ? 1u : 0u;
    y.v[1] = l1 - s1 - b0;
    if (l1 < s1 + b0) { // s1 + b0 <= 2: no wrap
        u32 bw = 1u;
#pragma unroll
        for (int i = 2; i < 8; i++) {
            const u32 t = y.v[i];
            y.v[i] = t - bw;
            bw = t < bw ? 1u : 0u;
        }
    }
}

// ---- quad-cooperative table reads landing in LDS directly (global_load_lds_dwordx4) ----------------------------------
// A lane reading its own 64-byte entry with four 16-byte loads makes every load instruction touch 64 different lines;
// the per-CU L1 miss path then sustains only ~19 G entries/s chip-wide (tools/gather_bench.hip).  Here the four lanes of
// a quad read one whole entry per instruction (16 lines per instruction, ~50 G entries/s), and the load unit hands the
// quarters to their owner through the wave's LDS slot.  Every lane of the wave must take part (no divergence around
// these calls).  (The per-lane and the register-staged forms of the sweep's table read, and its Jacobian accumulator,
// were A/B variants until round 3: profiles/archive/r01_table_read_experiments.txt; last in the sources at commit 4b24351.)
// No staging registers: the load unit writes lane i's 16 bytes to slot + 16 i.  Round r of a whole-entry read fetches,
// in lane L, quarter L & 3 of the entry owned by quad lane r; the owner then finds its 64 bytes contiguous in round
// (lane & 3) at lanes (lane & ~3) .. +3.  The x-only read does the same with lane pairs and halves of x.  A round is 65
// uint4 (64 lanes + one of padding, so that the owners' 16-byte reads of sixteen consecutive lanes hit distinct banks).
// slot must be wave-uniform; every lane of the wave takes part.  The data is in LDS once the wave's vmcnt has drained
// past the reads (the compiler tracks that); before re-using a slot the previous ds_reads must have returned
// (lds_gather_reuse_fence).
#define VK_LTILE_ROUND 65
#define VK_LTILE_FULL (4 * VK_LTILE_ROUND) /* uint4 per whole-entry slot */
typedef const void __attribute__((address_space(1))) *vk_gptr;
typedef void __attribute__((address_space(3))) *vk_lptr;
VK_DEV void lds_gather_reuse_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Address of round r = table + 64 * (entry of quad lane r) + 16 * (lane & 3): one DPP move (all lanes are written, so no old value
// has to be prepared) and one v_mad_u64_u32 with the lane's loop-invariant table + 16 * (lane & 3) as the addend -- as compiled from
// C++ it was a zeroing move, the DPP move, a 64-bit shift and a 64-bit add per round.  `slot` must be wave-uniform IN AN SGPR (the
// kernel derives it from readfirstlane(threadIdx.x >> 6)): M0 then comes from the scalar unit instead of four v_readfirstlane.
VK_DEV u64 table_entry_addr(u32 e, u64 tq)
{
    u64 a;
    asm("v_mad_u64_u32 %0, vcc, %1, 64, %2" : "=v"(a) : "v"(e), "v"(tq) : "vcc");
    return a;
}
VK_DEV void lds_gather_issue(uint4 *slot, const uint4 *__restrict__ table, u32 entry)
{
    const u64 tq = (u64)(table + (threadIdx.x & 3u));
    const u32 e0 = __builtin_amdgcn_mov_dpp(entry, 0x00, 0xF, 0xF, true);
    const u32 e1 = __builtin_amdgcn_mov_dpp(entry, 0x55, 0xF, 0xF, true);
    const u32 e2 = __builtin_amdgcn_mov_dpp(entry, 0xAA, 0xF, 0xF, true);
    const u32 e3 = __builtin_amdgcn_mov_dpp(entry, 0xFF, 0xF, 0xF, true);
    __builtin_amdgcn_global_load_lds((vk_gptr)table_entry_addr(e0, tq), (vk_lptr)(slot + 0 * VK_LTILE_ROUND), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((vk_gptr)table_entry_addr(e1, tq), (vk_lptr)(slot + 1 * VK_LTILE_ROUND), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((vk_gptr)table_entry_addr(e2, tq), (vk_lptr)(slot + 2 * VK_LTILE_ROUND), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((vk_gptr)table_entry_addr(e3, tq), (vk_lptr)(slot + 3 * VK_LTILE_ROUND), 16, 0, 0);
}
VK_DEV void lds_gather_read(fe &x, fe &y, const uint4 *slot)
{
    // the loads that fill the slot count in vmcnt like any other; the compiler tracks LDS-direct loads and places this
    // wait itself in the shipped kernels, but a read of stale LDS is silent, so it is spelled out
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u32 lane = threadIdx.x & 63u;
    const uint4 *m = slot + (lane & 3u) * VK_LTILE_ROUND + (lane & ~3u);
    const uint4 a = m[0], b = m[1], c = m[2], d = m[3];
    x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w;
    x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w;
    y.v[0] = c.x; y.v[1] = c.y; y.v[2] = c.z; y.v[3] = c.w;
    y.v[4] = d.x; y.v[5] = d.y; y.v[6] = d.z; y.v[7] = d.w;
}

// acc = k*G, k in [1, n).  Returns Jacobian (z = 1 possible).
// No special cases are needed in the adder: after windows 0..i-1 the accumulator is s*G with
// |s| < 2^off_i (s != 0 unless all digits so far were zero, in which case it is still "infinity"),
// the addend is t*G with |t| >= 2^off_i, so s != +-t, and |s + t| < n.
VK_DEV void ecmult_table(jac &acc, const u32 k[8], const uint4 *__restrict__ table, const vk_win_desc *__restrict__ wins, u32 nwin)
{
    bool inf = true, zone = false; // zone: the accumulator is a table entry (z = 1)
    fe_set_zero(acc.x);
    fe_set_zero(acc.y);
    fe_set_one(acc.z);
    // software prefetch of the next window's entry
    u32 carry = 0;
    bool neg;
    u32 d = window_digit(k, wins[0], carry, neg);
    fe qx, qy;
    table_load(qx, qy, table + ((wins[0].base + d) << 2));
#pragma unroll 1
    for (u32 w = 0; w < nwin; w++) {
        fe cx = qx, cy = qy;
        const u32 cd = d;
        const bool cneg = neg;
        if (w + 1 < nwin) {
            d = window_digit(k, wins[w + 1], carry, neg);
            table_load(qx, qy, table + ((wins[w + 1].base + d) << 2));
        }
        if (cd != 0) {
            fe_cond_negate(cy, cneg);
            if (inf) {
                acc.x = cx;
                acc.y = cy;
                inf = false;
                zone = true;
            } else if (zone) {
                jac_aadd(acc, acc.x, acc.y, cx, cy);
                zone = false;
            } else {
                jac_madd(acc, acc, cx, cy);
            }
        }
    }
}

// generic double-and-add over all 256 bits of s (MSB first), s in [1, n).  Table builder only.
VK_DEV void ecmult_generic(jac &acc, const u32 s[8])
{
    const fe gx = {{0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu, 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu}};
    const fe gy = {{0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u, 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u}};
    bool inf = true;
#pragma unroll 1
    for (int i = 255; i >= 0; i--) {
        if (!inf) jac_double(acc, acc);
        u32 limb = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) limb = ((i >> 5) == j) ? s[j] : limb;
        if ((limb >> (i & 31)) & 1u) {
            if (inf) {
                acc.x = gx;
                acc.y = gy;
                fe_set_one(acc.z);
                inf = false;
            } else {
                jac_madd(acc, acc, gx, gy);
            }
        }
    }
}

// (x, y) = affine(p) given zinv = 1/p.z; canonical limbs
VK_DEV void jac_to_affine(fe &x, fe &y, const jac &p, const fe &zinv)
{
    fe zi2, zi3;
    fe_sqr(zi2, zinv);
    fe_mul(zi3, zi2, zinv);
    fe_mul(x, p.x, zi2);
    fe_mul(y, p.y, zi3);
    fe_canon(x);
    fe_canon(y);
}

// ---- the sweep's accumulator in XYZZ coordinates ---------------------------------------------------------------------
// (X, Y, ZZ, ZZZ) with x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2.  Adding an affine point costs 8M + 2S instead of the
// 8M + 3S of the Jacobian mixed addition (no Z^2 to recompute: "madd-2008-s").  Leaving the chain costs the same as
// before: ZZZ goes through the batch inversion, y = Y / ZZZ, x = X * ZZ^2 / ZZZ^2 (ZZ^3 = ZZZ^2) -- 2S + 3M per key like
// the Jacobian 1S + 3M plus one squaring -- so a k*G with seven such additions saves about 6S: 4 % of the sweep.
// Round 5: the additions return the negative of the sum (xyzz_madd_flip), whose y is ONE fused pair of products.
struct xyzz {
    fe x, y, zz, zzz;
};
// r = -(r + (qx, qy)); r not infinity, q affine, r != +-q (same argument as jac_madd).   7M + 2S + one fused pair of products
// The NEGATIVE of the sum, because its y-coordinate is a sum of two products where the sum's own is a difference:
//   Y3 = R (Q - X3) - Y1 P^3      -(Y3) = R (X3 - Q) + Y1 P^3
// and a sum of two products is one trip through the multiplier's columns, one merge and one reduction (fe_mul2: 291 instructions,
// 119 multiply-adds) instead of two multiplications and a field subtraction (324 and 146).  The caller keeps track of the sign the
// accumulator stands for and hands in the table point with that sign (ecmult_table_lds_xyzz below): -(s A + s q) = -s (A + q).
VK_DEV void xyzz_madd_flip(xyzz &r, const fe &qx, const fe &qy)
{
    fe u2, s2, pp, ppp, q, t;
    fe_mul(u2, qx, r.zz);
    fe_mul(s2, qy, r.zzz);
    fe_sub(u2, u2, r.x);       // P
    fe_sub(s2, s2, r.y);       // R
    fe_sqr(pp, u2);
    fe_mul(ppp, pp, u2);
    fe_mul(q, r.x, pp);        // Q
    fe_mul(r.zz, r.zz, pp);
    fe_mul(r.zzz, r.zzz, ppp);
    fe_sqr(t, s2);
    fe_sub3(t, t, ppp, q);     // X3
    fe_sub(q, t, q);           // X3 - Q
    fe_mul2(r.y, s2, q, r.y, ppp); // -(Y3)
    r.x = t;
}
// r = -((r.x, r.y) + (qx, qy)) for two affine points with different x (r.zz = r.zzz = 1 on entry).   3M + 2S + one fused pair
VK_DEV void xyzz_aadd_flip(xyzz &r, const fe &qx, const fe &qy)
{
    fe h, rr, q, t;
    fe_sub(h, qx, r.x);
    fe_sub(rr, qy, r.y);
    fe_sqr(r.zz, h);
    fe_mul(r.zzz, r.zz, h);
    fe_mul(q, r.x, r.zz);
    fe_sqr(t, rr);
    fe_sub3(t, t, r.zzz, q);   // x3
    fe_sub(q, t, q);
    fe_mul2(r.y, rr, q, r.y, r.zzz);
    r.x = t;
}
// ecmult_table with the reads landing in LDS directly and the XYZZ accumulator (the whole wave calls it; lanes without a key
// pass k = 1).  VK_LDS_DEPTH reads are in flight per lane: slots = this wave's VK_LDS_DEPTH x VK_LTILE_FULL uint4, used as a ring.
#ifndef VK_LDS_DEPTH
#define VK_LDS_DEPTH 1 /* power of two; 2 and 4 measured 0.4 % slower than 1: the latency is covered, the gain is the cheaper miss path */
#endif
template <typename K>
VK_DEV void ecmult_table_lds_xyzz(xyzz &acc, const K &k, const uint4 *__restrict__ table, const vk_win_desc *__restrict__ wins, u32 nwin, uint4 *slots)
{
    bool inf = true, zone = false;
    // Every addition leaves the NEGATIVE of the sum in the accumulator (xyzz_madd_flip): `minus` says that the accumulator stands for
    // minus the point accumulated so far; a table point goes in with the accumulator's sign.  The first entry is loaded with the
    // sign that nwin - 1 additions turn into plus, so a key whose digits are all non-zero (all but one key in 2^25) ends with
    // minus == false; the others pay the negation at the end.
    bool minus = ((nwin - 1u) & 1u) != 0;
    fe_set_zero(acc.x);
    fe_set_zero(acc.y);
    fe_set_one(acc.zz);
    fe_set_one(acc.zzz);
    u32 carry = 0;
    u32 dq[VK_LDS_DEPTH];
    bool nq[VK_LDS_DEPTH];
    lds_gather_reuse_fence();
#pragma unroll
    for (u32 i = 0; i < VK_LDS_DEPTH; i++) {
        dq[i] = window_digit(k, wins[i], carry, nq[i]);
        lds_gather_issue(slots + i * VK_LTILE_FULL, table, (u32)(wins[i].base + dq[i]));
    }
#pragma unroll 1
    for (u32 w = 0; w < nwin; w++) {
        uint4 *slot = slots + (w & (VK_LDS_DEPTH - 1)) * VK_LTILE_FULL;
        fe cx, cy;
        lds_gather_read(cx, cy, slot);
        const u32 cd = dq[0];
        const bool cneg = nq[0];
#pragma unroll
        for (u32 i = 0; i + 1 < VK_LDS_DEPTH; i++) {
            dq[i] = dq[i + 1];
            nq[i] = nq[i + 1];
        }
        if (w + VK_LDS_DEPTH < nwin) {
            dq[VK_LDS_DEPTH - 1] = window_digit(k, wins[w + VK_LDS_DEPTH], carry, nq[VK_LDS_DEPTH - 1]);
            lds_gather_reuse_fence();
            lds_gather_issue(slot, table, (u32)(wins[w + VK_LDS_DEPTH].base + dq[VK_LDS_DEPTH - 1]));
        }
        if (cd != 0) {
            fe_cond_negate(cy, cneg != minus);
            if (inf) {
                acc.x = cx;
                acc.y = cy;
                inf = false;
                zone = true;
            } else if (zone) {
                xyzz_aadd_flip(acc, cx, cy);
                zone = false;
                minus = !minus;
            } else {
                xyzz_madd_flip(acc, cx, cy);
                minus = !minus;
            }
        }
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(minus) != 0, 0)) { // a zero digit somewhere (or a lane without a key: k = 1)
        fe ny;
        fe_set_zero(ny);
        fe_sub(ny, ny, acc.y);
#pragma unroll
        for (int i = 0; i < 8; i++) acc.y.v[i] = minus ? ny.v[i] : acc.y.v[i];
    }
}
