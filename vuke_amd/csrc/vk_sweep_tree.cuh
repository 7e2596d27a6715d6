// vk_sweep_tree.cuh -- the sweep with the first tree level of the table sum done in affine coordinates.
//
// k*G = T0[d0] + ... + T8[d8] (nine table points).  vk_sweep_kernel adds them as one Jacobian chain: 6 + 7 x 11 field
// multiplications.  Here the four pairs (T0+T1), (T2+T3), (T4+T5), (T6+T7) are affine + affine additions whose four
// denominators per key go through a batch inversion shared by all keys of the lane (6 multiplications per addition instead
// of 11), and only the chain P3 + P2 + P1 + P0 + T8 stays Jacobian: 70 field multiplications per key instead of 90.
// The price is memory: the eight entries are read twice (once for the denominators, once for the additions), every read a
// quad-cooperative gather (vk_ec.cuh), plus 11 quads of workspace per key for the digits and the prefix products.
//
//   pass A  forward over the B items of the lane: key -> digits -> x-difference of every pair -> running product,
//           digits and running products parked;
//   inv A   one field inversion;
//   pass B  backward: per pair the inverse of its x-difference falls out of the running products, the pair is added in
//           affine form and goes straight into the Jacobian chain; (X, Y, Z) and the running product of Z parked;
//   inv B   one field inversion;
//   pass C  forward: affine x, y -> hash160 -> probe / records, exactly as pass 2 of vk_sweep_kernel.
//
// Rare shapes are exact, not approximated: a zero digit makes its table point absent (the pair is then the other point,
// or nothing), lanes without a key carry all-zero digits; a pair never meets x1 = x2 (different windows).
#pragma once
#include "vk_kernels.cuh"

#define VK_TREE_QUADS 19 /* workspace quads per item: 3 digits + 8 prefix products (pass A), 8 X/Y/Z/c (pass B) */

// acc += q for an affine q (the state machine of ecmult_table)
VK_DEV void chain_add(jac &acc, bool &inf, bool &zone, const fe &qx, const fe &qy)
{
    if (inf) {
        acc.x = qx;
        acc.y = qy;
        fe_set_one(acc.z);
        inf = false;
        zone = true;
    } else if (zone) {
        jac_aadd(acc, acc.x, acc.y, qx, qy);
        zone = false;
    } else {
        jac_madd(acc, acc, qx, qy);
    }
}

template <u32 KIND>
__global__ void __launch_bounds__(VK_BLOCK, VK_SWEEP_WAVES) vk_sweep_tree_kernel(const vk_sweep_params p)
{
    const u32 tid = threadIdx.x;
    const u32 B = p.R * p.slots;
    u64 n_keys = 0, n_match = 0;
    u64 dg[4] = {0, 0, 0, 0};
    const u64 nth = VK_BLOCK;
    uint4 *wk = p.work + (u64)blockIdx.x * ((u64)B * VK_TREE_QUADS * VK_BLOCK) + tid;
    __shared__ u64 s_tile;
    __shared__ uint4 s_gather[VK_BLOCK * VK_GTILE_STRIDE];
    uint4 *gtile = s_gather + (tid & ~63u) * VK_GTILE_STRIDE;

#pragma unroll 1
    for (;;) {
        if (tid == 0) s_tile = atomicAdd((unsigned long long *)(p.stats + 7), 1ull);
        __syncthreads();
        const u64 tile = s_tile;
        __syncthreads();
        if (tile >= p.ntiles) break;
        const u64 tile_off = tile * ((u64)VK_BLOCK * p.R);

        // ------------------------------------------------------------------------------------------------ pass A
        // Software pipeline over the items: the eight reads of item b are issued as soon as its digits are known and are
        // only consumed after the key and digits of item b + 1 have been computed (the MT19937 / hash work of the next
        // key is what covers the read latency; the per-pair arithmetic here is far too short for that).
        fe c;
        fe_set_one(c);
        u32 r = 0, sub = 0;
        u32 e[9], en[9];
        quad_regs g0, g1, g2, g3, g4, g5, g6, g7;
        auto digits_of = [&](u32 out[9], u32 rr, u32 ss) -> bool {
            const u64 off = tile_off + (u64)rr * VK_BLOCK + tid;
            u32 kw[8], k[8];
            const bool have = item_key<KIND>(kw, p, off, ss);
#pragma unroll
            for (int i = 0; i < 8; i++) k[i] = have ? kw[7 - i] : (i == 0 ? 1u : 0u);
            scalar_normalize(k);
#pragma unroll
            for (int i = 0; i < 9; i++) out[i] = 0;
            u32 carry = 0;
#pragma unroll 1
            for (u32 w = 0; w < 9; w++) {
                bool neg;
                const u32 d = window_digit(k, p.wins[w], carry, neg);
                const u32 v = (d != 0 && have) ? (((u32)p.wins[w].base + d) | (neg ? 0x80000000u : 0u)) : 0u;
#pragma unroll
                for (int i = 0; i < 9; i++) out[i] = ((u32)i == w) ? v : out[i];
            }
            return have;
        };
        bool have = digits_of(e, r, sub);
        quad_gather_issue(g0, p.table, e[0] & 0x7FFFFFFFu);
        quad_gather_issue(g1, p.table, e[1] & 0x7FFFFFFFu);
        quad_gather_issue(g2, p.table, e[2] & 0x7FFFFFFFu);
        quad_gather_issue(g3, p.table, e[3] & 0x7FFFFFFFu);
        quad_gather_issue(g4, p.table, e[4] & 0x7FFFFFFFu);
        quad_gather_issue(g5, p.table, e[5] & 0x7FFFFFFFu);
        quad_gather_issue(g6, p.table, e[6] & 0x7FFFFFFFu);
        quad_gather_issue(g7, p.table, e[7] & 0x7FFFFFFFu);
#pragma unroll 1
        for (u32 b = 0; b < B; b++) {
            uint4 *it = wk + (u64)b * VK_TREE_QUADS * nth;
            if (++sub == p.slots) {
                sub = 0;
                r++;
            }
            bool have_next = false;
            if (b + 1 < B) have_next = digits_of(en, r, sub);
            it[0] = make_uint4(e[0], e[1], e[2], e[3]);
            it[nth] = make_uint4(e[4], e[5], e[6], e[7]);
            it[2 * nth] = make_uint4(e[8], have ? 1u : 0u, 0u, 0u);
            fe xa, ya, xb, yb, dx;
#define VK_TREE_PAIR(J, GA, GB)                                   \
            quad_gather_commit(xa, ya, GA, gtile);                 \
            quad_gather_commit(xb, yb, GB, gtile);                 \
            if (e[2 * J] != 0 && e[2 * J + 1] != 0) {              \
                fe_sub(dx, xb, xa);                                \
                fe_mul(c, c, dx);                                  \
            }                                                      \
            work_store(it + (3 + 2 * J) * nth, nth, c);
            VK_TREE_PAIR(0, g0, g1)
            VK_TREE_PAIR(1, g2, g3)
            VK_TREE_PAIR(2, g4, g5)
            VK_TREE_PAIR(3, g6, g7)
#undef VK_TREE_PAIR
            if (b + 1 < B) {
#pragma unroll
                for (int i = 0; i < 9; i++) e[i] = en[i];
                have = have_next;
                quad_gather_issue(g0, p.table, e[0] & 0x7FFFFFFFu);
                quad_gather_issue(g1, p.table, e[1] & 0x7FFFFFFFu);
                quad_gather_issue(g2, p.table, e[2] & 0x7FFFFFFFu);
                quad_gather_issue(g3, p.table, e[3] & 0x7FFFFFFFu);
                quad_gather_issue(g4, p.table, e[4] & 0x7FFFFFFFu);
                quad_gather_issue(g5, p.table, e[5] & 0x7FFFFFFFu);
                quad_gather_issue(g6, p.table, e[6] & 0x7FFFFFFFu);
                quad_gather_issue(g7, p.table, e[7] & 0x7FFFFFFFu);
            }
        }
        // ------------------------------------------------------------------------------------------------ inv A
        fe inv;
        fe_inv(inv, c);
        // ------------------------------------------------------------------------------------------------ pass B
        fe cz; // running product of Z, from the last item down
        fe_set_one(cz);
#pragma unroll 1
        for (u32 bb = B; bb > 0; bb--) {
            const u32 b = bb - 1;
            uint4 *it = wk + (u64)b * VK_TREE_QUADS * nth;
            const uint4 d0 = it[0], d1 = it[nth], d2 = it[2 * nth];
            const u32 e[9] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w, d2.x};
            jac acc;
            bool inf = true, zone = false;
            fe_set_zero(acc.x);
            fe_set_zero(acc.y);
            fe_set_one(acc.z);
            // five steps: the pairs (6,7), (4,5), (2,3), (0,1), then the ninth point alone.  One loop body, so the
            // Jacobian adders are instantiated once (four inlined copies did not fit the instruction cache).
            quad_regs g1, g2;
            quad_gather_issue(g1, p.table, e[6] & 0x7FFFFFFFu);
            quad_gather_issue(g2, p.table, e[7] & 0x7FFFFFFFu);
#pragma unroll 1
            for (u32 step = 0; step < 5; step++) {
                const u32 j = 3 - step; // pair index for step < 4
                // this step's entries, the next step's entries (step 3 -> the single ninth point, read into g1)
                u32 ea = 0, eb = 0, na = 0, nb = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    ea = ((u32)i == j) ? e[2 * i] : ea;
                    eb = ((u32)i == j) ? e[2 * i + 1] : eb;
                    na = ((u32)i + 1 == j) ? e[2 * i] : na;
                    nb = ((u32)i + 1 == j) ? e[2 * i + 1] : nb;
                }
                if (step == 3) na = e[8];
                if (step == 4) {
                    ea = e[8];
                    eb = 0;
                }
                // running product before this pair: slot j-1 of this item, or slot 3 of the item before, or 1.  Loaded
                // first: loads return in order, so a load issued after the next step's reads would wait for all of them.
                fe cp;
                work_load(cp, (step < 3) ? it + (3 + 2 * (j - 1)) * nth : (b > 0 ? it - VK_TREE_QUADS * nth + 9 * nth : it + 3 * nth), nth);
                fe x1, y1, x2, y2;
                quad_gather_commit(x1, y1, g1, gtile);
                if (step < 4) {
                    quad_gather_commit(x2, y2, g2, gtile);
                } else {
                    fe_set_zero(x2);
                    fe_set_zero(y2);
                }
                if (step < 4) quad_gather_issue(g1, p.table, na & 0x7FFFFFFFu);
                if (step < 3) quad_gather_issue(g2, p.table, nb & 0x7FFFFFFFu);
                fe_cond_negate(y1, (ea >> 31) != 0);
                fe_cond_negate(y2, (eb >> 31) != 0);
                bool present = ea != 0; // x1, y1 is the point to add unless both are there
                if (ea != 0 && eb != 0) {
                    fe dx, dxi, lam, t;
                    if (step == 3 && b == 0) fe_set_one(cp);
                    fe_sub(dx, x2, x1);
                    fe_mul(dxi, inv, cp);  // 1 / dx
                    fe_mul(inv, inv, dx);
                    fe_sub(t, y2, y1);
                    fe_mul(lam, t, dxi);   // slope
                    fe_sqr(t, lam);
                    fe_sub(t, t, x1);
                    fe_sub(t, t, x2);      // x3
                    fe_sub(x2, x1, t);
                    fe_mul(x2, x2, lam);
                    fe_sub(y1, x2, y1);    // y3
                    x1 = t;
                } else if (ea == 0 && eb != 0) {
                    x1 = x2;
                    y1 = y2;
                    present = true;
                }
                if (present) chain_add(acc, inf, zone, x1, y1);
            }
            if (inf) { // (0, 0, 1): no key in this lane for this item
                fe_set_zero(acc.x);
                fe_set_zero(acc.y);
                fe_set_one(acc.z);
            }
            fe_mul(cz, cz, acc.z);
            work_store(it + 11 * nth, nth, acc.x);
            work_store(it + 13 * nth, nth, acc.y);
            work_store(it + 15 * nth, nth, acc.z);
            work_store(it + 17 * nth, nth, cz);
        }
        // ------------------------------------------------------------------------------------------------ inv B
        fe_inv(inv, cz);
        // ------------------------------------------------------------------------------------------------ pass C
        r = 0;
        sub = 0;
#pragma unroll 1
        for (u32 b = 0; b < B; b++) {
            const uint4 *it = wk + (u64)b * VK_TREE_QUADS * nth;
            const u32 csub = sub;
            const u64 off = tile_off + (u64)r * VK_BLOCK + tid;
            if (++sub == p.slots) {
                sub = 0;
                r++;
            }
            jac P;
            fe zinv;
            work_load(P.x, it + 11 * nth, nth);
            work_load(P.y, it + 13 * nth, nth);
            work_load(P.z, it + 15 * nth, nth);
            if (b + 1 < B) {
                fe cn;
                work_load(cn, it + VK_TREE_QUADS * nth + 17 * nth, nth); // product of Z over the items after this one
                fe_mul(zinv, inv, cn);
                fe_mul(inv, inv, P.z);
            } else {
                zinv = inv;
            }
            u32 nz = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) nz |= P.x.v[i] | P.y.v[i];
            if (nz == 0) continue;
            fe x, y;
            jac_to_affine(x, y, P, zinv);
            n_keys++;
            u32 hc[5], hu[5];
            u32 bits_c = 0, bits_u = 0;
            if (p.flags & VK_F_COMPRESSED) {
                hash160_pub33(hc, x, y.v[0] & 1u);
                digest_fold(dg, hc);
                if (p.tgt) bits_c = probe_targets(p.tgt, p.tgt_mask, hc);
            }
            if (p.flags & VK_F_UNCOMPRESSED) {
                hash160_pub65(hu, x, y);
                digest_fold(dg, hu);
                if (p.tgt) bits_u = probe_targets(p.tgt, p.tgt_mask, hu);
            }
            int kind = -1;
            if (bits_c & 1u) kind = 0;
            else if (bits_u & 1u) kind = 1;
            else if (bits_c & 2u) kind = 2;
            const bool emit = (p.flags & VK_F_EMIT) != 0;
            if (kind >= 0 || emit) {
                u32 kw[8], k[8];
                item_key<KIND>(kw, p, off, csub);
#pragma unroll
                for (int i = 0; i < 8; i++) k[i] = kw[7 - i];
                scalar_normalize(k);
                if (kind >= 0) {
                    n_match++;
                    u64 slot = atomicAdd((unsigned long long *)&p.stats[2], 1ull);
                    if (slot < p.hit_cap) {
                        vk_hit_dev *h = p.hits + slot;
                        h->seed = p.lo + off;
                        h->sub_index = csub;
                        h->addr_kind = (u32)kind;
#pragma unroll
                        for (int i = 0; i < 8; i++) h->privkey[i] = bswap32(k[7 - i]);
#pragma unroll
                        for (int i = 0; i < 5; i++) h->h160[i] = (kind == 1) ? hu[i] : hc[i];
                        h->pad = 0;
                    }
                }
                if (emit) {
                    const u64 item = off * p.slots + csub;
                    if (p.out_valid) p.out_valid[item] = 1;
                    if (p.out_priv) {
#pragma unroll
                        for (int i = 0; i < 8; i++) p.out_priv[item * 8 + i] = bswap32(k[7 - i]);
                    }
                    if (p.out_h160c && (p.flags & VK_F_COMPRESSED)) {
#pragma unroll
                        for (int i = 0; i < 5; i++) p.out_h160c[item * 5 + i] = hc[i];
                    }
                    if (p.out_h160u && (p.flags & VK_F_UNCOMPRESSED)) {
#pragma unroll
                        for (int i = 0; i < 5; i++) p.out_h160u[item * 5 + i] = hu[i];
                    }
                    if (p.out_pub) {
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            p.out_pub[item * 16 + i] = bswap32(FE_BE(x, i));
                            p.out_pub[item * 16 + 8 + i] = bswap32(FE_BE(y, i));
                        }
                    }
                }
            }
        }
    }
    n_keys = wave_add(n_keys);
    n_match = wave_add(n_match);
    dg[0] = wave_xor(dg[0]);
    dg[1] = wave_xor(dg[1]);
    dg[2] = wave_xor(dg[2]);
    dg[3] = wave_add(dg[3]);
    if ((tid & 63) == 0) {
        atomicAdd((unsigned long long *)&p.stats[0], (unsigned long long)n_keys);
        atomicAdd((unsigned long long *)&p.stats[1], (unsigned long long)n_match);
        atomicXor((unsigned long long *)&p.stats[3], (unsigned long long)dg[0]);
        atomicXor((unsigned long long *)&p.stats[4], (unsigned long long)dg[1]);
        atomicXor((unsigned long long *)&p.stats[5], (unsigned long long)dg[2]);
        atomicAdd((unsigned long long *)&p.stats[6], (unsigned long long)dg[3]);
    }
}
