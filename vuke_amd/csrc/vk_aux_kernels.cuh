// vk_aux_kernels.cuh -- the small kernels around the sweep: k*G table builder, Transform::apply_batch for numeric and
// passphrase inputs, field test hook.  Compiled once (vk_engine.hip); the sweep kernel instantiations live in
// vk_sweep_tu.hip so the heavy translation units build in parallel.
#pragma once
#include "vk_kernels.cuh"

// ---------------------------------------------------------------- table builder
// entry (w, d) = (d << off_w) * G, affine canonical; d = 0 and unrepresentable entries = 0.
//
// Two phases, ONE launch each for all windows, so that even the 120 GB nine-window table builds in a fraction of a second:
//   seed     the 2^h "low" multiples d = j and the 2^(tb-h) "high" multiples d = i << h of every window, each by a generic
//            double-and-add (a few 10^4 scalar multiplications per window; windows below 2^12 entries are seeded whole);
//   combine  every other entry is T[i << h] + T[j]: one affine + affine addition (2M + 1S) under a shared inversion
//            (Montgomery's trick, 3M per entry + one fe_inv per RUN).  A wave owns 64 consecutive low digits j -- one per
//            lane, (xj, yj) stay in registers -- and walks a run of K consecutive high digits i, which are wave-uniform:
//            every store of a step is 64 consecutive entries = 4 KB, the high multiple is one broadcast line.  The K
//            running products of a lane are parked in the engine's batch-inversion workspace as [wave][k][half][lane]
//            uint4 rows (coalesced 1 KB rows; 64 B of workspace traffic per entry next to the 64 B of the entry itself),
//            so the run length is not bounded by registers: K = 256 puts the inversion at ~1 multiplication per entry.
//            Rounds 1-3 kept the products in registers (one lane = one high digit x 16 low digits): 298 VGPRs + 42 AGPR
//            spills, one fe_inv per 16 entries = 3/4 of the kernel's multiplications, 253 ms for the 120 GB table.
VK_DEV bool table_scalar(u32 s[8], u32 d, u32 shift, bool allow_2p256 = false)
{
    const u32 dbits = 32 - __builtin_clz(d | 1u);
    if (d == 0) return false;
    if (shift + dbits > 256) {
        // the one scalar beyond 256 bits that can occur: 2^256 (final carry of a signed layout) = 2^256 mod n
        if (allow_2p256 && shift + dbits == 257 && (d & (d - 1)) == 0) {
            const u32 r[8] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 1u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = r[i];
            return true;
        }
        return false;
    }
    const u32 limb = shift >> 5, sh = shift & 31;
    const u64 wide = (u64)d << sh;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 v = 0;
        v = (limb == (u32)i) ? (u32)wide : v;
        v = (limb + 1 == (u32)i) ? (u32)(wide >> 32) : v;
        s[i] = v;
    }
    return true;
}

VK_DEV void table_store(uint4 *dst, const fe &x, const fe &y)
{
    dst[0] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    dst[1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    dst[2] = make_uint4(y.v[0], y.v[1], y.v[2], y.v[3]);
    dst[3] = make_uint4(y.v[4], y.v[5], y.v[6], y.v[7]);
}

// what the two builder kernels know about a window: where its entry 0 lives, its scalar shift, its index bits, the split h
// (low digits: h bits) and the first work item of the combine phase that belongs to it
struct vk_tbl_win {
    u64 base;
    u32 off, tb, h, first_item;
};
struct vk_tbl_jobs {
    u32 nwin, with_max; // with_max: every window also holds the single entry 2^tb (signed layouts)
    u32 K, items;       // combine phase: run length (power of two, <= 2^(tb-h) of every combined window), work items in all
    vk_tbl_win w[VK_MAX_WIN];
};

// seed phase, blockIdx.y = window: the 2^h low digits, the 2^(tb-h) high digits i << h, and (with_max) the single entry 2^tb.
__global__ void __launch_bounds__(VK_BLOCK) vk_table_seed_kernel(uint4 *table, vk_tbl_jobs jobs)
{
    const vk_tbl_win &wd = jobs.w[blockIdx.y];
    uint4 *win = table + wd.base * 4;
    const u32 tb = wd.tb, h = wd.h, off = wd.off, with_max = jobs.with_max;
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    const u32 nlo = 1u << h, nhi = 1u << (tb - h);
    if (idx >= (u64)nlo + nhi + with_max) return;
    u32 d;
    if (idx < nlo) d = (u32)idx;
    else if (idx < (u64)nlo + nhi) d = (u32)(idx - nlo) << h;
    else d = 1u << tb;
    if (idx == nlo) return; // high digit 0 is entry 0 again
    uint4 *dst = win + (u64)d * 4;
    u32 s[8];
    if (!table_scalar(s, d, off, with_max && d == (1u << tb))) {
        dst[0] = dst[1] = dst[2] = dst[3] = make_uint4(0, 0, 0, 0);
        return;
    }
    jac P;
    ecmult_generic(P, s);
    fe zi, x, y;
    fe_inv(zi, P.z);
    jac_to_affine(x, y, P, zi);
    table_store(dst, x, y);
}

// combine phase, persistent waves striding over the work items of all windows: item = (window, block of 64 low digits, run
// of K high digits).  `ws` holds K x 2 rows of 64 uint4 per wave of the grid.
#ifndef VK_TBL_WAVES
#define VK_TBL_WAVES 3 /* 138 VGPRs with the inversion chain inlined: three waves per SIMD, no spills (128 for four would spill 10) */
#endif
__global__ void __launch_bounds__(VK_BLOCK, VK_TBL_WAVES) vk_table_combine_kernel(uint4 *table, vk_tbl_jobs jobs, uint4 *ws)
{
    const u32 lane = threadIdx.x & 63u;
    const u32 wave = __builtin_amdgcn_readfirstlane((blockIdx.x * VK_BLOCK + threadIdx.x) >> 6);
    const u32 nwaves = gridDim.x * (VK_BLOCK / 64);
    const u32 K = jobs.K;
    uint4 *my = ws + (u64)wave * K * 128 + lane;
    fe one;
    fe_set_one(one);
#pragma unroll 1
    for (u32 item = wave; item < jobs.items; item += nwaves) {
        u32 wi = 0;
#pragma unroll 1
        for (u32 t = 1; t < jobs.nwin; t++)
            if (jobs.w[t].first_item <= item) wi = t; // first_item is non-decreasing; windows without items repeat their successor's
        const vk_tbl_win &wd = jobs.w[wi];
        const u32 h = wd.h, off = wd.off;
        const u32 rel = item - wd.first_item, jblocks = (1u << h) >> 6;
        const u32 jb = rel % jblocks, i0 = (rel / jblocks) * K;
        uint4 *win = table + wd.base * 4;
        const u32 j = (jb << 6) + lane;
        fe xj, yj;
        table_load(xj, yj, win + (u64)j * 4);
        fe c = one;
        // forward: running products of the x differences, parked row by row
#pragma unroll 1
        for (u32 k = 0; k < K; k++) {
            const u32 i = i0 + k, hi_d = i << h;
            u32 tmp[8];
            const bool live = i != 0 && table_scalar(tmp, hi_d, off); // wave-uniform
            if (live) {
                const uint4 *src = win + (u64)hi_d * 4;
                const uint4 a = src[0], b = src[1];
                fe xi, dx;
                xi.v[0] = a.x; xi.v[1] = a.y; xi.v[2] = a.z; xi.v[3] = a.w;
                xi.v[4] = b.x; xi.v[5] = b.y; xi.v[6] = b.z; xi.v[7] = b.w;
                fe_sub(dx, xj, xi);
                if (j == 0) dx = one;
                fe_mul(c, c, dx);
            }
            my[(u64)k * 128] = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]);
            my[(u64)k * 128 + 64] = make_uint4(c.v[4], c.v[5], c.v[6], c.v[7]);
        }
        fe inv = fe_inv_chain(c); // inlined: the kernel's register budget covers it (as a call its 178 VGPRs set the kernel's occupancy)
        // backward: 1 / dx of step k = inv * (product before k); the addition; the entries of the step, 4 KB per wave
#pragma unroll 1
        for (u32 kk = K; kk > 0; kk--) {
            const u32 k = kk - 1, i = i0 + k, hi_d = i << h;
            if (i == 0) continue; // the seeded low multiples themselves
            uint4 *dst = win + ((u64)hi_d + j) * 4;
            u32 tmp[8];
            if (!table_scalar(tmp, hi_d, off)) { // digits whose scalar does not fit 256 bits are never looked up: zero
                if (j != 0) dst[0] = dst[1] = dst[2] = dst[3] = make_uint4(0, 0, 0, 0);
                continue;
            }
            fe xi, yi, dx, dinv;
            table_load(xi, yi, win + (u64)hi_d * 4);
            fe_sub(dx, xj, xi);
            if (j == 0) dx = one;
            if (k > 0) {
                const uint4 a = my[(u64)(k - 1) * 128], b = my[(u64)(k - 1) * 128 + 64];
                fe cp;
                cp.v[0] = a.x; cp.v[1] = a.y; cp.v[2] = a.z; cp.v[3] = a.w;
                cp.v[4] = b.x; cp.v[5] = b.y; cp.v[6] = b.z; cp.v[7] = b.w;
                fe_mul(dinv, inv, cp);
                fe_mul(inv, inv, dx);
            } else {
                dinv = inv;
            }
            // (xi, yi) + (xj, yj), distinct x guaranteed (distinct scalars below n, never negatives of each other)
            fe lam, x3, y3, t;
            fe_sub(t, yj, yi);
            fe_mul(lam, t, dinv);
            fe_sqr(x3, lam);
            fe_sub(x3, x3, xi);
            fe_sub(x3, x3, xj);
            fe_sub(t, xi, x3);
            fe_mul(y3, lam, t);
            fe_sub(y3, y3, yi);
            fe_canon(x3);
            fe_canon(y3);
            if (j != 0) table_store(dst, x3, y3); // j = 0 is the seeded high multiple itself
        }
    }
}

// ---------------------------------------------------------------- Transform::apply_batch
// seeds == nullptr: the inputs are the range base, base + 1, ... (the staged sweep of the chained transforms)
template <u32 KIND>
__global__ void __launch_bounds__(VK_BLOCK) vk_transform_kernel(vk_spec_dev spec, const u64 *seeds, u64 base, u64 n, u32 slots,
                                                                u32 *keys_out, unsigned char *valid, vk_ec_ctx ec)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n * slots) return;
    const u64 i = idx / slots;
    const u32 sub = (u32)(idx - i * slots);
    const u64 seed = seeds ? seeds[i] : base + i;
    u32 kw[8];
    bool ok = KIND == VK_T_ARMORY ? transform_armory(kw, seed, sub, ec) : transform_key<KIND>(kw, spec, seed, sub);
    uint4 *dst = reinterpret_cast<uint4 *>(keys_out) + idx * 2; // two 16-byte stores per key
    dst[0] = ok ? make_uint4(bswap32(kw[0]), bswap32(kw[1]), bswap32(kw[2]), bswap32(kw[3])) : make_uint4(0, 0, 0, 0);
    dst[1] = ok ? make_uint4(bswap32(kw[4]), bswap32(kw[5]), bswap32(kw[6]), bswap32(kw[7])) : make_uint4(0, 0, 0, 0);
    valid[idx] = ok ? 1 : 0;
}

// sha256_chain over a range, one lane per (input, seed form, variant) walking its chain once: the iterated variant is
// k_0 = SHA256(seed), k_i = SHA256(k_{i-1}) (src/sha256_chain.rs:88-145), so a lane per KEY would recompute i hashes for key i.
// Slot order = transform_sha256_chain's: form * (V * D) + v * D + idx.
__global__ void __launch_bounds__(VK_BLOCK) vk_sha256_chain_range_kernel(vk_spec_dev spec, u64 base, u64 n, u32 *keys_out, unsigned char *valid)
{
    const u32 nv = spec.variant == VK_VARIANT_ALL ? 4u : 1u, depth = spec.chain_depth, chains = 2u * nv;
    const u64 t = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (t >= n * chains) return;
    // chain-major lane order: the lanes of a wave walk the same (seed form, variant) for consecutive inputs -- no divergence
    // between the iterated, indexed and counter code paths
    const u32 c = (u32)(t / n);
    const u64 i = t - (u64)c * n;
    const u32 form = c / nv, vi = c - form * nv, v = spec.variant == VK_VARIANT_ALL ? vi : spec.variant;
    const u64 seed = base + i, first = (i * chains + c) * depth; // key index of this chain's idx 0
    u32 kw[8];
    bool ok = true;
#pragma unroll 1
    for (u32 idx = 0; idx < depth; idx++) {
        if (v == 0 && idx > 0) {
            u32 nx[8];
            sha256_of_digest(nx, kw);
#pragma unroll
            for (int j = 0; j < 8; j++) kw[j] = nx[j];
        } else {
            ok = transform_sha256_chain(kw, seed, form * nv * depth + vi * depth + idx, spec.variant, depth);
        }
        // two 16-byte stores per key (the lanes of a wave write keys 32 * depth bytes apart: fewer, wider requests)
        uint4 *dst = reinterpret_cast<uint4 *>(keys_out) + (first + idx) * 2;
        dst[0] = ok ? make_uint4(bswap32(kw[0]), bswap32(kw[1]), bswap32(kw[2]), bswap32(kw[3])) : make_uint4(0, 0, 0, 0);
        dst[1] = ok ? make_uint4(bswap32(kw[4]), bswap32(kw[5]), bswap32(kw[6]), bswap32(kw[7])) : make_uint4(0, 0, 0, 0);
        valid[first + idx] = ok ? 1 : 0;
    }
}

// armory over a range in a sweep, batched: the four chained public keys of a key (src/transform/armory.rs:84-101) come from the
// sweep kernel itself (raw keys in, affine X || Y out: table reads through LDS, one field inversion per 256 keys) instead of a
// k*G and an inversion per lane and step.  vk_armory_begin_kernel: seed -> (key, chaincode); four times { raw-key sweep with
// out_pub; vk_armory_step_kernel: s = dSHA256(04 || X || Y) xor chaincode, key = key * s mod n }.
// seeds32 != nullptr: one item per 32-byte root seed already in `keys` (string inputs: SHA256(string)); else the range form:
// item = (input, slot), slot 0 seed = SHA256(decimal string), slot 1 seed = 24 zero bytes || 8 BE bytes
__global__ void __launch_bounds__(VK_BLOCK) vk_armory_begin_kernel(u64 base, u64 n_items, const u32 *seeds32, u32 *keys, u32 *chaincodes, unsigned char *valid)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n_items) return;
    u32 s[8];
    bool ok = true;
    if (seeds32) {
        ok = valid[idx] != 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = bswap32(seeds32[idx * 8 + i]);
    } else if ((idx & 1) == 0) {
        ok = transform_sha256(s, base + (idx >> 1), 0);
    } else {
        const u64 seed = base + (idx >> 1);
#pragma unroll
        for (int i = 0; i < 6; i++) s[i] = 0;
        s[6] = (u32)(seed >> 32);
        s[7] = (u32)seed;
    }
    u32 h1[8], h2[8], cc[8], k[8];
    sha256_of_digest(h1, s);
    sha256_of_digest(h2, h1);
    armory_chaincode(cc, h2);
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = s[7 - i];
    ok = ok && scalar_is_key(k);
    uint4 *kd = reinterpret_cast<uint4 *>(keys) + idx * 2, *cd = reinterpret_cast<uint4 *>(chaincodes) + idx * 2;
    kd[0] = ok ? make_uint4(bswap32(s[0]), bswap32(s[1]), bswap32(s[2]), bswap32(s[3])) : make_uint4(0, 0, 0, 0);
    kd[1] = ok ? make_uint4(bswap32(s[4]), bswap32(s[5]), bswap32(s[6]), bswap32(s[7])) : make_uint4(0, 0, 0, 0);
    cd[0] = make_uint4(cc[0], cc[1], cc[2], cc[3]);
    cd[1] = make_uint4(cc[4], cc[5], cc[6], cc[7]);
    valid[idx] = ok ? 1 : 0;
}

__global__ void __launch_bounds__(VK_BLOCK) vk_armory_step_kernel(u64 items, const u32 *__restrict__ pub /* X || Y, raw bytes */, u32 *keys,
                                                                  const u32 *__restrict__ chaincodes, unsigned char *valid)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= items || !valid[idx]) return;
    fe ax, ay;
#pragma unroll
    for (int i = 0; i < 8; i++) { // FE_BE(f, j) = big-endian word j
        FE_BE(ax, i) = bswap32(pub[idx * 16 + i]);
        FE_BE(ay, i) = bswap32(pub[idx * 16 + 8 + i]);
    }
    u32 d1[8], d2[8], s[8], k[8], nk[8];
    sha256_pub65(d1, ax, ay);
    sha256_of_digest(d2, d1);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        s[i] = d2[7 - i] ^ chaincodes[idx * 8 + 7 - i];
        k[i] = bswap32(keys[idx * 8 + 7 - i]);
    }
    if (!scalar_is_key(s)) { // Scalar::from_be_bytes rejects >= n, mul_tweak rejects zero
        valid[idx] = 0;
        return;
    }
    scalar_mul_mod_n(nk, k, s);
#pragma unroll
    for (int i = 0; i < 8; i++) keys[idx * 8 + i] = bswap32(nk[7 - i]);
}

__global__ void __launch_bounds__(VK_BLOCK) vk_zero_invalid_keys_kernel(u64 items, u32 *keys, const unsigned char *valid)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= items || valid[idx]) return;
    uint4 *d = reinterpret_cast<uint4 *>(keys) + idx * 2;
    d[0] = d[1] = make_uint4(0, 0, 0, 0);
}

// Transform::apply_batch for Input::from_string inputs of the hash transforms: one lane per (string, slot)
__global__ void __launch_bounds__(VK_BLOCK) vk_strings_kernel(vk_spec_dev spec, const unsigned char *data, const u32 *offsets, u64 n, u32 slots,
                                                              u32 *keys_out, unsigned char *valid, vk_ec_ctx ec)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n * slots) return;
    const u64 i = idx / slots;
    const u32 sub = (u32)(idx - i * slots);
    u32 kw[8];
    const unsigned char *str = data + offsets[i];
    const u32 len = offsets[i + 1] - offsets[i];
    bool ok = spec.kind == VK_T_ARMORY ? transform_armory_string(kw, str, len, sub, ec) : transform_string(kw, spec, str, len, sub);
    uint4 *dst = reinterpret_cast<uint4 *>(keys_out) + idx * 2; // two 16-byte stores per key
    dst[0] = ok ? make_uint4(bswap32(kw[0]), bswap32(kw[1]), bswap32(kw[2]), bswap32(kw[3])) : make_uint4(0, 0, 0, 0);
    dst[1] = ok ? make_uint4(bswap32(kw[4]), bswap32(kw[5]), bswap32(kw[6]), bswap32(kw[7])) : make_uint4(0, 0, 0, 0);
    valid[idx] = ok ? 1 : 0;
}

// ---------------------------------------------------------------- field test hook (libvuke_hip_hooks.so only)
#ifdef VK_TEST_HOOKS
__global__ void vk_field_kernel(int op, const u32 *a, const u32 *b, u64 n, u32 *out)
{
    const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    fe x, y, r;
#pragma unroll
    for (int i = 0; i < 8; i++) { // operands arrive as big-endian bytes
        x.v[i] = bswap32(a[idx * 8 + 7 - i]);
        y.v[i] = b ? bswap32(b[idx * 8 + 7 - i]) : 0u;
    }
    switch (op) {
    case 0: fe_mul(r, x, y); break;
    case 1: fe_sqr(r, x); break;
    case 2: fe_inv(r, x); break;
    case 3: fe_add(r, x, y); break;
    case 5: fe_sub3(r, x, y, y); break;                      // x - 3y
    case 6: r = x; fe_sub3(r, r, y, r); break;               // -x - y, result and third operand aliasing the first
    case 7: fe_add(r, x, y); fe_sub3(r, x, y, r); break;     // -x - 3y, result aliasing the third
    // aliasing forms of the generated statements (the shapes the sweep uses: inv = inv * e, c = c * e, zinv = zinv^2, ...)
    case 8: r = x; fe_mul(r, r, y); break;                   // r = r * y
    case 9: r = y; fe_mul(r, x, r); break;                   // r = x * r
    case 10: r = x; fe_mul(r, r, r); break;                  // r = r * r through the general multiplier
    case 11: r = x; fe_sqr(r, r); break;                     // r = r^2
    case 12: r = x; fe_add(r, r, r); break;                  // r = 2x
    case 13: r = x; fe_sub(r, y, r); break;                  // r = y - x
    case 14: r = x; fe_sub3(r, r, r, r); break;              // r = x - x - 2x = -2x
    case 15: r = x; fe_sqr_n(r, r, 5); fe_mul(r, r, y); break; // x^32 * y: a loop-carried element feeding a multiply
    // the fused sum of two products (the y-coordinate of the sweep's point additions)
    case 16: fe_mul2(r, x, y, y, x); break;                        // 2xy
    case 17: fe_mul2(r, x, x, y, y); break;                        // x^2 + y^2
    case 18: r = x; fe_mul2(r, r, y, y, r); break;                 // result aliasing the first and the last operand
    case 19: r = y; fe_sqr(r, r); fe_mul2(r, x, r, r, y); break;   // x y^2 + y^3: result aliasing the inner operands
    default: fe_sub(r, x, y); break;
    }
    fe_canon(r);
#pragma unroll
    for (int i = 0; i < 8; i++) out[idx * 8 + 7 - i] = bswap32(r.v[i]);
}
#endif // VK_TEST_HOOKS
