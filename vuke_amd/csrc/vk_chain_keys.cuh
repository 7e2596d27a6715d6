// vk_chain_keys.cuh -- the last kernel of every key-derivation chain, i.e. the part that needs the curve: BIP-32 private-key
// derivation in registers for MultiBit-HD and Bitimage (seed -> master key -> CKDpriv along the path, serP(K) through the k*G
// table: src/multibit.rs:300-350, src/bitimage.rs:200-275) and Electrum's master public key, sequence hashes and child keys
// (src/electrum.rs:81-195).  The BIP-32 messages have a fixed layout (HMAC pads, the 37 bytes of a CKD message) and are built
// in registers; Electrum's "{i}:{c}:" || X || Y goes through the staging slice.
#pragma once
#include "vk_chain.cuh"
#include "vk_ec.cuh"
#include "vk_scalar.cuh"
#include "vk_electrum.cuh" // state layout of the Electrum stretch

// ------------------------------------------------------------------------------------------------ BIP-32 in registers
// A private key as scalar limbs k[8] (little-endian u32) and a chain code as four big-endian u64 words.
struct vk_xprv {
    u32 k[8];
    u64 c[4];
};

VK_DEV void limbs_from_be64(u32 k[8], const u64 w[4])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        k[7 - 2 * i] = (u32)(w[i] >> 32);
        k[6 - 2 * i] = (u32)w[i];
    }
}
VK_DEV void be64_from_limbs(u64 w[4], const u32 k[8])
{
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = pack64(k[6 - 2 * i], k[7 - 2 * i]);
}

// (k, c) = HMAC-SHA512("Bitcoin seed", seed)
VK_DEV void bip32_master(vk_xprv &m, const u64 seed[8])
{
    u64 key[16], ist[8], ost[8], in[8], I[8], w[16];
    key[0] = 0x426974636f696e20ull; // "Bitcoin "
    key[1] = 0x7365656400000000ull; // "seed"
#pragma unroll
    for (int i = 2; i < 16; i++) key[i] = 0;
    hmac512_pads(ist, ost, key);
    sha512_digest_block(w, seed);
#pragma unroll
    for (int i = 0; i < 8; i++) in[i] = ist[i];
    sha512_compress(in, w);
    hmac512_outer(I, ost, in);
    limbs_from_be64(m.k, I);
#pragma unroll
    for (int i = 0; i < 4; i++) m.c[i] = I[4 + i];
}

// HMAC pads of a parent's chain code: shared by all children of that parent
struct vk_ckd_pads {
    u64 ist[8], ost[8];
};
VK_DEV void ckd_pads(vk_ckd_pads &p, const u64 c[4])
{
    u64 key[16];
#pragma unroll
    for (int i = 0; i < 4; i++) key[i] = c[i];
#pragma unroll
    for (int i = 4; i < 16; i++) key[i] = 0;
    hmac512_pads(p.ist, p.ost, key);
}

// the 33 bytes that precede the index in a CKD message: 0x00 || k (hardened) or serP(K) = 02/03 || X (normal)
struct vk_ckd_data {
    u32 prefix;
    u64 x[4]; // big-endian words
};
VK_DEV void ckd_data_hardened(vk_ckd_data &d, const u32 k[8])
{
    d.prefix = 0;
    be64_from_limbs(d.x, k);
}
VK_DEV void ckd_data_normal(vk_ckd_data &d, const u32 k[8], const uint4 *table, const vk_win_desc *wins, u32 nwin)
{
    jac P;
    ecmult_table(P, k, table, wins, nwin);
    fe zi, x, y;
    fe_inv(zi, P.z);
    jac_to_affine(x, y, P, zi);
    d.prefix = 0x02u | (y.v[0] & 1u);
    be64_from_limbs(d.x, x.v);
}

// one CKDpriv step in the reference's spelling (src/multibit.rs:300-350, src/bitimage.rs:200-275):
// I = HMAC-SHA512(c_par, data33 || ser32(index)); child = IL + k_par mod n, chain code IR.
VK_DEV bool ckd_step(vk_xprv &child, const u32 kpar[8], const vk_ckd_pads &pads, const vk_ckd_data &d, u32 index)
{
    u64 w[16], in[8], I[8];
    w[0] = ((u64)d.prefix << 56) | (d.x[0] >> 8);
    w[1] = (d.x[0] << 56) | (d.x[1] >> 8);
    w[2] = (d.x[1] << 56) | (d.x[2] >> 8);
    w[3] = (d.x[2] << 56) | (d.x[3] >> 8);
    w[4] = (d.x[3] << 56) | ((u64)index << 24) | (0x80ull << 16);
#pragma unroll
    for (int i = 5; i < 15; i++) w[i] = 0;
    w[15] = (128 + 37) * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) in[i] = pads.ist[i];
    sha512_compress(in, w);
    hmac512_outer(I, pads.ost, in);
    u32 il[8];
    limbs_from_be64(il, I);
#pragma unroll
    for (int i = 0; i < 4; i++) child.c[i] = I[4 + i];
    if (!scalar_is_key(kpar)) return false;      // SecretKey::from_slice(parent)
    if (!scalar_lt_n(il)) return false;          // Scalar::from_be_bytes(IL)
    return scalar_add_mod_n(child.k, kpar, il);  // add_tweak: a zero result is an error
}

VK_DEV void store_key_be(u32 *__restrict__ out, const u32 k[8])
{
#pragma unroll
    for (int q = 0; q < 8; q++) out[q] = bswap32(k[7 - q]);
}

// ------------------------------------------------------------------------------------------------ MultiBit: final
__global__ void __launch_bounds__(VK_BLOCK) vk_mb_final_kernel(u64 n, u32 count, const u64 *__restrict__ state, const unsigned char *__restrict__ ok,
                                                                 const uint4 *__restrict__ table, const vk_win_desc *__restrict__ wins, u32 nwin,
                                                                 u32 *__restrict__ keys_out, unsigned char *__restrict__ valid_out)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n || !ok[idx]) return;
    u64 seed[8];
    pb_state_result(seed, state, idx);
    vk_xprv m, a, b;
    bip32_master(m, seed);
    vk_ckd_pads pads;
    vk_ckd_data d;
    // ---- m/0' (hardened)
    ckd_pads(pads, m.c);
    ckd_data_hardened(d, m.k);
    if (!ckd_step(a, m.k, pads, d, 0x80000000u)) return;
    // ---- m/0'/0 (normal): private_to_public needs a valid parent
    if (!scalar_is_key(a.k)) return;
    ckd_pads(pads, a.c);
    ckd_data_normal(d, a.k, table, wins, nwin);
    if (!ckd_step(b, a.k, pads, d, 0)) return;
    // ---- m/0'/0/i
    if (!scalar_is_key(b.k)) return;
    ckd_pads(pads, b.c);
    ckd_data_normal(d, b.k, table, wins, nwin);
    u32 *kout = keys_out + idx * count * 8;
    unsigned char *vout = valid_out + idx * count;
#pragma unroll 1
    for (u32 j = 0; j < count; j++) {
        vk_xprv ch;
        if (!ckd_step(ch, b.k, pads, d, j)) continue;
        store_key_be(kout + j * 8, ch.k);
        vout[j] = 1;
    }
}

// ------------------------------------------------------------------------------------------------------ bitimage: final
// the path's components (bit 31 = hardened) come from HBM: any depth, as the reference's derive_path walks any Vec (src/bitimage.rs:76-116;
// 16 levels in the kernel arguments until round 5)
struct vk_bip32_path {
    u32 depth;
    const u32 *comp;
};

__global__ void __launch_bounds__(VK_BLOCK) vk_bitimage_final_kernel(u64 n_items, const vk_bip32_path path, u32 count, const u64 *__restrict__ state,
                                                                       const uint4 *__restrict__ table, const vk_win_desc *__restrict__ wins, u32 nwin,
                                                                       u32 *__restrict__ keys_out, unsigned char *__restrict__ valid_out)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n_items) return;
    u64 seed[8];
    pb_state_result(seed, state, idx);
    vk_xprv cur;
    bip32_master(cur, seed);
    u32 *kout = keys_out + idx * count * 8;
    unsigned char *vout = valid_out + idx * count;
    if (path.depth == 0) { // "m": the master key as it is, derive_count times (increment_path_index leaves "m" alone)
#pragma unroll 1
        for (u32 j = 0; j < count; j++) {
            store_key_be(kout + j * 8, cur.k);
            vout[j] = 1;
        }
        return;
    }
    vk_ckd_pads pads;
    vk_ckd_data d;
    // ---- all levels but the last
#pragma unroll 1
    for (u32 lv = 0; lv + 1 < path.depth; lv++) {
        const u32 comp = path.comp[lv];
        if (!scalar_is_key(cur.k)) return; // SecretKey::from_slice(parent) in scalar_add / private_to_public
        ckd_pads(pads, cur.c);
        if (comp >> 31) ckd_data_hardened(d, cur.k);
        else ckd_data_normal(d, cur.k, table, wins, nwin);
        vk_xprv nx;
        if (!ckd_step(nx, cur.k, pads, d, comp)) return;
        cur = nx;
    }
    // ---- the last level: index, index + 1, ...; the first failure ends the item (the reference does not advance the
    // path after a failed derivation, so every later attempt repeats it)
    const u32 last = path.comp[path.depth - 1];
    if (!scalar_is_key(cur.k)) return;
    ckd_pads(pads, cur.c);
    if (last >> 31) ckd_data_hardened(d, cur.k);
    else ckd_data_normal(d, cur.k, table, wins, nwin);
#pragma unroll 1
    for (u32 j = 0; j < count; j++) {
        const u64 index = (u64)(last & 0x7FFFFFFFu) + j;
        if (index >= 0x80000000ull) return; // "exceeds BIP32 limit"
        vk_xprv ch;
        if (!ckd_step(ch, cur.k, pads, d, (u32)index | (last & 0x80000000u))) return;
        store_key_be(kout + j * 8, ch.k);
        vout[j] = 1;
    }
}

// ------------------------------------------------------------------------------------------------------ electrum: final
__global__ void __launch_bounds__(VK_BLOCK) vk_electrum_final_kernel(u64 n, u32 for_change, u32 count, const u32 *__restrict__ state,
                                                                       const unsigned char *__restrict__ ok, u64 *__restrict__ stage_all,
                                                                       const uint4 *__restrict__ table, const vk_win_desc *__restrict__ wins, u32 nwin,
                                                                       u32 *__restrict__ keys_out, unsigned char *__restrict__ valid_out)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n || !ok[idx]) return;
    const u32 *s = state + (idx >> 6) * (VK_EL_WORDS * 64) + (idx & 63u);
    // ---- master key: SecretKey::from_slice accepts 1 <= x < n only
    u32 m[8];
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] = s[64 * (7 - i)];
    if (!scalar_is_key(m)) return; // 0 or >= n
    jac P;
    ecmult_table(P, m, table, wins, nwin);
    fe zi, ax, ay;
    fe_inv(zi, P.z);
    jac_to_affine(ax, ay, P, zi);
    u64 xy[8]; // X || Y as byte strings packed for the stage writer (first byte in the low byte)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        xy[j] = pack64(__builtin_bswap32(FE_BE(ax, 2 * j)), __builtin_bswap32(FE_BE(ax, 2 * j + 1)));
        xy[4 + j] = pack64(__builtin_bswap32(FE_BE(ay, 2 * j)), __builtin_bswap32(FE_BE(ay, 2 * j + 1)));
    }
    u64 *stage = stage_all + idx * VK_STAGE_WORDS;
    u32 *kout = keys_out + idx * count * 8;
    unsigned char *vout = valid_out + idx * count;
    // ---- child keys
#pragma unroll 1
    for (u32 i = 0; i < count; i++) {
        // "{i}:{c}:" || X || Y
        vk_stage_writer wr;
        wr.init(stage);
        {
            u32 div = 1;
            while (i / div >= 10) div *= 10;
            for (; div; div /= 10) wr.put('0' + (i / div) % 10, 1);
            wr.put(':', 1);
            wr.put(for_change ? '1' : '0', 1);
            wr.put(':', 1);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) wr.put(xy[j], 8);
        const u32 len = wr.finish(16); // at most 13 + 64 bytes: two SHA-256 blocks
        u32 d1[8], seq[8];
        sha256_stage(d1, stage, len);
        sha256_of_digest(seq, d1);
        // Scalar::from_be_bytes: seq must be < n (zero allowed)
        u32 sc[8], ck[8];
#pragma unroll
        for (int j = 0; j < 8; j++) sc[j] = seq[7 - j];
        if (!scalar_lt_n(sc)) continue;
        // child = master + seq mod n; a zero result is rejected by add_tweak
        if (!scalar_add_mod_n(ck, m, sc)) continue;
#pragma unroll
        for (int j = 0; j < 8; j++) kout[i * 8 + j] = bswap32(ck[7 - j]);
        vout[i] = 1;
    }
}
