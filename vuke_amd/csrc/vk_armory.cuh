// vk_armory.cuh -- the armory transform on gfx950: src/transform/armory.rs:21-70,84-101.
//   generate_key(seed32): hash1 = SHA256(seed), hash2 = SHA256(hash1),
//                         chaincode = HMAC-SHA256(key = hash2, "Derive Chaincode from Root Key"),
//                         key = seed as a secret key (1 <= key < n, else nothing is emitted),
//                         4 x { P = key*G; s = SHA256(SHA256(04||X||Y)) xor chaincode; s must be < n and non-zero;
//                               key = key * s mod n }
//   apply_batch: slot 0 seed = SHA256(decimal string), slot 1 seed = 24 zero bytes || 8 BE bytes; a string input has slot 0 only.
// Four fixed-base multiplications with an affine conversion (one field inversion) each, so one armory key costs about
// five plain keys; the k*G table of the engine is reused.
#pragma once
#include "vk_ec.cuh"
#include "vk_scalar.cuh"
#include "vk_transforms.cuh"

#define VK_T_ARMORY 11u

struct vk_ec_ctx {
    const uint4 *table;
    const vk_win_desc *wins;
    u32 nwin;
};

// out[OL] = lo[8] + hi[HL] * (2^256 - n), limbs little-endian; the caller sizes OL so that nothing carries out
template <int HL, int OL>
VK_DEV void scalar_fold(u32 out[OL], const u32 lo[8], const u32 hi[HL])
{
    const u32 C[5] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 1u}; // 2^256 - n
    u64 acc = 0;
    u32 ah = 0;
#pragma unroll
    for (int k = 0; k < OL; k++) {
        if (k < 8) {
            acc += lo[k];
            ah += acc < (u64)lo[k] ? 1u : 0u;
        }
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int j = k - i;
            if (j >= 0 && j < HL) {
                const u64 p = (u64)hi[j] * C[i];
                acc += p;
                ah += acc < p ? 1u : 0u;
            }
        }
        out[k] = (u32)acc;
        acc = (acc >> 32) | ((u64)ah << 32);
        ah = 0;
    }
}

// out = a * b mod n (SecretKey::mul_tweak); a, b < n
VK_DEV void scalar_mul_mod_n(u32 out[8], const u32 a[8], const u32 b[8])
{
    u32 t[16];
    u64 acc = 0;
    u32 ah = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j >= 0 && j < 8) {
                const u64 p = (u64)a[i] * b[j];
                acc += p;
                ah += acc < p ? 1u : 0u;
            }
        }
        t[k] = (u32)acc;
        acc = (acc >> 32) | ((u64)ah << 32);
        ah = 0;
    }
    t[15] = (u32)acc;
    // 2^256 = C (mod n), C < 2^129: 512 -> 386 -> 260 -> 257 -> 256 bits
    u32 r1[13], r2[9], r3[9], r4[9];
    scalar_fold<8, 13>(r1, t, t + 8);
    scalar_fold<5, 9>(r2, r1, r1 + 8);
    scalar_fold<1, 9>(r3, r2, r2 + 8);
    scalar_fold<1, 9>(r4, r3, r3 + 8); // r3[8] is 0 or 1 and then r3 < 2^134: no carry out
    const u32 nl[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    u32 d[8];
    u64 bw = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u64 q = (u64)r4[j] - nl[j] - bw;
        d[j] = (u32)q;
        bw = (q >> 32) & 1;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) out[j] = bw ? r4[j] : d[j];
}

// HMAC-SHA256(key = 32 bytes, "Derive Chaincode from Root Key")
VK_DEV void armory_chaincode(u32 cc[8], const u32 key[8])
{
    u32 w[16], st[8] = VK_SHA256_IV;
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = (i < 8 ? key[i] : 0u) ^ 0x36363636u;
    sha256_compress(st, w);
    // "Deri" "ve C" "hain" "code" " fro" "m Ro" "ot K" "ey" + 0x80
    w[0] = 0x44657269u; w[1] = 0x76652043u; w[2] = 0x6861696eu; w[3] = 0x636f6465u;
    w[4] = 0x2066726fu; w[5] = 0x6d20526fu; w[6] = 0x6f74204bu; w[7] = 0x65798000u;
#pragma unroll
    for (int i = 8; i < 15; i++) w[i] = 0;
    w[15] = (64 + 30) * 8;
    sha256_compress(st, w);
    u32 inner[8];
#pragma unroll
    for (int i = 0; i < 8; i++) inner[i] = st[i];
    u32 so[8] = VK_SHA256_IV;
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = (i < 8 ? key[i] : 0u) ^ 0x5c5c5c5cu;
    sha256_compress(so, w);
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = inner[i];
    w[8] = 0x80000000u;
#pragma unroll
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = (64 + 32) * 8;
    sha256_compress(so, w);
#pragma unroll
    for (int i = 0; i < 8; i++) cc[i] = so[i];
}

// seed = 32 bytes as big-endian words; kw = the derived key
VK_DEV bool armory_generate_key(u32 kw[8], const u32 seed[8], const vk_ec_ctx &ec)
{
    u32 h1[8], h2[8], cc[8];
    sha256_of_digest(h1, seed);
    sha256_of_digest(h2, h1);
    armory_chaincode(cc, h2);
    u32 k[8];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = seed[7 - i];
    if (!scalar_is_key(k)) return false;
#pragma unroll 1
    for (int step = 0; step < 4; step++) {
        jac P;
        ecmult_table(P, k, ec.table, ec.wins, ec.nwin);
        fe zi, ax, ay;
        fe_inv(zi, P.z);
        jac_to_affine(ax, ay, P, zi);
        u32 d1[8], d2[8], s[8], nk[8];
        sha256_pub65(d1, ax, ay);
        sha256_of_digest(d2, d1);
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = d2[7 - i] ^ cc[7 - i];
        if (!scalar_is_key(s)) return false; // Scalar::from_be_bytes rejects >= n, mul_tweak rejects zero
        scalar_mul_mod_n(nk, k, s);
#pragma unroll
        for (int i = 0; i < 8; i++) k[i] = nk[i];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) kw[i] = k[7 - i];
    return true;
}

VK_DEV bool transform_armory(u32 kw[8], u64 seed, u32 sub, const vk_ec_ctx &ec)
{
    u32 s[8];
    if (sub == 0) {
        if (!transform_sha256(s, seed, 0)) return false;
    } else if (sub == 1) {
#pragma unroll
        for (int i = 0; i < 6; i++) s[i] = 0;
        s[6] = (u32)(seed >> 32);
        s[7] = (u32)seed;
    } else {
        return false;
    }
    return armory_generate_key(kw, s, ec);
}

VK_DEV bool transform_armory_string(u32 kw[8], const unsigned char *str, u32 len, u32 sub, const vk_ec_ctx &ec)
{
    if (sub != 0) return false;
    vk_bytes b;
    b.p = str;
    b.len = len;
    b.suf[0] = b.suf[1] = b.suf[2] = 0;
    b.suf_len = 0;
    u32 s[8];
    sha256_stream(s, b);
    return armory_generate_key(kw, s, ec);
}
