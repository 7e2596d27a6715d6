// vk_scalar.cuh -- scalars mod the secp256k1 group order n for the key-derivation chains
// (Electrum child keys, BIP32 CKD): the checks secp256k1::SecretKey / Scalar apply in the reference
// (src/electrum.rs:118-130, src/multibit.rs:334-350).
#pragma once
#include "vk_field.cuh"

// 1 <= k < n  (SecretKey::from_slice)
VK_DEV bool scalar_is_key(const u32 k[8])
{
    u32 t[8], diff = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = k[i];
    scalar_normalize(t); // changes k iff k == 0 or k >= n
#pragma unroll
    for (int i = 0; i < 8; i++) diff |= t[i] ^ k[i];
    return diff == 0;
}

// s < n, zero allowed  (Scalar::from_be_bytes)
VK_DEV bool scalar_lt_n(const u32 s[8])
{
    u32 any = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) any |= s[i];
    return any == 0 || scalar_is_key(s);
}

// out = a + b mod n for a, b < n; returns false if the sum is zero  (SecretKey::add_tweak)
VK_DEV bool scalar_add_mod_n(u32 out[8], const u32 a[8], const u32 b[8])
{
    const u32 nl[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    u32 c[8], d[8];
    u64 cy = 0, bw = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        cy += (u64)a[j] + b[j];
        c[j] = (u32)cy;
        cy >>= 32;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        u64 q = (u64)c[j] - nl[j] - bw;
        d[j] = (u32)q;
        bw = (q >> 32) & 1;
    }
    const bool ge = cy != 0 || bw == 0; // sum < 2n: one conditional subtraction
    u32 any = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        out[j] = ge ? d[j] : c[j];
        any |= out[j];
    }
    return any != 0;
}
