// vk_electrum.cuh -- Electrum v1 deterministic-wallet keys on gfx950, one seed string per lane.
//
// Reference behaviour (src/electrum.rs:57-195, src/transform/electrum.rs:43-67):
//   seed must be all ASCII hex digits, else the input is skipped;
//   x = SHA256(seed || seed); 99 999 times x = SHA256(x || seed);           (stretch_key)
//   master = x must be a valid scalar (1 <= x < n), else the input is skipped;
//   mpk = X || Y of master * G;
//   seq_i = SHA256(SHA256("{i}:{0|1}:" || mpk)); skipped if seq_i >= n;
//   key_i = master + seq_i mod n; skipped if zero.
// The 100 000-round stretch dominates (2 x 10^5 compressions per seed against one k*G), so the
// loop keeps the constant part of the message (seed bytes, padding, length) in registers and only
// swaps the 8 chaining words; one-off message assembly uses a small per-lane byte buffer.
#pragma once
#include "vk_ec.cuh"
#include "vk_hash.cuh"
#include "vk_scalar.cuh"

#define VK_ELECTRUM_MAX_SEED 64 /* bytes; longer strings are rejected by the host entry point */

VK_DEV u32 load_be32(const unsigned char *p) { return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3]; }

// SHA-256 of msg[0..len) (len <= 183) held in a lane-private byte buffer with room for padding
VK_DEV void sha256_private(u32 out[8], unsigned char *buf, u32 len)
{
    const u32 nblk = (len + 9 + 63) / 64;
    for (u32 i = len; i < nblk * 64; i++) buf[i] = 0;
    buf[len] = 0x80;
    const u32 bits = len * 8;
    buf[nblk * 64 - 4] = (unsigned char)(bits >> 24);
    buf[nblk * 64 - 3] = (unsigned char)(bits >> 16);
    buf[nblk * 64 - 2] = (unsigned char)(bits >> 8);
    buf[nblk * 64 - 1] = (unsigned char)bits;
    u32 st[8] = VK_SHA256_IV;
#pragma unroll 1
    for (u32 b = 0; b < nblk; b++) {
        u32 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = load_be32(buf + 64 * b + 4 * i);
        sha256_compress(st, w);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = st[i];
}

__global__ void __launch_bounds__(VK_BLOCK) vk_electrum_kernel(const unsigned char *__restrict__ data, const u32 *__restrict__ offs, u64 n,
                                                                 u32 for_change, u32 count, u32 rounds, const uint4 *__restrict__ table,
                                                                 const vk_win_desc *__restrict__ wins, u32 nwin, u32 *__restrict__ keys_out,
                                                                 unsigned char *__restrict__ valid_out)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n) return;
    const u32 len = offs[idx + 1] - offs[idx];
    const unsigned char *seed = data + offs[idx];
    unsigned char buf[192];
    bool ok = len <= VK_ELECTRUM_MAX_SEED;
    if (ok) {
        for (u32 i = 0; i < len; i++) {
            const unsigned char c = seed[i];
            ok = ok && ((c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F')); // is_ascii_hexdigit
            buf[i] = c;
            buf[len + i] = c;
        }
    }
    u32 *kout = keys_out + idx * count * 8;
    unsigned char *vout = valid_out + idx * count;
    if (!ok) return; // outputs were zeroed by the host
    // ---- stretch
    u32 x[8];
    sha256_private(x, buf, 2 * len);
    // constant tail of the message x || seed || pad: words 8.. of block 1 and all of block 2
    const bool two = (32 + len + 9) > 64;
    for (u32 i = 0; i < 96; i++) buf[i] = 0;
    for (u32 i = 0; i < len; i++) buf[i] = seed[i];
    buf[len] = 0x80;
    {
        const u32 bits = (32 + len) * 8, end = two ? 96 : 32;
        buf[end - 2] = (unsigned char)(bits >> 8);
        buf[end - 1] = (unsigned char)bits;
    }
    u32 tw[24];
#pragma unroll
    for (int i = 0; i < 24; i++) tw[i] = load_be32(buf + 4 * i);
#pragma unroll 1
    for (u32 r = 1; r < rounds; r++) {
        u32 st[8] = VK_SHA256_IV;
        u32 w[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            w[i] = x[i];
            w[8 + i] = tw[i];
        }
        sha256_compress(st, w);
        if (two) {
#pragma unroll
            for (int i = 0; i < 16; i++) w[i] = tw[8 + i];
            sha256_compress(st, w);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = st[i];
    }
    // ---- master key: SecretKey::from_slice accepts 1 <= x < n only
    u32 m[8];
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] = x[7 - i];
    if (!scalar_is_key(m)) return; // 0 or >= n
    jac P;
    ecmult_table(P, m, table, wins, nwin);
    fe zi, ax, ay;
    fe_inv(zi, P.z);
    jac_to_affine(ax, ay, P, zi);
    // ---- child keys
#pragma unroll 1
    for (u32 i = 0; i < count; i++) {
        // "{i}:{c}:" || X || Y
        u32 pl = 0;
        {
            unsigned char dig[10];
            u32 v = i, nd = 0;
            do {
                dig[nd++] = (unsigned char)('0' + v % 10);
                v /= 10;
            } while (v);
            while (nd) buf[pl++] = dig[--nd];
            buf[pl++] = ':';
            buf[pl++] = for_change ? '1' : '0';
            buf[pl++] = ':';
        }
        for (int j = 0; j < 8; j++) {
            const u32 wx = FE_BE(ax, j), wy = FE_BE(ay, j);
            buf[pl + 4 * j] = (unsigned char)(wx >> 24);
            buf[pl + 4 * j + 1] = (unsigned char)(wx >> 16);
            buf[pl + 4 * j + 2] = (unsigned char)(wx >> 8);
            buf[pl + 4 * j + 3] = (unsigned char)wx;
            buf[pl + 32 + 4 * j] = (unsigned char)(wy >> 24);
            buf[pl + 32 + 4 * j + 1] = (unsigned char)(wy >> 16);
            buf[pl + 32 + 4 * j + 2] = (unsigned char)(wy >> 8);
            buf[pl + 32 + 4 * j + 3] = (unsigned char)wy;
        }
        u32 d1[8], seq[8];
        sha256_private(d1, buf, pl + 64);
        sha256_of_digest(seq, d1);
        // Scalar::from_be_bytes: seq must be < n (zero allowed)
        u32 s[8], c[8];
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] = seq[7 - j];
        if (!scalar_lt_n(s)) continue;
        // child = master + seq mod n; a zero result is rejected by add_tweak
        if (!scalar_add_mod_n(c, m, s)) continue;
#pragma unroll
        for (int j = 0; j < 8; j++) kout[i * 8 + j] = bswap32(c[7 - j]);
        vout[i] = 1;
    }
}
