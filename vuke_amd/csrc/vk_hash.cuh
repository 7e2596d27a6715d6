// vk_hash.cuh -- SHA-256 and RIPEMD-160 compression for gfx950 lanes, fully unrolled with a
// 16-word rolling message schedule in VGPRs (no scratch), specialised for the message shapes
// on the hot path: 33-byte and 65-byte SEC1 public keys, 32-byte digests, and <= 55-byte
// transform inputs.  Stands in for bitcoin_hashes (hash160 inside Address::p2pkh,
// src/derive.rs:112-118) and sha2 (src/transform/sha256.rs:16-26) of the reference.
#pragma once
#include "vk_field.cuh"

VK_DEV u32 rotr32(u32 x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
VK_DEV u32 rotl32(u32 x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }
VK_DEV u32 bswap32(u32 x) { return __builtin_bswap32(x); }

// v_bitop3_b32: any 3-input boolean function in one instruction.  The table is the function evaluated on
// a = 0xF0, b = 0xCC, c = 0xAA.
#define VK_BITOP3(a, b, c, tbl) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tbl))
VK_DEV u32 xor3(u32 a, u32 b, u32 c) { return VK_BITOP3(a, b, c, 0x96); }
VK_DEV u32 bool_ch(u32 e, u32 f, u32 g) { return VK_BITOP3(e, f, g, 0xCA); }   // (e & f) ^ (~e & g)
VK_DEV u32 bool_maj(u32 a, u32 b, u32 c) { return VK_BITOP3(a, b, c, 0xE8); }  // majority

__device__ __constant__ const u32 VK_K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

#define VK_SHA256_IV {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u}

// one compression; w[16] is consumed (used as the rolling schedule)
VK_DEV void sha256_compress(u32 st[8], u32 w[16])
{
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            u32 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            u32 s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
            u32 s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        u32 S1 = xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
        u32 ch = bool_ch(e, f, g);
        u32 t1 = h + S1 + ch + VK_K256[i] + w[i & 15];
        u32 S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
        u32 mj = bool_maj(a, b, c);
        u32 t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// SHA-256 of a single pre-padded block
VK_DEV void sha256_oneblock(u32 out[8], u32 w[16])
{
    u32 st[8] = VK_SHA256_IV;
    sha256_compress(st, w);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = st[i];
}

// SHA-256 of a 32-byte message given as 8 big-endian words
VK_DEV void sha256_of_digest(u32 out[8], const u32 in[8])
{
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = in[i];
    w[8] = 0x80000000u;
#pragma unroll
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = 256;
    sha256_oneblock(out, w);
}

// big-endian word j (0 = most significant) of a canonical field element
#define FE_BE(f, j) ((f).v[7 - (j)])

// SHA-256(0x02|0x03 || X): the 33-byte compressed SEC1 key, one block
VK_DEV void sha256_pub33(u32 out[8], const fe &x, u32 y_odd)
{
    u32 w[16];
    w[0] = ((0x02u | y_odd) << 24) | (FE_BE(x, 0) >> 8);
#pragma unroll
    for (int i = 1; i < 8; i++) w[i] = (FE_BE(x, i - 1) << 24) | (FE_BE(x, i) >> 8);
    w[8] = (FE_BE(x, 7) << 24) | 0x00800000u;
#pragma unroll
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = 33 * 8;
    sha256_oneblock(out, w);
}

// SHA-256(0x04 || X || Y): the 65-byte uncompressed SEC1 key, two blocks
VK_DEV void sha256_pub65(u32 out[8], const fe &x, const fe &y)
{
    u32 w[16];
    u32 st[8] = VK_SHA256_IV;
    w[0] = (0x04u << 24) | (FE_BE(x, 0) >> 8);
#pragma unroll
    for (int i = 1; i < 8; i++) w[i] = (FE_BE(x, i - 1) << 24) | (FE_BE(x, i) >> 8);
    w[8] = (FE_BE(x, 7) << 24) | (FE_BE(y, 0) >> 8);
#pragma unroll
    for (int i = 1; i < 8; i++) w[8 + i] = (FE_BE(y, i - 1) << 24) | (FE_BE(y, i) >> 8);
    sha256_compress(st, w);
    w[0] = (FE_BE(y, 7) << 24) | 0x00800000u;
#pragma unroll
    for (int i = 1; i < 15; i++) w[i] = 0;
    w[15] = 65 * 8;
    sha256_compress(st, w);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = st[i];
}

// ---- the second block of a 65-byte SEC1 key from an LDS table (the sweep kernel, round 5) ------------------------------------------
// That block is Y's last byte, 0x80, zeros and the bit length 520: its 48 schedule words W[16..63] depend on ONE byte.  A workgroup
// builds W[t] + K[t] for all 256 values once per launch (row b = 52 words: 48 used, rows 16-byte aligned and spread over the banks);
// the block then costs its 64 rounds and twelve 16-byte LDS reads, no schedule arithmetic: -450 VALU instructions per uncompressed
// key, +2.1 % on config 3 same box (profiles/r05_sha_tail_lds_experiment.txt).  The table is DYNAMIC LDS (53 KB), requested only by
// launches that hash uncompressed keys: a compressed-only sweep (config 2) runs with the 24.8 KB it always had.
#define VK_TAIL_ROW4 13 /* 16-byte quads per row */
#define VK_TAIL_LDS_BYTES (256 * VK_TAIL_ROW4 * 16)
typedef u32 vk_u32x4 __attribute__((ext_vector_type(4)));
typedef vk_u32x4 __attribute__((address_space(3))) *vk_lds_u4p;
VK_DEV void sha256_tail65_build_row(vk_lds_u4p table, u32 b)
{
    u32 w[16];
    w[0] = (b << 24) | 0x00800000u;
#pragma unroll
    for (int i = 1; i < 15; i++) w[i] = 0;
    w[15] = 65 * 8;
    u32 q[4];
#pragma unroll
    for (int i = 16; i < 64; i++) {
        u32 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
        u32 s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
        u32 s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
        w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        q[i & 3] = w[i & 15] + VK_K256[i];
        if ((i & 3) == 3) table[b * VK_TAIL_ROW4 + (i - 16) / 4] = (vk_u32x4){q[0], q[1], q[2], q[3]};
    }
}
VK_DEV void sha256_compress_tail65(u32 st[8], u32 ylast, vk_lds_u4p table)
{
    const vk_lds_u4p row = table + ylast * VK_TAIL_ROW4;
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    vk_u32x4 q = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 64; i++) {
        u32 kw;
        if (i == 0) kw = ((ylast << 24) | 0x00800000u) + VK_K256[0];
        else if (i < 15) kw = VK_K256[i];
        else if (i == 15) kw = VK_K256[15] + 65u * 8u;
        else {
            if ((i & 3) == 0) q = row[(i - 16) / 4];
            kw = (i & 3) == 0 ? q.x : (i & 3) == 1 ? q.y : (i & 3) == 2 ? q.z : q.w;
        }
        u32 S1 = xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
        u32 ch = bool_ch(e, f, g);
        u32 t1 = h + S1 + ch + kw;
        u32 S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
        u32 mj = bool_maj(a, b, c);
        u32 t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
// SHA-256(0x04 || X || Y) with the second block from the table
VK_DEV void sha256_pub65_lds(u32 out[8], const fe &x, const fe &y, vk_lds_u4p table)
{
    u32 w[16];
    u32 st[8] = VK_SHA256_IV;
    w[0] = (0x04u << 24) | (FE_BE(x, 0) >> 8);
#pragma unroll
    for (int i = 1; i < 8; i++) w[i] = (FE_BE(x, i - 1) << 24) | (FE_BE(x, i) >> 8);
    w[8] = (FE_BE(x, 7) << 24) | (FE_BE(y, 0) >> 8);
#pragma unroll
    for (int i = 1; i < 8; i++) w[8 + i] = (FE_BE(y, i - 1) << 24) | (FE_BE(y, i) >> 8);
    sha256_compress(st, w);
    sha256_compress_tail65(st, FE_BE(y, 7) & 0xFFu, table);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = st[i];
}

// ------------------------------------------------------------------ RIPEMD-160
VK_DEV u32 rmd_f1(u32 x, u32 y, u32 z) { return xor3(x, y, z); }
VK_DEV u32 rmd_f2(u32 x, u32 y, u32 z) { return VK_BITOP3(x, y, z, 0xCA); } // (x&y)|(~x&z)
VK_DEV u32 rmd_f3(u32 x, u32 y, u32 z) { return VK_BITOP3(x, y, z, 0x59); } // (x|~y)^z
VK_DEV u32 rmd_f4(u32 x, u32 y, u32 z) { return VK_BITOP3(x, y, z, 0xE4); } // (x&z)|(y&~z)
VK_DEV u32 rmd_f5(u32 x, u32 y, u32 z) { return VK_BITOP3(x, y, z, 0x2D); } // x^(y|~z)

// Message-word order and rotate amounts of the two lines (RIPEMD-160 paper, tables r, r', s, s').
// The loops below are fully unrolled, so every table lookup folds to an immediate.
__device__ __constant__ const unsigned char VK_RMD_RL[80] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 7, 4, 13, 1, 10, 6, 15, 3, 12, 0, 9, 5, 2, 14, 11, 8,
    3, 10, 14, 4, 9, 15, 8, 1, 2, 7, 0, 6, 13, 11, 5, 12, 1, 9, 11, 10, 0, 8, 12, 4, 13, 3, 7, 15, 14, 5, 6, 2,
    4, 0, 5, 9, 7, 12, 2, 10, 14, 1, 3, 8, 11, 6, 15, 13};
__device__ __constant__ const unsigned char VK_RMD_RR[80] = {
    5, 14, 7, 0, 9, 2, 11, 4, 13, 6, 15, 8, 1, 10, 3, 12, 6, 11, 3, 7, 0, 13, 5, 10, 14, 15, 8, 12, 4, 9, 1, 2,
    15, 5, 1, 3, 7, 14, 6, 9, 11, 8, 12, 2, 10, 0, 4, 13, 8, 6, 4, 1, 3, 11, 15, 0, 5, 12, 2, 13, 9, 7, 10, 14,
    12, 15, 10, 4, 1, 5, 8, 7, 6, 2, 13, 14, 0, 3, 9, 11};
__device__ __constant__ const unsigned char VK_RMD_SL[80] = {
    11, 14, 15, 12, 5, 8, 7, 9, 11, 13, 14, 15, 6, 7, 9, 8, 7, 6, 8, 13, 11, 9, 7, 15, 7, 12, 15, 9, 11, 7, 13, 12,
    11, 13, 6, 7, 14, 9, 13, 15, 14, 8, 13, 6, 5, 12, 7, 5, 11, 12, 14, 15, 14, 15, 9, 8, 9, 14, 5, 6, 8, 6, 5, 12,
    9, 15, 5, 11, 6, 8, 13, 12, 5, 12, 13, 14, 11, 8, 5, 6};
__device__ __constant__ const unsigned char VK_RMD_SR[80] = {
    8, 9, 9, 11, 13, 15, 15, 5, 7, 7, 8, 11, 14, 14, 12, 6, 9, 13, 15, 7, 12, 8, 9, 11, 7, 7, 12, 7, 6, 15, 13, 11,
    9, 7, 15, 11, 8, 6, 6, 14, 12, 13, 5, 14, 13, 13, 7, 5, 15, 5, 8, 11, 14, 14, 6, 14, 6, 9, 12, 9, 12, 5, 15, 8,
    8, 5, 12, 9, 12, 5, 14, 6, 8, 13, 6, 5, 15, 13, 11, 11};

VK_DEV u32 rmd_f(int round, u32 x, u32 y, u32 z)
{
    switch (round) {
    case 0: return rmd_f1(x, y, z);
    case 1: return rmd_f2(x, y, z);
    case 2: return rmd_f3(x, y, z);
    case 3: return rmd_f4(x, y, z);
    default: return rmd_f5(x, y, z);
    }
}

// RIPEMD-160 of a 32-byte message: in[8] = SHA-256 state words (big-endian words of the digest).
// out[5] = state words; the 20 digest bytes are their little-endian serialisation.
VK_DEV void ripemd160_of_digest(u32 out[5], const u32 in[8])
{
    u32 x[16];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = bswap32(in[i]);
    x[8] = 0x80u;
#pragma unroll
    for (int i = 9; i < 14; i++) x[i] = 0;
    x[14] = 256;
    x[15] = 0;
    const u32 KL[5] = {0x00000000u, 0x5a827999u, 0x6ed9eba1u, 0x8f1bbcdcu, 0xa953fd4eu};
    const u32 KR[5] = {0x50a28be6u, 0x5c4dd124u, 0x6d703ef3u, 0x7a6d76e9u, 0x00000000u};
    const u32 h0 = 0x67452301u, h1 = 0xefcdab89u, h2 = 0x98badcfeu, h3 = 0x10325476u, h4 = 0xc3d2e1f0u;
    u32 al = h0, bl = h1, cl = h2, dl = h3, el = h4;
    u32 ar = h0, br = h1, cr = h2, dr = h3, er = h4;
#pragma unroll
    for (int j = 0; j < 80; j++) {
        u32 t = rotl32(al + rmd_f(j >> 4, bl, cl, dl) + x[VK_RMD_RL[j]] + KL[j >> 4], VK_RMD_SL[j]) + el;
        al = el; el = dl; dl = rotl32(cl, 10); cl = bl; bl = t;
        t = rotl32(ar + rmd_f(4 - (j >> 4), br, cr, dr) + x[VK_RMD_RR[j]] + KR[j >> 4], VK_RMD_SR[j]) + er;
        ar = er; er = dr; dr = rotl32(cr, 10); cr = br; br = t;
    }
    out[0] = h1 + cl + dr;
    out[1] = h2 + dl + er;
    out[2] = h3 + el + ar;
    out[3] = h4 + al + br;
    out[4] = h0 + bl + cr;
}

// hash160 words (RIPEMD-160 state) of the compressed / uncompressed SEC1 encodings
VK_DEV void hash160_pub33(u32 out[5], const fe &x, u32 y_odd)
{
    u32 d[8];
    sha256_pub33(d, x, y_odd);
    ripemd160_of_digest(out, d);
}
VK_DEV void hash160_pub65(u32 out[5], const fe &x, const fe &y)
{
    u32 d[8];
    sha256_pub65(d, x, y);
    ripemd160_of_digest(out, d);
}
VK_DEV void hash160_pub65_lds(u32 out[5], const fe &x, const fe &y, vk_lds_u4p table)
{
    u32 d[8];
    sha256_pub65_lds(d, x, y, table);
    ripemd160_of_digest(out, d);
}

// ---- MD5 (RFC 1321), for the md5 transform (src/transform/md5.rs:13-21; md-5 crate) -------------------------
__device__ __constant__ const u32 VK_MD5_K[64] = {
    0xd76aa478u, 0xe8c7b756u, 0x242070dbu, 0xc1bdceeeu, 0xf57c0fafu, 0x4787c62au, 0xa8304613u, 0xfd469501u,
    0x698098d8u, 0x8b44f7afu, 0xffff5bb1u, 0x895cd7beu, 0x6b901122u, 0xfd987193u, 0xa679438eu, 0x49b40821u,
    0xf61e2562u, 0xc040b340u, 0x265e5a51u, 0xe9b6c7aau, 0xd62f105du, 0x02441453u, 0xd8a1e681u, 0xe7d3fbc8u,
    0x21e1cde6u, 0xc33707d6u, 0xf4d50d87u, 0x455a14edu, 0xa9e3e905u, 0xfcefa3f8u, 0x676f02d9u, 0x8d2a4c8au,
    0xfffa3942u, 0x8771f681u, 0x6d9d6122u, 0xfde5380cu, 0xa4beea44u, 0x4bdecfa9u, 0xf6bb4b60u, 0xbebfbc70u,
    0x289b7ec6u, 0xeaa127fau, 0xd4ef3085u, 0x04881d05u, 0xd9d4d039u, 0xe6db99e5u, 0x1fa27cf8u, 0xc4ac5665u,
    0xf4292244u, 0x432aff97u, 0xab9423a7u, 0xfc93a039u, 0x655b59c3u, 0x8f0ccc92u, 0xffeff47du, 0x85845dd1u,
    0x6fa87e4fu, 0xfe2ce6e0u, 0xa3014314u, 0x4e0811a1u, 0xf7537e82u, 0xbd3af235u, 0x2ad7d2bbu, 0xeb86d391u};

#define VK_MD5_IV {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u}

// one compression; m[16] = the block as little-endian words
VK_DEV void md5_compress(u32 st[4], const u32 m[16])
{
    u32 a = st[0], b = st[1], c = st[2], d = st[3];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        u32 f;
        int g, r;
        if (i < 16) {
            f = VK_BITOP3(b, c, d, 0xCA); // (b & c) | (~b & d)
            g = i;
            r = (i & 3) == 0 ? 7 : (i & 3) == 1 ? 12 : (i & 3) == 2 ? 17 : 22;
        } else if (i < 32) {
            f = VK_BITOP3(d, b, c, 0xCA); // (d & b) | (~d & c)
            g = (5 * i + 1) & 15;
            r = (i & 3) == 0 ? 5 : (i & 3) == 1 ? 9 : (i & 3) == 2 ? 14 : 20;
        } else if (i < 48) {
            f = xor3(b, c, d);
            g = (3 * i + 5) & 15;
            r = (i & 3) == 0 ? 4 : (i & 3) == 1 ? 11 : (i & 3) == 2 ? 16 : 23;
        } else {
            f = VK_BITOP3(b, c, d, 0x39); // c ^ (b | ~d)
            g = (7 * i) & 15;
            r = (i & 3) == 0 ? 6 : (i & 3) == 1 ? 10 : (i & 3) == 2 ? 15 : 21;
        }
        const u32 t = a + f + VK_MD5_K[i] + m[g];
        a = d;
        d = c;
        c = b;
        b = b + rotl32(t, r);
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d;
}

// ---- streaming forms: a byte string in global memory followed by <= 12 suffix bytes held in registers -----
// (wordlist / stdin inputs of the hash transforms: passphrases of any length)
struct vk_bytes {
    const unsigned char *p;
    u32 len;
    u32 suf[3]; // suffix bytes, big-endian packed
    u32 suf_len;
};
VK_DEV u32 bytes_at(const vk_bytes &s, u32 i)
{
    if (i < s.len) return s.p[i];
    const u32 j = i - s.len;
    const u32 w = j < 4 ? s.suf[0] : j < 8 ? s.suf[1] : s.suf[2];
    return (w >> (24 - 8 * (j & 3))) & 0xffu;
}
// message byte q of the padded stream: data, then 0x80, then zeros (the length words are set by the caller)
VK_DEV u32 padded_at(const vk_bytes &s, u32 total, u32 q) { return q < total ? bytes_at(s, q) : (q == total ? 0x80u : 0u); }

VK_DEV void sha256_stream(u32 out[8], const vk_bytes &s)
{
    const u32 total = s.len + s.suf_len, nblk = (total + 9 + 63) / 64;
    u32 st[8] = VK_SHA256_IV;
#pragma unroll 1
    for (u32 blk = 0; blk < nblk; blk++) {
        u32 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const u32 q = blk * 64 + 4 * (u32)i;
            w[i] = q >= total + 1 ? 0u
                                  : (padded_at(s, total, q) << 24) | (padded_at(s, total, q + 1) << 16) | (padded_at(s, total, q + 2) << 8) |
                                        padded_at(s, total, q + 3);
        }
        if (blk == nblk - 1) {
            w[14] = total >> 29;
            w[15] = total << 3;
        }
        sha256_compress(st, w);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = st[i];
}

// MD5 digest as four little-endian state words (digest byte 4j+k = (out[j] >> 8k) & 0xff)
VK_DEV void md5_stream(u32 out[4], const vk_bytes &s)
{
    const u32 total = s.len + s.suf_len, nblk = (total + 9 + 63) / 64;
    u32 st[4] = VK_MD5_IV;
#pragma unroll 1
    for (u32 blk = 0; blk < nblk; blk++) {
        u32 m[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const u32 q = blk * 64 + 4 * (u32)i;
            m[i] = q >= total + 1 ? 0u
                                  : padded_at(s, total, q) | (padded_at(s, total, q + 1) << 8) | (padded_at(s, total, q + 2) << 16) |
                                        (padded_at(s, total, q + 3) << 24);
        }
        if (blk == nblk - 1) {
            m[14] = total << 3;
            m[15] = total >> 29;
        }
        md5_compress(st, m);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = st[i];
}
