// vk_multibit.cuh -- MultiBit HD "seed as entropy" key chain on gfx950, one mnemonic per lane.
//
// Reference behaviour (src/multibit.rs:87-350, src/transform/multibit.rs:45-61):
//   words = mnemonic.split_whitespace(); 12/15/18/21/24 of them, all in the BIP-39 list, else skip;
//   seed1 = PBKDF2-HMAC-SHA512(password = the mnemonic string as given, salt = "mnemonic", 2048);
//   48 words from the 512 bits of seed1 + 16 checksum bits (SHA-256), joined by single spaces;
//   seed2 = PBKDF2-HMAC-SHA512(password = that 48-word string, salt = "mnemonic", 2048);
//   (k, c) = HMAC-SHA512("Bitcoin seed", seed2); m/0' hardened, /0 and /i normal CKDpriv with the
//   reference's error rules (parent must be a valid key, IL < n, child != 0); i = 0..count-1.
// The BIP-39 list is supplied by the caller (vk_bip39_wordlist_set): 2048 words of <= 8 bytes, zero
// padded to 8.  The two PBKDF2 loops (4 x 2047 SHA-512 compressions) are the hot part and run on
// register-resident state; everything else goes through a small lane-private byte buffer.
#pragma once
#include "vk_ec.cuh"
#include "vk_hash.cuh"
#include "vk_scalar.cuh"
#include "vk_electrum.cuh" // sha256_private, load_be32

#define VK_MULTIBIT_MAX_MNEMONIC 256 /* bytes */

__device__ __constant__ const u64 VK_K512[80] = {
    0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull,
    0x59f111f1b605d019ull, 0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull, 0xd807aa98a3030242ull, 0x12835b0145706fbeull,
    0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull, 0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull,
    0xc19bf174cf692694ull, 0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull, 0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull,
    0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull, 0x983e5152ee66dfabull,
    0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull,
    0x06ca6351e003826full, 0x142929670a0e6e70ull, 0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull,
    0x53380d139d95b3dfull, 0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull,
    0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull, 0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull,
    0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull, 0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull,
    0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull, 0x5b9cca4f7763e373ull,
    0x682e6ff3d6b2b8a3ull, 0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
    0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull, 0xca273eceea26619cull,
    0xd186b8c721c0c207ull, 0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull,
    0x113f9804bef90daeull, 0x1b710b35131c471bull, 0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull,
    0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull, 0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};

#define VK_SHA512_IV {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull, \
                      0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull}

// 64-bit rotate as two v_alignbit_b32 (the compiler's own lowering of (x >> n) | (x << (64 - n)) is
// v_lshrrev_b64 + v_lshlrev_b64 + 2 x v_or: five issue slots instead of two)
template <int N>
VK_DEV u64 rotr64(u64 x)
{
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    u32 rl, rh;
    if constexpr (N < 32) {
        rl = __builtin_amdgcn_alignbit(hi, lo, N);
        rh = __builtin_amdgcn_alignbit(lo, hi, N);
    } else if constexpr (N == 32) {
        rl = hi;
        rh = lo;
    } else {
        rl = __builtin_amdgcn_alignbit(lo, hi, N - 32);
        rh = __builtin_amdgcn_alignbit(hi, lo, N - 32);
    }
    return ((u64)rh << 32) | rl;
}

VK_DEV u64 xor3_64(u64 a, u64 b, u64 c)
{
    return ((u64)xor3((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32)) << 32) | xor3((u32)a, (u32)b, (u32)c);
}
VK_DEV u64 ch64(u64 e, u64 f, u64 g)
{
    return ((u64)bool_ch((u32)(e >> 32), (u32)(f >> 32), (u32)(g >> 32)) << 32) | bool_ch((u32)e, (u32)f, (u32)g);
}
VK_DEV u64 maj64(u64 a, u64 b, u64 c)
{
    return ((u64)bool_maj((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32)) << 32) | bool_maj((u32)a, (u32)b, (u32)c);
}

// one SHA-512 compression; w[16] is consumed as the rolling schedule
VK_DEV void sha512_compress(u64 st[8], u64 w[16])
{
    u64 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 80; i++) {
        if (i >= 16) {
            u64 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            u64 s0 = xor3_64(rotr64<1>(w15), rotr64<8>(w15), w15 >> 7);
            u64 s1 = xor3_64(rotr64<19>(w2), rotr64<61>(w2), w2 >> 6);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        u64 S1 = xor3_64(rotr64<14>(e), rotr64<18>(e), rotr64<41>(e));
        u64 ch = ch64(e, f, g);
        u64 t1 = h + S1 + ch + VK_K512[i] + w[i & 15];
        u64 S0 = xor3_64(rotr64<28>(a), rotr64<34>(a), rotr64<39>(a));
        u64 mj = maj64(a, b, c);
        u64 t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

VK_DEV u64 load_be64(const unsigned char *p)
{
    u64 v = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
    return v;
}
VK_DEV void store_be64(unsigned char *p, u64 v)
{
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = (unsigned char)(v >> (56 - 8 * i));
}

// st = SHA-512 state after absorbing buf[0..len) continuing from `st` with `prefix` bytes already
// absorbed (a multiple of 128); writes the padding into buf (room for len + 17 rounded up to 128)
__device__ __noinline__ void sha512_absorb_final(u64 st[8], unsigned char *buf, u32 len, u32 prefix)
{
    const u32 nblk = (len + 17 + 127) / 128;
    for (u32 i = len; i < nblk * 128; i++) buf[i] = 0;
    buf[len] = 0x80;
    const u32 bits = (prefix + len) * 8;
    buf[nblk * 128 - 4] = (unsigned char)(bits >> 24);
    buf[nblk * 128 - 3] = (unsigned char)(bits >> 16);
    buf[nblk * 128 - 2] = (unsigned char)(bits >> 8);
    buf[nblk * 128 - 1] = (unsigned char)bits;
#pragma unroll 1
    for (u32 b = 0; b < nblk; b++) {
        u64 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = load_be64(buf + 128 * b + 8 * i);
        sha512_compress(st, w);
    }
}

// HMAC-SHA512 key schedule: states after the ipad / opad block.  key[0..klen), klen <= 128.
__device__ __noinline__ void hmac512_init(u64 ist[8], u64 ost[8], const unsigned char *key, u32 klen)
{
    const u64 iv[8] = VK_SHA512_IV;
    u64 w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) ist[i] = ost[i] = iv[i];
    for (int pass = 0; pass < 2; pass++) {
        const unsigned char pad = pass ? 0x5c : 0x36;
#pragma unroll 1
        for (int i = 0; i < 16; i++) {
            u64 v = 0;
            for (int j = 0; j < 8; j++) {
                const u32 pos = 8 * i + j;
                v = (v << 8) | (unsigned char)((pos < klen ? key[pos] : 0) ^ pad);
            }
            w[i] = v;
        }
        sha512_compress(pass ? ost : ist, w);
    }
}

// out = HMAC(data[0..dlen)) given the key schedule; data buffer needs room for padding (dlen + 17 -> x128)
__device__ __noinline__ void hmac512_final(u64 out[8], const u64 ist[8], const u64 ost[8], unsigned char *data, u32 dlen)
{
    u64 st[8];
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = ist[i];
    sha512_absorb_final(st, data, dlen, 128);
    unsigned char inner[128];
    for (int i = 0; i < 8; i++) store_be64(inner + 8 * i, st[i]);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = ost[i];
    sha512_absorb_final(out, inner, 64, 128);
}

// T = U1 ^ ... ^ U2048 with U1 = HMAC(salt || INT(1)) already in u: the hot loop, all in registers
__device__ __noinline__ void pbkdf2_tail_2048(u64 t[8], const u64 ist[8], const u64 ost[8], const u64 u1[8])
{
    u64 u[8], is_[8], os_[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u[i] = t[i] = u1[i];
        is_[i] = ist[i];
        os_[i] = ost[i];
    }
#pragma unroll 1
    for (u32 it = 1; it < 2048; it++) {
        u64 st[8], w[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            st[i] = is_[i];
            w[i] = u[i];
        }
        w[8] = 0x8000000000000000ull;
#pragma unroll
        for (int i = 9; i < 15; i++) w[i] = 0;
        w[15] = (128 + 64) * 8;
        sha512_compress(st, w);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            w[i] = st[i];
            st[i] = os_[i];
        }
        w[8] = 0x8000000000000000ull;
#pragma unroll
        for (int i = 9; i < 15; i++) w[i] = 0;
        w[15] = (128 + 64) * 8;
        sha512_compress(st, w);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            u[i] = st[i];
            t[i] ^= st[i];
        }
    }
}

// BIP-39 passphrase appended to the salt "mnemonic" (the transform passes none; the multibit-hd analyzer may): at most
// VK_MULTIBIT_MAX_PASS bytes so that salt || INT(1) stays inside one SHA-512 block
#define VK_MULTIBIT_MAX_PASS 96
struct vk_mb_pass {
    u32 len;
    unsigned char b[VK_MULTIBIT_MAX_PASS];
};

// PBKDF2-HMAC-SHA512(password, "mnemonic" || passphrase, 2048, 64 bytes); pw may be overwritten (padding).
// plen <= VK_BITIMAGE_MAX_PASS (salt || INT(1) may span three SHA-512 blocks).
#define VK_BITIMAGE_MAX_PASS 256
__device__ __noinline__ void pbkdf2_mnemonic_salt(u64 seed[8], unsigned char *pw, u32 pwlen, const unsigned char *pass, u32 pl)
{
    u64 ist[8], ost[8];
    unsigned char key[128];
    u32 klen = pwlen;
    if (pwlen > 128) { // RFC 2104: long keys are hashed first
        u64 st[8] = VK_SHA512_IV;
        sha512_absorb_final(st, pw, pwlen, 0);
        for (int i = 0; i < 8; i++) store_be64(key + 8 * i, st[i]);
        klen = 64;
    } else {
        for (u32 i = 0; i < pwlen; i++) key[i] = pw[i];
    }
    hmac512_init(ist, ost, key, klen);
    unsigned char salt[384];
    const char s[8] = {'m', 'n', 'e', 'm', 'o', 'n', 'i', 'c'};
    for (int i = 0; i < 8; i++) salt[i] = (unsigned char)s[i];
    for (u32 i = 0; i < pl; i++) salt[8 + i] = pass[i];
    salt[8 + pl] = 0; salt[9 + pl] = 0; salt[10 + pl] = 0; salt[11 + pl] = 1;
    u64 u1[8];
    hmac512_final(u1, ist, ost, salt, 12 + pl);
    pbkdf2_tail_2048(seed, ist, ost, u1);
}
VK_DEV void pbkdf2_mnemonic(u64 seed[8], unsigned char *pw, u32 pwlen, const vk_mb_pass &pass)
{
    unsigned char pb[VK_MULTIBIT_MAX_PASS];
    for (u32 i = 0; i < pass.len; i++) pb[i] = pass.b[i];
    pbkdf2_mnemonic_salt(seed, pw, pwlen, pb, pass.len);
}

// one CKDpriv step in the reference's spelling; data33 = 0x00 || k (hardened) or serP(K) (normal)
VK_DEV bool ckd_step(u32 ck[8], unsigned char cc[32], const u32 k[8], const unsigned char c[32], const unsigned char data33[33], u32 index)
{
    u64 ist[8], ost[8], I[8];
    hmac512_init(ist, ost, c, 32);
    unsigned char data[128];
    for (int i = 0; i < 33; i++) data[i] = data33[i];
    data[33] = (unsigned char)(index >> 24); data[34] = (unsigned char)(index >> 16); data[35] = (unsigned char)(index >> 8); data[36] = (unsigned char)index;
    hmac512_final(I, ist, ost, data, 37);
    u32 il[8];
#pragma unroll
    for (int i = 0; i < 4; i++) { // IL = first 32 bytes, as little-endian limbs
        il[7 - 2 * i] = (u32)(I[i] >> 32);
        il[6 - 2 * i] = (u32)I[i];
    }
    for (int i = 0; i < 4; i++) store_be64(cc + 8 * i, I[4 + i]);
    if (!scalar_is_key(k)) return false;   // SecretKey::from_slice(parent)
    if (!scalar_lt_n(il)) return false;    // Scalar::from_be_bytes(IL)
    return scalar_add_mod_n(ck, k, il);    // add_tweak: zero result is an error
}

// serP(k*G): 33-byte compressed SEC1 encoding
VK_DEV void ser_pubkey(unsigned char out[33], const u32 k[8], const uint4 *table, const vk_win_desc *wins, u32 nwin)
{
    jac P;
    ecmult_table(P, k, table, wins, nwin);
    fe zi, x, y;
    fe_inv(zi, P.z);
    jac_to_affine(x, y, P, zi);
    out[0] = (unsigned char)(0x02 | (y.v[0] & 1u));
    for (int j = 0; j < 8; j++) {
        const u32 w = FE_BE(x, j);
        out[1 + 4 * j] = (unsigned char)(w >> 24);
        out[2 + 4 * j] = (unsigned char)(w >> 16);
        out[3 + 4 * j] = (unsigned char)(w >> 8);
        out[4 + 4 * j] = (unsigned char)w;
    }
}

VK_DEV bool mb_is_ws(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c; }

__global__ void __launch_bounds__(VK_BLOCK) vk_multibit_kernel(const unsigned char *__restrict__ data, const u32 *__restrict__ offs, u64 n, u32 count,
                                                                 const u64 *__restrict__ wordlist /* 2048 x 8 bytes, zero padded, as LE u64 */,
                                                                 const uint4 *__restrict__ table, const vk_win_desc *__restrict__ wins, u32 nwin,
                                                                 u32 *__restrict__ keys_out, unsigned char *__restrict__ valid_out, const vk_mb_pass pass)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n) return;
    const u32 len = offs[idx + 1] - offs[idx];
    const unsigned char *mn = data + offs[idx];
    if (len > VK_MULTIBIT_MAX_MNEMONIC) return;
    // ---- split_whitespace + validate_mnemonic
    u32 nwords = 0, i = 0;
    bool ok = true;
    while (i < len && ok) {
        while (i < len && mb_is_ws(mn[i])) i++;
        const u32 s = i;
        while (i < len && !mb_is_ws(mn[i])) i++;
        if (i > s) {
            nwords++;
            if (i - s > 8) {
                ok = false;
                break;
            }
            u64 packed = 0;
            for (u32 j = s; j < i; j++) packed |= (u64)mn[j] << (8 * (j - s));
            bool found = false;
#pragma unroll 1
            for (u32 w = 0; w < 2048 && !found; w++) found = wordlist[w] == packed;
            ok = found;
        }
    }
    if (!ok || !(nwords == 12 || nwords == 15 || nwords == 18 || nwords == 21 || nwords == 24)) return;
    // ---- seed1 = PBKDF2(mnemonic as given)
    unsigned char buf[640];
    for (u32 j = 0; j < len; j++) buf[j] = mn[j];
    u64 seed[8];
    pbkdf2_mnemonic(seed, buf, len, pass);
    // ---- 512 + 16 checksum bits -> 48 words joined by spaces
    unsigned char bits[66];
    for (int j = 0; j < 8; j++) store_be64(bits + 8 * j, seed[j]);
    {
        unsigned char tmp[192];
        for (int j = 0; j < 64; j++) tmp[j] = bits[j];
        u32 h[8];
        sha256_private(h, tmp, 64);
        bits[64] = (unsigned char)(h[0] >> 24);
        bits[65] = (unsigned char)(h[0] >> 16);
    }
    u32 bl = 0;
#pragma unroll 1
    for (u32 w = 0; w < 48; w++) {
        u32 wi = 0;
        for (u32 j = 0; j < 11; j++) {
            const u32 b = w * 11 + j;
            wi = (wi << 1) | ((bits[b >> 3] >> (7 - (b & 7))) & 1u);
        }
        if (w) buf[bl++] = ' ';
        u64 packed = wordlist[wi];
        while (packed & 0xff) {
            buf[bl++] = (unsigned char)packed;
            packed >>= 8;
        }
    }
    // ---- seed2 = PBKDF2(48-word string); (k, c) = HMAC("Bitcoin seed", seed2)
    pbkdf2_mnemonic(seed, buf, bl, pass);
    u64 ist[8], ost[8], I[8];
    {
        const char bs[12] = {'B', 'i', 't', 'c', 'o', 'i', 'n', ' ', 's', 'e', 'e', 'd'};
        unsigned char key[12];
        for (int j = 0; j < 12; j++) key[j] = (unsigned char)bs[j];
        hmac512_init(ist, ost, key, 12);
        unsigned char dat[256];
        for (int j = 0; j < 8; j++) store_be64(dat + 8 * j, seed[j]);
        hmac512_final(I, ist, ost, dat, 64);
    }
    u32 k[8];
    unsigned char c[32], d33[33];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        k[7 - 2 * j] = (u32)(I[j] >> 32);
        k[6 - 2 * j] = (u32)I[j];
    }
    for (int j = 0; j < 4; j++) store_be64(c + 8 * j, I[4 + j]);
    // ---- m/0' (hardened): data = 0x00 || k
    u32 k0[8], k1[8];
    unsigned char c0[32], c1[32];
    d33[0] = 0;
    for (int j = 0; j < 8; j++) {
        const u32 w = k[7 - j];
        d33[1 + 4 * j] = (unsigned char)(w >> 24); d33[2 + 4 * j] = (unsigned char)(w >> 16);
        d33[3 + 4 * j] = (unsigned char)(w >> 8);  d33[4 + 4 * j] = (unsigned char)w;
    }
    if (!ckd_step(k0, c0, k, c, d33, 0x80000000u)) return;
    // ---- m/0'/0 (normal): data = serP(K)
    if (!scalar_is_key(k0)) return; // private_to_public
    ser_pubkey(d33, k0, table, wins, nwin);
    if (!ckd_step(k1, c1, k0, c0, d33, 0)) return;
    // ---- m/0'/0/i
    if (!scalar_is_key(k1)) return;
    ser_pubkey(d33, k1, table, wins, nwin);
    u32 *kout = keys_out + idx * count * 8;
    unsigned char *vout = valid_out + idx * count;
#pragma unroll 1
    for (u32 j = 0; j < count; j++) {
        u32 ck[8];
        unsigned char cc[32];
        if (!ckd_step(ck, cc, k1, c1, d33, j)) continue;
#pragma unroll
        for (int q = 0; q < 8; q++) kout[j * 8 + q] = bswap32(ck[7 - q]);
        vout[j] = 1;
    }
}

// ------------------------------------------------------------------------------------------------------ bitimage
// BitimageDeriver::from_file_bytes after the file has been reduced to its 32 bytes of entropy, then derive_path for
// derive_count paths that differ in the last index (reference src/bitimage.rs:38-74, src/transform/bitimage.rs:71-97):
// entropy -> 24-word mnemonic (8 checksum bits) -> PBKDF2 seed -> HMAC("Bitcoin seed") -> CKDpriv along the path.
// One lane per (file, passphrase); the parent of the last level is derived once.
#define VK_BIP32_MAX_DEPTH 16
struct vk_bip32_path {
    u32 depth;
    u32 comp[VK_BIP32_MAX_DEPTH]; // bit 31 = hardened
};

VK_DEV void ckd_parent_data(unsigned char d33[33], const u32 k[8], bool hardened, const uint4 *table, const vk_win_desc *wins, u32 nwin)
{
    if (hardened) { // 0x00 || k
        d33[0] = 0;
        for (int j = 0; j < 8; j++) {
            const u32 w = k[7 - j];
            d33[1 + 4 * j] = (unsigned char)(w >> 24); d33[2 + 4 * j] = (unsigned char)(w >> 16);
            d33[3 + 4 * j] = (unsigned char)(w >> 8);  d33[4 + 4 * j] = (unsigned char)w;
        }
    } else { // serP(K)
        ser_pubkey(d33, k, table, wins, nwin);
    }
}

__global__ void __launch_bounds__(VK_BLOCK) vk_bitimage_kernel(const unsigned char *__restrict__ entropy, u64 n_items, const unsigned char *__restrict__ pass_data,
                                                                 const u32 *__restrict__ pass_offs, u32 n_pass, const vk_bip32_path path, u32 count,
                                                                 const u64 *__restrict__ wordlist, const uint4 *__restrict__ table,
                                                                 const vk_win_desc *__restrict__ wins, u32 nwin, u32 *__restrict__ keys_out,
                                                                 unsigned char *__restrict__ valid_out)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n_items) return;
    const u64 file = idx / n_pass;
    const u32 pj = (u32)(idx % n_pass);
    // ---- entropy_to_mnemonic: 256 + 8 bits -> 24 words joined by single spaces
    unsigned char bits[33];
    for (int j = 0; j < 32; j++) bits[j] = entropy[file * 32 + j];
    {
        unsigned char tmp[128];
        for (int j = 0; j < 32; j++) tmp[j] = bits[j];
        u32 h[8];
        sha256_private(h, tmp, 32);
        bits[32] = (unsigned char)(h[0] >> 24);
    }
    unsigned char buf[640];
    u32 bl = 0;
#pragma unroll 1
    for (u32 w = 0; w < 24; w++) {
        u32 wi = 0;
        for (u32 j = 0; j < 11; j++) {
            const u32 b = w * 11 + j;
            wi = (wi << 1) | ((bits[b >> 3] >> (7 - (b & 7))) & 1u);
        }
        if (w) buf[bl++] = ' ';
        u64 packed = wordlist[wi];
        while (packed & 0xff) {
            buf[bl++] = (unsigned char)packed;
            packed >>= 8;
        }
    }
    // ---- seed = PBKDF2(mnemonic, "mnemonic" || passphrase); (k, c) = HMAC("Bitcoin seed", seed)
    u64 seed[8];
    {
        unsigned char pb[VK_BITIMAGE_MAX_PASS];
        const u32 po = pass_offs[pj], pl = pass_offs[pj + 1] - po;
        for (u32 j = 0; j < pl; j++) pb[j] = pass_data[po + j];
        pbkdf2_mnemonic_salt(seed, buf, bl, pb, pl);
    }
    u64 ist[8], ost[8], I[8];
    {
        const char bs[12] = {'B', 'i', 't', 'c', 'o', 'i', 'n', ' ', 's', 'e', 'e', 'd'};
        unsigned char key[12];
        for (int j = 0; j < 12; j++) key[j] = (unsigned char)bs[j];
        hmac512_init(ist, ost, key, 12);
        unsigned char dat[256];
        for (int j = 0; j < 8; j++) store_be64(dat + 8 * j, seed[j]);
        hmac512_final(I, ist, ost, dat, 64);
    }
    u32 k[8];
    unsigned char c[32], d33[33];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        k[7 - 2 * j] = (u32)(I[j] >> 32);
        k[6 - 2 * j] = (u32)I[j];
    }
    for (int j = 0; j < 4; j++) store_be64(c + 8 * j, I[4 + j]);
    u32 *kout = keys_out + idx * count * 8;
    unsigned char *vout = valid_out + idx * count;
    if (path.depth == 0) { // "m": the master key as it is, derive_count times (increment_path_index leaves "m" alone)
#pragma unroll 1
        for (u32 j = 0; j < count; j++) {
#pragma unroll
            for (int q = 0; q < 8; q++) kout[j * 8 + q] = bswap32(k[7 - q]);
            vout[j] = 1;
        }
        return;
    }
    // ---- all levels but the last
#pragma unroll 1
    for (u32 d = 0; d + 1 < path.depth; d++) {
        const u32 comp = path.comp[d];
        const bool hard = (comp >> 31) != 0;
        if (!scalar_is_key(k)) return; // SecretKey::from_slice(parent) in scalar_add / private_to_public
        ckd_parent_data(d33, k, hard, table, wins, nwin);
        u32 ck[8];
        unsigned char cc[32];
        if (!ckd_step(ck, cc, k, c, d33, comp)) return;
#pragma unroll
        for (int q = 0; q < 8; q++) k[q] = ck[q];
        for (int q = 0; q < 32; q++) c[q] = cc[q];
    }
    // ---- the last level: index, index + 1, ...; the first failure ends the item (the reference does not advance the
    // path after a failed derivation, so every later attempt repeats it)
    const u32 last = path.comp[path.depth - 1];
    const bool hard = (last >> 31) != 0;
    if (!scalar_is_key(k)) return;
    ckd_parent_data(d33, k, hard, table, wins, nwin);
#pragma unroll 1
    for (u32 j = 0; j < count; j++) {
        const u64 index = (u64)(last & 0x7FFFFFFFu) + j;
        if (index >= 0x80000000ull) return; // "exceeds BIP32 limit"
        u32 ck[8];
        unsigned char cc[32];
        if (!ckd_step(ck, cc, k, c, d33, (u32)index | (last & 0x80000000u))) return;
#pragma unroll
        for (int q = 0; q < 8; q++) kout[j * 8 + q] = bswap32(ck[7 - q]);
        vout[j] = 1;
    }
}
