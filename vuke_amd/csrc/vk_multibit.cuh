// vk_multibit.cuh -- MultiBit HD "seed as entropy" key chain and the Bitimage chain on gfx950, one input per lane, as a
// sequence of kernels around the time-sliced PBKDF2 stretch of vk_chain.cuh.
//
// Reference behaviour (src/multibit.rs:87-350, src/transform/multibit.rs:45-61):
//   words = mnemonic.split_whitespace(); 12/15/18/21/24 of them, all in the BIP-39 list, else skip;
//   seed1 = PBKDF2-HMAC-SHA512(password = the mnemonic string as given, salt = "mnemonic" || passphrase, 2048);
//   48 words from the 512 bits of seed1 + 16 checksum bits (SHA-256), joined by single spaces;
//   seed2 = PBKDF2-HMAC-SHA512(password = that 48-word string, same salt, 2048);
//   (k, c) = HMAC-SHA512("Bitcoin seed", seed2); m/0' hardened, /0 and /i normal CKDpriv with the
//   reference's error rules (parent must be a valid key, IL < n, child != 0); i = 0..count-1.
// The BIP-39 list is supplied by the caller (vk_bip39_wordlist_set): 2048 words of <= 8 bytes, zero padded to 8.
//
//   vk_mb_prepare_kernel   split + validate, stage the password, HMAC key schedule, U1            (cold)
//   vk_ring_kernel<pbkdf2> 2047 HMAC iterations in chunks, groups of 64 inputs circulating          (hot, vk_chain.cuh)
//   vk_mb_middle_kernel    seed1 -> 48 words -> staged password 2 -> key schedule, U1             (cold)
//   vk_ring_kernel<pbkdf2> again
//   vk_mb_final_kernel     master key, CKDpriv m/0'/0/i with serP through the k*G table            (cold, vk_bip32.cuh)
// No kernel keeps a byte array in private memory: variable-length strings go through the HBM staging slice of the input
// (vk_chain.cuh), fixed-layout messages (HMAC pads, the 37 bytes of a CKD message) are assembled in registers.
#pragma once
#include "vk_chain.cuh"


// ---- BIP-39 list as an open-addressed set (built by vk_bip39_wordlist_set): 4096 slots of packed words, 0 = empty
#define VK_WORDSET_SLOTS 4096u
VK_DEV u32 wordset_hash(u64 w) { return (u32)((w * 0x9E3779B97F4A7C15ull) >> 52); }
VK_DEV bool wordset_has(const u64 *__restrict__ set, u64 w)
{
    u32 i = wordset_hash(w);
    for (u32 probe = 0; probe < VK_WORDSET_SLOTS; probe++) {
        const u64 v = set[i];
        if (v == w) return true;
        if (v == 0) return false;
        i = (i + 1) & (VK_WORDSET_SLOTS - 1);
    }
    return false;
}

VK_DEV bool mb_is_ws(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0x0b || c == 0x0c; }

// bytes of a packed word (little-endian, zero padded to 8)
VK_DEV u32 packed_len(u64 w) { return 8u - ((u32)__builtin_clzll(w | 1ull) >> 3); }

// `nw` words of 11 bits from the big-endian bit string S[0] S[1] ... (each S[i] 64 bits, static indexing only), looked
// up in the list and joined by single spaces into the staging slice; returns the string's length
template <int NW, int NS>
VK_DEV u32 stage_words(u64 *stage, const u64 (&S)[NS], const u64 *__restrict__ wordlist)
{
    vk_stage_writer wr;
    wr.init(stage);
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const int q = (11 * k) >> 6, r = (11 * k) & 63;
        u32 wi;
        if (r + 11 <= 64) wi = (u32)(S[q] >> (64 - r - 11)) & 0x7FFu;
        else wi = (u32)((S[q] << (r + 11 - 64)) | (S[q + 1 < NS ? q + 1 : q] >> (128 - r - 11))) & 0x7FFu;
        if (k) wr.put(0x20, 1);
        const u64 w = wordlist[wi];
        wr.put(w, packed_len(w));
    }
    return wr.finish();
}

// ------------------------------------------------------------------------------------------------ MultiBit: prepare
__global__ void __launch_bounds__(VK_BLOCK) vk_mb_prepare_kernel(const unsigned char *__restrict__ data, const u32 *__restrict__ offs, u64 n,
                                                                   const u64 *__restrict__ wordset, const u64 *__restrict__ salt, u32 salt_nblk,
                                                                   u64 *__restrict__ stage_all, u64 *__restrict__ state, unsigned char *__restrict__ ok_out)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n) return;
    ok_out[idx] = 0;
    const u32 len = offs[idx + 1] - offs[idx];
    const unsigned char *mn = data + offs[idx];
    // ---- split_whitespace + validate_mnemonic; the password is the string as given, of ANY length (src/multibit.rs:87-101: the
    // reference hashes what it is given): up to 448 bytes it is staged, a longer line -- a valid mnemonic is at most 24 words of
    // at most 8 letters, so only hundreds of redundant blanks between the words get there -- is hashed from the input bytes
    const bool staged = len <= VK_STAGE_MAX_STRING;
    u64 *stage = stage_all + idx * VK_STAGE_WORDS;
    vk_stage_writer wr;
    wr.init(stage);
    u32 nwords = 0, wl = 0;
    u64 packed = 0;
    bool ok = true, nul = false;
    for (u32 i = 0; i <= len; i++) {
        const unsigned char c = i < len ? mn[i] : (unsigned char)' ';
        if (i < len && staged) wr.put(c, 1);
        if (mb_is_ws(c)) {
            if (wl) { // a word ends: in the list?  (no list word is longer than 8 bytes or holds a NUL byte)
                nwords++;
                ok = ok && wl <= 8 && !nul && wordset_has(wordset, packed);
            }
            wl = 0;
            packed = 0;
            nul = false;
        } else {
            if (wl < 8) packed |= (u64)c << (8 * wl);
            nul = nul || c == 0;
            wl++;
        }
    }
    wr.finish();
    if (!ok || !(nwords == 12 || nwords == 15 || nwords == 18 || nwords == 21 || nwords == 24)) return;
    u64 ist[8], ost[8], u1[8];
    pbkdf2_begin(ist, ost, u1, stage, len, salt, salt_nblk, mn);
    pb_state_store(state, idx, ist, ost, u1);
    ok_out[idx] = 1;
}

// ------------------------------------------------------------------------------------------------ MultiBit: middle
__global__ void __launch_bounds__(VK_BLOCK) vk_mb_middle_kernel(u64 n, const u64 *__restrict__ wordlist, const u64 *__restrict__ salt, u32 salt_nblk,
                                                                  u64 *__restrict__ stage_all, u64 *__restrict__ state, const unsigned char *__restrict__ ok)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n || !ok[idx]) return;
    // ---- 512 bits of seed1 + 16 checksum bits (first two bytes of SHA-256(seed1)) -> 48 words
    u64 S[9];
    pb_state_result(S, state, idx);
    {
        u32 st[8] = VK_SHA256_IV, w[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            w[2 * i] = (u32)(S[i] >> 32);
            w[2 * i + 1] = (u32)S[i];
        }
        sha256_compress(st, w);
        w[0] = 0x80000000u;
#pragma unroll
        for (int i = 1; i < 15; i++) w[i] = 0;
        w[15] = 512;
        sha256_compress(st, w);
        S[8] = (u64)(st[0] >> 16) << 48;
    }
    u64 *stage = stage_all + idx * VK_STAGE_WORDS;
    const u32 len = stage_words<48, 9>(stage, S, wordlist);
    u64 ist[8], ost[8], u1[8];
    pbkdf2_begin(ist, ost, u1, stage, len, salt, salt_nblk);
    pb_state_store(state, idx, ist, ost, u1);
}

// ------------------------------------------------------------------------------------------------------ bitimage
// BitimageDeriver::from_file_bytes after the file has been reduced to its 32 bytes of entropy, then derive_path for
// derive_count paths that differ in the last index (reference src/bitimage.rs:38-74, src/transform/bitimage.rs:71-97):
// entropy -> 24-word mnemonic (8 checksum bits) -> PBKDF2 seed -> HMAC("Bitcoin seed") -> CKDpriv along the path.
// One lane per (file, passphrase); the parent of the last level is derived once.
// salt blocks of passphrase j: words salt_off[j] .. salt_off[j + 1] of salts (16 per block; passphrases of any length)
__global__ void __launch_bounds__(VK_BLOCK) vk_bitimage_prepare_kernel(const unsigned char *__restrict__ entropy, u64 n_items, u32 n_pass,
                                                                         const u64 *__restrict__ wordlist, const u64 *__restrict__ salts,
                                                                         const u32 *__restrict__ salt_off, u64 *__restrict__ stage_all, u64 *__restrict__ state)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n_items) return;
    const u64 file = idx / n_pass;
    const u32 pj = (u32)(idx % n_pass);
    // ---- entropy_to_mnemonic: 256 + 8 bits -> 24 words joined by single spaces
    u64 S[5];
    {
        const u32 *e32 = (const u32 *)(entropy + file * 32); // 32-byte records of a hipMalloc'ed buffer: aligned
        u32 st[8] = VK_SHA256_IV, w[16];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = __builtin_bswap32(e32[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) S[i] = pack64(w[2 * i + 1], w[2 * i]);
        w[8] = 0x80000000u;
#pragma unroll
        for (int i = 9; i < 15; i++) w[i] = 0;
        w[15] = 256;
        sha256_compress(st, w);
        S[4] = (u64)(st[0] >> 24) << 56;
    }
    u64 *stage = stage_all + idx * VK_STAGE_WORDS;
    const u32 len = stage_words<24, 5>(stage, S, wordlist);
    u64 ist[8], ost[8], u1[8];
    pbkdf2_begin(ist, ost, u1, stage, len, salts + salt_off[pj], (salt_off[pj + 1] - salt_off[pj]) / 16u);
    pb_state_store(state, idx, ist, ost, u1);
}
