// vk_chain.cuh -- the long sequential key-derivation chains (BASELINE config 5) as time-sliced work on gfx950.
//
// A PBKDF2-HMAC-SHA512 stretch (2 x 2047 dependent HMACs per MultiBit-HD mnemonic, src/multibit.rs:171-200) or Electrum's
// 100 000-round SHA-256 stretch (src/electrum.rs:150-168) has no parallelism inside one input, and one wave carries 64
// inputs: 100 000 inputs are 1 563 waves for 1 024 SIMDs, so a kernel in which a wave owns its inputs from start to end
// leaves half the machine idle while the second round of waves finishes.  Here the chain is cut into CHUNKS of a few
// hundred microseconds; the state of a group of 64 inputs lives in HBM between chunks ([group][word][lane], one coalesced
// row per state word), and the groups circulate through a ring: a worker wave pops a group, advances it by one chunk,
// pushes it back.  Fast workers (alone on their SIMD) simply take more chunks, so all groups finish together and every
// SIMD stays busy until the end; with workers <= groups a pop never waits for more than the visibility of a push.
//
//   ring protocol (all counters device-scope atomics; data handed over with release / acquire fences at agent scope,
//   which on gfx950 write back / invalidate the per-XCD L2):
//     slots[t & mask] = (t + 1) << 32 | group      pushed with ticket t = tail++ after the group's chunk counter is stored
//     pop:  h = head++;  h >= total -> the worker is done;  wait until slots[h & mask] >> 32 == h + 1
//     total = groups x chunks tickets; a group is pushed again after every chunk but its last, so pushes == tickets
//   A worker is one wave; every ring call is wave-uniform by construction (see ring_pop).  Every wait is bounded: a worker that sees no progress for ~seconds raises ring->error and leaves (the host reports
//   it); with a correct protocol the bound is never reached (a pop of ticket h needs push h, which needs only work that
//   is already running -- see DESIGN.md section 4.4).
#pragma once
#include "vk_hash.cuh"

#ifndef VK_BLOCK
#define VK_BLOCK 256
#endif

// ------------------------------------------------------------------------------------------------ SHA-512
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

// A 64-bit value from its halves AS A REGISTER PAIR.  Written as (hi << 32) | lo the compiler sees two zero-extended
// summands, reassociates every SHA-512 addition into one 64-bit add per half and materialises the zero halves with
// v_mov: 8 003 VALU instructions per PBKDF2 iteration; as a bit-cast of a two-element vector the pair stays a pair and
// each addition is ONE v_lshl_add_u64 (4.2 issue cycles, like a 32-bit v_alignbit: profiles/archive/r03_ubench_valu_rates.txt):
// 6 602 instructions per iteration -- 3 086 v_alignbit, 1 757 v_bitop3, 1 478 v_lshl_add_u64, 236 shifts.
typedef u32 vk_u32x2 __attribute__((ext_vector_type(2)));
VK_DEV u64 pack64(u32 lo, u32 hi)
{
    const vk_u32x2 v = {lo, hi};
    return __builtin_bit_cast(u64, v);
}

// 64-bit rotate as two v_alignbit_b32
template <int N>
VK_DEV u64 rotr64(u64 x)
{
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    if constexpr (N < 32) return pack64(__builtin_amdgcn_alignbit(hi, lo, N), __builtin_amdgcn_alignbit(lo, hi, N));
    else if constexpr (N == 32) return pack64(hi, lo);
    else return pack64(__builtin_amdgcn_alignbit(lo, hi, N - 32), __builtin_amdgcn_alignbit(hi, lo, N - 32));
}
template <int N>
VK_DEV u64 shr64(u64 x) // N < 32
{
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    return pack64(__builtin_amdgcn_alignbit(hi, lo, N), hi >> N);
}
VK_DEV u64 xor3_64(u64 a, u64 b, u64 c) { return pack64(xor3((u32)a, (u32)b, (u32)c), xor3((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32))); }
VK_DEV u64 ch64(u64 e, u64 f, u64 g) { return pack64(bool_ch((u32)e, (u32)f, (u32)g), bool_ch((u32)(e >> 32), (u32)(f >> 32), (u32)(g >> 32))); }
VK_DEV u64 maj64(u64 a, u64 b, u64 c) { return pack64(bool_maj((u32)a, (u32)b, (u32)c), bool_maj((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32))); }
VK_DEV u64 bswap64(u64 x) { return pack64(__builtin_bswap32((u32)(x >> 32)), __builtin_bswap32((u32)x)); }

// one SHA-512 compression; w[16] is consumed as the rolling schedule
VK_DEV void sha512_compress(u64 st[8], u64 w[16])
{
    u64 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 80; i++) {
        if (i >= 16) {
            const u64 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const u64 s0 = xor3_64(rotr64<1>(w15), rotr64<8>(w15), shr64<7>(w15));
            const u64 s1 = xor3_64(rotr64<19>(w2), rotr64<61>(w2), shr64<6>(w2));
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const u64 S1 = xor3_64(rotr64<14>(e), rotr64<18>(e), rotr64<41>(e));
        const u64 t1 = h + S1 + ch64(e, f, g) + VK_K512[i] + w[i & 15];
        const u64 S0 = xor3_64(rotr64<28>(a), rotr64<34>(a), rotr64<39>(a));
        const u64 t2 = S0 + maj64(a, b, c);
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// the block that follows a 64-byte digest inside an HMAC: digest || 0x80 || 0.. || bit length of (128 + 64) bytes
VK_DEV void sha512_digest_block(u64 w[16], const u64 dg[8])
{
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = dg[i];
    w[8] = 0x8000000000000000ull;
#pragma unroll
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = (128 + 64) * 8;
}

// ------------------------------------------------------------------------------------------------ staging area
// Variable-length byte strings of the cold kernels (the mnemonic as given, the rebuilt 48-word mnemonic, "{i}:{c}:" ||
// public key) live in a lane-private slice of an HBM staging buffer instead of private arrays (which the compiler can
// only index dynamically through scratch): VK_STAGE_WORDS little-endian u64 words per input, zero filled behind the
// string, so a hash reads whole words and patches the padding in registers.
#define VK_STAGE_WORDS 64u
#define VK_STAGE_BYTES (8u * VK_STAGE_WORDS)
#define VK_STAGE_MAX_SHA512 (VK_STAGE_BYTES - 17u) /* longest string whose SHA-512 padding still fits the slice */
#define VK_STAGE_MAX_SHA256 (VK_STAGE_BYTES - 9u)

struct vk_stage_writer {
    u64 *p;
    u64 acc;
    u32 fill, pos;
    __device__ __forceinline__ void init(u64 *base)
    {
        p = base;
        acc = 0;
        fill = pos = 0;
    }
    // append the low n <= 8 bytes of v (the bytes of v above n must be zero); the caller keeps the total <= VK_STAGE_BYTES
    __device__ __forceinline__ void put(u64 v, u32 n)
    {
        acc |= v << (8 * fill);
        if (fill + n >= 8) {
            p[pos++] = acc;
            acc = fill ? (v >> (8 * (8 - fill))) : 0;
            fill = fill + n - 8;
        } else {
            fill += n;
        }
    }
    // flushes, zero fills the slice up to word `upto` (what the reader of this string will touch), returns the length in bytes
    __device__ __forceinline__ u32 finish(u32 upto = VK_STAGE_WORDS)
    {
        const u32 len = 8 * pos + fill;
        if (pos < VK_STAGE_WORDS) p[pos++] = acc;
        for (; pos < upto; pos++) p[pos] = 0;
        return len;
    }
};

// big-endian message word j (8 bytes) of the SHA-512 padding of the staged string: the string, 0x80, zeros, and in the
// last word of the last block the bit length of (prefix + len) bytes
VK_DEV u64 stage_sha512_word(const u64 *stage, u32 j, u32 len, u32 last_word, u32 prefix)
{
    u64 w = j < VK_STAGE_WORDS ? bswap64(stage[j]) : 0;
    const u32 rel = len - 8 * j; // position of the 0x80 byte inside this word, if 0..7
    if (rel < 8u) w |= 0x80ull << (56 - 8 * rel);
    if (j == last_word) w |= (u64)(prefix + len) * 8;
    return w;
}

// st = SHA-512 state after absorbing the staged string (and its padding) on top of `prefix` already absorbed bytes
VK_DEV void sha512_absorb_stage(u64 st[8], const u64 *stage, u32 len, u32 prefix)
{
    const u32 nblk = (len + 17 + 127) / 128;
#pragma unroll 1
    for (u32 b = 0; b < nblk; b++) {
        u64 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = stage_sha512_word(stage, 16 * b + i, len, 16 * nblk - 1, prefix);
        sha512_compress(st, w);
    }
}

// 32-bit word j of the staged bytes (little-endian in memory), as a big-endian message word.  The slice is only ever
// accessed as u64: it is written through u64 pointers, and reading it through a u32 pointer would let type-based alias
// analysis move the loads above those stores.
VK_DEV u32 stage_be32(const u64 *stage, u32 j)
{
    if (j >= 2 * VK_STAGE_WORDS) return 0;
    const u64 q = stage[j >> 1];
    return __builtin_bswap32((j & 1u) ? (u32)(q >> 32) : (u32)q);
}

// the same for SHA-256 over the staged string
VK_DEV void sha256_stage(u32 out[8], const u64 *stage, u32 len)
{
    const u32 nblk = (len + 9 + 63) / 64;
    u32 st[8] = VK_SHA256_IV;
#pragma unroll 1
    for (u32 b = 0; b < nblk; b++) {
        u32 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const u32 j = 16 * b + i;
            u32 v = stage_be32(stage, j);
            const u32 rel = len - 4 * j;
            if (rel < 4u) v |= 0x80u << (24 - 8 * rel);
            if (j == 16 * nblk - 1) v |= len * 8;
            w[i] = v;
        }
        sha256_compress(st, w);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = st[i];
}

// ------------------------------------------------------------------------------------------------ HMAC-SHA512 pieces
// states after the ipad / opad block of a key given as 16 big-endian words (zero padded)
VK_DEV void hmac512_pads(u64 ist[8], u64 ost[8], const u64 key[16])
{
    const u64 iv[8] = VK_SHA512_IV;
    u64 w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) ist[i] = ost[i] = iv[i];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = key[i] ^ 0x3636363636363636ull;
    sha512_compress(ist, w);
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = key[i] ^ 0x5c5c5c5c5c5c5c5cull;
    sha512_compress(ost, w);
}

// out = outer hash of an inner state that has absorbed its whole message
VK_DEV void hmac512_outer(u64 out[8], const u64 ost[8], const u64 inner[8])
{
    u64 w[16];
    sha512_digest_block(w, inner);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = ost[i];
    sha512_compress(out, w);
}

// PBKDF2-HMAC-SHA512 up to U1: key schedule of the staged password (RFC 2104: a key longer than the 128-byte block is
// hashed first) and U1 = HMAC(salt || INT(1)), the salt message given as `salt_nblk` ready-padded blocks of big-endian
// words (built on the host: "mnemonic" || passphrase || 00 00 00 01 || SHA-512 padding for 128 + len bytes).
// SHA-512 of `len` bytes read from the caller's input itself (any length, any alignment): the key of a password that does not
// fit the staging slice.  Rare (a mnemonic line of more than VK_STAGE_MAX_STRING bytes), one byte load per message byte.
VK_DEV void sha512_bytes(u64 st[8], const unsigned char *__restrict__ p, u32 len)
{
    const u32 nblk = (len + 17 + 127) / 128;
#pragma unroll 1
    for (u32 b = 0; b < nblk; b++) {
        u64 w[16];
#pragma unroll 1
        for (u32 i = 0; i < 16; i++) {
            u64 v = 0;
            for (u32 j = 0; j < 8; j++) {
                const u32 k = 128 * b + 8 * i + j;
                const u64 byte = k < len ? p[k] : (k == len ? 0x80u : 0u);
                v = (v << 8) | byte;
            }
            if (16 * b + i == 16 * nblk - 1) v |= (u64)len << 3;
            w[i] = v;
        }
        sha512_compress(st, w);
    }
}

#define VK_STAGE_MAX_STRING 448u /* bytes of a string the 512-byte staging slice holds with its SHA-512 padding */

// pw_bytes: the password where the caller gave it, read instead of the staging slice when pwlen > VK_STAGE_MAX_STRING (may be
// nullptr when the caller's strings always fit: the rebuilt mnemonics)
VK_DEV void pbkdf2_begin(u64 ist[8], u64 ost[8], u64 u1[8], const u64 *stage, u32 pwlen, const u64 *__restrict__ salt, u32 salt_nblk,
                         const unsigned char *__restrict__ pw_bytes = nullptr)
{
    u64 key[16];
    if (pwlen > 128) {
        u64 st[8] = VK_SHA512_IV;
        if (pwlen > VK_STAGE_MAX_STRING) sha512_bytes(st, pw_bytes, pwlen);
        else sha512_absorb_stage(st, stage, pwlen, 0);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            key[i] = st[i];
            key[8 + i] = 0;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) key[i] = bswap64(stage[i]); // zero filled behind the password
    }
    hmac512_pads(ist, ost, key);
    u64 in[8];
#pragma unroll
    for (int i = 0; i < 8; i++) in[i] = ist[i];
#pragma unroll 1
    for (u32 b = 0; b < salt_nblk; b++) {
        u64 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = salt[16 * b + i];
        sha512_compress(in, w);
    }
    hmac512_outer(u1, ost, in);
}

// Host side of pbkdf2_begin's salt argument: "mnemonic" || passphrase || INT(1) as the blocks the inner hash absorbs after
// the ipad block -- big-endian words, SHA-512 padding for 128 + len bytes.  A passphrase of any length (round 5; until then the
// message had to fit 48 words: 96 bytes for multibit, 256 for bitimage): vk_salt_nblk blocks of 16 words each.
static inline u32 vk_salt_nblk(size_t pl) { return (u32)((8 + pl + 4 + 17 + 127) / 128); }
static inline u32 vk_salt_blocks(u64 *out /* 16 * vk_salt_nblk(pl) words */, const unsigned char *pass, size_t pl)
{
    const char *tag = "mnemonic";
    const size_t len = 8 + pl + 4;
    const u32 nblk = vk_salt_nblk(pl);
    const u64 bits = (u64)(128 + len) * 8;
    const size_t total = (size_t)nblk * 128;
    for (size_t i = 0; i < total / 8; i++) {
        u64 w = 0;
        for (size_t j = 0; j < 8; j++) {
            const size_t k = 8 * i + j;
            unsigned char c = 0;
            if (k < 8) c = (unsigned char)tag[k];
            else if (k < 8 + pl) c = pass[k - 8];
            else if (k == len - 1) c = 1;
            else if (k == len) c = 0x80;
            else if (k >= total - 8) c = (unsigned char)(bits >> (8 * (total - 1 - k)));
            w = (w << 8) | c;
        }
        out[i] = w;
    }
    return nblk;
}

// ------------------------------------------------------------------------------------------------ the ring
struct vk_ring {
    u32 head;  // next pop ticket
    u32 tail;  // next push ticket (starts at `groups`: the host pre-fills one slot per group)
    u32 total; // tickets = groups x chunks
    u32 mask;  // slots - 1, slots a power of two >= 2 x groups
    u32 error; // a bounded wait ran out (never with a correct protocol)
    u32 pad[3];
};
#define VK_RING_SPINS (1u << 22) /* x ~1.7 us of s_sleep: ~7 s without progress before a worker gives up */

// The ring calls are WAVE-UNIFORM BY CONSTRUCTION: every lane executes every instruction, on the same addresses and
// values; loads are brought into SGPRs (readfirstlane) so that every branch is a scalar branch, the stores are 64 lanes
// storing the same word, and the one thing that must happen once per wave -- taking a ticket -- is an atomic add of
// (lane == 0 ? 1 : 0).  No `if (lane == 0)` region anywhere: round 3's first two versions had them, and with divergent
// regions inside the worker's loop the compiler is free to let the other 63 lanes run ahead into the next iteration while
// lane 0 is still pushing and popping (it did: first only lane 0's inputs were advanced, then the wave hung).
VK_DEV u32 ring_load32(const u32 *p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
VK_DEV u32 ring_ticket(u32 *counter)
{
    const u32 one = (threadIdx.x & 63u) == 0 ? 1u : 0u;
    return __builtin_amdgcn_readfirstlane(__hip_atomic_fetch_add(counter, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// pop the next group for this wave; -1 = no tickets left (or the ring failed)
VK_DEV int ring_pop(vk_ring *r, unsigned long long *slots)
{
    const u32 h = ring_ticket(&r->head);
    if (h >= ring_load32(&r->total)) return -1;
    unsigned long long *s = slots + (h & ring_load32(&r->mask));
    u32 g = 0xFFFFFFFFu;
    for (u32 spin = 0; spin < VK_RING_SPINS; spin++) {
        const unsigned long long v = __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
        if (hi == h + 1) {
            g = lo;
            break;
        }
        if (ring_load32(&r->error)) break;
        __builtin_amdgcn_s_sleep(64);
    }
    if (g == 0xFFFFFFFFu) __hip_atomic_store(&r->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(s, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // consumed: the slot may be reused
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // the group's state written by its previous worker is visible now
    return (int)g;
}

// hand the group back after its state and chunk counter have been stored
VK_DEV void ring_push(vk_ring *r, unsigned long long *slots, u32 g)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const u32 t = ring_ticket(&r->tail);
    unsigned long long *s = slots + (t & ring_load32(&r->mask));
    // slots >= 2 x groups: the previous occupant (ticket t - slots) was consumed long ago; checked, not assumed
    u32 spin = 0;
    for (; spin < VK_RING_SPINS; spin++) {
        const unsigned long long v = __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((__builtin_amdgcn_readfirstlane((u32)v) | __builtin_amdgcn_readfirstlane((u32)(v >> 32))) == 0) break;
        __builtin_amdgcn_s_sleep(64);
    }
    if (spin == VK_RING_SPINS) __hip_atomic_store(&r->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(s, ((unsigned long long)(t + 1) << 32) | g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// host side of the protocol: ring header + one pre-filled slot per group (ticket t = group t), counters cleared
__global__ void vk_ring_init_kernel(vk_ring *r, unsigned long long *slots, u32 *chunk_done, u32 groups, u32 chunks, u32 nslots)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        r->head = 0;
        r->tail = groups;
        r->total = groups * chunks;
        r->mask = nslots - 1;
        r->error = 0;
    }
    if (i < nslots) slots[i] = i < groups ? (((unsigned long long)(i + 1) << 32) | i) : 0ull;
    if (i < groups) chunk_done[i] = 0;
}

// The worker: Body::run(args, group, chunk) advances the 64 inputs of `group` by chunk number `chunk`.
template <class Body>
__global__ void __launch_bounds__(256) vk_ring_kernel(vk_ring *r, unsigned long long *slots, u32 *chunk_done, u32 chunks, typename Body::args a)
{
    for (;;) {
        const int g = ring_pop(r, slots);
        if (g < 0) return;
        const u32 c = __builtin_amdgcn_readfirstlane(chunk_done[g]); // the same address in every lane
        Body::run(a, (u32)g, c);
        chunk_done[g] = c + 1; // every lane, the same word
        if (c + 1 < chunks) ring_push(r, slots, (u32)g);
    }
}

// ------------------------------------------------------------------------------------------------ PBKDF2 chunks
// state of one group: 32 rows of 64 u64 -- ist[8], ost[8], u[8], t[8] -- row-major [word][lane]
#define VK_PB_WORDS 32u
struct vk_pbkdf2_body {
    struct args {
        u64 *state;
        u32 iters;       // HMAC iterations after U1 (2047 for c = 2048)
        u32 chunk_iters; // iterations per chunk
    };
    static __device__ __forceinline__ void run(const args &a, u32 g, u32 c)
    {
        u64 *s = a.state + (u64)g * (VK_PB_WORDS * 64) + (threadIdx.x & 63u);
        u64 is_[8], os_[8], u[8], t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            is_[i] = s[64 * i];
            os_[i] = s[64 * (8 + i)];
            u[i] = s[64 * (16 + i)];
            t[i] = s[64 * (24 + i)];
        }
        const u32 first = c * a.chunk_iters;
        const u32 n = min(a.chunk_iters, a.iters - first);
#pragma unroll 1
        for (u32 it = 0; it < n; it++) {
            u64 st[8], w[16];
#pragma unroll
            for (int i = 0; i < 8; i++) st[i] = is_[i];
            sha512_digest_block(w, u);
            sha512_compress(st, w);
            sha512_digest_block(w, st);
#pragma unroll
            for (int i = 0; i < 8; i++) st[i] = os_[i];
            sha512_compress(st, w);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                u[i] = st[i];
                t[i] ^= st[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            s[64 * (16 + i)] = u[i];
            s[64 * (24 + i)] = t[i];
        }
    }
};

VK_DEV void pb_state_store(u64 *state, u64 idx, const u64 ist[8], const u64 ost[8], const u64 u1[8])
{
    u64 *s = state + (idx >> 6) * (VK_PB_WORDS * 64) + (idx & 63u);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        s[64 * i] = ist[i];
        s[64 * (8 + i)] = ost[i];
        s[64 * (16 + i)] = u1[i];
        s[64 * (24 + i)] = u1[i];
    }
}
VK_DEV void pb_state_result(u64 t[8], const u64 *state, u64 idx)
{
    const u64 *s = state + (idx >> 6) * (VK_PB_WORDS * 64) + (idx & 63u);
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = s[64 * (24 + i)];
}
