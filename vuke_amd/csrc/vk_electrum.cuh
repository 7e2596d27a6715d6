// vk_electrum.cuh -- Electrum v1 deterministic-wallet keys on gfx950, one seed string per lane, as three kernels around
// the time-sliced stretch (vk_chain.cuh).
//
// Reference behaviour (src/electrum.rs:57-195, src/transform/electrum.rs:43-67):
//   seed must be all ASCII hex digits, else the input is skipped;
//   x = SHA256(seed || seed); 99 999 times x = SHA256(x || seed);           (stretch_key)
//   master = x must be a valid scalar (1 <= x < n), else the input is skipped;
//   mpk = X || Y of master * G;
//   seq_i = SHA256(SHA256("{i}:{0|1}:" || mpk)); skipped if seq_i >= n;
//   key_i = master + seq_i mod n; skipped if zero.
//
//   vk_electrum_prepare_kernel  hex check, x0, the constant tail of the stretch message (seed, padding, length)   (cold)
//   vk_ring_kernel<electrum>    99 999 rounds in chunks: only the 8 chaining words change, the tail stays in registers (hot)
//   vk_electrum_long_kernel     the same rounds for seeds of more than 64 bytes -- ANY length, as the reference hashes the whole
//                               slice (src/electrum.rs:57-78,150-168); such lines are rare: tail words read from the caller's
//                               input bytes every round, one lane per seed (ceil((41 + len) / 64) compressions per round), the
//                               rounds cut into launches of a bounded number of compressions (x parked in the state between
//                               them): a SHA-256 chain has no parallelism to give a wave, so a very long line costs what
//                               100 000 x len / 64 serial compressions cost one lane (~3 us each) -- but no single dispatch
//                               holds the GPU for more than about a second, and the process stays killable between them   (cold)
//   vk_electrum_final_kernel    master * G, 20 sequence hashes, child keys                                         (cold, vk_chain_keys.cuh)
#pragma once
#include "vk_chain.cuh"

#define VK_ELECTRUM_FAST_SEED 64 /* bytes: x || seed || padding then spans at most two SHA-256 blocks (register-resident tail) */

// big-endian word i of seed || 0x80 || 0.. || 64-bit bit length of (prefix + len) bytes, read from the input bytes themselves
// (any length, any alignment); `last` = index of the low word of the length field.  A word that lies entirely inside the seed
// comes from the two aligned dwords around it (v_alignbyte_b32 + byte swap): the second dword reaches up to four bytes
// beyond the seed (all four when the word's address is aligned: p[1] is then entirely behind it), which is why the device copy of
// the input carries VK_STRING_SLACK = 8 bytes behind the last string.
#define VK_STRING_SLACK 8
VK_DEV u32 electrum_tail_word_bytes(const unsigned char *__restrict__ seed, u32 i, u32 len, u32 last, u32 prefix)
{
    if (4 * i + 3 < len) {
        const u64 a = (u64)(seed + 4 * (u64)i);
        const u32 sh = (u32)a & 3u;
        const u32 *p = reinterpret_cast<const u32 *>(a - sh);
        return bswap32(__builtin_amdgcn_alignbyte(p[1], p[0], sh));
    }
    u32 v = 0;
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
        const u32 k = 4 * i + j;
        const u32 byte = k < len ? seed[k] : (k == len ? 0x80u : 0u);
        v = (v << 8) | byte;
    }
    if (i == last) v |= (prefix + len) << 3;
    if (i + 1 == last) v |= (prefix + len) >> 29;
    return v;
}

// big-endian word i of the constant tail seed || 0x80 || 0.. || bit length of (32 + len) bytes; `last` = index of the length word
VK_DEV u32 electrum_tail_word(const u64 *stage, u32 i, u32 len, u32 last)
{
    u32 v = 4u * i < len ? stage_be32(stage, i) : 0;
    const u32 rel = len - 4 * i; // bytes of the seed in this word, if 0..3: cut what follows off, set the 0x80 marker
    if (rel - 1u < 3u) v &= 0xFFFFFFFFu << (32 - 8 * rel);
    if (rel < 4u) v |= 0x80u << (24 - 8 * rel);
    if (i == last) v |= (32 + len) * 8;
    return v;
}

// state of one group: 33 rows of 64 u32 -- x[8], tail words tw[24], flags (bit 0: the message has a second block)
#define VK_EL_WORDS 33u

__global__ void __launch_bounds__(VK_BLOCK) vk_electrum_prepare_kernel(const unsigned char *__restrict__ data, const u32 *__restrict__ offs, u64 n,
                                                                         u64 *__restrict__ stage_all, u32 *__restrict__ state, unsigned char *__restrict__ ok_out)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n) return;
    ok_out[idx] = 0;
    const u32 len = offs[idx + 1] - offs[idx];
    const unsigned char *seed = data + offs[idx];
    u64 *stage = stage_all + idx * VK_STAGE_WORDS;
    const bool fast = len <= VK_ELECTRUM_FAST_SEED; // only the register-resident path reads the seed back from the staging slice
    vk_stage_writer wr;
    wr.init(stage);
    bool ok = true;
    for (u32 i = 0; i < len; i++) {
        const unsigned char c = seed[i];
        ok = ok && ((c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F')); // is_ascii_hexdigit
        if (fast) wr.put(c, 1);
    }
    wr.finish();
    if (!ok) return;
    // ---- x0 = SHA256(seed || seed): the doubled string is read byte by byte from the input (its second copy is unaligned)
    u32 x[8] = VK_SHA256_IV;
    {
        const u32 mlen = 2 * len, nblk = (mlen + 9 + 63) / 64;
#pragma unroll 1
        for (u32 b = 0; b < nblk; b++) {
            u32 w[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                u32 v = 0;
                for (u32 j = 0; j < 4; j++) {
                    const u32 k = 64 * b + 4 * i + j;
                    const u32 byte = k < mlen ? seed[k < len ? k : k - len] : (k == mlen ? 0x80u : 0u);
                    v = (v << 8) | byte;
                }
                if (16 * b + i == 16 * nblk - 1) v |= mlen << 3;
                if (16 * b + i == 16 * nblk - 2) v |= mlen >> 29;
                w[i] = v;
            }
            sha256_compress(x, w);
        }
    }
    // constant tail of the message x || seed || pad: words 8.. of block 1 and, if it exists, all of block 2
    const bool two = (32 + len + 9) > 64;
    u32 *s = state + (idx >> 6) * (VK_EL_WORDS * 64) + (idx & 63u);
#pragma unroll
    for (int i = 0; i < 8; i++) s[64 * i] = x[i];
    if (len > VK_ELECTRUM_FAST_SEED) { // stretched by vk_electrum_long_kernel; the ring carries the lane along with a harmless tail
#pragma unroll
        for (int i = 0; i < 24; i++) s[64 * (8 + i)] = 0;
        s[64 * 32] = 0;
        ok_out[idx] = 2;
        return;
    }
#pragma unroll
    for (int i = 0; i < 24; i++) s[64 * (8 + i)] = electrum_tail_word(stage, i, len, two ? 23u : 7u);
    s[64 * 32] = two ? 1u : 0u;
    ok_out[idx] = 1;
}

struct vk_electrum_body {
    struct args {
        u32 *state;
        u32 iters;       // rounds after x0 (99 999)
        u32 chunk_iters;
    };
    static __device__ __forceinline__ void run(const args &a, u32 g, u32 c)
    {
        u32 *s = a.state + (u64)g * (VK_EL_WORDS * 64) + (threadIdx.x & 63u);
        u32 x[8], tw[24];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = s[64 * i];
#pragma unroll
        for (int i = 0; i < 24; i++) tw[i] = s[64 * (8 + i)];
        const bool two = s[64 * 32] != 0;
        const u32 first = c * a.chunk_iters;
        const u32 n = min(a.chunk_iters, a.iters - first);
#pragma unroll 1
        for (u32 r = 0; r < n; r++) {
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
#pragma unroll
        for (int i = 0; i < 8; i++) s[64 * i] = x[i];
    }
};

// Seeds of more than 64 bytes, any length.  One lane per such seed; rounds [r_begin, r_end) of the rounds_total the stretch has
// (round 0 = x0 = SHA256(seed || seed), recomputed from the input: the state's copy was advanced by the ring with a harmless
// tail), x parked in the state rows between launches, ok[] stays 2 until the last slice has run.  Every message word is assembled
// from the caller's input bytes (no staging slice: no length limit); launched only when the host saw such a seed.
#define VK_EL_LONG_BUDGET (1u << 17) /* SHA-256 compressions of one lane per launch: a few tenths of a second */
__global__ void __launch_bounds__(VK_BLOCK) vk_electrum_long_kernel(const unsigned char *__restrict__ data, const u32 *__restrict__ offs, u64 n, u32 r_begin,
                                                                      u32 r_end, u32 rounds_total, u32 *__restrict__ state, unsigned char *__restrict__ ok)
{
    const u64 idx = (u64)blockIdx.x * VK_BLOCK + threadIdx.x;
    if (idx >= n || ok[idx] != 2) return;
    const u32 len = offs[idx + 1] - offs[idx];
    const unsigned char *seed = data + offs[idx];
    u32 *s = state + (idx >> 6) * (VK_EL_WORDS * 64) + (idx & 63u);
    u32 x[8] = VK_SHA256_IV;
    if (r_begin == 0) {
        const u32 mlen = 2 * len, nb0 = (mlen + 9 + 63) / 64;
#pragma unroll 1
        for (u32 b = 0; b < nb0; b++) {
            u32 w[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                u32 v = 0;
                for (u32 j = 0; j < 4; j++) {
                    const u32 k = 64 * b + 4 * i + j;
                    const u32 byte = k < mlen ? seed[k < len ? k : k - len] : (k == mlen ? 0x80u : 0u);
                    v = (v << 8) | byte;
                }
                if (16 * b + i == 16 * nb0 - 1) v |= mlen << 3;
                if (16 * b + i == 16 * nb0 - 2) v |= mlen >> 29;
                w[i] = v;
            }
            sha256_compress(x, w);
        }
        r_begin = 1;
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = s[64 * i];
    }
    const u32 nblk = (32 + len + 9 + 63) / 64, last = 16 * nblk - 9; // tail word index of the length field
#pragma unroll 1
    for (u32 r = r_begin; r < r_end; r++) {
        u32 st[8] = VK_SHA256_IV;
#pragma unroll 1
        for (u32 b = 0; b < nblk; b++) {
            u32 w[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (b == 0 && i < 8) w[i] = x[i];
                else w[i] = electrum_tail_word_bytes(seed, 16 * b + i - 8, len, last, 32u);
            }
            sha256_compress(st, w);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = st[i];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) s[64 * i] = x[i];
    if (r_end >= rounds_total) ok[idx] = 1;
}
