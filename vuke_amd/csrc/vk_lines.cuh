// vk_lines.cuh -- generate mode's compact CSV lines built on the device (reference src/output/console.rs:77-81):
//
//     {source},{transform},{private_key_hex},{p2pkh_compressed}\n
//
// source = the decimal seed plus the slot's label suffix (src/transform/lcg.rs:54-66, xorshift.rs:43-47,
// sha256_chain.rs:67), p2pkh_compressed = Base58Check(0x00 || hash160) (src/derive.rs:112-113).  In round 2 the device
// handed back raw keys and hash160s and 16 host threads did hex, two SHA-256 per address, the base-58 conversion and the
// line assembly: 1.7 of the 1.8 seconds of BASELINE config 1 (3 000 003 lines), with 3 ms of kernel time.  Here
//   vk_lines_kernel          one lane per (input, slot): the line into a fixed-stride slot of HBM, its length beside it
//   (exclusive scan of the lengths: hipcub)
//   vk_lines_compact_kernel  one wave per 64 lines: the bytes moved to their final place, 64 consecutive bytes per store
// and the host only writes the text.
#pragma once
#include "vk_chain.cuh" // vk_stage_writer, bswap64

#define VK_LINE_STRIDE 192u        /* bytes per line slot: 20 + 31 + 1 + 31 + 1 + 64 + 1 + 35 + 1 = 185 at most */
#define VK_LINES_MAX_ITEMS (1ull << 21) /* (input, slot) pairs per vk_generate_lines call: 400 MB of line slots */
#define VK_LINE_SUFFIX_BYTES 32u   /* per-slot label suffix / transform name: one length byte + up to 31 characters */

// Base58Check(0x00 || h160): up to 34 characters into the writer
VK_DEV void lines_put_p2pkh(vk_stage_writer &wr, const u32 h[5] /* hash160 as the 5 words the sweep stores (raw byte order) */)
{
    // payload bytes: 00 h0..h19; checksum = first 4 bytes of SHA256(SHA256(payload))
    u32 hb[5];
#pragma unroll
    for (int i = 0; i < 5; i++) hb[i] = __builtin_bswap32(h[i]); // big-endian words of the 20 bytes
    u32 w[16], d1[8], d2[8];
    w[0] = hb[0] >> 8;
    w[1] = (hb[0] << 24) | (hb[1] >> 8);
    w[2] = (hb[1] << 24) | (hb[2] >> 8);
    w[3] = (hb[2] << 24) | (hb[3] >> 8);
    w[4] = (hb[3] << 24) | (hb[4] >> 8);
    w[5] = (hb[4] << 24) | 0x00800000u;
#pragma unroll
    for (int i = 6; i < 15; i++) w[i] = 0;
    w[15] = 21 * 8;
    sha256_oneblock(d1, w);
    sha256_of_digest(d2, d1);
    // the 25 bytes as a 200-bit big-endian integer in seven 32-bit limbs (L[0] holds the top 8 bits = the version byte 0)
    u32 L[7] = {0u, hb[0], hb[1], hb[2], hb[3], hb[4], d2[0]};
    // repeated division by 58^5: five base-58 digits per pass, least significant first
    u32 dig[35];
#pragma unroll
    for (int pass = 0; pass < 7; pass++) {
        u64 rem = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const u64 cur = (rem << 32) | L[i];
            L[i] = (u32)(cur / 656356768u);
            rem = cur % 656356768u;
        }
        u32 r = (u32)rem;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            dig[5 * pass + k] = r % 58u;
            r /= 58u;
        }
    }
    // one '1' per leading zero BYTE of the payload (the version byte, then zero bytes of the hash160)
    u32 zeros = 1;
    {
        bool lead = true;
#pragma unroll
        for (int i = 0; i < 5; i++) {
#pragma unroll
            for (int k = 3; k >= 0; k--) {
                lead = lead && ((hb[i] >> (8 * k)) & 0xffu) == 0;
                zeros += lead ? 1u : 0u;
            }
        }
    }
    for (u32 i = 0; i < zeros; i++) wr.put('1', 1);
    bool started = false;
#pragma unroll
    for (int pos = 34; pos >= 0; pos--) {
        const u32 v = dig[pos];
        started = started || v != 0;
        if (started) {
            const u32 c = v < 9 ? '1' + v : v < 17 ? 'A' + (v - 9) : v < 22 ? 'J' + (v - 17) : v < 33 ? 'P' + (v - 22) : v < 44 ? 'a' + (v - 33) : 'm' + (v - 44);
            wr.put(c, 1);
        }
    }
}

// 8 lowercase hex characters of a 32-bit word whose bytes are in memory order (first byte = low byte)
VK_DEV u64 lines_hex8(u32 raw)
{
    u64 out = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const u32 byte = (raw >> (8 * b)) & 0xffu, hi = byte >> 4, lo = byte & 15u;
        const u64 ch = (u64)(hi < 10 ? '0' + hi : 'a' + hi - 10), cl = (u64)(lo < 10 ? '0' + lo : 'a' + lo - 10);
        out |= (ch | (cl << 8)) << (16 * b);
    }
    return out;
}

VK_DEV void lines_put_decimal(vk_stage_writer &wr, u64 v)
{
    u32 dig[20];
#pragma unroll
    for (int i = 0; i < 20; i++) {
        dig[i] = (u32)(v % 10u);
        v /= 10u;
    }
    bool started = false;
#pragma unroll
    for (int i = 19; i >= 0; i--) {
        started = started || dig[i] != 0 || i == 0;
        if (started) wr.put('0' + dig[i], 1);
    }
}

VK_DEV void lines_put_text(vk_stage_writer &wr, const unsigned char *__restrict__ entry /* length byte + characters */)
{
    const u32 n = entry[0];
    for (u32 i = 0; i < n; i++) wr.put(entry[1 + i], 1);
}

// one lane per item of a vk_generate chunk: item = input * slots + slot, input = lo + item / slots.  String sources
// (str_offs != nullptr): the label's first part is the input string itself -- it is not copied into the slot, the compaction
// kernel takes it from the input buffer; the slot holds what follows it.
__global__ void __launch_bounds__(256) vk_lines_kernel(u64 lo, u64 items, u32 slots, const u32 *__restrict__ priv, const u32 *__restrict__ h160c,
                                                      const unsigned char *__restrict__ valid,
                                                      const unsigned char *__restrict__ texts /* slots suffix entries, then the transform name */,
                                                      const u32 *__restrict__ str_offs, u64 *__restrict__ line_slots, u32 *__restrict__ lens)
{
    const u64 item = (u64)blockIdx.x * 256 + threadIdx.x;
    if (item >= items) return;
    if (!valid[item]) {
        lens[item] = 0;
        return;
    }
    const u64 input = item / slots;
    const u32 sub = (u32)(item - input * slots);
    vk_stage_writer wr;
    wr.init(line_slots + item * (VK_LINE_STRIDE / 8));
    u32 label_len = 0;
    if (str_offs) label_len = str_offs[input + 1] - str_offs[input];
    else lines_put_decimal(wr, lo + input);
    lines_put_text(wr, texts + (u64)sub * VK_LINE_SUFFIX_BYTES);
    wr.put(',', 1);
    lines_put_text(wr, texts + (u64)slots * VK_LINE_SUFFIX_BYTES);
    wr.put(',', 1);
#pragma unroll
    for (int i = 0; i < 8; i++) wr.put(lines_hex8(priv[item * 8 + i]), 8);
    wr.put(',', 1);
    u32 h[5];
#pragma unroll
    for (int i = 0; i < 5; i++) h[i] = h160c[item * 5 + i];
    lines_put_p2pkh(wr, h);
    wr.put('\n', 1);
    lens[item] = label_len + wr.finish(0);
}

// one wave per 64 consecutive lines: offs[] = exclusive scan of lens[]; the wave's output region is contiguous, every lane
// stores the bytes lane, lane + 64, ... of it (64 consecutive bytes per store), finding the line of a byte by a search over
// the 64 offsets of the group held one per lane
__global__ void __launch_bounds__(256) vk_lines_compact_kernel(u64 items, u32 slots, u32 stride /* bytes per slot */, const u64 *__restrict__ line_slots,
                                                              const u32 *__restrict__ lens,
                                                              const u32 *__restrict__ offs, const unsigned char *__restrict__ str_data,
                                                              const u32 *__restrict__ str_offs, unsigned char *__restrict__ text)
{
    const u32 lane = threadIdx.x & 63u;
    const u64 group = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6, first = group * 64;
    if (first >= items) return;
    const u64 mine = first + lane < items ? first + lane : items - 1;
    const u32 my_off = offs[mine];
    const u32 my_len = first + lane < items ? lens[mine] : 0u;
    const u32 begin = __shfl(my_off, 0, 64);
    const u32 last = (u32)min((u64)63, items - 1 - first);
    const u32 end = __shfl(my_off, last, 64) + __shfl(my_len, last, 64);
    const unsigned char *src = (const unsigned char *)line_slots;
    for (u32 base = begin; base < end; base += 64) { // wave-uniform trip count: every lane takes part in every shuffle
        const u32 pos = base + lane;
        // the last line of the group that starts at or before pos: binary search over the lanes' offsets (non-decreasing;
        // an empty line shares its offset with its successor, which is the one found)
        u32 l = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            const u32 cand = l + step;
            const u32 o = __shfl(my_off, cand < 64 ? cand : 63, 64);
            if (cand <= last && o <= pos) l = cand;
        }
        const u32 lo_off = __shfl(my_off, l, 64);
        if (pos < end) {
            u32 k = pos - lo_off;
            const u64 item = first + l;
            unsigned char byte;
            if (str_offs) { // a string source's line = the input string, then the slot
                const u64 input = item / slots;
                const u32 so = str_offs[input], ll = str_offs[input + 1] - so;
                byte = k < ll ? str_data[so + k] : src[item * stride + (k - ll)];
            } else {
                byte = src[item * stride + k];
            }
            text[pos] = byte;
        }
    }
}
