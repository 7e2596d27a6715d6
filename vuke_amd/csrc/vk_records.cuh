// vk_records.cuh -- generate mode's FULL record made on the device, column by column (reference: DerivedKey,
// src/derive.rs:12-41,94-165, as src/storage/schema.rs:137-262 lays it into the 19-column record batch).
//
// The host formatter of rounds 1-3 built a DerivedKey of fourteen strings per key and appended it to nineteen column buffers
// behind one mutex: 6.4 s for the 3 000 003 records of BASELINE config 1 (2.8 GB of Parquet pages), of which the device
// worked for milliseconds.  Here every value column of the record leaves the device in the form the Parquet writer puts into
// a PLAIN page -- BYTE_ARRAY values as a 4-byte little-endian length followed by the bytes, FIXED_LEN_BYTE_ARRAY(32) and
// INT32 values back to back -- and the host only frames, compresses and writes pages:
//   vk_records_key_kernel    one lane per (input, slot): source label, raw key, hex, decimal, binary, bit length, Hamming
//                            weight, leading zero nibbles, both WIFs -> one fixed-stride slot per column
//   vk_records_addr_kernel   both P2PKH addresses, P2WPKH (bech32), both SEC1 public keys as hex
//   (per column: exclusive scan of the value lengths, hipcub)  +  vk_lines_compact_kernel (vk_lines.cuh)
// Mainnet encodings (version bytes 0x80 / 0x00, hrp "bc"), as Network::Bitcoin in src/derive.rs:102-118.
#pragma once
#include "vk_lines.cuh"

// column ids: VK_RC_* of include/vuke_hip.h; the stride of one slot = the longest value incl. its length prefix, rounded up to 8 bytes
#define VK_RECORDS_MAX_ITEMS (1ull << 20) /* (input, slot) pairs per vk_generate_columns call: 1 GB of slots */

__host__ __device__ constexpr u32 vk_rc_stride(int col)
{
    return col == VK_RC_SOURCE        ? 56u  /* 4 + 20 digits + 31 suffix = 55 */
           : col == VK_RC_KEY_RAW     ? 32u
           : col == VK_RC_KEY_HEX     ? 72u  /* 4 + 64 */
           : col == VK_RC_KEY_DECIMAL ? 88u  /* 4 + 78 */
           : col == VK_RC_KEY_BINARY  ? 264u /* 4 + 256 */
           : col == VK_RC_PUBKEY_C    ? 72u  /* 4 + 66 */
           : col == VK_RC_PUBKEY_U    ? 136u /* 4 + 130 */
           : col == VK_RC_P2PKH_C     ? 40u  /* 4 + 34 */
           : col == VK_RC_P2PKH_U     ? 40u
           : col == VK_RC_P2WPKH      ? 48u /* 4 + 42 */
           : col == VK_RC_WIF_C       ? 56u /* 4 + 52 */
           : col == VK_RC_WIF_U       ? 56u /* 4 + 51 */
                                      : 8u; /* the three INT32 columns */
}

struct vk_record_slots {
    u64 *slot[VK_RC_COLUMNS]; // items x stride bytes
    u32 *lens[VK_RC_COLUMNS]; // items: bytes this item contributes to the column (0 = no record)
};

// the writer's flush without the spare word vk_stage_writer::finish() leaves behind (a value may fill its slot exactly)
VK_DEV u32 rec_finish(vk_stage_writer &wr)
{
    if (wr.fill) wr.p[wr.pos] = wr.acc;
    return 8 * wr.pos + wr.fill;
}

VK_DEV u32 base58_char(u32 v)
{
    return v < 9 ? '1' + v : v < 17 ? 'A' + (v - 9) : v < 22 ? 'J' + (v - 17) : v < 33 ? 'P' + (v - 22) : v < 44 ? 'a' + (v - 33) : 'm' + (v - 44);
}

// base-58 digits (least significant first) of the big-endian integer in L[0..NL), five per pass of division by 58^5
template <int NL, int NP>
VK_DEV void base58_digits(u32 (&L)[NL], u32 (&dig)[5 * NP])
{
#pragma unroll
    for (int pass = 0; pass < NP; pass++) {
        u64 rem = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
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
}

template <int ND>
VK_DEV u32 significant_digits(const u32 (&dig)[ND])
{
    u32 n = 0;
#pragma unroll
    for (int i = 0; i < ND; i++) n = dig[i] != 0 ? (u32)(i + 1) : n;
    return n;
}

// [len][Base58Check(0x00 || h160)] as a column value
VK_DEV u32 rec_put_p2pkh(u64 *slot, const u32 h[5])
{
    u32 hb[5];
#pragma unroll
    for (int i = 0; i < 5; i++) hb[i] = __builtin_bswap32(h[i]);
    u32 w[16], d1[8], d2[8];
    w[0] = hb[0] >> 8;
#pragma unroll
    for (int i = 1; i < 5; i++) w[i] = (hb[i - 1] << 24) | (hb[i] >> 8);
    w[5] = (hb[4] << 24) | 0x00800000u;
#pragma unroll
    for (int i = 6; i < 15; i++) w[i] = 0;
    w[15] = 21 * 8;
    sha256_oneblock(d1, w);
    sha256_of_digest(d2, d1);
    u32 L[7] = {0u, hb[0], hb[1], hb[2], hb[3], hb[4], d2[0]};
    u32 dig[35];
    base58_digits<7, 7>(L, dig);
    u32 zeros = 1; // one '1' per leading zero byte of the payload: the version byte, then zero bytes of the hash160
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
    const u32 nd = significant_digits(dig);
    vk_stage_writer wr;
    wr.init(slot);
    wr.put(zeros + nd, 4);
    for (u32 i = 0; i < zeros; i++) wr.put('1', 1);
#pragma unroll
    for (int pos = 34; pos >= 0; pos--)
        if ((u32)pos < nd) wr.put(base58_char(dig[pos]), 1);
    return rec_finish(wr);
}

// [len][WIF]: Base58Check(0x80 || key [|| 0x01]); kb = the key as 8 big-endian words.  52 / 51 characters always (the
// leading byte 0x80 fixes the magnitude: 'K' / 'L' and '5' first characters).
template <bool COMPRESSED>
VK_DEV void rec_emit_wif(vk_stage_writer &wr, const u32 kb[8])
{
    u32 w[16], d1[8], d2[8];
    w[0] = 0x80000000u | (kb[0] >> 8);
#pragma unroll
    for (int i = 1; i < 8; i++) w[i] = (kb[i - 1] << 24) | (kb[i] >> 8);
    w[8] = (kb[7] << 24) | (COMPRESSED ? 0x00018000u : 0x00800000u);
#pragma unroll
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = (COMPRESSED ? 34 : 33) * 8;
    sha256_oneblock(d1, w);
    sha256_of_digest(d2, d1);
    u32 L[10];
    if (COMPRESSED) { // 38 bytes = 304 bits: L[0] holds the top 16 bits
        L[0] = 0x8000u | (kb[0] >> 24);
#pragma unroll
        for (int j = 1; j < 8; j++) L[j] = (kb[j - 1] << 8) | (kb[j] >> 24);
        L[8] = (kb[7] << 8) | 0x01u;
    } else { // 37 bytes = 296 bits: L[0] holds the top 8 bits
        L[0] = 0x80u;
#pragma unroll
        for (int j = 1; j < 9; j++) L[j] = kb[j - 1];
    }
    L[9] = d2[0];
    u32 dig[55];
    base58_digits<10, 11>(L, dig);
    constexpr int N = COMPRESSED ? 52 : 51;
#pragma unroll
    for (int pos = N - 1; pos >= 0; pos--) wr.put(base58_char(dig[pos]), 1);
}
template <bool COMPRESSED>
VK_DEV u32 rec_put_wif(u64 *slot, const u32 kb[8])
{
    vk_stage_writer wr;
    wr.init(slot);
    wr.put(COMPRESSED ? 52 : 51, 4);
    rec_emit_wif<COMPRESSED>(wr, kb);
    return rec_finish(wr);
}

// [42]["bc1" + bech32 data of (witness version 0, hash160) + checksum]  (BIP-173; src/derive.rs:116-118)
static __device__ const char VK_BECH32_CHARSET[33] = "qpzry9x8gf2tvdw0s3jn54khce6mua7l";
VK_DEV void rec_emit_p2wpkh(vk_stage_writer &wr, const u32 h[5])
{
    u32 hb[6];
#pragma unroll
    for (int i = 0; i < 5; i++) hb[i] = __builtin_bswap32(h[i]);
    hb[5] = 0;
    u32 data[39];
    data[0] = 0;
#pragma unroll
    for (int g = 0; g < 32; g++) {
        const int p = 5 * g, wi = p >> 5, sh = p & 31;
        const u64 two = ((u64)hb[wi] << 32) | hb[wi + 1];
        data[1 + g] = (u32)(two >> (64 - sh - 5)) & 31u;
    }
    const u32 GEN[5] = {0x3b6a57b2u, 0x26508e6du, 0x1ea119fau, 0x3d4233ddu, 0x2a1462b3u};
    u32 chk = 1;
    auto step = [&](u32 v) {
        const u32 b = chk >> 25;
        chk = ((chk & 0x1ffffffu) << 5) ^ v;
#pragma unroll
        for (int i = 0; i < 5; i++) chk ^= ((b >> i) & 1u) ? GEN[i] : 0u;
    };
    step(3); // hrp "bc" expanded: high bits of 'b', 'c', 0, low bits
    step(3);
    step(0);
    step(2);
    step(3);
#pragma unroll
    for (int i = 0; i < 33; i++) step(data[i]);
#pragma unroll
    for (int i = 0; i < 6; i++) step(0);
    chk ^= 1u;
#pragma unroll
    for (int i = 0; i < 6; i++) data[33 + i] = (chk >> (5 * (5 - i))) & 31u;
    wr.put('b' | ('c' << 8) | ('1' << 16), 3);
#pragma unroll
    for (int i = 0; i < 39; i++) wr.put((u32)(unsigned char)VK_BECH32_CHARSET[data[i]], 1);
}
VK_DEV u32 rec_put_p2wpkh(u64 *slot, const u32 h[5])
{
    vk_stage_writer wr;
    wr.init(slot);
    wr.put(42, 4);
    rec_emit_p2wpkh(wr, h);
    return rec_finish(wr);
}

// decimal digits of the 256-bit big-endian integer in L, least significant first, nine per pass of division by 10^9
VK_DEV void decimal_digits256(u32 (&L)[8], u32 (&dig)[81])
{
#pragma unroll
    for (int pass = 0; pass < 9; pass++) {
        u64 rem = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u64 cur = (rem << 32) | L[i];
            L[i] = (u32)(cur / 1000000000u);
            rem = cur % 1000000000u;
        }
        u32 r = (u32)rem;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            dig[9 * pass + k] = r % 10u;
            r /= 10u;
        }
    }
}

VK_DEV u32 decimal_length64(u64 v)
{
    u32 n = 1;
    u64 t = 10;
#pragma unroll
    for (int i = 1; i < 20; i++) {
        n += v >= t ? 1u : 0u;
        t = i < 19 ? t * 10u : t;
    }
    return n;
}

__global__ void __launch_bounds__(256) vk_records_key_kernel(u64 lo, u64 items, u32 slots, const u32 *__restrict__ priv, const unsigned char *__restrict__ valid,
                                                            const unsigned char *__restrict__ texts /* slots label suffixes */, vk_record_slots out)
{
    const u64 item = (u64)blockIdx.x * 256 + threadIdx.x;
    if (item >= items) return;
    if (!valid[item]) {
        out.lens[VK_RC_SOURCE][item] = 0;
        out.lens[VK_RC_KEY_RAW][item] = 0;
        out.lens[VK_RC_KEY_HEX][item] = 0;
        out.lens[VK_RC_KEY_DECIMAL][item] = 0;
        out.lens[VK_RC_KEY_BINARY][item] = 0;
        out.lens[VK_RC_BIT_LENGTH][item] = 0;
        out.lens[VK_RC_HAMMING_WEIGHT][item] = 0;
        out.lens[VK_RC_LEADING_ZEROS][item] = 0;
        out.lens[VK_RC_WIF_C][item] = 0;
        out.lens[VK_RC_WIF_U][item] = 0;
        return;
    }
    const u64 input = item / slots;
    const u32 sub = (u32)(item - input * slots);
    u32 raw[8], kb[8]; // the key's bytes in memory order / as big-endian words
#pragma unroll
    for (int i = 0; i < 8; i++) {
        raw[i] = priv[item * 8 + i];
        kb[i] = __builtin_bswap32(raw[i]);
    }
    vk_stage_writer wr;
    if (!texts) {
        out.lens[VK_RC_SOURCE][item] = 0; // the caller's column (vk_derive_columns)
    } else { // source: "{input}{suffix}"
        const unsigned char *sfx = texts + (u64)sub * VK_LINE_SUFFIX_BYTES;
        wr.init(out.slot[VK_RC_SOURCE] + item * (vk_rc_stride(VK_RC_SOURCE) / 8));
        wr.put(decimal_length64(lo + input) + sfx[0], 4);
        lines_put_decimal(wr, lo + input);
        lines_put_text(wr, sfx);
        out.lens[VK_RC_SOURCE][item] = rec_finish(wr);
    }
    { // raw
        u64 *p = out.slot[VK_RC_KEY_RAW] + item * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) p[i] = (u64)raw[2 * i] | ((u64)raw[2 * i + 1] << 32);
        out.lens[VK_RC_KEY_RAW][item] = 32;
    }
    { // hex
        wr.init(out.slot[VK_RC_KEY_HEX] + item * (vk_rc_stride(VK_RC_KEY_HEX) / 8));
        wr.put(64, 4);
#pragma unroll
        for (int i = 0; i < 8; i++) wr.put(lines_hex8(raw[i]), 8);
        out.lens[VK_RC_KEY_HEX][item] = rec_finish(wr);
    }
    { // decimal (src/derive.rs:121)
        u32 L[8], dig[81];
#pragma unroll
        for (int i = 0; i < 8; i++) L[i] = kb[i];
        decimal_digits256(L, dig);
        u32 nd = significant_digits(dig);
        nd = nd ? nd : 1u;
        wr.init(out.slot[VK_RC_KEY_DECIMAL] + item * (vk_rc_stride(VK_RC_KEY_DECIMAL) / 8));
        wr.put(nd, 4);
#pragma unroll
        for (int pos = 77; pos >= 0; pos--)
            if ((u32)pos < nd) wr.put('0' + dig[pos], 1);
        out.lens[VK_RC_KEY_DECIMAL][item] = rec_finish(wr);
    }
    { // binary (src/derive.rs:124-131): 256 characters, most significant bit first
        wr.init(out.slot[VK_RC_KEY_BINARY] + item * (vk_rc_stride(VK_RC_KEY_BINARY) / 8));
        wr.put(256, 4);
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
            for (int b = 3; b >= 0; b--) {
                const u32 byte = (kb[i] >> (8 * b)) & 0xffu;
                u64 chars = 0x3030303030303030ull;
#pragma unroll
                for (int j = 0; j < 8; j++) chars |= (u64)((byte >> (7 - j)) & 1u) << (8 * j);
                wr.put(chars, 8);
            }
        }
        out.lens[VK_RC_KEY_BINARY][item] = rec_finish(wr);
    }
    { // bit length, Hamming weight, leading zero hex digits (src/derive.rs:134-148)
        u32 clz = 0, pop = 0;
        bool lead = true;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            pop += __builtin_popcount(kb[i]);
            if (lead) {
                clz += kb[i] ? (u32)__builtin_clz(kb[i]) : 32u;
                lead = kb[i] == 0;
            }
        }
        out.slot[VK_RC_BIT_LENGTH][item] = 256u - clz;
        out.slot[VK_RC_HAMMING_WEIGHT][item] = pop;
        out.slot[VK_RC_LEADING_ZEROS][item] = clz >> 2;
        out.lens[VK_RC_BIT_LENGTH][item] = 4;
        out.lens[VK_RC_HAMMING_WEIGHT][item] = 4;
        out.lens[VK_RC_LEADING_ZEROS][item] = 4;
    }
    out.lens[VK_RC_WIF_C][item] = rec_put_wif<true>(out.slot[VK_RC_WIF_C] + item * (vk_rc_stride(VK_RC_WIF_C) / 8), kb);
    out.lens[VK_RC_WIF_U][item] = rec_put_wif<false>(out.slot[VK_RC_WIF_U] + item * (vk_rc_stride(VK_RC_WIF_U) / 8), kb);
}

__global__ void __launch_bounds__(256) vk_records_addr_kernel(u64 items, const u32 *__restrict__ h160c, const u32 *__restrict__ h160u,
                                                             const u32 *__restrict__ pub /* X || Y, 16 words in byte order */,
                                                             const unsigned char *__restrict__ valid, vk_record_slots out)
{
    const u64 item = (u64)blockIdx.x * 256 + threadIdx.x;
    if (item >= items) return;
    if (!valid[item]) {
        out.lens[VK_RC_PUBKEY_C][item] = 0;
        out.lens[VK_RC_PUBKEY_U][item] = 0;
        out.lens[VK_RC_P2PKH_C][item] = 0;
        out.lens[VK_RC_P2PKH_U][item] = 0;
        out.lens[VK_RC_P2WPKH][item] = 0;
        return;
    }
    u32 hc[5], hu[5], xy[16];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        hc[i] = h160c[item * 5 + i];
        hu[i] = h160u[item * 5 + i];
    }
#pragma unroll
    for (int i = 0; i < 16; i++) xy[i] = pub[item * 16 + i];
    vk_stage_writer wr;
    { // SEC1 compressed: 02 / 03 by the parity of Y, then X (src/derive.rs:98,150)
        wr.init(out.slot[VK_RC_PUBKEY_C] + item * (vk_rc_stride(VK_RC_PUBKEY_C) / 8));
        wr.put(66, 4);
        wr.put('0' | ((((xy[15] >> 24) & 1u) ? '3' : '2') << 8), 2);
#pragma unroll
        for (int i = 0; i < 8; i++) wr.put(lines_hex8(xy[i]), 8);
        out.lens[VK_RC_PUBKEY_C][item] = rec_finish(wr);
    }
    { // SEC1 uncompressed: 04 X Y (src/derive.rs:99,151)
        wr.init(out.slot[VK_RC_PUBKEY_U] + item * (vk_rc_stride(VK_RC_PUBKEY_U) / 8));
        wr.put(130, 4);
        wr.put('0' | ('4' << 8), 2);
#pragma unroll
        for (int i = 0; i < 16; i++) wr.put(lines_hex8(xy[i]), 8);
        out.lens[VK_RC_PUBKEY_U][item] = rec_finish(wr);
    }
    out.lens[VK_RC_P2PKH_C][item] = rec_put_p2pkh(out.slot[VK_RC_P2PKH_C] + item * (vk_rc_stride(VK_RC_P2PKH_C) / 8), hc);
    out.lens[VK_RC_P2PKH_U][item] = rec_put_p2pkh(out.slot[VK_RC_P2PKH_U] + item * (vk_rc_stride(VK_RC_P2PKH_U) / 8), hu);
    out.lens[VK_RC_P2WPKH][item] = rec_put_p2wpkh(out.slot[VK_RC_P2WPKH] + item * (vk_rc_stride(VK_RC_P2WPKH) / 8), hc);
}

// ConsoleOutput's verbose block (src/output/console.rs:62-76), one lane per (input, slot) into a 512-byte slot:
//   ---\nsource: ..\ntransform: ..\nprivate_key: ..\nwif_compressed: ..\nwif_uncompressed: ..\np2pkh_compressed: ..\n
//   p2pkh_uncompressed: ..\np2wpkh: ..\n                                  (483 bytes at most)
#define VK_VERBOSE_STRIDE 512u
VK_DEV void put_literal(vk_stage_writer &wr, const char *s, u32 n) // n = strlen(s), a compile-time string
{
    for (u32 i = 0; i < n; i += 8) {
        u64 v = 0;
        const u32 m = n - i < 8 ? n - i : 8;
        for (u32 j = 0; j < m; j++) v |= (u64)(unsigned char)s[i + j] << (8 * j);
        wr.put(v, m);
    }
}
#define PUT_LIT(wr, s) put_literal(wr, s, sizeof(s) - 1)

__global__ void __launch_bounds__(256) vk_verbose_kernel(u64 lo, u64 items, u32 slots, const u32 *__restrict__ priv, const u32 *__restrict__ h160c,
                                                        const u32 *__restrict__ h160u, const unsigned char *__restrict__ valid,
                                                        const unsigned char *__restrict__ texts /* slots suffix entries, then the transform name */,
                                                        u64 *__restrict__ block_slots, u32 *__restrict__ lens)
{
    const u64 item = (u64)blockIdx.x * 256 + threadIdx.x;
    if (item >= items) return;
    if (!valid[item]) {
        lens[item] = 0;
        return;
    }
    const u64 input = item / slots;
    const u32 sub = (u32)(item - input * slots);
    u32 raw[8], kb[8], hc[5], hu[5];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        raw[i] = priv[item * 8 + i];
        kb[i] = __builtin_bswap32(raw[i]);
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
        hc[i] = h160c[item * 5 + i];
        hu[i] = h160u[item * 5 + i];
    }
    vk_stage_writer wr;
    wr.init(block_slots + item * (VK_VERBOSE_STRIDE / 8));
    PUT_LIT(wr, "---\nsource: ");
    lines_put_decimal(wr, lo + input);
    lines_put_text(wr, texts + (u64)sub * VK_LINE_SUFFIX_BYTES);
    PUT_LIT(wr, "\ntransform: ");
    lines_put_text(wr, texts + (u64)slots * VK_LINE_SUFFIX_BYTES);
    PUT_LIT(wr, "\nprivate_key: ");
#pragma unroll
    for (int i = 0; i < 8; i++) wr.put(lines_hex8(raw[i]), 8);
    PUT_LIT(wr, "\nwif_compressed: ");
    rec_emit_wif<true>(wr, kb);
    PUT_LIT(wr, "\nwif_uncompressed: ");
    rec_emit_wif<false>(wr, kb);
    PUT_LIT(wr, "\np2pkh_compressed: ");
    lines_put_p2pkh(wr, hc);
    PUT_LIT(wr, "\np2pkh_uncompressed: ");
    lines_put_p2pkh(wr, hu);
    PUT_LIT(wr, "\np2wpkh: ");
    rec_emit_p2wpkh(wr, hc);
    wr.put('\n', 1);
    lens[item] = rec_finish(wr);
}
