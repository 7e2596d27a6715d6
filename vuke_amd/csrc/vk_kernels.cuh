// vk_kernels.cuh -- the fused sweep kernel: seed -> key -> k*G -> hash160 -> probe, on gfx950.
//
// Shape of one workgroup tile (256 lanes, B = R * slots items per lane; tiles come from an atomic counter):
//   pass 1  for each item: transform -> normalise -> ecmult_table_lds_xyzz: the first two table entries are added in affine
//           form (4M + 2S), every later window adds its affine entry to an accumulator in XYZZ coordinates
//           (X, Y, ZZ, ZZZ: 8M + 2S); the entries arrive through the wave's LDS landing slot (global_load_lds_dwordx4,
//           four lanes fetch one 64-byte entry).  Parked per item in the workgroup's own slice of an HBM workspace laid out
//           [item][quad][lane] (every access a 16-byte-per-lane coalesced line): X*ZZ^2, Y, e = ZZZ and the running
//           product c_b = e_0..e_b.
//   invert  one field inversion of c_{B-1} per lane (Montgomery's trick: 3 multiplies per key instead of a 255-squaring
//           inversion per key); fe_inv_val is a real call with its operand and result in registers.
//   pass 2  walk the items backwards: w_b = inv * c_{b-1} = 1 / e_b; inv *= e_b; y = Y * w, x = (X*ZZ^2) * w^2;
//           SHA-256 + RIPEMD-160 of the SEC1 encoding(s); digest fold; open-address probe of the target table; hit
//           record / generate-mode record.  e and c of the next item are requested one item ahead.
// LDS: the tile index, one 4 160-byte landing slot per wave, the per-lane digest accumulators (8 KB) and, when uncompressed keys are hashed,
// the 53 KB table of the second SHA-256 block's schedule (dynamic).  No MFMA: the path is integer-VALU bound; the memory
// streams are the table gathers (64 B per window, random over the whole HBM table), the workspace (128 B written,
// 160 B read per key) and one 32-byte probe slot per hash160.  Every field element lives in VGPRs; the 16 bytes of scratch
// the code object reports are the argument frame of the rare exact-product calls (fe_mul_exact / fe_sqr_exact).
//
// Replaces the closure body of RangeSource::process (reference src/source/range.rs:42-70):
// Transform::apply_batch -> KeyDeriver::derive -> Matcher::check -> counters.
#pragma once
#include "vk_ec.cuh"
#include "vk_transforms.cuh"
#include "vk_armory.cuh"

#define VK_F_COMPRESSED 1u
#define VK_F_UNCOMPRESSED 2u
#define VK_F_EMIT 0x10000u /* internal: generate-mode records */

#define VK_BLOCK 256
#define VK_WIDE_BLOCK 768 /* the wide form of the sweep kernel: twelve waves, three per SIMD, one workgroup per CU */
#ifndef VK_PARK
#define VK_PARK 1 /* A/B switch of the LDS parking of c and k in the three-wave forms (vk_sweep_kernel) */
#endif
#ifndef VK_MILKSAD_PAIRS
#define VK_MILKSAD_PAIRS 1 /* pass 1 of the two-wave Milksad instantiation transforms its items two at a time (with three waves on a SIMD the serial
                              MT19937 chain is covered by the other two: pairs are 1.3 % SLOWER there, profiles/r05_three_waves_again.txt) */
#endif

struct vk_hit_dev { // must match vk_hit in include/vuke_hip.h (72 bytes)
    u64 seed;
    u32 sub_index;
    u32 addr_kind;
    u32 privkey[8]; // raw bytes, i.e. big-endian words already byte-swapped for memory
    u32 h160[5];
    u32 pad;
};

struct vk_sweep_params {
    vk_spec_dev spec;
    u64 lo;         // first input
    u64 span;       // last input - lo (inclusive range => span + 1 inputs)
    u64 ntiles;
    u32 slots;      // slots per input
    u32 R;          // inputs per lane per tile
    u32 flags;
    u32 nwin;
    u32 filt_mask;  // bits of the pre-filter - 1 (power of two), valid when tgt != nullptr
    u32 tgt_mask;   // capacity - 1 (power of two), valid when tgt != nullptr
    u32 hit_cap;
    u32 pad0;
    const uint4 *table;
    const uint4 *tgt;  // slots of 2 x uint4: {h0,h1,h2,h3}, {h4, kindbits, 0, 0}
    const u32 *filt;   // pre-filter of the target table: bit (h1 & filt_mask) is set for every target (probe_filter_word)
    vk_hit_dev *hits;
    u64 *stats;        // [0] keys [1] matches [2] hit cursor [3..6] digest [7] tile counter [8] shader cycles [9] 100 MHz ticks of workgroup 0
    uint4 *work;       // batch-inversion workspace
    const u32 *in_keys;   // VK_T_RAWKEYS: n x 8 words (raw bytes)
    const unsigned char *in_valid; // VK_T_RAWKEYS: optional n flags, 0 = no key in this slot (vk_scan_strings)
    u32 *out_priv;     // generate mode: n_items x 8 words (raw bytes)
    u32 *out_h160c;    // n_items x 5 words
    u32 *out_h160u;
    unsigned char *out_valid;
    u32 *out_pub;      // n_items x 16 words: affine X || Y, big-endian bytes (vk_derive_batch_pub)
    vk_win_desc wins[VK_MAX_WIN];
};

VK_DEV u32 probe_targets(const uint4 *__restrict__ tgt, u32 mask, const u32 h[5])
{
    u32 i = h[0] & mask;
#pragma unroll 1
    for (u32 n = 0; n <= mask; n++) {
        uint4 a = tgt[2 * (size_t)i], b = tgt[2 * (size_t)i + 1];
        if (b.y == 0) return 0; // empty slot
        if (a.x == h[0] && a.y == h[1] && a.z == h[2] && a.w == h[3] && b.x == h[4]) return b.y;
        i = (i + 1) & mask;
    }
    return 0;
}

// Pre-filter in front of the open-address table (round 6).  A miss in the table costs a CHAIN of dependent loads per lane -- with a
// million targets at load factor 1/2 an unsuccessful linear probe reads 2.4 slots on average, and a wave waits for its longest lane
// -- every one of them exposed, twice per key when both key forms are hashed: a quarter of config 3's wave-cycles were spent parked
// in those waits (profiles/r06_forms_pmc.txt).  The filter is a bitmap of >= 128 bits per target indexed by the hash160's second
// word: ONE independent 4-byte load per hash160, issued when the hash is done and looked at after the other key form has been hashed;
// 99.2 % of the keys stop there, the rest (true hits and the filter's false positives) take the exact probe.  What is reported is
// decided by the exact table alone: the filter only skips hash160s that cannot be in it.
VK_DEV u32 probe_filter_word(const u32 *__restrict__ filt, u32 fmask, const u32 h[5]) { return filt[(h[1] & fmask) >> 5]; }
VK_DEV bool probe_filter_hit(u32 word, const u32 h[5]) { return ((word >> (h[1] & 31u)) & 1u) != 0; }

// The digest accumulators live in LDS, one column per lane ([word][lane]: conflict-free 8-byte accesses): they are touched
// once per key, and as registers they were 8 VGPRs held across the whole tile loop -- the difference between 0 and a few
// spilled registers in the instantiations whose transform carries more state (mt64, raw keys, double_sha256, sha256_chain).
template <int TL>
VK_DEV void digest_fold(u64 *dg /* LDS, this lane's column, stride TL (the lanes of a tile) */, const u32 h[5])
{
    u64 a = (u64)h[0] | ((u64)h[1] << 32), b = (u64)h[2] | ((u64)h[3] << 32);
    dg[0] ^= a;
    dg[TL] ^= b;
    dg[2 * TL] ^= (u64)h[4];
    dg[3 * TL] += a + b + h[4];
}

VK_DEV void work_store(uint4 *base, u64 stride, const fe &f)
{
    base[0] = make_uint4(f.v[0], f.v[1], f.v[2], f.v[3]);
    base[stride] = make_uint4(f.v[4], f.v[5], f.v[6], f.v[7]);
}
VK_DEV void work_load(fe &f, const uint4 *base, u64 stride)
{
    uint4 a = base[0], b = base[stride];
    f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
    f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}

VK_DEV u64 wave_xor(u64 v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
    return v;
}
VK_DEV u64 wave_add(u64 v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// produce the key words for item (off, sub); returns false where the reference emits nothing
template <u32 KIND>
VK_DEV bool item_key(u32 kw[8], const vk_sweep_params &p, u64 off, u32 sub)
{
    if (off > p.span) return false;
    if (KIND == VK_T_RAWKEYS) {
        if (p.in_valid && !p.in_valid[off]) return false;
        const uint4 *src = reinterpret_cast<const uint4 *>(p.in_keys) + off * 2;
        uint4 a = src[0], b = src[1];
        kw[0] = bswap32(a.x); kw[1] = bswap32(a.y); kw[2] = bswap32(a.z); kw[3] = bswap32(a.w);
        kw[4] = bswap32(b.x); kw[5] = bswap32(b.y); kw[6] = bswap32(b.z); kw[7] = bswap32(b.w);
        return true;
    }
    if (KIND == VK_T_ARMORY) {
        const vk_ec_ctx ec = {p.table, p.wins, p.nwin};
        return transform_armory(kw, p.lo + off, sub, ec);
    }
    return transform_key<KIND>(kw, p.spec, p.lo + off, sub);
}

// pass 2 of one tile: walk the B items of this lane backwards from `inv` = 1 / (e_0 .. e_{B-1}); (r, sub) = the (input row, slot)
// counters as pass 1 left them.  e and the running product of the item before it are what the two multiplies at the top of the
// body wait for: they are requested one item ahead (16 registers), X and Y at the top of the body (needed two multiplies later).
template <u32 KIND, bool UNCOMP, int TL>
VK_DEV void sweep_pass2(const vk_sweep_params &p, const uint4 *wk, const u32 B, const u64 tile_off, fe inv, u32 r, u32 sub, u64 *dg, u32 &n_keys,
                        u32 &n_match, vk_lds_u4p tail)
{
    const u32 tid = TL == 64 ? (threadIdx.x & 63u) : threadIdx.x; // the lane's place in its tile
    const u64 nth = TL;
    fe nz, ncp;
    work_load(nz, wk + (u64)(B - 1) * 8 * nth + 4 * nth, nth);
    if (B > 1) work_load(ncp, wk + (u64)(B - 2) * 8 * nth + 6 * nth, nth);
#pragma unroll 1
    for (u32 bb = B; bb > 0; bb--) {
        const u32 b = bb - 1;
        if (sub == 0) {
            sub = p.slots - 1;
            r--;
        } else {
            sub--;
        }
        const uint4 *it = wk + (u64)b * 8 * nth;
        jac P;
        fe zinv;
        work_load(P.x, it, nth);
        work_load(P.y, it + 2 * nth, nth);
        P.z = nz;
        const fe cp = ncp;
        if (b > 0) {
            work_load(nz, it - 4 * nth, nth);                 // e of item b-1
            if (b > 1) work_load(ncp, it - 10 * nth, nth);    // c of item b-2
            fe_mul(zinv, inv, cp);
            fe_mul(inv, inv, P.z);
        } else {
            zinv = inv;
        }
        // a lane without a key parked (0, 0); a real Y is a non-zero residue (no curve point has y = 0), so Y alone tells them apart
        const u32 nzw = (P.y.v[0] | P.y.v[1] | P.y.v[2]) | (P.y.v[3] | P.y.v[4] | P.y.v[5]) | (P.y.v[6] | P.y.v[7]);
        if (nzw == 0) continue;
        fe x, y;
        fe_mul(y, P.y, zinv); // P.x = X*ZZ^2, P.y = Y, zinv = 1 / ZZZ
        fe_sqr(zinv, zinv);
        fe_mul(x, P.x, zinv);
        fe_canon(x);
        fe_canon(y);
        n_keys++;
        const u64 off = tile_off + (u64)r * TL + tid;
        u32 hc[5], hu[5];
        u32 bits_c = 0, bits_u = 0;
        u32 fw_c = 0, fw_u = 0; // the pre-filter's words: requested as soon as a hash160 exists, looked at after both
        if (!UNCOMP || (p.flags & VK_F_COMPRESSED)) { // an instantiation without the uncompressed form only runs compressed-only launches
            hash160_pub33(hc, x, y.v[0] & 1u);
            digest_fold<TL>(dg, hc);
            if (p.tgt) fw_c = probe_filter_word(p.filt, p.filt_mask, hc);
        }
        if (UNCOMP && (p.flags & VK_F_UNCOMPRESSED)) {
            hash160_pub65_lds(hu, x, y, tail);
            digest_fold<TL>(dg, hu);
            if (p.tgt) fw_u = probe_filter_word(p.filt, p.filt_mask, hu);
        }
        if (p.tgt) { // fw_* stay 0 for a key form that was not hashed
            if (fw_c && probe_filter_hit(fw_c, hc)) bits_c = probe_targets(p.tgt, p.tgt_mask, hc);
            if (UNCOMP && fw_u && probe_filter_hit(fw_u, hu)) bits_u = probe_targets(p.tgt, p.tgt_mask, hu);
        }
        // Matcher::check order (reference src/matcher.rs:72-98): p2pkh(c), p2pkh(u), p2wpkh
        int kind = -1;
        if (bits_c & 1u) kind = 0;
        else if (bits_u & 1u) kind = 1;
        else if (bits_c & 2u) kind = 2;
        const bool emit = (p.flags & VK_F_EMIT) != 0;
        if (kind >= 0 || emit) {
            // rare (or output-bound) path: regenerate the normalised key bytes
            u32 kw[8], k[8];
            item_key<KIND>(kw, p, off, sub);
#pragma unroll
            for (int i = 0; i < 8; i++) k[i] = kw[7 - i];
            scalar_normalize(k);
            if (kind >= 0) {
                n_match++;
                u64 slot = atomicAdd((unsigned long long *)&p.stats[2], 1ull);
                if (slot < p.hit_cap) {
                    vk_hit_dev *h = p.hits + slot;
                    h->seed = p.lo + off;
                    h->sub_index = sub;
                    h->addr_kind = (u32)kind;
#pragma unroll
                    for (int i = 0; i < 8; i++) h->privkey[i] = bswap32(k[7 - i]);
#pragma unroll
                    for (int i = 0; i < 5; i++) h->h160[i] = (UNCOMP && kind == 1) ? hu[i] : hc[i];
                    h->pad = 0;
                }
            }
            if (emit) {
                const u64 item = off * p.slots + sub;
                if (p.out_valid) p.out_valid[item] = 1;
                if (p.out_priv) {
#pragma unroll
                    for (int i = 0; i < 8; i++) p.out_priv[item * 8 + i] = bswap32(k[7 - i]);
                }
                if (p.out_h160c && (!UNCOMP || (p.flags & VK_F_COMPRESSED))) {
#pragma unroll
                    for (int i = 0; i < 5; i++) p.out_h160c[item * 5 + i] = hc[i];
                }
                if (UNCOMP && p.out_h160u && (p.flags & VK_F_UNCOMPRESSED)) {
#pragma unroll
                    for (int i = 0; i < 5; i++) p.out_h160u[item * 5 + i] = hu[i];
                }
                if (p.out_pub) {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        p.out_pub[item * 16 + i] = bswap32(FE_BE(x, i));
                        p.out_pub[item * 16 + 8 + i] = bswap32(FE_BE(y, i));
                    }
                }
            }
        }
    }
}

// counters of one workgroup into p.stats: one wave-reduced atomic per wave and counter
template <int TL>
VK_DEV void sweep_counters(const vk_sweep_params &p, const u64 *dg, u32 n_keys, u32 n_match)
{
    const u32 tid = threadIdx.x;
    const u64 w_keys = wave_add((u64)n_keys), w_match = wave_add((u64)n_match);
    const u64 d0 = wave_xor(dg[0]), d1 = wave_xor(dg[TL]), d2 = wave_xor(dg[2 * TL]), d3 = wave_add(dg[3 * TL]);
    if ((tid & 63) == 0) {
        atomicAdd((unsigned long long *)&p.stats[0], (unsigned long long)w_keys);
        atomicAdd((unsigned long long *)&p.stats[1], (unsigned long long)w_match);
        atomicXor((unsigned long long *)&p.stats[3], (unsigned long long)d0);
        atomicXor((unsigned long long *)&p.stats[4], (unsigned long long)d1);
        atomicXor((unsigned long long *)&p.stats[5], (unsigned long long)d2);
        atomicAdd((unsigned long long *)&p.stats[6], (unsigned long long)d3);
    }
}

// the parking area of the forms that have one; the others declare no LDS for it (not even a word: the dynamic schedule table behind
// the static LDS must stay 16-byte aligned)
template <int WORDS>
struct vk_park_lds {
    static __device__ __forceinline__ u32 *base()
    {
        __shared__ u32 s_park[WORDS];
        return s_park;
    }
};
template <>
struct vk_park_lds<0> {
    static __device__ __forceinline__ u32 *base() { return nullptr; }
};

// WAVES = the waves per SIMD the instantiation is register-budgeted for.  2: every field element in VGPRs (196 - 206 of them);
// 3: 168 VGPRs, the coldest values in scratch.  Round 4 measured three waves 4 % SLOWER than two (37 - 46 spilled registers then, and
// a part that gave the gain back in clock); after round 5's diet the same experiment is +2.2 % on the Milksad sweep and +2.8 % on the
// LCG sweep (profiles/r05_three_waves_again.txt).  Four waves: 315 spilled registers, -13 %.
// UNCOMP = the instantiation holds the code that hashes uncompressed keys (chosen per launch by p.flags); without it the kernel
// runs compressed-only launches and carries none of that code (round 6: a template parameter, not a launch flag over shared code).
// WIDE = one workgroup of 768 lanes per CU instead of three of 256 (round 6): three waves per SIMD for the launches that hash
// uncompressed keys, whose 53 KB schedule table does not fit a CU's LDS three times (3 x 78 KB) but fits once beside twelve waves'
// landing slots and digest columns (53 + 12 x 6.2 = 127 KB).  Twelve waves would meet in every barrier of the tile loop, so the
// wide form has none: a tile is 64 lanes x R inputs and every WAVE takes its own from the counter (TL = the lanes of a tile).
template <u32 KIND, int WAVES, bool UNCOMP, bool WIDE>
__global__ void __launch_bounds__(WIDE ? VK_WIDE_BLOCK : VK_BLOCK, WAVES) vk_sweep_kernel(const vk_sweep_params p)
{
    constexpr int NT = WIDE ? VK_WIDE_BLOCK : VK_BLOCK; // lanes of the workgroup
    constexpr int TL = WIDE ? 64 : VK_BLOCK;             // lanes of a tile = of a workspace slice = of a digest column block
    const u32 tid = WIDE ? (threadIdx.x & 63u) : threadIdx.x;                      // the lane's place in its tile
    const u32 unit = WIDE ? blockIdx.x * (NT / TL) + (threadIdx.x >> 6) : blockIdx.x; // which slice of the workspace
    const u32 B = p.R * p.slots;
    // workgroup 0 is resident for the whole launch (tiles come from a counter): its shader-cycle and constant-rate counters
    // give the clock the part actually held under this load
    const u64 clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    u32 n_keys = 0, n_match = 0; // per lane and launch: far below 2^32
    __shared__ u64 s_dg[4 * NT];
    u64 *dg = s_dg + (WIDE ? (threadIdx.x >> 6) * (4 * TL) : 0u) + tid;
#pragma unroll
    for (int i = 0; i < 4; i++) dg[i * TL] = 0;
    // workspace of this tile owner: item b, quad q, lane t at wk[(b*8 + q) * TL] -- one contiguous
    // B * TL * 128 B region (a few TLB entries), every access a fully coalesced 16 B per lane
    const u64 nth = TL;
    uint4 *wk = p.work + (u64)unit * ((u64)B * 8 * TL) + tid;
    // VK_PARK (the three-wave form of compressed-only launches): values that live across the whole elliptic-curve multiplication of an
    // item but are touched once per item -- the running product c and the scalar k, 8 words each -- wait in LDS ([word][lane] u32 columns
    // per wave, 16 KB per workgroup) instead of in registers the 168-VGPR budget does not have: 22 -> 12 spilled registers, Milksad sweep
    // +1.65 %, LCG sweep +-0, same box.  NOT in the wide form: parking c there (its LDS has no room for k as well) measured -6 % on the
    // sha256 scan (profiles/r06_lds_parking_experiment.txt).
    constexpr bool PARK_C = VK_PARK && WAVES >= 3 && !WIDE, PARK_K = PARK_C;
    constexpr int PARK_WORDS = (PARK_C ? 8 : 0) + (PARK_K ? 8 : 0);
    u32 *park = vk_park_lds<(NT / 64) * PARK_WORDS * 64>::base() + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * (PARK_WORDS * 64) + (threadIdx.x & 63u);
    __shared__ u64 s_tile;
    __shared__ uint4 s_gather[(NT / 64) * VK_LDS_DEPTH * VK_LTILE_FULL]; // whole-entry slots per wave (lds_gather_issue)
    // the wave's slot, as a scalar: the LDS-direct loads take their LDS address from M0
    uint4 *gtile = s_gather + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * (VK_LDS_DEPTH * VK_LTILE_FULL);
    // W[t] + K[t] of the second SHA-256 block of an uncompressed key, one row per value of Y's last byte (vk_hash.cuh): dynamic LDS,
    // VK_TAIL_LDS_BYTES of it when the launch hashes uncompressed keys (vk_sweep_tu.hip), none otherwise
    extern __shared__ vk_u32x4 s_tail[];
    const vk_lds_u4p tail = (vk_lds_u4p)s_tail;
    if (UNCOMP && (p.flags & VK_F_UNCOMPRESSED)) {
        if (!WIDE || threadIdx.x < 256) sha256_tail65_build_row(tail, threadIdx.x); // 256 rows
        if (WIDE) __syncthreads(); // the classic form's rows are made visible by the tile loop's first barrier
    }

#pragma unroll 1
    for (;;) {
        // tiles are handed out by an atomic counter: whoever runs ahead (clock, XCD, table locality) takes more
        u64 tile;
        if (WIDE) {
            u64 t = 0;
            if (tid == 0) t = atomicAdd((unsigned long long *)(p.stats + 7), 1ull);
            tile = ((u64)__builtin_amdgcn_readfirstlane((u32)(t >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)t);
        } else {
            if (tid == 0) s_tile = atomicAdd((unsigned long long *)(p.stats + 7), 1ull);
            __syncthreads();
            tile = s_tile;
            __syncthreads();
        }
        if (tile >= p.ntiles) break;
        const u64 tile_off = tile * ((u64)TL * p.R);
        // ---------------------------------------------------------------- pass 1
        fe c;
        u32 r = 0, sub = 0;
        u32 kw_next[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // Milksad: the key of item b + 1, made together with item b's (transform_milksad2)
        bool have_next = false;
#pragma unroll 1
        for (u32 b = 0; b < B; b++) {
            const u64 off = tile_off + (u64)r * TL + tid;
            u32 kw[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // stays zero where the reference emits nothing (item_key returns false)
            // parked per item: X*ZZ^2, Y, e = ZZZ and the running product of e: y = Y / e, x = X*ZZ^2 / e^2 (ZZ^3 = ZZZ^2)
            uint4 *it = wk + (u64)b * 8 * nth;
            {
                bool have;
                if (KIND == VK_T_MILKSAD && VK_MILKSAD_PAIRS && WAVES < 3) {
                    // one slot per input: item b + 1 is the input TL further on.  Even items transform both (two interleaved
                    // MT19937 chains), odd items take what the even one left
                    if ((b & 1u) == 0) {
                        const u64 off1 = off + TL, sa = p.lo + off, sb = p.lo + off1;
                        transform_milksad2(kw, kw_next, (u32)sa, (u32)sb);
                        have = off <= p.span && sa <= 0xFFFFFFFFull;          // transform_milksad's skip rule
                        have_next = off1 <= p.span && sb <= 0xFFFFFFFFull;
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; i++) kw[i] = kw_next[i];
                        have = have_next;
                    }
                } else {
                    have = item_key<KIND>(kw, p, off, sub);
                }
                u32 k[8];
#pragma unroll
                for (int i = 0; i < 8; i++) k[i] = kw[7 - i];
                if (__builtin_amdgcn_ballot_w64(!have) != 0) { // rare (the ragged end of a range, seeds beyond a ceiling): lanes without a key walk the table with k = 1
#pragma unroll
                    for (int i = 0; i < 8; i++) k[i] = have ? k[i] : (i == 0 ? 1u : 0u);
                }
                scalar_normalize(k);
                xyzz Q;
                if (PARK_K) {
#pragma unroll
                    for (int i = 0; i < 8; i++) park[(8 + i) * 64] = k[i];
                    const vk_kpark kp = {park + 8 * 64};
                    ecmult_table_lds_xyzz(Q, kp, p.table, p.wins, p.nwin, gtile);
                } else {
                    ecmult_table_lds_xyzz(Q, k, p.table, p.wins, p.nwin, gtile);
                }
                fe e;
                if (have) {
                    fe_sqr(e, Q.zz);
                    fe_mul(Q.x, Q.x, e);
                    e = Q.zzz;
                } else { // (0, 0) with e = 1 marks "no key here"; (0,0) is not on the curve
                    fe_set_zero(Q.x);
                    fe_set_zero(Q.y);
                    fe_set_one(e);
                }
                if (PARK_C) {
                    if (b == 0) {
                        c = e;
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; i++) c.v[i] = park[i * 64];
                        fe_mul(c, c, e);
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) park[i * 64] = c.v[i];
                } else {
                    if (b == 0) c = e;
                    else fe_mul(c, c, e);
                }
                work_store(it, nth, Q.x);
                work_store(it + 2 * nth, nth, Q.y);
                work_store(it + 4 * nth, nth, e);
                work_store(it + 6 * nth, nth, c);
            }
            if (++sub == p.slots) {
                sub = 0;
                r++;
            }
        }
        // ---------------------------------------------------------------- invert
        fe inv;
        if (PARK_C) {
#pragma unroll
            for (int i = 0; i < 8; i++) c.v[i] = park[i * 64];
        }
        fe_inv(inv, c);
        // ---------------------------------------------------------------- pass 2
        // (as a kernel of its own, budgeted for 3 or 4 waves per SIMD, with every tile's items kept in HBM until it had read them,
        // pass 2 made the Milksad sweep 4.8 % SLOWER and the sha256 scan 0 - 1.5 % slower, same box, three rounds each:
        // profiles/archive/r04_split_pass2_experiment.txt; the variant's last source is commit fe19f6c)
        sweep_pass2<KIND, UNCOMP, TL>(p, wk, B, tile_off, inv, r, sub, dg, n_keys, n_match, tail);
    }
    // ------------------------------------------------------------------- counters
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        p.stats[8] = __builtin_amdgcn_s_memtime() - clk0;
        p.stats[9] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
    sweep_counters<TL>(p, dg, n_keys, n_match);
}
