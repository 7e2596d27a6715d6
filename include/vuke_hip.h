/*
 * vuke_hip.h -- C ABI of the MI355X (gfx950) batch key-derivation engine.
 *
 * This is the drop-in boundary for vuke's hot path.  The reference (oritwoen/vuke v0.9.0)
 * has no FFI of its own (pure-Rust trait objects); each entry point below names the
 * reference interface it stands in for, and INTEGRATION.md shows the Rust `extern "C"`
 * block + `impl Source for HipRangeSource` a maintainer would add.
 *
 *   reference interface (file:line)                         replaced by
 *   ------------------------------------------------------  ---------------------------
 *   Source::process / RangeSource::process                  vk_sweep, vk_generate
 *     src/source/mod.rs:23-31, src/source/range.rs:26-79
 *   ProcessStats  src/source/mod.rs:35-39                   vk_sweep_stats
 *   Transform::apply_batch (numeric inputs)                 vk_transform_batch
 *     src/transform/mod.rs:39-65
 *   TransformType  src/transform/mod.rs:68-98               vk_transform_spec
 *   KeyDeriver::derive  src/derive.rs:78-165                vk_derive_batch (hash160s), vk_derive_columns
 *                                                           (every field of the record, as columns)
 *   ConsoleOutput::key  src/output/console.rs:62-81         vk_generate_lines, vk_generate_verbose,
 *                                                           vk_generate_lines_strings (text made on the device)
 *   records_to_batch  src/storage/schema.rs:137-262         vk_generate_columns, vk_derive_columns
 *   Matcher::{from_addresses,check}  src/matcher.rs:62-98   vk_targets_set + on-device probe
 *   MatchInfo/AddressType  src/matcher.rs:13-36             vk_hit.addr_kind
 *   GpuError  src/gpu/error.rs:7-20                         VK_E_* + vk_last_error
 *
 * Conventions: plain pointers and sizes, caller owns every host buffer, the engine owns all
 * device memory.  Return 0 on success, negative VK_E_* on failure; nothing throws or aborts
 * across this boundary.  One engine drives one GPU and is internally serialised (safe to
 * share between threads, like the reference's Send + Sync traits); use one engine per GPU.
 * There is NO CPU fallback in this library: without a usable gfx950 device every call fails
 * with VK_E_NO_DEVICE.
 */
#ifndef VUKE_HIP_H
#define VUKE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VK_ABI_VERSION 5 /* 2: vk_sweep_stats.shader_clock_ghz, vk_scan_strings; 3: vk_generate_lines, vk_last_warning,
                           vk_sweep_stats.inputs_skipped (was padding), string limits 448 bytes, vk_engine_opts.algo must be 0;
                           4: vk_generate_lines_strings, vk_generate_verbose,
                           vk_generate_columns / vk_derive_columns + vk_record_columns;
                           5: vk_engine_info_t gains runtime_init_seconds / alloc_seconds / table_wall_seconds; the vk_debug_*
                           test hooks leave this header and the shipped library (include/vuke_hip_test_hooks.h,
                           libvuke_hip_hooks.so); electrum seeds and multibit lines of any length (was 448 bytes: inputs_skipped is always 0) */

/* error codes */
#define VK_OK 0
#define VK_E_NO_DEVICE (-1)
#define VK_E_HIP (-2)
#define VK_E_INVALID_ARG (-3)
#define VK_E_UNSUPPORTED (-4)
#define VK_E_NOMEM (-5)

/* transform kinds (TransformType, src/transform/mod.rs:68-98) */
#define VK_T_SHA256 0u       /* src/transform/sha256.rs:13-29: 3 keys per numeric input */
#define VK_T_MILKSAD 1u      /* src/transform/milksad.rs:17-29: MT19937, u32 seeds only */
#define VK_T_MT64 2u         /* src/mt64.rs:15-20 */
#define VK_T_LCG 3u          /* src/lcg.rs:68-128; variants 0 glibc, 1 minstd, 2 msvc, 3 borland */
#define VK_T_XORSHIFT 4u     /* src/xorshift.rs:69-74; 0 xorshift64, 1 xorshift128, 2 128+, 3 xoroshiro128** */
#define VK_T_SHA256_CHAIN 5u /* src/sha256_chain.rs:74-79; 0 iterated, 1 indexed:be, 2 indexed:le, 3 counter */
#define VK_T_ELECTRUM 6u     /* src/electrum.rs, src/transform/electrum.rs: STRING inputs only (vk_transform_strings);
                                variant = 0 receiving / 1 change chain, chain_depth = derivation count (reference: 20) */
#define VK_T_MULTIBIT 7u     /* src/multibit.rs, src/transform/multibit.rs: STRING inputs only; chain_depth = derivation
                                count (reference: 20); needs vk_bip39_wordlist_set first */
#define VK_T_DOUBLE_SHA256 8u /* src/transform/double_sha256.rs:13-32: SHA256(SHA256(x)), x = decimal string, 8 BE bytes, 8 LE bytes */
#define VK_T_MD5 9u           /* src/transform/md5.rs:13-21: MD5(decimal string) twice, key[0..16] = key[16..32] */
#define VK_T_DIRECT 10u       /* src/transform/direct.rs:13-34: slot 0 = 8 BE bytes right-aligned, 1 = 8 LE bytes left-aligned,
                                2 = the string zero padded to 32 bytes (strings longer than 32 bytes emit nothing) */
#define VK_T_ARMORY 11u       /* src/transform/armory.rs:21-101: root seed -> chaincode -> 4 chained key*G multiplications;
                                slot 0 seed = SHA256(decimal string), slot 1 seed = 24 zero bytes || 8 BE bytes */
#define VK_T_BITIMAGE 12u     /* src/bitimage.rs, src/transform/bitimage.rs: BLOB inputs only (file contents) -- the device entry
                                 point is vk_bitimage_keys; vk_slots_per_input / vk_slots_per_string are 0 (the reference emits
                                 nothing for numeric or string inputs), every other call rejects the kind */
#define VK_VARIANT_ALL 0xFFFFFFFFu

typedef struct vk_transform_spec {
    uint32_t kind;        /* VK_T_* */
    uint32_t variant;     /* variant index, or VK_VARIANT_ALL (the reference's `None`) */
    uint32_t endian;      /* lcg: 0 big (default), 1 little  (LcgEndian, src/lcg.rs:9-14) */
    uint32_t chain_depth; /* sha256_chain: keys per variant (DEFAULT_CHAIN_DEPTH = 10) */
} vk_transform_spec;

/* sweep flags */
#define VK_F_COMPRESSED 1u   /* hash160 of the 33-byte SEC1 key: P2PKH(c) and P2WPKH probes */
#define VK_F_UNCOMPRESSED 2u /* hash160 of the 65-byte SEC1 key: P2PKH(u) probe */

/* address kinds */
#define VK_ADDR_P2PKH_C 0u
#define VK_ADDR_P2PKH_U 1u
#define VK_ADDR_P2WPKH 2u
/* target kinds for vk_targets_set */
#define VK_TARGET_P2PKH 0u
#define VK_TARGET_P2WPKH 1u

/* One match = one (input, slot) whose derived key hit the target set.  `sub_index` is the
 * position of the key among the keys the transform emits for that input (emit order of
 * apply_batch): sha256 0 string / 1 BE8 / 2 LE8; lcg, xorshift with VK_VARIANT_ALL: the
 * variant; sha256_chain: form*(V*depth) + v*depth + idx (form 0 decimal string, 1 BE32). */
typedef struct vk_hit {
    uint64_t seed;
    uint32_t sub_index;
    uint32_t addr_kind;  /* VK_ADDR_*; first match in the reference's order (src/matcher.rs:72-98) */
    uint8_t privkey[32]; /* normalised key bytes (DerivedKey.raw, src/derive.rs:94) */
    uint8_t h160[20];    /* the hash160 that matched */
    uint32_t _pad;
} vk_hit;

typedef struct vk_sweep_stats {
    uint64_t inputs;       /* ProcessStats.inputs_processed = hi - lo + 1 */
    uint64_t keys;         /* ProcessStats.keys_generated */
    uint64_t matches;      /* ProcessStats.matches_found (keys that matched, not addresses) */
    uint64_t hits_dropped; /* matches that did not fit hit_cap (they are still counted) */
    uint64_t digest[4];    /* order-independent fold of every hash160 produced */
    double kernel_seconds; /* hipEvent time of the device work on the engine's stream */
    uint32_t launches;     /* kernel launches this call made */
    uint32_t inputs_skipped; /* vk_scan_strings: input strings over the device limit for which the reference would have produced
                                keys (they are in `inputs`, their keys are not in `keys`); 0 everywhere else */
    double shader_clock_ghz; /* shader clock the sweep kernels ran at: s_memtime cycles / s_memrealtime of workgroup 0, over the
                                call's launches (the part lowers its clock under this load; 0 if no sweep kernel ran) */
} vk_sweep_stats;

typedef struct vk_engine vk_engine;

/* table layouts */
#define VK_LAYOUT_UNIFORM 0u
#define VK_LAYOUT_SIGNED 1u  /* = VK_LAYOUT_SIGNED9, the name of ABI 2-4 when nine windows were the only signed geometry */
#define VK_LAYOUT_SIGNED9 1u /* signed-digit windows: a digit above half its range is replaced by its negative (y negated on the fly)
                                and a carry, so a window of b bits needs 2^(b-1) + 1 entries.  window_bits = 0 (or 29): the nine
                                windows of the name (4 x 29 + 5 x 28 bits), 8 additions per key, 120 GB table.  ABI 5: any
                                window_bits in 8..29 -> ceil(256 / window_bits) windows, e.g. 26 -> ten windows, 9 additions, 18.3 GB
                                (the unsigned 26-bit table needs 43 GB for the same), 24 -> eleven windows, 10 additions, 4.0 GB */

/* summation of the table points */
#define VK_ALGO_CHAIN 0u /* one chain of additions: the first two table points in affine form, then an XYZZ accumulator */
/* (1 was VK_ALGO_TREE in ABI versions 1-2: removed in round 3 -- vk_engine_create_ex answers VK_E_UNSUPPORTED) */

typedef struct vk_engine_opts {
    uint32_t window_bits;  /* fixed-base window width of the k*G table: 4..28 for the uniform layout (0 = default 24: 11.8 GB),
                              8..29 for the signed layout (0 = 29: nine windows, 120 GB) */
    uint32_t batch;        /* keys per lane sharing one field inversion (0 = default 128; the workspace, 128 B x batch x
                              lanes, is allocated when the engine is created) */
    uint32_t blocks_per_cu;/* 256-thread workgroups per CU to launch.  0 = per launch, the form that measured fastest for the
                              transform kind and the key forms hashed (vk_sweep_tu.hip: vk_sweep_design_form): three workgroups
                              per CU (three waves per SIMD) for launches that hash compressed keys only -- two for the sha256
                              family --, and ONE workgroup of 768 threads per CU (three waves per SIMD around one LDS schedule
                              table, tiles per wave) where uncompressed keys are hashed; the workspace is sized for three tile
                              owners per SIMD row (6.4 GB at batch 256).  A value >= 3 given here
                              keeps those forms and sizes grid and workspace; a value below 3 runs every launch in the
                              two-waves-per-SIMD form on that many 256-thread workgroups per CU.  Tiles are handed out
                              dynamically, so the value never changes results */
    uint32_t layout;       /* VK_LAYOUT_*: 0 = uniform unsigned window_bits-wide windows (default) */
    void *stream;          /* hipStream_t to run on (NULL = engine creates its own) */
    uint32_t algo;         /* VK_ALGO_CHAIN (0); other values are rejected */
    uint32_t _pad;
} vk_engine_opts;

typedef struct vk_engine_info_t {
    int32_t device;
    uint32_t cu_count;
    uint32_t window_bits;   /* W: the width of the widest window of the table that was built.  Signed layout: windows are
                               floor / ceil(256 / windows) bits wide, so several requests name one geometry (27 and 28 build the
                               ten 26/25-bit windows of 26, 25 the eleven of 24) and the request's own number is not echoed */
    uint32_t windows;       /* table lookups per key (ceil(256 / requested width)); windows - 1 point additions */
    uint32_t batch;         /* keys per lane per field inversion */
    uint32_t blocks_per_cu; /* the value the engine was created with, or 3 (the widest grid of the automatic choice) */
    uint32_t layout;
    uint32_t _pad;
    uint64_t table_bytes;   /* table entries * 64 */
    double table_build_seconds;  /* hipEvent time of the two table kernels */
    /* where vk_engine_create's wall time went (ABI 5): HIP runtime start + small buffers; the hipMalloc of workspace and
     * table (waits for the driver's scrub when another process has just freed as much); table kernels launch -> sync */
    double runtime_init_seconds;
    double alloc_seconds;
    double table_wall_seconds;
} vk_engine_info_t;

int vk_abi_version(void);
int vk_device_count(void);
int vk_engine_create(int device, vk_engine **out);
int vk_engine_create_ex(int device, const vk_engine_opts *opts, vk_engine **out);
void vk_engine_destroy(vk_engine *e);
int vk_engine_info(const vk_engine *e, vk_engine_info_t *out);
/* Message for the last failing call on this engine (valid until the next call); e may be NULL
 * to read the message of a failed vk_engine_create. */
const char *vk_last_error(const vk_engine *e);
/* What the last SUCCESSFUL vk_transform_strings / vk_scan_strings call on this engine wants the caller to know (ABI 3-4: a
 * one-line "warning: N input string(s) longer than the device limit ..."; no such limit exists since ABI 5 and nothing sets a
 * warning today); empty otherwise -- every such call clears it on entry.  Copied out under the engine's lock:
 * returns the message length, writes at most cap - 1 bytes and a NUL into buf (buf may be NULL to ask for the length). */
size_t vk_last_warning(vk_engine *e, char *buf, size_t cap);

/* Matcher::from_addresses on the device.  h160 = n x 20 bytes, kind = n bytes (VK_TARGET_*).
 * Builds an open-address table in HBM; replaces any previous set; n = 0 clears it. */
int vk_targets_set(vk_engine *e, const uint8_t *h160, const uint8_t *kind, size_t n);

/* RangeSource::process in scan mode (matcher = Some) or counting mode (no targets set):
 * every input in [seed_lo, seed_hi] -> transform -> normalise -> k*G -> hash160 -> probe.
 * hits (optional, unsorted) receives up to hit_cap records. */
int vk_sweep(vk_engine *e, const vk_transform_spec *t, uint64_t seed_lo, uint64_t seed_hi_inclusive,
             uint32_t flags, vk_hit *hits, size_t hit_cap, vk_sweep_stats *out);

/* RangeSource::process in generate mode (matcher = None): one record per (input, slot), in
 * input-major order.  slots = vk_slots_per_input(t); n = (hi-lo+1)*slots.  Any output pointer
 * may be NULL.  valid[i] = 0 where the reference emits nothing for that slot. */
int vk_generate(vk_engine *e, const vk_transform_spec *t, uint64_t seed_lo, uint64_t seed_hi_inclusive,
                uint32_t flags, uint8_t *privkeys /* n*32 */, uint8_t *h160_c /* n*20 */,
                uint8_t *h160_u /* n*20 */, uint8_t *valid /* n */, vk_sweep_stats *out);

/* RangeSource::process in generate mode as ConsoleOutput's compact format writes it (src/output/console.rs:77-81,
 * src/source/range.rs:42-70): for every (input, slot) the reference emits, in input-major order, the line
 *     {input}{slot_suffix[slot]},{transform_name},{private key, 64 hex digits},{P2PKH address of the compressed key}\n
 * -- decimal input, hex, Base58Check (two SHA-256) and the line itself are produced on the device; the host only writes the
 * text.  slot_suffix = vk_slots_per_input(t) strings (NULL = all empty): what Transform::apply_batch appends to the input's
 * decimal string in the label of that slot (src/transform/lcg.rs:54-66, xorshift.rs:43-47, sha256_chain.rs:67); suffixes and
 * transform_name are at most 31 bytes.  At most 2^21 (input, slot) pairs per call; text_cap >= 192 bytes per pair. */
int vk_generate_lines(vk_engine *e, const vk_transform_spec *t, uint64_t seed_lo, uint64_t seed_hi_inclusive, const char *transform_name,
                      const char *const *slot_suffix, uint8_t *text, size_t text_cap, size_t *text_len, vk_sweep_stats *out);
/* The same inputs in ConsoleOutput's verbose format (src/output/console.rs:62-76): per key the block
 *     ---\nsource: ..\ntransform: ..\nprivate_key: ..\nwif_compressed: ..\nwif_uncompressed: ..\n
 *     p2pkh_compressed: ..\np2pkh_uncompressed: ..\np2wpkh: ..\n
 * (mainnet encodings).  text_cap >= 512 bytes per (input, slot) pair; otherwise as vk_generate_lines. */
int vk_generate_verbose(vk_engine *e, const vk_transform_spec *t, uint64_t seed_lo, uint64_t seed_hi_inclusive, const char *transform_name,
                        const char *const *slot_suffix, uint8_t *text, size_t text_cap, size_t *text_len, vk_sweep_stats *out);

/* The same for WordlistSource / StdinSource in generate mode (src/source/wordlist.rs:24-77, src/output/console.rs:77-81): one
 * line per (input string, slot) the reference emits,
 *     {label}{slot_suffix[slot]},{transform_name},{private key hex},{P2PKH address of the compressed key}\n
 * slot_suffix = vk_slots_per_string(t) strings.  label = the input string itself (label_data = label_offsets = NULL: what sha256,
 * double_sha256, md5, direct, armory and sha256_chain print) or string i of (label_data, label_offsets), laid out like
 * (data, offsets): electrum and multibit print shortened forms of the input (src/electrum.rs:199-211, src/multibit.rs:352-360),
 * which the caller makes once per input.  At most 2^21 (input, slot) pairs and 4 GB of text per call;
 * text_cap >= slots x the label bytes + 192 bytes per pair. */
int vk_generate_lines_strings(vk_engine *e, const vk_transform_spec *t, const uint8_t *data, const uint32_t *offsets, size_t n,
                              const uint8_t *label_data, const uint32_t *label_offsets, const char *transform_name,
                              const char *const *slot_suffix, uint8_t *text, size_t text_cap, size_t *text_len, vk_sweep_stats *out);

/* Generate mode's FULL record (DerivedKey, src/derive.rs:12-41, as src/storage/schema.rs:137-262 lays it out for the 19-column
 * record batch) made on the device, column by column, for the inputs [seed_lo, seed_hi_inclusive]: one row per (input, slot) the
 * reference emits, in input order.  Every column arrives as the bytes of a Parquet PLAIN page: string columns as a 4-byte
 * little-endian length followed by the characters, VK_RC_KEY_RAW as 32 bytes per row, the three integer columns as 4-byte
 * little-endian values.  Mainnet encodings (Network::Bitcoin).  The columns a writer still has to add are constant per call
 * (transform, chain, timestamp, matched_target = null).  At most 2^20 (input, slot) pairs per call; every data[c] must hold
 * vk_record_column_stride(c) bytes per pair.  slot_suffix as for vk_generate_lines. */
#define VK_RC_SOURCE 0         /* "{input}{slot_suffix[slot]}" */
#define VK_RC_KEY_RAW 1        /* private_key_raw, FixedSizeBinary(32) */
#define VK_RC_KEY_HEX 2        /* 64 characters */
#define VK_RC_KEY_DECIMAL 3    /* src/derive.rs:121 */
#define VK_RC_KEY_BINARY 4     /* 256 characters, src/derive.rs:124-131 */
#define VK_RC_BIT_LENGTH 5     /* INT32, src/derive.rs:134-141 */
#define VK_RC_HAMMING_WEIGHT 6 /* INT32, src/derive.rs:144 */
#define VK_RC_LEADING_ZEROS 7  /* INT32: leading '0' characters of the hex form, src/derive.rs:148 */
#define VK_RC_PUBKEY_C 8       /* SEC1 compressed as hex, 66 characters */
#define VK_RC_PUBKEY_U 9       /* SEC1 uncompressed as hex, 130 characters */
#define VK_RC_P2PKH_C 10
#define VK_RC_P2PKH_U 11
#define VK_RC_P2WPKH 12
#define VK_RC_WIF_C 13
#define VK_RC_WIF_U 14
#define VK_RC_COLUMNS 15
typedef struct vk_record_columns {
    uint8_t *data[VK_RC_COLUMNS]; /* caller's buffers */
    size_t cap[VK_RC_COLUMNS];    /* their sizes */
    size_t len[VK_RC_COLUMNS];    /* out: bytes written */
    uint64_t rows;                /* out: records */
} vk_record_columns;
size_t vk_record_column_stride(int column); /* most bytes one row can take in that column (0 for an unknown column) */
int vk_generate_columns(vk_engine *e, const vk_transform_spec *t, uint64_t seed_lo, uint64_t seed_hi_inclusive, const char *const *slot_suffix,
                        vk_record_columns *cols, vk_sweep_stats *out);
/* KeyDeriver::derive (src/derive.rs:78-165) for n <= 2^20 raw keys, the whole record as the same columns, one row per key in
 * input order (a key outside [1, n) is normalised like vk_derive_batch does).  The source column is the caller's:
 * data[VK_RC_SOURCE] is not touched and len[VK_RC_SOURCE] = 0. */
int vk_derive_columns(vk_engine *e, const uint8_t *keys, size_t n, vk_record_columns *cols);

/* KeyDeriver::derive for callers that already hold keys (wordlist / stdin sources): raw 32-byte
 * keys in, normalised keys and hash160s out.  Any output pointer may be NULL. */
int vk_derive_batch(vk_engine *e, const uint8_t *keys /* n*32 */, size_t n, uint8_t *normalised /* n*32 */,
                    uint8_t *h160_c /* n*20 */, uint8_t *h160_u /* n*20 */);

/* The same with the public key itself: pub_xy = n*64 bytes, affine X || Y big-endian (SEC1 compressed = 02|03 || X by the
 * parity of Y, uncompressed = 04 || X || Y: DerivedKey::pubkey_compressed / pubkey_uncompressed, src/derive.rs:98-99).
 * Any output pointer may be NULL. */
int vk_derive_batch_pub(vk_engine *e, const uint8_t *keys /* n*32 */, size_t n, uint8_t *normalised, uint8_t *h160_c, uint8_t *h160_u,
                        uint8_t *pub_xy /* n*64 */);

/* Transform::apply_batch for numeric inputs: keys_out = n*slots*32, valid = n*slots. */
int vk_transform_batch(vk_engine *e, const vk_transform_spec *t, const uint64_t *seeds, size_t n,
                       uint8_t *keys_out, uint8_t *valid);

/* The 2048-word BIP-39 list the multibit transform validates against and rebuilds mnemonics from: words = 2048 x 8
 * bytes, each word zero padded to 8 bytes (the English list has no longer word).  The list is an input of the
 * caller; the library embeds none. */
int vk_bip39_wordlist_set(vk_engine *e, const uint8_t *words /* 16384 bytes */);

/* BIP-39 passphrase of the multibit chain: salt = "mnemonic" || passphrase in both stretches (src/multibit.rs:87-101,
 * 165-168).  Empty by default -- MultibitTransform always passes "" (src/transform/multibit.rs:15-21); the multibit-hd
 * analyzer takes one (src/analyze/multibit.rs:47-52).  Any length (96 bytes at most until round 5); stays set until changed. */
int vk_multibit_passphrase_set(vk_engine *e, const uint8_t *passphrase, size_t len);

/* BitimageTransform::apply_batch once the caller has reduced every file to its entropy = SHA-256(base64(file bytes))
 * (src/bitimage.rs:38-44; file-size-bound host work).  Per (file, passphrase): entropy -> 24-word BIP-39 mnemonic ->
 * PBKDF2-HMAC-SHA512(mnemonic, "mnemonic" || passphrase, 2048) -> HMAC-SHA512("Bitcoin seed") -> CKDpriv along `path`
 * (src/bitimage.rs:45-74,119-275), then `count` keys whose last path index is index, index + 1, ... as
 * BitimageTransform::derive_keys_for_passphrase walks increment_path_index (src/transform/bitimage.rs:71-97,
 * src/bitimage.rs:277-293).  path[i] = index | 0x80000000 for hardened components; depth = 0 is "m" (count copies of
 * the master key); any depth (16 levels at most until round 5).  Passphrases: n_pass >= 1 strings in pass_data[pass_offsets[j] .. pass_offsets[j+1]),
 * of any length (256 bytes at most until round 5; the single --bitimage-passphrase, possibly empty, or the lines of the
 * passphrase wordlist).
 * keys_out = n_files * n_pass * count * 32 bytes in (file, passphrase, index) order; valid = one byte per key, 0 where
 * the reference emits nothing: any error of derive_path (parent key 0 or >= n, IL >= n, child key 0, last index
 * reaching 2^31) -- and every later index of that (file, passphrase), because the reference does not advance the path
 * after a failure.  Needs vk_bip39_wordlist_set. */
int vk_bitimage_keys(vk_engine *e, const uint8_t *entropy /* n_files*32 */, size_t n_files, const uint8_t *pass_data,
                     const uint32_t *pass_offsets /* n_pass+1 */, size_t n_pass, const uint32_t *path, uint32_t depth,
                     uint32_t count, uint8_t *keys_out, uint8_t *valid);

/* Transform::apply_batch for string inputs (Input::from_string; wordlist / stdin sources).  String i is
 * data[offsets[i] .. offsets[i+1]).  slots = vk_slots_per_string(t): the hash transforms (sha256, double_sha256, md5,
 * direct) and armory emit one key per string of any length, sha256_chain variants x depth, VK_T_ELECTRUM and VK_T_MULTIBIT
 * chain_depth keys -- for strings of ANY length since ABI 5, as the reference hashes whatever it is given (src/electrum.rs:57-78,
 * src/multibit.rs:87-101): an Electrum seed beyond 64 bytes and a MultiBit line beyond 448 bytes run on slower paths that read
 * the caller's bytes (ABI 3-4 skipped strings over 448 bytes and reported them).  The PRNG transforms need a numeric seed and
 * return VK_E_UNSUPPORTED.  keys_out = n*slots*32, valid = n*slots, slot order = emit order.  An input the reference skips (a
 * non-hex seed, an unknown word, a wrong word count) has invalid slots; it never fails the batch. */
int vk_transform_strings(vk_engine *e, const vk_transform_spec *t, const uint8_t *data, const uint32_t *offsets, size_t n,
                         uint8_t *keys_out, uint8_t *valid);

/* KeyDeriver::derive + Matcher::check over keys the caller already holds (string-input sources): like
 * vk_sweep, with hit.seed = index of the key in `keys` and hit.sub_index = 0. */
int vk_scan_keys(vk_engine *e, const uint8_t *keys /* n*32 */, size_t n, uint32_t flags, vk_hit *hits, size_t hit_cap,
                 vk_sweep_stats *out);

/* Source::process of the string-input sources in scan mode (src/source/wordlist.rs:24-77, src/source/stdin.rs): the
 * fused form of vk_transform_strings + vk_scan_keys -- strings in, transform on the device, the keys stay in HBM and go
 * straight through k*G -> hash160 -> probe.  hit.seed = index of the input string, hit.sub_index = its slot;
 * stats.inputs = n, stats.keys = keys the transform emitted, kernel_seconds = transform kernel + sweep kernel.
 * Same length rule and warning as vk_transform_strings. */
int vk_scan_strings(vk_engine *e, const vk_transform_spec *t, const uint8_t *data, const uint32_t *offsets, size_t n, uint32_t flags,
                    vk_hit *hits, size_t hit_cap, vk_sweep_stats *out);
/* vk_scan_strings once more over the keys its last call on this engine left on the device (flags may differ): only the
 * derive + probe sweep runs, not the transform -- for a caller whose hit buffer was too small (out->hits_dropped > 0): the KDF
 * chains cost seconds per batch, the sweep milliseconds.  VK_E_INVALID_ARG once any other call has reused the buffers. */
int vk_rescan_strings(vk_engine *e, uint32_t flags, vk_hit *hits, size_t hit_cap, vk_sweep_stats *out);

/* ---- analyzer brute force (reference src/analyze/{milksad,lcg,mt64,xorshift,sha256_chain}.rs)
 * Smallest seed in [seed_lo, seed_hi] whose generated key(s) satisfy the target.  Kinds: VK_T_MILKSAD, VK_T_MT64,
 * VK_T_LCG (one variant: spec.variant 0..3, spec.endian), VK_T_XORSHIFT (one variant, cascade only) and
 * VK_T_SHA256_CHAIN (u32 seeds hashed as 4 big-endian bytes; spec.variant 0..3 or VK_VARIANT_ALL; exact / masked look
 * at every key of the chain of spec.chain_depth, cascade compares chain[j] with target j).  The reference reports
 * whichever matching seed a worker thread finds first; this search reports the smallest, which is one of the
 * reference's possible answers; for sha256_chain the (variant, index) is the first in the reference's loop order.
 * VK_T_SHA256 / DOUBLE_SHA256 / MD5 / DIRECT (exact, masked): any key the transform emits for the seed, result.index =
 * its slot -- the transform-only sweep the reference's `vuke bench` harness times (src/benchmark.rs:11-67). */
#define VK_SEARCH_EXACT 0u   /* key == target.key                                     (analyze_exact) */
#define VK_SEARCH_MASKED 1u  /* ((last 8 key bytes, BE) & (2^bits-1)) | 2^(bits-1) == last 8 bytes of target.key  (analyze_masked) */
#define VK_SEARCH_CASCADE 2u /* draws j = 0..n-1 of one generator instance: masked(draw j, bits[j]) == targets[j]  (analyze_cascading) */
typedef struct vk_search_target {
    uint32_t mode;        /* VK_SEARCH_* */
    uint32_t n_cascade;   /* cascade: 1..8 draws */
    uint8_t key[32];      /* exact / masked target key */
    uint32_t mask_bits;   /* masked: 1..64 */
    uint32_t _pad;
    uint8_t bits[8];      /* cascade: mask width per draw, 1..64 */
    uint64_t targets[8];  /* cascade: masked target per draw */
} vk_search_target;
typedef struct vk_search_result {
    uint32_t found;
    uint32_t n_keys;        /* draws stored in keys: 1, or n_cascade */
    uint64_t seed;
    uint64_t seeds_checked; /* whole range when nothing was found */
    double kernel_seconds;
    uint8_t keys[8][32];    /* the generated key(s) of the reported seed */
    uint32_t variant;       /* sha256_chain: variant that matched (0 iterated, 1 indexed:be, 2 indexed:le, 3 counter) */
    uint32_t index;         /* sha256_chain exact / masked: chain index of keys[0] */
} vk_search_result;
int vk_search(vk_engine *e, const vk_transform_spec *t, uint64_t seed_lo, uint64_t seed_hi_inclusive,
              const vk_search_target *target, vk_search_result *out);

/* Slots one numeric input (Input::from_u64) occupies under `t` (0 if the spec is invalid). */
uint32_t vk_slots_per_input(const vk_transform_spec *t);
/* Slots one string input (Input::from_string: no numeric forms) occupies; 0 = the transform emits nothing for strings. */
uint32_t vk_slots_per_string(const vk_transform_spec *t);

/* The vk_debug_* test hooks of ABI versions 1-4 are not part of this interface any more: include/vuke_hip_test_hooks.h,
 * exported only by the test build libvuke_hip_hooks.so. */

#ifdef __cplusplus
}
#endif
#endif
