/*
 * vko.h -- CPU ORACLE for the seed -> privkey -> secp256k1 -> hash160 -> match path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The shipped library (vuke_amd/csrc -> libvuke_hip.so) never links, includes or
 * calls anything in this directory.
 *
 * What it restates (reference = oritwoen/vuke v0.9.0, paths relative to the
 * reference root; the reference is Rust and cannot be built here):
 *   - transforms   src/transform/{sha256,milksad,mt64,lcg,xorshift,sha256_chain}.rs
 *                  src/{lcg,xorshift,mt64,sha256_chain}.rs
 *   - derivation   src/derive.rs:78-165      (KeyDeriver::derive)
 *   - matching     src/matcher.rs:45-108     (Matcher)
 *   - range driver src/source/range.rs:26-79 (RangeSource::process)
 * Third-party arithmetic that is NOT in the reference tree (Cargo.lock pins:
 * secp256k1 0.29.1 / secp256k1-sys 0.10.1, bitcoin 0.32.8, bitcoin_hashes 0.14.1,
 * base58ck 0.1.0, bech32 0.11.1, rand_mt 4.2.2, sha2 0.10.9) is restated from the
 * published standards: SEC2 secp256k1, FIPS 180-4, RIPEMD-160 (Dobbertin et al.),
 * BIP-173, Base58Check, MT19937 / MT19937-64 (Matsumoto & Nishimura).
 *
 * Parity pinning: tests/test_oracle_*.py check every function here against the
 * reference's own known-answer tests (src/derive.rs:179-224, src/matcher.rs:117-149,
 * src/lcg.rs:135-229, src/analyze/key_parser.rs:275-287, README.md:413,438-441) and
 * against independent implementations (hashlib, numpy RandomState, python ints).
 */
#ifndef VKO_H
#define VKO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- hashes */
void vko_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
void vko_ripemd160(const uint8_t *msg, size_t len, uint8_t out[20]);
void vko_hash160(const uint8_t *msg, size_t len, uint8_t out[20]); /* RIPEMD160(SHA256(m)) */
void vko_md5(const uint8_t *msg, size_t len, uint8_t out[16]);
void vko_hmac_sha256(const uint8_t *key, size_t klen, const uint8_t *msg, size_t mlen, uint8_t out[32]);
void vko_sha512(const uint8_t *msg, size_t len, uint8_t out[64]);
void vko_hmac_sha512(const uint8_t *key, size_t klen, const uint8_t *msg, size_t mlen, uint8_t out[64]);
void vko_pbkdf2_hmac_sha512(const uint8_t *pw, size_t pwlen, const uint8_t *salt, size_t saltlen,
                            uint32_t iters, uint8_t *out, size_t outlen);

/* ------------------------------------------------------------ transforms */
/* Transform kinds / variants: numbering is shared with include/vuke_hip.h. */
enum {
    VKO_T_SHA256 = 0,       /* src/transform/sha256.rs:13-29 */
    VKO_T_MILKSAD = 1,      /* src/transform/milksad.rs:17-29 */
    VKO_T_MT64 = 2,         /* src/mt64.rs:15-20 */
    VKO_T_LCG = 3,          /* src/lcg.rs:107-128, src/transform/lcg.rs:43-72 */
    VKO_T_XORSHIFT = 4,     /* src/xorshift.rs:83-317, src/transform/xorshift.rs:33-52 */
    VKO_T_SHA256_CHAIN = 5, /* src/sha256_chain.rs:88-145, src/transform/sha256_chain.rs:60-94 */
    VKO_T_ELECTRUM = 6,     /* vko_electrum_keys (string inputs) */
    VKO_T_MULTIBIT = 7,     /* vko_multibit_keys (string inputs) */
    VKO_T_DOUBLE_SHA256 = 8, /* src/transform/double_sha256.rs:13-32 */
    VKO_T_MD5 = 9,          /* src/transform/md5.rs:13-21 */
    VKO_T_DIRECT = 10,      /* src/transform/direct.rs:13-34 */
    VKO_T_ARMORY = 11,      /* src/transform/armory.rs:21-101 */
};
#define VKO_VARIANT_ALL 0xFFFFFFFFu

typedef struct {
    uint32_t kind;        /* VKO_T_* */
    uint32_t variant;     /* index into the transform's variant list, or VKO_VARIANT_ALL */
    uint32_t endian;      /* lcg only: 0 = big (default), 1 = little */
    uint32_t chain_depth; /* sha256_chain only (reference default 10) */
} vko_spec;

/* single-key primitives (return 0 when the reference skips the input, else 1) */
int vko_milksad_key(uint64_t seed, uint8_t key[32]);
int vko_mt64_key(uint64_t seed, uint8_t key[32]);
int vko_lcg_key(uint64_t seed, uint32_t variant, uint32_t little_endian, uint8_t key[32]);
int vko_xorshift_key(uint64_t seed, uint32_t variant, uint8_t key[32]);
int vko_sha256_chain_key(const uint8_t *seed, size_t seedlen, uint32_t variant, uint32_t index,
                         uint8_t key[32]);
int vko_armory_key(const uint8_t seed[32], uint8_t key[32]); /* ArmoryTransform::generate_key for a 32-byte seed */

/* Number of (sub_index) slots one input can produce under `spec` (fixed per spec). */
uint32_t vko_slots_per_input(const vko_spec *spec);
/* Key for (seed, sub_index); returns 0 if the reference emits nothing for that slot. */
int vko_transform_key(const vko_spec *spec, uint64_t seed, uint32_t sub_index, uint8_t key[32]);
/* The same for Input::from_string inputs (wordlist / stdin sources: no numeric forms). */
uint32_t vko_slots_per_string(const vko_spec *spec);
int vko_transform_string(const vko_spec *spec, const uint8_t *str, size_t len, uint32_t sub_index, uint8_t key[32]);
/* Label the reference attaches to (seed, sub_index) -- the `source` column. */
int vko_transform_label(const vko_spec *spec, uint64_t seed, uint32_t sub_index, char *out, size_t cap);
/* Transform::name() */
const char *vko_transform_name(const vko_spec *spec);

/* analyzer brute force (src/analyze/{milksad,lcg,mt64,xorshift,sha256_chain}.rs): successive draws and the smallest matching seed */
void vko_milksad_draws(uint32_t seed, uint32_t ndraws, uint8_t *out);
void vko_mt64_draws(uint64_t seed, uint32_t ndraws, uint8_t *out);
void vko_xorshift_draws(uint64_t seed, uint32_t variant, uint32_t ndraws, uint8_t *out);
int vko_search(const vko_spec *spec, uint64_t lo, uint64_t hi, uint32_t mode, const uint8_t key[32], uint32_t n,
               const uint8_t *bits, const uint64_t *targets, uint64_t *seed_out, uint8_t *keys_out, uint32_t variant_index[2]);

/* Electrum v1 (src/electrum.rs, src/transform/electrum.rs); rounds = 100000 in the reference */
void vko_electrum_stretch(const uint8_t *seed, size_t len, uint32_t rounds, uint8_t out[32]);
int vko_electrum_keys(const uint8_t *seed, size_t len, uint32_t for_change, uint32_t count, uint32_t rounds,
                      uint8_t *keys, uint8_t *valid, uint8_t mpk_out[64]);
int vko_electrum_label(const char *seed, uint32_t for_change, uint32_t index, char *out, size_t cap);

/* MultiBit HD seed-as-entropy chain (src/multibit.rs, src/transform/multibit.rs); wl = 2048 words, caller-supplied */
size_t vko_multibit_buggy_mnemonic(const char *const *wl, const uint8_t seed[64], char *out, size_t cap);
int vko_multibit_keys(const char *const *wl, const uint8_t *mnemonic, size_t len, uint32_t count, uint8_t *keys, uint8_t *valid,
                      uint8_t seed1_out[64], uint8_t seed2_out[64]);
int vko_multibit_keys_pass(const char *const *wl, const uint8_t *mnemonic, size_t len, const uint8_t *pass, size_t passlen, uint32_t count,
                           uint8_t *keys, uint8_t *valid, uint8_t seed1_out[64], uint8_t seed2_out[64]);
int vko_multibit_label(const char *mnemonic, uint32_t index, char *out, size_t cap);

/* Bitimage (src/bitimage.rs, src/transform/bitimage.rs): file -> base64 -> SHA-256 -> BIP-39 -> BIP-32 path */
size_t vko_base64(const uint8_t *data, size_t n, char *out);
int vko_bitimage_entropy(const uint8_t *data, size_t n, uint8_t out[32]);
size_t vko_bip39_mnemonic(const char *const *wl, const uint8_t *entropy, size_t elen, char *out, size_t cap);
int vko_bip32_parse_path(const char *path, uint32_t *comps, int cap);
size_t vko_bip32_increment_path(const char *path, char *out, size_t cap);
int vko_bip32_derive(const uint8_t seed[64], const uint32_t *comps, int depth, uint8_t key[32]);
uint32_t vko_bitimage_keys(const char *const *wl, const uint8_t entropy[32], const uint8_t *pass, size_t passlen, const char *path,
                           uint32_t count, uint8_t *keys, uint8_t *valid);

/* ------------------------------------------------------- secp256k1/derive */
/* src/derive.rs:79-93: valid iff 1 <= k < n, else k mod n, 0 -> 1. */
void vko_normalize_key(const uint8_t in[32], uint8_t out[32]);
/* k*G by plain double-and-add (k must be normalised); slow, "obviously right". */
void vko_pubkey_slow(const uint8_t key[32], uint8_t pub33[33], uint8_t pub65[65]);
/* k*G by an 8-bit fixed-window table (checked against vko_pubkey_slow in tests). */
void vko_pubkey(const uint8_t key[32], uint8_t pub33[33], uint8_t pub65[65]);
/* k*G for n normalised keys at once: 16-bit windows + one shared field inversion (checked against the two above). */
void vko_pubkey_batch(const uint8_t *keys, size_t n, uint8_t *pub33s, uint8_t *pub65s);
/* master + tweak mod n (BIP32 / Electrum child keys); returns 0 if result is 0. */
int vko_scalar_add(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
int vko_scalar_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]); /* a * b mod n; 0 if the product is zero */
/* 1 if 1 <= k < n */
int vko_scalar_valid(const uint8_t k[32]);

typedef struct {
    uint8_t raw[32]; /* normalised private key */
    uint8_t pub33[33];
    uint8_t pub65[65];
    uint8_t h160_c[20];
    uint8_t h160_u[20];
    char private_key_hex[65];
    char private_key_decimal[80];
    char private_key_binary[257];
    uint16_t bit_length;
    uint16_t hamming_weight;
    uint8_t leading_zeros;
    char pubkey_compressed[67];
    char pubkey_uncompressed[131];
    char wif_compressed[60];
    char wif_uncompressed[60];
    char p2pkh_compressed[40];
    char p2pkh_uncompressed[40];
    char p2wpkh[50];
} vko_derived;

/* KeyDeriver::derive, mainnet (src/derive.rs:78-165). */
void vko_derive(const uint8_t key[32], vko_derived *out);
/* Only what the device path needs. */
void vko_derive_h160(const uint8_t key[32], uint8_t h160_c[20], uint8_t h160_u[20]);

/* ---------------------------------------------------------- text codecs */
/* base58check(version || payload); returns strlen or -1 */
int vko_base58check_encode(const uint8_t *data, size_t len, char *out, size_t cap);
/* returns decoded length (incl. version byte, excl. checksum) or -1 */
int vko_base58check_decode(const char *s, uint8_t *out, size_t cap);
/* BIP-173 segwit address, witness v0 */
int vko_segwit_encode(const char *hrp, int witver, const uint8_t *prog, size_t proglen, char *out, size_t cap);
int vko_segwit_decode(const char *hrp, const char *addr, int *witver, uint8_t *prog, size_t *proglen);

/* ---------------------------------------------------------------- matcher */
typedef struct vko_matcher vko_matcher;
/* Matcher::from_addresses: a set of exact strings (src/matcher.rs:62-66). */
vko_matcher *vko_matcher_new(void);
void vko_matcher_free(vko_matcher *m);
/* Matcher::load line handling: trim; skip empty and '#' lines (src/matcher.rs:50-56). */
void vko_matcher_add_line(vko_matcher *m, const char *line);
size_t vko_matcher_count(const vko_matcher *m);
/* Matcher::check: -1 = none, 0 = p2pkh_compressed, 1 = p2pkh_uncompressed, 2 = p2wpkh. */
int vko_matcher_check(const vko_matcher *m, const vko_derived *d);
/* Same decision taken from hash160s only (what the device does). */
int vko_matcher_check_h160(const vko_matcher *m, const uint8_t h160_c[20], const uint8_t h160_u[20]);

/* ------------------------------------------------------------------ sweep */
typedef struct {
    uint64_t seed;
    uint32_t sub_index;
    uint32_t addr_kind; /* 0 p2pkh_c, 1 p2pkh_u, 2 p2wpkh */
    uint8_t privkey[32];
    uint8_t h160[20];
} vko_hit;

typedef struct {
    uint64_t inputs, keys, matches;
    uint64_t digest[4];
} vko_stats;

#define VKO_F_COMPRESSED 1u
#define VKO_F_UNCOMPRESSED 2u
#define VKO_F_FAITHFUL 0x100u /* do everything KeyDeriver::derive does per key (CPU baseline) */
#define VKO_F_BATCHED 0x200u  /* k*G of a whole 1000-input chunk through vko_pubkey_batch (same results, ~4x faster) */

/* RangeSource::process over inclusive [lo, hi] (src/source/range.rs:26-79), nthreads
 * worker threads pulling chunks of 1000 inputs.  hits are sorted by (seed, sub_index). */
int vko_sweep(const vko_spec *spec, uint64_t lo, uint64_t hi, uint32_t flags, const vko_matcher *m,
              vko_hit *hits, size_t hit_cap, vko_stats *out, int nthreads);

/* Digest fold shared with the device: see DESIGN.md "sweep digest". */
void vko_digest_fold(uint64_t digest[4], const uint8_t h160[20]);

#ifdef __cplusplus
}
#endif
#endif
