// vuke_host.hpp -- host side above the C ABI (include/vuke_hip.h), mirroring the reference's
// plugin interfaces for the hot path.  The reference is Rust and no Rust toolchain exists in this
// pipeline, so the mirror is C++: same names, argument meaning and error behaviour as
//   Transform / TransformType   src/transform/mod.rs:39-215
//   Source / ProcessStats       src/source/mod.rs:23-39, RangeSource src/source/range.rs:14-79
//   KeyDeriver / DerivedKey     src/derive.rs:12-165
//   Matcher / MatchInfo         src/matcher.rs:13-108
//   Output / ConsoleOutput      src/output/mod.rs:22-37, src/output/console.rs:14-122
// All compute (transform, k*G, hash160, probing) happens on the GPU through libvuke_hip.so; this
// layer only parses, formats text (hex / Base58Check / bech32 / decimal) and drives launches.
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/vuke_hip.h"

namespace vuke {

// Rust string semantics the reference's line readers rely on (BufRead::lines + str::trim, src/source/wordlist.rs:24-30,
// src/source/stdin.rs:45-50, src/matcher.rs:50-56): a line must be valid UTF-8, and trim() strips the Unicode White_Space
// code points (U+0009-000D, U+0020, U+0085, U+00A0, U+1680, U+2000-200A, U+2028, U+2029, U+202F, U+205F, U+3000).
bool utf8_valid(const char *p, size_t n);
void trim_unicode(const char *&a, const char *&e);


using Key = std::array<uint8_t, 32>;

// ---------------------------------------------------------------- text codecs (codecs.cpp)
void sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
// BitimageDeriver::from_file_bytes, first half (src/bitimage.rs:38-44): SHA-256 of the base64 text (STANDARD alphabet, '='
// padding) of the file bytes; the text is streamed through the hash, never materialised
void bitimage_entropy(const uint8_t *data, size_t len, uint8_t out[32]);
// parse_derivation_path (src/bitimage.rs:77-117): components as index | 0x80000000 when hardened; throws
// std::invalid_argument("Invalid derivation path: ...") with the reference's messages
std::vector<uint32_t> parse_derivation_path(const std::string &path);
// increment_path_index (src/bitimage.rs:277-293)
std::string increment_path_index(const std::string &path);
std::string hex_encode(const uint8_t *b, size_t n);
std::string base58check_encode(const uint8_t *payload, size_t len);
bool base58check_decode(const std::string &s, std::vector<uint8_t> &out); // version byte + payload
std::string segwit_encode(const std::string &hrp, int witver, const uint8_t *prog, size_t len);
bool segwit_decode(const std::string &hrp, const std::string &addr, int &witver, std::vector<uint8_t> &prog);
std::string decimal_of_be(const uint8_t be[32]);

// ---------------------------------------------------------------- derive.rs
struct DerivedKey { // src/derive.rs:12-41
    Key raw{};
    std::string private_key_hex, private_key_decimal, private_key_binary;
    uint16_t bit_length = 0, hamming_weight = 0;
    uint8_t leading_zeros = 0;
    std::string pubkey_compressed, pubkey_uncompressed; // hex SEC1; filled by KeyDeriver::derive (vk_derive_batch_pub), empty in the bulk generate path
    std::string wif_compressed, wif_uncompressed;
    std::string p2pkh_compressed, p2pkh_uncompressed, p2wpkh;
    uint8_t h160_c[20]{}, h160_u[20]{};
    std::array<const std::string *, 3> addresses() const { return {&p2pkh_compressed, &p2pkh_uncompressed, &p2wpkh}; }
};

// KeyDeriver: the k*G + hash160 half runs on the device (vk_derive_batch / vk_generate / hits);
// this class assembles the reference's record from the device outputs (mainnet, src/derive.rs:62-67).
enum class Network { Bitcoin, Testnet, Signet, Regtest }; // bitcoin::Network as src/network.rs:6-17 maps it
Network parse_network(const std::string &s);                 // unknown names warn on stderr and mean Bitcoin

class KeyDeriver {
public:
    explicit KeyDeriver(vk_engine *engine, Network net = Network::Bitcoin) : engine_(engine), net_(net) {} // KeyDeriver::with_network
    // derive(&[u8;32]) for arbitrary keys: normalisation + hash160s on the GPU, text on the host
    std::vector<DerivedKey> derive(const std::vector<Key> &keys) const;
    DerivedKey derive(const Key &key) const { return derive(std::vector<Key>{key})[0]; }
    // assemble from values the device already produced
    static DerivedKey from_device(const uint8_t normalised[32], const uint8_t h160_c[20], const uint8_t h160_u[20], Network net = Network::Bitcoin);
    // pubkey_compressed / pubkey_uncompressed from the affine point X || Y (64 bytes, big-endian)
    static void set_pubkeys(DerivedKey &d, const uint8_t xy[64]);
    // the two fields the compact CSV line prints (private_key_hex, p2pkh_compressed); every other text field stays empty
    static DerivedKey from_device_compact(const uint8_t normalised[32], const uint8_t h160_c[20], const uint8_t h160_u[20]);

private:
    vk_engine *engine_;
    Network net_;
};

// ---------------------------------------------------------------- matcher.rs
enum class AddressType { P2pkhCompressed, P2pkhUncompressed, P2wpkh };
const char *address_type_str(AddressType t); // AddressType::as_str
struct MatchInfo {
    AddressType address_type;
    std::string address;
};

class Matcher {
public:
    static Matcher load(const std::string &path);                     // src/matcher.rs:45-59 (throws std::runtime_error on I/O error)
    static Matcher from_addresses(const std::vector<std::string> &a); // :62-66
    std::optional<MatchInfo> check(const DerivedKey &d) const;        // :72-98
    size_t count() const { return targets_.size(); }                  // :101-103
    bool is_empty() const { return targets_.empty(); }
    // Device form of the set: hash160 + kind for every line that is a valid mainnet P2PKH / P2WPKH
    // address (any other line can never equal a derived address, so it cannot match in the reference).
    void device_targets(std::vector<uint8_t> &h160, std::vector<uint8_t> &kind) const;
    // decodes the set now (e.g. on the thread that loaded the file, while the GPU engine is still starting); device_targets() then copies
    void prepare_device_targets();
    bool contains(const std::string &s) const { return targets_.count(s) != 0; }

private:
    void decode_all(std::vector<uint8_t> &h160, std::vector<uint8_t> &kind) const;
    std::unordered_set<std::string> targets_;
    std::vector<uint8_t> dev_h160_, dev_kind_;
    bool dev_ready_ = false;
};

// ---------------------------------------------------------------- transform/mod.rs
// BitimageTransform's builder state (src/transform/bitimage.rs:9-69)
struct BitimageConfig {
    std::string path = "m/84'/0'/0'/0/0";
    std::string passphrase;
    std::optional<std::vector<std::string>> passphrase_wordlist; // trimmed non-empty lines of --bitimage-passphrase-wordlist
    uint32_t derive_count = 1;
    static std::vector<std::string> load_passphrase_wordlist(const std::string &file); // with_passphrase_wordlist (:36-64)
};

class Transform {
public:
    explicit Transform(const vk_transform_spec &s) : spec_(s) {}
    // ---- bitimage (blob inputs: FilesSource)
    bool is_bitimage() const { return spec_.kind == VK_T_BITIMAGE; }
    BitimageConfig bitimage;
    // apply_batch for Input::from_blob inputs, on the device: entropies[i] = bitimage_entropy(file i), labels[i] = its
    // Input::string_val; appends ("{label}[{path}]" or "{label}:{passphrase}[{path}]", key) in the reference's order
    void apply_batch_blobs(vk_engine *e, const std::vector<std::array<uint8_t, 32>> &entropies, const std::vector<std::string> &labels,
                           std::vector<std::pair<std::string, Key>> &output) const;
    const char *name() const;                              // Transform::name()
    const vk_transform_spec &spec() const { return spec_; }
    uint32_t slots() const { return vk_slots_per_input(&spec_); }
    std::string label(uint64_t seed, uint32_t sub) const;  // the `source` string apply_batch pushes
    std::string label(const std::string &input, uint32_t sub) const; // same for Input::from_string inputs
    uint32_t string_slots() const { return vk_slots_per_string(&spec_); } // keys per Input::from_string input (0: none)
    // electrum / multibit read only Input::string_val: numeric sources hand them the decimal string
    bool takes_strings() const { return spec_.kind == VK_T_ELECTRUM || spec_.kind == VK_T_MULTIBIT; }
    // apply_batch for string inputs (wordlist / stdin sources), on the device
    void apply_batch(vk_engine *e, const std::vector<std::string> &inputs, std::vector<std::pair<std::string, Key>> &output) const;
    bool supports_gpu() const { return true; }
    // apply_batch for numeric inputs, on the device: appends (label, key) in the reference's order
    void apply_batch(vk_engine *e, const std::vector<uint64_t> &inputs, std::vector<std::pair<std::string, Key>> &output) const;

private:
    vk_transform_spec spec_;
};

struct TransformType {
    // TransformType::from_str (src/transform/mod.rs:170-215): throws std::invalid_argument with the
    // reference's message for unknown names.  "bitimage" comes back with the default BitimageConfig
    // (apply_bitimage_config in src/main.rs:48-70 = assigning Transform::bitimage afterwards).
    static Transform from_str(const std::string &s);
};

// ---------------------------------------------------------------- output/
class Output {
public:
    virtual ~Output() = default;
    virtual void key(const std::string &source, const std::string &transform, const DerivedKey &d) = 0;
    virtual void hit(const std::string &source, const std::string &transform, const DerivedKey &d, const MatchInfo &m) = 0;
    virtual void flush() = 0;
    // false: key() reads only private_key_hex and p2pkh_compressed (the reference builds all fourteen fields for every
    // key regardless; at three million keys that is most of generate's wall time)
    virtual bool needs_full_record() const { return true; }
    // true: key() also reads pubkey_compressed / pubkey_uncompressed (the bulk generate path fetches them only on request)
    virtual bool needs_pubkeys() const { return false; }
    // Ready-made output text: whole lines exactly as key() would have written them for a run of keys, in order (generate mode's
    // compact lines made on the device, vk_generate_lines).  false = this sink wants records: the caller then goes through key().
    virtual bool write_text(const char *, size_t) { return false; }
    virtual bool accepts_text() const { return false; }
    // The same for sinks that store the full record: a run of records as the value columns the device made (vk_generate_columns /
    // vk_derive_columns; source_col != nullptr replaces the source column with the caller's labels, same encoding).
    virtual bool write_columns(const vk_record_columns &, const std::string & /*transform*/, const uint8_t * /*source_col*/ = nullptr, size_t = 0) { return false; }
    // which of the two ready-made forms this sink takes INSTEAD of key() calls (0: it wants records)
    enum : unsigned { DEVICE_TEXT = 1, DEVICE_COLUMNS = 2, DEVICE_VERBOSE = 4 };
    virtual bool write_verbose(const char *, size_t) { return false; } // the same with ConsoleOutput's verbose blocks (vk_generate_verbose)
    virtual unsigned device_formats() const { return accepts_text() ? DEVICE_TEXT : 0u; }
};

// MultiOutput (src/output/multi.rs): the same record to every sink, in order
class MultiOutput : public Output {
public:
    explicit MultiOutput(std::vector<Output *> sinks) : sinks_(std::move(sinks)) {}
    void key(const std::string &source, const std::string &transform, const DerivedKey &d) override
    {
        for (Output *o : sinks_) o->key(source, transform, d);
    }
    void hit(const std::string &source, const std::string &transform, const DerivedKey &d, const MatchInfo &m) override
    {
        for (Output *o : sinks_) o->hit(source, transform, d, m);
    }
    void flush() override
    {
        for (Output *o : sinks_) o->flush();
    }
    bool needs_full_record() const override
    {
        for (Output *o : sinks_)
            if (o->needs_full_record()) return true;
        return false;
    }
    bool needs_pubkeys() const override
    {
        for (Output *o : sinks_)
            if (o->needs_pubkeys()) return true;
        return false;
    }
    // every sink must be servable without records; each then gets the form it takes
    unsigned device_formats() const override
    {
        unsigned all = 0;
        for (Output *o : sinks_) {
            const unsigned f = o->device_formats();
            if (!f) return 0;
            all |= f;
        }
        return all;
    }
    bool write_text(const char *text, size_t len) override
    {
        for (Output *o : sinks_)
            if (o->device_formats() & DEVICE_TEXT) o->write_text(text, len);
        return true;
    }
    bool write_verbose(const char *text, size_t len) override
    {
        for (Output *o : sinks_)
            if (o->device_formats() & DEVICE_VERBOSE) o->write_verbose(text, len);
        return true;
    }
    bool write_columns(const vk_record_columns &cols, const std::string &transform, const uint8_t *source_col = nullptr, size_t source_len = 0) override
    {
        for (Output *o : sinks_)
            if (o->device_formats() & DEVICE_COLUMNS) o->write_columns(cols, transform, source_col, source_len);
        return true;
    }

private:
    std::vector<Output *> sinks_;
};

// StorageOutput (src/output/storage.rs + src/storage/parquet_backend.rs): the 19-column result record
// (src/storage/schema.rs:78-111) as Parquet files under
// DIR/transform=<name>/date=<UTC date>/chunk_NNNN.parquet, a new chunk every max_chunk_records records or
// max_chunk_bytes of record data, whichever comes first (0 = no such limit; defaults 1 000 000 and 100 MiB as the reference).
class StorageOutput : public Output {
public:
    // compression: "zstd" (the reference's default, level 3), "gzip", "snappy", "lz4", "none"
    StorageOutput(const std::string &base_dir, const std::string &transform_name, uint64_t max_chunk_records = 1000000,
                  const std::string &compression = "zstd", int level = 3, uint64_t max_chunk_bytes = 100ull << 20);
    ~StorageOutput() override;
    void key(const std::string &source, const std::string &transform, const DerivedKey &d) override;
    void hit(const std::string &source, const std::string &transform, const DerivedKey &d, const MatchInfo &m) override;
    void flush() override;
    bool needs_pubkeys() const override { return true; }
    unsigned device_formats() const override { return DEVICE_COLUMNS; }
    bool write_columns(const vk_record_columns &cols, const std::string &transform, const uint8_t *source_col = nullptr, size_t source_len = 0) override;
    std::vector<std::string> finish(); // closes the open chunk; paths of every file written
    uint64_t records() const;

private:
    struct Impl;
    Impl *impl_;
};

class ConsoleOutput : public Output { // src/output/console.rs
public:
    ConsoleOutput() : fp_(stdout), own_(false), verbose_(false) {}
    static std::unique_ptr<ConsoleOutput> verbose();
    static std::unique_ptr<ConsoleOutput> to_file(const std::string &path, bool verbose = false); // throws on open failure
    ~ConsoleOutput() override;
    void key(const std::string &source, const std::string &transform, const DerivedKey &d) override;
    void hit(const std::string &source, const std::string &transform, const DerivedKey &d, const MatchInfo &m) override;
    void flush() override;
    bool needs_full_record() const override { return verbose_; }
    bool accepts_text() const override { return !verbose_; } // the compact format is one line per key
    unsigned device_formats() const override { return verbose_ ? DEVICE_VERBOSE : DEVICE_TEXT; }
    bool write_text(const char *text, size_t len) override;
    bool write_verbose(const char *text, size_t len) override;

private:
    ConsoleOutput(FILE *fp, bool own, bool verbose) : fp_(fp), own_(own), verbose_(verbose) {}
    FILE *fp_;
    bool own_, verbose_;
    std::mutex mu_;
};

// ---------------------------------------------------------------- source/
struct ProcessStats { // src/source/mod.rs:35-39
    uint64_t inputs_processed = 0, keys_generated = 0, matches_found = 0;
};

class Source {
public:
    virtual ~Source() = default;
    // process(&[Box<dyn Transform>], Option<&Matcher>, &dyn Output) -> Result<ProcessStats>
    virtual ProcessStats process(const std::vector<Transform> &transforms, const Matcher *matcher, Output &output) = 0;
};

// RangeSource on the GPU: inclusive [start, end] (src/source/range.rs:14-79).
class HipRangeSource : public Source {
public:
    HipRangeSource(vk_engine *engine, uint64_t start, uint64_t end) : start(start), end(end), engine_(engine) {}
    ProcessStats process(const std::vector<Transform> &transforms, const Matcher *matcher, Output &output) override;
    uint64_t start, end;

private:
    vk_engine *engine_;
};

// RangeSource over several GPUs: [start, end] cut into one contiguous block per engine, one host thread per
// engine, no exchange between them; counters summed.  The same partition bench.py uses across processes.
class HipMultiRangeSource : public Source {
public:
    HipMultiRangeSource(std::vector<vk_engine *> engines, uint64_t start, uint64_t end) : start(start), end(end), engines_(std::move(engines)) {}
    ProcessStats process(const std::vector<Transform> &transforms, const Matcher *matcher, Output &output) override;
    uint64_t start, end;

private:
    std::vector<vk_engine *> engines_;
};

// WordlistSource / StdinSource on the GPU (src/source/wordlist.rs, src/source/stdin.rs): one passphrase per line,
// trimmed, empty lines skipped, Input::from_string.  path "-" reads stdin.  Transforms that need a numeric seed
// emit nothing for these inputs, exactly like the reference.  With several engines every batch of lines is cut into one
// contiguous slice per GPU (the inputs are independent, nothing is exchanged); records come out in input order.
class HipWordlistSource : public Source {
public:
    HipWordlistSource(vk_engine *engine, std::string path) : path(std::move(path)), engines_{engine} {}
    HipWordlistSource(std::vector<vk_engine *> engines, std::string path) : path(std::move(path)), engines_(std::move(engines)) {}
    ProcessStats process(const std::vector<Transform> &transforms, const Matcher *matcher, Output &output) override;
    std::string path;

private:
    std::vector<vk_engine *> engines_;
};

// TimestampSource on the GPU (src/source/timestamps.rs): every second of [start_date 00:00:00, end_date 23:59:59] UTC
// and, with `microseconds`, the thousand values ts*1000 + 0..999 behind each second -- two (or one) range sweeps.
class HipTimestampSource : public Source {
public:
    HipTimestampSource(vk_engine *engine, const std::string &start_date, const std::string &end_date, bool microseconds); // throws on bad dates
    ProcessStats process(const std::vector<Transform> &transforms, const Matcher *matcher, Output &output) override;
    uint64_t start, end;
    bool microseconds;

private:
    vk_engine *engine_;
};

// FilesSource on the GPU (src/source/files.rs): one file, or every regular file under a directory (recursive, symlinks
// skipped).  Each file is one Input::from_blob(contents, path): bitimage consumes the contents, every other transform
// sees the path as Input::string_val.  Unreadable files are reported on stderr and skipped.
class HipFilesSource : public Source {
public:
    static HipFilesSource from_file(vk_engine *engine, const std::string &path); // throws: "File not found: ..", "Not a file: .."
    static HipFilesSource from_dir(vk_engine *engine, const std::string &path);  // throws: "Directory not found", "Not a directory", "No files found in directory"
    ProcessStats process(const std::vector<Transform> &transforms, const Matcher *matcher, Output &output) override;
    size_t file_count() const { return files.size(); }
    std::vector<std::string> files;

private:
    explicit HipFilesSource(vk_engine *engine) : engine_(engine) {}
    vk_engine *engine_;
};

// parquet.cpp's page compressor by Parquet codec id (0 none, 1 snappy, 2 gzip, 5 lz4, 6 zstd), exposed for the tests
std::string compress_page_for_test(int codec, const std::string &in);

// Load a BIP-39 word list file (2048 whitespace-separated words of <= 8 bytes) into the engine (multibit transform).
void load_bip39_wordlist(vk_engine *e, const std::string &path);

// ---------------------------------------------------------------- analyze/ 
enum class AnalysisStatus { Confirmed, Possible, NotFound, Unknown }; // src/analyze/mod.rs:48-59
struct AnalysisResult {                                               // :36-45
    const char *analyzer;
    AnalysisStatus status;
    std::string details;
    const char *symbol() const;
    const char *status_str() const;
};
struct AnalysisConfig { // :61-66
    std::optional<uint8_t> mask_bits;
    std::optional<std::vector<std::pair<uint8_t, uint64_t>>> cascade_targets;
};
class Analyzer { // :91-129
public:
    virtual ~Analyzer() = default;
    virtual const char *name() const = 0;
    virtual AnalysisResult analyze(const Key &key, const AnalysisConfig &config) const = 0;
    virtual bool supports_mask() const { return false; }
    virtual bool is_brute_force() const { return true; }
    virtual bool supports_gpu() const { return true; }
};
class MilksadAnalyzer : public Analyzer {
public:
    explicit MilksadAnalyzer(vk_engine *e) : engine_(e) {}
    const char *name() const override { return "milksad"; }
    bool supports_mask() const override { return true; }
    AnalysisResult analyze(const Key &key, const AnalysisConfig &config) const override;

private:
    vk_engine *engine_;
};
class LcgAnalyzer : public Analyzer {
public:
    LcgAnalyzer(vk_engine *e, uint32_t variant = VK_VARIANT_ALL, uint32_t little_endian = 0) : engine_(e), variant_(variant), endian_(little_endian) {}
    const char *name() const override { return "lcg"; }
    bool supports_mask() const override { return true; }
    AnalysisResult analyze(const Key &key, const AnalysisConfig &config) const override;

private:
    vk_engine *engine_;
    uint32_t variant_, endian_;
};
class Mt64Analyzer : public Analyzer {
public:
    // the reference walks 2^64 seeds until it finds one; `limit` bounds the sweep here
    explicit Mt64Analyzer(vk_engine *e, uint64_t limit = (1ull << 40) - 1) : engine_(e), limit_(limit) {}
    const char *name() const override { return "mt64"; }
    AnalysisResult analyze(const Key &key, const AnalysisConfig &config) const override;

private:
    vk_engine *engine_;
    uint64_t limit_;
};
class XorshiftAnalyzer : public Analyzer { // src/analyze/xorshift.rs: cascade only, every variant walks seeds upward
public:
    XorshiftAnalyzer(vk_engine *e, uint32_t variant = VK_VARIANT_ALL, uint64_t limit = (1ull << 40) - 1) : engine_(e), variant_(variant), limit_(limit) {}
    const char *name() const override { return "xorshift"; }
    bool supports_mask() const override { return true; }
    AnalysisResult analyze(const Key &key, const AnalysisConfig &config) const override;

private:
    vk_engine *engine_;
    uint32_t variant_;
    uint64_t limit_;
};
class Sha256ChainAnalyzer : public Analyzer { // src/analyze/sha256_chain.rs: all 2^32 seeds as 4 BE bytes
public:
    Sha256ChainAnalyzer(vk_engine *e, uint32_t variant = VK_VARIANT_ALL, uint32_t chain_depth = 10) : engine_(e), variant_(variant), depth_(chain_depth) {}
    const char *name() const override { return "sha256_chain"; }
    bool supports_mask() const override { return true; }
    AnalysisResult analyze(const Key &key, const AnalysisConfig &config) const override;

private:
    vk_engine *engine_;
    uint32_t variant_, depth_;
};
// multibit-hd (src/analyze/multibit.rs): does a mnemonic -- or any line of a mnemonic file -- reach the key through the
// seed-as-entropy chain at m/0'/0/0..depth-1?  The chains run on the device (vk_transform_strings, VK_T_MULTIBIT), a file
// in batches of 2^16 mnemonics; needs the BIP-39 word list loaded into the engine.
class MultibitAnalyzer : public Analyzer {
public:
    MultibitAnalyzer(vk_engine *e, std::optional<std::string> mnemonic, std::optional<std::string> mnemonic_file, std::string passphrase,
                     uint32_t depth = 20)
        : engine_(e), mnemonic_(std::move(mnemonic)), file_(std::move(mnemonic_file)), passphrase_(std::move(passphrase)), depth_(depth) {}
    const char *name() const override { return "multibit-hd"; }
    bool is_brute_force() const override { return file_.has_value(); }
    AnalysisResult analyze(const Key &key, const AnalysisConfig &config) const override;

private:
    vk_engine *engine_;
    std::optional<std::string> mnemonic_, file_;
    std::string passphrase_;
    uint32_t depth_;
};

// host-only analyzers of the reference's default set: cheap pattern checks on the key bytes, no search
class DirectAnalyzer : public Analyzer { // src/analyze/direct.rs
public:
    const char *name() const override { return "direct"; }
    bool is_brute_force() const override { return false; }
    bool supports_gpu() const override { return false; }
    AnalysisResult analyze(const Key &key, const AnalysisConfig &config) const override;
};
class HeuristicAnalyzer : public Analyzer { // src/analyze/heuristic.rs
public:
    const char *name() const override { return "heuristic"; }
    bool is_brute_force() const override { return false; }
    bool supports_gpu() const override { return false; }
    AnalysisResult analyze(const Key &key, const AnalysisConfig &config) const override;
};
uint16_t calculate_bit_length(const Key &key);                                     // src/analyze/mod.rs:30-35
std::string format_results_json(const Key &key, const std::vector<AnalysisResult> &results); // src/analyze/output.rs:27-60
Key parse_private_key(const std::string &input);                              // src/analyze/key_parser.rs:32-48
std::vector<std::pair<uint8_t, uint64_t>> parse_cascade(const std::string &input); // :131-166
std::string format_results(const Key &key, const std::vector<AnalysisResult> &results); // src/analyze/output.rs:3-25

// error carrying the C ABI's code + message
struct EngineError : std::runtime_error {
    int code;
    EngineError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
void check_rc(vk_engine *e, int rc);
std::string last_warning(vk_engine *e); // vk_last_warning of the last successful string call (empty = nothing to report)

// test access to the line reader of the wordlist / stdin sources (source.cpp)
std::string line_reader_batches(FILE *in, bool strict, size_t max_lines, size_t max_bytes, FILE *out);

} // namespace vuke
