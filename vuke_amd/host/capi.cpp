// capi.cpp -- small extern "C" test hooks over the host layer so pytest can drive the C++ code on a
// machine without a GPU (parsers, labels, text codecs, record assembly, output formats, matcher).
#include "vuke_host.hpp"

#include <cstring>
#include <stdexcept>

using namespace vuke;

static int put(const std::string &s, char *out, size_t cap)
{
    if (s.size() + 1 > cap) return -1;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}

extern "C" {

// parse a --transform value; on failure returns -1 and the reference-style message in err
int vkh_parse_transform(const char *s, vk_transform_spec *spec, char *name, size_t name_cap, char *err, size_t err_cap)
{
    try {
        Transform t = TransformType::from_str(s);
        *spec = t.spec();
        put(t.name(), name, name_cap);
        return 0;
    } catch (const std::exception &e) {
        put(e.what(), err, err_cap);
        return -1;
    }
}

int vkh_label(const vk_transform_spec *spec, uint64_t seed, uint32_t sub, char *out, size_t cap) { return put(Transform(*spec).label(seed, sub), out, cap); }

int vkh_base58check_encode(const uint8_t *p, size_t n, char *out, size_t cap) { return put(base58check_encode(p, n), out, cap); }
int vkh_base58check_decode(const char *s, uint8_t *out, size_t cap)
{
    std::vector<uint8_t> v;
    if (!base58check_decode(s, v) || v.size() > cap) return -1;
    memcpy(out, v.data(), v.size());
    return (int)v.size();
}
int vkh_segwit_encode(const char *hrp, int ver, const uint8_t *p, size_t n, char *out, size_t cap) { return put(segwit_encode(hrp, ver, p, n), out, cap); }
int vkh_segwit_decode(const char *hrp, const char *addr, int *ver, uint8_t *out, size_t cap)
{
    std::vector<uint8_t> v;
    if (!segwit_decode(hrp, addr, *ver, v) || v.size() > cap) return -1;
    memcpy(out, v.data(), v.size());
    return (int)v.size();
}
void vkh_sha256(const uint8_t *p, size_t n, uint8_t out[32]) { sha256(p, n, out); }

// the device batches the wordlist (strict = 0) / stdin (strict = 1) source cuts this file into: lines + "--" per batch into `out_path`;
// returns 0, or 1 with the reader's error in err (the run ends with it after the batches written)
int vkh_line_batches(const char *in_path, int strict, size_t max_lines, size_t max_bytes, const char *out_path, char *err, size_t err_cap)
{
    FILE *in = fopen(in_path, "rb"), *out = fopen(out_path, "wb");
    if (!in || !out) return -1;
    const std::string e = line_reader_batches(in, strict != 0, max_lines, max_bytes, out);
    fclose(in);
    fclose(out);
    put(e, err, err_cap);
    return e.empty() ? 0 : 1;
}

// what the wordlist / stdin sources make of one raw line: -2 = dropped (not UTF-8), else the length of the trimmed text in out
int vkh_line_filter(const char *line, size_t n, char *out, size_t cap)
{
    if (!utf8_valid(line, n)) return -2;
    const char *a = line, *e = line + n;
    trim_unicode(a, e);
    return put(std::string(a, e), out, cap);
}

// Assemble a DerivedKey from device outputs and render it: mode 0 compact line, 1 verbose block,
// 2 HIT block (kind = address type), 3 "decimal|binary|bit_length|hamming|leading_zeros|wif_c|wif_u|c|u|w"
int vkh_render(const uint8_t key[32], const uint8_t hc[20], const uint8_t hu[20], int mode, int kind, const char *source,
               const char *transform, const char *path)
{
    try {
        DerivedKey d = KeyDeriver::from_device(key, hc, hu);
        auto out = ConsoleOutput::to_file(path, mode == 1);
        if (mode == 0 || mode == 1) out->key(source, transform, d);
        else if (mode == 2) {
            MatchInfo mi{(AddressType)kind, *d.addresses()[kind]};
            out->hit(source, transform, d, mi);
        } else {
            FILE *fp = fopen(path, "w");
            if (!fp) return -1;
            fprintf(fp, "%s|%s|%u|%u|%u|%s|%s|%s|%s|%s", d.private_key_decimal.c_str(), d.private_key_binary.c_str(), d.bit_length,
                    d.hamming_weight, d.leading_zeros, d.wif_compressed.c_str(), d.wif_uncompressed.c_str(), d.p2pkh_compressed.c_str(),
                    d.p2pkh_uncompressed.c_str(), d.p2wpkh.c_str());
            fclose(fp);
            return 0;
        }
        out->flush();
        return 0;
    } catch (...) {
        return -1;
    }
}

// Matcher::load on a file: returns count(), and writes the device form (n x 20 bytes, n kinds)
long vkh_matcher_load(const char *path, uint8_t *h160, uint8_t *kind, size_t cap, size_t *n_device)
{
    try {
        Matcher m = Matcher::load(path);
        std::vector<uint8_t> h, k;
        m.device_targets(h, k);
        *n_device = k.size();
        if (!k.empty() && k.size() <= cap) { // (an empty vector's data() may be null: not an argument for memcpy)
            memcpy(h160, h.data(), h.size());
            memcpy(kind, k.data(), k.size());
        }
        return (long)m.count();
    } catch (...) {
        return -1;
    }
}

// Matcher::check on a record assembled from (key, hc, hu): -1 none, else AddressType
int vkh_matcher_check(const char *path, const uint8_t key[32], const uint8_t hc[20], const uint8_t hu[20])
{
    try {
        Matcher m = Matcher::load(path);
        auto r = m.check(KeyDeriver::from_device(key, hc, hu));
        return r ? (int)r->address_type : -1;
    } catch (...) {
        return -2;
    }
}
}

extern "C" {
// parse_private_key (src/analyze/key_parser.rs:32-48): 0 on success
// TimestampSource::from_dates (src/source/timestamps.rs:21-43): the inclusive second range of two dates; -1 on a bad date
int vkh_timestamp_range(const char *start_date, const char *end_date, uint64_t *lo, uint64_t *hi)
{
    try {
        HipTimestampSource s(nullptr, start_date, end_date, false);
        *lo = s.start;
        *hi = s.end;
        return 0;
    } catch (const std::exception &) {
        return -1;
    }
}

// bitimage host pieces (src/bitimage.rs:38-44, 77-117, 277-293)
void vkh_bitimage_entropy(const uint8_t *data, size_t n, uint8_t out[32]) { bitimage_entropy(data, n, out); }
// depth, or -1 with the reference's error text in err
int vkh_parse_derivation_path(const char *path, uint32_t *comps, int cap, char *err, size_t err_cap)
{
    try {
        std::vector<uint32_t> c = parse_derivation_path(path);
        if ((int)c.size() > cap) return -2;
        for (size_t i = 0; i < c.size(); i++) comps[i] = c[i];
        return (int)c.size();
    } catch (const std::exception &e) {
        snprintf(err, err_cap, "%s", e.what());
        return -1;
    }
}
size_t vkh_increment_path_index(const char *path, char *out, size_t cap) { return (size_t)snprintf(out, cap, "%s", increment_path_index(path).c_str()); }

// the page compressors written in parquet.cpp (codec 1 = Snappy raw, 5 = Hadoop-framed LZ4 block): bytes written, or -1
long vkh_compress_page(int codec, const uint8_t *data, size_t n, uint8_t *out, size_t cap)
{
    try {
        const std::string r = compress_page_for_test(codec, std::string((const char *)data, n));
        if (r.size() > cap) return -1;
        memcpy(out, r.data(), r.size());
        return (long)r.size();
    } catch (const std::exception &) {
        return -1;
    }
}

// StorageOutput without a device: n records (key, hash160s, X || Y, source, optional matched address) -> Parquet chunks
// under `dir`; returns the number of files written.  matched[i] = "" means a plain key record.
int vkh_storage_write(const char *dir, const char *transform, uint64_t chunk_records, size_t n, const uint8_t *keys, const uint8_t *hc,
                      const uint8_t *hu, const uint8_t *xy, const char *const *sources, const char *const *matched, const char *compression)
{
    try {
        StorageOutput out(dir, transform, chunk_records, compression, 3, 0);
        for (size_t i = 0; i < n; i++) {
            DerivedKey d = KeyDeriver::from_device(keys + 32 * i, hc + 20 * i, hu + 20 * i);
            KeyDeriver::set_pubkeys(d, xy + 64 * i);
            if (matched[i][0]) {
                MatchInfo m{AddressType::P2pkhCompressed, matched[i]};
                out.hit(sources[i], transform, d, m);
            } else {
                out.key(sources[i], transform, d);
            }
        }
        return (int)out.finish().size();
    } catch (const std::exception &) {
        return -1;
    }
}

// The same sink fed with ready-made columns (what vk_generate_columns / vk_derive_columns return): `batches` runs of rows, run b
// holding rows[b] rows whose column c is data[b * VK_RC_COLUMNS + c] (len[...] bytes).  Between the runs, `singles` records are
// added through key() (records 0.., taken from the key arrays), so that both ways of filling a chunk meet in one file.
int vkh_storage_write_columns(const char *dir, const char *transform, uint64_t chunk_records, uint64_t chunk_bytes, size_t batches, const uint64_t *rows,
                              const uint8_t *const *data, const size_t *len, size_t singles, const uint8_t *keys, const uint8_t *hc, const uint8_t *hu,
                              const uint8_t *xy, const char *const *sources, const char *compression)
{
    try {
        StorageOutput out(dir, transform, chunk_records, compression, 3, chunk_bytes);
        size_t next_single = 0;
        for (size_t b = 0; b < batches; b++) {
            vk_record_columns rc;
            for (int c = 0; c < VK_RC_COLUMNS; c++) {
                rc.data[c] = const_cast<uint8_t *>(data[b * VK_RC_COLUMNS + c]);
                rc.cap[c] = rc.len[c] = len[b * VK_RC_COLUMNS + c];
            }
            rc.rows = rows[b];
            out.write_columns(rc, transform);
            const size_t upto = singles * (b + 1) / batches;
            for (; next_single < upto; next_single++) {
                DerivedKey d = KeyDeriver::from_device(keys + 32 * next_single, hc + 20 * next_single, hu + 20 * next_single);
                KeyDeriver::set_pubkeys(d, xy + 64 * next_single);
                out.key(sources[next_single], transform, d);
            }
        }
        return (int)out.finish().size();
    } catch (const std::exception &) {
        return -1;
    }
}

int vkh_parse_key(const char *s, uint8_t out[32])
{
    try {
        Key k = parse_private_key(s);
        memcpy(out, k.data(), 32);
        return 0;
    } catch (...) {
        return -1;
    }
}
// parse_cascade (:131-166): number of targets, or -1 with the message in err
int vkh_parse_cascade(const char *s, uint8_t *bits, uint64_t *targets, size_t cap, char *err, size_t err_cap)
{
    try {
        auto t = parse_cascade(s);
        if (t.size() > cap) return -1;
        for (size_t i = 0; i < t.size(); i++) {
            bits[i] = t[i].first;
            targets[i] = t[i].second;
        }
        return (int)t.size();
    } catch (const std::exception &e) {
        put(e.what(), err, err_cap);
        return -1;
    }
}
}
