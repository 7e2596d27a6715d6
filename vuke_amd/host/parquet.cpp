// parquet.cpp -- the reference's 19-column result record as Parquet files: StorageOutput + a minimal
// Parquet writer (PLAIN encoding, one data page per column per row group, Thrift compact metadata).
//
// Schema and column semantics: src/storage/schema.rs:78-111 (result_schema) and :137-262 (records_to_batch);
// partition layout `DIR/transform=<first transform>/date=<UTC date>/chunk_NNNN.parquet` and record-count rotation:
// src/storage/parquet_backend.rs:86-140; console + storage fan-out: src/main.rs:666-672 (MultiOutput).
// Page compression (src/main.rs:70-83, 134-141, 640-649): none, snappy and lz4 (written here: greedy hash-table matchers
// emitting the raw Snappy format and Hadoop-framed LZ4 blocks, which is what the Parquet codec ids 1 and 5 mean), gzip
// (zlib) and zstd -- the reference's default, level 3 -- through libzstd.so.1 loaded at run time (the image ships the
// library without its header).  Cloud upload, Iceberg and the DuckDB query layer are out of scope.
#include "vuke_host.hpp"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstring>
#include <ctime>
#include <dlfcn.h>
#include <future>
#include <stdexcept>
#include <sys/stat.h>
#include <thread>
#include <zlib.h>

namespace vuke {

namespace {

// ---- Thrift compact protocol (just what parquet.thrift's metadata needs)
class Thrift {
public:
    std::string out;
    void begin() { stack_.push_back(last_); last_ = 0; }
    void end()
    {
        out.push_back(0);
        last_ = stack_.back();
        stack_.pop_back();
    }
    void field(int id, int type)
    {
        const int delta = id - last_;
        if (delta > 0 && delta <= 15) out.push_back((char)((delta << 4) | type));
        else {
            out.push_back((char)type);
            varint(zigzag((int64_t)id));
        }
        last_ = id;
    }
    void i32(int id, int32_t v) { field(id, 5); varint(zigzag(v)); }
    void i64(int id, int64_t v) { field(id, 6); varint(zigzag(v)); }
    void i8(int id, int8_t v) { field(id, 3); out.push_back((char)v); }
    void boolean(int id, bool v) { field(id, v ? 1 : 2); }
    void str(int id, const std::string &s) { field(id, 8); binary(s); }
    void binary(const std::string &s) { varint(s.size()); out += s; }
    void list(int id, int elem_type, size_t n)
    {
        field(id, 9);
        if (n < 15) out.push_back((char)((n << 4) | (unsigned)elem_type));
        else {
            out.push_back((char)(0xF0 | elem_type));
            varint(n);
        }
    }
    void struct_field(int id) { field(id, 12); begin(); }
    void raw_i32(int32_t v) { varint(zigzag(v)); } // list element
    void varint(uint64_t v)
    {
        while (v >= 0x80) {
            out.push_back((char)(v | 0x80));
            v >>= 7;
        }
        out.push_back((char)v);
    }

private:
    static uint64_t zigzag(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }
    int last_ = 0;
    std::vector<int> stack_;
};

enum PType { P_INT32 = 1, P_INT64 = 2, P_BYTE_ARRAY = 6, P_FIXED = 7 };
enum Logical { L_STRING, L_TS_MILLIS_UTC, L_UINT16, L_UINT8, L_NONE };
struct ColumnDef {
    const char *name;
    PType type;
    bool optional;
    Logical logical;
};
// src/storage/schema.rs:78-111, in order
const ColumnDef COLS[19] = {
    {"source", P_BYTE_ARRAY, false, L_STRING},
    {"transform", P_BYTE_ARRAY, false, L_STRING},
    {"chain", P_BYTE_ARRAY, false, L_STRING},
    {"timestamp", P_INT64, false, L_TS_MILLIS_UTC},
    {"matched_target", P_BYTE_ARRAY, true, L_STRING},
    {"private_key_raw", P_FIXED, false, L_NONE},
    {"private_key_hex", P_BYTE_ARRAY, false, L_STRING},
    {"private_key_decimal", P_BYTE_ARRAY, false, L_STRING},
    {"private_key_binary", P_BYTE_ARRAY, false, L_STRING},
    {"private_key_bit_length", P_INT32, false, L_UINT16},
    {"private_key_hamming_weight", P_INT32, false, L_UINT16},
    {"private_key_leading_zeros", P_INT32, false, L_UINT8},
    {"pubkey_compressed", P_BYTE_ARRAY, true, L_STRING},
    {"pubkey_uncompressed", P_BYTE_ARRAY, true, L_STRING},
    {"address_p2pkh_compressed", P_BYTE_ARRAY, true, L_STRING},
    {"address_p2pkh_uncompressed", P_BYTE_ARRAY, true, L_STRING},
    {"address_p2wpkh", P_BYTE_ARRAY, true, L_STRING},
    {"wif_compressed", P_BYTE_ARRAY, true, L_STRING},
    {"wif_uncompressed", P_BYTE_ARRAY, true, L_STRING},
};

void put_le32(std::string &s, uint32_t v)
{
    for (int i = 0; i < 4; i++) s.push_back((char)(v >> (8 * i)));
}
void put_le64(std::string &s, uint64_t v)
{
    for (int i = 0; i < 8; i++) s.push_back((char)(v >> (8 * i)));
}

// one column of one row group while it is being filled
struct ColumnBuf {
    std::string values;        // PLAIN-encoded non-null values
    std::vector<uint8_t> defs; // definition level per row (optional columns)
    void str(const std::string &v)
    {
        put_le32(values, (uint32_t)v.size());
        values += v;
    }
    void opt_str(const std::string *v)
    {
        defs.push_back(v ? 1 : 0);
        if (v) str(*v);
    }
};

std::string rle_levels(const std::vector<uint8_t> &defs)
{ // RLE / bit-packed hybrid, bit width 1, RLE runs only
    Thrift t;
    size_t i = 0;
    while (i < defs.size()) {
        size_t j = i;
        while (j < defs.size() && defs[j] == defs[i]) j++;
        t.varint((uint64_t)(j - i) << 1);
        t.out.push_back((char)defs[i]);
        i = j;
    }
    return t.out;
}

struct ChunkMeta {
    int64_t data_page_offset, size, raw_size, num_values;
};

static inline uint32_t load32(const uint8_t *p)
{
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}

// Snappy raw format (format_description.txt): varint length, then literals (tag 00) and copies with a 2-byte offset (tag 10)
std::string snappy_compress(const std::string &in)
{
    const uint8_t *p = (const uint8_t *)in.data();
    const size_t n = in.size();
    if (n > 0xFFFFFFFFull) throw std::runtime_error("snappy: page larger than 4 GiB");
    std::string out;
    out.reserve(n + n / 6 + 32);
    for (size_t v = n;;) {
        uint8_t b = v & 0x7F;
        v >>= 7;
        out.push_back((char)(b | (v ? 0x80 : 0)));
        if (!v) break;
    }
    auto literal = [&](size_t from, size_t to) {
        while (from < to) {
            const size_t len = std::min<size_t>(to - from, 1u << 24), l1 = len - 1;
            if (l1 < 60) out.push_back((char)(l1 << 2));
            else if (l1 < 256) { out.push_back((char)(60 << 2)); out.push_back((char)l1); }
            else if (l1 < 65536) { out.push_back((char)(61 << 2)); out.push_back((char)l1); out.push_back((char)(l1 >> 8)); }
            else { out.push_back((char)(62 << 2)); out.push_back((char)l1); out.push_back((char)(l1 >> 8)); out.push_back((char)(l1 >> 16)); }
            out.append((const char *)p + from, len);
            from += len;
        }
    };
    std::vector<uint32_t> table(1u << 14, 0); // position + 1 of the last occurrence of a 4-byte hash
    size_t i = 0, lit = 0;
    while (i + 4 <= n) {
        const uint32_t v = load32(p + i), h = (v * 0x1e35a7bdu) >> 18;
        const size_t cand = table[h];
        table[h] = (uint32_t)(i + 1);
        if (cand && i - (cand - 1) <= 65535 && load32(p + cand - 1) == v) {
            const size_t m = cand - 1, off = i - m;
            size_t len = 4;
            while (i + len < n && p[m + len] == p[i + len]) len++;
            literal(lit, i);
            for (size_t rem = len; rem;) {
                const size_t l = std::min<size_t>(rem, 64);
                out.push_back((char)(((l - 1) << 2) | 2));
                out.push_back((char)off);
                out.push_back((char)(off >> 8));
                rem -= l;
            }
            i += len;
            lit = i;
        } else {
            i++;
        }
    }
    literal(lit, n);
    return out;
}

// One LZ4 block (lz4_Block_format.md) in Hadoop framing: BE32 uncompressed size, BE32 compressed size, block -- the
// layout behind Parquet's codec id 5 ("LZ4"), which parquet-rs writes for Compression::LZ4
std::string lz4_hadoop_compress(const std::string &in)
{
    const uint8_t *p = (const uint8_t *)in.data();
    const size_t n = in.size();
    if (n > 0x7E000000ull) throw std::runtime_error("lz4: page too large for one block");
    std::string blk;
    blk.reserve(n + n / 255 + 32);
    auto lengths = [&](size_t v) { // the 255-run extension of a nibble that said 15
        for (; v >= 255; v -= 255) blk.push_back((char)255);
        blk.push_back((char)v);
    };
    std::vector<uint32_t> table(1u << 14, 0);
    size_t i = 0, lit = 0;
    const size_t mflimit = n >= 12 ? n - 12 : 0, matchend = n >= 5 ? n - 5 : 0; // last match starts 12 before the end, ends 5 before it
    while (n >= 13 && i <= mflimit) {
        const uint32_t v = load32(p + i), h = (v * 2654435761u) >> 18;
        const size_t cand = table[h];
        table[h] = (uint32_t)(i + 1);
        if (cand && i - (cand - 1) <= 65535 && load32(p + cand - 1) == v) {
            const size_t m = cand - 1, off = i - m;
            size_t len = 4;
            while (i + len < matchend && p[m + len] == p[i + len]) len++;
            const size_t L = i - lit, M = len - 4;
            blk.push_back((char)((std::min<size_t>(L, 15) << 4) | std::min<size_t>(M, 15)));
            if (L >= 15) lengths(L - 15);
            blk.append((const char *)p + lit, L);
            blk.push_back((char)off);
            blk.push_back((char)(off >> 8));
            if (M >= 15) lengths(M - 15);
            i += len;
            lit = i;
        } else {
            i++;
        }
    }
    const size_t L = n - lit;
    blk.push_back((char)(std::min<size_t>(L, 15) << 4));
    if (L >= 15) lengths(L - 15);
    blk.append((const char *)p + lit, L);
    std::string out;
    const uint32_t a = (uint32_t)n, b = (uint32_t)blk.size();
    for (int sh = 24; sh >= 0; sh -= 8) out.push_back((char)(a >> sh));
    for (int sh = 24; sh >= 0; sh -= 8) out.push_back((char)(b >> sh));
    out += blk;
    return out;
}

// parquet.thrift CompressionCodec: UNCOMPRESSED = 0, SNAPPY = 1, GZIP = 2, LZ4 = 5, ZSTD = 6
std::string compress_page(int codec, int level, const std::string &in)
{
    if (codec == 0) return in;
    if (codec == 1) return snappy_compress(in);
    if (codec == 5) return lz4_hadoop_compress(in);
    if (codec == 2) {
        z_stream z;
        memset(&z, 0, sizeof z);
        if (deflateInit2(&z, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("zlib: deflateInit2 failed");
        std::string out(deflateBound(&z, in.size()) + 32, '\0');
        z.next_in = (Bytef *)in.data();
        z.avail_in = (uInt)in.size();
        z.next_out = (Bytef *)&out[0];
        z.avail_out = (uInt)out.size();
        const int rc = deflate(&z, Z_FINISH);
        deflateEnd(&z);
        if (rc != Z_STREAM_END) throw std::runtime_error("zlib: deflate failed");
        out.resize(z.total_out);
        return out;
    }
    // zstd: the three entry points of the stable API, resolved once
    typedef size_t (*bound_fn)(size_t);
    typedef size_t (*comp_fn)(void *, size_t, const void *, size_t, int);
    typedef unsigned (*err_fn)(size_t);
    static void *lib = dlopen("libzstd.so.1", RTLD_NOW);
    static bound_fn bound = lib ? (bound_fn)dlsym(lib, "ZSTD_compressBound") : nullptr;
    static comp_fn comp = lib ? (comp_fn)dlsym(lib, "ZSTD_compress") : nullptr;
    static err_fn iserr = lib ? (err_fn)dlsym(lib, "ZSTD_isError") : nullptr;
    if (!bound || !comp || !iserr) throw std::runtime_error("--compression zstd: libzstd.so.1 is not available on this machine (use gzip or none)");
    std::string out(bound(in.size()), '\0');
    const size_t n = comp(&out[0], out.size(), in.data(), in.size(), level);
    if (iserr(n)) throw std::runtime_error("zstd: compression failed");
    out.resize(n);
    return out;
}
struct RowGroupMeta {
    std::vector<ChunkMeta> cols;
    int64_t rows, bytes;
};

} // namespace

struct StorageOutput::Impl {
    std::string base_dir, transform, date;
    uint64_t max_chunk_records, max_chunk_bytes = 0, chunk_bytes = 0;
    int codec = 0, level = 3;
    uint32_t chunk_index = 0;
    FILE *fp = nullptr;
    std::string path;
    uint64_t file_pos = 0, chunk_records = 0, total_records = 0;
    std::vector<RowGroupMeta> groups;
    ColumnBuf cols[19];
    uint64_t group_rows = 0;
    std::vector<std::string> completed;
    std::mutex mu;

    static void mkdirs(const std::string &p)
    {
        for (size_t i = 1; i <= p.size(); i++)
            if (i == p.size() || p[i] == '/') {
                const std::string d = p.substr(0, i);
                if (mkdir(d.c_str(), 0777) != 0 && errno != EEXIST) throw std::runtime_error("cannot create directory " + d);
            }
    }
    void open_chunk()
    { // generate_chunk_path, src/storage/parquet_backend.rs:86-108
        if (date.empty()) {
            const time_t now = time(nullptr);
            struct tm tmv;
            gmtime_r(&now, &tmv);
            char b[16];
            strftime(b, sizeof b, "%Y-%m-%d", &tmv);
            date = b;
        }
        const std::string dir = base_dir + "/transform=" + transform + "/date=" + date;
        mkdirs(dir);
        char name[32];
        snprintf(name, sizeof name, "/chunk_%04u.parquet", ++chunk_index);
        path = dir + name;
        fp = fopen(path.c_str(), "wb");
        if (!fp) throw std::runtime_error("cannot create " + path);
        fwrite("PAR1", 1, 4, fp);
        file_pos = 4;
        chunk_records = 0;
        chunk_bytes = 0;
        groups.clear();
    }
    void write(const std::string &s)
    {
        if (fwrite(s.data(), 1, s.size(), fp) != s.size()) throw std::runtime_error("write failed: " + path);
        file_pos += s.size();
    }
    // one row group: the nineteen column chunks (one PLAIN data page each) in schema order
    void emit_group(uint64_t rows, const size_t (&raw)[19] /* uncompressed page sizes */, const std::string (&packed)[19])
    {
        RowGroupMeta g;
        g.rows = (int64_t)rows;
        g.bytes = 0;
        for (int c = 0; c < 19; c++) {
            Thrift h; // PageHeader
            h.begin();
            h.i32(1, 0); // DATA_PAGE
            h.i32(2, (int32_t)raw[c]);
            h.i32(3, (int32_t)packed[c].size());
            h.struct_field(5); // DataPageHeader
            h.i32(1, (int32_t)rows);
            h.i32(2, 0); // PLAIN
            h.i32(3, 3); // RLE
            h.i32(4, 3);
            h.end();
            h.end();
            ChunkMeta m;
            m.data_page_offset = (int64_t)file_pos;
            m.size = (int64_t)(h.out.size() + packed[c].size());
            m.raw_size = (int64_t)(h.out.size() + raw[c]);
            m.num_values = (int64_t)rows;
            write(h.out);
            write(packed[c]);
            g.bytes += m.raw_size;
            g.cols.push_back(m);
        }
        groups.push_back(g);
    }
    void flush_group()
    {
        if (group_rows == 0) return;
        // page bodies are built and compressed column by column on their own threads, then written in schema order
        std::string body[19], packed[19];
        std::exception_ptr err[19];
        {
            std::vector<std::thread> th;
            for (int c = 0; c < 19; c++)
                th.emplace_back([&, c]() {
                    try {
                        if (COLS[c].optional) {
                            const std::string lv = rle_levels(cols[c].defs);
                            put_le32(body[c], (uint32_t)lv.size());
                            body[c] += lv;
                        }
                        body[c] += cols[c].values;
                        packed[c] = compress_page(codec, level, body[c]);
                    } catch (...) {
                        err[c] = std::current_exception();
                    }
                });
            for (auto &t : th) t.join();
        }
        for (int c = 0; c < 19; c++)
            if (err[c]) std::rethrow_exception(err[c]);
        size_t raw[19];
        for (int c = 0; c < 19; c++) raw[c] = body[c].size();
        emit_group(group_rows, raw, packed);
        for (int c = 0; c < 19; c++) {
            cols[c].values.clear();
            cols[c].defs.clear();
        }
        group_rows = 0;
    }
    void close_chunk()
    {
        if (!fp) return;
        flush_group();
        Thrift t; // FileMetaData
        t.begin();
        t.i32(1, 1);
        t.list(2, 12, 20); // schema: root + 19 leaves
        t.begin();
        t.str(4, "arrow_schema");
        t.i32(5, 19);
        t.end();
        for (const ColumnDef &c : COLS) {
            t.begin();
            t.i32(1, c.type);
            if (c.type == P_FIXED) t.i32(2, 32);
            t.i32(3, c.optional ? 1 : 0);
            t.str(4, c.name);
            if (c.logical == L_STRING) t.i32(6, 0);        // UTF8
            if (c.logical == L_TS_MILLIS_UTC) t.i32(6, 9); // TIMESTAMP_MILLIS
            if (c.logical == L_UINT16) t.i32(6, 12);       // UINT_16
            if (c.logical == L_UINT8) t.i32(6, 11);        // UINT_8
            if (c.logical != L_NONE) {
                t.struct_field(10); // LogicalType union
                if (c.logical == L_STRING) {
                    t.struct_field(1);
                    t.end();
                } else if (c.logical == L_TS_MILLIS_UTC) {
                    t.struct_field(8);
                    t.boolean(1, true);
                    t.struct_field(2); // TimeUnit
                    t.struct_field(1); // MILLIS
                    t.end();
                    t.end();
                    t.end();
                } else {
                    t.struct_field(10); // IntType
                    t.i8(1, c.logical == L_UINT16 ? 16 : 8);
                    t.boolean(2, false);
                    t.end();
                }
                t.end();
            }
            t.end();
        }
        int64_t rows = 0;
        for (auto &g : groups) rows += g.rows;
        t.i64(3, rows);
        t.list(4, 12, groups.size());
        for (auto &g : groups) {
            t.begin();
            t.list(1, 12, 19);
            for (int c = 0; c < 19; c++) {
                t.begin(); // ColumnChunk
                t.i64(2, g.cols[c].data_page_offset);
                t.struct_field(3); // ColumnMetaData
                t.i32(1, COLS[c].type);
                t.list(2, 5, 2);
                t.raw_i32(0);
                t.raw_i32(3);
                t.list(3, 8, 1);
                t.binary(COLS[c].name);
                t.i32(4, codec);
                t.i64(5, g.cols[c].num_values);
                t.i64(6, g.cols[c].raw_size);
                t.i64(7, g.cols[c].size);
                t.i64(9, g.cols[c].data_page_offset);
                t.end();
                t.end();
            }
            t.i64(2, g.bytes);
            t.i64(3, g.rows);
            t.end();
        }
        t.str(6, "vuke-hip (MI355X engine) parquet writer");
        t.end();
        std::string tail = t.out;
        put_le32(tail, (uint32_t)t.out.size());
        tail += "PAR1";
        write(tail);
        fclose(fp);
        fp = nullptr;
        completed.push_back(path);
    }
    // A run of records whose value columns were made on the device (vk_generate_columns / vk_derive_columns: Parquet PLAIN bytes per
    // column).  Same rotation rules as add(), applied row by row on the lengths; the pages of a row group are only framed,
    // compressed and written here, many at a time on the host's threads.
    void add_columns(const vk_record_columns &b, const std::string &tf, const uint8_t *source_col, size_t source_len)
    {
        std::lock_guard<std::mutex> lk(mu);
        const uint64_t R = b.rows;
        if (R == 0) return;
        flush_group(); // records added one by one so far form their own row group
        // device column -> schema column; bytes per row of the fixed-width ones (0 = variable: walk the length prefixes)
        static const int SCHEMA_OF[VK_RC_COLUMNS] = {0, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18};
        static const uint32_t WIDTH[VK_RC_COLUMNS] = {0, 32, 68, 0, 260, 4, 4, 4, 70, 134, 0, 0, 46, 56, 55};
        const uint8_t *data[VK_RC_COLUMNS];
        size_t len[VK_RC_COLUMNS];
        for (int c = 0; c < VK_RC_COLUMNS; c++) {
            data[c] = b.data[c];
            len[c] = b.len[c];
        }
        if (source_col) { // the caller's labels (vk_derive_columns leaves the column to it)
            data[VK_RC_SOURCE] = source_col;
            len[VK_RC_SOURCE] = source_len;
        }
        uint64_t fixed = 0;
        for (int c = 0; c < VK_RC_COLUMNS; c++) {
            fixed += WIDTH[c];
            if (WIDTH[c] && len[c] != R * WIDTH[c]) throw std::runtime_error("add_columns: a fixed-width column does not hold rows x width bytes");
        }
        fixed += 4 + tf.size() + 4 + 7 + 8; // transform, chain = "bitcoin", timestamp
        struct Piece {
            uint64_t row0, row1;
            size_t v0[VK_RC_COLUMNS], v1[VK_RC_COLUMNS];
            uint64_t records_after, bytes_after;
            bool close_after;
        };
        std::vector<Piece> pieces;
        size_t cur[VK_RC_COLUMNS] = {0};
        Piece pc;
        pc.row0 = 0;
        memset(pc.v0, 0, sizeof pc.v0);
        uint64_t sim_records = fp ? chunk_records : 0, sim_bytes = fp ? chunk_bytes : 0, grows = 0;
        for (uint64_t r = 0; r < R; r++) {
            uint64_t sz = fixed;
            for (int c = 0; c < VK_RC_COLUMNS; c++) {
                if (WIDTH[c]) continue;
                if (cur[c] + 4 > len[c]) throw std::runtime_error("add_columns: a string column ends inside a value");
                const uint32_t l = load32(data[c] + cur[c]);
                cur[c] += 4 + (size_t)l;
                sz += 4 + l;
            }
            sim_bytes += sz;
            sim_records++;
            grows++;
            const bool rotate = (max_chunk_records && sim_records >= max_chunk_records) || (max_chunk_bytes && sim_bytes >= max_chunk_bytes);
            if (grows == 65536 || rotate || r + 1 == R) {
                pc.row1 = r + 1;
                for (int c = 0; c < VK_RC_COLUMNS; c++) pc.v1[c] = WIDTH[c] ? (size_t)(r + 1) * WIDTH[c] : cur[c];
                pc.close_after = rotate;
                pc.records_after = sim_records;
                pc.bytes_after = sim_bytes;
                pieces.push_back(pc);
                pc.row0 = r + 1;
                memcpy(pc.v0, pc.v1, sizeof pc.v0);
                grows = 0;
                if (rotate) sim_records = sim_bytes = 0;
            }
        }
        for (int c = 0; c < VK_RC_COLUMNS; c++)
            if (!WIDTH[c] && cur[c] != len[c]) throw std::runtime_error("add_columns: a string column holds more values than rows");
        const int64_t now_ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        // pages of up to WAVE row groups at a time: built and compressed on the host's threads while the previous wave's pages are
        // being written (one writer at a time, in order)
        const size_t WAVE = 8;
        const unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
        struct Wave {
            size_t w0 = 0, wn = 0;
            std::vector<std::array<std::string, 19>> packed;
            std::vector<std::array<size_t, 19>> raw;
        };
        auto write_wave = [this, &pieces](Wave w) {
            for (size_t pi = 0; pi < w.wn; pi++) {
                const Piece &q = pieces[w.w0 + pi];
                if (!fp) open_chunk();
                std::string ps[19];
                size_t raw[19];
                for (int c = 0; c < 19; c++) {
                    ps[c].swap(w.packed[pi][c]);
                    raw[c] = w.raw[pi][c];
                }
                emit_group(q.row1 - q.row0, raw, ps);
                chunk_records = q.records_after; // what add() would have counted up to this row
                chunk_bytes = q.bytes_after;
                total_records += q.row1 - q.row0;
                if (q.close_after) close_chunk();
            }
        };
        std::future<void> writer;
        auto drain = [&]() {
            if (writer.valid()) writer.get(); // rethrows what the writer threw
        };
        try {
            for (size_t w0 = 0; w0 < pieces.size(); w0 += WAVE) {
                Wave w;
                w.w0 = w0;
                w.wn = std::min(WAVE, pieces.size() - w0);
                w.packed.resize(w.wn);
                w.raw.resize(w.wn);
                std::atomic<size_t> next{0};
                std::exception_ptr err;
                std::mutex err_mu;
                auto work = [&]() {
                    for (;;) {
                        const size_t task = next.fetch_add(1);
                        if (task >= w.wn * 19) return;
                        const size_t pi = task / 19;
                        const int col = (int)(task % 19);
                        const Piece &q = pieces[w0 + pi];
                        const uint64_t n = q.row1 - q.row0;
                        try {
                            std::string out;
                            if (COLS[col].optional) { // definition levels: one RLE run
                                Thrift t;
                                t.varint(n << 1);
                                t.out.push_back(col == 4 ? 0 : 1);
                                put_le32(out, (uint32_t)t.out.size());
                                out += t.out;
                            }
                            if (col == 1 || col == 2) {
                                std::string one;
                                const std::string &v = col == 1 ? tf : std::string("bitcoin");
                                put_le32(one, (uint32_t)v.size());
                                one += v;
                                out.reserve(out.size() + n * one.size());
                                for (uint64_t i = 0; i < n; i++) out += one;
                            } else if (col == 3) {
                                std::string one;
                                put_le64(one, (uint64_t)now_ms);
                                out.reserve(n * 8);
                                for (uint64_t i = 0; i < n; i++) out += one;
                            } else if (col != 4) {
                                int dc = 0;
                                while (SCHEMA_OF[dc] != col) dc++;
                                out.reserve(out.size() + (q.v1[dc] - q.v0[dc]));
                                out.append((const char *)data[dc] + q.v0[dc], q.v1[dc] - q.v0[dc]);
                            }
                            w.raw[pi][col] = out.size();
                            if (codec == 0) w.packed[pi][col] = std::move(out); // an uncompressed page is its body
                            else w.packed[pi][col] = compress_page(codec, level, out);
                        } catch (...) {
                            std::lock_guard<std::mutex> g(err_mu);
                            if (!err) err = std::current_exception();
                        }
                    }
                };
                std::vector<std::thread> th;
                for (unsigned t = 0; t < nt; t++) th.emplace_back(work);
                for (auto &t : th) t.join();
                if (err) std::rethrow_exception(err);
                drain();
                writer = std::async(std::launch::async, write_wave, std::move(w));
            }
            drain();
        } catch (...) {
            if (writer.valid()) {
                try {
                    writer.get();
                } catch (...) {
                }
            }
            throw;
        }
    }
    void add(const std::string &source, const std::string &tf, const DerivedKey &d, const std::string *matched)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!fp) open_chunk();
        size_t before = 0;
        for (const ColumnBuf &c : cols) before += c.values.size();
        const int64_t now_ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        cols[0].str(source);
        cols[1].str(tf);
        cols[2].str("bitcoin");
        put_le64(cols[3].values, (uint64_t)now_ms);
        cols[4].opt_str(matched);
        cols[5].values.append((const char *)d.raw.data(), 32);
        cols[6].str(d.private_key_hex);
        cols[7].str(d.private_key_decimal);
        cols[8].str(d.private_key_binary);
        put_le32(cols[9].values, d.bit_length);
        put_le32(cols[10].values, d.hamming_weight);
        put_le32(cols[11].values, d.leading_zeros);
        cols[12].opt_str(d.pubkey_compressed.empty() ? nullptr : &d.pubkey_compressed);
        cols[13].opt_str(d.pubkey_uncompressed.empty() ? nullptr : &d.pubkey_uncompressed);
        cols[14].opt_str(&d.p2pkh_compressed);
        cols[15].opt_str(&d.p2pkh_uncompressed);
        cols[16].opt_str(&d.p2wpkh);
        cols[17].opt_str(&d.wif_compressed);
        cols[18].opt_str(&d.wif_uncompressed);
        size_t after = 0;
        for (const ColumnBuf &c : cols) after += c.values.size();
        chunk_bytes += after - before; // the record's in-memory size (the reference adds get_array_memory_size of each batch, :212)
        group_rows++;
        chunk_records++;
        total_records++;
        if (group_rows == 65536) flush_group();
        if ((max_chunk_records && chunk_records >= max_chunk_records) || (max_chunk_bytes && chunk_bytes >= max_chunk_bytes))
            close_chunk(); // should_rotate, :110-131
    }
};

std::string compress_page_for_test(int codec, const std::string &in) { return compress_page(codec, codec == 2 ? 6 : 3, in); }

StorageOutput::StorageOutput(const std::string &base_dir, const std::string &transform_name, uint64_t max_chunk_records, const std::string &compression,
                             int level, uint64_t max_chunk_bytes)
    : impl_(new Impl)
{
    impl_->max_chunk_bytes = max_chunk_bytes;
    impl_->base_dir = base_dir;
    impl_->transform = transform_name;
    impl_->max_chunk_records = max_chunk_records;
    impl_->level = level;
    if (compression == "none") impl_->codec = 0;
    else if (compression == "snappy") impl_->codec = 1;
    else if (compression == "gzip") impl_->codec = 2;
    else if (compression == "lz4") impl_->codec = 5;
    else if (compression == "zstd") impl_->codec = 6;
    else {
        delete impl_;
        throw std::invalid_argument("invalid value '" + compression + "' for '--compression <COMPRESSION>' [possible values: none, snappy, gzip, lz4, zstd]");
    }
    if (impl_->codec == 2 && (level < 0 || level > 9)) impl_->level = 6;
    if (impl_->codec) (void)compress_page(impl_->codec, impl_->level, std::string("probe")); // fail now, not at the first page
}
StorageOutput::~StorageOutput()
{
    try {
        finish();
    } catch (...) {
    }
    delete impl_;
}
void StorageOutput::key(const std::string &source, const std::string &transform, const DerivedKey &d) { impl_->add(source, transform, d, nullptr); }
void StorageOutput::hit(const std::string &source, const std::string &transform, const DerivedKey &d, const MatchInfo &m)
{
    impl_->add(source, transform, d, &m.address);
}
void StorageOutput::flush() {}
bool StorageOutput::write_columns(const vk_record_columns &cols, const std::string &transform, const uint8_t *source_col, size_t source_len)
{
    impl_->add_columns(cols, transform, source_col, source_len);
    return true;
}
std::vector<std::string> StorageOutput::finish()
{
    std::lock_guard<std::mutex> lk(impl_->mu);
    impl_->close_chunk();
    return impl_->completed;
}
uint64_t StorageOutput::records() const { return impl_->total_records; }

} // namespace vuke
