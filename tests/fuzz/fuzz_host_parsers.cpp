// fuzz_host_parsers.cpp -- byte-level fuzz of the host-side parsers that read what a user supplies, built with
// -fsanitize=address,undefined -fno-sanitize-recover=all against the sanitizer build of libvuke_host.so (vuke_amd/host/Makefile: asan):
//   * the targets file      -> Matcher::load            (reference src/matcher.rs:45-59: one address per line, blank lines skipped)
//   * wordlist / stdin lines -> LineReader + the filter  (reference src/source/wordlist.rs:24-77, src/source/stdin.rs:45-60)
//   * derivation paths      -> parse_derivation_path     (reference src/bitimage.rs:77-117)
//   * and the small grammars beside them: transform names, private keys, cascade targets, segwit addresses, dates.
// Inputs: valid seeds mutated at byte level (flips, insertions, deletions, truncation, splices of other seeds, NULs, 0x80-0xFF runs) and
// pure random bytes.  What is checked: no sanitizer report, no exception leaving the C hooks, documented return ranges, and for the
// Matcher that a file of N distinct valid addresses plus arbitrary garbage lines counts >= the valid ones it was given.
//   usage: fuzz_host_parsers <cases per parser> <seed> <scratch dir>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct vk_transform_spec_c {
    uint32_t kind, variant, endian, chain_depth;
};
extern "C" {
int vkh_parse_transform(const char *s, vk_transform_spec_c *spec, char *name, size_t name_cap, char *err, size_t err_cap);
int vkh_segwit_decode(const char *hrp, const char *addr, int *ver, uint8_t *out, size_t cap);
int vkh_line_batches(const char *in_path, int strict, size_t max_lines, size_t max_bytes, const char *out_path, char *err, size_t err_cap);
int vkh_line_filter(const char *line, size_t n, char *out, size_t cap);
long vkh_matcher_load(const char *path, uint8_t *h160, uint8_t *kind, size_t cap, size_t *n_device);
int vkh_timestamp_range(const char *start_date, const char *end_date, uint64_t *lo, uint64_t *hi);
int vkh_parse_derivation_path(const char *path, uint32_t *comps, int cap, char *err, size_t err_cap);
size_t vkh_increment_path_index(const char *path, char *out, size_t cap);
int vkh_parse_key(const char *s, uint8_t out[32]);
int vkh_parse_cascade(const char *s, uint8_t *bits, uint64_t *targets, size_t cap, char *err, size_t err_cap);
}

static uint64_t g_s;
static uint64_t rnd()
{
    g_s ^= g_s << 13;
    g_s ^= g_s >> 7;
    g_s ^= g_s << 17;
    return g_s;
}
static uint32_t below(uint32_t n) { return n ? (uint32_t)(rnd() % n) : 0; }

static std::string mutate(const std::vector<std::string> &seeds)
{
    std::string s = below(8) == 0 ? std::string() : seeds[below((uint32_t)seeds.size())];
    if (s.empty()) {
        const uint32_t n = below(80);
        for (uint32_t i = 0; i < n; i++) s.push_back((char)rnd());
        return s;
    }
    if (below(6) == 0) return s; // the seed as it is: the accepting paths are walked too
    const uint32_t steps = 1 + below(6);
    for (uint32_t k = 0; k < steps; k++) {
        const uint32_t pos = below((uint32_t)s.size() + 1);
        switch (below(9)) {
        case 0: if (!s.empty()) s[below((uint32_t)s.size())] ^= (char)(1u << below(8)); break;
        case 1: s.insert(s.begin() + pos, (char)rnd()); break;
        case 2: if (!s.empty()) s.erase(s.begin() + below((uint32_t)s.size())); break;
        case 3: s.resize(below((uint32_t)s.size() + 1)); break;
        case 4: s.insert(pos, seeds[below((uint32_t)seeds.size())]); break;
        case 5: s.insert(s.begin() + pos, (char)(0x80 + below(0x80))); break;
        case 6: s.insert(pos, std::string(1 + below(40), "'/:0 \t9hm\xff"[below(10)])); break;
        case 7: if (!s.empty()) s[below((uint32_t)s.size())] = "\n\r\0 \t/'"[below(7)]; break;
        default: s.append(std::to_string(rnd())); break;
        }
    }
    return s;
}
// the hooks take C strings: a NUL inside ends the input there, which is one more mutation
static const char *cstr(const std::string &s) { return s.c_str(); }

#define CHECK(c)                                                                   \
    do {                                                                           \
        if (!(c)) {                                                                \
            fprintf(stderr, "fuzz_host_parsers: check failed: %s (line %d)\n", #c, __LINE__); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

int main(int argc, char **argv)
{
    const long cases = argc > 1 ? atol(argv[1]) : 10000;
    g_s = argc > 2 ? strtoull(argv[2], nullptr, 0) | 1 : 0x9E3779B97F4A7C15ull;
    const std::string dir = argc > 3 ? argv[3] : "/tmp";
    const std::vector<std::string> addrs = {"1BvBMSEYstWetqTFn5Au4m4GFg7xJaNVN2", "1JwSSubhmg6iPtRjtyqhUYYH7bZg3Lfy1T", "bc1qw508d6qejxtdg4y5r3zarvary0c5xw7kv8f3t4",
                                            "1A1zP1eP5QGefi2DMPTfTL5SLmv7DivfNa", "BC1QW508D6QEJXTDG4Y5R3ZARVARY0C5XW7KV8F3T4", "3J98t1WpEZ73CNmQviecrnyiWrnqRhWNLy",
                                            "bc1p5cyxnuxmeuwuvkwfem96lqzszd02n6xdcjrs20cac6yqjjwudpxqkedrcr", "  1BvBMSEYstWetqTFn5Au4m4GFg7xJaNVN2  ", "", "#comment"};
    const std::vector<std::string> paths = {"m/84'/0'/0'/0/0", "m/44h/0h/0h/0/1", "m", "m/0", "m/2147483647'", "m/2147483648", "M/0'/1", "m//0", "m/0'/", "84'/0'", "m/4294967295/1/2/3/4/5/6/7/8/9"};
    const std::vector<std::string> lines = {"correct horse battery staple", "  padded\t", "\xc3\xa9t\xc3\xa9", "\xe2\x80\x83""em space\xe2\x80\x83", "\xff\xfe", "", "a", "\xf0\x9f\x94\x91 key",
                                            "abandon abandon abandon abandon abandon abandon abandon abandon abandon abandon abandon about", "\xed\xa0\x80", "\xc0\xaf", "line\r"};
    const std::vector<std::string> names = {"sha256", "lcg:glibc:le", "xorshift:128+", "sha256_chain:indexed:le", "electrum:change", "multibit", "milksad", "lcg:a:b:c", "armory", "bitimage"};
    const std::vector<std::string> keys = {"c4bb5e8f1f6d6a9c2b7f0e4d3a1c9e8b7d6f5a4c3b2a19080706050403020100", "0x1", "1", "5KJvsngHeMpm884wtkJNzQGaCErckhHJBGFsvd3VyK5qMZXj3hS",
                                           "KwDiBf89QgGbjEhKnhXJuH7LrciVrZi3qYjgd9M7rFU73sVHnoWn", "ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff", "", "zz"};
    const std::vector<std::string> cascades = {"32:0xdeadbeef,32:0x80000001", "64:0x8000000000000001", "8:0x81,8:0xff, 8:128 ,8:200", "64:18446744073709551615", "0:1", "65:1", "32", ":", "32:0xfffffffff", "32:0x1"};
    const std::vector<std::string> dates = {"2011-01-01", "2011-13-01", "1970-01-01", "9999-12-31", "2011-02-30", "20110101", "2011-1-1", "-1-01-01"};
    std::vector<uint8_t> h160(20 * 65536), kind(65536);
    char buf[4096], err[512];
    long stats[8] = {0};

    for (long c = 0; c < cases; c++) {
        // ---- targets file -> Matcher::load
        {
            const std::string path = dir + "/fz_targets.txt";
            FILE *f = fopen(path.c_str(), "wb");
            CHECK(f);
            const uint32_t n = below(40);
            long valid_distinct = 0;
            bool seen[3] = {false, false, false};
            for (uint32_t i = 0; i < n; i++) {
                std::string ln;
                const uint32_t pick = below(6);
                if (pick < 3) { // an untouched valid address on a line of its own
                    ln = addrs[pick];
                    if (!seen[pick]) valid_distinct++;
                    seen[pick] = true;
                } else {
                    ln = mutate(addrs);
                    for (char &ch : ln)
                        if (ch == '\n' && below(2)) ch = 'x';
                }
                fwrite(ln.data(), 1, ln.size(), f);
                fputc('\n', f);
            }
            if (below(4) == 0) fwrite("1BvBMSEYstWetqTFn5Au", 1, 20, f); // no newline at the end
            fclose(f);
            size_t nd = 0;
            const long cnt = vkh_matcher_load(path.c_str(), h160.data(), kind.data(), kind.size(), &nd);
            CHECK(cnt >= -1);
            if (cnt >= 0) {
                CHECK(cnt >= valid_distinct);     // every line is an entry of the reference's HashSet<String>: garbage counts too
                CHECK(nd <= (size_t)cnt);         // the device form holds only what decodes
                stats[0]++;
            }
        }
        // ---- wordlist / stdin lines -> LineReader
        {
            const std::string in = dir + "/fz_lines.txt", out = dir + "/fz_lines.out";
            FILE *f = fopen(in.c_str(), "wb");
            CHECK(f);
            const uint32_t n = below(30);
            for (uint32_t i = 0; i < n; i++) {
                const std::string ln = mutate(lines);
                fwrite(ln.data(), 1, ln.size(), f);
                if (below(10)) fputc('\n', f);
            }
            fclose(f);
            const int strict = (int)below(2);
            const int rc = vkh_line_batches(in.c_str(), strict, 1 + below(8), 1 + below(200), out.c_str(), err, sizeof err);
            CHECK(rc == 0 || rc == 1);
            CHECK(strict || rc == 0);             // only the stdin source stops with the reader's error
            const std::string one = mutate(lines);
            const int fl = vkh_line_filter(one.data(), one.size(), buf, sizeof buf);
            CHECK(fl == -2 || (fl >= 0 && (size_t)fl <= one.size()));
            stats[1]++;
        }
        // ---- derivation paths
        {
            uint32_t comps[64];
            const std::string p = mutate(paths);
            const int d = vkh_parse_derivation_path(cstr(p), comps, 64, err, sizeof err);
            CHECK(d >= -2 && d <= 64);
            if (d >= 0) stats[2]++;
            const size_t n = vkh_increment_path_index(cstr(p), buf, sizeof buf);
            CHECK(n < 4096 + 64);
        }
        // ---- the small grammars
        {
            vk_transform_spec_c sp;
            char name[64];
            const std::string t = mutate(names);
            const int rc = vkh_parse_transform(cstr(t), &sp, name, sizeof name, err, sizeof err);
            CHECK(rc == 0 || rc != 0);
            if (rc == 0) stats[3]++;
            uint8_t k[32];
            const std::string ks = mutate(keys);
            if (vkh_parse_key(cstr(ks), k) == 0) stats[4]++;
            uint8_t bits[16];
            uint64_t tg[16];
            const std::string cs = mutate(cascades);
            const int nc = vkh_parse_cascade(cstr(cs), bits, tg, 16, err, sizeof err);
            CHECK(nc <= 16);
            if (nc > 0) stats[5]++;
            int ver = 0;
            uint8_t prog[64];
            const std::string a = mutate(addrs);
            const int pl = vkh_segwit_decode(below(4) ? "bc" : "tb", cstr(a), &ver, prog, sizeof prog);
            CHECK(pl <= 40);
            if (pl >= 0) stats[6]++;
            uint64_t lo, hi;
            const std::string d0 = mutate(dates), d1 = mutate(dates);
            if (vkh_timestamp_range(cstr(d0), cstr(d1), &lo, &hi) == 0) {
                CHECK(lo <= hi);
                stats[7]++;
            }
        }
    }
    printf("fuzz_host_parsers: %ld cases per parser, seed %llu: accepted -- targets files %ld, line files %ld, derivation paths %ld, transform names %ld, "
           "keys %ld, cascades %ld, segwit addresses %ld, date ranges %ld; no sanitizer report, every return value in its documented range\n",
           cases, (unsigned long long)(argc > 2 ? strtoull(argv[2], nullptr, 0) : 0), stats[0], stats[1], stats[2], stats[3], stats[4], stats[5], stats[6], stats[7]);
    return 0;
}
