// source.cpp -- HipRangeSource::process: the GPU replacement for RangeSource::process
// (src/source/range.rs:26-79).  Scan mode = vk_sweep (fused transform -> k*G -> hash160 -> probe)
// with the rare hits replayed through KeyDeriver + Output::hit on the host; generate mode =
// vk_generate with one Output::key per emitted key.  Counters follow the reference exactly:
// inputs = end - start + 1, keys = keys the transforms emitted, matches = keys that matched.
#include "vuke_host.hpp"

#include <chrono>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cerrno>
#include <exception>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <thread>

namespace vuke {

// The text encodings of a key (base58check x4, bech32, decimal, binary) cost more host time than the
// device spends deriving it, so generate mode formats a chunk on every host core (the reference does
// the same work under rayon, src/source/range.rs:44-70) and then writes the chunk in input order.
template <typename F>
static void parallel_for(size_t n, F fn)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 64) nt = 64;
    if (n < 4096 || nt == 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        const size_t a = (size_t)t * per, b = std::min(n, a + per);
        if (a >= b) break;
        th.emplace_back([=]() { fn(a, b); });
    }
    for (auto &x : th) x.join();
}

// shared by the string-capable paths: keys (with labels) -> derive on the device -> match / emit
static void emit_keys(vk_engine *engine, const Transform &t, const std::vector<std::pair<std::string, Key>> &batch, const Matcher *matcher,
                      Output &output, ProcessStats &st)
{
    if (batch.empty()) return;
    st.keys_generated += batch.size();
    std::vector<Key> keys;
    for (auto &lk : batch) keys.push_back(lk.second);
    KeyDeriver deriver(engine);
    if (matcher) {
        std::vector<uint8_t> flat(keys.size() * 32);
        for (size_t i = 0; i < keys.size(); i++) memcpy(&flat[32 * i], keys[i].data(), 32);
        std::vector<vk_hit> hits(keys.size());
        vk_sweep_stats ss;
        check_rc(engine, vk_scan_keys(engine, flat.data(), keys.size(), VK_F_COMPRESSED | VK_F_UNCOMPRESSED, hits.data(), hits.size(), &ss));
        hits.resize((size_t)ss.matches);
        std::sort(hits.begin(), hits.end(), [](const vk_hit &a, const vk_hit &b) { return a.seed < b.seed; });
        st.matches_found += ss.matches;
        std::vector<Key> hk;
        for (auto &h : hits) hk.push_back(keys[(size_t)h.seed]);
        std::vector<DerivedKey> derived = deriver.derive(hk);
        for (size_t i = 0; i < hits.size(); i++) {
            MatchInfo mi;
            mi.address_type = hits[i].addr_kind == VK_ADDR_P2PKH_C   ? AddressType::P2pkhCompressed
                              : hits[i].addr_kind == VK_ADDR_P2PKH_U ? AddressType::P2pkhUncompressed
                                                                     : AddressType::P2wpkh;
            mi.address = *derived[i].addresses()[hits[i].addr_kind];
            output.hit(batch[(size_t)hits[i].seed].first, t.name(), derived[i], mi);
        }
    } else {
        std::vector<DerivedKey> derived = deriver.derive(keys);
        for (size_t i = 0; i < keys.size(); i++) output.key(batch[i].first, t.name(), derived[i]);
    }
}

// One pass over a passphrase file (or stdin): lines are split and trimmed straight into the (data, offsets) form the
// C ABI takes -- no std::string per line; labels are only built for the keys that are printed.
namespace {
struct LineBatch {
    std::vector<uint8_t> data;
    std::vector<uint32_t> offs{0};
    size_t n() const { return offs.size() - 1; }
    std::string str(size_t i) const { return std::string(data.begin() + offs[i], data.begin() + offs[i + 1]); }
    uint32_t max_len() const
    {
        uint32_t m = 0;
        for (size_t i = 0; i + 1 < offs.size(); i++) m = std::max(m, offs[i + 1] - offs[i]);
        return m;
    }
    void clear()
    {
        data.clear();
        offs.assign(1, 0);
    }
};

class LineReader { // src/source/wordlist.rs:24-30, src/source/stdin.rs:45-50: trim, skip empty lines
public:
    // strict: the reference's StdinSource (`let line = line?;`, src/source/stdin.rs:45-46) stops at the first line that is not
    // UTF-8 and returns the error; lines are handed to the transforms in batches of 1000 there, so the (fewer than 1000) lines
    // read since the last full batch are never processed.  The wordlist source drops such a line and goes on
    // (`filter_map(Result::ok)`, src/source/wordlist.rs:33-36).
    //
    // Strict mode therefore reads a whole group of 1000 lines (the reference's `batch`), checks it, and only then hands its lines
    // out -- in as many device batches as the caller's limits ask for: a batch never exceeds max_lines (the device bound on
    // lines x slots) nor the 4 GiB the C ABI's uint32 offsets can address, whatever the lines' lengths, and a line that is not
    // UTF-8 still loses exactly the lines of its own group (until round 4 a batch grew to the next multiple of 1000).
    explicit LineReader(FILE *fp, bool strict = false) : fp_(fp), strict_(strict), buf_(8u << 20) {}
    bool failed() const { return failed_; }
    const std::string &error() const { return err_; }
    // appends lines to `b` until it holds max_lines or max_bytes (one line may exceed max_bytes on its own); false at end of input
    bool fill(LineBatch &b, size_t max_lines, size_t max_bytes)
    {
        for (;;) {
            // ---- lines that are checked and waiting
            while (ready_next_ < ready_.n()) {
                const size_t len = ready_.offs[ready_next_ + 1] - ready_.offs[ready_next_];
                const int a = admit(b, len, max_lines, max_bytes);
                if (a < 0) return false;
                if (a == 0) return true;
                b.data.insert(b.data.end(), ready_.data.begin() + (std::ptrdiff_t)ready_.offs[ready_next_], ready_.data.begin() + (std::ptrdiff_t)ready_.offs[ready_next_ + 1]);
                b.offs.push_back((uint32_t)b.data.size());
                ready_next_++;
            }
            if (failed_ || eof_) return false;
            if (b.n() >= max_lines || b.data.size() >= max_bytes) return true;
            // ---- the next line of the input
            if (pos_ == len_) {
                len_ = fread(buf_.data(), 1, buf_.size(), fp_);
                pos_ = 0;
                if (len_ == 0) {
                    if (!line_.empty()) take(b, line_.data(), line_.data() + line_.size(), max_lines, max_bytes);
                    line_.clear();
                    eof_ = true;
                    if (strict_ && !failed_) publish(); // `if !batch.is_empty()` after the loop: the last, shorter group
                    continue;
                }
            }
            const char *p = buf_.data() + pos_, *e = buf_.data() + len_;
            const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
            if (!nl) {
                line_.append(p, e);
                pos_ = len_;
                continue;
            }
            pos_ = (size_t)(nl - buf_.data()) + 1;
            if (line_.empty()) {
                take(b, p, nl, max_lines, max_bytes);
            } else {
                line_.append(p, nl);
                take(b, line_.data(), line_.data() + line_.size(), max_lines, max_bytes);
                line_.clear();
            }
        }
    }

private:
    // lines held back: strict mode's group being read (`group_`) and the checked lines not yet handed out (`ready_`: a complete
    // group, or in the other mode the one line that did not fit the batch it arrived in)
    struct Lines {
        std::vector<uint8_t> data;
        std::vector<size_t> offs{0};
        size_t n() const { return offs.size() - 1; }
        void clear()
        {
            data.clear();
            offs.assign(1, 0);
        }
        void add(const char *a, const char *e)
        {
            data.insert(data.end(), a, e);
            offs.push_back(data.size());
        }
    };
    static constexpr size_t kHardBytes = 0xFFFFFF00u; // one call's strings are addressed by uint32 offsets (include/vuke_hip.h)
    // 1: the line goes into this batch; 0: the batch is full, the line waits; -1: no batch can ever take it
    int admit(const LineBatch &b, size_t len, size_t max_lines, size_t max_bytes)
    {
        if (len > kHardBytes) {
            failed_ = true;
            err_ = "an input line of more than 4 GiB cannot be handed to the device";
            return -1;
        }
        if (b.n() >= max_lines) return 0;
        if (b.n() && b.data.size() + len > std::min(max_bytes, kHardBytes)) return 0;
        return 1;
    }
    void publish()
    {
        std::swap(ready_, group_);
        group_.clear();
        ready_next_ = 0;
    }
    void take(LineBatch &b, const char *a, const char *e, size_t max_lines, size_t max_bytes)
    {
        if (!utf8_valid(a, (size_t)(e - a))) {
            if (!strict_) return; // .lines().filter_map(Result::ok): a line that is not UTF-8 is dropped
            failed_ = true;       // `line?`: the lines since the last batch of 1000 are lost with it
            err_ = "stream did not contain valid UTF-8"; // io::Error of BufRead::lines(), src/source/stdin.rs:46
            group_.clear();
            return;
        }
        trim_unicode(a, e);                           // str::trim(): Unicode White_Space, not only ASCII
        if (a == e) return;
        if (strict_) {
            group_.add(a, e);
            if (group_.n() == 1000) publish();
            return;
        }
        if (admit(b, (size_t)(e - a), max_lines, max_bytes) == 1) {
            b.data.insert(b.data.end(), a, e);
            b.offs.push_back((uint32_t)b.data.size());
        } else if (!failed_) { // rare: the line would carry the batch past what uint32 offsets address -- it opens the next one
            ready_.clear();
            ready_.add(a, e);
            ready_next_ = 0;
        }
    }
    FILE *fp_;
    bool strict_, failed_ = false, eof_ = false;
    std::string err_;
    Lines group_, ready_;
    size_t ready_next_ = 0;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
    std::string line_; // a line that straddles two reads
};
} // namespace

// test access (capi.cpp: vkh_line_batches): the batches the wordlist / stdin source would hand to the device for this input,
// written one line per line with "--" between batches; returns the reader's error text (empty when it ended normally)
std::string line_reader_batches(FILE *in, bool strict, size_t max_lines, size_t max_bytes, FILE *out)
{
    LineReader reader(in, strict);
    LineBatch b;
    bool more = true;
    while (more) {
        b.clear();
        more = reader.fill(b, max_lines, max_bytes);
        if (b.n() == 0) continue;
        for (size_t i = 0; i < b.n(); i++) {
            fwrite(b.data.data() + b.offs[i], 1, b.offs[i + 1] - b.offs[i], out);
            fputc('\n', out);
        }
        fputs("--\n", out);
    }
    return reader.failed() ? reader.error() : std::string();
}

ProcessStats HipWordlistSource::process(const std::vector<Transform> &transforms, const Matcher *matcher, Output &output)
{
    FILE *fp = path == "-" ? stdin : fopen(path.c_str(), "rb");
    if (!fp) throw std::runtime_error("cannot open wordlist: " + path);
    struct Closer { FILE *f; bool own; ~Closer() { if (own) fclose(f); } } closer{fp, path != "-"};
    ProcessStats st;
    if (matcher) {
        std::vector<uint8_t> h160, kind;
        matcher->device_targets(h160, kind);
        for (vk_engine *e : engines_) check_rc(e, vk_targets_set(e, h160.data(), kind.data(), kind.size()));
    }
    const uint32_t flags = VK_F_COMPRESSED | VK_F_UNCOMPRESSED;
    const bool full = output.needs_full_record(), want_pub = output.needs_pubkeys();
    LineReader reader(fp, path == "-");
    LineBatch b;
    // what the device side of one slice of a batch produces: the emitted keys in emit order with the slot each came from,
    // then either the hits (scan) or the derived values of every key (generate)
    struct SliceOut {
        std::vector<uint8_t> flat, norm, hc, hu, xy;
        std::vector<uint32_t> slot_of; // generate: slot of every emitted key; scan: slot of every hit
        size_t nkeys = 0;              // keys the transform emitted for this slice
        std::string warning;           // over-long inputs the device skipped (vk_last_warning after VK_OK)
        std::vector<vk_hit> hits;
        std::vector<DerivedKey> hit_derived;
        std::vector<uint8_t> text;     // generate, compact console format: the lines as the device wrote them
        bool as_text = false;
    };
    // one vk_transform_strings / vk_scan_strings call carries at most 2^28 keys: size the batch by keys, not by lines
    uint32_t max_slots = 1;
    for (const Transform &t : transforms) max_slots = std::max(max_slots, t.string_slots());
    size_t max_lines = std::max<size_t>(1, std::min<size_t>(1u << 20, (1u << 27) / max_slots));
    // bytes per device batch (the CLI's knob, tests/test_golden.py cuts batches short with it; the library reads no environment)
    size_t batch_bytes = 256u << 20;
    if (const char *v = getenv("VUKE_HIP_BATCH_BYTES")) {
        const long long x = atoll(v);
        if (x >= 1024 && x <= (1ll << 31)) batch_bytes = (size_t)x;
    }
    bool more = true;
    while (more) {
        b.clear();
        more = reader.fill(b, max_lines, batch_bytes); // a million passphrases per batch, fewer when one line makes many keys
        const size_t n = b.n();
        if (n == 0) continue;
        st.inputs_processed += n;
        for (const Transform &t : transforms) {
            const uint32_t k = t.string_slots();
            if (k == 0) continue; // a transform that needs Input::u64_val pushes nothing for a passphrase
            // generate mode into a sink that takes text, from a transform whose label is the input plus a per-slot suffix
            // or into one that stores the full record (columns from vk_derive_columns, written slice by slice: one engine only)
            const size_t g = n < 4096 ? 1 : std::min(engines_.size(), n);
            const unsigned fmt = (matcher || getenv("VUKE_HIP_RECORD_PATH")) ? 0u : output.device_formats(); // (the variable: A/B knob of tools/probe_*)
            auto device_ok = [&](const Transform &tr) {
                if (fmt == 0 || (fmt & Output::DEVICE_VERBOSE)) return false;
                if ((fmt & Output::DEVICE_COLUMNS) && g != 1) return false;
                if (fmt & Output::DEVICE_TEXT) return strlen(tr.name()) < 32 && tr.string_slots() <= (1u << 16);
                return true;
            };
            auto run_slice = [&](vk_engine *eng, size_t first, size_t cnt, SliceOut &o) {
                std::vector<uint32_t> offs(cnt + 1);
                for (size_t i = 0; i <= cnt; i++) offs[i] = b.offs[first + i] - b.offs[first];
                if (matcher) {
                    // scan: strings in, hits out -- the keys stay on the device (vk_scan_strings)
                    o.hits.resize(1u << 16);
                    vk_sweep_stats ss;
                    check_rc(eng, vk_scan_strings(eng, &t.spec(), b.data.data() + b.offs[first], offs.data(), cnt, flags, o.hits.data(), o.hits.size(), &ss));
                    const std::string warned = last_warning(eng);
                    while (ss.hits_dropped) { // more hits than the buffer holds: only the probe sweep runs again, the keys are still on the device
                        o.hits.resize(o.hits.size() + (size_t)ss.hits_dropped + 1024);
                        check_rc(eng, vk_rescan_strings(eng, flags, o.hits.data(), o.hits.size(), &ss));
                    }
                    o.warning = warned;
                    o.nkeys = (size_t)ss.keys;
                    o.hits.resize((size_t)ss.matches);
                    std::sort(o.hits.begin(), o.hits.end(), [](const vk_hit &x, const vk_hit &y) {
                        return x.seed != y.seed ? x.seed < y.seed : x.sub_index < y.sub_index;
                    });
                    std::vector<Key> hk(o.hits.size());
                    for (size_t i = 0; i < o.hits.size(); i++) {
                        memcpy(hk[i].data(), o.hits[i].privkey, 32);
                        o.slot_of.push_back((uint32_t)((first + (size_t)o.hits[i].seed) * k + o.hits[i].sub_index));
                    }
                    if (!hk.empty()) o.hit_derived = KeyDeriver(eng).derive(hk);
                    return;
                }
                if (device_ok(t)) {
                    // compact console output: the lines come back as text (vk_generate_lines_strings: the label is the input
                    // string plus the slot's suffix), 2^21 (input, slot) pairs per call
                    std::vector<std::string> suffix(k);
                    std::vector<const char *> sp(k);
                    for (uint32_t s = 0; s < k; s++) {
                        suffix[s] = t.label(std::string(), s);
                        sp[s] = suffix[s].c_str();
                    }
                    o.as_text = true;
                    if (fmt & Output::DEVICE_COLUMNS) {
                        // the record's columns: keys from vk_transform_strings, every other value column from vk_derive_columns, the
                        // source column (the labels) made here in the same encoding; 2^20 keys per call
                        std::vector<uint8_t> keys(cnt * k * 32), valid(cnt * k);
                        check_rc(eng, vk_transform_strings(eng, &t.spec(), b.data.data() + b.offs[first], offs.data(), cnt, keys.data(), valid.data()));
                        o.warning = last_warning(eng);
                        std::vector<uint8_t> colbuf[VK_RC_COLUMNS], flat, src;
                        std::vector<size_t> slot_of;
                        size_t s0 = 0;
                        while (s0 < cnt * k) {
                            flat.clear();
                            slot_of.clear();
                            size_t s1 = s0;
                            for (; s1 < cnt * k && slot_of.size() < (1u << 20); s1++)
                                if (valid[s1]) {
                                    flat.insert(flat.end(), keys.begin() + 32 * s1, keys.begin() + 32 * s1 + 32);
                                    slot_of.push_back(s1);
                                }
                            s0 = s1;
                            const size_t m = slot_of.size();
                            if (m == 0) continue;
                            vk_record_columns rc;
                            for (int c = 0; c < VK_RC_COLUMNS; c++) {
                                const size_t need = m * vk_record_column_stride(c);
                                if (colbuf[c].size() < need) colbuf[c].resize(need);
                                rc.data[c] = colbuf[c].data();
                                rc.cap[c] = colbuf[c].size();
                            }
                            check_rc(eng, vk_derive_columns(eng, flat.data(), m, &rc));
                            src.clear();
                            for (size_t j = 0; j < m; j++) {
                                const std::string lab = t.label(b.str(first + slot_of[j] / k), (uint32_t)(slot_of[j] % k));
                                const uint32_t l = (uint32_t)lab.size();
                                for (int i = 0; i < 4; i++) src.push_back((uint8_t)(l >> (8 * i)));
                                src.insert(src.end(), lab.begin(), lab.end());
                            }
                            output.write_columns(rc, t.name(), src.data(), src.size());
                            o.nkeys += m;
                        }
                        if (!(fmt & Output::DEVICE_TEXT)) return;
                        o.nkeys = 0; // counted again below
                    }
                    // electrum / multibit print a shortened form of the input (src/electrum.rs:199-211, src/multibit.rs:352-360): made
                    // here once per input and handed over as the label; every other transform prints the input itself
                    const bool own_labels = t.spec().kind == VK_T_ELECTRUM || t.spec().kind == VK_T_MULTIBIT;
                    const size_t per = std::max<size_t>(1, (1u << 21) / k);
                    std::vector<uint32_t> sub, lab_offs;
                    std::vector<uint8_t> lab;
                    lab.reserve(64);
                    for (size_t a = 0; a < cnt; a += per) {
                        const size_t c = std::min(per, cnt - a);
                        sub.resize(c + 1);
                        for (size_t i = 0; i <= c; i++) sub[i] = offs[a + i] - offs[a];
                        if (own_labels) {
                            lab.clear();
                            lab_offs.assign(1, 0);
                            for (size_t i = 0; i < c; i++) {
                                const std::string l0 = t.label(b.str(first + a + i), 0);
                                lab.insert(lab.end(), l0.begin(), l0.end() - (std::ptrdiff_t)suffix[0].size());
                                lab_offs.push_back((uint32_t)lab.size());
                            }
                        }
                        const size_t at = o.text.size(), cap = (size_t)(own_labels ? lab_offs[c] : sub[c]) * k + c * k * 192;
                        o.text.resize(at + cap);
                        size_t len = 0;
                        vk_sweep_stats ss;
                        check_rc(eng, vk_generate_lines_strings(eng, &t.spec(), b.data.data() + b.offs[first] + offs[a], sub.data(), c, own_labels ? lab.data() : nullptr,
                                                                own_labels ? lab_offs.data() : nullptr, t.name(), sp.data(),
                                                                o.text.data() + at, cap, &len, &ss));
                        const std::string w = last_warning(eng);
                        if (!w.empty()) o.warning = w;
                        o.text.resize(at + len);
                        o.nkeys += (size_t)ss.keys;
                    }
                    return;
                }
                std::vector<uint8_t> keys(cnt * k * 32), valid(cnt * k);
                check_rc(eng, vk_transform_strings(eng, &t.spec(), b.data.data() + b.offs[first], offs.data(), cnt, keys.data(), valid.data()));
                o.warning = last_warning(eng);
                for (size_t s = 0; s < cnt * k; s++)
                    if (valid[s]) {
                        o.flat.insert(o.flat.end(), keys.begin() + 32 * s, keys.begin() + 32 * s + 32);
                        o.slot_of.push_back((uint32_t)(first * k + s));
                    }
                const size_t m = o.slot_of.size();
                o.nkeys = m;
                if (m == 0) return;
                {
                    o.norm.resize(m * 32);
                    o.hc.resize(m * 20);
                    o.hu.resize(m * 20);
                    o.xy.resize(want_pub ? m * 64 : 0);
                    check_rc(eng, vk_derive_batch_pub(eng, o.flat.data(), m, o.norm.data(), o.hc.data(), o.hu.data(), want_pub ? o.xy.data() : nullptr));
                }
            };
            // one contiguous slice of the batch per engine (independent inputs, no exchange); small batches stay on one
            std::vector<SliceOut> outs(g);
            if (g == 1) {
                run_slice(engines_[0], 0, n, outs[0]);
            } else {
                std::vector<std::exception_ptr> err(g);
                std::vector<std::thread> th;
                for (size_t i = 0; i < g; i++)
                    th.emplace_back([&, i]() {
                        try {
                            const size_t first = n * i / g, last = n * (i + 1) / g;
                            run_slice(engines_[i], first, last - first, outs[i]);
                        } catch (...) {
                            err[i] = std::current_exception();
                        }
                    });
                for (auto &x : th) x.join();
                for (auto &x : err)
                    if (x) std::rethrow_exception(x);
            }
            // emission, in input order
            for (SliceOut &o : outs) {
                const size_t m = o.slot_of.size();
                st.keys_generated += o.nkeys;
                if (o.warning.rfind("warning:", 0) == 0) fprintf(stderr, "%s (%s)\n", o.warning.c_str(), t.name());
                if (o.as_text) {
                    if (!o.text.empty()) output.write_text((const char *)o.text.data(), o.text.size());
                    continue;
                }
                if (m == 0) continue;
                auto label_of = [&](size_t j) { return t.label(b.str(o.slot_of[j] / k), o.slot_of[j] % k); };
                if (matcher) {
                    st.matches_found += o.hits.size();
                    for (size_t i = 0; i < o.hits.size(); i++) {
                        MatchInfo mi;
                        mi.address_type = o.hits[i].addr_kind == VK_ADDR_P2PKH_C   ? AddressType::P2pkhCompressed
                                          : o.hits[i].addr_kind == VK_ADDR_P2PKH_U ? AddressType::P2pkhUncompressed
                                                                                   : AddressType::P2wpkh;
                        mi.address = *o.hit_derived[i].addresses()[o.hits[i].addr_kind];
                        output.hit(label_of(i), t.name(), o.hit_derived[i], mi);
                    }
                } else {
                    for (size_t lo = 0; lo < m; lo += 1u << 18) { // bounded text buffers: ~1 KB per key
                        const size_t cnt = std::min<size_t>(1u << 18, m - lo);
                        std::vector<DerivedKey> derived(cnt);
                        std::vector<std::string> labels(cnt);
                        parallel_for(cnt, [&](size_t a0, size_t a1) {
                            for (size_t i = a0; i < a1; i++) {
                                derived[i] = full ? KeyDeriver::from_device(&o.norm[32 * (lo + i)], &o.hc[20 * (lo + i)], &o.hu[20 * (lo + i)])
                                                  : KeyDeriver::from_device_compact(&o.norm[32 * (lo + i)], &o.hc[20 * (lo + i)], &o.hu[20 * (lo + i)]);
                                if (want_pub) KeyDeriver::set_pubkeys(derived[i], &o.xy[64 * (lo + i)]);
                                labels[i] = label_of(lo + i);
                            }
                        });
                        for (size_t i = 0; i < cnt; i++) output.key(labels[i], t.name(), derived[i]);
                    }
                }
            }
        }
    }
    if (matcher)
        for (vk_engine *e : engines_) check_rc(e, vk_targets_set(e, nullptr, nullptr, 0));
    if (reader.failed()) throw std::runtime_error(reader.error()); // io::Error of BufRead::lines(), src/source/stdin.rs:46 (or a line no batch can carry)
    return st;
}

ProcessStats HipMultiRangeSource::process(const std::vector<Transform> &transforms, const Matcher *matcher, Output &output)
{
    const size_t g = engines_.size();
    if (g == 0) throw std::invalid_argument("no engines");
    // block i = [start + i*per, ...): ceil(N / G) inputs each, the last one shorter (N = 2^64 cannot occur: end - start <= 2^64 - 1)
    const uint64_t span = end - start; // inputs - 1
    const uint64_t per = span / g + 1;
    std::vector<ProcessStats> part(g);
    std::vector<std::exception_ptr> err(g);
    std::vector<std::thread> th;
    for (size_t i = 0; i < g; i++) {
        const uint64_t off = per * i;
        if (off > span) break; // fewer inputs than engines
        const uint64_t lo = start + off, hi = (span - off < per - 1) ? end : lo + per - 1;
        th.emplace_back([&, i, lo, hi]() {
            try {
                part[i] = HipRangeSource(engines_[i], lo, hi).process(transforms, matcher, output);
            } catch (...) {
                err[i] = std::current_exception();
            }
        });
    }
    for (auto &t : th) t.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
    ProcessStats st;
    for (auto &p : part) {
        st.inputs_processed += p.inputs_processed;
        st.keys_generated += p.keys_generated;
        st.matches_found += p.matches_found;
    }
    return st;
}

// days since 1970-01-01 of a proleptic Gregorian date (chrono NaiveDate::parse_from_str "%Y-%m-%d")
static int64_t days_from_civil(int64_t y, unsigned m, unsigned d)
{
    y -= m <= 2;
    const int64_t era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    return era * 146097 + (int64_t)doe - 719468;
}

static uint64_t parse_date(const std::string &s, bool end_of_day)
{
    int y = 0;
    unsigned m = 0, d = 0;
    char tail = 0;
    if (sscanf(s.c_str(), "%d-%u-%u%c", &y, &m, &d, &tail) != 3 || m < 1 || m > 12 || d < 1) throw std::invalid_argument("invalid date '" + s + "' (expected YYYY-MM-DD)");
    static const unsigned DIM[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    const bool leap = (y % 4 == 0 && y % 100 != 0) || y % 400 == 0;
    if (d > DIM[m - 1] + (m == 2 && leap ? 1u : 0u)) throw std::invalid_argument("invalid date '" + s + "' (expected YYYY-MM-DD)");
    const int64_t t = days_from_civil(y, m, d) * 86400 + (end_of_day ? 86399 : 0);
    return (uint64_t)t; // `as u64`, src/source/timestamps.rs:28,35
}

HipTimestampSource::HipTimestampSource(vk_engine *engine, const std::string &start_date, const std::string &end_date, bool us)
    : start(parse_date(start_date, false)), end(parse_date(end_date, true)), microseconds(us), engine_(engine)
{
    if (end < start) throw std::invalid_argument("timestamps: --end is before --start");
}

ProcessStats HipTimestampSource::process(const std::vector<Transform> &transforms, const Matcher *matcher, Output &output)
{
    // src/source/timestamps.rs:49-90: every second ts of the range, and with `microseconds` also ts*1000 + 0..999.
    // Both sets are contiguous, so they are two range sweeps.
    ProcessStats st = HipRangeSource(engine_, start, end).process(transforms, matcher, output);
    const uint64_t count = end - start + 1;
    st.inputs_processed = count; // :50-55
    if (microseconds) {
        if (end > (~0ull - 999) / 1000) throw std::invalid_argument("timestamps: range too large for --microseconds");
        ProcessStats ms = HipRangeSource(engine_, start * 1000, end * 1000 + 999).process(transforms, matcher, output);
        st.keys_generated += ms.keys_generated;
        st.matches_found += ms.matches_found;
        st.inputs_processed = count * 1000; // the reference reports count * 1000 although it walks 1001 values per second
    }
    return st;
}

ProcessStats HipRangeSource::process(const std::vector<Transform> &transforms, const Matcher *matcher, Output &output)
{
    ProcessStats st;
    st.inputs_processed = end - start + 1; // src/source/range.rs:32
    const uint32_t flags = VK_F_COMPRESSED | VK_F_UNCOMPRESSED;
    const bool full = output.needs_full_record(), want_pub = output.needs_pubkeys();
    if (matcher) {
        std::vector<uint8_t> h160, kind;
        matcher->device_targets(h160, kind);
        check_rc(engine_, vk_targets_set(engine_, h160.data(), kind.data(), kind.size()));
        // the reference walks the range chunk by chunk and, inside a chunk, transform by transform;
        // hits are rare, so here each transform sweeps a large block of the range in one launch
        const uint64_t block = 1ull << 30;
        for (uint64_t lo = start;; lo += block) {
            const uint64_t hi = (end - lo < block - 1) ? end : lo + block - 1;
            for (const Transform &t : transforms) {
                if (t.is_bitimage()) continue; // nothing for an input without a blob
                if (t.takes_strings()) { // e.g. electrum over a numeric range: the decimal string is the seed
                    for (uint64_t a = lo;; a += 16384) {
                        const uint64_t b = (hi - a < 16383) ? hi : a + 16383;
                        std::vector<uint64_t> inputs;
                        for (uint64_t v = a;; v++) { inputs.push_back(v); if (v == b) break; }
                        std::vector<std::pair<std::string, Key>> batch;
                        t.apply_batch(engine_, inputs, batch);
                        emit_keys(engine_, t, batch, matcher, output, st);
                        if (b == hi) break;
                    }
                    continue;
                }
                std::vector<vk_hit> hits(1 << 16);
                vk_sweep_stats ss;
                for (;;) {
                    check_rc(engine_, vk_sweep(engine_, &t.spec(), lo, hi, flags, hits.data(), hits.size(), &ss));
                    if (!ss.hits_dropped) break;
                    hits.resize(hits.size() + (size_t)ss.hits_dropped + 1024); // rerun with room for every hit
                }
                st.keys_generated += ss.keys;
                st.matches_found += ss.matches;
                hits.resize((size_t)ss.matches);
                std::sort(hits.begin(), hits.end(), [](const vk_hit &a, const vk_hit &b) {
                    return a.seed != b.seed ? a.seed < b.seed : a.sub_index < b.sub_index;
                });
                std::vector<Key> keys(hits.size());
                for (size_t i = 0; i < hits.size(); i++) memcpy(keys[i].data(), hits[i].privkey, 32);
                KeyDeriver deriver(engine_);
                std::vector<DerivedKey> derived = deriver.derive(keys);
                for (size_t i = 0; i < hits.size(); i++) {
                    const DerivedKey &d = derived[i];
                    MatchInfo mi;
                    mi.address_type = hits[i].addr_kind == VK_ADDR_P2PKH_C   ? AddressType::P2pkhCompressed
                                      : hits[i].addr_kind == VK_ADDR_P2PKH_U ? AddressType::P2pkhUncompressed
                                                                             : AddressType::P2wpkh;
                    mi.address = *d.addresses()[hits[i].addr_kind];
                    output.hit(t.label(hits[i].seed, hits[i].sub_index), t.name(), d, mi);
                }
            }
            if (hi == end) break;
        }
        check_rc(engine_, vk_targets_set(engine_, nullptr, nullptr, 0));
    } else {
        // inputs per vk_generate call, sized by KEYS (host: 72 B raw + ~1 KB of text per key): at most ~2^18 keys per call
        // whatever the slot count (sha256_chain with a large --chain-depth has thousands of slots per input)
        uint32_t max_slots = 1;
        for (const Transform &t : transforms) max_slots = std::max(max_slots, t.takes_strings() ? t.string_slots() : t.slots());
        // Compact console output of numeric transforms: the lines come back from the device as text (vk_generate_lines: decimal
        // label, hex, Base58Check and line assembly in HIP), 2^21 keys per call; anything else takes the record path below.
        // A sink that stores the full record (StorageOutput) takes the record's value columns the same way (vk_generate_columns, 2^20
        // keys per call); a MultiOutput of such sinks gets both.
        const unsigned fmt = getenv("VUKE_HIP_RECORD_PATH") ? 0u : output.device_formats(); // (the variable: A/B knob of tools/probe_*)
        bool text_path = fmt != 0 && max_slots <= (1u << 16);
        for (const Transform &t : transforms) text_path = text_path && !t.takes_strings() && strlen(t.name()) < 32;
        const bool want_text = (fmt & Output::DEVICE_TEXT) != 0, want_cols = (fmt & Output::DEVICE_COLUMNS) != 0, want_verbose = (fmt & Output::DEVICE_VERBOSE) != 0;
        const uint64_t chunk = text_path ? std::max<uint64_t>(1, (want_cols || want_verbose ? 1ull << 20 : 1ull << 21) / max_slots)
                                         : std::max<uint64_t>(1, std::min<uint64_t>(1ull << 16, (1ull << 18) / max_slots));
        std::vector<uint8_t> text;
        std::vector<uint8_t> colbuf[VK_RC_COLUMNS];
        double t_dev = 0, t_sink = 0; // VUKE_HIP_TIMING: seconds in the device calls / in the sinks
        auto now = []() { return std::chrono::steady_clock::now(); };
        auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
        for (uint64_t lo = start;; lo += chunk) {
            const uint64_t hi = (end - lo < chunk - 1) ? end : lo + chunk - 1;
            for (const Transform &t : transforms) {
                if (text_path) {
                    const uint32_t k = t.slots();
                    if (k == 0) continue;
                    std::vector<std::string> suffix(k);
                    std::vector<const char *> sp(k);
                    for (uint32_t s = 0; s < k; s++) {
                        suffix[s] = t.label(0, s).substr(1); // what follows the decimal input in this slot's label
                        sp[s] = suffix[s].c_str();
                    }
                    const size_t n = (size_t)(hi - lo + 1) * k;
                    vk_sweep_stats ss;
                    if (want_text) {
                        if (text.size() < n * 192) text.resize(n * 192);
                        size_t len = 0;
                        auto t0 = now();
                        check_rc(engine_, vk_generate_lines(engine_, &t.spec(), lo, hi, t.name(), sp.data(), text.data(), text.size(), &len, &ss));
                        t_dev += since(t0);
                        t0 = now();
                        output.write_text((const char *)text.data(), len);
                        t_sink += since(t0);
                    }
                    if (want_verbose) {
                        if (text.size() < n * 512) text.resize(n * 512);
                        size_t len = 0;
                        auto t0 = now();
                        check_rc(engine_, vk_generate_verbose(engine_, &t.spec(), lo, hi, t.name(), sp.data(), text.data(), text.size(), &len, &ss));
                        t_dev += since(t0);
                        t0 = now();
                        output.write_verbose((const char *)text.data(), len);
                        t_sink += since(t0);
                    }
                    if (want_cols) {
                        vk_record_columns rc;
                        for (int c = 0; c < VK_RC_COLUMNS; c++) {
                            const size_t need = n * vk_record_column_stride(c);
                            if (colbuf[c].size() < need) colbuf[c].resize(need);
                            rc.data[c] = colbuf[c].data();
                            rc.cap[c] = colbuf[c].size();
                        }
                        auto t0 = now();
                        check_rc(engine_, vk_generate_columns(engine_, &t.spec(), lo, hi, sp.data(), &rc, &ss));
                        t_dev += since(t0);
                        t0 = now();
                        output.write_columns(rc, t.name());
                        t_sink += since(t0);
                    }
                    st.keys_generated += ss.keys;
                    continue;
                }
                if (t.takes_strings()) {
                    std::vector<uint64_t> inputs;
                    for (uint64_t v = lo;; v++) { inputs.push_back(v); if (v == hi) break; }
                    std::vector<std::pair<std::string, Key>> batch;
                    t.apply_batch(engine_, inputs, batch);
                    emit_keys(engine_, t, batch, nullptr, output, st);
                    continue;
                }
                const uint32_t k = t.slots();
                if (k == 0) continue; // bitimage: nothing for an input without a blob
                const size_t n = (size_t)(hi - lo + 1) * k;
                std::vector<uint8_t> priv(n * 32), hc(n * 20), hu(n * 20), valid(n);
                vk_sweep_stats ss;
                check_rc(engine_, vk_generate(engine_, &t.spec(), lo, hi, flags, priv.data(), hc.data(), hu.data(), valid.data(), &ss));
                st.keys_generated += ss.keys;
                std::vector<uint8_t> xy;
                if (want_pub) { // the record carries the public keys: a second pass over the (already normalised) keys fetches X || Y
                    xy.resize(n * 64);
                    check_rc(engine_, vk_derive_batch_pub(engine_, priv.data(), n, nullptr, nullptr, nullptr, xy.data()));
                }
                std::vector<DerivedKey> derived(n);
                parallel_for(n, [&](size_t a, size_t b) {
                    for (size_t i = a; i < b; i++)
                        if (valid[i]) {
                            derived[i] = full ? KeyDeriver::from_device(&priv[32 * i], &hc[20 * i], &hu[20 * i])
                                              : KeyDeriver::from_device_compact(&priv[32 * i], &hc[20 * i], &hu[20 * i]);
                            if (want_pub) KeyDeriver::set_pubkeys(derived[i], &xy[64 * i]);
                        }
                });
                for (size_t i = 0; i < n; i++)
                    if (valid[i]) output.key(t.label(lo + i / k, (uint32_t)(i % k)), t.name(), derived[i]);
            }
            if (hi == end) break;
        }
        if (text_path && getenv("VUKE_HIP_TIMING")) fprintf(stderr, "[timing] generate: %.3f s in device calls, %.3f s in the sinks\n", t_dev, t_sink);
    }
    return st;
}

// ------------------------------------------------------------------------------------------------ FilesSource
HipFilesSource HipFilesSource::from_file(vk_engine *engine, const std::string &path)
{
    std::error_code ec;
    const std::filesystem::path p(path);
    if (!std::filesystem::exists(p, ec)) throw std::runtime_error("File not found: " + path);
    if (!std::filesystem::is_regular_file(p, ec)) throw std::runtime_error("Not a file: " + path);
    HipFilesSource s(engine);
    s.files.push_back(path);
    return s;
}

static void collect_files_inner(const std::filesystem::path &dir, std::vector<std::string> &files)
{
    for (const std::filesystem::directory_entry &entry : std::filesystem::directory_iterator(dir)) {
        if (entry.is_symlink()) continue; // skip symlinks: loops (src/source/files.rs:69-72)
        if (entry.is_regular_file()) files.push_back(entry.path().string());
        else if (entry.is_directory()) collect_files_inner(entry.path(), files);
    }
}

HipFilesSource HipFilesSource::from_dir(vk_engine *engine, const std::string &path)
{
    std::error_code ec;
    const std::filesystem::path p(path);
    if (!std::filesystem::exists(p, ec)) throw std::runtime_error("Directory not found: " + path);
    if (!std::filesystem::is_directory(p, ec)) throw std::runtime_error("Not a directory: " + path);
    HipFilesSource s(engine);
    collect_files_inner(p, s.files);
    if (s.files.empty()) throw std::runtime_error("No files found in directory: " + path);
    std::sort(s.files.begin(), s.files.end()); // read_dir order is unspecified (and the reference runs files in parallel): a fixed order here
    return s;
}

ProcessStats HipFilesSource::process(const std::vector<Transform> &transforms, const Matcher *matcher, Output &output)
{
    ProcessStats st;
    if (matcher) {
        std::vector<uint8_t> h160, kind;
        matcher->device_targets(h160, kind);
        check_rc(engine_, vk_targets_set(engine_, h160.data(), kind.data(), kind.size()));
    }
    bool any_bitimage = false;
    for (const Transform &t : transforms) any_bitimage |= t.is_bitimage();
    // the files in batches: read + base64 + SHA-256 on the host cores (file-size-bound), everything after on the device
    const size_t batch = 4096;
    for (size_t f0 = 0; f0 < files.size(); f0 += batch) {
        const size_t nf = std::min(batch, files.size() - f0);
        std::vector<std::array<uint8_t, 32>> ent(nf);
        std::vector<uint8_t> ok(nf, 0);
        std::vector<std::string> errs(nf);
        auto work = [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                FILE *fp = fopen(files[f0 + i].c_str(), "rb");
                if (!fp) {
                    errs[i] = strerror(errno);
                    continue;
                }
                std::vector<uint8_t> data;
                uint8_t buf[1 << 16];
                size_t got;
                while ((got = fread(buf, 1, sizeof buf, fp)) > 0) data.insert(data.end(), buf, buf + got);
                const bool bad = ferror(fp) != 0;
                if (bad) errs[i] = strerror(errno);
                fclose(fp);
                if (bad) continue;
                if (any_bitimage) bitimage_entropy(data.data(), data.size(), ent[i].data());
                ok[i] = 1;
            }
        };
        if (nf >= 2) {
            unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), (unsigned)std::min<size_t>(nf, 32));
            std::vector<std::thread> th;
            for (unsigned w = 0; w < nt; w++) th.emplace_back(work, nf * w / nt, nf * (w + 1) / nt);
            for (auto &t : th) t.join();
        } else {
            work(0, nf);
        }
        std::vector<std::array<uint8_t, 32>> ents;
        std::vector<std::string> labels;
        for (size_t i = 0; i < nf; i++) {
            if (!ok[i]) {
                fprintf(stderr, "Warning: Failed to read '%s': %s\n", files[f0 + i].c_str(), errs[i].c_str());
                continue;
            }
            ents.push_back(ent[i]);
            labels.push_back(files[f0 + i]); // path.display().to_string()
        }
        st.inputs_processed += labels.size();
        if (labels.empty()) continue;
        for (const Transform &t : transforms) {
            std::vector<std::pair<std::string, Key>> out;
            if (t.is_bitimage()) t.apply_batch_blobs(engine_, ents, labels, out);
            else t.apply_batch(engine_, labels, out); // Input::string_val = the label
            emit_keys(engine_, t, out, matcher, output, st);
        }
    }
    if (matcher) check_rc(engine_, vk_targets_set(engine_, nullptr, nullptr, 0));
    return st;
}

} // namespace vuke
