// main.cpp -- `vuke-hip`: the generate / scan subcommands of the reference CLI (src/main.rs:85-434,
// run_generate :602-768, run_scan :770-952) driving the GPU engine.  Same argument spelling for the
// options on the hot path, same stderr status lines, same stdout formats.
//
//   vuke-hip generate [--transform T ...] [-v] [-o FILE] range --start A --end B
//   vuke-hip scan --transform T ... --targets FILE [-o FILE] range --start A --end B
//   vuke-hip single --transform T VALUE            (numeric VALUE; src/main.rs:954-992 format)
// Extras: --device N, --window-bits W.
#include "vuke_host.hpp"

#include <future>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <sys/stat.h>
#include <thread>

using namespace vuke;

static int usage(const char *msg)
{
    if (msg) fprintf(stderr, "error: %s\n\n", msg);
    fprintf(stderr, "Usage: vuke-hip <generate|scan> [OPTIONS] <SOURCE>\n"
                    "       vuke-hip single <PASSPHRASE> --transform <T>\n"
                    "       vuke-hip bench [--transform <T>] [--json]      transform-only inputs/s (the reference's `vuke bench`)\n"
                    "       vuke-hip analyze <KEY> [--fast] [--json] [--mask BITS] [--cascade b:t,b:t,...] [--analyzer milksad|lcg[:variant][:endian]|mt64|xorshift[:variant]|sha256_chain[:variant]]...\n"
                    "  --transform <T>...   sha256 | double_sha256 | md5 | direct | armory | milksad | mt64 | lcg[:variant][:endian] | xorshift[:variant] |\n"
                    "                       sha256_chain[:variant] | electrum[:change] | multibit | bitimage\n"
                    "  sources: range --start A --end B | timestamps --start YYYY-MM-DD --end YYYY-MM-DD [--microseconds] |\n"
                    "           wordlist --file F | stdin   (passphrase sources feed the string form of each transform) |\n"
                    "           files --file F | files --dir D   (file contents feed bitimage; other transforms see the path)\n"
                    "  --bitimage-path <P>  HD derivation path (default m/84'/0'/0'/0/0); --bitimage-passphrase <S>;\n"
                    "                       --bitimage-passphrase-wordlist <F> one passphrase per line; --bitimage-derive-count <N> (default 1)\n"
                    "  --targets <FILE>     (scan) one address per line\n"
                    "  -v, --verbose        (generate) all key formats\n"
                    "  -o, --output <FILE>  write to FILE instead of stdout\n"
                    "  --storage <DIR>      also write the 19-column Parquet record (DIR/transform=T/date=D/chunk_NNNN.parquet,\n"
                    "                       --chunk-records <N> / --chunk-bytes <N[K|M|G]> per file (defaults 1000000 / 100M),\n"
                    "                       --compression none|snappy|gzip|lz4|zstd (default zstd),\n"
                    "                       --compression-level <L> (default 3)\n"
                    "  --network <N>        bitcoin | testnet | signet | regtest: used by `single`; parsed and ignored by generate / scan\n"
                    "                       exactly like the reference (src/main.rs:464,499)\n"
                    "  --bip39-wordlist <F> 2048-word BIP-39 list (required by --transform multibit)\n"
                    "  --device <N>         GPU index (default 0)\n"
                    "  --devices <LIST|all> several GPUs: a range source is cut into one contiguous block per GPU\n"
                    "  --window-bits <W>    k*G table window width (default 24)\n"
                    "  --layout <L>         auto (default: by the job's key count -- nine signed windows from 2^34 keys, ten from 2^31, else uniform) |\n"
                    "                       uniform (11.8 GB table, 10 additions per key) | signed9 (120 GB table, 8 additions per key) |\n"
                    "                       signed (with --window-bits W: ceil(256 / W) signed windows, e.g. 26 -> 18 GB, 9 additions)\n");
    return 2;
}

static uint64_t parse_u64(const char *s);
static int run_analyze(int argc, char **argv);

static uint64_t parse_u64(const char *s)
{
    char *end = nullptr;
    errno = 0;
    unsigned long long v = strtoull(s, &end, 10);
    if (errno || !end || *end || s[0] == '-' || !s[0]) throw std::invalid_argument(std::string("invalid value '") + s + "' for a u64 argument");
    return v;
}

static bool is_source_word(const char *a)
{
    for (const char *w : {"range", "wordlist", "stdin", "timestamps", "files"})
        if (strcmp(a, w) == 0) return true;
    return false;
}

// `vuke analyze KEY [--mask N] [--cascade b:t,...] [--analyzer milksad|lcg[:variant][:endian]|mt64]...` (src/main.rs:307-360)
static int run_analyze(int argc, char **argv)
{
    std::string key_s, cascade_s;
    std::vector<std::string> names;
    std::optional<uint8_t> mask;
    int device = 0;
    uint64_t mt64_limit = (1ull << 40) - 1;
    uint32_t chain_depth = 10; // DEFAULT_CHAIN_DEPTH (src/sha256_chain.rs:82)
    bool fast = false, json = false;
    std::optional<std::string> mnemonic, mnemonic_file;
    std::string passphrase, bip39_file;
    try {
        for (int i = 2; i < argc; i++) {
            std::string a = argv[i];
            auto need = [&](const char *opt) -> const char * {
                if (i + 1 >= argc) throw std::invalid_argument(std::string("a value is required for '") + opt + "'");
                return argv[++i];
            };
            if (a == "--mask") {
                int m = atoi(need("--mask"));
                if (m < 1 || m > 64) throw std::invalid_argument("--mask must be 1..64");
                mask = (uint8_t)m;
            } else if (a == "--cascade") cascade_s = need("--cascade");
            else if (a == "--fast") fast = true;
            else if (a == "--mnemonic") mnemonic = need("--mnemonic");
            else if (a == "--mnemonic-file") mnemonic_file = need("--mnemonic-file");
            else if (a == "--passphrase") passphrase = need("--passphrase");
            else if (a == "--bip39-wordlist") bip39_file = need("--bip39-wordlist");
            else if (a == "--json") json = true;
            else if (a == "--analyzer") names.push_back(need("--analyzer"));
            else if (a == "--device") device = atoi(need("--device"));
            else if (a == "--mt64-limit" || a == "--seed-limit") mt64_limit = parse_u64(need("--seed-limit")); // bound for the 64-bit seed spaces
            else if (a == "--chain-depth") chain_depth = (uint32_t)parse_u64(need("--chain-depth"));
            else if (key_s.empty() && a[0] != '-') key_s = a;
            else throw std::invalid_argument("unexpected argument '" + a + "'");
        }
        if (key_s.empty()) throw std::invalid_argument("analyze needs a KEY (hex, WIF or decimal)");
        // AnalyzerType::fast() / all() (src/analyze/mod.rs:199-218, src/main.rs:1534-1535)
        if (names.empty()) names = fast ? std::vector<std::string>{"direct", "heuristic"}
                                        : std::vector<std::string>{"milksad", "mt64", "lcg", "xorshift", "sha256_chain", "direct", "heuristic"};
        Key key = parse_private_key(key_s);
        if (mask && calculate_bit_length(key) > *mask) // src/main.rs:1495-1503
            fprintf(stderr, "Warning: key has %u bits but mask is %u bits. Key will be treated as already masked.\n", calculate_bit_length(key), *mask);
        AnalysisConfig cfg;
        cfg.mask_bits = mask;
        if (!cascade_s.empty()) cfg.cascade_targets = parse_cascade(cascade_s);
        vk_engine *engine = nullptr;
        bool device_needed = false, ec_needed = false; // direct / heuristic look at the key bytes only: `analyze --fast` runs without a GPU
        for (const std::string &n : names) {
            device_needed = device_needed || (n != "direct" && n != "heuristic");
            ec_needed = ec_needed || n == "multibit-hd" || n == "multibit"; // CKD needs serP(k*G)
        }
        if (device_needed) {
            vk_engine_opts opts{};
            opts.window_bits = ec_needed ? 16 : 8; // the search analyzers never touch the k*G table or the inversion workspace: keep both tiny
            opts.batch = 1;
            if (vk_engine_create_ex(device, &opts, &engine) != VK_OK) {
                fprintf(stderr, "GPU engine unavailable: %s\n", vk_last_error(nullptr));
                return 1;
            }
        }
        struct Guard { vk_engine *e; ~Guard() { vk_engine_destroy(e); } } guard{engine};
        std::vector<AnalysisResult> results;
        for (const std::string &n : names) {
            if (n == "multibit-hd" || n == "multibit") { // src/analyze/mod.rs:225-229
                if (bip39_file.empty()) throw std::invalid_argument("multibit-hd needs --bip39-wordlist <FILE> (the 2048-word BIP-39 English list)");
                load_bip39_wordlist(engine, bip39_file);
                results.push_back(MultibitAnalyzer(engine, mnemonic, mnemonic_file, passphrase).analyze(key, cfg));
            } else if (n == "direct") results.push_back(DirectAnalyzer().analyze(key, cfg));
            else if (n == "heuristic") results.push_back(HeuristicAnalyzer().analyze(key, cfg));
            else if (n == "milksad") results.push_back(MilksadAnalyzer(engine).analyze(key, cfg));
            else if (n == "mt64") results.push_back(Mt64Analyzer(engine, mt64_limit).analyze(key, cfg));
            else if (n == "lcg" || n.rfind("lcg:", 0) == 0) {
                Transform t = TransformType::from_str(n); // same grammar as the transform (src/lcg.rs:247-277)
                results.push_back(LcgAnalyzer(engine, t.spec().variant, t.spec().endian).analyze(key, cfg));
            } else if (n == "xorshift" || n.rfind("xorshift:", 0) == 0) {
                Transform t = TransformType::from_str(n); // XorshiftConfig::parse (src/xorshift.rs:334-357)
                results.push_back(XorshiftAnalyzer(engine, t.spec().variant, mt64_limit).analyze(key, cfg));
            } else if (n == "sha256_chain" || n.rfind("sha256_chain:", 0) == 0) {
                Transform t = TransformType::from_str(n); // Sha256ChainConfig::parse (src/sha256_chain.rs:187-217)
                results.push_back(Sha256ChainAnalyzer(engine, t.spec().variant, chain_depth).analyze(key, cfg));
            } else throw std::invalid_argument("analyzer '" + n + "' is not a known analyzer (milksad, lcg[:variant][:endian], mt64, xorshift[:variant], sha256_chain[:variant], multibit-hd, direct, heuristic)");
        }
        if (json) puts(format_results_json(key, results).c_str());
        else fputs(format_results(key, results).c_str(), stdout);
        return 0;
    } catch (const EngineError &e) {
        fprintf(stderr, "Error: GPU engine failure (%d): %s\n", e.code, e.what());
        return 1;
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}

// `vuke bench [--transform T] [--json]` (src/benchmark.rs:11-67): transform-only inputs per second -- no k*G, no hashing
// to an address, no match.  On the device that is a vk_search sweep against a key no input produces; the reference
// feeds one input a million times, here distinct seeds are swept (same work per input for every transform).
static int run_bench(int argc, char **argv)
{
    std::string tname = "sha256";
    bool json = false;
    int device = 0;
    double seconds = 5.0;
    try {
        for (int i = 2; i < argc; i++) {
            std::string a = argv[i];
            auto need = [&](const char *opt) -> const char * {
                if (i + 1 >= argc) throw std::invalid_argument(std::string("a value is required for '") + opt + "'");
                return argv[++i];
            };
            if (a == "--transform") tname = need("--transform");
            else if (a == "--json") json = true;
            else if (a == "--device") device = atoi(need("--device"));
            else if (a == "--seconds") seconds = atof(need("--seconds"));
            else throw std::invalid_argument("unexpected argument '" + a + "'");
        }
        Transform t = TransformType::from_str(tname);
        vk_transform_spec sp = t.spec();
        const uint32_t k = sp.kind;
        if (k == VK_T_ELECTRUM || k == VK_T_MULTIBIT || k == VK_T_ARMORY)
            throw std::invalid_argument(std::string("bench: '") + t.name() + "' has no transform-only sweep on the device (measure it with scan / generate)");
        if ((k == VK_T_LCG || k == VK_T_XORSHIFT) && sp.variant == VK_VARIANT_ALL) sp.variant = 0; // one generator per sweep
        vk_search_target tg;
        memset(&tg, 0, sizeof tg);
        if (k == VK_T_MT64 || k == VK_T_XORSHIFT) { // cascade-only generators: one draw against a 64-bit target (chance 2^-64 per seed)
            tg.mode = VK_SEARCH_CASCADE;
            tg.n_cascade = 1;
            tg.bits[0] = 64;
            tg.targets[0] = ~0ull;
        } else {
            tg.mode = VK_SEARCH_EXACT; // key = 0x55..55: not an output of any of these transforms over the swept seeds
            memset(tg.key, 0x55, 32);
        }
        vk_engine *engine = nullptr;
        vk_engine_opts opts{};
        opts.window_bits = 8;
        opts.batch = 1;
        if (vk_engine_create_ex(device, &opts, &engine) != VK_OK) {
            fprintf(stderr, "GPU engine unavailable: %s\n", vk_last_error(nullptr));
            return 1;
        }
        struct Guard { vk_engine *e; ~Guard() { vk_engine_destroy(e); } } guard{engine};
        if (!json) {
            printf("Running Benchmark for %s...\n", t.name());
            printf("Time: 1 warmup sweep + %.0fs measure (approx)\n", seconds);
        }
        const uint64_t chunk = (1ull << 30) - 1; // seeds lo .. lo + chunk, inside u32 for the 32-bit generators
        vk_search_result r;
        check_rc(engine, vk_search(engine, &sp, 0, (1ull << 24) - 1, &tg, &r));
        uint64_t count = 0, lo = 0;
        const auto t0 = std::chrono::steady_clock::now();
        double dur = 0;
        do {
            check_rc(engine, vk_search(engine, &sp, lo, lo + chunk, &tg, &r));
            if (r.found) throw std::runtime_error("bench: the sentinel key matched; pick another transform");
            count += r.seeds_checked;
            lo = (lo + chunk + 1) & 0x7FFFFFFFull; // stays below every generator's seed ceiling
            dur = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        } while (dur < seconds);
        const double speed = (double)count / dur;
        if (json)
            printf("{ \"name\": \"%s\", \"ops_per_sec\": %llu, \"total_ops\": %llu, \"duration_secs\": %f }\n", t.name(), (unsigned long long)speed,
                   (unsigned long long)count, dur);
        else
            printf("------------------------------------------------\nResult: %.2f Million Inputs/sec\nTotal:  %llu inputs in %.2fs\n"
                   "------------------------------------------------\n",
                   speed / 1e6, (unsigned long long)count, dur);
        return 0;
    } catch (const EngineError &e) {
        fprintf(stderr, "Error: GPU engine failure (%d): %s\n", e.code, e.what());
        return 1;
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}

int main(int argc, char **argv)
{
    if (argc < 2) return usage(nullptr);
    const std::string cmd = argv[1];
    if (cmd == "analyze") return run_analyze(argc, argv);
    if (cmd == "bench") return run_bench(argc, argv);
    if (cmd != "generate" && cmd != "scan" && cmd != "single") return usage("unknown subcommand");
    std::vector<std::string> tnames;
    std::string targets, outpath, single_value;
    bool verbose = false, have_range = false, have_start = false, have_end = false, have_wordlist = false, have_timestamps = false, microseconds = false;
    std::string wordlist_file, bip39_file, start_s, end_s, network = "bitcoin", devices_s, storage_dir;
    bool have_files = false;
    std::string files_dir, bitimage_pass_wordlist;
    BitimageConfig bitimage_cfg; // src/main.rs:174-188
    uint64_t chunk_records = 1000000; // DEFAULT_CHUNK_RECORDS, src/storage/parquet_backend.rs:13
    uint64_t chunk_bytes = 100ull << 20; // 100M, src/main.rs:130-132
    std::string compression = "zstd";   // src/main.rs:134-141
    int compression_level = 3;
    uint64_t start = 0, end = 0;
    int device = 0;
    unsigned window_bits = 0, layout = VK_LAYOUT_UNIFORM;
    bool layout_auto = true;
    try {
        for (int i = 2; i < argc; i++) {
            std::string a = argv[i];
            auto need = [&](const char *opt) -> const char * {
                if (i + 1 >= argc) throw std::invalid_argument(std::string("a value is required for '") + opt + "'");
                return argv[++i];
            };
            if (a == "--transform") {
                tnames.push_back(need("--transform"));
                while (cmd != "single" && i + 1 < argc && argv[i + 1][0] != '-' && !is_source_word(argv[i + 1])) tnames.push_back(argv[++i]); // num_args = 1..
            } else if (a == "--targets") targets = need("--targets");
            else if (a == "-o" || a == "--output") outpath = need("--output");
            else if (a == "-v" || a == "--verbose") verbose = true;
            else if (a == "--network") network = need("--network");
            else if (a == "--device") device = atoi(need("--device"));
            else if (a == "--devices") devices_s = need("--devices");
            else if (a == "--storage") storage_dir = need("--storage");
            else if (a == "--chunk-records") chunk_records = parse_u64(need("--chunk-records"));
            else if (a == "--chunk-bytes") { // parse_byte_size, src/main.rs:29-46
                std::string v = need("--chunk-bytes");
                uint64_t mul = 1;
                if (!v.empty() && strchr("kKmMgG", v.back())) {
                    mul = (v.back() == 'k' || v.back() == 'K') ? 1024ull : (v.back() == 'm' || v.back() == 'M') ? 1048576ull : 1073741824ull;
                    v.pop_back();
                }
                chunk_bytes = parse_u64(v.c_str()) * mul;
            }
            else if (a == "--compression") compression = need("--compression");
            else if (a == "--compression-level") compression_level = atoi(need("--compression-level"));
            else if (a == "--window-bits") window_bits = (unsigned)atoi(need("--window-bits"));
            else if (a == "--layout") {
                const std::string l = need("--layout");
                if (l != "auto" && l != "uniform" && l != "signed9" && l != "signed") throw std::invalid_argument("--layout takes auto, uniform, signed or signed9");
                layout_auto = l == "auto";
                layout = (l == "signed9" || l == "signed") ? VK_LAYOUT_SIGNED9 : VK_LAYOUT_UNIFORM; // signed + --window-bits W: ceil(256 / W) signed windows
            }
            else if (a == "range") have_range = true;
            else if (a == "wordlist") have_wordlist = true;
            else if (a == "stdin") { have_wordlist = true; wordlist_file = "-"; }
            else if (a == "timestamps") have_timestamps = true;
            else if (a == "files") have_files = true;
            else if (a == "--dir") files_dir = need("--dir");
            else if (a == "--bitimage-path") bitimage_cfg.path = need("--bitimage-path");
            else if (a == "--bitimage-passphrase") bitimage_cfg.passphrase = need("--bitimage-passphrase");
            else if (a == "--bitimage-passphrase-wordlist") bitimage_pass_wordlist = need("--bitimage-passphrase-wordlist");
            else if (a == "--bitimage-derive-count") {
                const uint64_t v = parse_u64(need("--bitimage-derive-count"));
                if (v > 0xFFFFFFFFull) throw std::invalid_argument("--bitimage-derive-count does not fit a u32");
                bitimage_cfg.derive_count = (uint32_t)v;
            }
            else if (a == "--microseconds") microseconds = true;
            else if (a == "--file") wordlist_file = need("--file");
            else if (a == "--bip39-wordlist") bip39_file = need("--bip39-wordlist");
            else if (a == "--start") { start_s = need("--start"); have_start = true; }
            else if (a == "--end") { end_s = need("--end"); have_end = true; }
            else if (cmd == "single" && single_value.empty() && a[0] != '-') single_value = a;
            else throw std::invalid_argument("unexpected argument '" + a + "'");
        }
        if (cmd != "single") (void)parse_network(network); // src/main.rs:464: parsed (an unknown name warns), then unused
        if (cmd == "generate" && tnames.empty()) tnames.push_back("sha256"); // default_value, src/main.rs:107
        if (tnames.empty()) throw std::invalid_argument("the following required arguments were not provided: --transform <TRANSFORM>...");
        std::vector<Transform> transforms;
        for (auto &n : tnames) transforms.push_back(TransformType::from_str(n));
        if (!bitimage_pass_wordlist.empty()) bitimage_cfg.passphrase_wordlist = BitimageConfig::load_passphrase_wordlist(bitimage_pass_wordlist);
        for (Transform &t : transforms)
            if (t.is_bitimage()) t.bitimage = bitimage_cfg; // apply_bitimage_config, src/main.rs:48-70
        if (cmd == "scan" && targets.empty()) throw std::invalid_argument("the following required arguments were not provided: --targets <TARGETS>");
        if (cmd != "single" && have_wordlist && wordlist_file.empty()) throw std::invalid_argument("wordlist needs --file <FILE>");
        if (cmd != "single" && have_files) {
            if (!wordlist_file.empty() && !files_dir.empty()) throw std::invalid_argument("the argument '--file <FILE>' cannot be used with '--dir <DIR>'");
            if (wordlist_file.empty() && files_dir.empty()) throw std::invalid_argument("Either --file or --dir must be specified"); // src/main.rs:1361
        }
        if (cmd != "single" && !have_wordlist && !have_files && ((!have_range && !have_timestamps) || !have_start || !have_end))
            throw std::invalid_argument("sources on the device path: `range --start A --end B`, `timestamps --start DATE --end DATE [--microseconds]`, "
                                        "`wordlist --file F`, `stdin`, `files --file F | --dir D`");
        if (have_range) {
            start = parse_u64(start_s.c_str());
            end = parse_u64(end_s.c_str());
        }
        if (cmd != "single" && have_range && end < start) throw std::invalid_argument("--end must be >= --start");

        // --devices 0,1,... | all: one engine per listed GPU (range sources split the range between them)
        std::vector<int> devs;
        if (devices_s == "all") {
            for (int i = 0; i < vk_device_count(); i++) devs.push_back(i);
        } else if (!devices_s.empty()) {
            size_t pos = 0;
            while (pos <= devices_s.size()) {
                size_t q = devices_s.find(',', pos);
                const std::string part = devices_s.substr(pos, q == std::string::npos ? std::string::npos : q - pos);
                if (part.empty() || part.find_first_not_of("0123456789") != std::string::npos) throw std::invalid_argument("--devices takes a comma separated list of GPU indices, or 'all'");
                devs.push_back(atoi(part.c_str()));
                if (q == std::string::npos) break;
                pos = q + 1;
            }
        }
        if (devs.empty()) devs.push_back(device);
        vk_engine_opts opts{};
        opts.window_bits = window_bits;
        // The table geometry follows the size of the job (keys = inputs x slots of every transform), by measured END-TO-END time
        // (profiles/archive/r02_strong_geometry.txt, profiles/archive/r03_probe_c2_cli.txt): the 120 GB signed-window table sweeps a 2^32-key scan in
        // 2.6 s instead of 3.4 s, but allocating and releasing 120 GB costs 0.4 s on a quiet box and 2 - 6 s when the memory has just
        // been freed by another process (the driver scrubs VRAM), so it only pays from several sweeps' worth of keys on; in between
        // ten signed windows (18 GB, built in 15 ms; until round 4 the unsigned W = 26 table, 43 GB, at the same sweep rate) are the
        // fastest whole run; small jobs keep the 11.8 GB default.
        if (layout_auto && window_bits == 0 && have_range) {
            long double keys = 0;
            for (const Transform &t : transforms) keys += (long double)(end - start) * (t.takes_strings() ? t.string_slots() : t.slots());
            if (keys >= 0x1p34L) layout = VK_LAYOUT_SIGNED9;
            else if (keys >= 0x1p31L) { // round 4: ten SIGNED windows (18 GB) sweep at the rate of the unsigned 26-bit table (43 GB)
                layout = VK_LAYOUT_SIGNED9;
                opts.window_bits = 26;
            }
        }
        opts.layout = layout;
        // scan: the targets file is read and decoded on its own thread while the engines start (the HIP runtime's start and the table
        // build leave the host cores idle); a failure surfaces where the reference reports it, after "Loading targets"
        std::future<Matcher> targets_loading;
        if (cmd == "scan" && !targets.empty())
            targets_loading = std::async(std::launch::async, [&targets]() {
                Matcher m = Matcher::load(targets);
                m.prepare_device_targets();
                return m;
            });
        std::vector<vk_engine *> engines(devs.size(), nullptr);
        const bool timing_env = getenv("VUKE_HIP_TIMING") != nullptr;
        const auto t_create = std::chrono::steady_clock::now();
        struct Guard {
            std::vector<vk_engine *> &v;
            bool timing;
            ~Guard()
            {
                const auto t0 = std::chrono::steady_clock::now();
                for (vk_engine *e : v) vk_engine_destroy(e);
                if (timing) fprintf(stderr, "[timing] engines destroyed in %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
        } guard{engines, timing_env};
        for (int attempt = 0;; attempt++) {
            std::vector<int> rcs(devs.size(), VK_OK);
            std::vector<std::string> msgs(devs.size());
            std::vector<std::thread> th; // the k*G tables are built side by side
            for (size_t i = 0; i < devs.size(); i++)
                th.emplace_back([&, i]() {
                    rcs[i] = vk_engine_create_ex(devs[i], &opts, &engines[i]);
                    if (rcs[i] != VK_OK) msgs[i] = vk_last_error(nullptr); // the creation error is thread local
                });
            for (auto &t : th) t.join();
            size_t bad = devs.size();
            for (size_t i = 0; i < devs.size(); i++)
                if (rcs[i] != VK_OK && bad == devs.size()) bad = i;
            if (bad == devs.size()) break;
            // the automatic choice of the 120 GB table must not cost the run on a smaller or shared device
            const bool no_device = rcs[bad] == VK_E_NO_DEVICE;
            if (attempt == 0 && layout_auto && opts.layout == VK_LAYOUT_SIGNED9 && !no_device) {
                for (vk_engine *&e : engines) {
                    vk_engine_destroy(e);
                    e = nullptr;
                }
                fprintf(stderr, "signed-window table unavailable (%s); using the default 24-bit uniform table\n", msgs[bad].c_str());
                opts.layout = VK_LAYOUT_UNIFORM;
                opts.window_bits = 0;
                continue;
            }
            fprintf(stderr, "GPU engine unavailable (device %d): %s\n", devs[bad], msgs[bad].c_str());
            return 1;
        }
        vk_engine *engine = engines[0];
        const bool timing = timing_env; // stage times on stderr
        if (timing) fprintf(stderr, "[timing] engines created in %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_create).count());
        const auto t_start = std::chrono::steady_clock::now();
        auto stamp = [&](const char *what) {
            if (timing) fprintf(stderr, "[timing] %s at %.3f s after engine creation\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
        };
        for (const Transform &t : transforms)
            if (t.spec().kind == VK_T_MULTIBIT || (t.is_bitimage() && have_files)) {
                if (bip39_file.empty()) throw std::invalid_argument(std::string(t.name()) + " needs --bip39-wordlist <FILE> (the 2048-word BIP-39 English list)");
                for (vk_engine *e : engines) load_bip39_wordlist(e, bip39_file);
                break;
            }

        if (cmd == "single") { // run_single, src/main.rs:954-992: the value is a passphrase (Input::from_string)
            if (single_value.empty()) throw std::invalid_argument("single needs a PASSPHRASE");
            KeyDeriver deriver(engine, parse_network(network)); // KeyDeriver::with_network, src/main.rs:957-958
            for (const Transform &t : transforms) {
                std::vector<std::pair<std::string, Key>> out;
                t.apply_batch(engine, std::vector<std::string>{single_value}, out);
                if (out.empty()) {
                    fprintf(stderr, "No key generated from passphrase.\n");
                    continue;
                }
                for (auto &[label, key] : out) {
                    DerivedKey d = deriver.derive(key);
                    printf("Passphrase: \"%s\"\nTransform: %s\nSource: %s\n---\nPrivate Key (hex):     %s\nPrivate Key (decimal): %s\n"
                           "Private Key (binary):  %s\nBit Length:            %u\nHamming Weight:        %u\nLeading Zeros (hex):   %u\n"
                           "WIF (compressed):      %s\nWIF (uncompressed):    %s\n---\n"
                           "P2PKH (compressed):   %s\nP2PKH (uncompressed): %s\nP2WPKH:               %s\n",
                           single_value.c_str(), t.name(), label.c_str(), d.private_key_hex.c_str(), d.private_key_decimal.c_str(),
                           d.private_key_binary.c_str(), d.bit_length, d.hamming_weight, d.leading_zeros, d.wif_compressed.c_str(),
                           d.wif_uncompressed.c_str(), d.p2pkh_compressed.c_str(), d.p2pkh_uncompressed.c_str(), d.p2wpkh.c_str());
                }
            }
            return 0;
        }

        std::unique_ptr<ConsoleOutput> out;
        if (!outpath.empty()) out = ConsoleOutput::to_file(outpath, verbose);
        else if (verbose) out = ConsoleOutput::verbose();
        else out = std::make_unique<ConsoleOutput>();
        // --storage DIR: the same records also go to Parquet chunks (MultiOutput, src/main.rs:666-672)
        std::unique_ptr<StorageOutput> storage;
        std::unique_ptr<MultiOutput> multi;
        Output *sink = out.get();
        if (!storage_dir.empty()) {
            storage = std::make_unique<StorageOutput>(storage_dir, transforms[0].name(), chunk_records, compression, compression_level, chunk_bytes);
            multi = std::make_unique<MultiOutput>(std::vector<Output *>{out.get(), storage.get()});
            sink = multi.get();
        }
        std::unique_ptr<Source> source_ptr;
        if (have_files)
            source_ptr = std::make_unique<HipFilesSource>(files_dir.empty() ? HipFilesSource::from_file(engine, wordlist_file) : HipFilesSource::from_dir(engine, files_dir));
        else if (have_wordlist) source_ptr = std::make_unique<HipWordlistSource>(engines, wordlist_file);
        else if (have_timestamps) source_ptr = std::make_unique<HipTimestampSource>(engine, start_s, end_s, microseconds);
        else if (engines.size() > 1) source_ptr = std::make_unique<HipMultiRangeSource>(engines, start, end);
        else source_ptr = std::make_unique<HipRangeSource>(engine, start, end);
        Source &source = *source_ptr;
        ProcessStats stats;
        if (cmd == "scan") {
            fprintf(stderr, "Loading targets from \"%s\"...\n", targets.c_str());
            Matcher m = targets_loading.valid() ? targets_loading.get() : Matcher::load(targets);
            fprintf(stderr, "Loaded %zu targets.\n", m.count());
            fprintf(stderr, "Scanning...\n");
            stamp("targets loaded");
            stats = source.process(transforms, &m, *sink);
            stamp("source processed");
        } else {
            fprintf(stderr, "Generating keys...\n");
            stats = source.process(transforms, nullptr, *sink);
        }
        sink->flush();
        fprintf(stderr, "Done. Inputs: %llu, Keys: %llu, Matches: %llu\n", (unsigned long long)stats.inputs_processed,
                (unsigned long long)stats.keys_generated, (unsigned long long)stats.matches_found);
        if (storage) { // print_storage_summary_inline, src/main.rs:995-1018
            const std::vector<std::string> paths = storage->finish();
            auto fmt = [](uint64_t b) {
                char buf[32];
                if (b >= (1ull << 30)) snprintf(buf, sizeof buf, "%.1f GB", b / 1073741824.0);
                else if (b >= (1ull << 20)) snprintf(buf, sizeof buf, "%.1f MB", b / 1048576.0);
                else if (b >= 1024) snprintf(buf, sizeof buf, "%.1f KB", b / 1024.0);
                else snprintf(buf, sizeof buf, "%llu B", (unsigned long long)b);
                return std::string(buf);
            };
            std::vector<uint64_t> sizes;
            uint64_t total = 0;
            for (auto &pth : paths) {
                struct stat sb;
                sizes.push_back(stat(pth.c_str(), &sb) == 0 ? (uint64_t)sb.st_size : 0);
                total += sizes.back();
            }
            if (!paths.empty()) {
                fprintf(stderr, "\nStorage: %zu files written (%s total)\n", paths.size(), fmt(total).c_str());
                for (size_t i = 0; i < paths.size(); i++) fprintf(stderr, "  %s (%s)\n", paths[i].c_str(), fmt(sizes[i]).c_str());
            }
        }
        return 0;
    } catch (const EngineError &e) {
        fprintf(stderr, "Error: GPU engine failure (%d): %s\n", e.code, e.what());
        return 1;
    } catch (const std::invalid_argument &e) {
        return usage(e.what());
    } catch (const std::exception &e) {
        fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
}
