// output.cpp -- ConsoleOutput (src/output/console.rs:62-122): compact CSV line, verbose block,
// HIT block; one writer behind a mutex like the reference.
#include "vuke_host.hpp"

#include <stdexcept>

namespace vuke {

std::unique_ptr<ConsoleOutput> ConsoleOutput::verbose() { return std::unique_ptr<ConsoleOutput>(new ConsoleOutput(stdout, false, true)); }

std::unique_ptr<ConsoleOutput> ConsoleOutput::to_file(const std::string &path, bool verbose)
{
    FILE *fp = fopen(path.c_str(), "w");
    if (!fp) throw std::runtime_error("cannot create output file: " + path);
    return std::unique_ptr<ConsoleOutput>(new ConsoleOutput(fp, true, verbose));
}

ConsoleOutput::~ConsoleOutput()
{
    if (own_ && fp_) fclose(fp_);
}

void ConsoleOutput::key(const std::string &source, const std::string &transform, const DerivedKey &d)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (verbose_) {
        fprintf(fp_, "---\nsource: %s\ntransform: %s\nprivate_key: %s\nwif_compressed: %s\nwif_uncompressed: %s\n"
                     "p2pkh_compressed: %s\np2pkh_uncompressed: %s\np2wpkh: %s\n",
                source.c_str(), transform.c_str(), d.private_key_hex.c_str(), d.wif_compressed.c_str(), d.wif_uncompressed.c_str(),
                d.p2pkh_compressed.c_str(), d.p2pkh_uncompressed.c_str(), d.p2wpkh.c_str());
    } else {
        fprintf(fp_, "%s,%s,%s,%s\n", source.c_str(), transform.c_str(), d.private_key_hex.c_str(), d.p2pkh_compressed.c_str());
    }
}

void ConsoleOutput::hit(const std::string &source, const std::string &transform, const DerivedKey &d, const MatchInfo &m)
{
    std::lock_guard<std::mutex> lk(mu_);
    fprintf(fp_, "\n========== HIT ==========\nSource: %s\nTransform: %s\nMatched: %s (%s)\n---\nPrivate Key: %s\n"
                 "WIF (compressed): %s\nWIF (uncompressed): %s\n---\nP2PKH (compressed): %s\nP2PKH (uncompressed): %s\nP2WPKH: %s\n"
                 "=========================\n",
            source.c_str(), transform.c_str(), m.address.c_str(), address_type_str(m.address_type), d.private_key_hex.c_str(),
            d.wif_compressed.c_str(), d.wif_uncompressed.c_str(), d.p2pkh_compressed.c_str(), d.p2pkh_uncompressed.c_str(), d.p2wpkh.c_str());
}

bool ConsoleOutput::write_text(const char *text, size_t len)
{
    if (verbose_) return false;
    std::lock_guard<std::mutex> lk(mu_);
    fwrite(text, 1, len, fp_);
    return true;
}

bool ConsoleOutput::write_verbose(const char *text, size_t len)
{
    if (!verbose_) return false;
    std::lock_guard<std::mutex> lk(mu_);
    fwrite(text, 1, len, fp_);
    return true;
}

void ConsoleOutput::flush()
{
    std::lock_guard<std::mutex> lk(mu_);
    fflush(fp_);
}

} // namespace vuke
