// host_chain_emu.cpp -- TEST INFRASTRUCTURE: the cold kernels and the chunk bodies of the key-derivation chains
// (vuke_amd/csrc/vk_chain.cuh, vk_multibit.cuh, vk_electrum.cuh minus its EC tail) compiled for the host, one lane at a time:
// HIP's kernel syntax and the handful of gfx950 builtins they use are shimmed below, the kernels are called as plain functions
// with threadIdx / blockIdx set by the driver loop.  tests/test_host_chain.py holds the results to hashlib (PBKDF2-HMAC-SHA512,
// SHA-256) and to a python restatement of the BIP-39 repacking, on a machine without a GPU.  What this checks is the LOGIC of the
// new code paths (staging slices, padding patched in registers, word extraction, salt blocks, the ring protocol's bookkeeping);
// memory ordering between waves and the compiler's device code generation are what the GPU parity tests check.
// Build (clang: the sources use ext_vector_type):  clang++ -std=c++17 -O1 -I vuke_amd/csrc -I tests/host_emu host_chain_emu.cpp
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hip_shims.h"

#include "vk_chain.cuh"
#include "vk_multibit.cuh"
#include "vk_electrum.cuh"

static std::vector<unsigned char> unhex(const char *h)
{
    std::vector<unsigned char> o;
    if (h[0] == '-' && h[1] == 0) return o; // "-" = empty
    for (size_t i = 0; h[i] && h[i + 1]; i += 2) {
        unsigned v;
        sscanf(h + i, "%2x", &v);
        o.push_back((unsigned char)v);
    }
    return o;
}
static void print_words(const char *tag, const u64 *w, int n)
{
    printf("%s ", tag);
    for (int i = 0; i < n; i++) printf("%016llx", (unsigned long long)w[i]);
    printf("\n");
}

// every lane of group 0 through all chunks of one stretch, popping and pushing through the ring as ONE worker would
template <class Body>
static void run_stretch(typename Body::args a, u32 groups, u32 iters, u32 chunk_iters, std::vector<std::pair<u32, u32>> *order = nullptr)
{
    const u32 chunks = (iters + chunk_iters - 1) / chunk_iters;
    u32 nslots = 1;
    while (nslots < 2 * groups) nslots <<= 1;
    vk_ring ring;
    std::vector<unsigned long long> slots(nslots);
    std::vector<u32> done(groups);
    blockDim.x = 256;
    for (u32 i = 0; i < nslots; i++) {
        blockIdx.x = i / 256;
        threadIdx.x = i % 256;
        vk_ring_init_kernel(&ring, slots.data(), done.data(), groups, chunks, nslots);
    }
    a.iters = iters;
    a.chunk_iters = chunk_iters;
    blockIdx.x = 0;
    for (;;) { // vk_ring_kernel's loop with the 64 lanes of the wave run one after the other around the wave-uniform ring calls
        threadIdx.x = 0;
        const int g = ring_pop(&ring, slots.data());
        if (g < 0) break;
        const u32 c = done[g];
        if (order) order->push_back({(u32)g, c});
        for (u32 lane = 0; lane < 64; lane++) {
            threadIdx.x = lane;
            Body::run(a, (u32)g, c);
        }
        threadIdx.x = 0;
        done[g] = c + 1;
        if (c + 1 < chunks) ring_push(&ring, slots.data(), (u32)g);
    }
    if (ring.error || ring.head != ring.total + 1 || ring.tail != ring.total) {
        printf("ring-error %u %u %u %u\n", ring.error, ring.head, ring.tail, ring.total);
        exit(3);
    }
}

int main(int argc, char **argv)
{
    // argv[1] = word list file (2048 lines), needed by the mnemonic commands
    std::vector<u64> wordlist(2048, 0), wordset(VK_WORDSET_SLOTS, 0);
    if (argc > 1) {
        FILE *f = fopen(argv[1], "r");
        char line[64];
        for (int i = 0; i < 2048 && fgets(line, sizeof line, f); i++) {
            size_t n = strcspn(line, "\r\n");
            u64 w = 0;
            memcpy(&w, line, std::min<size_t>(n, 8));
            wordlist[i] = w;
            u32 h = (u32)((w * 0x9E3779B97F4A7C15ull) >> 52); // as vk_bip39_wordlist_set builds it
            while (wordset[h] != 0 && wordset[h] != w) h = (h + 1) & (VK_WORDSET_SLOTS - 1);
            wordset[h] = w;
        }
        fclose(f);
    }
    static char line[40000], a1[16384], a2[16384];
    char cmd[32];
    while (fgets(line, sizeof line, stdin)) {
        a1[0] = a2[0] = 0;
        if (sscanf(line, "%31s %16383s %16383s", cmd, a1, a2) < 2) continue;
        const std::vector<unsigned char> x = unhex(a1), y = unhex(a2[0] ? a2 : "-");
        blockIdx.x = 0;
        threadIdx.x = 0;
        if (!strcmp(cmd, "sha256stage") || !strcmp(cmd, "sha512stage")) { // hash of a string written through the stage writer
            std::vector<u64> stage(VK_STAGE_WORDS, ~0ull);
            vk_stage_writer wr;
            wr.init(stage.data());
            for (unsigned char c : x) wr.put(c, 1);
            const u32 len = wr.finish();
            if (cmd[3] == '2') {
                u32 out[8];
                sha256_stage(out, stage.data(), len);
                for (int i = 0; i < 8; i++) printf("%08x", out[i]);
                printf("\n");
            } else {
                u64 st[8] = VK_SHA512_IV;
                sha512_absorb_stage(st, stage.data(), len, 0);
                print_words("", st, 8);
            }
        } else if (!strcmp(cmd, "multibit")) { // mnemonic (hex), passphrase (hex): ok, seed1, the 48-word string, seed2
            std::vector<unsigned char> data = x;
            const u32 offs[2] = {0, (u32)x.size()};
            std::vector<u64> saltv((size_t)vk_salt_nblk(y.size()) * 16);
            u64 *salt = saltv.data();
            const u32 nblk = vk_salt_blocks(salt, y.data(), y.size());
            std::vector<u64> stage(VK_STAGE_WORDS), state(VK_PB_WORDS * 64, 0);
            unsigned char ok = 7;
            data.push_back(0);
            vk_mb_prepare_kernel(data.data(), offs, 1, wordset.data(), salt, nblk, stage.data(), state.data(), &ok);
            printf("ok %u\n", ok);
            if (!ok) continue;
            vk_pbkdf2_body::args a = {state.data(), 0, 0};
            run_stretch<vk_pbkdf2_body>(a, 1, 2047, 64);
            u64 seed[8];
            pb_state_result(seed, state.data(), 0);
            print_words("seed1", seed, 8);
            vk_mb_middle_kernel(1, wordlist.data(), salt, nblk, stage.data(), state.data(), &ok);
            printf("words %s\n", std::string((const char *)stage.data()).c_str());
            run_stretch<vk_pbkdf2_body>(a, 1, 2047, 100);
            pb_state_result(seed, state.data(), 0);
            print_words("seed2", seed, 8);
        } else if (!strcmp(cmd, "bitimage")) { // entropy (32 bytes hex), passphrase (hex): mnemonic, seed
            std::vector<u64> saltv((size_t)vk_salt_nblk(y.size()) * 16);
            u64 *salt = saltv.data();
            const u32 nblk = vk_salt_blocks(salt, y.data(), y.size());
            std::vector<u64> stage(VK_STAGE_WORDS), state(VK_PB_WORDS * 64, 0);
            alignas(8) unsigned char ent[32];
            memcpy(ent, x.data(), 32);
            const u32 salt_off[2] = {0, 16 * nblk};
            vk_bitimage_prepare_kernel(ent, 1, 1, wordlist.data(), salt, salt_off, stage.data(), state.data());
            printf("words %s\n", std::string((const char *)stage.data()).c_str());
            vk_pbkdf2_body::args a = {state.data(), 0, 0};
            run_stretch<vk_pbkdf2_body>(a, 1, 2047, 2047);
            u64 seed[8];
            pb_state_result(seed, state.data(), 0);
            print_words("seed", seed, 8);
        } else if (!strcmp(cmd, "electrum")) { // seed string (hex), rounds (decimal in a2): ok, stretched key
            // "rounds" or "rounds@k": the seed starts k bytes into the input buffer (the aligned-dword reads of the long path)
            const u32 rounds = (u32)atoi(a2);
            const char *at = strchr(a2, '@');
            const u32 shift = at ? (u32)atoi(at + 1) : 0u;
            std::vector<unsigned char> data(shift, (unsigned char)'f');
            data.insert(data.end(), x.begin(), x.end());
            const u32 offs[2] = {shift, shift + (u32)x.size()};
            std::vector<u64> stage(VK_STAGE_WORDS);
            std::vector<u32> state(VK_EL_WORDS * 64, 0);
            unsigned char ok = 7;
            data.resize(data.size() + VK_STRING_SLACK, 0); // what the engine's device copy of the input carries behind the last string
            vk_electrum_prepare_kernel(data.data(), offs, 1, stage.data(), state.data(), &ok);
            printf("ok %u\n", ok);
            if (!ok) continue;
            vk_electrum_body::args a = {state.data(), 0, 0};
            if (rounds > 1) run_stretch<vk_electrum_body>(a, 1, rounds - 1, 37);
            if (ok == 2) { // the host's loop over bounded launches (vk_engine.hip), with slices of 7 rounds so that a short run has many
                for (u32 r = 0; r < rounds; r += 7) vk_electrum_long_kernel(data.data(), offs, 1, r, std::min(rounds, r + 7), rounds, state.data(), &ok);
            }
            printf("x ");
            for (int i = 0; i < 8; i++) printf("%08x", state[64 * i]);
            printf("\n");
        } else if (!strcmp(cmd, "ring")) { // groups, chunks: the order in which ONE worker visits (group, chunk)
            struct nop_body {
                struct args {
                    u32 *state;
                    u32 iters, chunk_iters;
                };
                static void run(const args &, u32, u32) {}
            };
            const u32 groups = (u32)atoi(a1), chunks = (u32)atoi(a2);
            std::vector<std::pair<u32, u32>> order;
            nop_body::args a = {nullptr, 0, 0};
            run_stretch<nop_body>(a, groups, chunks, 1, &order);
            for (auto &p : order) printf("%u:%u ", p.first, p.second);
            printf("\n");
        }
        fflush(stdout);
    }
    return 0;
}
