/* hip_range_driver.c -- TEST INFRASTRUCTURE.  The scan-mode call sequence of bindings/rust/src/hip_range.rs (`impl Source for
 * HipRangeSource`, the branch with a matcher) performed in strict C11 through include/vuke_hip.h against libvuke_hip.so: the Rust crate
 * cannot be compiled in this pipeline (no rustc / cargo), so what it DOES is executed here, statement for statement --
 *     vk_targets_set -> per spec { vk_slots_per_input; loop { vk_sweep; break when hits_dropped == 0; grow by hits_dropped + 1024 };
 *     truncate to matches; sort by (seed, sub_index); print seed + label_suffix(spec, sub_index) } -> vk_targets_set(NULL, NULL, 0)
 * -- and tests/test_gpu_parity.py compares its output with the oracle's hits and labels; tests/test_abi_layout.py holds this file and
 * hip_range.rs to each other as TEXT (the same vk_* calls in the same order, the same growth rule, the same three string tables and
 * format strings of label_suffix), so that an edit of one without the other fails on the CPU and an edit of both fails on the GPU.
 *
 *   usage: hip_range_driver <targets file: "hex40 kind" per line> <start> <end> <initial hit capacity> <kind variant endian depth> ...
 *   prints one line per hit: "<label> <addr_kind> <privkey hex>", then "stats <inputs> <keys> <matches> <sweep calls>" per spec */
#include "vuke_hip.h"
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int check(vk_engine *e, int rc, const char *what)
{
    if (rc == VK_OK) return 0;
    fprintf(stderr, "hip_range_driver: %s: vuke_hip error %d: %s\n", what, rc, vk_last_error(e));
    return 1;
}

/* label_suffix of hip_range.rs, transliterated (the tables and format strings are compared with the Rust text by the CPU test) */
static void label_suffix(const vk_transform_spec *spec, uint32_t sub, char *out, size_t cap)
{
    static const char *LCG[4] = {"glibc", "minstd", "msvc", "borland"};
    static const char *XS[4] = {"xorshift64", "xorshift128", "xorshift128+", "xoroshiro128**"};
    static const char *CH[4] = {"iterated", "indexed:be", "indexed:le", "counter"};
    const int all = spec->variant == VK_VARIANT_ALL;
    out[0] = 0;
    if (spec->kind == VK_T_LCG && all) {
        snprintf(out, cap, ":%s%s", LCG[sub & 3], spec->endian != 0 ? ":le" : "");
    } else if (spec->kind == VK_T_XORSHIFT && all) {
        snprintf(out, cap, ":%s", XS[sub & 3]);
    } else if (spec->kind == VK_T_SHA256_CHAIN) {
        const uint32_t nv = all ? 4 : 1, d = spec->chain_depth > 1 ? spec->chain_depth : 1;
        const uint32_t r = sub % (nv * d);
        const uint32_t v = all ? r / d : spec->variant;
        snprintf(out, cap, "[%s:%" PRIu32 "]", CH[v & 3], r % d);
    }
}

static int by_seed_sub(const void *a, const void *b)
{
    const vk_hit *x = (const vk_hit *)a, *y = (const vk_hit *)b;
    if (x->seed != y->seed) return x->seed < y->seed ? -1 : 1;
    return x->sub_index < y->sub_index ? -1 : x->sub_index > y->sub_index;
}

int main(int argc, char **argv)
{
    if (argc < 9 || (argc - 5) % 4) {
        fprintf(stderr, "usage: hip_range_driver targets start end hit_cap kind variant endian depth [...]\n");
        return 2;
    }
    const uint64_t start = strtoull(argv[2], NULL, 0), end = strtoull(argv[3], NULL, 0);
    const size_t cap0 = (size_t)strtoull(argv[4], NULL, 0);
    /* Matcher::hash160_targets(): the decoded P2PKH (kind 0) / P2WPKH (kind 1) lines */
    size_t n = 0, room = 1024;
    uint8_t *h160 = malloc(20 * room), *kind = malloc(room);
    FILE *f = fopen(argv[1], "r");
    if (!f || !h160 || !kind) return 2;
    char hex[64];
    unsigned k;
    while (fscanf(f, "%40s %u", hex, &k) == 2) {
        if (n == room) {
            room *= 2;
            h160 = realloc(h160, 20 * room);
            kind = realloc(kind, room);
            if (!h160 || !kind) return 2;
        }
        for (int i = 0; i < 20; i++) {
            unsigned b;
            sscanf(hex + 2 * i, "%2x", &b);
            h160[20 * n + i] = (uint8_t)b;
        }
        kind[n++] = (uint8_t)k;
    }
    fclose(f);

    vk_engine *e = NULL;
    if (check(NULL, vk_engine_create(0, &e), "vk_engine_create")) return 1;
    if (check(e, vk_targets_set(e, h160, kind, n), "vk_targets_set")) return 1;
    for (int a = 5; a < argc; a += 4) {
        vk_transform_spec spec;
        spec.kind = (uint32_t)strtoul(argv[a], NULL, 0);
        spec.variant = (uint32_t)strtoul(argv[a + 1], NULL, 0);
        spec.endian = (uint32_t)strtoul(argv[a + 2], NULL, 0);
        spec.chain_depth = (uint32_t)strtoul(argv[a + 3], NULL, 0);
        const uint32_t slots = vk_slots_per_input(&spec);
        (void)slots;
        size_t len = cap0;
        vk_hit *hits = calloc(len ? len : 1, sizeof(vk_hit));
        vk_sweep_stats st;
        unsigned calls = 0;
        for (;;) {
            memset(&st, 0, sizeof st);
            calls++;
            if (check(e, vk_sweep(e, &spec, start, end, VK_F_COMPRESSED | VK_F_UNCOMPRESSED, hits, len, &st), "vk_sweep")) return 1;
            if (st.hits_dropped == 0) break;
            len = len + (size_t)st.hits_dropped + 1024;
            hits = realloc(hits, len * sizeof(vk_hit));
            if (!hits) return 2;
        }
        const size_t matches = (size_t)st.matches;
        qsort(hits, matches, sizeof(vk_hit), by_seed_sub);
        for (size_t i = 0; i < matches; i++) {
            char suf[64];
            label_suffix(&spec, hits[i].sub_index, suf, sizeof suf);
            printf("%" PRIu64 "%s %" PRIu32 " ", hits[i].seed, suf, hits[i].addr_kind);
            for (int j = 0; j < 32; j++) printf("%02x", hits[i].privkey[j]);
            printf("\n");
        }
        printf("stats %" PRIu64 " %" PRIu64 " %" PRIu64 " %u\n", end - start + 1, st.keys, st.matches, calls);
        free(hits);
    }
    if (check(e, vk_targets_set(e, NULL, NULL, 0), "vk_targets_set(clear)")) return 1;
    vk_engine_destroy(e);
    free(h160);
    free(kind);
    return 0;
}
