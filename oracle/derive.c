/*
 * derive.c -- ORACLE (test infrastructure, see vko.h): KeyDeriver::derive
 * (src/derive.rs:78-165), Matcher (src/matcher.rs:45-108) and the RangeSource driver
 * (src/source/range.rs:26-79), restated on the CPU.
 */
#include "vko.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void hex_encode(const uint8_t *b, size_t n, char *out)
{
    static const char H[] = "0123456789abcdef";
    for (size_t i = 0; i < n; i++) { out[2 * i] = H[b[i] >> 4]; out[2 * i + 1] = H[b[i] & 15]; }
    out[2 * n] = 0;
}

/* BigUint::from_bytes_be(..).to_string() (src/derive.rs:121) */
static void decimal_of_be(const uint8_t b[32], char *out)
{
    uint32_t w[8];
    for (int i = 0; i < 8; i++) w[i] = (uint32_t)b[4 * i] << 24 | (uint32_t)b[4 * i + 1] << 16 | (uint32_t)b[4 * i + 2] << 8 | b[4 * i + 3];
    char tmp[96];
    int n = 0;
    for (;;) {
        int nz = 0;
        uint64_t rem = 0;
        for (int i = 0; i < 8; i++) {
            uint64_t cur = (rem << 32) | w[i];
            w[i] = (uint32_t)(cur / 10);
            rem = cur % 10;
            nz |= w[i] != 0;
        }
        tmp[n++] = (char)('0' + rem);
        if (!nz) break;
    }
    for (int i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
    out[n] = 0;
}

void vko_derive(const uint8_t key[32], vko_derived *d)
{
    memset(d, 0, sizeof *d);
    vko_normalize_key(key, d->raw);                    /* :79-94 */
    vko_pubkey(d->raw, d->pub33, d->pub65);            /* :97-99 */
    hex_encode(d->pub33, 33, d->pubkey_compressed);
    hex_encode(d->pub65, 65, d->pubkey_uncompressed);
    vko_hash160(d->pub33, 33, d->h160_c);
    vko_hash160(d->pub65, 65, d->h160_u);
    uint8_t buf[40];
    buf[0] = 0x00; memcpy(buf + 1, d->h160_c, 20);     /* :112 mainnet P2PKH version 0x00 */
    vko_base58check_encode(buf, 21, d->p2pkh_compressed, sizeof d->p2pkh_compressed);
    memcpy(buf + 1, d->h160_u, 20);                    /* :113 */
    vko_base58check_encode(buf, 21, d->p2pkh_uncompressed, sizeof d->p2pkh_uncompressed);
    vko_segwit_encode("bc", 0, d->h160_c, 20, d->p2wpkh, sizeof d->p2wpkh); /* :116-118 */
    buf[0] = 0x80; memcpy(buf + 1, d->raw, 32); buf[33] = 0x01; /* :102-105,159-160 WIF */
    vko_base58check_encode(buf, 34, d->wif_compressed, sizeof d->wif_compressed);
    vko_base58check_encode(buf, 33, d->wif_uncompressed, sizeof d->wif_uncompressed);
    decimal_of_be(d->raw, d->private_key_decimal);     /* :121 */
    for (int i = 0; i < 256; i++) d->private_key_binary[i] = (d->raw[i / 8] >> (7 - i % 8)) & 1 ? '1' : '0'; /* :124-131 */
    d->private_key_binary[256] = 0;
    int lz = 0;                                        /* :134-141 */
    while (lz < 256 && !((d->raw[lz / 8] >> (7 - lz % 8)) & 1)) lz++;
    d->bit_length = (uint16_t)(256 - lz);
    int hw = 0;                                        /* :144 */
    for (int i = 0; i < 32; i++) hw += __builtin_popcount(d->raw[i]);
    d->hamming_weight = (uint16_t)hw;
    hex_encode(d->raw, 32, d->private_key_hex);        /* :147-148 */
    int l0 = 0;
    while (l0 < 64 && d->private_key_hex[l0] == '0') l0++;
    d->leading_zeros = (uint8_t)l0;
}

/* ------------------------------------------------------------------ matcher */
struct vko_matcher {
    char **slots;
    size_t cap, count;
};

static uint64_t fnv1a(const char *s)
{
    uint64_t h = 1469598103934665603ULL;
    for (; *s; s++) { h ^= (uint8_t)*s; h *= 1099511628211ULL; }
    return h;
}

vko_matcher *vko_matcher_new(void)
{
    vko_matcher *m = (vko_matcher *)calloc(1, sizeof *m);
    m->cap = 1024;
    m->slots = (char **)calloc(m->cap, sizeof(char *));
    return m;
}

void vko_matcher_free(vko_matcher *m)
{
    if (!m) return;
    for (size_t i = 0; i < m->cap; i++) free(m->slots[i]);
    free(m->slots);
    free(m);
}

static int matcher_has(const vko_matcher *m, const char *s)
{
    size_t i = fnv1a(s) & (m->cap - 1);
    while (m->slots[i]) {
        if (strcmp(m->slots[i], s) == 0) return 1;
        i = (i + 1) & (m->cap - 1);
    }
    return 0;
}

static void matcher_put(vko_matcher *m, char *owned)
{
    size_t i = fnv1a(owned) & (m->cap - 1);
    while (m->slots[i]) i = (i + 1) & (m->cap - 1);
    m->slots[i] = owned;
}

void vko_matcher_add_line(vko_matcher *m, const char *line)
{
    /* str::trim(): strip leading/trailing whitespace; skip empty and '#' (src/matcher.rs:50-56) */
    while (*line == ' ' || *line == '\t' || *line == '\r' || *line == '\n' || *line == '\f' || *line == '\v') line++;
    size_t n = strlen(line);
    while (n && (line[n - 1] == ' ' || line[n - 1] == '\t' || line[n - 1] == '\r' || line[n - 1] == '\n' || line[n - 1] == '\f' || line[n - 1] == '\v')) n--;
    if (n == 0 || line[0] == '#') return;
    char *s = (char *)malloc(n + 1);
    memcpy(s, line, n);
    s[n] = 0;
    if (matcher_has(m, s)) { free(s); return; } /* HashSet: duplicates collapse */
    if ((m->count + 1) * 2 > m->cap) {
        char **old = m->slots;
        size_t oc = m->cap;
        m->cap *= 2;
        m->slots = (char **)calloc(m->cap, sizeof(char *));
        for (size_t i = 0; i < oc; i++) if (old[i]) matcher_put(m, old[i]);
        free(old);
    }
    matcher_put(m, s);
    m->count++;
}

size_t vko_matcher_count(const vko_matcher *m) { return m->count; }

/* src/matcher.rs:72-98: p2pkh_compressed, then p2pkh_uncompressed, then p2wpkh; first wins */
int vko_matcher_check(const vko_matcher *m, const vko_derived *d)
{
    if (matcher_has(m, d->p2pkh_compressed)) return 0;
    if (matcher_has(m, d->p2pkh_uncompressed)) return 1;
    if (matcher_has(m, d->p2wpkh)) return 2;
    return -1;
}

int vko_matcher_check_h160(const vko_matcher *m, const uint8_t h160_c[20], const uint8_t h160_u[20])
{
    uint8_t buf[21];
    char s[64];
    buf[0] = 0;
    if (h160_c) {
        memcpy(buf + 1, h160_c, 20);
        vko_base58check_encode(buf, 21, s, sizeof s);
        if (matcher_has(m, s)) return 0;
    }
    if (h160_u) {
        memcpy(buf + 1, h160_u, 20);
        vko_base58check_encode(buf, 21, s, sizeof s);
        if (matcher_has(m, s)) return 1;
    }
    if (h160_c) {
        vko_segwit_encode("bc", 0, h160_c, 20, s, sizeof s);
        if (matcher_has(m, s)) return 2;
    }
    return -1;
}

/* -------------------------------------------------------------------- sweep */
void vko_digest_fold(uint64_t dg[4], const uint8_t h[20])
{
    uint32_t w[5];
    for (int i = 0; i < 5; i++) w[i] = (uint32_t)h[4 * i] | (uint32_t)h[4 * i + 1] << 8 | (uint32_t)h[4 * i + 2] << 16 | (uint32_t)h[4 * i + 3] << 24;
    uint64_t a = (uint64_t)w[0] | (uint64_t)w[1] << 32, b = (uint64_t)w[2] | (uint64_t)w[3] << 32;
    dg[0] ^= a;
    dg[1] ^= b;
    dg[2] ^= (uint64_t)w[4];
    dg[3] += a + b + w[4];
}

typedef struct {
    const vko_spec *spec;
    uint64_t lo, hi;
    uint32_t flags;
    const vko_matcher *m;
    vko_hit *hits;
    size_t hit_cap;
    pthread_mutex_t mu;
    uint64_t next_chunk;
    size_t nhits;
    vko_stats st;
} sweep_job;

static void *sweep_worker(void *arg)
{
    sweep_job *j = (sweep_job *)arg;
    uint32_t slots = vko_slots_per_input(j->spec);
    vko_stats loc;
    memset(&loc, 0, sizeof loc);
    uint8_t *bk = NULL;
    uint64_t *bseed = NULL;
    uint32_t *bsub = NULL;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        uint64_t c = j->next_chunk++;
        pthread_mutex_unlock(&j->mu);
        /* par_chunks(1000), src/source/range.rs:42 */
        uint64_t span = j->hi - j->lo; /* inclusive range: count-1 */
        if (c > span / 1000) break;
        uint64_t s0 = j->lo + c * 1000, s1 = s0 + 999;
        if (s1 > j->hi || s1 < s0) s1 = j->hi;
        if ((j->flags & VKO_F_BATCHED) && !(j->flags & VKO_F_FAITHFUL)) {
            /* same chunk, same order, same counters; only the k*G of the chunk's keys is shared work */
            size_t cap = (size_t)1000 * slots, n = 0;
            if (!bk) {
                bk = (uint8_t *)malloc(cap * (32 + 33 + 65));
                bseed = (uint64_t *)malloc(cap * sizeof *bseed);
                bsub = (uint32_t *)malloc(cap * sizeof *bsub);
            }
            uint8_t *b33 = bk + cap * 32, *b65 = b33 + cap * 33;
            for (uint64_t seed = s0;; seed++) {
                for (uint32_t sub = 0; sub < slots; sub++) {
                    uint8_t key[32];
                    if (!vko_transform_key(j->spec, seed, sub, key)) continue;
                    vko_normalize_key(key, bk + 32 * n);
                    bseed[n] = seed; bsub[n] = sub; n++;
                }
                if (seed == s1) break;
            }
            vko_pubkey_batch(bk, n, b33, b65);
            loc.keys += n;
            for (size_t i = 0; i < n; i++) {
                uint8_t hc[20], hu[20];
                const uint8_t *pc = NULL, *pu = NULL;
                if (j->flags & VKO_F_COMPRESSED) { vko_hash160(b33 + 33 * i, 33, hc); vko_digest_fold(loc.digest, hc); pc = hc; }
                if (j->flags & VKO_F_UNCOMPRESSED) { vko_hash160(b65 + 65 * i, 65, hu); vko_digest_fold(loc.digest, hu); pu = hu; }
                int kind = j->m ? vko_matcher_check_h160(j->m, pc, pu) : -1;
                if (kind >= 0) {
                    loc.matches++;
                    pthread_mutex_lock(&j->mu);
                    if (j->nhits < j->hit_cap) {
                        vko_hit *h = &j->hits[j->nhits];
                        h->seed = bseed[i]; h->sub_index = bsub[i]; h->addr_kind = (uint32_t)kind;
                        memcpy(h->privkey, bk + 32 * i, 32);
                        memcpy(h->h160, kind == 1 ? hu : hc, 20);
                    }
                    j->nhits++;
                    pthread_mutex_unlock(&j->mu);
                }
            }
            continue;
        }
        for (uint64_t seed = s0;; seed++) {
            for (uint32_t sub = 0; sub < slots; sub++) {
                uint8_t key[32];
                if (!vko_transform_key(j->spec, seed, sub, key)) continue;
                loc.keys++;
                uint8_t hc[20], hu[20], nk[32];
                int kind;
                if (j->flags & VKO_F_FAITHFUL) {
                    vko_derived d;
                    vko_derive(key, &d);
                    /* derive.rs:108-109 recomputes the public key twice more */
                    uint8_t t33[33], t65[65];
                    vko_pubkey(d.raw, t33, NULL);
                    vko_pubkey(d.raw, NULL, t65);
                    memcpy(hc, d.h160_c, 20); memcpy(hu, d.h160_u, 20); memcpy(nk, d.raw, 32);
                    kind = j->m ? vko_matcher_check(j->m, &d) : -1;
                    if (j->flags & VKO_F_COMPRESSED) vko_digest_fold(loc.digest, hc);
                    if (j->flags & VKO_F_UNCOMPRESSED) vko_digest_fold(loc.digest, hu);
                } else {
                    uint8_t p33[33], p65[65];
                    vko_normalize_key(key, nk);
                    vko_pubkey(nk, p33, p65);
                    const uint8_t *pc = NULL, *pu = NULL;
                    if (j->flags & VKO_F_COMPRESSED) { vko_hash160(p33, 33, hc); vko_digest_fold(loc.digest, hc); pc = hc; }
                    if (j->flags & VKO_F_UNCOMPRESSED) { vko_hash160(p65, 65, hu); vko_digest_fold(loc.digest, hu); pu = hu; }
                    kind = j->m ? vko_matcher_check_h160(j->m, pc, pu) : -1;
                }
                if (kind >= 0) {
                    loc.matches++;
                    pthread_mutex_lock(&j->mu);
                    if (j->nhits < j->hit_cap) {
                        vko_hit *h = &j->hits[j->nhits];
                        h->seed = seed; h->sub_index = sub; h->addr_kind = (uint32_t)kind;
                        memcpy(h->privkey, nk, 32);
                        memcpy(h->h160, kind == 1 ? hu : hc, 20);
                    }
                    j->nhits++;
                    pthread_mutex_unlock(&j->mu);
                }
            }
            if (seed == s1) break;
        }
    }
    free(bk); free(bseed); free(bsub);
    pthread_mutex_lock(&j->mu);
    j->st.keys += loc.keys; j->st.matches += loc.matches;
    j->st.digest[0] ^= loc.digest[0]; j->st.digest[1] ^= loc.digest[1];
    j->st.digest[2] ^= loc.digest[2]; j->st.digest[3] += loc.digest[3];
    pthread_mutex_unlock(&j->mu);
    return NULL;
}

static int hit_cmp(const void *a, const void *b)
{
    const vko_hit *x = (const vko_hit *)a, *y = (const vko_hit *)b;
    if (x->seed != y->seed) return x->seed < y->seed ? -1 : 1;
    if (x->sub_index != y->sub_index) return x->sub_index < y->sub_index ? -1 : 1;
    return 0;
}

int vko_sweep(const vko_spec *spec, uint64_t lo, uint64_t hi, uint32_t flags, const vko_matcher *m,
              vko_hit *hits, size_t hit_cap, vko_stats *out, int nthreads)
{
    if (hi < lo || !out) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    { uint8_t k[32] = {0}, p[33]; k[31] = 1; vko_pubkey(k, p, NULL); if (flags & VKO_F_BATCHED) vko_pubkey_batch(k, 1, p, NULL); } /* build the tables before threads start */
    sweep_job j;
    memset(&j, 0, sizeof j);
    j.spec = spec; j.lo = lo; j.hi = hi; j.flags = flags; j.m = m; j.hits = hits; j.hit_cap = hits ? hit_cap : 0;
    pthread_mutex_init(&j.mu, NULL);
    pthread_t th[256];
    for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, sweep_worker, &j);
    for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    pthread_mutex_destroy(&j.mu);
    j.st.inputs = hi - lo + 1; /* src/source/range.rs:32 */
    size_t n = j.nhits < j.hit_cap ? j.nhits : j.hit_cap;
    if (n) qsort(hits, n, sizeof *hits, hit_cmp);
    *out = j.st;
    return 0;
}
