/*
 * secp256k1.c -- ORACLE (test infrastructure, see vko.h): the curve arithmetic the
 * reference obtains from the un-vendored crate secp256k1 0.29.1 / secp256k1-sys 0.10.1
 * (C libsecp256k1) at src/derive.rs:79,92,97,108,109.  Restated from SEC2 (curve
 * parameters) and textbook Jacobian formulas; deliberately simple.  Two k*G paths:
 * vko_pubkey_slow (double-and-add, every special case handled by the generic adder)
 * and vko_pubkey (8-bit fixed windows, mixed additions) -- tests check one against
 * the other, against python big-int EC, and against the reference's pinned vectors
 * (src/derive.rs:179-200, src/electrum.rs:218-221).
 */
#include "vko.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe; /* little-endian limbs, always fully reduced < p */

static const fe FE_P = {{0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}};
static const uint64_t ORDER_N[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL};
static const fe GX = {{0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL, 0x79BE667EF9DCBBACULL}};
static const fe GY = {{0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL, 0x483ADA7726A3C465ULL}};
#define P_C 0x1000003D1ULL /* 2^256 mod p */

static int ge4(const uint64_t a[4], const uint64_t b[4])
{
    for (int i = 3; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static uint64_t sub4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4])
{
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static uint64_t add4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4])
{
    uint64_t carry = 0;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a[i] + b[i] + carry;
        r[i] = (uint64_t)s;
        carry = (uint64_t)(s >> 64);
    }
    return carry;
}
static int fe_is_zero(const fe *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }

static void fe_add(fe *r, const fe *a, const fe *b)
{
    uint64_t c = add4(r->v, a->v, b->v);
    if (c || ge4(r->v, FE_P.v)) sub4(r->v, r->v, FE_P.v);
}
static void fe_sub(fe *r, const fe *a, const fe *b)
{
    if (sub4(r->v, a->v, b->v)) add4(r->v, r->v, FE_P.v);
}
static void fe_mul(fe *r, const fe *a, const fe *b)
{
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {
            u128 m = (u128)a->v[i] * b->v[j] + t[i + j] + carry;
            t[i + j] = (uint64_t)m;
            carry = (uint64_t)(m >> 64);
        }
        t[i + 4] = carry;
    }
    /* fold high half: 2^256 == P_C (mod p) */
    uint64_t lo[5];
    uint64_t carry = 0;
    for (int i = 0; i < 4; i++) {
        u128 m = (u128)t[i + 4] * P_C + t[i] + carry;
        lo[i] = (uint64_t)m;
        carry = (uint64_t)(m >> 64);
    }
    lo[4] = carry;
    u128 m = (u128)lo[4] * P_C + lo[0];
    r->v[0] = (uint64_t)m;
    carry = (uint64_t)(m >> 64);
    for (int i = 1; i < 4; i++) {
        u128 s = (u128)lo[i] + carry;
        r->v[i] = (uint64_t)s;
        carry = (uint64_t)(s >> 64);
    }
    if (carry) { /* wrapped past 2^256 once more: add P_C (cannot carry again) */
        uint64_t c2[4] = {P_C, 0, 0, 0};
        add4(r->v, r->v, c2);
    }
    if (ge4(r->v, FE_P.v)) sub4(r->v, r->v, FE_P.v);
}
static void fe_sqr(fe *r, const fe *a) { fe_mul(r, a, a); }
static void fe_inv(fe *r, const fe *a)
{
    /* a^(p-2) by square-and-multiply over the bits of p-2 */
    uint64_t e[4];
    uint64_t two[4] = {2, 0, 0, 0};
    sub4(e, FE_P.v, two);
    fe acc = {{1, 0, 0, 0}};
    for (int i = 255; i >= 0; i--) {
        fe_sqr(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fe_mul(&acc, &acc, a);
    }
    *r = acc;
}
static void fe_from_be(fe *r, const uint8_t b[32])
{
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int j = 0; j < 8; j++) v = (v << 8) | b[8 * (3 - i) + j];
        r->v[i] = v;
    }
}
static void fe_to_be(uint8_t b[32], const fe *a)
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) b[8 * (3 - i) + j] = (uint8_t)(a->v[i] >> (56 - 8 * j));
}

/* ------------------------------------------------------------------ points */
typedef struct { fe x, y, z; int inf; } jac;
typedef struct { fe x, y; } aff;

static void jac_double(jac *r, const jac *p)
{
    if (p->inf || fe_is_zero(&p->y)) { r->inf = 1; return; }
    fe a, b, c, d, e, f, t, x3, y3, z3;
    fe_sqr(&a, &p->x);            /* A = X^2 */
    fe_sqr(&b, &p->y);            /* B = Y^2 */
    fe_sqr(&c, &b);               /* C = B^2 */
    fe_add(&t, &p->x, &b); fe_sqr(&t, &t); fe_sub(&t, &t, &a); fe_sub(&t, &t, &c);
    fe_add(&d, &t, &t);           /* D = 2((X+B)^2 - A - C) */
    fe_add(&e, &a, &a); fe_add(&e, &e, &a); /* E = 3A */
    fe_sqr(&f, &e);               /* F = E^2 */
    fe_sub(&x3, &f, &d); fe_sub(&x3, &x3, &d);
    fe_sub(&t, &d, &x3); fe_mul(&y3, &e, &t);
    fe_add(&c, &c, &c); fe_add(&c, &c, &c); fe_add(&c, &c, &c); /* 8C */
    fe_sub(&y3, &y3, &c);
    fe_mul(&z3, &p->y, &p->z); fe_add(&z3, &z3, &z3);
    r->x = x3; r->y = y3; r->z = z3; r->inf = 0;
}

/* generic Jacobian + affine, all special cases */
static void jac_add_aff(jac *r, const jac *p, const aff *q)
{
    if (p->inf) { r->x = q->x; r->y = q->y; r->z = (fe){{1, 0, 0, 0}}; r->inf = 0; return; }
    fe z2, u2, s2, h, rr, h2, h3, v, t, x3, y3, z3;
    fe_sqr(&z2, &p->z);
    fe_mul(&u2, &q->x, &z2);
    fe_mul(&s2, &q->y, &z2); fe_mul(&s2, &s2, &p->z);
    fe_sub(&h, &u2, &p->x);
    fe_sub(&rr, &s2, &p->y);
    if (fe_is_zero(&h)) {
        if (fe_is_zero(&rr)) { jac_double(r, p); return; }
        r->inf = 1; return;
    }
    fe_sqr(&h2, &h); fe_mul(&h3, &h2, &h);
    fe_mul(&v, &p->x, &h2);
    fe_sqr(&x3, &rr); fe_sub(&x3, &x3, &h3); fe_sub(&x3, &x3, &v); fe_sub(&x3, &x3, &v);
    fe_sub(&t, &v, &x3); fe_mul(&y3, &rr, &t);
    fe_mul(&t, &p->y, &h3); fe_sub(&y3, &y3, &t);
    fe_mul(&z3, &p->z, &h);
    r->x = x3; r->y = y3; r->z = z3; r->inf = 0;
}

static void jac_to_aff(aff *r, const jac *p)
{
    fe zi, zi2, zi3;
    fe_inv(&zi, &p->z);
    fe_sqr(&zi2, &zi); fe_mul(&zi3, &zi2, &zi);
    fe_mul(&r->x, &p->x, &zi2);
    fe_mul(&r->y, &p->y, &zi3);
}

static void serialize(const aff *a, uint8_t pub33[33], uint8_t pub65[65])
{
    uint8_t xb[32], yb[32];
    fe_to_be(xb, &a->x); fe_to_be(yb, &a->y);
    if (pub33) { pub33[0] = 0x02 | (yb[31] & 1); memcpy(pub33 + 1, xb, 32); }
    if (pub65) { pub65[0] = 0x04; memcpy(pub65 + 1, xb, 32); memcpy(pub65 + 33, yb, 32); }
}

/* ----------------------------------------------------------------- scalars */
static void sc_from_be(uint64_t k[4], const uint8_t b[32])
{
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int j = 0; j < 8; j++) v = (v << 8) | b[8 * (3 - i) + j];
        k[i] = v;
    }
}
static void sc_to_be(uint8_t b[32], const uint64_t k[4])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) b[8 * (3 - i) + j] = (uint8_t)(k[i] >> (56 - 8 * j));
}

int vko_scalar_valid(const uint8_t kb[32])
{
    uint64_t k[4];
    sc_from_be(k, kb);
    if ((k[0] | k[1] | k[2] | k[3]) == 0) return 0;
    return !ge4(k, ORDER_N);
}

/* src/derive.rs:79-93.  2^256 < 2n, so "mod n" is at most one subtraction. */
void vko_normalize_key(const uint8_t in[32], uint8_t out[32])
{
    uint64_t k[4];
    sc_from_be(k, in);
    if (ge4(k, ORDER_N)) sub4(k, k, ORDER_N);
    if ((k[0] | k[1] | k[2] | k[3]) == 0) k[0] = 1;
    sc_to_be(out, k);
}

int vko_scalar_add(const uint8_t a[32], const uint8_t b[32], uint8_t out[32])
{
    uint64_t x[4], y[4], r[4];
    sc_from_be(x, a); sc_from_be(y, b);
    if (ge4(x, ORDER_N)) sub4(x, x, ORDER_N);
    if (ge4(y, ORDER_N)) sub4(y, y, ORDER_N);
    uint64_t c = add4(r, x, y);
    if (c || ge4(r, ORDER_N)) sub4(r, r, ORDER_N);
    sc_to_be(out, r);
    return (r[0] | r[1] | r[2] | r[3]) != 0;
}

/* a * b mod n by double-and-add over the bits of b (SecretKey::mul_tweak); returns 0 if the product is zero */
int vko_scalar_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32])
{
    uint64_t x[4], y[4], r[4] = {0, 0, 0, 0};
    sc_from_be(x, a); sc_from_be(y, b);
    if (ge4(x, ORDER_N)) sub4(x, x, ORDER_N);
    for (int bit = 255; bit >= 0; bit--) {
        uint64_t c = add4(r, r, r);
        if (c || ge4(r, ORDER_N)) sub4(r, r, ORDER_N);
        if ((y[bit / 64] >> (bit % 64)) & 1) {
            c = add4(r, r, x);
            if (c || ge4(r, ORDER_N)) sub4(r, r, ORDER_N);
        }
    }
    sc_to_be(out, r);
    return (r[0] | r[1] | r[2] | r[3]) != 0;
}

/* ------------------------------------------------------------------- k * G */
void vko_pubkey_slow(const uint8_t key[32], uint8_t pub33[33], uint8_t pub65[65])
{
    uint64_t k[4];
    sc_from_be(k, key);
    aff g = {GX, GY};
    jac acc; acc.inf = 1;
    for (int i = 255; i >= 0; i--) {
        jac_double(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) jac_add_aff(&acc, &acc, &g);
    }
    aff a;
    jac_to_aff(&a, &acc);
    serialize(&a, pub33, pub65);
}

static aff *g_table; /* [32][256]: (d * 256^w) * G, entry d = 0 unused */
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void build_table(void)
{
    g_table = (aff *)calloc(32 * 256, sizeof(aff));
    aff base = {GX, GY};
    for (int w = 0; w < 32; w++) {
        jac acc; acc.inf = 1;
        for (int d = 1; d < 256; d++) {
            jac_add_aff(&acc, &acc, &base);
            jac_to_aff(&g_table[w * 256 + d], &acc);
        }
        /* next base = 256 * base */
        jac_add_aff(&acc, &acc, &base);
        jac_to_aff(&base, &acc);
    }
}

void vko_pubkey(const uint8_t key[32], uint8_t pub33[33], uint8_t pub65[65])
{
    pthread_once(&g_once, build_table);
    jac acc; acc.inf = 1;
    for (int w = 0; w < 32; w++) {
        unsigned d = key[31 - w];
        if (d) jac_add_aff(&acc, &acc, &g_table[w * 256 + d]);
    }
    aff a;
    jac_to_aff(&a, &acc);
    serialize(&a, pub33, pub65);
}

/* Third k*G path, for whole chunks of keys (the "lean" CPU baseline and the generator of the
 * full-size golden digests, tests/golden/make_milksad_2p32.py): 16-bit fixed windows (16 x 65536
 * affine points, 64 MB) and one field inversion per chunk (Montgomery's simultaneous inversion).
 * Same generic adder as the other two paths, so every special case is still handled; checked
 * against vko_pubkey_slow / vko_pubkey in tests/test_oracle_derive.py. */
static aff *g_table16; /* [16][65536]: (d * 65536^w) * G, entry d = 0 unused */
static pthread_once_t g_once16 = PTHREAD_ONCE_INIT;

/* z[i] <- 1/z[i] for i < n (none zero); scratch holds n field elements */
static void fe_batch_inv(fe *z, size_t n, fe *scratch)
{
    if (!n) return;
    scratch[0] = z[0];
    for (size_t i = 1; i < n; i++) fe_mul(&scratch[i], &scratch[i - 1], &z[i]);
    fe inv;
    fe_inv(&inv, &scratch[n - 1]);
    for (size_t i = n - 1; i > 0; i--) {
        fe t;
        fe_mul(&t, &inv, &scratch[i - 1]);
        fe_mul(&inv, &inv, &z[i]);
        z[i] = t;
    }
    z[0] = inv;
}

static void build_table16(void)
{
    const size_t N = 65536;
    g_table16 = (aff *)calloc(16 * N, sizeof(aff));
    jac *row = (jac *)malloc(N * sizeof(jac));
    fe *zs = (fe *)malloc(N * sizeof(fe)), *scr = (fe *)malloc(N * sizeof(fe));
    aff base = {GX, GY};
    for (int w = 0; w < 16; w++) {
        jac acc; acc.inf = 1;
        for (size_t d = 1; d < N; d++) {
            jac_add_aff(&acc, &acc, &base);
            row[d] = acc;
            zs[d] = acc.z;
        }
        fe_batch_inv(zs + 1, N - 1, scr);
        for (size_t d = 1; d < N; d++) {
            fe zi2, zi3;
            fe_sqr(&zi2, &zs[d]); fe_mul(&zi3, &zi2, &zs[d]);
            fe_mul(&g_table16[(size_t)w * N + d].x, &row[d].x, &zi2);
            fe_mul(&g_table16[(size_t)w * N + d].y, &row[d].y, &zi3);
        }
        jac_add_aff(&acc, &acc, &base); /* 65536 * base */
        jac_to_aff(&base, &acc);
    }
    free(row); free(zs); free(scr);
}

/* keys: n x 32 B normalised scalars; pub33s / pub65s: n x 33 / n x 65 B (either may be NULL) */
void vko_pubkey_batch(const uint8_t *keys, size_t n, uint8_t *pub33s, uint8_t *pub65s)
{
    pthread_once(&g_once16, build_table16);
    if (!n) return;
    jac *acc = (jac *)malloc(n * sizeof(jac));
    fe *zs = (fe *)malloc(n * sizeof(fe)), *scr = (fe *)malloc(n * sizeof(fe));
    for (size_t i = 0; i < n; i++) {
        const uint8_t *k = keys + 32 * i;
        acc[i].inf = 1;
        for (int w = 0; w < 16; w++) {
            unsigned d = (unsigned)k[31 - 2 * w] | (unsigned)k[30 - 2 * w] << 8;
            if (d) jac_add_aff(&acc[i], &acc[i], &g_table16[(size_t)w * 65536 + d]);
        }
        zs[i] = acc[i].z; /* k in [1, n) => never the point at infinity */
    }
    fe_batch_inv(zs, n, scr);
    for (size_t i = 0; i < n; i++) {
        fe zi2, zi3;
        aff a;
        fe_sqr(&zi2, &zs[i]); fe_mul(&zi3, &zi2, &zs[i]);
        fe_mul(&a.x, &acc[i].x, &zi2);
        fe_mul(&a.y, &acc[i].y, &zi3);
        serialize(&a, pub33s ? pub33s + 33 * i : NULL, pub65s ? pub65s + 65 * i : NULL);
    }
    free(acc); free(zs); free(scr);
}

void vko_derive_h160(const uint8_t key[32], uint8_t h160_c[20], uint8_t h160_u[20])
{
    uint8_t k[32], p33[33], p65[65];
    vko_normalize_key(key, k);
    vko_pubkey(k, p33, p65);
    if (h160_c) vko_hash160(p33, 33, h160_c);
    if (h160_u) vko_hash160(p65, 65, h160_u);
}

/* used by tests to poke at field arithmetic directly */
void vko_fe_mul_be(const uint8_t a[32], const uint8_t b[32], uint8_t out[32])
{
    fe x, y, r;
    fe_from_be(&x, a); fe_from_be(&y, b);
    if (ge4(x.v, FE_P.v)) sub4(x.v, x.v, FE_P.v);
    if (ge4(y.v, FE_P.v)) sub4(y.v, y.v, FE_P.v);
    fe_mul(&r, &x, &y);
    fe_to_be(out, &r);
}
void vko_fe_inv_be(const uint8_t a[32], uint8_t out[32])
{
    fe x, r;
    fe_from_be(&x, a);
    if (ge4(x.v, FE_P.v)) sub4(x.v, x.v, FE_P.v);
    fe_inv(&r, &x);
    fe_to_be(out, &r);
}
