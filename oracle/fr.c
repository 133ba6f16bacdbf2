/*
 * oracle/fr.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).  See fr.h for what
 * this restates, where the algorithm comes from and how it is pinned.
 */
#include "fr.h"
#include <string.h>

typedef unsigned __int128 u128;

const fr_t FR_MODULUS = {{0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull}};
const fr_t FR_ZERO    = {{0, 0, 0, 0}};
const fr_t FR_ONE     = {{0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full}};
const fr_t FR_R2      = {{0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full, 0x0748d9d99f59ff11ull}};
static const uint64_t FR_INV = 0xfffffffeffffffffull; /* -q^{-1} mod 2^64 */

/* a + b + carry -> (lo, carry) */
static inline uint64_t adc(uint64_t a, uint64_t b, uint64_t *carry) {
    u128 t = (u128)a + b + *carry;
    *carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}
/* a - b - borrow -> (lo, borrow in {0,1}) */
static inline uint64_t sbb(uint64_t a, uint64_t b, uint64_t *borrow) {
    u128 t = (u128)a - b - *borrow;
    *borrow = (uint64_t)(t >> 64) & 1;
    return (uint64_t)t;
}
/* a + b*c + carry -> (lo, carry) */
static inline uint64_t mac(uint64_t a, uint64_t b, uint64_t c, uint64_t *carry) {
    u128 t = (u128)a + (u128)b * c + *carry;
    *carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}

/* r - q if r >= q (r < 2q on entry) */
static fr_t sub_modulus_if_ge(const uint64_t r[4], uint64_t top) {
    uint64_t bw = 0, d[4];
    for (int i = 0; i < 4; i++) d[i] = sbb(r[i], FR_MODULUS.l[i], &bw);
    /* borrow past the (virtual) top limb means r < q: keep r */
    uint64_t keep = (top < bw) ? 1 : 0;
    fr_t o;
    for (int i = 0; i < 4; i++) o.l[i] = keep ? r[i] : d[i];
    return o;
}

fr_t fr_add(fr_t a, fr_t b) {
    uint64_t c = 0, r[4];
    for (int i = 0; i < 4; i++) r[i] = adc(a.l[i], b.l[i], &c);
    return sub_modulus_if_ge(r, c);
}

fr_t fr_sub(fr_t a, fr_t b) {
    uint64_t bw = 0, d[4];
    for (int i = 0; i < 4; i++) d[i] = sbb(a.l[i], b.l[i], &bw);
    uint64_t mask = 0 - bw, c = 0;
    fr_t o;
    for (int i = 0; i < 4; i++) o.l[i] = adc(d[i], FR_MODULUS.l[i] & mask, &c);
    return o;
}

fr_t fr_neg(fr_t a) {
    uint64_t bw = 0, d[4];
    for (int i = 0; i < 4; i++) d[i] = sbb(FR_MODULUS.l[i], a.l[i], &bw);
    uint64_t nz = (a.l[0] | a.l[1] | a.l[2] | a.l[3]) ? ~0ull : 0;
    fr_t o;
    for (int i = 0; i < 4; i++) o.l[i] = d[i] & nz;
    return o;
}

static fr_t montgomery_reduce(uint64_t t[8]) {
    uint64_t carry2 = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t k = t[i] * FR_INV, carry = 0;
        (void)mac(t[i], k, FR_MODULUS.l[0], &carry);
        for (int j = 1; j < 4; j++) t[i + j] = mac(t[i + j], k, FR_MODULUS.l[j], &carry);
        t[i + 4] = adc(t[i + 4], carry2, &carry);
        carry2 = carry;
    }
    return sub_modulus_if_ge(t + 4, carry2);
}

fr_t fr_mul(fr_t a, fr_t b) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) t[i + j] = mac(t[i + j], a.l[i], b.l[j], &carry);
        t[i + 4] = carry;
    }
    return montgomery_reduce(t);
}

fr_t fr_square(fr_t a) { return fr_mul(a, a); }

fr_t fr_from_raw(const uint64_t v[4]) {
    fr_t x;
    memcpy(x.l, v, sizeof x.l);
    return fr_mul(x, FR_R2);
}

fr_t fr_from_u64(uint64_t v) {
    uint64_t raw[4] = {v, 0, 0, 0};
    return fr_from_raw(raw);
}

fr_t fr_reduce(fr_t a) {
    uint64_t t[8] = {a.l[0], a.l[1], a.l[2], a.l[3], 0, 0, 0, 0};
    return montgomery_reduce(t);
}

void fr_to_bytes(fr_t a, uint8_t out[32]) {
    fr_t c = fr_reduce(a);
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(c.l[i] >> (8 * b));
}

void fr_divn(fr_t *a, uint32_t n) {
    if (n >= 256) { *a = FR_ZERO; return; }
    while (n >= 64) {
        uint64_t t = 0;
        for (int i = 3; i >= 0; i--) { uint64_t s = a->l[i]; a->l[i] = t; t = s; }
        n -= 64;
    }
    if (n > 0) {
        uint64_t t = 0;
        for (int i = 3; i >= 0; i--) {
            uint64_t t2 = a->l[i] << (64 - n);
            a->l[i] = (a->l[i] >> n) | t;
            t = t2;
        }
    }
}

int fr_cmp_raw(fr_t a, fr_t b) {
    for (int i = 3; i >= 0; i--) {
        if (a.l[i] > b.l[i]) return 1;
        if (a.l[i] < b.l[i]) return -1;
    }
    return 0;
}

int fr_eq(fr_t a, fr_t b) { return fr_cmp_raw(a, b) == 0; }
int fr_is_zero(fr_t a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }

fr_t fr_pow(fr_t base, const uint64_t by[4]) {
    /* fixed 256 x (square, multiply, select): the cost structure the
     * reference pays per ladder bit at src/range.rs:146 */
    fr_t res = FR_ONE;
    for (int e = 3; e >= 0; e--) {
        for (int i = 63; i >= 0; i--) {
            res = fr_square(res);
            fr_t tmp = fr_mul(res, base);
            if ((by[e] >> i) & 1) res = tmp;
        }
    }
    return res;
}

fr_t fr_pow_of_2(uint64_t by) {
    fr_t two = fr_from_u64(2), res = FR_ONE;
    for (int i = 63; i >= 0; i--) {
        res = fr_square(res);
        fr_t tmp = fr_mul(res, two);
        if ((by >> i) & 1) res = tmp;
    }
    return res;
}

int fr_invert(fr_t a, fr_t *out) {
    /* a^(q-2); upstream uses an addition chain, the value is the same */
    static const uint64_t QM2[4] = {0xfffffffeffffffffull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    *out = fr_pow(a, QM2);
    return !fr_is_zero(a);
}
