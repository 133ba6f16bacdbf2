// fr.hpp -- BLS12-381 scalar field (BlsScalar) for the engine: 4 x 64-bit limbs,
// Montgomery form x*R mod q with R = 2^256, values always fully reduced in
// [0, q) so that equal field elements have identical limbs (the property the
// limb-for-limb parity tests rely on).  One source for host and gfx950 device
// code.  Replaces, for this path, the slice of dusk-bls12_381 `Scalar` the
// reference calls (call sites: /root/reference/src/range.rs:63,69,87,94,102,
// 130,146,152,163,174-177,187; src/scalar.rs:47,73,91,113,121-126).
#pragma once

#include "experiment.hpp"  // (first: it looks at build options before any header gives them a default)

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PG_HD __host__ __device__ __forceinline__
#else
#define PG_HD inline
#endif

namespace pg {

struct Fr {
    uint64_t l[4];
};

// q
#define PG_Q0 0xffffffff00000001ull
#define PG_Q1 0x53bda402fffe5bfeull
#define PG_Q2 0x3339d80809a1d805ull
#define PG_Q3 0x73eda753299d7d48ull
// -q^{-1} mod 2^64
#define PG_INV 0xfffffffeffffffffull

PG_HD Fr fr_modulus() { return Fr{{PG_Q0, PG_Q1, PG_Q2, PG_Q3}}; }
PG_HD Fr fr_zero() { return Fr{{0, 0, 0, 0}}; }
// mont(1) = R mod q
PG_HD Fr fr_one() { return Fr{{0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full}}; }
// mont(-1)
PG_HD Fr fr_neg_one() { return Fr{{0xfffffffd00000003ull, 0xfb38ec08fffb13fcull, 0x99ad88181ce5880full, 0x5bc8f5f97cd877d8ull}}; }
// R^2 mod q
PG_HD Fr fr_r2() { return Fr{{0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full, 0x0748d9d99f59ff11ull}}; }

PG_HD bool fr_is_zero(const Fr &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
PG_HD bool fr_eq(const Fr &a, const Fr &b) {
    return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1]) | (a.l[2] ^ b.l[2]) | (a.l[3] ^ b.l[3])) == 0;
}

typedef unsigned __int128 pg_u128;

PG_HD uint64_t adc64(uint64_t a, uint64_t b, uint64_t &carry) {
    pg_u128 t = (pg_u128)a + b + carry;
    carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}
PG_HD uint64_t sbb64(uint64_t a, uint64_t b, uint64_t &borrow) {
    pg_u128 t = (pg_u128)a - b - borrow;
    borrow = (uint64_t)(t >> 64) & 1;
    return (uint64_t)t;
}
PG_HD uint64_t mac64(uint64_t a, uint64_t b, uint64_t c, uint64_t &carry) {
    pg_u128 t = (pg_u128)a + (pg_u128)b * c + carry;
    carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}

// r (with a virtual 5th limb `top`, value < 2q) -> r mod q
PG_HD Fr fr_final_sub(const uint64_t r[4], uint64_t top) {
    uint64_t bw = 0;
    uint64_t d0 = sbb64(r[0], PG_Q0, bw), d1 = sbb64(r[1], PG_Q1, bw), d2 = sbb64(r[2], PG_Q2, bw),
             d3 = sbb64(r[3], PG_Q3, bw);
    bool keep = top < bw;  // borrowed past the top: r < q
    return Fr{{keep ? r[0] : d0, keep ? r[1] : d1, keep ? r[2] : d2, keep ? r[3] : d3}};
}

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 forms of the carry chains: eight 32-bit words, ONE asm statement each (seen by the device pass only; host code
// that names them in a .hip file is compiled from the generic forms below in the host pass).  Written through 128-bit integers (the host
// forms below) the compiler emulates every 64-bit carry step with three or four 32-bit instructions and pads every
// carry-in with an s_nop: 75 instructions for a subtraction that is 25.
#define PG_W(x, i) ((uint32_t)((x).l[(i) >> 1] >> (32 * ((i) & 1))))
PG_HD Fr fr_from_words(const uint32_t (&w)[8]) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 4; i++) o.l[i] = ((uint64_t)w[2 * i + 1] << 32) | w[2 * i];
    return o;
}

// a - b mod q: subtract, then add q back under the mask of the final borrow
PG_HD Fr fr_sub(const Fr &a, const Fr &b) {
    uint32_t d[8], m;
    asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
        "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
        "v_cndmask_b32 %8, 0, -1, vcc\n\t"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(m)
        : "v"(PG_W(a, 0)), "v"(PG_W(a, 1)), "v"(PG_W(a, 2)), "v"(PG_W(a, 3)), "v"(PG_W(a, 4)), "v"(PG_W(a, 5)), "v"(PG_W(a, 6)),
          "v"(PG_W(a, 7)), "v"(PG_W(b, 0)), "v"(PG_W(b, 1)), "v"(PG_W(b, 2)), "v"(PG_W(b, 3)), "v"(PG_W(b, 4)), "v"(PG_W(b, 5)),
          "v"(PG_W(b, 6)), "v"(PG_W(b, 7))
        : "vcc");
    // q = {0x00000001, 0xffffffff, 0xfffe5bfe, 0x53bda402, 0x09a1d805, 0x3339d808, 0x299d7d48, 0x73eda753}
    uint32_t t[7];
    t[0] = m & 1u;
    t[1] = m & 0xfffe5bfeu; t[2] = m & 0x53bda402u; t[3] = m & 0x09a1d805u; t[4] = m & 0x3339d808u; t[5] = m & 0x299d7d48u;
    t[6] = m & 0x73eda753u;
    asm("v_add_co_u32 %0, vcc, %0, %8\n\t"
        "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %10, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %11, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %4, %12, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %5, %13, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %6, %14, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %7, %15, vcc\n\t"
        : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
        : "v"(t[0]), "v"(m), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6])
        : "vcc");
    return fr_from_words(d);
}

// r (eight words) + top * 2^256 < 2q  ->  r mod q: subtract q, keep r if that borrowed past the top
PG_HD Fr fr_final_sub_words(const uint32_t (&r)[8], uint32_t top) {
    uint32_t d[8], keep;
    const uint32_t q2 = 0xfffe5bfeu, q3 = 0x53bda402u, q4 = 0x09a1d805u, q5 = 0x3339d808u, q6 = 0x299d7d48u, q7 = 0x73eda753u;
    asm("v_sub_co_u32 %0, vcc, %9, 1\n\t"
        "v_subb_co_u32 %1, vcc, %10, -1, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %11, %17, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %12, %18, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %13, %19, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %14, %20, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %15, %21, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %16, %22, vcc\n\t"
        "v_cndmask_b32 %8, 0, -1, vcc\n\t"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(keep)
        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(q2), "v"(q3), "v"(q4),
          "v"(q5), "v"(q6), "v"(q7)
        : "vcc");
    const bool below = keep != 0 && top == 0;  // borrowed past the top: r + top * 2^256 < q
    uint32_t o[8];
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = below ? r[i] : d[i];
    return fr_from_words(o);
}

PG_HD Fr fr_add(const Fr &a, const Fr &b) {
    uint32_t r[8], top;
    asm("v_add_co_u32 %0, vcc, %9, %17\n\t"
        "v_addc_co_u32 %1, vcc, %10, %18, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %12, %20, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %13, %21, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %14, %22, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %15, %23, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %16, %24, vcc\n\t"
        "v_cndmask_b32 %8, 0, 1, vcc\n\t"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(top)
        : "v"(PG_W(a, 0)), "v"(PG_W(a, 1)), "v"(PG_W(a, 2)), "v"(PG_W(a, 3)), "v"(PG_W(a, 4)), "v"(PG_W(a, 5)), "v"(PG_W(a, 6)),
          "v"(PG_W(a, 7)), "v"(PG_W(b, 0)), "v"(PG_W(b, 1)), "v"(PG_W(b, 2)), "v"(PG_W(b, 3)), "v"(PG_W(b, 4)), "v"(PG_W(b, 5)),
          "v"(PG_W(b, 6)), "v"(PG_W(b, 7))
        : "vcc");
    return fr_final_sub_words(r, top);
}
#undef PG_W
#else
PG_HD Fr fr_add(const Fr &a, const Fr &b) {
    uint64_t c = 0, r[4];
    for (int i = 0; i < 4; i++) r[i] = adc64(a.l[i], b.l[i], c);
    return fr_final_sub(r, c);
}

PG_HD Fr fr_sub(const Fr &a, const Fr &b) {
    uint64_t bw = 0, d[4];
    for (int i = 0; i < 4; i++) d[i] = sbb64(a.l[i], b.l[i], bw);
    uint64_t m = 0 - bw, c = 0;
    Fr o;
    o.l[0] = adc64(d[0], PG_Q0 & m, c);
    o.l[1] = adc64(d[1], PG_Q1 & m, c);
    o.l[2] = adc64(d[2], PG_Q2 & m, c);
    o.l[3] = adc64(d[3], PG_Q3 & m, c);
    return o;
}
#endif

PG_HD Fr fr_neg(const Fr &a) {
    uint64_t bw = 0;
    uint64_t d0 = sbb64(PG_Q0, a.l[0], bw), d1 = sbb64(PG_Q1, a.l[1], bw), d2 = sbb64(PG_Q2, a.l[2], bw),
             d3 = sbb64(PG_Q3, a.l[3], bw);
    uint64_t nz = fr_is_zero(a) ? 0 : ~0ull;
    return Fr{{d0 & nz, d1 & nz, d2 & nz, d3 & nz}};
}

// 8-limb product -> Montgomery reduction (4 rounds) -> final conditional subtraction
PG_HD Fr fr_mont_reduce(uint64_t t[8]) {
    const uint64_t Q[4] = {PG_Q0, PG_Q1, PG_Q2, PG_Q3};
    uint64_t carry2 = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t k = t[i] * PG_INV, carry = 0;
        (void)mac64(t[i], k, Q[0], carry);
        for (int j = 1; j < 4; j++) t[i + j] = mac64(t[i + j], k, Q[j], carry);
        t[i + 4] = adc64(t[i + 4], carry2, carry);
        carry2 = carry;
    }
    return fr_final_sub(t + 4, carry2);
}

// generic 4 x 64-bit schoolbook product + reduction (host code; also the reference the device form is tested against)
PG_HD Fr fr_mul64(const Fr &a, const Fr &b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) t[i + j] = mac64(t[i + j], a.l[i], b.l[j], carry);
        t[i + 4] = carry;
    }
    return fr_mont_reduce(t);
}

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 form: 8 x 32-bit limbs, product scanning (FIPS).  CDNA4 has no 64x64->128 vector multiply; its widest
// integer multiply-add is v_mad_u64_u32 (32 x 32 + 64 -> 64, carry-out to VCC).  Column by column a 96-bit
// accumulator {ex:hi:lo} takes one v_mad_u64_u32 + one v_addc per 32 x 32 product; the Montgomery quotient digit of
// column k is m_k = -lo because q = 1 (mod 2^32) (so -q^-1 mod 2^32 = 0xffffffff and m_k * q_0 = m_k needs no
// multiply).  120 multiply-adds per product (measured 1.2e11 products/s per MI355X, 1.85x the generic code, whose
// 64-bit products are re-assembled from 32-bit pieces and whose reduction loop stays rolled: tools/fr_mul_bench.hip).
// One asm statement per run of products of a column (at most eight): between two asm statements the compiler has to assume
// that the second reads the VCC the first wrote and pads with an s_nop; per product that was 360 of them in a loop body
// with three multiplications, a quarter of its instructions.
#define PG_MAC(p, q) "v_mad_u64_u32 %0, vcc, %" #p ", %" #q ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
// (the first product of a COLUMN starts the third word afresh: 0 + 0 + carry, so that nothing has to zero it)
#define PG_MAC0(p, q) "v_mad_u64_u32 %0, vcc, %" #p ", %" #q ", %0\n\tv_addc_co_u32 %1, vcc, 0, 0, vcc\n\t"
#define PG_MAC_FN(N, BODY, ...)                                                                               \
    __device__ __forceinline__ void mac96_##N(uint64_t &acc, uint32_t &ex, __VA_ARGS__) {                      \
        asm(BODY : "+v"(acc), "+v"(ex) : PG_MAC_IN_##N : "vcc");                                               \
    }
#define PG_MAC_FN0(N, BODY, ...)                                                                              \
    __device__ __forceinline__ void mac96f_##N(uint64_t &acc, uint32_t &ex, __VA_ARGS__) {                     \
        asm(BODY : "+v"(acc), "=&v"(ex) : PG_MAC_IN_##N : "vcc");                                              \
    }
#define PG_MAC_IN_1 "v"(p0), "v"(q0)
#define PG_MAC_IN_2 PG_MAC_IN_1, "v"(p1), "v"(q1)
#define PG_MAC_IN_3 PG_MAC_IN_2, "v"(p2), "v"(q2)
#define PG_MAC_IN_4 PG_MAC_IN_3, "v"(p3), "v"(q3)
#define PG_MAC_IN_5 PG_MAC_IN_4, "v"(p4), "v"(q4)
#define PG_MAC_IN_6 PG_MAC_IN_5, "v"(p5), "v"(q5)
#define PG_MAC_IN_7 PG_MAC_IN_6, "v"(p6), "v"(q6)
#define PG_MAC_IN_8 PG_MAC_IN_7, "v"(p7), "v"(q7)
#define PG_U2(i) uint32_t p##i, uint32_t q##i
PG_MAC_FN(1, PG_MAC(2, 3), PG_U2(0))
PG_MAC_FN(2, PG_MAC(2, 3) PG_MAC(4, 5), PG_U2(0), PG_U2(1))
PG_MAC_FN(3, PG_MAC(2, 3) PG_MAC(4, 5) PG_MAC(6, 7), PG_U2(0), PG_U2(1), PG_U2(2))
PG_MAC_FN(4, PG_MAC(2, 3) PG_MAC(4, 5) PG_MAC(6, 7) PG_MAC(8, 9), PG_U2(0), PG_U2(1), PG_U2(2), PG_U2(3))
PG_MAC_FN(5, PG_MAC(2, 3) PG_MAC(4, 5) PG_MAC(6, 7) PG_MAC(8, 9) PG_MAC(10, 11), PG_U2(0), PG_U2(1), PG_U2(2), PG_U2(3), PG_U2(4))
PG_MAC_FN(6, PG_MAC(2, 3) PG_MAC(4, 5) PG_MAC(6, 7) PG_MAC(8, 9) PG_MAC(10, 11) PG_MAC(12, 13), PG_U2(0), PG_U2(1), PG_U2(2), PG_U2(3),
          PG_U2(4), PG_U2(5))
PG_MAC_FN(7, PG_MAC(2, 3) PG_MAC(4, 5) PG_MAC(6, 7) PG_MAC(8, 9) PG_MAC(10, 11) PG_MAC(12, 13) PG_MAC(14, 15), PG_U2(0), PG_U2(1),
          PG_U2(2), PG_U2(3), PG_U2(4), PG_U2(5), PG_U2(6))
PG_MAC_FN(8, PG_MAC(2, 3) PG_MAC(4, 5) PG_MAC(6, 7) PG_MAC(8, 9) PG_MAC(10, 11) PG_MAC(12, 13) PG_MAC(14, 15) PG_MAC(16, 17), PG_U2(0),
          PG_U2(1), PG_U2(2), PG_U2(3), PG_U2(4), PG_U2(5), PG_U2(6), PG_U2(7))
PG_MAC_FN0(1, PG_MAC0(2, 3), PG_U2(0))
PG_MAC_FN0(2, PG_MAC0(2, 3) PG_MAC(4, 5), PG_U2(0), PG_U2(1))
PG_MAC_FN0(3, PG_MAC0(2, 3) PG_MAC(4, 5) PG_MAC(6, 7), PG_U2(0), PG_U2(1), PG_U2(2))
PG_MAC_FN0(4, PG_MAC0(2, 3) PG_MAC(4, 5) PG_MAC(6, 7) PG_MAC(8, 9), PG_U2(0), PG_U2(1), PG_U2(2), PG_U2(3))
PG_MAC_FN0(5, PG_MAC0(2, 3) PG_MAC(4, 5) PG_MAC(6, 7) PG_MAC(8, 9) PG_MAC(10, 11), PG_U2(0), PG_U2(1), PG_U2(2), PG_U2(3), PG_U2(4))
PG_MAC_FN0(6, PG_MAC0(2, 3) PG_MAC(4, 5) PG_MAC(6, 7) PG_MAC(8, 9) PG_MAC(10, 11) PG_MAC(12, 13), PG_U2(0), PG_U2(1), PG_U2(2), PG_U2(3),
           PG_U2(4), PG_U2(5))
PG_MAC_FN0(7, PG_MAC0(2, 3) PG_MAC(4, 5) PG_MAC(6, 7) PG_MAC(8, 9) PG_MAC(10, 11) PG_MAC(12, 13) PG_MAC(14, 15), PG_U2(0), PG_U2(1),
           PG_U2(2), PG_U2(3), PG_U2(4), PG_U2(5), PG_U2(6))
PG_MAC_FN0(8, PG_MAC0(2, 3) PG_MAC(4, 5) PG_MAC(6, 7) PG_MAC(8, 9) PG_MAC(10, 11) PG_MAC(12, 13) PG_MAC(14, 15) PG_MAC(16, 17), PG_U2(0),
           PG_U2(1), PG_U2(2), PG_U2(3), PG_U2(4), PG_U2(5), PG_U2(6), PG_U2(7))
#undef PG_U2

// acc:ex += sum over i in [lo, hi] of x[i] * y[k - i]   (hi - lo < 8); FIRST: the column's first run (ex comes in as 0)
template <bool FIRST = false>
__device__ __forceinline__ void mac_run(const uint32_t (&x)[8], const uint32_t (&y)[8], int k, int lo, int hi, uint64_t &acc, uint32_t &ex) {
#define PG_XY(j) x[lo + j], y[k - lo - j]
    if constexpr (FIRST) {
        switch (hi - lo + 1) {
            case 1: mac96f_1(acc, ex, PG_XY(0)); break;
            case 2: mac96f_2(acc, ex, PG_XY(0), PG_XY(1)); break;
            case 3: mac96f_3(acc, ex, PG_XY(0), PG_XY(1), PG_XY(2)); break;
            case 4: mac96f_4(acc, ex, PG_XY(0), PG_XY(1), PG_XY(2), PG_XY(3)); break;
            case 5: mac96f_5(acc, ex, PG_XY(0), PG_XY(1), PG_XY(2), PG_XY(3), PG_XY(4)); break;
            case 6: mac96f_6(acc, ex, PG_XY(0), PG_XY(1), PG_XY(2), PG_XY(3), PG_XY(4), PG_XY(5)); break;
            case 7: mac96f_7(acc, ex, PG_XY(0), PG_XY(1), PG_XY(2), PG_XY(3), PG_XY(4), PG_XY(5), PG_XY(6)); break;
            case 8: mac96f_8(acc, ex, PG_XY(0), PG_XY(1), PG_XY(2), PG_XY(3), PG_XY(4), PG_XY(5), PG_XY(6), PG_XY(7)); break;
            default: break;
        }
        return;
    }
    switch (hi - lo + 1) {
        case 1: mac96_1(acc, ex, PG_XY(0)); break;
        case 2: mac96_2(acc, ex, PG_XY(0), PG_XY(1)); break;
        case 3: mac96_3(acc, ex, PG_XY(0), PG_XY(1), PG_XY(2)); break;
        case 4: mac96_4(acc, ex, PG_XY(0), PG_XY(1), PG_XY(2), PG_XY(3)); break;
        case 5: mac96_5(acc, ex, PG_XY(0), PG_XY(1), PG_XY(2), PG_XY(3), PG_XY(4)); break;
        case 6: mac96_6(acc, ex, PG_XY(0), PG_XY(1), PG_XY(2), PG_XY(3), PG_XY(4), PG_XY(5)); break;
        case 7: mac96_7(acc, ex, PG_XY(0), PG_XY(1), PG_XY(2), PG_XY(3), PG_XY(4), PG_XY(5), PG_XY(6)); break;
        case 8: mac96_8(acc, ex, PG_XY(0), PG_XY(1), PG_XY(2), PG_XY(3), PG_XY(4), PG_XY(5), PG_XY(6), PG_XY(7)); break;
        default: break;
    }
#undef PG_XY
}

// (Two interleaved accumulator chains per column were measured 16 % SLOWER at every occupancy: tools/fr_mul_bench.hip.)
__device__ __forceinline__ Fr fr_mul(const Fr &x, const Fr &y) {
    const uint32_t Q[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
    uint32_t a[8], b[8], m[8], r[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        a[2 * i] = (uint32_t)x.l[i]; a[2 * i + 1] = (uint32_t)(x.l[i] >> 32);
        b[2 * i] = (uint32_t)y.l[i]; b[2 * i + 1] = (uint32_t)(y.l[i] >> 32);
    }
    uint64_t acc = 0;
    uint32_t ex;
    // columns 0..7 produce the quotient digits
#pragma unroll
    for (int k = 0; k < 8; k++) {
        mac_run<true>(a, b, k, 0, k, acc, ex);
        if (k > 0) mac_run(m, Q, k, 0, k - 1, acc, ex);
        // m_k = -lo; adding m_k * q_0 = m_k zeroes the low word (a carry iff lo != 0); then the accumulator moves down one word
        uint32_t nlo, nhi;
        asm("v_sub_u32 %0, 0, %3\n\t"
            "v_cmp_ne_u32 vcc, 0, %3\n\t"
            "v_addc_co_u32 %1, vcc, %4, 0, vcc\n\t"
            "v_addc_co_u32 %2, vcc, %5, 0, vcc\n\t"
            : "=&v"(m[k]), "=&v"(nlo), "=&v"(nhi)
            : "v"((uint32_t)acc), "v"((uint32_t)(acc >> 32)), "v"(ex)
            : "vcc");
        acc = (uint64_t)nlo | ((uint64_t)nhi << 32);
    }
    // columns 8..15 are the result words
#pragma unroll
    for (int k = 8; k < 15; k++) {
        mac_run<true>(a, b, k, k - 7, 7, acc, ex);
        mac_run(m, Q, k, k - 7, 7, acc, ex);
        r[k - 8] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)ex << 32);
    }
    r[7] = (uint32_t)acc;  // (column 15 has no products)
    // value = r + top * 2^256 < 2q: one conditional subtraction
    return fr_final_sub_words(r, (uint32_t)(acc >> 32));
}
#else
PG_HD Fr fr_mul(const Fr &a, const Fr &b) { return fr_mul64(a, b); }
#endif

PG_HD Fr fr_square(const Fr &a) { return fr_mul(a, a); }

// canonical integer (raw limbs, < q) -> Montgomery form
PG_HD Fr fr_to_mont(const Fr &raw) { return fr_mul(raw, fr_r2()); }
// Montgomery form -> canonical integer as raw limbs (Scalar::reduce / to_bytes)
PG_HD Fr fr_from_mont(const Fr &a) {
    uint64_t t[8] = {a.l[0], a.l[1], a.l[2], a.l[3], 0, 0, 0, 0};
    return fr_mont_reduce(t);
}
PG_HD Fr fr_from_u64(uint64_t v) { return fr_to_mont(Fr{{v, 0, 0, 0}}); }

// raw-limb helpers on canonical integers
PG_HD uint32_t raw_bit(const Fr &c, uint32_t i) { return (uint32_t)(c.l[i >> 6] >> (i & 63)) & 1u; }
// c mod 2^nbits, nbits in [0, 256]
PG_HD Fr raw_low_bits(const Fr &c, uint32_t nbits) {
    Fr o;
    for (int i = 0; i < 4; i++) {
        uint32_t lo = 64u * i;
        uint64_t m = nbits >= lo + 64 ? ~0ull : (nbits <= lo ? 0ull : ((1ull << (nbits - lo)) - 1));
        o.l[i] = c.l[i] & m;
    }
    return o;
}
// c >> nbits != 0, nbits in [0, 256]
PG_HD bool raw_has_high_bits(const Fr &c, uint32_t nbits) {
    Fr lo = raw_low_bits(c, nbits);
    return !fr_eq(lo, c);
}
// bit length of a canonical integer (0 for 0)
PG_HD uint32_t raw_bit_length(const Fr &c) {
    for (int i = 3; i >= 0; i--)
        if (c.l[i]) {
            uint32_t n = 0;
            uint64_t v = c.l[i];
            while (v) { n++; v >>= 1; }
            return 64u * i + n;
        }
    return 0;
}

// a^(q-2): plain square-and-multiply over the fixed exponent (383 multiplications).  Kept as the independent
// cross-check of fr_invert_or_zero (tests) -- the product path uses the divstep inversion below.
PG_HD Fr fr_invert_fermat(const Fr &a) {
    const uint64_t E[4] = {0xfffffffeffffffffull, PG_Q1, PG_Q2, PG_Q3};  // q - 2
    Fr res = fr_one();
    bool started = false;
    for (int e = 3; e >= 0; e--) {
        for (int i = 63; i >= 0; i--) {
            if (started) res = fr_square(res);
            if ((E[e] >> i) & 1) {
                res = started ? fr_mul(res, a) : a;
                started = true;
            }
        }
    }
    return res;
}

// ---- inversion by division steps (Bernstein-Yang "safegcd", the 32-bit formulation with 30 steps per batch) ------
// Algorithm: D. J. Bernstein, B.-Y. Yang, "Fast constant-time gcd computation and modular inversion" (TCHES 2019).  The
// formulation used here -- numbers as nine signed 30-bit limbs, 30 division steps per batch summarised by a 2x2 matrix
// (divsteps_30), the matrix applied to (f, g) exactly and to (d, e) modulo the prime (update_fg_30 / update_de_30), the
// final normalize_30 -- is the one Pieter Wuille wrote for libsecp256k1 (src/modinv32.h / modinv32_impl.h, "modinv32",
// MIT licence; explained in that project's doc/safegcd_implementation.md): same decomposition, same function roles and
// names.  It is re-expressed here for the BLS12-381 scalar modulus (other limb constants; q^-1 mod 2^30 = 1 removes the
// multiplication by the modular inverse), for Montgomery-form inputs, and for a GPU wave (the early exit when every lane
// has reached g = 0, v_mad_i64_i32 spelt out).  Nothing of it comes from the reference, whose dependency inverts by a
// Fermat power.
// 20 batches of 30 branch-free division steps on (f, g) = (q, a) reduce g to 0 and f to +-1; every batch is
// summarised by a 2x2 integer matrix that is then applied to the full-width (f, g) and, modulo q, to (d, e) with
// d a = f, e a = g (mod q), so that at the end a^-1 = +-d.  600 steps cover any 256-bit input (590 suffice); the
// count is fixed, the loop bodies have no data-dependent branches.  Cost: ~20 x (30 x 13 32-bit ops + ~100
// 32x32->64 multiply-adds), about a tenth of the 383 Montgomery multiplications of the Fermat power -- which matters
// because the batch inversion pre-pass (invert.hpp) is one sequential chain per lane.
// Numbers are 9 signed limbs of 30 bits.
struct Signed30 {
    int32_t v[9];
};
#define PG_M30 0x3fffffff
// q in 30-bit limbs; q^-1 mod 2^30 = 1 (q = 1 mod 2^32)
#define PG_Q30 {0x1, 0x3ffffffc, 0x3fe5bfef, 0x2f6900bf, 0x21d80553, 0x27602026, 0x17d48333, 0x29d4ca67, 0x73ed}

PG_HD void raw_to_signed30(const Fr &a, Signed30 &r) {
    r.v[0] = (int32_t)(a.l[0] & PG_M30);
    r.v[1] = (int32_t)((a.l[0] >> 30) & PG_M30);
    r.v[2] = (int32_t)(((a.l[0] >> 60) | (a.l[1] << 4)) & PG_M30);
    r.v[3] = (int32_t)((a.l[1] >> 26) & PG_M30);
    r.v[4] = (int32_t)(((a.l[1] >> 56) | (a.l[2] << 8)) & PG_M30);
    r.v[5] = (int32_t)((a.l[2] >> 22) & PG_M30);
    r.v[6] = (int32_t)(((a.l[2] >> 52) | (a.l[3] << 12)) & PG_M30);
    r.v[7] = (int32_t)((a.l[3] >> 18) & PG_M30);
    r.v[8] = (int32_t)(a.l[3] >> 48);
}
PG_HD Fr signed30_to_raw(const Signed30 &r) {  // r normalised: limbs in [0, 2^30), value < 2^256
    const uint64_t v0 = (uint32_t)r.v[0], v1 = (uint32_t)r.v[1], v2 = (uint32_t)r.v[2], v3 = (uint32_t)r.v[3], v4 = (uint32_t)r.v[4],
                   v5 = (uint32_t)r.v[5], v6 = (uint32_t)r.v[6], v7 = (uint32_t)r.v[7], v8 = (uint32_t)r.v[8];
    return Fr{{v0 | v1 << 30 | v2 << 60, v2 >> 4 | v3 << 26 | v4 << 56, v4 >> 8 | v5 << 22 | v6 << 52, v6 >> 12 | v7 << 18 | v8 << 48}};
}

// 30 division steps on the low limbs; t = {u, v, q, r} with 2^30 (f', g') = (u f + v g, q f + r g)
// zeta = -(delta + 1/2) of the paper, so that its sign bit is the branch condition
PG_HD int32_t divsteps_30(int32_t zeta, uint32_t f, uint32_t g, int32_t t[4]) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
#pragma unroll
    for (int i = 0; i < 30; i++) {
        uint32_t c1 = (uint32_t)(zeta >> 31);  // delta > 0
        const uint32_t c2 = 0u - (g & 1u);     // g odd
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // +-(f, u, v)
        g += x & c2;
        q += y & c2;
        r += z & c2;
        c1 &= c2;  // swap: delta > 0 and g odd
        zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
        f += g & c1;
        u += q & c1;
        v += r & c1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t[0] = (int32_t)u; t[1] = (int32_t)v; t[2] = (int32_t)q; t[3] = (int32_t)r;
    return zeta;
}

// c + a * b with 32-bit signed factors: ONE instruction on gfx950 (v_mad_i64_i32).  Written through int64 operands the
// compiler does not always see that both factors are sign-extended 32-bit values and emits a 64 x 32 product (a
// v_mad_u64_u32, a v_mul_lo_u32 and an add) -- 2.7 x the multiplies in the inversion's inner updates.
PG_HD int64_t smad32(int32_t a, int32_t b, int64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t r;
    uint64_t carry;  // the carry-out goes to a scratch SGPR pair, not VCC: nothing then has to wait for VCC
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return c + (int64_t)a * b;
#endif
}

// (f, g) <- t (f, g) / 2^30, exact
PG_HD void update_fg_30(Signed30 &f, Signed30 &g, const int32_t t[4]) {
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    int64_t cf = smad32(u, f.v[0], smad32(v, g.v[0], 0)), cg = smad32(q, f.v[0], smad32(r, g.v[0], 0));
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cf = smad32(u, f.v[i], smad32(v, g.v[i], cf));
        cg = smad32(q, f.v[i], smad32(r, g.v[i], cg));
        f.v[i - 1] = (int32_t)cf & PG_M30;
        g.v[i - 1] = (int32_t)cg & PG_M30;
        cf >>= 30;
        cg >>= 30;
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}

// (d, e) <- t (d, e) / 2^30 mod q: a multiple of q is added first so that the division is exact; d, e stay in (-2q, q)
PG_HD void update_de_30(Signed30 &d, Signed30 &e, const int32_t t[4]) {
    const int32_t Q30[9] = PG_Q30;
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (t[0] & sd) + (t[1] & se), me = (t[2] & sd) + (t[3] & se);
    int64_t cd = smad32(u, d.v[0], smad32(v, e.v[0], 0)), ce = smad32(q, d.v[0], smad32(r, e.v[0], 0));
    md -= (int32_t)(((uint32_t)cd + (uint32_t)md) & PG_M30);  // q^-1 mod 2^30 = 1
    me -= (int32_t)(((uint32_t)ce + (uint32_t)me) & PG_M30);
    cd = smad32(Q30[0], md, cd);
    ce = smad32(Q30[0], me, ce);
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cd = smad32(u, d.v[i], smad32(v, e.v[i], smad32(Q30[i], md, cd)));
        ce = smad32(q, d.v[i], smad32(r, e.v[i], smad32(Q30[i], me, ce)));
        d.v[i - 1] = (int32_t)cd & PG_M30;
        e.v[i - 1] = (int32_t)ce & PG_M30;
        cd >>= 30;
        ce >>= 30;
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
}

// r in (-2q, q) -> [0, q), negated first if `negate` is all ones
PG_HD void normalize_30(Signed30 &r, int32_t negate) {
    const int32_t Q30[9] = PG_Q30;
    int32_t add = r.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = ((r.v[i] + (Q30[i] & add)) ^ negate) - negate;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i + 1] += r.v[i] >> 30;
        r.v[i] &= PG_M30;
    }
    add = r.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] += Q30[i] & add;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.v[i + 1] += r.v[i] >> 30;
        r.v[i] &= PG_M30;
    }
}

// Montgomery form in, Montgomery form out: the steps run on the residue a = xR and give (xR)^-1 = x^-1 R^-1; one
// Montgomery multiplication by R^3 turns that into x^-1 R.  Returns 0 for 0 (CtOption::unwrap_or(zero) at
// scalar.rs:122; the caller tests for zero where the reference distinguishes, scalar.rs:73-80): with g = 0 the steps
// leave d = 0.
PG_HD Fr fr_invert_or_zero(const Fr &a) {
#ifdef PG_INVERT_FERMAT  // A/B build (tools/ab_emit.py)
    return fr_invert_fermat(a);
#endif
    const int32_t Q30[9] = PG_Q30;
    Signed30 d, e, f, g;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        d.v[i] = 0;
        e.v[i] = i == 0;
        f.v[i] = Q30[i];
    }
    raw_to_signed30(a, g);
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 20; it++) {
        int32_t t[4];
        zeta = divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        update_de_30(d, e, t);
        update_fg_30(f, g, t);
#if defined(__HIP_DEVICE_COMPILE__)
        // g = 0 ends the computation: further steps leave f and d as they are (u = 2^30, v = 0).  A wave stops when every
        // active lane is there -- 500-530 steps for random 255-bit inputs, i.e. 18 batches instead of the 20 the bound needs.
        const int32_t gnz = g.v[0] | g.v[1] | g.v[2] | g.v[3] | g.v[4] | g.v[5] | g.v[6] | g.v[7] | g.v[8];
        if (!__any(gnz != 0)) break;
#endif
    }
    normalize_30(d, f.v[8] >> 31);
    const Fr r3{{0xc62c1807439b73afull, 0x1b3e0d188cf06990ull, 0x73d13c71c7b5f418ull, 0x6e2a5bb9c8db33e9ull}};  // R^3 mod q
    return fr_mul(signed30_to_raw(d), r3);
}

// 2^by mod q in Montgomery form, by repeated doubling (BlsScalar::pow_of_2, range.rs:187)
PG_HD Fr fr_pow_of_2(uint64_t by) {
    Fr r = fr_one();
    for (uint64_t i = 0; i < by; i++) r = fr_add(r, r);
    return r;
}

// range.rs:173-181: bit length of the canonical value, minimum 1
PG_HD uint64_t bits_count(const Fr &s) {
    uint32_t n = raw_bit_length(fr_from_mont(s));
    return n < 1 ? 1 : n;
}
// range.rs:185-189
PG_HD uint64_t num_bits_closest_power_of_two(const Fr &s) { return bits_count(fr_pow_of_2(bits_count(s))); }

}  // namespace pg
