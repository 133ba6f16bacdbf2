/*
 * oracle/fr.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of the slice of `BlsScalar` (= dusk-bls12_381 `Scalar`,
 * the BLS12-381 scalar field in 4x64-bit-limb Montgomery form) that the
 * reference's gadget path calls (call sites: /root/reference/src/range.rs:63,
 * 69,87,94,102,130,146,152,163,174-177,187 and src/scalar.rs:47,73,91,113,
 * 121-126).
 *
 * The algorithm lives in the third-party crate dusk-bls12_381 (pulled in by
 * dusk-plonk "^0.8", /root/reference/Cargo.toml:20; no Cargo.lock is committed
 * so the exact patch version is not pinned).  Its source is NOT under
 * /root/reference; what follows restates its published algorithm (the
 * zkcrypto/bls12_381 scalar code dusk forked): schoolbook 4x4 multiply +
 * 4-round Montgomery reduction + one final conditional subtraction, values
 * always fully reduced in [0, q).
 *
 * PARITY UNPINNED for limb values: the reference holds no golden vectors for
 * field elements.  This file is pinned instead against (a) the constants of
 * SURVEY.md section 8a, (b) Python big-int arithmetic (tests/test_oracle_fr.py)
 * and (c) the reference's only pure-arithmetic KAT, `counting_scalar_bits`
 * (src/range.rs:196-203), through oracle/gadgets.c:bits_count.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything in this directory.  The shipped library never links it.
 */
#ifndef PG_ORACLE_FR_H
#define PG_ORACLE_FR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } fr_t; /* little-endian limbs, Montgomery form x*R mod q, R = 2^256 */

extern const fr_t FR_MODULUS; /* q, raw limbs */
extern const fr_t FR_ZERO;    /* mont(0) */
extern const fr_t FR_ONE;     /* mont(1) = R mod q */
extern const fr_t FR_R2;      /* R^2 mod q */

fr_t fr_add(fr_t a, fr_t b);
fr_t fr_sub(fr_t a, fr_t b);
fr_t fr_neg(fr_t a);
fr_t fr_mul(fr_t a, fr_t b);
fr_t fr_square(fr_t a);
fr_t fr_from_u64(uint64_t v);             /* BlsScalar::from(u64) */
fr_t fr_from_raw(const uint64_t v[4]);    /* BlsScalar::from_raw: v*R2 reduced */
fr_t fr_reduce(fr_t a);                   /* Scalar::reduce(): out of Montgomery form, raw limbs = canonical value */
void fr_to_bytes(fr_t a, uint8_t out[32]);/* Serializable::to_bytes(): canonical little-endian */
void fr_divn(fr_t *a, uint32_t n);        /* Scalar::divn on the raw limbs */
int  fr_cmp_raw(fr_t a, fr_t b);          /* Ord for Scalar: limb-wise from the top limb down */
int  fr_eq(fr_t a, fr_t b);
int  fr_is_zero(fr_t a);
fr_t fr_pow(fr_t base, const uint64_t by[4]); /* Scalar::pow: 256 squarings + 256 multiplies, always */
fr_t fr_pow_of_2(uint64_t by);            /* Scalar::pow_of_2 */
int  fr_invert(fr_t a, fr_t *out);        /* Scalar::invert -> CtOption: returns 1 and *out=a^-1, or 0 when a == 0 */

#ifdef __cplusplus
}
#endif
#endif
