/*
 * oracle/composer.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of the slice of dusk-plonk 0.8.x `StandardComposer`
 * that the reference's gadgets call (call sites listed per function below).
 * The crate's source is NOT under /root/reference (Cargo.toml:20 pulls
 * dusk-plonk "^0.8"; no Cargo.lock), so this restates its published
 * behaviour: every gate appends one element to each of 11 selector columns
 * and 4 wire columns, bumps `n`, and records the 4 wire positions of the row
 * in the permutation's Variable -> [WireData] map; witness assignments live in
 * a Variable -> BlsScalar hash map; Variables are numbered sequentially.
 *
 * PARITY UNPINNED for rows/indices: no test or fixture of the reference
 * records gate rows, wire indices or variable numbering.  What the reference
 * does pin -- the 0/1 outcome of each test case and satisfiability -- is
 * checked by tests/test_oracle_reference_cases.py through composer_check().
 * The cost structure (hash-map look-ups, one push per column per gate) is
 * kept on purpose: this is also bench.py's cpu_baseline ("port").
 */
#ifndef PG_ORACLE_COMPOSER_H
#define PG_ORACLE_COMPOSER_H

#include "fr.h"
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t var_t; /* Variable(usize) */

typedef struct composer composer_t;

/* StandardComposer::new(): zero_var = Variable(0) via
 * add_witness_to_circuit_description(0) (gate 0), then add_dummy_constraints()
 * (variables 1..4 = 6, 1, 7, -20 and two dummy gates): 3 gates, 5 variables. */
composer_t *composer_new(void);
/* same object with nothing pre-allocated except zero_var's row (for tests that
 * want to show the initial state is a parameter, not baked in) */
composer_t *composer_new_without_dummy(void);
void composer_free(composer_t *c);

size_t composer_circuit_size(const composer_t *c);  /* n: gates so far */
size_t composer_num_variables(const composer_t *c);
var_t composer_zero_var(const composer_t *c);

/* allocated_scalar.rs:28; range.rs:130; scalar.rs:69,77,123 */
var_t composer_add_input(composer_t *c, fr_t s);
/* range.rs:139; scalar.rs:41,83 */
var_t composer_add_witness_to_circuit_description(composer_t *c, fr_t v);
/* tests/range_gadgets_tests.rs:26,43; tests/scalar_gadgets_tests.rs:30,82,135 */
void composer_constrain_to_constant(composer_t *c, var_t a, fr_t constant, const fr_t *pi);
/* scalar.rs:71 */
void composer_assert_equal(composer_t *c, var_t a, var_t b);
/* scalar.rs:84-94 */
void composer_poly_gate(composer_t *c, var_t a, var_t b, var_t o, fr_t q_m, fr_t q_l, fr_t q_r, fr_t q_o, fr_t q_c,
                        const fr_t *pi);
/* range.rs:65,98,151; scalar.rs:45,53,116 */
var_t composer_add(composer_t *c, fr_t q_l, var_t a, fr_t q_r, var_t b, fr_t q_c, const fr_t *pi);
/* range.rs:42; scalar.rs:26,43,126 */
var_t composer_mul(composer_t *c, fr_t q_m, var_t a, var_t b, fr_t q_c, const fr_t *pi);
/* scalar.rs:137 */
void composer_mul_gate(composer_t *c, var_t a, var_t b, var_t o, fr_t q_m, fr_t q_o, fr_t q_c, const fr_t *pi);
/* range.rs:144 */
var_t composer_boolean_gate(composer_t *c, var_t a);

/* read-back for the checker / parity tests */
fr_t composer_value(const composer_t *c, var_t v);
/* column ids for composer_selector(): */
enum { PG_Q_M, PG_Q_L, PG_Q_R, PG_Q_O, PG_Q_C, PG_Q_4, PG_Q_ARITH, PG_Q_RANGE, PG_Q_LOGIC, PG_Q_FIXED, PG_Q_VARIABLE, PG_NSEL };
enum { PG_W_L, PG_W_R, PG_W_O, PG_W_4, PG_NWIRE };
const fr_t *composer_selector(const composer_t *c, int col);
const var_t *composer_wire(const composer_t *c, int col);
/* dense variable table [0, num_variables) in Variable order */
void composer_values_dense(const composer_t *c, fr_t *out);
/* number of wire positions recorded for a variable in the permutation map */
size_t composer_perm_count(const composer_t *c, var_t v);
/* Permutation::compute_sigma_permutations(n) [dusk-plonk 0.8, restated]: every Variable's wire positions, in the
 * order they were recorded (gate by gate; left, right, output, fourth within a gate), form one cycle; sigma maps each
 * position to the next one of its Variable.  Positions are encoded wire * padded_n + gate; rows >= circuit size map
 * to themselves.  out has 4 * padded_n entries. */
void composer_sigma(const composer_t *c, size_t padded_n, uint64_t *out);
/* construct_dense_pi_vec (tests/scalar_gadgets_tests.rs:151,173,211,229) */
void composer_dense_pi(const composer_t *c, fr_t *out /* n entries */);

/* q_m*a*b + q_l*a + q_r*b + q_o*c + q_4*d + q_c + PI == 0 on every row whose
 * q_arith = 1; returns the index of the first unsatisfied row or -1. */
long composer_check(const composer_t *c);

#ifdef __cplusplus
}
#endif
#endif
