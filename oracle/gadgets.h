/*
 * oracle/gadgets.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of the reference's gadget layer, function for function:
 *   /root/reference/src/allocated_scalar.rs:17-30   AllocatedScalar, allocate
 *   /root/reference/src/errors.rs:13-18             Error::NonExistingInverse
 *   /root/reference/src/range.rs:21-189             range_proof, range_check, min_bound, max_bound,
 *                                                   scalar_decomposition_gadget, scalar_to_bits,
 *                                                   bits_count, num_bits_closest_power_of_two
 *   /root/reference/src/scalar.rs:21-140            conditionally_select_zero/one, is_non_zero, maybe_equal
 * Same names, same argument order, same sequence of composer calls, same
 * per-bit `pow` (range.rs:146) -- it is the checker for the HIP path and the
 * timed "port" CPU baseline.  The shipped library never links it.
 */
#ifndef PG_ORACLE_GADGETS_H
#define PG_ORACLE_GADGETS_H

#include "composer.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { var_t var; fr_t scalar; } allocated_scalar_t;

enum { GADGET_OK = 0, GADGET_ERR_NON_EXISTING_INVERSE = 1 };

allocated_scalar_t allocated_scalar_allocate(composer_t *c, fr_t scalar);

var_t range_check(composer_t *c, fr_t min_range, fr_t max_range, allocated_scalar_t witness);
var_t max_bound(composer_t *c, fr_t max_range, allocated_scalar_t witness, uint64_t *num_bits_pow_2);
/* private in the reference; exposed so the tests can reach every function */
var_t min_bound(composer_t *c, fr_t min_range, allocated_scalar_t witness, uint64_t num_bits);
var_t range_proof(composer_t *c, allocated_scalar_t value, uint64_t num_bits);
/* returns is_equal; bit variables (first num_bits) written to bits_out if non-NULL.
 * num_bits > 256 panics in the reference (slice [..num_bits], range.rs:134): returns (var_t)-1. */
var_t scalar_decomposition_gadget(composer_t *c, size_t num_bits, allocated_scalar_t witness, var_t *bits_out);
void scalar_to_bits(fr_t scalar, uint8_t out[256]);
uint64_t bits_count(fr_t scalar);
uint64_t num_bits_closest_power_of_two(fr_t scalar);

var_t conditionally_select_zero(composer_t *c, var_t x, var_t select);
var_t conditionally_select_one(composer_t *c, var_t y, var_t selector);
int is_non_zero(composer_t *c, var_t var, fr_t value_assigned);
var_t maybe_equal(composer_t *c, allocated_scalar_t a, allocated_scalar_t b);

/* ---- batch drivers (what the HIP path is compared with) --------------- */

/* The 8 live columns + variable table the engine materialises
 * (SURVEY.md section 8a row a14).  Arrays are caller-allocated. */
typedef struct {
    fr_t *q_m, *q_l, *q_r, *q_o, *q_c;
    uint64_t *w_l, *w_r, *w_o;
    fr_t *var_values;
} oracle_columns_t;

/* for w in witnesses: allocate(w); range_check(min,max,w)  on one fresh
 * composer.  Copies rows [gate_base, n) / variables [var_base, nvars) into
 * `out` (may be NULL for timing only), result variables into result_vars (may
 * be NULL).  Returns 0, or -1 if some row is unsatisfied (check != 0 only). */
int oracle_range_check_batch(fr_t min_range, fr_t max_range, const fr_t *witness, size_t batch, int check,
                             oracle_columns_t *out, uint64_t *result_vars, uint64_t *gate_base, uint64_t *var_base,
                             uint64_t *n_gates, uint64_t *n_vars);

/* oracle/fast.c: the same output as oracle_range_check_batch's columns (rows relative to the call, Variables from
 * var_base), written directly at closed-form offsets by `threads` POSIX threads, mont(2^i) from a table. */
int oracle_range_check_fast(fr_t min_range, fr_t max_range, const fr_t *witness, size_t batch, uint64_t var_base,
                            int threads, oracle_columns_t *out, uint64_t *result_vars);

/* the same loop on witnesses allocated BEFORE it (the reference's own signature, range.rs:27-32): item i's Variable is
 * witness_vars[i]; 4n + 11 rows and 2n + 523 new Variables per item */
int oracle_range_check_allocated_fast(fr_t min_range, fr_t max_range, const fr_t *witness, const uint64_t *witness_vars,
                                      size_t batch, uint64_t var_base, int threads, oracle_columns_t *out,
                                      uint64_t *result_vars);

/* oracle/fast.c, ragged forms: the columns of oracle_max_bound_batch / oracle_scalar_mix_batch for items [lo, hi) of a
 * batch whose prefix sums (row_off / var_off, batch + 1 entries, from the *_plan calls) the caller holds: rows are
 * written at row_off[i] - row_off[lo], item i's first Variable is var_base + var_off[i] and its assignment lies at
 * var_off[i] - var_off[lo] of out->var_values; inputs are indexed by i, result_vars by i - lo. */
int oracle_max_bound_plan(const fr_t *max_range, size_t batch, int threads, uint64_t *num_bits, uint64_t *row_off,
                          uint64_t *var_off);
int oracle_max_bound_fast(const fr_t *max_range, const fr_t *witness, const uint64_t *num_bits, const uint64_t *row_off,
                          const uint64_t *var_off, size_t lo, size_t hi, uint64_t var_base, int threads,
                          oracle_columns_t *out, uint64_t *result_vars);
int oracle_scalar_mix_plan(const fr_t *v, size_t batch, uint64_t *row_off, uint64_t *var_off, uint8_t *err_mask);
int oracle_scalar_mix_fast(const fr_t *v, const fr_t *y, const fr_t *s, const fr_t *a, const fr_t *b,
                           const uint64_t *row_off, const uint64_t *var_off, size_t lo, size_t hi, uint64_t var_base,
                           uint64_t zero_var, int threads, oracle_columns_t *out, uint64_t *result_vars);

/* for i: allocate(witness[i]); max_bound(max_range[i], .) */
int oracle_max_bound_batch(const fr_t *max_range, const fr_t *witness, size_t batch, int check, oracle_columns_t *out,
                           uint64_t *result_vars, uint64_t *num_bits, uint64_t *gate_base, uint64_t *var_base,
                           uint64_t *n_gates, uint64_t *n_vars);

/* for i: v,y,s,a,b = add_input x5; is_non_zero(var v, v); conditionally_select_one(y, s); maybe_equal(a, b)
 * (SURVEY.md section 8d, config C3).  err_mask[i] = 1 where is_non_zero
 * returned NonExistingInverse (the partial emission stays in the columns, and
 * the remaining two gadgets of the item still run).  out3[i*2+0]=select var,
 * out3[i*2+1]=maybe_equal var. */
int oracle_scalar_mix_batch(const fr_t *v, const fr_t *y, const fr_t *s, const fr_t *a, const fr_t *b, size_t batch,
                            int check, oracle_columns_t *out, uint64_t *result_vars, uint8_t *err_mask,
                            uint64_t *gate_base, uint64_t *var_base, uint64_t *n_gates, uint64_t *n_vars);

/* oracle/fast.c: `for i: gadget(a[i], b[i])` on EXISTING Variables -- the scalar gadgets (scalar.rs:21-140) and the composer's add /
 * mul / poly_gate over arrays -- items [lo, hi), threaded; `table`: the assignments of the Variables [0, var_base) that exist before
 * the call.  is_non_zero is ragged (its plan gives the prefix sums), the others uniform. */
enum { ORACLE_SELECT_ZERO = 0, ORACLE_SELECT_ONE, ORACLE_MAYBE_EQUAL, ORACLE_IS_NON_ZERO, ORACLE_GATE_ADD, ORACLE_GATE_MUL, ORACLE_GATE_ROWS };
int oracle_is_non_zero_plan(const uint64_t *vars, const fr_t *table, size_t batch, uint64_t *row_off, uint64_t *var_off,
                            uint8_t *err_mask);
int oracle_small_batch_fast(int kind, const uint64_t *a, const uint64_t *b, const uint64_t *c, const fr_t *table,
                            const fr_t *selectors, const uint64_t *row_off, const uint64_t *var_off, size_t lo, size_t hi,
                            uint64_t var_base, uint64_t zero_var, int threads, oracle_columns_t *out, uint64_t *result_vars);

/* ---- the f-rows of a whole circuit from its wire columns (oracle/fast.c) -------------------------------
 * What composer_sigma() and composer_selector / composer_wire / composer_values_dense give for a composer.c composer,
 * computed from the four wire columns + the assignments, threaded, for row ranges -- so that circuits of hundreds of millions
 * of rows (assembled from the fast gadget forms above) can be checked word for word. */
typedef struct oracle_sigma_plan oracle_sigma_plan_t;
/* pass 1 over all n rows (first / last position of every Variable); wires must stay alive until _end */
oracle_sigma_plan_t *oracle_sigma_fast_begin(const uint64_t *w_l, const uint64_t *w_r, const uint64_t *w_o, const uint64_t *w_4,
                                             size_t n, size_t padded_n, size_t n_vars, int threads);
/* sigma[w][r0 .. r1) -> out[w][0 .. r1 - r0); chunks from the LAST row to the first: r1 = padded_n, then the previous r0 */
int oracle_sigma_fast_chunk(oracle_sigma_plan_t *plan, size_t r0, size_t r1, uint64_t *const out[4]);
void oracle_sigma_fast_end(oracle_sigma_plan_t *plan);

/* the columns pg_composer_materialize produces beyond the eight live ones */
typedef struct {
    fr_t *q_4, *q_arith, *q_range, *q_logic, *q_fixed_group_add, *q_variable_group_add;
    uint64_t *w_4;
    fr_t *w_l_value, *w_r_value, *w_o_value, *w_4_value;
} oracle_full_columns_t;
/* rows [r0, r1) -> out[..][0 .. r1 - r0); q_4 is zero except on the n_q4 listed rows (the dummy rows of composer_new()) */
int oracle_materialize_fast(const uint64_t *w_l, const uint64_t *w_r, const uint64_t *w_o, const uint64_t *w_4,
                            const fr_t *values, size_t n_vars, const uint64_t *q4_rows, const fr_t *q4_values, size_t n_q4,
                            size_t r0, size_t r1, int threads, const oracle_full_columns_t *out);

#ifdef __cplusplus
}
#endif
#endif
