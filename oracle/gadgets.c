/*
 * oracle/gadgets.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).  See gadgets.h.
 * Each function cites the reference lines it follows.
 */
#include "gadgets.h"
#include <stdlib.h>
#include <string.h>

/* allocated_scalar.rs:27-30 */
allocated_scalar_t allocated_scalar_allocate(composer_t *c, fr_t scalar) {
    allocated_scalar_t a;
    a.var = composer_add_input(c, scalar);
    a.scalar = scalar;
    return a;
}

/* range.rs:161-170 -- canonical LE bytes, bit i = (byte[i/8] >> (i%8)) & 1 */
void scalar_to_bits(fr_t scalar, uint8_t out[256]) {
    uint8_t bytes[32];
    fr_to_bytes(scalar, bytes);
    for (int byte = 0; byte < 32; byte++)
        for (int i = 0; i < 8; i++) out[8 * byte + i] = (bytes[byte] >> i) & 1;
}

/* range.rs:173-181 */
uint64_t bits_count(fr_t scalar) {
    scalar = fr_reduce(scalar);
    fr_t one_reduced = fr_reduce(FR_ONE);
    uint64_t counter = 1;
    while (fr_cmp_raw(scalar, one_reduced) > 0) {
        fr_divn(&scalar, 1);
        counter++;
    }
    return counter;
}

/* range.rs:185-189 */
uint64_t num_bits_closest_power_of_two(fr_t scalar) {
    uint64_t num_bits = bits_count(scalar);
    fr_t closest_pow_of_two = fr_pow_of_2(num_bits);
    return bits_count(closest_pow_of_two);
}

/* scalar.rs:105-140 */
var_t maybe_equal(composer_t *c, allocated_scalar_t a, allocated_scalar_t b) {
    /* u = a - b                                                  :111-117 */
    var_t u = composer_add(c, FR_ONE, a.var, fr_neg(FR_ONE), b.var, FR_ZERO, NULL);
    /* z = u^-1 or 0                                              :121-123 */
    fr_t u_scalar = fr_sub(a.scalar, b.scalar), u_inv_scalar;
    if (!fr_invert(u_scalar, &u_inv_scalar)) u_inv_scalar = FR_ZERO;
    var_t z = composer_add_input(c, u_inv_scalar);
    /* y = 1 - u z                                                :126 */
    var_t y = composer_mul(c, fr_neg(FR_ONE), z, u, FR_ONE, NULL);
    /* y u = 0                                                    :129-138 */
    composer_mul_gate(c, y, u, u, FR_ONE, FR_ZERO, FR_ZERO, NULL);
    return y;
}

/* range.rs:119-158 */
var_t scalar_decomposition_gadget(composer_t *c, size_t num_bits, allocated_scalar_t witness, var_t *bits_out) {
    uint8_t scalar_bits[256];
    var_t scalar_bits_var[256];
    scalar_to_bits(witness.scalar, scalar_bits);                                   /* :125 */
    for (int i = 0; i < 256; i++)                                                  /* :128-131: all 256 are allocated */
        scalar_bits_var[i] = composer_add_input(c, fr_from_u64(scalar_bits[i]));
    if (num_bits > 256) return (var_t)-1;                                          /* :134 would panic */

    allocated_scalar_t accumulator;                                                /* :138-141 */
    accumulator.var = composer_add_witness_to_circuit_description(c, FR_ZERO);
    accumulator.scalar = FR_ZERO;

    for (size_t power = 0; power < num_bits; power++) {                            /* :143-153 */
        var_t bit = scalar_bits_var[power];
        composer_boolean_gate(c, bit);
        uint64_t e[4] = {(uint64_t)power, 0, 0, 0};
        fr_t two_pow = fr_pow(fr_from_u64(2), e);                                  /* :146 -- full pow per bit */
        accumulator.var = composer_add(c, two_pow, bit, FR_ONE, accumulator.var, FR_ZERO, NULL);
        accumulator.scalar = fr_add(accumulator.scalar, fr_mul(two_pow, fr_from_u64(scalar_bits[power])));
    }
    if (bits_out) memcpy(bits_out, scalar_bits_var, num_bits * sizeof(var_t));
    return maybe_equal(c, accumulator, witness);                                   /* :155 */
}

/* range.rs:21-24 */
var_t range_proof(composer_t *c, allocated_scalar_t value, uint64_t num_bits) {
    return scalar_decomposition_gadget(c, (size_t)num_bits, value, NULL);
}

/* range.rs:82-113 */
var_t max_bound(composer_t *c, fr_t max_range, allocated_scalar_t witness, uint64_t *num_bits_pow_2_out) {
    max_range = fr_sub(max_range, FR_ONE);                                         /* :87 */
    uint64_t num_bits_pow_2 = num_bits_closest_power_of_two(max_range);            /* :90 */
    /* b - x: q_l = -1 on x, q_r = 0 on x (not zero_var), q_c = max-1  :93-99 */
    var_t b_minus_x_var = composer_add(c, fr_neg(FR_ONE), witness.var, FR_ZERO, witness.var, max_range, NULL);
    allocated_scalar_t b_prime_plus_x;
    b_prime_plus_x.var = b_minus_x_var;
    b_prime_plus_x.scalar = fr_sub(max_range, witness.scalar);                     /* :102 */
    if (num_bits_pow_2_out) *num_bits_pow_2_out = num_bits_pow_2;
    return range_proof(c, b_prime_plus_x, num_bits_pow_2);                         /* :110 */
}

/* range.rs:53-76 */
var_t min_bound(composer_t *c, fr_t min_range, allocated_scalar_t witness, uint64_t num_bits) {
    /* x - a: q_l = 1 on x, q_r = 0 on x, q_c = -min                  :60-66 */
    var_t x_min_a_var = composer_add(c, FR_ONE, witness.var, FR_ZERO, witness.var, fr_neg(min_range), NULL);
    allocated_scalar_t x_min_a;
    x_min_a.var = x_min_a_var;
    x_min_a.scalar = fr_sub(witness.scalar, min_range);                            /* :69 */
    return range_proof(c, x_min_a, num_bits);                                      /* :75 */
}

/* range.rs:27-43 */
var_t range_check(composer_t *c, fr_t min_range, fr_t max_range, allocated_scalar_t witness) {
    uint64_t num_bits_pow_2;
    var_t y1 = max_bound(c, max_range, witness, &num_bits_pow_2);                  /* :34 */
    var_t y2 = min_bound(c, min_range, witness, num_bits_pow_2);                   /* :37 */
    return composer_mul(c, FR_ONE, y1, y2, FR_ZERO, NULL);                         /* :42 */
}

/* scalar.rs:21-27 */
var_t conditionally_select_zero(composer_t *c, var_t x, var_t select) {
    return composer_mul(c, FR_ONE, x, select, FR_ZERO, NULL);
}

/* scalar.rs:36-59 */
var_t conditionally_select_one(composer_t *c, var_t y, var_t selector) {
    var_t one = composer_add_witness_to_circuit_description(c, FR_ONE);            /* :41 */
    var_t selector_y = composer_mul(c, FR_ONE, y, selector, FR_ZERO, NULL);        /* :43 */
    var_t one_min_selector = composer_add(c, FR_ONE, one, fr_neg(FR_ONE), selector, FR_ZERO, NULL); /* :45-50 */
    return composer_add(c, FR_ONE, selector_y, FR_ONE, one_min_selector, FR_ZERO, NULL);            /* :53-58 */
}

/* scalar.rs:63-97 */
int is_non_zero(composer_t *c, var_t var, fr_t value_assigned) {
    var_t var_assigned = composer_add_input(c, value_assigned);                    /* :69 */
    composer_assert_equal(c, var, var_assigned);                                   /* :71 */
    fr_t inverse;
    if (!fr_invert(value_assigned, &inverse))                                      /* :73-80 */
        return GADGET_ERR_NON_EXISTING_INVERSE;
    var_t inv = composer_add_input(c, inverse);                                    /* :77 */
    var_t one = composer_add_witness_to_circuit_description(c, FR_ONE);            /* :83 */
    composer_poly_gate(c, var, inv, one, FR_ONE, FR_ZERO, FR_ZERO, fr_neg(FR_ONE), FR_ZERO, NULL); /* :84-94 */
    return GADGET_OK;
}

/* ---- batch drivers ------------------------------------------------------ */

static void export_columns(const composer_t *c, uint64_t g0, uint64_t v0, oracle_columns_t *out) {
    size_t n = composer_circuit_size(c), nv = composer_num_variables(c);
    if (!out) return;
    memcpy(out->q_m, composer_selector(c, PG_Q_M) + g0, (n - g0) * sizeof(fr_t));
    memcpy(out->q_l, composer_selector(c, PG_Q_L) + g0, (n - g0) * sizeof(fr_t));
    memcpy(out->q_r, composer_selector(c, PG_Q_R) + g0, (n - g0) * sizeof(fr_t));
    memcpy(out->q_o, composer_selector(c, PG_Q_O) + g0, (n - g0) * sizeof(fr_t));
    memcpy(out->q_c, composer_selector(c, PG_Q_C) + g0, (n - g0) * sizeof(fr_t));
    memcpy(out->w_l, composer_wire(c, PG_W_L) + g0, (n - g0) * sizeof(uint64_t));
    memcpy(out->w_r, composer_wire(c, PG_W_R) + g0, (n - g0) * sizeof(uint64_t));
    memcpy(out->w_o, composer_wire(c, PG_W_O) + g0, (n - g0) * sizeof(uint64_t));
    fr_t *all = (fr_t *)malloc(nv * sizeof(fr_t));
    composer_values_dense(c, all);
    memcpy(out->var_values, all + v0, (nv - v0) * sizeof(fr_t));
    free(all);
}

static int finish(composer_t *c, int check, uint64_t g0, uint64_t v0, oracle_columns_t *out, uint64_t *gate_base,
                  uint64_t *var_base, uint64_t *n_gates, uint64_t *n_vars) {
    int rc = 0;
    if (check && composer_check(c) >= 0) rc = -1;
    export_columns(c, g0, v0, out);
    if (gate_base) *gate_base = g0;
    if (var_base) *var_base = v0;
    if (n_gates) *n_gates = composer_circuit_size(c) - g0;
    if (n_vars) *n_vars = composer_num_variables(c) - v0;
    composer_free(c);
    return rc;
}

int oracle_range_check_batch(fr_t min_range, fr_t max_range, const fr_t *witness, size_t batch, int check,
                             oracle_columns_t *out, uint64_t *result_vars, uint64_t *gate_base, uint64_t *var_base,
                             uint64_t *n_gates, uint64_t *n_vars) {
    composer_t *c = composer_new();
    uint64_t g0 = composer_circuit_size(c), v0 = composer_num_variables(c);
    for (size_t i = 0; i < batch; i++) {
        allocated_scalar_t w = allocated_scalar_allocate(c, witness[i]);
        var_t r = range_check(c, min_range, max_range, w);
        if (result_vars) result_vars[i] = r;
    }
    return finish(c, check, g0, v0, out, gate_base, var_base, n_gates, n_vars);
}

int oracle_max_bound_batch(const fr_t *max_range, const fr_t *witness, size_t batch, int check, oracle_columns_t *out,
                           uint64_t *result_vars, uint64_t *num_bits, uint64_t *gate_base, uint64_t *var_base,
                           uint64_t *n_gates, uint64_t *n_vars) {
    composer_t *c = composer_new();
    uint64_t g0 = composer_circuit_size(c), v0 = composer_num_variables(c);
    for (size_t i = 0; i < batch; i++) {
        allocated_scalar_t w = allocated_scalar_allocate(c, witness[i]);
        uint64_t nb;
        var_t r = max_bound(c, max_range[i], w, &nb);
        if (result_vars) result_vars[i] = r;
        if (num_bits) num_bits[i] = nb;
    }
    return finish(c, check, g0, v0, out, gate_base, var_base, n_gates, n_vars);
}

int oracle_scalar_mix_batch(const fr_t *v, const fr_t *y, const fr_t *s, const fr_t *a, const fr_t *b, size_t batch,
                            int check, oracle_columns_t *out, uint64_t *result_vars, uint8_t *err_mask,
                            uint64_t *gate_base, uint64_t *var_base, uint64_t *n_gates, uint64_t *n_vars) {
    composer_t *c = composer_new();
    uint64_t g0 = composer_circuit_size(c), v0 = composer_num_variables(c);
    for (size_t i = 0; i < batch; i++) {
        var_t vv = composer_add_input(c, v[i]);
        var_t yv = composer_add_input(c, y[i]);
        var_t sv = composer_add_input(c, s[i]);
        allocated_scalar_t aa = allocated_scalar_allocate(c, a[i]);
        allocated_scalar_t bb = allocated_scalar_allocate(c, b[i]);
        int e = is_non_zero(c, vv, v[i]);
        if (err_mask) err_mask[i] = (uint8_t)(e != GADGET_OK);
        var_t sel = conditionally_select_one(c, yv, sv);
        var_t eq = maybe_equal(c, aa, bb);
        if (result_vars) { result_vars[2 * i] = sel; result_vars[2 * i + 1] = eq; }
    }
    return finish(c, check, g0, v0, out, gate_base, var_base, n_gates, n_vars);
}
