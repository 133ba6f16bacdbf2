/* tests/cpp/oracle_sanitize.c -- the CPU oracle (the checker of every parity test) under -fsanitize=address,undefined:
 * every batch driver, the fast threaded form, the full composer with sigma / dense PI, on small inputs.
 * Built together with oracle/*.c and run by tests/test_fr_host_sanitizers.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gadgets.h"

static uint64_t s = 0x2545F4914F6CDD1Dull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static fr_t rnd_fr(void) {
    uint64_t raw[4] = {rnd(), rnd(), rnd(), rnd() % 0x73eda753299d7d48ull};
    return fr_from_raw(raw);
}

static oracle_columns_t alloc_cols(size_t g, size_t v) {
    oracle_columns_t c;
    c.q_m = malloc(g * 32); c.q_l = malloc(g * 32); c.q_r = malloc(g * 32); c.q_o = malloc(g * 32); c.q_c = malloc(g * 32);
    c.w_l = malloc(g * 8); c.w_r = malloc(g * 8); c.w_o = malloc(g * 8);
    c.var_values = malloc(v * 32);
    return c;
}
static void free_cols(oracle_columns_t *c) {
    free(c->q_m); free(c->q_l); free(c->q_r); free(c->q_o); free(c->q_c); free(c->w_l); free(c->w_r); free(c->w_o); free(c->var_values);
}

int main(void) {
    enum { B = 24 };
    fr_t wit[B], bounds[B], v[B], y[B], sel[B], a[B], b[B];
    for (int i = 0; i < B; i++) {
        wit[i] = i % 3 ? fr_from_u64(40000 + 9000 * (uint64_t)i) : rnd_fr();
        bounds[i] = fr_from_u64(1 + (rnd() >> (rnd() % 60)));
        v[i] = i == 5 ? fr_from_u64(0) : rnd_fr();
        y[i] = rnd_fr(); sel[i] = fr_from_u64(i & 1); a[i] = rnd_fr(); b[i] = i % 2 ? a[i] : rnd_fr();
    }
    uint64_t gb, vb, ng, nv, res[2 * B], nbits[B];
    uint8_t err[B];
    const fr_t mn = fr_from_u64(50000), mx = fr_from_u64(250000);
    /* range_check: sizes from a dry run, then with output */
    if (oracle_range_check_batch(mn, mx, wit, B, 1, NULL, NULL, &gb, &vb, &ng, &nv)) return 1;
    oracle_columns_t c = alloc_cols(ng, nv), f = alloc_cols(ng, nv);
    if (oracle_range_check_batch(mn, mx, wit, B, 1, &c, res, &gb, &vb, &ng, &nv)) return 2;
    if (oracle_range_check_fast(mn, mx, wit, B, vb, 3, &f, res + B)) return 3;
    if (memcmp(c.q_c, f.q_c, ng * 32) || memcmp(c.w_o, f.w_o, ng * 8) || memcmp(c.var_values, f.var_values, nv * 32) ||
        memcmp(res, res + B, B * 8)) return 4;
    free_cols(&c); free_cols(&f);
    if (oracle_max_bound_batch(bounds, wit, B, 1, NULL, NULL, NULL, &gb, &vb, &ng, &nv)) return 5;
    c = alloc_cols(ng, nv);
    if (oracle_max_bound_batch(bounds, wit, B, 1, &c, res, nbits, &gb, &vb, &ng, &nv)) return 6;
    {   /* the threaded ragged form, in two chunks at the numbering of the whole batch == the faithful loop */
        uint64_t *nb2 = malloc(B * 8), *ro = malloc((B + 1) * 8), *vo = malloc((B + 1) * 8), *r2 = malloc(B * 8);
        if (oracle_max_bound_plan(bounds, B, 3, nb2, ro, vo) || memcmp(nb2, nbits, B * 8) || ro[B] != ng || vo[B] != nv) return 20;
        const size_t cut = B / 3;
        for (int part = 0; part < 2; part++) {
            const size_t lo = part ? cut : 0, hi = part ? B : cut;
            oracle_columns_t g = alloc_cols(ro[hi] - ro[lo], vo[hi] - vo[lo]);
            if (oracle_max_bound_fast(bounds, wit, nb2, ro, vo, lo, hi, vb, 3, &g, r2 + lo)) return 21;
            if (memcmp(g.q_c, c.q_c + ro[lo], (ro[hi] - ro[lo]) * 32) || memcmp(g.w_r, c.w_r + ro[lo], (ro[hi] - ro[lo]) * 8) ||
                memcmp(g.var_values, c.var_values + vo[lo], (vo[hi] - vo[lo]) * 32)) return 22;
            free_cols(&g);
        }
        if (memcmp(r2, res, B * 8)) return 23;
        free(nb2); free(ro); free(vo); free(r2);
    }
    free_cols(&c);
    oracle_scalar_mix_batch(v, y, sel, a, b, B, 0, NULL, NULL, NULL, &gb, &vb, &ng, &nv);
    c = alloc_cols(ng, nv);
    oracle_scalar_mix_batch(v, y, sel, a, b, B, 0, &c, res, err, &gb, &vb, &ng, &nv);
    if (err[5] != 1) return 7;
    {   /* the threaded fused mix (batch inversion per block of items), chunked */
        uint64_t *ro = malloc((B + 1) * 8), *vo = malloc((B + 1) * 8), *r2 = malloc(2 * B * 8);
        uint8_t *e2 = malloc(B);
        if (oracle_scalar_mix_plan(v, B, ro, vo, e2) || memcmp(e2, err, B) || ro[B] != ng || vo[B] != nv) return 24;
        const size_t cut = B / 2 + 1;
        for (int part = 0; part < 2; part++) {
            const size_t lo = part ? cut : 0, hi = part ? B : cut;
            oracle_columns_t g = alloc_cols(ro[hi] - ro[lo], vo[hi] - vo[lo]);
            if (oracle_scalar_mix_fast(v, y, sel, a, b, ro, vo, lo, hi, vb, 0, 2, &g, r2 + 2 * lo)) return 25;
            if (memcmp(g.q_m, c.q_m + ro[lo], (ro[hi] - ro[lo]) * 32) || memcmp(g.w_o, c.w_o + ro[lo], (ro[hi] - ro[lo]) * 8) ||
                memcmp(g.var_values, c.var_values + vo[lo], (vo[hi] - vo[lo]) * 32)) return 26;
            free_cols(&g);
        }
        if (memcmp(r2, res, 2 * B * 8)) return 27;
        free(ro); free(vo); free(r2); free(e2);
    }
    free_cols(&c);
    /* one composer with everything, then sigma and the dense public inputs */
    composer_t *cs = composer_new();
    const fr_t pi = fr_from_u64(11);
    var_t x = composer_add_input(cs, fr_from_u64(9));
    var_t r = range_check(cs, mn, mx, allocated_scalar_allocate(cs, fr_from_u64(60000)));
    if (is_non_zero(cs, x, fr_from_u64(9)) != GADGET_OK) return 8;
    var_t yv = conditionally_select_one(cs, x, r);
    composer_constrain_to_constant(cs, yv, fr_from_u64(20), &pi);
    var_t m = maybe_equal(cs, allocated_scalar_allocate(cs, fr_from_u64(5)), allocated_scalar_allocate(cs, fr_from_u64(5)));
    (void)conditionally_select_zero(cs, m, r);
    (void)scalar_decomposition_gadget(cs, 9, allocated_scalar_allocate(cs, fr_from_u64(300)), NULL);
    (void)composer_boolean_gate(cs, m);
    if (composer_check(cs) != -1) return 9;
    const size_t n = composer_circuit_size(cs);
    size_t padded = 1;
    while (padded < n) padded <<= 1;
    uint64_t *sigma = malloc(4 * padded * 8);
    fr_t *dense = malloc(n * 32);
    composer_sigma(cs, padded, sigma);
    composer_dense_pi(cs, dense);
    uint64_t acc = 0;
    for (size_t i = 0; i < 4 * padded; i++) acc ^= sigma[i] ^ i;   /* a permutation of 0..4*padded-1 */
    {   /* the threaded f-row forms on this composer's own wire columns: sigma in ragged chunks from the last row down, and the
         * materialised columns == composer_sigma / the composer's columns and table */
        const size_t nvars = composer_num_variables(cs);
        const uint64_t *w[4] = {composer_wire(cs, PG_W_L), composer_wire(cs, PG_W_R), composer_wire(cs, PG_W_O), composer_wire(cs, PG_W_4)};
        fr_t *table = malloc(nvars * 32);
        composer_values_dense(cs, table);
        oracle_sigma_plan_t *sp = oracle_sigma_fast_begin(w[0], w[1], w[2], w[3], n, padded, nvars, 3);
        if (!sp) return 30;
        uint64_t *chunk = malloc(4 * 97 * 8);
        for (size_t r1 = padded; r1 > 0;) {
            const size_t r0 = r1 > 97 ? r1 - 97 : 0;
            uint64_t *const out[4] = {chunk, chunk + 97, chunk + 2 * 97, chunk + 3 * 97};
            if (oracle_sigma_fast_chunk(sp, r0, r1, out)) return 31;
            for (int k = 0; k < 4; k++)
                if (memcmp(out[k], sigma + (size_t)k * padded + r0, (r1 - r0) * 8)) return 32;
            r1 = r0;
        }
        oracle_sigma_fast_end(sp);
        free(chunk);
        oracle_full_columns_t fc;
        fr_t *buf = malloc(10 * n * 32);
        fc.q_4 = buf; fc.q_arith = buf + n; fc.q_range = buf + 2 * n; fc.q_logic = buf + 3 * n; fc.q_fixed_group_add = buf + 4 * n;
        fc.q_variable_group_add = buf + 5 * n; fc.w_l_value = buf + 6 * n; fc.w_r_value = buf + 7 * n; fc.w_o_value = buf + 8 * n;
        fc.w_4_value = buf + 9 * n;
        fc.w_4 = malloc(n * 8);
        const uint64_t q4_row = 1;
        const fr_t q4_val = composer_selector(cs, PG_Q_4)[1];
        if (oracle_materialize_fast(w[0], w[1], w[2], w[3], table, nvars, &q4_row, &q4_val, 1, 0, n, 4, &fc)) return 33;
        if (memcmp(fc.q_4, composer_selector(cs, PG_Q_4), n * 32) || memcmp(fc.q_arith, composer_selector(cs, PG_Q_ARITH), n * 32) ||
            memcmp(fc.q_logic, composer_selector(cs, PG_Q_LOGIC), n * 32) || memcmp(fc.w_4, w[3], n * 8)) return 34;
        for (size_t i = 0; i < n; i++)
            if (memcmp(&fc.w_o_value[i], &table[w[2][i]], 32) || memcmp(&fc.w_4_value[i], &table[w[3][i]], 32)) return 35;
        free(buf); free(fc.w_4); free(table);
    }
    {   /* range_check on witnesses allocated before the loop: the threaded form == the faithful calls */
        composer_t *c2 = composer_new();
        allocated_scalar_t al[B];
        uint64_t wv[B], r1[B], r2[B];
        for (int i = 0; i < B; i++) { al[i] = allocated_scalar_allocate(c2, wit[i]); wv[i] = al[i].var; }
        const size_t g0 = composer_circuit_size(c2), v0 = composer_num_variables(c2);
        for (int i = 0; i < B; i++) r1[i] = range_check(c2, mn, mx, al[i]);
        const size_t g1 = composer_circuit_size(c2), v1 = composer_num_variables(c2);
        oracle_columns_t g = alloc_cols(g1 - g0, v1 - v0);
        if (oracle_range_check_allocated_fast(mn, mx, wit, wv, B, v0, 3, &g, r2)) return 36;
        fr_t *t2 = malloc(v1 * 32);
        composer_values_dense(c2, t2);
        if (memcmp(g.w_l, composer_wire(c2, PG_W_L) + g0, (g1 - g0) * 8) || memcmp(g.q_c, composer_selector(c2, PG_Q_C) + g0, (g1 - g0) * 32) ||
            memcmp(g.var_values, t2 + v0, (v1 - v0) * 32) || memcmp(r1, r2, B * 8)) return 37;
        free(t2); free_cols(&g);
        composer_free(c2);
    }
    free(sigma); free(dense);
    composer_free(cs);
    printf("oracle under sanitizers: ok (%zu rows, xor %llu)\n", n, (unsigned long long)acc);
    return acc != 0;
}
