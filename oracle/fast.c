/*
 * oracle/fast.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, fast form).
 *
 * The same rows and variables as oracle/gadgets.c:oracle_range_check_batch -- i.e. as the reference loop
 *     for w in witnesses { AllocatedScalar::allocate(composer, w); range_check(composer, min, max, w) }
 * (/root/reference/src/allocated_scalar.rs:27, src/range.rs:27-158, src/scalar.rs:105-140) -- but written for speed:
 * flat output arrays at closed-form offsets, a table of mont(2^i) instead of the per-bit `pow` of range.rs:146, the
 * accumulator carried by one modular addition per bit (range.rs:152), no hash maps, and the items split over POSIX
 * threads.  It exists (a) as the "best-case CPU" figure bench.py reports beside the faithful port, and (b) as one
 * more independent statement of the path that is fast enough to check thousands of n = 255 items limb for limb.
 * Pinned by tests/test_oracle_fast.py against oracle/gadgets.c.
 */
#include "gadgets.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    fr_t qc_max, qc_min, neg_one, pow2[256];
    uint64_t n, gate_base, var_base;
    const fr_t *witness;
    oracle_columns_t out;
    uint64_t *result_vars;
    size_t lo, hi;
} job_t;

static void put_row(const job_t *J, uint64_t r, uint64_t a, uint64_t b, uint64_t c, fr_t qm, fr_t ql, fr_t qr, fr_t qo,
                    fr_t qc) {
    J->out.q_m[r] = qm; J->out.q_l[r] = ql; J->out.q_r[r] = qr; J->out.q_o[r] = qo; J->out.q_c[r] = qc;
    J->out.w_l[r] = a; J->out.w_r[r] = b; J->out.w_o[r] = c;
}

/* one bound block: add row + scalar_decomposition_gadget + maybe_equal; returns the Variable of y and its value */
static uint64_t bound_block(const job_t *J, uint64_t *row, uint64_t *var, uint64_t x, fr_t T, fr_t ql0, fr_t qc0, int *y_out) {
    const uint64_t n = J->n, vrel = J->var_base;
    fr_t *vals = J->out.var_values;
    uint64_t r = *row, v = *var;
    const uint64_t Tv = v++;                     /* composer.add -> new variable T */
    vals[Tv - vrel] = T;
    put_row(J, r++, x, x, Tv, FR_ZERO, ql0, FR_ZERO, J->neg_one, qc0);
    uint8_t bits[256];
    scalar_to_bits(T, bits);
    const uint64_t b0 = v;                       /* 256 bit variables */
    for (int i = 0; i < 256; i++) vals[v++ - vrel] = bits[i] ? FR_ONE : FR_ZERO;
    uint64_t acc_var = v++;                      /* accumulator = constant 0 */
    vals[acc_var - vrel] = FR_ZERO;
    put_row(J, r++, acc_var, acc_var, acc_var, FR_ZERO, FR_ONE, FR_ZERO, FR_ZERO, FR_ZERO);
    fr_t acc = FR_ZERO;
    for (uint64_t i = 0; i < n; i++) {
        put_row(J, r++, b0 + i, b0 + i, b0 + i, FR_ONE, FR_ZERO, FR_ZERO, J->neg_one, FR_ZERO);
        if (bits[i]) acc = fr_add(acc, J->pow2[i]);
        const uint64_t nv = v++;
        vals[nv - vrel] = acc;
        put_row(J, r++, b0 + i, acc_var, nv, FR_ZERO, J->pow2[i], FR_ONE, J->neg_one, FR_ZERO);
        acc_var = nv;
    }
    /* maybe_equal(accumulator, T) */
    const fr_t u = fr_sub(acc, T);
    const int y = fr_is_zero(u);
    fr_t z = FR_ZERO;
    if (!y) fr_invert(u, &z);  /* the exponentiation only when an inverse exists */
    const uint64_t uv = v++, zv = v++, yv = v++;
    vals[uv - vrel] = u; vals[zv - vrel] = z; vals[yv - vrel] = y ? FR_ONE : FR_ZERO;
    put_row(J, r++, acc_var, Tv, uv, FR_ZERO, FR_ONE, J->neg_one, J->neg_one, FR_ZERO);
    put_row(J, r++, zv, uv, yv, J->neg_one, FR_ZERO, FR_ZERO, J->neg_one, FR_ONE);
    put_row(J, r++, yv, uv, uv, FR_ONE, FR_ZERO, FR_ZERO, FR_ZERO, FR_ZERO);
    *row = r; *var = v; *y_out = y;
    return yv;
}

static void *worker(void *p) {
    const job_t *J = (const job_t *)p;
    const uint64_t n = J->n, G = 4 * n + 11, V = 2 * n + 524;
    for (size_t i = J->lo; i < J->hi; i++) {
        uint64_t row = i * G, var = J->var_base + i * V;
        const fr_t xval = J->witness[i];
        const uint64_t x = var++;
        J->out.var_values[x - J->var_base] = xval;
        int y1, y2;
        const uint64_t yv1 = bound_block(J, &row, &var, x, fr_sub(J->qc_max, xval), J->neg_one, J->qc_max, &y1);
        const uint64_t yv2 = bound_block(J, &row, &var, x, fr_add(xval, J->qc_min), FR_ONE, J->qc_min, &y2);
        const uint64_t rv = var++;
        J->out.var_values[rv - J->var_base] = (y1 && y2) ? FR_ONE : FR_ZERO;
        put_row(J, row++, yv1, yv2, rv, FR_ONE, FR_ZERO, FR_ZERO, J->neg_one, FR_ZERO);
        if (J->result_vars) J->result_vars[i] = rv;
    }
    return NULL;
}

int oracle_range_check_fast(fr_t min_range, fr_t max_range, const fr_t *witness, size_t batch, uint64_t var_base,
                            int threads, oracle_columns_t *out, uint64_t *result_vars) {
    job_t base;
    memset(&base, 0, sizeof base);
    base.qc_max = fr_sub(max_range, FR_ONE);
    base.qc_min = fr_neg(min_range);
    base.neg_one = fr_neg(FR_ONE);
    base.n = num_bits_closest_power_of_two(base.qc_max);
    base.pow2[0] = FR_ONE;
    for (int i = 1; i < 256; i++) base.pow2[i] = fr_add(base.pow2[i - 1], base.pow2[i - 1]);
    base.var_base = var_base;
    base.witness = witness;
    base.out = *out;
    base.result_vars = result_vars;
    if (threads < 1) threads = 1;
    if ((size_t)threads > batch) threads = batch ? (int)batch : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    job_t *jobs = (job_t *)malloc(sizeof(job_t) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = base;
        jobs[t].lo = batch * (size_t)t / (size_t)threads;
        jobs[t].hi = batch * (size_t)(t + 1) / (size_t)threads;
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
    return 0;
}
