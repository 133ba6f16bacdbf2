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
 *
 * oracle_max_bound_fast / oracle_scalar_mix_fast are the same thing for the loops of oracle_max_bound_batch
 * (allocate + max_bound with a bound per item: src/range.rs:82-113) and oracle_scalar_mix_batch (five add_input +
 * is_non_zero + conditionally_select_one + maybe_equal: src/scalar.rs:36-140), which are ragged: the caller passes the
 * prefix sums of rows and variables per item (oracle_max_bound_plan / oracle_scalar_mix_plan) and an item range
 * [lo, hi), so that a batch of 2^20 items can be produced chunk by chunk at its global numbering -- what the
 * exhaustive full-size parity tests stream to the GPU (tests/test_gpu_exhaustive.py).
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
    /* ragged forms: inputs indexed by the absolute item, outputs relative to item `first` of the call */
    const fr_t *max_range, *in_v, *in_y, *in_s, *in_a, *in_b;
    const uint64_t *num_bits, *row_off, *var_off;
    uint64_t row_rel, zero_var;
    size_t first;
} job_t;

static void put_row(const job_t *J, uint64_t r, uint64_t a, uint64_t b, uint64_t c, fr_t qm, fr_t ql, fr_t qr, fr_t qo,
                    fr_t qc) {
    r -= J->row_rel;
    J->out.q_m[r] = qm; J->out.q_l[r] = ql; J->out.q_r[r] = qr; J->out.q_o[r] = qo; J->out.q_c[r] = qc;
    J->out.w_l[r] = a; J->out.w_r[r] = b; J->out.w_o[r] = c;
}

/* one bound block: add row + scalar_decomposition_gadget + maybe_equal; returns the Variable of y and its value */
static uint64_t bound_block(const job_t *J, uint64_t n, uint64_t *row, uint64_t *var, uint64_t x, fr_t T, fr_t ql0, fr_t qc0,
                            int *y_out) {
    const uint64_t vrel = J->var_base;
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
        const uint64_t yv1 = bound_block(J, n, &row, &var, x, fr_sub(J->qc_max, xval), J->neg_one, J->qc_max, &y1);
        const uint64_t yv2 = bound_block(J, n, &row, &var, x, fr_add(xval, J->qc_min), FR_ONE, J->qc_min, &y2);
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

/* ---- shared driver: split [lo, hi) over threads ------------------------------------------------------------- */

static void run_threads(const job_t *base, size_t lo, size_t hi, int threads, void *(*fn)(void *)) {
    const size_t count = hi - lo;
    if (threads < 1) threads = 1;
    if ((size_t)threads > count) threads = count ? (int)count : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    job_t *jobs = (job_t *)malloc(sizeof(job_t) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = *base;
        jobs[t].lo = lo + count * (size_t)t / (size_t)threads;
        jobs[t].hi = lo + count * (size_t)(t + 1) / (size_t)threads;
        pthread_create(&th[t], NULL, fn, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

static void fill_pow2(job_t *J) {
    J->neg_one = fr_neg(FR_ONE);
    J->pow2[0] = FR_ONE;
    for (int i = 1; i < 256; i++) J->pow2[i] = fr_add(J->pow2[i - 1], J->pow2[i - 1]);
}

/* ---- max_bound with a bound per item (oracle_max_bound_batch's loop) --------------------------------------- */

typedef struct { const fr_t *max_range; uint64_t *num_bits; size_t lo, hi; } plan_job_t;

static void *plan_worker(void *p) {
    const plan_job_t *J = (const plan_job_t *)p;
    for (size_t i = J->lo; i < J->hi; i++)   /* range.rs:87-90 */
        J->num_bits[i] = num_bits_closest_power_of_two(fr_sub(J->max_range[i], FR_ONE));
    return NULL;
}

int oracle_max_bound_plan(const fr_t *max_range, size_t batch, int threads, uint64_t *num_bits, uint64_t *row_off,
                          uint64_t *var_off) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > batch) threads = batch ? (int)batch : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    plan_job_t *jobs = (plan_job_t *)malloc(sizeof(plan_job_t) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t].max_range = max_range;
        jobs[t].num_bits = num_bits;
        jobs[t].lo = batch * (size_t)t / (size_t)threads;
        jobs[t].hi = batch * (size_t)(t + 1) / (size_t)threads;
        pthread_create(&th[t], NULL, plan_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
    uint64_t r = 0, v = 0;
    for (size_t i = 0; i < batch; i++) {     /* allocate: 1 variable; max_bound: 2n+5 rows, n+261 variables */
        row_off[i] = r; var_off[i] = v;
        r += 2 * num_bits[i] + 5;
        v += num_bits[i] + 262;
    }
    row_off[batch] = r; var_off[batch] = v;
    return 0;
}

static void *max_bound_worker(void *p) {
    const job_t *J = (const job_t *)p;
    for (size_t i = J->lo; i < J->hi; i++) {
        uint64_t row = J->row_off[i], var = J->var_base + (J->var_off[i] - J->var_off[J->first]);
        const fr_t xval = J->witness[i];
        const uint64_t x = var++;                                   /* allocated_scalar.rs:27 */
        J->out.var_values[x - J->var_base] = xval;
        const fr_t m1 = fr_sub(J->max_range[i], FR_ONE);           /* range.rs:87 */
        int y;
        const uint64_t yv = bound_block(J, J->num_bits[i], &row, &var, x, fr_sub(m1, xval), J->neg_one, m1, &y);
        if (J->result_vars) J->result_vars[i - J->first] = yv;
    }
    return NULL;
}

int oracle_max_bound_fast(const fr_t *max_range, const fr_t *witness, const uint64_t *num_bits, const uint64_t *row_off,
                          const uint64_t *var_off, size_t lo, size_t hi, uint64_t var_base, int threads,
                          oracle_columns_t *out, uint64_t *result_vars) {
    job_t base;
    memset(&base, 0, sizeof base);
    fill_pow2(&base);
    base.max_range = max_range;
    base.witness = witness;
    base.num_bits = num_bits;
    base.row_off = row_off;
    base.var_off = var_off;
    base.first = lo;
    base.row_rel = row_off[lo];
    base.var_base = var_base + var_off[lo];   /* Variable of the chunk's first item = index 0 of out->var_values */
    base.out = *out;
    base.result_vars = result_vars;
    run_threads(&base, lo, hi, threads, max_bound_worker);
    return 0;
}

/* ---- the fused mix (oracle_scalar_mix_batch's loop) ------------------------------------------------------- */

int oracle_scalar_mix_plan(const fr_t *v, size_t batch, uint64_t *row_off, uint64_t *var_off, uint8_t *err_mask) {
    uint64_t r = 0, x = 0;
    for (size_t i = 0; i < batch; i++) {     /* is_non_zero stops after 1 variable + 1 row when v == 0: scalar.rs:69-79 */
        const int z = fr_is_zero(v[i]);
        row_off[i] = r; var_off[i] = x;
        r += z ? 8 : 10;
        x += z ? 13 : 15;
        if (err_mask) err_mask[i] = (uint8_t)z;
    }
    row_off[batch] = r; var_off[batch] = x;
    return 0;
}

/* inverses of e[0..m) (zeros stay zero: unwrap_or(zero), scalar.rs:122) by Montgomery's trick -- ONE exponentiation per
 * block instead of one per element; values are field elements kept fully reduced, so the limbs are those of fr_invert
 * (tests/test_oracle_fast.py compares them with the faithful form, which inverts one by one) */
static void invert_block(const fr_t *e, fr_t *inv, fr_t *prefix, size_t m) {
    fr_t run = FR_ONE;
    for (size_t k = 0; k < m; k++) {
        prefix[k] = run;
        if (!fr_is_zero(e[k])) run = fr_mul(run, e[k]);
    }
    fr_t r;
    fr_invert(run, &r);
    for (size_t k = m; k-- > 0;) {
        if (fr_is_zero(e[k])) { inv[k] = FR_ZERO; continue; }
        inv[k] = fr_mul(r, prefix[k]);
        r = fr_mul(r, e[k]);
    }
}

enum { MIX_BLOCK = 128 };

static void *mix_worker(void *p) {
    const job_t *J = (const job_t *)p;
    fr_t *vals = J->out.var_values;
    const uint64_t vrel = J->var_base, Z = J->zero_var;
    const fr_t m1 = J->neg_one;
    fr_t e[2 * MIX_BLOCK], einv[2 * MIX_BLOCK], prefix[2 * MIX_BLOCK];
    for (size_t i0 = J->lo; i0 < J->hi; i0 += MIX_BLOCK) {
    const size_t i1 = i0 + MIX_BLOCK < J->hi ? i0 + MIX_BLOCK : J->hi;
    for (size_t i = i0; i < i1; i++) {
        e[2 * (i - i0)] = J->in_v[i];                                /* is_non_zero inverts v: scalar.rs:73 */
        e[2 * (i - i0) + 1] = fr_sub(J->in_a[i], J->in_b[i]);        /* maybe_equal inverts a - b: scalar.rs:121-122 */
    }
    invert_block(e, einv, prefix, 2 * (i1 - i0));
    for (size_t i = i0; i < i1; i++) {
        uint64_t r = J->row_off[i], var = J->var_base + (J->var_off[i] - J->var_off[J->first]);
        const fr_t v = J->in_v[i], y = J->in_y[i], s = J->in_s[i], a = J->in_a[i], b = J->in_b[i];
        const uint64_t vv = var++, yv = var++, sv = var++, av = var++, bv = var++;  /* five add_input */
        vals[vv - vrel] = v; vals[yv - vrel] = y; vals[sv - vrel] = s; vals[av - vrel] = a; vals[bv - vrel] = b;
        /* is_non_zero(vv, v): scalar.rs:63-97 */
        const uint64_t assigned = var++;
        vals[assigned - vrel] = v;
        put_row(J, r++, vv, assigned, Z, FR_ZERO, FR_ONE, m1, FR_ZERO, FR_ZERO);             /* assert_equal :71 */
        if (!fr_is_zero(v)) {                                                                /* else Err: :79 */
            const uint64_t iv = var++, one = var++;
            vals[iv - vrel] = einv[2 * (i - i0)];                                            /* :77 */
            vals[one - vrel] = FR_ONE;                                                       /* :83 */
            put_row(J, r++, one, one, one, FR_ZERO, FR_ONE, FR_ZERO, FR_ZERO, m1);
            put_row(J, r++, vv, iv, one, FR_ONE, FR_ZERO, FR_ZERO, m1, FR_ZERO);             /* :84-94 */
        }
        /* conditionally_select_one(yv, sv): scalar.rs:36-59 */
        const uint64_t one = var++, sy = var++, oms = var++, sel = var++;
        vals[one - vrel] = FR_ONE;
        put_row(J, r++, one, one, one, FR_ZERO, FR_ONE, FR_ZERO, FR_ZERO, m1);               /* :41 */
        const fr_t sy_val = fr_mul(y, s), oms_val = fr_sub(FR_ONE, s);
        vals[sy - vrel] = sy_val;
        put_row(J, r++, yv, sv, sy, FR_ONE, FR_ZERO, FR_ZERO, m1, FR_ZERO);                  /* :43 */
        vals[oms - vrel] = oms_val;
        put_row(J, r++, one, sv, oms, FR_ZERO, FR_ONE, m1, m1, FR_ZERO);                     /* :45-50 */
        vals[sel - vrel] = fr_add(sy_val, oms_val);
        put_row(J, r++, sy, oms, sel, FR_ZERO, FR_ONE, FR_ONE, m1, FR_ZERO);                 /* :53-58 */
        /* maybe_equal(a, b): scalar.rs:105-140 */
        const uint64_t uv = var++, zv = var++, ev = var++;
        const fr_t u = e[2 * (i - i0) + 1], z = einv[2 * (i - i0) + 1];
        vals[uv - vrel] = u;
        put_row(J, r++, av, bv, uv, FR_ZERO, FR_ONE, m1, m1, FR_ZERO);                       /* :111-117 */
        vals[zv - vrel] = z;                                                                 /* :121-123 */
        vals[ev - vrel] = fr_sub(FR_ONE, fr_mul(u, z));                                      /* :126 */
        put_row(J, r++, zv, uv, ev, m1, FR_ZERO, FR_ZERO, m1, FR_ONE);
        put_row(J, r++, ev, uv, uv, FR_ONE, FR_ZERO, FR_ZERO, FR_ZERO, FR_ZERO);             /* :129-138 */
        if (J->result_vars) {
            J->result_vars[2 * (i - J->first)] = sel;
            J->result_vars[2 * (i - J->first) + 1] = ev;
        }
    }
    }
    return NULL;
}

int oracle_scalar_mix_fast(const fr_t *v, const fr_t *y, const fr_t *s, const fr_t *a, const fr_t *b,
                           const uint64_t *row_off, const uint64_t *var_off, size_t lo, size_t hi, uint64_t var_base,
                           uint64_t zero_var, int threads, oracle_columns_t *out, uint64_t *result_vars) {
    job_t base;
    memset(&base, 0, sizeof base);
    base.neg_one = fr_neg(FR_ONE);
    base.in_v = v; base.in_y = y; base.in_s = s; base.in_a = a; base.in_b = b;
    base.row_off = row_off;
    base.var_off = var_off;
    base.first = lo;
    base.row_rel = row_off[lo];
    base.var_base = var_base + var_off[lo];
    base.zero_var = zero_var;
    base.out = *out;
    base.result_vars = result_vars;
    run_threads(&base, lo, hi, threads, mix_worker);
    return 0;
}
