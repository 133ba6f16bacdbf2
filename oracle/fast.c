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
    const uint64_t *witness_vars; /* NULL: the loop allocates its witness (allocated_scalar.rs:27); else the Variable it was given */
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
    if (J->out.q_m) {   /* (NULL: the caller wants the wires and the assignments only -- tests/frows_oracle.py) */
        J->out.q_m[r] = qm; J->out.q_l[r] = ql; J->out.q_r[r] = qr; J->out.q_o[r] = qo; J->out.q_c[r] = qc;
    }
    if (J->out.w_l) {   /* (NULL: the assignments only -- the witness refresh) */
        J->out.w_l[r] = a; J->out.w_r[r] = b; J->out.w_o[r] = c;
    }
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
    const uint64_t n = J->n, G = 4 * n + 11, V = 2 * n + 523 + (J->witness_vars ? 0 : 1);
    for (size_t i = J->lo; i < J->hi; i++) {
        uint64_t row = i * G, var = J->var_base + i * V;
        const fr_t xval = J->witness[i];
        uint64_t x;
        if (J->witness_vars) x = J->witness_vars[i];   /* range_check(composer, min, max, witness) as the reference has it: range.rs:27-32 */
        else {
            x = var++;
            J->out.var_values[x - J->var_base] = xval;
        }
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

static int range_check_fast(fr_t min_range, fr_t max_range, const fr_t *witness, const uint64_t *witness_vars, size_t batch,
                            uint64_t var_base, int threads, oracle_columns_t *out, uint64_t *result_vars) {
    job_t base;
    memset(&base, 0, sizeof base);
    base.witness_vars = witness_vars;
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

int oracle_range_check_fast(fr_t min_range, fr_t max_range, const fr_t *witness, size_t batch, uint64_t var_base,
                            int threads, oracle_columns_t *out, uint64_t *result_vars) {
    return range_check_fast(min_range, max_range, witness, NULL, batch, var_base, threads, out, result_vars);
}

/* for i: range_check(min, max, AllocatedScalar { var: witness_vars[i], scalar: witness[i] }) on witnesses allocated before the
 * loop -- the reference's own signature (range.rs:27-32): 4n + 11 rows and 2n + 523 Variables per item, none for the witness */
int oracle_range_check_allocated_fast(fr_t min_range, fr_t max_range, const fr_t *witness, const uint64_t *witness_vars,
                                      size_t batch, uint64_t var_base, int threads, oracle_columns_t *out,
                                      uint64_t *result_vars) {
    if (!witness_vars) return -1;
    return range_check_fast(min_range, max_range, witness, witness_vars, batch, var_base, threads, out, result_vars);
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

/* ---- the scalar gadgets and the gate calls over arrays of EXISTING Variables ---------------------------------------------------
 * The loops `for i: gadget(composer, Variable a[i], Variable b[i])` of the batched appends on Variables allocated before the call
 * (pg_composer_conditionally_select_zero_batch ... pg_composer_boolean_gate_batch): the same composer calls as oracle/gadgets.c makes of
 * /root/reference/src/scalar.rs:21-140 (and composer.c's add / mul / poly_gate), threaded over the items.  `table` holds the
 * assignments of every Variable that exists before the call ([0, var_base)); an item's rows and Variables lie at i * L / var_base + i * V,
 * or -- is_non_zero, whose items stop after one row and one Variable where the value is 0 (scalar.rs:73-80) -- at the prefix sums of
 * oracle_is_non_zero_plan.  Pinned by tests/test_oracle_fast.py against the faithful calls. */
typedef struct {
    int kind;
    const uint64_t *a, *b, *c;
    const fr_t *table;
    fr_t q[5];   /* gate kinds: q_m, q_l, q_r, q_o, q_c of the batch */
} small_args_t;

static void *small_worker(void *p) {
    const job_t *J = (const job_t *)p;
    const small_args_t *S = (const small_args_t *)J->max_range; /* (carried in a spare pointer of the shared job record) */
    fr_t *vals = J->out.var_values;
    const uint64_t vrel = J->var_base, Z = J->zero_var;
    const fr_t m1 = J->neg_one;
    static const uint32_t rows_of[] = {1, 4, 3, 3, 1, 1, 1}, vars_of[] = {1, 4, 3, 3, 1, 1, 0};
    for (size_t i = J->lo; i < J->hi; i++) {
        uint64_t r, var;
        if (J->row_off) { r = J->row_off[i]; var = J->var_base + (J->var_off[i] - J->var_off[J->first]); }
        else { r = (uint64_t)i * rows_of[S->kind]; var = J->var_base + (uint64_t)(i - J->first) * vars_of[S->kind]; }
        const uint64_t av = S->a[i], bv = S->b ? S->b[i] : 0;
        const fr_t a = S->table[av], b = S->b ? S->table[bv] : FR_ZERO;
        uint64_t res = 0;
        switch (S->kind) {
        case ORACLE_SELECT_ZERO: {   /* scalar.rs:21-27: composer.mul(1, x, select, 0) */
            const uint64_t o = var++;
            vals[o - vrel] = fr_mul(a, b);
            put_row(J, r++, av, bv, o, FR_ONE, FR_ZERO, FR_ZERO, m1, FR_ZERO);
            res = o;
            break;
        }
        case ORACLE_SELECT_ONE: {    /* scalar.rs:36-59 */
            const uint64_t one = var++, sy = var++, oms = var++, o = var++;
            vals[one - vrel] = FR_ONE;
            put_row(J, r++, one, one, one, FR_ZERO, FR_ONE, FR_ZERO, FR_ZERO, m1);           /* :41 */
            const fr_t sy_val = fr_mul(a, b), oms_val = fr_sub(FR_ONE, b);
            vals[sy - vrel] = sy_val;
            put_row(J, r++, av, bv, sy, FR_ONE, FR_ZERO, FR_ZERO, m1, FR_ZERO);              /* :43 */
            vals[oms - vrel] = oms_val;
            put_row(J, r++, one, bv, oms, FR_ZERO, FR_ONE, m1, m1, FR_ZERO);                 /* :45-50 */
            vals[o - vrel] = fr_add(sy_val, oms_val);
            put_row(J, r++, sy, oms, o, FR_ZERO, FR_ONE, FR_ONE, m1, FR_ZERO);               /* :53-58 */
            res = o;
            break;
        }
        case ORACLE_MAYBE_EQUAL: {   /* scalar.rs:105-140 */
            const uint64_t uv = var++, zv = var++, ev = var++;
            const fr_t u = fr_sub(a, b);
            fr_t z = FR_ZERO;
            if (!fr_is_zero(u)) fr_invert(u, &z);                                            /* unwrap_or(zero) :121-122 */
            vals[uv - vrel] = u;
            put_row(J, r++, av, bv, uv, FR_ZERO, FR_ONE, m1, m1, FR_ZERO);                   /* :111-117 */
            vals[zv - vrel] = z;
            vals[ev - vrel] = fr_sub(FR_ONE, fr_mul(u, z));                                  /* :126 */
            put_row(J, r++, zv, uv, ev, m1, FR_ZERO, FR_ZERO, m1, FR_ONE);
            put_row(J, r++, ev, uv, uv, FR_ONE, FR_ZERO, FR_ZERO, FR_ZERO, FR_ZERO);         /* :129-138 */
            res = ev;
            break;
        }
        case ORACLE_IS_NON_ZERO: {   /* scalar.rs:63-97, value_assigned = the Variable's own assignment */
            const uint64_t assigned = var++;
            vals[assigned - vrel] = a;
            put_row(J, r++, av, assigned, Z, FR_ZERO, FR_ONE, m1, FR_ZERO, FR_ZERO);         /* assert_equal :71 */
            if (!fr_is_zero(a)) {                                                            /* else Err(NonExistingInverse) :79 */
                const uint64_t iv = var++, one = var++;
                fr_t inv;
                fr_invert(a, &inv);
                vals[iv - vrel] = inv;                                                       /* :77 */
                vals[one - vrel] = FR_ONE;                                                   /* :83 */
                put_row(J, r++, one, one, one, FR_ZERO, FR_ONE, FR_ZERO, FR_ZERO, m1);
                put_row(J, r++, av, iv, one, FR_ONE, FR_ZERO, FR_ZERO, m1, FR_ZERO);         /* :84-94 */
            }
            break;
        }
        case ORACLE_GATE_ADD: {      /* composer.c:composer_add -- q_l a + q_r b + q_c */
            const uint64_t o = var++;
            vals[o - vrel] = fr_add(fr_add(fr_mul(S->q[1], a), fr_mul(S->q[2], b)), S->q[4]);
            put_row(J, r++, av, bv, o, FR_ZERO, S->q[1], S->q[2], m1, S->q[4]);
            res = o;
            break;
        }
        case ORACLE_GATE_MUL: {      /* composer.c:composer_mul -- q_m a b + q_c */
            const uint64_t o = var++;
            vals[o - vrel] = fr_add(fr_mul(fr_mul(S->q[0], a), b), S->q[4]);
            put_row(J, r++, av, bv, o, S->q[0], FR_ZERO, FR_ZERO, m1, S->q[4]);
            res = o;
            break;
        }
        default:                     /* composer.c:composer_poly_gate: a row on three existing Variables, nothing created */
            put_row(J, r++, av, bv, S->c[i], S->q[0], S->q[1], S->q[2], S->q[3], S->q[4]);
            break;
        }
        if (J->result_vars) J->result_vars[i - J->first] = res;
    }
    return NULL;
}

int oracle_is_non_zero_plan(const uint64_t *vars, const fr_t *table, size_t batch, uint64_t *row_off, uint64_t *var_off,
                            uint8_t *err_mask) {
    uint64_t r = 0, x = 0;
    for (size_t i = 0; i < batch; i++) {
        const int z = fr_is_zero(table[vars[i]]);
        row_off[i] = r; var_off[i] = x;
        r += z ? 1 : 3;
        x += z ? 1 : 3;
        if (err_mask) err_mask[i] = (uint8_t)z;
    }
    row_off[batch] = r; var_off[batch] = x;
    return 0;
}

int oracle_small_batch_fast(int kind, const uint64_t *a, const uint64_t *b, const uint64_t *c, const fr_t *table,
                            const fr_t *selectors /* gate kinds: q_m, q_l, q_r, q_o, q_c; else NULL */, const uint64_t *row_off,
                            const uint64_t *var_off, size_t lo, size_t hi, uint64_t var_base, uint64_t zero_var, int threads,
                            oracle_columns_t *out, uint64_t *result_vars) {
    if (kind < 0 || kind > ORACLE_GATE_ROWS || !a || !table || (kind == ORACLE_IS_NON_ZERO) != (row_off != NULL)) return -1;
    if (kind != ORACLE_IS_NON_ZERO && !b) return -1;
    if (kind == ORACLE_GATE_ROWS && !c) return -1;
    small_args_t S;
    memset(&S, 0, sizeof S);
    S.kind = kind; S.a = a; S.b = b; S.c = c; S.table = table;
    if (selectors) memcpy(S.q, selectors, sizeof S.q);
    static const uint32_t rows_of[] = {1, 4, 3, 3, 1, 1, 1}, vars_of[] = {1, 4, 3, 3, 1, 1, 0};
    job_t base;
    memset(&base, 0, sizeof base);
    base.neg_one = fr_neg(FR_ONE);
    base.max_range = (const fr_t *)&S;
    base.row_off = row_off;
    base.var_off = var_off;
    base.first = lo;
    base.row_rel = row_off ? row_off[lo] : (uint64_t)lo * rows_of[kind];
    base.var_base = var_base + (var_off ? var_off[lo] : (uint64_t)lo * vars_of[kind]);
    base.zero_var = zero_var;
    base.out = *out;
    base.result_vars = result_vars;
    run_threads(&base, lo, hi, threads, small_worker);
    return 0;
}

/* ==== the f-rows of a WHOLE circuit, from its wire columns ================================================
 *
 * oracle/composer.c keeps, like dusk-plonk 0.8's Permutation, a Variable -> [WireData] map that every row updates, and
 * composer_sigma() walks it; composer_values_dense() + the wire columns give the wire-VALUE columns the prover takes.
 * Both are a function of the four wire columns (and the assignments) alone:
 *   sigma[p]   = the next position, in recording order (gate by gate; left, right, output, fourth within a gate), that holds
 *                the Variable of position p -- the Variable's first position when p is its last;
 *   w_X_value  = values[w_X[row]];   q_arith = 1, q_range = q_logic = q_fixed = q_variable = 0 on every row push_arith_row
 *                makes (all rows of this path), q_4 = 0 except where the caller lists it (the dummy rows of composer_new()).
 * The forms below compute exactly that for row ranges of a circuit of hundreds of millions of rows, threaded, so that
 * tests/test_gpu_frows_exhaustive.py can compare EVERY word of pg_composer_permutation / pg_composer_materialize at the sizes
 * bench.py times them.  The wire columns they start from are assembled from the fast gadget forms above (tests/frows_oracle.py).
 * Pinned by tests/test_oracle_fast.py: == composer_sigma / composer_selector / composer_wire / composer_values_dense of
 * faithful composer.c programs (every gadget, foreign Variables, random gate programs), for any chunking and thread count.
 */

#include <stdio.h>

typedef struct { uint64_t *k, *v; size_t cap, len; } u64map_t; /* open addressing; key stored + 1 (0 = empty) */

static void map_init(u64map_t *m, size_t cap) {
    m->cap = cap; m->len = 0;
    m->k = (uint64_t *)calloc(cap, sizeof(uint64_t));
    m->v = (uint64_t *)malloc(cap * sizeof(uint64_t));
}
static void map_free(u64map_t *m) { free(m->k); free(m->v); m->k = m->v = NULL; m->cap = m->len = 0; }
static inline size_t map_hash(uint64_t key) { uint64_t x = key * 0x9e3779b97f4a7c15ull; return (size_t)(x ^ (x >> 31)); }
static uint64_t *map_get(const u64map_t *m, uint64_t key) {
    size_t i = map_hash(key) & (m->cap - 1);
    while (m->k[i]) {
        if (m->k[i] == key + 1) return &m->v[i];
        i = (i + 1) & (m->cap - 1);
    }
    return NULL;
}
static void map_put(u64map_t *m, uint64_t key, uint64_t val);
static void map_grow(u64map_t *m) {
    u64map_t n;
    map_init(&n, m->cap * 2);
    for (size_t i = 0; i < m->cap; i++)
        if (m->k[i]) map_put(&n, m->k[i] - 1, m->v[i]);
    map_free(m);
    *m = n;
}
static void map_put(u64map_t *m, uint64_t key, uint64_t val) {
    if ((m->len + 1) * 2 > m->cap) map_grow(m);
    size_t i = map_hash(key) & (m->cap - 1);
    while (m->k[i]) {
        if (m->k[i] == key + 1) { m->v[i] = val; return; }
        i = (i + 1) & (m->cap - 1);
    }
    m->k[i] = key + 1; m->v[i] = val; m->len++;
}

enum { SIGMA_BLOCK = 1 << 14 };
#define SIGMA_NONE UINT64_MAX

struct oracle_sigma_plan {
    const uint64_t *w[4];
    size_t n, padded, n_vars;
    int threads;
    uint64_t *first; /* recording-order key 4 * gate + wire of a Variable's first position; SIGMA_NONE: never on a wire */
    uint64_t *last;  /* key + 1 of its last position; 0: never */
    uint64_t *seen_before; /* per block of SIGMA_BLOCK rows: 1 + the largest Variable on a wire of an EARLIER row (0: none) */
    size_t n_blocks;
    u64map_t carry;  /* Variable -> its first position (sigma encoding) in the rows the chunks so far have covered */
    size_t next_r1;
    int bad;
};

typedef struct {
    struct oracle_sigma_plan *P;
    size_t a, b;           /* rows */
    uint64_t va, vb;       /* this range's own Variables: on no wire of an earlier row */
    size_t r0;             /* first row of out */
    uint64_t *const *out;
    u64map_t foreign;      /* a Variable seen here that may be on earlier rows -> its first position in [a, b) */
    uint64_t *tails; size_t n_tails, cap_tails; /* (Variable, wire, gate): last position in [a, b), but not the Variable's last */
    int bad;
} sigma_job_t;

static inline void atomic_min_u64(uint64_t *p, uint64_t x) {
    uint64_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (x < cur && !__atomic_compare_exchange_n(p, &cur, x, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
static inline void atomic_max_u64(uint64_t *p, uint64_t x) {
    uint64_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (x > cur && !__atomic_compare_exchange_n(p, &cur, x, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

/* pass 1 over blocks [a, b): first / last position of every Variable, the largest Variable of every block */
static void *sigma_scan_worker(void *p) {
    sigma_job_t *J = (sigma_job_t *)p;
    struct oracle_sigma_plan *P = J->P;
    for (size_t blk = J->a; blk < J->b; blk++) {
        const size_t g0 = blk * SIGMA_BLOCK, g1 = g0 + SIGMA_BLOCK < P->n ? g0 + SIGMA_BLOCK : P->n;
        uint64_t top = 0;
        for (size_t g = g0; g < g1; g++)
            for (int w = 0; w < 4; w++) {
                const uint64_t v = P->w[w][g];
                if (v >= P->n_vars) { J->bad = 1; continue; }
                if (v + 1 > top) top = v + 1;
                atomic_min_u64(&P->first[v], 4 * (uint64_t)g + (uint64_t)w);
            }
        P->seen_before[blk + 1] = top; /* (made a running maximum by the caller) */
    }
    for (size_t blk = J->b; blk-- > J->a;) {
        const size_t g0 = blk * SIGMA_BLOCK, g1 = g0 + SIGMA_BLOCK < P->n ? g0 + SIGMA_BLOCK : P->n;
        for (size_t g = g1; g-- > g0;)
            for (int w = 4; w-- > 0;) {
                const uint64_t v = P->w[w][g];
                if (v < P->n_vars) atomic_max_u64(&P->last[v], 4 * (uint64_t)g + (uint64_t)w + 1);
            }
    }
    return NULL;
}

struct oracle_sigma_plan *oracle_sigma_fast_begin(const uint64_t *w_l, const uint64_t *w_r, const uint64_t *w_o,
                                                  const uint64_t *w_4, size_t n, size_t padded_n, size_t n_vars, int threads) {
    if (padded_n < n || threads < 1) return NULL;
    struct oracle_sigma_plan *P = (struct oracle_sigma_plan *)calloc(1, sizeof *P);
    P->w[0] = w_l; P->w[1] = w_r; P->w[2] = w_o; P->w[3] = w_4;
    P->n = n; P->padded = padded_n; P->n_vars = n_vars; P->threads = threads;
    P->n_blocks = (n + SIGMA_BLOCK - 1) / SIGMA_BLOCK;
    P->first = (uint64_t *)malloc((n_vars ? n_vars : 1) * sizeof(uint64_t));
    P->last = (uint64_t *)calloc(n_vars ? n_vars : 1, sizeof(uint64_t));
    P->seen_before = (uint64_t *)calloc(P->n_blocks + 1, sizeof(uint64_t));
    memset(P->first, 0xff, (n_vars ? n_vars : 1) * sizeof(uint64_t));
    map_init(&P->carry, 1024);
    P->next_r1 = padded_n;
    int T = threads;
    if ((size_t)T > P->n_blocks) T = P->n_blocks ? (int)P->n_blocks : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)T);
    sigma_job_t *jobs = (sigma_job_t *)calloc((size_t)T, sizeof(sigma_job_t));
    for (int t = 0; t < T; t++) {
        jobs[t].P = P;
        jobs[t].a = P->n_blocks * (size_t)t / (size_t)T;
        jobs[t].b = P->n_blocks * (size_t)(t + 1) / (size_t)T;
        pthread_create(&th[t], NULL, sigma_scan_worker, &jobs[t]);
    }
    for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); P->bad |= jobs[t].bad; }
    free(th); free(jobs);
    for (size_t k = 1; k <= P->n_blocks; k++)
        if (P->seen_before[k] < P->seen_before[k - 1]) P->seen_before[k] = P->seen_before[k - 1];
    if (P->bad) { fprintf(stderr, "oracle_sigma_fast: a wire holds a Variable >= n_vars\n"); }
    return P;
}

void oracle_sigma_fast_end(struct oracle_sigma_plan *P) {
    if (!P) return;
    free(P->first); free(P->last); free(P->seen_before);
    map_free(&P->carry);
    free(P);
}

/* 1 + the largest Variable on a wire of rows [0, a) */
static uint64_t sigma_seen_before(const struct oracle_sigma_plan *P, size_t a) {
    const size_t blk = a / SIGMA_BLOCK;
    uint64_t top = P->seen_before[blk];
    for (size_t g = blk * SIGMA_BLOCK; g < a; g++)
        for (int w = 0; w < 4; w++)
            if (P->w[w][g] + 1 > top) top = P->w[w][g] + 1;
    return top;
}

static void *sigma_chunk_worker(void *p) {
    sigma_job_t *J = (sigma_job_t *)p;
    const struct oracle_sigma_plan *P = J->P;
    const size_t own = (size_t)(J->vb - J->va), padded = P->padded;
    uint64_t *next = (uint64_t *)malloc((own ? own : 1) * sizeof(uint64_t));
    memset(next, 0xff, (own ? own : 1) * sizeof(uint64_t));
    for (size_t g = J->b; g-- > J->a;)
        for (int w = 4; w-- > 0;) {
            const uint64_t v = P->w[w][g], here = (uint64_t)w * padded + g, key = 4 * (uint64_t)g + (uint64_t)w;
            uint64_t nx = SIGMA_NONE;
            if (v >= J->va) {               /* (v < vb: every Variable of these rows is below seen_before(b)) */
                nx = next[v - J->va];
                next[v - J->va] = here;
            } else {
                uint64_t *s = map_get(&J->foreign, v);
                if (s) { nx = *s; *s = here; } else map_put(&J->foreign, v, here);
            }
            if (nx == SIGMA_NONE) {
                if (P->last[v] == key + 1) {            /* the Variable's last position: the cycle closes on its first */
                    const uint64_t f = P->first[v];
                    nx = (f & 3) * padded + (f >> 2);
                } else {                                 /* it comes again after row b: the caller knows where */
                    if (J->n_tails == J->cap_tails) {
                        J->cap_tails = J->cap_tails ? 2 * J->cap_tails : 64;
                        J->tails = (uint64_t *)realloc(J->tails, 3 * J->cap_tails * sizeof(uint64_t));
                    }
                    uint64_t *t = &J->tails[3 * J->n_tails++];
                    t[0] = v; t[1] = (uint64_t)w; t[2] = g;
                    continue;
                }
            }
            J->out[w][g - J->r0] = nx;
        }
    free(next);
    return NULL;
}

/* sigma[w][r0 .. r1) -> out[w][0 .. r1 - r0), w = 0..3.  Chunks are asked for from the LAST row to the first: the first call
 * has r1 = padded_n, every later one r1 = the previous call's r0 (a position's successor lies at or after it -- or is the
 * Variable's first position, which pass 1 has).  Returns 0; -1 on a call out of order or an inconsistency. */
int oracle_sigma_fast_chunk(struct oracle_sigma_plan *P, size_t r0, size_t r1, uint64_t *const out[4]) {
    if (!P || P->bad || r1 != P->next_r1 || r0 >= r1) return -1;
    P->next_r1 = r0;
    for (int w = 0; w < 4; w++)      /* rows past the circuit map to themselves */
        for (size_t g = r0 > P->n ? r0 : P->n; g < r1; g++) out[w][g - r0] = (uint64_t)w * P->padded + g;
    const size_t e = r1 < P->n ? r1 : P->n;
    if (r0 >= e) return 0;
    int T = P->threads;
    if ((size_t)T > (e - r0 + 255) / 256) T = (int)((e - r0 + 255) / 256);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)T);
    sigma_job_t *jobs = (sigma_job_t *)calloc((size_t)T, sizeof(sigma_job_t));
    for (int t = 0; t < T; t++) {
        jobs[t].P = P;
        jobs[t].a = r0 + (e - r0) * (size_t)t / (size_t)T;
        jobs[t].b = r0 + (e - r0) * (size_t)(t + 1) / (size_t)T;
        jobs[t].r0 = r0;
        jobs[t].out = out;
    }
    for (int t = 0; t < T; t++) {
        jobs[t].va = sigma_seen_before(P, jobs[t].a);
        jobs[t].vb = t + 1 < T ? 0 : sigma_seen_before(P, jobs[t].b);
        if (t) jobs[t - 1].vb = jobs[t].va;
        map_init(&jobs[t].foreign, 256);
    }
    for (int t = 0; t < T; t++) pthread_create(&th[t], NULL, sigma_chunk_worker, &jobs[t]);
    for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
    int rc = 0;
    for (int t = T; t-- > 0;)        /* a range's open ends: the first position in a later range, else in a later chunk */
        for (size_t k = 0; k < jobs[t].n_tails; k++) {
            const uint64_t *tl = &jobs[t].tails[3 * k];
            const uint64_t *s = NULL;
            for (int u = t + 1; u < T && !s; u++) s = map_get(&jobs[u].foreign, tl[0]);
            if (!s) s = map_get(&P->carry, tl[0]);
            if (!s) { rc = -1; continue; }
            out[tl[1]][tl[2] - r0] = *s;
        }
    for (int t = T; t-- > 0;) {      /* descending: the smallest position of a Variable is written last */
        for (size_t i = 0; i < jobs[t].foreign.cap; i++)
            if (jobs[t].foreign.k[i]) map_put(&P->carry, jobs[t].foreign.k[i] - 1, jobs[t].foreign.v[i]);
        map_free(&jobs[t].foreign);
        free(jobs[t].tails);
    }
    free(th); free(jobs);
    if (rc) P->bad = 1;
    return rc;
}

/* ---- materialised columns of rows [r0, r1) ------------------------------------------------------------------------------ */

typedef struct {
    const uint64_t *w[4];
    const fr_t *values;
    size_t n_vars, r0, a, b;
    oracle_full_columns_t out;
    int bad;
} mat_job_t;

static void *mat_worker(void *p) {
    mat_job_t *J = (mat_job_t *)p;
    fr_t *const vals[4] = {J->out.w_l_value, J->out.w_r_value, J->out.w_o_value, J->out.w_4_value};
    for (size_t g = J->a; g < J->b; g++) {
        const size_t k = g - J->r0;
        J->out.q_4[k] = FR_ZERO;           /* composer.c:push_arith_row; the caller lists the rows whose q_4 is not 0 */
        J->out.q_arith[k] = FR_ONE;
        J->out.q_range[k] = FR_ZERO; J->out.q_logic[k] = FR_ZERO;
        J->out.q_fixed_group_add[k] = FR_ZERO; J->out.q_variable_group_add[k] = FR_ZERO;
        J->out.w_4[k] = J->w[3][g];
        for (int w = 0; w < 4; w++) {
            const uint64_t v = J->w[w][g];
            if (v >= J->n_vars) { J->bad = 1; vals[w][k] = FR_ZERO; continue; }
            vals[w][k] = J->values[v];
        }
    }
    return NULL;
}

int oracle_materialize_fast(const uint64_t *w_l, const uint64_t *w_r, const uint64_t *w_o, const uint64_t *w_4,
                            const fr_t *values, size_t n_vars, const uint64_t *q4_rows, const fr_t *q4_values, size_t n_q4,
                            size_t r0, size_t r1, int threads, const oracle_full_columns_t *out) {
    if (r1 < r0 || threads < 1) return -1;
    int T = threads;
    if ((size_t)T > (r1 - r0 + 255) / 256) T = r1 > r0 ? (int)((r1 - r0 + 255) / 256) : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)T);
    mat_job_t *jobs = (mat_job_t *)calloc((size_t)T, sizeof(mat_job_t));
    for (int t = 0; t < T; t++) {
        mat_job_t *J = &jobs[t];
        J->w[0] = w_l; J->w[1] = w_r; J->w[2] = w_o; J->w[3] = w_4;
        J->values = values; J->n_vars = n_vars; J->r0 = r0; J->out = *out;
        J->a = r0 + (r1 - r0) * (size_t)t / (size_t)T;
        J->b = r0 + (r1 - r0) * (size_t)(t + 1) / (size_t)T;
        pthread_create(&th[t], NULL, mat_worker, J);
    }
    int bad = 0;
    for (int t = 0; t < T; t++) { pthread_join(th[t], NULL); bad |= jobs[t].bad; }
    free(th); free(jobs);
    for (size_t i = 0; i < n_q4; i++)
        if (q4_rows[i] >= r0 && q4_rows[i] < r1) out->q_4[q4_rows[i] - r0] = q4_values[i];
    return bad ? -1 : 0;
}
