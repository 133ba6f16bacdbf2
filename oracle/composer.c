/*
 * oracle/composer.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).  See composer.h.
 */
#include "composer.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- growable columns ("Vec::push") ---------------------------------- */
typedef struct { fr_t *p; size_t len, cap; } fr_vec;
typedef struct { var_t *p; size_t len, cap; } var_vec;

static void die(const char *m) { fprintf(stderr, "oracle/composer: %s\n", m); abort(); }

static void fr_push(fr_vec *v, fr_t x) {
    if (v->len == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 16;
        v->p = (fr_t *)realloc(v->p, v->cap * sizeof(fr_t));
        if (!v->p) die("out of memory");
    }
    v->p[v->len++] = x;
}
static void var_push(var_vec *v, var_t x) {
    if (v->len == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 16;
        v->p = (var_t *)realloc(v->p, v->cap * sizeof(var_t));
        if (!v->p) die("out of memory");
    }
    v->p[v->len++] = x;
}

/* ---- Variable -> slot hash map (hashbrown stand-in, open addressing) -- */
typedef struct { uint8_t wire; uint64_t gate; } wire_data; /* WireData::{Left,Right,Output,Fourth}(gate) */
typedef struct { wire_data *p; uint32_t len, cap; } wd_vec;
typedef struct {
    var_t key;
    int used;
    fr_t value;  /* variables: HashMap<Variable, BlsScalar> */
    wd_vec perm; /* perm.variable_map: HashMap<Variable, Vec<WireData>> */
} slot_t;
typedef struct { slot_t *s; size_t cap, len; } map_t;

static inline size_t hash_var(var_t k) {
    uint64_t x = k * 0x9e3779b97f4a7c15ull;
    return (size_t)(x ^ (x >> 29));
}
static slot_t *map_find(const map_t *m, var_t k) {
    if (!m->cap) return NULL;
    size_t i = hash_var(k) & (m->cap - 1);
    while (m->s[i].used) {
        if (m->s[i].key == k) return &m->s[i];
        i = (i + 1) & (m->cap - 1);
    }
    return NULL;
}
static void map_grow(map_t *m) {
    size_t ncap = m->cap ? 2 * m->cap : 64;
    slot_t *ns = (slot_t *)calloc(ncap, sizeof(slot_t));
    if (!ns) die("out of memory");
    for (size_t i = 0; i < m->cap; i++) {
        if (!m->s[i].used) continue;
        size_t j = hash_var(m->s[i].key) & (ncap - 1);
        while (ns[j].used) j = (j + 1) & (ncap - 1);
        ns[j] = m->s[i];
    }
    free(m->s);
    m->s = ns;
    m->cap = ncap;
}
static slot_t *map_insert(map_t *m, var_t k) {
    if ((m->len + 1) * 4 > m->cap * 3) map_grow(m);
    size_t i = hash_var(k) & (m->cap - 1);
    while (m->s[i].used) i = (i + 1) & (m->cap - 1);
    m->s[i].used = 1;
    m->s[i].key = k;
    m->len++;
    return &m->s[i];
}

struct composer {
    size_t n;
    fr_vec sel[PG_NSEL];
    var_vec wire[PG_NWIRE];
    /* public_inputs_sparse_store: BTreeMap<usize, BlsScalar> (rarely used) */
    size_t *pi_pos; fr_t *pi_val; size_t pi_len, pi_cap;
    var_t zero_var;
    map_t vars;
    size_t next_var;
};

static slot_t *must_find(const composer_t *c, var_t v) {
    slot_t *s = map_find(&c->vars, v);
    if (!s) die("unknown Variable (the reference would panic on the HashMap index)");
    return s;
}

size_t composer_circuit_size(const composer_t *c) { return c->n; }
size_t composer_num_variables(const composer_t *c) { return c->next_var; }
var_t composer_zero_var(const composer_t *c) { return c->zero_var; }
fr_t composer_value(const composer_t *c, var_t v) { return must_find(c, v)->value; }
const fr_t *composer_selector(const composer_t *c, int col) { return c->sel[col].p; }
const var_t *composer_wire(const composer_t *c, int col) { return c->wire[col].p; }
size_t composer_perm_count(const composer_t *c, var_t v) { return must_find(c, v)->perm.len; }

void composer_values_dense(const composer_t *c, fr_t *out) {
    for (size_t v = 0; v < c->next_var; v++) out[v] = must_find(c, v)->value;
}

void composer_dense_pi(const composer_t *c, fr_t *out) {
    for (size_t i = 0; i < c->n; i++) out[i] = FR_ZERO;
    for (size_t i = 0; i < c->pi_len; i++) out[c->pi_pos[i]] = c->pi_val[i];
}

/* add_input: perm.new_variable() (fresh Vec::with_capacity(16) in the map) +
 * variables.insert(var, s) */
var_t composer_add_input(composer_t *c, fr_t s) {
    var_t v = c->next_var++;
    slot_t *sl = map_insert(&c->vars, v);
    sl->value = s;
    sl->perm.cap = 16;
    sl->perm.len = 0;
    sl->perm.p = (wire_data *)malloc(16 * sizeof(wire_data));
    if (!sl->perm.p) die("out of memory");
    return v;
}

static void perm_add(composer_t *c, var_t v, uint8_t wire, uint64_t gate) {
    slot_t *s = must_find(c, v);
    if (s->perm.len == s->perm.cap) {
        s->perm.cap *= 2;
        s->perm.p = (wire_data *)realloc(s->perm.p, s->perm.cap * sizeof(wire_data));
        if (!s->perm.p) die("out of memory");
    }
    s->perm.p[s->perm.len].wire = wire;
    s->perm.p[s->perm.len].gate = gate;
    s->perm.len++;
}

static void set_pi(composer_t *c, const fr_t *pi) {
    if (!pi) return;
    if (c->pi_len == c->pi_cap) {
        c->pi_cap = c->pi_cap ? 2 * c->pi_cap : 8;
        c->pi_pos = (size_t *)realloc(c->pi_pos, c->pi_cap * sizeof(size_t));
        c->pi_val = (fr_t *)realloc(c->pi_val, c->pi_cap * sizeof(fr_t));
    }
    c->pi_pos[c->pi_len] = c->n;
    c->pi_val[c->pi_len] = *pi;
    c->pi_len++;
}

/* one arithmetic row: 15 column pushes + 4 permutation updates + n += 1 */
static void push_arith_row(composer_t *c, var_t a, var_t b, var_t o, var_t d, fr_t q_m, fr_t q_l, fr_t q_r, fr_t q_o,
                           fr_t q_c, fr_t q_4, const fr_t *pi) {
    var_push(&c->wire[PG_W_L], a);
    var_push(&c->wire[PG_W_R], b);
    var_push(&c->wire[PG_W_O], o);
    var_push(&c->wire[PG_W_4], d);
    fr_push(&c->sel[PG_Q_M], q_m);
    fr_push(&c->sel[PG_Q_L], q_l);
    fr_push(&c->sel[PG_Q_R], q_r);
    fr_push(&c->sel[PG_Q_O], q_o);
    fr_push(&c->sel[PG_Q_C], q_c);
    fr_push(&c->sel[PG_Q_4], q_4);
    fr_push(&c->sel[PG_Q_ARITH], FR_ONE);
    fr_push(&c->sel[PG_Q_RANGE], FR_ZERO);
    fr_push(&c->sel[PG_Q_LOGIC], FR_ZERO);
    fr_push(&c->sel[PG_Q_FIXED], FR_ZERO);
    fr_push(&c->sel[PG_Q_VARIABLE], FR_ZERO);
    set_pi(c, pi);
    perm_add(c, a, 0, c->n);
    perm_add(c, b, 1, c->n);
    perm_add(c, o, 2, c->n);
    perm_add(c, d, 3, c->n);
    c->n++;
}

void composer_poly_gate(composer_t *c, var_t a, var_t b, var_t o, fr_t q_m, fr_t q_l, fr_t q_r, fr_t q_o, fr_t q_c,
                        const fr_t *pi) {
    push_arith_row(c, a, b, o, c->zero_var, q_m, q_l, q_r, q_o, q_c, FR_ZERO, pi);
}

void composer_constrain_to_constant(composer_t *c, var_t a, fr_t constant, const fr_t *pi) {
    composer_poly_gate(c, a, a, a, FR_ZERO, FR_ONE, FR_ZERO, FR_ZERO, fr_neg(constant), pi);
}

var_t composer_add_witness_to_circuit_description(composer_t *c, fr_t v) {
    var_t var = composer_add_input(c, v);
    composer_constrain_to_constant(c, var, v, NULL);
    return var;
}

void composer_assert_equal(composer_t *c, var_t a, var_t b) {
    composer_poly_gate(c, a, b, c->zero_var, FR_ZERO, FR_ONE, fr_neg(FR_ONE), FR_ZERO, FR_ZERO, NULL);
}

/* add = big_add with no fourth advice wire: q_4 = 0 on zero_var */
var_t composer_add(composer_t *c, fr_t q_l, var_t a, fr_t q_r, var_t b, fr_t q_c, const fr_t *pi) {
    fr_t q_4 = FR_ZERO, q_o = fr_neg(FR_ONE);
    var_t d = c->zero_var;
    fr_t a_eval = must_find(c, a)->value, b_eval = must_find(c, b)->value, d_eval = must_find(c, d)->value;
    fr_t c_eval = fr_add(fr_add(fr_add(fr_mul(q_l, a_eval), fr_mul(q_r, b_eval)), fr_mul(q_4, d_eval)), q_c);
    if (pi) c_eval = fr_add(c_eval, *pi);
    var_t o = composer_add_input(c, c_eval);
    push_arith_row(c, a, b, o, d, FR_ZERO, q_l, q_r, q_o, q_c, q_4, pi);
    return o;
}

/* mul = big_mul with no fourth advice wire */
var_t composer_mul(composer_t *c, fr_t q_m, var_t a, var_t b, fr_t q_c, const fr_t *pi) {
    fr_t q_4 = FR_ZERO, q_o = fr_neg(FR_ONE);
    var_t d = c->zero_var;
    fr_t a_eval = must_find(c, a)->value, b_eval = must_find(c, b)->value, d_eval = must_find(c, d)->value;
    fr_t c_eval = fr_add(fr_add(fr_mul(fr_mul(q_m, a_eval), b_eval), fr_mul(q_4, d_eval)), q_c);
    if (pi) c_eval = fr_add(c_eval, *pi);
    var_t o = composer_add_input(c, c_eval);
    push_arith_row(c, a, b, o, d, q_m, FR_ZERO, FR_ZERO, q_o, q_c, q_4, pi);
    return o;
}

void composer_mul_gate(composer_t *c, var_t a, var_t b, var_t o, fr_t q_m, fr_t q_o, fr_t q_c, const fr_t *pi) {
    push_arith_row(c, a, b, o, c->zero_var, q_m, FR_ZERO, FR_ZERO, q_o, q_c, FR_ZERO, pi);
}

var_t composer_boolean_gate(composer_t *c, var_t a) {
    push_arith_row(c, a, a, a, c->zero_var, FR_ONE, FR_ZERO, FR_ZERO, fr_neg(FR_ONE), FR_ZERO, FR_ZERO, NULL);
    return a;
}

static void add_dummy_constraints(composer_t *c) {
    var_t six = composer_add_input(c, fr_from_u64(6));
    var_t one = composer_add_input(c, fr_from_u64(1));
    var_t seven = composer_add_input(c, fr_from_u64(7));
    var_t min_twenty = composer_add_input(c, fr_neg(fr_from_u64(20)));
    push_arith_row(c, six, seven, min_twenty, one, fr_from_u64(1), fr_from_u64(2), fr_from_u64(3), fr_from_u64(4),
                   fr_from_u64(4), FR_ONE, NULL);
    push_arith_row(c, min_twenty, six, seven, c->zero_var, fr_from_u64(1), fr_from_u64(1), fr_from_u64(1),
                   fr_from_u64(1), fr_from_u64(127), FR_ZERO, NULL);
}

composer_t *composer_new_without_dummy(void) {
    composer_t *c = (composer_t *)calloc(1, sizeof(composer_t));
    if (!c) die("out of memory");
    c->zero_var = 0;
    c->zero_var = composer_add_witness_to_circuit_description(c, FR_ZERO);
    return c;
}

composer_t *composer_new(void) {
    composer_t *c = composer_new_without_dummy();
    add_dummy_constraints(c);
    return c;
}

void composer_free(composer_t *c) {
    if (!c) return;
    for (int i = 0; i < PG_NSEL; i++) free(c->sel[i].p);
    for (int i = 0; i < PG_NWIRE; i++) free(c->wire[i].p);
    for (size_t i = 0; i < c->vars.cap; i++)
        if (c->vars.s[i].used) free(c->vars.s[i].perm.p);
    free(c->vars.s);
    free(c->pi_pos);
    free(c->pi_val);
    free(c);
}

void composer_sigma(const composer_t *c, size_t padded_n, uint64_t *out) {
    for (size_t w = 0; w < 4; w++)
        for (size_t i = 0; i < padded_n; i++) out[w * padded_n + i] = w * padded_n + i;
    for (size_t v = 0; v < c->next_var; v++) {
        const slot_t *s = must_find(c, v);
        for (uint32_t k = 0; k < s->perm.len; k++) {
            const wire_data cur = s->perm.p[k], nxt = s->perm.p[(k + 1 == s->perm.len) ? 0 : k + 1];
            out[(size_t)cur.wire * padded_n + cur.gate] = (uint64_t)nxt.wire * padded_n + nxt.gate;
        }
    }
}

long composer_check(const composer_t *c) {
    fr_t *pi = (fr_t *)malloc((c->n ? c->n : 1) * sizeof(fr_t));
    composer_dense_pi(c, pi);
    long bad = -1;
    for (size_t i = 0; i < c->n && bad < 0; i++) {
        fr_t a = composer_value(c, c->wire[PG_W_L].p[i]), b = composer_value(c, c->wire[PG_W_R].p[i]);
        fr_t o = composer_value(c, c->wire[PG_W_O].p[i]), d = composer_value(c, c->wire[PG_W_4].p[i]);
        fr_t t = fr_mul(fr_mul(c->sel[PG_Q_M].p[i], a), b);
        t = fr_add(t, fr_mul(c->sel[PG_Q_L].p[i], a));
        t = fr_add(t, fr_mul(c->sel[PG_Q_R].p[i], b));
        t = fr_add(t, fr_mul(c->sel[PG_Q_O].p[i], o));
        t = fr_add(t, fr_mul(c->sel[PG_Q_4].p[i], d));
        t = fr_add(t, c->sel[PG_Q_C].p[i]);
        t = fr_add(t, pi[i]);
        t = fr_mul(t, c->sel[PG_Q_ARITH].p[i]);
        if (!fr_is_zero(t)) bad = (long)i;
    }
    free(pi);
    return bad;
}
