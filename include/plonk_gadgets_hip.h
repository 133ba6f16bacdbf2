/*
 * plonk_gadgets_hip.h -- C ABI of libplonk_gadgets_hip.so, the MI355X (gfx950)
 * batched constraint-evaluation engine for the gadget hot path of
 * dusk-network/plonk_gadgets.
 *
 * The reference has no FFI of its own: it is a safe-Rust library whose whole
 * interface is six public gadget functions + `AllocatedScalar` + `Error`
 * (/root/reference/src/lib.rs:37-45), every one taking
 * `&mut StandardComposer`.  Each entry point below names the reference
 * function(s) it replaces; INTEGRATION.md shows the Rust `extern "C"` shim a
 * maintainer would add on the reference side.
 *
 * Conventions
 *   - plain C types only; no HIP or torch types in any signature.
 *   - `pg_scalar` is bit-identical to BlsScalar's inner `[u64; 4]`
 *     (Montgomery limbs, little-endian, fully reduced).
 *   - pointers named `d_*` and every pointer inside `pg_columns` are DEVICE
 *     pointers on the engine's GPU; everything else is host memory.
 *     (Speed, not correctness: the emitters line their wave stores up with
 *     128-byte lines from the address of the FIRST array of each kind; arrays
 *     that all start on lines -- hipMalloc's do -- are written 11-17 % faster
 *     than arrays that start at different offsets inside a line.)
 *   - the caller allocates and frees every output buffer (sizes come from the
 *     `pg_*_layout` / `pg_*_plan` queries); the engine owns only its constant
 *     table and grow-only scratch (inverses of the current call, prefix-sum
 *     temporaries).  Because that scratch is per engine, calls on one engine
 *     must be issued on ONE stream at a time (one engine per stream/thread).
 *     A call on another stream than the previous one is ordered behind it by
 *     an event recorded on the previous stream: keep a stream alive until the
 *     next call on the engine has been issued (a stream that has been
 *     destroyed costs that call a device synchronisation instead).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     Batch calls enqueue and return; synchronise the stream (or call
 *     pg_engine_sync) before reading results.
 *   - every function returns a pg_status; nothing aborts.  Conditions on
 *     which the reference panics map to PG_ERR_INVALID_ARGUMENT.
 *   - an engine handle is not thread-safe (the reference is single-threaded
 *     per composer by its `&mut` borrow, e.g. src/range.rs:27-32): use one
 *     handle per host thread.
 */
#ifndef PLONK_GADGETS_HIP_H
#define PLONK_GADGETS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pg_status {
    PG_OK = 0,
    PG_ERR_NON_EXISTING_INVERSE = 1, /* Error::NonExistingInverse, src/errors.rs:17 (raised at src/scalar.rs:79) */
    PG_ERR_INVALID_ARGUMENT = 2,     /* NULL / misaligned pointer, num_bits > 256 (src/range.rs:134 panics), ... */
    PG_ERR_NO_DEVICE = 3,            /* no gfx950 device / HIP runtime failure at engine creation */
    PG_ERR_HIP = 4,                  /* a HIP call failed; pg_last_error() has the text */
    PG_ERR_CAPACITY = 5,             /* composer buffers too small for the append */
    PG_ERR_BAD_ENCODING = 6          /* pg_scalars_from_canonical_batch: some 32-byte encodings are >= q (BlsScalar::from_bytes -> Err) */
} pg_status;

typedef struct pg_scalar { uint64_t l[4]; } pg_scalar; /* BlsScalar */
typedef uint64_t pg_variable;                         /* Variable(usize) */

/* AllocatedScalar, src/allocated_scalar.rs:17-23.  The Rust struct has no
 * #[repr(C)]; a shim converts field by field. */
typedef struct pg_allocated_scalar {
    pg_variable var;
    pg_scalar scalar;
} pg_allocated_scalar;

/* The live columns a gate row occupies on this path (SURVEY.md section 8a row
 * a14) plus the variable-assignment table.  Row r of a batch call is gate
 * `gate_base + r`; entry v of var_values is Variable(var_base + v).  Every
 * row has q_4 = 0, q_arith = 1, the other selectors 0 and w_4 = zero_var;
 * those constant columns are not materialised here (pg_composer does). */
typedef struct pg_columns {
    pg_scalar *q_m, *q_l, *q_r, *q_o, *q_c; /* device, 16-byte aligned */
    uint64_t *w_l, *w_r, *w_o;              /* device, 8-byte aligned   */
    pg_scalar *var_values;                  /* device, 16-byte aligned */
} pg_columns;

typedef struct pg_layout {
    uint64_t num_bits;       /* ladder length n (0 for gadgets without a ladder / ragged batches) */
    uint64_t gates_per_item; /* rows one item emits (0 if ragged)      */
    uint64_t vars_per_item;  /* variables one item creates (0 if ragged) */
    uint64_t n_gates;        /* total rows of the batch                 */
    uint64_t n_vars;         /* total variables of the batch            */
} pg_layout;

typedef struct pg_engine pg_engine;

/* ---- engine ----------------------------------------------------------- */
pg_status pg_engine_create(int device, pg_engine **out);
void pg_engine_destroy(pg_engine *e);
pg_status pg_engine_sync(pg_engine *e, void *stream);
const char *pg_status_string(pg_status s);
const char *pg_last_error(void);
/* "gfx950" etc.; what the library was built for and what it runs on */
const char *pg_build_arch(void);

/* ---- BlsScalar helpers (host) ------------------------------------------
 * what a host needs to form the public arguments: BlsScalar::from(u64),
 * from_raw/to canonical, neg/sub/add/mul, and the two pure functions of
 * src/range.rs:173-189. */
void pg_scalar_from_u64(uint64_t v, pg_scalar *out);
void pg_scalar_from_canonical(const uint64_t raw[4], pg_scalar *out);
void pg_scalar_to_canonical(const pg_scalar *s, uint64_t raw[4]);
void pg_scalar_add(const pg_scalar *a, const pg_scalar *b, pg_scalar *out);
void pg_scalar_sub(const pg_scalar *a, const pg_scalar *b, pg_scalar *out);
void pg_scalar_neg(const pg_scalar *a, pg_scalar *out);
void pg_scalar_mul(const pg_scalar *a, const pg_scalar *b, pg_scalar *out);
/* BlsScalar::invert (used at src/scalar.rs:73,121): PG_ERR_NON_EXISTING_INVERSE and *out = 0 for zero.  The same
 * division-step inversion the device pre-pass runs; pg_scalar_invert_fermat is a^(q-2), its independent cross-check. */
pg_status pg_scalar_invert(const pg_scalar *a, pg_scalar *out);
pg_status pg_scalar_invert_fermat(const pg_scalar *a, pg_scalar *out);
uint64_t pg_bits_count(const pg_scalar *s);                    /* src/range.rs:173-181 */
uint64_t pg_num_bits_closest_power_of_two(const pg_scalar *s); /* src/range.rs:185-189 */

/* ---- the data format on the way in and out -------------------------------------------
 * A BlsScalar travels as 32 little-endian bytes of its canonical value (dusk-bytes Serializable, what
 * src/range.rs:162 reads back with to_bytes); the kernels want Montgomery limbs.  Both conversions in bulk, device to
 * device, 16-byte aligned buffers of 32 * batch bytes:
 *   from_canonical  BlsScalar::from_bytes per element; an encoding >= q (from_bytes returns Err) becomes 0, is flagged in
 *                   d_bad_mask (may be NULL) and counted; PG_ERR_BAD_ENCODING if any -- the output is complete all the
 *                   same -- while PG_ERR_INVALID_ARGUMENT keeps meaning a bad pointer (synchronises `stream`)
 *   to_canonical    BlsScalar::to_bytes per element */
pg_status pg_scalars_from_canonical_batch(pg_engine *e, const void *d_bytes, uint64_t batch, pg_scalar *d_out,
                                          uint8_t *d_bad_mask, uint64_t *bad_count /* may be NULL */, void *stream);
pg_status pg_scalars_to_canonical_batch(pg_engine *e, const pg_scalar *d_scalars, uint64_t batch, void *d_bytes, void *stream);

/* ---- range gadgets, batched --------------------------------------------
 * pg_range_check_batch: for every witness i, in order,
 *     w = AllocatedScalar::allocate(composer, witness[i]);   src/allocated_scalar.rs:27
 *     result[i] = range_check(composer, min, max, w);        src/range.rs:27-43
 * emitting 4n+11 rows and 2n+524 variables per witness, rows and variables
 * numbered exactly as that loop numbers them starting at (gate_base,
 * var_base).  n = num_bits_closest_power_of_two(max - 1). */
pg_status pg_range_check_layout(const pg_scalar *min_range, const pg_scalar *max_range, uint64_t batch,
                                pg_layout *out);
pg_status pg_range_check_batch(pg_engine *e, const pg_scalar *min_range, const pg_scalar *max_range,
                               const pg_scalar *d_witness, uint64_t batch, uint64_t gate_base, uint64_t var_base,
                               const pg_columns *out, pg_variable *d_result_vars /* may be NULL */, void *stream);

/* The rows of pg_range_check_batch WITHOUT the witnesses: selectors and wire indices are a function of the public
 * bounds and of the numbering alone, so a process that is sent only another process's variable table (32 B per
 * variable instead of 184 B per row on top) regenerates the rest locally.  out->var_values is not written and may be
 * NULL.  The result Variable of item i is var_base + (i + 1) * vars_per_item - 1. */
pg_status pg_range_check_structure_batch(pg_engine *e, const pg_scalar *min_range, const pg_scalar *max_range, uint64_t batch,
                                         uint64_t gate_base, uint64_t var_base, const pg_columns *out, void *stream);

/* ---- witness refresh: the variable assignments of a batched call, and nothing else ---------------------------------
 * The reference's prover flow builds a circuit once, preprocesses it, and then rebuilds the SAME circuit with other witnesses
 * (prover.clear_witness(); gadget calls again; prove -- tests/scalar_gadgets_tests.rs:108-119, 168-177): selectors, wire
 * indices and numbering are a function of the public inputs (the verifier builds them from other witnesses altogether, :36 vs
 * :43), only the assignments change.  A *_values_batch call writes exactly what the full call writes into out->var_values --
 * d_var_values[k] = the assignment of the call's k-th variable, limb for limb -- and touches no row: 32 B per variable instead
 * of 184 B per row on top (33 of the 223 KB a 256-bit range_check item weighs).  Numbering does not enter (no gate_base /
 * var_base: an assignment does not depend on its Variable's index). */
pg_status pg_range_check_values_batch(pg_engine *e, const pg_scalar *min_range, const pg_scalar *max_range,
                                      const pg_scalar *d_witness, uint64_t batch, pg_scalar *d_var_values, void *stream);
pg_status pg_max_bound_values_batch(pg_engine *e, const pg_scalar *max_range, const pg_scalar *d_witness, uint64_t batch,
                                    pg_scalar *d_var_values, void *stream);
/* ragged: the plan of the bounds (pg_max_bound_ragged_plan) is public structure and is reused as it is */
pg_status pg_max_bound_ragged_values_batch(pg_engine *e, const pg_scalar *d_max_range, const pg_scalar *d_witness, uint64_t batch,
                                           const uint32_t *d_num_bits, const uint64_t *d_row_off, const uint64_t *d_var_off,
                                           pg_scalar *d_var_values, void *stream);
/* the fused mix: an item's SHAPE depends on its witness (is_non_zero stops at v = 0, src/scalar.rs:79), so the refresh plans
 * again -- d_row_off / d_var_off / d_err_mask (may be NULL) are OUTPUTS as in pg_scalar_mix_planned_batch, the totals come
 * from pg_plan_result -- and the caller compares them with the circuit it preprocessed: other totals or another error mask
 * mean another circuit. */
pg_status pg_scalar_mix_values_batch(pg_engine *e, const pg_scalar *d_v, const pg_scalar *d_y, const pg_scalar *d_s,
                                     const pg_scalar *d_a, const pg_scalar *d_b, uint64_t batch, uint64_t *d_row_off,
                                     uint64_t *d_var_off, uint8_t *d_err_mask /* may be NULL */, pg_scalar *d_var_values,
                                     void *stream);

/* pg_range_check_allocated_batch: the gadget alone, on witnesses that are ALREADY allocated --
 *     result[i] = range_check(composer, min, max, AllocatedScalar { var: d_witness_var[i], scalar: d_witness[i] });
 * 4n+11 rows and 2n+523 variables per item (no allocate). */
pg_status pg_range_check_allocated_batch(pg_engine *e, const pg_scalar *min_range, const pg_scalar *max_range,
                                         const pg_variable *d_witness_var, const pg_scalar *d_witness, uint64_t batch,
                                         uint64_t gate_base, uint64_t var_base, const pg_columns *out,
                                         pg_variable *d_result_vars /* may be NULL */, void *stream);

/* scalar_decomposition_gadget(composer, num_bits, witness) -> (is_equal, bit Variables)   src/range.rs:119-158
 * (private in the reference, exercised by its unit test at :205-233): per item 2*num_bits+4 rows and num_bits+260
 * variables; the bit Variables of item i are var_base + i*(num_bits+260) + [0, num_bits), is_equal is the item's
 * last variable.  num_bits > 256 -> PG_ERR_INVALID_ARGUMENT (the reference panics on the slice, :134). */
pg_status pg_scalar_decomposition_layout(uint64_t num_bits, uint64_t batch, pg_layout *out);
pg_status pg_scalar_decomposition_batch(pg_engine *e, uint64_t num_bits, const pg_variable *d_witness_var,
                                        const pg_scalar *d_witness, uint64_t batch, uint64_t gate_base, uint64_t var_base,
                                        const pg_columns *out, pg_variable *d_result_vars /* may be NULL */, void *stream);

/* pg_max_bound_batch: one public bound for the whole batch.  For every witness i,
 *     w = AllocatedScalar::allocate(composer, witness[i]);
 *     (result[i], n) = max_bound(composer, max_range, w);             src/range.rs:82-113
 * 2n+5 rows and n+262 variables per witness; layout->num_bits is the u64 the reference returns. */
pg_status pg_max_bound_layout(const pg_scalar *max_range, uint64_t batch, pg_layout *out);
pg_status pg_max_bound_batch(pg_engine *e, const pg_scalar *max_range, const pg_scalar *d_witness, uint64_t batch,
                             uint64_t gate_base, uint64_t var_base, const pg_columns *out,
                             pg_variable *d_result_vars /* may be NULL */, void *stream);

/* same, on already-allocated witnesses: 2n+5 rows and n+261 variables per item */
pg_status pg_max_bound_allocated_batch(pg_engine *e, const pg_scalar *max_range, const pg_variable *d_witness_var,
                                       const pg_scalar *d_witness, uint64_t batch, uint64_t gate_base, uint64_t var_base,
                                       const pg_columns *out, pg_variable *d_result_vars /* may be NULL */, void *stream);

/* Ragged max_bound: one public bound PER ITEM (device array), so the ladder length n_i and with it the rows
 * (2 n_i + 5) and variables (n_i + 262) of an item depend on public data.  The plan computes n_i
 * (src/range.rs:87-90) and the exclusive prefix sums of rows / variables on the device, returns the totals
 * (it synchronises `stream`), and the batch call emits at those offsets.
 * Caller-allocated device buffers: d_num_bits[batch] (u32), d_row_off[batch+1], d_var_off[batch+1] (u64). */
pg_status pg_max_bound_ragged_plan(pg_engine *e, const pg_scalar *d_max_range, uint64_t batch, uint32_t *d_num_bits,
                                   uint64_t *d_row_off, uint64_t *d_var_off, pg_layout *out, void *stream);
pg_status pg_max_bound_ragged_batch(pg_engine *e, const pg_scalar *d_max_range, const pg_scalar *d_witness,
                                    uint64_t batch, const uint32_t *d_num_bits, const uint64_t *d_row_off,
                                    const uint64_t *d_var_off, uint64_t gate_base, uint64_t var_base,
                                    const pg_columns *out, pg_variable *d_result_vars /* may be NULL */, void *stream);

/* Asynchronous plans: enqueue and return at once; the totals (and the error count) land in engine-owned pinned host
 * memory and are read with pg_plan_result once the stream has been synchronised.  For callers that pre-allocate
 * worst-case buffers (10 rows / 15 variables per mix item, 515 rows / 517 variables per max_bound item) and do not
 * want a host round trip between the plan and the emit call. */
pg_status pg_max_bound_ragged_plan_async(pg_engine *e, const pg_scalar *d_max_range, uint64_t batch, uint32_t *d_num_bits,
                                         uint64_t *d_row_off, uint64_t *d_var_off, void *stream);
pg_status pg_scalar_mix_plan_async(pg_engine *e, const pg_scalar *d_v, uint64_t batch, uint64_t *d_row_off,
                                   uint64_t *d_var_off, uint8_t *d_err_mask /* may be NULL */, void *stream);
/* totals of the engine's most recent plan; PG_ERR_NON_EXISTING_INVERSE when that plan found failing items */
pg_status pg_plan_result(pg_engine *e, pg_layout *out, uint64_t *err_count /* may be NULL */);

/* ---- scalar gadgets, batched ----------------------------------------------
 * The stand-alone gadgets take EXISTING Variables: d_*_var are their indices, d_*_val their assignments (what
 * the reference reads from composer.variables).  Each emits, per item i and in order, the rows/variables of
 *   conditionally_select_zero(composer, x, select) -> Variable          src/scalar.rs:21-27   (1 row, 1 var)
 *   conditionally_select_one(composer, y, selector) -> Variable         src/scalar.rs:36-59   (4 rows, 4 vars)
 *   maybe_equal(composer, a, b) -> Variable                             src/scalar.rs:105-140 (3 rows, 3 vars)
 *   is_non_zero(composer, var, value_assigned) -> Result<(), Error>     src/scalar.rs:63-97   (3 rows, 3 vars) */
pg_status pg_conditionally_select_zero_batch(pg_engine *e, const pg_variable *d_x_var, const pg_scalar *d_x_val,
                                             const pg_variable *d_select_var, const pg_scalar *d_select_val,
                                             uint64_t batch, uint64_t gate_base, uint64_t var_base,
                                             const pg_columns *out, pg_variable *d_result_vars, void *stream);
pg_status pg_conditionally_select_one_batch(pg_engine *e, const pg_variable *d_y_var, const pg_scalar *d_y_val,
                                            const pg_variable *d_selector_var, const pg_scalar *d_selector_val,
                                            uint64_t batch, uint64_t gate_base, uint64_t var_base,
                                            const pg_columns *out, pg_variable *d_result_vars, void *stream);
pg_status pg_maybe_equal_batch(pg_engine *e, const pg_variable *d_a_var, const pg_scalar *d_a_val,
                               const pg_variable *d_b_var, const pg_scalar *d_b_val, uint64_t batch,
                               uint64_t gate_base, uint64_t var_base, const pg_columns *out,
                               pg_variable *d_result_vars, void *stream);
/* is_non_zero returns Err(NonExistingInverse) AFTER one variable and one row were pushed (src/scalar.rs:69-79):
 * an item whose value is 0 therefore occupies 1 row / 1 variable instead of 3 / 3 and the batch is ragged.
 * The plan writes the per-item error mask (1 = Err), the prefix sums and the totals; it returns
 * PG_ERR_NON_EXISTING_INVERSE when any item failed (the layout is still valid and the batch may be emitted:
 * it reproduces a loop that records the error and carries on) and synchronises `stream`. */
pg_status pg_is_non_zero_plan(pg_engine *e, const pg_scalar *d_value_assigned, uint64_t batch, uint64_t *d_row_off,
                              uint64_t *d_var_off, uint8_t *d_err_mask /* may be NULL */, pg_layout *out,
                              uint64_t *err_count /* may be NULL */, void *stream);
pg_status pg_is_non_zero_batch(pg_engine *e, const pg_variable *d_var, const pg_scalar *d_value_assigned,
                               uint64_t batch, const uint64_t *d_row_off, const uint64_t *d_var_off,
                               uint64_t gate_base, uint64_t var_base, pg_variable zero_var, const pg_columns *out,
                               void *stream);

/* The fused mix (BASELINE.json config 3), one launch: per item
 *     v, y, s = composer.add_input x3; a, b = AllocatedScalar::allocate x2;
 *     is_non_zero(composer, v_var, v); conditionally_select_one(composer, y_var, s_var); maybe_equal(composer, a, b)
 * 10 rows + 15 variables per item (8 + 13 where v = 0).  d_result_vars[2i] = select_one's Variable,
 * d_result_vars[2i+1] = maybe_equal's.  Plan semantics as pg_is_non_zero_plan. */
pg_status pg_scalar_mix_plan(pg_engine *e, const pg_scalar *d_v, uint64_t batch, uint64_t *d_row_off,
                             uint64_t *d_var_off, uint8_t *d_err_mask /* may be NULL */, pg_layout *out,
                             uint64_t *err_count /* may be NULL */, void *stream);
pg_status pg_scalar_mix_batch(pg_engine *e, const pg_scalar *d_v, const pg_scalar *d_y, const pg_scalar *d_s,
                              const pg_scalar *d_a, const pg_scalar *d_b, uint64_t batch, const uint64_t *d_row_off,
                              const uint64_t *d_var_off, uint64_t gate_base, uint64_t var_base, pg_variable zero_var,
                              const pg_columns *out, pg_variable *d_result_vars /* may be NULL */, void *stream);
/* Plan and emit in ONE call, for callers whose buffers hold the worst case (`out`: 10 rows and 15 variables per item;
 * d_row_off / d_var_off: batch + 1 entries, written by the call).  Same result as pg_scalar_mix_plan_async followed by
 * pg_scalar_mix_batch on the same stream; the totals and the error count are read with pg_plan_result after the stream has
 * been synchronised.  What the single call buys: no plan launch at all -- the launch that inverts reads every v anyway and
 * makes the prefix sums on its way (~7 % of a 2^20-item step). */
pg_status pg_scalar_mix_planned_batch(pg_engine *e, const pg_scalar *d_v, const pg_scalar *d_y, const pg_scalar *d_s,
                                      const pg_scalar *d_a, const pg_scalar *d_b, uint64_t batch, uint64_t *d_row_off,
                                      uint64_t *d_var_off, uint8_t *d_err_mask /* may be NULL */, uint64_t gate_base,
                                      uint64_t var_base, pg_variable zero_var, const pg_columns *out,
                                      pg_variable *d_result_vars /* may be NULL */, void *stream);

/* ---- pg_composer: the composer-gate API surface, device resident ----------------------------
 * The counterpart of the slice of dusk-plonk's `StandardComposer` the reference's gadgets call.  Columns and the
 * variable table live in HBM with fixed capacities chosen at creation (PG_ERR_CAPACITY when an append does not fit);
 * every call appends on the composer's stream and returns immediately -- Variables are numbered on the host, their
 * values exist only on the device.  With the single-gadget entry points below (the reference's exact signatures,
 * argument order and error behaviour) host code written against the reference reads the same:
 *
 *     reference (tests/range_gadgets_tests.rs:36-43)                 this library
 *     let w = AllocatedScalar::allocate(composer, witness);          pg_allocated_scalar_allocate(c, &witness, &w);
 *     let r = range_check(composer, min, max, w);                    pg_range_check(c, &min, &max, &w, &r);
 *     composer.constrain_to_constant(r, outcome, None);              pg_composer_constrain_to_constant(c, r, &outcome, NULL);
 */
typedef struct pg_composer pg_composer;

/* StandardComposer::new() [with_dummy = 1]: zero_var = Variable(0) + its constant row, then the two dummy
 * constraints (variables 1..4 = 6, 1, 7, -20): 3 rows, 5 variables.  with_dummy = 0 stops after zero_var. */
pg_status pg_composer_create(pg_engine *e, uint64_t gate_capacity, uint64_t var_capacity, int with_dummy, void *stream,
                             pg_composer **out);
void pg_composer_destroy(pg_composer *c);
uint64_t pg_composer_circuit_size(const pg_composer *c);  /* StandardComposer::circuit_size(): rows so far */
uint64_t pg_composer_num_variables(const pg_composer *c);
pg_variable pg_composer_zero_var(const pg_composer *c);
/* device pointers to the live columns (row 0 = gate 0, var_values[0] = Variable(0)); they move when the composer grows */
pg_status pg_composer_columns(const pg_composer *c, pg_columns *out);
/* Capacity.  The reference's composer is a set of Vecs that grow on their own; here growth is an explicit decision
 * because it re-allocates HBM and copies the live part (device to device, on the composer's stream):
 *   pg_composer_reserve    make room for at least that many rows / Variables in total (never shrinks);
 *   pg_composer_auto_grow  on = an append that does not fit doubles the capacity it ran out of (at least to what the
 *                          append needs) instead of returning PG_ERR_CAPACITY; off (the default) = fixed capacity. */
pg_status pg_composer_reserve(pg_composer *c, uint64_t gate_capacity, uint64_t var_capacity);
pg_status pg_composer_auto_grow(pg_composer *c, int on);
/* Where the composer's columns lie (see pg_columns_slab_layout below): stride_bytes > 0 moves the nine arrays into ONE
 * allocation with the selector columns that far apart (live rows and variables are copied; later growth keeps the layout and
 * needs room for the old block and the new one at once); 0 goes back to nine separate allocations.  For a composer of a few
 * GB on a card with room to spare: 24 GiB.  The columns move: pointers from pg_composer_columns are stale afterwards. */
pg_status pg_composer_spread_columns(pg_composer *c, uint64_t stride_bytes);
uint64_t pg_composer_gate_capacity(const pg_composer *c);
uint64_t pg_composer_var_capacity(const pg_composer *c);
pg_status pg_composer_sync(pg_composer *c);
/* The command queue.  The reference's usage is ONE composer call at a time (tests/range_gadgets_tests.rs:29-44:
 * allocate, range_check, constrain_to_constant, ...); a launch per call would make that loop launch-bound.  Variables
 * are numbered on the host and their assignments live on the device, so the single calls -- the gate calls below,
 * pg_allocated_scalar_allocate, pg_range_check, pg_max_bound, and the four scalar gadgets (which are sequences of those
 * gate calls, src/scalar.rs, and are composed of them here) -- are RECORDED (arguments checked, numbering advanced,
 * results returned at once) and reach the composer's stream, in order, when anything needs them: pg_composer_sync /
 * _columns / _copy_out / _read_value / _check / _dense_pi / _materialize / _permutation, any batched append,
 * pg_scalar_decomposition_gadget, a capacity change, a full queue (8192 entries), or pg_composer_flush.  A run of gate calls is ONE launch
 * (the outputs of add / mul computed in command order, level by level); a run of `allocate + range_check` (or
 * max_bound) pairs with the same public bounds -- the reference's loop -- is ONE batched emit launch, and leaves the
 * batched append's footprint for pg_composer_materialize / _permutation (runs of >= 4096 rows).  The loop as the
 * reference's tests write it -- the pair, then up to eight gate calls that create no Variable (constrain_to_constant on
 * the result, assert_equal, ...), per witness -- is ONE emit launch too (its items that many rows apart) plus one run of
 * gates: a row's place is assigned when its call is recorded, and a rows-only gate reads no assignment.  Where those
 * gates constrain the gadget's own result (its Variable on all three wires) the loop leaves a footprint as well.
 *   pg_composer_queue(c, 0) flushes and turns recording off (one launch per call, as before); on by default. */
pg_status pg_composer_queue(pg_composer *c, int on);
pg_status pg_composer_flush(pg_composer *c);
/* entries waiting, flushes so far, launches those flushes took (any pointer may be NULL) */
pg_status pg_composer_queue_stats(const pg_composer *c, uint64_t *pending, uint64_t *flushes, uint64_t *launches);

/* Witness refresh.  The reference's tests prove twice with one preprocessed circuit: prover.clear_witness(), the same
 * gadget calls on other witnesses, prove again (tests/scalar_gadgets_tests.rs:108-119, 168-177, 226-235; dusk-plonk's
 * Prover::clear_witness() = a fresh StandardComposer).  pg_composer_clear_witness does the same -- rows and Variables
 * count from StandardComposer::new()'s state again, public inputs are forgotten -- but the device columns stay where
 * they are: an append that repeats, at the same place in the call sequence, what the build before it did there (the same
 * call with the same public parameters, landing on the same first row, first Variable and zero_var) finds its rows in the
 * columns and writes only its assignments (32 B per variable instead of 184 B per row on top; pg_*_values_batch).  That
 * holds for every append: single composer gate calls and single gadget calls, the batched appends on witness scalars, and the
 * batched appends whose rows depend on device arrays (Variables, per-item bounds), which are signed by a 128-bit digest of
 * those arrays taken in the pass that validates them.  The two gadgets whose SHAPE depends on witness values --
 * is_non_zero and the fused mix, which stop at a zero (src/scalar.rs:79) -- are found in place only when no item fails, in
 * either build.  The first append that differs from the previous build ends the refresh: from there on everything is emitted
 * in full, so another circuit is built correctly too -- only slower. */
pg_status pg_composer_clear_witness(pg_composer *c);
/* since the last pg_composer_clear_witness: rows found in place, rows written again, still matching (any pointer may be NULL) */
pg_status pg_composer_refresh_stats(const pg_composer *c, uint64_t *rows_in_place, uint64_t *rows_rewritten, int *refreshing);

/* composer calls used by the gadgets (same argument order as dusk-plonk 0.8; `pi` may be NULL = None) */
pg_status pg_composer_add_input(pg_composer *c, const pg_scalar *s, pg_variable *out);
pg_status pg_composer_add_witness_to_circuit_description(pg_composer *c, const pg_scalar *value, pg_variable *out);
pg_status pg_composer_constrain_to_constant(pg_composer *c, pg_variable a, const pg_scalar *constant, const pg_scalar *pi);
pg_status pg_composer_assert_equal(pg_composer *c, pg_variable a, pg_variable b);
pg_status pg_composer_poly_gate(pg_composer *c, pg_variable a, pg_variable b, pg_variable o, const pg_scalar *q_m,
                                const pg_scalar *q_l, const pg_scalar *q_r, const pg_scalar *q_o, const pg_scalar *q_c,
                                const pg_scalar *pi);
pg_status pg_composer_add(pg_composer *c, const pg_scalar *q_l, pg_variable a, const pg_scalar *q_r, pg_variable b,
                          const pg_scalar *q_c, const pg_scalar *pi, pg_variable *out);
pg_status pg_composer_mul(pg_composer *c, const pg_scalar *q_m, pg_variable a, pg_variable b, const pg_scalar *q_c,
                          const pg_scalar *pi, pg_variable *out);
pg_status pg_composer_mul_gate(pg_composer *c, pg_variable a, pg_variable b, pg_variable o, const pg_scalar *q_m,
                               const pg_scalar *q_o, const pg_scalar *q_c, const pg_scalar *pi);
pg_status pg_composer_boolean_gate(pg_composer *c, pg_variable a);

/* the reference's public API, one call = one gadget (src/lib.rs:42-45) */
pg_status pg_allocated_scalar_allocate(pg_composer *c, const pg_scalar *scalar, pg_allocated_scalar *out); /* src/allocated_scalar.rs:27 */
pg_status pg_range_check(pg_composer *c, const pg_scalar *min_range, const pg_scalar *max_range,
                         const pg_allocated_scalar *witness, pg_variable *out);                           /* src/range.rs:27-43 */
pg_status pg_max_bound(pg_composer *c, const pg_scalar *max_range, const pg_allocated_scalar *witness, pg_variable *out,
                       uint64_t *num_bits);                                                               /* src/range.rs:82-113 */
/* src/range.rs:119-123; d_bits_out (host, may be NULL) receives the first num_bits bit Variables */
pg_status pg_scalar_decomposition_gadget(pg_composer *c, uint64_t num_bits, const pg_allocated_scalar *witness,
                                         pg_variable *is_equal, pg_variable *bits_out);
pg_status pg_conditionally_select_zero(pg_composer *c, pg_variable x, pg_variable select, pg_variable *out); /* src/scalar.rs:21-27 */
pg_status pg_conditionally_select_one(pg_composer *c, pg_variable y, pg_variable selector, pg_variable *out); /* src/scalar.rs:36-59 */
/* PG_ERR_NON_EXISTING_INVERSE after the first variable + row were appended, like the reference (src/scalar.rs:69-79) */
pg_status pg_is_non_zero(pg_composer *c, pg_variable var, const pg_scalar *value_assigned);              /* src/scalar.rs:63-97 */
pg_status pg_maybe_equal(pg_composer *c, const pg_allocated_scalar *a, const pg_allocated_scalar *b, pg_variable *out); /* src/scalar.rs:105-140 */

/* batched appends: the loop  for w in d_witness { allocate; range_check }  emitted at the composer's end */
pg_status pg_composer_range_check_batch(pg_composer *c, const pg_scalar *min_range, const pg_scalar *max_range,
                                        const pg_scalar *d_witness, uint64_t batch, pg_variable *d_result_vars);

/* AllocatedScalar::allocate of a batch (src/allocated_scalar.rs:27): the loop  for s in d_scalars { add_input(s) };
 * the Variables are *first_var, *first_var + 1, ...  (d_scalars: device, reduced) */
pg_status pg_composer_add_input_batch(pg_composer *c, const pg_scalar *d_scalars, uint64_t batch, pg_variable *first_var);
/* the loop  for i { range_check(composer, min, max, AllocatedScalar { var: d_witness_var[i], scalar: d_witness[i] }) }
 * on witnesses allocated before (device arrays; every d_witness_var[i] must be a Variable of this composer -- the
 * reference panics on an unknown one; here EVERY batched append that takes Variables checks its arrays on the device
 * first (one small reduction and one host synchronisation) and returns PG_ERR_INVALID_ARGUMENT with nothing appended) */
pg_status pg_composer_range_check_allocated_batch(pg_composer *c, const pg_scalar *min_range, const pg_scalar *max_range,
                                                  const pg_variable *d_witness_var, const pg_scalar *d_witness,
                                                  uint64_t batch, pg_variable *d_result_vars);

/* the other uniform gadgets as batched appends, each the loop over i of the reference call named:
 *   max_bound:            allocate(d_witness[i]); max_bound(composer, max_range, w)        src/range.rs:82-113
 *   max_bound_allocated:  max_bound(composer, max_range, AllocatedScalar { d_witness_var[i], d_witness[i] })
 *   scalar_decomposition: scalar_decomposition_gadget(composer, num_bits, AllocatedScalar { .. })   src/range.rs:119-158
 *                         (bit Variables of item i: first new Variable + i * (num_bits + 260) + [0, num_bits))
 *   conditionally_select_zero / _one, maybe_equal: on EXISTING Variables (device index arrays); their assignments
 *                         are read from the composer's own variable table, as the reference reads composer.variables
 * *num_bits (may be NULL) is the u64 max_bound returns. */
pg_status pg_composer_max_bound_batch(pg_composer *c, const pg_scalar *max_range, const pg_scalar *d_witness, uint64_t batch,
                                      pg_variable *d_result_vars, uint64_t *num_bits);
pg_status pg_composer_max_bound_allocated_batch(pg_composer *c, const pg_scalar *max_range, const pg_variable *d_witness_var,
                                                const pg_scalar *d_witness, uint64_t batch, pg_variable *d_result_vars,
                                                uint64_t *num_bits);
pg_status pg_composer_scalar_decomposition_batch(pg_composer *c, uint64_t num_bits, const pg_variable *d_witness_var,
                                                 const pg_scalar *d_witness, uint64_t batch, pg_variable *d_result_vars);
pg_status pg_composer_conditionally_select_zero_batch(pg_composer *c, const pg_variable *d_x_var,
                                                      const pg_variable *d_select_var, uint64_t batch,
                                                      pg_variable *d_result_vars);
pg_status pg_composer_conditionally_select_one_batch(pg_composer *c, const pg_variable *d_y_var,
                                                     const pg_variable *d_selector_var, uint64_t batch,
                                                     pg_variable *d_result_vars);
pg_status pg_composer_maybe_equal_batch(pg_composer *c, const pg_variable *d_a_var, const pg_variable *d_b_var, uint64_t batch,
                                        pg_variable *d_result_vars);

/* the composer's gate calls over device arrays of Variables, one set of selectors for the whole batch (no public
 * inputs): the loops
 *   poly_gate:             for i { composer.poly_gate(a[i], b[i], c[i], q_m, q_l, q_r, q_o, q_c, None) }
 *   add / mul:             for i { out[i] = composer.add((q_l, a[i]), (q_r, b[i]), q_c, None) }  /  mul(q_m, a[i], b[i], q_c, None)
 *   constrain_to_constant: for i { composer.constrain_to_constant(a[i], constant, None) }
 *   boolean_gate:          for i { composer.boolean_gate(a[i]) }
 * every index must be a Variable of this composer (the reference panics on an unknown one; here the arrays are checked on
 * the device first: PG_ERR_INVALID_ARGUMENT, nothing appended). */
pg_status pg_composer_poly_gate_batch(pg_composer *c, const pg_variable *d_a, const pg_variable *d_b, const pg_variable *d_c,
                                      const pg_scalar *q_m, const pg_scalar *q_l, const pg_scalar *q_r, const pg_scalar *q_o,
                                      const pg_scalar *q_c, uint64_t batch);
pg_status pg_composer_add_batch(pg_composer *c, const pg_scalar *q_l, const pg_variable *d_a, const pg_scalar *q_r,
                                const pg_variable *d_b, const pg_scalar *q_c, uint64_t batch, pg_variable *d_out_vars);
pg_status pg_composer_mul_batch(pg_composer *c, const pg_scalar *q_m, const pg_variable *d_a, const pg_variable *d_b,
                                const pg_scalar *q_c, uint64_t batch, pg_variable *d_out_vars);
pg_status pg_composer_constrain_to_constant_batch(pg_composer *c, const pg_variable *d_a, const pg_scalar *constant, uint64_t batch);
pg_status pg_composer_boolean_gate_batch(pg_composer *c, const pg_variable *d_a, uint64_t batch);

/* ragged batched appends: the composer plans the call itself (one host synchronisation for the totals), keeps the
 * per-item offsets for the permutation, and emits.
 *   max_bound_ragged: for i { allocate(d_witness[i]); max_bound(composer, d_max_range[i], w) } -- one public bound per
 *                     item; d_num_bits_out (device u32[batch], may be NULL) receives the u64 each call returns
 *   is_non_zero:      for i { is_non_zero(composer, d_var[i], value of d_var[i]) }; an item whose value is 0 stops after
 *                     its first variable + row (src/scalar.rs:69-79); returns PG_ERR_NON_EXISTING_INVERSE when any did
 *                     (everything is appended all the same, like a loop that records the error and carries on);
 *                     d_err_mask (device u8[batch]) and err_count may be NULL
 *   scalar_mix:       the fused item of pg_scalar_mix_batch (BASELINE config 3) */
pg_status pg_composer_max_bound_ragged_batch(pg_composer *c, const pg_scalar *d_max_range, const pg_scalar *d_witness,
                                             uint64_t batch, pg_variable *d_result_vars, uint32_t *d_num_bits_out);
pg_status pg_composer_is_non_zero_batch(pg_composer *c, const pg_variable *d_var, uint64_t batch, uint8_t *d_err_mask,
                                        uint64_t *err_count);
pg_status pg_composer_scalar_mix_batch(pg_composer *c, const pg_scalar *d_v, const pg_scalar *d_y, const pg_scalar *d_s,
                                       const pg_scalar *d_a, const pg_scalar *d_b, uint64_t batch,
                                       pg_variable *d_result_vars, uint8_t *d_err_mask, uint64_t *err_count);

/* copy rows [gate_first, gate_first + n_gates) of the live columns and variables [var_first, var_first + n_vars)
 * into caller-owned device buffers (any member of dst may be NULL); enqueued on the composer's stream */
pg_status pg_composer_copy_out(pg_composer *c, uint64_t gate_first, uint64_t n_gates, uint64_t var_first, uint64_t n_vars,
                               const pg_columns *dst);

/* read-back and checks */
pg_status pg_composer_read_value(pg_composer *c, pg_variable v, pg_scalar *out); /* synchronises */
/* every row satisfies q_m a b + q_l a + q_r b + q_o c + q_4 d + q_c + PI = 0?  *first_bad = -1 or the first failing row */
pg_status pg_composer_check(pg_composer *c, int64_t *first_bad);
/* construct_dense_pi_vec (tests/scalar_gadgets_tests.rs:151): d_out[circuit_size] */
pg_status pg_composer_dense_pi(pg_composer *c, pg_scalar *d_out);

/* everything else a prover-ready row holds.  Constant selector columns (q_4, q_arith = 1,
 * q_range = q_logic = q_fixed_group_add = q_variable_group_add = 0), w_4 (zero_var except on the dummy rows) and
 * the wire VALUE columns (variables[w_x[i]]).  Any pointer may be NULL (skipped); all are device arrays of
 * circuit_size entries. */
typedef struct pg_full_columns {
    pg_scalar *q_4, *q_arith, *q_range, *q_logic, *q_fixed_group_add, *q_variable_group_add;
    uint64_t *w_4;
    pg_scalar *w_l_value, *w_r_value, *w_o_value, *w_4_value;
} pg_full_columns;
pg_status pg_composer_materialize(pg_composer *c, const pg_full_columns *out);

/* the copy permutation the composer's bookkeeping implies (dusk-plonk's
 * Permutation::compute_sigma_permutations).  d_sigma[4 * padded_n] (device): sigma of position (wire, gate) at
 * index wire * padded_n + gate, encoded the same way; wire 0..3 = left, right, output, fourth; rows >= circuit_size
 * map to themselves.  padded_n >= circuit_size (the prover pads to a power of two).  Enqueued on the composer's
 * stream (one host synchronisation inside, to size the sorted list); scratch is kept by the composer and only grows. */
pg_status pg_composer_permutation(pg_composer *c, uint64_t padded_n, uint64_t *d_sigma);
/* Size of the sorted list of wire positions that refer to Variables created outside their own item (plus every position
 * of rows appended by single calls) for the NEXT pg_composer_permutation call; 0 = the composer's own estimate.  A
 * figure that is too small costs a second pass (the first reports the size needed), never a wrong result. */
pg_status pg_composer_permutation_reserve(pg_composer *c, uint64_t sparse_positions);

/* Satisfiability of the rows of ONE batch call whose wires all point into its own variables (the allocate-style
 * batches: range_check, max_bound, scalar_mix): q_m a b + q_l a + q_r b + q_o c + q_c = 0 on every row, with
 * a/b/c = var_values[wire - var_base] (a wire equal to zero_var reads as 0: is_non_zero's assert_equal row).
 * *first_bad = -1, or the first row that fails or references any other Variable outside
 * [var_base, var_base + n_vars).  Synchronises `stream`.  A size-independent property check: it scales to
 * the full 2^20-item batches where a limb-for-limb comparison with the CPU oracle does not. */
pg_status pg_check_rows(pg_engine *e, const pg_columns *cols, uint64_t n_gates, uint64_t var_base, uint64_t n_vars,
                        pg_variable zero_var, int64_t *first_bad, void *stream);

/* ---- multi-GPU: shards, packed chunks, the all-gather (BASELINE.json config 5) -------------
 * The reference has no counterpart: it is single-threaded (`&mut StandardComposer`, src/range.rs:27-32).  What is
 * sharded is the loop  for w in witnesses { allocate; range_check }  of tests/range_gadgets_tests.rs:29-44: items are
 * independent and item i owns rows [gate_base + i*G, +G) and variables [var_base + i*V, +V), so rank r of P emits its
 * contiguous range [lo, hi) straight at its final global numbering -- no exchange is needed to place anything.  One
 * process per GPU; the host (the Rust host of north_star, or Python) calls the same entry points on every rank. */
typedef struct pg_shard {
    uint32_t rank, world;
    uint64_t lo, hi;              /* this rank's items of the whole batch: [lo, hi); the first total % world ranks get one more */
    uint64_t gate_base, var_base; /* global number of the shard's first row / first variable */
    uint64_t n_gates, n_vars;     /* rows / variables the shard emits */
} pg_shard;
pg_status pg_shard_range(uint64_t total, uint32_t rank, uint32_t world, uint64_t *lo, uint64_t *hi);
pg_status pg_range_check_shard_layout(const pg_scalar *min_range, const pg_scalar *max_range, uint64_t total, uint32_t rank,
                                      uint32_t world, uint64_t gate_base, uint64_t var_base, pg_shard *out);
/* pg_range_check_batch on this rank's shard of a `total`-item batch whose first row / variable are (gate_base,
 * var_base): d_witness_local holds the hi - lo witnesses of the shard, `out` buffers sized by the shard layout.
 * No communication.  `shard` (may be NULL) receives the placement. */
pg_status pg_range_check_sharded_batch(pg_engine *e, const pg_scalar *min_range, const pg_scalar *max_range,
                                       const pg_scalar *d_witness_local, uint64_t total, uint32_t rank, uint32_t world,
                                       uint64_t gate_base, uint64_t var_base, const pg_columns *out,
                                       pg_variable *d_result_vars /* may be NULL */, pg_shard *shard, void *stream);

/* A packed chunk: the nine arrays of one call back to back in ONE buffer (offsets in 8-byte words, every section
 * starting on a 128-byte line of a buffer that does), so that a chunk is a single collective.  pg_columns_in_packed gives the pg_columns view to emit into. */
typedef struct pg_packed {
    uint64_t q_words[5], w_words[3], var_words; /* offsets of q_m..q_c, w_l..w_o, var_values */
    uint64_t total_words, n_gates, n_vars;
} pg_packed;
pg_status pg_packed_layout(uint64_t n_gates, uint64_t n_vars, pg_packed *out);
pg_status pg_columns_in_packed(void *d_packed, uint64_t n_gates, uint64_t n_vars, pg_columns *out);

/* The communicator: RCCL over xGMI, bound at run time (dlsym in the process, then dlopen of librccl.so.1; PG_RCCL_LIB
 * overrides) -- the library itself does not link RCCL.  Rank 0 calls pg_comm_unique_id and hands the 128 bytes to the
 * other ranks out of band (the host's own channel: a file, a socket, MPI, torch.distributed); every rank then calls
 * pg_comm_create with the same id (collective: returns when all `world` ranks have joined).  A host that already owns
 * an ncclComm_t passes it to pg_comm_adopt instead (not destroyed by pg_comm_destroy). */
#define PG_COMM_ID_BYTES 128
typedef struct pg_comm pg_comm;
pg_status pg_comm_unique_id(uint8_t id[PG_COMM_ID_BYTES]);
/* path of the RCCL the library bound ("" if none was found): inside a PyTorch process it must be PyTorch's own copy */
const char *pg_comm_library(void);
pg_status pg_comm_create(pg_engine *e, const uint8_t id[PG_COMM_ID_BYTES], uint32_t rank, uint32_t world, pg_comm **out);
pg_status pg_comm_adopt(pg_engine *e, void *nccl_comm, pg_comm **out);
void pg_comm_destroy(pg_comm *c);
uint32_t pg_comm_rank(const pg_comm *c);
uint32_t pg_comm_world(const pg_comm *c);
/* "a single RCCL all-gather of the emitted gate columns" (north_star), enqueued on `stream` after the emission:
 *   pg_allgather_bytes    d_recv[r * bytes_per_rank ..] = rank r's d_send (one ncclAllGather; a packed chunk, a variable
 *                         table, or the 16 bytes of (rows, variables) totals a ragged sharded batch exchanges);
 *   pg_allgather_columns  the nine arrays of equal-sized shards, one grouped launch; rank order == witness order, so
 *                         every gathered column is the whole batch's column (n_gates / n_vars: per rank). */
pg_status pg_allgather_bytes(pg_comm *c, const void *d_send, void *d_recv, uint64_t bytes_per_rank /* multiple of 8 */,
                             void *stream);
pg_status pg_allgather_columns(pg_comm *c, const pg_columns *local, uint64_t n_gates, uint64_t n_vars,
                               const pg_columns *gathered, void *stream);

/* BASELINE config 5 as ONE object: the chunked gather pipeline.  2^23 x 223 KB = 1.87 TB fits no GPU, so a sharded batch is
 * streamed in chunks of `chunk` witnesses per rank: chunk k + 1 is emitted on the caller's stream while chunk k is in
 * ncclAllGather on the pipeline's communication stream (two slots, events between the streams).  Numbering: rank r's
 * item i is item r * total_per_rank + i of the whole batch (contiguous witness shards).
 *   variables_only = 0  every rank's packed chunk travels (all nine arrays back to back, pg_packed_layout): one collective
 *                       of 223 KB per witness;
 *   variables_only = 1  only the variable tables travel (33 KB per witness); the other ranks' selectors and wire indices --
 *                       a function of the public bounds and the numbering -- are regenerated locally
 *                       (pg_range_check_structure_batch).
 * `consume` is called on the host, once per chunk and in order, when the chunk is complete for `stream`: parts[r] are the
 * nine arrays of rank r's chunk (device pointers owned by the pipeline, n_gates rows / n_vars variables each); whatever
 * the callback enqueues on `stream` reads them safely, and they are overwritten two chunks later.  The communicator must
 * outlive every run of the pipeline (not its destruction).  The reference has no
 * counterpart (one composer, one thread: src/range.rs:27-32); the shape is BASELINE.json config 5's. */
typedef struct pg_gather_pipeline pg_gather_pipeline;
typedef void (*pg_chunk_consumer)(void *user, uint64_t chunk_index, uint32_t world, const pg_columns *parts, uint64_t n_gates,
                                  uint64_t n_vars, void *stream);
pg_status pg_range_check_gather_pipeline_create(pg_comm *c, const pg_scalar *min_range, const pg_scalar *max_range,
                                                uint64_t chunk, uint32_t variables_only, pg_gather_pipeline **out);
/* bytes one rank puts on the links per chunk */
uint64_t pg_range_check_gather_pipeline_bytes_per_chunk(const pg_gather_pipeline *p);
pg_status pg_range_check_gather_pipeline_run(pg_gather_pipeline *p, const pg_scalar *d_witness_local,
                                             uint64_t total_per_rank /* a multiple of chunk */, uint64_t gate_base,
                                             uint64_t var_base, pg_chunk_consumer consume /* may be NULL */, void *user,
                                             void *stream);
void pg_range_check_gather_pipeline_destroy(pg_gather_pipeline *p);

/* Ragged sharded batches (one public bound per item: rows per item differ) have ONE real exchange step: every rank plans
 * its contiguous shard, the 16 bytes of (rows, variables) totals are all-gathered, and an exclusive prefix sum over the
 * ranks gives each rank the global numbering of its first row and variable.
 *   pg_max_bound_ragged_sharded_plan   plan + exchange; `shard` receives this rank's placement (gate_base / var_base
 *                                      already global, n_gates / n_vars its totals); gates_per_rank / vars_per_rank
 *                                      ([world] each, may be NULL) every rank's totals.  Synchronises `stream`.
 *   pg_max_bound_ragged_sharded_batch  the same followed by the emission (pg_max_bound_ragged_batch) at that numbering, for
 *                                      callers whose `out` holds the worst case (515 rows, 517 variables per item). */
pg_status pg_max_bound_ragged_sharded_plan(pg_comm *c, const pg_scalar *d_max_range_local, uint64_t batch_local,
                                           uint32_t *d_num_bits, uint64_t *d_row_off, uint64_t *d_var_off, uint64_t gate_base,
                                           uint64_t var_base, pg_shard *shard, uint64_t *gates_per_rank, uint64_t *vars_per_rank,
                                           void *stream);
pg_status pg_max_bound_ragged_sharded_batch(pg_comm *c, const pg_scalar *d_max_range_local, const pg_scalar *d_witness_local,
                                            uint64_t batch_local, uint32_t *d_num_bits, uint64_t *d_row_off, uint64_t *d_var_off,
                                            uint64_t gate_base, uint64_t var_base, const pg_columns *out,
                                            pg_variable *d_result_vars /* may be NULL */, pg_shard *shard, void *stream);

/* ---- where to put the columns ---------------------------------------------
 * pg_columns is nine device pointers and the library writes where it is told; on MI355X it matters where.  The emitters write
 * the same row of the five selector columns at once, and five such streams inside a few GiB of physical memory run 10-18 %
 * slower than five streams tens of GiB apart.  Columns of tens of GB each lie that far apart by
 * themselves; for a smaller circuit, allocate ONE block of *total_bytes and put the arrays at the offsets this call returns:
 * q_m, q_l, q_r, q_o, q_c `stride_bytes` apart (rounded up to 2 MiB; at least a column's own size), then w_l, w_r, w_o and the
 * variable table back to back behind q_c -- of the layouts measured the best (profiles/NOTES_r04.md section 7).  offsets[] in
 * the order of pg_columns' fields, every one a multiple of 2 MiB; the block itself should start on a 2-MiB boundary too (any
 * large hipMalloc does).  24 GiB is what bench.py's fused-mix configuration uses.  Host arithmetic only. */
pg_status pg_columns_slab_layout(uint64_t n_gates, uint64_t n_vars, uint64_t stride_bytes, uint64_t offsets[9],
                                 uint64_t *total_bytes);

/* ---- diagnostics ----------------------------------------------------------
 * Bare store streams, the comparison points bench.py measures in the same process as every workload (SURVEY.md section
 * 8d; `roofline.bare_fill` of its line).
 * pg_fill_columns: the emitters' store stream with nothing behind it -- a workgroup owns a tile of `rows_per_tile`
 * consecutive rows (0: 32768; a multiple of 8) and sweeps the five selector columns in lock step, then the three wire columns,
 * then its share of the variable table, 16 B per lane, one constant.  On the arrays a workload writes it is that workload's
 * store ceiling on those arrays (where lock-step streams lie decides 10-18 % on MI355X).  All nine
 * pointers 16-byte aligned.
 * pg_fill_bytes: one buffer.  streams = 1..16: written as that many equal parts advanced together by long-lived
 * workgroups; streams = 0: one short-lived workgroup per 8 KiB, two resident per CU (one moving window of a few MiB:
 * the shape that reaches 7.1-7.2 TB/s wherever the buffer lies). */
pg_status pg_fill_columns(pg_engine *e, const pg_columns *out, uint64_t n_gates, uint64_t n_vars, uint64_t rows_per_tile,
                          uint64_t pattern, void *stream);
pg_status pg_fill_bytes(pg_engine *e, void *d_dst /* 16-byte aligned */, uint64_t bytes /* multiple of 16 */,
                        uint32_t streams, uint64_t pattern, void *stream);

#ifdef __cplusplus
}
#endif
#endif
