// plonk_gadgets.hpp -- C++ host side above the C ABI (include/plonk_gadgets_hip.h): the reference's public
// interface, name for name, over a device-resident composer.
//
// The reference is a Rust library; no Rust toolchain exists in this build environment, so the host layer a Rust
// user would get from the shim in INTEGRATION.md is provided in C++ instead (header only, links only against
// libplonk_gadgets_hip.so).  Mirrors /root/reference/src/lib.rs:37-45:
//     pub use allocated_scalar::AllocatedScalar;        -> plonk_gadgets::AllocatedScalar
//     pub use errors::Error;                            -> plonk_gadgets::Error
//     pub use range as RangeGadgets;                    -> plonk_gadgets::RangeGadgets::{range_check, max_bound}
//     pub use scalar as ScalarGadgets;                  -> plonk_gadgets::ScalarGadgets::{conditionally_select_zero,
//                                                          conditionally_select_one, is_non_zero, maybe_equal}
// plus the slice of dusk_plonk::prelude the gadgets and their tests use (BlsScalar, Variable, StandardComposer).
// Same argument order, same return shapes (Result<(), Error> is `Result`), same error behaviour; conditions on
// which the reference panics throw std::runtime_error.  Every call runs on the GPU; nothing is computed here
// except BlsScalar host arithmetic for forming public arguments.
#pragma once

#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "plonk_gadgets_hip.h"

namespace plonk_gadgets {

// src/errors.rs:13-18
enum class Error { NonExistingInverse };

// Result<(), Error>
struct Result {
    std::optional<Error> err;
    bool is_ok() const { return !err.has_value(); }
    bool is_err() const { return err.has_value(); }
};

inline void pg_throw(pg_status st, const char *where) {
    if (st != PG_OK) throw std::runtime_error(std::string(where) + ": " + pg_status_string(st) + " (" + pg_last_error() + ")");
}

// dusk_plonk::bls12_381::BlsScalar: 4 x u64 Montgomery limbs
class BlsScalar {
  public:
    pg_scalar s{};
    BlsScalar() = default;
    explicit BlsScalar(const pg_scalar &raw) : s(raw) {}
    static BlsScalar from(uint64_t v) { BlsScalar r; pg_scalar_from_u64(v, &r.s); return r; }
    static BlsScalar zero() { return from(0); }
    static BlsScalar one() { return from(1); }
    // canonical little-endian limbs -> scalar
    static BlsScalar from_raw(const uint64_t raw[4]) { BlsScalar r; pg_scalar_from_canonical(raw, &r.s); return r; }
    // BlsScalar::from(2).pow(&[e, 0, 0, 0]) as the reference's tests write it
    static BlsScalar pow_of_2(uint64_t e) {
        BlsScalar r = one(), two = from(2);
        for (uint64_t i = 0; i < e; i++) r = r * two;
        return r;
    }
    BlsScalar operator+(const BlsScalar &o) const { BlsScalar r; pg_scalar_add(&s, &o.s, &r.s); return r; }
    BlsScalar operator-(const BlsScalar &o) const { BlsScalar r; pg_scalar_sub(&s, &o.s, &r.s); return r; }
    BlsScalar operator*(const BlsScalar &o) const { BlsScalar r; pg_scalar_mul(&s, &o.s, &r.s); return r; }
    BlsScalar operator-() const { BlsScalar r; pg_scalar_neg(&s, &r.s); return r; }
    // BlsScalar::invert(): false (and *out = 0) for zero, the reference's CtOption::is_none
    bool invert(BlsScalar *out) const { return pg_scalar_invert(&s, &out->s) == PG_OK; }
    bool operator==(const BlsScalar &o) const {
        return s.l[0] == o.s.l[0] && s.l[1] == o.s.l[1] && s.l[2] == o.s.l[2] && s.l[3] == o.s.l[3];
    }
    bool operator!=(const BlsScalar &o) const { return !(*this == o); }
};

// dusk_plonk::constraint_system::Variable(usize)
struct Variable {
    uint64_t index = 0;
    bool operator==(const Variable &o) const { return index == o.index; }
};

class Engine {
  public:
    pg_engine *h = nullptr;
    explicit Engine(int device = 0) { pg_throw(pg_engine_create(device, &h), "pg_engine_create"); }
    ~Engine() { pg_engine_destroy(h); }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
};

// dusk_plonk::constraint_system::StandardComposer, the slice the gadgets and their tests call
class StandardComposer {
  public:
    pg_composer *h = nullptr;
    // StandardComposer::new()
    explicit StandardComposer(Engine &e, uint64_t gate_capacity = 1 << 16, uint64_t var_capacity = 1 << 16) {
        pg_throw(pg_composer_create(e.h, gate_capacity, var_capacity, 1, nullptr, &h), "pg_composer_create");
    }
    ~StandardComposer() { pg_composer_destroy(h); }
    StandardComposer(const StandardComposer &) = delete;
    StandardComposer &operator=(const StandardComposer &) = delete;

    // capacity is explicit here (the reference's Vecs grow on their own): reserve, or let appends double it
    void reserve(uint64_t gate_capacity, uint64_t var_capacity) { pg_throw(pg_composer_reserve(h, gate_capacity, var_capacity), "reserve"); }
    void auto_grow(bool on = true) { pg_throw(pg_composer_auto_grow(h, on ? 1 : 0), "auto_grow"); }
    // the columns in ONE allocation, the selector columns stride_bytes apart (0: nine allocations): plonk_gadgets_hip.h
    void spread_columns(uint64_t stride_bytes) { pg_throw(pg_composer_spread_columns(h, stride_bytes), "spread_columns"); }

    // the command queue behind the single calls (plonk_gadgets_hip.h): calls are validated and numbered at once and
    // recorded; anything that needs the device state flushes them, in order, as few launches.  queue(false) = one launch
    // per call; flush() sends what is pending; sync() flushes and waits for the composer's stream.
    void queue(bool on = true) { pg_throw(pg_composer_queue(h, on ? 1 : 0), "queue"); }
    void flush() { pg_throw(pg_composer_flush(h), "flush"); }
    void sync() { pg_throw(pg_composer_sync(h), "sync"); }
    struct QueueStats { uint64_t pending, flushes, launches; };
    QueueStats queue_stats() const {
        QueueStats q{};
        pg_throw(pg_composer_queue_stats(h, &q.pending, &q.flushes, &q.launches), "queue_stats");
        return q;
    }

    // Prover::clear_witness() of the reference's tests (tests/scalar_gadgets_tests.rs:110,170,228): a fresh composer again; the
    // same calls on other witnesses find their rows in place and write only the assignments
    void clear_witness() { pg_throw(pg_composer_clear_witness(h), "clear_witness"); }
    struct RefreshStats { uint64_t rows_in_place, rows_rewritten; bool refreshing; };
    RefreshStats refresh_stats() const {
        RefreshStats r{};
        int on = 0;
        pg_throw(pg_composer_refresh_stats(h, &r.rows_in_place, &r.rows_rewritten, &on), "refresh_stats");
        r.refreshing = on != 0;
        return r;
    }

    uint64_t circuit_size() const { return pg_composer_circuit_size(h); }
    uint64_t num_variables() const { return pg_composer_num_variables(h); }
    Variable zero_var() const { return Variable{pg_composer_zero_var(h)}; }

    Variable add_input(const BlsScalar &s) {
        Variable v;
        pg_throw(pg_composer_add_input(h, &s.s, &v.index), "add_input");
        return v;
    }
    Variable add_witness_to_circuit_description(const BlsScalar &value) {
        Variable v;
        pg_throw(pg_composer_add_witness_to_circuit_description(h, &value.s, &v.index), "add_witness_to_circuit_description");
        return v;
    }
    void constrain_to_constant(Variable a, const BlsScalar &constant, const std::optional<BlsScalar> &pi) {
        pg_throw(pg_composer_constrain_to_constant(h, a.index, &constant.s, pi ? &pi->s : nullptr), "constrain_to_constant");
    }
    void assert_equal(Variable a, Variable b) { pg_throw(pg_composer_assert_equal(h, a.index, b.index), "assert_equal"); }
    void poly_gate(Variable a, Variable b, Variable c, const BlsScalar &q_m, const BlsScalar &q_l, const BlsScalar &q_r,
                   const BlsScalar &q_o, const BlsScalar &q_c, const std::optional<BlsScalar> &pi) {
        pg_throw(pg_composer_poly_gate(h, a.index, b.index, c.index, &q_m.s, &q_l.s, &q_r.s, &q_o.s, &q_c.s, pi ? &pi->s : nullptr),
                 "poly_gate");
    }
    Variable add(std::pair<BlsScalar, Variable> q_l_a, std::pair<BlsScalar, Variable> q_r_b, const BlsScalar &q_c,
                 const std::optional<BlsScalar> &pi) {
        Variable v;
        pg_throw(pg_composer_add(h, &q_l_a.first.s, q_l_a.second.index, &q_r_b.first.s, q_r_b.second.index, &q_c.s,
                                 pi ? &pi->s : nullptr, &v.index), "add");
        return v;
    }
    Variable mul(const BlsScalar &q_m, Variable a, Variable b, const BlsScalar &q_c, const std::optional<BlsScalar> &pi) {
        Variable v;
        pg_throw(pg_composer_mul(h, &q_m.s, a.index, b.index, &q_c.s, pi ? &pi->s : nullptr, &v.index), "mul");
        return v;
    }
    void mul_gate(Variable a, Variable b, Variable c, const BlsScalar &q_m, const BlsScalar &q_o, const BlsScalar &q_c,
                  const std::optional<BlsScalar> &pi) {
        pg_throw(pg_composer_mul_gate(h, a.index, b.index, c.index, &q_m.s, &q_o.s, &q_c.s, pi ? &pi->s : nullptr), "mul_gate");
    }
    Variable boolean_gate(Variable a) {
        pg_throw(pg_composer_boolean_gate(h, a.index), "boolean_gate");
        return a;
    }

    // what prove + verify establish in the reference's tests, restricted to gate satisfiability:
    // -1 when every row holds, else the first failing row
    int64_t check() {
        int64_t bad = 0;
        pg_throw(pg_composer_check(h, &bad), "check");
        return bad;
    }
    BlsScalar value(Variable v) {
        BlsScalar r;
        pg_throw(pg_composer_read_value(h, v.index, &r.s), "read_value");
        return r;
    }
    // what dusk-plonk's preprocessing consumes next: the remaining columns of a prover-ready row and
    // the copy permutation; device buffers of circuit_size() / 4 * padded_n entries
    void materialize(const pg_full_columns &out) { pg_throw(pg_composer_materialize(h, &out), "materialize"); }
    void permutation(uint64_t padded_n, uint64_t *d_sigma) { pg_throw(pg_composer_permutation(h, padded_n, d_sigma), "permutation"); }
};

// src/allocated_scalar.rs:17-30
struct AllocatedScalar {
    Variable var;
    BlsScalar scalar;
    static AllocatedScalar allocate(StandardComposer &composer, const BlsScalar &scalar) {
        pg_allocated_scalar out;
        pg_throw(pg_allocated_scalar_allocate(composer.h, &scalar.s, &out), "AllocatedScalar::allocate");
        return AllocatedScalar{Variable{out.var}, BlsScalar(out.scalar)};
    }
    pg_allocated_scalar c() const { return pg_allocated_scalar{var.index, scalar.s}; }
};

// The loops a caller of the reference writes around the gadgets -- for w in witnesses { allocate; gadget } -- as single
// appends at the composer's end.  Arrays are DEVICE pointers (scalars: reduced Montgomery limbs; Variables: u64);
// result arrays may be NULL.  Same rows, Variables and numbering as the loop of single calls.
namespace Batched {
// for s in scalars { AllocatedScalar::allocate(composer, s) } -> the first Variable, the others follow
inline Variable allocate(StandardComposer &c, const pg_scalar *d_scalars, uint64_t batch) {
    Variable first;
    pg_throw(pg_composer_add_input_batch(c.h, d_scalars, batch, &first.index), "Batched::allocate");
    return first;
}
inline void range_check(StandardComposer &c, const BlsScalar &min_range, const BlsScalar &max_range, const pg_scalar *d_witness,
                        uint64_t batch, pg_variable *d_result_vars) {
    pg_throw(pg_composer_range_check_batch(c.h, &min_range.s, &max_range.s, d_witness, batch, d_result_vars), "Batched::range_check");
}
inline void range_check_allocated(StandardComposer &c, const BlsScalar &min_range, const BlsScalar &max_range,
                                  const pg_variable *d_witness_var, const pg_scalar *d_witness, uint64_t batch,
                                  pg_variable *d_result_vars) {
    pg_throw(pg_composer_range_check_allocated_batch(c.h, &min_range.s, &max_range.s, d_witness_var, d_witness, batch, d_result_vars),
             "Batched::range_check_allocated");
}
inline uint64_t max_bound(StandardComposer &c, const BlsScalar &max_range, const pg_scalar *d_witness, uint64_t batch,
                          pg_variable *d_result_vars) {
    uint64_t num_bits = 0;
    pg_throw(pg_composer_max_bound_batch(c.h, &max_range.s, d_witness, batch, d_result_vars, &num_bits), "Batched::max_bound");
    return num_bits;
}
inline uint64_t max_bound_allocated(StandardComposer &c, const BlsScalar &max_range, const pg_variable *d_witness_var,
                                    const pg_scalar *d_witness, uint64_t batch, pg_variable *d_result_vars) {
    uint64_t num_bits = 0;
    pg_throw(pg_composer_max_bound_allocated_batch(c.h, &max_range.s, d_witness_var, d_witness, batch, d_result_vars, &num_bits),
             "Batched::max_bound_allocated");
    return num_bits;
}
// one public bound per item
inline void max_bound_ragged(StandardComposer &c, const pg_scalar *d_max_range, const pg_scalar *d_witness, uint64_t batch,
                             pg_variable *d_result_vars, uint32_t *d_num_bits) {
    pg_throw(pg_composer_max_bound_ragged_batch(c.h, d_max_range, d_witness, batch, d_result_vars, d_num_bits), "Batched::max_bound_ragged");
}
inline void scalar_decomposition(StandardComposer &c, uint64_t num_bits, const pg_variable *d_witness_var, const pg_scalar *d_witness,
                                 uint64_t batch, pg_variable *d_result_vars) {
    pg_throw(pg_composer_scalar_decomposition_batch(c.h, num_bits, d_witness_var, d_witness, batch, d_result_vars),
             "Batched::scalar_decomposition");
}
inline void conditionally_select_zero(StandardComposer &c, const pg_variable *d_x, const pg_variable *d_select, uint64_t batch,
                                      pg_variable *d_result_vars) {
    pg_throw(pg_composer_conditionally_select_zero_batch(c.h, d_x, d_select, batch, d_result_vars), "Batched::conditionally_select_zero");
}
inline void conditionally_select_one(StandardComposer &c, const pg_variable *d_y, const pg_variable *d_selector, uint64_t batch,
                                     pg_variable *d_result_vars) {
    pg_throw(pg_composer_conditionally_select_one_batch(c.h, d_y, d_selector, batch, d_result_vars), "Batched::conditionally_select_one");
}
inline void maybe_equal(StandardComposer &c, const pg_variable *d_a, const pg_variable *d_b, uint64_t batch, pg_variable *d_result_vars) {
    pg_throw(pg_composer_maybe_equal_batch(c.h, d_a, d_b, batch, d_result_vars), "Batched::maybe_equal");
}
// Ok, or Error::NonExistingInverse when some item's value was zero (all items are appended either way, the failing
// ones as far as the reference gets before it returns the error); *err_count = how many
inline Result is_non_zero(StandardComposer &c, const pg_variable *d_var, uint64_t batch, uint8_t *d_err_mask, uint64_t *err_count) {
    const pg_status st = pg_composer_is_non_zero_batch(c.h, d_var, batch, d_err_mask, err_count);
    if (st == PG_ERR_NON_EXISTING_INVERSE) return Result{Error::NonExistingInverse};
    pg_throw(st, "Batched::is_non_zero");
    return Result{};
}
// the composer's own gate calls over arrays of Variables, one set of selectors per batch
inline void poly_gate(StandardComposer &c, const pg_variable *d_a, const pg_variable *d_b, const pg_variable *d_c, const BlsScalar &q_m,
                      const BlsScalar &q_l, const BlsScalar &q_r, const BlsScalar &q_o, const BlsScalar &q_c, uint64_t batch) {
    pg_throw(pg_composer_poly_gate_batch(c.h, d_a, d_b, d_c, &q_m.s, &q_l.s, &q_r.s, &q_o.s, &q_c.s, batch), "Batched::poly_gate");
}
inline void add(StandardComposer &c, const BlsScalar &q_l, const pg_variable *d_a, const BlsScalar &q_r, const pg_variable *d_b,
                const BlsScalar &q_c, uint64_t batch, pg_variable *d_out_vars) {
    pg_throw(pg_composer_add_batch(c.h, &q_l.s, d_a, &q_r.s, d_b, &q_c.s, batch, d_out_vars), "Batched::add");
}
inline void mul(StandardComposer &c, const BlsScalar &q_m, const pg_variable *d_a, const pg_variable *d_b, const BlsScalar &q_c,
                uint64_t batch, pg_variable *d_out_vars) {
    pg_throw(pg_composer_mul_batch(c.h, &q_m.s, d_a, d_b, &q_c.s, batch, d_out_vars), "Batched::mul");
}
inline void constrain_to_constant(StandardComposer &c, const pg_variable *d_a, const BlsScalar &constant, uint64_t batch) {
    pg_throw(pg_composer_constrain_to_constant_batch(c.h, d_a, &constant.s, batch), "Batched::constrain_to_constant");
}
inline void boolean_gate(StandardComposer &c, const pg_variable *d_a, uint64_t batch) {
    pg_throw(pg_composer_boolean_gate_batch(c.h, d_a, batch), "Batched::boolean_gate");
}
}  // namespace Batched

namespace RangeGadgets {
// src/range.rs:27-32
inline Variable range_check(StandardComposer &composer, const BlsScalar &min_range, const BlsScalar &max_range,
                            const AllocatedScalar &witness) {
    Variable out;
    const pg_allocated_scalar w = witness.c();
    pg_throw(pg_range_check(composer.h, &min_range.s, &max_range.s, &w, &out.index), "range_check");
    return out;
}
// src/range.rs:82-86
inline std::pair<Variable, uint64_t> max_bound(StandardComposer &composer, const BlsScalar &max_range,
                                               const AllocatedScalar &witness) {
    Variable out;
    uint64_t num_bits = 0;
    const pg_allocated_scalar w = witness.c();
    pg_throw(pg_max_bound(composer.h, &max_range.s, &w, &out.index, &num_bits), "max_bound");
    return {out, num_bits};
}
// src/range.rs:119-123 -- private in the reference (its in-file unit test calls it); exposed for the same purpose
inline std::pair<Variable, std::vector<Variable>> scalar_decomposition_gadget(StandardComposer &composer, size_t num_bits,
                                                                              const AllocatedScalar &witness) {
    Variable is_equal;
    std::vector<uint64_t> bits(num_bits ? num_bits : 1);
    const pg_allocated_scalar w = witness.c();
    pg_throw(pg_scalar_decomposition_gadget(composer.h, num_bits, &w, &is_equal.index, bits.data()), "scalar_decomposition_gadget");
    std::vector<Variable> out;
    for (size_t i = 0; i < num_bits; i++) out.push_back(Variable{bits[i]});
    return {is_equal, out};
}
}  // namespace RangeGadgets

namespace ScalarGadgets {
// src/scalar.rs:21-25
inline Variable conditionally_select_zero(StandardComposer &composer, Variable x, Variable select) {
    Variable out;
    pg_throw(pg_conditionally_select_zero(composer.h, x.index, select.index, &out.index), "conditionally_select_zero");
    return out;
}
// src/scalar.rs:36-40
inline Variable conditionally_select_one(StandardComposer &composer, Variable y, Variable selector) {
    Variable out;
    pg_throw(pg_conditionally_select_one(composer.h, y.index, selector.index, &out.index), "conditionally_select_one");
    return out;
}
// src/scalar.rs:63-67
inline Result is_non_zero(StandardComposer &composer, Variable var, const BlsScalar &value_assigned) {
    const pg_status st = pg_is_non_zero(composer.h, var.index, &value_assigned.s);
    if (st == PG_ERR_NON_EXISTING_INVERSE) return Result{Error::NonExistingInverse};
    pg_throw(st, "is_non_zero");
    return Result{};
}
// src/scalar.rs:105-109
inline Variable maybe_equal(StandardComposer &composer, const AllocatedScalar &a, const AllocatedScalar &b) {
    Variable out;
    const pg_allocated_scalar ca = a.c(), cb = b.c();
    pg_throw(pg_maybe_equal(composer.h, &ca, &cb, &out.index), "maybe_equal");
    return out;
}
}  // namespace ScalarGadgets

}  // namespace plonk_gadgets
