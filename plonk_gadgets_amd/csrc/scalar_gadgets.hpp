// scalar_gadgets.hpp -- the scalar gadgets of /root/reference/src/scalar.rs as
// policies of the streaming writer (emit.hpp), plus the fused mix of BASELINE
// config C3 (five add_input, then is_non_zero + conditionally_select_one +
// maybe_equal per item, one launch).
//
// Rows (w_l, w_r, w_o ; q_m, q_l, q_r, q_o, q_c) and new variables per gadget:
//   conditionally_select_zero(x, sel)            scalar.rs:21-27     1 row, 1 var
//       (x, sel, out ; 1, 0, 0, -1, 0)                               out = x*sel
//   conditionally_select_one(y, sel)             scalar.rs:36-59     4 rows, 4 vars
//       (one,one,one ; 0, 1, 0, 0, -1)                               one = 1          :41
//       (y, sel, sy  ; 1, 0, 0, -1, 0)                               sy = y*sel       :43
//       (one, sel, oms ; 0, 1, -1, -1, 0)                            oms = 1 - sel    :45-50
//       (sy, oms, out ; 0, 1, 1, -1, 0)                              out = sy + oms   :53-58
//   is_non_zero(var, value)                      scalar.rs:63-97     3 rows, 3 vars (Err: 1 row, 1 var)
//       (var, va, zero_var ; 0, 1, -1, 0, 0)                         va = value       :69-71
//       -- value == 0: Err(NonExistingInverse), nothing more is emitted             :73-80
//       (one,one,one ; 0, 1, 0, 0, -1)                               inv = value^-1 :77, one = 1 :83
//       (var, inv, one ; 1, 0, 0, -1, 0)                                              :84-94
//   maybe_equal(a, b)                            scalar.rs:105-140   3 rows, 3 vars
//       (a, b, u ; 0, 1, -1, -1, 0)                                  u = a - b        :111-117
//       (z, u, y ; -1, 0, 0, -1, 1)                                  z = u^-1 or 0 :121-123, y = 1 - u z :126
//       (y, u, u ; 1, 0, 0, 0, 0)                                                     :129-138
#pragma once

#include "emit.hpp"
#include "invert.hpp"

namespace pg {

struct RowOut {
    uint32_t id[5];
    uint64_t w[3];
};

__device__ __forceinline__ void set_row(RowOut &r, uint32_t qm, uint32_t ql, uint32_t qr, uint32_t qo, uint32_t qc,
                                        uint64_t a, uint64_t b, uint64_t c) {
    r.id[0] = qm; r.id[1] = ql; r.id[2] = qr; r.id[3] = qo; r.id[4] = qc;
    r.w[0] = a; r.w[1] = b; r.w[2] = c;
}

// constrain_to_constant(one, 1) has q_c = -1: table slot T_NEG1
__device__ __forceinline__ void select_zero_row(uint64_t x, uint64_t sel, uint64_t v0, RowOut &r) {
    set_row(r, T_ONE, T_ZERO, T_ZERO, T_NEG1, T_ZERO, x, sel, v0);
}

__device__ __forceinline__ void select_one_row(uint32_t j, uint64_t y, uint64_t sel, uint64_t v0, RowOut &r) {
    const uint64_t one = v0, sy = v0 + 1, oms = v0 + 2, out = v0 + 3;
    if (j == 0) set_row(r, T_ZERO, T_ONE, T_ZERO, T_ZERO, T_NEG1, one, one, one);
    else if (j == 1) set_row(r, T_ONE, T_ZERO, T_ZERO, T_NEG1, T_ZERO, y, sel, sy);
    else if (j == 2) set_row(r, T_ZERO, T_ONE, T_NEG1, T_NEG1, T_ZERO, one, sel, oms);
    else set_row(r, T_ZERO, T_ONE, T_ONE, T_NEG1, T_ZERO, sy, oms, out);
}

__device__ __forceinline__ void is_non_zero_row(uint32_t j, uint64_t var, uint64_t v0, uint64_t zero_var, RowOut &r) {
    const uint64_t va = v0, inv = v0 + 1, one = v0 + 2;
    if (j == 0) set_row(r, T_ZERO, T_ONE, T_NEG1, T_ZERO, T_ZERO, var, va, zero_var);
    else if (j == 1) set_row(r, T_ZERO, T_ONE, T_ZERO, T_ZERO, T_NEG1, one, one, one);
    else set_row(r, T_ONE, T_ZERO, T_ZERO, T_NEG1, T_ZERO, var, inv, one);
}

__device__ __forceinline__ void maybe_equal_row(uint32_t j, uint64_t a, uint64_t b, uint64_t v0, RowOut &r) {
    const uint64_t u = v0, z = v0 + 1, y = v0 + 2;
    if (j == 0) set_row(r, T_ZERO, T_ONE, T_NEG1, T_NEG1, T_ZERO, a, b, u);
    else if (j == 1) set_row(r, T_NEG1, T_ZERO, T_ZERO, T_NEG1, T_ONE, z, u, y);
    else set_row(r, T_ONE, T_ZERO, T_ZERO, T_ZERO, T_ZERO, y, u, u);
}

__device__ __forceinline__ Fr load_fr(const uint4 *p, uint64_t i) {
    FrVec t;
    t.v[0] = p[2 * i];
    t.v[1] = p[2 * i + 1];
    return t.f;
}

__device__ __forceinline__ void row_values(const RowOut &r, const uint4 *table, uint32_t h, uint4 out[5]) {
#pragma unroll
    for (int c = 0; c < 5; c++) out[c] = table[2 * r.id[c] + h];
}

// Inputs of the stand-alone scalar gadgets: existing Variables (indices) and their assignments.
struct ScalarArgs {
    const uint64_t *a_var, *b_var;  // x / y / var / a      and      select / selector / - / b
    const uint4 *a_val, *b_val;
    uint64_t *result_vars;
    uint8_t *err_mask;  // is_non_zero only
};

// ---- conditionally_select_zero ------------------------------------------------
struct SelectZeroGD {
    using Args = ScalarArgs;
    struct alignas(16) ItemRec { Fr out; };
    static constexpr int W = 256;
    static constexpr int kInv = 0;
    __device__ static bool is_inv_slot(const Args &, const ItemRec &, uint32_t) { return false; }
    static constexpr bool kRagged = false, kRecInRows = false, kUsePow2 = false;
    __device__ static uint32_t rows_per_item(const Args &) { return 1; }
    __device__ static uint32_t vars_per_item(const Args &) { return 1; }
    __device__ static void fill_table(const Args &, uint4 *, uint32_t) {}
    __device__ static void item(const Args &A, const EmitOut &O, uint64_t item, const uint4 *, ItemRec &R) {
        R.out = fr_mul(load_fr(A.a_val, item), load_fr(A.b_val, item));
        if (A.result_vars) A.result_vars[item] = O.var_base + item;
    }
    __device__ static void selectors(const Args &, const ItemRec &, uint32_t, const uint4 *table, uint32_t h, uint4 out[5]) {
        RowOut r;
        select_zero_row(0, 0, 0, r);
        row_values(r, table, h, out);
    }
    __device__ static void wires(const Args &A, const EmitOut &, const ItemRec &, uint64_t item, uint64_t vbase, uint32_t,
                                 uint64_t out[3]) {
        RowOut r;
        select_zero_row(A.a_var[item], A.b_var[item], vbase, r);
        out[0] = r.w[0]; out[1] = r.w[1]; out[2] = r.w[2];
    }
    __device__ static Fr var_value(const Args &, const ItemRec &R, const uint4 *, uint32_t) { return R.out; }
};

// ---- conditionally_select_one -------------------------------------------------
struct SelectOneGD {
    using Args = ScalarArgs;
    struct alignas(16) ItemRec { Fr sy, oms; };
    static constexpr int W = 256;
    static constexpr int kInv = 0;
    __device__ static bool is_inv_slot(const Args &, const ItemRec &, uint32_t) { return false; }
    static constexpr bool kRagged = false, kRecInRows = false, kUsePow2 = false;
    __device__ static uint32_t rows_per_item(const Args &) { return 4; }
    __device__ static uint32_t vars_per_item(const Args &) { return 4; }
    __device__ static void fill_table(const Args &, uint4 *, uint32_t) {}
    __device__ static void item(const Args &A, const EmitOut &O, uint64_t item, const uint4 *, ItemRec &R) {
        const Fr sel = load_fr(A.b_val, item);
        R.sy = fr_mul(load_fr(A.a_val, item), sel);
        R.oms = fr_sub(fr_one(), sel);
        if (A.result_vars) A.result_vars[item] = O.var_base + item * 4 + 3;
    }
    __device__ static void selectors(const Args &, const ItemRec &, uint32_t j, const uint4 *table, uint32_t h, uint4 out[5]) {
        RowOut r;
        select_one_row(j, 0, 0, 0, r);
        row_values(r, table, h, out);
    }
    __device__ static void wires(const Args &A, const EmitOut &, const ItemRec &, uint64_t item, uint64_t vbase, uint32_t j,
                                 uint64_t out[3]) {
        RowOut r;
        select_one_row(j, A.a_var[item], A.b_var[item], vbase, r);
        out[0] = r.w[0]; out[1] = r.w[1]; out[2] = r.w[2];
    }
    __device__ static Fr var_value(const Args &, const ItemRec &R, const uint4 *, uint32_t k) {
        if (k == 0) return fr_one();
        if (k == 1) return R.sy;
        if (k == 2) return R.oms;
        return fr_add(R.sy, R.oms);
    }
};

// ---- maybe_equal ----------------------------------------------------------------
struct MaybeEqualGD {
    using Args = ScalarArgs;
    struct alignas(16) ItemRec { Fr u; };
    static constexpr int W = 256;
    static constexpr int kInv = 1;
    __device__ static Fr inv_element(const Args &A, uint64_t item, uint32_t) {
        return fr_sub(load_fr(A.a_val, item), load_fr(A.b_val, item));  // scalar.rs:121
    }
    // variables of an item: u, z, y -- z is the pre-pass's (scalar.rs:122-123)
    __device__ static uint4 *inv_slot(const Args &, const EmitOut &O, uint64_t item, uint32_t) { return O.vars + 2 * (item * 3 + 1); }
    __device__ static bool is_inv_slot(const Args &, const ItemRec &, uint32_t k) { return k == 1; }
    static constexpr bool kRagged = false, kRecInRows = false, kUsePow2 = false;
    __device__ static uint32_t rows_per_item(const Args &) { return 3; }
    __device__ static uint32_t vars_per_item(const Args &) { return 3; }
    __device__ static void fill_table(const Args &, uint4 *, uint32_t) {}
    __device__ static void item(const Args &A, const EmitOut &O, uint64_t item, const uint4 *, ItemRec &R) {
        R.u = fr_sub(load_fr(A.a_val, item), load_fr(A.b_val, item));
        if (A.result_vars) A.result_vars[item] = O.var_base + item * 3 + 2;
    }
    __device__ static void selectors(const Args &, const ItemRec &, uint32_t j, const uint4 *table, uint32_t h, uint4 out[5]) {
        RowOut r;
        maybe_equal_row(j, 0, 0, 0, r);
        row_values(r, table, h, out);
    }
    __device__ static void wires(const Args &A, const EmitOut &, const ItemRec &, uint64_t item, uint64_t vbase, uint32_t j,
                                 uint64_t out[3]) {
        RowOut r;
        maybe_equal_row(j, A.a_var[item], A.b_var[item], vbase, r);
        out[0] = r.w[0]; out[1] = r.w[1]; out[2] = r.w[2];
    }
    __device__ static Fr var_value(const Args &, const ItemRec &R, const uint4 *, uint32_t k) {
        if (k == 0) return R.u;
        return fr_is_zero(R.u) ? fr_one() : fr_zero();  // k == 2 (k == 1 is z: the pre-pass's)
    }
};

// ---- is_non_zero (ragged: an item whose value is 0 stops after 1 row / 1 variable) ----------
struct IsNonZeroGD {
    using Args = ScalarArgs;
    struct alignas(16) ItemRec { Fr value; };
    static constexpr int W = 256;
    static constexpr int kInv = 1;
    __device__ static Fr inv_element(const Args &A, uint64_t item, uint32_t) { return load_fr(A.b_val, item); }  // scalar.rs:73
    // variables of an item: var_assigned, inv, one -- an item whose value is 0 stopped before `inv` existed (scalar.rs:79)
    __device__ static uint4 *inv_slot(const Args &A, const EmitOut &O, uint64_t item, uint32_t) {
        if (fr_is_zero(load_fr(A.b_val, item))) return nullptr;
        return O.vars + 2 * (O.var_off[item] + 1);
    }
    __device__ static bool is_inv_slot(const Args &, const ItemRec &, uint32_t k) { return k == 1; }
    static constexpr bool kRagged = true, kRecInRows = false, kUsePow2 = false;
    __device__ static void fill_table(const Args &, uint4 *, uint32_t) {}
    __device__ static void item(const Args &A, const EmitOut &, uint64_t item, const uint4 *, ItemRec &R) {
        R.value = load_fr(A.b_val, item);
    }
    __device__ static void selectors(const Args &, const ItemRec &, uint32_t j, const uint4 *table, uint32_t h, uint4 out[5]) {
        RowOut r;
        is_non_zero_row(j, 0, 0, 0, r);
        row_values(r, table, h, out);
    }
    __device__ static void wires(const Args &A, const EmitOut &O, const ItemRec &, uint64_t item, uint64_t vbase, uint32_t j,
                                 uint64_t out[3]) {
        RowOut r;
        is_non_zero_row(j, A.a_var[item], vbase, O.zero_var, r);
        out[0] = r.w[0]; out[1] = r.w[1]; out[2] = r.w[2];
    }
    __device__ static Fr var_value(const Args &, const ItemRec &R, const uint4 *, uint32_t k) {
        if (k == 0) return R.value;
        return fr_one();  // k == 2 (k == 1 is inv: the pre-pass's)
    }
};

__global__ __launch_bounds__(kThreads) void is_non_zero_plan_kernel(const uint4 *value, uint64_t batch, uint32_t *rows,
                                                                   uint32_t *vars, uint8_t *err_mask, uint32_t *err_count) {
    const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= batch) return;
    const bool err = fr_is_zero(load_fr(value, i));
    rows[i] = err ? 1 : 3;
    vars[i] = err ? 1 : 3;
    if (err_mask) err_mask[i] = err ? 1 : 0;
    if (err) atomicAdd(err_count, 1u);
}

// ---- the fused mix (BASELINE config C3) ---------------------------------------------
// per item: v, y, s, a, b = 5 x add_input; is_non_zero(var v, v); conditionally_select_one(y, s);
// maybe_equal(a, b).  Variables: [v y s a b | va inv one | one' sy oms out | u z yeq] (15; 13 on Err),
// rows: [3 | 4 | 3] (10; 8 on Err).
struct ScalarMixArgs {
    const uint4 *v, *y, *s, *a, *b;
    uint64_t *result_vars;  // [batch][2]: select_one's output, maybe_equal's output
};

struct ScalarMixGD {
    using Args = ScalarMixArgs;
    // every variable of the item, in emission order, is computed once by the item's lane: the variable sweep is then
    // a plain LDS -> HBM copy (a wave that met one slot needing arithmetic would pay it for all 64 lanes)
    struct alignas(16) ItemRec {
        Fr vals[15];
        uint32_t err, pad[3];
    };
#ifndef PG_MIX_W
#define PG_MIX_W 64
#endif
    static constexpr int W = PG_MIX_W;
    static constexpr int kInv = 2;
    __device__ static Fr inv_element(const Args &A, uint64_t item, uint32_t e) {
        return e == 0 ? load_fr(A.v, item) : fr_sub(load_fr(A.a, item), load_fr(A.b, item));
    }
    // item variables: [v y s a b | va inv one | one' sy oms out | u z yeq]; an item with v = 0 has no inv / one
    __device__ static uint4 *inv_slot(const Args &A, const EmitOut &O, uint64_t item, uint32_t e) {
        const bool err = fr_is_zero(load_fr(A.v, item));
        if (e == 0) return err ? nullptr : O.vars + 2 * (O.var_off[item] + 6);
        return O.vars + 2 * (O.var_off[item] + 5 + (err ? 1 : 3) + 4 + 1);
    }
    __device__ static bool is_inv_slot(const Args &, const ItemRec &R, uint32_t k) {
        const uint32_t nz = R.err ? 1 : 3;
        return (!R.err && k == 6) || k == 5 + nz + 4 + 1;
    }
    static constexpr bool kRagged = true, kRecInRows = true, kUsePow2 = false;
    __device__ static void fill_table(const Args &, uint4 *, uint32_t) {}
    __device__ static void item(const Args &A, const EmitOut &O, uint64_t item, const uint4 *, ItemRec &R) {
        const Fr v = load_fr(A.v, item), y = load_fr(A.y, item), s = load_fr(A.s, item), a = load_fr(A.a, item),
                 b = load_fr(A.b, item);
        const uint32_t err = fr_is_zero(v) ? 1u : 0u;
        R.err = err;
        uint32_t k = 0;
        R.vals[k++] = v; R.vals[k++] = y; R.vals[k++] = s; R.vals[k++] = a; R.vals[k++] = b;  // 5 x add_input
        R.vals[k++] = v;                                                                     // var_assigned, scalar.rs:69
        if (!err) {
            R.vals[k++] = fr_zero();                                                         // inverse: the pre-pass's slot
            R.vals[k++] = fr_one();                                                          // one, scalar.rs:83
        }
        const Fr sy = fr_mul(y, s), oms = fr_sub(fr_one(), s);
        R.vals[k++] = fr_one();                                                              // scalar.rs:41
        R.vals[k++] = sy;                                                                    // scalar.rs:43
        R.vals[k++] = oms;                                                                   // scalar.rs:45-50
        R.vals[k++] = fr_add(sy, oms);                                                       // scalar.rs:53-58
        const Fr u = fr_sub(a, b);
        R.vals[k++] = u;                                                                     // scalar.rs:111-117
        R.vals[k++] = fr_zero();                                                             // z: the pre-pass's slot
        R.vals[k++] = fr_is_zero(u) ? fr_one() : fr_zero();                                  // scalar.rs:126
        if (A.result_vars) {
            const uint64_t vb = O.var_base + O.var_off[item];
            const uint64_t nz = err ? 1 : 3;
            A.result_vars[2 * item] = vb + 5 + nz + 3;
            A.result_vars[2 * item + 1] = vb + 5 + nz + 4 + 2;
        }
    }
    __device__ static void row(const ItemRec &R, uint64_t vbase, uint64_t zero_var, uint32_t j, RowOut &r) {
        const uint32_t nz = R.err ? 1 : 3;
        if (j < nz) is_non_zero_row(j, vbase + 0, vbase + 5, zero_var, r);
        else if (j < nz + 4) select_one_row(j - nz, vbase + 1, vbase + 2, vbase + 5 + nz, r);
        else maybe_equal_row(j - nz - 4, vbase + 3, vbase + 4, vbase + 5 + nz + 4, r);
    }
    __device__ static void selectors(const Args &, const ItemRec &R, uint32_t j, const uint4 *table, uint32_t h, uint4 out[5]) {
        RowOut r;
        row(R, 0, 0, j, r);
        row_values(r, table, h, out);
    }
    __device__ static void wires(const Args &, const EmitOut &O, const ItemRec &R, uint64_t, uint64_t vbase, uint32_t j,
                                 uint64_t out[3]) {
        RowOut r;
        row(R, vbase, O.zero_var, j, r);
        out[0] = r.w[0]; out[1] = r.w[1]; out[2] = r.w[2];
    }
    __device__ static Fr var_value(const Args &, const ItemRec &R, const uint4 *, uint32_t k) { return R.vals[k]; }
};

__global__ __launch_bounds__(kThreads) void scalar_mix_plan_kernel(const uint4 *v, uint64_t batch, uint32_t *rows, uint32_t *vars,
                                                                  uint8_t *err_mask, uint32_t *err_count) {
    const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= batch) return;
    const bool err = fr_is_zero(load_fr(v, i));
    rows[i] = err ? 8 : 10;
    vars[i] = err ? 13 : 15;
    if (err_mask) err_mask[i] = err ? 1 : 0;
    if (err) atomicAdd(err_count, 1u);
}

}  // namespace pg
