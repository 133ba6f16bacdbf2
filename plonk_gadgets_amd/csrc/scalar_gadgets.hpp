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
    static constexpr int kInvGroup = PG_INV_GRP;
    __device__ static void inv_operands(const Args &A, const EmitOut &, uint64_t item, uint32_t, FrVec &p, FrVec &q, uint32_t &) {
        p.f = load_fr(A.a_val, item);
        q.f = load_fr(A.b_val, item);
    }
    __device__ static Fr inv_combine(const Args &, uint32_t, const Fr &p, const Fr &q, uint32_t) { return fr_sub(p, q); }  // scalar.rs:121
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
    static constexpr int kInvGroup = PG_INV_GRP;
    __device__ static void inv_operands(const Args &A, const EmitOut &, uint64_t item, uint32_t, FrVec &p, FrVec &q, uint32_t &) {
        p.f = load_fr(A.b_val, item);  // scalar.rs:73
        q.f = p.f;
    }
    __device__ static Fr inv_combine(const Args &, uint32_t, const Fr &p, const Fr &, uint32_t) { return p; }
    // variables of an item: var_assigned, inv, one -- an item whose value is 0 stopped before `inv` existed (scalar.rs:79)
    __device__ static uint4 *inv_slot(const Args &A, const EmitOut &O, uint64_t item, uint32_t) {
        if (fr_is_zero(load_fr(A.b_val, item))) return nullptr;
        return O.vars + 2 * (O.var_off[item] + 1);
    }
    __device__ static bool is_inv_slot(const Args &, const ItemRec &, uint32_t k) { return k == 1; }
    static constexpr bool kRagged = true, kRecInRows = false, kUsePow2 = false;
    static constexpr uint32_t kUniformRows = 3, kUniformVars = 3;  // an item whose value is not 0
    __device__ static void fill_table(const Args &, uint4 *, uint32_t) {}
    __device__ static void item_rows(const Args &, const EmitOut &, uint64_t, const uint4 *, ItemRec &) {}
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

// plan kernels: per-item counts + the block sums of the prefix-sum pass in one launch (grid = blocks of kScanBlock items)
__global__ __launch_bounds__(kThreads) void is_non_zero_plan_kernel(const uint4 *value, uint64_t batch, uint32_t *rows,
                                                                   uint32_t *vars, uint8_t *err_mask, uint32_t *err_count,
                                                                   const PlanScan P) {
    uint32_t r[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x * 4 + k;
        if (i < batch) {
            const bool err = fr_is_zero(load_fr(value, i));
            r[k] = v[k] = err ? 1 : 3;
            if (err_mask) err_mask[i] = err ? 1 : 0;
            if (err) atomicAdd(err_count, 1u);
        }
    }
    plan_store(P, r, v, batch, rows, vars);
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
    // what the item's lane computes once: the five inputs and the four derived values that are not constants (the
    // variable sweep is then an LDS -> HBM copy: a wave that met one slot needing arithmetic would pay it for all 64
    // lanes).  Kept to 368 bytes: the records bound how many workgroups share a CU, and the variable-table launch lives
    // on that overlap (6 per CU instead of 4 with one slot per variable).
    struct alignas(16) ItemRec {
        Fr d[11];  // v y s a b | sy oms out | u | v^-1 (a-b)^-1  (the inverses: read back from the pre-pass's compact output)
        uint32_t err, yeq, pad[2];
    };
#ifndef PG_MIX_W
#define PG_MIX_W 64
#endif
    static constexpr int W = PG_MIX_W;
    static constexpr int kInv = 2;
    static constexpr int kInvGroup = PG_INV_GRP;
    // element 0: v (is_non_zero, scalar.rs:73); element 1: a - b (maybe_equal, scalar.rs:121) -- pointers selected, not branches
    __device__ static void inv_operands(const Args &A, const EmitOut &, uint64_t item, uint32_t e, FrVec &p, FrVec &q, uint32_t &) {
        const uint4 *pp = e ? A.a : A.v, *qp = e ? A.b : A.v;
        p.f = load_fr(pp, item);
        q.f = load_fr(qp, item);
    }
    __device__ static Fr inv_combine(const Args &, uint32_t e, const Fr &p, const Fr &q, uint32_t) {
        const Fr d = fr_sub(p, q);
        Fr r;
#pragma unroll
        for (int i = 0; i < 4; i++) r.l[i] = e ? d.l[i] : p.l[i];
        return r;
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
    static constexpr uint32_t kUniformRows = 10, kUniformVars = 15;  // an item whose v is not 0
    // full-shape items: selectors are a function of the row, wires are the item's own variables (+ zero_var)
    static constexpr bool kPeriodic = true;
    // emitted as two launches (emit.hpp, EmitMode): the rows -- which depend on one bit per item -- as a lean store-only
    // launch over tiles of kRowsW items, then the variable table
    static constexpr bool kSplit = true;
    static constexpr int kRowsW = 256;
    struct RowRec {
        uint32_t err;
    };
    __device__ static void fill_table(const Args &, uint4 *, uint32_t) {}
    // the rows of an item depend on one bit: did is_non_zero stop at its error (scalar.rs:79)?
    template <class R>
    __device__ static void item_rows(const Args &A, const EmitOut &, uint64_t item, const uint4 *, R &rec) {
        rec.err = fr_is_zero(load_fr(A.v, item)) ? 1u : 0u;
    }
    // what an item's lane reads from memory, apart from the arithmetic on it
    struct Loads {
        Fr v, y, s, a, b, inv0, inv1;
        uint64_t var_off;
    };
    __device__ static void item_load(const Args &A, const EmitOut &O, uint64_t item, Loads &L) {
        L.v = load_fr(A.v, item);
        L.y = load_fr(A.y, item);
        L.s = load_fr(A.s, item);
        L.a = load_fr(A.a, item);
        L.b = load_fr(A.b, item);
        L.var_off = O.var_off[item];
        // the item's two inverses, if the pre-pass has run (the variables-only launch of the split): element e of item i
        // sits at [e * batch + i] of its compact output
        const uint4 *inv = O.inv ? O.inv : A.v;  // (no pre-pass output: any readable address, the values are not used)
        L.inv0 = load_fr(inv, O.inv ? item : 0);
        L.inv1 = load_fr(inv, O.inv ? O.batch + item : 0);
    }
    __device__ static void item_from(const Args &A, const EmitOut &O, uint64_t item, ItemRec &R, const Loads &L) {
        const uint32_t err = fr_is_zero(L.v) ? 1u : 0u;
        R.err = err;  // what item_rows writes when there are rows to emit; a variables-only launch has no item_rows
        R.d[0] = L.v; R.d[1] = L.y; R.d[2] = L.s; R.d[3] = L.a; R.d[4] = L.b;  // 5 x add_input (and var_assigned = v, scalar.rs:69)
        const Fr sy = fr_mul(L.y, L.s), oms = fr_sub(fr_one(), L.s);
        R.d[5] = sy;                                                  // scalar.rs:43
        R.d[6] = oms;                                                 // scalar.rs:45-50
        R.d[7] = fr_add(sy, oms);                                     // scalar.rs:53-58
        const Fr u = fr_sub(L.a, L.b);
        R.d[8] = u;                                                   // scalar.rs:111-117
        R.d[9] = L.inv0;                                              // scalar.rs:77 (0 for an item that stopped at its error: never stored)
        R.d[10] = L.inv1;                                             // scalar.rs:122-123
        R.yeq = fr_is_zero(u) ? 1u : 0u;                              // y = 1 - u z, scalar.rs:126
        if (A.result_vars) {
            const uint64_t vb = O.var_base + L.var_off;
            const uint64_t nz = err ? 1 : 3;
            A.result_vars[2 * item] = vb + 5 + nz + 3;
            A.result_vars[2 * item + 1] = vb + 5 + nz + 4 + 2;
        }
    }
    __device__ static void item(const Args &A, const EmitOut &O, uint64_t item, const uint4 *, ItemRec &R) {
        Loads L;
        item_load(A, O, item, L);
        // every load of the item (and the tile's offsets, issued just before) is in flight before the first use: left to
        // itself the scheduler interleaves the record's LDS writes with the loads and the ten loads leave in three
        // batches, each waited for -- three global round trips where one will do, in a workgroup that lives for little else
        __builtin_amdgcn_sched_barrier(0);
        item_from(A, O, item, R, L);
    }
    template <class R_>
    __device__ static void row(const R_ &R, uint64_t vbase, uint64_t zero_var, uint32_t j, RowOut &r) {
        const uint32_t nz = R.err ? 1 : 3;
        if (j < nz) is_non_zero_row(j, vbase + 0, vbase + 5, zero_var, r);
        else if (j < nz + 4) select_one_row(j - nz, vbase + 1, vbase + 2, vbase + 5 + nz, r);
        else maybe_equal_row(j - nz - 4, vbase + 3, vbase + 4, vbase + 5 + nz + 4, r);
    }
    template <class R_>
    __device__ static void selectors(const Args &, const R_ &R, uint32_t j, const uint4 *table, uint32_t h, uint4 out[5]) {
        RowOut r;
        row(R, 0, 0, j, r);
        row_values(r, table, h, out);
    }
    template <class R_>
    __device__ static void wires(const Args &, const EmitOut &O, const R_ &R, uint64_t, uint64_t vbase, uint32_t j,
                                 uint64_t out[3]) {
        RowOut r;
        row(R, vbase, O.zero_var, j, r);
        out[0] = r.w[0]; out[1] = r.w[1]; out[2] = r.w[2];
    }
    // the variable table as an image (emit.hpp, vars_image_kernel): the item's slots, shared out over four waves.  `img` is
    // where the item's first variable lies in the tile's image, `nvars` how many it has (13: it stopped at its error and
    // has no inv / one), `vb` its first Variable.
#ifndef PG_IMAGE_W
#define PG_IMAGE_W 64
#endif
    static constexpr int kImageW = PG_IMAGE_W, kImageParts = 4;
    __device__ static void put(uint4 *img, uint32_t slot, const Fr &x) {
        FrVec t;
        t.f = x;
        img[2 * slot] = t.v[0];
        img[2 * slot + 1] = t.v[1];
    }
    // what a part's lane reads from memory (image_load: loads only, so that a tile's reads can be in flight while the
    // tile before it is built and stored) and what it makes of it (image_build)
    struct ImageLoads {
        Fr f0, f1;
    };
    __device__ static void image_load(const Args &A, const EmitOut &O, uint64_t item, uint32_t part, ImageLoads &L) {
        if (part == 0) {
            L.f0 = load_fr(A.v, item);
            L.f1 = L.f0;
        } else if (part == 1) {
            L.f0 = load_fr(A.y, item);
            L.f1 = load_fr(A.s, item);
        } else if (part == 2) {
            L.f0 = load_fr(A.a, item);
            L.f1 = load_fr(A.b, item);
        } else {  // the two inverses, from the pre-pass's compact output: element e of item i at [e * batch + i]
            L.f0 = load_fr(O.inv, item);
            L.f1 = load_fr(O.inv, O.batch + item);
        }
    }
    __device__ static void image_build(const Args &A, uint64_t item, uint32_t part, uint32_t nvars, uint4 *img, uint64_t vb,
                                       const ImageLoads &L) {
        const bool err = nvars != kUniformVars;
        const uint32_t tail = err ? 6 : 8;  // one' sy oms out | u z yeq
        if (part == 0) {  // v, var_assigned = v (scalar.rs:69), the two constants (scalar.rs:83, :41), the results' Variables
            put(img, 0, L.f0);
            put(img, 5, L.f0);
            if (!err) put(img, 7, fr_one());
            put(img, tail, fr_one());
            if (A.result_vars) {
                A.result_vars[2 * item] = vb + tail + 3;
                A.result_vars[2 * item + 1] = vb + tail + 6;
            }
        } else if (part == 1) {  // y, s and select_one's three values
            put(img, 1, L.f0);
            put(img, 2, L.f1);
            const Fr sy = fr_mul(L.f0, L.f1), oms = fr_sub(fr_one(), L.f1);  // scalar.rs:43, :45-50
            put(img, tail + 1, sy);
            put(img, tail + 2, oms);
            put(img, tail + 3, fr_add(sy, oms));                              // scalar.rs:53-58
        } else if (part == 2) {  // a, b and maybe_equal's difference and result
            put(img, 3, L.f0);
            put(img, 4, L.f1);
            const Fr u = fr_sub(L.f0, L.f1);                                  // scalar.rs:111-117
            put(img, tail + 4, u);
            put(img, tail + 6, fr_is_zero(u) ? fr_one() : fr_zero());         // y = 1 - u z, scalar.rs:126
        } else {
            if (!err) put(img, 6, L.f0);                                      // scalar.rs:77
            put(img, tail + 5, L.f1);                                         // scalar.rs:122-123 (0 when a = b)
        }
    }
    // variable kc of a full-shape item: [v y s a b | va inv one | one' sy oms out | u z yeq]
    __device__ static Fr var_value_full(const Args &, const ItemRec &R, const uint4 *, uint32_t kc) {
        if (kc < 5) return R.d[kc];
        if (kc == 5) return R.d[0];                          // var_assigned
        if (kc == 7 || kc == 8) return fr_one();             // scalar.rs:83, :41
        if (kc >= 9 && kc <= 12) return R.d[kc - 4];         // sy oms out u
        if (kc == 14) return R.yeq ? fr_one() : fr_zero();
        return R.d[kc == 6 ? 9 : 10];                        // 6: v^-1, 13: (a - b)^-1 (asked for by the variables-only launch)
    }
    // an item that stopped at its error has no inv / one: its variables 6.. are the full shape's 8..
    __device__ static Fr var_value(const Args &A, const ItemRec &R, const uint4 *t, uint32_t k) {
        return var_value_full(A, R, t, (R.err && k >= 6) ? k + 2 : k);
    }
};

__global__ __launch_bounds__(kThreads) void scalar_mix_plan_kernel(const uint4 *v_in, uint64_t batch, uint32_t *rows, uint32_t *vars,
                                                                  uint8_t *err_mask, uint32_t *err_count, const PlanScan P) {
    uint32_t r[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x * 4 + k;
        if (i < batch) {
            const bool err = fr_is_zero(load_fr(v_in, i));
            r[k] = err ? 8 : 10;
            v[k] = err ? 13 : 15;
            if (err_mask) err_mask[i] = err ? 1 : 0;
            if (err) atomicAdd(err_count, 1u);
        }
    }
    plan_store(P, r, v, batch, rows, vars);
}

}  // namespace pg
