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
    struct alignas(16) ItemRec { Fr u, z; };  // (z: when the pre-pass has left the call's inverses dense, EmitOut::inv_in_place)
    static constexpr int W = 256;
    static constexpr int kInv = 1;
    static constexpr bool kInvDense = true;
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
        fetch_inverse(O, item, &R.z);
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
        if (k == 1) return R.z;                         // (asked for only when the record holds it)
        return fr_is_zero(R.u) ? fr_one() : fr_zero();  // scalar.rs:126
    }
};

// ---- is_non_zero (ragged: an item whose value is 0 stops after 1 row / 1 variable) ----------
struct IsNonZeroGD {
    using Args = ScalarArgs;
    struct alignas(16) ItemRec { Fr value, inv; };  // (inv: when the pre-pass has left the call's inverses dense)
    static constexpr int W = 256;
    static constexpr int kInv = 1;
    static constexpr bool kInvDense = true;
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
    __device__ static void item(const Args &A, const EmitOut &O, uint64_t item, const uint4 *, ItemRec &R) {
        fetch_inverse(O, item, &R.inv);
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
        if (k == 1) return R.inv;  // (asked for only when the record holds it)
        return fr_one();
    }
};

// plan kernels: per-item counts + the block sums of the prefix-sum pass in one launch (grid = blocks of kScanBlock items)
__global__ __launch_bounds__(kThreads) void is_non_zero_plan_kernel(const uint4 *value, uint64_t batch, uint32_t *rows,
                                                                   uint32_t *vars, uint8_t *err_mask, uint32_t *err_count,
                                                                   const PlanScan P) {
    uint32_t r[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0}, errs = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x * 4 + k;
        if (i < batch) {
            const bool err = fr_is_zero(load_fr(value, i));
            r[k] = v[k] = err ? 1 : 3;
            if (err_mask) err_mask[i] = err ? 1 : 0;
            errs += err ? 1 : 0;
        }
    }
    plan_store(P, r, v, batch, rows, vars, errs);
}

// ---- the fused mix (BASELINE config C3) ---------------------------------------------
// per item: v, y, s, a, b = 5 x add_input; is_non_zero(var v, v); conditionally_select_one(y, s);
// maybe_equal(a, b).  Variables: [v y s a b | va inv one | one' sy oms out | u z yeq] (15; 13 on Err),
// rows: [3 | 4 | 3] (10; 8 on Err).
struct ScalarMixArgs {
    const uint4 *v, *y, *s, *a, *b;
    uint64_t *result_vars;  // [batch][2]: select_one's output, maybe_equal's output
};

// The mix is a split gadget (emit.hpp): its ROWS depend on one bit per item (did is_non_zero stop at its error,
// scalar.rs:79) and are written by the two rows launches; its VARIABLES by scalar_mix_vars_kernel below.  This policy
// class is the rows' view.
struct ScalarMixGD {
    using Args = ScalarMixArgs;
    struct RowRec {
        uint32_t err;
    };
    using ItemRec = RowRec;
    static constexpr int W = 256, kRowsW = 256;
    static constexpr bool kRagged = true, kRecInRows = true, kUsePow2 = false;
    static constexpr uint32_t kUniformRows = 10, kUniformVars = 15;  // an item whose v is not 0
    static constexpr bool kPeriodic = true, kSplit = true;
    __device__ static void fill_table(const Args &, uint4 *, uint32_t) {}
    // the shape of an item, read off the call's prefix sums (the plan made them from v = 0, scalar.rs:73)
    __device__ static void item_rows(const Args &, const EmitOut &O, uint64_t item, const uint4 *, RowRec &rec) {
        rec.err = O.row_off[item + 1] - O.row_off[item] != kUniformRows ? 1u : 0u;
    }
    // item variables: [v y s a b | va inv one | one' sy oms out | u z yeq]; an item with v = 0 has no inv / one
    __device__ static void row(const RowRec &R, uint64_t vbase, uint64_t zero_var, uint32_t j, RowOut &r) {
        const uint32_t nz = R.err ? 1 : 3;
        if (j < nz) is_non_zero_row(j, vbase + 0, vbase + 5, zero_var, r);
        else if (j < nz + 4) select_one_row(j - nz, vbase + 1, vbase + 2, vbase + 5 + nz, r);
        else maybe_equal_row(j - nz - 4, vbase + 3, vbase + 4, vbase + 5 + nz + 4, r);
    }
    __device__ static void selectors(const Args &, const RowRec &R, uint32_t j, const uint4 *table, uint32_t h, uint4 out[5]) {
        RowOut r;
        row(R, 0, 0, j, r);
        row_values(r, table, h, out);
    }
    __device__ static void wires(const Args &, const EmitOut &O, const RowRec &R, uint64_t, uint64_t vbase, uint32_t j,
                                 uint64_t out[3]) {
        RowOut r;
        row(R, vbase, O.zero_var, j, r);
        out[0] = r.w[0]; out[1] = r.w[1]; out[2] = r.w[2];
    }
};

// ---- the mix's variable table: inversion and writing in ONE launch ---------------------------------------------------
// Every item needs two inverses (v^-1 for is_non_zero, scalar.rs:73-77; (a - b)^-1 for maybe_equal, scalar.rs:121-123) and
// 15 (13) variables that are its inputs, those inverses and three cheap values.  A WAVE owns 32 * ipl consecutive items
// and works on 32 of them per step, TWO LANES PER ITEM: lane p < 32 has the item's v side (v, y, s; the chain of the v's),
// lane 32 + p its a - b side (a, b; the chain of the differences).  Every lane runs Montgomery's trick on its own chain of
// ipl elements without the elements or their inverses ever leaving the lane:
//   forward  step m: load the element; the lane's running product BEFORE it goes to `scratch` (32 bytes per element: the
//            one thing that is parked in memory); multiply the element in (zeros are skipped and keep a zero inverse,
//            unwrap_or(zero), scalar.rs:122)
//   one inversion per lane (fr_invert_or_zero)
//   backward step m: load the element again with the rest of the item's inputs and the parked product; two multiplications
//            give the inverse and move the running inverse on, one more (v side) gives y * s; between them the two lanes now
//            hold ALL of the item's variables
//   the wave lays the variables of its 32 items out in a private LDS image exactly as they lie in memory (ragged: an item
//   that stopped at its error has 13; ballots give every lane its item's offset), and copies the image out linearly -- 16
//   bytes per lane, every wave store one contiguous KiB.
// Waves meet only around the inversion, and only in PAIRS (one inversion per pair of waves, two LDS flags: below).  Two lanes per item halve what a lane carries (half an
// item and its prefetch: 226 registers with the forward pass's four prefetch sets) and make every load and store of a step
// a full-width access.  The launch is built for TWO waves per SIMD (amdgpu_num_vgpr(256), 120 KB of LDS = one workgroup per
// CU): the multiplier needs two (one wave alone issues every other cycle), and the rows launches do NOT run beside it --
// they follow it on the same stream (capi.hip, launch_mix; side by side they lose in every arrangement measured,
// profiles/NOTES_r03.md section 2).  Inputs are read once per pass, the parked products are 64 B per item each way: against
// the three-launch form this replaced (pre-pass with 128 B per element of scratch traffic, a compact inverse array, a
// variable-table launch that read all five inputs again) the step's reads fell from 3.5 x the inputs to 2 x and nothing
// waits for a pre-pass any more.  Geometry: ipl = 16 at 2^20 items = 2048 waves = two per SIMD.  What bounds the launch
// (profiles/NOTES_r04.md): the forward pass is bound by the chip's memory system, not by a CU (half of the workgroups alone
// run it twice as fast), the inversion by latency with HBM idle, the backward pass by HBM.
constexpr uint32_t kMixMaxIpl = 32;  // steps per wave at most (beyond: more workgroups than fit at once)

__device__ __forceinline__ void mix_load16(FrVec &d, const uint4 *col, uint64_t i) {
    d.v[0] = col[2 * i];
    d.v[1] = col[2 * i + 1];
}
__device__ __forceinline__ Fr fr_select(bool c, const Fr &a, const Fr &b) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}
__device__ __forceinline__ void mix_put(uint4 *img, uint32_t slot, const Fr &x) {
    FrVec t;
    t.f = x;
    img[2 * slot] = t.v[0];
    img[2 * slot + 1] = t.v[1];
}

#ifndef PG_MIX_VARS_VGPRS
#define PG_MIX_VARS_VGPRS 256  // two waves per SIMD (the 15 KB images allow no more)
#endif
// A note on waiting.  gfx950 counts a wave's loads AND stores in one counter (vmcnt), in issue order: "wait for the data
// I prefetched a step ago" also waits for every store issued before that prefetch, and -- unless the number of stores
// issued since is a compile-time constant -- for those as well.  Beside the rows launch, which keeps the memory system's
// write queues full, an acknowledged store is tens of microseconds away; a step that waits for its own last stores
// crawls (this launch: 250 us alone, 370-580 us beside the rows, before).  So every step issues a FIXED number of memory
// instructions, none of them under a lane mask the compiler would branch around (a lane with nothing to store stores to
// `sink`, 16 bytes per lane that nobody reads), and the loops are entered with nothing in flight: the compiler can then
// count, and waits for a step's prefetch with the step's stores still on their way (s_waitcnt vmcnt(17), not 0).
//
// PLAN: the launch also makes the call's prefix sums (pg_scalar_mix_planned_batch), which are nothing but the count of
// items with v = 0 before each item: rows before item i = 10 i - 2 e_i, variables 15 i - 2 e_i.  A wave counts its own in
// the forward pass; the workgroup's last wave publishes the workgroup's count and adds up its predecessors' -- the
// single-pass scan of the plan kernels (emit.hpp, plan_finish: one 64-bit word per workgroup that carries the count and two
// flag bits, touched by relaxed atomics only; a workgroup waits for workgroups of lower index, which were dispatched
// before it; every wait is bounded; the last one to finish zeroes the words) -- while the inversions run and it would
// only wait.  The offsets are written beside the variables, the totals by the workgroup that owns the last item.
struct MixPlan {
    unsigned long long *agg;      // per workgroup: flags | count of failing items (zero between launches); [cap] = workgroups done
    uint64_t *row_off, *var_off;  // the call's outputs, batch + 1 entries each
    uint8_t *err_mask;            // per item: it stopped at is_non_zero's error (optional)
    PlanTotals *host;
    uint32_t cap, nwaves;         // index of the done counter; workgroups of the launch
};

// waves per workgroup: one workgroup per CU (8 x 15 KB of images), two waves per SIMD; waves w and w + 4 share an inversion
constexpr int kMixWaves = 8;

#if defined(PG_MIX_STAMPS)  // timing build (tools/mix_phases.py): time the waves spend in each phase, summed over the launch
__device__ unsigned long long g_mix_phase_ticks[12];
#define PG_STAMP(k)                                                                                              \
    do {                                                                                                         \
        const unsigned long long now_ = wall_clock64();                                                          \
        if (lane == 0) atomicAdd(&g_mix_phase_ticks[k], now_ - stamp_);                                          \
        stamp_ = now_;                                                                                           \
    } while (0)
#else
#define PG_STAMP(k) do {} while (0)
#endif

template <bool PLAN>
__global__ __launch_bounds__(kMixWaves * 64) __attribute__((amdgpu_num_vgpr(PG_MIX_VARS_VGPRS))) void scalar_mix_vars_kernel(
    const ScalarMixArgs A, const EmitOut O, uint32_t ipl, uint4 *scratch, uint4 *sink, const MixPlan P) {
    __shared__ uint4 s_img[kMixWaves][32 * 15 * 2];
    __shared__ uint64_t s_errs[kMixWaves], s_before;  // PLAN: failing items of the workgroup's waves, and before the workgroup
    __shared__ uint4 s_consts[T_POW * 2];             // the rows' constants (early rows, below)
    __shared__ uint32_t s_before_ready;               // PLAN: s_before is written (every wave waits for it before its backward pass)
    __shared__ uint32_t s_pair_in[4], s_pair_out[4];  // pair p: the upper wave's products are in LDS / the lower wave's answer is
    __shared__ uint32_t s_arrived;                    // PLAN: waves whose count of failing items is in s_errs
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 31;
    const bool bside = lane >= 32;  // the a - b side of item p
    if (threadIdx.x < 4) s_pair_in[threadIdx.x] = s_pair_out[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_arrived = s_before_ready = 0;
    fill_common_table(s_consts, nullptr, threadIdx.x, T_POW);
    __syncthreads();  // (the only workgroup barrier of the launch: the flags and the rows' constants, before any wave's first load)
    const uint64_t gw = (uint64_t)blockIdx.x * kMixWaves + wave;  // the wave's index in the launch
    const uint64_t chunk0 = gw * 32 * ipl;                         // the wave's first item
    // (the launch's last workgroup may hold waves without items: they take no step, but stand at the two barriers)
    if (chunk0 >= O.batch) ipl = 0;
    const uint64_t last = O.batch - 1;
    uint4 *img = s_img[wave];
    // (a lane's loads are unconditional: where a side has nothing to read it re-reads what it has -- a cache hit)
    const uint4 *in0 = bside ? A.a : A.v, *in1 = bside ? A.b : A.y, *in2 = bside ? A.b : A.s, *fw1 = bside ? A.b : A.v;
    uint4 *park = scratch + (bside ? O.batch : 0);  // [2][2][batch] 16-byte halves: a half-wave's store is 512 contiguous bytes
    sink += lane;

#if defined(PG_MIX_STAMPS)
    unsigned long long stamp_ = wall_clock64();
#endif
    // ---- forward: running products ------------------------------------------------------------------------------
    Fr acc = fr_one();
    uint32_t errs = 0;  // PLAN: the wave's items with v = 0
    {
        // A forward step is short -- one multiplication, ~1.2 us with two waves per SIMD -- and a load from HBM is not: the
        // elements are fetched THREE steps ahead into four sets of registers that take turns (the loop is unrolled by four so
        // that no set is ever copied into another: a copy would make its step wait for the load it has just issued).  Steps
        // past the wave's last one (ipl is rounded up to a multiple of four) fetch the last step again and change nothing.
        FrVec b0[4], b1[4];
        auto fetch = [&](uint32_t m, FrVec &d0, FrVec &d1) {
            m = m < ipl ? m : ipl - 1;
            uint64_t i = chunk0 + (uint64_t)m * 32 + p;
            i = i < last ? i : last;  // lanes past the end re-read the last item (masked below)
            mix_load16(d0, in0, i);
            mix_load16(d1, fw1, i);
        };
        auto step = [&](uint32_t m, const FrVec &e0, const FrVec &e1) {
            const uint64_t i = chunk0 + (uint64_t)m * 32 + p;
            const bool valid = m < ipl && i <= last;
            {
                FrVec t;
                t.f = acc;
                store16(valid ? park + i : sink, t.v[0]);
                store16(valid ? park + 2 * O.batch + i : sink, t.v[1]);
            }
            const Fr x = bside ? fr_sub(e0.f, e1.f) : e0.f;  // scalar.rs:121 / :73
            const bool nz = valid && !fr_is_zero(x);
            if constexpr (PLAN) errs += (uint32_t)__popcll(__ballot(!bside && valid && !nz));  // scalar.rs:73-80
            const Fr t = fr_mul(acc, x);
            acc = fr_select(nz, t, acc);
        };
        if (ipl) {
            fetch(0, b0[0], b1[0]);
            fetch(1, b0[1], b1[1]);
            fetch(2, b0[2], b1[2]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (uint32_t m = 0; m < ipl; m += 4) {
            fetch(m + 3, b0[3], b1[3]);
            step(m, b0[0], b1[0]);
            fetch(m + 4, b0[0], b1[0]);
            step(m + 1, b0[1], b1[1]);
            fetch(m + 5, b0[1], b1[1]);
            step(m + 2, b0[2], b1[2]);
            fetch(m + 6, b0[2], b1[2]);
            step(m + 3, b0[3], b1[3]);
        }
    }

    PG_STAMP(0);  // forward
    // ---- what the backward pass loads per step (declared here: its first step is fetched while the inversion runs) ------
    struct In {
        FrVec f0, f1, f2, pk;  // v y s | a b -, the parked product
        uint64_t base;
    };
    In c, n;
    auto fetch = [&](uint32_t m, In &d) {
        const uint64_t i0 = chunk0 + (uint64_t)m * 32;
        uint64_t i = i0 + p;
        i = i < last ? i : last;
        mix_load16(d.f0, in0, i);
        mix_load16(d.f1, in1, i);
        mix_load16(d.f2, in2, i);
        d.pk.v[0] = park[i];
        d.pk.v[1] = park[2 * O.batch + i];
        if constexpr (!PLAN) d.base = O.var_off[i0 < O.batch ? i0 : O.batch];  // first variable of the step's 32 items, relative to the call
    };
    // (PLAN: the failing items before this wave are counted while the inversions run, below)
    uint64_t errs_before = 0;

    PG_STAMP(1);  // look-back
    // ---- one inversion per PAIR of waves --------------------------------------------------------------------------
    // The inversion is 20 k vector instructions against the 25 k of everything else a wave does here, and a wave pays it
    // whether one lane wants an inverse or sixty-four.  Waves w and w + 4 of the workgroup -- the two that share a SIMD --
    // therefore share one: the upper wave hands its lanes' products to the lower one (through LDS), which multiplies them to
    // its own, inverts the product and takes the two inverses apart again (three multiplications).  Every SIMD then runs exactly
    // one inversion, alone -- the chain of dependent instructions that it is gains nothing from a second wave beside it, and one
    // inversion for all eight waves (14 multiplications by a single wave first) measured 15 us slower.
    Fr accinv;
    {
        // Waves w and w + 4 -- the two that share a SIMD -- share ONE inversion (above), and that pair is all that has to meet:
        // the upper wave hands its lanes' products over through its own image area and raises a flag, the lower one inverts
        // the product of both, hands the upper wave's inverse back and raises another.  No workgroup barrier: a pair whose
        // forward passes end early inverts early and is in its backward pass -- which is bound by HBM -- while slower pairs
        // still invert with HBM idle (a barrier made every wave wait 19 us on average for the workgroup's slowest).  PLAN: the one
        // thing all waves do wait for is the count of failing items before the workgroup (s_before_ready), which the look-back
        // delivers ~35 us after the workgroup's last forward pass -- every workgroup before this one has to be through its own.
        static_assert(kMixWaves == 8, "waves w and w + 4 pair up");
        const uint32_t pair = wave & 3;
        uint4 *slot = &s_img[4 + pair][0] + lane;  // [half][lane] in the upper wave's image area (idle until its backward pass)
        auto get = [&]() {
            FrVec t;
            t.v[0] = slot[0];
            t.v[1] = slot[64];
            return t.f;
        };
        auto set = [&](const Fr &x) {
            FrVec t;
            t.f = x;
            slot[0] = t.v[0];
            slot[64] = t.v[1];
        };
        auto raise = [&](uint32_t *flag) {  // (the wave's LDS writes above are ordered before it)
            if (lane == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        auto await = [&](uint32_t *flag, uint32_t want) {
            while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(4);
        };
#if defined(PG_MIX_STAMPS)
        const unsigned long long after_b1_ = wall_clock64();
#endif
        if (PLAN && lane == 0) {
            s_errs[wave] = errs;  // (a wave without items counted none)
            __hip_atomic_fetch_add(&s_arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (wave < 4) {
            await(&s_pair_in[pair], 1);
#if defined(PG_MIX_STAMPS)
            const unsigned long long t0_ = wall_clock64();
            if (lane == 0) atomicAdd(&g_mix_phase_ticks[4], t0_ - after_b1_);  // a lower wave's wait for its partner's products
#endif
            const Fr other = get();
            const Fr t = fr_invert_or_zero(fr_mul(acc, other));  // (a product of non-zero elements, or mont(1))
            accinv = fr_mul(t, other);
            set(fr_mul(t, acc));
            raise(&s_pair_out[pair]);
#if defined(PG_MIX_STAMPS)
            if (lane == 0) atomicAdd(&g_mix_phase_ticks[5], wall_clock64() - t0_);  // the inversion itself (lower waves)
#endif
        } else {
            set(acc);
            raise(&s_pair_in[pair]);
            if (PLAN && wave == kMixWaves - 1) {
                await(&s_arrived, kMixWaves);  // every wave's count is in s_errs
                // PLAN: failing items in the WORKGROUPS before this one, by the last wave while it would only wait for its
                // inverse: the decoupled look-back of the plan kernels, one word per workgroup, 64 predecessors per round.
                // (Per wave instead -- 2048 words, every wave adding up its own predecessors ahead of the inversion -- it took
                // every wave 21 us: thirty-two rounds of agent-scope atomics, all waves of the chip at once.)
                const uint32_t b = blockIdx.x;
                uint64_t mine = 0;
#pragma unroll
                for (int w = 0; w < kMixWaves; w++) mine += s_errs[w];
                if (lane == 0 && b > 0) __hip_atomic_exchange(&P.agg[b], kAggA | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint64_t before = 0;
                bool gave_up = false;
                int64_t back = (int64_t)b - 1;
                for (bool more = b > 0; more; back -= 64) {
                    const int64_t j = back - (int64_t)lane;
                    unsigned long long w = kAggP;  // before workgroup 0: the empty prefix
                    if (j >= 0) {
                        w = plan_rmw_read(&P.agg[j]);
                        for (uint32_t polls = 0; !(w >> 62); w = plan_rmw_read(&P.agg[j])) {
                            if (++polls > kPlanSpinLimit) { gave_up = true; w = kAggP; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    const uint64_t has_prefix = __ballot((w & kAggP) != 0);
                    const uint32_t first = has_prefix ? (uint32_t)__ffsll((unsigned long long)has_prefix) - 1 : 64u;  // the nearest one
                    before += wave_sum(lane <= first ? w & 0xffffffffull : 0);
                    more = has_prefix == 0;
                }
                if (lane == 0) {
                    if (gave_up) P.host->pad = 1;
                    __hip_atomic_exchange(&P.agg[b], kAggP | (before + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_before = before;
                    __hip_atomic_store(&s_before_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (b + 1 == P.nwaves) {  // the workgroup of the last item: the totals
                        const uint64_t e_all = before + mine;
                        P.row_off[O.batch] = 10 * O.batch - 2 * e_all;
                        P.var_off[O.batch] = 15 * O.batch - 2 * e_all;
                        P.host->n_gates = 10 * O.batch - 2 * e_all;
                        P.host->n_vars = 15 * O.batch - 2 * e_all;
                        P.host->errs = (uint32_t)e_all;
                    }
                }
                // the last workgroup to get here has every look-back behind it: the words go back to zero
                unsigned long long done = 0;
                if (lane == 0) done = __hip_atomic_fetch_add(&P.agg[P.cap], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                done = __shfl(done, 0, 64);
                if (done == P.nwaves - 1) {
                    for (uint32_t j = lane; j < P.nwaves; j += 64) __hip_atomic_exchange(&P.agg[j], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lane == 0) __hip_atomic_exchange(&P.agg[P.cap], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            // ---- early rows: what the waiting waves do meanwhile --------------------------------------------------------
            // A pair's inversion takes ~42 us during which nothing of the pair moves: the lower wave computes a chain of dependent
            // instructions, the upper one waits, and HBM is idle whenever most pairs are there.  A store stream needs neither
            // many waves nor the multiplier (four waves per CU with a tight store loop write at the chip's rate, and 240 MB
            // written beside a 50-us arithmetic phase cost it 5 us: tools/probes/store_occupancy.hip, hidden_stores.hip) -- so
            // the waiting waves write the rows of the first early_tiles row tiles of the workgroup's items, and the rows launch
            // that follows leaves those alone (emit.hpp, early_rows_candidate: both sides decide from the same thing, "no item
            // of the span stopped early").  PLAN: where those rows lie is only known with the prefix (~32-35 us after the
            // workgroup's last forward pass), which leaves room for ONE tile; a second one costs the launch more than it saves
            // the rows launch (profiles/NOTES_r05.md).
            if (O.early_tiles) {
                using GD = ScalarMixGD;
                constexpr uint32_t W = GD::kRowsW, R = GD::kUniformRows, VV = GD::kUniformVars;
                const uint64_t s0 = (uint64_t)blockIdx.x * O.early_span;
                if (s0 < O.batch) {
                    const uint64_t s1 = s0 + O.early_span < O.batch ? s0 + O.early_span : O.batch;
                    bool full;
                    uint64_t row0, var0;
                    if constexpr (PLAN) {
                        await(&s_before_ready, 1);  // (implies every wave's count: the look-back waited for them)
                        uint64_t mine = 0;
#pragma unroll
                        for (int w = 0; w < kMixWaves; w++) mine += s_errs[w];
                        full = mine == 0;
#if defined(PG_MIX_STAMPS)
                        if (lane == 0) atomicAdd(&g_mix_phase_ticks[7], wall_clock64() - after_b1_);  // until the prefix is known (upper waves)
#endif
                        const uint64_t before = full ? s_before : 0;
                        row0 = R * s0 - 2 * before;
                        var0 = VV * s0 - 2 * before;
                    } else {
                        row0 = O.row_off[s0];
                        var0 = O.var_off[s0];
                        full = O.row_off[s1] - row0 == (s1 - s0) * R;
                    }
                    if (full) {
                        const uint32_t tid = (wave - 4) * 64 + lane;
                        uint4 v[5];
                        periodic_lane_selectors<GD>(A, s_consts, tid, v);
                        const uint32_t tiles = (uint32_t)((s1 - s0) / W) < O.early_tiles ? (uint32_t)((s1 - s0) / W) : O.early_tiles;  // complete tiles only
                        for (uint32_t t = 0; t < tiles; t++) {
#if defined(PG_MIX_STAMPS)
                            const unsigned long long t0_ = wall_clock64();
#endif
                            periodic_tile_rows<GD>(A, O, v, tid, s0 + (uint64_t)t * W, W, row0 + (uint64_t)t * W * R, var0 + (uint64_t)t * W * VV);
#if defined(PG_MIX_STAMPS)
                            if (lane == 0) atomicAdd(&g_mix_phase_ticks[8 + (t < 3 ? t : 3)], wall_clock64() - t0_);  // issuing tile t's stores
#endif
                        }
                    }
                }
            }
#if defined(PG_MIX_STAMPS)
            if (lane == 0) atomicAdd(&g_mix_phase_ticks[6], wall_clock64() - after_b1_);  // look-back / early rows (upper waves)
#endif
            await(&s_pair_out[pair], 1);
            accinv = get();
        }
        if constexpr (PLAN) {  // before this wave = before the workgroup + in its earlier waves
            await(&s_before_ready, 1);
            errs_before = s_before;
            for (uint32_t w = 0; w < wave; w++) errs_before += s_errs[w];
        }
    }

    PG_STAMP(2);  // inversion (and waiting for it)
    // ---- backward: inverses, the item's variables, the image ------------------------------------------------------
    if (ipl) fetch(ipl - 1, c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // The image of a step is copied out DURING THE NEXT STEP, five stores after each of its three multiplications: all
    // waves of the chip run in step, and 15 KB per wave at once is 31 MB that the memory system takes 5 us to absorb --
    // with the copy at the end of its own step the waves stood at their stores for a third of the step (issue-stalled 24 %
    // of their cycles, profiles/NOTES_r03.md).  A step therefore keeps what it computes in registers and lays its image
    // down at its end, when the previous one has been read out.
    uint32_t ptotal = 0;            // scalars of the image that waits to be copied out (0: none yet)
    uint4 *pdst = sink;
    auto flush = [&](uint32_t k0, uint32_t k1) {
#pragma unroll
        for (uint32_t k = k0; k < k1; k++) {  // (a fixed number of stores whatever the items' shapes)
            const uint32_t o = lane + 64 * k;
            store16(o < ptotal * 2 ? pdst + o : sink, img[o]);
        }
    };
    for (uint32_t mm = ipl; mm-- > 0;) {
        fetch(mm > 0 ? mm - 1 : 0, n);  // (the last step fetches itself again)
        const uint64_t i = chunk0 + (uint64_t)mm * 32 + p;
        const bool valid = i <= last;
        const Fr x = bside ? fr_sub(c.f0.f, c.f1.f) : c.f0.f;  // a - b (scalar.rs:111-117, :121) | v
        const bool nz = valid && !fr_is_zero(x);
        // which items stopped at is_non_zero's error (scalar.rs:79): the v sides know, the a - b sides need it for their slots
        const uint32_t vmask = (uint32_t)__ballot(valid), emask = (uint32_t)__ballot(!bside && valid && !nz);
        const uint32_t below = (1u << p) - 1;
        const uint32_t off = 15u * (uint32_t)__popc(vmask & below) - 2u * (uint32_t)__popc(emask & below);
        const uint32_t total = 15u * (uint32_t)__popc(vmask) - 2u * (uint32_t)__popc(emask);
        const bool err = (emask >> p) & 1;
        if constexpr (PLAN) {  // failing items before this step's = before the wave + in the wave's earlier steps
            errs -= (uint32_t)__popc(emask);
            const uint64_t eb = errs_before + errs, i0 = chunk0 + (uint64_t)mm * 32;
            c.base = 15 * i0 - 2 * eb;
            const uint64_t e_i = eb + (uint64_t)__popc(emask & below);  // ... and before this item
            uint64_t *po = reinterpret_cast<uint64_t *>(sink);
            if (valid) po = bside ? P.var_off + i : P.row_off + i;
            *po = (bside ? 15 : 10) * i - 2 * e_i;
            if (P.err_mask) {
                uint8_t *pm = reinterpret_cast<uint8_t *>(sink);
                if (valid && !bside) pm = P.err_mask + i;
                *pm = err ? 1 : 0;
            }
        }
        const uint32_t tail = err ? 6 : 8;  // item variables: [v y s a b | va inv one | one' sy oms out | u z yeq]
        const Fr inv = fr_select(nz, fr_mul(accinv, c.pk.f), fr_zero());  // scalar.rs:77 | :122-123
        flush(0, 5);
        accinv = fr_select(nz, fr_mul(accinv, x), accinv);
        flush(5, 10);
        const Fr sy = fr_mul(c.f1.f, c.f2.f);  // scalar.rs:43 (the v sides')
        flush(10, 15);
        asm volatile("" ::: "memory");  // (LDS executes a wave's instructions in order: the reads above are ahead of the writes below)
        uint4 *it = img + 2 * off;
        if (valid) {
            if (!bside) {
                mix_put(it, 0, c.f0.f);  // 5 x add_input
                mix_put(it, 1, c.f1.f);
                mix_put(it, 2, c.f2.f);
                mix_put(it, 5, c.f0.f);  // var_assigned, scalar.rs:69
                if (!err) mix_put(it, 6, inv);
                const Fr oms = fr_sub(fr_one(), c.f2.f);  // scalar.rs:45-50
                mix_put(it, tail + 1, sy);
                mix_put(it, tail + 2, oms);
                mix_put(it, tail + 3, fr_add(sy, oms));   // scalar.rs:53-58
            } else {
                mix_put(it, 3, c.f0.f);
                mix_put(it, 4, c.f1.f);
                if (!err) mix_put(it, 7, fr_one());  // scalar.rs:83
                mix_put(it, tail, fr_one());         // scalar.rs:41
                mix_put(it, tail + 4, x);            // u
                mix_put(it, tail + 5, inv);
                mix_put(it, tail + 6, nz ? fr_zero() : fr_one());  // y = 1 - u z, scalar.rs:126
            }
        }
        {   // the results' Variables: select_one's output (the v side stores it), maybe_equal's (the a - b side)
            uint64_t *rv = reinterpret_cast<uint64_t *>(sink);
            if (valid && A.result_vars) rv = A.result_vars + 2 * i + (bside ? 1 : 0);
            *rv = O.var_base + c.base + off + tail + (bside ? 6 : 3);
        }
        ptotal = total;
        pdst = O.vars + c.base * 2;
        asm volatile("" ::: "memory");
        c = n;
    }
    if (ipl) flush(0, 15);  // the wave's last image
    PG_STAMP(3);  // backward
}

__global__ __launch_bounds__(kThreads) void scalar_mix_plan_kernel(const uint4 *v_in, uint64_t batch, uint32_t *rows, uint32_t *vars,
                                                                  uint8_t *err_mask, uint32_t *err_count, const PlanScan P) {
    uint32_t r[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0}, errs = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x * 4 + k;
        if (i < batch) {
            const bool err = fr_is_zero(load_fr(v_in, i));
            r[k] = err ? 8 : 10;
            v[k] = err ? 13 : 15;
            if (err_mask) err_mask[i] = err ? 1 : 0;
            errs += err ? 1 : 0;
        }
    }
    plan_store(P, r, v, batch, rows, vars, errs);
}

}  // namespace pg
