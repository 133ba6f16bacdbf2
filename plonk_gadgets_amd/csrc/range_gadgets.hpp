// range_gadgets.hpp -- range_check / max_bound as policies of the streaming
// writer (emit.hpp).
//
// One "bound block" is what max_bound (/root/reference/src/range.rs:82-113)
// or min_bound (:53-76) appends: one add row producing T (= max-1-x or x-min)
// followed by scalar_decomposition_gadget (:119-158) on T, which ends in
// maybe_equal (/root/reference/src/scalar.rs:105-140).  With n ladder bits a
// block has L = 2n+5 rows and VB = n+261 variables:
//     row 0        bound add   (x, x, T ; 0, -1|+1, 0, -1, mont(max-1) | mont(-min))
//     row 1        A0 const    (A0,A0,A0 ; 0, 1, 0, 0, 0)
//     row 2+2i     boolean     (b_i,b_i,b_i ; 1, 0, 0, -1, 0)
//     row 3+2i     ladder add  (b_i, A_i, A_{i+1} ; 0, mont(2^i), 1, -1, 0)
//     row 2n+2     u = A_n - T (A_n, T, U ; 0, 1, -1, -1, 0)
//     row 2n+3     y = 1 - u z (Z, U, Y ; -1, 0, 0, -1, 1)
//     row 2n+4     y u = 0     (Y, U, U ; 1, 0, 0, 0, 0)
//     var 0 = T, 1..256 = the 256 bits of canonical(T) (all allocated, range.rs:128-131),
//     257+i = A_i (i = 0..n), 258+n = U, 259+n = Z, 260+n = Y
//
// range_check (range.rs:27-43), preceded by AllocatedScalar::allocate
// (allocated_scalar.rs:27): variables [x | max block | min block | R], rows
// [max block | min block | (Y1, Y2, R ; 1, 0, 0, -1, 0)].
// max_bound alone: variables [x | block], rows [block].
#pragma once

#include "emit.hpp"
#include "invert.hpp"

namespace pg {

struct alignas(16) BoundRec {
    Fr Tm;  // T in Montgomery form (the block's first variable)
    Fr Tc;  // canonical integer of T
    Fr UZ[2];  // u = A_n - T, and its inverse z -- the pre-pass's (invert.hpp): held here when it has left the call's inverses
               // in the scratch array (!EmitOut::inv_in_place), else not used
};

// z of element s of the call, from the pre-pass's dense output
__device__ __forceinline__ void bound_fetch_z(const EmitOut &O, uint64_t s, BoundRec &b) { fetch_inverse(O, s, &b.UZ[1]); }

// u = accumulator - witness (scalar.rs:121) of a bound block: A_n - T with A_n = mont(T mod 2^n); 0 when T fits n bits
__device__ __forceinline__ Fr bound_u(const Fr &Tm, const Fr &Tc, uint32_t n) {
    if (!raw_has_high_bits(Tc, n)) return fr_zero();
    return fr_sub(fr_to_mont(raw_low_bits(Tc, n)), Tm);
}

// per-item arithmetic of one bound block (z = u^-1 or 0, scalar.rs:122, is the pre-pass's business)
__device__ __forceinline__ uint32_t bound_item(const Fr &Tm, uint32_t n, BoundRec &b) {
    const Fr Tc = fr_from_mont(Tm);  // scalar_to_bits -> to_bytes, range.rs:163
    b.Tm = Tm;
    b.Tc = Tc;
    const Fr u = bound_u(Tm, Tc, n);
    b.UZ[0] = u;
    return fr_is_zero(u) ? 1u : 0u;  // y = 1 - u z
}

// offset of z inside a bound block's variables
__device__ __forceinline__ uint32_t bound_z_offset(uint32_t n) { return 259 + n; }

// selector table ids of block row jj; is_min selects the min_bound flavour of row 0
__device__ __forceinline__ void bound_selector_ids(uint32_t jj, uint32_t n, bool is_min, uint32_t id[5]) {
    uint32_t qm = T_ZERO, ql = T_ZERO, qr = T_ZERO, qo = T_NEG1, qc = T_ZERO;
    if (jj >= 2 && jj < 2 * n + 2) {
        if (jj & 1) {  // ladder add, range.rs:147-151
            ql = T_POW + ((jj - 3) >> 1);
            qr = T_ONE;
        } else {  // boolean_gate, range.rs:144
            qm = T_ONE;
        }
    } else if (jj == 0) {  // range.rs:93-99 / :60-66
        ql = is_min ? T_ONE : T_NEG1;
        qc = is_min ? T_QC_B : T_QC_A;
    } else if (jj == 1) {  // add_witness_to_circuit_description(0), range.rs:139
        ql = T_ONE;
        qo = T_ZERO;
    } else if (jj == 2 * n + 2) {  // scalar.rs:111-117
        ql = T_ONE;
        qr = T_NEG1;
    } else if (jj == 2 * n + 3) {  // scalar.rs:126
        qm = T_NEG1;
        qc = T_ONE;
    } else {  // scalar.rs:129-138
        qm = T_ONE;
        qo = T_ZERO;
    }
    id[0] = qm; id[1] = ql; id[2] = qr; id[3] = qo; id[4] = qc;
}

constexpr uint32_t kWitnessWire = 0xffffffffu;  // "the witness Variable" (allocated by the item, or an existing one)

// wire variable offsets of block row jj; vb = offset of the block's first variable, x = kWitnessWire
__device__ __forceinline__ void bound_wire_offsets(uint32_t jj, uint32_t n, uint32_t vb, uint32_t x, uint32_t off[3]) {
    uint32_t a, b, c;
    if (jj >= 2 && jj < 2 * n + 2) {
        const uint32_t i = (jj - 2) >> 1;
        a = vb + 1 + i;  // b_i
        if (jj & 1) { b = vb + 257 + i; c = vb + 258 + i; }
        else { b = a; c = a; }
    } else if (jj == 0) {
        a = x; b = x; c = vb;  // w_r = x with q_r = 0 (not zero_var), range.rs:62,95
    } else if (jj == 1) {
        a = b = c = vb + 257;
    } else if (jj == 2 * n + 2) {
        a = vb + 257 + n; b = vb; c = vb + 258 + n;
    } else if (jj == 2 * n + 3) {
        a = vb + 259 + n; b = vb + 258 + n; c = vb + 260 + n;
    } else {
        a = vb + 260 + n; b = vb + 258 + n; c = b;
    }
    off[0] = a; off[1] = b; off[2] = c;
}

// value of block variable kk
__device__ __forceinline__ Fr bound_var_value(const BoundRec &B, uint32_t y, uint32_t kk, uint32_t n) {
    if (kk == 0) return B.Tm;
    if (kk <= 256) return raw_bit(B.Tc, kk - 1) ? fr_one() : fr_zero();  // range.rs:128-131
    if (kk <= 257 + n) {                                                  // A_i = mont(T mod 2^i), range.rs:152
        const uint32_t i = kk - 257;
        return i ? fr_to_mont(raw_low_bits(B.Tc, i)) : fr_zero();
    }
    if (kk <= 259 + n) {  // u, z (z is asked for only when the record holds it: !EmitOut::inv_in_place)
        const uint4 *p = reinterpret_cast<const uint4 *>(B.UZ) + 2 * (kk - (258 + n));
        FrVec v;
        v.v[0] = p[0];
        v.v[1] = p[1];
        return v.f;
    }
    return y ? fr_one() : fr_zero();  // scalar.rs:126
}

__device__ __forceinline__ void ids_to_values(const uint32_t id[5], const uint4 *table, uint32_t h, uint4 out[5]) {
#pragma unroll
    for (int c = 0; c < 5; c++) out[c] = table[2 * id[c] + h];
}

// ---- range_check ------------------------------------------------------------
struct RangeCheckGD {
    struct Args {
        Fr min_range, max_range;  // Montgomery form (public inputs)
        uint32_t n;               // ladder bits
        const uint4 *witness;
        const uint64_t *witness_vars;  // NULL: every item allocates its witness (AllocatedScalar::allocate) as its
                                       // first variable; else: the existing Variables the items range-check
        uint64_t *result_vars;
        const uint4 *pow2;
    };
    static constexpr int kInv = 2;
    static constexpr bool kInvDense = true;  // item() fetches the inverses when the pre-pass has left them dense
    static constexpr int kInvGroup = 4;  // elements a lane of the pre-pass fetches per round trip (inv_combine is register-hungry here)
    // element e of item: u of the max block (e = 0) / min block (e = 1)
    __device__ static void inv_operands(const Args &A, const EmitOut &, uint64_t item, uint32_t, FrVec &p, FrVec &q, uint32_t &) {
        p.v[0] = A.witness[item * 2];
        p.v[1] = A.witness[item * 2 + 1];
        q.f = p.f;
    }
    __device__ static Fr inv_combine(const Args &A, uint32_t e, const Fr &x, const Fr &, uint32_t) {
        // both bound blocks' T, the wanted one selected limb by limb (a ternary over two struct temporaries goes through
        // private memory)
        const Fr t0 = fr_sub(fr_sub(A.max_range, fr_one()), x), t1 = fr_sub(x, A.min_range);
        Fr Tm;
#pragma unroll
        for (int i = 0; i < 4; i++) Tm.l[i] = e == 0 ? t0.l[i] : t1.l[i];
        return bound_u(Tm, fr_from_mont(Tm), A.n);
    }
    // z of the max block (e = 0) / min block (e = 1)
    __device__ static uint4 *inv_slot(const Args &A, const EmitOut &O, uint64_t item, uint32_t e) {
        const uint64_t V = 2 * A.n + 523 + (A.witness_vars ? 0u : 1u), VB = A.n + 261;
        return O.vars + 2 * (item * V + (A.witness_vars ? 0u : 1u) + e * VB + bound_z_offset(A.n));
    }
    struct alignas(16) ItemRec {
        Fr x;
        BoundRec b[2];
        uint32_t y[2];
        uint32_t pad[2];
    };
    __device__ static bool is_inv_slot(const Args &A, const ItemRec &, uint32_t k) {
        const uint32_t x0 = A.witness_vars ? 0u : 1u, VB = A.n + 261, z = bound_z_offset(A.n);
        return k == x0 + z || k == x0 + VB + z;
    }
#ifndef PG_RC_W
#define PG_RC_W 32
#endif
    static constexpr int W = PG_RC_W;
    static constexpr bool kRagged = false, kRecInRows = false, kUsePow2 = true;

    __device__ static const uint4 *pow2(const Args &A) { return A.pow2; }
    __device__ static uint32_t rows_per_item(const Args &A) { return 4 * A.n + 11; }
    __device__ static uint32_t xo(const Args &A) { return A.witness_vars ? 0u : 1u; }  // variables the witness takes
    __device__ static uint32_t vars_per_item(const Args &A) { return 2 * A.n + 523 + xo(A); }

    __device__ static void fill_table(const Args &A, uint4 *table, uint32_t tid) {
        if (tid == T_QC_A || tid == T_QC_B) {
            FrVec t;
            t.f = tid == T_QC_A ? fr_sub(A.max_range, fr_one())  // range.rs:87
                                : fr_neg(A.min_range);           // range.rs:63
            table[2 * tid] = t.v[0];
            table[2 * tid + 1] = t.v[1];
        }
    }

    __device__ static void item(const Args &A, const EmitOut &O, uint64_t item, const uint4 *table, ItemRec &R) {
        FrVec x;
        x.v[0] = A.witness[item * 2];
        x.v[1] = A.witness[item * 2 + 1];
        R.x = x.f;
        // T = (max-1) - x  (range.rs:102)   |   T = x - min  (range.rs:69)
        bound_fetch_z(O, item, R.b[0]);  // (elements are numbered e-major, invert.hpp)
        bound_fetch_z(O, O.batch + item, R.b[1]);
        R.y[0] = bound_item(fr_sub(lds_fr(table, T_QC_A), x.f), A.n, R.b[0]);
        R.y[1] = bound_item(fr_add(x.f, lds_fr(table, T_QC_B)), A.n, R.b[1]);
        const uint64_t V = vars_per_item(A);
        if (A.result_vars) A.result_vars[item] = O.var_base + item * V + (V - 1);
    }

    __device__ static void selectors(const Args &A, const ItemRec &, uint32_t j, const uint4 *table, uint32_t h, uint4 out[5]) {
        const uint32_t L = 2 * A.n + 5;
        uint32_t id[5];
        if (j == 2 * L) {  // y1 * y2, range.rs:42
            id[0] = T_ONE; id[1] = T_ZERO; id[2] = T_ZERO; id[3] = T_NEG1; id[4] = T_ZERO;
        } else {
            const bool is_min = j >= L;
            bound_selector_ids(is_min ? j - L : j, A.n, is_min, id);
        }
        ids_to_values(id, table, h, out);
    }

    __device__ static void wires(const Args &A, const EmitOut &, const ItemRec &, uint64_t item, uint64_t vbase, uint32_t j,
                                 uint64_t out[3]) {
        const uint32_t n = A.n, L = 2 * n + 5, VB = n + 261, x0 = xo(A);
        uint32_t off[3];
        if (j == 2 * L) {
            off[0] = x0 + 260 + n;       // Y1
            off[1] = x0 + VB + 260 + n;  // Y2
            off[2] = x0 + 2 * VB;        // R
        } else {
            const bool is_min = j >= L;
            bound_wire_offsets(is_min ? j - L : j, n, is_min ? x0 + VB : x0, kWitnessWire, off);
        }
        const uint64_t xvar = A.witness_vars ? A.witness_vars[item] : vbase;
#pragma unroll
        for (int c = 0; c < 3; c++) out[c] = off[c] == kWitnessWire ? xvar : vbase + off[c];
    }

    __device__ static Fr var_value(const Args &A, const ItemRec &R, const uint4 *, uint32_t k) {
        const uint32_t n = A.n, VB = n + 261;
        if (!A.witness_vars) {
            if (k == 0) return R.x;
            k -= 1;
        }
        if (k == 2 * VB) return (R.y[0] & R.y[1]) ? fr_one() : fr_zero();  // range.rs:42
        uint32_t kk = k, blk = 0;
        if (kk >= VB) { kk -= VB; blk = 1; }
        return bound_var_value(R.b[blk], R.y[blk], kk, n);
    }
    // the region sweep of a witness refresh (emit.hpp): blocks, and where their variables lie in the item
    // ([x] | T bits accumulators U z y | T bits accumulators U z y | R; z is the pre-pass's)
    static constexpr bool kRegionVars = true;
    static constexpr uint32_t kBlocks = 2;
    __device__ static uint32_t region_n(const Args &A, const ItemRec &) { return A.n; }
    __device__ static const BoundRec &region_block(const ItemRec &R, uint32_t b) { return R.b[b]; }
    __device__ static uint32_t region_k(const Args &A, const ItemRec &, uint32_t b, uint32_t kk) { return xo(A) + b * (A.n + 261) + kk; }
};

// ---- max_bound: one public bound for the whole batch, or one bound per item ----
template <bool RAGGED>
struct MaxBoundGD {
    struct Args {
        Fr max_range;               // uniform: the bound (Montgomery form)
        uint32_t n;                 // uniform: ladder bits
        const uint4 *max_range_v;   // ragged: per-item bounds
        const uint32_t *num_bits_v; // ragged: per-item ladder bits (from the plan)
        const uint4 *witness;
        const uint64_t *witness_vars;  // NULL: items allocate their witness; else existing Variables (uniform only)
        uint64_t *result_vars;
        const uint4 *pow2;
    };
    static constexpr int kInv = 1;
    static constexpr bool kInvDense = true;
    static constexpr int kInvGroup = 4;  // elements a lane of the pre-pass fetches per round trip (inv_combine is register-hungry here)
    __device__ static uint4 *inv_slot(const Args &A, const EmitOut &O, uint64_t item, uint32_t) {
        const uint32_t n = RAGGED ? A.num_bits_v[item] : A.n;
        const uint32_t x0 = (!RAGGED && A.witness_vars) ? 0u : 1u;
        const uint64_t first = RAGGED ? O.var_off[item] : item * (uint64_t)(n + 261 + x0);
        return O.vars + 2 * (first + x0 + bound_z_offset(n));
    }
    // operands: the witness, the item's bound and ladder length (ragged: per item)
    __device__ static void inv_operands(const Args &A, const EmitOut &, uint64_t item, uint32_t, FrVec &p, FrVec &q, uint32_t &n) {
        p.v[0] = A.witness[item * 2];
        p.v[1] = A.witness[item * 2 + 1];
        n = A.n;
        q.f = A.max_range;
        if constexpr (RAGGED) {
            q.v[0] = A.max_range_v[item * 2];
            q.v[1] = A.max_range_v[item * 2 + 1];
            n = A.num_bits_v[item];
        }
    }
    __device__ static Fr inv_combine(const Args &, uint32_t, const Fr &x, const Fr &m, uint32_t n) {
        const Fr Tm = fr_sub(fr_sub(m, fr_one()), x);
        return bound_u(Tm, fr_from_mont(Tm), n);
    }
    struct alignas(16) ItemRec {
        Fr x;
        Fr qc;  // mont(max - 1) of this item (ragged: a per-item selector constant)
        BoundRec b;
        uint32_t y, n;
        uint32_t pad[2];
    };
#ifndef PG_MB_W
#define PG_MB_W 16
#endif
    static constexpr int W = PG_MB_W;
    static constexpr bool kRagged = RAGGED, kRecInRows = RAGGED, kUsePow2 = true;

    __device__ static const uint4 *pow2(const Args &A) { return A.pow2; }
    __device__ static uint32_t rows_per_item(const Args &A) { return 2 * A.n + 5; }
    __device__ static uint32_t xo(const Args &A) { return (!RAGGED && A.witness_vars) ? 0u : 1u; }
    __device__ static uint32_t vars_per_item(const Args &A) { return A.n + 261 + xo(A); }

    __device__ static void fill_table(const Args &A, uint4 *table, uint32_t tid) {
        if (tid == T_QC_A) {
            FrVec t;
            t.f = RAGGED ? fr_zero() : fr_sub(A.max_range, fr_one());  // range.rs:87
            table[2 * tid] = t.v[0];
            table[2 * tid + 1] = t.v[1];
        } else if (tid == T_QC_B) {
            table[2 * tid] = make_uint4(0, 0, 0, 0);
            table[2 * tid + 1] = make_uint4(0, 0, 0, 0);
        }
    }

    static constexpr uint32_t kUniformRows = 0, kUniformVars = 0;  // ragged: ladder lengths differ from item to item
    // ragged: what the rows of an item depend on -- its ladder length (from the plan) and q_c = mont(max_i - 1), both
    // functions of the PUBLIC bound alone
    __device__ static void item_rows(const Args &A, const EmitOut &, uint64_t item, const uint4 *, ItemRec &R) {
        FrVec m;
        m.v[0] = A.max_range_v[item * 2];
        m.v[1] = A.max_range_v[item * 2 + 1];
        R.qc = fr_sub(m.f, fr_one());  // range.rs:87
        R.n = A.num_bits_v[item];
    }

    __device__ static void item(const Args &A, const EmitOut &O, uint64_t item, const uint4 *table, ItemRec &R) {
        FrVec x;
        x.v[0] = A.witness[item * 2];
        x.v[1] = A.witness[item * 2 + 1];
        R.x = x.f;
        bound_fetch_z(O, item, R.b);
        uint32_t n = A.n;
        Fr qc;
        if constexpr (RAGGED) {  // R.qc / R.n are item_rows' (same lane, earlier): read, never rewritten
            qc = R.qc;
            n = R.n;
        } else {
            qc = lds_fr(table, T_QC_A);
            R.qc = qc;
            R.n = n;
        }
        R.y = bound_item(fr_sub(qc, x.f), n, R.b);  // range.rs:102
        if (A.result_vars) {
            const uint64_t first = RAGGED ? O.var_off[item] : item * (uint64_t)(n + 261 + xo(A));
            A.result_vars[item] = O.var_base + first + (n + 260 + xo(A));  // Y is the item's last variable
        }
    }

    __device__ static void selectors(const Args &A, const ItemRec &R, uint32_t j, const uint4 *table, uint32_t h, uint4 out[5]) {
        uint32_t id[5];
        bound_selector_ids(j, RAGGED ? R.n : A.n, false, id);
        ids_to_values(id, table, h, out);
        if constexpr (RAGGED) {
            if (j == 0) {  // the only data-dependent selector: q_c = mont(max_i - 1)
                FrVec q;
                q.f = R.qc;
                out[4] = q.v[h];
            }
        }
    }

    __device__ static void wires(const Args &A, const EmitOut &, const ItemRec &R, uint64_t item, uint64_t vbase, uint32_t j,
                                 uint64_t out[3]) {
        uint32_t off[3];
        bound_wire_offsets(j, RAGGED ? R.n : A.n, xo(A), kWitnessWire, off);
        const uint64_t xvar = xo(A) ? vbase : A.witness_vars[item];
#pragma unroll
        for (int c = 0; c < 3; c++) out[c] = off[c] == kWitnessWire ? xvar : vbase + off[c];
    }

    __device__ static bool is_inv_slot(const Args &A, const ItemRec &R, uint32_t k) {
        return k == xo(A) + bound_z_offset(RAGGED ? R.n : A.n);
    }
    __device__ static Fr var_value(const Args &A, const ItemRec &R, const uint4 *, uint32_t k) {
        if (xo(A)) {
            if (k == 0) return R.x;
            k -= 1;
        }
        return bound_var_value(R.b, R.y, k, RAGGED ? R.n : A.n);
    }
    static constexpr bool kRegionVars = true;
    static constexpr uint32_t kBlocks = 1;
    __device__ static uint32_t region_n(const Args &A, const ItemRec &R) { return RAGGED ? R.n : A.n; }
    __device__ static const BoundRec &region_block(const ItemRec &R, uint32_t) { return R.b; }
    __device__ static uint32_t region_k(const Args &A, const ItemRec &, uint32_t, uint32_t kk) { return xo(A) + kk; }
};

// ---- scalar_decomposition_gadget alone (range.rs:119-158): the bound block without its add row ------------------
// Private in the reference but exercised directly by its unit test (range.rs:205-233).  The witness is an existing
// AllocatedScalar; rows: A0 const, n x (boolean, ladder add), then maybe_equal(accumulator, witness): 2n+4 rows,
// n+260 variables [256 bits | A_0..A_n | u z y].  Implemented as the bound block shifted by one row / one variable:
// with the block base at -1 the block's "T" slot is exactly the sentinel kWitnessWire.
struct DecompositionGD {
    struct Args {
        uint32_t n;                    // num_bits, 0..256
        const uint4 *witness;          // assignments of the witnesses
        const uint64_t *witness_vars;  // their Variables
        uint64_t *result_vars;         // is_equal per item
        const uint4 *pow2;
    };
    static constexpr int kInv = 1;
    static constexpr bool kInvDense = true;
    static constexpr int kInvGroup = 4;  // elements a lane of the pre-pass fetches per round trip (inv_combine is register-hungry here)
    __device__ static void inv_operands(const Args &A, const EmitOut &, uint64_t item, uint32_t, FrVec &p, FrVec &q, uint32_t &) {
        p.v[0] = A.witness[item * 2];
        p.v[1] = A.witness[item * 2 + 1];
        q.f = p.f;
    }
    __device__ static Fr inv_combine(const Args &A, uint32_t, const Fr &x, const Fr &, uint32_t) {
        return bound_u(x, fr_from_mont(x), A.n);
    }
    // the block shifted by one variable: z sits at (259 + n) - 1
    __device__ static uint4 *inv_slot(const Args &A, const EmitOut &O, uint64_t item, uint32_t) {
        return O.vars + 2 * (item * (uint64_t)(A.n + 260) + bound_z_offset(A.n) - 1);
    }
    struct alignas(16) ItemRec {
        BoundRec b;
        uint32_t y, pad[3];
    };
    static constexpr int W = 32;
    static constexpr bool kRagged = false, kRecInRows = false, kUsePow2 = true;
    __device__ static const uint4 *pow2(const Args &A) { return A.pow2; }
    __device__ static uint32_t rows_per_item(const Args &A) { return 2 * A.n + 4; }
    __device__ static uint32_t vars_per_item(const Args &A) { return A.n + 260; }
    __device__ static void fill_table(const Args &, uint4 *table, uint32_t tid) {
        if (tid == T_QC_A || tid == T_QC_B) {
            table[2 * tid] = make_uint4(0, 0, 0, 0);
            table[2 * tid + 1] = make_uint4(0, 0, 0, 0);
        }
    }
    __device__ static void item(const Args &A, const EmitOut &O, uint64_t item, const uint4 *, ItemRec &R) {
        FrVec x;
        x.v[0] = A.witness[item * 2];
        x.v[1] = A.witness[item * 2 + 1];
        bound_fetch_z(O, item, R.b);
        R.y = bound_item(x.f, A.n, R.b);
        const uint64_t V = vars_per_item(A);
        if (A.result_vars) A.result_vars[item] = O.var_base + item * V + (V - 1);
    }
    __device__ static void selectors(const Args &A, const ItemRec &, uint32_t j, const uint4 *table, uint32_t h, uint4 out[5]) {
        uint32_t id[5];
        bound_selector_ids(j + 1, A.n, false, id);
        ids_to_values(id, table, h, out);
    }
    __device__ static void wires(const Args &A, const EmitOut &, const ItemRec &, uint64_t item, uint64_t vbase, uint32_t j,
                                 uint64_t out[3]) {
        uint32_t off[3];
        bound_wire_offsets(j + 1, A.n, kWitnessWire, kWitnessWire, off);
        const uint64_t xvar = A.witness_vars[item];
#pragma unroll
        for (int c = 0; c < 3; c++) out[c] = off[c] == kWitnessWire ? xvar : vbase + off[c];
    }
    __device__ static bool is_inv_slot(const Args &A, const ItemRec &, uint32_t k) { return k + 1 == bound_z_offset(A.n); }
    __device__ static Fr var_value(const Args &A, const ItemRec &R, const uint4 *, uint32_t k) {
        return bound_var_value(R.b, R.y, k + 1, A.n);
    }
    static constexpr bool kRegionVars = true;
    static constexpr uint32_t kBlocks = 1;
    __device__ static uint32_t region_n(const Args &A, const ItemRec &) { return A.n; }
    __device__ static const BoundRec &region_block(const ItemRec &R, uint32_t) { return R.b; }
    __device__ static uint32_t region_k(const Args &, const ItemRec &, uint32_t, uint32_t kk) { return kk - 1; }  // (no T: the block starts at its bits)
    // (num_bits may be 256 here, range.rs:134: 257 accumulators, one more than a wave-pass of four per lane holds -- the last goes with U z y)
};

// plan of a ragged max_bound batch: ladder bits and row/variable counts per item (range.rs:87-90)
__global__ __launch_bounds__(kThreads) void max_bound_plan_kernel(const uint4 *max_range, uint64_t batch, const uint4 *pow2,
                                                                 uint32_t *num_bits, uint32_t *rows, uint32_t *vars, const PlanScan P) {
    uint32_t r[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x * 4 + k;
        if (i < batch) {
            FrVec m, p;
            m.v[0] = max_range[i * 2];
            m.v[1] = max_range[i * 2 + 1];
            const Fr mm1 = fr_sub(m.f, fr_one());
            uint32_t nb = raw_bit_length(fr_from_mont(mm1));  // bits_count, range.rs:173-181
            if (nb < 1) nb = 1;
            // bits_count(BlsScalar::pow_of_2(nb)), range.rs:187-188 (nb <= 255)
            p.v[0] = pow2[nb * 2];
            p.v[1] = pow2[nb * 2 + 1];
            uint32_t n = raw_bit_length(fr_from_mont(p.f));
            if (n < 1) n = 1;
            num_bits[i] = n;
            r[k] = 2 * n + 5;
            v[k] = n + 262;
        }
    }
    plan_store(P, r, v, batch, rows, vars);  // the prefix sums, same launch
}

// engine table: mont(2^i) by repeated doubling (one thread; runs once per engine)
__global__ void pow2_table_kernel(uint4 *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    FrVec p;
    p.f = fr_one();
    for (int i = 0; i < 256; i++) {
        out[2 * i] = p.v[0];
        out[2 * i + 1] = p.v[1];
        p.f = fr_add(p.f, p.f);
    }
}

}  // namespace pg
