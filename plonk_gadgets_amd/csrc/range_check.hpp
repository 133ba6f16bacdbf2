// range_check.hpp -- gfx950 kernel for the batched range_check gadget.
//
// For every witness x (one item), in order, the kernel materialises exactly
// what this loop appends to a dusk-plonk StandardComposer:
//     w = AllocatedScalar::allocate(composer, x)          /root/reference/src/allocated_scalar.rs:27
//     range_check(composer, min, max, w)                  /root/reference/src/range.rs:27-43
// i.e. max_bound (range.rs:82-113) -> min_bound (:53-76) -> mul (:42), each
// bound being one add row + scalar_decomposition_gadget (:119-158) which ends
// in maybe_equal (scalar.rs:105-140).
//
// Per item, with n = ladder bits, L = 2n+5 rows and VB = n+261 variables per
// bound block:
//   rows   G = 2L+1:  [max block | min block | final mul]
//     block row 0        bound add   (x, x, T ; 0, -1|+1, 0, -1, mont(max-1) | mont(-min))
//     block row 1        A0 const    (A0,A0,A0 ; 0, 1, 0, 0, 0)
//     block row 2+2i     boolean     (b_i,b_i,b_i ; 1, 0, 0, -1, 0)
//     block row 3+2i     ladder add  (b_i, A_i, A_{i+1} ; 0, mont(2^i), 1, -1, 0)
//     block row 2n+2     u = A_n - T (A_n, T, U ; 0, 1, -1, -1, 0)
//     block row 2n+3     y = 1 - u z (Z, U, Y ; -1, 0, 0, -1, 1)
//     block row 2n+4     y u = 0     (Y, U, U ; 1, 0, 0, 0, 0)
//     row 2L             y1 y2       (Y1, Y2, R ; 1, 0, 0, -1, 0)
//   variables V = 2VB+2:  [x | max block | min block | R]
//     block var 0 = T, 1..256 = bits of canonical(T), 257+i = A_i (i = 0..n),
//     258+n = U, 259+n = Z, 260+n = Y
//
// Data layout in HBM (struct-of-arrays, the composer's own columns): five
// selector columns of 32-byte scalars, three wire columns of 8-byte Variable
// indices, one variable table of 32-byte scalars; items are laid out one
// after the other (witness-major), so every column of a tile of W consecutive
// items is ONE contiguous byte range.  The kernel is a pure streaming writer:
// 184 B per row + 32 B per variable, 32 B read per item.
//
// Mapping: a 256-thread workgroup owns a tile of W consecutive items and
// sweeps each column's contiguous range with 16-byte-per-lane stores (lane i
// at base + 16 i: 1 KiB per wave instruction).  Selectors come from a 264-entry
// constant table in LDS (0, 1, -1, q_c constants, mont(2^i)); wire indices are
// affine in (item, row); the only field arithmetic is per item (canonical
// form of the two differences, the rare inversion) and one Montgomery
// multiplication per accumulator variable A_i = mont(T mod 2^i).
#pragma once

#include "fr.hpp"

namespace pg {

constexpr int kThreads = 256;
constexpr int kTableEntries = 8 + 256;
enum : uint32_t { T_ZERO = 0, T_ONE = 1, T_NEG1 = 2, T_QC_MAX = 3, T_QC_MIN = 4, T_POW = 8 };

struct RangeCheckArgs {
    Fr min_range, max_range;  // Montgomery form (public inputs)
    uint32_t n;               // ladder bits, 2..255
    uint32_t tiles;
    uint64_t batch, gate_base, var_base;
    uint4 *q[5];
    uint64_t *w[3];
    uint4 *vars;
    const uint4 *witness;
    uint64_t *result_vars;
    const uint4 *pow2;  // engine table: mont(2^i), i < 256
};

union FrVec {
    Fr f;
    uint4 v[2];
};

// per-item record kept in LDS between the item phase and the variable sweep
struct alignas(16) BoundRec {
    Fr Tm;  // T in Montgomery form (the block's first variable)
    Fr Tc;  // canonical integer of T
    Fr U;   // A_n - T
    Fr Z;   // U^-1 or 0
};
struct alignas(16) ItemRec {
    Fr x;
    BoundRec b[2];
    uint32_t y[2];
    uint32_t pad[2];
};

__device__ __forceinline__ void store16(uint4 *p, uint4 v) {
#if defined(PG_NT_STORES)
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// ---- row / variable classification -------------------------------------

// selector table ids (q_m,q_l,q_r,q_o,q_c) of item-row j
__device__ __forceinline__ void selector_ids(uint32_t j, uint32_t n, uint32_t L, uint32_t id[5]) {
    uint32_t blk = 0, jj = j;
    if (jj >= L) { jj -= L; blk = 1; }
    uint32_t qm = T_ZERO, ql = T_ZERO, qr = T_ZERO, qo = T_NEG1, qc = T_ZERO;
    if (j == 2 * L) {  // y1*y2
        qm = T_ONE;
    } else if (jj >= 2 && jj < 2 * n + 2) {
        if (jj & 1) {  // ladder add, i = (jj-3)/2
            ql = T_POW + ((jj - 3) >> 1);
            qr = T_ONE;
        } else {  // boolean
            qm = T_ONE;
        }
    } else if (jj == 0) {
        ql = blk ? T_ONE : T_NEG1;
        qc = blk ? T_QC_MIN : T_QC_MAX;
    } else if (jj == 1) {
        ql = T_ONE;
        qo = T_ZERO;
    } else if (jj == 2 * n + 2) {
        ql = T_ONE;
        qr = T_NEG1;
    } else if (jj == 2 * n + 3) {
        qm = T_NEG1;
        qc = T_ONE;
    } else {  // jj == 2n+4
        qm = T_ONE;
        qo = T_ZERO;
    }
    id[0] = qm; id[1] = ql; id[2] = qr; id[3] = qo; id[4] = qc;
}

// wire variable offsets (relative to the item's first variable) of item-row j
__device__ __forceinline__ void wire_offsets(uint32_t j, uint32_t n, uint32_t L, uint32_t VB, uint32_t off[3]) {
    uint32_t blk = 0, jj = j;
    if (jj >= L) { jj -= L; blk = 1; }
    const uint32_t vb = 1 + blk * VB;  // block's first variable (T)
    uint32_t a, b, c;
    if (j == 2 * L) {
        a = 1 + 260 + n;       // Y1
        b = 1 + VB + 260 + n;  // Y2
        c = 1 + 2 * VB;        // R
    } else if (jj >= 2 && jj < 2 * n + 2) {
        uint32_t i = (jj - 2) >> 1;
        a = vb + 1 + i;  // b_i
        if (jj & 1) {
            b = vb + 257 + i;
            c = vb + 258 + i;
        } else {
            b = a;
            c = a;
        }
    } else if (jj == 0) {
        a = 0; b = 0; c = vb;
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

// ---- the kernel ----------------------------------------------------------

template <int W>
__global__ __launch_bounds__(kThreads) void range_check_kernel(const RangeCheckArgs A) {
    __shared__ uint4 s_table[kTableEntries * 2];
    __shared__ ItemRec s_item[W];

    const uint32_t tid = threadIdx.x;
    const uint32_t n = A.n, L = 2 * n + 5, G = 2 * L + 1, VB = n + 261, V = 2 * VB + 2;

    // constant table -> LDS (once per workgroup)
    for (uint32_t e = tid; e < kTableEntries; e += kThreads) {
        FrVec t;
        t.f = fr_zero();
        if (e == T_ONE) t.f = fr_one();
        else if (e == T_NEG1) t.f = fr_neg_one();
        else if (e == T_QC_MAX) t.f = fr_sub(A.max_range, fr_one());  // range.rs:87
        else if (e == T_QC_MIN) t.f = fr_neg(A.min_range);            // range.rs:63
        else if (e >= T_POW) { t.v[0] = A.pow2[(e - T_POW) * 2]; t.v[1] = A.pow2[(e - T_POW) * 2 + 1]; }
        s_table[2 * e] = t.v[0];
        s_table[2 * e + 1] = t.v[1];
    }
    __syncthreads();

    for (uint32_t tile = blockIdx.x; tile < A.tiles; tile += gridDim.x) {
        const uint64_t w0 = (uint64_t)tile * W;
        const uint32_t Wt = (uint32_t)((A.batch - w0) < (uint64_t)W ? (A.batch - w0) : (uint64_t)W);
        const uint64_t row0 = w0 * G;  // first row of the tile, relative to gate_base
        const uint64_t var0 = w0 * V;  // first variable of the tile, relative to var_base

        // ---- item phase: one lane per item ------------------------------
        if (tid < Wt) {
            FrVec x;
            x.v[0] = A.witness[(w0 + tid) * 2];
            x.v[1] = A.witness[(w0 + tid) * 2 + 1];
            ItemRec &R = s_item[tid];
            R.x = x.f;
            FrVec qc0, qc1;
            qc0.v[0] = s_table[2 * T_QC_MAX]; qc0.v[1] = s_table[2 * T_QC_MAX + 1];
            qc1.v[0] = s_table[2 * T_QC_MIN]; qc1.v[1] = s_table[2 * T_QC_MIN + 1];
#pragma unroll 1
            for (int blk = 0; blk < 2; blk++) {
                // T = (max-1) - x  (range.rs:102)   |   T = x - min  (range.rs:69)
                Fr Tm = blk == 0 ? fr_sub(qc0.f, x.f) : fr_add(x.f, qc1.f);
                Fr Tc = fr_from_mont(Tm);
                bool hi = raw_has_high_bits(Tc, n);
                // A_n = mont(T mod 2^n); equals T when T fits n bits
                Fr U = fr_zero(), Z = fr_zero();
                if (hi) {
                    U = fr_sub(fr_to_mont(raw_low_bits(Tc, n)), Tm);  // scalar.rs:121
                    Z = fr_invert_or_zero(U);                         // scalar.rs:122
                }
                R.b[blk].Tm = Tm;
                R.b[blk].Tc = Tc;
                R.b[blk].U = U;
                R.b[blk].Z = Z;
                R.y[blk] = hi ? 0u : 1u;
            }
            if (A.result_vars) A.result_vars[w0 + tid] = A.var_base + (w0 + tid) * V + (V - 1);
        }

        // ---- selector sweep: 16 B per lane, 128 rows x 5 columns per pass
        {
            const uint32_t total = Wt * G * 2;  // half-scalars
            const uint32_t h = tid & 1;
            uint32_t j = (tid >> 1) % G;
            for (uint32_t idx = tid; idx < total; idx += kThreads) {
                uint32_t id[5];
                selector_ids(j, n, L, id);
#pragma unroll
                for (int c = 0; c < 5; c++) store16(A.q[c] + (row0 * 2 + idx), s_table[2 * id[c] + h]);
                j += kThreads / 2;
                while (j >= G) j -= G;
            }
        }

        // ---- wire sweep: two rows (16 B) per lane per column ------------
#pragma unroll
        for (int c = 0; c < 3; c++) {
            uint64_t *col = A.w[c] + row0;
            // rows are paired so that each pair starts 16-byte aligned
            const uint32_t shift = (uint32_t)((reinterpret_cast<uintptr_t>(col) >> 3) & 1);
            const uint32_t rows = Wt * G;
            const uint32_t pairs = (rows + shift + 1) >> 1;
            for (uint32_t p = tid; p < pairs; p += kThreads) {
                int64_t r0 = (int64_t)2 * p - shift;
                uint64_t val[2];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    int64_t r = r0 + k;
                    uint32_t rr = r < 0 ? 0u : (uint32_t)r;
                    uint32_t wl = rr / G, j = rr - wl * G;
                    uint32_t off[3];
                    wire_offsets(j, n, L, VB, off);
                    val[k] = A.var_base + (w0 + wl) * V + off[c];
                }
                if (r0 >= 0 && r0 + 1 < (int64_t)rows) {
                    store16(reinterpret_cast<uint4 *>(col + r0),
                            make_uint4((uint32_t)val[0], (uint32_t)(val[0] >> 32), (uint32_t)val[1],
                                       (uint32_t)(val[1] >> 32)));
                } else {
                    if (r0 >= 0) col[r0] = val[0];
                    if (r0 + 1 < (int64_t)rows) col[r0 + 1] = val[1];
                }
            }
        }

        __syncthreads();  // item records visible

        // ---- variable sweep: one scalar (2 x 16 B) per lane -------------
        {
            const uint32_t total = Wt * V;
            uint32_t wl = tid / V, k = tid - wl * V;
            for (uint32_t s = tid; s < total; s += kThreads) {
                const ItemRec &R = s_item[wl];
                FrVec val;
                val.f = fr_zero();
                if (k == 0) {
                    val.f = R.x;
                } else if (k == V - 1) {
                    if (R.y[0] & R.y[1]) val.f = fr_one();  // range.rs:42
                } else {
                    uint32_t kk = k - 1, blk = 0;
                    if (kk >= VB) { kk -= VB; blk = 1; }
                    const BoundRec &B = R.b[blk];
                    if (kk == 0) {
                        val.f = B.Tm;
                    } else if (kk <= 256) {  // bit variables, range.rs:128-131
                        if (raw_bit(B.Tc, kk - 1)) val.f = fr_one();
                    } else if (kk <= 257 + n) {  // A_i, i = kk-257 (A_0 = 0), range.rs:152
                        uint32_t i = kk - 257;
                        if (i) val.f = fr_to_mont(raw_low_bits(B.Tc, i));
                    } else if (kk == 258 + n) {
                        val.f = B.U;
                    } else if (kk == 259 + n) {
                        val.f = B.Z;
                    } else {
                        if (R.y[blk]) val.f = fr_one();  // y = 1 - u z, scalar.rs:126
                    }
                }
                uint4 *dst = A.vars + (var0 + s) * 2;
                store16(dst, val.v[0]);
                store16(dst + 1, val.v[1]);
                k += kThreads;
                if (k >= V) { k -= V; wl++; }
            }
        }
        __syncthreads();  // records are rewritten by the next tile
    }
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
