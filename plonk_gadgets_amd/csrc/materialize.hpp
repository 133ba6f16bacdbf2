// materialize.hpp -- SURVEY section 8f1 for the rows of BATCHED calls: wire-value columns without a gather from memory.
//
// pg_composer_materialize writes, per row, seven constant columns, w_4 and the three wire-VALUE columns out[i] =
// variables[w[i]].  The generic kernel (composer.hpp, materialize_kernel) fetches every value by its index: two dependent
// loads per value, 32 random bytes each -- its gathers ran at 4.6 TB/s beside constant columns at 6.9.  But the rows of a
// batched call (PermSeg: `items` items of L rows that created V Variables each, one after the other) reference almost
// nothing but their own item's Variables -- a run of V consecutive entries of the variable table.  So a workgroup takes a
// GROUP of consecutive items whose Variables fit its LDS window, reads that run LINEARLY (coalesced 16-byte loads, every
// byte of the table exactly once), and serves the rows' look-ups from LDS; the wire indices themselves are read linearly
// too (they are what makes the kernel gadget-agnostic: any batched append, uniform or ragged, present or future).  What
// is left for memory is three linear streams in and eleven out.  A reference outside the window (the allocated witness of
// an `_allocated_batch` call, zero_var) goes to memory as before.
#pragma once

#include "permutation.hpp"

namespace pg {

#ifndef PG_MAT_THREADS
#define PG_MAT_THREADS 512
#endif
#ifndef PG_MAT_UNROLL
#define PG_MAT_UNROLL 2
#endif
constexpr int kMatThreads = PG_MAT_THREADS;
constexpr uint32_t kMatWindowVars = 1040;  // 33 280 B of LDS: four workgroups per CU; range_check's 1034 Variables per item fit

// CLOSED: the segment's wires are known in closed form (PermSeg::wire_kind, a uniform ladder gadget): a row's three Variables are
// computed from its position in its item, and only a reference to a witness allocated elsewhere is read from the wire column
template <bool CLOSED>
__global__ __launch_bounds__(kMatThreads) void materialize_items_kernel(const ComposerCols C, const MaterializeOut M, const PermSeg S,
                                                                        uint32_t group, uint64_t zero_var) {
    __shared__ uint4 s_win[2 * kMatWindowVars];
    const uint32_t tid = threadIdx.x;
    FrVec one;
    one.f = fr_one();
    const uint4 v1 = (tid & 1) ? one.v[1] : one.v[0], v0 = make_uint4(0, 0, 0, 0);
    const uint64_t n_groups = (S.items + group - 1) / group;
    for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint64_t i0 = g * group, i1 = i0 + group < S.items ? i0 + group : S.items;
        const uint64_t r0 = S.gate_base + perm_rows_before(S, i0), r1 = S.gate_base + perm_rows_before(S, i1);
        const uint64_t w0 = S.var_base + perm_vars_before(S, i0);
        const uint32_t nv = (uint32_t)(S.var_base + perm_vars_before(S, i1) - w0);
        for (uint32_t u = tid; u < 2 * nv; u += kMatThreads) s_win[u] = C.vars[2 * w0 + u];
        __syncthreads();
        // The group writes WHOLE LINES of every output column: its range of 16-byte units is cut at multiples of 32 (= 16 rows:
        // four lines of a scalar column, one line of w_4) instead of at its own first and last row, except at the two ends of
        // the call.  A line that two workgroups fill at different times reaches memory in pieces, which costs about twelve
        // lines' worth -- with items of 1031 rows that would be two of every 258 lines.  The few rows
        // of the next group that come along find their Variables outside the window and fetch them from memory.
        const uint64_t ubeg = g == 0 ? 2 * r0 : (2 * r0 + 31) & ~31ull, uend = g + 1 == n_groups ? 2 * r1 : (2 * r1 + 31) & ~31ull;
        constexpr int U = PG_MAT_UNROLL;  // units per lane and pass: the index loads of all of them are in flight together
        for (uint64_t ub = (ubeg & ~7ull) + tid; ub < uend; ub += (uint64_t)U * kMatThreads) {
            uint64_t idx[U][3];
            uint4 got[U][3];
#pragma unroll
            for (int j = 0; j < U; j++) {
                const uint64_t u = ub + (uint64_t)j * kMatThreads;
                const uint64_t r = (u < ubeg ? ubeg : (u < uend ? u : uend - 1)) >> 1;
                if constexpr (CLOSED) {
                    // the row's item (of the group, or -- the rows that come along with the last line -- the one after it) and
                    // its place in it; Variables relative to the window's first
                    uint32_t rr = (uint32_t)(r - r0), vb = 0;
                    for (; rr >= S.L; rr -= S.L) vb += S.V;
                    uint32_t off[3];
                    seg_wire_offsets(S.wire_kind, S.wire_n, rr, off);
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        idx[j][k] = w0 + vb + off[k];
                        if (off[k] == kWitnessWire) idx[j][k] = M.val[k] ? C.w[k][r] : w0;  // (an `_allocated` call's witness)
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 3; k++) idx[j][k] = M.val[k] ? C.w[k][r] : w0;
                }
            }
#pragma unroll
            for (int j = 0; j < U; j++) {
                const uint32_t half = (uint32_t)((ub + (uint64_t)j * kMatThreads) & 1);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const uint64_t rel = idx[j][k] - w0;
                    got[j][k] = rel < nv ? s_win[2 * (uint32_t)rel + half] : C.vars[2 * idx[j][k] + half];
                }
            }
#pragma unroll
            for (int j = 0; j < U; j++) {
                const uint64_t u = ub + (uint64_t)j * kMatThreads;
                if (u >= ubeg && u < uend) {
#pragma unroll
                    for (int k = 0; k < 7; k++)
                        if (M.konst[k]) store16(M.konst[k] + u, k == 1 ? v1 : v0);
                    if (M.w_4 && !(u & 1)) M.w_4[u >> 1] = zero_var;
#pragma unroll
                    for (int k = 0; k < 3; k++)
                        if (M.val[k]) store16(M.val[k] + u, got[j][k]);
                }
            }
        }
        __syncthreads();  // the window is the next group's
    }
}

}  // namespace pg
