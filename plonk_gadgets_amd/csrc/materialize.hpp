// materialize.hpp -- SURVEY section 8f1 for the rows of BATCHED calls: wire-value columns without a gather from memory.
//
// pg_composer_materialize writes, per row, seven constant columns, w_4 and the three wire-VALUE columns out[i] =
// variables[w[i]].  The generic kernel (composer.hpp, materialize_kernel) fetches every value by its index: two dependent
// loads per value, 32 random bytes each -- its gathers ran at 4.6 TB/s beside constant columns at 6.9.  But the rows of a
// batched call (PermSeg: `items` items of L rows that created V Variables each, one after the other) reference almost
// nothing but their own item's Variables -- a run of V consecutive entries of the variable table.  So a workgroup takes a
// GROUP of consecutive items whose Variables fit its LDS window, reads that run LINEARLY (coalesced 16-byte loads, every
// byte of the table at most once -- the 256 bit Variables per bound block of the uniform ladder kinds not at all: they are made
// from the block's T, round 6), and serves the rows' look-ups from LDS.  Where the call's wires are known in closed form
// (PermSeg::wire_kind: the five ladder kinds, per-item bounds, the complete scalar mix) the rows' Variables are computed from
// their place in their item and nothing else is read: one linear stream in and eleven out (MAT_SELF).  Any other batched
// append, present or future, has its wire indices read linearly too, and a reference outside the window fetched from memory
// as before (MAT_READ_WIRES).
#pragma once

#include "permutation.hpp"

namespace pg {

#ifndef PG_MAT_THREADS
#define PG_MAT_THREADS 512
#endif
#ifndef PG_MAT_UNROLL
#define PG_MAT_UNROLL 2
#endif
#ifndef PG_MAT_LOADS
#define PG_MAT_LOADS 8  // (4, 8 or 16)
#endif
// kMatStoreThreads lanes write; ONE more wave only reads: it fetches the NEXT group's window while the others write the current
// group's rows.  A wave waits for a load with s_waitcnt vmcnt -- the counter its stores are counted on as well -- so a wave that
// both loads and stores drains its own store stream at every wait, and a pass that only MIGHT load (behind a branch no lane takes)
// still does: the compiler puts the wait where the paths meet.  That was the kernel up to round 5's first half: one drain per pass
// and two per group (without its reads: 12.4-12.8 ms per 270 M rows, with them 15.2-15.7; the reads' share of the traffic is 1.2).
// The loader's counter is its own.
constexpr int kMatStoreThreads = PG_MAT_THREADS, kMatThreads = kMatStoreThreads + 64;
constexpr uint32_t kMatWindowVars = 1040;  // two windows of 33 280 B: two workgroups per CU (fewer, fatter store streams: -3 %); range_check's 1034 Variables per item fit
constexpr uint32_t kMatTailRows = 16;      // rows of the NEXT group that come along with a group's last line of w_4 (at most 15)

// how the store waves learn a row's three Variables
enum : int {
    MAT_READ_WIRES = 0,  // from the wire columns (any batched call)
    MAT_SELF = 2,        // in closed form (PermSeg::wire_kind = KIND, compiled in: one instantiation per kind keeps the wire functions
                         // of the others -- and their registers -- out), and the store waves load NOTHING: the rows that come along
                         // from the next group and the ONE Variable per item that may come from elsewhere (the witness of the
                         // `_allocated` kinds and of scalar_decomposition) get their values from the loader; such groups have at most
                         // kMatWitItems items
};
constexpr uint32_t kMatWitItems = 4;  // (these kinds create >= 257 Variables per item: a window holds at most four)

// RAGGED (MAT_SELF, WIRES_MAX_BOUND only): per-item bounds -- rows and Variables by the call's prefix sums, the ladder length of an item
// from its row count (L = 2 n + 5).  The loader reads the group's prefix sums and leaves them in LDS for the store waves.
template <int MODE, uint32_t KIND, bool RAGGED = false>
__global__ __launch_bounds__(kMatThreads) void materialize_items_kernel(const ComposerCols C, const MaterializeOut M, const PermSeg S,
                                                                        uint32_t group, uint64_t zero_var) {
    constexpr bool CLOSED = MODE != MAT_READ_WIRES;
    constexpr bool kForeign = KIND == WIRES_RANGE_CHECK_ALLOCATED || KIND == WIRES_MAX_BOUND_ALLOCATED || KIND == WIRES_DECOMPOSITION;
    __shared__ uint4 s_win[2][2 * kMatWindowVars];
    __shared__ uint4 s_tail[2][MODE == MAT_SELF ? kMatTailRows * 6 : 1];
    __shared__ uint4 s_wit[2][MODE == MAT_SELF ? kMatWitItems * 2 : 1];  // the items' witnesses, when they are Variables from elsewhere
    // BITS (the uniform ladder kinds whose blocks start with T): half of an item's Variables are the 256 bits of canonical(T) per block
    // (range.rs:128-131), i.e. a function of ONE Variable the window holds anyway.  The loader leaves them in memory -- it copies the
    // runs between them and puts canonical(T) of every block beside the window -- and the store waves make a bit's assignment from that:
    // the call reads half of what it did (what its reads cost against the saturated store stream is what separates it from the ceiling).
    constexpr bool BITS = MODE == MAT_SELF && (KIND == WIRES_MAX_BOUND || (!RAGGED && (KIND == WIRES_RANGE_CHECK ||
                                                KIND == WIRES_RANGE_CHECK_ALLOCATED || KIND == WIRES_MAX_BOUND_ALLOCATED)));
    constexpr uint32_t kBlocks = (KIND == WIRES_RANGE_CHECK || KIND == WIRES_RANGE_CHECK_ALLOCATED) ? 2u : 1u;
    constexpr uint32_t kX0 = (KIND == WIRES_RANGE_CHECK || KIND == WIRES_MAX_BOUND) ? 1u : 0u;  // the item's own witness Variable
    __shared__ uint4 s_tc[2][BITS ? kMatWitItems * 2 * 2 : 1];  // canonical T of block b of item i of the group: [(i * kBlocks + b) * 2 + half]
    // RAGGED: rows before item k of the group, k = 0 .. items, then Variables before it (max_bound: at most kMatWitItems items of >= 264
    // Variables; the fused mix with failing items: up to 80 items of 13 or 15 -- a power of two of entries, for the store waves' search)
    constexpr uint32_t kOff = KIND == WIRES_MIX ? 128 : kMatWitItems + 1;
    __shared__ uint32_t s_off[2][RAGGED ? 2 * kOff : 1];
    __shared__ uint64_t s_base[2][RAGGED ? 2 : 1];                         // RAGGED: rows / Variables of the call before the group
    static_assert(!RAGGED || (MODE == MAT_SELF && (KIND == WIRES_MAX_BOUND || KIND == WIRES_MIX)),
                  "ragged closed forms: max_bound with per-item bounds, the fused mix with items that stopped at their error");
    const uint32_t tid = threadIdx.x;
    const bool loader = tid >= (uint32_t)kMatStoreThreads;
    FrVec one;
    one.f = fr_one();
    const uint4 v1 = (tid & 1) ? one.v[1] : one.v[0], v0 = make_uint4(0, 0, 0, 0);
    const uint64_t n_groups = (S.items + group - 1) / group;
    struct Group { uint64_t r0, r1, w0, ubeg, uend; uint32_t nv, items; };
    // where an item's witness stands when it is a Variable from elsewhere (kWitnessWire: this kind allocates it itself)
    const uint32_t foreign_row = KIND == WIRES_DECOMPOSITION ? 2 * S.wire_n + 1
                                 : KIND == WIRES_RANGE_CHECK_ALLOCATED || KIND == WIRES_MAX_BOUND_ALLOCATED ? 0u : kWitnessWire;
    const uint32_t foreign_wire = KIND == WIRES_DECOMPOSITION ? 1u : 0u;
    const uint32_t recip_L = (uint32_t)(((1ull << 32) + S.L - 1) / (S.L ? S.L : 1));
    // lds_buf >= 0: a store wave asks (RAGGED: the loader has left the group's prefix sums in that buffer -- a load here would drain the
    // wave's stores once per group); < 0: the loader asks
    auto group_of = [&](uint64_t g, int lds_buf) {
        Group G;
        const uint64_t i0 = g * group, i1 = i0 + group < S.items ? i0 + group : S.items;
        G.items = (uint32_t)(i1 - i0);
        if constexpr (RAGGED) {
            if (lds_buf >= 0) {
                G.r0 = S.gate_base + s_base[lds_buf][0];
                G.r1 = G.r0 + s_off[lds_buf][G.items];
                G.w0 = S.var_base + s_base[lds_buf][1];
                G.nv = s_off[lds_buf][kOff + G.items];
            } else {
                G.r0 = S.gate_base + S.row_off[i0];
                G.r1 = S.gate_base + S.row_off[i1];
                G.w0 = S.var_base + S.var_off[i0];
                G.nv = (uint32_t)(S.var_off[i1] - S.var_off[i0]);
            }
        } else if constexpr (CLOSED) {  // (uniform items: no prefix sums to fetch)
            G.r0 = S.gate_base + i0 * S.L;
            G.r1 = S.gate_base + i1 * S.L;
            G.w0 = S.var_base + i0 * S.V;
            G.nv = (uint32_t)(i1 - i0) * S.V;
        } else {
            G.r0 = S.gate_base + perm_rows_before(S, i0);
            G.r1 = S.gate_base + perm_rows_before(S, i1);
            G.w0 = S.var_base + perm_vars_before(S, i0);
            G.nv = (uint32_t)(S.var_base + perm_vars_before(S, i1) - G.w0);
        }
        // The group writes WHOLE LINES of every output column: its range of 16-byte units is cut at multiples of 32 (= 16 rows:
        // four lines of a scalar column, one line of w_4) instead of at its own first and last row, except at the two ends of
        // the call.  A line that two workgroups fill at different times reaches memory in pieces, which costs about twelve
        // lines' worth -- with items of 1031 rows that would be two of every 258 lines.
        const uint64_t last = 2 * S.gate_end;
        G.ubeg = g == 0 ? 2 * G.r0 : (2 * G.r0 + 31) & ~31ull;
        G.uend = g + 1 == n_groups ? 2 * G.r1 : (2 * G.r1 + 31) & ~31ull;
        G.ubeg = G.ubeg < last ? G.ubeg : last;  // (a last group of fewer than 16 rows: the call's own rows, no others)
        G.uend = G.uend < last ? G.uend : last;
        return G;
    };
    // the loader's part of group g: the Variables its items created -- a run of the table, read linearly -- and (MAT_SELF) the
    // values of the rows that come along from the items after it
    auto fetch = [&](uint64_t g, uint32_t buf) {
        const Group G = group_of(g, -1);
        const uint32_t units = 2 * G.nv, lane = tid - kMatStoreThreads;
        const uint4 *src = C.vars + 2 * G.w0;
        uint4 *win = s_win[buf];
        // PG_MAT_LOADS loads in flight per lane (named registers: as an array this staging area went through scratch memory in two of
        // the three instantiations)
        // (BITS: the runs between the items' bit Variables, one after the other -- run r of item i: [rb, re) in units of the window)
        constexpr uint32_t kRuns = BITS ? kBlocks + 1 : 1;
        const uint32_t n_runs = BITS ? G.items * kRuns : 1;
        const uint32_t VBk = S.wire_n + 261;
#pragma unroll 1
        for (uint32_t run = 0; run < n_runs; run++) {
        uint32_t rb = 0, re = units;
        if constexpr (BITS) {
            const uint32_t it = run / kRuns, r = run - it * kRuns;
            uint32_t vb = it * S.V, Vi = S.V;
            if constexpr (RAGGED) {  // (max_bound with a bound per item: the item's Variables from the call's prefix sums)
                const uint64_t i0 = g * group;
                vb = (uint32_t)(S.var_off[i0 + it] - S.var_off[i0]);
                Vi = (uint32_t)(S.var_off[i0 + it + 1] - S.var_off[i0 + it]);
            }
            // r = 0: [x] T_0 | r = 1: A.. U z y of block 0 (and T_1) | r = 2: A.. U z y of block 1, R
            rb = 2 * (vb + (r == 0 ? 0 : kX0 + (r - 1) * VBk + 257));
            re = 2 * (vb + (r == kBlocks ? Vi : kX0 + r * VBk + 1));
        }
        uint32_t u = rb + lane;
        const uint32_t units = re;
        for (; u + (PG_MAT_LOADS - 1) * 64 < units; u += PG_MAT_LOADS * 64) {
            const uint4 *q = src + u;
            uint4 *d = win + u;
#if PG_MAT_LOADS >= 8
            const uint4 t0 = q[0], t1 = q[64], t2 = q[128], t3 = q[192], t4 = q[256], t5 = q[320], t6 = q[384], t7 = q[448];
#if PG_MAT_LOADS >= 16
            const uint4 t8 = q[512], t9 = q[576], ta = q[640], tb = q[704], tc = q[768], td = q[832], te = q[896], tf = q[960];
#endif
            d[0] = t0; d[64] = t1; d[128] = t2; d[192] = t3; d[256] = t4; d[320] = t5; d[384] = t6; d[448] = t7;
#if PG_MAT_LOADS >= 16
            d[512] = t8; d[576] = t9; d[640] = ta; d[704] = tb; d[768] = tc; d[832] = td; d[896] = te; d[960] = tf;
#endif
#else
            const uint4 t0 = q[0], t1 = q[64], t2 = q[128], t3 = q[192];
            d[0] = t0; d[64] = t1; d[128] = t2; d[192] = t3;
#endif
        }
        for (; u < units; u += 64) win[u] = src[u];
        }
        if constexpr (BITS) {  // canonical(T) of every block of the group, from the T the window has just received (this wave's own writes)
            if (lane < G.items * kBlocks) {
                const uint32_t it = lane / kBlocks, b = lane - it * kBlocks;
                uint32_t vb = it * S.V;
                if constexpr (RAGGED) vb = (uint32_t)(S.var_off[g * group + it] - S.var_off[g * group]);
                const uint4 *tp = win + 2 * (vb + kX0 + b * VBk);
                FrVec t;
                t.v[0] = tp[0];
                t.v[1] = tp[1];
                t.f = fr_from_mont(t.f);  // scalar_to_bits -> to_bytes, range.rs:163
                s_tc[buf][2 * lane] = t.v[0];
                s_tc[buf][2 * lane + 1] = t.v[1];
            }
        }
        if constexpr (MODE == MAT_SELF) {
            const uint32_t rows = (uint32_t)((G.uend - 2 * G.r1) >> 1);  // (2 r1 <= uend: whole rows, at most 15)
            if (lane < 3 * rows) {
                uint32_t rr = lane / 3, vb = 0, n = S.wire_n;
                const uint32_t k = lane - 3 * rr;
                if constexpr (RAGGED) {  // the item of that row, from the prefix sums behind the group's last item
                    uint64_t it = g * group + G.items;
                    const uint64_t rows0 = S.row_off[it], vars0 = S.var_off[it];
                    while (S.row_off[it + 1] - rows0 <= rr) it++;  // (the row lies inside the call: uend stops at its last row)
                    const uint64_t first = S.row_off[it];
                    n = KIND == WIRES_MIX ? (S.row_off[it + 1] - first == 10 ? 0u : 1u)   // (seg_wire_offsets: the short item's table)
                                          : (uint32_t)(S.row_off[it + 1] - first - 5) >> 1;
                    rr -= (uint32_t)(first - rows0);
                    vb = (uint32_t)(S.var_off[it] - vars0);
                } else {
                    for (; rr >= S.L; rr -= S.L) vb += S.V;
                }
                uint32_t off[3];
                seg_wire_offsets(KIND, n, rr, off, S.tail);
                const uint32_t o = k == 0 ? off[0] : k == 1 ? off[1] : off[2];
                const uint64_t var = o == kWitnessWire ? C.w[k][G.r1 + lane / 3] : o == kZeroWire ? zero_var : G.w0 + G.nv + vb + o;
                s_tail[buf][2 * lane] = C.vars[2 * var];
                s_tail[buf][2 * lane + 1] = C.vars[2 * var + 1];
            }
            if constexpr (RAGGED) {  // the group's prefix sums for the store waves (entries behind the last item: never reached)
                const uint64_t i0 = g * group;
                for (uint32_t e = lane; e < kOff; e += 64) {
                    s_off[buf][e] = e <= G.items ? (uint32_t)(S.row_off[i0 + e] - S.row_off[i0]) : 0xffffffffu;
                    s_off[buf][kOff + e] = e <= G.items ? (uint32_t)(S.var_off[i0 + e] - S.var_off[i0]) : 0u;
                }
                if (lane == 47) {
                    s_base[buf][0] = S.row_off[i0];
                    s_base[buf][1] = S.var_off[i0];
                }
            }
            // lanes 48 ..: the witness of item (lane - 48) of the group, read from the row and wire that hold it
            if (foreign_row != kWitnessWire && lane >= 48 && lane - 48 < G.items) {
                const uint64_t var = C.w[foreign_wire][G.r0 + (uint64_t)(lane - 48) * S.L + foreign_row];
                s_wit[buf][2 * (lane - 48)] = C.vars[2 * var];
                s_wit[buf][2 * (lane - 48) + 1] = C.vars[2 * var + 1];
            }
        }
    };
    uint64_t g = blockIdx.x;
    if (loader && g < n_groups) fetch(g, 0);
    __syncthreads();
    for (uint32_t buf = 0; g < n_groups; g += gridDim.x, buf ^= 1) {
        if (loader) {
            if (g + gridDim.x < n_groups) fetch(g + gridDim.x, buf ^ 1);
            lds_barrier();
            continue;
        }
        const uint4 *win = s_win[buf];
        const Group G = group_of(g, (int)buf);
        const uint64_t r0 = G.r0, w0 = G.w0, ubeg = G.ubeg, uend = G.uend;
        const uint32_t nv = G.nv;
        constexpr int U = PG_MAT_UNROLL;  // units per lane and pass
        for (uint64_t ub = (ubeg & ~7ull) + tid; ub < uend; ub += (uint64_t)U * kMatStoreThreads) {
            uint64_t idx[U][3];
            uint4 got[U][3];
#pragma unroll
            for (int j = 0; j < U; j++) {
                const uint64_t u = ub + (uint64_t)j * kMatStoreThreads;
                const uint64_t r = (u < ubeg ? ubeg : (u < uend ? u : uend - 1)) >> 1;
                const uint32_t half = (uint32_t)(u & 1);
                if constexpr (CLOSED) {
                    // the row's item (of the group, or -- the rows that come along with the last line -- the one after it) and
                    // its place in it; Variables relative to the window's first
                    // (floor(x / L) = umulhi(x, ceil(2^32 / L)), exact while x * L < 2^32: a group has a few thousand rows at most)
                    uint32_t rr = (uint32_t)(r - r0), it, vb, n = S.wire_n;
                    if constexpr (RAGGED) {  // compare with the prefix sums the loader left
                        it = 0;
                        if constexpr (KIND == WIRES_MIX) {  // (up to 80 items: the last entry that is <= rr; entries past the group hold ~0)
#pragma unroll
                            for (uint32_t step = kOff / 2; step; step >>= 1)
                                if (s_off[buf][it + step] <= rr) it += step;
                        } else {                             // (at most kMatWitItems items)
#pragma unroll
                            for (uint32_t e = 1; e < kMatWitItems; e++) it += rr >= s_off[buf][e] ? 1u : 0u;
                        }
                        const uint32_t first = s_off[buf][it];
                        n = KIND == WIRES_MIX ? (s_off[buf][it + 1] - first == 10 ? 0u : 1u) : (s_off[buf][it + 1] - first - 5) >> 1;
                        vb = s_off[buf][kOff + it];
                        rr -= first;
                    } else {
                        it = __umulhi(rr, recip_L);
                        vb = it * S.V;
                        rr -= it * S.L;
                    }
                    uint32_t off[3];
                    seg_wire_offsets(KIND, n, rr, off, S.tail);
                    {
                        const bool along = r >= G.r1;
                        const uint32_t t = along ? (uint32_t)(r - G.r1) * 6 + half : 0;
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            // ONE LDS read per value: from the window, the witness table or the side table of the rows that came along
                            const uint32_t rel = vb + off[k];
                            const uint4 *src = win + 2 * (rel < nv ? rel : 0) + half;
                            if constexpr (kForeign)
                                if (off[k] == kWitnessWire) src = s_wit[buf] + 2 * (it < kMatWitItems ? it : 0) + half;
                            if (along) src = s_tail[buf] + t + 2 * k;
                            got[j][k] = *src;
                            if constexpr (BITS) {  // a bit of canonical(T): mont(1) or mont(0), from the block's T (range.rs:128-131)
                                uint32_t kk = off[k] - kX0, blk = 0;
                                if (kBlocks == 2 && kk >= S.wire_n + 261 && off[k] != kWitnessWire) { kk -= S.wire_n + 261; blk = 1; }
                                if (kk - 1 < 256u && off[k] != kWitnessWire && !along) {
                                    const uint32_t bit = kk - 1, w = bit >> 5;
                                    const uint4 q = s_tc[buf][2 * ((it < kMatWitItems ? it : 0) * kBlocks + blk) + (w >> 2)];
                                    const uint32_t word = (w & 3) == 0 ? q.x : (w & 3) == 1 ? q.y : (w & 3) == 2 ? q.z : q.w;
                                    got[j][k] = (word >> (bit & 31)) & 1u ? v1 : v0;
                                }
                            }
                            if constexpr (KIND == WIRES_MIX)
                                if (off[k] == kZeroWire && !along) got[j][k] = make_uint4(0, 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 3; k++) idx[j][k] = M.val[k] ? C.w[k][r] : w0;
                }
            }
            if constexpr (MODE != MAT_SELF) {
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const uint32_t half = (uint32_t)((ub + (uint64_t)j * kMatStoreThreads) & 1);
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const uint64_t rel = idx[j][k] - w0;
                        got[j][k] = rel < nv ? win[2 * (uint32_t)rel + half] : C.vars[2 * idx[j][k] + half];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < U; j++) {
                const uint64_t u = ub + (uint64_t)j * kMatStoreThreads;
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
        lds_barrier();  // this window is read, the next one written: they swap
    }
}

}  // namespace pg
