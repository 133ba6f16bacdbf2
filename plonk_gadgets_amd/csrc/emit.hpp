// emit.hpp -- the streaming writer every batched gadget shares (gfx950).
//
// A batched gadget call appends, item after item, exactly the rows and
// variables the reference's sequential composer calls would append
// (witness-major order).  Because items are laid out one after the other,
// every column of a TILE of consecutive items is one contiguous byte range.
// A 256-thread workgroup owns a tile and
//   1. runs the gadget's per-item arithmetic with one lane per item, leaving
//      a small record per item in LDS,
//   2. sweeps the five selector columns (16 B per lane: lanes 2k, 2k+1 store
//      the two halves of row k's scalar, 1 KiB per wave store),
//   3. sweeps the three wire columns (two rows = 16 B per lane),
//   4. sweeps the variable table (one 32-byte scalar per lane).
// Steps 2-4 are pure streaming stores: the kernel's cost is the HBM write of
// 184 B per row + 32 B per variable.  The gadget itself is a policy class GD
// (see range_gadgets.hpp / scalar_gadgets.hpp) that answers, for item-row j or
// item-variable k, "which selector constants / which Variables / which value".
//
// Ragged batches (rows per item depend on public per-item data) pass
// exclusive prefix sums of rows and variables per item; uniform batches use
// the closed form item * G.
#pragma once

#include <type_traits>

#include "fr.hpp"

namespace pg {

constexpr int kThreads = 256;

union FrVec {
    Fr f;
    uint4 v[2];
};

struct EmitOut {
    uint4 *q[5];
    uint64_t *w[3];
    uint4 *vars;
    uint64_t gate_base, var_base;  // numbering of the call's first row / first variable
    uint64_t zero_var;             // composer.zero_var (fourth wire / assert_equal's output wire)
    const uint64_t *row_off;       // ragged only: [batch+1] exclusive prefix sums, relative to the call
    const uint64_t *var_off;
    uint64_t batch;
    uint32_t tiles;
    // Rows from one item's first to the next one's; 0: the items' rows follow one another.  With a stride (the composer's queue: a loop
    // of allocate + gadget + a gate or two per witness leaves ROWS of other calls between the items -- never Variables) a tile is ONE
    // item, whatever the gadget's W: tiles == batch.
    uint32_t stride_rows;
    // The inverses of the call (invert.hpp).  inv_in_place: the pre-pass runs BESIDE the emitter and writes every inverse at
    // its final slot, which the emitter skips -- a 32-byte hole in a 128-byte line, i.e. two partial line writes to HBM: at
    // 2^20 x range_check that alone is 0.6 ms of a witness refresh's 7.4.  Big calls (and tiny ones) run the pre-pass to
    // completion BEFORE the emitter on the same stream, and it leaves the inverses in the engine's scratch array: two planes of
    // 16-byte halves, element s = e * batch + item at inv_dense[s] and inv_dense[inv_elems + s]; the emitter (GD::kInvDense
    // gadgets) fetches them with its items' inputs and writes the variable itself, with its neighbours.  (inv_dense is
    // readable either way for those gadgets -- what is read beside a running pre-pass is not used.)
    const uint4 *inv_dense;
    uint64_t inv_elems;
    uint32_t inv_in_place;
    // Split gadgets (the fused mix): the launch that inverts writes the rows of the first `early_tiles` row tiles of every
    // `early_span` items itself, while its inversions leave HBM idle (scalar_gadgets.hpp); the rows launches leave those tiles alone.
    // 0: no such tiles.
    uint32_t early_span, early_tiles;
};
// is row tile `tile` (kRowsW = W items from item w0, Wt of them inside the batch) one of the tiles the inverting launch wrote?
// Both launches decide from the call's prefix sums alone: the tile is complete, among the first early_tiles of its span, and NO item of
// the span stopped early (the span's rows are then 10 per item -- R -- and the inverting launch, which counts failing items per
// span, knows the same thing without reading them).
// (two steps, so that the caller can issue the two loads with its own and wait once: candidate -> the span's ends; then the test)
__device__ __forceinline__ bool early_rows_candidate(const EmitOut &O, uint64_t w0, uint32_t Wt, uint32_t W, uint64_t &s0, uint64_t &s1) {
    s0 = s1 = 0;
    if (!O.early_tiles || Wt != W) return false;
    s0 = w0 - w0 % O.early_span;
    s1 = s0 + O.early_span < O.batch ? s0 + O.early_span : O.batch;
    return (w0 - s0) / W < O.early_tiles;
}

// the inverse of element s of the call, from the pre-pass's dense output into an item record (16-byte halves straight into
// the record: a copy through an Fr temporary goes through private memory).  Unconditional -- beside a running pre-pass
// (inv_in_place) the bytes are readable and not used
__device__ __forceinline__ void fetch_inverse(const EmitOut &O, uint64_t s, void *rec_fr) {
    uint4 *z = reinterpret_cast<uint4 *>(rec_fr);
    z[0] = O.inv_dense[s];
    z[1] = O.inv_dense[O.inv_elems + s];
}

// constant-table slots every gadget shares
enum : uint32_t { T_ZERO = 0, T_ONE = 1, T_NEG1 = 2, T_QC_A = 3, T_QC_B = 4, T_POW = 8 };
constexpr int kTableEntries = 8 + 256;

typedef unsigned int pg_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store16(uint4 *p, uint4 v) {
#if defined(PG_NT_STORES)
    pg_u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<pg_u32x4 *>(p));
#else
    *p = v;
#endif
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier:
// it also waits until every global STORE the wave has issued is acknowledged by memory.  The emit kernel's waves exchange
// nothing through global memory -- only item records, offsets and the constant table, all in LDS -- and they have just
// issued a sweep of stores when they reach a barrier; draining those costs microseconds per barrier under load, which is
// most of the life of a small tile's workgroup (the fused mix's variable-table launch: 8 stores per wave, 16.8 us wave
// lifetime, 68 % of it waiting -- profiles/r02f_c3_prepass_counters.json).  Here: LDS operations done, then the barrier.
__device__ __forceinline__ void lds_barrier() {
#if defined(PG_FULL_BARRIERS)
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// a value every lane of the wave holds, moved to scalar registers
__device__ __forceinline__ uint64_t uniform64(uint64_t x) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x) |
           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32;
}

__device__ __forceinline__ Fr lds_fr(const uint4 *table, uint32_t id) {
    FrVec t;
    t.v[0] = table[2 * id];
    t.v[1] = table[2 * id + 1];
    return t.f;
}

__device__ __forceinline__ void fill_common_table(uint4 *table, const uint4 *pow2, uint32_t tid, uint32_t entries) {
    for (uint32_t e = tid; e < entries; e += kThreads) {
        FrVec t;
        t.f = fr_zero();
        if (e == T_ONE) t.f = fr_one();
        else if (e == T_NEG1) t.f = fr_neg_one();
        else if (e >= T_POW && pow2) { t.v[0] = pow2[(e - T_POW) * 2]; t.v[1] = pow2[(e - T_POW) * 2 + 1]; }
        if (e != T_QC_A && e != T_QC_B) {
            table[2 * e] = t.v[0];
            table[2 * e + 1] = t.v[1];
        }
    }
}

// ragged batches: item that owns tile-relative position r (off[] = exclusive prefix sums in LDS, off[Wt] = total).
// Items much larger than a sweep step are found by walking on from the previous item; small items (a step skips
// many of them) by bisection -- unless every item of the tile has the gadget's full size UNI (no item stopped early:
// the common case), when the owner is a division by a constant and LDS is not touched at all.
template <bool SMALL_ITEMS, uint32_t UNI>
__device__ __forceinline__ uint32_t find_item(const uint32_t *off, uint32_t Wt, uint32_t r, uint32_t it, bool uniform) {
    if constexpr (UNI != 0) {
        if (uniform) return r / UNI;
    }
    if constexpr (SMALL_ITEMS) {
        uint32_t lo = it, hi = Wt;  // invariant: off[lo] <= r < off[hi]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (off[mid] <= r) lo = mid; else hi = mid;
        }
        return lo;
    } else {
        while (r >= off[it + 1]) it++;
        return it;
    }
}

// GD interface (all static, all __device__):
//   struct Args; struct ItemRec;
//   static constexpr int  W;             items per tile (<= 256)
//   static constexpr bool kRagged;       rows/vars per item vary
//   static constexpr bool kRecInRows;    selectors or wires read the item record
//   static constexpr bool kUsePow2;      table needs mont(2^i)
//   uint32_t rows_per_item(A), vars_per_item(A)                      (uniform only)
//   void fill_table(A, table, tid)       gadget constants into T_QC_A / T_QC_B
//   void item_rows(A, O, item, table, rec)  kRagged || kRecInRows: the part of the record the ROWS read (cheap: an
//                                        item's shape), written before anything else so that the waves that do not
//                                        run the item phase start storing rows at once
//   void item(A, O, item, table, rec)    per-item arithmetic -> the rest of rec (+ per-item outputs); never rewrites
//                                        what item_rows wrote
//   uint32_t kUniformRows, kUniformVars  ragged only: rows / variables of an item that did not stop early (0: none)
//   void selectors(A, rec, j, table, h, uint4 out[5])
//   void wires(A, O, rec, item, item_var_base, j, uint64_t out[3])
//   Fr   var_value(A, rec, table, k)
//   bool is_inv_slot(A, rec, k)          item-variable k holds an inverse: the pre-pass's, written in place and skipped here,
//                                        unless the call's inverses are in the records (kInvDense and !O.inv_in_place)
//   kInvDense                            optional: item() fetches the item's inverses from the pre-pass's dense output
//                                        (EmitOut::inv_dense) and var_value returns them
//   kRegionVars / kBlocks / region_n / region_block / region_k
//                                        optional: the item's variables as runs of bits and accumulators of bound blocks --
//                                        the witness refresh sweeps them by region (emit_kernel<GD, EMIT_VALUES>)
//   kSplit / RowRec / kRowsW             optional: the rows are written by launches of their own (EmitMode, below)
//   kPeriodic                            optional: the rows of full-shape tiles by rows_periodic_kernel (below)
//   int  kInv; inv_operands / inv_combine / inv_slot                        the pre-pass's view (invert.hpp)
template <class GD, bool RAGGED = GD::kRagged>
struct UniformShape {
    static constexpr uint32_t rows = 0, vars = 0;
};
template <class GD>
struct UniformShape<GD, true> {
    static constexpr uint32_t rows = GD::kUniformRows, vars = GD::kUniformVars;
};

// kPeriodic (optional, ragged gadgets with small items): tiles of full-shape items have their rows written by
// rows_periodic_kernel (below)
template <class GD, class = void>
struct Periodic {
    static constexpr bool ok = false;
};
template <class GD>
struct Periodic<GD, std::void_t<decltype(GD::kPeriodic)>> {
    static constexpr bool ok = GD::kPeriodic;
};
// largest lane count <= kThreads whose pass of two rows per lane is a whole number of items of R rows
constexpr uint32_t periodic_wire_lanes(uint32_t R) {
    uint32_t L = 256;
    while ((2 * L) % R) L--;
    return L;
}

#ifndef PG_EMIT_WAVES_PER_SIMD
#define PG_EMIT_WAVES_PER_SIMD 1  // __launch_bounds__ second argument (waves per SIMD the register allocator must allow)
#endif

// What one launch of the emit kernel writes:
//   EMIT_ALL        rows and variables of every item (inverse slots excepted: the pre-pass writes those in place)
//   EMIT_STRUCTURE  selectors and wire indices alone, no item phase at all -- for a gadget whose rows do not read the
//                   item record they are a function of the public inputs and the numbering, not of the witnesses, so a
//                   rank can regenerate another rank's rows instead of receiving them (distributed.VariablesOnlyPipeline)
//   EMIT_ROWS       selectors and wire indices of a gadget whose rows DO depend on its inputs, but only through a cheap
//                   per-item shape (GD::RowRec, GD::item_rows), over tiles of GD::kRowsW items.  For a periodic gadget
//                   (below) this launch writes only the tiles that hold an item of another shape; the full-shape tiles --
//                   all of them, unless an item stopped at its error -- are rows_periodic_kernel's
//   EMIT_VALUES     the variable assignments alone (item phase + variable sweep; the pre-pass still writes the inverse slots):
//                   what changes when the SAME circuit is rebuilt with other witnesses -- the reference's prover flow
//                   after clear_witness() (tests/scalar_gadgets_tests.rs:108-119); 33 of the 223 KB a 256-bit range_check
//                   item weighs
// A gadget with GD::kSplit (small items: the fused mix) is emitted as ONE launch of its own that inverts and writes the
// variable table (scalar_gadgets.hpp, scalar_mix_vars_kernel), followed on the same stream by rows launches: for small items the all-in-one
// launch is a poor streaming writer -- its tile is bounded by the LDS the item records take (64 items = 148 KB of output
// for the fused mix), so a workgroup's global round trips before its first store are never amortised -- while four
// fifths of its bytes (the rows) need no record at all.
enum EmitMode : int { EMIT_ALL = 0, EMIT_STRUCTURE = 1, EMIT_ROWS = 2, EMIT_VALUES = 3 };

template <class GD, class = void>
struct Split {
    static constexpr bool ok = false;
};
template <class GD>
struct Split<GD, std::void_t<decltype(GD::kSplit)>> {
    static constexpr bool ok = GD::kSplit;
};
// kRegionVars (optional, gadgets made of ladder blocks): in a witness refresh the variable sweep goes by REGION -- a wave-pass
// is the 256 bit variables or the n + 1 accumulators of one block of one item, four consecutive ones per lane -- see the sweep
template <class GD, class = void>
struct RegionVars {
    static constexpr bool ok = false;
};
template <class GD>
struct RegionVars<GD, std::void_t<decltype(GD::kRegionVars)>> {
    static constexpr bool ok = GD::kRegionVars;
};
// kInvDense (optional): the gadget's item phase can take the call's inverses from the pre-pass's dense output (EmitOut::inv_dense)
template <class GD, class = void>
struct InvDense {
    static constexpr bool ok = false;
};
template <class GD>
struct InvDense<GD, std::void_t<decltype(GD::kInvDense)>> {
    static constexpr bool ok = GD::kInvDense;
};
template <class GD, int MODE>
struct EmitShape {
    using Rec = typename GD::ItemRec;
    static constexpr int W = GD::W;
};
template <class GD>
struct EmitShape<GD, EMIT_ROWS> {
    using Rec = typename GD::RowRec;
    static constexpr int W = GD::kRowsW;
};

// the region sweep's stage (emit_kernel<GD, EMIT_VALUES>), per wave, in units of 16 bytes: 8 units for what precedes the run in
// the pass's first line (at most three variables), the half's 256, 16 for what follows it (U z y, one of T / R / A_256, at most
// three variables of the next run); one unit of padding after every 8
constexpr uint32_t kRegionLead = 8, kRegionStage = (kRegionLead + 256 + 16) / 8 * 9;

// Nothing an EMIT_ALL launch writes depends on a field inversion: the variables that hold inverses (z of maybe_equal,
// inv of is_non_zero) are written, at their final slots, by the inversion pre-pass (invert.hpp), which runs
// concurrently on the engine's side stream; the variable sweep here skips exactly those slots (GD::is_inv_slot) -- or,
// in a big call, ahead of this launch, which then writes them itself (EmitOut::inv_in_place).
template <class GD, int MODE = EMIT_ALL>
__global__ __launch_bounds__(kThreads, PG_EMIT_WAVES_PER_SIMD) void emit_kernel(const typename GD::Args A, const EmitOut O) {
    static_assert(MODE != EMIT_STRUCTURE || !(GD::kRagged || GD::kRecInRows), "rows of this gadget depend on its inputs");
    constexpr bool kVars = MODE == EMIT_ALL || MODE == EMIT_VALUES;  // item arithmetic and the variable sweep
    constexpr bool kRows = MODE != EMIT_VALUES;                      // selector and wire sweeps
    constexpr int W = EmitShape<GD, MODE>::W;
    using Rec = typename EmitShape<GD, MODE>::Rec;
    constexpr uint32_t kTable = GD::kUsePow2 ? kTableEntries : T_POW;  // gadgets without a ladder need the 8 constants only
    __shared__ uint4 s_table[kTable * 2];
    __shared__ Rec s_item[W];
    __shared__ uint32_t s_roff[W + 1], s_voff[W + 1];

    const uint32_t tid = threadIdx.x;
    {
        const uint4 *p2 = nullptr;
        if constexpr (GD::kUsePow2) p2 = GD::pow2(A);
        fill_common_table(s_table, p2, tid, kTable);
    }
    GD::fill_table(A, s_table, tid);
    lds_barrier();

    for (uint32_t t = blockIdx.x; t < O.tiles; t += gridDim.x) {
#if defined(PG_XCD_REMAP)
        // workgroups b and b + 8 share an XCD (round-robin dispatch): give each XCD one contiguous eighth of the tiles
        // (bijective for any tile count; speed only, nothing depends on the placement)
        const uint32_t q8 = O.tiles >> 3, r8 = O.tiles & 7, x = t & 7, k = t >> 3;
        const uint32_t tile = x * q8 + (x < r8 ? x : r8) + k;
#elif defined(PG_TILE_SPREAD)
        // A/B build: consecutive workgroups take tiles 1 / PG_TILE_SPREAD of the call apart (the resident workgroups then write all
        // over the arrays instead of inside one moving window)
        const uint32_t qs = O.tiles / PG_TILE_SPREAD, rs = O.tiles % PG_TILE_SPREAD, x = t % PG_TILE_SPREAD, k = t / PG_TILE_SPREAD;
        const uint32_t tile = x * qs + (x < rs ? x : rs) + k;
#else
        const uint32_t tile = t;
#endif
        const uint64_t w0 = O.stride_rows ? (uint64_t)tile : (uint64_t)tile * W;
        const uint32_t Wt = O.stride_rows ? 1u : (uint32_t)((O.batch - w0) < (uint64_t)W ? (O.batch - w0) : (uint64_t)W);
        if constexpr (MODE == EMIT_ROWS && Periodic<GD>::ok) {  // a tile of full-shape items: rows_periodic_kernel's
            if (O.row_off[w0 + Wt] - O.row_off[w0] == (uint64_t)Wt * UniformShape<GD>::rows) continue;
        }
        uint64_t row0, var0;  // tile's first row / variable relative to the call
        uint32_t G = 0, V = 0;
        // The tile's global loads are ISSUED together and waited for once: its offsets (ragged), then whatever the item
        // phase reads.  Under a saturated memory system a round trip costs microseconds, and a small-item tile has only a
        // few tens of microseconds of stores to hide it behind.
        uint64_t my_r = 0, my_v = 0, tail_r = 0, tail_v = 0;
        if constexpr (GD::kRagged) {
            row0 = O.row_off[w0];
            var0 = O.var_off[w0];
            if (tid <= Wt) {
                my_r = O.row_off[w0 + tid];
                my_v = O.var_off[w0 + tid];
            }
            if constexpr (W == kThreads) {  // Wt + 1 entries, one more than there are threads
                if (tid == 0) {
                    tail_r = O.row_off[w0 + Wt];
                    tail_v = O.var_off[w0 + Wt];
                }
            }
        } else {
            G = GD::rows_per_item(A);
            V = GD::vars_per_item(A);
            row0 = w0 * (O.stride_rows ? O.stride_rows : G);
            var0 = w0 * V;
        }

        // ---- item phase: one lane per item --------------------------------
        // With rows to write: first what they depend on (an item's shape: ladder length, stopped-early flag), so that
        // after ONE barrier every wave can store rows; the witness-dependent arithmetic (loads, Montgomery conversions)
        // then occupies the first lanes only, beside the other waves' selector and wire sweeps.
        if constexpr ((GD::kRagged || GD::kRecInRows) && MODE != EMIT_STRUCTURE) {
            for (uint32_t i = tid; i < Wt; i += kThreads) GD::item_rows(A, O, w0 + i, s_table, s_item[i]);
        }
        if constexpr (GD::kRagged) {
            if (tid <= Wt) {
                s_roff[tid] = (uint32_t)(my_r - row0);
                s_voff[tid] = (uint32_t)(my_v - var0);
            }
            if constexpr (W == kThreads) {
                if (tid == 0) {
                    s_roff[Wt] = (uint32_t)(tail_r - row0);
                    s_voff[Wt] = (uint32_t)(tail_v - var0);
                }
            }
        }
        if constexpr (GD::kRagged || GD::kRecInRows) lds_barrier();
        if constexpr (kVars) {
            if (tid < Wt) GD::item(A, O, w0 + tid, s_table, s_item[tid]);
        }

        const uint32_t total_rows = GD::kRagged ? s_roff[Wt] : Wt * G;
        const uint32_t total_vars = GD::kRagged ? s_voff[Wt] : Wt * V;
        constexpr uint32_t kUniR = UniformShape<GD>::rows, kUniV = UniformShape<GD>::vars;
        const bool uni_rows = kUniR != 0 && total_rows == Wt * kUniR, uni_vars = kUniV != 0 && total_vars == Wt * kUniV;

        if constexpr (kRows) {
            // ---- selector sweep: 16 B per lane, 128 rows x 5 columns per pass ---
            {
                const uint32_t total = total_rows * 2;
                // A wave's store is one contiguous KiB; it should also START on a 128-byte line.  A tile begins wherever its
                // first row falls (any multiple of 32 bytes when items are ragged), and a stream of KiB pieces that straddle
                // lines writes 11-17 % slower than one that does not (tools/fill_stride.py).  So the sweep runs over the
                // tile's 16-byte units counted from the line boundary before its first one: `mis` lanes of the first pass
                // have nothing to store, and every wave store of every pass starts on a line.
#if defined(PG_UNALIGNED_SWEEPS)  // A/B build
                const uint32_t mis = 0;
#else
                const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(O.q[0] + row0 * 2) >> 4) & 7u);
#endif
                const uint32_t v0 = tid < mis ? tid + kThreads : tid;  // the first unit (counted from the line) this lane stores
                const uint32_t h = (v0 - mis) & 1;
                uint32_t it = 0, j = (v0 - mis) >> 1;
                if constexpr (!GD::kRagged) { it = j / G; j -= it * G; }
                for (uint32_t u = v0; u < total + mis; u += kThreads) {
                    const uint32_t idx = u - mis;
                    if constexpr (GD::kRagged) {
                        const uint32_t r = idx >> 1;
                        it = find_item<W >= 64, kUniR>(s_roff, Wt, r, it, uni_rows);
                        j = uni_rows ? r - it * kUniR : r - s_roff[it];
                    }
                    uint4 v[5];
                    GD::selectors(A, s_item[it], j, s_table, h, v);
#pragma unroll
                    for (int c = 0; c < 5; c++) store16(O.q[c] + (row0 * 2 + idx), v[c]);
                    if constexpr (!GD::kRagged) {
                        j += kThreads / 2;
                        if (j >= G) { const uint32_t d = j / G; it += d; j -= d * G; }
                    }
                }
            }

            // ---- wire sweep: two rows (16 B) per lane per column ---------------
#pragma unroll
            for (int c = 0; c < 3; c++) {
                uint64_t *col = O.w[c] + row0;
                // pair rows so that every pair starts on a 16-byte boundary
                const uint32_t shift = (uint32_t)((reinterpret_cast<uintptr_t>(col) >> 3) & 1);
                const uint32_t pairs = (total_rows + shift + 1) >> 1;
                // ... and pairs are counted from the 128-byte line before the first one (see the selector sweep)
#if defined(PG_UNALIGNED_SWEEPS)
                const uint32_t pmis = 0;
#else
                const uint32_t pmis = (uint32_t)(((reinterpret_cast<uintptr_t>(col) - 8 * shift) >> 4) & 7u);
#endif
                uint32_t it = 0;
                for (uint32_t pv = tid < pmis ? tid + kThreads : tid; pv < pairs + pmis; pv += kThreads) {
                    const uint32_t p = pv - pmis;
                    const int64_t r0 = (int64_t)2 * p - shift;
                    uint64_t val[2];
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const int64_t r = r0 + k;
                        uint32_t rr = r < 0 ? 0u : (uint32_t)r;
                        if (rr >= total_rows) rr = total_rows - 1;
                        uint32_t j, vo;
                        if constexpr (GD::kRagged) {
                            it = find_item<W >= 64, kUniR>(s_roff, Wt, rr, it, uni_rows);
                            j = uni_rows ? rr - it * kUniR : rr - s_roff[it];
                            vo = (uni_rows && uni_vars) ? it * kUniV : s_voff[it];
                        } else {
                            it = rr / G;
                            j = rr - it * G;
                            vo = it * V;
                        }
                        uint64_t out[3];
                        GD::wires(A, O, s_item[it], w0 + it, O.var_base + var0 + vo, j, out);
                        val[k] = out[c];
                    }
                    if (r0 >= 0 && r0 + 1 < (int64_t)total_rows) {
                        store16(reinterpret_cast<uint4 *>(col + r0),
                                make_uint4((uint32_t)val[0], (uint32_t)(val[0] >> 32), (uint32_t)val[1],
                                           (uint32_t)(val[1] >> 32)));
                    } else {
                        if (r0 >= 0) col[r0] = val[0];
                        if (r0 + 1 < (int64_t)total_rows) col[r0 + 1] = val[1];
                    }
                }
            }
        }

        lds_barrier();  // item records visible

        // ---- variable sweep: one scalar (2 x 16 B) per lane ----------------
        if constexpr (kVars) {
            const bool inv_here = InvDense<GD>::ok && !O.inv_in_place;  // the inverses are in the records: written here
            // slots are counted from the 128-byte line before the tile's first one (see the selector sweep): a wave's two
            // stores then cover whole lines between them
#if defined(PG_UNALIGNED_SWEEPS)
            const uint32_t smis = 0;
#else
            const uint32_t smis = (uint32_t)((reinterpret_cast<uintptr_t>(O.vars + var0 * 2) >> 5) & 3u);
#endif
#if !defined(PG_VAR_SWEEP_SINGLE)
            if constexpr (RegionVars<GD>::ok && MODE == EMIT_VALUES) {
                // A launch that writes ONLY variables is bound by instruction issue, not by HBM, when it sweeps them as the full
                // emission does (one scalar per lane, slot by slot: 0.54 of peak): half of a block's variables are accumulators,
                // A_i = mont(T mod 2^i), one Montgomery multiplication each, and a wave-instruction is paid per WAVE -- a pass
                // that holds one accumulator runs the multiplication for all 64 lanes.  So the sweep goes by region: a
                // wave-pass is the 256 bit variables, or the n + 1 accumulators, of ONE block of ONE item; a lane takes four
                // consecutive ones, the first accumulator by the multiplication, the next three by the reference's own
                // update A_{i+1} = A_i + b_i mont(2^i) (range.rs:152: a modular addition of a table entry; values fully reduced,
                // so the limbs are those of the closed form).  One multiplication per 256 accumulators instead of four; a pass
                // over bits runs none.  A lane's four scalars are 128 contiguous bytes; the wave's 8 KiB leave in two halves
                // through a wave-private LDS buffer as stores of one contiguous KiB that start on a line.
                __shared__ uint4 s_reg[4 * kRegionStage];
                const uint32_t wave = tid >> 6, lane = tid & 63;
                uint4 *wp = s_reg + wave * kRegionStage;
                constexpr uint32_t B2 = 2 * GD::kBlocks;
                const uint32_t units = Wt * B2;
                const uint32_t tile_mis = (uint32_t)((reinterpret_cast<uintptr_t>(O.vars + var0 * 2) >> 5) & 3u);  // variables into its line
                for (uint32_t q = wave; q < units; q += kThreads / 64) {
                    const uint32_t it = q / B2, r = q - it * B2, b = r >> 1;
                    const uint32_t kind = (r + (GD::kBlocks == 2 ? it : it >> 1)) & 1;  // bits / accumulators alternate per wave
                    const Rec &R = s_item[it];
                    const auto &Bk = GD::region_block(R, b);  // (a BoundRec: its canonical T is what the run is made from)
                    const uint32_t n = GD::region_n(A, R);
                    const uint32_t count = kind ? (n + 1 < 256 ? n + 1 : 256) : 256;
                    uint32_t vitem, vend;  // the item's first variable and the one after its last, relative to the tile
                    if constexpr (GD::kRagged) {
                        vitem = uni_vars ? it * kUniV : s_voff[it];
                        vend = uni_vars ? vitem + kUniV : s_voff[it + 1];
                    } else {
                        vitem = it * V;
                        vend = vitem + V;
                    }
                    const uint32_t kfirst = GD::region_k(A, R, b, kind ? 257 : 1);  // the run's first variable, in the item
                    // WHO WRITES WHICH LINE.  A 128-byte line (four variables) that reaches HBM in two pieces costs far more than
                    // its share -- the pre-pass's 32-byte z alone, written in place, was 0.6 ms of 7.4 -- so every line of the
                    // tile is written by ONE pass, in one store instruction.  The variables outside the runs (x, T, U, z, y, R,
                    // A_256) go with a neighbouring run: what precedes the item's first run with that run, what follows an
                    // accumulator run with it.  The lines two passes would share belong to the ACCUMULATOR pass on both sides: it
                    // starts at the line that holds its first variable (up to three bits of its own block before it) and ends
                    // with the line that holds its last (up to three variables of the next run: bits of the next block, or x, T,
                    // bits of the next item); a bit pass keeps to the whole lines in between.  Only the tile's first and last line
                    // are shared, with other workgroups.
                    const uint32_t mis0 = (tile_mis + vitem) & 3u;  // the item's first variable, in variables into its line
                    uint32_t ks, ke;                                // the variables the pass writes: [ks, ke) of the item
                    if (kind == 0) {
                        ks = b == 0 ? 0 : kfirst;
                        if (it != 0 || b != 0) ks += (4u - ((mis0 + ks) & 3u)) & 3u;  // (the tile's first line: from where the tile starts)
                        ke = kfirst + 256;
                        ke -= (mis0 + ke) & 3u;
                    } else {
                        ks = kfirst - ((mis0 + kfirst) & 3u);
                        ke = b + 1 < GD::kBlocks ? GD::region_k(A, R, b + 1, 1) : vend - vitem;
                        if (it + 1 != Wt || b + 1 < GD::kBlocks) ke += (4u - ((mis0 + ke) & 3u)) & 3u;  // (the tile's last: to where it ends)
                    }
                    const uint32_t h_end = (count - 1) >> 7;  // the half that holds the run's end
                    uint32_t hole = ~0u;                      // the variable the pre-pass has written in place (z), if any
                    if (!inv_here)
                        for (uint32_t k = kfirst + count; k < ke && k < vend - vitem; k++)
                            if (GD::is_inv_slot(A, R, k)) hole = k;
                    FrVec v[4];
                    const uint32_t i0 = 4 * lane;
                    if (kind == 0) {  // range.rs:128-131
#pragma unroll
                        for (int t = 0; t < 4; t++) v[t].f = raw_bit(Bk.Tc, i0 + t) ? fr_one() : fr_zero();
                    } else {
                        v[0].f = fr_zero();
                        if (i0 && i0 < count) v[0].f = fr_to_mont(raw_low_bits(Bk.Tc, i0));
#pragma unroll
                        for (int t = 1; t < 4; t++) {
                            v[t].f = v[t - 1].f;
                            if (i0 + t < count && raw_bit(Bk.Tc, i0 + t - 1)) {
                                FrVec p;
                                p.v[0] = s_table[2 * (T_POW + i0 + t - 1)];
                                p.v[1] = s_table[2 * (T_POW + i0 + t - 1) + 1];
                                v[t].f = fr_add(v[t - 1].f, p.f);
                            }
                        }
                    }
                    // the variables of the pass outside its run: lanes 0 .. npre + npost - 1 (records in the LDS; no arithmetic
                    // but for A_256)
                    const uint32_t npre = ks < kfirst ? kfirst - ks : 0, npost = ke > kfirst + count ? ke - (kfirst + count) : 0;
                    FrVec nb;
                    uint32_t nb_at = ~0u;  // where the lane's goes in the stage (units; ~0: the lane has none)
                    if (lane < npre + npost) {
                        const uint32_t k = lane < npre ? ks + lane : kfirst + count + (lane - npre);
                        const uint32_t vitem_n = vend - vitem;  // past the item's last variable: the next item's first ones
                        nb.f = k < vitem_n ? GD::var_value(A, R, s_table, k) : GD::var_value(A, s_item[it + 1], s_table, k - vitem_n);
                        nb_at = kRegionLead + 2 * (k - kfirst) - (lane < npre ? 0 : 256 * h_end);
                    }
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        // the stage holds the half's 256 units of 16 bytes from unit kRegionLead on, the other variables before
                        // and after them; unit u lives at wp[u + u / 8]: the lanes of a 16-lane group, 128 bytes apart, then write
                        // 16 different 16-byte columns of the LDS
                        if ((lane >> 5) == (uint32_t)h) {
                            const uint32_t u0 = kRegionLead + 8 * (lane & 31), w0 = u0 + (u0 >> 3);
#pragma unroll
                            for (int t = 0; t < 4; t++) {
                                wp[w0 + 2 * t] = v[t].v[0];
                                wp[w0 + 2 * t + 1] = v[t].v[1];
                            }
                        }
                        if (h == 1 && lane == 31) {  // the first half's last line is completed by the second: its tail, before the lead
#pragma unroll
                            for (int t = 0; t < 4; t++) {
                                wp[kRegionLead - 8 + 2 * t] = v[t].v[0];
                                wp[kRegionLead - 8 + 2 * t + 1] = v[t].v[1];
                            }
                        }
                        if (nb_at != ~0u && (lane < npre ? h == 0 : (uint32_t)h == h_end)) {
                            wp[nb_at + (nb_at >> 3)] = nb.v[0];
                            wp[nb_at + 1 + ((nb_at + 1) >> 3)] = nb.v[1];
                        }
                        // (wave-private, and the LDS runs a wave's instructions in order: the reads below see the other lanes'
                        // writes, the next half's writes come after these reads -- the compiler must only keep the order)
                        asm volatile("" ::: "memory");
                        // the half's variables [lo, hi) of the item
                        const uint32_t mid = kfirst + 128 - ((mis0 + kfirst + 128) & 3u);  // (the halves meet on a line boundary)
                        const uint32_t lo = h == 0 ? ks : mid, hi = (uint32_t)h == h_end ? ke : (h == 0 ? mid : lo);
                        const uint32_t total = hi > lo ? 2 * (hi - lo) : 0;
                        uint4 *dst = O.vars + ((uint64_t)var0 + vitem + lo) * 2;
                        const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(dst) >> 4) & 7u);  // units into its line
                        const uint32_t lead = kRegionLead + 2 * (lo - kfirst) - 256 * h;  // (lo < kfirst: below the lead, never below 0)
#pragma unroll
                        for (int j = 0; j < 5; j++) {
                            const uint32_t g = 64 * j + lane - mis;  // (wraps below zero for the lanes before the first unit)
                            const uint32_t u = g + lead;
                            if (g < total && lo + (g >> 1) != hole) store16(dst + g, wp[u + (u >> 3)]);
                        }
                        asm volatile("" ::: "memory");
                    }
                }
            } else
#endif
            {
            const uint32_t sv0 = tid < smis ? tid + kThreads : tid;
                uint32_t it = 0, k = sv0 - smis;
                if constexpr (!GD::kRagged) { it = k / V; k -= it * V; }
                for (uint32_t sv = sv0; sv < total_vars + smis; sv += kThreads) {
                    const uint32_t s = sv - smis;
                    if constexpr (GD::kRagged) {
                        it = find_item<W >= 64, kUniV>(s_voff, Wt, s, it, uni_vars);
                        k = uni_vars ? s - it * kUniV : s - s_voff[it];
                    }
                    uint4 *dst = O.vars + (var0 + s) * 2;
                    if (inv_here || !GD::is_inv_slot(A, s_item[it], k)) {  // (the pre-pass's, written in place)
                        FrVec val;
                        val.f = GD::var_value(A, s_item[it], s_table, k);
                        store16(dst, val.v[0]);
                        store16(dst + 1, val.v[1]);
                    }
                    if constexpr (!GD::kRagged) {
                        k += kThreads;
                        if (k >= V) { const uint32_t d = k / V; it += d; k -= d * V; }
                    }
                }
            }
            }
        lds_barrier();  // records and offsets are rewritten by the next tile
    }
}

// ---- the rows of a periodic gadget's full-shape tiles ----------------------------------------------------------------
// In a tile whose items all have the gadget's full shape (GD::kPeriodic: ragged gadgets with small items), selectors depend
// on the row-within-item alone and wires are `constant` or `item's first variable + constant`.  The sweeps use a lane
// count that is a multiple of the item size, so that a lane meets the SAME row-within-item on every pass: selector values
// are computed once per workgroup, wire offsets once per tile, and the loops are bare stores (the generic sweeps spend
// ~30 vector + ~20 scalar instructions per 16-byte store on finding the item, the row and the constants again).  The
// launch is a pure store stream: no item phase, no per-item record, 256 bytes of LDS (the constants) and 62 registers: eight
// waves per SIMD, which a store stream needs (a wave keeps only a handful of stores in flight).  It runs AFTER the gadget's
// inverting launch, on the same stream.
// Tiles that hold an item of another shape (is_non_zero stopped at its error, scalar.rs:79) are left to
// emit_kernel<GD, EMIT_ROWS>, which skips the ones written here: both read the shape off the call's prefix sums.
#ifndef PG_ROWS_WAVES_PER_SIMD
#define PG_ROWS_WAVES_PER_SIMD 8
#endif
// the rows of ONE tile of full-shape items (Wt items from item w0; first row row0 and first Variable var0, relative to the call) by
// 256 lanes `tid` -- the periodic launch's workgroup, or the four waiting waves of the inverting launch (scalar_gadgets.hpp).  v[5]: the
// lane's selector halves (lanes 2r, 2r + 1 hold the halves of row r of a pass of IPP whole items: the same on every pass)
template <class GD>
__device__ __forceinline__ void periodic_tile_rows(const typename GD::Args &A, const EmitOut &O, const uint4 v[5], uint32_t tid, uint64_t w0,
                                                   uint32_t Wt, uint64_t row0, uint64_t var0) {
    constexpr uint32_t R = GD::kUniformRows, VV = GD::kUniformVars;
    constexpr uint32_t IPP = (kThreads / 2) / R, LS = IPP * R * 2;
    constexpr uint32_t LW = periodic_wire_lanes(R), IPW = 2 * LW / R;  // wires: two rows per lane, IPW whole items per pass
    typename GD::RowRec full{};  // the shape record of a full item
    const uint32_t total_rows = Wt * R;
    if (tid < LS) {
        // A lane's five stores of one pass would go to the same row of five arrays of the same size: whether those five
        // addresses fall on the same memory channel is then decided by the arrays' base addresses, once, for the whole
        // call (up to 25 % between placements of the same columns, profiles/NOTES_r02.md).  The values of a lane do not
        // depend on the pass, so column c starts c fifths of the tile further on and wraps: the five streams are that far
        // apart in their arrays, by an amount that is not a power of two (-3 % on the fused mix's step).
        const uint32_t passes = (total_rows * 2 + LS - 1) / LS;
        uint4 *base[5];
        uint32_t at[5];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            base[c] = O.q[c] + (row0 * 2 + tid);
            at[c] = (uint32_t)(((uint64_t)c * passes) / 5);
        }
        for (uint32_t p = 0; p < passes; p++) {
#pragma unroll
            for (int c = 0; c < 5; c++) {
                const uint32_t off = at[c] * LS;
                if (off + tid < total_rows * 2) store16(base[c] + off, v[c]);
                at[c] = at[c] + 1 == passes ? 0 : at[c] + 1;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        uint64_t *col = O.w[c] + row0;
        // pair rows so that every pair starts on a 16-byte boundary
        const uint32_t shift = (uint32_t)((reinterpret_cast<uintptr_t>(col) >> 3) & 1);
        if (tid < LW) {
            uint64_t wbase[2], slope[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                // (shifted columns: lane 0's first row is row -1 of the pass -- masked in the tile's first pass, but the LAST
                // row of the item before the pass's first in every later one: item -1, row R - 1, not item 0, row 0)
                const int32_t r = (int32_t)(2 * tid + k) - (int32_t)shift;
                const int32_t it = r < 0 ? -1 : r / (int32_t)R;
                const uint32_t j = (uint32_t)(r - it * (int32_t)R);
                const uint64_t vb = O.var_base + var0 + (uint64_t)(int64_t)it * VV;
                uint64_t a[3], b[3];
                GD::wires(A, O, full, w0 + (uint64_t)(int64_t)it, vb, j, a);
                GD::wires(A, O, full, w0 + (uint64_t)(int64_t)it, vb + 1, j, b);
                wbase[k] = a[c];
                slope[k] = b[c] - a[c];  // 1: a Variable of the item; 0: a constant (zero_var)
            }
            const uint64_t step = (uint64_t)IPW * VV;
            uint64_t adv = 0;
            for (int64_t r0 = (int64_t)2 * tid - shift; r0 < (int64_t)total_rows; r0 += 2 * LW, adv += step) {
                const uint64_t v0 = wbase[0] + slope[0] * adv, v1 = wbase[1] + slope[1] * adv;
                if (r0 >= 0 && r0 + 1 < (int64_t)total_rows) {
                    store16(reinterpret_cast<uint4 *>(col + r0),
                            make_uint4((uint32_t)v0, (uint32_t)(v0 >> 32), (uint32_t)v1, (uint32_t)(v1 >> 32)));
                } else {
                    if (r0 >= 0) col[r0] = v0;
                    if (r0 + 1 < (int64_t)total_rows) col[r0 + 1] = v1;
                }
            }
        }
    }
}
// the lane's selector halves for periodic_tile_rows (table: the eight common constants, fill_common_table)
template <class GD>
__device__ __forceinline__ void periodic_lane_selectors(const typename GD::Args &A, const uint4 *table, uint32_t tid, uint4 v[5]) {
    constexpr uint32_t R = GD::kUniformRows, IPP = (kThreads / 2) / R, LS = IPP * R * 2;
    typename GD::RowRec full{};
    const uint32_t rl = (tid < LS ? tid : 0) >> 1;
    GD::selectors(A, full, rl - (rl / R) * R, table, tid & 1, v);
}

template <class GD>
__global__ __launch_bounds__(kThreads, PG_ROWS_WAVES_PER_SIMD) void rows_periodic_kernel(const typename GD::Args A, const EmitOut O) {
    constexpr int W = GD::kRowsW;
    constexpr uint32_t R = GD::kUniformRows;
    __shared__ uint4 s_table[T_POW * 2];
    const uint32_t tid = threadIdx.x;
#if defined(PG_ROWS_SETPRIO)
    __builtin_amdgcn_s_setprio(PG_ROWS_SETPRIO);
#endif
    fill_common_table(s_table, nullptr, tid, T_POW);
    GD::fill_table(A, s_table, tid);
    lds_barrier();
    // selectors: lanes 2r, 2r+1 hold the halves of row r; IPP whole items per pass.  What a lane stores is the same on
    // every pass of every tile
    uint4 v[5];
    periodic_lane_selectors<GD>(A, s_table, tid, v);

    for (uint32_t tile = blockIdx.x; tile < O.tiles; tile += gridDim.x) {
        const uint64_t w0 = (uint64_t)tile * W;
        const uint32_t Wt = (uint32_t)((O.batch - w0) < (uint64_t)W ? (O.batch - w0) : (uint64_t)W);
        // (the three offsets are the same for every lane, but the compiler cannot load them through the scalar cache -- the
        // columns' stores might alias them -- and would make the wire sweep wait for them behind EVERY selector store of the
        // tile: vmcnt counts loads and stores alike.  Made scalar here, they are waited for before the first store.)
        uint64_t s0, s1;
        const bool cand = early_rows_candidate(O, w0, Wt, W, s0, s1);
        const uint64_t span_rows = cand ? O.row_off[s1] - O.row_off[s0] : 0;  // (issued with the three below, waited for once)
        const uint64_t row0 = uniform64(O.row_off[w0]), var0 = uniform64(O.var_off[w0]), row1 = uniform64(O.row_off[w0 + Wt]);
        if (row1 - row0 != Wt * R) continue;  // an item of another shape: the generic launch's tile
        if (cand && uniform64(span_rows) == (s1 - s0) * R) continue;  // written by the inverting launch while it inverted
        periodic_tile_rows<GD>(A, O, v, tid, w0, Wt, row0, var0);
    }
}

// bare streaming fill: the practical write ceiling the emitters are compared with.  The buffer is split into
// `streams` equal contiguous parts that every workgroup advances together, 4 KiB per part per pass -- the emitters'
// shape (five selector columns at once); one single linear stream measures ~18 % lower on MI355X.
__global__ __launch_bounds__(kThreads) void fill_kernel(uint4 *dst, uint64_t n16, uint32_t streams, uint64_t pattern) {
    const uint4 v = make_uint4((uint32_t)pattern, (uint32_t)(pattern >> 32), (uint32_t)~pattern, (uint32_t)(~pattern >> 32));
    const uint64_t part = n16 / streams;       // uint4 per part (the remainder is filled by the last part's tail loop)
    constexpr uint64_t kPiece = 65536;         // 1 MiB of each part per workgroup piece
    for (uint64_t base = (uint64_t)blockIdx.x * kPiece; base < part; base += (uint64_t)gridDim.x * kPiece) {
        const uint64_t end = base + kPiece < part ? base + kPiece : part;
        for (uint64_t i = base + threadIdx.x; i < end; i += kThreads)
            for (uint32_t s = 0; s < streams; s++) store16(dst + (uint64_t)s * part + i, v);
    }
    if (blockIdx.x == 0)
        for (uint64_t i = part * streams + threadIdx.x; i < n16; i += kThreads) store16(dst + i, v);
}

// the same fill as short-lived workgroups: one 8 KiB block per workgroup (two 16-byte stores per lane), no loop, started in address
// order by the dispatcher, and only TWO resident per CU (kFillOneshotLds bytes of dynamic LDS each, unused): at any moment the chip writes
// one moving window of a few MiB.  This is the shape that does not care where its array lies: 7.14-7.21 TB/s on every one of six 34.7-GB
// tables where long-lived workgroups reach 5.6-7.1 by table (tools/probes/single_table_fill.hip; four stores per lane at full residency,
// this kernel until the end of round 5: 5.9-6.5).
constexpr uint32_t kFillOneshotLds = 160 * 1024 / 2 - 1024;
constexpr uint32_t kFillOneshotUnits = 2 * kThreads;  // 16-byte units per workgroup
__global__ __launch_bounds__(kThreads) void fill_oneshot_kernel(uint4 *dst, uint64_t n16, uint64_t pattern) {
    extern __shared__ uint4 fill_oneshot_pad[];
    const uint4 v = make_uint4((uint32_t)pattern, (uint32_t)(pattern >> 32), (uint32_t)~pattern, (uint32_t)(~pattern >> 32));
    const uint64_t base = (uint64_t)blockIdx.x * kFillOneshotUnits + threadIdx.x;
#pragma unroll
    for (uint32_t k = 0; k < kFillOneshotUnits / kThreads; k++)
        if (base + k * kThreads < n16) store16(dst + base + k * kThreads, v);
    if (n16 == ~0ull) fill_oneshot_pad[threadIdx.x] = v;  // (keeps the allocation)
}

// the emitters' store stream with nothing behind it: a workgroup owns a tile of consecutive rows (and the matching share of
// the variable table) and sweeps the five selector columns in lock step (16 B per lane, 4 KiB of each column per pass), then
// the three wire columns, then the variable table -- exactly the emit kernel's sweeps, one constant instead of the table
// look-ups and the arithmetic.  What it reaches on the arrays a workload writes is that workload's store ceiling ON THOSE
// ARRAYS (where lock-step streams lie decides 10-18 % on MI355X: DESIGN.md section 2): bench.py times it beside every
// workload.  rows_per_tile is even and a multiple of 8 (whole 128-byte lines of every column), vars_per_tile a multiple of 4.
__global__ __launch_bounds__(kThreads) void fill_columns_kernel(EmitOut O, uint64_t n_gates, uint64_t n_vars, uint64_t rows_per_tile,
                                                                uint64_t vars_per_tile, uint64_t pattern) {
    const uint4 v = make_uint4((uint32_t)pattern, (uint32_t)(pattern >> 32), (uint32_t)~pattern, (uint32_t)(~pattern >> 32));
    for (uint64_t tile = blockIdx.x; tile < O.tiles; tile += gridDim.x) {
        const uint64_t r0 = tile * rows_per_tile, r1 = r0 + rows_per_tile < n_gates ? r0 + rows_per_tile : n_gates;
        const uint64_t v0 = tile * vars_per_tile, v1 = tile + 1 == O.tiles ? n_vars : (v0 + vars_per_tile < n_vars ? v0 + vars_per_tile : n_vars);
        if (r0 < r1) {
            for (uint64_t u = 2 * r0 + threadIdx.x; u < 2 * r1; u += kThreads) {
#pragma unroll
                for (int c = 0; c < 5; c++) store16(O.q[c] + u, v);
            }
            for (uint64_t u = r0 / 2 + threadIdx.x; u < r1 / 2; u += kThreads) {
#pragma unroll
                for (int c = 0; c < 3; c++) store16(reinterpret_cast<uint4 *>(O.w[c]) + u, v);
            }
            if ((r1 & 1) && threadIdx.x == 0)  // (only the call's last row can be an odd one out)
                for (int c = 0; c < 3; c++) O.w[c][r1 - 1] = pattern;
        }
        for (uint64_t u = 2 * v0 + threadIdx.x; u < 2 * v1; u += kThreads) store16(O.vars + u, v);
    }
}

// ---- BlsScalar <-> its 32-byte little-endian canonical encoding, in bulk ----------------
// (dusk-bytes Serializable: from_bytes rejects values >= q, to_bytes leaves Montgomery form; src/range.rs:162)
__global__ __launch_bounds__(kThreads) void from_canonical_kernel(const uint4 *raw, uint64_t n, uint4 *out, uint8_t *bad_mask,
                                                                 uint32_t *bad_count) {
    const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    FrVec x;
    x.v[0] = raw[2 * i];
    x.v[1] = raw[2 * i + 1];
    const Fr q = fr_modulus();
    bool ge = true;  // x >= q ?
    for (int k = 3; k >= 0; k--)
        if (x.f.l[k] != q.l[k]) { ge = x.f.l[k] > q.l[k]; break; }
    FrVec r;
    r.f = ge ? fr_zero() : fr_to_mont(x.f);
    out[2 * i] = r.v[0];
    out[2 * i + 1] = r.v[1];
    if (bad_mask) bad_mask[i] = ge ? 1 : 0;
    if (ge) atomicAdd(bad_count, 1u);
}
__global__ __launch_bounds__(kThreads) void to_canonical_kernel(const uint4 *in, uint64_t n, uint4 *raw) {
    const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    FrVec x, r;
    x.v[0] = in[2 * i];
    x.v[1] = in[2 * i + 1];
    r.f = fr_from_mont(x.f);
    raw[2 * i] = r.v[0];
    raw[2 * i + 1] = r.v[1];
}

// ---- exclusive prefix sums for ragged batches ---------------------------
// counts[i] (rows, vars of item i) -> off[i], off[batch] = total.  The plan kernel leaves the per-item counts AND
// their per-block sums (plan_block_sums); one more launch turns them into offsets (two above 4 M items).
constexpr int kScanBlock = 1024;  // items per block (256 threads x 4)

__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t *s_warp, uint64_t &block_total) {
    // 256 threads = 4 waves of 64
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t t = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += t;
    }
    if (lane == 63) s_warp[wave] = incl;
    __syncthreads();
    uint64_t base = 0;
    for (uint32_t wv = 0; wv < wave; wv++) base += s_warp[wv];
    block_total = s_warp[0] + s_warp[1] + s_warp[2] + s_warp[3];
    __syncthreads();
    return base + incl - v;
}

// single block: in-place exclusive scan of the block sums (nblk arbitrary: loops in chunks of 256)
__global__ __launch_bounds__(kThreads) void scan_top_kernel(uint64_t *blk_rows, uint64_t *blk_vars, uint32_t nblk) {
    __shared__ uint64_t s_warp[4];
    uint64_t carry_r = 0, carry_v = 0;
    for (uint32_t base = 0; base < nblk; base += kThreads) {
        uint32_t i = base + threadIdx.x;
        uint64_t r = i < nblk ? blk_rows[i] : 0, v = i < nblk ? blk_vars[i] : 0, tr, tv;
        uint64_t er = block_exclusive_scan(r, s_warp, tr);
        uint64_t ev = block_exclusive_scan(v, s_warp, tv);
        if (i < nblk) { blk_rows[i] = carry_r + er; blk_vars[i] = carry_v + ev; }
        carry_r += tr;
        carry_v += tv;
    }
}

// The totals of the plan -- rows, variables, and the plan kernel's count of failing items (err_count, or NULL for a plan
// that has none) -- go straight to the engine's pinned result record `host` (a device-visible host address: three
// 4..8-byte copies cost three copy launches, ~15 us of a 0.65 ms step); err_count is left at zero for the next plan.
struct PlanTotals {
    uint64_t n_gates, n_vars;
    uint32_t errs, pad;
};


// ---- the plan in ONE launch (batches up to kPlanFusedBlocks blocks) -------------------------------------------------
// A plan kernel that has its threads' four counts in registers finishes the prefix sums itself: every block publishes its
// totals, adds up the totals of the blocks before it back to the nearest one that has already published its inclusive
// prefix (decoupled look-back: a block waits only for blocks of lower index, which were dispatched before it and publish
// before they wait, so the lowest unpublished block can always run), publishes its own inclusive prefix and writes its
// items' offsets; the block of the last item writes the totals.  The block that finishes last writes the error count to
// the host record and puts the published words, the counter and the error count back to zero: nothing to clear between launches, so a launch can be replayed from a HIP
// graph.  Saves the second launch, its dependency and the per-item counts (8 B per item written and read back).
//
// NO FENCES.  On this part a release / acquire at agent scope writes back / invalidates the whole L2 of the XCD (the L2s
// of the eight XCDs are not coherent with each other for ordinary memory): with one per block the first version of this
// kernel took 470 us beside the rows' store stream.  So a block's totals and its "published" bit travel in ONE 64-bit
// word that is only ever touched by relaxed read-modify-write atomics (which execute at the coherence point): nothing
// has to be ordered against anything else.  Every wait is bounded (kPlanSpinLimit polls); a plan that gives up says so in
// PlanTotals::pad.
struct PlanScan {
    unsigned long long *agg;      // per block: flags | rows << 31 | variables (zero between launches); [blocks_cap] = blocks done
    uint64_t *blk_rows, *blk_vars;  // block sums for the two-launch scan (fused == 0)
    uint64_t *row_off, *var_off;  // the call's outputs (batch + 1 entries each)
    PlanTotals *host;
    uint32_t *err_count;          // the plan kernel's count of failing items, or NULL
    uint32_t blocks_cap;          // index of the done counter in agg
    uint32_t fused;               // 0: leave the counts and block sums to scan_final_kernel (more blocks than kPlanFusedBlocks)
};
// a published word: kAggA = the block's own totals, kAggP = its INCLUSIVE prefix (a look-back stops there); 31 bits each
// for rows and variables, which bounds the fused path: 2048 blocks x 1024 items x 517 < 2^31
constexpr unsigned long long kAggA = 1ull << 63, kAggP = 1ull << 62;
constexpr uint32_t kPlanFusedBlocks = 2048;
constexpr uint32_t kPlanSpinLimit = 1u << 22;

__device__ __forceinline__ unsigned long long plan_rmw_read(unsigned long long *p) {
    return __hip_atomic_fetch_add(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void plan_publish(unsigned long long *p, unsigned long long flag, uint64_t rows, uint64_t vars) {
    __hip_atomic_exchange(p, flag | rows << 31 | vars, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t wave_sum(uint64_t x) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

__device__ __forceinline__ void plan_finish(const PlanScan &P, const uint32_t r[4], const uint32_t v[4], uint64_t batch, uint32_t my_errs) {
    __shared__ uint64_t s_w[4], s_pref[2];
    __shared__ uint32_t s_last;
    const uint32_t tid = threadIdx.x, b = blockIdx.x, nblk = gridDim.x;
    uint64_t tr, tv, te;
    const uint64_t er = block_exclusive_scan((uint64_t)r[0] + r[1] + r[2] + r[3], s_w, tr);
    const uint64_t ev = block_exclusive_scan((uint64_t)v[0] + v[1] + v[2] + v[3], s_w, tv);
    (void)block_exclusive_scan((uint64_t)my_errs, s_w, te);  // the block's failing items: ONE atomic per block, below
    if (tid == 0 && b > 0) plan_publish(&P.agg[b], kAggA, tr, tv);
    if (tid < 64) {  // wave 0 looks back, 64 predecessors at a time, nearest first, until one of them holds a prefix
        uint64_t pr = 0, pv = 0;
        bool gave_up = false;
        int64_t base = (int64_t)b - 1;
        for (bool more = b > 0; more; base -= 64) {
            const int64_t j = base - (int64_t)tid;
            unsigned long long x = kAggP;  // before block 0: the empty prefix
            if (j >= 0) {
                x = plan_rmw_read(&P.agg[j]);
                for (uint32_t polls = 0; !(x >> 62); x = plan_rmw_read(&P.agg[j])) {
                    if (++polls > kPlanSpinLimit) { gave_up = true; x = kAggP; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            const uint64_t has_prefix = __ballot((x & kAggP) != 0);
            const uint32_t first = has_prefix ? (uint32_t)__ffsll((unsigned long long)has_prefix) - 1 : 64u;  // the nearest one
            const bool take = tid <= first;
            pr += wave_sum(take ? (x >> 31) & 0x7fffffffull : 0);
            pv += wave_sum(take ? x & 0x7fffffffull : 0);
            more = has_prefix == 0;
        }
        if (__ballot(gave_up) && tid == 0) P.host->pad = 1;
        if (tid == 0) {
            s_pref[0] = pr;
            s_pref[1] = pv;
            plan_publish(&P.agg[b], kAggP, pr + tr, pv + tv);
        }
    }
    __syncthreads();
    uint64_t ro = s_pref[0] + er, vo = s_pref[1] + ev;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t i = (uint64_t)b * kScanBlock + tid * 4 + k;
        if (i < batch) { P.row_off[i] = ro; P.var_off[i] = vo; }
        ro += r[k];
        vo += v[k];
        if (i + 1 == batch) {  // the last item: the totals
            P.row_off[batch] = ro;
            P.var_off[batch] = vo;
            P.host->n_gates = ro;
            P.host->n_vars = vo;
        }
    }
    // The last block to get here must have every block's error count behind it.  A block adds its count with ONE atomic
    // whose RESULT it consumes before it counts itself done: a returning atomic has been performed at the coherence point
    // when its value arrives, so the two are ordered by a data dependency (a non-returning add could still be in flight
    // when the done counter is bumped; a workgroup-scope fence does not wait for it).
    if (tid == 0) {
        if (P.err_count && te) {
            uint32_t before = __hip_atomic_fetch_add(P.err_count, (uint32_t)te, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(before));  // the value is needed here: the wave waits for the atomic to return
        }
        s_last = __hip_atomic_fetch_add(&P.agg[P.blocks_cap], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    for (uint32_t j = tid; j < nblk; j += kThreads) __hip_atomic_exchange(&P.agg[j], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
        P.host->errs = P.err_count ? __hip_atomic_exchange(P.err_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        __hip_atomic_exchange(&P.agg[P.blocks_cap], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// what every plan kernel ends with: its threads' four row / variable counts -> offsets (fused) or counts + block sums
__device__ __forceinline__ void plan_store(const PlanScan &P, const uint32_t r[4], const uint32_t v[4], uint64_t batch, uint32_t *rows,
                                           uint32_t *vars, uint32_t my_errs = 0);

// block sums straight from a plan kernel: thread t of plan block b owns items b * kScanBlock + 4 t .. + 3 (the
// indexing of scan_final_kernel) and hands in the sums of its four counts -- the separate block-sums launch is gone
__device__ __forceinline__ void plan_block_sums(uint64_t r, uint64_t v, uint64_t *blk_rows, uint64_t *blk_vars) {
    __shared__ uint64_t s_plan_warp[4];
    uint64_t tr, tv;
    block_exclusive_scan(r, s_plan_warp, tr);
    block_exclusive_scan(v, s_plan_warp, tv);
    if (threadIdx.x == 0) { blk_rows[blockIdx.x] = tr; blk_vars[blockIdx.x] = tv; }
}

// my_errs: this thread's failing items (counted into P.err_count: by one atomic per block on the fused path, per thread on the other)
__device__ __forceinline__ void plan_store(const PlanScan &P, const uint32_t r[4], const uint32_t v[4], uint64_t batch, uint32_t *rows,
                                           uint32_t *vars, uint32_t my_errs) {
    if (P.fused) {
        plan_finish(P, r, v, batch, my_errs);
        return;
    }
    if (my_errs && P.err_count) atomicAdd(P.err_count, my_errs);  // (read by scan_final_kernel, a later launch)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x * 4 + k;
        if (i < batch) { rows[i] = r[k]; vars[i] = v[k]; }
    }
    plan_block_sums((uint64_t)r[0] + r[1] + r[2] + r[3], (uint64_t)v[0] + v[1] + v[2] + v[3], P.blk_rows, P.blk_vars);
}

// blk_prefixed = 0: blk_* hold the block SUMS and every block adds up the ones before it itself (no scan_top launch;
// the host takes this route up to kScanDirectBlocks blocks = 4 M items); 1: scan_top_kernel has turned them into
// exclusive prefix sums
constexpr uint32_t kScanDirectBlocks = 4096;

__global__ __launch_bounds__(kThreads) void scan_final_kernel(const uint32_t *rows, const uint32_t *vars, uint64_t n,
                                                             const uint64_t *blk_rows, const uint64_t *blk_vars,
                                                             uint64_t *row_off, uint64_t *var_off, uint32_t blk_prefixed,
                                                             PlanTotals *host, uint32_t *err_count) {
    __shared__ uint64_t s_warp[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanBlock;
    uint64_t br, bv;
    if (blk_prefixed) {
        br = blk_rows[blockIdx.x];
        bv = blk_vars[blockIdx.x];
    } else {
        uint64_t pr = 0, pv = 0;
        for (uint32_t b = threadIdx.x; b < blockIdx.x; b += kThreads) { pr += blk_rows[b]; pv += blk_vars[b]; }
        block_exclusive_scan(pr, s_warp, br);
        block_exclusive_scan(pv, s_warp, bv);
    }
    uint64_t r[4], v[4], sr = 0, sv = 0;
    for (int k = 0; k < 4; k++) {
        uint64_t i = base + threadIdx.x * 4 + k;
        r[k] = i < n ? rows[i] : 0;
        v[k] = i < n ? vars[i] : 0;
        sr += r[k];
        sv += v[k];
    }
    uint64_t tr, tv;
    uint64_t er = br + block_exclusive_scan(sr, s_warp, tr);
    uint64_t ev = bv + block_exclusive_scan(sv, s_warp, tv);
    for (int k = 0; k < 4; k++) {
        uint64_t i = base + threadIdx.x * 4 + k;
        if (i < n) { row_off[i] = er; var_off[i] = ev; }
        er += r[k];
        ev += v[k];
        if (i + 1 == n) {
            row_off[n] = er;
            var_off[n] = ev;
            host->n_gates = er;
            host->n_vars = ev;
            host->errs = err_count ? *err_count : 0u;
            if (err_count) *err_count = 0;
            __threadfence_system();
        }
    }
}

}  // namespace pg
