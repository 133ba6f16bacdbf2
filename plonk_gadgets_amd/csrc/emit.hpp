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
};

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

__device__ __forceinline__ Fr lds_fr(const uint4 *table, uint32_t id) {
    FrVec t;
    t.v[0] = table[2 * id];
    t.v[1] = table[2 * id + 1];
    return t.f;
}

__device__ __forceinline__ void fill_common_table(uint4 *table, const uint4 *pow2, uint32_t tid) {
    for (uint32_t e = tid; e < kTableEntries; e += kThreads) {
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
// many of them) by bisection.
template <bool SMALL_ITEMS>
__device__ __forceinline__ uint32_t find_item(const uint32_t *off, uint32_t Wt, uint32_t r, uint32_t it) {
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
//   void item(A, O, item, table, rec)    per-item arithmetic -> rec (+ per-item outputs)
//   void selectors(A, rec, j, table, h, uint4 out[5])
//   void wires(A, O, rec, item, item_var_base, j, uint64_t out[3])
//   Fr   var_value(A, rec, table, k)
//   bool is_inv_slot(A, rec, k)          item-variable k holds an inverse (written by the pre-pass, not here)
//   int  kInv; Fr inv_element(A, item, e); uint4 *inv_slot(A, O, item, e)   the pre-pass's view (invert.hpp)
#ifndef PG_EMIT_WAVES_PER_SIMD
#define PG_EMIT_WAVES_PER_SIMD 1  // __launch_bounds__ second argument (waves per SIMD the register allocator must allow)
#endif
// Nothing this kernel writes depends on a field inversion: the variables that hold inverses (z of maybe_equal, inv of
// is_non_zero) are written, at their final slots, by the inversion pre-pass (invert.hpp), which runs concurrently on
// the engine's side stream; the variable sweep here skips exactly those slots (GD::is_inv_slot).
//
// kStructureOnly: selectors and wire indices alone -- for a gadget whose rows do not read the item record they are a
// function of the public inputs and the numbering, not of the witnesses, so a rank can regenerate another rank's rows
// instead of receiving them (distributed.VariablesOnlyPipeline); the item phase and the variable sweep are compiled out.
template <class GD, bool kStructureOnly = false>
__global__ __launch_bounds__(kThreads, PG_EMIT_WAVES_PER_SIMD) void emit_kernel(const typename GD::Args A, const EmitOut O) {
    static_assert(!kStructureOnly || !(GD::kRagged || GD::kRecInRows), "rows of this gadget depend on its inputs");
    constexpr int W = GD::W;
    __shared__ uint4 s_table[kTableEntries * 2];
    __shared__ typename GD::ItemRec s_item[W];
    __shared__ uint32_t s_roff[W + 1], s_voff[W + 1];

    const uint32_t tid = threadIdx.x;
    {
        const uint4 *p2 = nullptr;
        if constexpr (GD::kUsePow2) p2 = GD::pow2(A);
        fill_common_table(s_table, p2, tid);
    }
    GD::fill_table(A, s_table, tid);
    __syncthreads();

    for (uint32_t t = blockIdx.x; t < O.tiles; t += gridDim.x) {
#if defined(PG_XCD_REMAP)
        // workgroups b and b + 8 share an XCD (round-robin dispatch): give each XCD one contiguous eighth of the tiles
        // (bijective for any tile count; speed only, nothing depends on the placement)
        const uint32_t q8 = O.tiles >> 3, r8 = O.tiles & 7, x = t & 7, k = t >> 3;
        const uint32_t tile = x * q8 + (x < r8 ? x : r8) + k;
#else
        const uint32_t tile = t;
#endif
        const uint64_t w0 = (uint64_t)tile * W;
        const uint32_t Wt = (uint32_t)((O.batch - w0) < (uint64_t)W ? (O.batch - w0) : (uint64_t)W);
        uint64_t row0, var0;  // tile's first row / variable relative to the call
        uint32_t G = 0, V = 0;
        if constexpr (GD::kRagged) {
            row0 = O.row_off[w0];
            var0 = O.var_off[w0];
            // Wt + 1 entries: W may equal the block size, so stride over them
            for (uint32_t i = tid; i <= Wt; i += kThreads) {
                s_roff[i] = (uint32_t)(O.row_off[w0 + i] - row0);
                s_voff[i] = (uint32_t)(O.var_off[w0 + i] - var0);
            }
        } else {
            G = GD::rows_per_item(A);
            V = GD::vars_per_item(A);
            row0 = w0 * G;
            var0 = w0 * V;
        }

        // ---- item phase: one lane per item --------------------------------
        if constexpr (!kStructureOnly) {
            if (tid < Wt) GD::item(A, O, w0 + tid, s_table, s_item[tid]);
        }
        if constexpr (GD::kRagged || GD::kRecInRows) __syncthreads();

        const uint32_t total_rows = GD::kRagged ? s_roff[Wt] : Wt * G;
        const uint32_t total_vars = GD::kRagged ? s_voff[Wt] : Wt * V;

        // ---- selector sweep: 16 B per lane, 128 rows x 5 columns per pass ---
        {
            const uint32_t total = total_rows * 2;
            const uint32_t h = tid & 1;
            uint32_t it = 0, j = tid >> 1;
            if constexpr (!GD::kRagged) { it = j / G; j -= it * G; }
            for (uint32_t idx = tid; idx < total; idx += kThreads) {
                if constexpr (GD::kRagged) {
                    const uint32_t r = idx >> 1;
                    it = find_item<GD::W >= 64>(s_roff, Wt, r, it);
                    j = r - s_roff[it];
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
            uint32_t it = 0;
            for (uint32_t p = tid; p < pairs; p += kThreads) {
                const int64_t r0 = (int64_t)2 * p - shift;
                uint64_t val[2];
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int64_t r = r0 + k;
                    uint32_t rr = r < 0 ? 0u : (uint32_t)r;
                    if (rr >= total_rows) rr = total_rows - 1;
                    uint32_t j, vo;
                    if constexpr (GD::kRagged) {
                        it = find_item<GD::W >= 64>(s_roff, Wt, rr, it);
                        j = rr - s_roff[it];
                        vo = s_voff[it];
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

        if constexpr (!(GD::kRagged || GD::kRecInRows)) __syncthreads();  // item records visible

        // ---- variable sweep: one scalar (2 x 16 B) per lane ----------------
        if constexpr (!kStructureOnly) {
            uint32_t it = 0, k = tid;
            if constexpr (!GD::kRagged) { it = k / V; k -= it * V; }
            for (uint32_t s = tid; s < total_vars; s += kThreads) {
                if constexpr (GD::kRagged) {
                    it = find_item<GD::W >= 64>(s_voff, Wt, s, it);
                    k = s - s_voff[it];
                }
                if (!GD::is_inv_slot(A, s_item[it], k)) {  // inverse slots belong to the pre-pass
                    FrVec val;
                    val.f = GD::var_value(A, s_item[it], s_table, k);
                    uint4 *dst = O.vars + (var0 + s) * 2;
                    store16(dst, val.v[0]);
                    store16(dst + 1, val.v[1]);
                }
                if constexpr (!GD::kRagged) {
                    k += kThreads;
                    if (k >= V) { const uint32_t d = k / V; it += d; k -= d * V; }
                }
            }
        }
        __syncthreads();  // records and offsets are rewritten by the next tile
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

// the same fill as short-lived workgroups: one 16 KiB block per workgroup (four 16-byte stores per lane), no loop --
// at any moment the resident workgroups cover one moving window of a few tens of MiB
__global__ __launch_bounds__(kThreads) void fill_oneshot_kernel(uint4 *dst, uint64_t n16, uint64_t pattern) {
    const uint4 v = make_uint4((uint32_t)pattern, (uint32_t)(pattern >> 32), (uint32_t)~pattern, (uint32_t)(~pattern >> 32));
    const uint64_t base = (uint64_t)blockIdx.x * (4 * kThreads) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (base + k * kThreads < n16) store16(dst + base + k * kThreads, v);
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
// counts[i] (rows, vars of item i) -> off[i], off[batch] = total.  Three small
// kernels: per-block sums, scan of the block sums (one block), final scan.
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

__global__ __launch_bounds__(kThreads) void scan_block_sums_kernel(const uint32_t *rows, const uint32_t *vars, uint64_t n,
                                                                  uint64_t *blk_rows, uint64_t *blk_vars) {
    __shared__ uint64_t s_warp[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanBlock;
    uint64_t r = 0, v = 0;
    for (int k = 0; k < 4; k++) {
        uint64_t i = base + threadIdx.x * 4 + k;
        if (i < n) { r += rows[i]; v += vars[i]; }
    }
    uint64_t tr, tv;
    block_exclusive_scan(r, s_warp, tr);
    block_exclusive_scan(v, s_warp, tv);
    if (threadIdx.x == 0) { blk_rows[blockIdx.x] = tr; blk_vars[blockIdx.x] = tv; }
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

__global__ __launch_bounds__(kThreads) void scan_final_kernel(const uint32_t *rows, const uint32_t *vars, uint64_t n,
                                                             const uint64_t *blk_rows, const uint64_t *blk_vars,
                                                             uint64_t *row_off, uint64_t *var_off) {
    __shared__ uint64_t s_warp[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanBlock;
    uint64_t r[4], v[4], sr = 0, sv = 0;
    for (int k = 0; k < 4; k++) {
        uint64_t i = base + threadIdx.x * 4 + k;
        r[k] = i < n ? rows[i] : 0;
        v[k] = i < n ? vars[i] : 0;
        sr += r[k];
        sv += v[k];
    }
    uint64_t tr, tv;
    uint64_t er = blk_rows[blockIdx.x] + block_exclusive_scan(sr, s_warp, tr);
    uint64_t ev = blk_vars[blockIdx.x] + block_exclusive_scan(sv, s_warp, tv);
    for (int k = 0; k < 4; k++) {
        uint64_t i = base + threadIdx.x * 4 + k;
        if (i < n) { row_off[i] = er; var_off[i] = ev; }
        er += r[k];
        ev += v[k];
        if (i + 1 == n) { row_off[n] = er; var_off[n] = ev; }
    }
}

}  // namespace pg
