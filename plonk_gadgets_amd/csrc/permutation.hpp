// permutation.hpp -- SURVEY section 8f2: the permutation bookkeeping of the composer, on the device.
//
// dusk-plonk records, per gate, the four wire positions of the row under their Variables
// (perm.add_variables_to_map) and at preprocess time turns every Variable's position list -- in recording order:
// gate by gate, left/right/output/fourth inside a gate -- into one cycle of the copy permutation
// (compute_sigma_permutations).  Recording per call is pointless on a GPU; the same sigma is derived at the end from
// the wire columns alone, in one pass over them and almost without sorting anything:
//
//   items   a batched gadget call leaves `batch` items of L rows / V Variables each (PermSeg).  One workgroup per item
//           reads the item's 3L wires once (perm_item_kernel) and
//             - links every position whose Variable belongs to the item: counting sort of the positions by Variable
//               in LDS, successors staged in LDS, coalesced store of sigma;
//             - links the cycle of zero_var, which sits on the fourth wire of (nearly) every row, by looking at the
//               following row(s);
//             - appends what is left (references to Variables created elsewhere, live fourth wires) to the sparse list;
//   gaps    rows appended by single composer calls: zero chain as above, everything else to the sparse list
//           (perm_gap_kernel);
//   sparse  the list -- key = Variable << 33 | position -- is sorted (rocprim::radix_sort_keys, the only library call
//           on this path) and neighbours with one Variable are linked (perm_sparse_link_kernel).  A Variable can only
//           be referenced after it was created, so the sparse positions of an item's Variable all come after the
//           item's rows: its cycle is local positions, then sparse positions, spliced by perm_splice_kernel.
//
// Positions are ordered p = 4*gate + wire (== recording order); sigma is encoded wire * padded_n + gate.
#pragma once

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "composer.hpp"
#include "range_gadgets.hpp"

namespace pg {

// footprint of a batched call: `items` items, each owning a run of rows and a run of Variables it created itself.
// Uniform calls: L rows / V Variables per item.  Ragged calls (per-item public bounds, is_non_zero items that stop at
// their error): row_off / var_off are the call's exclusive prefix sums (items + 1 entries, relative to the bases) and
// L, V the largest item.  `group` consecutive items are linked by one workgroup as if they were one item.
struct PermSeg {
    uint64_t gate_base, gate_end, var_base, var_end;
    uint32_t L, V;
    uint64_t items;
    const uint64_t *row_off, *var_off;
    uint32_t group;
    // The call's wires in closed form (0: not known -- read the wire columns): the ladder gadgets' rows reference the item's own
    // Variables at offsets that are a function of the row and the ladder length alone (range_gadgets.hpp, bound_wire_offsets:
    // what the emitter wrote them from), so whoever needs the Variables of a row -- the wire-value columns of
    // pg_composer_materialize -- can compute them instead of reading 24 bytes per row back.
    uint32_t wire_kind, wire_n;
    // perm_ladder_kernel: the segment's slots on the sparse list -- ladder_foreign_per_item(wire_kind) per item, in closed form (set by
    // pg_composer_permutation for the pass it launches)
    uint64_t sparse_base;
    // Rows behind every item's own that hold its RESULT Variable on all three wires (constrain_to_constant / boolean_gate on the result:
    // the loop of the reference's tests, recorded call by call and flushed as one launch -- capi_composer.inc, flush): they count as
    // the item's rows (L includes them), their wires and their place in the result's cycle are closed forms like the others.  Kinds that
    // allocate their witness only.
    uint32_t tail;
    // perm_ladder_kernel<true> (per-item bounds): the item that holds the first row of every piece of kPermLadderRows rows (set by
    // pg_composer_permutation for the pass it launches; perm_piece_items_kernel fills it from the call's prefix sums)
    const uint32_t *piece_item;
};
enum : uint32_t { WIRES_UNKNOWN = 0, WIRES_RANGE_CHECK = 1, WIRES_MAX_BOUND = 2, WIRES_RANGE_CHECK_ALLOCATED = 3, WIRES_MAX_BOUND_ALLOCATED = 4,
                  WIRES_DECOMPOSITION = 5,
                  WIRES_MIX = 6,    // the fused scalar mix (ten rows, fifteen Variables: ScalarMixGD::row; eight / thirteen where v = 0)
                  // the small gadgets on Variables from elsewhere (scalar_gadgets.hpp) and the gate batches (composer.hpp): SegTemplate below
                  WIRES_SELECT_ZERO = 7, WIRES_SELECT_ONE = 8, WIRES_MAYBE_EQUAL = 9, WIRES_IS_NON_ZERO = 10, WIRES_GATE_OUT = 11,
                  WIRES_GATE_ROWS = 12, WIRES_KINDS = 13 };
constexpr uint32_t kZeroWire = 0xfffffffeu;  // "zero_var" (the composer's Variable with value 0: is_non_zero's first row has it)
// offsets (from the item's first own Variable) of the three wires of item-row j; kWitnessWire: the witness, which is the item's
// first Variable for the kinds that allocate it and a Variable from elsewhere (read it from the wire column) for the others
__device__ __forceinline__ void seg_wire_offsets(uint32_t kind, uint32_t n, uint32_t j, uint32_t off[3], uint32_t tail = 0) {
    const uint32_t x0 = kind == WIRES_RANGE_CHECK || kind == WIRES_MAX_BOUND ? 1u : 0u;
    if (tail && kind >= WIRES_RANGE_CHECK && kind <= WIRES_MAX_BOUND_ALLOCATED) {
        // (PermSeg::tail) the rows behind the gadget's own: its result -- the item's last Variable -- three times
        const bool rc = kind == WIRES_RANGE_CHECK || kind == WIRES_RANGE_CHECK_ALLOCATED;
        const uint32_t own = rc ? 4 * n + 11 : 2 * n + 5, res = x0 + (rc ? 2 * n + 522 : n + 260);
        if (j >= own) {
            off[0] = off[1] = off[2] = res;
            return;
        }
    }
    if (kind == WIRES_MIX) {
        // ScalarMixGD::row of an item whose v is not 0, as a table: Variables [v y s a b | va inv one | one' sy oms out | u z yeq] = 0 .. 14,
        // 15 = zero_var; four bits per wire, rows 0 .. 4 in the first word and 5 .. 9 in the second (a RowOut here cost the caller a
        // scratch array).  tests/test_gpu_composer.py compares what this yields with the wires the emitter wrote, row for row.
        //   0: (v, va, 0)   1: (one, one, one)   2: (v, inv, one)   3: (one', one', one')   4: (y, s, sy)
        //   5: (one', s, oms)   6: (sy, oms, out)   7: (a, b, u)   8: (z, u, yeq)   9: (yeq, u, u)
        constexpr uint64_t lo = 0xF50ull | 0x777ull << 12 | 0x760ull << 24 | 0x888ull << 36 | 0x921ull << 48;
        constexpr uint64_t hi = 0xA28ull | 0xBA9ull << 12 | 0xC43ull << 24 | 0xECDull << 36 | 0xCCEull << 48;
        // n != 0: the SHORT item (v = 0: is_non_zero stopped after its first row, scalar.rs:73-80) -- eight rows, thirteen Variables
        // [v y s a b | va | one' sy oms out | u z yeq] = 0 .. 12:
        //   0: (v, va, 0)   1: (one', one', one')   2: (y, s, sy)   3: (one', s, oms)   4: (sy, oms, out)   5: (a, b, u)   6: (z, u, yeq)   7: (yeq, u, u)
        constexpr uint64_t slo = 0xF50ull | 0x666ull << 12 | 0x721ull << 24 | 0x826ull << 36 | 0x987ull << 48;
        constexpr uint64_t shi = 0xA43ull | 0xCABull << 12 | 0xAACull << 24;
        const uint32_t row = (uint32_t)((j < 5 ? (n ? slo : lo) : (n ? shi : hi)) >> (12 * (j < 5 ? j : j - 5))) & 0xFFFu;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const uint32_t o = row >> (4 * c) & 15u;
            off[c] = o == 15u ? kZeroWire : o;
        }
        return;
    }
    if (kind == WIRES_RANGE_CHECK || kind == WIRES_RANGE_CHECK_ALLOCATED) {  // RangeCheckGD::wires
        const uint32_t L = 2 * n + 5, VB = n + 261;
        if (j == 2 * L) {
            off[0] = x0 + 260 + n; off[1] = x0 + VB + 260 + n; off[2] = x0 + 2 * VB;
        } else {
            const bool is_min = j >= L;
            bound_wire_offsets(is_min ? j - L : j, n, is_min ? x0 + VB : x0, kWitnessWire, off);
        }
    } else if (kind == WIRES_DECOMPOSITION) {  // DecompositionGD::wires
        bound_wire_offsets(j + 1, n, kWitnessWire, kWitnessWire, off);
    } else {  // MaxBoundGD<false>::wires
        bound_wire_offsets(j, n, x0, kWitnessWire, off);
    }
    if (x0)
        for (int c = 0; c < 3; c++) off[c] = off[c] == kWitnessWire ? 0u : off[c];
}
__device__ __forceinline__ uint64_t perm_rows_before(const PermSeg &s, uint64_t i) { return s.row_off ? s.row_off[i] : i * s.L; }
__device__ __forceinline__ uint64_t perm_vars_before(const PermSeg &s, uint64_t i) { return s.var_off ? s.var_off[i] : i * s.V; }
// item owning Variable v of the segment
__device__ __forceinline__ uint64_t perm_item_of_var(const PermSeg &s, uint64_t v) {
    const uint64_t rel = v - s.var_base;
    if (!s.var_off) return rel / s.V;
    uint64_t lo = 0, hi = s.items;  // last i with var_off[i] <= rel
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (s.var_off[mid] <= rel) lo = mid; else hi = mid;
    }
    return lo;
}

struct PermCtx {
    ComposerCols C;
    uint64_t n, padded_n, zero_var;
    const PermSeg *segs;
    uint32_t n_segs;
    const FourthWire *fw;
    uint32_t n_fw;
    uint64_t fw_lo, fw_span;  // live fourth wires only on gates [fw_lo, fw_lo + fw_span]
    uint32_t pos_bits;        // bits of a position 4 * gate + wire; sparse key = Variable << pos_bits | position
    uint64_t hole_key;        // every sorted bit set (no Variable has that number): a reserved slot of the list that holds nothing
};

constexpr uint32_t kPermIters = 16;                               // gates per thread of the gap kernel
constexpr uint64_t kPermChunk = (uint64_t)kThreads * kPermIters;  // gates per workgroup of the gap kernel
constexpr uint32_t kPermRowsPerThread = 4;  // item kernel: rows loaded per thread before any is processed (5 spilled at the 72 registers seven workgroups per CU leave)
constexpr uint32_t kPermLocalLdsLimit = 64 * 1024 - 256;
constexpr uint32_t kPermNone = 0xFFFF, kPermDone = 0xFFFE;

__device__ __forceinline__ uint64_t perm_encode(uint64_t gate, uint32_t wire, uint64_t padded_n) { return wire * padded_n + gate; }
__device__ __forceinline__ uint64_t perm_encode_pos(uint64_t p, uint64_t padded_n) { return (p & 3) * padded_n + (p >> 2); }

__device__ __forceinline__ uint64_t perm_fourth_var(const PermCtx &X, uint64_t g) {
    if (g - X.fw_lo > X.fw_span) return X.zero_var;
    for (uint32_t k = 0; k < X.n_fw; k++)
        if (X.fw[k].gate == g) return X.fw[k].w_4;
    return X.zero_var;
}

__device__ __forceinline__ uint32_t perm_zero_mask(const PermCtx &X, uint64_t g) {
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (X.C.w[k][g] == X.zero_var) m |= 1u << k;
    if (perm_fourth_var(X, g) == X.zero_var) m |= 8u;
    return m;
}

// first position of zero_var in gate g or later, wrapping past the end (the caller holds one, so there is one)
__device__ uint64_t perm_next_zero_from(const PermCtx &X, uint64_t g) {
    for (uint64_t tries = 0; tries <= X.n; tries++, g++) {
        if (g >= X.n) g = 0;
        const uint32_t m = perm_zero_mask(X, g);
        if (m) return perm_encode(g, (uint32_t)__ffs((int)m) - 1, X.padded_n);
    }
    return 0;
}

__device__ __forceinline__ int perm_home_seg(const PermCtx &X, uint64_t var) {
    uint32_t lo = 0, hi = X.n_segs;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (X.segs[mid].var_end <= var) lo = mid + 1; else hi = mid;
    }
    return lo < X.n_segs && X.segs[lo].var_base <= var ? (int)lo : -1;
}

struct PermSparse {
    uint64_t *keys;               // Variable << pos_bits | position, in any order
    unsigned long long *count;    // entries wanted so far (may pass cap: the host then grows the list and runs again); count[1]: how many
                                  // of them are holes (hole_key: they sort to the end and are not looked at)
    uint64_t cap;
};

__global__ void perm_count_init_kernel(unsigned long long *count, unsigned long long reserved) {
    count[0] = reserved;  // (slots handed out in closed form, perm_ladder_kernel; the counter continues behind them)
    count[1] = 0;
}

__device__ __forceinline__ void perm_sparse_put(const PermCtx &X, const PermSparse &Q, uint64_t at, uint64_t var, uint64_t gate,
                                                uint32_t wire) {
    if (at < Q.cap) Q.keys[at] = var << X.pos_bits | (4 * gate + wire);
}

// rows [g_begin, g_end) appended by single composer calls: zero chain + sparse list
__global__ __launch_bounds__(kThreads) void perm_gap_kernel(const PermCtx X, uint64_t g_begin, uint64_t g_end, const PermSparse Q,
                                                           uint64_t *sigma) {
    __shared__ uint64_t s_warp[4];
    __shared__ uint64_t s_base;
    for (uint32_t it = 0; it < kPermIters; it++) {
        const uint64_t g = g_begin + (uint64_t)blockIdx.x * kPermChunk + (uint64_t)it * kThreads + threadIdx.x;
        uint64_t var[4];
        uint32_t zero = 0, sparse = 0;
        if (g < g_end) {
#pragma unroll
            for (int w = 0; w < 3; w++) var[w] = X.C.w[w][g];
            var[3] = perm_fourth_var(X, g);
#pragma unroll
            for (uint32_t w = 0; w < 4; w++) {
                if (var[w] == X.zero_var) zero |= 1u << w; else sparse |= 1u << w;
            }
            if (zero) {
                uint64_t succ = perm_next_zero_from(X, g + 1);
                for (uint32_t m = zero; m;) {  // wires from the highest down: each links to the one found before
                    const uint32_t w = 31 - __clz((int)m);
                    m &= ~(1u << w);
                    sigma[perm_encode(g, w, X.padded_n)] = succ;
                    succ = perm_encode(g, w, X.padded_n);
                }
            }
        }
        uint64_t tot;
        uint64_t at = block_exclusive_scan(__popc(sparse), s_warp, tot);
        if (threadIdx.x == 0) s_base = tot ? atomicAdd(Q.count, (unsigned long long)tot) : 0;
        __syncthreads();
        at += s_base;
        for (uint32_t w = 0; w < 4; w++)
            if (sparse >> w & 1) perm_sparse_put(X, Q, at++, var[w], g, w);
        __syncthreads();
    }
}

// one workgroup per item (or group of small items) of a batched segment.  Dynamic LDS, kept small because the kernel
// lives on occupancy (halving the resident workgroups costs 1.7x): cnt[V] (u32, LDS atomics), off[V+1] (u16),
// lw[3L] (u16: local Variable id per position, later reused for the successors of those positions), pos[3L] (u16),
// sig4[L] (u16: successors of the fourth-wire positions), zm[L] (u8: wires holding zero_var | foreign wires << 4)
__host__ __device__ inline uint32_t perm_local_lds_bytes(uint32_t L, uint32_t V) {
    return 4 * V + 2 * (V + 2) + 2 * (3 * L + 3 * L + L) + L + 16;
}

__global__ __launch_bounds__(kThreads, 7) void perm_item_kernel(const PermCtx X, const PermSeg S, uint64_t groups, const PermSparse Q,
                                                            uint64_t *sigma) {
    extern __shared__ uint32_t perm_lds[];
    __shared__ uint64_t s_warp[4];
    __shared__ uint32_t s_foreign, s_rank;
    __shared__ unsigned long long s_base;
    const uint32_t Lmax = S.L * S.group, Vmax = S.V * S.group, tid = threadIdx.x;
    uint32_t *cnt = perm_lds;
    uint16_t *off = reinterpret_cast<uint16_t *>(cnt + Vmax), *lw = off + ((Vmax + 2) & ~1u), *pos = lw + 3 * Lmax, *sig4 = pos + 3 * Lmax;
    uint16_t *sig3 = lw;  // the ids are dead once the positions are scattered
    uint8_t *zm = reinterpret_cast<uint8_t *>(sig4 + Lmax);
    for (uint64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const uint64_t ib = grp * S.group, ie = ib + S.group < S.items ? ib + S.group : S.items;
        const uint64_t rb0 = perm_rows_before(S, ib), vb0 = perm_vars_before(S, ib);
        const uint64_t g0 = S.gate_base + rb0, v0 = S.var_base + vb0;
        const uint32_t L = (uint32_t)(perm_rows_before(S, ie) - rb0), V = (uint32_t)(perm_vars_before(S, ie) - vb0), n3 = 3 * L;
        for (uint32_t id = tid; id < V; id += kThreads) cnt[id] = 0;
        if (tid == 0) s_foreign = s_rank = 0;
        __syncthreads();
        for (uint32_t rb = 0; rb < L; rb += kPermRowsPerThread * kThreads) {  // 12 loads in flight per thread
            uint64_t var[kPermRowsPerThread][3];
#pragma unroll
            for (uint32_t u = 0; u < kPermRowsPerThread; u++) {
                const uint32_t r = rb + u * kThreads + tid;
#pragma unroll
                for (uint32_t w = 0; w < 3; w++) var[u][w] = r < L ? X.C.w[w][g0 + r] : 0;
            }
#pragma unroll
            for (uint32_t u = 0; u < kPermRowsPerThread; u++) {
                const uint32_t r = rb + u * kThreads + tid;
                if (r >= L) continue;
                uint32_t mask = 0;  // bits 0..3: wire holds zero_var; bits 4..6: wire references a Variable created elsewhere
#pragma unroll
                for (uint32_t w = 0; w < 3; w++) {
                    const uint64_t rel = var[u][w] - v0;
                    uint32_t id = kPermNone;
                    if (var[u][w] == X.zero_var) mask |= 1u << w;
                    else if (rel < (uint64_t)V) atomicAdd(&cnt[id = (uint32_t)rel], 1u);
                    else mask |= 16u << w;
                    lw[3 * r + w] = (uint16_t)id;
                }
                if (mask >> 4) atomicAdd(&s_foreign, (uint32_t)__popc(mask >> 4));
                const uint64_t var4 = perm_fourth_var(X, g0 + r);
                if (var4 == X.zero_var) mask |= 8u;
                else perm_sparse_put(X, Q, atomicAdd(Q.count, 1ull), var4, g0 + r, 3);
                zm[r] = (uint8_t)mask;
            }
        }
        __syncthreads();
        if (s_foreign) {  // one reservation in the sparse list per workgroup
            if (tid == 0) s_base = atomicAdd(Q.count, (unsigned long long)s_foreign);
            __syncthreads();
            for (uint32_t r = tid; r < L; r += kThreads)
                for (uint32_t m = zm[r] >> 4; m; m &= m - 1) {
                    const uint32_t w = (uint32_t)__ffs((int)m) - 1;
                    perm_sparse_put(X, Q, s_base + atomicAdd(&s_rank, 1u), X.C.w[w][g0 + r], g0 + r, w);
                }
        }
        {  // off = exclusive scan of cnt; cnt becomes the scatter cursor
            const uint32_t per = (V + kThreads - 1) / kThreads, b = tid * per, e = b + per < V ? b + per : V;
            uint64_t sum = 0, tot;
            for (uint32_t id = b; id < e; id++) sum += cnt[id];
            uint32_t run = (uint32_t)block_exclusive_scan(sum, s_warp, tot);
            for (uint32_t id = b; id < e; id++) {
                off[id] = (uint16_t)run;
                run += cnt[id];
                cnt[id] = 0;
            }
            if (tid == 0) off[V] = (uint16_t)tot;
        }
        __syncthreads();
        for (uint32_t j = tid; j < n3; j += kThreads) {
            const uint32_t id = lw[j];
            if (id != kPermNone) pos[off[id] + atomicAdd(&cnt[id], 1u)] = (uint16_t)j;
        }
        __syncthreads();  // lw is dead from here on: its storage holds the successors (sig3)
        for (uint32_t id = tid; id < V; id += kThreads) {
            const uint32_t b = off[id], k = off[id + 1] - b;
            for (uint32_t i = 1; i < k; i++) {  // the atomics scattered in any order: restore position order
                const uint16_t x = pos[b + i];
                uint32_t h = i;
                while (h > 0 && pos[b + h - 1] > x) {
                    pos[b + h] = pos[b + h - 1];
                    h--;
                }
                pos[b + h] = x;
            }
            for (uint32_t i = 0; i < k; i++) {  // successors are stored as 4 r + w
                const uint32_t z = pos[b + (i + 1 == k ? 0 : i + 1)];
                sig3[pos[b + i]] = (uint16_t)(z + z / 3);
            }
        }
        for (uint32_t r = tid; r < L; r += kThreads) {  // zero chain
            const uint32_t zero = zm[r] & 15u;
            if (!zero) continue;
            uint32_t nr = r + 1;
            while (nr < L && (zm[nr] & 15u) == 0) nr++;
            uint32_t succ = nr < L ? 4 * nr + (uint32_t)__ffs((int)(zm[nr] & 15u)) - 1 : kPermDone;
            for (uint32_t m = zero; m;) {
                const uint32_t w = 31 - __clz((int)m);
                m &= ~(1u << w);
                if (succ == kPermDone) sigma[perm_encode(g0 + r, w, X.padded_n)] = perm_next_zero_from(X, g0 + L);
                if (w == 3) sig4[r] = (uint16_t)succ; else sig3[3 * r + w] = (uint16_t)succ;
                succ = 4 * r + w;
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t w = 0; w < 4; w++)
            for (uint32_t r = tid; r < L; r += kThreads) {
                const uint32_t m = zm[r];
                if (w < 3 ? (m >> (4 + w) & 1u) : !(m & 8u)) continue;  // that position is on the sparse list
                const uint32_t s = w < 3 ? sig3[3 * r + w] : sig4[r];
                if (s < kPermDone) sigma[perm_encode(g0 + r, w, X.padded_n)] = perm_encode(g0 + (s >> 2), s & 3, X.padded_n);
            }
        __syncthreads();
    }
}

// ---- sigma of a ladder gadget's rows in closed form --------------------------------------------------------------------------
// For the uniform ladder gadgets (PermSeg::wire_kind) not only the wires of a row are a function of its place in its item
// (seg_wire_offsets) -- so are the CYCLES: which positions hold one Variable, in recording order.  A bound block
// (range_gadgets.hpp: rows jj = 0 .. 2n + 4 of max_bound / min_bound; Variables T, b_0.., A_0.., U, z, y):
//     T    (0, o) -> (2n+2, r)                                   b - x, then the right wire of  u = A_n - T
//     A_0  (1, l) -> (1, r) -> (1, o) -> (3, r)                  its constant row, then the first ladder step
//     b_i  (2+2i, l) -> (2+2i, r) -> (2+2i, o) -> (3+2i, l)      boolean row, then its ladder step
//     A_i  (1+2i, o) -> (3+2i, r)      (0 < i < n)               made by step i - 1, used by step i
//     A_n  (2n+1, o) -> (2n+2, l)
//     U    (2n+2, o) -> (2n+3, r) -> (2n+4, r) -> (2n+4, o)      z: (2n+3, l) alone      y: (2n+3, o) -> (2n+4, l) [-> the caller's row]
//     x    (0, l) -> (0, r)            the witness: the item's own first Variable, or one from elsewhere (the sparse list)
// and every position's successor is the next one of its line, the last one's the first.  The fourth wire of every such row holds
// zero_var: its successor is the fourth wire of the next row.  One lane per row, nothing read, 32 bytes written per row: the
// counting sort of perm_item_kernel (3.4 ms of the 4.8 a 270 M-row circuit took) is not needed for these rows: 1.9-2.4 ms by box, at
// the rate four lock-step streams 4.3 GB apart take (sigma's columns are padded_n entries apart); one column per pass -- the successors
// computed four times -- is slower (5.1 against 4.1 ms for the whole call: the arithmetic is not free).
// successors of the three positions of block-row jj: (j2[w], w2[w]) in block rows.  Returned mask: bit w = the position is not linked
// inside the block -- jj = 0: w = 0, 1 hold the witness x; jj = 2n + 2: w = 1 is the END of T's line; jj = 2n + 4: w = 0 the end of y's
__device__ __forceinline__ uint32_t ladder_block_row(uint32_t jj, uint32_t n, uint32_t j2[3], uint32_t w2[3]) {
    if (jj >= 2 && jj < 2 * n + 2) {  // the ladder: one classification, the rest is selects (odd and even rows alternate along a wave)
        const uint32_t i = (jj - 2) >> 1;
        const bool step = jj & 1, first = i == 0, last = i + 1 >= n;
        // boolean row (b_i, b_i, b_i): (jj, r), (jj, o), (jj + 1, l)   |   step (b_i, A_i, A_{i+1}): b_i back to its boolean row; A_0's line
        // ends here, A_i goes back to where it was made; A_{i+1} on to the next step's right wire, A_n to the left wire of u = A_n - T
        j2[0] = step ? jj - 1 : jj;
        w2[0] = step ? 0u : 1u;
        j2[1] = step ? (first ? 1u : jj - 2) : jj;
        w2[1] = step ? (first ? 0u : 2u) : 2u;
        j2[2] = step ? (last ? 2 * n + 2 : jj + 2) : jj + 1;
        w2[2] = step ? (last ? 0u : 1u) : 0u;
        return 0;
    }
    j2[0] = j2[1] = j2[2] = jj;
    w2[0] = 0; w2[1] = 1; w2[2] = 2;
    if (jj == 0) {  // (x, x, T)
        j2[2] = 2 * n + 2; w2[2] = 1;
        return 3;
    }
    if (jj == 1) {  // (A_0, A_0, A_0)
        w2[0] = 1; w2[1] = 2; j2[2] = 3; w2[2] = 1;
        return 0;
    }
    if (jj == 2 * n + 2) {  // (A_n, T, U)
        j2[0] = 2 * n + 1; w2[0] = 2; j2[2] = 2 * n + 3; w2[2] = 1;
        return 2;
    }
    if (jj == 2 * n + 3) {  // (z, U, y): z alone on its line
        j2[1] = 2 * n + 4; w2[1] = 1; j2[2] = 2 * n + 4; w2[2] = 0;
        return 0;
    }
    // jj == 2n + 4: (y, U, U)
    w2[1] = 2; j2[2] = 2 * n + 2; w2[2] = 2;
    return 1;
}
// successors of the three positions of item-row j of an item of kind `kind`, in item rows; returned mask: bit w = the position holds a
// Variable created elsewhere (the sparse list links it, or the zero chain if it is zero_var)
__device__ __forceinline__ uint32_t ladder_row(uint32_t kind, uint32_t n, uint32_t j, uint32_t j2[3], uint32_t w2[3], uint32_t tail = 0) {
    const uint32_t L = 2 * n + 5;
    // PermSeg::tail rows (result, result, result) behind the item's own: the result's line runs on through them in recording order
    // and closes from the last one back to where the result was made -- (2L, o) of range_check, (2n + 3, o) of max_bound
    const bool rc = kind == WIRES_RANGE_CHECK || kind == WIRES_RANGE_CHECK_ALLOCATED;
    const uint32_t own = rc ? 2 * L + 1 : L;
    if (tail && j >= own) {
        j2[0] = j; w2[0] = 1; j2[1] = j; w2[1] = 2;
        if (j + 1 < own + tail) { j2[2] = j + 1; w2[2] = 0; }
        else { j2[2] = rc ? 2 * L : 2 * n + 3; w2[2] = 2; }
        return 0;
    }
    if (kind == WIRES_RANGE_CHECK || kind == WIRES_RANGE_CHECK_ALLOCATED) {
        if (j == 2 * L) {  // (y1, y2, R): the end of either block's y line; R alone (or on to the first of the tail rows)
            j2[0] = 2 * n + 3; w2[0] = 2; j2[1] = L + 2 * n + 3; w2[1] = 2; j2[2] = tail ? j + 1 : j; w2[2] = tail ? 0 : 2;
            return 0;
        }
        const uint32_t blk = j >= L ? 1u : 0u, base = blk * L, jj = j - base;
        const uint32_t open = ladder_block_row(jj, n, j2, w2);
#pragma unroll
        for (int w = 0; w < 3; w++) j2[w] += base;
        if (jj == 0) {  // x: (0, l) -> (0, r) -> (L, l) -> (L, r) -> (0, l), unless it comes from elsewhere
            if (kind == WIRES_RANGE_CHECK_ALLOCATED) return 3;
            j2[0] = j; w2[0] = 1; j2[1] = blk ? 0 : L; w2[1] = 0;
        } else if (open == 2) { j2[1] = base; w2[1] = 2; }           // T's line closes
        else if (open == 1) { j2[0] = 2 * L; w2[0] = blk; }          // y goes on to the product row
        return 0;
    }
    if (kind == WIRES_DECOMPOSITION) {  // a block without its row 0, on a witness from elsewhere (T)
        const uint32_t open = ladder_block_row(j + 1, n, j2, w2);
#pragma unroll
        for (int w = 0; w < 3; w++) j2[w] -= 1;
        if (open == 2) return 2;
        if (open == 1) { j2[0] = 2 * n + 2; w2[0] = 2; }
        return 0;
    }
    const uint32_t open = ladder_block_row(j, n, j2, w2);  // max_bound
    if (j == 0) {
        if (kind == WIRES_MAX_BOUND_ALLOCATED) return 3;
        j2[0] = 0; w2[0] = 1; j2[1] = 0; w2[1] = 0;
    } else if (open == 2) { j2[1] = 0; w2[1] = 2; }
    else if (open == 1) { j2[0] = tail ? L : 2 * n + 3; w2[0] = tail ? 0 : 2; }  // y's line closes, or goes on to the first tail row
    return 0;
}

// wires (bits 0..2) of item-row j that hold a Variable from elsewhere: the witness of an `_allocated` call (x on both input wires of a
// bound block's first row), the witness of a decomposition (T, the right wire of u = A_n - T)
__device__ __forceinline__ uint32_t ladder_foreign_wires(uint32_t kind, uint32_t n, uint32_t j) {
    if (kind == WIRES_RANGE_CHECK_ALLOCATED) return j == 0 || j == 2 * n + 5 ? 3u : 0u;
    if (kind == WIRES_MAX_BOUND_ALLOCATED) return j == 0 ? 3u : 0u;
    if (kind == WIRES_DECOMPOSITION) return j == 2 * n + 1 ? 2u : 0u;
    return 0u;
}

// positions of an item that hold a witness allocated elsewhere, and the rank of (item-row j, wire w) among them in recording order:
// every item owns that many consecutive slots of the sparse list, so a lane knows where its entry goes without asking anybody
__host__ __device__ inline uint32_t ladder_foreign_per_item(uint32_t kind) {
    return kind == WIRES_RANGE_CHECK_ALLOCATED ? 4u : kind == WIRES_MAX_BOUND_ALLOCATED ? 2u : kind == WIRES_DECOMPOSITION ? 1u : 0u;
}
__device__ __forceinline__ uint32_t ladder_foreign_rank(uint32_t kind, uint32_t j, uint32_t w) {
    if (kind == WIRES_DECOMPOSITION) return 0u;
    return (j ? 2u : 0u) + w;  // rows 0 (and 2n + 5 of range_check): wires 0 and 1
}

#ifndef PG_PERM_LADDER_ROWS
#define PG_PERM_LADDER_ROWS 512
#endif
#ifndef PG_PERM_LADDER_LDS
#define PG_PERM_LADDER_LDS 39936
#endif
// Short-lived workgroups: ONE pass of 512 rows each (16 KiB written: 4 KiB of each of sigma's four columns), started in address order by
// the dispatcher, four resident per CU (an unused dynamic LDS allocation bounds them) -- the same finding as perm_identity_kernel's: what is
// under way at any moment should be a narrow window.  The whole call on a 270 M-row circuit, one process, same placement
// (tools/perm_variants.py): pieces of 8192 rows walked by long-lived workgroups 3.77 ms; 512 rows, full residency 3.45; 512 rows, 4 per CU
// 3.2; 3 per CU 3.55; 2 per CU 4.4 (too few waves for the arithmetic); 1024 rows, 4 per CU 3.26.
constexpr uint32_t kPermLadderRows = PG_PERM_LADDER_ROWS;  // rows per workgroup piece (a multiple of 2 * kThreads)
constexpr uint32_t kPermLadderLds = PG_PERM_LADDER_LDS;    // dynamic LDS per workgroup, unused: bounds how many are resident per CU
// RAGGED (max_bound with a bound per item, WIRES_MAX_BOUND: range.rs:82-113 with bound_i): an item's ladder length follows from its row
// count, L_i = 2 n_i + 5, and its place from the call's prefix sums -- the piece's first item from S.piece_item, the <= 58 items a piece
// of 512 rows can hold (an item has at least nine) from a window of the prefix sums in LDS that every lane searches (six steps).
// Before round 6 such segments went through perm_item_kernel's counting sort in LDS: 3.9 against 3.2 ms per 268 M rows.
constexpr uint32_t kPermRaggedWindow = 64;  // prefix sums per piece: kPermLadderRows / 9 + 2 items and one entry more, rounded up
// the item that holds row `first + p * kPermLadderRows` of the segment, for every piece p (one thread per item: an item is shorter than a piece)
__global__ __launch_bounds__(kThreads) void perm_piece_items_kernel(const PermSeg S, uint32_t rows_per_piece, uint32_t *piece_item) {
    const uint64_t d = S.gate_base & 1;  // pieces are counted from the even gate at or before the segment's first
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < S.items; i += (uint64_t)gridDim.x * kThreads) {
        const uint64_t r0 = S.row_off[i], r1 = S.row_off[i + 1];
        if (i == 0) piece_item[0] = 0;
        for (uint64_t p = (r0 + d + rows_per_piece - 1) / rows_per_piece; p * rows_per_piece < r1 + d; p++)
            if (p) piece_item[p] = (uint32_t)i;
    }
}

template <bool RAGGED>
__global__ __launch_bounds__(kThreads) void perm_ladder_kernel(const PermCtx X, const PermSeg S, const PermSparse Q, uint64_t *sigma) {
    extern __shared__ uint4 perm_ladder_pad[];
    __shared__ uint32_t s_off[RAGGED ? kPermRaggedWindow : 1];  // RAGGED: rows before items item0 .. of the piece, from item0's first
    static_assert(!RAGGED || kPermLadderRows / 9 + 3 <= kPermRaggedWindow, "a piece's items fit the window of prefix sums");
    const uint32_t lane = threadIdx.x & 63, kind = S.wire_kind;
    uint32_t n = S.wire_n;
    if (X.padded_n == 1) perm_ladder_pad[threadIdx.x] = make_uint4(0, 0, 0, 0);  // (keeps the allocation)
    // a lane takes the TWO gates 2k, 2k + 1 (16 bytes of each of sigma's four columns: one store each where sigma is 16-byte aligned)
    const uint64_t first = S.gate_base & ~1ull, total = S.gate_end - first;  // (rows counted from the even gate at or before the segment's first)
    const bool wide = (reinterpret_cast<uintptr_t>(sigma) & 15) == 0 && !(X.padded_n & 1);
    // wire * padded_n + gate: a shift when padded_n is a power of two (it is for whoever pads as dusk-plonk does) -- the successor's
    // wire is data, so the product cannot be hoisted, and a 64 x 64-bit multiplication per position is most of what a lane would do
    const uint32_t recip = (uint32_t)(((1ull << 32) + S.L - 1) / S.L);  // floor(x / L) = umulhi(x, ceil(2^32 / L)) for x < 2^16 / ... (below)
    const bool pow2 = (X.padded_n & (X.padded_n - 1)) == 0;
    const uint32_t sh = 63u - (uint32_t)__clzll((long long)X.padded_n);
    auto enc = [&](uint64_t gate, uint32_t wire) { return pow2 ? ((uint64_t)wire << sh) + gate : perm_encode(gate, wire, X.padded_n); };
    // which of wires 0..2 of gate g (item-row j) hold zero_var: only a witness from elsewhere can
    auto zero_wires = [&](uint64_t g, uint32_t j) {
        uint32_t z = 0;
        for (uint32_t m = ladder_foreign_wires(kind, n, j); m; m &= m - 1) {
            const uint32_t w = (uint32_t)__ffs((int)m) - 1;
            if (X.C.w[w][g] == X.zero_var) z |= 1u << w;
        }
        return z;
    };
    for (uint64_t base = (uint64_t)blockIdx.x * kPermLadderRows; base < total; base += (uint64_t)gridDim.x * kPermLadderRows) {
        // item and item-row of the piece's first gate (one wide division per piece; the rows' own are 32-bit)
        const uint64_t g_first = first + base, rel = g_first < S.gate_base ? 0 : g_first - S.gate_base;
        uint64_t item0, rows0 = 0;  // the piece's first item, the rows of the call before it
        if constexpr (RAGGED) {
            item0 = S.piece_item[base / kPermLadderRows];
            rows0 = S.row_off[item0];
            __syncthreads();  // (a workgroup's previous piece is read out)
            if (threadIdx.x < kPermRaggedWindow) {
                const uint64_t i = item0 + threadIdx.x < S.items ? item0 + threadIdx.x : S.items;
                s_off[threadIdx.x] = (uint32_t)(S.row_off[i] - rows0);
            }
            __syncthreads();
        } else {
            item0 = rel / S.L;
            rows0 = item0 * S.L;
        }
        const uint32_t j0 = (uint32_t)(rel - rows0), skip = (uint32_t)(g_first < S.gate_base ? S.gate_base - g_first : 0);
        for (uint32_t t = 2 * threadIdx.x; t < kPermLadderRows; t += 2 * kThreads) {  // (whole waves: the ballot below)
            uint64_t out[4][2], fitem[2] = {0, 0};
            uint32_t foreign[2] = {0, 0}, fslots[2] = {0, 0}, fj[2] = {0, 0};
            bool live[2];
#pragma unroll
            for (uint32_t h = 0; h < 2; h++) {
                const uint64_t g = g_first + t + h;
                live[h] = g >= S.gate_base && g < S.gate_end;
                const uint32_t tt = t + h - skip;  // rows past the piece's first row of the segment
                uint32_t q, j;
                uint64_t g_item;
#pragma unroll
                for (uint32_t w = 0; w < 4; w++) out[w][h] = 0;
                if constexpr (RAGGED) {
                    if (!live[h]) continue;
                    const uint32_t x = j0 + tt;
                    uint32_t lo = 0;  // the last entry of the window that is <= x
#pragma unroll
                    for (uint32_t step = kPermRaggedWindow / 2; step; step >>= 1)
                        if (s_off[lo + step] <= x) lo += step;
                    q = lo;
                    j = x - s_off[lo];
                    n = (s_off[lo + 1] - s_off[lo] - 5) >> 1;  // L_i = 2 n_i + 5 (range.rs:82-113: one bound block)
                    g_item = S.gate_base + rows0 + s_off[lo];
                } else {
                    q = __umulhi(j0 + tt, recip);
                    j = (j0 + tt) - q * S.L;  // (exact: j0 + tt < 2^14, L < 2^11)
                    g_item = S.gate_base + (item0 + q) * S.L;
                }
                if (!live[h]) continue;
                uint32_t j2[3], w2[3];
                const uint32_t fw = ladder_row(kind, n, j, j2, w2, S.tail), zw = fw ? zero_wires(g, j) : 0u;
                foreign[h] = fw & ~zw;
                fslots[h] = fw;
                fitem[h] = item0 + q;
                fj[h] = j;
#pragma unroll
                for (uint32_t w = 0; w < 3; w++) {
                    if (fw >> w & 1) {  // zero_var here: the next wire of this row that holds it (the fourth one at the latest)
                        const uint32_t later = (zw | 8u) & ~((2u << w) - 1);
                        out[w][h] = enc(g, (uint32_t)__ffs((int)later) - 1);
                    } else {
                        out[w][h] = enc(g_item + j2[w], w2[w]);
                    }
                }
                // the fourth wire holds zero_var: on to the first wire of the next row that does
                if (g + 1 < S.gate_end) {
                    const uint32_t jn = j + 1 == S.L ? 0 : j + 1;
                    const uint32_t zn = ladder_foreign_wires(kind, n, jn) ? zero_wires(g + 1, jn) : 0u;
                    out[3][h] = enc(g + 1, (uint32_t)__ffs((int)(zn | 8u)) - 1);
                } else {
                    out[3][h] = perm_next_zero_from(X, S.gate_end);
                }
            }
            const uint64_t g = g_first + t;
#pragma unroll
            for (uint32_t w = 0; w < 4; w++) {
                const bool skip0 = !live[0] || (foreign[0] >> w & 1), skip1 = !live[1] || (foreign[1] >> w & 1);
                uint64_t *dst = sigma + enc(g, w);
                if (wide && !skip0 && !skip1)
                    store16(reinterpret_cast<uint4 *>(dst), make_uint4((uint32_t)out[w][0], (uint32_t)(out[w][0] >> 32), (uint32_t)out[w][1],
                                                                      (uint32_t)(out[w][1] >> 32)));
                else {
                    if (!skip0) dst[0] = out[w][0];
                    if (!skip1) dst[1] = out[w][1];
                }
            }
            // the positions that hold a Variable from elsewhere go to the sparse list, each into its item's own slot (closed form: no
            // counter, no shuffle -- a reservation per wave cost an agent-scope atomic's round trip in a quarter of all wave passes, 6.4 ms
            // against 1.5 for 270 M rows); one that holds zero_var belongs to the zero chain and leaves a hole
            if (fslots[0] | fslots[1]) {
                const uint32_t F = ladder_foreign_per_item(kind);
#pragma unroll
                for (uint32_t h = 0; h < 2; h++)
                    for (uint32_t m = fslots[h]; m; m &= m - 1) {
                        const uint32_t w = (uint32_t)__ffs((int)m) - 1;
                        const uint64_t at = S.sparse_base + fitem[h] * F + ladder_foreign_rank(kind, fj[h], w);
                        if (foreign[h] >> w & 1) perm_sparse_put(X, Q, at, X.C.w[w][g + h], g + h, w);
                        else {
                            if (at < Q.cap) Q.keys[at] = X.hole_key;
                            atomicAdd(Q.count + 1, 1ull);
                        }
                    }
            }
        }
    }
}

// ---- sigma of the small gadgets' rows in closed form ---------------------------------------------------------------------------
// The scalar gadgets (/root/reference/src/scalar.rs) and the gate batches leave items of one to ten rows whose wires are, row by row,
// one of the item's OWN Variables (an offset from its first), one of the call's INPUT Variables (foreign: a Variable from elsewhere,
// read back from the wire column) or zero_var.  SegTemplate says which, per kind, for the full item and for the SHORT one that
// is_non_zero leaves when it stops at its error (scalar.rs:73-80: one row, one Variable -- in the fused mix eight rows, thirteen
// Variables).  Everything sigma needs follows from that table: an own Variable's positions are the table's entries that name it, in
// row-major order, so a position's successor is the next such entry (the first, from the last) -- derived once per workgroup into LDS;
// a foreign position goes to the sparse list at a slot that is a function of (item, rank of the position in its item); a zero_var
// position -- the table's, or a foreign Variable that IS zero_var -- links to the next one of its row, the fourth wire's to the first
// of the next row.  One lane per two gates, like perm_ladder_kernel; ragged segments (failing items) find a row's item in a window of
// the call's prefix sums.  This replaces perm_item_kernel's counting sort in LDS for every batched append that is not a ladder.
constexpr uint8_t kTmplForeign = 0xF0, kTmplZero = 0xFE, kTmplNone = 0xFF;  // wire codes (0 .. 14: the item's own Variable)
constexpr uint32_t kTmplRows = 10;
struct SegTemplate {
    uint8_t L[2], V[2];            // rows / Variables of a full item, of a short one (0: the kind has none)
    uint8_t wire[2][kTmplRows][3];
};
#define PG_T3(a, b, c) {a, b, c}
#define PG_TF0 0xF0
#define PG_TF1 0xF1
#define PG_TF2 0xF2
#define PG_TZ 0xFE
#define PG_TX PG_T3(0xFF, 0xFF, 0xFF)
constexpr SegTemplate kSegTemplates[WIRES_KINDS - WIRES_MIX] = {
    // WIRES_MIX: [v y s a b | va inv one | one' sy oms out | u z yeq] (ScalarMixGD::row); short: [v y s a b | va | one' sy oms out | u z yeq]
    {{10, 8}, {15, 13},
     {{PG_T3(0, 5, PG_TZ), PG_T3(7, 7, 7), PG_T3(0, 6, 7), PG_T3(8, 8, 8), PG_T3(1, 2, 9), PG_T3(8, 2, 10), PG_T3(9, 10, 11), PG_T3(3, 4, 12),
       PG_T3(13, 12, 14), PG_T3(14, 12, 12)},
      {PG_T3(0, 5, PG_TZ), PG_T3(6, 6, 6), PG_T3(1, 2, 7), PG_T3(6, 2, 8), PG_T3(7, 8, 9), PG_T3(3, 4, 10), PG_T3(11, 10, 12), PG_T3(12, 10, 10),
       PG_TX, PG_TX}}},
    // WIRES_SELECT_ZERO: (x, select, out)                                                         scalar.rs:21-27
    {{1, 0}, {1, 0}, {{PG_T3(PG_TF0, PG_TF1, 0), PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX}, {PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX}}},
    // WIRES_SELECT_ONE: [one sy oms out]: (one,one,one) (y,sel,sy) (one,sel,oms) (sy,oms,out)      scalar.rs:36-59
    {{4, 0}, {4, 0}, {{PG_T3(0, 0, 0), PG_T3(PG_TF0, PG_TF1, 1), PG_T3(0, PG_TF1, 2), PG_T3(1, 2, 3), PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX}, {PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX}}},
    // WIRES_MAYBE_EQUAL: [u z y]: (a,b,u) (z,u,y) (y,u,u)                                          scalar.rs:105-140
    {{3, 0}, {3, 0}, {{PG_T3(PG_TF0, PG_TF1, 0), PG_T3(1, 0, 2), PG_T3(2, 0, 0), PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX}, {PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX}}},
    // WIRES_IS_NON_ZERO: [va inv one]: (var,va,zero) (one,one,one) (var,inv,one); short: [va]: (var,va,zero)   scalar.rs:63-97
    {{3, 1}, {3, 1}, {{PG_T3(PG_TF0, 0, PG_TZ), PG_T3(2, 2, 2), PG_T3(PG_TF0, 1, 2), PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX}, {PG_T3(PG_TF0, 0, PG_TZ), PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX}}},
    // WIRES_GATE_OUT: add / mul over arrays of Variables: (a, b, out)
    {{1, 0}, {1, 0}, {{PG_T3(PG_TF0, PG_TF1, 0), PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX}, {PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX}}},
    // WIRES_GATE_ROWS: poly_gate / constrain_to_constant / boolean_gate over arrays: (a, b, c), no Variable
    {{1, 0}, {0, 0}, {{PG_T3(PG_TF0, PG_TF1, PG_TF2), PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX}, {PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX, PG_TX}}},
};
#undef PG_T3
#undef PG_TF0
#undef PG_TF1
#undef PG_TF2
#undef PG_TZ
#undef PG_TX
// positions of a full item that hold a Variable from elsewhere (= its slots on the sparse list; a short item leaves the rest as holes)
__host__ __device__ inline uint32_t template_foreign_per_item(uint32_t kind) {
    return kind == WIRES_SELECT_ZERO || kind == WIRES_MAYBE_EQUAL || kind == WIRES_IS_NON_ZERO || kind == WIRES_GATE_OUT ? 2u
           : kind == WIRES_SELECT_ONE || kind == WIRES_GATE_ROWS ? 3u : 0u;
}
__host__ __device__ inline bool is_template_kind(uint32_t kind) { return kind >= WIRES_MIX && kind < WIRES_KINDS; }

// What the kernels take: the table and what follows from it, one 64-bit word per (shape, row) -- 16 bits per wire:
//   bits 0-1  type   0 none, 1 an own Variable, 2 a Variable from elsewhere, 3 zero_var
//   bits 2-5  own    the Variable's offset from the item's first                                (type 1)
//   bits 6-11 succ   the next position of that Variable in its item, 4 * row + wire             (type 1)
//   bits 12-13 rank  of this position among the item's foreign positions, in recording order    (type 2)
// derived on the host (template_rows), passed by value, copied to LDS by the workgroup.
struct TemplateRows {
    uint64_t row[2][kTmplRows];
    uint8_t L[2], V[2], F[2];  // rows, Variables, foreign positions of a full / a short item
    uint8_t Fmax, pad;
};
inline TemplateRows template_rows(uint32_t kind) {
    const SegTemplate &T = kSegTemplates[kind - WIRES_MIX];
    TemplateRows R{};
    for (uint32_t sh = 0; sh < 2; sh++) {
        R.L[sh] = T.L[sh];
        R.V[sh] = T.V[sh];
        uint32_t foreign = 0;
        const uint32_t n = (uint32_t)T.L[sh] * 3;
        for (uint32_t p = 0; p < n; p++) {
            const uint8_t code = T.wire[sh][p / 3][p % 3];
            uint64_t b = 0;
            if (code < kTmplForeign) {  // an own Variable: the next entry that names it, or -- from the last -- the first
                uint32_t first = 0xff, next = 0xff;
                for (uint32_t q = 0; q < n; q++)
                    if (T.wire[sh][q / 3][q % 3] == code) {
                        if (first == 0xff) first = q;
                        if (q > p && next == 0xff) next = q;
                    }
                const uint32_t z = next != 0xff ? next : first;
                b = 1u | (uint64_t)code << 2 | (uint64_t)(4 * (z / 3) + z % 3) << 6;
            } else if (code < kTmplZero) {
                b = 2u | (uint64_t)foreign++ << 12;
            } else if (code == kTmplZero) {
                b = 3u;
            }
            R.row[sh][p / 3] |= b << (16 * (p % 3));
        }
        R.F[sh] = (uint8_t)foreign;
    }
    R.Fmax = (uint8_t)template_foreign_per_item(kind);
    return R;
}

constexpr uint32_t kTmplWindow = 1024;  // ragged: prefix sums per piece (a piece of 512 rows holds at most 513 items; a power of two)
template <bool RAGGED>
__global__ __launch_bounds__(kThreads) void perm_template_kernel(const PermCtx X, const PermSeg S, const TemplateRows T, const PermSparse Q,
                                                                uint64_t *sigma) {
    extern __shared__ uint4 perm_template_pad[];
    __shared__ uint64_t s_row[2][kTmplRows];
    __shared__ uint32_t s_off[RAGGED ? kTmplWindow : 1];        // RAGGED: rows before items item0 .. of the piece, from item0's first
    static_assert(kPermLadderRows + 3 <= kTmplWindow, "a piece's items fit the window of prefix sums");
    if (X.padded_n == 1) perm_template_pad[threadIdx.x] = make_uint4(0, 0, 0, 0);  // (keeps the allocation)
    const uint32_t Lf = T.L[0], Ls = T.L[1], Fmax = T.Fmax, Ff = T.F[0], Fs = T.F[1];
    if (threadIdx.x < 2 * kTmplRows) s_row[threadIdx.x / kTmplRows][threadIdx.x % kTmplRows] = T.row[threadIdx.x / kTmplRows][threadIdx.x % kTmplRows];
    __syncthreads();
    const uint64_t first = S.gate_base & ~1ull, total = S.gate_end - first;  // (rows counted from the even gate at or before the segment's first)
    const bool wide = (reinterpret_cast<uintptr_t>(sigma) & 15) == 0 && !(X.padded_n & 1);
    const uint32_t recip = (uint32_t)(((1ull << 32) + Lf - 1) / Lf);
    const bool pow2 = (X.padded_n & (X.padded_n - 1)) == 0;
    const uint32_t sh_bits = 63u - (uint32_t)__clzll((long long)X.padded_n);
    auto enc = [&](uint64_t gate, uint32_t wire) { return pow2 ? ((uint64_t)wire << sh_bits) + gate : perm_encode(gate, wire, X.padded_n); };
    for (uint64_t base = (uint64_t)blockIdx.x * kPermLadderRows; base < total; base += (uint64_t)gridDim.x * kPermLadderRows) {
        const uint64_t g_first = first + base, rel = g_first < S.gate_base ? 0 : g_first - S.gate_base;
        uint64_t item0, rows0;
        if constexpr (RAGGED) {
            item0 = S.piece_item[base / kPermLadderRows];
            rows0 = S.row_off[item0];
            __syncthreads();  // (a workgroup's previous piece is read out)
            for (uint32_t e = threadIdx.x; e < kTmplWindow; e += kThreads) {
                const uint64_t i = item0 + e < S.items ? item0 + e : S.items;
                s_off[e] = (uint32_t)(S.row_off[i] - rows0);
            }
            __syncthreads();
        } else {
            item0 = rel / Lf;
            rows0 = item0 * Lf;
        }
        const uint32_t j0 = (uint32_t)(rel - rows0), skip = (uint32_t)(g_first < S.gate_base ? S.gate_base - g_first : 0);
        // a row of the piece: its item (counted from item0), its place in the item, the item's shape and first gate, the table's word
        struct Row { uint32_t item, j, sh, rows; uint64_t g_item, word; };
        auto locate = [&](uint32_t x, Row &R) {  // x: rows from the piece's first item's first row
            if constexpr (RAGGED) {
                uint32_t lo = 0;  // the last entry of the window that is <= x
#pragma unroll
                for (uint32_t step = kTmplWindow / 2; step; step >>= 1)
                    if (s_off[lo + step] <= x) lo += step;
                R.item = lo;
                R.j = x - s_off[lo];
                R.rows = s_off[lo + 1] - s_off[lo];
                R.sh = R.rows == Lf ? 0u : 1u;
                R.g_item = S.gate_base + rows0 + s_off[lo];
            } else {
                R.item = Lf == 1 ? x : __umulhi(x, recip);   // (exact: x < 2^14, 2 <= L <= 10; ceil(2^32 / 1) does not fit)
                R.j = x - R.item * Lf;
                R.sh = 0;
                R.rows = Lf;
                R.g_item = S.gate_base + rows0 + (uint64_t)R.item * Lf;
            }
            R.word = s_row[R.sh][R.j];
        };
        auto step = [&](Row &R) {  // the row behind R
            if (R.j + 1 < R.rows) R.j++;
            else {
                R.g_item += R.rows;
                R.item++;
                R.j = 0;
                if constexpr (RAGGED) {
                    R.rows = s_off[R.item + 1] - s_off[R.item];
                    R.sh = R.rows == Lf ? 0u : 1u;
                }
            }
            R.word = s_row[R.sh][R.j];
        };
        // wires 0..2 of gate g that hold zero_var: the table's, and a Variable from elsewhere that is zero_var
        auto zero_mask = [&](uint64_t g, uint64_t word) {
            uint32_t z = 0;
#pragma unroll
            for (uint32_t w = 0; w < 3; w++) {
                const uint32_t type = (uint32_t)(word >> (16 * w)) & 3u;
                if (type == 3u) z |= 1u << w;
                else if (type == 2u && X.C.w[w][g] == X.zero_var) z |= 1u << w;
            }
            return z;
        };
        for (uint32_t t = 2 * threadIdx.x; t < kPermLadderRows; t += 2 * kThreads) {
            uint64_t out[4][2];
            uint32_t keep[2] = {0, 0};  // wires whose sigma entry is written here (the others are the sparse list's)
            bool live[3];
            Row R[3];                   // the lane's two gates and the one behind them (whose zero wires the second one's chain needs)
            uint32_t zm[3] = {0, 0, 0};
#pragma unroll
            for (uint32_t h = 0; h < 3; h++) {
                const uint64_t g = g_first + t + h;
                live[h] = g >= S.gate_base && g < S.gate_end;
                if (h == 0 || !live[h - 1]) {
                    R[h] = Row{0, 0, 0, 1, 0, 0};
                    if (live[h]) locate(j0 + t + h - skip, R[h]);
                } else {
                    R[h] = R[h - 1];
                    if (live[h]) step(R[h]);
                }
                if (live[h]) zm[h] = zero_mask(g, R[h].word);
            }
#pragma unroll
            for (uint32_t h = 0; h < 2; h++) {
                const uint64_t g = g_first + t + h;
#pragma unroll
                for (uint32_t w = 0; w < 4; w++) out[w][h] = 0;
                if (!live[h]) continue;
                const uint32_t zw = zm[h];
                const uint64_t slot0 = S.sparse_base + (item0 + R[h].item) * Fmax;
#pragma unroll
                for (uint32_t w = 0; w < 3; w++) {
                    const uint32_t b = (uint32_t)(R[h].word >> (16 * w)) & 0xffffu, type = b & 3u;
                    if (zw >> w & 1) {  // zero_var: on to the next wire of this row that holds it (the fourth at the latest)
                        const uint32_t later = (zw | 8u) & ~((2u << w) - 1);
                        out[w][h] = enc(g, (uint32_t)__ffs((int)later) - 1);
                        keep[h] |= 1u << w;
                        if (type == 2u) {  // a Variable from elsewhere that is zero_var: its slot stays a hole
                            const uint64_t at = slot0 + (b >> 12 & 3u);
                            if (at < Q.cap) Q.keys[at] = X.hole_key;
                            atomicAdd(Q.count + 1, 1ull);
                        }
                    } else if (type == 1u) {
                        const uint32_t sc = b >> 6 & 63u;
                        out[w][h] = enc(R[h].g_item + (sc >> 2), sc & 3);
                        keep[h] |= 1u << w;
                    } else if (type == 2u) {
                        perm_sparse_put(X, Q, slot0 + (b >> 12 & 3u), X.C.w[w][g], g, w);
                    }
                }
                if (R[h].j == 0)  // a short item uses fewer slots than it owns: the rest are holes
                    for (uint32_t k = R[h].sh ? Fs : Ff; k < Fmax; k++) {
                        if (slot0 + k < Q.cap) Q.keys[slot0 + k] = X.hole_key;
                        atomicAdd(Q.count + 1, 1ull);
                    }
                // the fourth wire holds zero_var: on to the first wire of the next row that does
                if (g + 1 < S.gate_end) out[3][h] = enc(g + 1, (uint32_t)__ffs((int)(zm[h + 1] | 8u)) - 1);
                else out[3][h] = perm_next_zero_from(X, S.gate_end);
                keep[h] |= 8u;
            }
            const uint64_t g = g_first + t;
#pragma unroll
            for (uint32_t w = 0; w < 4; w++) {
                const bool k0 = live[0] && (keep[0] >> w & 1), k1 = live[1] && (keep[1] >> w & 1);
                uint64_t *dst = sigma + enc(g, w);
                if (wide && k0 && k1)
                    store16(reinterpret_cast<uint4 *>(dst), make_uint4((uint32_t)out[w][0], (uint32_t)(out[w][0] >> 32), (uint32_t)out[w][1],
                                                                      (uint32_t)(out[w][1] >> 32)));
                else {
                    if (k0) dst[0] = out[w][0];
                    if (k1) dst[1] = out[w][1];
                }
            }
        }
    }
    (void)Ls;
}

// the sorted sparse list: neighbours with one Variable are consecutive positions of it; the last one wraps to the
// first unless the Variable belongs to a batched item (perm_splice_kernel closes that cycle)
__global__ __launch_bounds__(kThreads) void perm_sparse_link_kernel(const PermCtx X, const uint64_t *keys, uint64_t nS,
                                                                   uint64_t *sigma) {
    const uint64_t pos_mask = (1ull << X.pos_bits) - 1;
    for (uint64_t j = (uint64_t)blockIdx.x * kThreads + threadIdx.x; j < nS; j += (uint64_t)gridDim.x * kThreads) {
        const uint64_t key = keys[j], v = key >> X.pos_bits;
        uint64_t succ;
        if (j + 1 < nS && keys[j + 1] >> X.pos_bits == v) succ = keys[j + 1];
        else {
            if (perm_home_seg(X, v) >= 0) continue;
            uint64_t h = j;
            while (h > 0 && keys[h - 1] >> X.pos_bits == v) h--;
            succ = keys[h];
        }
        sigma[perm_encode_pos(key & pos_mask, X.padded_n)] = perm_encode_pos(succ & pos_mask, X.padded_n);
    }
}

// sparse runs whose Variable belongs to a batched item: every lane tests one entry (run head with a home segment?), the
// wave then takes the hits one by one, searches the item's rows for the Variable's local positions (already one closed
// cycle, ascending) and joins the two lists: local first..last -> sparse first..last -> local first
__global__ __launch_bounds__(kThreads) void perm_splice_kernel(const PermCtx X, const uint64_t *keys, uint64_t nS, uint64_t *sigma) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t pos_mask = (1ull << X.pos_bits) - 1;
    const uint64_t waves = (uint64_t)gridDim.x * (kThreads / 64);
    for (uint64_t jb = ((uint64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * 64; jb < nS; jb += waves * 64) {
        const uint64_t jm = jb + lane;
        int my_seg = -1;
        if (jm < nS) {
            const uint64_t vm = keys[jm] >> X.pos_bits;
            if (jm == 0 || keys[jm - 1] >> X.pos_bits != vm) my_seg = perm_home_seg(X, vm);
        }
        uint64_t hits = __ballot(my_seg >= 0);
        while (hits) {
            const uint32_t src = (uint32_t)__ffsll((long long)hits) - 1;
            hits &= hits - 1;
            const uint64_t j = jb + src;
            const uint64_t v = keys[j] >> X.pos_bits;
            const PermSeg s = X.segs[__shfl(my_seg, (int)src, 64)];
            const uint64_t ib = perm_item_of_var(s, v) / s.group * s.group, ie = ib + s.group < s.items ? ib + s.group : s.items;
            const uint64_t rb0 = perm_rows_before(s, ib), g0 = s.gate_base + rb0;
            const uint32_t sL = (uint32_t)(perm_rows_before(s, ie) - rb0);
            uint64_t lo = ~0ull, hi = 0;
            bool found = false;
            for (uint32_t w = 0; w < 3; w++)
                for (uint32_t r = lane; r < sL; r += 64)
                    if (X.C.w[w][g0 + r] == v) {
                        const uint64_t p = 4 * (g0 + r) + w;
                        lo = p < lo ? p : lo;
                        hi = p > hi ? p : hi;
                        found = true;
                    }
            for (int d = 32; d; d >>= 1) {
                const uint64_t olo = __shfl_xor(lo, d, 64), ohi = __shfl_xor(hi, d, 64);
                lo = olo < lo ? olo : lo;
                hi = ohi > hi ? ohi : hi;
            }
            found = __any(found);
            if (lane == 0) {
                uint64_t e = j;
                while (e + 1 < nS && keys[e + 1] >> X.pos_bits == v) e++;
                const uint64_t first = perm_encode_pos(keys[j] & pos_mask, X.padded_n),
                               last = perm_encode_pos(keys[e] & pos_mask, X.padded_n);
                if (!found) sigma[last] = first;
                else {
                    sigma[perm_encode_pos(hi, X.padded_n)] = first;
                    sigma[last] = perm_encode_pos(lo, X.padded_n);
                }
            }
        }
    }
}

// rows >= circuit size keep the identity: 16-byte stores over the four runs [wire * padded_n + n, (wire + 1) * padded_n).
// SHORT-LIVED workgroups of 8 KiB each (two stores per lane), block b -> run b % 4 (four fronts that advance together), started in
// address order by the dispatcher, and only two of them resident per CU (kPermIdentityLds bytes of dynamic LDS each, unused): what is
// under way at any moment is a window of a few MiB per run.  Long-lived workgroups walking 1-MiB pieces took 1.64-1.69 ms for the 8.5 GB
// of a 270 M-row circuit padded to 2^29, this shape 1.37 (tools/probes/identity_fill.hip; single_table_fill.hip has the general finding:
// a store stream whose window is small does not care where its array lies).
constexpr uint32_t kPermIdentityLds = 160 * 1024 / 2 - 1024;
constexpr uint32_t kPermIdentityUnits = 2 * kThreads;  // 16-byte units per workgroup
__global__ __launch_bounds__(kThreads) void perm_identity_kernel(uint64_t *sigma, uint64_t n, uint64_t padded_n) {
    typedef unsigned long long __attribute__((ext_vector_type(2))) u64x2;
    extern __shared__ uint4 perm_identity_pad[];
    const uint32_t wire = blockIdx.x & 3, b = blockIdx.x >> 2;
    uint64_t lo = wire * padded_n + n, hi = (wire + 1) * padded_n;
    if (b == 0 && threadIdx.x == 0) {
        if (lo & 1) sigma[lo] = lo;
        if (hi & 1) sigma[hi - 1] = hi - 1;
    }
    lo += lo & 1;
    hi -= hi & 1;
    const uint64_t base = lo + ((uint64_t)b * kPermIdentityUnits + threadIdx.x) * 2;
#pragma unroll
    for (uint32_t k = 0; k < kPermIdentityUnits / kThreads; k++) {
        const uint64_t i = base + (uint64_t)k * 2 * kThreads;
        if (i < hi) *reinterpret_cast<u64x2 *>(sigma + i) = u64x2{i, i + 1};
    }
    if (padded_n == 1) perm_identity_pad[threadIdx.x] = make_uint4(0, 0, 0, 0);  // (keeps the allocation; padded_n == 1 has no padding)
}

}  // namespace pg
