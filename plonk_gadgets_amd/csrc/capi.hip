// capi.hip -- the C ABI of libplonk_gadgets_hip.so (include/plonk_gadgets_hip.h).
// Host-side validation and layout arithmetic + kernel launches.  There is no
// CPU fallback anywhere in this file: a call either runs on the GPU or
// returns an error status.
#include "../../include/plonk_gadgets_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <new>
#include <functional>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "emit.hpp"
#include "fr.hpp"
#include "range_gadgets.hpp"
#include "scalar_gadgets.hpp"
#include "composer.hpp"
#include "permutation.hpp"
#include "materialize.hpp"

#ifndef PG_GRID_BLOCKS_PER_CU
// more workgroups than can be resident: the dispatcher back-fills CUs as tiles finish (+6 % over a persistent
// 8-per-CU grid on the C2 shape, tools/ab_emit.py)
#define PG_GRID_BLOCKS_PER_CU 64
#endif
#ifndef PG_EMIT_GRID_BLOCKS_PER_CU
// the full emission of a big-item gadget: at most this many tiles' worth of workgroups, i.e. ONE tile per workgroup, in dispatch order,
// up to 2^20 items of C4's shape (65536 tiles; with 64 per CU a workgroup walked four tiles 16384 apart): 17.45 -> 17.07 ms and
// 18.34 -> 17.90 on two boxes; C2's shape (32768 tiles) 33.31 / 33.27, C3's unchanged.  The witness refresh keeps 64: one tile per
// workgroup costs it 5.6 -> 6.15 ms on the tables that are fast (tools/ab_emit.py run_values, `ga`)
#define PG_EMIT_GRID_BLOCKS_PER_CU 256
#endif

namespace {

thread_local std::string g_last_error;

pg_status fail(pg_status s, const std::string &msg) {
    g_last_error = msg;
    return s;
}

#define PG_HIP_TRY(expr)                                                                \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess)                                                           \
            return fail(PG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define PG_TRY(expr)                   \
    do {                               \
        pg_status _s = (expr);         \
        if (_s != PG_OK) return _s;    \
    } while (0)

inline pg::Fr to_fr(const pg_scalar *s) {
    pg::Fr f;
    std::memcpy(f.l, s->l, sizeof f.l);
    return f;
}
inline void from_fr(const pg::Fr &f, pg_scalar *out) { std::memcpy(out->l, f.l, sizeof f.l); }

inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// a BlsScalar is always fully reduced; reject limbs >= q instead of computing garbage
bool is_reduced(const pg::Fr &f) {
    const uint64_t Q[4] = {PG_Q0, PG_Q1, PG_Q2, PG_Q3};
    for (int i = 3; i >= 0; i--) {
        if (f.l[i] < Q[i]) return true;
        if (f.l[i] > Q[i]) return false;
    }
    return false;
}

pg_status check_columns(const pg_columns *c) {
    if (!c) return fail(PG_ERR_INVALID_ARGUMENT, "columns is NULL");
    const void *sc[6] = {c->q_m, c->q_l, c->q_r, c->q_o, c->q_c, c->var_values};
    for (const void *p : sc)
        if (!p || !aligned(p, 16)) return fail(PG_ERR_INVALID_ARGUMENT, "scalar column NULL or not 16-byte aligned");
    const void *wc[3] = {c->w_l, c->w_r, c->w_o};
    for (const void *p : wc)
        if (!p || !aligned(p, 8)) return fail(PG_ERR_INVALID_ARGUMENT, "wire column NULL or not 8-byte aligned");
    return PG_OK;
}

pg_status check_scalars(const void *p, const char *what) {
    if (!p || !aligned(p, 16)) return fail(PG_ERR_INVALID_ARGUMENT, std::string(what) + " NULL or not 16-byte aligned");
    return PG_OK;
}
pg_status check_u64s(const void *p, const char *what, bool nullable = false) {
    if (!p && nullable) return PG_OK;
    if (!p || !aligned(p, 8)) return fail(PG_ERR_INVALID_ARGUMENT, std::string(what) + " NULL or not 8-byte aligned");
    return PG_OK;
}

}  // namespace

struct pg_engine {
    int device = -1;
    int num_cus = 0;
    uint4 *d_pow2 = nullptr;  // mont(2^i), i < 256
    // scratch of the ragged plans (grow-only): per-item counts, block sums, error counter
    uint32_t *d_rows = nullptr, *d_vars = nullptr;
    uint64_t *d_blk_rows = nullptr, *d_blk_vars = nullptr;
    unsigned long long *d_blk_agg = nullptr;  // single-launch plans (emit.hpp, PlanScan): published block totals; [blocks] = blocks done
    uint32_t *d_err_count = nullptr;
    uint64_t scratch_items = 0, scratch_blocks = 0;
    // totals of the last plan, written by async copies into pinned host memory (read after a synchronisation)
    using PlanResult = pg::PlanTotals;
    PlanResult *h_plan = nullptr;
    // scratch of the inversions (grow-only): the pre-pass parks an element and its running product (64 B per element),
    // the fused mix one running product per item (32 B)
    uint4 *d_prefix = nullptr;
    uint64_t inv_elems = 0;
    // the pre-pass runs on its own stream beside the rows-only emit launch
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_inv = nullptr;
    // the stream the last call was issued on (see enter_stream)
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    hipEvent_t ev_switch = nullptr;
};

namespace {

pg_status ensure_scratch(pg_engine *e, uint64_t batch) {
    if (batch <= e->scratch_items && e->d_err_count) return PG_OK;
    PG_HIP_TRY(hipSetDevice(e->device));
    if (e->d_rows) { (void)hipFree(e->d_rows); e->d_rows = nullptr; }
    if (e->d_vars) { (void)hipFree(e->d_vars); e->d_vars = nullptr; }
    if (e->d_blk_rows) { (void)hipFree(e->d_blk_rows); e->d_blk_rows = nullptr; }
    if (e->d_blk_vars) { (void)hipFree(e->d_blk_vars); e->d_blk_vars = nullptr; }
    if (e->d_blk_agg) { (void)hipFree(e->d_blk_agg); e->d_blk_agg = nullptr; }
    e->scratch_items = 0;
    const uint64_t items = batch < 1024 ? 1024 : batch;
    // (the published words: one per block of a plan kernel, or per wave of the fused mix's planning launch -- at most 2048
    // of those up to 2 M items, one per 1024 items beyond)
    const uint64_t nblk = std::max<uint64_t>((items + pg::kScanBlock - 1) / pg::kScanBlock, 2048);
    PG_HIP_TRY(hipMalloc(&e->d_rows, items * sizeof(uint32_t)));
    PG_HIP_TRY(hipMalloc(&e->d_vars, items * sizeof(uint32_t)));
    PG_HIP_TRY(hipMalloc(&e->d_blk_rows, nblk * sizeof(uint64_t)));
    PG_HIP_TRY(hipMalloc(&e->d_blk_vars, nblk * sizeof(uint64_t)));
    PG_HIP_TRY(hipMalloc(&e->d_blk_agg, (nblk + 1) * sizeof(unsigned long long)));
    PG_HIP_TRY(hipMemset(e->d_blk_agg, 0, (nblk + 1) * sizeof(unsigned long long)));  // zero between launches: the plan kernels leave them so
    e->scratch_blocks = nblk;
    if (!e->d_err_count) {
        PG_HIP_TRY(hipMalloc(&e->d_err_count, sizeof(uint32_t)));
        PG_HIP_TRY(hipMemset(e->d_err_count, 0, sizeof(uint32_t)));  // zero between calls (see error_plan)
    }
    e->scratch_items = items;
    return PG_OK;
}

// what a plan kernel is handed to finish the prefix sums itself (up to kPlanFusedBlocks blocks; beyond, fused = 0 and
// scan_counts launches the two-level scan)
pg::PlanScan plan_scan(pg_engine *e, uint64_t batch, uint64_t *d_row_off, uint64_t *d_var_off, bool with_errs) {
    const uint64_t nblk = (batch + pg::kScanBlock - 1) / pg::kScanBlock;
    pg::PlanScan P{};
    P.agg = e->d_blk_agg;
    P.blocks_cap = (uint32_t)e->scratch_blocks;
    P.blk_rows = e->d_blk_rows;
    P.blk_vars = e->d_blk_vars;
    P.row_off = d_row_off;
    P.var_off = d_var_off;
    P.host = e->h_plan;
    P.err_count = with_errs ? e->d_err_count : nullptr;
#if defined(PG_PLAN_TWO_LAUNCHES)  // A/B build
    P.fused = 0;
#else
    P.fused = nblk <= pg::kPlanFusedBlocks ? 1u : 0u;
#endif
    return P;
}

pg_status plan_totals(pg_engine *e, uint64_t *n_rows, uint64_t *n_vars) {
    if (e->h_plan->pad) {
        e->h_plan->pad = 0;
        return fail(PG_ERR_HIP, "a plan's look-back gave up waiting for another block's totals");
    }
    *n_rows = e->h_plan->n_gates;
    *n_vars = e->h_plan->n_vars;
    return PG_OK;
}

// counts in e->d_rows / e->d_vars and their block sums in e->d_blk_* (both left by the plan kernel) -> exclusive prefix
// sums; totals copied back (the synchronous form synchronises the stream).  Nothing to launch after a fused plan.
pg_status scan_counts(pg_engine *e, uint64_t batch, uint64_t *d_row_off, uint64_t *d_var_off, uint64_t *n_rows,
                      uint64_t *n_vars, hipStream_t st, bool with_errs) {
    const uint32_t nblk = (uint32_t)((batch + pg::kScanBlock - 1) / pg::kScanBlock);
    if (!plan_scan(e, batch, d_row_off, d_var_off, with_errs).fused) {
        const uint32_t prefixed = nblk > pg::kScanDirectBlocks ? 1u : 0u;
        if (prefixed)
            hipLaunchKernelGGL(pg::scan_top_kernel, dim3(1), dim3(pg::kThreads), 0, st, e->d_blk_rows, e->d_blk_vars, nblk);
        hipLaunchKernelGGL(pg::scan_final_kernel, dim3(nblk), dim3(pg::kThreads), 0, st, e->d_rows, e->d_vars, batch,
                           e->d_blk_rows, e->d_blk_vars, d_row_off, d_var_off, prefixed, e->h_plan, with_errs ? e->d_err_count : nullptr);
        PG_HIP_TRY(hipGetLastError());
    }
    if (n_rows) {  // synchronous form
        PG_HIP_TRY(hipStreamSynchronize(st));
        PG_TRY(plan_totals(e, n_rows, n_vars));
    }
    return PG_OK;
}

pg::EmitOut make_out(const pg_columns *c, uint64_t batch, int W, uint64_t gate_base, uint64_t var_base, uint64_t zero_var,
                     const uint64_t *row_off, const uint64_t *var_off) {
    pg::EmitOut O;
    O.q[0] = reinterpret_cast<uint4 *>(c->q_m);
    O.q[1] = reinterpret_cast<uint4 *>(c->q_l);
    O.q[2] = reinterpret_cast<uint4 *>(c->q_r);
    O.q[3] = reinterpret_cast<uint4 *>(c->q_o);
    O.q[4] = reinterpret_cast<uint4 *>(c->q_c);
    O.w[0] = c->w_l;
    O.w[1] = c->w_r;
    O.w[2] = c->w_o;
    O.vars = reinterpret_cast<uint4 *>(c->var_values);
    O.gate_base = gate_base;
    O.var_base = var_base;
    O.zero_var = zero_var;
    O.row_off = row_off;
    O.var_off = var_off;
    O.batch = batch;
    O.tiles = (uint32_t)((batch + W - 1) / W);
    O.stride_rows = 0;
    O.inv_dense = nullptr;
    O.inv_elems = 0;
    O.inv_in_place = 1;
    O.early_span = O.early_tiles = 0;
    return O;
}

pg_status ensure_inv_scratch(pg_engine *e, uint64_t elems) {
    if (elems <= e->inv_elems) return PG_OK;
    PG_HIP_TRY(hipSetDevice(e->device));
    if (e->d_prefix) { (void)hipFree(e->d_prefix); e->d_prefix = nullptr; }
    e->inv_elems = 0;
    // per element: the element and its running product; + 64 x 16 bytes that the fused mix's masked lanes store to
    PG_HIP_TRY(hipMalloc(&e->d_prefix, (elems * 4 + 64 + 4 * 32 * 32 * 8) * sizeof(uint4)));  // (+ the fused mix's last workgroup: whole steps of whole waves)
    e->inv_elems = elems;
    return PG_OK;
}

// the engine's events order streams of ONE device against each other and are never waited for by the host: without the
// system-scope fence a default event performs when it completes (cache write-back and invalidation: microseconds on the
// stream, after every call)
constexpr unsigned kOrderingEvent = hipEventDisableTiming | hipEventDisableSystemFence;

// The engine's scratch (plan counts, pre-pass products, the pinned plan result) is shared by consecutive calls and is
// ordered only by the stream they are issued on.  A caller that moves to ANOTHER stream is made to wait for everything
// the engine still has in flight on the previous one (and on the side stream, which joins it): every call records, when
// it has enqueued its last work, an event on ITS OWN stream (StreamScope), and a call that finds itself on another
// stream than the last waits for that event.  The previous stream's handle is only ever compared, never used: the caller
// may have destroyed that stream since (the header allows it), and a handle that has been recycled for a new stream would
// otherwise have an event recorded on it that orders nothing.
pg_status enter_stream(pg_engine *e, hipStream_t st) {
    PG_HIP_TRY(hipSetDevice(e->device));
    if (e->have_last && e->last_stream != st) PG_HIP_TRY(hipStreamWaitEvent(st, e->ev_switch, 0));
    e->last_stream = st;
    e->have_last = true;
    return PG_OK;
}
struct StreamScope {  // declared right after enter_stream succeeds; its destructor runs on every way out of the call
    pg_engine *e;
    hipStream_t st;
    ~StreamScope() { (void)hipEventRecord(e->ev_switch, st); }
};

// Once a call has forked its pre-pass to the engine's side stream, the caller's stream waits for the side stream on EVERY way
// out of the call, the failing ones included: StreamScope (declared before, so destroyed after) then records ev_switch behind
// that wait, and a following call on another stream is ordered after the pre-pass, which writes e->d_prefix and variable slots.
struct SideJoin {
    pg_engine *e;
    hipStream_t st;
    bool forked = false, joined = false;
    ~SideJoin() {
        if (!forked || joined) return;
        (void)hipEventRecord(e->ev_inv, e->side);
        (void)hipStreamWaitEvent(st, e->ev_inv, 0);
    }
};

#ifndef PG_MIX_EARLY_TILES  // row tiles (256 items) per workgroup of the fused mix's arithmetic launch written during its inversions
#define PG_MIX_EARLY_TILES 1
#endif
#ifndef PG_INV_LANES_PER_CU  // lanes of the pre-pass per CU (256 = one wave per SIMD)
#define PG_INV_LANES_PER_CU 256
#endif
#ifndef PG_INV_MAX_PER_LANE
#define PG_INV_MAX_PER_LANE 32
#endif

// One batched gadget call = the emit kernel on the caller's stream and, for gadgets that invert, the inversion pre-pass
// on the engine's high-priority side stream.  The two write disjoint bytes (the pre-pass owns the inverse slots of
// the variable table), so they run concurrently; the caller's stream is made to wait for both before the call's
// results can be consumed.
// A split gadget (small items: the fused mix) = ONE launch that inverts and writes the variable table
// (scalar_mix_vars_kernel: integer arithmetic, bound by the multiplier) and then the rows, a pure store stream of 1.9 GB per
// 2^20 items.  The rows need nothing the arithmetic computes, and still the launches run ONE AFTER THE OTHER: side by
// side on the same compute units each slows the other down by more than the overlap gains, in every order and priority
// tried -- even with the arithmetic launch stripped of all its memory traffic -- and on disjoint compute units (CU-masked
// streams) the rows lack store bandwidth, which is per CU (profiles/NOTES_r03.md has the measurements).
// `planned`: the arithmetic launch makes the call's prefix sums itself (d_row_off / d_var_off are outputs, as are the
// engine's plan totals); otherwise it reads them.
pg_status launch_mix(pg_engine *e, const pg::ScalarMixArgs &A, const pg_columns *c, uint64_t batch, uint64_t gate_base,
                     uint64_t var_base, uint64_t zero_var, const uint64_t *row_off, const uint64_t *var_off, hipStream_t st,
                     const pg::MixPlan *planned, bool values_only = false) {
    using GD = pg::ScalarMixGD;
    PG_TRY(ensure_inv_scratch(e, batch));
    // geometry of the arithmetic launch: a wave owns 32 * ipl consecutive items; two waves per SIMD is what it is built for
    const uint64_t waves_wanted = (uint64_t)e->num_cus * 4 * 2;
    uint64_t ipl = (batch + waves_wanted * 32 - 1) / (waves_wanted * 32);
    if (ipl < 1) ipl = 1;
    if (ipl > pg::kMixMaxIpl) ipl = pg::kMixMaxIpl;
    const uint64_t waves = (batch + 32 * ipl - 1) / (32 * ipl);
    pg::EmitOut V = make_out(c, batch, GD::W, gate_base, var_base, zero_var, row_off, var_off);
    pg::EmitOut R = make_out(c, batch, GD::kRowsW, gate_base, var_base, zero_var, row_off, var_off);
    // early rows: the arithmetic launch's waiting waves write the first PG_MIX_EARLY_TILES row tiles of every workgroup's items
    // during its inversions (scalar_gadgets.hpp); the rows launches leave them alone.  Only where a workgroup holds enough tiles
    // for that to be a fraction of its rows (a small call's inversions are not worth hiding)
    const uint64_t span = (uint64_t)pg::kMixWaves * 32 * ipl;
    if (!values_only && ipl >= 8 && span <= 0xffffffffull) {
        V.early_span = R.early_span = (uint32_t)span;
        V.early_tiles = R.early_tiles = PG_MIX_EARLY_TILES;
    }
    const uint32_t max_blocks = (uint32_t)e->num_cus * PG_GRID_BLOCKS_PER_CU;
    const dim3 grid(R.tiles < max_blocks ? R.tiles : max_blocks), vgrid((uint32_t)((waves + pg::kMixWaves - 1) / pg::kMixWaves));
    const dim3 vblock(pg::kMixWaves * 64);
    uint4 *sink = e->d_prefix + e->inv_elems * 4 + 4 * 32 * 32 * 8;
    if (planned) {
        pg::MixPlan P = *planned;
        P.nwaves = vgrid.x;  // (the look-back is over workgroups)
        hipLaunchKernelGGL(pg::scalar_mix_vars_kernel<true>, vgrid, vblock, 0, st, A, V, (uint32_t)ipl, e->d_prefix, sink, P);
    } else
        hipLaunchKernelGGL(pg::scalar_mix_vars_kernel<false>, vgrid, vblock, 0, st, A, V, (uint32_t)ipl, e->d_prefix, sink,
                           pg::MixPlan{});
    if (values_only) {  // a witness refresh: the variable table (and, planned, the prefix sums) alone
        PG_HIP_TRY(hipGetLastError());
        return PG_OK;
    }
    hipLaunchKernelGGL(pg::rows_periodic_kernel<GD>, grid, dim3(pg::kThreads), 0, st, A, R);
    // the tiles that hold an item of another shape (none, unless an item stopped at its error): every other workgroup of
    // this launch reads two offsets and leaves
    hipLaunchKernelGGL((pg::emit_kernel<GD, pg::EMIT_ROWS>), grid, dim3(pg::kThreads), 0, st, A, R);
    PG_HIP_TRY(hipGetLastError());
    return PG_OK;
}

// the witness refresh (EMIT_VALUES: the variable assignments alone) exists for every gadget of the one-launch emitter; the split
// gadget (the fused mix) has its own: the launch that writes its variable table, alone (launch_mix)
template <class GD> struct ValuesMode { static constexpr bool ok = !pg::Split<GD>::ok; };

template <class GD>
pg_status launch(pg_engine *e, const typename GD::Args &A, const pg_columns *c, uint64_t batch, uint64_t gate_base,
                 uint64_t var_base, uint64_t zero_var, const uint64_t *row_off, const uint64_t *var_off, void *stream,
                 const pg::MixPlan *planned = nullptr, bool values_only = false, uint32_t stride_rows = 0) {
    if ((batch + GD::W - 1) / GD::W > 0xffffffffull) return fail(PG_ERR_INVALID_ARGUMENT, "batch too large for one call");
    if (stride_rows && (batch > 0xffffffffull || pg::Split<GD>::ok || GD::kRagged))
        return fail(PG_ERR_INVALID_ARGUMENT, "a row stride is for uniform one-launch gadgets of at most 2^32 - 1 items");
    hipStream_t st = static_cast<hipStream_t>(stream);
    PG_TRY(enter_stream(e, st));
    StreamScope scope{e, st};
    if constexpr (pg::Split<GD>::ok) {
        return launch_mix(e, A, c, batch, gate_base, var_base, zero_var, row_off, var_off, st, planned, values_only);
    } else {
        pg::EmitOut O = make_out(c, batch, GD::W, gate_base, var_base, zero_var, row_off, var_off);
        if (stride_rows) {  // rows of other calls between the items: one item per tile (emit.hpp)
            O.stride_rows = stride_rows;
            O.tiles = (uint32_t)batch;
        }
        bool side = false;  // the inversion pre-pass runs on the engine's side stream
        SideJoin join{e, st};
        if constexpr (GD::kInv > 0) {
            constexpr int GRP = GD::kInvGroup;
            const uint64_t elems = batch * GD::kInv;
            PG_TRY(ensure_inv_scratch(e, elems));
            const uint64_t lanes_wanted = (uint64_t)e->num_cus * PG_INV_LANES_PER_CU;
            uint64_t per_lane = (elems + lanes_wanted - 1) / lanes_wanted;
            if (per_lane < 1) per_lane = 1;
            if (per_lane > PG_INV_MAX_PER_LANE) per_lane = PG_INV_MAX_PER_LANE;
            const uint64_t groups = (per_lane + GRP - 1) / GRP;  // a lane owns groups * GRP elements
            const uint64_t lanes = (elems + groups * GRP - 1) / (groups * GRP);
            const uint32_t blocks = (uint32_t)((lanes + pg::kThreads - 1) / pg::kThreads);
            // a handful of elements (the single-gadget calls of pg_composer): two event hops cost more than the overlap buys.
            // A big call (an emit launch of milliseconds): the pre-pass goes FIRST on the caller's stream -- beside the
            // emitter its round trips crawl through a saturated memory system and it ends with the emitter (17.58 of 17.64 ms
            // at 2^20 ragged max_bound items), one wave per SIMD taken from the store stream all the while; alone it takes
            // 0.15 ms and the step is the same within 0.4 % (profiles/NOTES_r04.md, tools/c4_timeline.sh).
            side = elems >= 2048 && elems < (1ull << 18);
            hipStream_t inv_st = st;
            if (side) {
                PG_HIP_TRY(hipEventRecord(e->ev_fork, st));  // the pre-pass reads the call's inputs: order it after the stream
                PG_HIP_TRY(hipStreamWaitEvent(e->side, e->ev_fork, 0));
                join.forked = true;
                inv_st = e->side;
            }
            if constexpr (pg::InvDense<GD>::ok) {
                O.inv_dense = e->d_prefix;
                O.inv_elems = elems;
                O.inv_in_place = side;  // else the pre-pass is through before the emitter starts: it leaves the inverses where it computed them
            }
            hipLaunchKernelGGL((pg::batch_invert_kernel<GD, GRP>), dim3(blocks), dim3(pg::kThreads), 0, inv_st, A, O, elems,
                               (uint32_t)groups, e->d_prefix);
            PG_HIP_TRY(hipGetLastError());
            if (side) PG_HIP_TRY(hipEventRecord(e->ev_inv, e->side));
        }
        const uint32_t max_blocks = (uint32_t)e->num_cus * (values_only ? PG_GRID_BLOCKS_PER_CU : PG_EMIT_GRID_BLOCKS_PER_CU);
        const dim3 egrid(O.tiles < max_blocks ? O.tiles : max_blocks);
        if (values_only) {
            if constexpr (ValuesMode<GD>::ok) hipLaunchKernelGGL((pg::emit_kernel<GD, pg::EMIT_VALUES>), egrid, dim3(pg::kThreads), 0, st, A, O);
            else return fail(PG_ERR_INVALID_ARGUMENT, "this gadget has no values-only emission");
        } else {
            hipLaunchKernelGGL(pg::emit_kernel<GD>, egrid, dim3(pg::kThreads), 0, st, A, O);
        }
        PG_HIP_TRY(hipGetLastError());
        if (side) {
            PG_HIP_TRY(hipStreamWaitEvent(st, e->ev_inv, 0));  // join
            join.joined = true;
        }
        return PG_OK;
    }
}

pg_status scalar_args(const pg_variable *a_var, const pg_scalar *a_val, const pg_variable *b_var,
                             const pg_scalar *b_val, pg_variable *res, pg::ScalarArgs *A) {
    PG_TRY(check_u64s(a_var, "first Variable array"));
    PG_TRY(check_u64s(b_var, "second Variable array"));
    PG_TRY(check_scalars(a_val, "first value array"));
    PG_TRY(check_scalars(b_val, "second value array"));
    PG_TRY(check_u64s(res, "d_result_vars", true));
    A->a_var = a_var;
    A->b_var = b_var;
    A->a_val = reinterpret_cast<const uint4 *>(a_val);
    A->b_val = reinterpret_cast<const uint4 *>(b_val);
    A->result_vars = res;
    A->err_mask = nullptr;
    return PG_OK;
}

// the launches of an asynchronous plan of a gadget with failing items (scratch ensured, arguments checked by the caller)
template <class PlanKernel>
pg_status error_plan_launch(pg_engine *e, PlanKernel kernel, const pg_scalar *d_value, uint64_t batch, uint64_t *d_row_off,
                            uint64_t *d_var_off, uint8_t *d_err_mask, hipStream_t st) {
    const uint32_t grid = (uint32_t)((batch + pg::kScanBlock - 1) / pg::kScanBlock);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(pg::kThreads), 0, st, reinterpret_cast<const uint4 *>(d_value), batch,
                       e->d_rows, e->d_vars, d_err_mask, e->d_err_count, plan_scan(e, batch, d_row_off, d_var_off, true));
    PG_HIP_TRY(hipGetLastError());
    return scan_counts(e, batch, d_row_off, d_var_off, nullptr, nullptr, st, true);
}

// out == NULL: asynchronous form (results through pg_plan_result)
template <class PlanKernel>
pg_status error_plan(pg_engine *e, PlanKernel kernel, const pg_scalar *d_value, uint64_t batch, uint64_t *d_row_off,
                            uint64_t *d_var_off, uint8_t *d_err_mask, pg_layout *out, uint64_t *err_count, void *stream) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (out) std::memset(out, 0, sizeof *out);
    if (err_count) *err_count = 0;
    if (batch == 0) {  // stream-ordered like every other write of the plan result
        PG_TRY(enter_stream(e, static_cast<hipStream_t>(stream)));
        StreamScope scope{e, static_cast<hipStream_t>(stream)};
        PG_HIP_TRY(hipMemsetAsync(e->h_plan, 0, sizeof(pg_engine::PlanResult), static_cast<hipStream_t>(stream)));
        return PG_OK;
    }
    PG_TRY(check_scalars(d_value, "value array"));
    PG_TRY(check_u64s(d_row_off, "d_row_off"));
    PG_TRY(check_u64s(d_var_off, "d_var_off"));
    PG_TRY(ensure_scratch(e, batch));
    hipStream_t st = static_cast<hipStream_t>(stream);
    PG_TRY(enter_stream(e, st));
    StreamScope scope{e, st};
    if (!out) return error_plan_launch(e, kernel, d_value, batch, d_row_off, d_var_off, d_err_mask, st);
    // e->d_err_count is zero between calls: whoever reads it (scan_final_kernel, the bulk decoder) leaves it so
    const uint32_t grid = (uint32_t)((batch + pg::kScanBlock - 1) / pg::kScanBlock);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(pg::kThreads), 0, st, reinterpret_cast<const uint4 *>(d_value), batch,
                       e->d_rows, e->d_vars, d_err_mask, e->d_err_count, plan_scan(e, batch, d_row_off, d_var_off, true));
    PG_HIP_TRY(hipGetLastError());
    PG_TRY(scan_counts(e, batch, d_row_off, d_var_off, &out->n_gates, &out->n_vars, st, true));
    const uint32_t errs = e->h_plan->errs;
    if (err_count) *err_count = errs;
    if (errs) return fail(PG_ERR_NON_EXISTING_INVERSE, std::to_string(errs) + " item(s) have no inverse (value = 0)");
    return PG_OK;
}

}  // namespace

extern "C" {

const char *pg_status_string(pg_status s) {
    switch (s) {
        case PG_OK: return "ok";
        case PG_ERR_NON_EXISTING_INVERSE: return "non-existing inverse";
        case PG_ERR_INVALID_ARGUMENT: return "invalid argument";
        case PG_ERR_NO_DEVICE: return "no usable gfx950 device";
        case PG_ERR_HIP: return "HIP runtime error";
        case PG_ERR_CAPACITY: return "composer capacity exceeded";
        case PG_ERR_BAD_ENCODING: return "encoding not below the modulus";
    }
    return "unknown status";
}

const char *pg_last_error(void) { return g_last_error.c_str(); }
const char *pg_build_arch(void) { return "gfx950"; }

pg_status pg_engine_create(int device, pg_engine **out) {
    if (!out) return fail(PG_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(PG_ERR_NO_DEVICE, "hipGetDeviceCount found no device: this library has no CPU path");
    if (device < 0 || device >= count) return fail(PG_ERR_INVALID_ARGUMENT, "device index out of range");
    hipDeviceProp_t prop;
    PG_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PG_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", library is built for gfx950 only");
    PG_HIP_TRY(hipSetDevice(device));
    pg_engine *e = new (std::nothrow) pg_engine();
    if (!e) return fail(PG_ERR_HIP, "out of host memory");
    e->device = device;
    e->num_cus = prop.multiProcessorCount;
    if (hipMalloc(&e->d_pow2, 256 * 2 * sizeof(uint4)) != hipSuccess) {
        delete e;
        return fail(PG_ERR_HIP, "hipMalloc(pow2 table) failed");
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&e->h_plan), sizeof(pg_engine::PlanResult), hipHostMallocDefault) != hipSuccess) {
        pg_engine_destroy(e);
        return fail(PG_ERR_HIP, "hipHostMalloc(plan result) failed");
    }
    *e->h_plan = pg_engine::PlanResult{0, 0, 0, 0};
    // the pre-pass is the critical path of a call with small items: highest priority, so its waves are placed ahead of
    // the rows-only emit launch it runs beside
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
#if defined(PG_SIDE_STREAM_NORMAL_PRIORITY)
    prio_hi = 0;
#endif
    if (hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_fork, kOrderingEvent) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_inv, kOrderingEvent) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_switch, kOrderingEvent) != hipSuccess) {
        pg_engine_destroy(e);
        return fail(PG_ERR_HIP, "creating the engine's side stream / events failed");
    }
    hipLaunchKernelGGL(pg::pow2_table_kernel, dim3(1), dim3(64), 0, nullptr, e->d_pow2);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) {
        pg_engine_destroy(e);
        return fail(PG_ERR_HIP, std::string("pow2 table kernel: ") + hipGetErrorString(err));
    }
    *out = e;
    return PG_OK;
}

void pg_engine_destroy(pg_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->d_pow2) (void)hipFree(e->d_pow2);
    if (e->d_rows) (void)hipFree(e->d_rows);
    if (e->d_vars) (void)hipFree(e->d_vars);
    if (e->d_blk_rows) (void)hipFree(e->d_blk_rows);
    if (e->d_blk_vars) (void)hipFree(e->d_blk_vars);
    if (e->d_err_count) (void)hipFree(e->d_err_count);
    if (e->d_blk_agg) (void)hipFree(e->d_blk_agg);
    if (e->d_prefix) (void)hipFree(e->d_prefix);
    if (e->side) { (void)hipStreamSynchronize(e->side); (void)hipStreamDestroy(e->side); }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_inv) (void)hipEventDestroy(e->ev_inv);
    if (e->ev_switch) (void)hipEventDestroy(e->ev_switch);
    if (e->h_plan) (void)hipHostFree(e->h_plan);
    delete e;
}

pg_status pg_engine_sync(pg_engine *e, void *stream) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    PG_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return PG_OK;
}

/* ---- encodings, in bulk ---------------------------------------------------- */
pg_status pg_scalars_from_canonical_batch(pg_engine *e, const void *d_bytes, uint64_t batch, pg_scalar *d_out, uint8_t *d_bad_mask,
                                          uint64_t *bad_count, void *stream) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    if (bad_count) *bad_count = 0;
    if (batch == 0) return PG_OK;
    PG_TRY(check_scalars(d_bytes, "d_bytes"));
    PG_TRY(check_scalars(d_out, "d_out"));
    PG_TRY(ensure_scratch(e, 1));
    hipStream_t st = static_cast<hipStream_t>(stream);
    PG_HIP_TRY(hipSetDevice(e->device));
    PG_HIP_TRY(hipMemsetAsync(e->d_err_count, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(pg::from_canonical_kernel, dim3((uint32_t)((batch + pg::kThreads - 1) / pg::kThreads)), dim3(pg::kThreads), 0, st,
                       static_cast<const uint4 *>(d_bytes), batch, reinterpret_cast<uint4 *>(d_out), d_bad_mask, e->d_err_count);
    PG_HIP_TRY(hipGetLastError());
    PG_HIP_TRY(hipMemcpyAsync(&e->h_plan->errs, e->d_err_count, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PG_HIP_TRY(hipMemsetAsync(e->d_err_count, 0, sizeof(uint32_t), st));  // zero between calls (see error_plan)
    PG_HIP_TRY(hipStreamSynchronize(st));
    const uint32_t bad = e->h_plan->errs;
    if (bad_count) *bad_count = bad;
    if (bad) return fail(PG_ERR_BAD_ENCODING, std::to_string(bad) + " encoding(s) are not below the modulus");
    return PG_OK;
}

pg_status pg_scalars_to_canonical_batch(pg_engine *e, const pg_scalar *d_scalars, uint64_t batch, void *d_bytes, void *stream) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    if (batch == 0) return PG_OK;
    PG_TRY(check_scalars(d_scalars, "d_scalars"));
    PG_TRY(check_scalars(d_bytes, "d_bytes"));
    PG_HIP_TRY(hipSetDevice(e->device));
    hipLaunchKernelGGL(pg::to_canonical_kernel, dim3((uint32_t)((batch + pg::kThreads - 1) / pg::kThreads)), dim3(pg::kThreads), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const uint4 *>(d_scalars), batch, static_cast<uint4 *>(d_bytes));
    PG_HIP_TRY(hipGetLastError());
    return PG_OK;
}

/* ---- host scalar helpers ------------------------------------------------ */
void pg_scalar_from_u64(uint64_t v, pg_scalar *out) { from_fr(pg::fr_from_u64(v), out); }
void pg_scalar_from_canonical(const uint64_t raw[4], pg_scalar *out) {
    from_fr(pg::fr_to_mont(pg::Fr{{raw[0], raw[1], raw[2], raw[3]}}), out);
}
void pg_scalar_to_canonical(const pg_scalar *s, uint64_t raw[4]) {
    pg::Fr c = pg::fr_from_mont(to_fr(s));
    std::memcpy(raw, c.l, sizeof c.l);
}
void pg_scalar_add(const pg_scalar *a, const pg_scalar *b, pg_scalar *out) { from_fr(pg::fr_add(to_fr(a), to_fr(b)), out); }
void pg_scalar_sub(const pg_scalar *a, const pg_scalar *b, pg_scalar *out) { from_fr(pg::fr_sub(to_fr(a), to_fr(b)), out); }
void pg_scalar_neg(const pg_scalar *a, pg_scalar *out) { from_fr(pg::fr_neg(to_fr(a)), out); }
void pg_scalar_mul(const pg_scalar *a, const pg_scalar *b, pg_scalar *out) { from_fr(pg::fr_mul(to_fr(a), to_fr(b)), out); }
pg_status pg_scalar_invert(const pg_scalar *a, pg_scalar *out) {
    if (!a || !out) return fail(PG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!is_reduced(to_fr(a))) return fail(PG_ERR_INVALID_ARGUMENT, "scalar is not a reduced BlsScalar");
    from_fr(pg::fr_invert_or_zero(to_fr(a)), out);
    return pg::fr_is_zero(to_fr(a)) ? fail(PG_ERR_NON_EXISTING_INVERSE, "zero has no inverse") : PG_OK;
}
pg_status pg_scalar_invert_fermat(const pg_scalar *a, pg_scalar *out) {
    if (!a || !out) return fail(PG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!is_reduced(to_fr(a))) return fail(PG_ERR_INVALID_ARGUMENT, "scalar is not a reduced BlsScalar");
    from_fr(pg::fr_invert_fermat(to_fr(a)), out);
    return pg::fr_is_zero(to_fr(a)) ? fail(PG_ERR_NON_EXISTING_INVERSE, "zero has no inverse") : PG_OK;
}
uint64_t pg_bits_count(const pg_scalar *s) { return pg::bits_count(to_fr(s)); }
uint64_t pg_num_bits_closest_power_of_two(const pg_scalar *s) { return pg::num_bits_closest_power_of_two(to_fr(s)); }

/* ---- range_check ---------------------------------------------------------- */
pg_status pg_range_check_layout(const pg_scalar *min_range, const pg_scalar *max_range, uint64_t batch, pg_layout *out) {
    if (!min_range || !max_range || !out) return fail(PG_ERR_INVALID_ARGUMENT, "NULL argument");
    pg::Fr mn = to_fr(min_range), mx = to_fr(max_range);
    if (!is_reduced(mn) || !is_reduced(mx)) return fail(PG_ERR_INVALID_ARGUMENT, "bound is not a reduced BlsScalar");
    // range.rs:87-90
    uint64_t n = pg::num_bits_closest_power_of_two(pg::fr_sub(mx, pg::fr_one()));
    out->num_bits = n;
    out->gates_per_item = 4 * n + 11;
    out->vars_per_item = 2 * n + 524;  // 2n+523 of range_check + 1 of allocate
    out->n_gates = out->gates_per_item * batch;
    out->n_vars = out->vars_per_item * batch;
    return PG_OK;
}

static pg_status range_check_common(pg_engine *e, const pg_scalar *min_range, const pg_scalar *max_range,
                                    const pg_variable *d_witness_var, const pg_scalar *d_witness, uint64_t batch,
                                    uint64_t gate_base, uint64_t var_base, const pg_columns *out,
                                    pg_variable *d_result_vars, void *stream, bool values_only = false, uint32_t stride_rows = 0) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    pg_layout lay;
    PG_TRY(pg_range_check_layout(min_range, max_range, batch, &lay));
    if (batch == 0) return PG_OK;
    PG_TRY(check_scalars(d_witness, "d_witness"));
    PG_TRY(check_u64s(d_result_vars, "d_result_vars", true));
    PG_TRY(check_columns(out));
    if (lay.num_bits < 2 || lay.num_bits > 255) return fail(PG_ERR_INVALID_ARGUMENT, "ladder length out of range");
    pg::RangeCheckGD::Args A{};
    A.min_range = to_fr(min_range);
    A.max_range = to_fr(max_range);
    A.n = (uint32_t)lay.num_bits;
    A.witness = reinterpret_cast<const uint4 *>(d_witness);
    A.witness_vars = d_witness_var;
    A.result_vars = d_result_vars;
    A.pow2 = e->d_pow2;
    return launch<pg::RangeCheckGD>(e, A, out, batch, gate_base, var_base, 0, nullptr, nullptr, stream, nullptr, values_only, stride_rows);
}

// a witness refresh's `out`: only var_values is written (EMIT_VALUES); the row pointers merely have to pass the checks
static pg_columns values_columns(pg_scalar *d_var_values) {
    pg_columns c;
    c.q_m = c.q_l = c.q_r = c.q_o = c.q_c = c.var_values = d_var_values;
    c.w_l = c.w_r = c.w_o = reinterpret_cast<uint64_t *>(d_var_values);
    return c;
}

pg_status pg_range_check_values_batch(pg_engine *e, const pg_scalar *min_range, const pg_scalar *max_range, const pg_scalar *d_witness,
                                      uint64_t batch, pg_scalar *d_var_values, void *stream) {
    const pg_columns c = values_columns(d_var_values);
    return range_check_common(e, min_range, max_range, nullptr, d_witness, batch, 0, 0, &c, nullptr, stream, true);
}

pg_status pg_range_check_structure_batch(pg_engine *e, const pg_scalar *min_range, const pg_scalar *max_range, uint64_t batch,
                                         uint64_t gate_base, uint64_t var_base, const pg_columns *out, void *stream) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    pg_layout lay;
    PG_TRY(pg_range_check_layout(min_range, max_range, batch, &lay));
    if (batch == 0) return PG_OK;
    if (!out) return fail(PG_ERR_INVALID_ARGUMENT, "columns are NULL");
    pg_columns c = *out;
    if (!c.var_values) c.var_values = c.q_m;  // never written: only has to pass the pointer checks
    PG_TRY(check_columns(&c));
    if (lay.num_bits < 2 || lay.num_bits > 255) return fail(PG_ERR_INVALID_ARGUMENT, "ladder length out of range");
    if ((batch + pg::RangeCheckGD::W - 1) / pg::RangeCheckGD::W > 0xffffffffull) return fail(PG_ERR_INVALID_ARGUMENT, "batch too large for one call");
    PG_HIP_TRY(hipSetDevice(e->device));
    pg::RangeCheckGD::Args A{};
    A.min_range = to_fr(min_range);
    A.max_range = to_fr(max_range);
    A.n = (uint32_t)lay.num_bits;
    A.pow2 = e->d_pow2;
    const pg::EmitOut O = make_out(&c, batch, pg::RangeCheckGD::W, gate_base, var_base, 0, nullptr, nullptr);
    const uint32_t max_blocks = (uint32_t)e->num_cus * PG_GRID_BLOCKS_PER_CU;
    hipLaunchKernelGGL((pg::emit_kernel<pg::RangeCheckGD, pg::EMIT_STRUCTURE>), dim3(O.tiles < max_blocks ? O.tiles : max_blocks), dim3(pg::kThreads),
                       0, static_cast<hipStream_t>(stream), A, O);
    PG_HIP_TRY(hipGetLastError());
    return PG_OK;
}

pg_status pg_range_check_batch(pg_engine *e, const pg_scalar *min_range, const pg_scalar *max_range,
                               const pg_scalar *d_witness, uint64_t batch, uint64_t gate_base, uint64_t var_base,
                               const pg_columns *out, pg_variable *d_result_vars, void *stream) {
    return range_check_common(e, min_range, max_range, nullptr, d_witness, batch, gate_base, var_base, out, d_result_vars, stream);
}

pg_status pg_range_check_allocated_batch(pg_engine *e, const pg_scalar *min_range, const pg_scalar *max_range,
                                         const pg_variable *d_witness_var, const pg_scalar *d_witness, uint64_t batch,
                                         uint64_t gate_base, uint64_t var_base, const pg_columns *out,
                                         pg_variable *d_result_vars, void *stream) {
    if (batch) PG_TRY(check_u64s(d_witness_var, "d_witness_var"));
    return range_check_common(e, min_range, max_range, d_witness_var, d_witness, batch, gate_base, var_base, out,
                              d_result_vars, stream);
}

/* ---- scalar_decomposition_gadget -------------------------------------------- */
pg_status pg_scalar_decomposition_layout(uint64_t num_bits, uint64_t batch, pg_layout *out) {
    if (!out) return fail(PG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (num_bits > 256) return fail(PG_ERR_INVALID_ARGUMENT, "num_bits > 256 (the reference panics: src/range.rs:134)");
    out->num_bits = num_bits;
    out->gates_per_item = 2 * num_bits + 4;
    out->vars_per_item = num_bits + 260;
    out->n_gates = out->gates_per_item * batch;
    out->n_vars = out->vars_per_item * batch;
    return PG_OK;
}

static pg_status decomposition_common(pg_engine *e, uint64_t num_bits, const pg_variable *d_witness_var,
                                      const pg_scalar *d_witness, uint64_t batch, uint64_t gate_base, uint64_t var_base,
                                      const pg_columns *out, pg_variable *d_result_vars, void *stream, bool values_only) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    pg_layout lay;
    PG_TRY(pg_scalar_decomposition_layout(num_bits, batch, &lay));
    if (batch == 0) return PG_OK;
    PG_TRY(check_u64s(d_witness_var, "d_witness_var"));
    PG_TRY(check_scalars(d_witness, "d_witness"));
    PG_TRY(check_u64s(d_result_vars, "d_result_vars", true));
    PG_TRY(check_columns(out));
    pg::DecompositionGD::Args A{};
    A.n = (uint32_t)num_bits;
    A.witness = reinterpret_cast<const uint4 *>(d_witness);
    A.witness_vars = d_witness_var;
    A.result_vars = d_result_vars;
    A.pow2 = e->d_pow2;
    return launch<pg::DecompositionGD>(e, A, out, batch, gate_base, var_base, 0, nullptr, nullptr, stream, nullptr, values_only);
}

pg_status pg_scalar_decomposition_batch(pg_engine *e, uint64_t num_bits, const pg_variable *d_witness_var,
                                        const pg_scalar *d_witness, uint64_t batch, uint64_t gate_base, uint64_t var_base,
                                        const pg_columns *out, pg_variable *d_result_vars, void *stream) {
    return decomposition_common(e, num_bits, d_witness_var, d_witness, batch, gate_base, var_base, out, d_result_vars, stream, false);
}

/* ---- max_bound ------------------------------------------------------------ */
pg_status pg_max_bound_layout(const pg_scalar *max_range, uint64_t batch, pg_layout *out) {
    if (!max_range || !out) return fail(PG_ERR_INVALID_ARGUMENT, "NULL argument");
    pg::Fr mx = to_fr(max_range);
    if (!is_reduced(mx)) return fail(PG_ERR_INVALID_ARGUMENT, "bound is not a reduced BlsScalar");
    uint64_t n = pg::num_bits_closest_power_of_two(pg::fr_sub(mx, pg::fr_one()));
    out->num_bits = n;
    out->gates_per_item = 2 * n + 5;
    out->vars_per_item = n + 262;  // n+261 of max_bound + 1 of allocate
    out->n_gates = out->gates_per_item * batch;
    out->n_vars = out->vars_per_item * batch;
    return PG_OK;
}

static pg_status max_bound_common(pg_engine *e, const pg_scalar *max_range, const pg_variable *d_witness_var,
                                  const pg_scalar *d_witness, uint64_t batch, uint64_t gate_base, uint64_t var_base,
                                  const pg_columns *out, pg_variable *d_result_vars, void *stream, bool values_only = false,
                                  uint32_t stride_rows = 0) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    pg_layout lay;
    PG_TRY(pg_max_bound_layout(max_range, batch, &lay));
    if (batch == 0) return PG_OK;
    PG_TRY(check_scalars(d_witness, "d_witness"));
    PG_TRY(check_u64s(d_result_vars, "d_result_vars", true));
    PG_TRY(check_columns(out));
    pg::MaxBoundGD<false>::Args A{};
    A.max_range = to_fr(max_range);
    A.n = (uint32_t)lay.num_bits;
    A.witness = reinterpret_cast<const uint4 *>(d_witness);
    A.witness_vars = d_witness_var;
    A.result_vars = d_result_vars;
    A.pow2 = e->d_pow2;
    return launch<pg::MaxBoundGD<false>>(e, A, out, batch, gate_base, var_base, 0, nullptr, nullptr, stream, nullptr, values_only, stride_rows);
}

pg_status pg_max_bound_values_batch(pg_engine *e, const pg_scalar *max_range, const pg_scalar *d_witness, uint64_t batch,
                                    pg_scalar *d_var_values, void *stream) {
    const pg_columns c = values_columns(d_var_values);
    return max_bound_common(e, max_range, nullptr, d_witness, batch, 0, 0, &c, nullptr, stream, true);
}

pg_status pg_max_bound_batch(pg_engine *e, const pg_scalar *max_range, const pg_scalar *d_witness, uint64_t batch,
                             uint64_t gate_base, uint64_t var_base, const pg_columns *out, pg_variable *d_result_vars,
                             void *stream) {
    return max_bound_common(e, max_range, nullptr, d_witness, batch, gate_base, var_base, out, d_result_vars, stream);
}

pg_status pg_max_bound_allocated_batch(pg_engine *e, const pg_scalar *max_range, const pg_variable *d_witness_var,
                                       const pg_scalar *d_witness, uint64_t batch, uint64_t gate_base, uint64_t var_base,
                                       const pg_columns *out, pg_variable *d_result_vars, void *stream) {
    if (batch) PG_TRY(check_u64s(d_witness_var, "d_witness_var"));
    return max_bound_common(e, max_range, d_witness_var, d_witness, batch, gate_base, var_base, out, d_result_vars, stream);
}

static pg_status max_bound_ragged_plan_common(pg_engine *e, const pg_scalar *d_max_range, uint64_t batch, uint32_t *d_num_bits,
                                              uint64_t *d_row_off, uint64_t *d_var_off, pg_layout *out, void *stream) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "NULL argument");
    if (out) std::memset(out, 0, sizeof *out);
    if (batch == 0) {
        PG_TRY(enter_stream(e, static_cast<hipStream_t>(stream)));
        StreamScope scope{e, static_cast<hipStream_t>(stream)};
        PG_HIP_TRY(hipMemsetAsync(e->h_plan, 0, sizeof(pg_engine::PlanResult), static_cast<hipStream_t>(stream)));
        return PG_OK;
    }
    PG_TRY(check_scalars(d_max_range, "d_max_range"));
    if (!d_num_bits || !aligned(d_num_bits, 4)) return fail(PG_ERR_INVALID_ARGUMENT, "d_num_bits NULL or misaligned");
    PG_TRY(check_u64s(d_row_off, "d_row_off"));
    PG_TRY(check_u64s(d_var_off, "d_var_off"));
    PG_TRY(ensure_scratch(e, batch));
    hipStream_t st = static_cast<hipStream_t>(stream);
    PG_TRY(enter_stream(e, st));
    StreamScope scope{e, st};
    const uint32_t grid = (uint32_t)((batch + pg::kScanBlock - 1) / pg::kScanBlock);
    hipLaunchKernelGGL(pg::max_bound_plan_kernel, dim3(grid), dim3(pg::kThreads), 0, st,
                       reinterpret_cast<const uint4 *>(d_max_range), batch, e->d_pow2, d_num_bits, e->d_rows, e->d_vars,
                       plan_scan(e, batch, d_row_off, d_var_off, false));
    PG_HIP_TRY(hipGetLastError());
    // a max_bound plan has no failing items: the scan writes errs = 0 with the totals, in stream order
    if (!out) return scan_counts(e, batch, d_row_off, d_var_off, nullptr, nullptr, st, false);
    PG_TRY(scan_counts(e, batch, d_row_off, d_var_off, &out->n_gates, &out->n_vars, st, false));
    return PG_OK;
}

pg_status pg_max_bound_ragged_plan(pg_engine *e, const pg_scalar *d_max_range, uint64_t batch, uint32_t *d_num_bits,
                                   uint64_t *d_row_off, uint64_t *d_var_off, pg_layout *out, void *stream) {
    if (!out) return fail(PG_ERR_INVALID_ARGUMENT, "NULL argument");
    return max_bound_ragged_plan_common(e, d_max_range, batch, d_num_bits, d_row_off, d_var_off, out, stream);
}

pg_status pg_max_bound_ragged_plan_async(pg_engine *e, const pg_scalar *d_max_range, uint64_t batch, uint32_t *d_num_bits,
                                         uint64_t *d_row_off, uint64_t *d_var_off, void *stream) {
    return max_bound_ragged_plan_common(e, d_max_range, batch, d_num_bits, d_row_off, d_var_off, nullptr, stream);
}

pg_status pg_plan_result(pg_engine *e, pg_layout *out, uint64_t *err_count) {
    if (!e || !out) return fail(PG_ERR_INVALID_ARGUMENT, "NULL argument");
    std::memset(out, 0, sizeof *out);
    PG_TRY(plan_totals(e, &out->n_gates, &out->n_vars));
    if (err_count) *err_count = e->h_plan->errs;
    return e->h_plan->errs ? fail(PG_ERR_NON_EXISTING_INVERSE, std::to_string(e->h_plan->errs) + " item(s) have no inverse")
                           : PG_OK;
}

static pg_status max_bound_ragged_common(pg_engine *e, const pg_scalar *d_max_range, const pg_scalar *d_witness, uint64_t batch,
                                         const uint32_t *d_num_bits, const uint64_t *d_row_off, const uint64_t *d_var_off,
                                         uint64_t gate_base, uint64_t var_base, const pg_columns *out,
                                         pg_variable *d_result_vars, void *stream, bool values_only) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    if (batch == 0) return PG_OK;
    PG_TRY(check_scalars(d_max_range, "d_max_range"));
    PG_TRY(check_scalars(d_witness, "d_witness"));
    if (!d_num_bits) return fail(PG_ERR_INVALID_ARGUMENT, "d_num_bits is NULL");
    PG_TRY(check_u64s(d_row_off, "d_row_off"));
    PG_TRY(check_u64s(d_var_off, "d_var_off"));
    PG_TRY(check_u64s(d_result_vars, "d_result_vars", true));
    PG_TRY(check_columns(out));
    pg::MaxBoundGD<true>::Args A{};
    A.max_range_v = reinterpret_cast<const uint4 *>(d_max_range);
    A.num_bits_v = d_num_bits;
    A.witness = reinterpret_cast<const uint4 *>(d_witness);
    A.result_vars = d_result_vars;
    A.pow2 = e->d_pow2;
    return launch<pg::MaxBoundGD<true>>(e, A, out, batch, gate_base, var_base, 0, d_row_off, d_var_off, stream, nullptr, values_only);
}

pg_status pg_max_bound_ragged_batch(pg_engine *e, const pg_scalar *d_max_range, const pg_scalar *d_witness, uint64_t batch,
                                    const uint32_t *d_num_bits, const uint64_t *d_row_off, const uint64_t *d_var_off,
                                    uint64_t gate_base, uint64_t var_base, const pg_columns *out,
                                    pg_variable *d_result_vars, void *stream) {
    return max_bound_ragged_common(e, d_max_range, d_witness, batch, d_num_bits, d_row_off, d_var_off, gate_base, var_base, out,
                                   d_result_vars, stream, false);
}

pg_status pg_max_bound_ragged_values_batch(pg_engine *e, const pg_scalar *d_max_range, const pg_scalar *d_witness, uint64_t batch,
                                           const uint32_t *d_num_bits, const uint64_t *d_row_off, const uint64_t *d_var_off,
                                           pg_scalar *d_var_values, void *stream) {
    const pg_columns c = values_columns(d_var_values);
    return max_bound_ragged_common(e, d_max_range, d_witness, batch, d_num_bits, d_row_off, d_var_off, 0, 0, &c, nullptr, stream, true);
}

/* ---- scalar gadgets ------------------------------------------------------- */
}  // extern "C"

// the three gadgets on two existing Variables each; values_only: a witness refresh (pg_composer_clear_witness)
template <class GD>
static pg_status two_input_common(pg_engine *e, const pg_variable *d_a_var, const pg_scalar *d_a_val, const pg_variable *d_b_var,
                                  const pg_scalar *d_b_val, uint64_t batch, uint64_t gate_base, uint64_t var_base,
                                  const pg_columns *out, pg_variable *d_result_vars, void *stream, bool values_only) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    if (batch == 0) return PG_OK;
    pg::ScalarArgs A;
    PG_TRY(scalar_args(d_a_var, d_a_val, d_b_var, d_b_val, d_result_vars, &A));
    PG_TRY(check_columns(out));
    return launch<GD>(e, A, out, batch, gate_base, var_base, 0, nullptr, nullptr, stream, nullptr, values_only);
}

extern "C" {

pg_status pg_conditionally_select_zero_batch(pg_engine *e, const pg_variable *d_x_var, const pg_scalar *d_x_val,
                                             const pg_variable *d_select_var, const pg_scalar *d_select_val, uint64_t batch,
                                             uint64_t gate_base, uint64_t var_base, const pg_columns *out,
                                             pg_variable *d_result_vars, void *stream) {
    return two_input_common<pg::SelectZeroGD>(e, d_x_var, d_x_val, d_select_var, d_select_val, batch, gate_base, var_base, out,
                                              d_result_vars, stream, false);
}

pg_status pg_conditionally_select_one_batch(pg_engine *e, const pg_variable *d_y_var, const pg_scalar *d_y_val,
                                            const pg_variable *d_selector_var, const pg_scalar *d_selector_val,
                                            uint64_t batch, uint64_t gate_base, uint64_t var_base, const pg_columns *out,
                                            pg_variable *d_result_vars, void *stream) {
    return two_input_common<pg::SelectOneGD>(e, d_y_var, d_y_val, d_selector_var, d_selector_val, batch, gate_base, var_base, out,
                                             d_result_vars, stream, false);
}

pg_status pg_maybe_equal_batch(pg_engine *e, const pg_variable *d_a_var, const pg_scalar *d_a_val, const pg_variable *d_b_var,
                               const pg_scalar *d_b_val, uint64_t batch, uint64_t gate_base, uint64_t var_base,
                               const pg_columns *out, pg_variable *d_result_vars, void *stream) {
    return two_input_common<pg::MaybeEqualGD>(e, d_a_var, d_a_val, d_b_var, d_b_val, batch, gate_base, var_base, out, d_result_vars,
                                              stream, false);
}

pg_status pg_is_non_zero_plan(pg_engine *e, const pg_scalar *d_value_assigned, uint64_t batch, uint64_t *d_row_off,
                              uint64_t *d_var_off, uint8_t *d_err_mask, pg_layout *out, uint64_t *err_count, void *stream) {
    if (!out) return fail(PG_ERR_INVALID_ARGUMENT, "NULL argument");
    return error_plan(e, pg::is_non_zero_plan_kernel, d_value_assigned, batch, d_row_off, d_var_off, d_err_mask, out,
                      err_count, stream);
}

static pg_status is_non_zero_common(pg_engine *e, const pg_variable *d_var, const pg_scalar *d_value_assigned, uint64_t batch,
                                    const uint64_t *d_row_off, const uint64_t *d_var_off, uint64_t gate_base, uint64_t var_base,
                                    pg_variable zero_var, const pg_columns *out, void *stream, bool values_only) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    if (batch == 0) return PG_OK;
    PG_TRY(check_u64s(d_var, "d_var"));
    PG_TRY(check_scalars(d_value_assigned, "d_value_assigned"));
    PG_TRY(check_u64s(d_row_off, "d_row_off"));
    PG_TRY(check_u64s(d_var_off, "d_var_off"));
    PG_TRY(check_columns(out));
    pg::ScalarArgs A{};
    A.a_var = d_var;
    A.b_val = reinterpret_cast<const uint4 *>(d_value_assigned);
    return launch<pg::IsNonZeroGD>(e, A, out, batch, gate_base, var_base, zero_var, d_row_off, d_var_off, stream, nullptr, values_only);
}

pg_status pg_is_non_zero_batch(pg_engine *e, const pg_variable *d_var, const pg_scalar *d_value_assigned, uint64_t batch,
                               const uint64_t *d_row_off, const uint64_t *d_var_off, uint64_t gate_base, uint64_t var_base,
                               pg_variable zero_var, const pg_columns *out, void *stream) {
    return is_non_zero_common(e, d_var, d_value_assigned, batch, d_row_off, d_var_off, gate_base, var_base, zero_var, out, stream, false);
}

pg_status pg_scalar_mix_plan(pg_engine *e, const pg_scalar *d_v, uint64_t batch, uint64_t *d_row_off, uint64_t *d_var_off,
                             uint8_t *d_err_mask, pg_layout *out, uint64_t *err_count, void *stream) {
    if (!out) return fail(PG_ERR_INVALID_ARGUMENT, "NULL argument");
    return error_plan(e, pg::scalar_mix_plan_kernel, d_v, batch, d_row_off, d_var_off, d_err_mask, out, err_count, stream);
}

pg_status pg_scalar_mix_plan_async(pg_engine *e, const pg_scalar *d_v, uint64_t batch, uint64_t *d_row_off,
                                   uint64_t *d_var_off, uint8_t *d_err_mask, void *stream) {
    return error_plan(e, pg::scalar_mix_plan_kernel, d_v, batch, d_row_off, d_var_off, d_err_mask, nullptr, nullptr, stream);
}

static pg_status scalar_mix_common(pg_engine *e, const pg_scalar *d_v, const pg_scalar *d_y, const pg_scalar *d_s,
                                   const pg_scalar *d_a, const pg_scalar *d_b, uint64_t batch, const uint64_t *d_row_off,
                                   const uint64_t *d_var_off, uint64_t gate_base, uint64_t var_base, pg_variable zero_var,
                                   const pg_columns *out, pg_variable *d_result_vars, void *stream, bool values_only) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    if (batch == 0) return PG_OK;
    const pg_scalar *in[5] = {d_v, d_y, d_s, d_a, d_b};
    for (const pg_scalar *p : in) PG_TRY(check_scalars(p, "input array"));
    PG_TRY(check_u64s(d_row_off, "d_row_off"));
    PG_TRY(check_u64s(d_var_off, "d_var_off"));
    PG_TRY(check_u64s(d_result_vars, "d_result_vars", true));
    PG_TRY(check_columns(out));
    pg::ScalarMixArgs A{};
    A.v = reinterpret_cast<const uint4 *>(d_v);
    A.y = reinterpret_cast<const uint4 *>(d_y);
    A.s = reinterpret_cast<const uint4 *>(d_s);
    A.a = reinterpret_cast<const uint4 *>(d_a);
    A.b = reinterpret_cast<const uint4 *>(d_b);
    A.result_vars = d_result_vars;
    return launch<pg::ScalarMixGD>(e, A, out, batch, gate_base, var_base, zero_var, d_row_off, d_var_off, stream, nullptr, values_only);
}

pg_status pg_scalar_mix_batch(pg_engine *e, const pg_scalar *d_v, const pg_scalar *d_y, const pg_scalar *d_s,
                              const pg_scalar *d_a, const pg_scalar *d_b, uint64_t batch, const uint64_t *d_row_off,
                              const uint64_t *d_var_off, uint64_t gate_base, uint64_t var_base, pg_variable zero_var,
                              const pg_columns *out, pg_variable *d_result_vars, void *stream) {
    return scalar_mix_common(e, d_v, d_y, d_s, d_a, d_b, batch, d_row_off, d_var_off, gate_base, var_base, zero_var, out, d_result_vars,
                             stream, false);
}

static pg_status scalar_mix_planned_common(pg_engine *e, const pg_scalar *d_v, const pg_scalar *d_y, const pg_scalar *d_s,
                                           const pg_scalar *d_a, const pg_scalar *d_b, uint64_t batch, uint64_t *d_row_off,
                                           uint64_t *d_var_off, uint8_t *d_err_mask, uint64_t gate_base, uint64_t var_base,
                                           pg_variable zero_var, const pg_columns *out, pg_variable *d_result_vars, void *stream,
                                           bool values_only) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    if (batch == 0) return pg_scalar_mix_plan_async(e, d_v, 0, d_row_off, d_var_off, d_err_mask, stream);
    const pg_scalar *in[5] = {d_v, d_y, d_s, d_a, d_b};
    for (const pg_scalar *p : in) PG_TRY(check_scalars(p, "input array"));
    PG_TRY(check_u64s(d_row_off, "d_row_off"));
    PG_TRY(check_u64s(d_var_off, "d_var_off"));
    PG_TRY(check_u64s(d_result_vars, "d_result_vars", true));
    PG_TRY(check_columns(out));
    PG_TRY(ensure_scratch(e, batch));
    pg::ScalarMixArgs A{};
    A.v = reinterpret_cast<const uint4 *>(d_v);
    A.y = reinterpret_cast<const uint4 *>(d_y);
    A.s = reinterpret_cast<const uint4 *>(d_s);
    A.a = reinterpret_cast<const uint4 *>(d_a);
    A.b = reinterpret_cast<const uint4 *>(d_b);
    A.result_vars = d_result_vars;
    pg::MixPlan P{};  // the call plans itself: the launch that inverts also makes the prefix sums (scalar_gadgets.hpp)
    P.agg = e->d_blk_agg;
    P.cap = (uint32_t)e->scratch_blocks;
    P.row_off = d_row_off;
    P.var_off = d_var_off;
    P.err_mask = d_err_mask;
    P.host = e->h_plan;
    return launch<pg::ScalarMixGD>(e, A, out, batch, gate_base, var_base, zero_var, d_row_off, d_var_off, stream, &P, values_only);
}

pg_status pg_scalar_mix_planned_batch(pg_engine *e, const pg_scalar *d_v, const pg_scalar *d_y, const pg_scalar *d_s,
                                      const pg_scalar *d_a, const pg_scalar *d_b, uint64_t batch, uint64_t *d_row_off,
                                      uint64_t *d_var_off, uint8_t *d_err_mask, uint64_t gate_base, uint64_t var_base,
                                      pg_variable zero_var, const pg_columns *out, pg_variable *d_result_vars, void *stream) {
    return scalar_mix_planned_common(e, d_v, d_y, d_s, d_a, d_b, batch, d_row_off, d_var_off, d_err_mask, gate_base, var_base, zero_var,
                                     out, d_result_vars, stream, false);
}

pg_status pg_scalar_mix_values_batch(pg_engine *e, const pg_scalar *d_v, const pg_scalar *d_y, const pg_scalar *d_s,
                                     const pg_scalar *d_a, const pg_scalar *d_b, uint64_t batch, uint64_t *d_row_off,
                                     uint64_t *d_var_off, uint8_t *d_err_mask, pg_scalar *d_var_values, void *stream) {
    const pg_columns c = values_columns(d_var_values);
    return scalar_mix_planned_common(e, d_v, d_y, d_s, d_a, d_b, batch, d_row_off, d_var_off, d_err_mask, 0, 0, 0, &c, nullptr, stream,
                                     true);
}

#if defined(PG_MIX_STAMPS)  // timing build only: ticks (100 MHz) the fused mix's waves spent per phase since the last call; not in the header
extern "C" int pg_debug_mix_phases(unsigned long long out[12]) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pg::g_mix_phase_ticks), 12 * sizeof(unsigned long long)) != hipSuccess) return 1;
    const unsigned long long zero[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(pg::g_mix_phase_ticks), zero, sizeof zero) == hipSuccess ? 0 : 1;
}
#endif

pg_status pg_columns_slab_layout(uint64_t n_gates, uint64_t n_vars, uint64_t stride_bytes, uint64_t offsets[9], uint64_t *total_bytes) {
    if (!offsets || !total_bytes) return fail(PG_ERR_INVALID_ARGUMENT, "offsets / total_bytes is NULL");
    if (n_gates > (1ull << 40) || n_vars > (1ull << 40) || stride_bytes > (1ull << 44))
        return fail(PG_ERR_INVALID_ARGUMENT, "circuit or stride too large");
    constexpr uint64_t al = 2ull << 20;
    auto up = [](uint64_t x) { return (x + al - 1) / al * al; };
    const uint64_t ssz = up(n_gates * 32), wsz = up(n_gates * 8), vsz = up(n_vars * 32);
    const uint64_t stride = up(stride_bytes) > ssz ? up(stride_bytes) : ssz;
    for (int c = 0; c < 5; c++) offsets[c] = (uint64_t)c * stride;
    const uint64_t tail = 4 * stride + ssz;
    for (int c = 0; c < 3; c++) offsets[5 + c] = tail + (uint64_t)c * wsz;
    offsets[8] = tail + 3 * wsz;
    *total_bytes = tail + 3 * wsz + vsz;
    return PG_OK;
}

pg_status pg_fill_bytes(pg_engine *e, void *d_dst, uint64_t bytes, uint32_t streams, uint64_t pattern, void *stream) {
    if (!e) return fail(PG_ERR_INVALID_ARGUMENT, "engine is NULL");
    if (!d_dst || !aligned(d_dst, 16) || (bytes & 15)) return fail(PG_ERR_INVALID_ARGUMENT, "dst/bytes not 16-byte aligned");
    if (bytes == 0) return PG_OK;
    PG_HIP_TRY(hipSetDevice(e->device));
    if (streams == 0) {  // short-lived workgroups, 8 KiB each, two resident per CU
        const uint64_t blocks = (bytes / 16 + pg::kFillOneshotUnits - 1) / pg::kFillOneshotUnits;
        if (blocks > 0x7fffffffull) return fail(PG_ERR_INVALID_ARGUMENT, "buffer too large for one launch");
        PG_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pg::fill_oneshot_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)pg::kFillOneshotLds));
        hipLaunchKernelGGL(pg::fill_oneshot_kernel, dim3((uint32_t)blocks), dim3(pg::kThreads), pg::kFillOneshotLds, static_cast<hipStream_t>(stream),
                           static_cast<uint4 *>(d_dst), bytes / 16, pattern);
        PG_HIP_TRY(hipGetLastError());
        return PG_OK;
    }
    if (streams < 1 || streams > 16) return fail(PG_ERR_INVALID_ARGUMENT, "streams must be in [0, 16]");
    uint64_t pieces = (bytes / 16 / streams + 65535) / 65536;
    const uint64_t cap = (uint64_t)e->num_cus * PG_GRID_BLOCKS_PER_CU;
    if (pieces < 1) pieces = 1;  // (fewer units than streams: the remainder loop of workgroup 0 writes them)
    hipLaunchKernelGGL(pg::fill_kernel, dim3((uint32_t)(pieces < cap ? pieces : cap)), dim3(pg::kThreads), 0,
                       static_cast<hipStream_t>(stream), static_cast<uint4 *>(d_dst), bytes / 16, streams, pattern);
    PG_HIP_TRY(hipGetLastError());
    return PG_OK;
}

pg_status pg_fill_columns(pg_engine *e, const pg_columns *out, uint64_t n_gates, uint64_t n_vars, uint64_t rows_per_tile,
                          uint64_t pattern, void *stream) {
    if (!e || !out) return fail(PG_ERR_INVALID_ARGUMENT, "NULL argument");
    PG_HIP_TRY(hipSetDevice(e->device));
    if (n_gates == 0 && n_vars == 0) return PG_OK;
    PG_TRY(check_columns(out));
    for (const void *p : {(const void *)out->w_l, (const void *)out->w_r, (const void *)out->w_o})
        if (!aligned(p, 16)) return fail(PG_ERR_INVALID_ARGUMENT, "pg_fill_columns: wire columns must be 16-byte aligned");
    if (rows_per_tile == 0) rows_per_tile = 32768;  // (about the emitters' tiles: 32 x 1031 rows of range_check, 16 x ~511 of max_bound)
    if (rows_per_tile % 8) return fail(PG_ERR_INVALID_ARGUMENT, "rows_per_tile must be a multiple of 8 (whole lines of every column)");
    uint64_t tiles = n_gates ? (n_gates + rows_per_tile - 1) / rows_per_tile : 1;
    if (tiles > 0xffffffffull) return fail(PG_ERR_INVALID_ARGUMENT, "too many tiles for one call");
    const uint64_t vars_per_tile = ((n_vars + tiles - 1) / tiles + 3) / 4 * 4;
    pg::EmitOut O = make_out(out, 0, 1, 0, 0, 0, nullptr, nullptr);
    O.tiles = (uint32_t)tiles;
    const uint64_t cap = (uint64_t)e->num_cus * PG_GRID_BLOCKS_PER_CU;
    hipLaunchKernelGGL(pg::fill_columns_kernel, dim3((uint32_t)(tiles < cap ? tiles : cap)), dim3(pg::kThreads), 0,
                       static_cast<hipStream_t>(stream), O, n_gates, n_vars, rows_per_tile, vars_per_tile, pattern);
    PG_HIP_TRY(hipGetLastError());
    return PG_OK;
}

}  // extern "C"

#include "capi_composer.inc"
#include "capi_dist.inc"
