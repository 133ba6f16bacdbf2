// invert.hpp -- batched field inversion pre-pass (Montgomery's trick), gfx950.
//
// Every gadget on this path that inverts (maybe_equal: (a-b)^-1, scalar.rs:121-122; is_non_zero: value^-1,
// scalar.rs:73) would pay a full inversion (fr_invert_or_zero: 600 division steps) per item.  The pre-pass
// computes all of a call's inverses first: every lane owns `groups * GRP` elements (strided by the grid, so loads
// coalesce), multiplies them up, inverts ONE product, and unwinds:
//     inv(x_k) = inv(x_0..x_k) * (x_0..x_{k-1}),   inv(x_0..x_{k-1}) = inv(x_0..x_k) * x_k
// 3 multiplications per element + one inversion per lane.  Zero elements are skipped and come out as zero (the
// reference's unwrap_or(zero) at scalar.rs:122).  Where the inverses go depends on the call's size (EmitOut::inv_in_place):
// a call of a few thousand elements runs the pre-pass BESIDE the emitter, and each inverse goes straight to its final
// slot in the call's variable table (GD::inv_slot; NULL when the item has no such variable -- an is_non_zero item that
// stopped at its error), which the emit kernel leaves alone: the two kernels write disjoint bytes, so they run
// concurrently on two streams and the call joins them at the end.  A big call runs it FIRST and leaves the inverses
// in the scratch array, dense; the emitter reads them with its items' inputs and writes them as it writes every other
// variable (no 32-byte holes in its lines, no scattered 32-byte stores here).
//
// What the kernel is built around is MEMORY LATENCY, not arithmetic: it runs beside a writer that saturates HBM, so a
// load takes several microseconds, and a lane's chain is strictly sequential.  Round 1 loaded one element per step
// (2 x per_lane dependent round trips per lane, 64 at the C3 size).  Here a lane fetches a GROUP of GRP elements with
// independent loads before it touches the first (one round trip per group), the running products AND the elements
// go to an engine-owned scratch array in the forward pass (stores do not stall), and the unwind reads both back a
// group at a time -- the elements are never recomputed from the gadget's inputs (inv_combine costs up to two
// multiplications for the range gadgets).  Elements are numbered e-major (all items' element 0, then all items'
// element 1): a wave's lanes then read consecutive items of ONE input array -- coalesced, no divergence.
#pragma once

#include "emit.hpp"

namespace pg {

#ifndef PG_INV_GRP
// elements fetched per lane per round trip (registers: ~40 x GRP with the software pipeline).  4 measured best for the
// fused mix in the pre-pass-first order (0.627 ms against 0.650 for 8 and 0.633 for 2 with two waves per SIMD;
// tools/ab_emit.py run_c3): 168 registers leave the dependent multiplication chains more room than 253
#define PG_INV_GRP 4
#endif

// element s of a call -> (item, e), e-major
template <int KINV>
__device__ __forceinline__ void inv_locate(uint64_t s, uint64_t batch, uint64_t &item, uint32_t &e) {
    e = 0;
    item = s;
#pragma unroll
    for (int q = 1; q < KINV; q++)
        if (s >= (uint64_t)q * batch) { e = q; item = s - (uint64_t)q * batch; }
}

// scratch: per element 64 B = {x, running product after x}, as four planes of 16-byte halves
//
// GD's view of its inverses (all static, all __device__):
//   int  kInv                                     inverses per item
//   void inv_operands(A, O, item, e, p, q, aux)   LOADS ONLY, no branches: what element (item, e) is computed from
//   Fr   inv_combine(A, e, p, q, aux)             arithmetic only: the element (0 = "no inverse wanted": stays 0)
//   uint4 *inv_slot(A, O, item, e)                where the inverse goes (NULL: nowhere)
// The split is what lets a lane have a whole group's loads in flight at once: every load of the group is issued
// unconditionally (out-of-range lanes re-read the call's last element and are masked later) before the first
// dependent instruction.
template <class GD, int GRP>
__global__ __launch_bounds__(kThreads) void batch_invert_kernel(const typename GD::Args A, const EmitOut O, uint64_t n_elems,
                                                               uint32_t groups, uint4 *scratch) {
    static_assert(GRP % 2 == 0, "the unwind works on half groups");
    constexpr int HB = GRP / 2;  // the unwind's group: x AND its prefix are live, twice (current + prefetched)
    const uint64_t T = (uint64_t)gridDim.x * kThreads;
    const uint64_t gtid = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    const uint64_t batch = O.batch, last = n_elems - 1;
    Fr acc = fr_one();
    bool any = false;

    // Software pipeline: the loads of group g + 1 are in flight while group g is multiplied up, so that (after the first
    // group) memory time hides behind arithmetic instead of adding to it -- with one wave per SIMD nothing else would.
    FrVec p[GRP], q[GRP];
    uint32_t aux[GRP];
    auto fetch_operands = [&](uint32_t g) {
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            uint64_t s = ((uint64_t)g * GRP + j) * T + gtid, item;
            uint32_t e;
            s = s < last ? s : last;
            inv_locate<GD::kInv>(s, batch, item, e);
            GD::inv_operands(A, O, item, e, p[j], q[j], aux[j]);
        }
    };
    // ---- forward: running products ------------------------------------------------------------------
    fetch_operands(0);
    for (uint32_t g = 0; g < groups; g++) {
        Fr x[GRP];
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            uint64_t s = ((uint64_t)g * GRP + j) * T + gtid, item;
            uint32_t e;
            const bool live = s < n_elems;
            s = s < last ? s : last;
            inv_locate<GD::kInv>(s, batch, item, e);
            x[j] = GD::inv_combine(A, e, p[j].f, q[j].f, aux[j]);
            if (!live) x[j] = fr_zero();
        }
        if (g + 1 < groups) fetch_operands(g + 1);
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            const uint64_t s = ((uint64_t)g * GRP + j) * T + gtid;
            if (!fr_is_zero(x[j])) {
                acc = any ? fr_mul(acc, x[j]) : x[j];
                any = true;
            }
            if (s < n_elems) {
                FrVec xv, pv;
                xv.f = x[j];
                pv.f = acc;
                uint4 *dst = scratch + s;  // four planes of n_elems x 16 B: a wave's store is one contiguous KiB
                dst[0] = xv.v[0];
                dst[n_elems] = xv.v[1];
                dst[2 * n_elems] = pv.v[0];
                dst[3 * n_elems] = pv.v[1];
            }
        }
    }

    // ---- unwind, in half groups; the first one is fetched BEFORE the inversion, which hides it completely ------------
    FrVec x[HB], prev[HB], xn[HB], prevn[HB];
    auto fetch_back = [&](int hg, FrVec(&xs)[HB], FrVec(&ps)[HB]) {
#pragma unroll
        for (int j = 0; j < HB; j++) {
            const uint64_t k = (uint64_t)hg * HB + j;
            uint64_t s = k * T + gtid;
            // running product before x: the previous element of this lane, T elements back (k = 0, or a lane past the
            // end of the call: anything readable -- its own slot -- the value is not used)
            const bool has_prev = k > 0 && s < n_elems;
            s = s < last ? s : last;
            const uint4 *src = scratch + s;
            const uint4 *psrc = has_prev ? src - T : src;
            xs[j].v[0] = src[0];
            xs[j].v[1] = src[n_elems];
            ps[j].v[0] = psrc[2 * n_elems];
            ps[j].v[1] = psrc[3 * n_elems];
        }
    };
    const int half_groups = (int)groups * 2;
    fetch_back(half_groups - 1, x, prev);
    Fr accinv = fr_one();
    if (any) accinv = fr_invert_or_zero(acc);
    for (int hg = half_groups - 1; hg >= 0; hg--) {
        if (hg > 0) fetch_back(hg - 1, xn, prevn);
#pragma unroll
        for (int j = HB - 1; j >= 0; j--) {
            const uint64_t k = (uint64_t)hg * HB + j, s = k * T + gtid;
            if (s >= n_elems) continue;
            FrVec o;
            o.f = fr_zero();
            if (!fr_is_zero(x[j].f)) {
                o.f = k > 0 ? fr_mul(accinv, prev[j].f) : accinv;  // a prefix that is still mont(1) multiplies as the identity
                accinv = fr_mul(accinv, x[j].f);
            }
            uint64_t item;
            uint32_t e;
            inv_locate<GD::kInv>(s, batch, item, e);
            if (!O.inv_in_place) {
                // over x's planes (this lane's own element, already in registers): a wave's store is one contiguous KiB, and
                // the emitter writes the variable with the rest of its line
                uint4 *dst = scratch + s;
                dst[0] = o.v[0];
                dst[n_elems] = o.v[1];
                continue;
            }
            uint4 *slot = GD::inv_slot(A, O, item, e);
            if (slot) {
                slot[0] = o.v[0];
                slot[1] = o.v[1];
            }
        }
#pragma unroll
        for (int j = 0; j < HB; j++) {
            x[j] = xn[j];
            prev[j] = prevn[j];
        }
    }
}

}  // namespace pg
