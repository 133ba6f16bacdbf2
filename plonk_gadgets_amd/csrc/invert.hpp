// invert.hpp -- batched field inversion pre-pass (Montgomery's trick), gfx950.
//
// Every gadget on this path that inverts (maybe_equal: (a-b)^-1, scalar.rs:121-122; is_non_zero: value^-1,
// scalar.rs:73) would pay a full inversion (fr_invert_or_zero: 600 division steps) per item.  The pre-pass
// computes all of a call's inverses first: every lane owns up to G elements (strided, so loads coalesce), multiplies
// them up keeping the running products in a scratch array, inverts ONE product, and unwinds:
//     inv(x_k) = inv(x_0..x_k) * (x_0..x_{k-1}),   inv(x_0..x_{k-1}) = inv(x_0..x_k) * x_k
// 3 multiplications per element + one inversion per G.  Zero elements are skipped and come out as zero (the reference's
// unwrap_or(zero) at scalar.rs:122).  Each inverse goes straight to its final slot in the call's variable table
// (GD::inv_slot; NULL when the item has no such variable -- an is_non_zero item that stopped at its error), which the
// emit kernel leaves alone: the two kernels write disjoint bytes, so they run concurrently on two streams and the
// call joins them at the end.  The element values themselves are recomputed from the gadget's inputs by the policy
// (GD::inv_element) in both passes instead of being staged in memory.
#pragma once

#include "emit.hpp"

namespace pg {

template <class GD>
__global__ __launch_bounds__(kThreads) void batch_invert_kernel(const typename GD::Args A, const EmitOut O, uint64_t n_elems,
                                                               uint32_t per_lane, uint4 *prefix) {
    const uint64_t T = (uint64_t)gridDim.x * kThreads;
    const uint64_t gtid = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    Fr acc = fr_one();
    bool any = false;
    for (uint32_t k = 0; k < per_lane; k++) {
        const uint64_t s = (uint64_t)k * T + gtid;
        if (s >= n_elems) break;
        const Fr x = GD::inv_element(A, s / GD::kInv, (uint32_t)(s % GD::kInv));
        if (!fr_is_zero(x)) {
            acc = any ? fr_mul(acc, x) : x;
            any = true;
        }
        FrVec p;
        p.f = acc;
        prefix[2 * s] = p.v[0];
        prefix[2 * s + 1] = p.v[1];
    }
    Fr accinv = fr_one();
    if (any) accinv = fr_invert_or_zero(acc);
    for (int k = (int)per_lane - 1; k >= 0; k--) {
        const uint64_t s = (uint64_t)k * T + gtid;
        if (s >= n_elems) continue;
        const Fr x = GD::inv_element(A, s / GD::kInv, (uint32_t)(s % GD::kInv));
        FrVec o;
        o.f = fr_zero();
        if (!fr_is_zero(x)) {
            // is there an earlier non-zero element in this lane?  prefix[k-1] == running product before x
            Fr prev = fr_one();
            bool have_prev = false;
            if (k > 0) {
                FrVec p;
                const uint64_t sp = (uint64_t)(k - 1) * T + gtid;
                p.v[0] = prefix[2 * sp];
                p.v[1] = prefix[2 * sp + 1];
                prev = p.f;
                // a prefix that is still mont(1) before any non-zero element: multiplying by it is the identity
                have_prev = true;
            }
            o.f = have_prev ? fr_mul(accinv, prev) : accinv;
            accinv = fr_mul(accinv, x);
        }
        uint4 *slot = GD::inv_slot(A, O, s / GD::kInv, (uint32_t)(s % GD::kInv));
        if (slot) {
            slot[0] = o.v[0];
            slot[1] = o.v[1];
        }
    }
}

}  // namespace pg
