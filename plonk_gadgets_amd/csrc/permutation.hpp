// permutation.hpp -- SURVEY section 8f2: the permutation bookkeeping of the composer, on the device.
//
// dusk-plonk records, per gate, the four wire positions of the row under their Variables
// (perm.add_variables_to_map) and at preprocess time turns every Variable's position list -- in recording order:
// gate by gate, left/right/output/fourth inside a gate -- into one cycle of the copy permutation
// (compute_sigma_permutations).  Recording per call is pointless on a GPU; the same sigma is obtained at the end
// from the wire columns alone:
//   1. key[p] = Variable at position p, value[p] = p, with p = 4*gate + wire (ascending p == recording order);
//   2. stable LSD radix sort of (key, value) by key  -- rocprim::radix_sort_pairs (AMD's device primitive library;
//      the only library call on this path);
//   3. neighbours in the sorted order with equal keys are consecutive positions of one Variable; the last position
//      of a Variable wraps to the first (first_of[var]);
//   4. scatter sigma[wire][gate] = wire' * padded_n + gate'.
#pragma once

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "composer.hpp"

namespace pg {

__global__ __launch_bounds__(kThreads) void perm_keys_kernel(const ComposerCols C, uint64_t n, uint64_t zero_var, uint32_t *keys,
                                                            uint64_t *vals) {
    for (uint64_t p = (uint64_t)blockIdx.x * kThreads + threadIdx.x; p < 4 * n; p += (uint64_t)gridDim.x * kThreads) {
        const uint64_t gate = p >> 2;
        const uint32_t wire = (uint32_t)(p & 3);
        keys[p] = (uint32_t)(wire == 3 ? zero_var : C.w[wire][gate]);
        vals[p] = p;
    }
}

__global__ void perm_patch_fourth_kernel(const FourthWire *fw, uint32_t n_fw, uint32_t *keys) {
    if (threadIdx.x < n_fw) keys[4 * fw[threadIdx.x].gate + 3] = (uint32_t)fw[threadIdx.x].w_4;
}

// heads of the runs of equal keys publish the Variable's first position
__global__ __launch_bounds__(kThreads) void perm_heads_kernel(const uint32_t *keys, const uint64_t *vals, uint64_t P,
                                                             uint64_t *first_of) {
    for (uint64_t j = (uint64_t)blockIdx.x * kThreads + threadIdx.x; j < P; j += (uint64_t)gridDim.x * kThreads)
        if (j == 0 || keys[j - 1] != keys[j]) first_of[keys[j]] = vals[j];
}

__device__ __forceinline__ uint64_t perm_encode(uint64_t p, uint64_t padded_n) { return (p & 3) * padded_n + (p >> 2); }

__global__ __launch_bounds__(kThreads) void perm_link_kernel(const uint32_t *keys, const uint64_t *vals, uint64_t P,
                                                            const uint64_t *first_of, uint64_t padded_n, uint64_t *sigma) {
    for (uint64_t j = (uint64_t)blockIdx.x * kThreads + threadIdx.x; j < P; j += (uint64_t)gridDim.x * kThreads) {
        const uint64_t succ = (j + 1 < P && keys[j + 1] == keys[j]) ? vals[j + 1] : first_of[keys[j]];
        sigma[perm_encode(vals[j], padded_n)] = perm_encode(succ, padded_n);
    }
}

// rows >= circuit size keep the identity
__global__ __launch_bounds__(kThreads) void perm_identity_kernel(uint64_t *sigma, uint64_t n, uint64_t padded_n) {
    const uint64_t pad = padded_n - n;
    for (uint64_t t = (uint64_t)blockIdx.x * kThreads + threadIdx.x; t < 4 * pad; t += (uint64_t)gridDim.x * kThreads) {
        const uint64_t wire = t / pad, gate = n + t % pad;
        sigma[wire * padded_n + gate] = wire * padded_n + gate;
    }
}

}  // namespace pg
