// tools/probes/identity_fill.hip -- the padding of sigma (perm_identity_kernel): four runs [w * P + n, (w + 1) * P) of a 4 * P-entry array get
// sigma[i] = i.  Long-lived workgroups with 1-MiB pieces (round 4's kernel) against short-lived ones of 4 / 8 KiB in dispatch order
// (tools/probes/single_table_fill.hip: the shapes that are indifferent to where the array lies).  GB/s over the 8.5 GB written.
//   hipcc --offload-arch=gfx950 -O3 -o identity_fill identity_fill.hip && ./identity_fill
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned long long __attribute__((ext_vector_type(2))) u64x2;

__global__ __launch_bounds__(256) void pieces(uint64_t *sigma, uint64_t n, uint64_t P) {
    constexpr uint64_t kPiece = 131072;
    for (uint32_t wire = 0; wire < 4; wire++) {
        uint64_t lo = wire * P + n, hi = (wire + 1) * P;
        lo += lo & 1; hi -= hi & 1;
        for (uint64_t base = lo + (uint64_t)blockIdx.x * kPiece; base < hi; base += (uint64_t)gridDim.x * kPiece) {
            const uint64_t end = base + kPiece < hi ? base + kPiece : hi;
            for (uint64_t i = base + 2 * threadIdx.x; i < end; i += 2 * 256) *reinterpret_cast<u64x2 *>(sigma + i) = u64x2{i, i + 1};
        }
    }
}

// K stores per lane, workgroup b of run w covers K * 256 consecutive 16-byte units; MODE 0: the runs one after the other in block order,
// MODE 1: the four runs interleaved (block b -> run b % 4): four fronts
template <int K, int MODE>
__global__ __launch_bounds__(256) void oneshot(uint64_t *sigma, uint64_t n, uint64_t P, uint32_t blocks_per_run) {
    extern __shared__ uint4 pad[];
    const uint32_t wire = MODE ? blockIdx.x & 3 : blockIdx.x / blocks_per_run, b = MODE ? blockIdx.x >> 2 : blockIdx.x % blocks_per_run;
    uint64_t lo = wire * P + n, hi = (wire + 1) * P;
    lo += lo & 1; hi -= hi & 1;
    const uint64_t base = lo + ((uint64_t)b * K * 256 + threadIdx.x) * 2;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint64_t i = base + (uint64_t)k * 512;
        if (i < hi) *reinterpret_cast<u64x2 *>(sigma + i) = u64x2{i, i + 1};
    }
    if (P == 1) pad[threadIdx.x] = make_uint4(0, 0, 0, 0);
}

int main() {
    const uint64_t n = 270270467, P = 536870912;
    uint64_t *sigma;
    if (hipMalloc(&sigma, 4 * P * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const double bytes = 4.0 * (P - n) * 8;
    auto timed = [&](const char *name, auto launch) {
        std::vector<float> ms;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1);
            if (rep) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-64s %7.3f ms  %6.0f GB/s\n", name, ms[2], bytes / (ms[2] * 1e-3) / 1e9);
    };
    const uint64_t units = (P - n) / 2 + 1;
    for (int round = 0; round < 2; round++) {
        timed("long-lived, 1-MiB pieces, 64 workgroups per CU (round 4)", [&] { hipLaunchKernelGGL(pieces, dim3(prop.multiProcessorCount * 64), dim3(256), 0, 0, sigma, n, P); });
        { const uint32_t bpr = (uint32_t)((units + 255) / 256);
          timed("short-lived, 4 KiB, runs in turn", [&] { hipLaunchKernelGGL((oneshot<1, 0>), dim3(4 * bpr), dim3(256), 0, 0, sigma, n, P, bpr); });
          timed("short-lived, 4 KiB, four fronts", [&] { hipLaunchKernelGGL((oneshot<1, 1>), dim3(4 * bpr), dim3(256), 0, 0, sigma, n, P, bpr); }); }
        { const uint32_t bpr = (uint32_t)((units + 511) / 512);
          const size_t lds = 160 * 1024 / 2 - 1024;
          hipFuncSetAttribute(reinterpret_cast<const void *>(oneshot<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipFuncSetAttribute(reinterpret_cast<const void *>(oneshot<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          timed("short-lived, 8 KiB, runs in turn, full residency", [&] { hipLaunchKernelGGL((oneshot<2, 0>), dim3(4 * bpr), dim3(256), 0, 0, sigma, n, P, bpr); });
          timed("short-lived, 8 KiB, runs in turn, 2 per CU", [&] { hipLaunchKernelGGL((oneshot<2, 0>), dim3(4 * bpr), dim3(256), lds, 0, sigma, n, P, bpr); });
          timed("short-lived, 8 KiB, four fronts, 2 per CU", [&] { hipLaunchKernelGGL((oneshot<2, 1>), dim3(4 * bpr), dim3(256), lds, 0, sigma, n, P, bpr); }); }
    }
    if (hipGetLastError() != hipSuccess) { printf("a launch failed\n"); return 1; }
    return 0;
}
