// tools/probes/store_occupancy.hip -- how many waves per CU does a pure store stream need on MI355X?
// A fill of 5 arrays in lock step (the rows launch's shape), by workgroups of 256 / 512 threads whose residency is bounded by a
// dynamic LDS allocation: 1, 2, 4, 8 workgroups per CU = 4 ... 32 waves per CU.  Each lane issues UNROLL 16-byte stores per loop
// trip with nothing between them (no address dependency, no wait).  Prints GB/s per configuration.
//   hipcc --offload-arch=gfx950 -O3 -o store_occupancy store_occupancy.hip && ./store_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int UNROLL>
__global__ void fill5(uint4 *base, size_t stride16, size_t n16, uint32_t pieces_per_block) {
    extern __shared__ uint4 pad[];
    const uint4 v = make_uint4(1, 2, 3, 4);
    const size_t piece = (size_t)blockDim.x * UNROLL;
    for (size_t p = (size_t)blockIdx.x; p * piece < n16; p += gridDim.x) {
        const size_t at = p * piece + threadIdx.x;
#pragma unroll
        for (int c = 0; c < 5; c++) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const size_t i = at + (size_t)u * blockDim.x;
                if (i < n16) base[c * stride16 + i] = v;
            }
        }
    }
    if (pieces_per_block == 0xffffffffu) pad[threadIdx.x] = v;  // (keeps the allocation)
}

int main() {
    const size_t per_col = (size_t)3 << 30;  // 3 GiB per column, 15 GiB in all
    const size_t n16 = per_col / 16;
    uint4 *buf;
    if (hipMalloc(&buf, 5 * per_col) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("CUs %d\n", cus);
    struct Cfg { int threads, wg_per_cu; };
    const Cfg cfgs[] = {{256, 1}, {256, 2}, {512, 1}, {256, 4}, {512, 2}, {1024, 1}, {256, 8}, {512, 4}, {1024, 2}};
    for (const Cfg &c : cfgs) {
        // residency through LDS: 160 KiB per CU / wg_per_cu, a little less
        const size_t lds = c.wg_per_cu == 1 ? 100 * 1024 : (160 * 1024 / c.wg_per_cu) - 1024;
        for (int persistent = 0; persistent < 2; persistent++) {
            const size_t piece = (size_t)c.threads * 4;
            const size_t pieces = (n16 + piece - 1) / piece;
            const unsigned grid = persistent ? (unsigned)(cus * c.wg_per_cu) : (unsigned)(pieces < 0x7fffffff ? pieces : 0x7fffffff);
            hipFuncSetAttribute(reinterpret_cast<const void *>(fill5<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(fill5<4>, dim3(grid), dim3(c.threads), lds, 0, buf, n16, n16, 0u);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
            printf("threads %4d  workgroups/CU %d  = %2d waves/CU  %s  %7.1f GB/s\n", c.threads, c.wg_per_cu, c.threads / 64 * c.wg_per_cu,
                   persistent ? "persistent grid " : "one piece per WG", 5.0 * per_col / best / 1e6);
        }
    }
    hipFree(buf);
    return 0;
}
