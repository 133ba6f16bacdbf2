// tools/probes/region_blocks.hip -- round 6: would a witness refresh made of ONE 256-THREAD WORKGROUP PER 8-KiB REGION, two 16-byte stores per
// lane, dispatched in address order, be indifferent to where its table lies -- WITH the region's arithmetic in front of the stores?
// (round 5: bare short-lived workgroups of 8 KiB, two resident per CU, write 7.1-7.2 TB/s on every table; one wave x eight stores does not.)
// A 34.7-GB table; every workgroup reads its item's 64-byte record (uniform: scalar loads), its first wave runs `alu` dependent 64-bit
// multiply-adds (the accumulators' Montgomery multiplication + three additions, four accumulators per lane, are ~30 of them; all four
// waves: the one-multiplication-per-lane form), the result goes through LDS to all four waves, which store 2 x 16 bytes per lane.
// Six tables alive; GB/s per table.      hipcc --offload-arch=gfx950 -O3 -o region_blocks region_blocks.hip && ./region_blocks
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <bool ALL_WAVES, bool STORE>
__global__ __launch_bounds__(256) void region(uint4 *dst, size_t n16, const uint4 *rec, int alu) {
    extern __shared__ uint4 pad[];
    __shared__ uint4 stage[64];
    const size_t b = blockIdx.x;
    const uint4 r = rec[((b / 4) & 0xfffff) * 4];  // (uniform: scalar loads; 2^20 records of 64 bytes, four regions per item)
    unsigned long long x = r.x | 1, c = r.y + (threadIdx.x & 63);
    if (ALL_WAVES || threadIdx.x < 64)
        for (int i = 0; i < alu; i++) x = x * x + c;  // dependent 64-bit multiply-adds
    if (threadIdx.x < 64) stage[threadIdx.x] = make_uint4((uint32_t)x, (uint32_t)(x >> 32), r.z, r.w);
    __syncthreads();
    uint4 v = stage[threadIdx.x & 63];
    if (ALL_WAVES) v.z ^= (uint32_t)x;
    const size_t base = b * 512 + threadIdx.x;
    if (STORE) {
        if (base < n16) dst[base] = v;
        if (base + 256 < n16) dst[base + 256] = v;
    } else if (v.x == 0x12345 && v.y == 0x6789) dst[base] = v;  // (timing of the arithmetic alone: never taken)
    if (alu < 0) pad[threadIdx.x] = v;
}

int main() {
    const int tables = 6;
    const size_t bytes = (size_t)1034 * 32 << 20, n16 = bytes / 16;
    std::vector<uint4 *> tab(tables);
    for (auto &t : tab) if (hipMalloc(&t, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    uint4 *rec;
    hipMalloc(&rec, (size_t)64 << 20);  // 2^20 items x 64 B
    hipMemset(rec, 1, (size_t)64 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](auto launch) {
        std::vector<float> ms;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1);
            if (rep) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        return ms[1];
    };
    const unsigned grid = (unsigned)((n16 + 511) / 512);
    printf("%u workgroups of 256 threads x 8 KiB per table\n", grid);
    for (int all = 0; all < 2; all++)
        for (int alu : {0, 16, 32, 64, 128}) {
            if (all && alu == 0) continue;
            for (int per_cu : {0, 4, 2}) {  // residency bound through dynamic LDS (0: none)
                const size_t lds = per_cu ? 160 * 1024 / per_cu - 2048 : 0;
                const void *fn = all ? reinterpret_cast<const void *>(region<true, true>) : reinterpret_cast<const void *>(region<false, true>);
                hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds ? lds : 1024));
                printf("%-11s %3d multiply-adds, %-14s GB/s:", all ? "all waves" : "first wave", alu, per_cu == 0 ? "full residency" : per_cu == 4 ? "4 per CU" : "2 per CU");
                for (int t = 0; t < tables; t++) {
                    const float ms = timed([&] {
                        if (all) hipLaunchKernelGGL((region<true, true>), dim3(grid), dim3(256), lds, 0, tab[t], n16, rec, alu);
                        else hipLaunchKernelGGL((region<false, true>), dim3(grid), dim3(256), lds, 0, tab[t], n16, rec, alu);
                    });
                    printf("  %6.0f", bytes / (ms * 1e-3) / 1e9);
                    fflush(stdout);
                }
                if (per_cu == 0) {  // what the arithmetic alone takes (no stores): ms per table's worth of workgroups
                    const float ms = timed([&] {
                        if (all) hipLaunchKernelGGL((region<true, false>), dim3(grid), dim3(256), lds, 0, tab[0], n16, rec, alu);
                        else hipLaunchKernelGGL((region<false, false>), dim3(grid), dim3(256), lds, 0, tab[0], n16, rec, alu);
                    });
                    printf("   | arithmetic alone %.2f ms", ms);
                }
                printf("\n");
            }
        }
    if (hipGetLastError() != hipSuccess) { printf("a launch failed\n"); return 1; }
    return 0;
}
