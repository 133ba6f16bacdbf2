// tools/probes/region_waves.hip -- would a witness refresh made of ONE WAVE PER 8-KiB REGION be indifferent to where its table lies?
// A 34.7-GB table written by single-wave workgroups in address order: each reads a 64-byte record (scalar loads, one per item of four
// regions), spins ALU dependent multiply-adds (the region's arithmetic), then issues eight 1-KiB stores.  Six tables alive; GB/s per table.
//   hipcc --offload-arch=gfx950 -O3 -o region_waves region_waves.hip && ./region_waves
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int THREADS, int STORES>
__global__ __launch_bounds__(THREADS) void region(uint4 *dst, size_t n16, const uint4 *rec, int alu) {
    extern __shared__ uint4 pad[];
    const size_t b = blockIdx.x;
    const uint4 r = rec[((b / 4) & 0xfffff) * 4];  // (uniform: scalar loads; 2^20 records of 64 bytes)
    unsigned long long x = r.x | 1, c = r.y + threadIdx.x;
    for (int i = 0; i < alu; i++) x = x * x + c;  // dependent 64-bit multiply-adds: ~6 instructions each
    const uint4 v = make_uint4((uint32_t)x, (uint32_t)(x >> 32), r.z, r.w);
    const size_t base = b * (size_t)(THREADS * STORES) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < STORES; k++)
        if (base + (size_t)k * THREADS < n16) dst[base + (size_t)k * THREADS] = v;
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
        return bytes / (ms[1] * 1e-3) / 1e9;
    };
    auto row = [&](const char *name, auto launch_on) {
        printf("%-72s", name);
        for (int t = 0; t < tables; t++) { printf("  %6.0f", timed([&] { launch_on(tab[t]); })); fflush(stdout); }
        printf("\n");
    };
    static char nm[64][96];
    int k = 0;
    for (int alu : {0, 40, 80, 160}) {
        for (int per_cu : {0, 16, 8, 4}) {  // residency bound through dynamic LDS (0: none)
            const size_t lds = per_cu ? 160 * 1024 / per_cu - 1024 : 0;
            hipFuncSetAttribute(reinterpret_cast<const void *>(region<64, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds ? lds : 1024));
            snprintf(nm[k], 96, "1 wave x 8 KiB, %3d multiply-adds, %s", alu, per_cu == 0 ? "full residency" : per_cu == 16 ? "16 per CU" : per_cu == 8 ? "8 per CU" : "4 per CU");
            const unsigned grid = (unsigned)((n16 + 511) / 512);
            row(nm[k], [&](uint4 *d) { hipLaunchKernelGGL((region<64, 8>), dim3(grid), dim3(64), lds, 0, d, n16, rec, alu); });
            k++;
        }
    }
    for (int alu : {0, 80}) {
        for (int per_cu : {0, 4, 2}) {
            const size_t lds = per_cu ? 160 * 1024 / per_cu - 1024 : 0;
            hipFuncSetAttribute(reinterpret_cast<const void *>(region<256, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds ? lds : 1024));
            snprintf(nm[k], 96, "4 waves x 32 KiB (an item), %3d multiply-adds, %s", alu, per_cu == 0 ? "full residency" : per_cu == 4 ? "4 per CU" : "2 per CU");
            const unsigned grid = (unsigned)((n16 + 2047) / 2048);
            row(nm[k], [&](uint4 *d) { hipLaunchKernelGGL((region<256, 8>), dim3(grid), dim3(256), lds, 0, d, n16, rec, alu); });
            k++;
        }
    }
    if (hipGetLastError() != hipSuccess) { printf("a launch failed\n"); return 1; }
    return 0;
}
