// tools/probes/hidden_stores.hip -- can a store stream hide behind an arithmetic phase that leaves HBM idle?
// The fused mix's arithmetic launch spends ~50 us in its inversions (one wave per SIMD computes, the other waits, nothing moves);
// the rows launch that follows is a pure store stream.  If the waiting waves wrote part of the rows meanwhile, would the bytes
// reach HBM during the window -- or sit dirty in L2 / the Infinity Cache and be paid for by the HBM-bound phase that follows?
//   A  : 256 workgroups x 8 waves (one per CU), waves 0-3 spin on dependent integer multiply-adds for ~SPIN_US, waves 4-7 idle
//   A' : the same, waves 4-7 write W MB in all (a tight loop of 16-byte stores) while waves 0-3 spin
//   B  : a store stream of S MB at full occupancy (the phase that follows: the backward pass / the rows launch)
// Timed: A;B(S)   A';B(S)   A;B(S+W)   over fresh addresses every round (a 6 GiB ring), median of 15 rounds.
//   hipcc --offload-arch=gfx950 -O3 -o hidden_stores hidden_stores.hip && ./hidden_stores
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512) void phase_a(uint4 *dst, size_t n16_per_block, unsigned spin, unsigned *sink, int writers) {
    extern __shared__ uint4 pad[];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        unsigned long long x = threadIdx.x + 1;
        for (unsigned i = 0; i < spin; i++) x = x * 6364136223846793005ull + 1442695040888963407ull;  // a dependent chain
        if (x == 42) sink[0] = (unsigned)x;
    } else if (writers) {
        const uint4 v = make_uint4(wave, lane, 3, 4);
        uint4 *p = dst + (size_t)blockIdx.x * n16_per_block;
        const size_t per_wave = n16_per_block / 4;
        p += (size_t)(wave - 4) * per_wave;
        for (size_t i = lane; i < per_wave; i += 64 * 8) {
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (i + 64 * u < per_wave) p[i + 64 * u] = v;
        }
    }
    if (spin == 0xffffffffu) pad[threadIdx.x] = make_uint4(0, 0, 0, 0);
}

__global__ __launch_bounds__(256) void phase_b(uint4 *dst, size_t n16) {
    const uint4 v = make_uint4(9, 9, 9, 9);
    const size_t piece = 4096;  // 64 KiB per workgroup piece
    for (size_t base = (size_t)blockIdx.x * piece; base < n16; base += (size_t)gridDim.x * piece)
        for (size_t i = base + threadIdx.x; i < base + piece && i < n16; i += 256) dst[i] = v;
}

int main() {
    const size_t ring = (size_t)6 << 30;
    uint4 *buf; unsigned *sink;
    if (hipMalloc(&buf, ring) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipFuncSetAttribute(reinterpret_cast<const void *>(phase_a), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time_of = [&](auto fn) {
        std::vector<float> ms;
        for (int r = 0; r < 16; r++) {
            hipDeviceSynchronize();
            hipEventRecord(e0, 0); fn(r); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1);
            if (r) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        return ms[ms.size() / 2] * 1e3f;  // us
    };
    // calibrate the spin to ~50 us
    unsigned spin = 20000;
    for (int k = 0; k < 6; k++) {
        const float us = time_of([&](int) { hipLaunchKernelGGL(phase_a, dim3(256), dim3(512), 100 * 1024, 0, buf, (size_t)0, spin, sink, 0); });
        spin = (unsigned)(spin * 50.0f / us);
    }
    const float a_alone = time_of([&](int) { hipLaunchKernelGGL(phase_a, dim3(256), dim3(512), 100 * 1024, 0, buf, (size_t)0, spin, sink, 0); });
    printf("A alone (spin %u): %.1f us\n", spin, a_alone);
    const size_t S = (size_t)770 << 20;
    for (size_t Wmb : {60, 120, 240, 360}) {
        const size_t W = Wmb << 20, per_block16 = W / 256 / 16;
        auto at = [&](int r, size_t need) { return buf + ((size_t)r * ((size_t)1200 << 20) % (ring - need - W)) / 16; };
        const float a_w = time_of([&](int r) { hipLaunchKernelGGL(phase_a, dim3(256), dim3(512), 100 * 1024, 0, at(r, 0), per_block16, spin, sink, 1); });
        const float ab = time_of([&](int r) {
            hipLaunchKernelGGL(phase_a, dim3(256), dim3(512), 100 * 1024, 0, at(r, S), per_block16, spin, sink, 0);
            hipLaunchKernelGGL(phase_b, dim3(256 * 32), dim3(256), 0, 0, at(r, S) + W / 16, S / 16);
        });
        const float awb = time_of([&](int r) {
            hipLaunchKernelGGL(phase_a, dim3(256), dim3(512), 100 * 1024, 0, at(r, S), per_block16, spin, sink, 1);
            hipLaunchKernelGGL(phase_b, dim3(256 * 32), dim3(256), 0, 0, at(r, S) + W / 16, S / 16);
        });
        const float abw = time_of([&](int r) {
            hipLaunchKernelGGL(phase_a, dim3(256), dim3(512), 100 * 1024, 0, at(r, S), per_block16, spin, sink, 0);
            hipLaunchKernelGGL(phase_b, dim3(256 * 32), dim3(256), 0, 0, at(r, S), (S + W) / 16);
        });
        printf("W = %3zu MB:  A' alone %.1f us | A;B(S) %.1f | A';B(S) %.1f | A;B(S+W) %.1f   -> hidden %.0f %% of W's cost\n", Wmb, a_w, ab, awb, abw,
               100.0 * (abw - awb) / (abw - ab));
    }
    return 0;
}
