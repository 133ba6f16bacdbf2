// tools/probes/single_table_fill.hip -- ONE array of 34.7 GB (the witness refresh's table) written by different store shapes,
// on several tables alive together (where a table lies decides 5.5 against 6.6 ms for the refresh: NOTES_r05 section 6).
// Which shapes are indifferent to the placement?  Prints GB/s per table and shape (median of 3 after one untimed launch).
//   hipcc --offload-arch=gfx950 -O3 -o single_table_fill single_table_fill.hip && ./single_table_fill [tables]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

// lanes 16 B apart, K stores per lane, each a whole 4-KiB (256-thread) row of the block apart
template <int K>
__global__ void strided(uint4 *dst, size_t n16) {
    const uint4 v = make_uint4(1, 2, 3, 4);
    const size_t base = (size_t)blockIdx.x * (K * blockDim.x) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++)
        if (base + (size_t)k * blockDim.x < n16) dst[base + (size_t)k * blockDim.x] = v;
}

// every lane writes K x 16 contiguous bytes (an elementwise kernel's vector store of K/2 64-bit pairs)
template <int K>
__global__ void contiguous(uint4 *dst, size_t n16) {
    const uint4 v = make_uint4(1, 2, 3, 4);
    const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * K;
#pragma unroll
    for (int k = 0; k < K; k++)
        if (base + k < n16) dst[base + k] = v;
}

// long-lived workgroups: `piece16` consecutive 16-byte units per trip, grid-stride over the pieces
__global__ void pieces(uint4 *dst, size_t n16, size_t piece16) {
    const uint4 v = make_uint4(1, 2, 3, 4);
    for (size_t p = blockIdx.x; p * piece16 < n16; p += gridDim.x) {
        const size_t end = std::min(n16, (p + 1) * piece16);
        for (size_t i = p * piece16 + threadIdx.x; i < end; i += blockDim.x) dst[i] = v;
    }
}

// one store per lane, 4 KiB per workgroup, with the block -> chunk map bent: MODE 0 chunk = block (+ shift chunks), MODE 1 every
// group of 8 consecutive blocks (one per XCD under round-robin dispatch) writes 8 chunks that lie an eighth of the table apart
// (an XCD then writes ONE contiguous eighth), MODE 2 the 8 blocks of a group write consecutive chunks in REVERSED order
template <int MODE>
__global__ void bent(uint4 *dst, size_t n16, size_t shift) {
    const uint4 v = make_uint4(1, 2, 3, 4);
    const size_t chunks = n16 / 256, b = blockIdx.x;
    size_t c = b;
    if (MODE == 0) c = b + shift;
    if (MODE == 1) c = (b % 8) * (chunks / 8) + b / 8;
    if (MODE == 2) c = (b / 8) * 8 + (7 - b % 8);
    if (c < chunks) dst[c * 256 + threadIdx.x] = v;
}

// long-lived workgroups, XCD-affine: a workgroup on XCD x (HW_REG_XCC_ID) only writes 4-KiB chunks c with c % 8 == (x + shift) % 8;
// the j-th workgroup to arrive on its XCD takes chunks 8 * k + x for k = j, j + per_xcd, ...
__global__ void xcd_affine(uint4 *dst, size_t n16, unsigned *arrivals, unsigned per_xcd, unsigned shift, unsigned run) {
    const uint4 v = make_uint4(1, 2, 3, 4);
    __shared__ unsigned s_j;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    if (threadIdx.x == 0) s_j = atomicAdd(arrivals + xcc * 32, 1u);
    __syncthreads();
    const size_t chunks = n16 / 256;
    // `run` consecutive own chunks per trip (run x 4 KiB written, 8 x run x 4 KiB of address range covered)
    for (size_t k = (size_t)s_j * run; k * 8 < chunks; k += (size_t)per_xcd * run)
        for (unsigned u = 0; u < run; u++) {
            const size_t c = (k + u) * 8 + ((xcc + shift) & 7);
            if (c < chunks) dst[c * 256 + threadIdx.x] = v;
        }
}

// short-lived workgroups of blockDim.x threads that write `piece16` consecutive units each; a dynamic LDS allocation bounds how many
// are resident per CU -- the chip-wide window of addresses under way is CUs x residency x piece
__global__ void windowed(uint4 *dst, size_t n16, size_t piece16) {
    extern __shared__ uint4 pad[];
    const uint4 v = make_uint4(1, 2, 3, 4);
    const size_t p = blockIdx.x, end = std::min(n16, (p + 1) * piece16);
    for (size_t i = p * piece16 + threadIdx.x; i < end; i += blockDim.x) dst[i] = v;
    if (piece16 == 1) pad[threadIdx.x] = v;
}

// long-lived workgroups that never have more than INFLIGHT + 1 stores per wave under way: a piece order that is either fixed
// (grid-stride) or taken from a ticket counter (the pieces then start in address order, whichever workgroup is free)
template <int INFLIGHT, bool TICKET>
__global__ void throttled(uint4 *dst, size_t n16, size_t piece16, unsigned long long *ticket) {
    const uint4 v = make_uint4(1, 2, 3, 4);
    __shared__ unsigned long long s_p;
    size_t p = blockIdx.x;
    for (;;) {
        if (TICKET) {
            __syncthreads();
            if (threadIdx.x == 0) s_p = atomicAdd(ticket, 1ull);
            __syncthreads();
            p = s_p;
        }
        if (p * piece16 >= n16) break;
        const size_t end = std::min(n16, (p + 1) * piece16);
        for (size_t i = p * piece16 + threadIdx.x; i < end; i += blockDim.x) {
            dst[i] = v;
            if (INFLIGHT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (INFLIGHT == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
        if (!TICKET) p += gridDim.x;
    }
}

int main(int argc, char **argv) {
    const int tables = argc > 1 ? atoi(argv[1]) : 6;
    const size_t bytes = (size_t)1034 * 32 << 20;  // 2^20 items x 1034 variables x 32 B
    const size_t n16 = bytes / 16;
    std::vector<uint4 *> tab(tables);
    for (auto &t : tab)
        if (hipMalloc(&t, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    auto timed = [&](auto launch) {
        std::vector<float> ms;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0, 0);
            launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1);
            if (rep) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        return bytes / (ms[1] * 1e-3) / 1e9;
    };
    auto grid_for = [&](size_t per_block16) { return (unsigned)((n16 + per_block16 - 1) / per_block16); };
    printf("%-52s", "shape \\ table");
    for (int t = 0; t < tables; t++) printf("  %6d", t);
    printf("\n");
    struct Row { const char *name; double gbps[16]; };
    auto row = [&](const char *name, auto launch_on) {
        printf("%-52s", name);
        for (int t = 0; t < tables; t++) { printf("  %6.0f", timed([&] { launch_on(tab[t]); })); fflush(stdout); }
        printf("\n");
    };
    row("strided x4, 256 threads (16 KiB / workgroup)", [&](uint4 *d) { hipLaunchKernelGGL(strided<4>, dim3(grid_for(4 * 256)), dim3(256), 0, 0, d, n16); });
    row("strided x1, 256 threads (4 KiB)", [&](uint4 *d) { hipLaunchKernelGGL(strided<1>, dim3(grid_for(256)), dim3(256), 0, 0, d, n16); });
    row("strided x2, 256 threads (8 KiB)", [&](uint4 *d) { hipLaunchKernelGGL(strided<2>, dim3(grid_for(2 * 256)), dim3(256), 0, 0, d, n16); });
    row("strided x4, 1024 threads (64 KiB)", [&](uint4 *d) { hipLaunchKernelGGL(strided<4>, dim3(grid_for(4 * 1024)), dim3(1024), 0, 0, d, n16); });
    row("strided x16, 256 threads (64 KiB)", [&](uint4 *d) { hipLaunchKernelGGL(strided<16>, dim3(grid_for(16 * 256)), dim3(256), 0, 0, d, n16); });
    row("contiguous 32 B / lane, 256 threads (8 KiB)", [&](uint4 *d) { hipLaunchKernelGGL(contiguous<2>, dim3(grid_for(2 * 256)), dim3(256), 0, 0, d, n16); });
    row("contiguous 64 B / lane, 256 threads (16 KiB)", [&](uint4 *d) { hipLaunchKernelGGL(contiguous<4>, dim3(grid_for(4 * 256)), dim3(256), 0, 0, d, n16); });
    row("pieces of 1 MiB, 64 workgroups / CU", [&](uint4 *d) { hipLaunchKernelGGL(pieces, dim3(cus * 64), dim3(256), 0, 0, d, n16, (size_t)65536); });
    row("pieces of 128 KiB, 64 workgroups / CU", [&](uint4 *d) { hipLaunchKernelGGL(pieces, dim3(cus * 64), dim3(256), 0, 0, d, n16, (size_t)8192); });
    row("pieces of 32 KiB, 64 workgroups / CU", [&](uint4 *d) { hipLaunchKernelGGL(pieces, dim3(cus * 64), dim3(256), 0, 0, d, n16, (size_t)2048); });
    row("pieces of 1 MiB, 4 workgroups / CU", [&](uint4 *d) { hipLaunchKernelGGL(pieces, dim3(cus * 4), dim3(256), 0, 0, d, n16, (size_t)65536); });
    row("pieces of 32 KiB, 4 workgroups / CU", [&](uint4 *d) { hipLaunchKernelGGL(pieces, dim3(cus * 4), dim3(256), 0, 0, d, n16, (size_t)2048); });
    row("strided x1, 64 threads (1 KiB)", [&](uint4 *d) { hipLaunchKernelGGL(strided<1>, dim3(grid_for(64)), dim3(64), 0, 0, d, n16); });
    row("strided x1, 512 threads (8 KiB)", [&](uint4 *d) { hipLaunchKernelGGL(strided<1>, dim3(grid_for(512)), dim3(512), 0, 0, d, n16); });
    row("strided x1, 1024 threads (16 KiB)", [&](uint4 *d) { hipLaunchKernelGGL(strided<1>, dim3(grid_for(1024)), dim3(1024), 0, 0, d, n16); });
    unsigned long long *ticket;
    hipMalloc(&ticket, 8);
    const size_t K32 = 2048, M1 = 65536;
#define THROTTLED(NAME, INF, TICK, WGPC, PIECE) \
    row(NAME, [&](uint4 *d) { hipMemsetAsync(ticket, 0, 8, 0); hipLaunchKernelGGL((throttled<INF, TICK>), dim3(cus * WGPC), dim3(256), 0, 0, d, n16, PIECE, ticket); });
    THROTTLED("fixed order, 1 store under way, 8 wg/CU, 32 KiB", 0, false, 8, K32)
    THROTTLED("fixed order, 2 stores under way, 8 wg/CU, 32 KiB", 1, false, 8, K32)
    THROTTLED("fixed order, 4 stores under way, 8 wg/CU, 32 KiB", 3, false, 8, K32)
    THROTTLED("fixed order, 1 store under way, 8 wg/CU, 1 MiB", 0, false, 8, M1)
    THROTTLED("fixed order, 1 store under way, 4 wg/CU, 1 MiB", 0, false, 4, M1)
    THROTTLED("fixed order, 2 stores under way, 4 wg/CU, 1 MiB", 1, false, 4, M1)
    THROTTLED("ticket order, unthrottled, 8 wg/CU, 4 KiB", 9, true, 8, (size_t)256)
    THROTTLED("ticket order, unthrottled, 8 wg/CU, 32 KiB", 9, true, 8, K32)
    THROTTLED("ticket order, 1 store under way, 8 wg/CU, 32 KiB", 0, true, 8, K32)
    THROTTLED("ticket order, unthrottled, 4 wg/CU, 1 MiB", 9, true, 4, M1)
    for (size_t sh = 0; sh < 8; sh++) {
        static char nm[8][64];
        snprintf(nm[sh], 64, "x1 256 threads, table shifted by %zu chunks", sh);
        row(nm[sh], [&](uint4 *d) { hipLaunchKernelGGL(bent<0>, dim3(grid_for(256)), dim3(256), 0, 0, d, n16 - 8 * 256, sh); });
    }
    row("x1 256 threads, an XCD writes one eighth", [&](uint4 *d) { hipLaunchKernelGGL(bent<1>, dim3(grid_for(256)), dim3(256), 0, 0, d, n16, (size_t)0); });
    row("x1 256 threads, groups of 8 reversed", [&](uint4 *d) { hipLaunchKernelGGL(bent<2>, dim3(grid_for(256)), dim3(256), 0, 0, d, n16, (size_t)0); });
    unsigned *arrivals;
    hipMalloc(&arrivals, 8 * 32 * 4);
    for (unsigned run : {1u, 8u, 64u})
        for (unsigned sh = 0; sh < 8; sh += (run == 8u ? 1 : 4)) {
            static char nm2[3][8][80];
            const int ri = run == 1 ? 0 : run == 8 ? 1 : 2;
            snprintf(nm2[ri][sh], 80, "long-lived, XCD-affine, run %u, shift %u, 8 wg/CU", run, sh);
            row(nm2[ri][sh], [&](uint4 *d) {
                hipMemsetAsync(arrivals, 0, 8 * 32 * 4, 0);
                hipLaunchKernelGGL(xcd_affine, dim3(cus * 8), dim3(256), 0, 0, d, n16, arrivals, (unsigned)(cus * 8 / 8), sh, run);
            });
        }
    {
        struct W { int threads, per_cu; size_t piece_kib; };
        const W ws[] = {{256, 1, 32}, {256, 2, 32}, {256, 4, 32}, {256, 8, 32}, {1024, 1, 32}, {1024, 2, 32}, {1024, 1, 128}, {512, 1, 32}, {512, 2, 32},
                        {256, 1, 8}, {256, 2, 8}, {256, 4, 8}, {256, 2, 1024}, {256, 4, 1024}};
        static char nm3[16][96];
        int k = 0;
        for (const W &w : ws) {
            const size_t lds = w.per_cu == 1 ? 100 * 1024 : 160 * 1024 / w.per_cu - 1024;
            hipFuncSetAttribute(reinterpret_cast<const void *>(windowed), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            snprintf(nm3[k], 96, "short-lived, %zu KiB each, %d threads, %d per CU", w.piece_kib, w.threads, w.per_cu);
            const size_t piece16 = w.piece_kib * 64;
            row(nm3[k], [&](uint4 *d) { hipLaunchKernelGGL(windowed, dim3(grid_for(piece16)), dim3(w.threads), lds, 0, d, n16, piece16); });
            k++;
        }
    }
    row("hipMemsetD32Async", [&](uint4 *d) { hipMemsetD32Async((hipDeviceptr_t)d, 0x5a5a5a5a, bytes / 4, 0); });
    if (hipGetLastError() != hipSuccess) { printf("a launch failed\n"); return 1; }
    return 0;
}
