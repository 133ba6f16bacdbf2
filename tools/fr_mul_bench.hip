// tools/fr_mul_bench.hip -- micro-benchmark of the Montgomery multiplication variants on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/fr_mul_bench tools/fr_mul_bench.hip && ./tools/fr_mul_bench
// variant 0 = generic 4 x 64-bit code (fr_mul64), variant 1 = the gfx950 form (fr_mul on the device)
// Prints multiplications/s per variant and checks that the variants agree.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../plonk_gadgets_amd/csrc/fr.hpp"

using namespace pg;

template <int VARIANT>
__global__ __launch_bounds__(256) void chain_kernel(const Fr *in, Fr *out, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr a = in[2 * i], b = in[2 * i + 1];
    for (int k = 0; k < iters; k++) {
        if constexpr (VARIANT == 0) { a = fr_mul64(a, b); b = fr_mul64(b, a); }
        else { a = fr_mul(a, b); b = fr_mul(b, a); }
    }
    out[i] = fr_add(a, b);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256 * 8, threads = argc > 2 ? atoi(argv[2]) : 256, iters = 2000;
    const size_t n = (size_t)blocks * threads;
    std::vector<Fr> h(2 * n);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    for (auto &f : h) {
        for (int k = 0; k < 4; k++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; f.l[k] = s; }
        f.l[3] %= 0x73eda753299d7d48ull;
    }
    Fr *d_in, *d_out0, *d_out1;
    CK(hipMalloc(&d_in, 2 * n * sizeof(Fr)));
    CK(hipMalloc(&d_out0, n * sizeof(Fr)));
    CK(hipMalloc(&d_out1, n * sizeof(Fr)));
    CK(hipMemcpy(d_in, h.data(), 2 * n * sizeof(Fr), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; variant++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            if (variant == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(blocks), dim3(threads), 0, 0, d_in, d_out0, iters);
            else hipLaunchKernelGGL(chain_kernel<1>, dim3(blocks), dim3(threads), 0, 0, d_in, d_out1, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("variant %d (%d blocks x %d): %.3f ms, %.3e mont-mul/s\n", variant, blocks, threads, ms, 2.0 * iters * n / (ms * 1e-3));
        }
    }
    std::vector<Fr> o0(n), o1(n);
    CK(hipMemcpy(o0.data(), d_out0, n * sizeof(Fr), hipMemcpyDeviceToHost));
    CK(hipMemcpy(o1.data(), d_out1, n * sizeof(Fr), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 4; k++) bad += o0[i].l[k] != o1[i].l[k];
    // host check of a few lanes against the 64-bit host code
    size_t hbad = 0;
    for (size_t i = 0; i < 64; i++) {
        Fr a = h[2 * i], b = h[2 * i + 1];
        for (int k = 0; k < iters; k++) { a = fr_mul64(a, b); b = fr_mul64(b, a); }
        Fr r = fr_add(a, b);
        for (int k = 0; k < 4; k++) hbad += r.l[k] != o1[i].l[k];
    }
    printf("mismatching limbs between variants: %zu; vs host: %zu\n", bad, hbad);
    return bad || hbad ? 2 : 0;
}
