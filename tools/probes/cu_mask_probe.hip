// tools/probes/cu_mask_probe.hip -- does a stream created with hipExtStreamCreateWithCUMask keep its workgroups on the
// masked compute units, and how do mask bits map to (XCC, SE, CU)?   hipcc --offload-arch=gfx950 -o cu_mask_probe cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

__global__ void where(uint32_t *out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup alive for a while so that the grid spreads over every CU the stream may use
    uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < (uint64_t)spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device CUs: %d\n", cus);
    const int nblk = 4096;
    uint32_t *d;
    CK(hipMalloc(&d, nblk * 8));
    std::vector<uint32_t> h(nblk * 2);
    for (int test = 0; test < 4; test++) {
        std::vector<uint32_t> mask((cus + 31) / 32, 0);
        const char *what;
        if (test == 0) { what = "all"; for (int i = 0; i < cus; i++) mask[i / 32] |= 1u << (i % 32); }
        else if (test == 1) { what = "first half"; for (int i = 0; i < cus / 2; i++) mask[i / 32] |= 1u << (i % 32); }
        else if (test == 2) { what = "second half"; for (int i = cus / 2; i < cus; i++) mask[i / 32] |= 1u << (i % 32); }
        else { what = "bits 0..15"; for (int i = 0; i < 16; i++) mask[i / 32] |= 1u << (i % 32); }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask -> %s\n", what, hipGetErrorString(e)); continue; }
        CK(hipMemsetAsync(d, 0xff, nblk * 8, s));
        hipLaunchKernelGGL(where, dim3(nblk), dim3(256), 0, s, d, 2000);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, nblk * 8, hipMemcpyDeviceToHost));
        std::map<uint32_t, std::set<uint32_t>> per_xcc;
        for (int b = 0; b < nblk; b++) {
            const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            per_xcc[xcc].insert(se << 8 | sh << 4 | cu);
        }
        size_t total = 0;
        printf("%-12s:", what);
        for (auto &kv : per_xcc) { printf(" xcc%u=%zu", kv.first, kv.second.size()); total += kv.second.size(); }
        printf("  distinct CUs %zu\n", total);
        CK(hipStreamDestroy(s));
    }
    return 0;
}
