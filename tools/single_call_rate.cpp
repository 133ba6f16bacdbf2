// tools/single_call_rate.cpp -- the reference-shaped path from C++ (include/plonk_gadgets.hpp): one AllocatedScalar::allocate +
// range_check per iteration on the device composer, no batching.  g++ -std=c++17 -O2 -Iinclude tools/single_call_rate.cpp
//   -Lplonk_gadgets_amd -lplonk_gadgets_hip -Wl,-rpath,$PWD/plonk_gadgets_amd -o tools/single_call_rate_cpp
#include <chrono>
#include <cstdio>

#include "plonk_gadgets.hpp"

using namespace plonk_gadgets;

int main() {
    Engine e(0);
    StandardComposer c(e, 1 << 22, 1 << 22);
    const BlsScalar mn = BlsScalar::from(0), mx = BlsScalar::pow_of_2(254);
    for (int i = 0; i < 50; i++) RangeGadgets::range_check(c, mn, mx, AllocatedScalar::allocate(c, BlsScalar::from(i)));
    pg_composer_sync(c.h);
    const int N = 2000;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) RangeGadgets::range_check(c, mn, mx, AllocatedScalar::allocate(c, BlsScalar::from(1000 + i)));
    pg_composer_sync(c.h);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("C++ single range_check calls: %.1f us per call, %.3g constraints/s\n", dt / N * 1e6, N * 1031.0 / dt);
    return c.check() == -1 ? 0 : 1;
}
