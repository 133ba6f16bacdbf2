// tools/single_call_rate.cpp -- the reference-shaped path from C++ (include/plonk_gadgets.hpp): one AllocatedScalar::allocate +
// range_check per iteration on the device composer, no batching by the caller; with the composer's command queue off (one
// launch per call) and on (calls recorded, flushed as few launches).
//   g++ -std=c++17 -O2 -Iinclude tools/single_call_rate.cpp -Lplonk_gadgets_amd -lplonk_gadgets_hip
//       -Wl,-rpath,$PWD/plonk_gadgets_amd -o tools/single_call_rate_cpp
#include <chrono>
#include <cstdio>

#include "plonk_gadgets.hpp"

using namespace plonk_gadgets;

int main() {
    Engine e(0);
    const BlsScalar mn = BlsScalar::from(0), mx = BlsScalar::pow_of_2(254);
    int bad = 0;
    for (int queued = 0; queued < 2; queued++) {
        StandardComposer c(e, 1 << 23, 1 << 23);
        pg_composer_queue(c.h, queued);
        for (int i = 0; i < 50; i++) RangeGadgets::range_check(c, mn, mx, AllocatedScalar::allocate(c, BlsScalar::from(i)));
        pg_composer_sync(c.h);
        const int N = 4000;
        for (int rep = 0; rep < 2; rep++) {  // the second pass is the steady state: the first one also maps fresh HBM pages
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) RangeGadgets::range_check(c, mn, mx, AllocatedScalar::allocate(c, BlsScalar::from(1000 + i)));
        const double t_calls = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        pg_composer_sync(c.h);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("C++ queue %-3s allocate + range_check: %6.2f us per pair (%.2f us in the calls, the rest in the final flush + sync), "
                    "%.3g constraints/s%s\n", queued ? "on" : "off", dt / N * 1e6, t_calls / N * 1e6, N * 1031.0 / dt,
                    rep ? "" : "  [first pass over fresh memory]");
        }
        auto t0 = std::chrono::steady_clock::now();
        double dt;
        for (int i = 0; i < N; i++) c.boolean_gate(Variable{0});
        pg_composer_sync(c.h);
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("C++ queue %-3s gate calls:             %6.2f us per call\n", queued ? "on" : "off", dt / N * 1e6);
        bad |= c.check() != -1;
    }
    return bad;
}
